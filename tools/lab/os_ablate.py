"""Times one weight-gradient product (3072 x 512 over 65 536 tokens, and the lm_head chunk's 151 936 x 512 over
32 768) on whatever library MGEN_LIB names (ablation builds of csrc/gemm_os.hip: results may be wrong on purpose)."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "symbolic-music-generation_amd"))
from mgen import ops

dev = "cuda:0"
g = torch.Generator().manual_seed(0)
out = []
for M, N, K in [(3072, 512, 65536), (151936, 512, 32768)]:
    a = (torch.randn(K, M, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    b = (torch.randn(K, N, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    c = torch.zeros(M, N, dtype=torch.float32, device=dev)
    n = 20 if M < 10000 else 5
    for _ in range(2):
        ops.gemm(a, b, c, M, N, K, M, N, N, True, True, ops.OUT_F32_ACC)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.gemm(a, b, c, M, N, K, M, N, N, True, True, ops.OUT_F32_ACC)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / n * 1e3
    out.append(f"{M}x{N}x{K}: {t:.1f} us ({2.0 * M * N * K / t / 1e9:.3f} PF)")
    del a, b, c
print(os.path.basename(os.environ.get("MGEN_LIB", "default")), " | ".join(out), flush=True)
