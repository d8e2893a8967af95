"""Input-gradient form (A [M][K] rows reduction-contiguous, B [K][N], bf16 out) at M = 65536, N = 512 over K:
fixed cost per launch and time per 1024 reduction indices of the output-stationary kernel, next to the vendor library."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "symbolic-music-generation_amd"))
from mgen import ops  # noqa: E402

dev = "cuda"
M, N = 65536, 512


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for K in (512, 1024, 2048, 3072, 4096, 8192):
    a = (torch.randn(M, K + 64, device=dev) * 0.5).bfloat16()[:, :K]
    b = (torch.randn(K, N, device=dev) * 0.05).bfloat16()
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: ops.gemm(a, b, c, M, N, K, a.stride(0), N, N, False, True, ops.OUT_BF16))
    ac = a.contiguous()
    tv = timeit(lambda: torch.matmul(ac, b, out=c))
    fl = 2.0 * M * N * K
    print(f"K={K:5d}: this library {t:7.1f} us ({fl / t / 1e9:4.2f} PF) | vendor {tv:7.1f} us ({fl / tv / 1e9:4.2f} PF)", flush=True)
