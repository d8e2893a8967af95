"""GEMM micro-benchmark on the GPU box: the model's shapes through mgen_gemm_bf16, random data, HIP-event timing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "symbolic-music-generation_amd"))
import torch
from mgen import ops

dev = torch.device("cuda:0")
T = 65536
shapes = [  # name, M, N, K, ta, tb, mode
    ("qkv fwd", T, 3072, 512, 0, 0, ops.OUT_BF16), ("gu fwd", T, 4096, 512, 0, 0, ops.OUT_BF16),
    ("o fwd+res", T, 512, 1024, 0, 0, ops.OUT_F32_RESID_BF16ROUND), ("down fwd+res", T, 512, 2048, 0, 0, ops.OUT_F32_RESID_BF16ROUND),
    ("lm fwd", 8192, 151936, 512, 0, 0, ops.OUT_BF16),
    ("qkv dgrad", T, 512, 3072, 0, 1, ops.OUT_BF16), ("gu dgrad", T, 512, 4096, 0, 1, ops.OUT_BF16),
    ("o dgrad", T, 1024, 512, 0, 1, ops.OUT_BF16), ("down dgrad", T, 2048, 512, 0, 1, ops.OUT_BF16),
    ("lm dgrad", 8192, 512, 151936, 0, 1, ops.OUT_BF16),
    ("qkv wgrad", 3072, 512, T, 1, 1, ops.OUT_F32_ACC), ("gu wgrad", 4096, 512, T, 1, 1, ops.OUT_F32_ACC),
    ("o wgrad", 512, 1024, T, 1, 1, ops.OUT_F32_ACC), ("down wgrad", 512, 2048, T, 1, 1, ops.OUT_F32_ACC),
    ("lm wgrad", 151936, 512, 8192, 1, 1, ops.OUT_F32_ACC),
    ("square 4096", 4096, 4096, 4096, 0, 0, ops.OUT_BF16), ("square 8192", 8192, 8192, 8192, 0, 0, ops.OUT_BF16),
]
tot_t = 0.0
if len(sys.argv) > 1:   # keep the shapes whose name contains one of the arguments
    shapes = [s for s in shapes if any(a in s[0] for a in sys.argv[1:])]
for name, M, N, K, ta, tb, mode in shapes:
    PAD = int(os.environ.get("BENCH_LD_PAD", "0"))   # extra elements per operand row (leading-dimension experiment)
    a = (torch.randn(((K, M + PAD) if ta else (M, K + PAD)), device=dev) * 0.5).to(torch.bfloat16)
    b = (torch.randn(((K, N + PAD) if tb else (N, K + PAD)), device=dev) * 0.5).to(torch.bfloat16)
    c = torch.zeros(M, N, dtype=torch.bfloat16 if mode == ops.OUT_BF16 else torch.float32, device=dev)
    r = torch.zeros(M, N, dtype=torch.float32, device=dev) if mode == ops.OUT_F32_RESID_BF16ROUND else None
    def run():
        ops.gemm(a, b, c, M, N, K, a.stride(0), b.stride(0), N, bool(ta), bool(tb), mode, r, N if r is not None else 0)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name:14s} M={M:6d} N={N:6d} K={K:6d}  {ms*1e3:9.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s", flush=True)
