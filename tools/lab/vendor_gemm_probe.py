"""Reference point, not part of the product: what the vendor GEMM library behind torch.matmul (hipBLASLt / rocBLAS)
reaches on the training step's product shapes, next to this library's kernels on the same box.
    python tools/lab/vendor_gemm_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "symbolic-music-generation_amd"))
from mgen import ops  # noqa: E402

dev = "cuda"


def timeit(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us


def case(name, M, N, K, kind):
    """kind: 'nt' C[M,N] = A[M,K] B[N,K]^T (forward); 'nn' C[M,N] = A[M,K] B[K,N] (input gradient);
    'tn' C[M,N] = A[K,M]^T B[K,N] fp32 (weight gradient)"""
    g = torch.Generator(device=dev).manual_seed(0)
    if kind == "nt":
        a = torch.randn(M, K, device=dev, generator=g).bfloat16()
        b = torch.randn(N, K, device=dev, generator=g).bfloat16()
        c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        mine = lambda: ops.gemm(a, b, c, M, N, K, K, K, N, False, False, ops.OUT_BF16)
        vend = lambda: torch.matmul(a, b.t(), out=c)
    elif kind == "nn":
        a = torch.randn(M, K, device=dev, generator=g).bfloat16()
        b = torch.randn(K, N, device=dev, generator=g).bfloat16()
        c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        mine = lambda: ops.gemm(a, b, c, M, N, K, K, N, N, False, True, ops.OUT_BF16)
        vend = lambda: torch.matmul(a, b, out=c)
    else:
        a = torch.randn(K, M, device=dev, generator=g).bfloat16()
        b = torch.randn(K, N, device=dev, generator=g).bfloat16()
        c = torch.zeros(M, N, device=dev, dtype=torch.float32)
        cb = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        mine = lambda: ops.gemm(a, b, c, M, N, K, M, N, N, True, True, ops.OUT_F32_ACC)
        vend = lambda: torch.matmul(a.t(), b, out=cb)   # (bf16 output: the library's fastest form)
    t_m, t_v = timeit(mine), timeit(vend)
    fl = 2.0 * M * N * K
    print(f"{name:34s} {M:6d} x {N:6d} x {K:6d}: this library {t_m:8.1f} us {fl / t_m / 1e9:6.2f} PF | vendor {t_v:8.1f} us "
          f"{fl / t_v / 1e9:6.2f} PF", flush=True)


case("QKV forward", 65536, 3072, 512, "nt")
case("gate/up forward", 65536, 4096, 512, "nt")
case("lm_head logits (chunk)", 32768, 151936, 512, "nt")
case("lm_head input gradient", 32768, 512, 151936, "nn")
case("lm_head weight gradient", 151936, 512, 32768, "tn")
case("QKV input gradient", 65536, 512, 3072, "nn")
case("gate/up input gradient", 65536, 512, 4096, "nn")
case("gate/up weight gradient", 4096, 512, 65536, "tn")
case("o_proj forward", 65536, 512, 1024, "nt")
case("down_proj forward", 65536, 512, 2048, "nt")
case("square 8192", 8192, 8192, 8192, "nt")
