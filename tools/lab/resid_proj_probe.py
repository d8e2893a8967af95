"""Main-loop speed of the candidates for the residual projections (65536 x 512 x K, K = 1024 / 2048), bf16 output:
tile loop on [N][K] weights, output-stationary kernel on the transposed copy [K][N], and the tile loop with the fp32
residual epilogue (the form the step runs)."""
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


PAD = int(os.environ.get("PAD", "0"))   # elements of row padding of the activation operand
for K in (1024, 2048):
    a = (torch.randn(M, K + PAD, device=dev) * 0.5).bfloat16()[:, :K]
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    wt = w.t().contiguous()
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    h = torch.randn(M, N, device=dev)
    o = torch.empty(M, N, device=dev)
    fl = 2.0 * M * N * K
    t1 = timeit(lambda: ops.gemm(a, w, c, M, N, K, a.stride(0), K, N, False, False, ops.OUT_BF16))
    t2 = timeit(lambda: ops.gemm(a, wt, c, M, N, K, a.stride(0), N, N, False, True, ops.OUT_BF16))
    t3 = timeit(lambda: ops.gemm(a, w, o, M, N, K, a.stride(0), K, N, False, False, ops.OUT_F32_RESID_BF16ROUND, h, N))
    x = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    wn = torch.ones(N, device=dev)
    rs = torch.empty(M, device=dev)
    t4 = timeit(lambda: ops.rmsnorm_fwd(o, wn, x, rs, 1e-6))
    print(f"pad {PAD} K={K}: tile loop bf16 {t1:6.1f} us ({fl / t1 / 1e9:4.2f} PF) | output-stationary on W^T bf16 {t2:6.1f} us "
          f"({fl / t2 / 1e9:4.2f} PF) | tile loop + fp32 residual {t3:6.1f} us | rmsnorm forward {t4:5.1f} us", flush=True)
