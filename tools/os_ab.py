"""Output-stationary weight-gradient GEMM against the tile-loop kernel on the GPU box (MGEN_GEMM_OS=0 switches the
new kernel off, read on every call): agreement up to the summation order, and timing."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "symbolic-music-generation_amd"))
import torch
from mgen import ops
dev = torch.device("cuda:0")


def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


ok = True
for name, M, N, K, acc in [("qkv wgrad", 3072, 512, 65536, True), ("gu wgrad", 4096, 512, 65536, True),
                           ("o wgrad", 512, 1024, 65536, True), ("down wgrad", 512, 2048, 65536, True),
                           ("lm wgrad 8k", 151936, 512, 8192, True), ("lm wgrad 24k", 151936, 512, 24576, True), ("lm wgrad 32k", 151936, 512, 32768, True),
                           ("lm wgrad store", 151936, 512, 8192, False), ("ragged", 1000, 520, 4096, True)]:
    a = (torch.randn(K, M, device=dev) * 0.5).to(torch.bfloat16)
    b = (torch.randn(K, N, device=dev) * 0.5).to(torch.bfloat16)
    outs, ts = [], []
    for flag in ("0", "1"):
        os.environ["MGEN_GEMM_OS"] = flag
        c = torch.full((M, N), 1.0, dtype=torch.float32, device=dev)
        run = lambda: ops.gemm(a, b, c, M, N, K, M, N, N, True, True, ops.OUT_F32_ACC if acc else ops.OUT_F32)
        run(); torch.cuda.synchronize()
        outs.append(c.clone())
        ts.append(timeit(run))
    err = float((outs[0] - outs[1]).abs().max() / outs[0].abs().max())
    fl = 2.0 * M * N * K
    good = err < 1e-5
    ok &= good
    print(f"{name:16s} M={M:6d} N={N:5d} K={K:6d}: old {ts[0]:8.1f} us ({fl/ts[0]/1e6:7.1f} TF/s)  new {ts[1]:8.1f} us "
          f"({fl/ts[1]/1e6:7.1f} TF/s)  rel diff {err:.1e} {'ok' if good else 'BAD'}", flush=True)

# ---- products with 512 output columns: input gradients (B stored [K][N]) and residual projections
print("--- dgrad / residual projections (bitwise against the tile-loop kernel)")
T = 65536
for name, M, N, K, tb, mode in [("qkv dgrad", T, 512, 3072, True, ops.OUT_BF16), ("gu dgrad", T, 512, 4096, True, ops.OUT_BF16),
                                ("lm dgrad", 32768, 512, 151936, True, ops.OUT_BF16),
                                ("o fwd+res", T, 512, 1024, False, ops.OUT_F32_RESID_BF16ROUND),
                                ("down fwd+res", T, 512, 2048, False, ops.OUT_F32_RESID_BF16ROUND)]:
    a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    b = (torch.randn((K, N) if tb else (N, K), device=dev) * 0.1).to(torch.bfloat16)
    res = torch.randn(M, N, device=dev) if mode != ops.OUT_BF16 else None
    outs, ts = [], []
    for flag in ("0", "1"):
        os.environ["MGEN_GEMM_OS"] = flag
        c = torch.zeros(M, N, dtype=torch.bfloat16 if mode == ops.OUT_BF16 else torch.float32, device=dev)
        run = lambda: ops.gemm(a, b, c, M, N, K, a.stride(0), b.stride(0), N, False, tb, mode, res, N if res is not None else 0)
        ts.append(timeit(run))
        outs.append(c)
    same = torch.equal(outs[0], outs[1])
    ok &= same
    fl = 2.0 * M * N * K
    print(f"{name:16s} M={M:6d} N={N:5d} K={K:6d}: old {ts[0]:8.1f} us ({fl/ts[0]/1e6:7.1f} TF/s)  new {ts[1]:8.1f} us "
          f"({fl/ts[1]/1e6:7.1f} TF/s)  equal={same}", flush=True)
print("ALL OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
