"""Register-panel GEMM (K = 512) against the tile-loop kernel on the GPU box: bit-for-bit equality and timing.
MGEN_GEMM_RS=0 (read on every call) switches the new kernel off."""
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


def plain(M, N, K=512, check=True):
    a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.5).to(torch.bfloat16)
    c0 = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    c1 = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=dev)
    run = lambda c: ops.gemm(a, b, c, M, N, K, K, K, N, False, False, ops.OUT_BF16)
    os.environ["MGEN_GEMM_RS"] = "0"; t0 = timeit(lambda: run(c0))
    os.environ["MGEN_GEMM_RS"] = "1"; t1 = timeit(lambda: run(c1))
    same = bool(torch.equal(c0, c1)) if check else None
    if check and not same:
        d = (c0.float() - c1.float()).abs()
        bad = (d > 0).nonzero()
        print("   mismatches:", bad.shape[0], "first:", bad[:4].tolist(), "max", float(d.max()))
    fl = 2.0 * M * N * K
    print(f"plain  M={M} N={N}: old {t0:8.1f} us ({fl/t0/1e6:7.1f} TF/s)  new {t1:8.1f} us ({fl/t1/1e6:7.1f} TF/s)  equal={same}", flush=True)
    return same


def swiglu(M, I, K=512):
    x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn(2 * I, K, device=dev) * 0.1).to(torch.bfloat16)
    outs = []
    ts = []
    for flag in ("0", "1"):
        os.environ["MGEN_GEMM_RS"] = flag
        gu = torch.zeros(M, 2 * I, dtype=torch.bfloat16, device=dev)
        act = torch.zeros(M, I, dtype=torch.bfloat16, device=dev)
        ts.append(timeit(lambda: ops.linear_swiglu_fwd(x, w, gu, act)))
        outs.append((gu, act))
    same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    fl = 2.0 * M * 2 * I * K
    print(f"swiglu M={M} I={I}: old {ts[0]:8.1f} us ({fl/ts[0]/1e6:7.1f} TF/s)  new {ts[1]:8.1f} us ({fl/ts[1]/1e6:7.1f} TF/s)  equal={same}", flush=True)
    return same


ok = True
ok &= plain(16384, 1024)
ok &= plain(65536, 3072)
ok &= plain(65536, 1536)
ok &= swiglu(65536, 2048)
ok &= plain(32768, 151936)
ok &= plain(8192, 151936)
print("ALL EQUAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
