"""A/B of two builds of the kernel library on one GPU, attention kernels only: the candidate's outputs are compared
with the baseline's on the same inputs (several shapes, ragged S included), then both are timed interleaved in one
process (HIP events, median over rounds).

    python tools/attn_ab.py BASE.so CAND.so [fwd|bwd|both] [B S H D]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "symbolic-music-generation_amd"))
import torch
from mgen._lib import SIGNATURES

dev = torch.device("cuda:0")
bf = torch.bfloat16


def load(path):
    l = ctypes.CDLL(os.path.abspath(path))
    for name in ("mgen_attn_fwd", "mgen_attn_bwd"):
        fn = getattr(l, name)
        fn.argtypes = SIGNATURES[name]
        fn.restype = ctypes.c_int
    l.mgen_last_error.restype = ctypes.c_char_p
    return l


def make(B, S, H, KVH, D, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    T = B * S
    q = torch.randn(T, H * D, device=dev, generator=g).to(bf)
    k = torch.randn(T, KVH * D, device=dev, generator=g).to(bf)
    v = torch.randn(T, KVH * D, device=dev, generator=g).to(bf)
    do = torch.randn(T, H * D, device=dev, generator=g).to(bf)
    return q, k, v, do


class Run:
    def __init__(self, lib, B, S, H, KVH, D, t):
        self.lib, self.dims = lib, (B, S, H, KVH, D)
        self.q, self.k, self.v, self.do = t
        T = B * S
        self.o = torch.zeros(T, H * D, dtype=bf, device=dev)
        self.lse = torch.zeros(B, H, S, device=dev)
        self.delta = torch.zeros(B, H, S, device=dev)
        self.dq = torch.zeros(T, H * D, dtype=bf, device=dev)
        self.dk = torch.zeros(T, KVH * D, dtype=bf, device=dev)
        self.dv = torch.zeros(T, KVH * D, dtype=bf, device=dev)

    def fwd(self):
        B, S, H, KVH, D = self.dims
        st = torch.cuda.current_stream().cuda_stream
        rc = self.lib.mgen_attn_fwd(self.q.data_ptr(), self.k.data_ptr(), self.v.data_ptr(), self.o.data_ptr(),
                                    self.lse.data_ptr(), H * D, KVH * D, KVH * D, H * D, B, S, H, KVH, D, D ** -0.5, st)
        assert rc == 0, self.lib.mgen_last_error()

    def bwd(self):
        B, S, H, KVH, D = self.dims
        st = torch.cuda.current_stream().cuda_stream
        rc = self.lib.mgen_attn_bwd(self.q.data_ptr(), self.k.data_ptr(), self.v.data_ptr(), self.o.data_ptr(),
                                    self.do.data_ptr(), self.lse.data_ptr(), self.delta.data_ptr(), self.dq.data_ptr(),
                                    self.dk.data_ptr(), self.dv.data_ptr(), H * D, KVH * D, KVH * D, H * D, H * D,
                                    H * D, KVH * D, KVH * D, B, S, H, KVH, D, D ** -0.5, st)
        assert rc == 0, self.lib.mgen_last_error()


def rel(a, b):
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-30))


def timeit(fn, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    base, cand = load(sys.argv[1]), load(sys.argv[2])
    what = sys.argv[3] if len(sys.argv) > 3 else "both"
    shapes = [(2, 77, 2, 2, 128), (1, 300, 4, 2, 128), (2, 130, 4, 4, 64), (1, 67, 2, 2, 64), (2, 1024, 8, 8, 128),
              (1, 513, 4, 1, 128), (2, 512, 6, 6, 64)]
    ok = True
    for (B, S, H, KVH, D) in shapes:
        t = make(B, S, H, KVH, D, seed=S)
        a, b = Run(base, B, S, H, KVH, D, t), Run(cand, B, S, H, KVH, D, t)
        a.fwd(); b.fwd()
        errs = {"o": rel(b.o, a.o), "lse": rel(b.lse, a.lse)}
        if what in ("bwd", "both"):
            b.o.copy_(a.o); b.lse.copy_(a.lse)   # same forward results into both backward passes
            a.bwd(); b.bwd()
            errs.update(dq=rel(b.dq, a.dq), dk=rel(b.dk, a.dk), dv=rel(b.dv, a.dv))
        torch.cuda.synchronize()
        bad = any(not (e < 8e-3) for e in errs.values())   # bf16 outputs: one ulp of the largest element is 4e-3
        ok &= not bad
        print(("BAD " if bad else "ok  "), (B, S, H, KVH, D), {k: float("%.2e" % e) for k, e in errs.items()}, flush=True)
    if len(sys.argv) >= 8:
        bench_shapes = [tuple(int(x) for x in sys.argv[4:8])]
    else:
        bench_shapes = [(32, 2048, 8, 128), (8, 4096, 24, 64)]
    for (B, S, H, D) in bench_shapes:
        t = make(B, S, H, H, D, seed=1)
        a, b = Run(base, B, S, H, H, D, t), Run(cand, B, S, H, H, D, t)
        fl = 4.0 * B * H * S * S / 2 * D
        for name, mult in (("fwd", 1.0), ("bwd", 2.5)):
            if what not in (name, "both"):
                continue
            fa, fb = getattr(a, name), getattr(b, name)
            fa(); fb(); fa(); fb()
            torch.cuda.synchronize()
            ta, tb = [], []
            for _ in range(5):
                ta.append(timeit(fa)); tb.append(timeit(fb))
            ta.sort(); tb.sort()
            print(f"{name} B{B} S{S} H{H} D{D}: base {ta[2]:8.1f} us ({fl*mult/ta[2]/1e6:7.1f} TF/s)   "
                  f"cand {tb[2]:8.1f} us ({fl*mult/tb[2]/1e6:7.1f} TF/s)   x{ta[2]/tb[2]:.3f}", flush=True)
    print("ALL OK" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)


main()
