"""RMSNorm backward alone at the row lengths of the two models, for every variant library under tools/lab/_build/
(norm.hip + runtime.hip built with other launch shapes): time per launch and the bytes it has to move."""
import ctypes
import glob
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
shapes = [(32768, 1536), (16384, 1536), (65536, 512)]
dev = "cuda"
for path in sorted(glob.glob(os.path.join(HERE, "_build", "norm_*.so"))):
    lib = ctypes.CDLL(path)
    f = lib.mgen_rmsnorm_bwd
    f.argtypes = [P, P, P, P, P, P, P, L, I, I, I, P]
    f.restype = I
    for T, d in shapes:
        dy = torch.randn(T, d, device=dev).bfloat16()
        x = torch.randn(T, d, device=dev)
        w = torch.ones(d, device=dev)
        rstd = torch.rand(T, device=dev) + 0.5
        dres = torch.randn(T, d, device=dev)
        lp = torch.empty(T, d, device=dev, dtype=torch.bfloat16)
        dw = torch.zeros(d, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        call = lambda: f(dy.data_ptr(), x.data_ptr(), w.data_ptr(), rstd.data_ptr(), dres.data_ptr(), lp.data_ptr(),
                         dw.data_ptr(), T, d, 1, 1, st)
        for _ in range(3):
            assert call() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        nbytes = T * d * (2 + 4 + 4 + 4 + 2)
        print(f"{os.path.basename(path):22s} T {T:6d} d {d:5d}: {us:7.1f} us  {nbytes / us / 1e6:5.2f} TB/s", flush=True)
