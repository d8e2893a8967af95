"""Time of one forward product (x W^T, bf16 out) against the reduction length: the slope is the cost of a K step, the
intercept the per-tile cost outside the K loop (first load, epilogue)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "symbolic-music-generation_amd"))
import torch
from mgen import ops

dev = torch.device("cuda:0")
M, N = 65536, 3072
for mode, nm in ((ops.OUT_BF16, "bf16 out"), (ops.OUT_F32, "f32 out")):
    for K in (64, 128, 256, 512, 1024, 2048, 4096):
        a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
        b = (torch.randn(N, K, device=dev) * 0.5).to(torch.bfloat16)
        c = torch.zeros(M, N, dtype=torch.bfloat16 if mode == ops.OUT_BF16 else torch.float32, device=dev)
        run = lambda: ops.gemm(a, b, c, M, N, K, K, K, N, False, False, mode)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        tiles_per_cu = (M // 256) * (N // 256) / 256
        print(f"{nm} K={K:5d} {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s  per tile {ms*1e3/tiles_per_cu:6.2f} us "
              f"({K//64} steps)", flush=True)
