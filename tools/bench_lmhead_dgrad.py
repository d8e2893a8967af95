"""lm_head input gradient dX[T, 512] = dlogits[T, V] @ W[V, 512] at the training shape (T = 32768, V = 151936):
time per launch with HIP events.  MGEN_GEMM_LONGK=0|1 selects the tiling."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "symbolic-music-generation_amd"))
import torch
from mgen import ops
dev = torch.device("cuda:0")
T, V, d = 32768, 151936, 512
g = torch.Generator(device=dev).manual_seed(0)
dl = (torch.randn(T, V, device=dev, generator=g, dtype=torch.float32) * 1e-3).to(torch.bfloat16) if False else torch.empty(T, V, dtype=torch.bfloat16, device=dev).normal_(0, 1e-3)
W = torch.empty(V, d, dtype=torch.bfloat16, device=dev).normal_(0, 0.02)
dx = torch.empty(T, d, dtype=torch.bfloat16, device=dev)
for _ in range(2):
    ops.linear_dgrad(dl, W, dx)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ops.linear_dgrad(dl, W, dx)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"LONGK={os.environ.get('MGEN_GEMM_LONGK', '1')}: {ms:.3f} ms  {2.0 * T * V * d / ms / 1e9:.1f} TFLOP/s  checksum {float(dx.float().abs().sum()):.6e}")
