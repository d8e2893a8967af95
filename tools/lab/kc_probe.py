"""The MFMA-bound GEMM kernels on the step's shapes, a few launches each, for counter passes:
   rocprofv3 --pmc <counters> -- python3 tools/lab/kc_probe.py
(kc: input-gradient form of the output-stationary kernel; os: weight gradient; rs: register panel)."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "symbolic-music-generation_amd"))
from mgen import ops

dev = "cuda:0"
g = torch.Generator().manual_seed(0)
T = 65536


scale = float(os.environ.get("DATA_SCALE", "1"))   # 0: all-zero operands (what the chip does without toggling data)


def rnd(*shape, s=0.5):
    return (torch.randn(*shape, generator=g) * (s * scale)).to(torch.bfloat16).to(dev)


reps = int(os.environ.get("REPS", "5"))
# kc: dx[T,512] = dy[T,3072] @ W[3072,512]
dy, w = rnd(T, 3072), rnd(3072, 512, s=0.02)
dx = torch.empty(T, 512, dtype=torch.bfloat16, device=dev)
# os wgrad: dW[3072,512] += dy^T x
x = rnd(T, 512)
dw = torch.zeros(3072, 512, dtype=torch.float32, device=dev)
# rs: qkv[T,3072] = x @ Wq[3072,512]^T
wq = rnd(3072, 512, s=0.02)
qkv = torch.empty(T, 3072, dtype=torch.bfloat16, device=dev)
for _ in range(reps):
    ops.gemm(dy, w, dx, T, 512, 3072, 3072, 512, 512, False, True, ops.OUT_BF16)
    ops.gemm(dy, x, dw, 3072, 512, T, 3072, 512, 512, True, True, ops.OUT_F32_ACC)
    ops.gemm(x, wq, qkv, T, 3072, 512, 512, 512, 3072, False, False, ops.OUT_BF16)
torch.cuda.synchronize()
print("done")
