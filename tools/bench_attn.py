"""Attention micro-benchmark on the GPU box (B=32, S=2048, H=8, D=128 by default), HIP-event timing per kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "symbolic-music-generation_amd"))
import torch
from mgen import ops
from mgen._lib import call

B, S, H, D = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (32, 2048, 8, 128)))
dev = torch.device("cuda:0")
T = B * S
bf = torch.bfloat16
# DATA_SCALE=0: all-zero operands (same instruction stream, least switching power: separates what the clock under
# load costs from what instruction issue costs)
_sc = float(os.environ.get("DATA_SCALE", "1"))
q = (torch.randn(T, H * D, device=dev) * _sc).to(bf); k = (torch.randn(T, H * D, device=dev) * _sc).to(bf)
v = (torch.randn(T, H * D, device=dev) * _sc).to(bf); do = (torch.randn(T, H * D, device=dev) * _sc).to(bf)
o = torch.empty(T, H * D, dtype=bf, device=dev); lse = torch.empty(B, H, S, device=dev)
dq = torch.empty_like(q); dk = torch.empty_like(q); dv = torch.empty_like(q); delta = torch.empty(B, H, S, device=dev)
HD = H * D
scale = D ** -0.5
def fwd(): ops.attn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse, HD, HD, HD, HD, B, S, H, H, D, scale)
def bwd(): ops.attn_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse, delta, dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), HD, HD, HD, HD, HD, HD, HD, HD, B, S, H, H, D, scale)
flops_fwd = 4.0 * B * H * S * S / 2 * D
for name, fn, fl in (("fwd", fwd, flops_fwd), ("bwd(dkv+dq+delta)", bwd, flops_fwd * 3.5)):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name:20s} {ms*1e3:9.1f} us  {fl/ms/1e9:8.1f} TFLOP/s", flush=True)
