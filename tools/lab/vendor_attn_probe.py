"""Reference point, not part of the product: torch.nn.functional.scaled_dot_product_attention (the flash kernels
PyTorch-ROCm ships) on the training step's attention shape, next to this library's kernels on the same box."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "symbolic-music-generation_amd"))
from mgen import ops  # noqa: E402

B, S, H, D = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (32, 2048, 8, 128)))
dev = "cuda"


def timeit(f, n=5):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


q = torch.randn(B, H, S, D, device=dev, dtype=torch.bfloat16, requires_grad=True)
k = torch.randn(B, H, S, D, device=dev, dtype=torch.bfloat16, requires_grad=True)
v = torch.randn(B, H, S, D, device=dev, dtype=torch.bfloat16, requires_grad=True)
do = torch.randn(B, H, S, D, device=dev, dtype=torch.bfloat16)
fl = 4.0 * B * H * S * S / 2 * D
try:
    t_f = timeit(lambda: F.scaled_dot_product_attention(q, k, v, is_causal=True))

    def fb():
        o = F.scaled_dot_product_attention(q, k, v, is_causal=True)
        o.backward(do)
        q.grad = k.grad = v.grad = None
    t_fb = timeit(fb)
    print(f"torch SDPA  B{B} S{S} H{H} D{D}: fwd {t_f:8.1f} us {fl / t_f / 1e6:7.1f} TF/s | bwd {t_fb - t_f:8.1f} us "
          f"{2.5 * fl / (t_fb - t_f) / 1e6:7.1f} TF/s (5-product count)")
except Exception as e:   # noqa: BLE001
    print("torch SDPA failed:", repr(e)[:300])
T, HD = B * S, H * D
bf = torch.bfloat16
q2 = torch.randn(T, HD, device=dev).to(bf); k2 = torch.randn(T, HD, device=dev).to(bf)
v2 = torch.randn(T, HD, device=dev).to(bf); do2 = torch.randn(T, HD, device=dev).to(bf)
o2 = torch.empty(T, HD, dtype=bf, device=dev); lse = torch.empty(B, H, S, device=dev)
dq = torch.empty_like(q2); dk = torch.empty_like(q2); dv = torch.empty_like(q2); delta = torch.empty(B, H, S, device=dev)
sc = D ** -0.5
t_f = timeit(lambda: ops.attn_fwd(q2.data_ptr(), k2.data_ptr(), v2.data_ptr(), o2.data_ptr(), lse, HD, HD, HD, HD, B, S, H, H, D, sc))
t_b = timeit(lambda: ops.attn_bwd(q2.data_ptr(), k2.data_ptr(), v2.data_ptr(), o2.data_ptr(), do2.data_ptr(), lse, delta,
                                  dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), HD, HD, HD, HD, HD, HD, HD, HD, B, S, H, H, D, sc))
print(f"this library B{B} S{S} H{H} D{D}: fwd {t_f:8.1f} us {fl / t_f / 1e6:7.1f} TF/s | bwd {t_b:8.1f} us "
      f"{2.5 * fl / t_b / 1e6:7.1f} TF/s (5-product count)")
