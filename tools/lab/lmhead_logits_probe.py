"""Why is the lm_head logits product slower inside the step (4.4 ms per 32768-row chunk) than timed alone (3.9 ms)?
Same kernel, different operand statistics and neighbours."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "symbolic-music-generation_amd"))
from mgen import ops  # noqa: E402

dev = "cuda"
M, N, K = 32768, 151936, 512


def timeit(f, n=6):
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


g = torch.Generator(device=dev).manual_seed(0)
c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for name, sa, sb in (("N(0,1) x N(0,1)", 1.0, 1.0), ("N(0,1) x N(0,0.02)", 1.0, 0.02), ("zeros", 0.0, 0.0)):
    a = (torch.randn(M, K, device=dev, generator=g) * sa).bfloat16()
    b = (torch.randn(N, K, device=dev, generator=g) * sb).bfloat16()
    us = timeit(lambda: ops.gemm(a, b, c, M, N, K, K, K, N, False, False, ops.OUT_BF16))
    print(f"{name:22s}: {us:8.1f} us  {2.0 * M * N * K / us / 1e9:5.2f} PF", flush=True)
# with the cross entropy (HBM-bound, rewrites the output) between the products, as in the step
a = torch.randn(M, K, device=dev, generator=g).bfloat16()
b = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
labels = torch.randint(0, 30000, (16, 2048), device=dev)
inv = torch.ones(1, device=dev) / M
rl = torch.empty(M, device=dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]


def pair():
    ev[0].record()
    ops.gemm(a, b, c, M, N, K, K, K, N, False, False, ops.OUT_BF16)
    ev[1].record()
    ops.ce_fwd_bwd(c, labels, 0, M, N, N, 2048, inv, 1.0, rl, None, write_grad=True)
    ev[2].record()


for _ in range(2):
    pair()
tg = tc = 0.0
for _ in range(5):
    pair()
    torch.cuda.synchronize()
    tg += ev[0].elapsed_time(ev[1])
    tc += ev[1].elapsed_time(ev[2])
print(f"alternating with the cross entropy: product {tg / 5 * 1e3:8.1f} us, cross entropy {tc / 5 * 1e3:8.1f} us", flush=True)
