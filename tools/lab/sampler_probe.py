"""The sampler alone on rows shaped like the decode bench's (random-init logits: nearly every token survives top-p
0.95): time per launch for 128 and 256 rows, sampled and greedy."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "symbolic-music-generation_amd"))
from mgen import ops  # noqa: E402

V, ld = 151936, 151936
for B in (128, 256):
    for scale in (0.3, 3.0):
        logits = (torch.randn(B, ld, device="cuda") * scale).bfloat16()
        out = torch.empty(B, dtype=torch.int64, device="cuda")
        for name, kw in (("greedy", dict(do_sample=False)), ("T.97 p.95", dict(do_sample=True, temperature=0.97, top_p=0.95)),
                         ("T.97 p1", dict(do_sample=True, temperature=0.97, top_p=1.0)),
                         ("k50 p.95", dict(do_sample=True, temperature=0.97, top_k=50, top_p=0.95))):
            f = lambda: ops.sample(logits, V, out, seed=1, offset=3, **kw)
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                f()
            e1.record()
            torch.cuda.synchronize()
            print(f"rows {B} sigma {scale}: {name:10s} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us", flush=True)
