"""Find which tokens the sampler kernel keeps (noise trick) and compare with the oracle warpers (GPU box only)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "symbolic-music-generation_amd"))
from mgen import ops  # noqa: E402
from oracle import ref_cpu as R  # noqa: E402

V, B = 1024, 6
g = torch.Generator().manual_seed(V)
logits = (torch.randn(B, V, generator=g) * 2.5).to(torch.bfloat16)
logits[1] = (torch.randn(V, generator=g) * 0.3).to(torch.bfloat16)
logits[2, 123] = 30.0
kw = dict(temperature=float(sys.argv[1]) if len(sys.argv) > 1 else 0.7, top_k=0,
          top_p=float(sys.argv[2]) if len(sys.argv) > 2 else 0.1)
warped = R.warp_logits(logits.float(), **kw)
lg = logits.cuda()
out = torch.empty(B, dtype=torch.int64, device="cuda")
for row in range(B):
    kept_ref = set(torch.nonzero(torch.isfinite(warped[row])).flatten().tolist())
    order = torch.argsort(logits[row].float(), descending=True)[:max(40, len(kept_ref) + 10)].tolist()
    kept = set()
    for idx in order:
        noise = torch.full((B, V), 1.0)
        noise[row, idx] = 1e-30
        ops.sample(lg, V, out, do_sample=True, noise=noise.cuda(), **kw)
        if int(out[row]) == idx:
            kept.add(idx)
    print(row, "ref", len(kept_ref), "kernel", len(kept), "only_ref", sorted(kept_ref - kept)[:8], "only_kernel",
          sorted(kept - kept_ref)[:8])
