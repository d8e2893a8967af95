"""Time the decode attention kernel alone (GPU box only): 256 streams x 8 heads x head_dim 128 (HEADS=H,KVH,D for another shape), bf16 K/V cache rows.
Arguments: context length(s); MAXLEN=<n> sets the cache's row capacity (the pitch between streams)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "symbolic-music-generation_amd"))
from mgen import ops  # noqa: E402

B = int(os.environ.get("STREAMS", "256"))
H, KVH, D = (int(x) for x in os.environ.get("HEADS", "8,8,128").split(","))   # HEADS=24,24,64: the 650M model
dev = "cuda"
lens = [int(a) for a in sys.argv[1:]] or [1920]
for max_len in [int(x) for x in os.environ.get("MAXLEN", "2048").split(",")]:
    k = torch.randn(B, KVH, max_len, D, device=dev).bfloat16()
    v = torch.randn(B, KVH, max_len, D, device=dev).bfloat16()
    q = torch.randn(B, H * D, device=dev).bfloat16()
    out = torch.empty(B, H * D, device=dev, dtype=torch.bfloat16)
    for L in lens:
        kv_len = torch.full((B,), L, dtype=torch.int32, device=dev)
        for nsplit in [int(x) for x in os.environ.get("NSPLIT", "1,2,4").split(",")]:
            ws = torch.empty(max(1, ops.attn_decode_workspace(B, H, D, nsplit)), dtype=torch.float32, device=dev)   # counted in floats
            f = lambda: ops.attn_decode(q.data_ptr(), H * D, k, v, kv_len, out, H * D, ws, B, H, KVH, D, max_len, nsplit,
                                        D ** -0.5)
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            n = 20
            for _ in range(n):
                f()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            nbytes = 2.0 * B * KVH * L * D * 2
            print(f"max_len {max_len:5d} len {L:5d} nsplit {nsplit}: {us:7.1f} us  {nbytes / us / 1e6:6.2f} TB/s", flush=True)
