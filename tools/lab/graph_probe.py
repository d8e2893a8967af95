"""Can a decode step (ctypes launches on torch's current stream) be captured into a HIP graph through
torch.cuda.graph and replayed? Prints the replay's logits difference from the eager step and the time per replay."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "symbolic-music-generation_amd"))
from mgen import MidiLM, default_decoder, ops  # noqa: E402
from mgen.generation import KVCache, _DecodeBuffers  # noqa: E402

dev = torch.device("cuda:0")
cfg = default_decoder()
m = MidiLM(cfg, device=dev, seed=0)
m.sync_shadow()
B, total, t = int(os.environ.get("STREAMS", "1")), 512, 100
cache = KVCache(cfg, B, total, dev)
for x in cache.k + cache.v:
    x.normal_()
lens = torch.arange(total + 1, dtype=torch.int32, device=dev)[:, None].expand(total + 1, B).contiguous()
m._rope_table(total)
buf = _DecodeBuffers(cfg, B, dev)
buf.next.fill_(17)
ws = torch.empty(max(1, ops.attn_decode_workspace(B, cfg.num_attention_heads, cfg.head_dim, 64)), dtype=torch.float32,
                 device=dev)
eager = m._decode_step(buf.next, cache, t, buf, lens, ws).clone()
torch.cuda.synchronize()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    m._decode_step(buf.next, cache, t, buf, lens, ws)
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        m._decode_step(buf.next, cache, t, buf, lens, ws)
    buf.logits.zero_()
    g.replay()
    s.synchronize()
    print("max |replay - eager| =", float((buf.logits.float() - eager.float()).abs().max()), flush=True)
    for name, f in (("replay", g.replay), ("eager", lambda: m._decode_step(buf.next, cache, t, buf, lens, ws))):
        for _ in range(5):
            f()
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            f()
        s.synchronize()
        print(f"{name}: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us per step ({B} streams, context {t})", flush=True)
