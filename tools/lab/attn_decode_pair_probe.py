"""Two decode-attention kernels of 128 streams each on two HIP streams, against one kernel of 256 streams: the aggregate
K/V read rate when the kernels of the two batch parts overlap (context 1920, 8 heads x 128)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "symbolic-music-generation_amd"))
from mgen import ops  # noqa: E402

H = KVH = 8
D, L, max_len = 128, int(os.environ.get("CTX", "1920")), 2048
dev = "cuda"


NL = int(os.environ.get("LAYERS", "1"))   # distinct cache tensors a launch sequence cycles through


def setup(B):
    if NL > 1:
        fs = [setup_one(B) for _ in range(NL)]
        it = [0]

        def f():
            fs[it[0] % NL]()
            it[0] += 1
        return f
    return setup_one(B)


def setup_one(B):
    k = torch.randn(B, KVH, max_len, D, device=dev).bfloat16()
    v = torch.randn(B, KVH, max_len, D, device=dev).bfloat16()
    q = torch.randn(B, H * D, device=dev).bfloat16()
    out = torch.empty(B, H * D, device=dev, dtype=torch.bfloat16)
    kv_len = torch.full((B,), L, dtype=torch.int32, device=dev)
    ws = torch.empty(max(1, ops.attn_decode_workspace(B, H, D, 1)), dtype=torch.float32, device=dev)
    if os.environ.get("PREP") == "1":   # the form the decode step runs: new token's q/k/v prepared inside the launch
        qkv = torch.randn(B, (H + 2 * KVH) * D, device=dev).bfloat16()
        wq, wk = torch.ones(D, device=dev), torch.ones(D, device=dev)
        cs = torch.randn(max_len, D, device=dev)
        pos = torch.full((B,), L - 1, dtype=torch.int32, device=dev)
        return lambda: ops.attn_decode_prep(qkv, (H + 2 * KVH) * D, wq, wk, cs, pos, 1e-6, False, k, v, kv_len, out, H * D,
                                            ws, B, H, KVH, D, max_len, 1, D ** -0.5)
    return lambda: ops.attn_decode(q.data_ptr(), H * D, k, v, kv_len, out, H * D, ws, B, H, KVH, D, max_len, 1, D ** -0.5)


n = 42
one = setup(256)
for _ in range(3):
    one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    one()
torch.cuda.synchronize()
t1 = time.perf_counter() - t0
nbytes = 2.0 * 256 * KVH * L * D * 2
print(f"one kernel of 256 streams: {t1 / n * 1e6:7.1f} us per launch, {nbytes * n / t1 / 1e12:5.2f} TB/s", flush=True)
a, b = setup(128), setup(128)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for offset in (0, 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        ops.use_stream(sa.cuda_stream)
        a()
        ops.use_stream(sb.cuda_stream)
        b()
    ops.use_stream(None)
    torch.cuda.synchronize()
    t2 = time.perf_counter() - t0
    print(f"two kernels of 128 streams on two HIP streams: {t2 / n * 1e6:7.1f} us per pair, {nbytes * n / t2 / 1e12:5.2f} TB/s",
          flush=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    a()
    b()
torch.cuda.synchronize()
t3 = time.perf_counter() - t0
print(f"the same two kernels on one stream: {t3 / n * 1e6:7.1f} us per pair, {nbytes * n / t3 / 1e12:5.2f} TB/s", flush=True)
