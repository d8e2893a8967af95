"""Is an MFMA-bound GEMM limited by the number of CUs or by the clock the chip holds? The tile-loop kernel on
8192^3 bf16 operands (many more workgroups than CUs) on streams restricted to 256 / 224 / 192 / 128 CUs
(hipExtStreamCreateWithCUMask): PFLOP/s per run and per active CU."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "symbolic-music-generation_amd"))
from mgen import ops  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
M = N = K = 8192
a = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
os.environ["MGEN_GEMM_OS"] = "0"
os.environ["MGEN_GEMM_RS"] = "0"


def masked_stream(ncu):
    # CU i of XCD x is bit x + 8 * i of the mask (round-robin numbering): keep the first ncu / 8 CUs of every XCD
    words = (ctypes.c_uint32 * 8)()
    per = ncu // 8
    for x in range(8):
        for i in range(per):
            bit = x + 8 * i
            words[bit // 32] |= 1 << (bit % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return s


for ncu in (256, 224, 192, 128):
    s = masked_stream(ncu)
    ops.use_stream(s.value)
    f = lambda: ops.gemm(a, w, c, M, N, K, K, K, N)
    for _ in range(3):
        f()
    hip.hipStreamSynchronize(s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    es = torch.cuda.ExternalStream(s.value)
    e0.record(es)
    n = 20
    for _ in range(n):
        f()
    e1.record(es)
    hip.hipStreamSynchronize(s)
    ops.use_stream(None)
    ms = e0.elapsed_time(e1) / n
    pf = 2.0 * M * N * K / ms / 1e12
    print(f"{ncu:3d} CUs: {ms:6.3f} ms  {pf:5.3f} PFLOP/s  {pf / ncu * 1e3:5.2f} TFLOP/s per CU", flush=True)
