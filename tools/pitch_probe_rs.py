"""Does the output row pitch matter for the register-panel GEMM? QKV and gate/up + SwiGLU with dense and padded rows."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "symbolic-music-generation_amd"))
import torch
from mgen import ops
from mgen._lib import call
dev = torch.device("cuda:0")
M, K, I = 65536, 512, 2048


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
for pad in (0, 64):
    for N in (3072, 2048, 1024):
        w = (torch.randn(N, K, device=dev) * 0.1).to(torch.bfloat16)
        c = torch.zeros(M, N + pad, dtype=torch.bfloat16, device=dev)
        us = timeit(lambda: ops.gemm(x, w, c, M, N, K, K, K, N + pad, False, False, ops.OUT_BF16))
        print(f"plain N={N} pitch {2*(N+pad)} B: {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TF/s", flush=True)
    wg = (torch.randn(2 * I, K, device=dev) * 0.1).to(torch.bfloat16)
    gu = torch.zeros(M, 2 * I + pad, dtype=torch.bfloat16, device=dev)
    act = torch.zeros(M, I + pad, dtype=torch.bfloat16, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    us = timeit(lambda: call("mgen_gemm_bf16_swiglu", x.data_ptr(), wg.data_ptr(), gu.data_ptr(), act.data_ptr(), M, I, K,
                             K, K, 2 * I + pad, I + pad, s))
    print(f"swiglu pitches {2*(2*I+pad)} / {2*(I+pad)} B: {us:7.1f} us {2.0*M*2*I*K/us/1e6:7.1f} TF/s", flush=True)
