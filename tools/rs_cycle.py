"""Register-panel GEMM with the outputs cycling over many buffers (as in the model, where every layer keeps its
activations) against the same launches into one buffer set."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "symbolic-music-generation_amd"))
import torch
from mgen import ops
dev = torch.device("cuda:0")
M, K, I = 65536, 512, 2048
NB = 14
xs = [(torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16) for _ in range(NB)]
w = (torch.randn(2 * I, K, device=dev) * 0.1).to(torch.bfloat16)
wq = (torch.randn(3072, K, device=dev) * 0.1).to(torch.bfloat16)
gus = [torch.zeros(M, 2 * I, dtype=torch.bfloat16, device=dev) for _ in range(NB)]
acts = [torch.zeros(M, I, dtype=torch.bfloat16, device=dev) for _ in range(NB)]
qkvs = [torch.zeros(M, 3072, dtype=torch.bfloat16, device=dev) for _ in range(NB)]


def timeit(fn, n):
    fn(0); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for flag in ("1", "0"):
    os.environ["MGEN_GEMM_RS"] = flag
    a = timeit(lambda i: ops.linear_swiglu_fwd(xs[0], w, gus[0], acts[0]), 28)
    b = timeit(lambda i: ops.linear_swiglu_fwd(xs[i % NB], w, gus[i % NB], acts[i % NB]), 28)
    c = timeit(lambda i: ops.gemm(xs[0], wq, qkvs[0], M, 3072, K, K, K, 3072, False, False, ops.OUT_BF16), 28)
    d = timeit(lambda i: ops.gemm(xs[i % NB], wq, qkvs[i % NB], M, 3072, K, K, K, 3072, False, False, ops.OUT_BF16), 28)
    print(f"RS={flag}: swiglu one buffer {a:7.1f} us, cycling {b:7.1f} us | qkv one buffer {c:7.1f} us, cycling {d:7.1f} us", flush=True)
