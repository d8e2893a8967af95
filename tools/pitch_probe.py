"""Does the row pitch of the reduction-contiguous operand matter for the input-gradient kernel? Same product with
the A rows 8192 bytes apart and 8320 bytes apart (64 padding elements)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "symbolic-music-generation_amd"))
import torch
from mgen import ops
dev = torch.device("cuda:0")
M, N = 65536, 512
for K in (4096, 3072, 2048):
    for pad in (0, 64):
        a = (torch.randn(M, K + pad, device=dev) * 0.5).to(torch.bfloat16)
        b = (torch.randn(K, N, device=dev) * 0.1).to(torch.bfloat16)
        c = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        run = lambda: ops.gemm(a, b, c, M, N, K, a.stride(0), N, N, False, True, ops.OUT_BF16)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"K={K} pitch {2*(K+pad)} B: {us:7.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s", flush=True)
