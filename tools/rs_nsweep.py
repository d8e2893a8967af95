"""Register-panel GEMM: time against the number of output columns (library built with EXTRA=-DMGEN_RS_EXPERIMENTS)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "symbolic-music-generation_amd"))
import torch
from mgen import ops
dev = torch.device("cuda:0")
M, K = 65536, 512
a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
for N in (1024, 1536, 2048, 3072, 4096, 6144, 12288):
    b = (torch.randn(N, K, device=dev) * 0.5).to(torch.bfloat16)
    c = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    run = lambda: ops.gemm(a, b, c, M, N, K, K, K, N, False, False, ops.OUT_BF16)
    res = []
    for dbg in sys.argv[1:] or ["0", "1"]:
        os.environ["MGEN_RS_DBG"] = dbg
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        res.append(f"dbg={dbg}: {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TF/s")
    print(f"N={N:6d} blocks {N//32:4d}  " + "   ".join(res), flush=True)
