"""o_proj / down_proj as a bf16-output product on the output-stationary kernel (through the transposed weight copy
the training step already keeps) against the fp32 residual form on the tile loop: time per launch."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "symbolic-music-generation_amd"))
from mgen import ops

dev = "cuda:0"
for M, N, K in [(65536, 512, 1024), (65536, 512, 2048)]:
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16).to(dev)
    wt = w.t().contiguous()
    res = torch.randn(M, N, generator=g).to(dev)
    c32 = torch.zeros(M, N, dtype=torch.float32, device=dev)
    c16 = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)

    def t(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20 * 1e3
    t_res = t(lambda: ops.gemm(a, w, c32, M, N, K, K, K, N, False, False, ops.OUT_F32_RESID_BF16ROUND, res, N))
    t_kc = t(lambda: ops.gemm(a, wt, c16, M, N, K, K, N, N, False, True, ops.OUT_BF16))
    t_tl = t(lambda: ops.gemm(a, w, c16, M, N, K, K, K, N, False, False, ops.OUT_BF16))
    ref = c32 - res
    print(f"K={K}: residual form (tile loop) {t_res:.1f} us | bf16 out, transposed weights (output-stationary) {t_kc:.1f} us "
          f"| bf16 out (tile loop / register panel) {t_tl:.1f} us | kc == rounded product: "
          f"{torch.equal(c16.float(), ref) if False else float((c16.float() - ref).abs().max()):.3e}", flush=True)
