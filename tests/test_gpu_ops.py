"""Per-kernel parity tests: every C-ABI kernel against plain torch fp32 arithmetic on the CPU for the same
(bf16-rounded) inputs. Run on the GPU box with -m gpu."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R


def dev():
    return torch.device("cuda:0")


def bf(x):
    return x.to(torch.bfloat16)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.fixture(scope="module")
def ops():
    from mgen import ops as o
    return o


# ----------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 512), (300, 200, 136), (1000, 72, 1096),
                                   # enough row tiles for the 256-wide workgroup tiles (64x128 per wave), one with a
                                   # ragged last column tile and a ragged last row tile
                                   (16384, 1024, 128), (32700, 600, 192),
                                   # 256x256 workgroup tiles (8 waves, double-buffered), accumulate epilogue included,
                                   # a single K step
                                   (33000, 1096, 64), (16500, 2040, 320)])
def test_gemm_layouts(ops, ta, tb, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = bf(torch.randn(M, K, generator=g))
    B = bf(torch.randn(K, N, generator=g))
    ref = A.float() @ B.float()
    ld = lambda n: (n + 7) // 8 * 8
    # stored forms, zero padded to a multiple of 8 columns
    def stored(mat):
        r, c = mat.shape
        out = torch.zeros(r, ld(c), dtype=torch.bfloat16)
        out[:, :c] = mat
        return out.to(dev())
    a_st = stored(A.t().contiguous()) if ta else stored(A)
    b_st = stored(B) if tb else stored(B.t().contiguous())
    ldc = ld(N) + 8
    c = torch.full((M, ldc), 7.0, dtype=torch.float32, device=dev())
    ops.gemm(a_st, b_st, c, M, N, K, a_st.stride(0), b_st.stride(0), ldc, ta, tb, ops.OUT_F32)
    assert rel(c[:, :N], ref) < 2e-5
    assert bool((c[:, N:] == 7.0).all()), "wrote outside the N columns"
    cb = torch.zeros(M, ldc, dtype=torch.bfloat16, device=dev())
    ops.gemm(a_st, b_st, cb, M, N, K, a_st.stride(0), b_st.stride(0), ldc, ta, tb, ops.OUT_BF16)
    assert rel(cb[:, :N].float(), ref) < 1e-2
    assert bool((cb[:, N:] == 0).all())
    # accumulate and residual epilogues
    ops.gemm(a_st, b_st, c, M, N, K, a_st.stride(0), b_st.stride(0), ldc, ta, tb, ops.OUT_F32_ACC)
    assert rel(c[:, :N], 2 * ref) < 2e-5
    res = torch.randn(M, N, generator=g).to(dev())
    out = torch.empty(M, N, dtype=torch.float32, device=dev())
    ops.gemm(a_st, b_st, out, M, N, K, a_st.stride(0), b_st.stride(0), N, ta, tb, ops.OUT_F32_RESID_BF16ROUND, res, N)
    assert rel(out, res.cpu() + bf(ref).float()) < 1e-2


@pytest.mark.parametrize("M,N,K", [(256, 128, 8192), (512, 1024, 4096 + 64), (100, 72, 5000),
                                   (520, 1000, 8192 + 64)])   # the last: 128x256 tiles with split-K atomics
def test_gemm_weight_gradient_split_k(ops, M, N, K):
    """dW[M,N] += dY[K,M]^T X[K,N] with few output tiles and a long reduction: the split-K (atomic) path."""
    g = torch.Generator().manual_seed(K)
    A = bf(torch.randn(K, M, generator=g) * 0.5)
    B = bf(torch.randn(K, N, generator=g) * 0.5)
    ref = A.float().t() @ B.float()
    ldm, ldn = (M + 7) // 8 * 8, (N + 7) // 8 * 8
    a_st = torch.zeros(K, ldm, dtype=torch.bfloat16)
    a_st[:, :M] = A
    b_st = torch.zeros(K, ldn, dtype=torch.bfloat16)
    b_st[:, :N] = B
    a_st, b_st = a_st.to(dev()), b_st.to(dev())
    prev = torch.randn(M, N, generator=g)
    c = prev.clone().to(dev())
    ops.gemm(a_st, b_st, c, M, N, K, ldm, ldn, N, True, True, ops.OUT_F32_ACC)
    assert rel(c, prev + ref) < 1e-5
    ops.gemm(a_st, b_st, c, M, N, K, ldm, ldn, N, True, True, ops.OUT_F32_ACC)
    assert rel(c, prev + 2 * ref) < 1e-5


@pytest.mark.parametrize("M,N,K,ldc", [(300, 200, 32, 200), (1000, 512, 32, 1024), (257, 72, 64, 76)])
def test_gemm_accumulate_into_bf16(ops, M, N, K, ldc):
    """out_mode 6: C (bf16) = bf16(C + A B^T), the low-rank adapter epilogue; vector path (ldc % 8 == 0) and the
    element-wise one, C a column block of a wider matrix."""
    g = torch.Generator().manual_seed(M + N)
    A = bf(torch.randn(M, K, generator=g))
    B = bf(torch.randn(N, K, generator=g) * 0.3)
    C0 = bf(torch.randn(M, ldc, generator=g))
    c = C0.clone().to(dev())
    ops.gemm(A.to(dev()), B.to(dev()), c, M, N, K, K, K, ldc, out_mode=ops.OUT_BF16_ACC)
    want = C0.float().clone()
    want[:, :N] += A.float() @ B.float().t()
    assert rel(c[:, :N].float(), want[:, :N]) < 1e-2
    assert torch.equal(c[:, N:].cpu(), C0[:, N:]), "wrote outside the N columns"


def test_gemm_exact_identity_asymmetric(ops):
    """A = I against an asymmetric integer B: catches swapped rows/columns and a wrong k order exactly."""
    n = 128
    A = torch.eye(n)
    B = (torch.arange(n)[:, None] * 3 + torch.arange(n)[None, :] * 1).float() % 251  # B[k][j], asymmetric
    for ta in (False, True):
        for tb in (False, True):
            a_st = bf(A).to(dev())
            b_st = bf(B if tb else B.t().contiguous()).to(dev())
            c = torch.empty(n, n, dtype=torch.float32, device=dev())
            ops.gemm(a_st, b_st, c, n, n, n, n, n, n, ta, tb, ops.OUT_F32)
            assert torch.equal(c.cpu(), B), (ta, tb)
    # and a non-trivial permutation on the A side (row i of A picks row (5i+3) mod n of B)
    perm = (torch.arange(n) * 5 + 3) % n
    A = torch.zeros(n, n)
    A[torch.arange(n), perm] = 1.0
    for ta in (False, True):
        a_st = bf(A.t().contiguous() if ta else A).to(dev())
        b_st = bf(B).to(dev())
        c = torch.empty(n, n, dtype=torch.float32, device=dev())
        ops.gemm(a_st, b_st, c, n, n, n, n, n, n, ta, True, ops.OUT_F32)
        assert torch.equal(c.cpu(), B[perm]), ta


def test_gemm_rejects_bad_arguments(ops):
    from mgen._lib import MgenError
    a = torch.zeros(8, 8, dtype=torch.bfloat16, device=dev())
    c = torch.zeros(8, 8, dtype=torch.float32, device=dev())
    with pytest.raises(MgenError):
        ops.gemm(a, a, c, 8, 8, 8, 4, 8, 8)  # lda not a multiple of 8
    with pytest.raises(MgenError):
        ops.gemm(a, a, c, 8, 8, 8, 8, 8, 8, out_mode=99)


# ----------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("T,d", [(7, 512), (1000, 256), (33, 1536)])
def test_rmsnorm_fwd_bwd(ops, T, d):
    g = torch.Generator().manual_seed(T + d)
    x = torch.randn(T, d, generator=g) * 2
    w = 1 + 0.1 * torch.randn(d, generator=g)
    dy = bf(torch.randn(T, d, generator=g))
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y = R.rmsnorm(xr, wr, 1e-6)
    y.backward(dy.float())
    xd, wd = x.to(dev()), w.to(dev())
    yb = torch.empty(T, d, dtype=torch.bfloat16, device=dev())
    yf = torch.empty(T, d, dtype=torch.float32, device=dev())
    rstd = torch.empty(T, dtype=torch.float32, device=dev())
    ops.rmsnorm_fwd(xd, wd, yf, rstd, 1e-6)
    assert rel(yf, y.detach()) < 1e-5
    ops.rmsnorm_fwd(xd, wd, yb, rstd, 1e-6)
    assert torch.equal(yb.cpu(), bf(yf.cpu())) or rel(yb.float(), y.detach()) < 1e-2
    prev = torch.randn(T, d, generator=g)
    dres = prev.clone().to(dev())
    dlp = torch.empty(T, d, dtype=torch.bfloat16, device=dev())
    dw = torch.zeros(d, dtype=torch.float32, device=dev())
    ops.rmsnorm_bwd(dy.to(dev()), xd, wd, rstd, dres, dlp, dw, accumulate=True)
    assert rel(dres, prev + xr.grad) < 1e-5
    assert rel(dw, wr.grad) < 1e-4
    assert rel(dlp.float(), prev + xr.grad) < 1e-2
    ops.rmsnorm_bwd(dy.to(dev()), xd, wd, rstd, dres, None, dw, accumulate=False)
    assert rel(dres, xr.grad) < 1e-5


def test_rmsnorm_bwd_overlapping_streams(ops):
    """Launches that overlap on one GPU (one per stream, several rounds) keep their gain gradients apart: each
    (device, stream) owns a set of partial-sum rows; more streams than sets fall back to plain atomics."""
    T, d, NS = 20000, 512, 20
    g = torch.Generator().manual_seed(5)
    streams = [torch.cuda.Stream() for _ in range(NS)]
    cases = []
    for i in range(NS):
        x = (torch.randn(T, d, generator=g) * (1 + i)).to(dev())
        dy = bf(torch.randn(T, d, generator=g)).to(dev())
        w = (1 + 0.1 * torch.randn(d, generator=g)).to(dev())
        rstd = torch.rsqrt((x * x).mean(-1) + 1e-6)
        want = (dy.float() * x * rstd[:, None]).sum(0)
        cases.append((x, dy, w, rstd, want, torch.zeros(d, device=dev()), torch.empty(T, d, device=dev())))
    torch.cuda.synchronize()
    for rnd in range(3):
        for st, (x, dy, w, rstd, want, dw, dres) in zip(streams, cases):
            with torch.cuda.stream(st):
                ops.rmsnorm_bwd(dy, x, w, rstd, dres, None, dw, accumulate=False)
    torch.cuda.synchronize()
    for x, dy, w, rstd, want, dw, dres in cases:
        assert rel(dw, 3 * want) < 1e-4


# ----------------------------------------------------------------------------------------------- rope
# il: adjacent rotation pairs (2i, 2i+1) instead of the split-half pairs
@pytest.mark.parametrize("il", [False, True])
@pytest.mark.parametrize("D,norm", [(128, True), (64, False), (128, False), (64, True)])
def test_qknorm_rope_fwd_bwd(ops, D, norm, il):
    rope = R.apply_rope_interleaved if il else R.apply_rope
    B, S, HQ, HK = 2, 37, 3, 2
    T = B * S
    g = torch.Generator().manual_seed(D + norm)
    NQ = (HQ + 2 * HK) * D
    qkv = bf(torch.randn(T, NQ, generator=g))
    wq = 1 + 0.2 * torch.randn(D, generator=g)
    wk = 1 + 0.2 * torch.randn(D, generator=g)
    theta = 1e6
    pos = torch.arange(S)[None].expand(B, S)
    cos, sin = R.rope_cos_sin(pos, D, theta)

    def ref(qkv_f, wq_, wk_):
        q = qkv_f[:, :HQ * D].view(B, S, HQ, D)
        k = qkv_f[:, HQ * D:(HQ + HK) * D].view(B, S, HK, D)
        if norm:
            q = R.rmsnorm(bf(q), wq_, 1e-6)
            k = R.rmsnorm(bf(k), wk_, 1e-6)
        q = rope(q.transpose(1, 2).float(), cos, sin).transpose(1, 2)
        k = rope(k.transpose(1, 2).float(), cos, sin).transpose(1, 2)
        return q.reshape(T, HQ * D), k.reshape(T, HK * D)

    q_ref, k_ref = ref(qkv.float(), wq, wk)
    inv = R.rope_inv_freq(D, theta).to(dev())
    cs = ops.rope_table(inv, 64)
    assert rel(cs[:S, :, 0], cos[0, :, :D // 2]) < 1e-6 and rel(cs[:S, :, 1], sin[0, :, :D // 2]) < 1e-6
    QK = (HQ + HK) * D
    out = torch.empty(T, QK, dtype=torch.bfloat16, device=dev())
    qd = qkv.to(dev())
    wqd, wkd = (wq.to(dev()), wk.to(dev())) if norm else (None, None)
    ops.qknorm_rope_fwd(qd, NQ, out.data_ptr(), out.data_ptr() + 2 * HQ * D, QK, QK, wqd, wkd, cs, None, S, T, HQ,
                        HK, D, 1e-6, interleaved=il)
    assert rel(out[:, :HQ * D].float(), q_ref) < 1e-2
    assert rel(out[:, HQ * D:].float(), k_ref) < 1e-2
    # explicit positions give the same result
    posd = pos.reshape(-1).to(torch.int32).to(dev())
    out2 = torch.empty_like(out)
    ops.qknorm_rope_fwd(qd, NQ, out2.data_ptr(), out2.data_ptr() + 2 * HQ * D, QK, QK, wqd, wkd, cs, posd, S, T, HQ,
                        HK, D, 1e-6, interleaved=il)
    assert torch.equal(out, out2)

    # backward against autograd of the fp32 formula (rounding of the normalised value treated as identity)
    xf = qkv.float().clone().requires_grad_(True)
    wq_r = wq.clone().requires_grad_(True)
    wk_r = wk.clone().requires_grad_(True)
    q = xf[:, :HQ * D].view(B, S, HQ, D)
    k = xf[:, HQ * D:(HQ + HK) * D].view(B, S, HK, D)
    if norm:
        q = R.rmsnorm(q, wq_r, 1e-6)
        k = R.rmsnorm(k, wk_r, 1e-6)
    q = rope(q.transpose(1, 2), cos, sin).transpose(1, 2).reshape(T, HQ * D)
    k = rope(k.transpose(1, 2), cos, sin).transpose(1, 2).reshape(T, HK * D)
    dqk = bf(torch.randn(T, NQ, generator=g))
    (q * dqk[:, :HQ * D].float()).sum().backward(retain_graph=True)
    (k * dqk[:, HQ * D:QK].float()).sum().backward()
    buf = dqk.clone().to(dev())
    dwq = torch.zeros(D, dtype=torch.float32, device=dev())
    dwk = torch.zeros(D, dtype=torch.float32, device=dev())
    p = buf.data_ptr()
    ops.qknorm_rope_bwd(p, NQ, p, p + 2 * HQ * D, NQ, NQ, qd, NQ, wqd, wkd, dwq if norm else None,
                        dwk if norm else None, cs, None, S, T, HQ, HK, D, 1e-6, interleaved=il)
    assert rel(buf[:, :QK].float(), xf.grad[:, :QK]) < 1.5e-2
    assert torch.equal(buf[:, QK:].cpu(), dqk[:, QK:]), "v section must be untouched"
    if norm:
        assert rel(dwq, wq_r.grad) < 1e-3 and rel(dwk, wk_r.grad) < 1e-3


# ----------------------------------------------------------------------------------------------- attention
def _attn_ref(q, k, v, scale):
    # q,k,v [B,S,H,D] fp32 leaf tensors
    o = R.causal_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), scale)
    return o  # [B,S,H*D]


@pytest.mark.parametrize("B,S,H,KVH,D", [(1, 128, 1, 1, 128), (2, 200, 2, 2, 128), (1, 384, 4, 2, 128),
                                         (2, 64, 2, 1, 128), (1, 777, 1, 1, 128), (2, 200, 4, 4, 64),
                                         (1, 384, 2, 2, 64), (1, 64, 6, 6, 64), (1, 1000, 2, 2, 64)])
def test_attention_fwd_bwd(ops, B, S, H, KVH, D):
    g = torch.Generator().manual_seed(B * 1000 + S + H)
    T = B * S
    q = bf(torch.randn(B, S, H, D, generator=g))
    k = bf(torch.randn(B, S, KVH, D, generator=g))
    v = bf(torch.randn(B, S, KVH, D, generator=g))
    do = bf(torch.randn(B, S, H * D, generator=g))
    scale = D ** -0.5
    qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    o_ref = _attn_ref(qr, kr, vr, scale)
    o_ref.backward(do.float())
    # fused layout as the model uses it: q|k in one buffer, v inside the projection output
    QK = (H + KVH) * D
    NQ = (H + 2 * KVH) * D
    qk = torch.cat([q.reshape(T, H * D), k.reshape(T, KVH * D)], dim=1).contiguous().to(dev())
    qkv = torch.zeros(T, NQ, dtype=torch.bfloat16)
    qkv[:, QK:] = v.reshape(T, KVH * D)
    qkv = qkv.to(dev())
    out = torch.zeros(T, H * D, dtype=torch.bfloat16, device=dev())
    lse = torch.zeros(B, H, S, dtype=torch.float32, device=dev())
    qp, kp, vp = qk.data_ptr(), qk.data_ptr() + 2 * H * D, qkv.data_ptr() + 2 * QK
    ops.attn_fwd(qp, kp, vp, out.data_ptr(), lse, QK, QK, NQ, H * D, B, S, H, KVH, D, scale)
    assert rel(out.float().view(B, S, H * D), o_ref.detach()) < 1e-2
    # LSE check
    rep = H // KVH
    kk = k.float().repeat_interleave(rep, dim=2)
    sc = torch.einsum("bqhd,bkhd->bhqk", q.float(), kk) * scale
    sc = sc.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool), 1), float("-inf"))
    assert rel(lse, torch.logsumexp(sc, dim=-1)) < 1e-4

    dqkv = torch.zeros(T, NQ, dtype=torch.bfloat16, device=dev())
    delta = torch.empty(B, H, S, dtype=torch.float32, device=dev())
    dod = do.reshape(T, H * D).to(dev())
    dp = dqkv.data_ptr()
    ops.attn_bwd(qp, kp, vp, out.data_ptr(), dod.data_ptr(), lse, delta, dp, dp + 2 * H * D, dp + 2 * QK,
                 QK, QK, NQ, H * D, H * D, NQ, NQ, NQ, B, S, H, KVH, D, scale)
    assert rel(dqkv[:, :H * D].float(), qr.grad.reshape(T, H * D)) < 2e-2
    assert rel(dqkv[:, H * D:QK].float(), kr.grad.reshape(T, KVH * D)) < 2e-2
    assert rel(dqkv[:, QK:].float(), vr.grad.reshape(T, KVH * D)) < 2e-2


def test_attention_online_softmax_rescale(ops):
    """One key row spiked against one query row so that the running max jumps at a late tile."""
    B, S, H, D = 1, 256, 1, 128
    g = torch.Generator().manual_seed(5)
    q = torch.randn(B, S, H, D, generator=g) * 0.5
    k = torch.randn(B, S, H, D, generator=g) * 0.5
    v = torch.randn(B, S, H, D, generator=g)
    k[0, 190] = q[0, 250] * 6.0     # huge score for query 250 (and large for others) at key 190 (third tile)
    k[0, 3] = q[0, 100] * 5.0
    q, k, v = bf(q), bf(k), bf(v)
    scale = D ** -0.5
    o_ref = _attn_ref(q.float(), k.float(), v.float(), scale)
    T = B * S
    qd, kd, vd = (t.reshape(T, D).contiguous().to(dev()) for t in (q, k, v))
    out = torch.zeros(T, D, dtype=torch.bfloat16, device=dev())
    lse = torch.zeros(B, H, S, dtype=torch.float32, device=dev())
    ops.attn_fwd(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), out.data_ptr(), lse, D, D, D, D, B, S, H, H, D, scale)
    assert rel(out.float().view(B, S, D), o_ref) < 1e-2


def test_gemm_long_reduction_512_columns(ops):
    """dX[T, 512] = dY[T, K] @ W[K, 512] with a very long K (the lm_head input gradient's shape class): sampled rows
    against fp32 arithmetic, ragged last row tile included."""
    M, N, K = 16384 + 72, 512, 32768 + 64
    g = torch.Generator().manual_seed(5)
    a = bf(torch.randn(M, K, generator=g) * 0.05).to(dev())
    b = bf(torch.randn(K, N, generator=g) * 0.05).to(dev())
    c = torch.full((M, N), 3.0, dtype=torch.bfloat16, device=dev())
    ops.gemm(a, b, c, M, N, K, K, N, N, trans_a=False, trans_b=True)
    rows = torch.cat([torch.arange(0, 40), torch.arange(8100, 8140), torch.arange(M - 90, M)])
    ref = a[rows.to(dev())].float().cpu() @ b.float().cpu()
    assert rel(c[rows.to(dev())].float(), ref) < 8e-3
    assert torch.isfinite(c.float()).all()


# ----------------------------------------------------------------------------------------------- elementwise
def test_swiglu(ops):
    T, I = 37, 256
    g = torch.Generator().manual_seed(3)
    gu = bf(torch.randn(T, 2 * I, generator=g) * 2)
    da = bf(torch.randn(T, I, generator=g))
    gr = gu.float().clone().requires_grad_(True)
    act_ref = torch.nn.functional.silu(gr[:, :I]) * gr[:, I:]
    act_ref.backward(da.float())
    act = torch.empty(T, I, dtype=torch.bfloat16, device=dev())
    ops.swiglu_fwd(gu.to(dev()), act)
    emul = bf(bf(torch.nn.functional.silu(gu[:, :I].float())).float() * gu[:, I:].float())
    assert rel(act.float(), emul.float()) < 8e-3
    dgu = torch.empty(T, 2 * I, dtype=torch.bfloat16, device=dev())
    ops.swiglu_bwd(gu.to(dev()), da.to(dev()), dgu)
    assert rel(dgu.float(), gr.grad) < 1e-2


@pytest.mark.parametrize("T,I,K", [(4096, 2048, 512), (2100, 256, 128), (65536, 2048, 512)])
def test_gate_up_product_with_swiglu_epilogue(ops, T, I, K):
    """The fused launch against the two separate ones (mgen_gemm_bf16, mgen_swiglu_fwd): same [gate | up] bits, same
    activation bits; a ragged last row tile included."""
    g = torch.Generator().manual_seed(11)
    x = bf(torch.randn(T, K, generator=g)).to(dev())
    w = bf(torch.randn(2 * I, K, generator=g) * K ** -0.5).to(dev())
    gu0 = torch.empty(T, 2 * I, dtype=torch.bfloat16, device=dev())
    act0 = torch.empty(T, I, dtype=torch.bfloat16, device=dev())
    ops.linear_fwd(x, w, gu0)
    ops.swiglu_fwd(gu0, act0)
    gu1 = torch.full((T, 2 * I), 7.0, dtype=torch.bfloat16, device=dev())
    act1 = torch.full((T, I), 7.0, dtype=torch.bfloat16, device=dev())
    assert ops.linear_swiglu_fwd(x, w, gu1, act1)
    assert torch.equal(gu1, gu0)
    assert torch.equal(act1, act0)
    if T <= 4096:   # and the product itself against fp32 arithmetic
        ref = x.float().cpu() @ w.float().cpu().T
        assert rel(gu1.float(), ref) < 8e-3


def test_embedding(ops):
    V, d, T = 1000, 512, 300
    g = torch.Generator().manual_seed(4)
    table = torch.randn(V, d, generator=g)
    ids = torch.randint(0, V, (T,), generator=g)
    ids[:50] = 7  # heavy collisions
    out = torch.empty(T, d, dtype=torch.float32, device=dev())
    err = torch.zeros(1, dtype=torch.int32, device=dev())
    ops.embedding_fwd(ids.to(dev()), table.to(dev()), out, err)
    assert torch.equal(out.cpu(), table[ids]) and int(err) == 0
    dout = torch.randn(T, d, generator=g)
    dt = torch.zeros(V, d, dtype=torch.float32, device=dev())
    ops.embedding_bwd(ids.to(dev()), dout.to(dev()), dt)
    ref = torch.zeros(V, d).index_add_(0, ids, dout)
    assert rel(dt, ref) < 1e-5
    bad = ids.clone()
    bad[3] = V + 5
    ops.embedding_fwd(bad.to(dev()), table.to(dev()), out, err)
    assert int(err) == 1


# V >= 30000: row-in-registers kernel; (151941, 130): more rows than CUs, i.e. persistent workgroups that stage half of
# the next row through LDS; (151936, 130): the same without a ragged last vector
@pytest.mark.parametrize("V,S", [(1024, 17), (1031, 17), (30000, 17), (65531, 17), (151941, 17), (151941, 130),
                                 (151936, 130)])
def test_cross_entropy(ops, V, S):
    B = 3
    T = B * S
    g = torch.Generator().manual_seed(V)
    ld = (V + 7) // 8 * 8
    logits = bf(torch.randn(T, V, generator=g) * 3)
    labels = torch.randint(0, V, (B, S), generator=g)
    labels[0, 3:6] = -100
    labels[2, :] = -100
    lr = logits.float().clone().requires_grad_(True)
    loss_ref = R.causal_lm_loss(lr.view(B, S, V), labels)
    loss_ref.backward()
    buf = torch.full((T, ld), 5.0, dtype=torch.bfloat16)
    buf[:, :V] = logits
    buf = buf.to(dev())
    lab = labels.to(dev())
    inv = torch.empty(1, dtype=torch.float32, device=dev())
    cnt = torch.empty(1, dtype=torch.int32, device=dev())
    ops.ce_count(lab, inv, cnt)
    n_valid = int((torch.nn.functional.pad(labels, (0, 1), value=-100)[:, 1:] != -100).sum())
    assert int(cnt) == n_valid
    row_loss = torch.empty(T, dtype=torch.float32, device=dev())
    ops.ce_fwd_bwd(buf, lab, 0, T, V, ld, S, inv, 1.0, row_loss)
    loss = torch.empty((), dtype=torch.float32, device=dev())
    ops.ce_finalize(row_loss, inv, loss)
    assert abs(float(loss) - float(loss_ref)) < 1e-5 * abs(float(loss_ref))
    assert rel(buf[:, :V].float(), lr.grad) < 1e-2
    assert bool((buf[:, V:] == 0).all())
    # chunked call (tok0 offset) gives the same rows
    buf2 = torch.full((T, ld), 5.0, dtype=torch.bfloat16)
    buf2[:, :V] = logits
    buf2 = buf2.to(dev())
    h = 20 if S < 100 else 300
    ops.ce_fwd_bwd(buf2[:h], lab, 0, h, V, ld, S, inv, 1.0, row_loss)
    ops.ce_fwd_bwd(buf2[h:], lab, h, T - h, V, ld, S, inv, 1.0, row_loss)
    assert torch.equal(buf, buf2)


def test_adamw_and_clip(ops):
    n = 4096 * 3 + 64
    g = torch.Generator().manual_seed(9)
    p = torch.randn(n, generator=g)
    grad = torch.randn(n, generator=g) * 0.3
    pr, m, v = p.clone(), torch.zeros(n), torch.zeros(n)
    pd, gd = p.clone().to(dev()), grad.to(dev())
    md, vd = torch.zeros(n, device=dev()), torch.zeros(n, device=dev())
    sh = torch.zeros(n, dtype=torch.bfloat16, device=dev())
    ws = torch.empty(1024, dtype=torch.float64, device=dev())
    nc = torch.empty(2, dtype=torch.float32, device=dev())
    for step in range(1, 4):
        total, coef = R.clip_coef([grad], 1.0)
        R.adamw_step(pr, grad * coef, m, v, step, 3e-4)
        ops.grad_norm_clip(gd, 1.0, ws, nc)
        assert abs(float(nc[0]) - float(total)) < 1e-5 * float(total)
        assert abs(float(nc[1]) - float(coef)) < 1e-5 * float(coef)
        ops.adamw_step(pd, gd, md, vd, sh, nc, 3e-4, 0.9, 0.95, 1e-5, 0.1, step)
        assert rel(pd, pr) < 1e-6
    assert rel(md, m) < 1e-5 and rel(vd, v) < 1e-5
    assert torch.equal(sh.cpu(), bf(pd.cpu()))


def test_cast_scale_axpy(ops):
    n = 1000 * 8
    x = torch.randn(n)
    xd = x.to(dev())
    o = torch.empty(n, dtype=torch.bfloat16, device=dev())
    ops.cast(xd, o)
    assert torch.equal(o.cpu(), bf(x))
    y = torch.ones(n, device=dev())
    a = torch.tensor(0.5, device=dev())
    ops.axpy_(y, xd, 2.0, a)
    assert rel(y, 1 + x) < 1e-6
    ops.scale_bf16_(o, 1.0, a)
    assert rel(o.float(), bf(x).float() * 0.5) < 1e-2


@pytest.mark.parametrize("M,N,K,K2,tb", [(300, 200, 128, 64, False), (16500, 2040, 512, 64, False),
                                         (33000, 1096, 64, 64, True), (1000, 512, 1024, 128, True)])
def test_gemm_two_operand_pairs(ops, M, N, K, K2, tb):
    """C = [A | A2] [B | B2]^T in one launch (a projection plus its low-rank adapter product); 128x128, 128x256 and
    256x256 tiles, ragged edges, both layouts of the B side, residual epilogue."""
    g = torch.Generator().manual_seed(M + N + K)
    A = bf(torch.randn(M, K, generator=g) * 0.5)
    A2 = bf(torch.randn(M, K2, generator=g) * 0.5)
    B = bf(torch.randn(N, K, generator=g) * 0.5)
    B2 = bf(torch.randn(N, K2, generator=g) * 0.5)
    ref = A.float() @ B.float().t() + A2.float() @ B2.float().t()
    ldn = (N + 7) // 8 * 8
    def side(mat):   # stored [N][k] or, transposed, [k][ldn]
        if not tb:
            return mat.contiguous().to(dev())
        out = torch.zeros(mat.shape[1], ldn, dtype=torch.bfloat16)
        out[:, :N] = mat.t()
        return out.to(dev())
    b_st, b2_st = side(B), side(B2)
    ldc = ldn + 8
    c = torch.full((M, ldc), 3.0, dtype=torch.bfloat16, device=dev())
    ops.gemm_seg(A.to(dev()), b_st, A2.to(dev()), b2_st, c, M, N, K, K2, K, b_st.stride(0), K2, b2_st.stride(0), ldc,
                 tb, ops.OUT_BF16)
    assert rel(c[:, :N].float(), ref) < 1e-2
    assert bool((c[:, N:] == 3.0).all()), "wrote outside the N columns"
    res = torch.randn(M, N, generator=g)
    out = torch.empty(M, N, dtype=torch.float32, device=dev())
    ops.gemm_seg(A.to(dev()), b_st, A2.to(dev()), b2_st, out, M, N, K, K2, K, b_st.stride(0), K2, b2_st.stride(0), N,
                 tb, ops.OUT_F32_RESID_BF16ROUND, res.to(dev()), N)
    assert rel(out, res + bf(ref).float()) < 1e-2
    # the plain entry point is untouched by the second build of the kernels
    c2 = torch.empty(M, ldc, dtype=torch.float32, device=dev())
    ops.gemm(A.to(dev()), b_st, c2, M, N, K, K, b_st.stride(0), ldc, False, tb, ops.OUT_F32)
    assert rel(c2[:, :N], A.float() @ B.float().t()) < 2e-5
    from mgen._lib import MgenError
    with pytest.raises(MgenError):   # reduction lengths must be multiples of 64
        ops.gemm_seg(A.to(dev()), b_st, A2.to(dev()), b2_st, c, M, N, K, 32, K, b_st.stride(0), K2, b2_st.stride(0),
                     ldc, tb, ops.OUT_BF16)


# ----------------------------------------------------------------------------------------------- lm_head + CE in one call
def test_lmhead_ce_entry_point_equals_the_written_out_protocol(ops):
    """mgen_lmhead_ce_fwd_bwd (chunk loop inside the library) against the same protocol issued call by call: logits
    GEMM, cross entropy in place, input gradient, weight gradient (store, then accumulate); ragged last chunk, ignored
    labels, V not a multiple of 8; and against fp32 autograd."""
    from mgen._lib import MgenError
    B, S, V, d, chunk = 3, 50, 1031, 256, 64
    T = B * S
    g = torch.Generator().manual_seed(12)
    x = bf(torch.randn(T, d, generator=g)).to(dev())
    w = bf(torch.randn(V, d, generator=g) * 0.05).to(dev())
    labels = torch.randint(0, V, (B, S), generator=g)
    labels[0, :7] = -100
    labels = labels.to(dev())
    ws = ops.lmhead_ce_workspace(T, V, chunk, dev())
    loss = torch.empty((), device=dev())
    dx = torch.empty(T, d, dtype=torch.bfloat16, device=dev())
    dw = torch.full((V, d), 7.0, device=dev())          # overwritten, not added to
    ops.lmhead_ce_fwd_bwd(x, w, labels, V, chunk, ws, loss, dx, dw)
    # fp32 autograd on the same bf16 inputs (logits rounded to bf16 as the kernel stores them)
    xr, wr = x.float().requires_grad_(True), w.float().requires_grad_(True)
    logits = (xr @ wr.t()).to(torch.bfloat16).float() + 0 * (xr @ wr.t())
    lab = torch.nn.functional.pad(labels, (0, 1), value=-100)[:, 1:].reshape(-1)
    ref = torch.nn.functional.cross_entropy((xr @ wr.t()), lab, ignore_index=-100)
    ref.backward()
    assert abs(float(loss) - float(ref)) < 2e-3 * float(ref)
    assert rel(dx.float(), xr.grad) < 2e-2 and rel(dw, wr.grad) < 2e-2
    # forward only
    loss2 = torch.empty((), device=dev())
    ops.lmhead_ce_fwd_bwd(x, w, labels, V, chunk, ws, loss2, None, None)
    assert torch.equal(loss, loss2)
    # frozen head: dx without dW
    dx3 = torch.empty_like(dx)
    ops.lmhead_ce_fwd_bwd(x, w, labels, V, chunk, ws, loss2, dx3, None)
    assert torch.equal(dx, dx3)
    with pytest.raises(MgenError):
        ops.lmhead_ce_fwd_bwd(x, w, labels, V, chunk, ws[:1024], loss2, None, None)
