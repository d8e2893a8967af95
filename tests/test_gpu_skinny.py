"""The decode-step projection kernel (csrc/gemm_skinny.hip) against fp32 arithmetic on the same bf16 inputs and against
the tile-loop GEMM + the separate SwiGLU kernel it replaces; and the fused q/k-norm + rotation + cache append against
the two launches it replaces (bit for bit)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from mgen import ops as o
    return o


@pytest.mark.parametrize("M", [1, 37, 64, 128, 256, 300])
@pytest.mark.parametrize("N", [512, 3072])
def test_bf16_output(ops, M, N):
    K = 512
    g = torch.Generator().manual_seed(M * 7 + N)
    x = (torch.randn(M, K, generator=g)).to(torch.bfloat16).to(dev())
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(dev())
    assert ops.skinny_fits(M, N, K, K, K, N, False)
    out = torch.full((M + 1, N), 7.0, dtype=torch.bfloat16, device=dev())
    ops.linear_skinny(x, w, out[:M])
    ref = x.float() @ w.float().t()
    assert float((out[:M].float() - ref).abs().max() / ref.abs().max()) < 1e-2
    assert bool((out[M] == 7.0).all()), "wrote past the M rows"
    old = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
    ops.linear_fwd(x, w, old)
    assert torch.equal(old, out[:M]), "differs from the tile-loop kernel (same MFMA order over k expected)"


@pytest.mark.parametrize("M,K", [(256, 1024), (128, 2048), (50, 1536)])
def test_slab_output(ops, M, K):
    N = 512
    g = torch.Generator().manual_seed(M + K)
    x = (torch.randn(M, K, generator=g)).to(torch.bfloat16).to(dev())
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(dev())
    assert ops.skinny_fits(M, N, K, K, K, N, True)
    ns = K // 512
    slabs = torch.full((ns + 1, M, N), 7.0, dtype=torch.float32, device=dev())
    ops.linear_skinny(x, w, slabs, slabs=True)
    assert bool((slabs[ns] == 7.0).all())
    for z in range(ns):
        ref = x[:, z * 512:(z + 1) * 512].float() @ w[:, z * 512:(z + 1) * 512].float().t()
        assert float((slabs[z] - ref).abs().max() / ref.abs().max()) < 1e-5


@pytest.mark.parametrize("M", [3, 128, 256])
def test_gated_operand(ops, M):
    """A operand = silu(gate) * up formed from gu while it is loaded: same bits as mgen_swiglu_fwd + the plain kernel."""
    I, N = 2048, 512
    g = torch.Generator().manual_seed(M)
    gu = (torch.randn(M, 2 * I, generator=g) * 2).to(torch.bfloat16).to(dev())
    w = (torch.randn(N, I, generator=g) * 0.05).to(torch.bfloat16).to(dev())
    act = torch.empty(M, I, dtype=torch.bfloat16, device=dev())
    ops.swiglu_fwd(gu, act)
    a = torch.empty(I // 512, M, N, dtype=torch.float32, device=dev())
    b = torch.empty(I // 512, M, N, dtype=torch.float32, device=dev())
    ops.linear_skinny(act, w, a, slabs=True)
    ops.linear_skinny(gu, w, b, slabs=True, a_swiglu=True)
    assert torch.equal(a, b)


@pytest.mark.parametrize("M,N,K", [(128, 4608, 1536), (256, 1024, 2048), (37, 12288, 1536)])
def test_bf16_output_over_several_chunks(ops, M, N, K):
    """A reduction longer than 512 with a bf16 output: the workgroup walks its chunks in ascending order (the 650M
    model's QKV and gate/up projections); same bits as the tile-loop kernel."""
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g)).to(torch.bfloat16).to(dev())
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(dev())
    assert ops.skinny_fits(M, N, K, K, K, N, 0)
    out = torch.full((M + 1, N), 7.0, dtype=torch.bfloat16, device=dev())
    ops.linear_skinny(x, w, out[:M])
    ref = x.float() @ w.float().t()
    assert float((out[:M].float() - ref).abs().max() / ref.abs().max()) < 1e-2
    assert bool((out[M] == 7.0).all()), "wrote past the M rows"
    old = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
    ops.linear_fwd(x, w, old)
    assert torch.equal(old, out[:M])


@pytest.mark.parametrize("M,N,K,gated", [(128, 1536, 6144, True), (128, 1536, 1536, False), (256, 512, 2048, True)])
def test_fewer_slabs_than_chunks(ops, M, N, K, gated):
    """Slabs that cover several 512-index chunks each (the count mgen_gemm_skinny_slabs picks, and every other
    divisor): each slab against fp32 arithmetic on its reduction range; the gated operand as with one chunk per slab."""
    g = torch.Generator().manual_seed(M + N + K)
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(dev())
    if gated:
        gu = (torch.randn(M, 2 * K, generator=g)).to(torch.bfloat16).to(dev())
        x = torch.empty(M, K, dtype=torch.bfloat16, device=dev())
        ops.swiglu_fwd(gu, x)
    else:
        x = (torch.randn(M, K, generator=g)).to(torch.bfloat16).to(dev())
    chunks = K // 512
    picked = ops.skinny_slabs(M, N, K)
    assert chunks % picked == 0 and (picked == chunks or ((M + 31) // 32) * (N // 64) * picked >= 256)
    for ns in sorted({picked} | {s for s in range(1, chunks + 1) if chunks % s == 0}):
        assert ops.skinny_fits(M, N, K, 2 * K if gated else K, K, N, ns)
        slabs = torch.full((ns + 1, M, N), 7.0, dtype=torch.float32, device=dev())
        ops.linear_skinny(gu if gated else x, w, slabs, slabs=ns, a_swiglu=gated)
        assert bool((slabs[ns] == 7.0).all())
        per = K // ns
        for z in range(ns):
            ref = x[:, z * per:(z + 1) * per].float() @ w[:, z * per:(z + 1) * per].float().t()
            assert float((slabs[z] - ref).abs().max() / ref.abs().max()) < 1e-5, (ns, z)


def test_shapes_not_taken(ops):
    from mgen._lib import MgenError
    assert not ops.skinny_fits(256, 3072, 1536, 1536, 1536, 3072, 2)       # 2 slabs do not divide 3 chunks
    assert not ops.skinny_fits(256, 1000, 512, 512, 512, 1000, False)      # N % 64
    assert not ops.skinny_fits(4096, 512, 512, 512, 512, 512, False)       # too many rows
    x = torch.zeros(8, 768, dtype=torch.bfloat16, device=dev())
    w = torch.zeros(64, 768, dtype=torch.bfloat16, device=dev())
    with pytest.raises(MgenError):
        ops.linear_skinny(x, w, torch.zeros(8, 64, dtype=torch.bfloat16, device=dev()))


@pytest.mark.parametrize("D,norm,il", [(128, True, False), (64, False, False), (128, True, True), (64, True, True)])
def test_decode_qkv_prep_equals_rope_then_append(ops, D, norm, il):
    B, H, KVH, max_len = 37, 4, 2, 24
    NQ, QK = (H + 2 * KVH) * D, (H + KVH) * D
    g = torch.Generator().manual_seed(D + norm + il)
    qkv = torch.randn(B, NQ, generator=g).to(torch.bfloat16).to(dev())
    wq = (1 + 0.2 * torch.randn(D, generator=g)).to(dev()) if norm else None
    wk = (1 + 0.2 * torch.randn(D, generator=g)).to(dev()) if norm else None
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, dtype=torch.float) / D))
    cs = ops.rope_table(inv.to(dev()), max_len)
    pos = torch.randint(0, max_len, (B,), generator=g).to(torch.int32).to(dev())
    qk = torch.zeros(B, QK, dtype=torch.bfloat16, device=dev())
    kc0 = torch.randn(B, KVH, max_len, D, generator=g).to(torch.bfloat16).to(dev())
    vc0 = torch.randn(B, KVH, max_len, D, generator=g).to(torch.bfloat16).to(dev())
    kc1, vc1 = kc0.clone(), vc0.clone()
    ops.qknorm_rope_fwd(qkv, NQ, qk.data_ptr(), qk.data_ptr() + 2 * H * D, QK, QK, wq, wk, cs, pos, 1, B, H, KVH, D, 1e-6,
                        interleaved=il)
    ops.kv_append(qk.data_ptr() + 2 * H * D, qkv.data_ptr() + 2 * QK, QK, NQ, kc0, vc0, pos, B, 1, KVH, D, max_len)
    q1 = torch.zeros(B, QK, dtype=torch.bfloat16, device=dev())
    ops.decode_qkv_prep(qkv, NQ, q1.data_ptr(), QK, kc1, vc1, wq, wk, cs, pos, B, H, KVH, D, max_len, 1e-6, interleaved=il)
    assert torch.equal(q1[:, :H * D], qk[:, :H * D])
    assert torch.equal(kc1, kc0) and torch.equal(vc1, vc0)


@pytest.mark.parametrize("D,H,KVH,norm,il,nsplit", [(128, 8, 8, True, False, 1), (128, 4, 2, True, False, 3),
                                                   (64, 6, 2, False, False, 1), (128, 4, 4, True, True, 2),
                                                   (64, 4, 4, True, True, 1)])
def test_attention_from_raw_projection_rows(ops, D, H, KVH, norm, il, nsplit):
    """mgen_attn_decode_prep == mgen_decode_qkv_prep + mgen_attn_decode: outputs and caches bit for bit, streams at
    different positions (incl. a stream whose cache is empty before this token), GQA, key splits."""
    B, max_len = 9, 300
    NQ, QK = (H + 2 * KVH) * D, (H + KVH) * D
    g = torch.Generator().manual_seed(D + H + nsplit)
    qkv = torch.randn(B, NQ, generator=g).to(torch.bfloat16).to(dev())
    wq = (1 + 0.2 * torch.randn(D, generator=g)).to(dev()) if norm else None
    wk = (1 + 0.2 * torch.randn(D, generator=g)).to(dev()) if norm else None
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, dtype=torch.float) / D))
    cs = ops.rope_table(inv.to(dev()), max_len)
    pos = torch.tensor([0, 1, 5, 63, 64, 129, 255, 298, 299], dtype=torch.int32).to(dev())
    kv_len = pos + 1
    kc0 = torch.randn(B, KVH, max_len, D, generator=g).to(torch.bfloat16).to(dev())
    vc0 = torch.randn(B, KVH, max_len, D, generator=g).to(torch.bfloat16).to(dev())
    kc1, vc1 = kc0.clone(), vc0.clone()
    ws = torch.empty(max(ops.attn_decode_workspace(B, H, D, nsplit), 1), dtype=torch.float32, device=dev())
    q0 = torch.zeros(B, QK, dtype=torch.bfloat16, device=dev())
    o0 = torch.zeros(B, H * D, dtype=torch.bfloat16, device=dev())
    o1 = torch.zeros(B, H * D, dtype=torch.bfloat16, device=dev())
    ops.decode_qkv_prep(qkv, NQ, q0.data_ptr(), QK, kc0, vc0, wq, wk, cs, pos, B, H, KVH, D, max_len, 1e-6, interleaved=il)
    ops.attn_decode(q0.data_ptr(), QK, kc0, vc0, kv_len, o0, H * D, ws, B, H, KVH, D, max_len, nsplit, D ** -0.5)
    ops.attn_decode_prep(qkv, NQ, wq, wk, cs, pos, 1e-6, il, kc1, vc1, kv_len, o1, H * D, ws, B, H, KVH, D, max_len,
                         nsplit, D ** -0.5)
    assert torch.equal(kc1, kc0) and torch.equal(vc1, vc0)
    # the new key enters the softmax through the merge step instead of the key loop: same terms, another order
    assert float((o1.float() - o0.float()).abs().max()) <= 2e-2 * float(o0.float().abs().max())
    assert float((o1.float() - o0.float()).abs().mean()) < 1e-3 * float(o0.float().abs().mean()) + 1e-6
