"""The K = 512 GEMM kernel that keeps the token panel in registers (csrc/gemm_rs.hip), through the C-ABI:
against torch fp32 arithmetic on the same bf16 inputs, bit for bit against the tile-loop kernel it replaces
(MGEN_GEMM_RS=0 is read on every call), with padded leading dimensions, in the SwiGLU form, behind the transposed
weight copies of the input-gradient products, and inside one training step of the default decoder."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from mgen import ops as o
    return o


@pytest.fixture(autouse=True)
def _rs_on():
    os.environ.pop("MGEN_GEMM_RS", None)
    yield
    os.environ.pop("MGEN_GEMM_RS", None)


def _both(run):
    """Runs `run(out_tag)` with the register-panel kernel and with the tile-loop kernel."""
    os.environ["MGEN_GEMM_RS"] = "1"
    a = run()
    os.environ["MGEN_GEMM_RS"] = "0"
    b = run()
    os.environ.pop("MGEN_GEMM_RS", None)
    torch.cuda.synchronize()
    return a, b


@pytest.mark.parametrize("M,N,pad", [(16384, 1024, 0), (16384, 1088, 8), (32768, 3072, 0), (16640, 1536, 64)])
def test_plain_against_fp32_and_tile_loop(ops, M, N, pad):
    K = 512
    g = torch.Generator().manual_seed(M + N)
    a = (torch.randn(M, K + pad, generator=g) * 0.5).to(torch.bfloat16).to(dev())
    b = (torch.randn(N, K + pad, generator=g) * 0.5).to(torch.bfloat16).to(dev())
    ldc = N + pad

    def run():
        c = torch.full((M, ldc), 7.0, dtype=torch.bfloat16, device=dev())
        ops.gemm(a, b, c, M, N, K, a.stride(0), b.stride(0), ldc, False, False, ops.OUT_BF16)
        return c

    new, old = _both(run)
    assert torch.equal(new, old), "register-panel kernel differs from the tile-loop kernel"
    if pad:
        assert bool((new[:, N:] == 7.0).all()), "wrote outside the N columns"
    rows = torch.arange(0, M, 97, device=dev())   # a strided sample of rows against fp32 on the same inputs
    ref = a[rows, :K].float() @ b[:, :K].float().t()
    err = float((new[rows, :N].float() - ref).abs().max() / ref.abs().max())
    assert err < 1e-2, err


def test_column_runs_when_rows_do_not_fill_the_chip(ops):
    """Few row panels: the column range is cut into runs (one workgroup per panel and run); an odd number of column
    pairs per run and the rotated block order are covered by 1088 columns over 4 runs."""
    M, N, K = 16384, 2176, 512
    g = torch.Generator().manual_seed(5)
    a = (torch.randn(M, K, generator=g)).to(torch.bfloat16).to(dev())
    b = (torch.randn(N, K, generator=g) * 0.1).to(torch.bfloat16).to(dev())

    def run():
        c = torch.zeros(M, N, dtype=torch.bfloat16, device=dev())
        ops.gemm(a, b, c, M, N, K, K, K, N, False, False, ops.OUT_BF16)
        return c

    new, old = _both(run)
    assert torch.equal(new, old)


def test_swiglu_form(ops):
    M, I, K = 16384, 1024, 512
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(dev())
    w = (torch.randn(2 * I, K, generator=g) * 0.1).to(torch.bfloat16).to(dev())

    def run():
        gu = torch.zeros(M, 2 * I, dtype=torch.bfloat16, device=dev())
        act = torch.zeros(M, I, dtype=torch.bfloat16, device=dev())
        assert ops.linear_swiglu_fwd(x, w, gu, act)
        return gu, act

    os.environ["MGEN_GEMM_RS"] = "1"
    gu1, act1 = run()
    os.environ["MGEN_GEMM_RS"] = "0"
    gu0, act0 = run()
    os.environ.pop("MGEN_GEMM_RS", None)
    assert torch.equal(gu1, gu0) and torch.equal(act1, act0)
    # and against the separate kernels' arithmetic in torch: silu in fp32 from the bf16 gate, rounded, times up
    gf = gu1[:, :I].float()
    ref = ((gf * torch.sigmoid(gf)).to(torch.bfloat16).float() * gu1[:, I:].float()).to(torch.bfloat16)
    diff = (act1.float() - ref.float()).abs().max() / ref.float().abs().max()
    assert float(diff) < 1e-2


def test_transposed_weight_copies(ops):
    L, R, C = 3, 512, 1024
    g = torch.Generator().manual_seed(3)
    src = torch.randn(L, R * C + 64, generator=g).to(torch.bfloat16).to(dev())   # matrices 64 elements apart
    dst = torch.zeros(L, C, R, dtype=torch.bfloat16, device=dev())
    ops.transpose_bf16(src, dst, R, C, C, R, L, src.stride(0), C * R)
    for l in range(L):
        assert torch.equal(dst[l], src[l, :R * C].view(R, C).t())
    # ragged sizes
    s2 = torch.randn(70, 130, generator=g).to(torch.bfloat16).to(dev())
    d2 = torch.zeros(130, 72, dtype=torch.bfloat16, device=dev())
    ops.transpose_bf16(s2, d2, 70, 130, 130, 72)
    assert torch.equal(d2[:, :70], s2.t()) and bool((d2[:, 70:] == 0).all())


def test_training_step_identical_with_and_without(ops):
    """Default decoder, 8 x 2048 tokens: the loss is bit-identical and every gradient equal up to the summation order
    of the weight-gradient atomics whether the K = 512 products
    (QKV, gate/up + SwiGLU, lm_head logits, o_proj / down_proj input gradients through the transposed copies) run on
    the register-panel kernel or on the tile-loop kernel."""
    from mgen import MidiLM
    from mgen.config import default_decoder
    cfg = default_decoder()
    cfg.num_hidden_layers = 2
    cfg.lm_head_chunk_tokens = 16384
    model = MidiLM(cfg, device="cuda:0", seed=1)
    g = torch.Generator().manual_seed(9)
    ids = torch.randint(3, cfg.vocab_size, (8, 2048), generator=g).to(dev())

    def run():
        model.zero_grad()
        loss = model(input_ids=ids, labels=ids).loss
        loss.backward()
        return loss.detach().clone(), model.flat_grad.clone()

    (l1, g1), (l0, g0) = _both(run)
    assert torch.equal(l1, l0), (float(l1), float(l0))
    # (the weight-gradient products add their K slices with fp32 atomics in arrival order: not bit-reproducible
    # from run to run with either kernel)
    err = float((g1 - g0).abs().max() / g0.abs().max())
    assert err < 1e-5, err
    assert model._wt is not None and model._wt_epoch == model._shadow_epoch
    # initial loss of a randomly initialised decoder: ln V + d * 0.02^2 / 2 within a few percent
    assert abs(float(l1) - 11.93) < 0.4


# ---------------------------------------------------------------- output-stationary weight-gradient kernel
@pytest.mark.parametrize("M,N,K,acc", [(512, 1024, 4096, True), (3072, 512, 8192, True), (1000, 520, 4096, True),
                                       (2048, 768, 2048, False), (9000, 512, 2080, True)])
def test_weight_gradient_kernel(ops, M, N, K, acc):
    """csrc/gemm_os.hip (both operands stored [reduction][output index], fp32 output, with and without reduction
    slices, ragged tile edges) against fp32 arithmetic on the same bf16 inputs and against the tile-loop kernel."""
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(K, M, generator=g) * 0.5).to(torch.bfloat16).to(dev())
    b = (torch.randn(K, N, generator=g) * 0.5).to(torch.bfloat16).to(dev())
    ldc = N + 8
    outs = []
    for flag in ("1", "0"):
        os.environ["MGEN_GEMM_OS"] = flag
        c = torch.full((M, ldc), 3.0, dtype=torch.float32, device=dev())
        ops.gemm(a, b, c, M, N, K, M, N, ldc, True, True, ops.OUT_F32_ACC if acc else ops.OUT_F32)
        outs.append(c)
    os.environ.pop("MGEN_GEMM_OS", None)
    new, old = outs
    assert bool((new[:, N:] == 3.0).all()), "wrote outside the N columns"
    ref = a.float().t() @ b.float() + (3.0 if acc else 0.0)
    scale = float(ref.abs().max())
    assert float((new[:, :N] - ref).abs().max()) / scale < 2e-5
    assert float((new[:, :N] - old[:, :N]).abs().max()) / scale < 2e-6


@pytest.mark.parametrize("K,acc", [(4096, False), (8192, True)])
def test_weight_gradient_kernel_walks_long_reductions_in_segments(ops, K, acc):
    """An operand whose reduction range spans more than 4 GiB (here: a row pitch of 2 MiB, as the lm_head's
    dlogits chunk of 32 768 x 151 936 has 10 GB) is walked in segments with a buffer descriptor each. The plain store
    must equal, bit for bit, the same product taken from a compact copy of the operand (one segment, same summation
    order); the accumulating form (reduction slices of 2048 = two segments each) up to the order of its atomics."""
    M, N, lda = 512, 768, 1 << 20
    g = torch.Generator().manual_seed(K)
    wide = torch.zeros(K, lda, dtype=torch.bfloat16, device=dev())
    assert wide.numel() * 2 > (1 << 32)
    wide[:, :M] = (torch.randn(K, M, generator=g) * 0.5).to(torch.bfloat16).to(dev())
    wide[:, M:M + 256] = 7.0   # columns past M are fetched with the ragged tile and must not reach the output
    compact = wide[:, :M].contiguous()
    b = (torch.randn(K, N, generator=g) * 0.5).to(torch.bfloat16).to(dev())
    mode = ops.OUT_F32_ACC if acc else ops.OUT_F32
    c_wide = torch.full((M, N), 3.0, dtype=torch.float32, device=dev())
    c_comp = torch.full((M, N), 3.0, dtype=torch.float32, device=dev())
    ops.gemm(wide, b, c_wide, M, N, K, lda, N, N, True, True, mode)
    ops.gemm(compact, b, c_comp, M, N, K, M, N, N, True, True, mode)
    ref = compact.float().t() @ b.float() + (3.0 if acc else 0.0)
    scale = float(ref.abs().max())
    assert float((c_wide - ref).abs().max()) / scale < 2e-5
    if acc:
        assert float((c_wide - c_comp).abs().max()) / scale < 2e-6
    else:
        assert torch.equal(c_wide, c_comp)
    # the same with the wide operand on the other side
    c2 = torch.full((N, M), 3.0, dtype=torch.float32, device=dev())
    ops.gemm(b, wide, c2, N, M, K, N, lda, M, True, True, mode)
    assert float((c2 - ref.t()).abs().max()) / scale < 2e-5


@pytest.mark.parametrize("kind", ["dgrad", "proj"])
def test_output_stationary_row_major_outputs(ops, kind):
    """Input-gradient form (A reduction-contiguous, B stored [K][N], bf16 output) and residual-projection form (both
    operands reduction-contiguous, fp32 = residual + bf16(product)) of csrc/gemm_os.hip: bit for bit against the
    tile-loop kernel and against fp32 arithmetic."""
    M, N, K = 65536, 512, 1024
    g = torch.Generator().manual_seed(21)
    a = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(dev())
    if kind == "dgrad":
        b = (torch.randn(K, N, generator=g) * 0.1).to(torch.bfloat16).to(dev())
        res, mode, tb, dt = None, ops.OUT_BF16, True, torch.bfloat16
    else:
        b = (torch.randn(N, K, generator=g) * 0.1).to(torch.bfloat16).to(dev())
        res, mode, tb, dt = torch.randn(M, N, generator=g).to(dev()), ops.OUT_F32_RESID_BF16ROUND, False, torch.float32
    outs = []
    for flag in ("2", "0"):   # 2: also the residual projections, which are not routed to this kernel by default
        os.environ["MGEN_GEMM_OS"] = flag
        c = torch.zeros(M, N, dtype=dt, device=dev())
        ops.gemm(a, b, c, M, N, K, K, b.stride(0), N, False, tb, mode, res, N if res is not None else 0)
        outs.append(c)
    os.environ.pop("MGEN_GEMM_OS", None)
    assert torch.equal(outs[0], outs[1])
    rows = torch.arange(0, M, 509, device=dev())
    prod = a[rows].float() @ (b.float() if tb else b.float().t())
    ref = prod if res is None else prod.to(torch.bfloat16).float() + res[rows]
    err = float((outs[0][rows].float() - ref).abs().max() / ref.abs().max())
    assert err < 1e-2, err


def test_grouped_weight_gradients(ops):
    """mgen_gemm_bf16_wgrad_group: four products over the same tokens in one launch against the four separate calls
    (the grouping changes only how the reduction range is cut, i.e. the summation order)."""
    T = 8192
    g = torch.Generator().manual_seed(31)
    shapes = [(512, 2048), (4096, 512), (512, 1024), (3072, 512)]   # (N_i, K_i) of down, gate/up, o, qkv
    items, refs = [], []
    for n, k in shapes:
        dy = (torch.randn(T, n, generator=g) * 0.5).to(torch.bfloat16).to(dev())
        x = (torch.randn(T, k, generator=g) * 0.5).to(torch.bfloat16).to(dev())
        dw = torch.full((n, k), 2.0, dtype=torch.float32, device=dev())
        items.append((dy, x, dw))
        refs.append(dy.float().t() @ x.float() + 2.0)
    ops.linear_wgrad_group(items)
    for (dy, x, dw), ref in zip(items, refs):
        sep = torch.full_like(dw, 2.0)
        ops.linear_wgrad(dy, x, sep)
        scale = float(ref.abs().max())
        assert float((dw - ref).abs().max()) / scale < 2e-5
        assert float((dw - sep).abs().max()) / scale < 2e-6
    # products that do not fit the kernel fall back to separate calls
    small = [(items[0][0][:96], items[0][1][:96], torch.zeros_like(items[0][2]))]
    ops.linear_wgrad_group(small)
    ref = small[0][0].float().t() @ small[0][1].float()
    assert float((small[0][2] - ref).abs().max()) / float(ref.abs().max()) < 2e-5


def test_residual_add_with_norm_equals_residual_epilogue_then_norm(ops):
    """mgen_resid_rmsnorm_fwd on a bf16 projection output against the projection with the fp32 residual epilogue
    followed by mgen_rmsnorm_fwd: residual stream, normed rows and rstd bit for bit (also with a padded row pitch)."""
    T, d, K = 4096, 512, 1024
    g = torch.Generator().manual_seed(77)
    a = (torch.randn(T, K, generator=g) * 0.5).to(torch.bfloat16).to(dev())
    w = (torch.randn(d, K, generator=g) * 0.05).to(torch.bfloat16).to(dev())
    res = torch.randn(T, d, generator=g).to(dev())
    gain = (1 + 0.1 * torch.randn(d, generator=g)).to(dev())
    h_ref = torch.empty(T, d, dtype=torch.float32, device=dev())
    ops.gemm(a, w, h_ref, T, d, K, K, K, d, False, False, ops.OUT_F32_RESID_BF16ROUND, res, d)
    x_ref = torch.empty(T, d, dtype=torch.bfloat16, device=dev())
    r_ref = torch.empty(T, dtype=torch.float32, device=dev())
    ops.rmsnorm_fwd(h_ref, gain, x_ref, r_ref, 1e-6)
    for pad in (0, 64):
        y = torch.empty(T, d + pad, dtype=torch.bfloat16, device=dev())[:, :d]
        ops.gemm(a, w.t().contiguous(), y, T, d, K, K, d, y.stride(0), False, True, ops.OUT_BF16)
        h = torch.empty_like(h_ref)
        x = torch.empty_like(x_ref)
        r = torch.empty_like(r_ref)
        ops.resid_rmsnorm_fwd(res, y, gain, h, x, r, 1e-6)
        assert torch.equal(h, h_ref) and torch.equal(x, x_ref) and torch.equal(r, r_ref)


def test_split_residual_projections_change_nothing(ops):
    """Default decoder, 16 x 2048 tokens (the size from which the split path is taken): o_proj / down_proj as bf16
    products through the transposed weight copies + residual addition inside the following norm, against the fp32
    residual epilogue + separate norm: loss bit for bit, gradients up to the order of the weight-gradient atomics."""
    from mgen import MidiLM
    import mgen.model as mm
    from mgen.config import default_decoder
    cfg = default_decoder()
    cfg.num_hidden_layers = 2
    cfg.lm_head_chunk_tokens = 16384
    model = MidiLM(cfg, device="cuda:0", seed=3)
    g = torch.Generator().manual_seed(11)
    ids = torch.randint(3, cfg.vocab_size, (16, 2048), generator=g).to(dev())
    saved = mm._SPLIT_PROJ
    outs = []
    try:
        for flag in (True, False):
            mm._SPLIT_PROJ = flag
            model.zero_grad()
            loss = model(input_ids=ids, labels=ids).loss
            loss.backward()
            outs.append((loss.detach().clone(), model.flat_grad.clone()))
    finally:
        mm._SPLIT_PROJ = saved
    (l1, g1), (l0, g0) = outs
    assert torch.equal(l1, l0), (float(l1), float(l0))
    err = float((g1 - g0).abs().max() / g0.abs().max())
    assert err < 1e-5, err
    # and without gradients (evaluation / prefill use the same forward)
    with torch.no_grad():
        model.eval()
        le = model(input_ids=ids, labels=ids).loss
        model.train()
    assert torch.equal(le, l1)


def test_row_padding_changes_nothing(ops):
    """The 128 bytes of row padding that the training step gives qkv, the rotated q|k, the attention output and the
    gradient activations (pitches that would be multiples of 2 KiB) must not change a bit of the loss; gradients agree
    up to the summation order of the weight-gradient atomics."""
    from mgen import MidiLM
    import mgen.model as mm
    from mgen.config import default_decoder
    cfg = default_decoder()
    cfg.num_hidden_layers = 2
    cfg.lm_head_chunk_tokens = 16384
    model = MidiLM(cfg, device="cuda:0", seed=2)
    g = torch.Generator().manual_seed(10)
    ids = torch.randint(3, cfg.vocab_size, (8, 2048), generator=g).to(dev())
    saved = (mm._ROW_PAD, mm._ATT_PAD)
    outs = []
    try:
        for pad in (64, 0):
            mm._ROW_PAD = mm._ATT_PAD = pad
            model.zero_grad()
            loss = model(input_ids=ids, labels=ids).loss
            loss.backward()
            outs.append((loss.detach().clone(), model.flat_grad.clone()))
    finally:
        mm._ROW_PAD, mm._ATT_PAD = saved
    (l1, g1), (l0, g0) = outs
    assert torch.equal(l1, l0), (float(l1), float(l0))
    err = float((g1 - g0).abs().max() / g0.abs().max())
    assert err < 1e-5, err
