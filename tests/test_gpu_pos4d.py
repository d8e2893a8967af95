"""GPU parity of the 4-D position model: kernels against the CPU oracle on seeded inputs, the whole model against
the fixture made from the reference's own module (tests/golden/tiny_pos4d.npz). bf16 tolerance 2e-2 relative
(max-norm); fp32 pieces (table, loss on fp32 pairs, note decoding) are checked to 1e-5."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R
from oracle import ref_pos4d as P
from golden_util import TINY_POS4D as OCFG, load, tiny_params_pos4d, rel_err

BF16_TOL = 2e-2


def _cfg():
    from mgen.config import LMConfig
    kw = {k: getattr(OCFG, k) for k in R.LMConfig.__dataclass_fields__}
    return LMConfig(pos4d=True, **kw)


def _rows(gen, T):
    start = torch.rand(T, generator=gen) * 300.0
    dur = torch.rand(T, generator=gen) * 4.0
    pitch = torch.randint(0, 128, (T,), generator=gen).float()
    vel = torch.randint(0, 128, (T,), generator=gen).float()
    return torch.stack([start, dur, pitch, vel], 1)


@pytest.mark.parametrize("D", [128, 64])
def test_rope4d_table(D):
    from mgen import ops
    g = torch.Generator().manual_seed(1)
    rows = _rows(g, 777)
    f = P.rope4d_freqs(D)
    cs = ops.rope4d_table(rows.cuda(), f.cuda(), D).cpu()
    ang = P.rope4d_angles(rows[None], D)[0]          # [T, D], each angle twice
    np.testing.assert_allclose(cs[..., 0].numpy(), ang[:, 0::2].cos().numpy(), atol=2e-6)
    np.testing.assert_allclose(cs[..., 1].numpy(), ang[:, 0::2].sin().numpy(), atol=2e-6)


@pytest.mark.parametrize("D,H", [(128, 2), (64, 4)])
def test_qkv_rotation_fwd_bwd(D, H):
    from mgen import ops
    g = torch.Generator().manual_seed(2)
    B, S = 2, 67
    T = B * S
    rows = _rows(g, T)
    NQ = 3 * H * D
    qkv = (torch.randn(T, NQ, generator=g) * 1.5).bfloat16()
    wq = 1.0 + 0.1 * torch.randn(D, generator=g)
    wk = 1.0 + 0.1 * torch.randn(D, generator=g)
    dout = torch.randn(T, NQ, generator=g).bfloat16()
    # oracle (autograd through the bf16-mode arithmetic)
    ang = P.rope4d_angles(rows.view(B, S, 4), D)
    cos, sin = ang.cos(), ang.sin()
    x = qkv.float().requires_grad_(True)
    wq_, wk_ = wq.clone().requires_grad_(True), wk.clone().requires_grad_(True)
    q, k, v = [t.view(B, S, H, D) for t in x.split(H * D, dim=1)]
    with torch.no_grad():   # forward value with the bf16 rounding of the normalised rows, as under autocast
        qn = R.rmsnorm(q.bfloat16(), wq, 1e-6)
        kn = R.rmsnorm(k.bfloat16(), wk, 1e-6)
        ref = torch.cat([P.rotate(t.transpose(1, 2), cos, sin).transpose(1, 2).reshape(T, H * D)
                         for t in (qn, kn, v)], 1)
    # gradient through the unrounded twin
    qn2 = wq_ * (q * torch.rsqrt(q.pow(2).mean(-1, keepdim=True) + 1e-6))
    kn2 = wk_ * (k * torch.rsqrt(k.pow(2).mean(-1, keepdim=True) + 1e-6))
    ref2 = torch.cat([P.rotate(t.transpose(1, 2), cos, sin).transpose(1, 2).reshape(T, H * D) for t in (qn2, kn2, v)], 1)
    (ref2 * dout.float()).sum().backward()

    dev = "cuda"
    cs = ops.rope4d_table(rows.to(dev), P.rope4d_freqs(D).to(dev), D)
    qkv_d = qkv.to(dev)
    out = torch.empty_like(qkv_d)
    o = out.data_ptr()
    ops.qkvnorm_rope4d_fwd(qkv_d.data_ptr(), NQ, o, o + 2 * H * D, o + 4 * H * D, NQ, NQ, NQ, wq.to(dev), wk.to(dev), cs,
                           T, H, H, H, D, 1e-6)
    assert rel_err(out.float().cpu().numpy(), ref.detach().numpy()) < BF16_TOL
    dq = dout.to(dev).clone()
    dwq = torch.zeros(D, device=dev)
    dwk = torch.zeros(D, device=dev)
    p = dq.data_ptr()
    ops.qkvnorm_rope4d_bwd(p, NQ, p, p + 2 * H * D, p + 4 * H * D, NQ, NQ, NQ, qkv_d, NQ, wq.to(dev), wk.to(dev), dwq,
                           dwk, cs, T, H, H, H, D, 1e-6)
    assert rel_err(dq.float().cpu().numpy(), x.grad.numpy()) < BF16_TOL
    assert rel_err(dwq.cpu().numpy(), wq_.grad.numpy()) < BF16_TOL
    assert rel_err(dwk.cpu().numpy(), wk_.grad.numpy()) < BF16_TOL


def test_plain_rotation_and_inverse():
    from mgen import ops
    g = torch.Generator().manual_seed(3)
    T, H, D = 300, 2, 128
    rows = _rows(g, T)
    x = torch.randn(T, H * D, generator=g).bfloat16()
    ang = P.rope4d_angles(rows[None], D)
    cos, sin = ang.cos(), ang.sin()
    xr = x.float().view(1, T, H, D).transpose(1, 2)
    fwd = P.rotate(xr, cos, sin).transpose(1, 2).reshape(T, H * D)
    inv = P.rotate_T(xr, cos, sin).transpose(1, 2).reshape(T, H * D)
    cs = ops.rope4d_table(rows.cuda(), P.rope4d_freqs(D).cuda(), D)
    xd = x.cuda()
    y = torch.empty_like(xd)
    ops.rotate4d(xd, y, cs, T, H, D, inverse=False)
    assert rel_err(y.float().cpu().numpy(), fwd.numpy()) < 4e-3   # one bf16 rounding of the result
    ops.rotate4d(xd, y, cs, T, H, D, inverse=True)
    assert rel_err(y.float().cpu().numpy(), inv.numpy()) < 4e-3
    # in place, and inverse(forward(x)) == x up to two bf16 roundings
    ops.rotate4d(xd, xd, cs, T, H, D, inverse=False)
    ops.rotate4d(xd, xd, cs, T, H, D, inverse=True)
    assert rel_err(xd.float().cpu().numpy(), x.float().numpy()) < 1e-2


def _labels(gen, T, masked_every=5):
    lab = _rows(gen, T)
    lab[:, 0] = torch.rand(T, generator=gen) * 3.0     # start deltas
    lab[::masked_every] = -100.0
    lab[1, 2] = -100.0                                  # one -100 in a row masks the row
    return lab


def test_cosine_loss_and_gradient_fp32():
    from mgen import ops
    g = torch.Generator().manual_seed(4)
    T = 1000
    pairs = torch.randn(T, 16, generator=g)
    lab = _labels(g, T)
    x = pairs.clone().requires_grad_(True)
    pr = x.view(1, T, 8, 2)
    pred = pr / (torch.norm(pr, dim=-1, keepdim=True) + 1e-8)
    loss = P.cosine_loss(pred, lab.view(1, T, 4))
    loss.backward()
    dev = "cuda"
    inv = torch.empty(1, device=dev)
    cnt = torch.empty(1, dtype=torch.int32, device=dev)
    ops.cos4d_count(lab.to(dev), inv, cnt)
    n_valid = int((lab != -100).all(-1).sum())
    assert int(cnt) == n_valid and abs(float(inv) - 1.0 / (8 * n_valid)) < 1e-9
    pred_d = torch.empty(T, 16, device=dev)
    rl = torch.empty(T, device=dev)
    dp = torch.empty(T, 16, device=dev)
    ops.cos4d_fwd_bwd(pairs.to(dev), lab.to(dev), inv, 1.0, pred_d, rl, dp)
    out = torch.empty((), device=dev)
    ops.ce_finalize(rl, inv, out)
    assert abs(float(out) - float(loss)) < 1e-5 * abs(float(loss))
    np.testing.assert_allclose(pred_d.cpu().numpy(), pred.detach().view(T, 16).numpy(), atol=1e-6)
    assert rel_err(dp.cpu().numpy(), x.grad.numpy()) < 1e-5
    # targets alone: encoding of the labels (rows that count)
    # all rows masked -> zero loss and zero gradient
    lab2 = torch.full((T, 4), -100.0)
    ops.cos4d_count(lab2.to(dev), inv, cnt)
    ops.cos4d_fwd_bwd(pairs.to(dev), lab2.to(dev), inv, 1.0, pred_d, rl, dp)
    ops.ce_finalize(rl, inv, out)
    assert float(out) == 0.0 and int(cnt) == 0 and float(dp.abs().max()) == 0.0


def test_component_losses():
    # the eight values the reference logs beside its loss, against the oracle on unit pairs and masked labels
    from mgen import ops
    g = torch.Generator().manual_seed(14)
    for T in (1, 77, 5000):
        pr = torch.randn(T, 8, 2, generator=g)
        pred = pr / (torch.norm(pr, dim=-1, keepdim=True) + 1e-8)
        lab = _labels(g, T) if T > 1 else _rows(g, T)
        want = P.component_losses(pred, lab)
        out = torch.empty(8, device="cuda")
        ops.cos4d_components(pred.view(T, 16).cuda(), lab.cuda(), out)
        np.testing.assert_allclose(out.cpu().numpy(), want.numpy(), rtol=1e-5, atol=1e-6)
        assert abs(float(out.mean()) - float(P.cosine_loss(pred, lab))) < 1e-5
    ops.cos4d_components(pred.view(T, 16).cuda(), torch.full((T, 4), -100.0).cuda(), out)
    assert float(out.abs().max()) == 0.0


def test_next_note_decoding():
    from mgen import ops
    g = torch.Generator().manual_seed(5)
    B = 257
    ang = torch.rand(B, 8, generator=g) * 2 * torch.pi - torch.pi / 2
    pred = torch.stack([ang.cos(), ang.sin()], -1).view(B, 16)
    prev = _rows(g, B)
    want = torch.stack([P.process_last_logits(pred[b].view(8, 2)) for b in range(B)])
    want[:, 0] += prev[:, 0]
    pd, pv = pred.cuda(), prev.cuda()
    nxt = torch.zeros(B, 4, device="cuda")
    ops.pos4d_next(pd.data_ptr(), 16, pv.data_ptr(), 4, nxt.data_ptr(), 4, B)
    got = nxt.cpu()
    # rounding decisions of the coarse parts may flip when an angle sits on a boundary (atan2 differs in the last
    # ulp between libraries): require exact agreement on all but a handful of entries
    close = torch.isclose(got, want, rtol=1e-5, atol=1e-4)
    assert close.float().mean() > 0.995, (got[~close], want[~close])


# ---------------------------------------------------------------------------------------------------------
# whole model against the fixture
# ---------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def tiny():
    from mgen import MidiLM
    g = load("tiny_pos4d")
    p = tiny_params_pos4d()
    m = MidiLM(_cfg(), device="cuda:0", seed=None)
    m.load_state_dict(p)
    return m, g, p


def test_state_dict_names(tiny):
    m, g, p = tiny
    sd = m.state_dict()
    assert set(sd) == set(p)
    assert torch.equal(sd["model.embed_vector"].cpu(), p["model.embed_vector"])
    assert tuple(sd["lm_head.weight"].shape) == (16, OCFG.hidden_size)


def test_forward_pred_matches_reference(tiny):
    m, g, p = tiny
    pos = torch.from_numpy(g["position_tensors"]).cuda()
    out = m(input_ids=None, position_tensors=pos)
    assert out.loss is None and tuple(out.logits.shape) == (2, 128, 8, 2)
    pred = out.logits.cpu().numpy()
    assert rel_err(pred, g["fp32_pred"]) < BF16_TOL
    assert rel_err(pred, g["bf16_pred"]) < BF16_TOL
    _, po = P.forward_pos4d(p, OCFG, pos.cpu(), None, "bf16")
    assert rel_err(pred, po.float().numpy()) < BF16_TOL


def test_loss_and_grads_match_reference(tiny):
    m, g, p = tiny
    pos = torch.from_numpy(g["position_tensors"]).cuda()
    lab = torch.from_numpy(g["labels"]).cuda()
    mask = torch.from_numpy(g["attention_mask"]).cuda()
    m.zero_grad()
    out = m(input_ids=torch.zeros(2, 128, dtype=torch.long, device="cuda"), attention_mask=mask, labels=lab,
            position_tensors=pos)
    out.loss.backward()
    loss = float(out.loss)
    assert abs(loss - float(g["fp32_loss"])) < 2e-3 * abs(float(g["fp32_loss"]))
    assert abs(loss - float(g["bf16_loss"])) < 2e-3 * abs(float(g["bf16_loss"]))
    grads = {n: q.grad.detach().float().cpu().numpy() for n, q in m.named_parameters()}
    worst = {}
    for k in g.files:
        if k.startswith("fp32_grad/"):
            name = k.split("/", 1)[1]
            worst[name] = rel_err(grads[name], g[k])
    print({k: round(v, 5) for k, v in worst.items()})
    assert max(worst.values()) < BF16_TOL, worst
    with torch.no_grad():
        l2 = float(m(position_tensors=pos, labels=lab).loss)
    assert abs(l2 - loss) < 1e-6 * abs(loss)
    comp = m.component_losses()
    assert list(comp) == ["loss/s_0", "loss/s_1", "loss/d_0", "loss/d_1", "loss/p_0", "loss/p_1", "loss/v_0", "loss/v_1"]
    cv = torch.stack(list(comp.values())).cpu()
    assert abs(float(cv.mean()) - l2) < 1e-5 * abs(l2)
    want = P.component_losses(m._last_pred.view(2, 128, 8, 2).cpu(), lab.cpu())
    np.testing.assert_allclose(cv.numpy(), want.numpy(), rtol=1e-5, atol=1e-6)
    # all rows masked: zero loss, zero gradients
    m.zero_grad()
    o = m(position_tensors=pos, labels=torch.full_like(lab, -100.0))
    o.loss.backward()
    assert float(o.loss) == 0.0 and float(m.flat_grad.abs().max()) == 0.0


def test_training_steps_follow_reference_curve():
    from mgen import MidiLM, get_optim
    g = load("tiny_pos4d")
    m = MidiLM(_cfg(), device="cuda:0", seed=None)
    m.load_state_dict(tiny_params_pos4d())
    pos = torch.from_numpy(g["position_tensors"]).cuda()
    lab = torch.from_numpy(g["labels"]).cuda()
    opt, sched = get_optim(m, 5e-4, 12, 20, warmup=10, max_grad_norm=5.0)
    losses, lrs = [], []
    for step in range(14):
        opt.zero_grad()
        loss = m(position_tensors=pos, labels=lab).loss
        loss.backward()
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
        losses.append(float(loss))
    np.testing.assert_allclose(lrs, g["train_lrs"], rtol=1e-9)
    print(losses, list(g["train_losses"]))
    np.testing.assert_allclose(losses, g["train_losses"], rtol=2e-2)
    assert losses[-1] < 0.35 * losses[0]


def test_generated_preds_teacher_forced_on_reference_rows(tiny):
    """Causality makes the pred at row i a function of rows 0..i: run the reference's generated rows through the
    model once and compare every pred with the one the reference computed while generating."""
    m, g, p = tiny
    rows = torch.from_numpy(g["gen_rows"])[None, :48].cuda()
    pred = m(position_tensors=rows).logits[0].cpu().numpy()
    assert rel_err(pred, g["gen_preds"]) < BF16_TOL


@pytest.mark.parametrize("use_cache", [True, False])
def test_generate_music_self_consistent(tiny, use_cache):
    m, g, p = tiny
    first = torch.from_numpy(g["position_tensors"])[0:1, 0:1, :].cuda()
    rows, preds = m.generate_music(first, 40, use_cache=use_cache, return_preds=True)
    assert tuple(rows.shape) == (1, 41, 4) and tuple(preds.shape) == (1, 40, 8, 2)
    rows_c, preds_c = rows[0].cpu(), preds[0].cpu()
    # every appended row is the decoding of the pred that produced it (start accumulated on the previous row)
    for i in range(40):
        want = P.process_last_logits(preds_c[i])
        want[0] += rows_c[i, 0]
        np.testing.assert_allclose(rows_c[i + 1].numpy(), want.numpy(), rtol=1e-5, atol=1e-4)
    # and the preds are what the oracle computes from the same rows
    _, po = P.forward_pos4d(p, OCFG, rows_c[None, :40], None, "bf16")
    assert rel_err(preds_c.numpy(), po[0].float().numpy()) < BF16_TOL
    # start times never decrease, pitch/velocity are integers
    assert bool((rows_c[1:, 0] >= rows_c[:-1, 0]).all())
    assert torch.equal(rows_c[:, 2:], rows_c[:, 2:].round())


def test_generate_music_batch_of_streams(tiny):
    m, g, p = tiny
    prompt = torch.from_numpy(g["position_tensors"])[:, :5, :].cuda()
    rows = m.generate_music(prompt, 12)
    assert tuple(rows.shape) == (2, 17, 4)
    assert torch.equal(rows[:, :5], prompt)
    solo = m.generate_music(prompt[1:2], 12)
    np.testing.assert_allclose(rows[1].cpu().numpy(), solo[0].cpu().numpy(), rtol=1e-3, atol=1e-2)


# ---------------------------------------------------------------------------------------------------------
# fp32 parity mode: within 1e-4 of the fp32 outputs of the reference's own module
def test_fp32_mode_matches_the_reference_module():
    from mgen import MidiLM
    g = load("tiny_pos4d")
    m = MidiLM(_cfg(), device="cuda:0", seed=None, compute_dtype=torch.float32)
    m.load_state_dict(tiny_params_pos4d())
    pos = torch.from_numpy(g["position_tensors"]).cuda()
    lab = torch.from_numpy(g["labels"]).cuda()
    mask = torch.from_numpy(g["attention_mask"]).cuda()
    with torch.no_grad():
        pred = m(position_tensors=pos).logits.float().cpu().numpy()
    assert rel_err(pred, g["fp32_pred"]) < 1e-4
    m.zero_grad()
    out = m(position_tensors=pos, attention_mask=mask, labels=lab)
    out.loss.backward()
    assert abs(float(out.loss.detach()) - float(g["fp32_loss"])) < 1e-4 * abs(float(g["fp32_loss"]))
    grads = {n: q.grad.detach().float().cpu().numpy() for n, q in m.named_parameters()}
    worst = {}
    for k in g.files:
        if k.startswith("fp32_grad/"):
            name = k.split("/", 1)[1]
            worst[name] = rel_err(grads[name], g[k])
    print({k: float("%.2e" % v) for k, v in worst.items()})
    assert max(worst.values()) < 1e-4, worst
    # the reference's generated rows, teacher-forced, and cache-less generation
    rows = torch.from_numpy(g["gen_rows"])[None, :48].cuda()
    with torch.no_grad():
        pr = m(position_tensors=rows).logits[0].float().cpu().numpy()
    assert rel_err(pr, g["gen_preds"]) < 1e-4
    gen, preds = m.generate_music(rows[:, :1], 6, return_preds=True)      # (no KV cache in this mode)
    assert tuple(gen.shape) == (1, 7, 4)
    assert rel_err(preds[0, 0].cpu().numpy(), g["gen_preds"][0]) < 1e-4
    np.testing.assert_allclose(gen[0, 1].cpu().numpy(), g["gen_rows"][1], rtol=1e-4, atol=1e-3)
