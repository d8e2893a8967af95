"""Error behaviour and call-surface semantics of the GPU path that the reference gets from torch / transformers:
out-of-range ids and labels raise, attention_mask must be right padding, two forwards may precede one backward,
the optimizer's state survives a checkpoint."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R
from golden_util import CASES, load, tiny_params, rel_err


def _cfg(ocfg, **kw):
    from mgen.config import LMConfig
    return LMConfig(**{k: getattr(ocfg, k) for k in R.LMConfig.__dataclass_fields__}, **kw)


@pytest.fixture(scope="module")
def tiny():
    from mgen import MidiLM
    ocfg = CASES["tiny_qwen3"]
    m = MidiLM(_cfg(ocfg), device="cuda:0", seed=None)
    m.load_state_dict(tiny_params(ocfg))
    return m, ocfg, load("tiny_qwen3")


def test_bad_token_id_raises_in_training_and_does_not_poison_generate(tiny):
    """The reference's embedding lookup fails on an id outside the vocabulary. Here the kernel clamps and raises a
    device flag: the training step reports it after backward, and the flag does not leak into later calls."""
    m, ocfg, g = tiny
    ids = torch.from_numpy(g["input_ids"]).cuda()
    bad = ids.clone()
    bad[0, 3] = ocfg.vocab_size + 5
    m.zero_grad()
    loss = m(input_ids=bad, labels=ids).loss
    with pytest.raises(IndexError):
        loss.backward()
    # evaluation forward: raised at the end of the call
    with pytest.raises(IndexError):
        with torch.no_grad():
            m(input_ids=bad, labels=ids)
    # a clean generate() after it works, and a bad prompt raises in generate() itself, once
    out = m.generate(ids[:1, :8], max_new_tokens=4, do_sample=False)
    assert out.shape == (1, 12)
    with pytest.raises(IndexError):
        m.generate(bad[:1, :8], max_new_tokens=4, do_sample=False)
    assert m.generate(ids[:1, :8], max_new_tokens=4, do_sample=False).shape == (1, 12)
    m.zero_grad()
    m(input_ids=ids, labels=ids).loss.backward()   # and a clean training step does not raise


def test_label_out_of_range_raises_and_is_not_counted(tiny):
    from mgen import ops
    m, ocfg, g = tiny
    ids = torch.from_numpy(g["input_ids"]).cuda()
    lab = ids.clone()
    lab[1, 5] = ocfg.vocab_size      # torch's cross_entropy asserts here
    m.zero_grad()
    loss = m(input_ids=ids, labels=lab).loss
    with pytest.raises(IndexError):
        loss.backward()
    # the count leaves the label out exactly as the loss kernels leave it out of the sum
    inv = torch.empty(1, device="cuda")
    cnt = torch.empty(1, dtype=torch.int32, device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.ce_count(lab, inv, cnt, V=ocfg.vocab_size, err_flag=flag)
    B, S = lab.shape
    assert int(cnt) == B * (S - 1) - 1 and int(flag) == 4
    flag.zero_()
    ops.ce_count(ids, inv, cnt, V=ocfg.vocab_size, err_flag=flag)
    assert int(cnt) == B * (S - 1) and int(flag) == 0


def test_attention_mask_must_be_right_padding(tiny):
    m, ocfg, g = tiny
    ids = torch.from_numpy(g["input_ids"]).cuda()
    right = torch.from_numpy(g["pad_mask"]).cuda()
    labels = torch.from_numpy(g["pad_labels"]).cuda()
    with torch.no_grad():
        ok = m(input_ids=ids, attention_mask=right, labels=labels).loss     # the collators' layout: accepted
    assert abs(float(ok) - float(g["pad_loss"])) < 5e-3 * abs(float(g["pad_loss"]))
    left = torch.flip(right, dims=[1])
    assert bool((left[:, 0] == 0).any())
    with pytest.raises(ValueError):
        with torch.no_grad():
            m(input_ids=ids, attention_mask=left, labels=labels)
    hole = torch.ones_like(right)
    hole[0, 4] = 0
    m.zero_grad()
    loss = m(input_ids=ids, attention_mask=hole, labels=ids).loss
    with pytest.raises(ValueError):
        loss.backward()
    with pytest.raises(ValueError):
        m(input_ids=ids, attention_mask=right[:, :-1])


@pytest.mark.parametrize("pos4d", [False, True])
def test_two_forwards_before_one_backward(pos4d):
    """loss_a + loss_b followed by one backward: the head's weight gradient of each forward must survive the
    other forward (it is formed during forward)."""
    from mgen import MidiLM
    if pos4d:
        from golden_util import TINY_POS4D, tiny_params_pos4d
        m = MidiLM(_cfg(TINY_POS4D, pos4d=True), device="cuda:0", seed=None)
        m.load_state_dict(tiny_params_pos4d())
        gp = load("tiny_pos4d")
        pt = torch.from_numpy(gp["position_tensors"]).cuda()
        lb = torch.from_numpy(gp["labels"]).cuda()
        batches = [dict(position_tensors=pt[:1], labels=lb[:1]), dict(position_tensors=pt[1:], labels=lb[1:])]
    else:
        ocfg = CASES["tiny_llama"]
        m = MidiLM(_cfg(ocfg), device="cuda:0", seed=None)
        m.load_state_dict(tiny_params(ocfg))
        ids = torch.from_numpy(load("tiny_llama")["input_ids"]).cuda()
        batches = [dict(input_ids=ids[:1], labels=ids[:1]), dict(input_ids=ids[1:], labels=ids[1:])]
    sep = None
    for bt in batches:            # reference: one backward per forward, gradients accumulate
        m.zero_grad() if sep is None else None
        m(**bt).loss.backward()
        sep = True
    want = m.flat_grad.clone()
    m.zero_grad()
    la = m(**batches[0]).loss
    lb_ = m(**batches[1]).loss
    (la + lb_).backward()
    got = m.flat_grad
    err = float((got - want).abs().max() / want.abs().max())
    assert err < 1e-6, err


def test_optimizer_state_dict_round_trip(tiny):
    from mgen import MidiLM, get_optim
    _, ocfg, g = tiny
    ids = torch.from_numpy(g["input_ids"]).cuda()

    def make():
        m = MidiLM(_cfg(ocfg), device="cuda:0", seed=None)
        m.load_state_dict(tiny_params(ocfg))
        return (m,) + get_optim(m, 5e-4, 12, 20, warmup=2)

    def step(m, opt, sched):
        opt.zero_grad()
        m(input_ids=ids, labels=ids).loss.backward()
        opt.step()
        sched.step()

    m1, o1, s1 = make()
    for _ in range(3):
        step(m1, o1, s1)
    saved = {"model": {k: v.clone() for k, v in m1.state_dict().items()}, "opt": o1.state_dict(), "sched": s1.state_dict()}
    for _ in range(2):
        step(m1, o1, s1)
    m2, o2, s2 = make()
    m2.load_state_dict(saved["model"])
    o2.load_state_dict(saved["opt"])
    s2.load_state_dict(saved["sched"])
    assert o2.step_count == 3
    for _ in range(2):
        step(m2, o2, s2)
    # (weight gradients are summed with float atomics: two runs agree to rounding, not bit for bit)
    diff = lambda a, b: float((a.flat_param - b.flat_param).abs().max() / a.flat_param.abs().max())
    assert diff(m1, m2) < 1e-5
    # without the optimizer's state the resumed run restarts the moments and the bias correction: visibly different
    m3, o3, s3 = make()
    m3.load_state_dict(saved["model"])
    s3.load_state_dict(saved["sched"])
    for _ in range(2):
        step(m3, o3, s3)
    assert diff(m1, m3) > 100 * diff(m1, m2) and diff(m1, m3) > 1e-4
    with pytest.raises(KeyError):
        o3.load_state_dict({k: v for k, v in saved["opt"].items() if k != "fused"})
