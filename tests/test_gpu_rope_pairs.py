"""Token models with adjacent rotation pairs (LMConfig.rope_interleaved) on the GPU. The split-half fixtures pin it
through the row permutation of golden_util.to_interleaved (the permuted model is the same function, see
tests/test_oracle_rope_pairs.py): fp32 mode within 1e-4 of the fixtures, bf16-mixed within the bf16 budget (2e-2
max-norm relative), KV-cached greedy decoding equal to the cache-less forward."""
import dataclasses

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R
from golden_util import CASES, load, tiny_params, slim, rel_err, to_interleaved, rows_from_interleaved

BF16_TOL = 2e-2


def _model(case, dtype=None):
    from mgen import MidiLM
    from mgen.config import LMConfig
    ocfg = dataclasses.replace(CASES[case], rope_interleaved=True)
    cfg = LMConfig(**{k: getattr(ocfg, k) for k in R.LMConfig.__dataclass_fields__})
    kw = {} if dtype is None else {"compute_dtype": dtype}
    m = MidiLM(cfg, device="cuda:0", seed=None, **kw)
    p = to_interleaved(tiny_params(CASES[case]), CASES[case])
    m.load_state_dict(p)
    return m, ocfg, p


def _grads_in_fixture_order(m, ocfg):
    out = {}
    for n, q in m.named_parameters():
        gq = q.grad.detach().float().cpu()
        if n.endswith(("q_proj.weight", "k_proj.weight", "q_norm.weight", "k_norm.weight")):
            gq = rows_from_interleaved(gq, ocfg)
        out[n] = gq.numpy()
    return out


@pytest.mark.parametrize("case", list(CASES))
def test_fp32_mode_reproduces_the_fixture(case):
    m, ocfg, p = _model(case, torch.float32)
    g = load(case)
    ids = torch.from_numpy(g["input_ids"]).cuda()
    logits = m(input_ids=ids).logits.float().cpu().numpy()
    assert rel_err(slim(logits), g["fp32_logits"]) < 1e-4
    m.zero_grad()
    loss = m(input_ids=ids, labels=ids).loss
    loss.backward()
    assert abs(float(loss) - float(g["fp32_loss"])) < 1e-4 * abs(float(g["fp32_loss"]))
    grads = _grads_in_fixture_order(m, ocfg)
    for k in g.files:
        if k.startswith("fp32_grad/"):
            name = k.split("/", 1)[1]
            assert rel_err(slim(grads[name]), g[k]) < 1e-4, name


@pytest.mark.parametrize("case", list(CASES))
def test_bf16_mode_loss_logits_grads(case):
    m, ocfg, p = _model(case)
    g = load(case)
    ids = torch.from_numpy(g["input_ids"]).cuda()
    logits = m(input_ids=ids).logits.float().cpu().numpy()
    assert rel_err(slim(logits), g["fp32_logits"]) < BF16_TOL
    m.zero_grad()
    loss = m(input_ids=ids, labels=ids).loss
    loss.backward()
    assert abs(float(loss) - float(g["fp32_loss"])) < 2e-3 * abs(float(g["fp32_loss"]))
    grads = _grads_in_fixture_order(m, ocfg)
    worst = {k.split("/", 1)[1]: rel_err(slim(grads[k.split("/", 1)[1]]), g[k]) for k in g.files
             if k.startswith("fp32_grad/")}
    assert max(worst.values()) < BF16_TOL, worst


def test_cached_greedy_decoding_follows_the_fixture_tokens():
    m, ocfg, p = _model("tiny_llama")
    g = load("tiny_llama")
    want = torch.from_numpy(g["greedy_tokens"])
    out = m.generate(want[:, :8].cuda(), max_new_tokens=56, do_sample=False)
    got = out.cpu()
    # bf16 decoding may leave the fp32 reference at a near-tie; up to there the tokens are the fixture's
    same = int((got[0, :64] == want[0, :64]).long().cumprod(0).sum())
    assert same >= 32, same
    # and the KV-cached path agrees with the cache-less forward of the same model at every generated position
    logits = m(input_ids=got[:, :-1].cuda()).logits.float()
    pred = logits[0, 7:].argmax(-1).cpu()
    agree = float((pred == got[0, 8:]).float().mean())
    assert agree >= 0.95, agree
