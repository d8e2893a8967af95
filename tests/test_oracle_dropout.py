"""Dropout on the CPU: the mask functions (pure integer arithmetic, restated in oracle/ref_dropout.py) and the GPT-2
oracle in training mode against the fixture made from transformers' GPT2LMHeadModel with the same masks injected
(tests/golden/tiny_gpt2_dropout.npz, tools/make_golden.py:gpt2_dropout_case)."""
import numpy as np
import pytest
import torch

from oracle import ref_dropout as RD
from oracle import ref_gpt2 as G
from golden_util import TINY_GPT2 as CFG, load, tiny_params_gpt2, checksum, slim, rel_err


def test_fmix32_known_answers():
    # murmur3's 32-bit finaliser: fixed points and published avalanche values
    got = RD.fmix32(np.array([0, 1, 0xFFFFFFFF, 0x12345678], dtype=np.uint32))
    assert got[0] == 0
    assert [int(x) for x in got[1:]] == [0x514E28B7, 0x81F16F39, 0xE37CD1BC]


def test_site_keys_are_distinct_and_stable():
    keys = {RD.site_key(5, c) for c in range(1000)} | {RD.site_key(6, c) for c in range(1000)}
    assert len(keys) == 2000
    assert RD.site_key(1, 0) == 0x910A2DEC89025CC1     # splitmix64(1 + golden ratio)


@pytest.mark.parametrize("p", [0.05, 0.1, 0.5])
def test_keep_probability_and_independence(p):
    key = RD.site_key(3, 17)
    n = 1 << 20
    keep = RD.keep_flat(key, n, p).numpy()
    sigma = np.sqrt(p * (1 - p) / n)
    assert abs(keep.mean() - (1 - p)) < 5 * sigma
    # neighbours and strided neighbours are uncorrelated
    for lag in (1, 2, 8, 256, 4096):
        a, b = keep[:-lag].astype(np.float64), keep[lag:].astype(np.float64)
        corr = np.corrcoef(a, b)[0, 1]
        assert abs(corr) < 6 / np.sqrt(n), (lag, corr)
    other = RD.keep_flat(RD.site_key(3, 18), n, p).numpy()
    assert abs(np.corrcoef(keep, other)[0, 1]) < 6 / np.sqrt(n)
    att = RD.keep_attn(key, 2, 4, 256, p).numpy()
    m = att.size
    assert abs(att.mean() - (1 - p)) < 5 * np.sqrt(p * (1 - p) / m)
    # along keys, along queries, across heads
    for a, b in ((att[..., :-1], att[..., 1:]), (att[:, :, :-1, :], att[:, :, 1:, :]), (att[:, :-1], att[:, 1:])):
        assert abs(np.corrcoef(a.ravel(), b.ravel())[0, 1]) < 6 / np.sqrt(a.size)


def test_p_zero_keeps_everything_and_threshold():
    assert bool(RD.keep_flat(RD.site_key(0, 0), 10000, 0.0).all())
    assert RD.threshold(0.5) == 1 << 31 and RD.threshold(0.0) == 0


def test_apply_is_torch_dropout_arithmetic():
    x = torch.randn(4096)
    keep = RD.keep_flat(1, 4096, 0.1)
    y = RD.apply(x, keep, 0.1)
    assert torch.equal(y[~keep], torch.zeros(int((~keep).sum())))
    torch.testing.assert_close(y[keep], x[keep] / 0.9, rtol=1e-6, atol=0)
    yb = RD.apply(x.bfloat16(), keep, 0.1)
    assert yb.dtype == torch.bfloat16


@pytest.fixture(scope="module")
def gold():
    g = load("tiny_gpt2_dropout")
    p = tiny_params_gpt2()
    assert abs(checksum(p) - float(g["param_checksum"])) < 1e-6 * float(g["param_checksum"])
    return g, p


def test_gpt2_training_mode_matches_transformers_with_the_same_masks(gold):
    g, p = gold
    ids = torch.from_numpy(g["input_ids"])
    pd = float(g["pdrop"])
    plan = RD.DropPlan(int(g["seed"]), 0, CFG.num_hidden_layers, pd, pd, pd)
    loss, grads = G.loss_and_grads(p, CFG, ids, None, ids, "fp32", drop=plan)
    assert abs(float(loss) - float(g["fp32_loss"])) < 1e-5 * abs(float(g["fp32_loss"]))
    _, logits = G.forward(p, CFG, ids, None, None, "fp32", drop=plan)
    assert rel_err(slim(logits.detach().numpy()), g["fp32_logits"]) < 1e-4
    for k in g.files:
        if k.startswith("fp32_grad/"):
            assert rel_err(slim(grads[k.split("/", 1)[1]].numpy()), g[k]) < 1e-4, k
    # and the masks matter: the eval-mode loss is a different number
    loss0, _ = G.forward(p, CFG, ids, None, ids, "fp32")
    assert abs(float(loss0) - float(loss)) > 1e-3
