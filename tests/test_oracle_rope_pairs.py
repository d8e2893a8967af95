"""Adjacent-pair rotary embedding (rope_interleaved) of the token models, pinned through an identity: a split-half
model whose q/k projection rows (and q/k norm gains) are permuted [0, Dh/2, 1, Dh/2 + 1, ...] within each head is the
same function under the adjacent pairing. So the oracle with rope_interleaved=True on the permuted weights must
reproduce the fixtures made from the stock transformers classes (tests/golden/tiny_qwen3.npz, tiny_llama.npz)."""
import dataclasses

import pytest
import torch

from oracle import ref_cpu as R
from golden_util import CASES, load, tiny_params, slim, rel_err, to_interleaved, rows_from_interleaved


@pytest.mark.parametrize("case", list(CASES))
def test_permuted_model_reproduces_the_fixture(case):
    cfg = CASES[case]
    cfg_il = dataclasses.replace(cfg, rope_interleaved=True)
    g = load(case)
    p = to_interleaved(tiny_params(cfg), cfg)
    ids = torch.from_numpy(g["input_ids"])
    loss, grads = R.loss_and_grads(p, cfg_il, ids, None, ids, "fp32")
    assert abs(float(loss) - float(g["fp32_loss"])) < 1e-5 * abs(float(g["fp32_loss"]))
    _, logits = R.forward(p, cfg_il, ids, None, None, "fp32")
    assert rel_err(slim(logits.numpy()), g["fp32_logits"]) < 1e-4
    for k in g.files:
        if k.startswith("fp32_grad/"):
            name = k.split("/", 1)[1]
            got = grads[name]
            if name.endswith(("q_proj.weight", "k_proj.weight", "q_norm.weight", "k_norm.weight")):
                got = rows_from_interleaved(got, cfg)
            assert rel_err(slim(got.numpy()), g[k]) < 1e-4, name
    # the plain pairing on the permuted weights is a different function
    _, wrong = R.forward(p, cfg, ids, None, None, "fp32")
    assert rel_err(slim(wrong.numpy()), g["fp32_logits"]) > 1e-2


def test_greedy_tokens_of_the_permuted_model():
    cfg = CASES["tiny_llama"]
    cfg_il = dataclasses.replace(cfg, rope_interleaved=True)
    g = load("tiny_llama")
    p = to_interleaved(tiny_params(cfg), cfg)
    want = torch.from_numpy(g["greedy_tokens"])
    out = R.generate(p, cfg_il, want[:, :8], 24, mode="fp32")
    assert torch.equal(out, want[:, :32])
