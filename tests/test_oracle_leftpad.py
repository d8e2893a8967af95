"""The CPU restatement's generation loop with LEFT-padded prompts against tokens transformers' Qwen3ForCausalLM.generate
produced for the same batch (tests/golden/leftpad_qwen3.npz, written by tools/make_golden.py --leftpad-only)."""
import numpy as np
import torch

import golden_util as G
from oracle import ref_cpu as R


def _setup():
    fx = G.load("leftpad_qwen3")
    cfg = G.TINY_QWEN3
    p = G.tiny_params(cfg)
    assert abs(G.checksum(p) - float(fx["param_checksum"])) < 1e-6 * float(fx["param_checksum"])
    return fx, cfg, p


def test_greedy_tokens_from_left_padded_prompts():
    fx, cfg, p = _setup()
    ids, mask = torch.from_numpy(fx["input_ids"]), torch.from_numpy(fx["attention_mask"])
    want = fx["greedy_tokens"]
    got = R.generate(p, cfg, ids, want.shape[1] - ids.shape[1], attention_mask=mask).numpy()
    margin = fx["greedy_margin"]
    for b in range(ids.shape[0]):
        # equal up to the first step whose top-2 gap is at rounding level (none in this fixture: min gap 5e-3)
        tie = np.nonzero(margin[b] < 1e-4)[0]
        upto = ids.shape[1] + (int(tie[0]) if len(tie) else margin.shape[1])
        assert np.array_equal(got[b, :upto], want[b, :upto]), b


def test_sampled_and_eos_runs_from_left_padded_prompts():
    fx, cfg, p = _setup()
    ids, mask = torch.from_numpy(fx["input_ids"]), torch.from_numpy(fx["attention_mask"])
    want = fx["sample_tokens"]
    steps = want.shape[1] - ids.shape[1]
    noise = torch.empty(steps, ids.shape[0], cfg.vocab_size).exponential_(
        1.0, generator=torch.Generator().manual_seed(int(fx["noise_seed"])))
    got = R.generate(p, cfg, ids, steps, do_sample=True, temperature=0.97, top_k=50, top_p=0.95, noise=noise,
                     attention_mask=mask).numpy()
    assert np.array_equal(got, want)
    want = fx["eos_tokens"]
    got = R.generate(p, cfg, ids, want.shape[1] - ids.shape[1], eos_token_id=int(fx["eos_id"]), pad_token_id=0,
                     attention_mask=mask).numpy()
    assert np.array_equal(got, want)
