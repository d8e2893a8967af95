"""generate() on prompts of different lengths batched with LEFT padding (attention_mask rows of zeros then ones).

fp32 parity mode: tokens equal to what transformers' Qwen3ForCausalLM.generate produced for the same padded batch
(tests/golden/leftpad_qwen3.npz: greedy, sampled with injected noise, EOS bookkeeping). bf16 cached decode: every
stream must produce what it produces alone from its unpadded prompt, also when the batch is decoded in two parts."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R
from golden_util import TINY_QWEN3, load, tiny_params


def _cfg(ocfg, **kw):
    from mgen.config import LMConfig
    return LMConfig(**{k: getattr(ocfg, k) for k in R.LMConfig.__dataclass_fields__}, **kw)


@pytest.fixture(scope="module")
def fx():
    return load("leftpad_qwen3")


@pytest.fixture(scope="module")
def tiny32():
    from mgen import MidiLM
    m = MidiLM(_cfg(TINY_QWEN3), device="cuda:0", seed=None, compute_dtype=torch.float32)
    m.load_state_dict(tiny_params(TINY_QWEN3))
    return m


@pytest.fixture(scope="module")
def tiny16():
    from mgen import MidiLM
    m = MidiLM(_cfg(TINY_QWEN3), device="cuda:0", seed=None)
    m.load_state_dict(tiny_params(TINY_QWEN3))
    return m


def test_fp32_greedy_tokens_equal_transformers(tiny32, fx):
    ids, mask = torch.from_numpy(fx["input_ids"]).cuda(), torch.from_numpy(fx["attention_mask"]).cuda()
    want = fx["greedy_tokens"]
    P = ids.shape[1]
    got = tiny32.generate(ids, attention_mask=mask, max_new_tokens=want.shape[1] - P, do_sample=False).cpu().numpy()
    assert got.shape == want.shape and np.array_equal(got[:, :P], want[:, :P])
    for b in range(ids.shape[0]):
        same = got[b, P:] == want[b, P:]
        if not same.all():     # only a top-2 gap at rounding level may part the two runs
            first_bad = int(np.argmin(same))
            assert fx["greedy_margin"][b, first_bad] < 1e-4, (b, first_bad, fx["greedy_margin"][b, first_bad])


def test_fp32_sampled_and_eos_tokens_equal_transformers(tiny32, fx):
    ids, mask = torch.from_numpy(fx["input_ids"]).cuda(), torch.from_numpy(fx["attention_mask"]).cuda()
    want = torch.from_numpy(fx["sample_tokens"])
    steps, B = want.shape[1] - ids.shape[1], ids.shape[0]
    noise = torch.empty(steps, B, TINY_QWEN3.vocab_size).exponential_(
        1.0, generator=torch.Generator().manual_seed(int(fx["noise_seed"])))
    got = tiny32.generate(ids, attention_mask=mask, max_new_tokens=steps, do_sample=True, temperature=0.97, top_p=0.95,
                          noise=noise.cuda()).cpu()
    assert torch.equal(got, want), (got != want).nonzero()[:4].tolist()
    want = torch.from_numpy(fx["eos_tokens"])
    got = tiny32.generate(ids, attention_mask=mask, max_new_tokens=steps, do_sample=False,
                          eos_token_id=int(fx["eos_id"]), pad_token_id=0).cpu()
    assert torch.equal(got, want), (got != want).nonzero()[:4].tolist()


def _rows_alone(model, ids, mask, steps):
    outs = []
    for b in range(ids.shape[0]):
        n = int(mask[b].sum())
        o, lg = model.generate(ids[b:b + 1, ids.shape[1] - n:].contiguous(), max_new_tokens=steps, do_sample=False,
                               return_step_logits=True)
        outs.append((o[0, n:].cpu(), lg[:, 0].cpu()))
    return outs


def _assert_rows_as_alone(tok, logits, alone, P):
    """tok [B, P + steps], logits [steps, B, V] of the padded batch against each row's own run: same tokens, and the
    same logits up to bf16 rounding, until a step whose top-2 gap is at that rounding level parts them."""
    for b, (tb, lb) in enumerate(alone):
        mine = tok[b, P:].cpu()
        same = (mine == tb).numpy()
        upto = len(same) if same.all() else int(np.argmin(same))
        if upto < len(same):
            top2 = torch.topk(logits[upto, b].cpu(), 2)[0]
            assert float(top2[0] - top2[1]) < 0.06, (b, upto, float(top2[0] - top2[1]))
        n = min(upto + 1, len(same))
        assert float((logits[:n, b].cpu() - lb[:n]).abs().max()) < 0.08, b


def test_bf16_cached_decode_rows_as_alone(tiny16, fx):
    ids, mask = torch.from_numpy(fx["input_ids"]).cuda(), torch.from_numpy(fx["attention_mask"]).cuda()
    steps = 24
    tok, logits = tiny16.generate(ids, attention_mask=mask, max_new_tokens=steps, do_sample=False,
                                  return_step_logits=True)
    assert torch.equal(tok[:, :ids.shape[1]], ids)
    _assert_rows_as_alone(tok, logits, _rows_alone(tiny16, ids, mask, steps), ids.shape[1])


def test_two_part_decode_with_ragged_prompts(tiny16):
    """66 streams with prompt lengths 1..9 of 9 columns: the two-part path (own position table per part) equals the
    one-part path, sampled and greedy, and a few rows equal their own unpadded runs."""
    B, P, steps = 66, 9, 20
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(3, TINY_QWEN3.vocab_size, (B, P), generator=g)
    n = torch.randint(1, P + 1, (B,), generator=g)
    n[0], n[1] = P, 1
    mask = (torch.arange(P)[None, :] >= (P - n)[:, None]).long()
    ids = ids * mask
    ids, mask = ids.cuda(), mask.cuda()

    def run(**kw):
        return tiny16.generate(ids, attention_mask=mask, max_new_tokens=steps, **kw)

    os.environ["MGEN_DECODE_PARTS"] = "1"
    try:
        one = run(do_sample=False)
        one_s = run(do_sample=True, top_k=0, top_p=0.9, seed=3)
    finally:
        os.environ.pop("MGEN_DECODE_PARTS", None)
    two, logits = run(do_sample=False, return_step_logits=True)
    assert torch.equal(one, two)
    assert torch.equal(one_s, run(do_sample=True, top_k=0, top_p=0.9, seed=3))
    pick = [0, 1, 32, 33, 65]
    alone = _rows_alone(tiny16, ids[pick], mask[pick], steps)
    _assert_rows_as_alone(two[pick], logits[:, pick], alone, P)


def test_mask_forms_that_are_refused_or_trivial(tiny16):
    ids = torch.randint(3, 1000, (2, 6), generator=torch.Generator().manual_seed(1)).cuda()
    ones = torch.ones(2, 6, dtype=torch.long).cuda()
    assert torch.equal(tiny16.generate(ids, attention_mask=ones, max_new_tokens=5),
                       tiny16.generate(ids, max_new_tokens=5))
    right = ones.clone()
    right[1, 4:] = 0
    with pytest.raises(ValueError, match="LEFT"):
        tiny16.generate(ids, attention_mask=right, max_new_tokens=5)
    hole = ones.clone()
    hole[0, 2] = 0
    with pytest.raises(ValueError, match="LEFT"):
        tiny16.generate(ids, attention_mask=hole, max_new_tokens=5)
    with pytest.raises(ValueError, match="LEFT"):
        tiny16.generate(ids, attention_mask=torch.zeros_like(ones), max_new_tokens=5)
    with pytest.raises(ValueError, match="shape"):
        tiny16.generate(ids, attention_mask=ones[:, :5], max_new_tokens=5)


def test_gpt2_left_padded_rows_as_alone_and_wide_batches():
    """GPT-2 (learned positions): a left-padded stream reads the position rows of its real tokens; a batch of 64
    streams or more decodes through the model's own one-token step."""
    from mgen.gpt2 import MidiGPT2LM
    from mgen.config import LMConfig
    cfg = LMConfig(arch="gpt2", vocab_size=1000, hidden_size=256, intermediate_size=1024, num_hidden_layers=2,
                   num_attention_heads=4, num_key_value_heads=4, head_dim=64, max_position_embeddings=256,
                   tie_word_embeddings=True, qk_norm=False)
    m = MidiGPT2LM(cfg, device="cuda:0", seed=2)
    m.eval()
    B, P, steps = 5, 10, 16
    g = torch.Generator().manual_seed(8)
    ids = torch.randint(3, 1000, (B, P), generator=g)
    n = torch.tensor([10, 3, 7, 1, 9])
    mask = (torch.arange(P)[None, :] >= (P - n)[:, None]).long()
    ids, mask = (ids * mask).cuda(), mask.cuda()
    tok, logits = m.generate(ids, attention_mask=mask, max_new_tokens=steps, do_sample=False, return_step_logits=True)
    _assert_rows_as_alone(tok, logits, _rows_alone(m, ids, mask, steps), P)
    wide = torch.randint(3, 1000, (64, 4), generator=g).cuda()
    out = m.generate(wide, max_new_tokens=6, do_sample=False)
    assert torch.equal(out[:3], m.generate(wide[:3].contiguous(), max_new_tokens=6, do_sample=False))
