"""Batches of 64 streams or more are decoded as two halves on two HIP streams. Every stream's tokens must be what
the one-part path produces: greedy, sampled from the Philox stream (row base of the second half), sampled from explicit
noise, with EOS / pad bookkeeping, with an odd number of streams, and with the per-step logits returned."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    from mgen import MidiLM, LMConfig
    cfg = LMConfig(vocab_size=2048, hidden_size=512, intermediate_size=2048, num_hidden_layers=2,
                   num_attention_heads=4, num_key_value_heads=2, head_dim=128)
    return MidiLM(cfg, device="cuda:0", seed=3)


def _both(fn):
    os.environ["MGEN_DECODE_PARTS"] = "1"
    try:
        one = fn()
    finally:
        os.environ.pop("MGEN_DECODE_PARTS", None)
    two = fn()
    torch.cuda.synchronize()
    return one, two


@pytest.mark.parametrize("B", [64, 65, 130])
def test_greedy_and_philox_sampling(model, B):
    g = torch.Generator().manual_seed(B)
    prompt = torch.randint(3, 2048, (B, 5), generator=g).cuda()
    one, two = _both(lambda: model.generate(prompt, max_new_tokens=24, do_sample=False))
    assert torch.equal(one, two)
    one, two = _both(lambda: model.generate(prompt, max_new_tokens=24, do_sample=True, temperature=0.9, top_k=0,
                                            top_p=0.9, seed=77))
    assert torch.equal(one, two)
    assert not torch.equal(one[:B // 2, 5:], one[B - B // 2:, 5:]), "the two halves drew the same noise"


def test_explicit_noise_and_step_logits(model):
    B, V, n = 64, 2048, 6
    g = torch.Generator().manual_seed(1)
    prompt = torch.randint(3, V, (B, 3), generator=g).cuda()
    noise = torch.empty(n, B, V).exponential_(1.0, generator=g).cuda()
    (o1, l1), (o2, l2) = _both(lambda: model.generate(prompt, max_new_tokens=n, do_sample=True, top_k=20, noise=noise,
                                                      return_step_logits=True))
    assert torch.equal(o1, o2)
    assert l1.shape == l2.shape == (n, B, V) and torch.equal(l1, l2)


def test_eos_and_pad_bookkeeping(model):
    B = 96
    g = torch.Generator().manual_seed(2)
    prompt = torch.randint(3, 2048, (B, 4), generator=g).cuda()
    free = model.generate(prompt, max_new_tokens=40, do_sample=True, temperature=1.0, top_k=0, seed=5)
    eos = int(free[3, 10])     # a token stream 3 produces early becomes the EOS id: rows finish at different steps
    one, two = _both(lambda: model.generate(prompt, max_new_tokens=40, do_sample=True, temperature=1.0, top_k=0, seed=5,
                                            eos_token_id=eos, pad_token_id=0, eos_check_every=4))
    assert torch.equal(one, two)
    row = two[3, 4:]
    first = int((row == eos).nonzero()[0])
    assert first <= 6 and bool((row[first + 1:] == 0).all())


@pytest.mark.parametrize("B", [3, 64, 130])
def test_layer_in_one_call_equals_launch_by_launch(model, B, monkeypatch):
    """mgen_decode_layer issues the same seven launches a decode layer makes one by one: same tokens, same logits."""
    from mgen import generation
    g = torch.Generator().manual_seed(10 + B)
    prompt = torch.randint(3, 2048, (B, 7), generator=g).cuda()
    assert generation._LAYER_IN_ONE_CALL
    o1, l1 = model.generate(prompt, max_new_tokens=12, do_sample=False, return_step_logits=True)
    monkeypatch.setattr(generation, "_LAYER_IN_ONE_CALL", False)
    o2, l2 = model.generate(prompt, max_new_tokens=12, do_sample=False, return_step_logits=True)
    assert torch.equal(o1, o2) and torch.equal(l1, l2)


def test_decode_layer_rejects_shapes_it_does_not_take():
    from mgen import ops
    from mgen._lib import lib
    assert lib().mgen_decode_layer_fits(128, 512, 8, 8, 128, 2048) == 1
    assert lib().mgen_decode_layer_fits(128, 768, 8, 8, 128, 2048) == 0      # hidden size not one 512 slice
    assert lib().mgen_decode_layer_fits(128, 512, 8, 8, 128, 1000) == 0      # intermediate size not whole slices
    x = torch.zeros(8, device="cuda")
    with pytest.raises(RuntimeError, match="shapes not taken"):
        ops.decode_layer([x.data_ptr()] * 17, x.data_ptr(), x.data_ptr(), x.data_ptr(),
                         [x.data_ptr(), x.data_ptr(), 4, 768, 8, 8, 128, 2048, 64], 1, 1e-6, 0)


def test_reserved_kv_cache_gives_the_same_tokens(model):
    """generate() inside a reserved cache (more streams and a longer row capacity than the call needs) equals
    generate() on a cache of its own, for a whole batch and for two parts; a call that does not fit allocates."""
    g = torch.Generator().manual_seed(21)
    for B in (3, 70):
        prompt = torch.randint(3, 2048, (B, 6), generator=g).cuda()
        want = model.generate(prompt, max_new_tokens=20, do_sample=True, top_k=0, top_p=0.9, seed=4)
        model.reserve_kv_cache(80, 64)
        try:
            got = model.generate(prompt, max_new_tokens=20, do_sample=True, top_k=0, top_p=0.9, seed=4)
            assert torch.equal(got, want)
            too_long = model.generate(prompt, max_new_tokens=70, do_sample=False)      # 76 > 64: own cache
            assert too_long.shape == (B, 76)
        finally:
            model.release_kv_cache()
    assert model._kv_reserved is None


@pytest.mark.parametrize("B", [2, 70])
def test_graph_replay_equals_launch_by_launch(model, B, monkeypatch):
    """Decode steps replayed from a captured HIP graph (positions, step index and EOS state in device buffers) give the
    tokens of the launch-by-launch loop: greedy, Philox sampling, EOS / pad bookkeeping with early stop, left-padded
    prompts, and a run long enough for the attention split count (hence the captured graph) to change."""
    from mgen import generation
    g = torch.Generator().manual_seed(40 + B)
    prompt = torch.randint(3, 2048, (B, 6), generator=g).cuda()
    mask = torch.ones(B, 6, dtype=torch.long)
    mask[1, :4] = 0
    mask = mask.cuda()
    free = model.generate(prompt, max_new_tokens=30, do_sample=True, top_k=0, top_p=0.9, seed=9)
    eos = int(free[0, 6 + 9])
    runs = [dict(max_new_tokens=40, do_sample=False),
            dict(max_new_tokens=40, do_sample=True, temperature=0.9, top_k=0, top_p=0.9, seed=9),
            dict(max_new_tokens=40, do_sample=True, top_k=0, top_p=0.9, seed=9, eos_token_id=eos, pad_token_id=0,
                 eos_check_every=4),
            dict(max_new_tokens=40, do_sample=True, top_k=20, top_p=0.9, seed=3, attention_mask=mask),
            dict(max_new_tokens=600 if B == 2 else 8, do_sample=False)]
    monkeypatch.setattr(generation, "_DECODE_GRAPH", True)
    with_graph = [model.generate(prompt, **kw) for kw in runs]
    monkeypatch.setattr(generation, "_DECODE_GRAPH", False)
    without = [model.generate(prompt, **kw) for kw in runs]
    for kw, a, b in zip(runs, with_graph, without):
        assert a.shape == b.shape and torch.equal(a, b), kw
    assert with_graph[2].shape[1] <= 6 + 40
