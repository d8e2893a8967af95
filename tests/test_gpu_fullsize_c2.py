"""The headline workload at its literal size: the 14-layer default decoder, batch 32 x 2048 tokens, bf16-mixed, seed 0,
token ids uniform in [3, 30000) (what bench.py times). The CPU oracle cannot run this size in seconds, so the chain to it
is: fp32 parity mode == transformers fixtures at 1e-4 (tests/test_gpu_fp32.py, incl. the full-size decoder at 4 x 1024)
and, here, bf16-mixed against that fp32 mode on the same 65 536 tokens. This is also the only size at which the
lm_head runs more than one chunk (the second chunk's accumulating weight-gradient launch), the register-panel and
output-stationary GEMM kernels see their real shapes, and the grouped weight gradients run with 14 layers.

Reference: the training step of /root/reference/src/train.py:61-72 (MidiQwen, bf16-mixed)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

B, S = 32, 2048


def _ids():
    g = torch.Generator().manual_seed(1234)
    return torch.randint(3, 30000, (B, S), generator=g).cuda()


def _slices(model):
    """The gradient slices the fixtures of the small models pin: layer-0 q_proj, last down_proj, final norm,
    1000 embedding rows (every 30th of the first 30 000: rows that received gradient from both the lookup and the tied
    head) plus 64 rows no token touches (head gradient only)."""
    L = model.cfg.num_hidden_layers
    emb = model.model.embed_tokens.weight.grad
    return {
        "q_proj.0": model.model.layers[0].self_attn.q_proj.weight.grad.clone(),
        f"down_proj.{L - 1}": model.model.layers[L - 1].mlp.down_proj.weight.grad.clone(),
        "norm": model.model.norm.weight.grad.clone(),
        "embed[::30]": emb[0:30000:30].clone(),
        "embed[100000:100064]": emb[100000:100064].clone(),
    }


def _step(model, ids):
    model.zero_grad()
    loss = model(input_ids=ids, labels=ids).loss
    loss.backward()
    torch.cuda.synchronize()
    return loss.detach().clone()


@pytest.fixture(scope="module")
def bf16_run():
    from mgen import MidiLM, default_decoder
    cfg = default_decoder()
    assert (cfg.num_hidden_layers, cfg.vocab_size, cfg.lm_head_chunk_tokens) == (14, 151936, 32768)
    model = MidiLM(cfg, device="cuda:0", seed=0)
    ids = _ids()
    loss = _step(model, ids)
    out = {"model": model, "ids": ids, "loss": loss, "slices": _slices(model), "flat": model.flat_grad.clone()}
    yield out
    out.clear()
    del model
    torch.cuda.empty_cache()


def test_bf16_mixed_against_fp32_mode(bf16_run):
    """Loss within 2e-3 (relative) and the gradient slices within 2e-2 (max-norm relative) of the fp32 parity mode on
    the same batch and the same weights."""
    from mgen import MidiLM, default_decoder
    ref = MidiLM(default_decoder(), device="cuda:0", seed=None, compute_dtype=torch.float32)
    with torch.no_grad():
        ref.flat_param.copy_(bf16_run["model"].flat_param)
    ref._shadow_version = -1
    loss32 = _step(ref, bf16_run["ids"])
    want = _slices(ref)
    del ref
    torch.cuda.empty_cache()
    got_loss = float(bf16_run["loss"])
    assert abs(got_loss - float(loss32)) < 2e-3 * abs(float(loss32)), (got_loss, float(loss32))
    # a randomly initialised decoder: ln V + d * 0.02^2 / 2
    assert abs(float(loss32) - 12.03) < 0.05, float(loss32)
    for k, w in want.items():
        g = bf16_run["slices"][k]
        err = float((g - w).abs().max() / w.abs().max())
        print(f"{k}: rel err {err:.3e}")
        assert err < 2e-2, (k, err)


def test_loss_bit_identical_on_the_tile_loop_kernels(bf16_run):
    """MGEN_GEMM_RS=0 MGEN_GEMM_OS=0 route every product through the tile-loop kernel (the one the small-model
    fixtures exercise): same MFMA order over the reduction, so the loss is bit-identical and the gradients agree up to
    the summation order of the weight-gradient atomics."""
    model, ids = bf16_run["model"], bf16_run["ids"]
    os.environ["MGEN_GEMM_RS"] = "0"
    os.environ["MGEN_GEMM_OS"] = "0"
    try:
        loss = _step(model, ids)
        flat = model.flat_grad.clone()
    finally:
        os.environ.pop("MGEN_GEMM_RS", None)
        os.environ.pop("MGEN_GEMM_OS", None)
    assert torch.equal(loss, bf16_run["loss"]), (float(loss), float(bf16_run["loss"]))
    ref = bf16_run["flat"]
    err = float((flat - ref).abs().max() / ref.abs().max())
    assert err < 1e-5, err


def test_lm_head_chunks_of_8192_tokens(bf16_run):
    """Eight lm_head chunks instead of two: the accumulating weight-gradient launches after the first chunk, the chunk
    offsets of the cross entropy and of the input gradient. Each token's logits, loss term and gradients are the
    same numbers; only the order in which fp32 partial sums meet differs."""
    model, ids = bf16_run["model"], bf16_run["ids"]
    model.cfg.lm_head_chunk_tokens = 8192
    try:
        loss = _step(model, ids)
        flat = model.flat_grad.clone()
    finally:
        model.cfg.lm_head_chunk_tokens = 32768
    assert abs(float(loss) - float(bf16_run["loss"])) < 1e-6 * abs(float(loss))
    ref = bf16_run["flat"]
    err = float((flat - ref).abs().max() / ref.abs().max())
    assert err < 1e-5, err


def test_one_optimizer_step_learns(bf16_run):
    """Clip + AdamW on the full-size gradients: finite norm, and the same batch has a lower loss afterwards."""
    import math
    from mgen.trainer import Trainer
    model, ids = bf16_run["model"], bf16_run["ids"]
    tr = Trainer(model, lr=1e-3, num_epochs=1, steps_per_epoch=100, warmup=1, max_grad_norm=1.0)
    l0 = float(tr.training_step({"input_ids": ids, "labels": ids}))
    norm = float(tr.opt.norm_and_coef[0])
    assert math.isfinite(norm) and norm > 0
    for _ in range(3):
        l1 = float(tr.training_step({"input_ids": ids, "labels": ids}))
    assert abs(l0 - float(bf16_run["loss"])) < 1e-6 * l0
    assert l1 < l0 - 0.05, (l0, l1)
