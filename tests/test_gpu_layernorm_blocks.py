"""The decoder block with LayerNorm (gain + bias) instead of RMSNorm and adjacent-pair RoPE: the recalled shape of the
Aria block behind /root/reference/src/model/model.py:191,239,263-269. The checkpoint's remote code is not available
offline, so there is no reference fixture: parity unpinned against the real Aria; checked against the CPU oracle's
restatement (torch.nn.functional.layer_norm in fp32) - fp32 mode at 1e-4, bf16-mixed at 2e-2, greedy decoding through
the KV cache token for token."""
import pytest
import torch

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu


def _setup(seed=3):
    from mgen import LMConfig
    ocfg = R.LMConfig(vocab_size=1031, hidden_size=256, intermediate_size=512, num_hidden_layers=3,
                      num_attention_heads=4, num_key_value_heads=4, head_dim=64, qk_norm=False, rope_interleaved=True,
                      tie_word_embeddings=False, rope_theta=10000.0, norm_kind="layer")
    cfg = LMConfig(**{k: getattr(ocfg, k) for k in R.LMConfig.__dataclass_fields__})
    params = R.init_params(ocfg, seed=seed)
    g = torch.Generator().manual_seed(seed)
    for k in params:                       # gains and biases away from their initial 1 / 0
        if k.endswith("norm.weight"):
            params[k] = 1 + 0.2 * torch.randn(params[k].shape, generator=g)
        elif k.endswith(".bias"):
            params[k] = 0.1 * torch.randn(params[k].shape, generator=g)
    ids = torch.randint(3, ocfg.vocab_size, (2, 77), generator=g)
    labels = ids.clone()
    labels[0, :9] = -100
    return ocfg, cfg, params, ids, labels


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("mode,tol_loss,tol_grad", [("fp32", 1e-4, 1e-4), ("bf16", 2e-3, 2e-2)])
def test_loss_and_gradients_match_the_oracle(mode, tol_loss, tol_grad):
    from mgen import MidiLM
    ocfg, cfg, params, ids, labels = _setup()
    m = MidiLM(cfg, device="cuda:0", seed=None,
               compute_dtype=torch.float32 if mode == "fp32" else torch.bfloat16)
    assert "model.layers.0.input_layernorm.bias" in dict(m.named_parameters()) and "model.norm.bias" in m.state_dict()
    m.load_state_dict(params)
    m.zero_grad()
    loss = m(input_ids=ids.cuda(), labels=labels.cuda()).loss
    loss.backward()
    ref_loss, ref_grads = R.loss_and_grads(params, ocfg, ids, None, labels, mode)
    assert abs(float(loss) - float(ref_loss)) < tol_loss * abs(float(ref_loss)), (float(loss), float(ref_loss))
    got = dict(m.named_parameters())
    for k in ("model.layers.0.input_layernorm.weight", "model.layers.0.input_layernorm.bias",
              "model.layers.2.post_attention_layernorm.bias", "model.norm.weight", "model.norm.bias",
              "model.layers.0.self_attn.q_proj.weight", "model.layers.2.mlp.down_proj.weight", "lm_head.weight",
              "model.embed_tokens.weight"):
        assert _rel(got[k].grad, ref_grads[k]) < tol_grad, k


def test_initial_parameters():
    from mgen import MidiLM
    _, cfg, _, _, _ = _setup()
    m = MidiLM(cfg, device="cuda:0", seed=0)
    assert bool((m.model.norm.bias == 0).all()) and bool((m.model.layers[1].input_layernorm.weight == 1).all())


def test_greedy_decoding_through_the_cache():
    from mgen import MidiLM
    ocfg, cfg, params, ids, _ = _setup(seed=5)
    m = MidiLM(cfg, device="cuda:0", seed=None)
    m.load_state_dict(params)
    prompt = ids[:, :8]
    out = m.generate(prompt.cuda(), max_new_tokens=24, do_sample=False)
    nocache = m.generate(prompt.cuda(), max_new_tokens=24, do_sample=False, use_cache=False)
    ref = R.generate(params, ocfg, prompt, 24, mode="bf16")
    assert out.shape == (2, 32)
    agree = (out.cpu() == ref).float().mean()
    # (bf16 near-ties may pick another token once and the continuations part ways: the teacher-forced check below
    # is the strict one)
    assert float(agree) > 0.9 or torch.equal(out[:, :12].cpu(), ref[:, :12])
    assert torch.equal(out, nocache) or float((out == nocache).float().mean()) > 0.9
    # teacher forcing: logits of the cached path against the oracle's on the oracle's own tokens
    with torch.no_grad():
        lg = m(input_ids=ref.cuda()).logits.float().cpu()
    with torch.no_grad():
        _, want = R.forward(params, ocfg, ref, mode="bf16")
    assert _rel(lg, want.float()) < 2e-2
