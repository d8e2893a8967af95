"""CPU tests of the host-side logic that needs no kernels: parameter layout, state-dict naming, LR schedule,
gradient bucket ranges."""
import torch

from mgen.config import LMConfig, default_decoder, aria_style_llama
from mgen.model import param_layout, MidiLM, CausalLMOutput
from oracle import ref_cpu as R


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


def test_param_counts_match_reference_models():
    assert sum(_numel(s) for _, s in param_layout(default_decoder())) == 151_209_984
    assert sum(_numel(s) for _, s in param_layout(aria_style_llama())) == 658_487_808


def test_layout_names_match_oracle_and_fused_views_are_contiguous():
    cfg = LMConfig(vocab_size=256, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                   num_attention_heads=2, num_key_value_heads=1, head_dim=64)
    m = MidiLM(cfg, device="cpu", seed=None)
    ocfg = R.LMConfig(**{k: v for k, v in cfg.to_dict().items() if k in R.LMConfig.__dataclass_fields__})
    ref = R.init_params(ocfg)
    own = dict(m.named_parameters())
    assert set(own) == set(ref)
    for k in ref:
        assert tuple(own[k].shape) == tuple(ref[k].shape), k
    sd = m.state_dict()
    assert "lm_head.weight" in sd and sd["lm_head.weight"].data_ptr() == sd["model.embed_tokens.weight"].data_ptr()
    # parameters alias the flat buffer; q,k,v and gate,up are back to back
    m.load_state_dict(ref)
    v = m._layers[1]
    q = own["model.layers.1.self_attn.q_proj.weight"]
    k = own["model.layers.1.self_attn.k_proj.weight"]
    vv = own["model.layers.1.self_attn.v_proj.weight"]
    assert torch.equal(v.w_qkv, torch.cat([q, k, vv], 0))
    assert torch.equal(v.w_gu, torch.cat([own["model.layers.1.mlp.gate_proj.weight"],
                                          own["model.layers.1.mlp.up_proj.weight"]], 0))
    for name, p in own.items():
        assert p.grad is not None and p.grad.shape == p.shape
    # in-place edits through torch are seen by the shadow-refresh check
    ver = m.flat_param._version
    with torch.no_grad():
        q.mul_(2.0)
    assert m.flat_param._version != ver


def test_grad_ranges_cover_the_flat_buffer_once():
    cfg = LMConfig(vocab_size=256, hidden_size=64, intermediate_size=128, num_hidden_layers=3,
                   num_attention_heads=2, num_key_value_heads=2, head_dim=64, tie_word_embeddings=False)
    m = MidiLM(cfg, device="cpu", seed=None)
    r = sorted((s, e) for _, s, e in m.grad_ranges())
    assert r[0][0] == 0 and r[-1][1] == m._total
    for (s0, e0), (s1, e1) in zip(r, r[1:]):
        assert e0 == s1
    assert [t for t, _, _ in m.grad_ranges()] == ["head", 2, 1, 0, "embed"]


def test_schedule_matches_oracle_closed_form():
    from mgen.optim import get_optim
    cfg = LMConfig(vocab_size=64, hidden_size=64, intermediate_size=64, num_hidden_layers=1, num_attention_heads=1,
                   num_key_value_heads=1, head_dim=64)
    m = MidiLM(cfg, device="cpu", seed=None)
    try:
        opt, sched = get_optim(m, 5e-5, 12, 20, warmup=10)
    except Exception as e:  # the optimizer allocates device workspaces through the library
        import pytest
        pytest.skip(f"needs the kernel library: {e}")
    for t in range(30):
        assert abs(opt.param_groups[0]["lr"] - R.lr_at(t, 5e-5, 10, 240)) < 1e-18 + 1e-12 * 5e-5
        opt.step_count += 1
        sched.step()


def test_output_object_supports_attribute_and_index_access():
    o = CausalLMOutput(loss=torch.tensor(1.0), logits=torch.zeros(2))
    assert o[0] is o.loss and o[1] is o.logits
    o = CausalLMOutput(loss=None, logits=torch.zeros(2))
    assert o[0] is o.logits
