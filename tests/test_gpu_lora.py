"""Low-rank adapters against the plain model with merged weights (the plain path is pinned by the golden fixtures):
same loss, adapter gradients = the merged model's weight gradients projected on the adapters."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SITES = {"qkv": "self_attn.qkv_proj", "o": "self_attn.o_proj", "gate": "mlp.gate_proj", "up": "mlp.up_proj",
         "down": "mlp.down_proj"}


def _cfg(kind="qwen"):
    from mgen.config import LMConfig
    if kind == "qwen":
        return LMConfig(vocab_size=1000, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                        num_attention_heads=2, num_key_value_heads=2, head_dim=128)
    return LMConfig(vocab_size=1031, hidden_size=256, intermediate_size=768, num_hidden_layers=2,
                    num_attention_heads=4, num_key_value_heads=4, head_dim=64, rope_theta=10000.0,
                    tie_word_embeddings=False, qk_norm=False)


def _base_weight(m, l, site):
    p = f"model.layers.{l}."
    sd = dict(m.named_parameters())
    if site == "qkv":
        return torch.cat([sd[p + f"self_attn.{x}_proj.weight"] for x in "qkv"], 0)
    return sd[p + SITES[site] + ".weight"]


def _set_base_weight(m, l, site, W):
    p = f"model.layers.{l}."
    sd = dict(m.named_parameters())
    with torch.no_grad():
        if site == "qkv":
            o = 0
            for x in "qkv":
                w = sd[p + f"self_attn.{x}_proj.weight"]
                w.copy_(W[o:o + w.shape[0]])
                o += w.shape[0]
        else:
            sd[p + SITES[site] + ".weight"].copy_(W)


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


@pytest.mark.parametrize("kind", ["qwen", "llama"])
def test_adapters_match_the_merged_plain_model(kind):
    from mgen import MidiLM
    from oracle import ref_lora as R
    cfg = _cfg(kind)
    m1 = MidiLM(cfg, device="cuda:0", seed=0)
    lo = m1.to_lora(r=32, lora_alpha=64, seed=1)
    assert lo.scale == 2.0 and all(not p.requires_grad for p in m1.parameters())
    ad = dict(lo.named_parameters())
    # B starts at zero (adapters are a no-op at first): give it values so that every product is exercised
    g = torch.Generator(device="cuda").manual_seed(3)
    with torch.no_grad():
        for k, v in ad.items():
            if "lora_B" in k:
                v.normal_(0.0, 0.02, generator=g)
    ids = torch.randint(3, cfg.vocab_size, (2, 128), generator=torch.Generator().manual_seed(4)).cuda()
    lab = ids.clone()
    lab[1, 100:] = -100
    base_before = m1.flat_param.clone()
    loss1 = m1(input_ids=ids, labels=lab).loss
    loss1.backward()
    assert torch.equal(m1.flat_param, base_before)

    m2 = MidiLM(cfg, device="cuda:0", seed=0)
    for l in range(cfg.num_hidden_layers):
        for site, nm in SITES.items():
            A = ad[f"model.layers.{l}.{nm}.lora_A.weight"]
            B = ad[f"model.layers.{l}.{nm}.lora_B.weight"]
            _set_base_weight(m2, l, site, R.merged_weight(_base_weight(m2, l, site), A, B, lo.scale).float())
    m2.zero_grad()
    loss2 = m2(input_ids=ids, labels=lab).loss
    loss2.backward()
    assert abs(float(loss1) - float(loss2)) < 2e-3 * abs(float(loss2))
    grads2 = {k: p.grad for k, p in m2.named_parameters()}
    for l in range(cfg.num_hidden_layers):
        for site, nm in SITES.items():
            A = ad[f"model.layers.{l}.{nm}.lora_A.weight"]
            B = ad[f"model.layers.{l}.{nm}.lora_B.weight"]
            p = f"model.layers.{l}."
            if site == "qkv":
                dW = torch.cat([grads2[p + f"self_attn.{x}_proj.weight"] for x in "qkv"], 0)
            else:
                dW = grads2[p + nm + ".weight"]
            dA, dB = R.adapter_grads_from_merged(dW, A, B, lo.scale)
            gA = lo.layers[l][site].gA
            gB = lo.layers[l][site].gB
            assert rel(gA, dA) < 3e-2, (l, site, "A")
            assert rel(gB, dB) < 3e-2, (l, site, "B")
    # logits of the adapted model = logits of the merged plain model
    with torch.no_grad():
        lg1 = m1(input_ids=ids).logits.float()
        lg2 = m2(input_ids=ids).logits.float()
    assert rel(lg1, lg2) < 2e-2

    # folding the adapters in gives the merged model, and decoding works again
    with pytest.raises(RuntimeError):
        m1.generate(ids[:, :8], max_new_tokens=4)
    m1.merge_lora()
    assert m1.lora is None and all(p.requires_grad for p in m1.parameters())
    with torch.no_grad():
        lg3 = m1(input_ids=ids).logits.float()
    assert rel(lg3, lg2) < 2e-2
    out = m1.generate(ids[:, :8], max_new_tokens=4)
    assert out.shape == (2, 12)


def test_adapter_training_leaves_the_base_alone_and_learns():
    from mgen import MidiLM
    from mgen.trainer import Trainer
    cfg = _cfg("qwen")
    m = MidiLM(cfg, device="cuda:0", seed=0)
    lo = m.to_lora()
    ids = torch.randint(3, cfg.vocab_size, (2, 128), generator=torch.Generator().manual_seed(4)).cuda()
    base = m.flat_param.clone()
    a0 = lo.flat_param.clone()
    with torch.no_grad():
        first = float(m(input_ids=ids, labels=ids).loss)
        plain = float(MidiLM(cfg, device="cuda:0", seed=0)(input_ids=ids, labels=ids).loss)
    assert abs(first - plain) < 1e-6 * plain          # B = 0: the adapted model starts as the base model
    tr = Trainer(m, lr=2e-3, num_epochs=1, steps_per_epoch=100, warmup=1, max_grad_norm=1.0)
    losses = [float(tr.training_step({"input_ids": ids, "labels": ids})) for _ in range(12)]
    assert torch.equal(m.flat_param, base)
    assert not torch.equal(lo.flat_param, a0)
    assert losses[-1] < losses[0] - 0.05
    sd = lo.state_dict()
    assert len(sd) == cfg.num_hidden_layers * 10 and sd["model.layers.0.mlp.up_proj.lora_B.weight"].shape == (512, 32)
    lo2 = MidiLM(cfg, device="cuda:0", seed=0).to_lora()
    lo2.load_state_dict(sd)
    assert torch.equal(lo2.flat_param, lo.flat_param)
