"""CPU: the oracle's LayerNorm block variant against torch.nn.LayerNorm modules written out independently."""
import torch

from oracle import ref_cpu as R


def test_block_norm_is_nn_layernorm_in_fp32():
    g = torch.Generator().manual_seed(0)
    cfg = R.LMConfig(hidden_size=64, norm_kind="layer")
    ln = torch.nn.LayerNorm(64, eps=cfg.layer_norm_eps)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.3 * torch.randn(64, generator=g))
        ln.bias.copy_(0.2 * torch.randn(64, generator=g))
    p = {"n.weight": ln.weight.detach(), "n.bias": ln.bias.detach()}
    x = torch.randn(5, 7, 64, generator=g) * 3
    assert torch.allclose(R.block_norm(x, p, "n", cfg), ln(x), atol=1e-6)
    rcfg = R.LMConfig(hidden_size=64)
    assert torch.equal(R.block_norm(x, {"n.weight": p["n.weight"]}, "n", rcfg), R.rmsnorm(x, p["n.weight"], rcfg.rms_norm_eps))


def test_parameters_and_a_training_step_of_the_variant():
    cfg = R.LMConfig(vocab_size=97, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                     num_key_value_heads=2, head_dim=32, qk_norm=False, rope_interleaved=True, norm_kind="layer",
                     tie_word_embeddings=False)
    p = R.init_params(cfg, seed=1)
    assert "model.layers.1.post_attention_layernorm.bias" in p and "model.norm.bias" in p
    ids = torch.randint(3, 97, (2, 16), generator=torch.Generator().manual_seed(2))
    loss, grads = R.loss_and_grads(p, cfg, ids, None, ids, "fp32")
    assert torch.isfinite(loss) and float(grads["model.norm.bias"].abs().max()) > 0
    assert abs(float(loss) - 4.57) < 0.3        # ln 97 at a random initialisation
