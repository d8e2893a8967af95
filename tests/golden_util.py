"""Helpers shared by the oracle and the GPU parity tests: load a golden fixture and rebuild its weights."""
import os

import numpy as np
import torch

from oracle import ref_cpu as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TINY_QWEN3 = R.LMConfig(vocab_size=1024, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                        num_attention_heads=2, num_key_value_heads=2, head_dim=128)
TINY_LLAMA = R.LMConfig(vocab_size=1031, hidden_size=256, intermediate_size=768, num_hidden_layers=2,
                        num_attention_heads=4, num_key_value_heads=4, head_dim=64, rms_norm_eps=1e-6,
                        rope_theta=10000.0, tie_word_embeddings=False, qk_norm=False)
CASES = {"tiny_qwen3": TINY_QWEN3, "tiny_llama": TINY_LLAMA}


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def tiny_params(cfg):
    """Same recipe as tools/make_golden.py: seed-0 init, norm gains perturbed with seed 7."""
    params = R.init_params(cfg, seed=0)
    g = torch.Generator().manual_seed(7)
    for k in params:
        if "norm" in k:
            params[k] = 1.0 + 0.1 * torch.randn(params[k].shape, generator=g)
    return params


TINY_POS4D = R.LMConfig(vocab_size=1, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                        num_attention_heads=2, num_key_value_heads=2, head_dim=128, tie_word_embeddings=False)


def tiny_params_pos4d(cfg=TINY_POS4D):
    from oracle import ref_pos4d as P
    params = P.init_params_pos4d(cfg, seed=0)
    g = torch.Generator().manual_seed(7)
    for k in params:
        if "norm" in k:
            params[k] = 1.0 + 0.1 * torch.randn(params[k].shape, generator=g)
    return params


TINY_GPT2 = R.LMConfig(vocab_size=1000, hidden_size=256, intermediate_size=1024, num_hidden_layers=2,
                       num_attention_heads=4, num_key_value_heads=4, head_dim=64, max_position_embeddings=1024,
                       tie_word_embeddings=True, qk_norm=False)


def tiny_params_gpt2(cfg=TINY_GPT2):
    from oracle import ref_gpt2 as G
    params = G.init_params_gpt2(cfg, seed=0)
    g = torch.Generator().manual_seed(7)
    for k in params:
        if ".ln_" in k or k.endswith(".bias"):
            params[k] = params[k] + 0.1 * torch.randn(params[k].shape, generator=g)
    return params


def checksum(params):
    return float(sum(v.double().abs().sum() for v in params.values()))


def slim(a):
    a = np.asarray(a)
    if a.ndim == 2 and a.size > 65536:
        return a[::4]
    if a.ndim == 3:
        return a[:, ::4, :]
    return a


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def pair_perm(Dh):
    """Row order that turns a split-half head into the adjacent-pair head with the same rotation pairs:
    new[2i] = old[i], new[2i + 1] = old[i + Dh/2]."""
    idx = torch.empty(Dh, dtype=torch.long)
    idx[0::2] = torch.arange(Dh // 2)
    idx[1::2] = torch.arange(Dh // 2) + Dh // 2
    return idx


def to_interleaved(params, cfg):
    """The adjacent-pair (rope_interleaved) twin of a split-half model: q/k projection rows and q/k norm gains
    permuted within each head; every other tensor unchanged. Both models compute the same function."""
    idx = pair_perm(cfg.head_dim)
    out = dict(params)
    for k, v in params.items():
        if k.endswith("q_proj.weight") or k.endswith("k_proj.weight"):
            H = v.shape[0] // cfg.head_dim
            out[k] = v.view(H, cfg.head_dim, -1)[:, idx, :].reshape(v.shape).contiguous()
        elif k.endswith("q_norm.weight") or k.endswith("k_norm.weight"):
            out[k] = v[idx].contiguous()
    return out


def rows_from_interleaved(t, cfg):
    """Inverse of the row permutation of to_interleaved for a q/k projection weight (or its gradient)."""
    t = torch.as_tensor(t)
    inv = torch.argsort(pair_perm(cfg.head_dim))
    if t.dim() == 1:
        return t[inv]
    H = t.shape[0] // cfg.head_dim
    return t.view(H, cfg.head_dim, -1)[:, inv, :].reshape(t.shape)
