"""CPU oracle for dropout: the mask functions of the HIP kernels restated in numpy, and torch.nn.Dropout's arithmetic
for a given mask.

THIS IS TEST INFRASTRUCTURE. Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may import
it. It is the checker, never the product.

What the reference does: GPT2LMHeadModel in training mode applies torch.nn.Dropout(0.1) to the summed embeddings
(GPT2Model.forward, `self.drop`), to the attention probabilities after the softmax (eager_attention_forward /
`attn_dropout`), and to the outputs of both residual projections (`resid_dropout`, GPT2MLP.dropout) —
transformers models/gpt2/modeling_gpt2.py, probabilities from configuration_gpt2.py:85-87, which
/root/reference/src/model/model.py:54-64 does not override; its adapters use lora_dropout 0.05 on the adapter input
(/root/reference/src/model/model.py:251-271, peft's Linear.forward: base(x) + B(A(dropout(x))) * scaling).
torch.nn.Dropout: y = x * mask / (1 - p), mask ~ Bernoulli(1 - p) per element, result rounded to x's dtype.

Parity status: the ARITHMETIC for a given mask is torch's (restated in `apply`); the mask STREAM is parity unpinned —
the reference draws its masks from torch's device generator, which no other implementation reproduces; only the
keep probability and independence can be (and are) checked statistically. The functions below define the stream
this build uses, independently of the kernels (pure integer arithmetic), so that tests can hand the very same masks
to a plain-torch model.
"""
from __future__ import annotations

import numpy as np
import torch

_M64 = (1 << 64) - 1


def fmix32(h: np.ndarray) -> np.ndarray:
    """32-bit murmur3 finaliser on uint32 arrays."""
    h = h.astype(np.uint32)
    h = h ^ (h >> np.uint32(16))
    h = (h * np.uint32(0x85EBCA6B)).astype(np.uint32)
    h = h ^ (h >> np.uint32(13))
    h = (h * np.uint32(0xC2B2AE35)).astype(np.uint32)
    return h ^ (h >> np.uint32(16))


def threshold(p: float) -> int:
    t = float(np.float32(p)) * 4294967296.0
    return 0xFFFFFFFF if t >= 4294967295.0 else int(t)


def site_key(seed: int, counter: int) -> int:
    """splitmix64 of seed + (counter + 1) * golden ratio: one 64-bit key per dropout site."""
    z = (int(seed) + (int(counter) + 1) * 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def keep_flat(key: int, n: int, p: float) -> torch.Tensor:
    """bool [n]: element i of a contiguous tensor is kept."""
    with np.errstate(over="ignore"):
        i = np.arange(n, dtype=np.uint64)
        lo, hi = (i & np.uint64(0xFFFFFFFF)).astype(np.uint32), (i >> np.uint64(32)).astype(np.uint32)
        r = fmix32(lo ^ np.uint32(key & 0xFFFFFFFF))
        r = fmix32(r + np.uint32(key >> 32) + hi * np.uint32(0x9E3779B1))
    return torch.from_numpy(r >= np.uint32(threshold(p)))


def keep_attn(key: int, B: int, H: int, S: int, p: float) -> torch.Tensor:
    """bool [B, H, S(query), S(key)]: attention probability (b, h, q, k) is kept."""
    with np.errstate(over="ignore"):
        bh = np.arange(B * H, dtype=np.uint32)[:, None]
        q = np.arange(S, dtype=np.uint32)[None, :]
        rowkey = fmix32((bh * np.uint32(0x9E3779B1) + q) ^ np.uint32(key & 0xFFFFFFFF)) + np.uint32(key >> 32)
        k = np.arange(S, dtype=np.uint32)[None, None, :]
        r = fmix32(rowkey[:, :, None] ^ k)
    return torch.from_numpy(r >= np.uint32(threshold(p))).view(B, H, S, S)


def apply(x: torch.Tensor, keep: torch.Tensor, p: float) -> torch.Tensor:
    """torch.nn.Dropout's training-mode arithmetic for a given mask, in x's dtype."""
    scale = float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))   # fp32 arithmetic, as the kernels compute it
    return (x.float() * keep.view(x.shape).float() * scale).to(x.dtype)


class DropPlan:
    """Masks of one training forward of the GPT-2 decoder: site 0 = summed embeddings, then per layer l the attention
    probabilities (3l + 1), the attention projection output (3l + 2) and the MLP projection output (3l + 3); the
    site counter of forward number `step` starts at step * (3 * n_layer + 1)."""

    def __init__(self, seed: int, step: int, n_layer: int, embd_p: float, attn_p: float, resid_p: float):
        self.seed, self.base = seed, step * (3 * n_layer + 1)
        self.embd_p, self.attn_p, self.resid_p = embd_p, attn_p, resid_p

    def key(self, site: int) -> int:
        return site_key(self.seed, self.base + site)
