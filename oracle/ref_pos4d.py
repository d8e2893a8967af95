"""CPU oracle for the reference's 4-D position model (notes as continuous [start, duration, pitch, velocity] rows).

THIS IS TEST INFRASTRUCTURE. Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may import
it. It is the checker, never the product.

Plain-torch restatement of the in-tree arithmetic of /root/reference, read as text; nothing is imported from it:

  rope4d_angles        src/model/modeling.py:498-545  (Qwen3Model.rot_pos_emb; theta fixed at 10000, 4 zero "NoPE"
                                                      frequencies per axis)
  rotate_pairs         src/model/modeling.py:87-98    (adjacent pairing (x0,x1) -> (-x1,x0))
  rotate / rotate_T    src/model/modeling.py:164-180  (apply_rotary_pos_emb, _1 and _T)
  attention block      src/model/modeling.py:252-311  (q/k norm, rotation of q, k AND v, inverse rotation of the
                                                      attention output before o_proj)
  constant input       src/model/modeling.py:389-394,426-428 (embed_vector [1,0,1,0,...] at every position)
  head + loss          src/model/modeling.py:603-660  (16-wide head, unit-normalised pairs, cosine loss)
  compute_encoding     src/model/modeling.py:105-141
  process_last_logits  generate_music.py:110-142
  generate_music       generate_music.py:146-179      (re-runs the whole prefix for every new note)

Parity pin: tests/golden/tiny_pos4d.npz was produced in the build container by tools/make_golden.py from the
reference's own src/model/modeling.py module (imported there with a stub for the absent `wandb` logger);
tests/test_oracle_golden.py checks the functions here against it.

mode "bf16" restates what autocast does to this network on a GPU (precision="bf16-mixed",
src/train_model_new.py:383): as for the token model, plus fp32 cos/sin tables, v rotated in fp32 from the bf16
projection and rounded at the attention input, the attention output rotated back in fp32 and rounded at o_proj, the
head output in bf16 and the pair norm in fp32. (CPU autocast, which made the bf16 numbers of the fixture, runs
torch.norm in bf16; `cpu_autocast_head=True` reproduces that.)
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from .ref_cpu import LMConfig, KVCache, rmsnorm, linear, causal_attention, _bf

HEAD_OUT = 16          # 8 (cos, sin) pairs: a fine and a coarse angle for each of the 4 axes
ROPE4D_THETA = 10000.0
ROPE4D_NOPE = 4


def init_params_pos4d(cfg: LMConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """N(0, 0.02) linears, unit norm gains; no token embedding, a [16, d] head, the constant input vector."""
    g = torch.Generator().manual_seed(seed)
    d, H, KVH, Dh, I = (cfg.hidden_size, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim,
                        cfg.intermediate_size)

    def lin(o, i):
        return torch.randn(o, i, generator=g) * 0.02

    p: Dict[str, torch.Tensor] = {"model.embed_vector": embed_vector(d)}
    for l in range(cfg.num_hidden_layers):
        pre = f"model.layers.{l}."
        p[pre + "input_layernorm.weight"] = torch.ones(d)
        p[pre + "self_attn.q_proj.weight"] = lin(H * Dh, d)
        p[pre + "self_attn.k_proj.weight"] = lin(KVH * Dh, d)
        p[pre + "self_attn.v_proj.weight"] = lin(KVH * Dh, d)
        p[pre + "self_attn.o_proj.weight"] = lin(d, H * Dh)
        p[pre + "self_attn.q_norm.weight"] = torch.ones(Dh)
        p[pre + "self_attn.k_norm.weight"] = torch.ones(Dh)
        p[pre + "post_attention_layernorm.weight"] = torch.ones(d)
        p[pre + "mlp.gate_proj.weight"] = lin(I, d)
        p[pre + "mlp.up_proj.weight"] = lin(I, d)
        p[pre + "mlp.down_proj.weight"] = lin(d, I)
    p["model.norm.weight"] = torch.ones(d)
    p["lm_head.weight"] = lin(HEAD_OUT, d)
    return p


def embed_vector(d: int) -> torch.Tensor:
    return (1 - torch.arange(d) % 2).float()


def rope4d_freqs(head_dim: int) -> torch.Tensor:
    pairs = head_dim // 8
    rot = pairs - ROPE4D_NOPE
    f = 1.0 / (ROPE4D_THETA ** (torch.arange(0, rot * 2, 2).float() / head_dim))
    return torch.cat([f, torch.zeros(ROPE4D_NOPE)], dim=-1)


def rope4d_angles(position_tensors: torch.Tensor, head_dim: int) -> torch.Tensor:
    """[B,S,4] fp32 -> angles [B,S,head_dim]: axis a owns elements [a*Dh/4, (a+1)*Dh/4), each frequency twice."""
    fb = rope4d_freqs(head_dim)
    out = []
    for a in range(4):
        out.append((position_tensors[:, :, a:a + 1] * fb).repeat_interleave(2, dim=-1))
    return torch.cat(out, dim=-1)


def rotate_pairs(x: torch.Tensor) -> torch.Tensor:
    return torch.stack([-x[..., 1::2], x[..., 0::2]], dim=-1).reshape_as(x)


def rotate(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """x [B,H,S,Dh], cos/sin [B,S,Dh] fp32 (never cast to x's dtype: bf16 x promotes to fp32)."""
    return x * cos.unsqueeze(1) + rotate_pairs(x) * sin.unsqueeze(1)


def rotate_T(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    return x * cos.unsqueeze(1) - rotate_pairs(x) * sin.unsqueeze(1)


def compute_encoding(pos: torch.Tensor) -> torch.Tensor:
    """[...,4] -> [...,8,2] (cos, sin) of a fine and a coarse angle per axis."""
    def fl(v, c):
        return v // c * c
    ang = []
    for a in range(4):
        x = pos[..., a:a + 1]
        if a < 2:
            ang += [torch.pi * ((x % 1) / 1), torch.pi * (fl(x, 1) / 32)]
        else:
            ang += [torch.pi * ((x % 12) / 12), torch.pi * (fl(x, 12) / 128)]
    ang = torch.cat(ang, dim=-1)
    return torch.stack((ang.cos(), ang.sin()), dim=-1)


def decoder_layer4d(h, p, pre, cfg: LMConfig, cos, sin, mode: str, cache: Optional[KVCache], layer: int):
    B, S, d = h.shape
    H, KVH, Dh = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    x = rmsnorm(h, p[pre + "input_layernorm.weight"], cfg.rms_norm_eps)
    q = linear(x, p[pre + "self_attn.q_proj.weight"], mode).view(B, S, H, Dh)
    k = linear(x, p[pre + "self_attn.k_proj.weight"], mode).view(B, S, KVH, Dh)
    v = linear(x, p[pre + "self_attn.v_proj.weight"], mode).view(B, S, KVH, Dh)
    q = rmsnorm(q, p[pre + "self_attn.q_norm.weight"], cfg.rms_norm_eps).transpose(1, 2)
    k = rmsnorm(k, p[pre + "self_attn.k_norm.weight"], cfg.rms_norm_eps).transpose(1, 2)
    v = v.transpose(1, 2)
    q, k, v = rotate(q, cos, sin), rotate(k, cos, sin), rotate(v, cos, sin)
    if mode == "bf16":
        q, k, v = _bf(q), _bf(k), _bf(v)
    q_offset = 0
    if cache is not None:
        if cache.k[layer] is not None:
            q_offset = cache.k[layer].shape[2]
            k = torch.cat([cache.k[layer], k], dim=2)
            v = torch.cat([cache.v[layer], v], dim=2)
        cache.k[layer], cache.v[layer] = k, v
    if mode == "bf16":
        a = _bf(causal_attention(q.float(), k.float(), v.float(), Dh ** -0.5, None, q_offset))
    else:
        a = causal_attention(q, k, v, Dh ** -0.5, None, q_offset)
    a = a.view(B, S, H, Dh).transpose(1, 2)
    a = rotate_T(a, cos, sin).transpose(1, 2).reshape(B, S, H * Dh)
    h = h + linear(a, p[pre + "self_attn.o_proj.weight"], mode)
    x = rmsnorm(h, p[pre + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
    g = linear(x, p[pre + "mlp.gate_proj.weight"], mode)
    u = linear(x, p[pre + "mlp.up_proj.weight"], mode)
    return h + linear(F.silu(g) * u, p[pre + "mlp.down_proj.weight"], mode)


def cosine_loss(pred: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """pred [B,S,8,2], labels [B,S,4]; rows whose 4 label columns are all != -100 count; no such row -> 0."""
    target = compute_encoding(labels)
    mask = (labels != -100).all(dim=-1)
    if not bool(mask.any()):
        return torch.tensor(0.0)
    cos_sim = (pred[mask] * target[mask]).sum(dim=-1)
    return (1 - cos_sim).mean()


def component_losses(pred: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """The eight values the reference logs as loss/s_0, s_1, d_0, d_1, p_0, p_1, v_0, v_1
    (/root/reference/src/model/modeling.py:646-655): column means of 1 - cos_sim over the rows that count; [8]."""
    target = compute_encoding(labels)
    mask = (labels != -100).all(dim=-1)
    if not bool(mask.any()):
        return torch.zeros(8)
    cos_sim = (pred[mask] * target[mask]).sum(dim=-1)
    return (1 - cos_sim).mean(dim=0)


def forward_pos4d(p: Dict[str, torch.Tensor], cfg: LMConfig, position_tensors: torch.Tensor,
                  labels: Optional[torch.Tensor] = None, mode: str = "fp32", cache: Optional[KVCache] = None,
                  cpu_autocast_head: bool = False):
    """Returns (loss or None, pred [B,S,8,2]). With a cache, position_tensors holds only the new rows."""
    B, S = position_tensors.shape[:2]
    if cache is not None and not cache.k:
        cache.k = [None] * cfg.num_hidden_layers
        cache.v = [None] * cfg.num_hidden_layers
    h = p["model.embed_vector"][None, None, :].expand(B, S, -1)
    ang = rope4d_angles(position_tensors.float(), cfg.head_dim)
    cos, sin = ang.cos(), ang.sin()
    for l in range(cfg.num_hidden_layers):
        h = decoder_layer4d(h, p, f"model.layers.{l}.", cfg, cos, sin, mode, cache, l)
    hN = rmsnorm(h, p["model.norm.weight"], cfg.rms_norm_eps)
    pairs = linear(hN, p["lm_head.weight"], mode).view(B, S, 8, 2)
    if mode == "bf16" and not cpu_autocast_head:
        pred = pairs.float() / (torch.norm(pairs.float(), dim=-1, keepdim=True) + 1e-8)
    else:
        pred = pairs / (torch.norm(pairs, dim=-1, keepdim=True) + 1e-8)
    loss = cosine_loss(pred, labels) if labels is not None else None
    return loss, pred


def loss_and_grads_pos4d(p, cfg, position_tensors, labels, mode="fp32"):
    q = {k: v.detach().clone().requires_grad_(k != "model.embed_vector") for k, v in p.items()}
    loss, _ = forward_pos4d(q, cfg, position_tensors, labels, mode)
    loss.backward()
    return loss.detach(), {k: v.grad for k, v in q.items() if v.grad is not None}


def process_last_logits(pairs: torch.Tensor) -> torch.Tensor:
    """pairs [8,2] (unit vectors) -> next row [4]: start delta, duration, pitch, velocity."""
    ang = (torch.atan2(pairs[..., 1], pairs[..., 0]) / torch.pi).clamp(0, 1)
    scale = (1, 32, 1, 32, 12, 128, 12, 128)
    pos = [ang[i] * scale[i] for i in range(8)]

    def rnd(v, c):
        return (v / c).round() * c
    f0 = rnd(pos[1], 1) + pos[0]
    f1 = rnd(pos[3], 1) + pos[2]
    f2 = rnd(pos[5], 12) + pos[4]
    f3 = rnd(pos[7], 12) + pos[6]
    return torch.stack([f0, f1, f2.round(), f3.round()])


def generate_music(p, cfg: LMConfig, first_row: torch.Tensor, total_length: int, mode: str = "fp32",
                   use_cache: bool = False, return_preds: bool = False):
    """first_row [1,1,4]. Returns generated [1+total_length, 4] (absolute start times). use_cache=False re-runs the
    prefix each step as the reference does; use_cache=True feeds only the newest row to a KV cache (same numbers up
    to summation order)."""
    generated = first_row.clone().float()
    cache = KVCache() if use_cache else None
    preds = []
    with torch.no_grad():
        for i in range(total_length):
            if use_cache:
                _, pred = forward_pos4d(p, cfg, generated[:, -1:, :], None, mode, cache)
                last = pred[0, 0]
            else:
                _, pred = forward_pos4d(p, cfg, generated, None, mode)
                last = pred[0, i]
            preds.append(last.float())
            nxt = process_last_logits(last.float())
            nxt[0] = nxt[0] + generated[0, -1, 0]
            generated = torch.cat([generated, nxt[None, None, :]], dim=1)
    if return_preds:
        return generated[0], torch.stack(preds, 0)
    return generated[0]
