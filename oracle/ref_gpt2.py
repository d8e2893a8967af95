"""CPU oracle for the reference's GPT-2 style decoder.

THIS IS TEST INFRASTRUCTURE. Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may import
it. It is the checker, never the product.

Plain-torch restatement, nothing imported from `transformers` or /root/reference. The reference builds
transformers' GPT2LMHeadModel (/root/reference/src/model/model.py:54-65; transformers pinned 4.57.1 by its uv.lock);
the functions below follow, read as text (TF = the installed transformers package):

  embeddings + blocks   TF/models/gpt2/modeling_gpt2.py  GPT2Model.forward (wte + wpe), GPT2Block.forward
                        (ln_1 -> attn -> residual, ln_2 -> mlp -> residual), GPT2Attention (c_attn split into q|k|v,
                        heads of n_embd/n_head, scale 1/sqrt(head_dim), causal), GPT2MLP (c_fc, act, c_proj)
  Conv1D                TF/pytorch_utils.py  Conv1D.forward: addmm(bias, x, weight), weight stored [in, out]
  activation            TF/activations.py  NewGELUActivation
  loss                  TF/loss/loss_utils.py:49-71 (ForCausalLMLoss), shared with oracle.ref_cpu
  init                  GPT2PreTrainedModel._init_weights (c_proj weights scaled by 1/sqrt(2 n_layer))

Parity pin: tests/golden/tiny_gpt2.npz, produced in the build container by tools/make_golden.py from the stock
GPT2LMHeadModel with all dropout probabilities set to 0; tests/test_oracle_gpt2.py checks the functions here
against it. The reference ships no tests or golden vectors of its own for this path.

mode "bf16" restates autocast on a GPU: LayerNorm in fp32 on the fp32 residual stream, every Conv1D with bf16
operands (bias included) and a bf16 result, attention on bf16 operands, the activation in fp32 from the bf16 c_fc
output, logits in bf16 and upcast for the loss.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .ref_cpu import LMConfig, KVCache, causal_attention, causal_lm_loss, _bf, warp_logits, sample_with_noise
from . import ref_dropout as RD

LN_EPS = 1e-5


def init_params_gpt2(cfg: LMConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    d, I, L = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    std_proj = 0.02 / math.sqrt(2 * L)

    def rnd(*shape, std=0.02):
        return torch.randn(*shape, generator=g) * std

    p = {"transformer.wte.weight": rnd(cfg.vocab_size, d),
         "transformer.wpe.weight": rnd(cfg.max_position_embeddings, d)}
    for l in range(L):
        pre = f"transformer.h.{l}."
        p[pre + "ln_1.weight"], p[pre + "ln_1.bias"] = torch.ones(d), torch.zeros(d)
        p[pre + "attn.c_attn.weight"], p[pre + "attn.c_attn.bias"] = rnd(d, 3 * d), torch.zeros(3 * d)
        p[pre + "attn.c_proj.weight"], p[pre + "attn.c_proj.bias"] = rnd(d, d, std=std_proj), torch.zeros(d)
        p[pre + "ln_2.weight"], p[pre + "ln_2.bias"] = torch.ones(d), torch.zeros(d)
        p[pre + "mlp.c_fc.weight"], p[pre + "mlp.c_fc.bias"] = rnd(d, I), torch.zeros(I)
        p[pre + "mlp.c_proj.weight"], p[pre + "mlp.c_proj.bias"] = rnd(I, d, std=std_proj), torch.zeros(d)
    p["transformer.ln_f.weight"], p["transformer.ln_f.bias"] = torch.ones(d), torch.zeros(d)
    return p


def conv1d(x, w, b, mode):
    if mode == "bf16":
        return torch.addmm(_bf(b), _bf(x).reshape(-1, x.shape[-1]), _bf(w)).view(*x.shape[:-1], w.shape[1])
    return torch.addmm(b, x.reshape(-1, x.shape[-1]), w).view(*x.shape[:-1], w.shape[1])


def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def block(h, p, pre, cfg: LMConfig, mode, kpm, cache: Optional[KVCache], layer: int, drop=None):
    """drop: an oracle.ref_dropout.DropPlan (training-mode forward with its masks) or None (eval arithmetic)."""
    B, S, d = h.shape
    H = cfg.num_attention_heads
    Dh = d // H
    x = F.layer_norm(h, (d,), p[pre + "ln_1.weight"], p[pre + "ln_1.bias"], LN_EPS)
    qkv = conv1d(x, p[pre + "attn.c_attn.weight"], p[pre + "attn.c_attn.bias"], mode)
    q, k, v = [t.view(B, S, H, Dh).transpose(1, 2) for t in qkv.split(d, dim=2)]
    q_offset = 0
    if cache is not None:
        if cache.k[layer] is not None:
            q_offset = cache.k[layer].shape[2]
            k = torch.cat([cache.k[layer], k], dim=2)
            v = torch.cat([cache.v[layer], v], dim=2)
        cache.k[layer], cache.v[layer] = k, v
    keep, ap = None, 0.0
    if drop is not None and drop.attn_p > 0:
        keep, ap = RD.keep_attn(drop.key(3 * layer + 1), B, H, S, drop.attn_p), drop.attn_p
    if mode == "bf16":
        a = _bf(causal_attention(q.float(), k.float(), v.float(), Dh ** -0.5, kpm, q_offset, keep, ap))
    else:
        a = causal_attention(q, k, v, Dh ** -0.5, kpm, q_offset, keep, ap)

    def resid_drop(y, site):
        if drop is None or drop.resid_p == 0:
            return y
        return RD.apply(y, RD.keep_flat(drop.key(site), y.numel(), drop.resid_p), drop.resid_p)

    h = h + resid_drop(conv1d(a, p[pre + "attn.c_proj.weight"], p[pre + "attn.c_proj.bias"], mode), 3 * layer + 2)
    x = F.layer_norm(h, (d,), p[pre + "ln_2.weight"], p[pre + "ln_2.bias"], LN_EPS)
    f = conv1d(x, p[pre + "mlp.c_fc.weight"], p[pre + "mlp.c_fc.bias"], mode)
    act = gelu_new(f.float()) if mode == "bf16" else gelu_new(f)
    return h + resid_drop(conv1d(act, p[pre + "mlp.c_proj.weight"], p[pre + "mlp.c_proj.bias"], mode), 3 * layer + 3)


def forward_hidden(p, cfg: LMConfig, input_ids, attention_mask=None, mode="fp32", cache: Optional[KVCache] = None,
                   drop=None):
    B, S = input_ids.shape
    past = cache.length() if cache is not None else 0
    if cache is not None and not cache.k:
        cache.k = [None] * cfg.num_hidden_layers
        cache.v = [None] * cfg.num_hidden_layers
    pos = torch.arange(past, past + S)
    h = F.embedding(input_ids, p["transformer.wte.weight"]) + p["transformer.wpe.weight"][pos][None]
    if drop is not None and drop.embd_p > 0:
        h = RD.apply(h, RD.keep_flat(drop.key(0), h.numel(), drop.embd_p), drop.embd_p)
    kpm = attention_mask if (attention_mask is not None and not bool(attention_mask.bool().all())) else None
    for l in range(cfg.num_hidden_layers):
        h = block(h, p, f"transformer.h.{l}.", cfg, mode, kpm, cache, l, drop)
    return F.layer_norm(h, (cfg.hidden_size,), p["transformer.ln_f.weight"], p["transformer.ln_f.bias"], LN_EPS)


def forward(p, cfg: LMConfig, input_ids, attention_mask=None, labels=None, mode="fp32", cache=None, drop=None):
    hN = forward_hidden(p, cfg, input_ids, attention_mask, mode, cache, drop)
    W = p["transformer.wte.weight"]
    logits = F.linear(_bf(hN), _bf(W)) if mode == "bf16" else F.linear(hN, W)
    loss = causal_lm_loss(logits, labels) if labels is not None else None
    return loss, logits


def loss_and_grads(p, cfg, input_ids, attention_mask, labels, mode="fp32", drop=None):
    q = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    loss, _ = forward(q, cfg, input_ids, attention_mask, labels, mode, drop=drop)
    loss.backward()
    return loss.detach(), {k: v.grad for k, v in q.items()}


def generate(p, cfg: LMConfig, input_ids, max_new_tokens, do_sample=False, temperature=1.0, top_k=0, top_p=1.0,
             noise=None, mode="fp32"):
    cache = KVCache()
    ids, cur = input_ids, input_ids
    for t in range(max_new_tokens):
        with torch.no_grad():
            _, logits = forward(p, cfg, cur, None, None, mode, cache)
        logits = logits[:, -1, :].float()
        if do_sample:
            nxt = sample_with_noise(warp_logits(logits, temperature, top_k, top_p), noise[t])
        else:
            nxt = logits.argmax(dim=-1)
        ids = torch.cat([ids, nxt[:, None]], dim=1)
        cur = nxt[:, None]
    return ids
