"""CPU oracle: a plain-torch restatement of the reference's decoder-only LM hot path.

THIS IS TEST INFRASTRUCTURE. Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of
bench.py may import it. It is the checker, never the product: the product path lives in
symbolic-music-generation_amd/ and runs hand-written HIP kernels only.

Every op is written out explicitly with torch CPU tensor arithmetic; nothing from `transformers`
or from /root/reference is imported. The arithmetic of the reference lives in the third-party
package `transformers` (pinned 4.57.1 by the reference's uv.lock:2784-2785); the functions below
follow these sources, read as text (TF = the installed transformers package):

  rmsnorm            src/model/modeling.py:60-65      == TF/models/qwen3/modeling_qwen3.py:59-64
  rope_cos_sin       TF/models/qwen3/modeling_qwen3.py:126-137 (inv_freq :120-123)
  apply_rope         TF/models/qwen3/modeling_qwen3.py:140-170 (split-half pairing)
  attention          TF/models/qwen3/modeling_qwen3.py:185-207 (eager twin of the sdpa path)
  mlp                TF/models/qwen3/modeling_qwen3.py:81-83
  decoder layer      TF/models/qwen3/modeling_qwen3.py:294-323
  causal_lm_loss     TF/loss/loss_utils.py:49-71
  clip_grad_norm     torch/nn/utils/clip_grad.py (total L2 norm, coef = max_norm/(norm+1e-6) clamped to 1)
  adamw / lr schedule src/model/model.py:11-46
  sampling warpers   TF/generation/logits_process.py:301-303 (T), :590-595 (top-k), :527-539 (top-p)
  sampling loop      TF/generation/utils.py:2783-2941

Parity pin: tests/golden/*.npz were produced in the build container by tools/make_golden.py from the
stock `transformers` Qwen3ForCausalLM / LlamaForCausalLM modules (the reference builds exactly these through
AutoModelForCausalLM.from_config, src/model/model.py:136); tests/test_oracle_golden.py checks every function
here against them. The reference ships no tests or golden vectors of its own for this path.

The "bf16" mode restates what `precision="bf16-mixed"` (src/train.py:71) does to dtypes: fp32 parameters and
residual stream, bf16 rounding at every GEMM input/output, q/k-norm result rounded to bf16 before the fp32
gain, attention operands rounded to bf16, silu and gate*up in bf16, logits bf16 then upcast for the loss.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F


@dataclass
class LMConfig:
    """Hyper-parameters the hot path reads (mirror of the Qwen3Config fields used by
    src/model/model.py:126-136)."""
    vocab_size: int = 151936
    hidden_size: int = 512
    intermediate_size: int = 2048
    num_hidden_layers: int = 14
    num_attention_heads: int = 8
    num_key_value_heads: int = 8
    head_dim: int = 128
    max_position_embeddings: int = 4096
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1e6
    tie_word_embeddings: bool = True
    qk_norm: bool = True          # Qwen3: per-head RMSNorm on q and k; Llama: none
    rope_interleaved: bool = False  # rotation pairs (2i, 2i+1) instead of the split-half pairs (i, i + Dh/2)
    norm_kind: str = "rms"        # "layer": nn.LayerNorm (gain + bias, eps = layer_norm_eps) in place of every RMSNorm
    layer_norm_eps: float = 1e-5
    bos_token_id: int = 1
    eos_token_id: int = 2
    pad_token_id: int = 0


def init_params(cfg: LMConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """N(0, 0.02) linears/embedding, norm gains = 1 (HF `_init_weights`), HF key names."""
    g = torch.Generator().manual_seed(seed)
    p: Dict[str, torch.Tensor] = {}
    d, H, KVH, Dh, I = (cfg.hidden_size, cfg.num_attention_heads, cfg.num_key_value_heads,
                        cfg.head_dim, cfg.intermediate_size)

    def lin(o, i):
        return torch.randn(o, i, generator=g) * 0.02

    p["model.embed_tokens.weight"] = torch.randn(cfg.vocab_size, d, generator=g) * 0.02
    for l in range(cfg.num_hidden_layers):
        pre = f"model.layers.{l}."
        p[pre + "input_layernorm.weight"] = torch.ones(d)
        if cfg.norm_kind == "layer":
            p[pre + "input_layernorm.bias"] = torch.zeros(d)
            p[pre + "post_attention_layernorm.bias"] = torch.zeros(d)
        p[pre + "self_attn.q_proj.weight"] = lin(H * Dh, d)
        p[pre + "self_attn.k_proj.weight"] = lin(KVH * Dh, d)
        p[pre + "self_attn.v_proj.weight"] = lin(KVH * Dh, d)
        p[pre + "self_attn.o_proj.weight"] = lin(d, H * Dh)
        if cfg.qk_norm:
            p[pre + "self_attn.q_norm.weight"] = torch.ones(Dh)
            p[pre + "self_attn.k_norm.weight"] = torch.ones(Dh)
        p[pre + "post_attention_layernorm.weight"] = torch.ones(d)
        p[pre + "mlp.gate_proj.weight"] = lin(I, d)
        p[pre + "mlp.up_proj.weight"] = lin(I, d)
        p[pre + "mlp.down_proj.weight"] = lin(d, I)
    p["model.norm.weight"] = torch.ones(d)
    if cfg.norm_kind == "layer":
        p["model.norm.bias"] = torch.zeros(d)
    if not cfg.tie_word_embeddings:
        p["lm_head.weight"] = lin(cfg.vocab_size, d)
    return p


# --------------------------------------------------------------------------------------------
# elementary ops
# --------------------------------------------------------------------------------------------
def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    in_dtype = x.dtype
    xf = x.to(torch.float32)
    var = xf.pow(2).mean(-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    return w * xf.to(in_dtype)


def block_norm(h: torch.Tensor, p: Dict[str, torch.Tensor], name: str, cfg: "LMConfig") -> torch.Tensor:
    """The decoder block's norm `name` (.weight / .bias in p): RMSNorm, or torch.nn.LayerNorm evaluated in fp32 (what
    autocast does to layer_norm; the residual stream is fp32 in both precisions)."""
    if cfg.norm_kind == "layer":
        return F.layer_norm(h.to(torch.float32), (h.shape[-1],), p[name + ".weight"], p[name + ".bias"],
                            cfg.layer_norm_eps)
    return rmsnorm(h, p[name + ".weight"], cfg.rms_norm_eps)


def rope_inv_freq(head_dim: int, theta: float) -> torch.Tensor:
    return 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))


def rope_cos_sin(position_ids: torch.Tensor, head_dim: int, theta: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """position_ids [B,S] int64 -> cos, sin [B,S,head_dim] fp32 (emb = cat(freqs, freqs))."""
    inv = rope_inv_freq(head_dim, theta)
    freqs = position_ids[:, :, None].float() * inv[None, None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """x [B,H,S,Dh]; cos/sin [B,S,Dh]."""
    c = cos.unsqueeze(1).to(x.dtype)
    s = sin.unsqueeze(1).to(x.dtype)
    return x * c + rotate_half(x) * s


def apply_rope_interleaved(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """Adjacent-pair form: elements 2i, 2i+1 rotate by the angle of frequency i (cos/sin as from rope_cos_sin, whose
    first half holds the Dh/2 frequencies). The same rotation as apply_rope on head dims permuted
    [0, 2, 4, ..., 1, 3, 5, ...]; the pair arithmetic is that of the 4-D model's rotation
    (/root/reference/src/model/modeling.py:143-180)."""
    h = x.shape[-1] // 2
    c = cos[..., :h].unsqueeze(1).to(x.dtype)
    s = sin[..., :h].unsqueeze(1).to(x.dtype)
    x1, x2 = x[..., 0::2], x[..., 1::2]
    return torch.stack((x1 * c - x2 * s, x2 * c + x1 * s), dim=-1).flatten(-2)


def causal_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float,
                     key_padding_mask: Optional[torch.Tensor] = None, q_offset: int = 0,
                     drop_keep: Optional[torch.Tensor] = None, drop_p: float = 0.0) -> torch.Tensor:
    """q [B,H,Sq,Dh], k/v [B,KVH,Sk,Dh] -> [B,Sq,H*Dh]. Softmax in fp32. Query i sits at absolute position
    q_offset+i and may see keys 0..q_offset+i. key_padding_mask [B,Sk] (1 = real token). drop_keep (bool
    [B,H,Sq,Sk]) with drop_p: dropout on the probabilities after the softmax (GPT-2's attn_dropout)."""
    B, H, Sq, Dh = q.shape
    KVH, Sk = k.shape[1], k.shape[2]
    if KVH != H:
        rep = H // KVH
        k = k[:, :, None].expand(B, KVH, rep, Sk, Dh).reshape(B, H, Sk, Dh)
        v = v[:, :, None].expand(B, KVH, rep, Sk, Dh).reshape(B, H, Sk, Dh)
    w = torch.matmul(q, k.transpose(2, 3)) * scale
    qi = torch.arange(Sq)[:, None] + q_offset
    kj = torch.arange(Sk)[None, :]
    allowed = (kj <= qi)[None, None]
    if key_padding_mask is not None:
        allowed = allowed & key_padding_mask.bool()[:, None, None, :]
    w = w.masked_fill(~allowed, torch.finfo(w.dtype).min)
    w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    if drop_keep is not None:
        w = w * drop_keep.to(w.dtype) * (1.0 / (1.0 - drop_p))
    o = torch.matmul(w, v)
    return o.transpose(1, 2).reshape(B, Sq, H * Dh)


def _bf(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16)


def linear(x: torch.Tensor, w: torch.Tensor, mode: str) -> torch.Tensor:
    """x @ w.T. In "bf16" mode both operands are rounded to bf16 and the result is bf16 (fp32 accumulate)."""
    if mode == "bf16":
        return F.linear(_bf(x), _bf(w))
    return F.linear(x, w)


# --------------------------------------------------------------------------------------------
# model
# --------------------------------------------------------------------------------------------
@dataclass
class KVCache:
    k: List[Optional[torch.Tensor]] = field(default_factory=list)
    v: List[Optional[torch.Tensor]] = field(default_factory=list)

    def length(self) -> int:
        return 0 if not self.k or self.k[0] is None else self.k[0].shape[2]


def decoder_layer(h: torch.Tensor, p: Dict[str, torch.Tensor], pre: str, cfg: LMConfig,
                  cos: torch.Tensor, sin: torch.Tensor, mode: str,
                  key_padding_mask: Optional[torch.Tensor], cache: Optional[KVCache], layer: int,
                  taps: Optional[dict] = None) -> torch.Tensor:
    B, S, d = h.shape
    H, KVH, Dh = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    x = block_norm(h, p, pre + "input_layernorm", cfg)
    q = linear(x, p[pre + "self_attn.q_proj.weight"], mode).view(B, S, H, Dh)
    k = linear(x, p[pre + "self_attn.k_proj.weight"], mode).view(B, S, KVH, Dh)
    v = linear(x, p[pre + "self_attn.v_proj.weight"], mode).view(B, S, KVH, Dh)
    if cfg.qk_norm:
        q = rmsnorm(q, p[pre + "self_attn.q_norm.weight"], cfg.rms_norm_eps)
        k = rmsnorm(k, p[pre + "self_attn.k_norm.weight"], cfg.rms_norm_eps)
    q, k, v = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
    rope = apply_rope_interleaved if cfg.rope_interleaved else apply_rope
    q = rope(q, cos, sin)
    k = rope(k, cos, sin)
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
        # fused attention kernels keep fp32 scores/accumulators over bf16 operands and round the output once
        a = _bf(causal_attention(q.float(), k.float(), v.float(), Dh ** -0.5, key_padding_mask, q_offset))
    else:
        a = causal_attention(q, k, v, Dh ** -0.5, key_padding_mask, q_offset)
    if taps is not None and layer == 0:
        taps["q_rot0"], taps["k_rot0"], taps["attn0"] = q, k, a
    h = h + linear(a, p[pre + "self_attn.o_proj.weight"], mode)
    x = block_norm(h, p, pre + "post_attention_layernorm", cfg)
    g = linear(x, p[pre + "mlp.gate_proj.weight"], mode)
    u = linear(x, p[pre + "mlp.up_proj.weight"], mode)
    h = h + linear(F.silu(g) * u, p[pre + "mlp.down_proj.weight"], mode)
    return h


def forward_hidden(p: Dict[str, torch.Tensor], cfg: LMConfig, input_ids: torch.Tensor,
                   attention_mask: Optional[torch.Tensor] = None, mode: str = "fp32",
                   cache: Optional[KVCache] = None, position_ids: Optional[torch.Tensor] = None,
                   taps: Optional[dict] = None) -> torch.Tensor:
    B, S = input_ids.shape
    past = cache.length() if cache is not None else 0
    if cache is not None and not cache.k:
        cache.k = [None] * cfg.num_hidden_layers
        cache.v = [None] * cfg.num_hidden_layers
    if position_ids is None:
        position_ids = (torch.arange(S) + past)[None, :].expand(B, S)
    h = F.embedding(input_ids, p["model.embed_tokens.weight"])
    cos, sin = rope_cos_sin(position_ids, cfg.head_dim, cfg.rope_theta)
    kpm = attention_mask if (attention_mask is not None and not bool(attention_mask.bool().all())) else None
    for l in range(cfg.num_hidden_layers):
        h = decoder_layer(h, p, f"model.layers.{l}.", cfg, cos, sin, mode, kpm, cache, l, taps)
        if taps is not None and l == 0:
            taps["h_after0"] = h
    return block_norm(h, p, "model.norm", cfg)


def lm_head_weight(p: Dict[str, torch.Tensor], cfg: LMConfig) -> torch.Tensor:
    return p["model.embed_tokens.weight"] if cfg.tie_word_embeddings else p["lm_head.weight"]


def causal_lm_loss(logits: torch.Tensor, labels: torch.Tensor, ignore_index: int = -100) -> torch.Tensor:
    V = logits.shape[-1]
    logits = logits.float()
    labels = F.pad(labels, (0, 1), value=ignore_index)
    shift = labels[..., 1:].contiguous()
    return F.cross_entropy(logits.view(-1, V), shift.view(-1), ignore_index=ignore_index, reduction="mean")


def forward(p: Dict[str, torch.Tensor], cfg: LMConfig, input_ids: torch.Tensor,
            attention_mask: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None,
            mode: str = "fp32", cache: Optional[KVCache] = None, taps: Optional[dict] = None):
    """Returns (loss or None, logits [B,S,V])."""
    hN = forward_hidden(p, cfg, input_ids, attention_mask, mode, cache, taps=taps)
    logits = linear(hN, lm_head_weight(p, cfg), mode)
    loss = causal_lm_loss(logits, labels) if labels is not None else None
    return loss, logits


def loss_and_grads(p: Dict[str, torch.Tensor], cfg: LMConfig, input_ids, attention_mask, labels, mode="fp32"):
    """Reference backward = autograd over the explicit forward above. Returns (loss, {name: grad fp32})."""
    q = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    loss, _ = forward(q, cfg, input_ids, attention_mask, labels, mode)
    loss.backward()
    return loss.detach(), {k: v.grad for k, v in q.items()}


# --------------------------------------------------------------------------------------------
# optimizer: clip + AdamW + schedule (src/model/model.py:11-46, src/train.py:66)
# --------------------------------------------------------------------------------------------
def lr_at(step: int, lr0: float, warmup: int, total_steps: int, end_ratio: float = 0.1,
          start_factor: float = 1e-6) -> float:
    """lr used AT optimizer step `step` (0-based): LinearLR(start_factor->1 over warmup) then
    LinearLR(1->end_ratio over total_steps-warmup), chained by SequentialLR."""
    if step < warmup:
        return lr0 * (start_factor + (1.0 - start_factor) * step / warmup)
    D = max(total_steps - warmup, 1)
    return lr0 * (1.0 - (1.0 - end_ratio) * min(step - warmup, D) / D)


def clip_coef(grads: List[torch.Tensor], max_norm: float) -> Tuple[torch.Tensor, torch.Tensor]:
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g.float(), 2) for g in grads]), 2)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return total, coef


def adamw_step(param, grad, m, v, step: int, lr: float, beta1=0.9, beta2=0.95, eps=1e-5, wd=0.1):
    """One decoupled-weight-decay Adam update, step is 1-based; updates in place (torch.optim.AdamW order)."""
    param.mul_(1.0 - lr * wd)
    m.lerp_(grad, 1.0 - beta1)
    v.mul_(beta2).addcmul_(grad, grad, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    param.addcdiv_(m, denom, value=-(lr / bc1))


# --------------------------------------------------------------------------------------------
# sampling
# --------------------------------------------------------------------------------------------
def warp_logits(scores: torch.Tensor, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0) -> torch.Tensor:
    """Temperature -> top-k -> top-p, in that order; filtered entries become -inf."""
    s = scores.float()
    if temperature != 1.0:
        s = s / temperature
    if top_k and top_k > 0:
        k = min(top_k, s.shape[-1])
        kth = torch.topk(s, k)[0][..., -1, None]
        s = s.masked_fill(s < kth, -float("inf"))
    if top_p < 1.0:
        sl, si = torch.sort(s, descending=False)
        cum = sl.softmax(dim=-1).cumsum(dim=-1)
        rem = cum <= (1 - top_p)
        rem[..., -1:] = False
        rem = rem.scatter(1, si, rem)
        s = s.masked_fill(rem, -float("inf"))
    return s


def sample_with_noise(scores: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """torch.multinomial(softmax(scores), 1) drawn with explicit Exp(1) noise q: argmax(p / q)."""
    probs = F.softmax(scores, dim=-1)
    return (probs / noise).argmax(dim=-1)


def generate(p, cfg: LMConfig, input_ids: torch.Tensor, max_new_tokens: int, do_sample: bool = False,
             temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0, eos_token_id: Optional[int] = None,
             pad_token_id: Optional[int] = None, noise: Optional[torch.Tensor] = None, mode: str = "fp32",
             return_logits: bool = False, attention_mask: Optional[torch.Tensor] = None):
    """KV-cached sampling loop. noise [steps,B,V] Exp(1) draws when do_sample. Returns ids [B, prompt+new].
    attention_mask [B, prompt] (1 = real token; left padding for batched prompts): padded keys stay masked at every
    step, the mask grows by a column of ones per token, and positions count real tokens only
    (position_ids = cumsum(mask) - 1, 1 where masked: transformers generation/utils.py
    _update_model_kwargs_for_generation and prepare_inputs_for_generation)."""
    B = input_ids.shape[0]
    cache = KVCache()
    unfinished = torch.ones(B, dtype=torch.long)
    ids = input_ids
    cur = input_ids
    all_logits = []
    mask = attention_mask.long() if attention_mask is not None else None
    for t in range(max_new_tokens):
        pos = None
        if mask is not None:
            pos = (mask.cumsum(-1) - 1).masked_fill(mask == 0, 1)[:, -cur.shape[1]:]
        with torch.no_grad():
            hN = forward_hidden(p, cfg, cur, mask, mode, cache, position_ids=pos)
            logits = linear(hN[:, -1:, :], lm_head_weight(p, cfg), mode)[:, 0, :].float()
        if return_logits:
            all_logits.append(logits)
        if do_sample:
            nxt = sample_with_noise(warp_logits(logits, temperature, top_k, top_p), noise[t])
        else:
            nxt = logits.argmax(dim=-1)
        if eos_token_id is not None:
            nxt = nxt * unfinished + (pad_token_id if pad_token_id is not None else 0) * (1 - unfinished)
            unfinished = unfinished & (nxt != eos_token_id).long()
        ids = torch.cat([ids, nxt[:, None]], dim=1)
        cur = nxt[:, None]
        if mask is not None:
            mask = torch.cat([mask, torch.ones(B, 1, dtype=torch.long)], dim=1)
        if eos_token_id is not None and int(unfinished.max()) == 0:
            break
    if return_logits:
        return ids, torch.stack(all_logits, 0)
    return ids
