"""MidiGPT2LM: the GPT-2 style decoder of the reference on the same flat-parameter engine as MidiLM.

Call surface of the reference's MidiGPT2 (/root/reference/src/model/model.py:49-72, which wraps transformers'
GPT2LMHeadModel): `model(input_ids, attention_mask=None, labels=None)` -> `.loss`, `.logits`; `generate(...)`;
`state_dict()` uses the HF names (`transformer.wte.weight`, `transformer.h.0.attn.c_attn.weight`, ...), Conv1D
weights are stored [in, out].

Block arithmetic (transformers models/gpt2/modeling_gpt2.py, GPT2Block.forward): pre-LayerNorm residual blocks,
fused c_attn producing q | k | v, causal attention over 64-wide heads scaled by 1/sqrt(64), c_proj, gelu_new MLP,
learned absolute positions, head tied to wte. In training mode (`model.train()`, the default) with non-zero
`embd_pdrop` / `resid_pdrop` / `attn_pdrop` the three dropouts of GPT2Model are applied: on the summed embeddings, on
the output of both residual projections, and on the attention probabilities inside the attention kernels. Masks
are a hash of (key, element index); every dropout site of every forward gets its own key from `dropout_seed` and a
running counter, the backward pass recomputes the masks from the keys (nothing is stored).
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import torch

from . import ops
from .config import LMConfig
from .model import MidiLM, _LayerViews, _register


def gpt2_param_layout(cfg: LMConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    d, I = cfg.hidden_size, cfg.intermediate_size
    out = [("transformer.wte.weight", (cfg.vocab_size, d)),
           ("transformer.wpe.weight", (cfg.max_position_embeddings, d))]
    for l in range(cfg.num_hidden_layers):
        p = f"transformer.h.{l}."
        out += [(p + "ln_1.weight", (d,)), (p + "ln_1.bias", (d,)),
                (p + "attn.c_attn.weight", (d, 3 * d)), (p + "attn.c_attn.bias", (3 * d,)),
                (p + "attn.c_proj.weight", (d, d)), (p + "attn.c_proj.bias", (d,)),
                (p + "ln_2.weight", (d,)), (p + "ln_2.bias", (d,)),
                (p + "mlp.c_fc.weight", (d, I)), (p + "mlp.c_fc.bias", (I,)),
                (p + "mlp.c_proj.weight", (I, d)), (p + "mlp.c_proj.bias", (d,))]
    out += [("transformer.ln_f.weight", (d,)), ("transformer.ln_f.bias", (d,))]
    return out


_LAYER_FIELDS = {"ln1w": "ln_1.weight", "ln1b": "ln_1.bias", "attn": "attn.c_attn.weight", "attn_b": "attn.c_attn.bias",
                 "proj": "attn.c_proj.weight", "proj_b": "attn.c_proj.bias", "ln2w": "ln_2.weight",
                 "ln2b": "ln_2.bias", "fc": "mlp.c_fc.weight", "fc_b": "mlp.c_fc.bias", "mproj": "mlp.c_proj.weight",
                 "mproj_b": "mlp.c_proj.bias"}


class MidiGPT2LM(MidiLM):
    def __init__(self, cfg: LMConfig, device="cuda", seed: Optional[int] = 0, compute_dtype=torch.bfloat16):
        if cfg.arch != "gpt2":
            raise ValueError("MidiGPT2LM needs a config with arch='gpt2' (mgen.config.gpt2_decoder())")
        if cfg.hidden_size != cfg.num_attention_heads * cfg.head_dim or \
                cfg.num_key_value_heads != cfg.num_attention_heads or not cfg.tie_word_embeddings:
            raise ValueError("GPT-2: hidden = heads * head_dim, no grouped keys, tied head")
        super().__init__(cfg, device=device, seed=seed, compute_dtype=compute_dtype)
        self.dropout_seed = int(seed or 0)     # give every data-parallel rank its own
        self._dropout_calls = 0                # training-mode forwards so far

    def _drop_keys(self, cache, pos):
        """Keys of this forward's dropout sites, or None when dropout is off (eval mode, decoding, all p = 0).
        Site 0 = embeddings, then per layer: attention probabilities, attention projection, MLP projection."""
        cfg = self.cfg
        if not self.training or cache is not None or pos is not None or \
                (cfg.embd_pdrop == 0 and cfg.resid_pdrop == 0 and cfg.attn_pdrop == 0):
            return None
        if self.act_dtype == torch.float32:
            raise NotImplementedError("dropout runs in the bf16-mixed mode only: use eval() or p = 0 in fp32 mode")
        n = 3 * cfg.num_hidden_layers + 1
        base = self._dropout_calls * n
        self._dropout_calls += 1
        return [ops.dropout_key(self.dropout_seed, base + i) for i in range(n)]

    # ---- parameter store -----------------------------------------------------------------------------------
    def _param_layout(self):
        return gpt2_param_layout(self.cfg)

    def _post_register(self, dev):
        _register(self, "lm_head.weight", self.transformer.wte.weight)

    def _make_layer_views(self, l):
        p = f"transformer.h.{l}."
        v = _LayerViews()
        for tag, flat in (("w_", self.flat_param), ("g_", self.flat_grad), ("s_", self.flat_shadow)):
            for short, name in _LAYER_FIELDS.items():
                setattr(v, tag + short, self._view(flat, p + name))
        if self.flat_shadow_lo is not None:   # fp32 parity mode: split weights, fp32 biases
            for short, name in _LAYER_FIELDS.items():
                setattr(v, "s_" + short, getattr(v, "w_" + short) if short.endswith("_b") else self._w(p + name))
        return v

    def _lm_head_name(self):
        return "transformer.wte.weight"

    def init_weights(self, seed: int = 0):
        """GPT2PreTrainedModel._init_weights: N(0, 0.02) weights and embeddings, zero biases, unit LayerNorm gains,
        and N(0, 0.02 / sqrt(2 * n_layer)) for the two residual projections (c_proj) of every block."""
        g = torch.Generator(device=self.flat_param.device).manual_seed(seed)
        std_proj = 0.02 / math.sqrt(2 * self.cfg.num_hidden_layers)
        with torch.no_grad():
            for name, prm in self.named_parameters():
                if name == "lm_head.weight":
                    continue
                if name.endswith(".bias"):
                    prm.zero_()
                elif ".ln_" in name:
                    prm.fill_(1.0)
                else:
                    prm.normal_(0.0, std_proj if name.endswith("c_proj.weight") else 0.02, generator=g)
        self._shadow_version = -1

    def grad_ranges(self):
        cfg, o = self.cfg, self._offsets
        out = [("head", o["transformer.ln_f.weight"], self._total)]
        for l in range(cfg.num_hidden_layers - 1, -1, -1):
            start = o[f"transformer.h.{l}.ln_1.weight"]
            end = o[f"transformer.h.{l + 1}.ln_1.weight"] if l + 1 < cfg.num_hidden_layers else \
                o["transformer.ln_f.weight"]
            out.append((l, start, end))
        out.append(("embed", 0, o["transformer.h.0.ln_1.weight"]))
        return out

    # ---- forward -------------------------------------------------------------------------------------------
    def _run_layers(self, input_ids, save: Optional[list], pos: Optional[torch.Tensor] = None, cache=None):
        cfg = self.cfg
        B, S = input_ids.shape
        T = B * S
        d, H, Dh, I = cfg.hidden_size, cfg.num_attention_heads, cfg.head_dim, cfg.intermediate_size
        if pos is None and S > cfg.max_position_embeddings:
            raise ValueError(f"sequence length {S} exceeds n_positions {cfg.max_position_embeddings}")
        dev, bf, f32 = input_ids.device, self.act_dtype, torch.float32
        full = bf == torch.float32            # fp32 parity mode
        esz = 4 if full else 2                # bytes per activation element (pointer offsets into qkv)
        eps = cfg.layer_norm_eps
        scale = Dh ** -0.5
        h = torch.empty(T, d, dtype=f32, device=dev)
        ops.embed_pos_fwd(input_ids, self.transformer.wte.weight, self.transformer.wpe.weight, pos, h, S,
                          self._err_flag)
        keys = self._drop_keys(cache, pos)
        p_e, p_r, p_a = (cfg.embd_pdrop, cfg.resid_pdrop, cfg.attn_pdrop) if keys else (0.0, 0.0, 0.0)
        if p_e > 0:
            ops.dropout(h, h, p_e, keys[0])

        def proj_resid(x, w, b, resid, key):
            """resid + bf16(x w + b), the projection output dropped first when resid_pdrop is on"""
            out = torch.empty(T, d, dtype=f32, device=dev)
            if p_r > 0:
                y = torch.empty(T, d, dtype=bf, device=dev)
                ops.conv1d_fwd(x, w, b, y)
                ops.dropout_resid(y, resid, out, p_r, key)
            else:
                ops.conv1d_fwd(x, w, b, out, ops.OUT_F32_RESID_BF16ROUND, resid)
            return out

        for l, v in enumerate(self._layers):
            x1 = torch.empty(T, d, dtype=bf, device=dev)
            mean1 = torch.empty(T, dtype=f32, device=dev)
            rstd1 = torch.empty(T, dtype=f32, device=dev)
            ops.layernorm_fwd(h, v.w_ln1w, v.w_ln1b, x1, mean1, rstd1, eps)
            qkv = torch.empty(T, 3 * d, dtype=bf, device=dev)
            ops.conv1d_fwd(x1, v.s_attn, v.s_attn_b, qkv)
            qp, kp, vp = qkv.data_ptr(), qkv.data_ptr() + esz * d, qkv.data_ptr() + 2 * esz * d
            if cache is not None:
                ops.kv_append(kp, vp, 3 * d, 3 * d, cache.k[l], cache.v[l], cache.zero_pos, B, S, H, Dh,
                              cache.max_len)
            att = torch.empty(T, d, dtype=bf, device=dev)
            lse = torch.empty(B, H, S, dtype=f32, device=dev)
            ops.attn_fwd(qp, kp, vp, att.data_ptr(), lse, 3 * d, 3 * d, 3 * d, d, B, S, H, H, Dh, scale, f32=full,
                         drop_p=p_a, drop_key=keys[3 * l + 1] if keys else 0)
            h_mid = proj_resid(att, v.s_proj, v.s_proj_b, h, keys[3 * l + 2] if keys else 0)
            x2 = torch.empty(T, d, dtype=bf, device=dev)
            mean2 = torch.empty(T, dtype=f32, device=dev)
            rstd2 = torch.empty(T, dtype=f32, device=dev)
            ops.layernorm_fwd(h_mid, v.w_ln2w, v.w_ln2b, x2, mean2, rstd2, eps)
            fc = torch.empty(T, I, dtype=bf, device=dev)
            ops.conv1d_fwd(x2, v.s_fc, v.s_fc_b, fc)
            act = torch.empty(T, I, dtype=bf, device=dev)
            ops.gelu_new_fwd(fc, act)
            h_out = proj_resid(act, v.s_mproj, v.s_mproj_b, h_mid, keys[3 * l + 3] if keys else 0)
            if save is not None:
                save.append((h, x1, mean1, rstd1, qkv, att, lse, h_mid, x2, mean2, rstd2, fc, act))
            h = h_out
        xN = torch.empty(T, d, dtype=bf, device=dev)
        meanN = torch.empty(T, dtype=f32, device=dev)
        rstdN = torch.empty(T, dtype=f32, device=dev)
        ops.layernorm_fwd(h, self.transformer.ln_f.weight, self.transformer.ln_f.bias, xN, meanN, rstdN, eps)
        return xN, (h, meanN, rstdN, keys)

    # ---- backward ------------------------------------------------------------------------------------------
    def _backward_layers(self, input_ids, saved, final, dxN, grad_hook=None):
        cfg = self.cfg
        B, S = input_ids.shape
        T = B * S
        d, H, Dh, I = cfg.hidden_size, cfg.num_attention_heads, cfg.head_dim, cfg.intermediate_size
        dev, bf, f32 = input_ids.device, self.act_dtype, torch.float32
        full = bf == torch.float32
        esz = 4 if full else 2
        scale = Dh ** -0.5
        h_last, meanN, rstdN, keys = final
        p_e, p_r, p_a = (cfg.embd_pdrop, cfg.resid_pdrop, cfg.attn_pdrop) if keys else (0.0, 0.0, 0.0)
        dy = torch.empty(T, d, dtype=bf, device=dev) if p_r > 0 else None   # gradient of a dropped projection output
        dh = torch.empty(T, d, dtype=f32, device=dev)
        dh_lp = torch.empty(T, d, dtype=bf, device=dev)
        # The four weight gradients of a layer as ONE launch at the end of the layer (a quarter of the reduction slices
        # and atomics of four launches): the output gradients the MLP branch saw must then outlive the ones the
        # attention branch sees, hence second buffers.
        group = not full
        dh_lp2 = torch.empty(T, d, dtype=bf, device=dev) if group else dh_lp
        dy2 = torch.empty(T, d, dtype=bf, device=dev) if (group and p_r > 0) else dy
        ops.layernorm_bwd(dxN, h_last, self.transformer.ln_f.weight, meanN, rstdN, dh, dh_lp,
                          self._view(self.flat_grad, "transformer.ln_f.weight"),
                          self._view(self.flat_grad, "transformer.ln_f.bias"), accumulate=False)
        if grad_hook is not None:
            grad_hook("head")
        dact = torch.empty(T, I, dtype=bf, device=dev)
        dfc = torch.empty(T, I, dtype=bf, device=dev)
        dx = torch.empty(T, d, dtype=bf, device=dev)
        datt = torch.empty(T, d, dtype=bf, device=dev)
        dqkv = torch.empty(T, 3 * d, dtype=bf, device=dev)
        delta = torch.empty(B, H, S, dtype=f32, device=dev)
        for l in range(cfg.num_hidden_layers - 1, -1, -1):
            v = self._layers[l]
            h_in, x1, mean1, rstd1, qkv, att, lse, h_mid, x2, mean2, rstd2, fc, act = saved[l]
            # ---- MLP branch
            g = dh_lp
            if p_r > 0:
                ops.dropout(dh_lp, dy, p_r, keys[3 * l + 3])
                g = dy
            ops.conv1d_dgrad(g, v.s_mproj, dact)
            wgrad = ops.conv1d_bias_grad if group else ops.conv1d_wgrad   # (grouped: only the bias column sums now)
            wgrad(g, act, v.g_mproj, v.g_mproj_b)
            ops.gelu_new_bwd(fc, dact, dfc)
            ops.conv1d_dgrad(dfc, v.s_fc, dx)
            wgrad(dfc, x2, v.g_fc, v.g_fc_b)
            g_mlp, dh_mlp = g, dh_lp
            dh_lp = dh_lp2            # (the same buffer unless the weight gradients are grouped)
            ops.layernorm_bwd(dx, h_mid, v.w_ln2w, mean2, rstd2, dh, dh_lp, v.g_ln2w, v.g_ln2b, accumulate=True)
            # ---- attention branch
            g = dh_lp
            if p_r > 0:
                ops.dropout(dh_lp, dy2, p_r, keys[3 * l + 2])
                g = dy2
            ops.conv1d_dgrad(g, v.s_proj, datt)
            wgrad(g, att, v.g_proj, v.g_proj_b)
            qp, kp, vp = qkv.data_ptr(), qkv.data_ptr() + esz * d, qkv.data_ptr() + 2 * esz * d
            dqp, dkp, dvp = dqkv.data_ptr(), dqkv.data_ptr() + esz * d, dqkv.data_ptr() + 2 * esz * d
            ops.attn_bwd(qp, kp, vp, att.data_ptr(), datt.data_ptr(), lse, delta, dqp, dkp, dvp,
                         3 * d, 3 * d, 3 * d, d, d, 3 * d, 3 * d, 3 * d, B, S, H, H, Dh, scale, f32=full,
                         drop_p=p_a, drop_key=keys[3 * l + 1] if keys else 0)
            ops.conv1d_dgrad(dqkv, v.s_attn, dx)
            wgrad(dqkv, x1, v.g_attn, v.g_attn_b)
            if group:   # Conv1D weights are [in][out]: dW += x^T dy
                ops.linear_wgrad_group([(act, g_mlp, v.g_mproj), (x2, dfc, v.g_fc), (att, g, v.g_proj),
                                        (x1, dqkv, v.g_attn)])
            dh_lp = dh_mlp            # the layer below reads (and keeps) this one
            ops.layernorm_bwd(dx, h_in, v.w_ln1w, mean1, rstd1, dh, dh_lp if l > 0 else None, v.g_ln1w, v.g_ln1b,
                              accumulate=True)
            saved[l] = None
            if grad_hook is not None:
                grad_hook(l)
        if p_e > 0:
            ops.dropout(dh, dh, p_e, keys[0])
        ops.embedding_bwd(input_ids, dh, self._view(self.flat_grad, "transformer.wte.weight"))
        pos_ids = torch.arange(S, device=dev, dtype=torch.int64).repeat(B)
        ops.embedding_bwd(pos_ids, dh, self._view(self.flat_grad, "transformer.wpe.weight"))
        if grad_hook is not None:
            grad_hook("embed")

    # ---- decode --------------------------------------------------------------------------------------------
    def _decode_step(self, tokens, cache, t: int, buf, lens, ws):
        cfg = self.cfg
        B = tokens.shape[0]
        d, H, Dh, I = cfg.hidden_size, cfg.num_attention_heads, cfg.head_dim, cfg.intermediate_size
        if t >= cfg.max_position_embeddings:
            raise ValueError(f"position {t} exceeds n_positions {cfg.max_position_embeddings}")
        eps = cfg.layer_norm_eps
        scale = Dh ** -0.5
        pos, kv_len = lens[t], lens[t + 1]
        nsplit = self._decode_split(B, t + 1)
        h = buf.h0
        ops.embed_pos_fwd(tokens, self.transformer.wte.weight, self.transformer.wpe.weight, pos, h, 1, self._err_flag)
        qp = buf.qkv.data_ptr()
        kp, vp = qp + 2 * d, qp + 4 * d
        fc = buf.gu.view(-1)[:B * I].view(B, I)   # the [B, 2I] gate-up area of the shared buffers, first half
        for l, v in enumerate(self._layers):
            ops.layernorm_fwd(h, v.w_ln1w, v.w_ln1b, buf.x, None, None, eps)
            ops.conv1d_fwd(buf.x, v.s_attn, v.s_attn_b, buf.qkv)
            ops.kv_append(kp, vp, 3 * d, 3 * d, cache.k[l], cache.v[l], pos, B, 1, H, Dh, cache.max_len)
            ops.attn_decode(qp, 3 * d, cache.k[l], cache.v[l], kv_len, buf.att, d, ws, B, H, H, Dh, cache.max_len,
                            nsplit, scale)
            ops.conv1d_fwd(buf.att, v.s_proj, v.s_proj_b, buf.h1, ops.OUT_F32_RESID_BF16ROUND, h)
            ops.layernorm_fwd(buf.h1, v.w_ln2w, v.w_ln2b, buf.x, None, None, eps)
            ops.conv1d_fwd(buf.x, v.s_fc, v.s_fc_b, fc)
            ops.gelu_new_fwd(fc, buf.act)
            nxt = buf.h2 if h is buf.h0 else buf.h0
            ops.conv1d_fwd(buf.act, v.s_mproj, v.s_mproj_b, nxt, ops.OUT_F32_RESID_BF16ROUND, buf.h1)
            h = nxt
        ops.layernorm_fwd(h, self.transformer.ln_f.weight, self.transformer.ln_f.bias, buf.x, None, None, eps)
        W = self._view(self.flat_shadow, self._lm_head_name())
        ops.gemm(buf.x, W, buf.logits, B, cfg.vocab_size, d, d, W.stride(0), buf.ldv)
        return buf.logits
