"""MidiLM: the decoder-only transformer over tokenized MIDI, running on hand-written gfx950 kernels.

Call surface of the reference's wrapped model (/root/reference/src/model/model.py:142-143, which forwards to
transformers' Qwen3ForCausalLM.forward): `model(input_ids, attention_mask=None, labels=None)` returns an object with
`.loss` (0-dim fp32 tensor, differentiable) and `.logits`; `state_dict()` uses the HF parameter names
(`model.layers.0.self_attn.q_proj.weight`, ...), `nn.Linear` weights are [out, in].

Design for MI355X: all parameters live in ONE flat fp32 buffer (`flat_param`) with a flat fp32 gradient buffer and a
flat bf16 shadow copy next to it; q/k/v and gate/up are adjacent in that buffer, so the fused QKV and gate-up GEMMs
read them as single matrices. The optimizer, the gradient-norm reduction and the data-parallel all-reduce work on
contiguous slices of these buffers. The whole network is one autograd node: forward saves the activations
(288 GB of HBM: nothing is recomputed), the lm_head + cross-entropy forward and backward are fused per chunk of
tokens so that the [T, V] logits are never alive at once, and backward walks the layers in reverse calling the
dgrad/wgrad GEMMs and the fused norm/rope/attention backward kernels.
"""
from __future__ import annotations

import math
import os
import weakref
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from .config import LMConfig
from .generation import GenerationMixin

_ALIGN = 64  # every tensor starts at a multiple of 64 elements in the flat buffers
POS4D_HEAD_OUT = 16  # 4-D position model: 8 (cos, sin) pairs (/root/reference/src/model/modeling.py:560)


# MGEN_LM_WGRAD_8K=1: the lm_head weight gradient in launches of 8192 tokens (same-box comparisons)
_LM_WGRAD_8K = os.environ.get("MGEN_LM_WGRAD_8K", "0") == "1"
# MGEN_SPLIT_PROJ=0: o_proj / down_proj with the fp32 residual epilogue + a separate norm (same-box comparisons)
_SPLIT_PROJ = os.environ.get("MGEN_SPLIT_PROJ", "1") != "0"
# MGEN_ROW_PAD=0: no padding of the gradient activations' rows (same-box comparisons)
_ROW_PAD = int(os.environ.get("MGEN_ROW_PAD", "64"))
_ATT_PAD = int(os.environ.get("MGEN_ATT_PAD", str(_ROW_PAD)))


class CausalLMOutput:
    """Minimal stand-in for transformers' CausalLMOutputWithPast: attribute and index access."""

    def __init__(self, loss=None, logits=None):
        self.loss = loss
        self.logits = logits

    def __getitem__(self, i):
        return tuple(x for x in (self.loss, self.logits) if x is not None)[i]

    def __iter__(self):
        return iter(tuple(x for x in (self.loss, self.logits) if x is not None))


def param_layout(cfg: LMConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    d, H, KVH, Dh, I = (cfg.hidden_size, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim,
                        cfg.intermediate_size)
    out = [] if cfg.pos4d else [("model.embed_tokens.weight", (cfg.vocab_size, d))]
    for l in range(cfg.num_hidden_layers):
        p = f"model.layers.{l}."
        out.append((p + "input_layernorm.weight", (d,)))
        if cfg.norm_kind == "layer":
            out.append((p + "input_layernorm.bias", (d,)))
        out.append((p + "self_attn.q_proj.weight", (H * Dh, d)))     # q, k, v adjacent = one [qkv_width, d] matrix
        out.append((p + "self_attn.k_proj.weight", (KVH * Dh, d)))
        out.append((p + "self_attn.v_proj.weight", (KVH * Dh, d)))
        if cfg.qk_norm:
            out.append((p + "self_attn.q_norm.weight", (Dh,)))
            out.append((p + "self_attn.k_norm.weight", (Dh,)))
        out.append((p + "self_attn.o_proj.weight", (d, H * Dh)))
        out.append((p + "post_attention_layernorm.weight", (d,)))
        if cfg.norm_kind == "layer":
            out.append((p + "post_attention_layernorm.bias", (d,)))
        out.append((p + "mlp.gate_proj.weight", (I, d)))             # gate, up adjacent = one [2I, d] matrix
        out.append((p + "mlp.up_proj.weight", (I, d)))
        out.append((p + "mlp.down_proj.weight", (d, I)))
    out.append(("model.norm.weight", (d,)))
    if cfg.norm_kind == "layer":
        out.append(("model.norm.bias", (d,)))
    if cfg.pos4d:
        out.append(("lm_head.weight", (POS4D_HEAD_OUT, d)))
    elif not cfg.tie_word_embeddings:
        out.append(("lm_head.weight", (cfg.vocab_size, d)))
    return out


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


def _register(root: nn.Module, dotted: str, param: nn.Parameter):
    parts = dotted.split(".")
    mod = root
    for i, p in enumerate(parts[:-1]):
        if p not in mod._modules:
            if isinstance(mod, nn.ModuleList):
                while len(mod) <= int(p):
                    mod.append(nn.Module())
            else:
                # a container whose children are numbered (model.layers.N) is a ModuleList, as in HF models
                numbered = i + 1 < len(parts) - 1 and parts[i + 1].isdigit()
                mod.add_module(p, nn.ModuleList() if numbered else nn.Module())
        mod = mod._modules[p]
    mod.register_parameter(parts[-1], param)


class _LayerViews:
    """Per-layer views into the flat buffers (fp32 master, fp32 grad, bf16 shadow)."""
    pass


class MidiLM(GenerationMixin, nn.Module):
    def __init__(self, cfg: LMConfig, device="cuda", seed: Optional[int] = 0, compute_dtype=torch.bfloat16):
        super().__init__()
        if compute_dtype not in (torch.bfloat16, torch.float32):
            raise NotImplementedError("compute_dtype is torch.bfloat16 (bf16-mixed, the training path) or torch.float32 "
                                      "(parity mode)")
        # bf16: what precision="bf16-mixed" computes. fp32: no rounding anywhere; products run on the bf16 MFMA GEMM
        # from operands split into hi + lo halves (three products per GEMM), attention and the gate in plain fp32
        self.act_dtype = compute_dtype
        self.cfg = cfg
        self.config = cfg
        self.device_ = torch.device(device)
        d = cfg.hidden_size
        assert cfg.head_dim in (64, 128) and d % 8 == 0 and cfg.intermediate_size % 8 == 0
        if cfg.norm_kind not in ("rms", "layer"):
            raise ValueError("norm_kind is 'rms' or 'layer'")
        if cfg.norm_kind == "layer" and cfg.pos4d:
            raise ValueError("the 4-D position model has RMSNorm blocks")
        self._ln = cfg.norm_kind == "layer"
        self._norm_eps = cfg.layer_norm_eps if self._ln else cfg.rms_norm_eps
        layout = self._param_layout()
        offs, off = {}, 0
        for name, shape in layout:
            # q/k/v and gate/up must stay back to back: their sizes are multiples of _ALIGN already
            n = _numel(shape)
            offs[name] = off
            off += (n + _ALIGN - 1) // _ALIGN * _ALIGN
        self._offsets, self._shapes, self._total = offs, dict(layout), off
        dev = self.device_
        self.flat_param = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_shadow = torch.zeros(off, dtype=torch.bfloat16, device=dev)
        self.flat_shadow_lo = torch.zeros(off, dtype=torch.bfloat16, device=dev) \
            if compute_dtype == torch.float32 else None
        self._shadow_version = -1
        self._shadow_epoch = 0     # counts rewrites of the bf16 shadow (transposed copies follow it)
        self._wt = None            # transposed bf16 copies of o_proj / down_proj, see _dgrad_weights_t
        self._wt_epoch = -1
        for name, shape in layout:
            prm = nn.Parameter(self._view(self.flat_param, name), requires_grad=True)
            prm.grad = self._view(self.flat_grad, name)
            _register(self, name, prm)
        self._post_register(dev)
        # layer views
        self._layers = []
        for l in range(cfg.num_hidden_layers):
            self._layers.append(self._make_layer_views(l))
        self._rope_cs = None
        self._rope_npos = 0
        self._lm_grad_tmp = None       # head weight gradient of the one forward that owns it (see _head_grad_buffer)
        self._lm_grad_owner = None
        self.lora = None        # low-rank adapters (to_lora): then the base weights are frozen
        # device-side error bits raised by kernels: 1 = token id out of range, 2 = attention_mask is not right
        # padding, 4 = label outside [0, V). Read (one synchronisation) and cleared by _check_device_flags.
        self._err_flag = torch.zeros(1, dtype=torch.int32, device=dev) if dev.type == "cuda" else None
        # defer_input_checks: backward does not wait for the device to read the error bits; it starts a copy of them
        # to the host, and raise_pending_errors() (called by the Trainer once the optimizer step is queued) raises.
        # A step loop then never stalls the host between backward and the optimizer, and under data parallelism
        # every rank has issued its collectives before one of them raises.
        self.defer_input_checks = False
        self._err_pending = None      # (pinned host copy, event) of a deferred check
        if seed is not None:
            self.init_weights(seed)

    # ------------------------------------------------------------------------------------------------
    # parameter store
    # ------------------------------------------------------------------------------------------------
    def _param_layout(self):
        return param_layout(self.cfg)

    def _post_register(self, dev):
        """Buffers and tied names that go with the parameters."""
        cfg = self.cfg
        if cfg.pos4d:
            if not cfg.qk_norm or cfg.tie_word_embeddings or cfg.head_dim % 8:
                raise ValueError("the 4-D position model has q/k norms and an untied 16-wide head")
            # constant input vector [1,0,1,0,...] (/root/reference/src/model/modeling.py:389-394)
            self.model.register_buffer("embed_vector", (1 - torch.arange(cfg.hidden_size, device=dev) % 2).float())
            self._rope4d_f = None
        elif cfg.tie_word_embeddings:
            _register(self, "lm_head.weight", self.model.embed_tokens.weight)

    def _check_device_flags(self):
        """Raise what the reference raises at the faulty call: its embedding lookup fails on a token id outside the
        vocabulary and its cross entropy on a label outside it (both device-side asserts there); an attention_mask
        other than right padding would need masked attention, which this path does not compute. The kernels keep
        going with a clamped id / an ignored label, so the check may come later than the fault: after backward, at
        the end of an evaluation forward and of generate()."""
        if self._err_flag is None:
            return
        bits = int(self._err_flag)
        self._raise_for_bits(bits)

    def _check_device_flags_deferred(self):
        """End of backward with defer_input_checks: start the copy of the error bits, do not wait."""
        if self._err_flag is None:
            return
        if not self.defer_input_checks:
            return self._check_device_flags()
        if self._err_pending is None:
            host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._err_pending = [host, torch.cuda.Event(), False]
        host, ev, _ = self._err_pending
        host.copy_(self._err_flag, non_blocking=True)
        ev.record()
        self._err_pending[2] = True

    def raise_pending_errors(self):
        """Raise what a deferred check found (waits for the copy started at the end of backward)."""
        if self._err_pending is None or not self._err_pending[2]:
            return
        host, ev, _ = self._err_pending
        ev.synchronize()
        self._err_pending[2] = False
        self._raise_for_bits(int(host))

    def _raise_for_bits(self, bits):
        if bits == 0:
            return
        self._err_flag.zero_()
        if bits & 1:
            raise IndexError("token id out of range [0, vocab_size)")
        if bits & 4:
            raise IndexError("label outside [0, vocab_size) that is not ignore_index (-100)")
        raise ValueError("attention_mask must be right padding (ones followed by zeros in every row): causal "
                         "attention then needs no mask for the real tokens; other masks are not supported")

    def _head_grad_buffer(self, shape, token):
        """fp32 buffer for the head's weight gradient, formed during forward and added to the gradient in
        backward. One buffer is kept on the model and lent to one autograd graph at a time; a forward that starts
        while the previous graph has not run its backward yet (two losses summed before one backward) gets a
        buffer of its own."""
        owner = self._lm_grad_owner() if self._lm_grad_owner is not None else None
        if getattr(self, "_head_grad_sink", None) is not None and self._lm_grad_busy():
            return torch.empty(shape, dtype=torch.float32, device=self.flat_param.device)
        if owner is not None and not owner.done:
            return torch.empty(shape, dtype=torch.float32, device=self.flat_param.device)
        if self._lm_grad_tmp is None or tuple(self._lm_grad_tmp.shape) != tuple(shape):
            self._lm_grad_tmp = torch.empty(shape, dtype=torch.float32, device=self.flat_param.device)
        self._lm_grad_owner = weakref.ref(token)
        return self._lm_grad_tmp

    def _lm_grad_busy(self):
        """True while a gradient reducer still holds the shared head-gradient buffer of an earlier backward."""
        busy = getattr(self, "_head_grad_in_flight", None)
        return bool(busy and busy())

    def _anchor(self):
        """A parameter handed to the autograd node so that it has a differentiable input."""
        if self.lora is not None:
            return self.lora.anchor
        return self.lm_head.weight

    # ------------------------------------------------------------------------------------------------
    # low-rank adapters
    # ------------------------------------------------------------------------------------------------
    def to_lora(self, r: int = 32, lora_alpha: int = 64, seed: int = 0, lora_dropout: float = 0.0):
        """`MidiAria.to_lora` (/root/reference/src/model/model.py:251-271): freeze every base weight and train
        rank-r adapters on the fused QKV, attention output, gate, up and down projections. Returns the adapter
        store (its flat buffers are what the fused optimizer and the gradient reducer then work on). `lora_dropout`
        (0.05 in the reference) masks each adapter's input in training mode."""
        from .lora import LoraAdapters
        if self.lora is not None:
            raise RuntimeError("the model already carries adapters")
        for prm in self.parameters():
            prm.requires_grad = False
        self.lora = LoraAdapters(self, r, lora_alpha, seed, dropout=lora_dropout)
        return self.lora

    def merge_lora(self):
        """Fold the adapters into the base weights (W += (alpha / r) B A, fp32) and drop them: the form generate()
        and checkpoints of the plain model use."""
        lo = self.lora
        if lo is None:
            return
        lo.sync_shadow()
        for l, v in enumerate(self._layers):
            st = lo.layers[l]
            I = self.cfg.intermediate_size
            targets = (("qkv", v.w_qkv), ("o", v.w_o), ("gate", v.w_gu[:I]), ("up", v.w_gu[I:]), ("down", v.w_down))
            for k, W in targets:
                n_in, n_out = lo.dims[k]
                # W[out, in] += s * B[out, r] A[r, in]: A is stored [K = r][N = in]
                ops.gemm(st[k].sB, st[k].sA, W, n_out, n_in, lo.r, lo.r, n_in, W.stride(0), False, True,
                         ops.OUT_F32_ACC, alpha=lo.scale)
        self.lora = None
        self._shadow_version = -1
        for prm in self.parameters():
            prm.requires_grad = True

    def _view(self, flat, name, shape=None, count=None):
        shape = shape or self._shapes[name]
        n = count or _numel(shape)
        o = self._offsets[name]
        return flat[o:o + n].view(shape)

    def _make_layer_views(self, l):
        cfg = self.cfg
        p = f"model.layers.{l}."
        d, I = cfg.hidden_size, cfg.intermediate_size
        v = _LayerViews()
        for tag, flat in (("w", self.flat_param), ("g", self.flat_grad), ("s", None)):
            view = (lambda name, shape=None: self._w(name, shape)) if flat is None else \
                (lambda name, shape=None, flat=flat: self._view(flat, name, shape))
            setattr(v, tag + "_ln1", view(p + "input_layernorm.weight"))
            setattr(v, tag + "_ln1b", view(p + "input_layernorm.bias") if cfg.norm_kind == "layer" else None)
            setattr(v, tag + "_ln2b", view(p + "post_attention_layernorm.bias") if cfg.norm_kind == "layer" else None)
            setattr(v, tag + "_qkv", view(p + "self_attn.q_proj.weight", (cfg.qkv_width, d)))
            setattr(v, tag + "_o", view(p + "self_attn.o_proj.weight"))
            setattr(v, tag + "_ln2", view(p + "post_attention_layernorm.weight"))
            setattr(v, tag + "_gu", view(p + "mlp.gate_proj.weight", (2 * I, d)))
            setattr(v, tag + "_down", view(p + "mlp.down_proj.weight"))
            if cfg.qk_norm:
                setattr(v, tag + "_qn", view(p + "self_attn.q_norm.weight"))
                setattr(v, tag + "_kn", view(p + "self_attn.k_norm.weight"))
            else:
                setattr(v, tag + "_qn", None)
                setattr(v, tag + "_kn", None)
        return v

    def _w(self, name, shape=None):
        """The GEMM operand form of a weight: its bf16 shadow, or the (hi, lo) pair in the fp32 parity mode."""
        hi = self._view(self.flat_shadow, name, shape)
        if self.flat_shadow_lo is None:
            return hi
        return ops.W2(hi, self._view(self.flat_shadow_lo, name, shape))

    def init_weights(self, seed: int = 0):
        """HF `_init_weights`: N(0, 0.02) for linear and embedding weights, ones for the norm gains."""
        g = torch.Generator(device=self.flat_param.device).manual_seed(seed)
        with torch.no_grad():
            for name, prm in self.named_parameters():
                if name.endswith(".bias"):
                    prm.zero_()
                elif "norm" in name:
                    prm.fill_(1.0)
                else:
                    prm.normal_(0.0, 0.02, generator=g)
        self._shadow_version = -1

    # the block's norm: RMSNorm (gain) or LayerNorm (gain + bias); `stats` = rstd, or (mean, rstd)
    def _norm_fwd(self, h, w, b, y, want_stats=True):
        T = h.shape[0]
        dev = h.device
        if self._ln:
            mean = torch.empty(T, dtype=torch.float32, device=dev)
            rstd = torch.empty(T, dtype=torch.float32, device=dev)
            ops.layernorm_fwd(h, w, b, y, mean, rstd, self._norm_eps)
            return (mean, rstd)
        rstd = torch.empty(T, dtype=torch.float32, device=dev) if want_stats else None
        ops.rmsnorm_fwd(h, w, y, rstd, self._norm_eps)
        return rstd

    def _norm_bwd(self, dy, x, w, stats, dres, dres_lp, gw, gb, accumulate):
        if self._ln:
            ops.layernorm_bwd(dy, x, w, stats[0], stats[1], dres, dres_lp, gw, gb, accumulate)
        else:
            ops.rmsnorm_bwd(dy, x, w, stats, dres, dres_lp, gw, accumulate)

    def _final_norm_bias(self, flat=None):
        if not self._ln:
            return None
        return self.model.norm.bias if flat is None else self._view(flat, "model.norm.bias")

    def _lm_head_name(self):
        return "model.embed_tokens.weight" if self.cfg.tie_word_embeddings else "lm_head.weight"

    def sync_shadow(self):
        """Refresh the bf16 shadow from the fp32 master if the master changed through torch (the fused optimizer
        writes both itself and calls mark_shadow_fresh)."""
        if self._shadow_version != self.flat_param._version:
            if self.flat_shadow_lo is not None:
                ops.split_bf16_into(self.flat_param, self.flat_shadow, self.flat_shadow_lo)
            else:
                ops.cast(self.flat_param, self.flat_shadow)
            self._shadow_version = self.flat_param._version
            self._shadow_epoch += 1

    def mark_shadow_fresh(self):
        """Called by the fused optimizer after it rewrote the master weights and the bf16 shadow in one kernel."""
        # (that kernel refreshes only the hi half: the fp32 parity mode splits again at the next forward)
        self._shadow_version = self.flat_param._version if self.flat_shadow_lo is None else -1
        self._shadow_epoch += 1

    def _dgrad_weights_t(self):
        """Transposed bf16 copies [in_features][hidden] of every layer's o_proj and down_proj weights, or None.
        With hidden_size 512 the input gradients of those two projections (dx = dy @ W, reduction over the 512
        outputs) run on the GEMM kernel that keeps a 256-token panel of dy in registers, which wants both operands
        with the reduction index contiguous. The forward products of the two projections read the same copies as
        [reduction][output column] operands of the output-stationary kernel. Two batched transpose launches per
        optimizer step."""
        cfg = self.cfg
        if cfg.hidden_size != 512 or self.flat_shadow_lo is not None or cfg.arch != "llama":
            return None
        d, HD, I, L = cfg.hidden_size, cfg.num_attention_heads * cfg.head_dim, cfg.intermediate_size, \
            cfg.num_hidden_layers
        if self._wt is None:
            self._wt = (torch.empty(L, HD, d, dtype=torch.bfloat16, device=self.flat_shadow.device),
                        torch.empty(L, I, d, dtype=torch.bfloat16, device=self.flat_shadow.device))
            self._wt_epoch = -1
        if self._wt_epoch != self._shadow_epoch:
            o0, o1 = (self._offsets[f"model.layers.{l}.self_attn.o_proj.weight"] for l in (0, min(1, L - 1)))
            d0 = self._offsets["model.layers.0.mlp.down_proj.weight"]
            stride = o1 - o0   # layers are laid out identically, one after the other
            ops.transpose_bf16(self.flat_shadow[o0:], self._wt[0], d, HD, HD, d, L, stride, HD * d)
            ops.transpose_bf16(self.flat_shadow[d0:], self._wt[1], d, I, I, d, L, stride, I * d)
            self._wt_epoch = self._shadow_epoch
        return self._wt

    def zero_grad(self, set_to_none: bool = False):
        self.flat_grad.zero_()
        for name, prm in self.named_parameters():
            if prm.grad is None:
                prm.grad = self._view(self.flat_grad, name)

    def _apply(self, fn, recurse=True):
        # parameters are views of the flat buffers: moving them one by one would break the aliasing
        probe = fn(torch.empty(0, device=self.flat_param.device))
        if probe.device != self.flat_param.device or probe.dtype != torch.float32:
            raise RuntimeError("MidiLM lives on the device it was created on in fp32 master precision; "
                               "construct it with device=... instead of calling .to()/.half()")
        return self

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        own = dict(self.named_parameters())
        own_names = set(own) | ({"lm_head.weight"} if self.cfg.tie_word_embeddings else set()) | \
            ({"model.embed_vector"} if self.cfg.pos4d else set())
        missing = [k for k in own if k not in state_dict]
        unexpected = [k for k in state_dict if k not in own_names]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing}, unexpected {unexpected}")
        with torch.no_grad():
            for k, prm in own.items():
                if k in state_dict:
                    src = state_dict[k]
                    if tuple(src.shape) != tuple(prm.shape):
                        raise RuntimeError(f"size mismatch for {k}: {tuple(src.shape)} vs {tuple(prm.shape)}")
                    prm.copy_(src.to(device=prm.device, dtype=torch.float32))
        self._shadow_version = -1
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    # ------------------------------------------------------------------------------------------------
    # rope table
    # ------------------------------------------------------------------------------------------------
    def _rope_table(self, npos: int):
        if self._rope_cs is None or self._rope_npos < npos:
            cfg = self.cfg
            npos = max(npos, cfg.max_position_embeddings)
            # same expression as the reference (fp32 on the host), so inv_freq is bit-identical
            inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, cfg.head_dim, 2, dtype=torch.float) / cfg.head_dim))
            self._rope_cs = ops.rope_table(inv.to(self.flat_param.device), npos)
            self._rope_npos = npos
        return self._rope_cs

    def _rope4d_table(self, rows):
        """cos/sin of the 4-axis angles of every note row ([T,4] fp32); the frequency vector is built with the
        reference's expression (/root/reference/src/model/modeling.py:521-525) on the host so that it is
        bit-identical."""
        cfg = self.cfg
        if self._rope4d_f is None:
            Dh = cfg.head_dim
            rot = Dh // 8 - cfg.rope4d_nope
            f = 1.0 / (cfg.rope4d_theta ** (torch.arange(0, rot * 2, 2).float() / Dh))
            self._rope4d_f = torch.cat([f, torch.zeros(cfg.rope4d_nope)]).to(self.flat_param.device)
        return ops.rope4d_table(rows, self._rope4d_f, cfg.head_dim)

    # ------------------------------------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------------------------------------
    def forward(self, input_ids=None, attention_mask=None, labels=None, num_logits_to_keep: int = 0,
                position_tensors=None, **unused):
        """attention_mask: with right padding (what the reference's collators produce) causal attention already keeps
        real tokens from seeing pads, so loss and gradients do not depend on it and logits at pad positions are
        unspecified; any other mask raises ValueError (checked on the device, reported after backward or at the end
        of an evaluation forward)."""
        if self.cfg.pos4d:
            return self._forward_pos4d(position_tensors, labels)
        if input_ids is None:
            raise ValueError("You must specify input_ids")
        if input_ids.dim() != 2:
            raise ValueError("input_ids must be [batch, seq]")
        if not input_ids.is_cuda:
            raise RuntimeError("MidiLM runs on the GPU only: input_ids must be a CUDA/HIP tensor")
        input_ids = input_ids.contiguous()
        if labels is not None:
            labels = labels.contiguous().to(torch.int64)
        if attention_mask is not None:
            if tuple(attention_mask.shape) != tuple(input_ids.shape):
                raise ValueError("attention_mask must have the shape of input_ids")
            ops.check_right_padding(attention_mask.to(device=input_ids.device, dtype=torch.int64).contiguous(),
                                    self._err_flag)
        self.sync_shadow()
        if self.lora is not None:
            self.lora.sync_shadow()
        if labels is not None and torch.is_grad_enabled():
            loss = _LMTrainFunction.apply(self, input_ids, labels, self._anchor())
            return CausalLMOutput(loss=loss, logits=None)
        with torch.no_grad():
            hN, _ = self._run_layers(input_ids, save=None)
            loss = None
            if labels is not None:
                loss, _ = self._lm_head_loss(hN, labels, with_grad=False)
                self._check_device_flags()
                return CausalLMOutput(loss=loss, logits=None)
            B, S = input_ids.shape
            if num_logits_to_keep:
                hN = hN.view(B, S, -1)[:, -num_logits_to_keep:, :].reshape(-1, hN.shape[-1]).contiguous()
            logits = self._logits(hN).view(B, -1, self.cfg.vocab_size)
            self._check_device_flags()
            return CausalLMOutput(loss=None, logits=logits)

    def _forward_pos4d(self, position_tensors, labels):
        """Call surface of the reference's 4-D model (/root/reference/src/model/model.py:296-303 ->
        src/model/modeling.py:567-668): `position_tensors` float [B,S,4] note rows, `labels` float [B,S,4] (a row with
        a -100 in it is ignored); returns `.loss` and `.logits` = the unit angle pairs [B,S,8,2] in fp32."""
        if position_tensors is None or position_tensors.dim() != 3 or position_tensors.shape[-1] != 4:
            raise ValueError("position_tensors must be [batch, seq, 4]")
        if not position_tensors.is_cuda:
            raise RuntimeError("MidiLM runs on the GPU only: position_tensors must be a CUDA/HIP tensor")
        pt = position_tensors.contiguous().to(torch.float32)
        B, S = pt.shape[:2]
        if labels is not None:
            if tuple(labels.shape) != (B, S, 4):
                raise ValueError("labels must be [batch, seq, 4]")
            labels = labels.contiguous().to(torch.float32)
        self.sync_shadow()
        if labels is not None and torch.is_grad_enabled():
            loss = _LMTrainFunction.apply(self, pt, labels, self.lm_head.weight)
        else:
            with torch.no_grad():
                xN, _ = self._run_layers(pt, save=None)
                loss, _ = self._cos_head_loss(xN, labels, with_grad=False)
        return CausalLMOutput(loss=loss, logits=self._last_pred.view(B, S, 8, 2))

    def _cos_head_loss(self, xN, labels, with_grad: bool, grad_buf=None):
        """16-wide head, unit-normalised pairs (kept in self._last_pred, fp32 [T,16]) and the cosine loss. With
        with_grad, also returns d(loss)/d(xN) (bf16) and leaves d(loss)/d(W_head) in grad_buf ([16, d] fp32), both
        for upstream grad 1."""
        cfg = self.cfg
        T, d = xN.shape[0], cfg.hidden_size
        dev = xN.device
        W = self._w("lm_head.weight")
        xN = ops.presplit(xN)
        pairs = torch.empty(T, POS4D_HEAD_OUT, dtype=self.act_dtype, device=dev)
        ops.gemm_any(xN, W, pairs, T, POS4D_HEAD_OUT, d, xN.stride(0), W.stride(0), POS4D_HEAD_OUT)
        pred = torch.empty(T, POS4D_HEAD_OUT, dtype=torch.float32, device=dev)
        self._last_pred = pred
        self._last_labels4d = labels
        if labels is None:
            ops.cos4d_fwd_bwd(pairs, None, None, 1.0, pred, None, None)
            return None, None
        inv_count = torch.empty(1, dtype=torch.float32, device=dev)
        ops.cos4d_count(labels, inv_count)
        row_loss = torch.empty(T, dtype=torch.float32, device=dev)
        dpairs = torch.empty(T, POS4D_HEAD_OUT, dtype=self.act_dtype, device=dev) if with_grad else None
        ops.cos4d_fwd_bwd(pairs, labels, inv_count, 1.0, pred, row_loss, dpairs)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        ops.ce_finalize(row_loss, inv_count, loss)
        dxN = None
        if with_grad:
            dxN = torch.empty(T, d, dtype=self.act_dtype, device=dev)
            dps = ops.presplit(dpairs)
            ops.gemm_any(dps, W, dxN, T, d, POS4D_HEAD_OUT, POS4D_HEAD_OUT, W.stride(0), d, False, True, ops.OUT_BF16)
            grad_buf.zero_()   # accumulate mode lets the [16, d] output use the split-K path over T
            ops.gemm_any(dps, xN, grad_buf, POS4D_HEAD_OUT, d, T, POS4D_HEAD_OUT, xN.stride(0), d, True, True,
                         ops.OUT_F32_ACC)
        return loss, dxN

    COMPONENT_LOSS_NAMES = ("loss/s_0", "loss/s_1", "loss/d_0", "loss/d_1", "loss/p_0", "loss/p_1", "loss/v_0",
                            "loss/v_1")

    def component_losses(self):
        """The eight per-pair losses of the last labelled forward of the 4-D model, under the names the reference logs
        them with (/root/reference/src/model/modeling.py:646-655): fine and coarse pair of start, duration, pitch,
        velocity; fp32 [8] on the device. Their mean is the loss of that forward."""
        if not self.cfg.pos4d:
            raise ValueError("component_losses: only the 4-D position model has per-component losses")
        if getattr(self, "_last_labels4d", None) is None:
            raise RuntimeError("component_losses: no forward with labels has run")
        out = torch.empty(8, dtype=torch.float32, device=self._last_pred.device)
        ops.cos4d_components(self._last_pred, self._last_labels4d, out)
        return dict(zip(self.COMPONENT_LOSS_NAMES, out.unbind(0)))

    def _head_grad_shape(self):
        return (POS4D_HEAD_OUT if self.cfg.pos4d else self.cfg.vocab_size, self.cfg.hidden_size)

    def _head_loss(self, xN, labels, with_grad: bool, grad_buf=None):
        if self.cfg.pos4d:
            return self._cos_head_loss(xN, labels, with_grad, grad_buf)
        return self._lm_head_loss(xN, labels, with_grad, grad_buf)

    def _logits(self, hN):
        cfg = self.cfg
        V = cfg.vocab_size
        W = self._w(self._lm_head_name())
        T = hN.shape[0]
        ld = (V + 7) // 8 * 8
        buf = torch.empty(T, ld, dtype=self.act_dtype, device=hN.device)
        ops.gemm_any(hN, W, buf, T, V, cfg.hidden_size, hN.stride(0), W.stride(0), ld)
        return buf[:, :V]

    def _run_layers(self, inp, save: Optional[list], pos: Optional[torch.Tensor] = None, cache=None):
        """inp: token ids int64 [B,S], or note rows fp32 [B,S,4] for the 4-D position model."""
        cfg = self.cfg
        B, S = inp.shape[:2]
        T = B * S
        d, H, KVH, Dh, I = (cfg.hidden_size, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim,
                            cfg.intermediate_size)
        dev = inp.device
        bf = self.act_dtype                     # dtype of the activations the GEMMs read and write
        f32 = bf == torch.float32
        esz = 4 if f32 else 2                   # bytes per element, for the pointer offsets into qkv / qk
        NQ = cfg.qkv_width
        QK = (H + KVH) * Dh
        scale = Dh ** -0.5
        h = torch.empty(T, d, dtype=torch.float32, device=dev)
        if cfg.pos4d:
            cs = self._rope4d_table(inp.view(T, 4))
            ops.fill_rows(self.model.embed_vector, h)
        else:
            cs = self._rope_table(S)
            ops.embedding_fwd(inp, self.model.embed_tokens.weight, h, self._err_flag)
        lo = self.lora
        lkeys = lo.drop_keys(self.training and cache is None and pos is None) if lo is not None else None
        # Split residual projections: o_proj and down_proj as bf16-output products through the transposed weight copies
        # (reduction-contiguous tokens against [reduction][512] weights: the fastest form of these shapes), and the
        # residual addition in the same pass as the RMSNorm that follows. Same bits as the fp32 residual epilogue
        # followed by the norm; taken when the products fill the output-stationary kernel's whole tiles. (Hidden size
        # 512 only, where the copies exist for the backward anyway: with copies made for it the 650M model's step
        # measured 176.7-177.6 against 177.6-177.9 ms at 8 x 4096 tokens and the same at 1 x 16 384.)
        wt = None
        if (_SPLIT_PROJ and lo is None and not f32 and not cfg.pos4d and not self._ln and T % 256 == 0
                and (T // 256) * (d // 256) >= 256):
            wt = self._dgrad_weights_t()
        nxt = None   # (x1, rstd1) of the coming layer when the previous layer's last kernel has formed them
        for l, v in enumerate(self._layers):
            la = lo.layers[l] if lo is not None else None
            lk = lkeys[l] if lkeys is not None else (None,) * 5   # adapter dropout keys: qkv, o, gate, up, down
            if nxt is None:
                x1 = torch.empty(T, d, dtype=bf, device=dev)
                rstd1 = self._norm_fwd(h, v.w_ln1, v.w_ln1b, x1)
            else:
                x1, rstd1 = nxt
            # (training without adapters: 128 bytes of row padding, so that the 6 KiB pitch of the rows the QKV product
            # stores by column segments is not a multiple of 2 KiB - 210 -> 179 us for that product)
            qpad = _ROW_PAD if (save is not None and lo is None and not f32 and not cfg.pos4d and cache is None
                                and (NQ * esz) % 2048 == 0) else 0
            qkv = torch.empty(T, NQ + qpad, dtype=bf, device=dev)[:, :NQ]
            LQ = qkv.stride(0)
            fused = la is not None   # adapters: y = [x | t] [W | B]^T, t = s dropout(x) A^T
            if fused:
                t_qkv = lo.fwd_fused(x1, v.s_qkv, la, "qkv", qkv, d, NQ, key=lk[0])
            else:
                ops.linear_fwd(x1, v.s_qkv, qkv)
            if cfg.pos4d:
                # q, k (normalised) and v all rotated: one [T, NQ] buffer in the layout of qkv
                qk = torch.empty(T, NQ, dtype=bf, device=dev)
                qp, kp, vp = qk.data_ptr(), qk.data_ptr() + esz * H * Dh, qk.data_ptr() + esz * QK
                sq = sk = sv = NQ
                ops.qkvnorm_rope4d_fwd(qkv.data_ptr(), NQ, qp, kp, vp, NQ, NQ, NQ, v.w_qn, v.w_kn, cs, T, H, KVH, KVH,
                                       Dh, cfg.rms_norm_eps, is_bf16=not f32)
            else:
                qk = torch.empty(T, QK + (qpad if (QK * esz) % 2048 == 0 else 0), dtype=bf, device=dev)[:, :QK]
                LK = qk.stride(0)
                qp, kp, vp = qk.data_ptr(), qk.data_ptr() + esz * H * Dh, qkv.data_ptr() + esz * QK
                sq, sk, sv = LK, LK, LQ
                ops.qknorm_rope_fwd(qkv, LQ, qp, kp, LK, LK, v.w_qn, v.w_kn, cs, pos, S, T, H, KVH, Dh,
                                    cfg.rms_norm_eps, interleaved=cfg.rope_interleaved)
            if cache is not None:  # prefill: rotated keys and values of every prompt position go into the cache
                ops.kv_append(kp, vp, sk, sv, cache.k[l], cache.v[l], cache.zero_pos, B, S, KVH, Dh, cache.max_len)
            apad = _ATT_PAD if (qpad and (H * Dh * esz) % 2048 == 0) else 0
            att = torch.empty(T, H * Dh + apad, dtype=bf, device=dev)[:, :H * Dh]
            lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
            ops.attn_fwd(qp, kp, vp, att.data_ptr(), lse, sq, sk, sv, att.stride(0), B, S, H, KVH, Dh, scale, f32=f32)
            att_o = att
            if cfg.pos4d:  # the attention output is rotated back by each query's own angles before o_proj
                att_o = torch.empty(T, H * Dh, dtype=bf, device=dev)
                ops.rotate4d(att, att_o, cs, T, H, Dh, inverse=True)
            h_mid = torch.empty(T, d, dtype=torch.float32, device=dev)
            x2 = torch.empty(T, d, dtype=bf, device=dev)
            if wt is not None:
                yb = torch.empty(T, d, dtype=bf, device=dev)
                # att_o @ W_o^T with W_o^T stored [H * Dh][d]: the input-gradient form of the GEMM
                ops.gemm(att_o, wt[0][l], yb, T, d, H * Dh, att_o.stride(0), d, d, False, True, ops.OUT_BF16)
                rstd2 = torch.empty(T, dtype=torch.float32, device=dev)
                ops.resid_rmsnorm_fwd(h, yb, v.w_ln2, h_mid, x2, rstd2, self._norm_eps)
            else:
                if fused:
                    t_o = lo.fwd_fused(att_o, v.s_o, la, "o", h_mid, H * Dh, d, ops.OUT_F32_RESID_BF16ROUND, h,
                                       key=lk[1])
                else:
                    ops.linear_fwd(att_o, v.s_o, h_mid, ops.OUT_F32_RESID_BF16ROUND, h)
                rstd2 = self._norm_fwd(h_mid, v.w_ln2, v.w_ln2b, x2)
            gu = torch.empty(T, 2 * I, dtype=bf, device=dev)
            act = torch.empty(T, I, dtype=bf, device=dev)   # (row padding measured here too: 0.3-0.8 ms slower)
            if fused:   # t = [t_gate | t_up], then one product with the block-diagonal [B_gate, B_up] behind [Wg; Wu]
                t_gu = lo.fwd_fused_gu(x2, v.s_gu, la, gu, d, I, keys=(lk[2], lk[3]))
                ops.swiglu_fwd(gu, act)
            elif f32 or la is not None or not ops.linear_swiglu_fwd(x2, v.s_gu, gu, act):
                ops.linear_fwd(x2, v.s_gu, gu)
                ops.swiglu_fwd(gu, act)
            h_out = torch.empty(T, d, dtype=torch.float32, device=dev)
            if wt is not None:
                ops.gemm(act, wt[1][l], yb, T, d, I, act.stride(0), d, d, False, True, ops.OUT_BF16)
                last = l + 1 == len(self._layers)
                xn = torch.empty(T, d, dtype=bf, device=dev)
                rn = torch.empty(T, dtype=torch.float32, device=dev)
                ops.resid_rmsnorm_fwd(h_mid, yb, self.model.norm.weight if last else self._layers[l + 1].w_ln1, h_out,
                                      xn, rn, self._norm_eps)
                nxt = (xn, rn)
            elif fused:
                t_down = lo.fwd_fused(act, v.s_down, la, "down", h_out, I, d, ops.OUT_F32_RESID_BF16ROUND, h_mid,
                                      key=lk[4])
            else:
                ops.linear_fwd(act, v.s_down, h_out, ops.OUT_F32_RESID_BF16ROUND, h_mid)
            if save is not None:
                save.append((h, x1, rstd1, qkv, qk, att, lse, h_mid, x2, rstd2, gu, act, att_o,
                             (t_qkv, t_o, t_gu, t_down, lk) if la is not None else None))
            h = h_out
        if nxt is not None:
            xN, rstdN = nxt
        else:
            xN = torch.empty(T, d, dtype=bf, device=dev)
            rstdN = self._norm_fwd(h, self.model.norm.weight, self._final_norm_bias(), xN)
        return xN, (h, rstdN, cs)

    def _lm_head_loss(self, xN, labels, with_grad: bool, grad_buf=None):
        """Fused lm_head + shifted cross entropy over chunks of tokens. With with_grad, also returns
        d(loss)/d(xN) (bf16) and leaves d(loss)/d(W_lm_head) in grad_buf ([V, d] fp32; not needed with adapters),
        both for upstream grad 1."""
        cfg = self.cfg
        B, S = labels.shape
        T = B * S
        V, d = cfg.vocab_size, cfg.hidden_size
        dev = xN.device
        W = self._w(self._lm_head_name())
        ld = (V + 7) // 8 * 8
        chunk = min(cfg.lm_head_chunk_tokens, T)
        if self.act_dtype == torch.bfloat16 and os.environ.get("MGEN_LMHEAD_C", "1") != "0":
            # the chunk loop runs inside the library (mgen_lmhead_ce_fwd_bwd); the fp32 parity mode below keeps the
            # same protocol written out with its three-product GEMMs
            ws = ops.lmhead_ce_workspace(T, V, chunk, dev)
            loss = torch.empty((), dtype=torch.float32, device=dev)
            dxN = torch.empty(T, d, dtype=torch.bfloat16, device=dev) if with_grad else None
            ops.lmhead_ce_fwd_bwd(xN, W, labels, V, chunk, ws, loss, dxN,
                                  grad_buf if (with_grad and self.lora is None) else None, self._err_flag)
            return loss, dxN
        inv_count = torch.empty(1, dtype=torch.float32, device=dev)
        ops.ce_count(labels, inv_count, V=V, err_flag=self._err_flag)
        row_loss = torch.empty(T, dtype=torch.float32, device=dev)
        logits = torch.empty(chunk, ld, dtype=self.act_dtype, device=dev)
        dxN = torch.empty(T, d, dtype=self.act_dtype, device=dev) if with_grad else None
        first = True
        for t0 in range(0, T, chunk):
            n = min(chunk, T - t0)
            xc = ops.presplit(xN[t0:t0 + n])
            ops.gemm_any(xc, W, logits, n, V, d, xc.stride(0), W.stride(0), ld)
            ops.ce_fwd_bwd(logits, labels, t0, n, V, ld, S, inv_count, 1.0, row_loss, None, write_grad=with_grad)
            if with_grad:
                # dxN[c] = dlogits[c] @ W ; dW += dlogits[c]^T @ xN[c]
                dl = ops.presplit(logits)
                ops.gemm_any(dl, W, dxN[t0:t0 + n], n, d, V, ld, W.stride(0), d, False, True, ops.OUT_BF16)
                # the weight-gradient product: one launch per chunk; the first stores, every later one is added (the
                # kernel walks the token range in segments of less than 4 GiB of dlogits rows)
                if self.lora is not None:
                    pass   # frozen base: no weight gradient for the head
                elif isinstance(dl, ops.W2):
                    ops.gemm_any(dl, xc, grad_buf, V, d, n, ld, xc.stride(0), d, True, True,
                                 ops.OUT_F32 if first else ops.OUT_F32_ACC)
                    first = False
                else:
                    r0 = 0
                    while r0 < n:
                        rn = min(8192, n - r0) if _LM_WGRAD_8K else n - r0
                        ops.gemm(dl[r0:r0 + rn], xc[r0:r0 + rn], grad_buf, V, d, rn, ld, xc.stride(0), d,
                                 True, True, ops.OUT_F32 if first else ops.OUT_F32_ACC)
                        first = False
                        r0 += rn
        loss = torch.empty((), dtype=torch.float32, device=dev)
        ops.ce_finalize(row_loss, inv_count, loss)
        return loss, dxN

    # ------------------------------------------------------------------------------------------------
    # backward through the layer stack; gradients are ACCUMULATED into flat_grad
    # ------------------------------------------------------------------------------------------------
    def _backward_layers(self, inp, saved, final, dxN, grad_hook=None):
        cfg = self.cfg
        B, S = inp.shape[:2]
        T = B * S
        d, H, KVH, Dh, I = (cfg.hidden_size, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim,
                            cfg.intermediate_size)
        dev = inp.device
        bf = self.act_dtype
        f32 = bf == torch.float32
        esz = 4 if f32 else 2
        NQ = cfg.qkv_width
        QK = (H + KVH) * Dh
        scale = Dh ** -0.5
        h_last, rstdN, cs = final
        lo = self.lora
        dh = torch.empty(T, d, dtype=torch.float32, device=dev)
        dh_lp = torch.empty(T, d, dtype=bf, device=dev)
        # Without adapters the four weight gradients of a layer run as ONE launch at the end of the layer (a quarter of
        # the reduction slices and atomics of four launches): the low-precision residual gradient the MLP branch saw
        # must then outlive the one the attention branch sees, hence a second buffer.
        group = lo is None and not f32
        dh_lp_att = torch.empty(T, d, dtype=bf, device=dev) if group else dh_lp
        g_norm = self._view(self.flat_grad, "model.norm.weight")
        self._norm_bwd(dxN, h_last, self.model.norm.weight, rstdN, dh, dh_lp, g_norm,
                       self._final_norm_bias(self.flat_grad), accumulate=False)
        if grad_hook is not None:
            grad_hook("head")
        # Gradient activations that the GEMM kernels read or write by column segments of many rows at a time get 64
        # elements of padding per row when their pitch would be a multiple of 2 KiB (all those rows would sit on the
        # same memory channels; measured on the gate/up input gradient: 322 us at 8192 bytes, 278 us at 8320).
        pad = _ROW_PAD if (lo is None and not f32) else 0
        padded = lambda cols: torch.empty(T, cols + (pad if (cols * esz) % 2048 == 0 else 0), dtype=bf,
                                          device=dev)[:, :cols]
        dact = torch.empty(T, I, dtype=bf, device=dev)   # (written by the register-panel GEMM, which rotates instead)
        dgu = padded(2 * I)
        dx = torch.empty(T, d, dtype=bf, device=dev)
        datt = torch.empty(T, H * Dh + (_ATT_PAD if pad and not cfg.pos4d and (H * Dh * esz) % 2048 == 0 else 0), dtype=bf,
                           device=dev)[:, :H * Dh]
        dqkv = padded(NQ)
        LDQ = dqkv.stride(0)
        delta = torch.empty(B, H, S, dtype=torch.float32, device=dev)
        wt = self._dgrad_weights_t() if (lo is None and not f32) else None
        for l in range(cfg.num_hidden_layers - 1, -1, -1):
            v = self._layers[l]
            h_in, x1, rstd1, qkv, qk, att, lse, h_mid, x2, rstd2, gu, act, att_o, lts = saved[l]
            la = lo.layers[l] if lo is not None else None   # adapters: the base weights are frozen (no wgrad)
            fused = la is not None and lts[4][0] is None   # adapters without dropout (as in the forward pass)
            # ---- MLP branch
            if fused:
                lo.bwd_fused(dh_lp, act, lts[3], v.s_down, la, "down", dact, I, d)
            elif wt is not None:
                ops.linear_dgrad_t(dh_lp, wt[1][l], dact)
            else:
                ops.linear_dgrad(dh_lp, v.s_down, dact)
            if la is None and not group:
                ops.linear_wgrad(dh_lp, act, v.g_down)
            elif la is not None and not fused:
                lo.bwd(dh_lp, act, lts[3], la["down"].sA, la["down"].sB, la["down"].gA, la["down"].gB, dact, I, d,
                       key=lts[4][4])
            ops.swiglu_bwd(gu, dact, dgu)
            if fused:
                lo.bwd_fused_gu(dgu, x2, lts[2], v.s_gu, la, dx, d, I)
            else:
                ops.linear_dgrad(dgu, v.s_gu, dx)
            if la is None and not group:
                ops.linear_wgrad(dgu, x2, v.g_gu)
            elif la is not None and not fused:
                r_ = lo.r
                lo.bwd(dgu[:, :I], x2, lts[2][:, :r_], la["gate"].sA, la["gate"].sB, la["gate"].gA, la["gate"].gB,
                       dx, d, I, key=lts[4][2])
                lo.bwd(dgu[:, I:], x2, lts[2][:, r_:2 * r_], la["up"].sA, la["up"].sB, la["up"].gA, la["up"].gB, dx, d, I,
                       key=lts[4][3])
            dh_mlp = dh_lp            # what the MLP branch's weight gradients were taken from
            dh_lp = dh_lp_att         # (the same buffer unless the weight gradients are grouped)
            self._norm_bwd(dx, h_mid, v.w_ln2, rstd2, dh, dh_lp, v.g_ln2, v.g_ln2b, accumulate=True)
            # ---- attention branch
            if fused:
                lo.bwd_fused(dh_lp, att_o, lts[1], v.s_o, la, "o", datt, H * Dh, d)
            elif wt is not None:
                ops.linear_dgrad_t(dh_lp, wt[0][l], datt)
            else:
                ops.linear_dgrad(dh_lp, v.s_o, datt)
            if la is None and not group:
                ops.linear_wgrad(dh_lp, att_o, v.g_o)
            elif la is not None and not fused:
                lo.bwd(dh_lp, att_o, lts[1], la["o"].sA, la["o"].sB, la["o"].gA, la["o"].gB, datt, H * Dh, d,
                       key=lts[4][1])
            dqp, dkp, dvp = dqkv.data_ptr(), dqkv.data_ptr() + esz * H * Dh, dqkv.data_ptr() + esz * QK
            if cfg.pos4d:
                # datt is the gradient of the rotated-back output: rotate it forward to reach the attention output
                ops.rotate4d(datt, datt, cs, T, H, Dh, inverse=False)
                qp, kp, vp = qk.data_ptr(), qk.data_ptr() + esz * H * Dh, qk.data_ptr() + esz * QK
                sq = sk = sv = NQ
            else:
                qp, kp, vp = qk.data_ptr(), qk.data_ptr() + esz * H * Dh, qkv.data_ptr() + esz * QK
                sq, sk, sv = qk.stride(0), qk.stride(0), qkv.stride(0)
            ops.attn_bwd(qp, kp, vp, att.data_ptr(), datt.data_ptr(), lse, delta, dqp, dkp, dvp,
                         sq, sk, sv, att.stride(0), datt.stride(0), LDQ, LDQ, LDQ, B, S, H, KVH, Dh, scale, f32=f32)
            if cfg.pos4d:
                ops.qkvnorm_rope4d_bwd(dqp, LDQ, dqp, dkp, dvp, LDQ, LDQ, LDQ, qkv, qkv.stride(0), v.w_qn, v.w_kn, v.g_qn, v.g_kn,
                                       cs, T, H, KVH, KVH, Dh, cfg.rms_norm_eps, is_bf16=not f32)
            else:
                ops.qknorm_rope_bwd(dqp, LDQ, dqp, dkp, LDQ, LDQ, qkv, qkv.stride(0), v.w_qn, v.w_kn, v.g_qn, v.g_kn, cs, None, S,
                                    T, H, KVH, Dh, cfg.rms_norm_eps, is_bf16=not f32, interleaved=cfg.rope_interleaved)
            if fused:
                lo.bwd_fused(dqkv, x1, lts[0], v.s_qkv, la, "qkv", dx, d, NQ)
            else:
                ops.linear_dgrad(dqkv, v.s_qkv, dx)
            if la is None and not group:
                ops.linear_wgrad(dqkv, x1, v.g_qkv)
            elif la is not None and not fused:
                lo.bwd(dqkv, x1, lts[0], la["qkv"].sA, la["qkv"].sB, la["qkv"].gA, la["qkv"].gB, dx, d, NQ,
                       key=lts[4][0])
            if group:
                ops.linear_wgrad_group([(dh_mlp, act, v.g_down), (dgu, x2, v.g_gu), (dh_lp, att_o, v.g_o),
                                        (dqkv, x1, v.g_qkv)])
            dh_lp = dh_mlp            # the next layer's MLP branch reads (and keeps) this one
            self._norm_bwd(dx, h_in, v.w_ln1, rstd1, dh, dh_lp if l > 0 else None, v.g_ln1, v.g_ln1b, accumulate=True)
            saved[l] = None
            if grad_hook is not None:
                grad_hook(l)
        if not cfg.pos4d and lo is None:  # (4-D model: constant input; adapters: frozen embedding)
            ops.embedding_bwd(inp, dh, self._view(self.flat_grad, "model.embed_tokens.weight"))
            if grad_hook is not None:
                grad_hook("embed")

    # layout helpers for the data-parallel reducer -------------------------------------------------------
    def grad_ranges(self):
        """Contiguous [start, end) slices of flat_grad in the order backward finishes them: head (final norm and an
        untied lm_head), layers L-1..0, embedding."""
        cfg = self.cfg
        o = self._offsets
        total = self._total
        out = [("head", o["model.norm.weight"], total)]
        for l in range(cfg.num_hidden_layers - 1, -1, -1):
            start = o[f"model.layers.{l}.input_layernorm.weight"]
            end = o[f"model.layers.{l + 1}.input_layernorm.weight"] if l + 1 < cfg.num_hidden_layers else \
                o["model.norm.weight"]
            out.append((l, start, end))
        if not cfg.pos4d:
            out.append(("embed", 0, o["model.layers.0.input_layernorm.weight"] if cfg.num_hidden_layers else
                        o["model.norm.weight"]))
        return out


class _GraphToken:
    """Identity of one forward's autograd graph (see MidiLM._head_grad_buffer)."""

    def __init__(self):
        self.done = False


class _LMTrainFunction(torch.autograd.Function):
    """The whole network as one autograd node (see module docstring)."""

    @staticmethod
    def forward(ctx, model: MidiLM, input_ids, labels, _anchor):
        saved: list = []
        xN, final = model._run_layers(input_ids, save=saved)
        ctx.token = _GraphToken()
        ctx.lm_grad = None if model.lora is not None else model._head_grad_buffer(model._head_grad_shape(), ctx.token)
        loss, dxN = model._head_loss(xN, labels, with_grad=True, grad_buf=ctx.lm_grad)
        ctx.model, ctx.input_ids, ctx.saved, ctx.final, ctx.dxN = model, input_ids, saved, final, dxN
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        model: MidiLM = ctx.model
        if ctx.saved is None:
            raise RuntimeError("MidiLM backward called twice on the same forward")
        # the upstream gradient stays on the device: everything below is linear in it
        go = grad_out.to(torch.float32).contiguous()
        dxN = ctx.dxN
        if dxN.dtype == torch.float32:
            ops.scale_(dxN, 1.0, go)
        else:
            ops.scale_bf16_(dxN, 1.0, go)
        if ctx.lm_grad is not None:   # (frozen base: the head's weight gradient was not formed)
            name = model._lm_head_name()
            sink = getattr(model, "_head_grad_sink", None)
            if sink is not None and model.cfg.tie_word_embeddings and not model.cfg.pos4d:
                # data parallel, tied head: the head's weight gradient is final NOW, at the start of backward, while
                # the embedding lookup's share of the same tensor only arrives at its end. It is handed over as a
                # gradient bucket of its own (reduced under the whole backward) and added to the reduced embedding
                # share by whoever took it.
                ops.scale_(ctx.lm_grad.view(-1), 1.0, go)
                sink(ctx.lm_grad, model._view(model.flat_grad, name))
            else:
                ops.axpy_(model._view(model.flat_grad, name).view(-1), ctx.lm_grad.view(-1), 1.0, go)
        model._backward_layers(ctx.input_ids, ctx.saved, ctx.final, dxN, getattr(model, "_grad_hook", None))
        ctx.saved = None
        ctx.lm_grad = None
        ctx.token.done = True
        model._check_device_flags_deferred()
        for pname, prm in model.named_parameters():
            if prm.grad is None:
                prm.grad = model._view(model.flat_grad, pname)
        return None, None, None, None
