"""Low-rank adapters (LoRA) on the decoder's projections: the other training mode of the reference.

Reference: `MidiAria.to_lora` (/root/reference/src/model/model.py:251-271) freezes every base weight and wraps the
model with peft's `LoraConfig(r=32, lora_alpha=64, lora_dropout=0.05, bias="none")` on the fused QKV projection, the
attention output projection and the gate / up / down projections. peft's linear layer computes
`base(x) + lora_B(lora_A(dropout(x))) * (alpha / r)` with A ~ kaiming_uniform(a=sqrt(5)) = U(-1/sqrt(in), 1/sqrt(in))
and B = 0. peft is not installed here: the arithmetic is restated from that formula (parity unpinned). The adapter
dropout (`dropout=`, training mode only) masks the adapter's input with a hash of (key, element index); every site of
every forward has its own key and the backward pass recomputes the masks, as peft's per-layer Dropout modules draw
independent masks. With dropout on, the gate and up adapters no longer share one product (their inputs differ).

Here the adapters live in one small flat fp32 buffer (+ gradient, + bf16 shadow) next to the frozen model. Without
dropout a site's forward is t = s * x A^T (N = r) and then ONE product y = [x | t] [W | B]^T whose reduction runs over
the base weight and then over the adapter (`mgen_gemm_bf16_seg`; the rank is padded to 64 with zeros); its backward
is dt = s * dy B, dB += dy^T t, dA += dt^T x and ONE product dx = [dy | dt] [W ; A]. With dropout the adapter sees a
masked input, so its products stay separate (y += t B^T through the accumulate epilogue, dx += dropout'(dt A)).
The base weights get no gradient products at all.
"""
import math

import torch

from . import ops

SITES = ("qkv", "o", "gate", "up", "down")


def _up64(n):
    return (n + 63) // 64 * 64
_NAMES = {"qkv": "self_attn.qkv_proj", "o": "self_attn.o_proj", "gate": "mlp.gate_proj", "up": "mlp.up_proj",
          "down": "mlp.down_proj"}


class _Site:
    __slots__ = ("A", "B", "gA", "gB", "sA", "sB")


class LoraAdapters:
    def __init__(self, model, r=32, alpha=64, seed=0, dropout=0.0):
        cfg = model.cfg
        if r % 8 or r <= 0:
            raise ValueError("the adapter rank must be a positive multiple of 8")
        if cfg.arch != "llama" or cfg.pos4d or model.act_dtype != torch.bfloat16:
            raise NotImplementedError("adapters cover the bf16-mixed token models with RMSNorm/RoPE/SwiGLU blocks")
        if not 0.0 <= dropout < 1.0:
            raise ValueError("the adapter dropout probability must be in [0, 1)")
        self.r, self.alpha, self.scale = r, alpha, alpha / r
        self.dropout_p, self.dropout_seed, self._dropout_calls = float(dropout), int(seed), 0
        self._scratch = {}
        d, I = cfg.hidden_size, cfg.intermediate_size
        NQ, HD = cfg.qkv_width, cfg.num_attention_heads * cfg.head_dim
        # (in, out) of every site; gate and up sit back to back so that one product with [A_gate; A_up] serves both
        self.dims = {"qkv": (d, NQ), "o": (HD, d), "gate": (d, I), "up": (d, I), "down": (I, d)}
        per_layer = sum(r * i + o * r for i, o in self.dims.values())
        L = cfg.num_hidden_layers
        dev = model.flat_param.device
        self.flat_param = torch.zeros(L * per_layer, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros_like(self.flat_param)
        self.flat_shadow = torch.zeros(L * per_layer, dtype=torch.bfloat16, device=dev)
        self._shadow_version = -1
        self.layers = []
        self._names = {}
        off = 0
        g = torch.Generator(device=dev).manual_seed(seed)
        for l in range(L):
            sites = {}
            # A blocks first (gate and up adjacent), then the B blocks
            order = [("qkv", "A"), ("o", "A"), ("gate", "A"), ("up", "A"), ("down", "A"),
                     ("qkv", "B"), ("o", "B"), ("gate", "B"), ("up", "B"), ("down", "B")]
            for k in SITES:
                sites[k] = _Site()
            for k, which in order:
                i, o = self.dims[k]
                shape = (r, i) if which == "A" else (o, r)
                n = shape[0] * shape[1]
                for tag, flat in (("", self.flat_param), ("g", self.flat_grad), ("s", self.flat_shadow)):
                    setattr(sites[k], tag + which, flat[off:off + n].view(shape))
                if which == "A":
                    bound = 1.0 / math.sqrt(i)
                    sites[k].A.uniform_(-bound, bound, generator=g)
                self._names[f"model.layers.{l}.{_NAMES[k]}.lora_{which}.weight"] = getattr(sites[k], which)
                off += n
            # [A_gate; A_up] as one [2r, d] operand (the two blocks are adjacent in the flat buffers)
            sites["gu_sA"] = self.flat_shadow[sites["gate"].sA.storage_offset():][:2 * r * d].view(2 * r, d)
            sites["gu_gA"] = self.flat_grad[sites["gate"].gA.storage_offset():][:2 * r * d].view(2 * r, d)
            # zero-padded bf16 operands of the two-pair products: B as [out, Rp], A as [Rp, in]; gate and up share
            # one pair (block-diagonal B over [gate | up] outputs, [A_gate; A_up] stacked)
            Rp, Rp2 = _up64(r), _up64(2 * r)
            bf = torch.bfloat16
            pad = {}
            for k in ("qkv", "o", "down"):
                i, o = self.dims[k]
                pad[k] = (torch.zeros(Rp, i, dtype=bf, device=dev), torch.zeros(o, Rp, dtype=bf, device=dev))
            pad["gu"] = (torch.zeros(Rp2, d, dtype=bf, device=dev), torch.zeros(2 * I, Rp2, dtype=bf, device=dev))
            sites["pad"] = pad
            self.layers.append(sites)
        # a differentiable input for the model's autograd node while every base parameter is frozen
        self.anchor = torch.nn.Parameter(self.flat_param[:8])

    # ---- parameter store protocol shared with MidiLM (what FusedAdamW and the gradient reducer use) ----------
    def sync_shadow(self):
        if self._shadow_version != self.flat_param._version:
            ops.cast(self.flat_param, self.flat_shadow)
            self._refresh_padded()
            self._shadow_version = self.flat_param._version

    def mark_shadow_fresh(self):
        """The fused optimizer has rewritten master and bf16 shadow in one kernel."""
        self._refresh_padded()
        self._shadow_version = self.flat_param._version

    def _refresh_padded(self):
        r = self.r
        I = self.dims["gate"][1]
        for st in self.layers:
            for k in ("qkv", "o", "down"):
                pA, pB = st["pad"][k]
                pA[:r].copy_(st[k].sA)
                pB[:, :r].copy_(st[k].sB)
            pA, pB = st["pad"]["gu"]
            pA[:2 * r].copy_(st["gu_sA"])
            pB[:I, :r].copy_(st["gate"].sB)
            pB[I:, r:2 * r].copy_(st["up"].sB)

    def zero_grad(self):
        self.flat_grad.zero_()

    def state_dict(self):
        return {k: v.detach().clone() for k, v in self._names.items()}

    def load_state_dict(self, sd):
        missing = [k for k in self._names if k not in sd]
        if missing:
            raise RuntimeError(f"adapter state dict misses {missing[:4]} ...")
        with torch.no_grad():
            for k, v in self._names.items():
                v.copy_(sd[k].to(device=v.device, dtype=torch.float32))
        self._shadow_version = -1

    def named_parameters(self):
        return list(self._names.items())

    # ---- dropout ---------------------------------------------------------------------------------------------
    def drop_keys(self, training: bool):
        """Per layer, the five mask keys (SITES order) of this forward; None when the adapter dropout is off."""
        if not training or self.dropout_p == 0.0:
            return None
        n = len(SITES) * len(self.layers)
        base = self._dropout_calls * n
        self._dropout_calls += 1
        return [[ops.dropout_key(self.dropout_seed, base + len(SITES) * l + i) for i in range(len(SITES))]
                for l in range(len(self.layers))]

    def _buf(self, like):
        b = self._scratch.get(tuple(like.shape))
        if b is None or b.device != like.device:
            b = self._scratch[tuple(like.shape)] = torch.empty_like(like)
        return b

    def _dropped(self, x, key):
        xd = self._buf(x)
        ops.dropout(x, xd, self.dropout_p, key)
        return xd

    # ---- one site, adapter folded into the base product ------------------------------------------------------
    def fwd_fused(self, x, W, st, k, y, n_in, n_out, out_mode=ops.OUT_BF16, resid=None, key=None):
        """y = x W^T + (s dropout(x) A^T) B^T (+ resid) in two launches; returns t = s dropout(x) A^T as [T, Rp]
        (zero beyond r). key: the site's dropout key, None = no dropout."""
        T, r = x.shape[0], self.r
        pA, pB = st["pad"][k]
        Rp = pB.shape[1]
        t = (torch.zeros if Rp != r else torch.empty)(T, Rp, dtype=torch.bfloat16, device=x.device)
        xin = self._dropped(x, key) if key is not None else x
        ops.gemm(xin, st[k].sA, t, T, r, n_in, xin.stride(0), st[k].sA.stride(0), Rp, alpha=self.scale)
        ops.gemm_seg(x, W, t, pB, y, T, n_out, n_in, Rp, x.stride(0), W.stride(0), Rp, Rp, y.stride(0), False, out_mode,
                     resid, resid.stride(0) if resid is not None else 0)
        return t

    def fwd_fused_gu(self, x, W, st, gu, d, I, keys=(None, None)):
        """[gate | up] = x [Wg; Wu]^T + [t_g B_g^T | t_u B_u^T]; returns [t_g | t_u] as [T, Rp2]. keys: the dropout
        keys of the gate and up adapters (their inputs are then masked independently)."""
        T, r = x.shape[0], self.r
        pA, pB = st["pad"]["gu"]
        Rp2 = pB.shape[1]
        t = (torch.zeros if Rp2 != 2 * r else torch.empty)(T, Rp2, dtype=torch.bfloat16, device=x.device)
        if keys[0] is None:
            ops.gemm(x, st["gu_sA"], t, T, 2 * r, d, x.stride(0), st["gu_sA"].stride(0), Rp2, alpha=self.scale)
        else:
            for j, k in enumerate(("gate", "up")):
                xin = self._dropped(x, keys[j])
                ops.gemm(xin, st[k].sA, t[:, j * r:(j + 1) * r], T, r, d, xin.stride(0), st[k].sA.stride(0), Rp2,
                         alpha=self.scale)
        ops.gemm_seg(x, W, t, pB, gu, T, 2 * I, d, Rp2, x.stride(0), W.stride(0), Rp2, Rp2, gu.stride(0))
        return t

    def bwd_fused(self, dy, x, t, W, st, k, dx, n_in, n_out):
        """Adapter gradients of site k and dx = dy W + (s dy B) A in one product (dx is overwritten)."""
        T, r = dy.shape[0], self.r
        site = st[k]
        pA, pB = st["pad"][k]
        Rp = pA.shape[0]
        dt = (torch.zeros if Rp != r else torch.empty)(T, Rp, dtype=torch.bfloat16, device=dy.device)
        ops.gemm(dy, site.sB, dt, T, r, n_out, dy.stride(0), site.sB.stride(0), Rp, trans_b=True, alpha=self.scale)
        ops.gemm(dy, t, site.gB, n_out, r, T, dy.stride(0), t.stride(0), site.gB.stride(0), True, True, ops.OUT_F32_ACC)
        ops.gemm(dt, x, site.gA, r, n_in, T, Rp, x.stride(0), site.gA.stride(0), True, True, ops.OUT_F32_ACC)
        ops.gemm_seg(dy, W, dt, pA, dx, T, n_in, n_out, Rp, dy.stride(0), W.stride(0), Rp, pA.stride(0), dx.stride(0),
                     True, ops.OUT_BF16)

    def bwd_fused_gu(self, dgu, x, t, W, st, dx, d, I):
        T, r = dgu.shape[0], self.r
        pA, pB = st["pad"]["gu"]
        Rp2 = pA.shape[0]
        dt = (torch.zeros if Rp2 != 2 * r else torch.empty)(T, Rp2, dtype=torch.bfloat16, device=dgu.device)
        for j, k in enumerate(("gate", "up")):
            dy, site = dgu[:, j * I:(j + 1) * I], st[k]
            dtj, tj = dt[:, j * r:(j + 1) * r], t[:, j * r:(j + 1) * r]
            ops.gemm(dy, site.sB, dtj, T, r, I, dy.stride(0), site.sB.stride(0), Rp2, trans_b=True, alpha=self.scale)
            ops.gemm(dy, tj, site.gB, I, r, T, dy.stride(0), Rp2, site.gB.stride(0), True, True, ops.OUT_F32_ACC)
        ops.gemm(dt, x, st["gu_gA"], 2 * r, d, T, Rp2, x.stride(0), st["gu_gA"].stride(0), True, True, ops.OUT_F32_ACC)
        ops.gemm_seg(dgu, W, dt, pA, dx, T, d, 2 * I, Rp2, dgu.stride(0), W.stride(0), Rp2, pA.stride(0), dx.stride(0),
                     True, ops.OUT_BF16)

    # ---- one site ------------------------------------------------------------------------------------------
    def fwd(self, x, sA, sB, y, n_in, n_out, key=None, t=None):
        """t = s * dropout(x) A^T (returned, bf16 [T, rows of A]); y[:, :n_out] += t B^T. `y` is the base projection's
        output: bf16, or the fp32 residual stream. key: the site's dropout key (None = no dropout); t: where to put
        the product (a column block of a wider buffer) instead of a new tensor."""
        T = x.shape[0]
        rr = sA.shape[0]
        if key is not None:
            x = self._dropped(x, key)
        if t is None:
            t = torch.empty(T, rr, dtype=torch.bfloat16, device=x.device)
        ops.gemm(x, sA, t, T, rr, n_in, x.stride(0), sA.stride(0), t.stride(0), alpha=self.scale)
        if sB is not None:
            self.add_out(t, sB, y, n_out)
        return t

    def add_out(self, t, sB, y, n_out):
        T = t.shape[0]
        mode = ops.OUT_F32_ACC if y.dtype == torch.float32 else ops.OUT_BF16_ACC
        ops.gemm(t, sB, y, T, n_out, self.r, t.stride(0), sB.stride(0), y.stride(0), out_mode=mode)

    def bwd(self, dy, x, t, sA, sB, gA, gB, dx, n_in, n_out, key=None):
        """dy [T, n_out] (bf16), x [T, n_in], t from fwd. Accumulates gA, gB; adds the adapter's share to dx (which
        already holds dy W). key: the dropout key the forward call of this site used."""
        T = dy.shape[0]
        r = self.r
        dt = torch.empty(T, r, dtype=torch.bfloat16, device=dy.device)
        ops.gemm(dy, sB, dt, T, r, n_out, dy.stride(0), sB.stride(0), r, trans_b=True, alpha=self.scale)
        ops.gemm(dy, t, gB, n_out, r, T, dy.stride(0), t.stride(0), gB.stride(0), True, True, ops.OUT_F32_ACC)
        xin = self._dropped(x, key) if key is not None else x
        ops.gemm(dt, xin, gA, r, n_in, T, r, xin.stride(0), gA.stride(0), True, True, ops.OUT_F32_ACC)
        if dx is not None and key is None:
            ops.gemm(dt, sA, dx, T, n_in, r, r, sA.stride(0), dx.stride(0), False, True, ops.OUT_BF16_ACC)
        elif dx is not None:   # the adapter's input gradient passes the mask before it joins the base path's
            tmp = self._buf(x)     # (stream order: the product above has consumed the dropped input by now)
            ops.gemm(dt, sA, tmp, T, n_in, r, r, sA.stride(0), tmp.stride(0), False, True, ops.OUT_BF16)
            ops.dropout_add(tmp, dx, self.dropout_p, key)
        return dt
