"""Autoregressive sampling for MidiLM: `generate()` with a pre-allocated KV cache.

Call surface of the reference's sampling sites (`model.model.generate(input_ids, max_new_tokens | max_length,
do_sample, temperature, top_k, top_p, pad_token_id, eos_token_id, use_cache)`, /root/reference/src/test_tokenizer.py:
129-136 and notebooks/test.ipynb), i.e. transformers' GenerationMixin._sample (generation/utils.py:2783-2941):
prefill the prompt, then one token per iteration; temperature -> top-k -> top-p -> multinomial, or argmax; finished
rows are padded with pad_token_id; returns int64 [B, prompt + new]. As in transformers, `do_sample=True` without an
explicit top_k runs with top_k = 50.

MI355X design: the cache is allocated once per call ([B, KVH, max_len, D] bf16 per layer, head-major so that a
stream's keys are contiguous), every step runs M = B token rows through the same GEMM kernels as training plus the
split-key decode-attention kernel, and the sampler kernel writes the new ids straight into the output matrix.
Streams are independent, so a large batch is decoded as two halves on two HIP streams: a step is a chain of one long
HBM-bound kernel per layer (the attention over the cache) and ~9 short latency-bound ones (projections at M = B rows,
norms, sampler); with two chains in flight the short kernels of one half run under the attention of the other. The
multinomial draw uses explicit Exp(1) noise (argmax p/q): a `noise` tensor [steps, B, V] for parity checks, or a
counter-based Philox stream keyed by `seed` (offset = step).
"""
import os
import time
from typing import Optional

import torch

from . import ops


class KVCache:
    def __init__(self, cfg, batch: int, max_len: int, device):
        self.max_len = max_len
        shape = (batch, cfg.num_key_value_heads, max_len, cfg.head_dim)
        self.k = [torch.empty(shape, dtype=torch.bfloat16, device=device) for _ in range(cfg.num_hidden_layers)]
        self.v = [torch.empty(shape, dtype=torch.bfloat16, device=device) for _ in range(cfg.num_hidden_layers)]
        self.zero_pos = torch.zeros(batch, dtype=torch.int32, device=device)

    def bytes_per_token(self):
        return 2 * len(self.k) * self.k[0].shape[1] * self.k[0].shape[3] * 2

    def first_streams(self, batch: int):
        """The cache of the first `batch` streams (views; the row capacity, i.e. the pitch between heads, stays)."""
        if batch == self.k[0].shape[0]:
            return self
        return _CachePart(self, 0, batch)


class _CachePart:
    """Streams [b0, b1) of a KVCache (views: a stream's rows are contiguous)."""

    def __init__(self, cache: KVCache, b0: int, b1: int):
        self.max_len = cache.max_len
        self.k = [t[b0:b1] for t in cache.k]
        self.v = [t[b0:b1] for t in cache.v]
        self.zero_pos = cache.zero_pos[b0:b1]


class _Part:
    """One independently decoded part of the batch: streams [b0, b1), its buffers, cache views and HIP stream."""

    def __init__(self, cfg, cache, b0, b1, nxt, ws_floats, dev, stream):
        self.b0, self.b1, self.stream = b0, b1, stream
        self.cache = _CachePart(cache, b0, b1)
        self.buf = _DecodeBuffers(cfg, b1 - b0, dev, nxt[b0:b1])
        self.ws = torch.empty(max(ws_floats, 1), dtype=torch.float32, device=dev)


# batches of at least this many streams are decoded as two halves on two HIP streams (MGEN_DECODE_PARTS=1: never)
_TWO_PART_MIN_STREAMS = 64
# MGEN_DECODE_LAYER_C=0: issue the launches of a decode layer one by one from Python (for measurements)
_LAYER_IN_ONE_CALL = os.environ.get("MGEN_DECODE_LAYER_C", "1") != "0"
# MGEN_DECODE_GRAPH=1: replay every decode step from a captured HIP graph instead of issuing its launches one by one.
# Off by default: on this ROCm the replay of the ~200 kernel nodes of a two-part step costs the host as much as the
# launches themselves (1.31 against 1.25 ms per step) and the per-step join of the two parts costs the GPU time
# (contexts 1792-2048: 4.83 against 4.53 ms per step).
_DECODE_GRAPH = os.environ.get("MGEN_DECODE_GRAPH", "0") == "1"


class _DecodeBuffers:
    def __init__(self, cfg, B, dev, nxt=None):
        bf, f32 = torch.bfloat16, torch.float32
        d, I = cfg.hidden_size, cfg.intermediate_size
        H, KVH, Dh = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        self.h0 = torch.empty(B, d, dtype=f32, device=dev)
        self.h1 = torch.empty(B, d, dtype=f32, device=dev)
        self.h2 = torch.empty(B, d, dtype=f32, device=dev)
        self.x = torch.empty(B, d, dtype=bf, device=dev)
        self.qkv = torch.empty(B, cfg.qkv_width, dtype=bf, device=dev)
        self.qk = torch.empty(B, (H + KVH) * Dh, dtype=bf, device=dev)
        self.att = torch.empty(B, H * Dh, dtype=bf, device=dev)
        self.gu = torch.empty(B, 2 * I, dtype=bf, device=dev)
        self.act = torch.empty(B, I, dtype=bf, device=dev)
        self.ldv = (cfg.vocab_size + 7) // 8 * 8
        self.logits = torch.empty(B, self.ldv, dtype=bf, device=dev)
        self.next = nxt if nxt is not None else torch.empty(B, dtype=torch.int64, device=dev)
        # The four projections of a layer run on the one-round-trip kernel for few token rows when their shapes
        # fit it (512-long reduction slices); the down projection then forms silu(gate) * up while it loads gu.
        NQ, HD = cfg.qkv_width, H * Dh
        sk = not cfg.pos4d
        self.sk_qkv = sk and ops.skinny_fits(B, NQ, d, d, d, NQ, False)
        self.sk_gu = sk and ops.skinny_fits(B, 2 * I, d, d, d, 2 * I, False)
        self.sk_o = sk and HD % 512 == 0 and ops.skinny_fits(B, d, HD, HD, HD, d, ops.skinny_slabs(B, d, HD))
        self.sk_down = sk and I % 512 == 0 and ops.skinny_fits(B, d, I, 2 * I, I, d, ops.skinny_slabs(B, d, I))
        # split-K slabs of the two residual projections (K = H*Dh and K = I)
        self.ns_o = ops.skinny_slabs(B, d, HD) if self.sk_o else ops.gemm_slab_count(HD)
        self.ns_d = ops.skinny_slabs(B, d, I) if self.sk_down else ops.gemm_slab_count(I)
        self.slabs = torch.empty(max(self.ns_o, self.ns_d), B, d, dtype=f32, device=dev)
        self.layer_args = {}
        if cfg.pos4d:
            self.rot = torch.empty(B, cfg.qkv_width, dtype=bf, device=dev)
            self.pairs = torch.empty(B, 16, dtype=bf, device=dev)
            self.pred = torch.empty(B, 16, dtype=f32, device=dev)
            self.cur = torch.empty(B, 4, dtype=f32, device=dev)


class GenerationMixin:
    """Mixed into MidiLM."""

    def reserve_kv_cache(self, batch: int, max_len: int):
        """Serving: allocate the KV cache for `batch` streams of up to `max_len` tokens once and write to all of it, so
        that generate() calls that fit (batch and total length not larger) neither allocate nor take the first-touch
        page faults of fresh device memory (about 170 ms for the 60 GB of 256 streams x 4096 tokens of the default
        decoder) inside their first step. Held until release_kv_cache()."""
        dev = self.flat_shadow.device
        cache = KVCache(self.cfg, batch, max_len, dev)
        for t in cache.k + cache.v:
            t.zero_()
        self._kv_reserved = cache

    def release_kv_cache(self):
        self._kv_reserved = None

    def _decode_split(self, B, length):
        cfg = self.cfg
        blocks = B * cfg.num_attention_heads
        want = max(1, 1024 // max(blocks, 1))
        return max(1, min(64, want, (length + 255) // 256))

    def _decode_begin(self, tokens, buf: _DecodeBuffers):
        """Start of a one-token step: embedding row per stream and the first layer's input norm."""
        ops.embedding_fwd(tokens, self.model.embed_tokens.weight, buf.h0, self._err_flag)
        self._norm_fwd(buf.h0, self._layers[0].w_ln1, self._layers[0].w_ln1b, buf.x, want_stats=False)
        buf.h = buf.h0

    def _decode_layer(self, l: int, cache, t: int, buf: _DecodeBuffers, lens, ws):
        """Decoder layer l of a one-token step at position t (all streams share t). The residual stream is in buf.h,
        the normed input of the layer in buf.x."""
        cfg = self.cfg
        B = buf.x.shape[0]
        d, H, KVH, Dh, I = (cfg.hidden_size, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim,
                            cfg.intermediate_size)
        NQ, QK = cfg.qkv_width, (H + KVH) * Dh
        eps = cfg.rms_norm_eps
        L = cfg.num_hidden_layers
        v = self._layers[l]
        if isinstance(lens, tuple):
            pos, kv_len = lens     # live buffers a launch at the end of the step moves on (graph replay)
        else:
            pos = lens[t]          # int32 [>= B] = t
            kv_len = lens[t + 1]   # int32 [>= B] = t + 1
        h = buf.h
        if _LAYER_IN_ONE_CALL and buf.sk_qkv and buf.sk_gu and buf.sk_o and buf.sk_down and not self._ln:
            # every launch of the layer issued by one library call (the pointer arguments are collected once per
            # layer and buffer set: only the position, the split count and the stream change from step to step)
            key = (l, cache.k[l].data_ptr())
            fixed = buf.layer_args.get(key)
            nxt = buf.h2 if h is buf.h0 else buf.h0
            if fixed is None:
                w_next = self._layers[l + 1].w_ln1 if l + 1 < L else self.model.norm.weight
                fixed = buf.layer_args[key] = (
                    [t_.data_ptr() for t_ in (buf.x, h, buf.h1, nxt, buf.qkv, buf.att, buf.gu, buf.slabs, ws, v.s_qkv)]
                    + [0 if v.w_qn is None else v.w_qn.data_ptr(), 0 if v.w_kn is None else v.w_kn.data_ptr()]
                    + [t_.data_ptr() for t_ in (v.s_o, v.w_ln2, v.s_gu, v.s_down, w_next)],
                    [cache.k[l].data_ptr(), cache.v[l].data_ptr(), B, d, H, KVH, Dh, I, cache.max_len])
            ops.decode_layer(fixed[0], self._rope_cs.data_ptr(), pos.data_ptr(), kv_len.data_ptr(), fixed[1],
                             self._decode_split(B, t + 1), eps, int(cfg.rope_interleaved))
            buf.h = nxt
            return
        if buf.sk_qkv:
            ops.linear_skinny(buf.x, v.s_qkv, buf.qkv)
        else:
            ops.linear_fwd(buf.x, v.s_qkv, buf.qkv)
        # attention over the cache; q/k norm + rotation of the new token and its cache append happen in the same launch
        ops.attn_decode_prep(buf.qkv, NQ, v.w_qn, v.w_kn, self._rope_cs, pos, eps, cfg.rope_interleaved, cache.k[l],
                             cache.v[l], kv_len, buf.att, H * Dh, ws, B, H, KVH, Dh, cache.max_len,
                             self._decode_split(B, t + 1), Dh ** -0.5)
        # The two residual projections have few output tiles at M = B rows: they run split along K into fp32 slabs,
        # and one kernel sums the slabs, adds the residual and applies the norm that follows.
        if buf.sk_o:
            ops.linear_skinny(buf.att, v.s_o, buf.slabs, slabs=buf.ns_o)
        else:
            ops.linear_slabs(buf.att, v.s_o, buf.slabs, buf.ns_o)
        self._slab_resid_norm(buf.slabs, buf.ns_o, h, buf.h1, v.w_ln2, v.w_ln2b, buf.x)
        if buf.sk_gu:
            ops.linear_skinny(buf.x, v.s_gu, buf.gu)
        else:
            ops.linear_fwd(buf.x, v.s_gu, buf.gu)
        if buf.sk_down:
            ops.linear_skinny(buf.gu, v.s_down, buf.slabs, slabs=buf.ns_d, a_swiglu=True)
        else:
            ops.swiglu_fwd(buf.gu, buf.act)
            ops.linear_slabs(buf.act, v.s_down, buf.slabs, buf.ns_d)
        nxt = buf.h2 if h is buf.h0 else buf.h0
        w_next = self._layers[l + 1].w_ln1 if l + 1 < L else self.model.norm.weight
        b_next = self._layers[l + 1].w_ln1b if l + 1 < L else self._final_norm_bias()
        self._slab_resid_norm(buf.slabs, buf.ns_d, buf.h1, nxt, w_next, b_next, buf.x)
        buf.h = nxt

    def _slab_resid_norm(self, slabs, ns, resid, h_out, w, b, y):
        """h_out = resid + bf16(sum of the projection's slabs); y = the following norm of h_out."""
        if self._ln:   # LayerNorm blocks: the slab sum + residual in one launch, the norm in the next
            ops.slab_resid_rmsnorm(slabs, ns, resid, h_out, None, None, self._norm_eps)
            self._norm_fwd(h_out, w, b, y)
        else:
            ops.slab_resid_rmsnorm(slabs, ns, resid, h_out, w, y, self._norm_eps)

    def _decode_end(self, buf: _DecodeBuffers):
        """lm_head on the final-normed rows; returns bf16 logits [B, ldv]."""
        cfg = self.cfg
        W = self._view(self.flat_shadow, self._lm_head_name())
        ops.gemm(buf.x, W, buf.logits, buf.x.shape[0], cfg.vocab_size, cfg.hidden_size, cfg.hidden_size, W.stride(0),
                 buf.ldv)
        return buf.logits

    def _decode_step(self, tokens, cache: KVCache, t: int, buf: _DecodeBuffers, lens, ws):
        """One token per stream at position t (all streams share t); returns bf16 logits [B, ldv]."""
        self._rope_table(t + 1)
        self._decode_begin(tokens, buf)
        for l in range(self.cfg.num_hidden_layers):
            self._decode_layer(l, cache, t, buf, lens, ws)
        return self._decode_end(buf)

    @torch.no_grad()
    def generate(self, input_ids=None, max_new_tokens: Optional[int] = None, max_length: Optional[int] = None,
                 do_sample: bool = False, temperature: float = 1.0, top_k: Optional[int] = None, top_p: float = 1.0,
                 pad_token_id: Optional[int] = None, eos_token_id: Optional[int] = None, use_cache: bool = True,
                 noise: Optional[torch.Tensor] = None, seed: int = 0, eos_check_every: int = 16,
                 return_step_logits: bool = False, attention_mask: Optional[torch.Tensor] = None, **unused):
        """attention_mask [B, prompt_len] (1 = real token): prompts of different lengths batched with LEFT padding, as
        transformers' generate() expects of a decoder-only model (generation/utils.py: the mask grows by a column of
        ones per token, position ids count real tokens only). Each stream then produces what it would produce alone
        from its unpadded prompt: internally the rows are shifted to start at column 0, the prefill runs on the shifted
        batch (causal attention keeps the tail a shorter row carries out of its real tokens' sight) and every stream
        decodes at its own position. Right padding in generate() is refused (the reference library would continue
        such a row from a padding token)."""
        if input_ids is None or input_ids.dim() != 2:
            raise ValueError("generate() needs input_ids of shape [batch, prompt_len]")
        if not input_ids.is_cuda:
            raise RuntimeError("MidiLM runs on the GPU only: input_ids must be a CUDA/HIP tensor")
        if getattr(self, "lora", None) is not None and use_cache:
            # the one-token decode path reads the base weights only: fold the adapters in, or decode without a cache
            raise RuntimeError("the model carries low-rank adapters: call merge_lora() before generate(), or pass "
                               "use_cache=False")
        cfg = self.cfg
        B, P = input_ids.shape
        if max_new_tokens is None:
            max_new_tokens = (max_length if max_length is not None else 20) - P
        if max_new_tokens <= 0:
            return input_ids.clone()
        if do_sample:
            if not (isinstance(temperature, (int, float)) and temperature > 0):
                raise ValueError(f"`temperature` (={temperature}) has to be a strictly positive float")
            if top_k is None:
                top_k = 50  # transformers' generation default whenever sampling is on
            if top_k is not None and top_k < 0:
                raise ValueError(f"`top_k` has to be a non-negative integer, but is {top_k}")
            if not (0 < top_p <= 1.0):
                raise ValueError(f"`top_p` has to be a float > 0 and <= 1, but is {top_p}")
        if eos_token_id is not None and pad_token_id is None:
            pad_token_id = eos_token_id
        dev = input_ids.device
        V = cfg.vocab_size
        total = P + max_new_tokens
        self._check_device_flags()   # bits an earlier call left behind (a forward whose backward never ran) are raised
        self.sync_shadow()
        out = torch.empty(B, total, dtype=torch.int64, device=dev)
        out[:, :P] = input_ids
        # ---- left-padded prompts: n_real[b] real tokens at the END of row b; `prompt` holds them from column 0 on
        n_real = None          # int64 [B], None when every row is P long
        prompt = input_ids
        rows = None
        if attention_mask is not None:
            if tuple(attention_mask.shape) != (B, P):
                raise ValueError("attention_mask must have the shape of input_ids")
            am = attention_mask.to(device=dev, dtype=torch.int64)
            n = am.sum(1)
            zeros_then_ones = (am[:, 1:] >= am[:, :-1]).all() if P > 1 else torch.ones((), dtype=torch.bool, device=dev)
            n_min, n_max, ok = torch.stack([n.min(), n.max(), zeros_then_ones.long()]).tolist()
            if not (ok and n_min >= 1 and int(am.max()) <= 1 and int(am.min()) >= 0):
                raise ValueError("generate() takes prompts padded on the LEFT: every attention_mask row must be zeros "
                                 "followed by at least one 1")
            if n_min < P:
                n_real = n
                rows = torch.arange(B, device=dev)
                shift = (torch.arange(P, device=dev)[None, :] + (P - n)[:, None]).clamp_(max=P - 1)
                prompt = input_ids.gather(1, shift)   # past its real tokens a row repeats its last token (never seen)
        if not use_cache or getattr(self, "act_dtype", torch.bfloat16) == torch.float32:
            # no cache (and the fp32 parity mode, whose cache-less arithmetic is the reference's own): every step
            # runs the whole prefix through the training forward and keeps the last row of logits
            nxt = torch.empty(B, dtype=torch.int64, device=dev)
            unfinished = torch.ones(B, dtype=torch.int64, device=dev) if eos_token_id is not None else None
            step_logits = [] if return_step_logits else None
            n_done = 0
            seq = out
            if n_real is not None:     # the shifted rows grow at their own ends; columns past an end hold token 0
                seq = torch.zeros(B, total, dtype=torch.int64, device=dev)
                seq[:, :P] = prompt
            for step in range(max_new_tokens):
                xN, _ = self._run_layers(seq[:, :P + step].contiguous(), save=None)
                if n_real is None:
                    last = xN.view(B, P + step, -1)[:, -1, :].contiguous()
                else:
                    last = xN.view(B, P + step, -1)[rows, n_real + step - 1].contiguous()
                logits = self._logits(last)
                ld = logits.stride(0)
                full = logits.as_strided((B, ld), (ld, 1))
                nz = noise[step] if (noise is not None and do_sample) else None
                ops.sample(full, V, nxt, do_sample, temperature, top_k or 0, top_p, nz, seed, step)
                if return_step_logits:
                    step_logits.append(logits.float().clone())
                tok = nxt
                if unfinished is not None:
                    tok = nxt * unfinished + pad_token_id * (1 - unfinished)
                    unfinished.mul_((tok != eos_token_id).long())
                out[:, P + step] = tok
                if n_real is not None:
                    seq[rows, n_real + step] = tok
                n_done = step + 1
                if unfinished is not None and int(unfinished.max()) == 0:
                    break
            out = out[:, :P + n_done]
            self._check_device_flags()
            return (out, torch.stack(step_logits, 0)) if return_step_logits else out
        held = getattr(self, "_kv_reserved", None)
        if held is not None and B <= held.k[0].shape[0] and total <= held.max_len and held.k[0].device == dev:
            cache = held.first_streams(B)
        else:
            cache = KVCache(cfg, B, total, dev)
        # lens[t] = int32 [B] filled with t: per-stream positions / key counts without per-step host->device traffic
        lens = torch.arange(total + 1, dtype=torch.int32, device=dev)[:, None].expand(total + 1, B).contiguous()
        if n_real is not None:   # stream b is (P - n_real[b]) positions behind the column count
            lens = (lens - (P - n_real).to(torch.int32)[None, :]).clamp_(min=0)
        self._rope_table(total)          # sized once, before any side stream reads it
        # (MGEN_DECODE_PARTS=n: n parts of at least 32 streams, for measurements; 3 and 4 parts measured no better than 2)
        nparts = max(1, min(int(os.environ.get("MGEN_DECODE_PARTS", "2")), B // 32)) if B >= _TWO_PART_MIN_STREAMS else 1
        if type(self)._decode_step is not GenerationMixin._decode_step:
            nparts = 1           # a model with its own one-token step (GPT-2) decodes the batch whole
        main = torch.cuda.current_stream()
        nxt_all = torch.empty(B, dtype=torch.int64, device=dev)
        parts = []
        for i in range(nparts):
            b0, b1 = B * i // nparts, B * (i + 1) // nparts
            ws_n = ops.attn_decode_workspace(b1 - b0, cfg.num_attention_heads, cfg.head_dim,
                                             self._decode_split(b1 - b0, total))
            parts.append(_Part(cfg, cache, b0, b1, nxt_all, ws_n, dev,
                               main if nparts == 1 else torch.cuda.Stream(device=dev)))
            parts[-1].lens = lens[:, b0:b1]      # row t of the view: positions of the part's streams, contiguous
        unfinished = torch.ones(B, dtype=torch.int64, device=dev) if eos_token_id is not None else None
        step_logits = [[] for _ in parts] if return_step_logits else None

        def pick(logits, step, b0, b1, nxt, log=None, step_dev=None):
            """Rows [b0, b1) of the batch: choose the token of generation step `step` from their logits (step_dev: the
            step is read from this device counter when the launch runs, `step` is then ignored)."""
            nz = noise[step][b0:b1] if (noise is not None and do_sample) else None
            first_col = P if step_dev is not None else P + step
            col = out.data_ptr() + 8 * (b0 * total + first_col)
            ops.sample(logits, V, nxt, do_sample, temperature, top_k or 0, top_p, nz, seed,
                       0 if step_dev is not None else step, out2=col, out2_stride=total, row0=b0, step=step_dev,
                       unfinished=unfinished[b0:b1] if unfinished is not None else None,
                       eos_id=eos_token_id if unfinished is not None else 0,
                       pad_id=pad_token_id if unfinished is not None else 0)
            if log is not None:
                log.append(logits[:, :V].float().clone())

        # ---- prefill: the training forward on the prompt, K/V written into the cache
        xN, _ = self._run_layers(prompt.contiguous(), save=None, cache=cache)
        if n_real is None:
            last = xN.view(B, P, -1)[:, -1, :].contiguous()
        else:
            last = xN.view(B, P, -1)[rows, n_real - 1].contiguous()
        W = self._view(self.flat_shadow, self._lm_head_name())
        ldv = parts[0].buf.ldv
        logits0 = parts[0].buf.logits if nparts == 1 else torch.empty(B, ldv, dtype=torch.bfloat16, device=dev)
        ops.gemm(last, W, logits0, B, V, cfg.hidden_size, last.stride(0), W.stride(0), ldv)
        for i, pt in enumerate(parts):
            pick(logits0[pt.b0:pt.b1], 0, pt.b0, pt.b1, pt.buf.next, step_logits[i] if return_step_logits else None)
        if nparts > 1:
            for pt in parts:
                pt.stream.wait_stream(main)
        n_done = 1
        L = cfg.num_hidden_layers
        t_issue = time.perf_counter()

        def issue_step(step, t, live):
            """The launches of one decode step: every part's layers, lm_head and token choice. live: positions and
            step index come from device buffers that the last launch of the step moves on, so the same launches serve
            every step (what a captured graph replays); otherwise they are arguments of this step's launches."""
            if nparts == 1:
                pt = parts[0]
                logits = self._decode_step(pt.buf.next, pt.cache, t, pt.buf, pt.live if live else pt.lens, pt.ws)
                pick(logits, step, 0, B, pt.buf.next, step_logits[0] if return_step_logits else None,
                     pt.step_dev if live else None)
                if live:
                    ops.decode_advance(pt.live[0], pt.live[1], pt.step_dev)
                return
            # The launches of the two halves are issued layer by layer in turn, each half on its own stream.
            # (Chaining the attention launches of the two halves through events, so that they never run at the
            # same time, measured slower than leaving the two streams alone: 5.68 against 5.45 ms per step.)
            try:
                for pt in parts:
                    ops.use_stream(pt.stream.cuda_stream)
                    self._decode_begin(pt.buf.next, pt.buf)
                for l in range(L):
                    for pt in parts:
                        ops.use_stream(pt.stream.cuda_stream)
                        self._decode_layer(l, pt.cache, t, pt.buf, pt.live if live else pt.lens, pt.ws)
                for pt in parts:
                    ops.use_stream(pt.stream.cuda_stream)
                    self._decode_end(pt.buf)
            finally:
                ops.use_stream(None)
            for i, pt in enumerate(parts):
                with torch.cuda.stream(pt.stream):
                    pick(pt.buf.logits, step, pt.b0, pt.b1, pt.buf.next,
                         step_logits[i] if return_step_logits else None, pt.step_dev if live else None)
                    if live:
                        ops.decode_advance(pt.live[0], pt.live[1], pt.step_dev)

        # A decode step as a HIP graph (opt-in, see _DECODE_GRAPH): the step's launches are captured once (per
        # attention split count) with positions and step index in device buffers, and replayed.
        graphs = None
        if (_DECODE_GRAPH and max_new_tokens > 3 and noise is None and not return_step_logits
                and type(self)._decode_step is GenerationMixin._decode_step and not self._ln
                and all(pt.buf.sk_qkv and pt.buf.sk_gu and pt.buf.sk_o and pt.buf.sk_down for pt in parts)):
            graphs = {}
            for pt in parts:
                pt.live = (pt.lens[P].clone(), pt.lens[P + 1].clone())      # positions / key counts of step 1
                pt.step_dev = torch.ones(1, dtype=torch.int32, device=dev)  # generation step of the next launch
            cap_stream = torch.cuda.Stream(device=dev)
        for step in range(1, max_new_tokens):
            if unfinished is not None and step % eos_check_every == 0:
                if nparts > 1:
                    for pt in parts:
                        main.wait_stream(pt.stream)
                if int(unfinished.max()) == 0:
                    break
            t = P + step - 1
            if graphs is None:
                issue_step(step, t, False)
            elif step == 1:
                issue_step(step, t, True)    # (not captured: the first launches of a process set kernel attributes)
                if nparts > 1:
                    for pt in parts:
                        main.wait_stream(pt.stream)
            else:
                key = tuple(self._decode_split(pt.b1 - pt.b0, t + 1) for pt in parts)
                g = graphs.get(key)
                if g is None:
                    g = graphs[key] = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=cap_stream):
                        if nparts > 1:
                            for pt in parts:
                                pt.stream.wait_stream(cap_stream)
                        issue_step(step, t, True)
                        if nparts > 1:
                            for pt in parts:
                                cap_stream.wait_stream(pt.stream)
                g.replay()
            n_done = step + 1
        # host time spent issuing the decode steps (the GPU runs behind): for measurements
        self.last_decode_issue_s = time.perf_counter() - t_issue
        if nparts > 1:
            for pt in parts:
                main.wait_stream(pt.stream)
        if unfinished is not None and n_done < max_new_tokens:
            out = out[:, :P + n_done]
        if unfinished is not None:
            # trim trailing columns produced after every row had finished (the reference stops at that point)
            fin = (out[:, P:] == eos_token_id).long().cumsum(1).clamp(max=1)
            all_done = fin.min(0).values
            if bool(all_done.any()):
                first = int(torch.nonzero(all_done)[0])
                out = out[:, :P + first + 1]
        self._check_device_flags()
        if return_step_logits:
            n = min(len(x) for x in step_logits)
            return out, torch.cat([torch.stack(x[:n], 0) for x in step_logits], 1)
        return out

    # ------------------------------------------------------------------------------------------------
    # 4-D position model: note-by-note generation
    # ------------------------------------------------------------------------------------------------
    def _decode_step4d(self, rows, cache: KVCache, t: int, buf: _DecodeBuffers, lens, ws):
        """One note row per stream ([B,4] fp32, contiguous) at sequence index t; leaves the unit pairs in buf.pred."""
        cfg = self.cfg
        B = rows.shape[0]
        d, H, KVH, Dh, I = (cfg.hidden_size, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim,
                            cfg.intermediate_size)
        NQ, QK = cfg.qkv_width, (H + KVH) * Dh
        cs = self._rope4d_table(rows)
        scale = Dh ** -0.5
        pos, kv_len = lens[t], lens[t + 1]
        nsplit = self._decode_split(B, t + 1)
        h = buf.h0
        ops.fill_rows(self.model.embed_vector, h)
        qp = buf.rot.data_ptr()
        kp, vp = qp + 2 * H * Dh, qp + 2 * QK
        eps = cfg.rms_norm_eps
        L = cfg.num_hidden_layers
        ops.rmsnorm_fwd(h, self._layers[0].w_ln1, buf.x, None, eps)
        for l, v in enumerate(self._layers):
            ops.linear_fwd(buf.x, v.s_qkv, buf.qkv)
            ops.qkvnorm_rope4d_fwd(buf.qkv.data_ptr(), NQ, qp, kp, vp, NQ, NQ, NQ, v.w_qn, v.w_kn, cs, B, H, KVH, KVH,
                                   Dh, eps)
            ops.kv_append(kp, vp, NQ, NQ, cache.k[l], cache.v[l], pos, B, 1, KVH, Dh, cache.max_len)
            ops.attn_decode(qp, NQ, cache.k[l], cache.v[l], kv_len, buf.att, H * Dh, ws, B, H, KVH, Dh, cache.max_len,
                            nsplit, scale)
            ops.rotate4d(buf.att, buf.att, cs, B, H, Dh, inverse=True)
            ops.linear_slabs(buf.att, v.s_o, buf.slabs, buf.ns_o)
            ops.slab_resid_rmsnorm(buf.slabs, buf.ns_o, h, buf.h1, v.w_ln2, buf.x, eps)
            ops.linear_fwd(buf.x, v.s_gu, buf.gu)
            ops.swiglu_fwd(buf.gu, buf.act)
            ops.linear_slabs(buf.act, v.s_down, buf.slabs, buf.ns_d)
            nxt = buf.h2 if h is buf.h0 else buf.h0
            w_next = self._layers[l + 1].w_ln1 if l + 1 < L else self.model.norm.weight
            ops.slab_resid_rmsnorm(buf.slabs, buf.ns_d, buf.h1, nxt, w_next, buf.x, eps)
            h = nxt
        W = self._view(self.flat_shadow, "lm_head.weight")
        ops.gemm(buf.x, W, buf.pairs, B, 16, d, d, W.stride(0), 16)
        ops.cos4d_fwd_bwd(buf.pairs, None, None, 1.0, buf.pred, None, None)

    @torch.no_grad()
    def generate_music(self, position_tensors, total_length: int = 200, use_cache: bool = True,
                       return_preds: bool = False):
        """Note-by-note generation of the 4-D position model (/root/reference/generate_music.py:146-179): every
        step decodes the unit pairs predicted at the last row into [start delta, duration, pitch, velocity]
        (process_last_logits, :116-142), adds the previous start time and appends the row.

        position_tensors: fp32 [B, P, 4] starting rows (the reference starts from the single row
        batch['position_tensors'][0:1, 0:1]). Returns fp32 [B, P + total_length, 4].
        The reference re-runs the whole prefix for every new note; with use_cache (default) the rotated keys and
        values stay in a pre-allocated cache and each step costs one row. use_cache=False keeps the prefix re-run."""
        cfg = self.cfg
        if not cfg.pos4d:
            raise RuntimeError("generate_music() belongs to the 4-D position model (config.pos4d)")
        if position_tensors.dim() != 3 or position_tensors.shape[-1] != 4 or position_tensors.shape[1] < 1:
            raise ValueError("position_tensors must be [batch, prompt_len >= 1, 4]")
        if not position_tensors.is_cuda:
            raise RuntimeError("MidiLM runs on the GPU only: position_tensors must be a CUDA/HIP tensor")
        dev = position_tensors.device
        B, P = position_tensors.shape[:2]
        total = P + total_length
        self.sync_shadow()
        out = torch.zeros(B, total, 4, dtype=torch.float32, device=dev)
        out[:, :P] = position_tensors
        preds = [] if return_preds else None
        row_bytes, ld = 16, total * 4

        def decode_into(pred, ld_pred, t):
            """row t+1 <- note decoded from pred, start accumulated on row t"""
            ops.pos4d_next(pred, ld_pred, out.data_ptr() + row_bytes * t, ld, out.data_ptr() + row_bytes * (t + 1),
                           ld, B)

        if not use_cache or getattr(self, "act_dtype", torch.bfloat16) == torch.float32:
            for i in range(total_length):
                t = P + i - 1
                xN, _ = self._run_layers(out[:, :t + 1].contiguous(), save=None)
                last = xN.view(B, t + 1, -1)[:, -1, :].contiguous()
                self._cos_head_loss(last, None, with_grad=False)
                if return_preds:
                    preds.append(self._last_pred.clone())
                decode_into(self._last_pred.data_ptr(), 16, t)
            return (out, torch.stack(preds, 1).view(B, -1, 8, 2)) if return_preds else out

        cache = KVCache(cfg, B, total, dev)
        buf = _DecodeBuffers(cfg, B, dev)
        lens = torch.arange(total + 1, dtype=torch.int32, device=dev)[:, None].expand(total + 1, B).contiguous()
        ws_n = ops.attn_decode_workspace(B, cfg.num_attention_heads, cfg.head_dim, self._decode_split(B, total))
        ws = torch.empty(max(ws_n, 1), dtype=torch.float32, device=dev)
        # prefill on the starting rows
        xN, _ = self._run_layers(out[:, :P].contiguous(), save=None, cache=cache)
        last = xN.view(B, P, -1)[:, -1, :].contiguous()
        self._cos_head_loss(last, None, with_grad=False)
        if return_preds:
            preds.append(self._last_pred.clone())
        decode_into(self._last_pred.data_ptr(), 16, P - 1)
        for i in range(1, total_length):
            t = P + i - 1
            buf.cur.copy_(out[:, t])
            self._decode_step4d(buf.cur, cache, t, buf, lens, ws)
            if return_preds:
                preds.append(buf.pred.clone())
            decode_into(buf.pred.data_ptr(), 16, t)
        return (out, torch.stack(preds, 1).view(B, -1, 8, 2)) if return_preds else out
