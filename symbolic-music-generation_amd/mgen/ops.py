"""Tensor-level wrappers over the C-ABI kernels: take torch CUDA(HIP) tensors, pass raw device pointers, sizes,
strides and the current HIP stream. torch is used for device memory and streams only; every computation below runs
in libmgen_hip.so."""
import os

import torch

from ._lib import call

# GEMM epilogues (must match csrc/gemm.hip)
OUT_BF16, OUT_F32, OUT_F32_ACC, OUT_F32_RESID_BF16ROUND, OUT_F32_RESID = 0, 1, 2, 3, 4
OUT_BF16_ACC = 6   # bf16 C = bf16(C + product)


_stream_override = None


def use_stream(handle):
    """Launch on this raw HIP stream handle instead of torch's current stream (None: back to torch's). For code that
    feeds several streams in turn without switching torch's stream context for every launch."""
    global _stream_override
    _stream_override = handle


def _stream():
    return _stream_override if _stream_override is not None else torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("mgen ops need tensors on the GPU (no CPU fallback exists)")


def gemm(a, b, c, M, N, K, lda, ldb, ldc, trans_a=False, trans_b=False, out_mode=OUT_BF16, resid=None, ldr=0,
         alpha=1.0):
    """c[M,N] (+)= op(a)[M,K] @ op(b)[K,N]; a is [M,K] (or [K,M] if trans_a), b is [N,K] (or [K,N] if trans_b)."""
    _need_gpu(a, b, c)
    call("mgen_gemm_bf16", _p(a), _p(b), _p(c), _p(resid), M, N, K, lda, ldb, ldc, ldr, int(trans_a), int(trans_b),
         out_mode, alpha, _stream())


class W2:
    """A weight of the fp32 parity mode: hi = bf16(w), lo = bf16(w - hi), two bf16 tensors of w's shape."""

    def __init__(self, hi, lo):
        self.hi, self.lo = hi, lo
        self.shape = hi.shape

    def stride(self, i):
        return self.hi.stride(i)


def split_bf16(x):
    """fp32 tensor (contiguous memory) -> (hi, lo) bf16 tensors of the same shape and strides."""
    _need_gpu(x)
    hi = torch.empty_like(x, dtype=torch.bfloat16)
    lo = torch.empty_like(x, dtype=torch.bfloat16)
    call("mgen_split_bf16", _p(x), _p(hi), _p(lo), x.numel(), _stream())
    return hi, lo


def split_bf16_into(x, hi, lo):
    _need_gpu(x, hi, lo)
    call("mgen_split_bf16", _p(x), _p(hi), _p(lo), x.numel(), _stream())


def presplit(t):
    """An fp32 activation that feeds several products: split once. bf16 tensors pass through."""
    return W2(*split_bf16(t)) if (isinstance(t, torch.Tensor) and t.dtype == torch.float32) else t


def _parts(t):
    if isinstance(t, W2):
        return t.hi, t.lo
    if t.dtype == torch.float32:
        return split_bf16(t)
    return t, None


def gemm_any(a, b, c, M, N, K, lda, ldb, ldc, trans_a=False, trans_b=False, out_mode=OUT_BF16, resid=None, ldr=0):
    """gemm() whose operands may be fp32 tensors or W2 pairs: then the product is hi*hi + hi*lo + lo*hi accumulated
    in fp32 into an fp32 `c` (bf16-store and bf16-round epilogues become their fp32 counterparts)."""
    ah, al = _parts(a)
    bh, bl = _parts(b)
    if al is None and bl is None:
        return gemm(a, b, c, M, N, K, lda, ldb, ldc, trans_a, trans_b, out_mode, resid, ldr)
    if al is None or bl is None:
        raise RuntimeError("gemm_any: both operands must be bf16 or both fp32 / split")
    mode = {OUT_BF16: OUT_F32, OUT_F32_RESID_BF16ROUND: OUT_F32_RESID}.get(out_mode, out_mode)
    gemm(ah, bh, c, M, N, K, lda, ldb, ldc, trans_a, trans_b, mode, resid, ldr)
    gemm(ah, bl, c, M, N, K, lda, ldb, ldc, trans_a, trans_b, OUT_F32_ACC)
    gemm(al, bh, c, M, N, K, lda, ldb, ldc, trans_a, trans_b, OUT_F32_ACC)


def gemm_seg(a, b, a2, b2, c, M, N, K, K2, lda, ldb, lda2, ldb2, ldc, trans_b=False, out_mode=OUT_BF16, resid=None,
             ldr=0):
    """c = [a | a2] [b | b2]^T: one launch for a projection plus its low-rank adapter product (K, K2 multiples of
    64). trans_b: b stored [K, N] and b2 stored [K2, N]."""
    _need_gpu(a, b, a2, b2, c)
    call("mgen_gemm_bf16_seg", _p(a), _p(b), _p(a2), _p(b2), _p(c), _p(resid), M, N, K, K2, lda, ldb, lda2, ldb2, ldc,
         ldr, int(trans_b), out_mode, _stream())


def gemm_slab_count(K, max_slabs=16):
    from ._lib import lib
    return lib().mgen_gemm_slab_count(K, max_slabs)


def linear_slabs(x, w, slabs, nslabs):
    """slabs[z, T, N] = partial products of x[T,K] @ w[N,K]^T over the z-th K slice (fp32, no atomics)."""
    _need_gpu(x, w, slabs)
    T, K = x.shape
    N = w.shape[0]
    call("mgen_gemm_bf16_slabs", _p(x), _p(w), _p(slabs), T, N, K, x.stride(0), w.stride(0), 0, 0, nslabs, _stream())


def skinny_fits(M, N, K, lda, ldw, ldc, slabs):
    from ._lib import lib
    return os.environ.get("MGEN_GEMM_SKINNY", "1") != "0" and bool(lib().mgen_gemm_skinny_fits(M, N, K, lda, ldw, ldc,
                                                                                                int(slabs)))


def skinny_slabs(M, N, K):
    """Number of fp32 slabs the library cuts a K-long reduction into for mgen_slab_resid_rmsnorm to sum."""
    from ._lib import lib
    return int(lib().mgen_gemm_skinny_slabs(M, N, K))


def linear_skinny(x, w, out, slabs=0, a_swiglu=False):
    """Decode-step projection out = x @ w^T (x [M, K] bf16, or gu [M, 2K] with a_swiglu: the operand is
    silu(gate) * up); out bf16 [M, N], or with slabs = s fp32 [s, M, N] partial products over K / s reduction indices
    each (s divides K / 512; True = one per 512 indices)."""
    _need_gpu(x, w, out)
    M = x.shape[0]
    N, K = w.shape
    if slabs is True:
        slabs = K // 512
    call("mgen_gemm_bf16_skinny", _p(x), _p(w), _p(out), M, N, K, x.stride(0), w.stride(0), N, int(slabs),
         M * N, int(a_swiglu), _stream())


def slab_resid_rmsnorm(slabs, nslabs, resid, h_out, w, y, eps, round_bf16=True):
    T, d = resid.shape
    call("mgen_slab_resid_rmsnorm", _p(slabs), nslabs, T * d, _p(resid), _p(h_out), _p(w), _p(y), T, d, eps,
         int(round_bf16), _stream())


def linear_fwd(x, w, out, out_mode=OUT_BF16, resid=None):
    """out[T,N] = x[T,K] @ w[N,K]^T for contiguous 2-D tensors."""
    T, K = x.shape
    N = w.shape[0]
    gemm_any(x, w, out, T, N, K, x.stride(0), w.stride(0), out.stride(0), False, False, out_mode, resid,
             resid.stride(0) if resid is not None else 0)


def linear_dgrad(dy, w, dx):
    """dx[T,K] = dy[T,N] @ w[N,K]."""
    T, N = dy.shape
    K = w.shape[1]
    gemm_any(dy, w, dx, T, K, N, dy.stride(0), w.stride(0), dx.stride(0), False, True, OUT_BF16)


def linear_wgrad(dy, x, dw, accumulate=True, n_rows=None, n_cols=None):
    """dw[N,K] (+)= dy[T,N]^T @ x[T,K] (fp32 output)."""
    T = dy.shape[0]
    N = n_rows if n_rows is not None else dy.shape[1]
    K = n_cols if n_cols is not None else x.shape[1]
    gemm_any(dy, x, dw, N, K, T, dy.stride(0), x.stride(0), dw.stride(0), True, True,
             OUT_F32_ACC if accumulate else OUT_F32)


def linear_wgrad_group(items):
    """items = [(dy [T, N_i], x [T, K_i], dw [N_i, K_i] fp32), ...] (at most 4, same T): dw_i += dy_i^T @ x_i in one
    launch when every product fits the output-stationary kernel, else one by one."""
    import ctypes
    n = len(items)
    T = items[0][0].shape[0]
    ok = 1 <= n <= 4 and os.environ.get("MGEN_WGRAD_GROUP", "1") != "0" and \
        os.environ.get("MGEN_GEMM_OS", "1") != "0" and T % 32 == 0 and T >= 2048
    for dy, x, dw in items:
        ok = ok and dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dw.dtype == torch.float32 and \
            dy.shape[0] == T and x.shape[0] == T and dy.shape[1] >= 256 and x.shape[1] >= 256 and \
            dy.shape[1] % 8 == 0 and x.shape[1] % 8 == 0 and dy.stride(0) % 8 == 0 and x.stride(0) % 8 == 0 and \
            T * dy.stride(0) * 2 < (1 << 32) and T * x.stride(0) * 2 < (1 << 32)
    if not ok:
        for dy, x, dw in items:
            linear_wgrad(dy, x, dw)
        return
    _need_gpu(*[t for it in items for t in it])
    vp = ctypes.c_void_p * n
    ci = ctypes.c_int * n
    cl = ctypes.c_long * n
    call("mgen_gemm_bf16_wgrad_group", n, vp(*[_p(i[0]) for i in items]), vp(*[_p(i[1]) for i in items]),
         vp(*[_p(i[2]) for i in items]), ci(*[i[0].shape[1] for i in items]), ci(*[i[1].shape[1] for i in items]), T,
         cl(*[i[0].stride(0) for i in items]), cl(*[i[1].stride(0) for i in items]),
         cl(*[i[2].stride(0) for i in items]), _stream())


def gemm_bias(a, b, c, bias, M, N, K, lda, ldb, ldc, trans_a=False, trans_b=False, out_mode=OUT_BF16, resid=None,
              ldr=0):
    """gemm() with a bf16 bias row [N] added to the accumulators before the output rounding."""
    _need_gpu(a, b, c, bias)
    call("mgen_gemm_bf16_bias", _p(a), _p(b), _p(c), _p(resid), _p(bias), M, N, K, lda, ldb, ldc, ldr, int(trans_a),
         int(trans_b), out_mode, _stream())


def conv1d_fwd(x, w, bias, out, out_mode=OUT_BF16, resid=None):
    """out[T,N] = x[T,K] @ w[K,N] + bias[N]  (GPT-2 Conv1D: weight stored [in, out])."""
    T, K = x.shape
    N = w.shape[1]
    if isinstance(w, W2):   # fp32 parity mode: three bf16 products into an fp32 output, then the fp32 bias
        gemm_any(x, w, out, T, N, K, x.stride(0), w.stride(0), out.stride(0), False, True, out_mode, resid,
                 resid.stride(0) if resid is not None else 0)
        call("mgen_bias_add_f32", _p(out), out.stride(0), _p(bias), T, N, _stream())
        return
    gemm_bias(x, w, out, bias, T, N, K, x.stride(0), w.stride(0), out.stride(0), False, True, out_mode, resid,
              resid.stride(0) if resid is not None else 0)


def conv1d_dgrad(dy, w, dx):
    """dx[T,K] = dy[T,N] @ w[K,N]^T."""
    T, N = dy.shape
    K = w.shape[0]
    gemm_any(dy, w, dx, T, K, N, dy.stride(0), w.stride(0), dx.stride(0), False, False, OUT_BF16)


def conv1d_wgrad(dy, x, dw, db):
    """dw[K,N] += x[T,K]^T @ dy[T,N];  db[N] += column sums of dy."""
    T, N = dy.shape
    K = x.shape[1]
    gemm_any(x, dy, dw, K, N, T, x.stride(0), dy.stride(0), dw.stride(0), True, True, OUT_F32_ACC)
    call("mgen_colsum_f32" if dy.dtype == torch.float32 else "mgen_colsum_bf16", _p(dy), dy.stride(0), _p(db), T, N,
         _stream())


def conv1d_bias_grad(dy, x, dw, db):
    """db[N] += column sums of dy (the signature of conv1d_wgrad: its weight gradient is taken elsewhere)."""
    T, N = dy.shape
    call("mgen_colsum_f32" if dy.dtype == torch.float32 else "mgen_colsum_bf16", _p(dy), dy.stride(0), _p(db), T, N,
         _stream())


def layernorm_fwd(x, w, b, y, mean, rstd, eps):
    T, d = x.shape
    call("mgen_layernorm_fwd", _p(x), _p(w), _p(b), _p(y), _p(mean), _p(rstd), T, d, eps,
         int(y.dtype == torch.bfloat16), _stream())


def layernorm_bwd(dy, x, w, mean, rstd, dres, dres_lp, dw, db, accumulate):
    T, d = x.shape
    call("mgen_layernorm_bwd", _p(dy), _p(x), _p(w), _p(mean), _p(rstd), _p(dres), _p(dres_lp), _p(dw), _p(db), T, d,
         int(accumulate), int(dy.dtype == torch.bfloat16), _stream())


def gelu_new_fwd(h, act):
    _need_gpu(h, act)
    call("mgen_gelu_new_f32_fwd" if h.dtype == torch.float32 else "mgen_gelu_new_fwd", _p(h), _p(act), h.numel(),
         _stream())


def gelu_new_bwd(h, dact, dh):
    _need_gpu(h, dact, dh)
    call("mgen_gelu_new_f32_bwd" if h.dtype == torch.float32 else "mgen_gelu_new_bwd", _p(h), _p(dact), _p(dh),
         h.numel(), _stream())


def embed_pos_fwd(ids, wte, wpe, pos, out, S, err_flag=None):
    _need_gpu(ids, wte, wpe, out)
    T, d = out.shape
    call("mgen_embed_pos_fwd", _p(ids), _p(wte), _p(wpe), _p(pos), _p(out), T, S, d, wte.shape[0], wpe.shape[0],
         _p(err_flag), _stream())


def rmsnorm_fwd(x, w, y, rstd, eps):
    T, d = x.shape
    call("mgen_rmsnorm_fwd", _p(x), _p(w), _p(y), _p(rstd), T, d, eps, int(y.dtype == torch.bfloat16), _stream())


def resid_rmsnorm_fwd(resid, y, w, h_out, x_out, rstd, eps):
    """h_out = resid + float(y) (y bf16 [T, d], any row pitch); x_out = RMSNorm(h_out) in bf16; rstd [T] or None."""
    T, d = resid.shape
    call("mgen_resid_rmsnorm_fwd", _p(resid), _p(y), y.stride(0), _p(w), _p(h_out), _p(x_out), _p(rstd), T, d, eps,
         _stream())


def rmsnorm_bwd(dy, x, w, rstd, dres, dres_lp, dw, accumulate):
    T, d = x.shape
    call("mgen_rmsnorm_bwd", _p(dy), _p(x), _p(w), _p(rstd), _p(dres), _p(dres_lp), _p(dw), T, d, int(accumulate),
         int(dy.dtype == torch.bfloat16), _stream())


def rope_table(inv_freq, npos):
    half = inv_freq.numel()
    cs = torch.empty(npos, half, 2, dtype=torch.float32, device=inv_freq.device)
    call("mgen_rope_table", _p(inv_freq), _p(cs), npos, half, _stream())
    return cs


def qknorm_rope_fwd(qkv, ld_in, qout, kout, ld_q, ld_k, wq, wk, cs, pos, S, T, HQ, HK, D, eps, interleaved=False):
    """interleaved: adjacent rotation pairs (2i, 2i+1) instead of the split-half pairs (i, i + D/2)."""
    call("mgen_qknorm_rope_il_fwd" if interleaved else "mgen_qknorm_rope_fwd", _p(qkv), ld_in, qout, kout, ld_q, ld_k, _p(wq), _p(wk), _p(cs), _p(pos), S, T, HQ,
         HK, D, eps, int(qkv.dtype == torch.bfloat16), _stream())


def qknorm_rope_bwd(dqk, ld_in, dq_raw, dk_raw, ld_q, ld_k, raw, ld_raw, wq, wk, dwq, dwk, cs, pos, S, T, HQ, HK, D,
                    eps, is_bf16=True, interleaved=False):
    call("mgen_qknorm_rope_il_bwd" if interleaved else "mgen_qknorm_rope_bwd", dqk, ld_in, dq_raw, dk_raw, ld_q, ld_k, _p(raw), ld_raw, _p(wq), _p(wk), _p(dwq),
         _p(dwk), _p(cs), _p(pos), S, T, HQ, HK, D, eps, int(is_bf16), _stream())


def rope4d_table(pos4, freqs, D):
    """pos4 [T,4] fp32 -> cos/sin table [T, D/2, 2] fp32."""
    _need_gpu(pos4, freqs)
    T = pos4.shape[0]
    cs = torch.empty(T, D // 2, 2, dtype=torch.float32, device=pos4.device)
    call("mgen_rope4d_table", _p(pos4), _p(freqs), _p(cs), T, D, _stream())
    return cs


def qkvnorm_rope4d_fwd(qkv, ld_in, qout, kout, vout, ld_q, ld_k, ld_v, wq, wk, cs, T, HQ, HK, HV, D, eps,
                       is_bf16=True):
    """qkv/qout/kout/vout are raw device pointers (ints)."""
    call("mgen_qkvnorm_rope4d_fwd", qkv, ld_in, qout, kout, vout, ld_q, ld_k, ld_v, _p(wq), _p(wk), _p(cs), T, HQ, HK,
         HV, D, eps, int(is_bf16), _stream())


def qkvnorm_rope4d_bwd(dqkv, ld_in, dq, dk, dv, ld_q, ld_k, ld_v, raw, ld_raw, wq, wk, dwq, dwk, cs, T, HQ, HK, HV, D,
                       eps, is_bf16=True):
    call("mgen_qkvnorm_rope4d_bwd", dqkv, ld_in, dq, dk, dv, ld_q, ld_k, ld_v, _p(raw), ld_raw, _p(wq), _p(wk),
         _p(dwq), _p(dwk), _p(cs), T, HQ, HK, HV, D, eps, int(is_bf16), _stream())


def rotate4d(x, y, cs, T, H, D, inverse):
    """y[T, H*D] = x rotated pairwise by +angle (inverse=False) or -angle (inverse=True); may be in place."""
    _need_gpu(x, y, cs)
    ld_x, ld_y = x.stride(0), y.stride(0)
    if inverse:
        call("mgen_qkvnorm_rope4d_bwd", x.data_ptr(), ld_x, 0, 0, y.data_ptr(), 0, 0, ld_y, 0, 0, 0, 0, 0, 0, _p(cs),
             T, 0, 0, H, D, 0.0, int(x.dtype == torch.bfloat16), _stream())
    else:
        call("mgen_qkvnorm_rope4d_fwd", x.data_ptr(), ld_x, 0, 0, y.data_ptr(), 0, 0, ld_y, 0, 0, _p(cs), T, 0, 0, H, D,
             0.0, int(x.dtype == torch.bfloat16), _stream())


def fill_rows(vec, out):
    _need_gpu(vec, out)
    T, d = out.shape
    call("mgen_fill_rows", _p(vec), _p(out), T, d, _stream())


def cos4d_count(labels, inv_count, count=None):
    _need_gpu(labels, inv_count)
    call("mgen_cos4d_count", _p(labels), labels.numel() // 4, _p(inv_count), _p(count), _stream())


def cos4d_fwd_bwd(pairs, labels, inv_count, grad_scale, pred_out, row_loss, dpairs):
    _need_gpu(pairs)
    T = pairs.shape[0]
    call("mgen_cos4d_fwd_bwd", _p(pairs), _p(labels), _p(inv_count), grad_scale, _p(pred_out), _p(row_loss),
         _p(dpairs), T, int(pairs.dtype == torch.bfloat16), _stream())


def cos4d_components(pred, labels, out8):
    """out8[j] = mean over the counted rows of 1 - cos_sim of pair j (reference src/model/modeling.py:646-655)."""
    _need_gpu(pred, labels, out8)
    call("mgen_cos4d_components", _p(pred), _p(labels), labels.numel() // 4, _p(out8), _stream())


def pos4d_next(pred, ld_pred, prev, ld_prev, nxt, ld_next, B):
    """pred/prev/nxt are raw device pointers (ints) to fp32 rows."""
    call("mgen_pos4d_next", pred, ld_pred, prev, ld_prev, nxt, ld_next, B, _stream())


def collate_tokens(flat, offsets, prompt_len, ids, mask, labels, pad_id, mask_pads):
    """flat int64 [sum len], offsets int64 [B+1], prompt_len int32 [B] or None; ids/mask/labels int64 [B, S]."""
    _need_gpu(flat, offsets, prompt_len, ids, mask, labels)
    B, S = ids.shape
    call("mgen_collate_tokens", _p(flat), _p(offsets), _p(prompt_len), _p(ids), _p(mask), _p(labels), B, S,
         int(pad_id), int(bool(mask_pads)), _stream())


def collate_pos4d(rows, chunk_ids, pos, labels, mask, ids):
    """rows fp32 [total, 4], chunk_ids int64 [B]; pos/labels fp32 [B, S, 4]; mask/ids int64 [B, S] or None."""
    _need_gpu(rows, chunk_ids, pos, labels, mask, ids)
    B, S = pos.shape[0], pos.shape[1]
    call("mgen_collate_pos4d", _p(rows), rows.shape[0], _p(chunk_ids), _p(pos), _p(labels), _p(mask), _p(ids), B, S,
         _stream())


def notes_to_events(rows, ticks_per_second, ticks, msgs):
    """rows fp32 [n, 4]; ticks, msgs int32 [2n]."""
    _need_gpu(rows, ticks, msgs)
    call("mgen_notes_to_events", _p(rows), rows.shape[0], float(ticks_per_second), _p(ticks), _p(msgs), _stream())


def attn_fwd(q, k, v, out, lse, sq, sk, sv, so, B, S, H, KVH, D, scale, f32=False, drop_p=0.0, drop_key=0):
    """q/k/v/out are raw device pointers (ints) of token-major views (bf16, or fp32 with f32=True); strides in
    elements. drop_p > 0: dropout on the attention probabilities with the mask of `drop_key` (bf16 kernels only)."""
    if drop_p > 0.0:
        if f32:
            raise NotImplementedError("attention dropout exists for the bf16 kernels only")
        call("mgen_attn_fwd_dropout", q, k, v, out, _p(lse), sq, sk, sv, so, B, S, H, KVH, D, scale, float(drop_p),
             int(drop_key), _stream())
        return
    call("mgen_attn_f32_fwd" if f32 else "mgen_attn_fwd", q, k, v, out, _p(lse), sq, sk, sv, so, B, S, H, KVH, D,
         scale, _stream())


def attn_bwd(q, k, v, out, dout, lse, delta, dq, dk, dv, sq, sk, sv, so, sdo, sdq, sdk, sdv, B, S, H, KVH, D, scale,
             f32=False, drop_p=0.0, drop_key=0):
    if drop_p > 0.0:
        if f32:
            raise NotImplementedError("attention dropout exists for the bf16 kernels only")
        call("mgen_attn_bwd_dropout", q, k, v, out, dout, _p(lse), _p(delta), dq, dk, dv, sq, sk, sv, so, sdo, sdq,
             sdk, sdv, B, S, H, KVH, D, scale, float(drop_p), int(drop_key), _stream())
        return
    call("mgen_attn_f32_bwd" if f32 else "mgen_attn_bwd", q, k, v, out, dout, _p(lse), _p(delta), dq, dk, dv, sq, sk,
         sv, so, sdo, sdq, sdk, sdv, B, S, H, KVH, D, scale, _stream())


def dropout(x, y, p, key):
    """y = dropout(x) with the mask of `key` (x, y contiguous bf16 or fp32 of one type; y may be x). The gradient is
    the same call on the upstream gradient."""
    _need_gpu(x, y)
    assert x.dtype == y.dtype and x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel()
    call("mgen_dropout", _p(x), _p(y), x.numel(), int(x.dtype == torch.bfloat16), float(p), int(key), _stream())


def dropout_add(x, y, p, key):
    """y += dropout(x) with the mask of `key` (same type, contiguous, not aliased)."""
    _need_gpu(x, y)
    assert x.dtype == y.dtype and x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel()
    call("mgen_dropout_add", _p(x), _p(y), x.numel(), int(x.dtype == torch.bfloat16), float(p), int(key), _stream())


def dropout_resid(y, resid, out, p, key):
    """out (fp32) = resid (fp32) + bf16-rounded dropout(y) (y bf16)."""
    _need_gpu(y, resid, out)
    assert y.dtype == torch.bfloat16 and resid.dtype == torch.float32 and out.dtype == torch.float32
    assert y.is_contiguous() and resid.is_contiguous() and out.is_contiguous()
    call("mgen_dropout_resid", _p(y), _p(resid), _p(out), y.numel(), float(p), int(key), _stream())


_M64 = (1 << 64) - 1


def dropout_key(seed: int, counter: int) -> int:
    """One 64-bit mask key per (seed, dropout site): splitmix64 of seed + counter * golden ratio."""
    z = (int(seed) + (int(counter) + 1) * 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def swiglu_fwd(gu, act, round_silu=True):
    T, I = act.shape
    if gu.dtype == torch.float32:
        call("mgen_swiglu_f32_fwd", _p(gu), _p(act), T, I, _stream())
    else:
        call("mgen_swiglu_fwd", _p(gu), _p(act), T, I, int(round_silu), _stream())


def linear_swiglu_fwd(x, w_gu, gu, act):
    """gu[T, 2I] = x[T, K] @ w_gu[2I, K]^T and act[T, I] = silu(gate) * up in one launch (bf16; K % 64 == 0,
    I % 128 == 0). Returns False when the shapes do not fit: the caller then runs the two separate launches."""
    T, K = x.shape
    I = act.shape[1]
    if x.dtype != torch.bfloat16 or not isinstance(w_gu, torch.Tensor) or K % 64 or I % 128 or T < 2048:
        return False
    _need_gpu(x, w_gu, gu, act)
    call("mgen_gemm_bf16_swiglu", _p(x), _p(w_gu), _p(gu), _p(act), T, I, K, x.stride(0), w_gu.stride(0), gu.stride(0),
         act.stride(0), _stream())
    return True


def swiglu_bwd(gu, dact, dgu):
    T, I = dact.shape
    if gu.dtype == torch.float32:
        call("mgen_swiglu_f32_bwd", _p(gu), _p(dact), _p(dgu), T, I, _stream())
    elif gu.stride(0) != 2 * I or dact.stride(0) != I or dgu.stride(0) != 2 * I:
        call("mgen_swiglu_bwd_ld", _p(gu), gu.stride(0), _p(dact), dact.stride(0), _p(dgu), dgu.stride(0), T, I,
             _stream())
    else:
        call("mgen_swiglu_bwd", _p(gu), _p(dact), _p(dgu), T, I, _stream())


def embedding_fwd(ids, table, out, err_flag=None):
    T = ids.numel()
    V, d = table.shape
    call("mgen_embedding_fwd", _p(ids), _p(table), _p(out), T, d, V, _p(err_flag), _stream())


def embedding_bwd(ids, dout, dtable):
    T = ids.numel()
    V, d = dtable.shape
    call("mgen_embedding_bwd", _p(ids), _p(dout), _p(dtable), T, d, V, _stream())


def cast(src, dst):
    kind = 0 if src.dtype == torch.float32 else 1
    call("mgen_cast", _p(src), _p(dst), src.numel(), kind, _stream())


def transpose_bf16(src, dst, R, C, ldin, ldout, nbatch=1, bs_in=0, bs_out=0):
    """dst[b] [C][R] = src[b] [R][C]^T (bf16), matrices bs_in / bs_out elements apart."""
    call("mgen_transpose_bf16", _p(src), _p(dst), R, C, ldin, ldout, nbatch, bs_in, bs_out, _stream())


def linear_dgrad_t(dy, w_t, dx):
    """dx[T,K] = dy[T,N] @ w[N,K] given w_t = w^T stored [K][N] (the reduction index contiguous in both operands)."""
    T, N = dy.shape
    K = w_t.shape[0]
    gemm(dy, w_t, dx, T, K, N, dy.stride(0), w_t.stride(0), dx.stride(0), False, False, OUT_BF16)


def ce_count(labels, inv_count, count=None, ignore_index=-100, V=0, err_flag=None):
    """V > 0: labels outside [0, V) are not counted and set bit 2 of err_flag."""
    B, S = labels.shape
    call("mgen_ce_count", _p(labels), B, S, ignore_index, _p(inv_count), _p(count), V, _p(err_flag), _stream())


def check_right_padding(mask, err_flag):
    """Sets bit 1 of err_flag if a row of the int64 mask [B, S] is anything but ones followed by zeros."""
    B, S = mask.shape
    call("mgen_check_right_padding", _p(mask), B, S, _p(err_flag), _stream())


def ce_fwd_bwd(logits, labels, tok0, rows, V, ld, S, inv_count, grad_scale, row_loss, lse_out=None, write_grad=True,
               ignore_index=-100):
    call("mgen_ce_fwd_bwd", _p(logits), _p(labels), tok0, rows, V, ld, S, ignore_index, _p(inv_count), grad_scale,
         _p(row_loss), _p(lse_out), int(write_grad), int(logits.dtype == torch.bfloat16), _stream())


def lmhead_ce_workspace(T, V, chunk_tokens, device):
    from ._lib import lib
    n = lib().mgen_lmhead_ce_workspace_bytes(T, V, chunk_tokens)
    return torch.empty(n, dtype=torch.uint8, device=device)


def lmhead_ce_fwd_bwd(x, w, labels, V, chunk_tokens, workspace, loss, dx, dw, err_flag=None, ignore_index=-100):
    """Chunked lm_head + cross entropy (+ backward when dx is given; dw may be None for a frozen head), one call."""
    _need_gpu(x, w, labels, workspace, loss, dx, dw)
    B, S = labels.shape
    d = x.shape[1]
    call("mgen_lmhead_ce_fwd_bwd", _p(x), x.stride(0), _p(w), w.stride(0), _p(labels), B, S, V, d, ignore_index,
         chunk_tokens, _p(workspace), workspace.numel(), _p(loss), _p(dx), dx.stride(0) if dx is not None else 0, _p(dw),
         int(dx is not None), _p(err_flag), _stream())


def ce_finalize(row_loss, inv_count, loss):
    call("mgen_ce_finalize", _p(row_loss), row_loss.numel(), _p(inv_count), _p(loss), _stream())


def grad_norm_clip(grad, max_norm, workspace, out):
    call("mgen_grad_norm_clip", _p(grad), grad.numel(), max_norm, _p(workspace), _p(out), _stream())


def adamw_step(p, g, m, v, shadow, clip, lr, beta1, beta2, eps, wd, step):
    call("mgen_adamw_step", _p(p), _p(g), _p(m), _p(v), _p(shadow), p.numel(), _p(clip), lr, beta1, beta2, eps, wd,
         step, _stream())


def scale_(x, s=1.0, a_dev=None):
    call("mgen_scale_f32", _p(x), x.numel(), s, _p(a_dev), _stream())


def axpy_(y, x, a=1.0, a_dev=None):
    call("mgen_axpy_f32", _p(y), _p(x), x.numel(), a, _p(a_dev), _stream())


def kv_append(k_src, v_src, ld_k, ld_v, k_cache, v_cache, pos0, B, n_new, KVH, D, max_len):
    """k_src/v_src are raw device pointers of token-major rows; caches are [B, KVH, max_len, D] bf16 tensors."""
    call("mgen_kv_append", k_src, v_src, ld_k, ld_v, _p(k_cache), _p(v_cache), _p(pos0), B, n_new, KVH, D, max_len,
         _stream())


def attn_decode(q, ld_q, k_cache, v_cache, kv_len, out, ld_o, workspace, B, H, KVH, D, max_len, nsplit, scale):
    call("mgen_attn_decode", q, ld_q, _p(k_cache), _p(v_cache), _p(kv_len), _p(out), ld_o, _p(workspace), B, H, KVH, D,
         max_len, nsplit, scale, _stream())


def attn_decode_prep(qkv, ld_in, wq, wk, cs, pos, eps, interleaved, k_cache, v_cache, kv_len, out, ld_o, workspace, B, H,
                     KVH, D, max_len, nsplit, scale):
    """decode_qkv_prep + attn_decode in one launch (q, new key and value taken from the raw projection rows)."""
    call("mgen_attn_decode_prep", _p(qkv), ld_in, _p(wq), _p(wk), _p(cs), _p(pos), eps, int(interleaved), _p(k_cache),
         _p(v_cache), _p(kv_len), _p(out), ld_o, _p(workspace), B, H, KVH, D, max_len, nsplit, scale, _stream())


def decode_layer(ptrs17, cos_sin, pos, kv_len, cache_and_shape, nsplit, eps, interleaved):
    """One decoder layer of the one-token step in one library call. ptrs17: raw pointers x, h_in, h_mid, h_out, qkv,
    att, gu, slabs, attention workspace, w_qkv, w_qn, w_kn, w_o, w_ln2, w_gu, w_down, w_next; cache_and_shape:
    k_cache, v_cache pointers then B, d, H, KVH, D, I, max_len."""
    call("mgen_decode_layer", *ptrs17, cos_sin, pos, kv_len, *cache_and_shape, nsplit, eps, interleaved, _stream())


def attn_decode_workspace(B, H, D, nsplit):
    from ._lib import lib
    return lib().mgen_attn_decode_workspace(B, H, D, nsplit)


def sample(logits, V, out, do_sample=False, temperature=1.0, top_k=0, top_p=1.0, noise=None, seed=0, offset=0,
           out2=None, out2_stride=0, row0=0, step=None, unfinished=None, eos_id=0, pad_id=0):
    """logits [B, ld] (bf16 or fp32) -> out [B] int64; out2 = raw pointer of an optional strided second destination.
    row0: the rows are rows [row0, row0 + B) of a larger batch (Philox stream of row i = that of row row0 + i).
    step (int32 device tensor): added to `offset` and to the out2 column when the kernel runs; unfinished ([B] int64
    device tensor): finished streams receive pad_id, a stream that draws eos_id becomes finished."""
    B = logits.shape[0]
    call("mgen_sample_step", _p(logits), logits.stride(0), B, V, int(logits.dtype == torch.bfloat16), int(do_sample),
         float(temperature), int(top_k or 0), float(top_p), _p(noise), noise.stride(0) if noise is not None else 0,
         int(seed), int(offset), int(row0), _p(out), out2 or 0, out2_stride, _p(step), _p(unfinished), int(eos_id),
         int(pad_id), _stream())


def decode_advance(pos, kv_len, step):
    """pos += 1, kv_len += 1 (int32 device tensors of one entry per stream), step[0] += 1."""
    call("mgen_decode_advance", _p(pos), _p(kv_len), pos.numel(), _p(step), _stream())


def decode_qkv_prep(qkv, ld_in, qout, ld_q, k_cache, v_cache, wq, wk, cs, pos, B, HQ, HK, D, max_len, eps,
                    interleaved=False):
    """One decode step: q/k norm + rotation, q to `qout` (raw pointer), k and v into their cache slots."""
    call("mgen_decode_qkv_prep", _p(qkv), ld_in, qout, ld_q, _p(k_cache), _p(v_cache), _p(wq), _p(wk), _p(cs),
         _p(pos), B, HQ, HK, D, max_len, eps, int(interleaved), _stream())


def philox_exponential(out, V, seed, offset):
    B = out.shape[0]
    call("mgen_philox_exponential", _p(out), out.stride(0), B, V, int(seed), int(offset), _stream())


def scale_bf16_(x, a=1.0, a_dev=None):
    call("mgen_scale_bf16", _p(x), x.numel(), a, _p(a_dev), _stream())
