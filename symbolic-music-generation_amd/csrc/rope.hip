// Per-head q/k RMSNorm + rotary position embedding, forward and backward, and the cos/sin table builders.
// HBM-bound; 16 lanes per (token, head) row of 128 (or 8 lanes per row of 64), 16-byte accesses.
//
// Two pairings:
//
//  split-half (token models). Reference: q_norm/k_norm + apply_rotary_pos_emb in Qwen3Attention.forward
//  (transformers models/qwen3/modeling_qwen3.py:252-257; rotary table :126-137; rotate_half :140-145):
//    xn  = (x.float() * rsqrt(mean(x^2) + eps)).to(bf16)         (x = bf16 projection output under autocast)
//    y   = w * xn                                                 (fp32)
//    out = y * cos + rotate_half(y) * sin,  rotate_half(y) = cat(-y[D/2:], y[:D/2])
//  and the attention kernel then reads `out` rounded to bf16 (autocast at scaled_dot_product_attention).
//  For Llama-style blocks (no q/k norm) w is null and the row is only rotated. The table holds cos/sin of
//  pos * inv_freq[i] for i < D/2 (the two halves of the reference's `emb` are equal), one row per position.
//
//  adjacent pairs (4-D position model). Reference: /root/reference/src/model/modeling.py:87-98 (rotate_half on
//  (x0,x1) -> (-x1,x0)), :164-180 (rotation of q, k, of v, and the inverse rotation), :264-270, :300-302 (where they
//  are applied), :498-545 (angles: 4 axes x D/8 frequencies, each angle used for one adjacent pair). The table has
//  one row of D/2 (cos, sin) pairs per TOKEN (angles depend on the note, not on the index in the sequence). Value
//  heads are rotated without a norm; a launch with only "value" rows is the plain rotation used on the attention
//  output (backward direction = inverse rotation) and on its gradient (forward direction).
#include "common.h"

namespace {

__global__ void rope_table_kernel(const float* __restrict__ inv_freq, float* __restrict__ cs, int npos, int half) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npos * half) return;
    const int pos = i / half, f = i % half;
    const float ang = (float)pos * inv_freq[f];
    cs[2 * (long)i] = cosf(ang);
    cs[2 * (long)i + 1] = sinf(ang);
}

// cs[t][a * nf + j] = (cos, sin)(pos4[t][a] * freqs[j]),  a < 4 axes, j < nf = D/8 frequencies
__global__ void rope4d_table_kernel(const float* __restrict__ pos4, const float* __restrict__ freqs,
                                    float* __restrict__ cs, long T, int nf) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * 4 * nf) return;
    const long t = i / (4 * nf);
    const int r = (int)(i % (4 * nf));
    const float ang = pos4[t * 4 + r / nf] * freqs[r % nf];
    cs[2 * i] = cosf(ang);
    cs[2 * i + 1] = sinf(ang);
}

// One row = one (token, head). LPR lanes per row, each lane owns 8 consecutive elements.
// Split-half: the rotation partner of element i < D/2 is i + D/2, held by lane (l ^ LPR/2).
// Adjacent pairs (IL): the partner of element e is e^1, held by the same lane.
//  in : qkv  [T, ld_in] (q heads then k heads then v heads; the first HQ+HK+HV heads are touched)
//  out: qout [T, ld_q], kout [T, ld_k], vout [T, ld_v]
template <int D, typename IoT, bool FWD, bool IL>
__global__ __launch_bounds__(256) void qknorm_rope_kernel(const IoT* __restrict__ in, long ld_in,
                                                          IoT* __restrict__ qout, IoT* __restrict__ kout,
                                                          IoT* __restrict__ vout, long ld_q, long ld_k, long ld_v,
                                                          const float* __restrict__ wq, const float* __restrict__ wk,
                                                          const float* __restrict__ cs, const int* __restrict__ pos,
                                                          int S, long T, int HQ, int HK, int HV, float eps,
                                                          const IoT* __restrict__ raw, long ld_raw,
                                                          float* __restrict__ dwq, float* __restrict__ dwk) {
    constexpr int LPR = D / 8;             // lanes per row
    constexpr int RPB = 256 / LPR;         // rows per block per iteration
    const int sub = threadIdx.x % LPR;     // lane within row
    const int rib = threadIdx.x / LPR;     // row within block
    const long NH = HQ + HK + HV;
    const long nrows = T * NH;
    const bool hi = sub >= LPR / 2;
    const int fi = (sub % (LPR / 2)) * 8;  // first frequency index of this lane (split-half)
    float dwacc_q[8], dwacc_k[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) dwacc_q[e] = dwacc_k[e] = 0.f;

    // the LPR lanes of a row share `row`, so they leave the loop together (the shuffles stay inside a row)
    for (long row = (long)blockIdx.x * RPB + rib; row < nrows; row += (long)gridDim.x * RPB) {
        const long t = row / NH;
        const int hh = (int)(row % NH);
        const bool isq = hh < HQ;
        const bool isk = !isq && hh < HQ + HK;
        const float* w = isq ? wq : (isk ? wk : nullptr);
        float cosv[8], sinv[8];
        if (IL) {
            const float* c = cs + (t * (D / 2) + sub * 4) * 2;
            f32x4 v0 = *reinterpret_cast<const f32x4*>(c);
            f32x4 v1 = *reinterpret_cast<const f32x4*>(c + 4);
            cosv[0] = cosv[1] = v0[0]; sinv[0] = sinv[1] = v0[1];
            cosv[2] = cosv[3] = v0[2]; sinv[2] = sinv[3] = v0[3];
            cosv[4] = cosv[5] = v1[0]; sinv[4] = sinv[5] = v1[1];
            cosv[6] = cosv[7] = v1[2]; sinv[6] = sinv[7] = v1[3];
        } else {
            const int p = pos ? pos[t] : (int)(t % S);
            const float* c = cs + ((long)p * (D / 2) + fi) * 2;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                f32x4 v = *reinterpret_cast<const f32x4*>(c + 2 * e);
                cosv[e] = v[0]; sinv[e] = v[1]; cosv[e + 1] = v[2]; sinv[e + 1] = v[3];
            }
        }
        IoT* outp = isq ? qout + t * ld_q + (long)hh * D + sub * 8
                        : (isk ? kout + t * ld_k + (long)(hh - HQ) * D + sub * 8
                               : vout + t * ld_v + (long)(hh - HQ - HK) * D + sub * 8);
        float wv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) wv[e] = w ? w[sub * 8 + e] : 1.f;

        if (FWD) {
            float x[8];
            const IoT* src = in + t * ld_in + (long)hh * D + sub * 8;
            if (sizeof(IoT) == 2) {
                bf16x8 v = *reinterpret_cast<const bf16x8*>(src);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = (float)v[e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = (float)src[e];
            }
            float y[8];
            if (w) {
                float ss = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) ss += x[e] * x[e];
                ss = group_sum<LPR>(ss);
                const float rs = rsqrtf(ss / (float)D + eps);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float xn = x[e] * rs;
                    if (sizeof(IoT) == 2) xn = round_bf(xn);
                    y[e] = wv[e] * xn;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = x[e];
            }
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (IL) {
                    o[e] = (e & 1) ? (y[e] * cosv[e] + y[e ^ 1] * sinv[e]) : (y[e] * cosv[e] - y[e ^ 1] * sinv[e]);
                } else {
                    const float other = __shfl_xor(y[e], LPR / 2, 64);
                    o[e] = hi ? (y[e] * cosv[e] + other * sinv[e]) : (y[e] * cosv[e] - other * sinv[e]);
                }
            }
            if (sizeof(IoT) == 2) {
                bf16x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (bf16)o[e];
                *reinterpret_cast<bf16x8*>(outp) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) outp[e] = (IoT)o[e];
            }
        } else {
            // backward: `in` holds d(out) ([T, ld_in], same head order); result written to qout/kout/vout (may
            // alias `in`: each row is read and written by the same lanes). `raw` = forward input.
            float dout[8];
            const IoT* src = in + t * ld_in + (long)hh * D + sub * 8;
            if (sizeof(IoT) == 2) {
                bf16x8 v = *reinterpret_cast<const bf16x8*>(src);
#pragma unroll
                for (int e = 0; e < 8; ++e) dout[e] = (float)v[e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) dout[e] = (float)src[e];
            }
            float dy[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                // transpose of the rotation: dy1 = d1*cos + d2*sin ; dy2 = d2*cos - d1*sin
                if (IL) {
                    dy[e] = (e & 1) ? (dout[e] * cosv[e] - dout[e ^ 1] * sinv[e])
                                    : (dout[e] * cosv[e] + dout[e ^ 1] * sinv[e]);
                } else {
                    const float other = __shfl_xor(dout[e], LPR / 2, 64);
                    dy[e] = hi ? (dout[e] * cosv[e] - other * sinv[e]) : (dout[e] * cosv[e] + other * sinv[e]);
                }
            }
            float dx[8];
            if (w) {
                float x[8];
                const IoT* rsrc = raw + t * ld_raw + (long)hh * D + sub * 8;
                if (sizeof(IoT) == 2) {
                    bf16x8 v = *reinterpret_cast<const bf16x8*>(rsrc);
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = (float)v[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = (float)rsrc[e];
                }
                float ss = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) ss += x[e] * x[e];
                ss = group_sum<LPR>(ss);
                const float rs = rsqrtf(ss / (float)D + eps);
                float dot = 0.f;
                float xh[8], gx[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xh[e] = x[e] * rs;
                    gx[e] = dy[e] * wv[e];
                    dot += gx[e] * xh[e];
                    if (isq) dwacc_q[e] += dy[e] * xh[e];
                    else dwacc_k[e] += dy[e] * xh[e];
                }
                dot = group_sum<LPR>(dot) / (float)D;
#pragma unroll
                for (int e = 0; e < 8; ++e) dx[e] = (gx[e] - xh[e] * dot) * rs;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) dx[e] = dy[e];
            }
            if (sizeof(IoT) == 2) {
                bf16x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (bf16)dx[e];
                *reinterpret_cast<bf16x8*>(outp) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) outp[e] = (IoT)dx[e];
            }
        }
    }
    if (!FWD && wq) {
        // reduce the per-thread gain gradients over the rows of the block (threads with equal `sub`)
        __shared__ float red[2][256][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[0][threadIdx.x][e] = dwacc_q[e];
            red[1][threadIdx.x][e] = dwacc_k[e];
        }
        __syncthreads();
        if (threadIdx.x < 2 * D) {
            const int which = threadIdx.x / D, el = threadIdx.x % D;
            float s = 0.f;
            for (int rr = 0; rr < RPB; ++rr) s += red[which][rr * LPR + el / 8][el % 8];
            atomicAdd((which ? dwk : dwq) + el, s);
        }
    }
}

}  // namespace

extern "C" int mgen_rope_table(const float* inv_freq, float* cos_sin, int npos, int half_dim, hipStream_t stream) {
    MGEN_REQUIRE(inv_freq && cos_sin && npos > 0 && half_dim > 0, "mgen_rope_table: bad arguments");
    const long n = (long)npos * half_dim;
    hipLaunchKernelGGL(rope_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, inv_freq, cos_sin,
                       npos, half_dim);
    return mgen_check_launch("mgen_rope_table");
}

extern "C" int mgen_rope4d_table(const float* pos4, const float* freqs, float* cos_sin, long T, int D,
                                 hipStream_t stream) {
    MGEN_REQUIRE(pos4 && freqs && cos_sin && T > 0, "mgen_rope4d_table: bad arguments");
    MGEN_REQUIRE(D == 128 || D == 64, "mgen_rope4d_table: head_dim must be 64 or 128 (got %d)", D);
    const long n = T * (D / 2);
    hipLaunchKernelGGL(rope4d_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, pos4, freqs,
                       cos_sin, T, D / 8);
    return mgen_check_launch("mgen_rope4d_table");
}

template <bool FWD, bool IL>
static int qknorm_rope_launch(const void* in, long ld_in, void* qout, void* kout, void* vout, long ld_q, long ld_k,
                              long ld_v, const float* wq, const float* wk, const float* cs, const int* pos, int S,
                              long T, int HQ, int HK, int HV, int D, float eps, const void* raw, long ld_raw,
                              float* dwq, float* dwk, int is_bf16, hipStream_t stream, const char* what) {
    const long nrows = T * (HQ + HK + HV);
    const int rpb = 256 / (D / 8);
    long nb = (nrows + rpb - 1) / rpb;
    if (nb > 4096) nb = 4096;
#define LAUNCH(DD, TT)                                                                                         \
    hipLaunchKernelGGL((qknorm_rope_kernel<DD, TT, FWD, IL>), dim3((unsigned)nb), dim3(256), 0, stream,        \
                       (const TT*)in, ld_in, (TT*)qout, (TT*)kout, (TT*)vout, ld_q, ld_k, ld_v, wq, wk, cs,    \
                       pos, S, T, HQ, HK, HV, eps, (const TT*)raw, ld_raw, dwq, dwk)
    if (D == 128 && is_bf16) LAUNCH(128, bf16);
    else if (D == 128) LAUNCH(128, float);
    else if (D == 64 && is_bf16) LAUNCH(64, bf16);
    else LAUNCH(64, float);
#undef LAUNCH
    return mgen_check_launch(what);
}

extern "C" int mgen_qknorm_rope_fwd(const void* qkv, long ld_in, void* qout, void* kout, long ld_q, long ld_k,
                                    const float* wq, const float* wk, const float* cos_sin, const int* pos, int S,
                                    long T, int HQ, int HK, int D, float eps, int is_bf16, hipStream_t stream) {
    MGEN_REQUIRE(qkv && qout && kout && cos_sin, "mgen_qknorm_rope_fwd: null pointer");
    MGEN_REQUIRE(D == 128 || D == 64, "mgen_qknorm_rope_fwd: head_dim must be 64 or 128 (got %d)", D);
    MGEN_REQUIRE((wq == nullptr) == (wk == nullptr), "mgen_qknorm_rope_fwd: wq and wk must both be given or both null");
    MGEN_REQUIRE(ld_in % 8 == 0 && ld_q % 8 == 0 && ld_k % 8 == 0, "mgen_qknorm_rope_fwd: strides must be multiples of 8");
    MGEN_REQUIRE(T > 0 && S > 0 && HQ > 0 && HK > 0, "mgen_qknorm_rope_fwd: bad sizes");
    return qknorm_rope_launch<true, false>(qkv, ld_in, qout, kout, nullptr, ld_q, ld_k, 0, wq, wk, cos_sin, pos, S, T,
                                           HQ, HK, 0, D, eps, nullptr, 0, nullptr, nullptr, is_bf16, stream,
                                           "mgen_qknorm_rope_fwd");
}

extern "C" int mgen_qknorm_rope_bwd(const void* dqk, long ld_in, void* dq_raw, void* dk_raw, long ld_q, long ld_k,
                                    const void* raw_qkv, long ld_raw, const float* wq, const float* wk, float* dwq,
                                    float* dwk, const float* cos_sin, const int* pos, int S, long T, int HQ, int HK,
                                    int D, float eps, int is_bf16, hipStream_t stream) {
    MGEN_REQUIRE(dqk && dq_raw && dk_raw && cos_sin, "mgen_qknorm_rope_bwd: null pointer");
    MGEN_REQUIRE(D == 128 || D == 64, "mgen_qknorm_rope_bwd: head_dim must be 64 or 128 (got %d)", D);
    MGEN_REQUIRE((wq == nullptr) == (wk == nullptr), "mgen_qknorm_rope_bwd: wq and wk must both be given or both null");
    if (wq) MGEN_REQUIRE(raw_qkv && dwq && dwk, "mgen_qknorm_rope_bwd: norm backward needs raw_qkv, dwq, dwk");
    MGEN_REQUIRE(ld_in % 8 == 0 && ld_q % 8 == 0 && ld_k % 8 == 0 && ld_raw % 8 == 0,
                 "mgen_qknorm_rope_bwd: strides must be multiples of 8");
    MGEN_REQUIRE(T > 0 && S > 0 && HQ > 0 && HK > 0, "mgen_qknorm_rope_bwd: bad sizes");
    return qknorm_rope_launch<false, false>(dqk, ld_in, dq_raw, dk_raw, nullptr, ld_q, ld_k, 0, wq, wk, cos_sin, pos,
                                            S, T, HQ, HK, 0, D, eps, raw_qkv, ld_raw, dwq, dwk, is_bf16, stream,
                                            "mgen_qknorm_rope_bwd");
}

// Adjacent-pair rotation with a per-token table [T, D/2, 2]. Head counts may be zero: with HQ = HK = 0 the call is a
// plain rotation of HV heads (fwd: by +angle, bwd: by -angle).
extern "C" int mgen_qkvnorm_rope4d_fwd(const void* qkv, long ld_in, void* qout, void* kout, void* vout, long ld_q,
                                       long ld_k, long ld_v, const float* wq, const float* wk, const float* cos_sin,
                                       long T, int HQ, int HK, int HV, int D, float eps, int is_bf16,
                                       hipStream_t stream) {
    MGEN_REQUIRE(qkv && cos_sin, "mgen_qkvnorm_rope4d_fwd: null pointer");
    MGEN_REQUIRE((HQ == 0 || qout) && (HK == 0 || kout) && (HV == 0 || vout), "mgen_qkvnorm_rope4d_fwd: null output");
    MGEN_REQUIRE(D == 128 || D == 64, "mgen_qkvnorm_rope4d_fwd: head_dim must be 64 or 128 (got %d)", D);
    MGEN_REQUIRE((wq == nullptr) == (wk == nullptr), "mgen_qkvnorm_rope4d_fwd: wq and wk must both be given or both null");
    MGEN_REQUIRE(ld_in % 8 == 0 && ld_q % 8 == 0 && ld_k % 8 == 0 && ld_v % 8 == 0,
                 "mgen_qkvnorm_rope4d_fwd: strides must be multiples of 8");
    MGEN_REQUIRE(T > 0 && HQ >= 0 && HK >= 0 && HV >= 0 && HQ + HK + HV > 0, "mgen_qkvnorm_rope4d_fwd: bad sizes");
    return qknorm_rope_launch<true, true>(qkv, ld_in, qout, kout, vout, ld_q, ld_k, ld_v, wq, wk, cos_sin, nullptr, 1,
                                          T, HQ, HK, HV, D, eps, nullptr, 0, nullptr, nullptr, is_bf16, stream,
                                          "mgen_qkvnorm_rope4d_fwd");
}

extern "C" int mgen_qkvnorm_rope4d_bwd(const void* dqkv, long ld_in, void* dq_raw, void* dk_raw, void* dv_raw,
                                       long ld_q, long ld_k, long ld_v, const void* raw_qkv, long ld_raw,
                                       const float* wq, const float* wk, float* dwq, float* dwk,
                                       const float* cos_sin, long T, int HQ, int HK, int HV, int D, float eps,
                                       int is_bf16, hipStream_t stream) {
    MGEN_REQUIRE(dqkv && cos_sin, "mgen_qkvnorm_rope4d_bwd: null pointer");
    MGEN_REQUIRE((HQ == 0 || dq_raw) && (HK == 0 || dk_raw) && (HV == 0 || dv_raw),
                 "mgen_qkvnorm_rope4d_bwd: null output");
    MGEN_REQUIRE(D == 128 || D == 64, "mgen_qkvnorm_rope4d_bwd: head_dim must be 64 or 128 (got %d)", D);
    MGEN_REQUIRE((wq == nullptr) == (wk == nullptr), "mgen_qkvnorm_rope4d_bwd: wq and wk must both be given or both null");
    if (wq && HQ + HK > 0) MGEN_REQUIRE(raw_qkv && dwq && dwk, "mgen_qkvnorm_rope4d_bwd: norm backward needs raw_qkv, dwq, dwk");
    MGEN_REQUIRE(ld_in % 8 == 0 && ld_q % 8 == 0 && ld_k % 8 == 0 && ld_v % 8 == 0 && ld_raw % 8 == 0,
                 "mgen_qkvnorm_rope4d_bwd: strides must be multiples of 8");
    MGEN_REQUIRE(T > 0 && HQ >= 0 && HK >= 0 && HV >= 0 && HQ + HK + HV > 0, "mgen_qkvnorm_rope4d_bwd: bad sizes");
    return qknorm_rope_launch<false, true>(dqkv, ld_in, dq_raw, dk_raw, dv_raw, ld_q, ld_k, ld_v, wq, wk, cos_sin,
                                           nullptr, 1, T, HQ, HK, HV, D, eps, raw_qkv, ld_raw, dwq, dwk, is_bf16,
                                           stream, "mgen_qkvnorm_rope4d_bwd");
}
