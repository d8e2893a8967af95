// Per-head q/k RMSNorm + rotary position embedding, forward and backward, and the cos/sin table builders.
// HBM-bound: every (token, head) row is read once and written once with 16-byte accesses; the cos/sin values of a
// token are loaded once into registers and reused for all of its heads.
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

template <typename IoT> __device__ __forceinline__ void load8(const IoT* p, float* x) {
    if (sizeof(IoT) == 2) {
        bf16x8 v = ld_stream(reinterpret_cast<const bf16x8*>(p));
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = (float)v[e];
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = (float)p[e];
    }
}
template <typename IoT> __device__ __forceinline__ void store8(IoT* p, const float* x) {
    if (sizeof(IoT) == 2) {
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16)x[e];
        st_stream(reinterpret_cast<bf16x8*>(p), v);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) p[e] = (IoT)x[e];
    }
}

// D/16 lanes per token. A lane owns two groups of 8 elements of every head row of its token, a[0..7] at offset o1 and
// a[8..15] at offset o2, chosen so that both members of every rotation pair sit in the same lane:
//   split-half: o1 = 8*sub, o2 = D/2 + 8*sub; element e < 8 pairs with e + 8; both use table entry 8*sub + e
//   adjacent  : o1 = 16*sub, o2 = 16*sub + 8; element e pairs with e ^ 1; table entry 8*sub + e/2
// so the lane needs 8 consecutive (cos, sin) entries of its token: 64 bytes, loaded once and used for all heads.
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
                                                          float* __restrict__ dwq, float* __restrict__ dwk,
                                                          int per_token) {
    constexpr int LPR = D / 16;            // lanes per token
    constexpr int TPB = 256 / LPR;         // tokens per block per iteration
    const int sub = threadIdx.x % LPR;
    const int tib = threadIdx.x / LPR;
    const int NH = HQ + HK + HV;
    const int o1 = IL ? 16 * sub : 8 * sub;
    const int o2 = IL ? 16 * sub + 8 : D / 2 + 8 * sub;
    const bool norm = wq != nullptr;
    float gq[16], gk[16], dgq[16], dgk[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int off = e < 8 ? o1 + e : o2 + e - 8;
        gq[e] = norm ? wq[off] : 1.f;
        gk[e] = norm ? wk[off] : 1.f;
        dgq[e] = dgk[e] = 0.f;
    }

    // the LPR lanes of a token share `t`, so they leave the loop together (the row reductions stay inside a token)
    for (long t = (long)blockIdx.x * TPB + tib; t < T; t += (long)gridDim.x * TPB) {
        float c[8], s[8];
        {
            const long trow = per_token ? t : (long)(pos ? pos[t] : (int)(t % S));   // table row: the token's own, or its position's
            const float* cp = cs + (trow * (D / 2) + 8 * sub) * 2;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(cp + 2 * e);
                c[e] = v[0]; s[e] = v[1]; c[e + 1] = v[2]; s[e + 1] = v[3];
            }
        }
        // the rows of head hh: a = input (backward: d(out)), x = forward input (backward with a norm)
        auto request = [&](int hh, float (&a)[16], float (&x)[16]) {
            const bool nrm = norm && hh < HQ + HK;
            const IoT* src = in + t * ld_in + (long)hh * D;
            load8(src + o1, a);
            load8(src + o2, a + 8);
            if (!FWD && nrm) {
                const IoT* rsrc = raw + t * ld_raw + (long)hh * D;
                load8(rsrc + o1, x);
                load8(rsrc + o2, x + 8);
            }
        };
        auto finish = [&](int hh, float (&a)[16], float (&x)[16]) {
            const bool isq = hh < HQ;
            const bool isk = !isq && hh < HQ + HK;
            const bool nrm = norm && (isq || isk);
            IoT* outp = isq ? qout + t * ld_q + (long)hh * D
                            : (isk ? kout + t * ld_k + (long)(hh - HQ) * D : vout + t * ld_v + (long)(hh - HQ - HK) * D);
            float o[16];
            if (FWD) {
                if (nrm) {
                    float ss = 0.f;
#pragma unroll
                    for (int e = 0; e < 16; ++e) ss += a[e] * a[e];
                    ss = group_sum<LPR>(ss);
                    const float rs = rsqrtf(ss / (float)D + eps);
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float xn = a[e] * rs;
                        if (sizeof(IoT) == 2) xn = round_bf(xn);
                        a[e] = (isq ? gq[e] : gk[e]) * xn;
                    }
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int pe = IL ? (e ^ 1) : (e ^ 8);                 // partner element
                    const int ci = IL ? (e >> 1) : (e & 7);                // table entry
                    const bool second = IL ? (e & 1) : (e >= 8);           // second member of its pair
                    o[e] = second ? (a[e] * c[ci] + a[pe] * s[ci]) : (a[e] * c[ci] - a[pe] * s[ci]);
                }
            } else {
                // backward: `in` holds d(out) (same head order); the result may be written over it (each row is read
                // and written by the same lanes). `raw` = forward input. Transpose of the rotation:
                // dy1 = d1*cos + d2*sin ; dy2 = d2*cos - d1*sin
                float dy[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int pe = IL ? (e ^ 1) : (e ^ 8);
                    const int ci = IL ? (e >> 1) : (e & 7);
                    const bool second = IL ? (e & 1) : (e >= 8);
                    dy[e] = second ? (a[e] * c[ci] - a[pe] * s[ci]) : (a[e] * c[ci] + a[pe] * s[ci]);
                }
                if (nrm) {
                    float ss = 0.f;
#pragma unroll
                    for (int e = 0; e < 16; ++e) ss += x[e] * x[e];
                    ss = group_sum<LPR>(ss);
                    const float rs = rsqrtf(ss / (float)D + eps);
                    float dot = 0.f;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        x[e] *= rs;                                        // normalised input
                        const float gx = dy[e] * (isq ? gq[e] : gk[e]);
                        dot += gx * x[e];
                        if (isq) dgq[e] += dy[e] * x[e];
                        else dgk[e] += dy[e] * x[e];
                        o[e] = gx;
                    }
                    dot = group_sum<LPR>(dot) / (float)D;
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[e] = (o[e] - x[e] * dot) * rs;
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[e] = dy[e];
                }
            }
            store8(outp + o1, o);
            store8(outp + o2, o + 8);
        };
        // gridDim.y > 1 (few tokens: decode) spreads the heads of a token over workgroups. Several heads per
        // iteration: the rows of all of them are requested before the first one's row reduction.
        // (written out for two heads: the same loop over an array of heads compiled to a forward kernel that ran at
        // 129 us instead of 110)
        const int hstep = gridDim.y;
        for (int hh = blockIdx.y; hh < NH; hh += 2 * hstep) {
            const int h1 = hh + hstep;
            float a0[16], x0[16], a1[16], x1[16];
            request(hh, a0, x0);
            if (h1 < NH) request(h1, a1, x1);
            finish(hh, a0, x0);
            if (h1 < NH) finish(h1, a1, x1);
        }
    }
    if (!FWD && norm) {
        // reduce the per-lane gain gradients over the tokens of the block (lanes with equal `sub`)
        __shared__ float red[256][17];
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 16; ++e) red[threadIdx.x][e] = which ? dgk[e] : dgq[e];
            __syncthreads();
            if (threadIdx.x < D) {
                const int el = threadIdx.x;
                int ls, le;   // owning lane and slot of element el
                if (IL) { ls = el / 16; le = el % 16; }
                else if (el < D / 2) { ls = el / 8; le = el % 8; }
                else { ls = (el - D / 2) / 8; le = 8 + (el - D / 2) % 8; }
                float acc = 0.f;
                for (int tt = 0; tt < TPB; ++tt) acc += red[tt * LPR + ls][le];
                atomicAdd((which ? dwk : dwq) + el, acc);
            }
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
                              float* dwq, float* dwk, int is_bf16, hipStream_t stream, const char* what,
                              int per_token = IL ? 1 : 0) {
    const int tpb = 256 / (D / 16);
    long nb = (T + tpb - 1) / tpb;
    // (the backward ends with a block-wide reduction of the gain gradients and one atomic per gain element: fewer,
    // longer blocks amortise it)
    const long cap = (!FWD && wq) ? 512 : 4096;   // 175 -> 160 us at 65536 tokens x 16 heads (1024: 167, 256: 241)
    if (nb > cap) nb = cap;
    const int nh = HQ + HK + HV;
    const unsigned ny = nb >= 512 ? 1u : (unsigned)(nh < 32 ? nh : 32);   // few tokens: one head (group) per workgroup
#define LAUNCH(DD, TT)                                                                                         \
    hipLaunchKernelGGL((qknorm_rope_kernel<DD, TT, FWD, IL>), dim3((unsigned)nb, ny), dim3(256), 0, stream,    \
                       (const TT*)in, ld_in, (TT*)qout, (TT*)kout, (TT*)vout, ld_q, ld_k, ld_v, wq, wk, cs,    \
                       pos, S, T, HQ, HK, HV, eps, (const TT*)raw, ld_raw, dwq, dwk, per_token)
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

// The same two with ADJACENT pairs (elements 2i, 2i+1 rotate by the angle of frequency i: the GPT-J / "interleaved"
// pairing) and the position-indexed table of mgen_rope_table. Equal to the split-half form on q/k projections whose
// output rows are permuted [0, 2, 4, ..., 1, 3, 5, ...] within each head.
extern "C" int mgen_qknorm_rope_il_fwd(const void* qkv, long ld_in, void* qout, void* kout, long ld_q, long ld_k,
                                       const float* wq, const float* wk, const float* cos_sin, const int* pos, int S,
                                       long T, int HQ, int HK, int D, float eps, int is_bf16, hipStream_t stream) {
    MGEN_REQUIRE(qkv && qout && kout && cos_sin, "mgen_qknorm_rope_il_fwd: null pointer");
    MGEN_REQUIRE(D == 128 || D == 64, "mgen_qknorm_rope_il_fwd: head_dim must be 64 or 128 (got %d)", D);
    MGEN_REQUIRE((wq == nullptr) == (wk == nullptr), "mgen_qknorm_rope_il_fwd: wq and wk must both be given or both null");
    MGEN_REQUIRE(ld_in % 8 == 0 && ld_q % 8 == 0 && ld_k % 8 == 0, "mgen_qknorm_rope_il_fwd: strides must be multiples of 8");
    MGEN_REQUIRE(T > 0 && S > 0 && HQ > 0 && HK > 0, "mgen_qknorm_rope_il_fwd: bad sizes");
    return qknorm_rope_launch<true, true>(qkv, ld_in, qout, kout, nullptr, ld_q, ld_k, 0, wq, wk, cos_sin, pos, S, T,
                                          HQ, HK, 0, D, eps, nullptr, 0, nullptr, nullptr, is_bf16, stream,
                                          "mgen_qknorm_rope_il_fwd", 0);
}

extern "C" int mgen_qknorm_rope_il_bwd(const void* dqk, long ld_in, void* dq_raw, void* dk_raw, long ld_q, long ld_k,
                                       const void* raw_qkv, long ld_raw, const float* wq, const float* wk, float* dwq,
                                       float* dwk, const float* cos_sin, const int* pos, int S, long T, int HQ, int HK,
                                       int D, float eps, int is_bf16, hipStream_t stream) {
    MGEN_REQUIRE(dqk && dq_raw && dk_raw && cos_sin, "mgen_qknorm_rope_il_bwd: null pointer");
    MGEN_REQUIRE(D == 128 || D == 64, "mgen_qknorm_rope_il_bwd: head_dim must be 64 or 128 (got %d)", D);
    MGEN_REQUIRE((wq == nullptr) == (wk == nullptr), "mgen_qknorm_rope_il_bwd: wq and wk must both be given or both null");
    if (wq) MGEN_REQUIRE(raw_qkv && dwq && dwk, "mgen_qknorm_rope_il_bwd: norm backward needs raw_qkv, dwq, dwk");
    MGEN_REQUIRE(ld_in % 8 == 0 && ld_q % 8 == 0 && ld_k % 8 == 0 && ld_raw % 8 == 0,
                 "mgen_qknorm_rope_il_bwd: strides must be multiples of 8");
    MGEN_REQUIRE(T > 0 && S > 0 && HQ > 0 && HK > 0, "mgen_qknorm_rope_il_bwd: bad sizes");
    return qknorm_rope_launch<false, true>(dqk, ld_in, dq_raw, dk_raw, nullptr, ld_q, ld_k, 0, wq, wk, cos_sin, pos,
                                           S, T, HQ, HK, 0, D, eps, raw_qkv, ld_raw, dwq, dwk, is_bf16, stream,
                                           "mgen_qknorm_rope_il_bwd", 0);
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
