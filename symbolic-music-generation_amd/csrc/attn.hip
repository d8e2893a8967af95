// Causal self-attention on bf16 MFMA for gfx950, head_dim 128: forward (online softmax, saves LSE), and backward
// as two kernels (dK/dV per key block, dQ per query block) that recompute P from Q, K and LSE.
//
// Reference: the attention call in Qwen3Attention.forward — transformers integrations/sdpa_attention.py:154-163
// (is_causal, scale = head_dim^-0.5) and its eager twin models/qwen3/modeling_qwen3.py:185-207 (softmax in fp32) —
// plus its autograd backward. Operands are the bf16-rounded q, k (after q/k-norm + RoPE) and v.
//
// Layout: q/k/v/o/dq/dk/dv are token-major [B*S, heads, 128] views with an explicit token stride (so q, k, v may
// live inside one fused projection output). LSE and delta are [B, H, S] fp32.
//
// Structure shared by the three kernels: 256 threads = 4 waves; each wave owns 32 rows of the block's 128 (queries
// in fwd/dQ, keys in dK/dV) and keeps its own operand rows in registers as MFMA B-fragments; the other side is
// streamed in 64-row tiles through two LDS buffers (global -> registers -> LDS, next tile's loads issued before
// the current tile's MFMAs, one barrier per tile). Scores are computed transposed where that puts the softmax row
// on the lane (S^T = K Q^T in fwd/dQ), so row max / row sum need one cross-half exchange only, and the fp32 score
// tile is re-used in place as the bf16 operand of the next product (no LDS round trip for P or dS).
#include "common.h"
#include <type_traits>

namespace {

constexpr int D = 128;
constexpr int TILE = 64 * D * 2;  // bytes of one 64-row bf16 tile
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float NEG_BIG = -1.0e30f;

struct AttnArgs {
    const bf16* Q; const bf16* K; const bf16* V; const bf16* O; const bf16* dO;
    bf16* Out; bf16* dQ; bf16* dK; bf16* dV;
    float* LSE; const float* Delta;
    long sq, sk, sv, so, sdo, sdq, sdk, sdv;  // token strides in elements
    int B, S, H, KVH;
    float scale;
};

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// 64 x 128 tile: 4 x 16 bytes per thread; rows clamped to S-1 (their results are masked or never stored)
__device__ __forceinline__ void tile_load(u32x4 (&st)[4], const bf16* __restrict__ base, long tok_stride, long tok0,
                                          int row0, int S, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i, row = c >> 4, ch = c & 15;
        const int gr = min(row0 + row, S - 1);
        st[i] = *reinterpret_cast<const u32x4*>(base + (tok0 + gr) * tok_stride + ch * 8);
    }
}
__device__ __forceinline__ void tile_store(const u32x4 (&st)[4], char* lds, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i, row = c >> 4, ch = c & 15;
        lds_write_b128(lds, img256_off(row, ch), st[i]);
    }
}

// The same 64 x 128 tile straight into its LDS image by LDS-DMA (no staging registers, no ds_write): a wave
// instruction fills one contiguous 1 KiB piece (4 rows), so the image's chunk swizzle goes on the source address.
__device__ __forceinline__ void tile_glds(const bf16* __restrict__ base, long tok_stride, long tok0, int row0, int S,
                                          char* lds, int tid) {
    const int wave = tid >> 6, l = tid & 63;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int piece = wave * 4 + j;
        const int row = piece * 4 + (l >> 4);
        const int ch = (l & 15) ^ swz256(row);
        const int gr = min(row0 + row, S - 1);
        const bf16* src = base + (tok0 + gr) * tok_stride + ch * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + piece * 1024), 16, 0, 0);
    }
}

// rows of this wave as B-operand fragments: lane (r, hh) holds row (row0 + r), columns 16ks + 8hh .. +7
__device__ __forceinline__ void load_own_rows(bf16x8 (&f)[8], const bf16* __restrict__ base, long tok_stride,
                                              long tok0, int row0, int S, int lane) {
    const int gr = min(row0 + (lane & 31), S - 1);
    const bf16* p = base + (tok0 + gr) * tok_stride + 8 * (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) f[ks] = *reinterpret_cast<const bf16x8*>(p + 16 * ks);
}

// The register-resident operand rows are loaded once, before the main loop. Drain those loads here and pass the
// registers through an empty asm so that the compiler's wait-count bookkeeping no longer ties the loop's MFMAs to
// vector-memory loads: otherwise every MFMA of the first phase waits for the NEXT tile's prefetch loads issued just
// before it, and the prefetch hides nothing.
__device__ __forceinline__ void settle_frags(bf16x8 (&f)[8]) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(f[i]));
}

__device__ __forceinline__ void zero16(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// Store a transposed accumulator set acc[db] = X^T[d = 32db + acc_row(reg)][row = lane&31] as X[row][d] (bf16)
__device__ __forceinline__ void store_rows_T(const f32x16 (&acc)[4], bf16* __restrict__ dst, float mul, int lane) {
    const int hh = lane >> 5;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16)(acc[db][4 * g4 + e] * mul);
            *reinterpret_cast<bf16x4*>(dst + 32 * db + 8 * g4 + 4 * hh) = o;
        }
}

// ===================================================================================================
// forward
// ===================================================================================================
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][K tile | V tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qb = gridDim.x - 1 - blockIdx.x;  // heaviest (latest) query blocks first
    const int h = blockIdx.y, b = blockIdx.z;
    const int kvh = h / (a.H / a.KVH);
    const int S = a.S;
    const long tok0 = (long)b * S;
    const int q0 = qb * 128, qw0 = q0 + wave * 32;
    const bf16* Kb = a.K + (long)kvh * D;
    const bf16* Vb = a.V + (long)kvh * D;

    bf16x8 qf[8];
    load_own_rows(qf, a.Q + (long)h * D, a.sq, tok0, qw0, S, lane);

    f32x16 o[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) zero16(o[db]);
    float m = NEG_BIG, l = 0.f;
    const float sl2 = a.scale * LOG2E;
    const int my_q = qw0 + (lane & 31);

    const int last_key = min(q0 + 127, S - 1);
    const int ntiles = last_key / 64 + 1;

    tile_glds(Kb, a.sk, tok0, 0, S, smem, tid);
    tile_glds(Vb, a.sv, tok0, 0, S, smem + TILE, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    settle_frags(qf);

    for (int j = 0; j < ntiles; ++j) {
        const int cur = j & 1;
        const int kv0 = j * 64;
        if (j + 1 < ntiles) {  // next tile into the other buffer (last read one iteration ago, before a barrier)
            char* nb = smem + (cur ^ 1) * 2 * TILE;
            tile_glds(Kb, a.sk, tok0, kv0 + 64, S, nb, tid);
            tile_glds(Vb, a.sv, tok0, kv0 + 64, S, nb + TILE, tid);
        }
        const char* Kt = smem + cur * 2 * TILE;
        const char* Vt = Kt + TILE;
        if (kv0 <= qw0 + 31) {  // wave-uniform: this wave has at least one visible key in the tile
            // MASKED is a compile-time flag: only tiles that straddle the diagonal pay for the key > query test
            auto tile = [&](auto masked) {
                constexpr bool MASKED = decltype(masked)::value;
                f32x16 s[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    zero16(s[t]);
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks)
                        s[t] = mfma32(frag_row_img256(Kt, 32 * t, ks, lane), qf[ks], s[t]);
                }
                // running max in the RAW score domain (the scale is positive); the scale is folded into the
                // exponent's fma below
                float mx = NEG_BIG;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (MASKED) {
                            const int key = kv0 + 32 * t + acc_row(r, lane);
                            if (key > my_q) s[t][r] = NEG_BIG;
                        }
                        mx = fmaxf(mx, s[t][r]);
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                if (__any(mx > m)) {  // wave-uniform; when no row's max grew the rescale factor is exactly 1
                    const float m_new = fmaxf(m, mx);
                    const float alpha = fast_exp2((m - m_new) * sl2);
                    m = m_new;
                    l *= alpha;
#pragma unroll
                    for (int db = 0; db < 4; ++db)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
                }
                const float nm = -m * sl2;
                float psum = 0.f;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float p = fast_exp2(__builtin_fmaf(s[t][r], sl2, nm));
                        s[t][r] = p;
                        psum += p;
                    }
                l += psum;
                bf16x8 pf[2][2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    pf[t][0] = acc_to_frag<0>(s[t]);
                    pf[t][1] = acc_to_frag<1>(s[t]);
                }
                // O^T[d][q] += V^T[d][key] * P^T[key][q]
#pragma unroll
                for (int db = 0; db < 4; ++db)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2)
                            o[db] = mfma32(frag_col_img256<true>(Vt, 32 * t + 16 * s2, 32 * db, lane), pf[t][s2],
                                           o[db]);
            };
            if (kv0 + 63 > qw0) tile(std::true_type{});
            else tile(std::false_type{});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    const float lt = l + __shfl_xor(l, 32, 64);
    if (my_q < S) {
        const float inv = 1.f / lt;
        store_rows_T(o, a.Out + (tok0 + my_q) * a.so + (long)h * D, inv, lane);
        if (lane < 32 && a.LSE) a.LSE[((long)b * a.H + h) * S + my_q] = (m * sl2 + __log2f(lt)) * LN2;
    }
}

// ===================================================================================================
// backward, part 0: delta[b,h,q] = sum_d dO * O
// ===================================================================================================
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16* __restrict__ O, const bf16* __restrict__ dO,
                                                         long so, long sdo, float* __restrict__ delta, int B, int S,
                                                         int H) {
    const long nrows = (long)B * S * H;
    const int sub = threadIdx.x & 15;
    for (long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4); row < nrows; row += (long)gridDim.x * 16) {
        const long tok = row / H;
        const int h = (int)(row % H);
        const bf16x8 ov = *reinterpret_cast<const bf16x8*>(O + tok * so + (long)h * D + sub * 8);
        const bf16x8 dv = *reinterpret_cast<const bf16x8*>(dO + tok * sdo + (long)h * D + sub * 8);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)ov[e] * (float)dv[e];
        s = group_sum<16>(s);
        if (sub == 0) {
            const long b = tok / S, q = tok % S;
            delta[(b * H + h) * S + q] = s;
        }
    }
}

// ===================================================================================================
// backward, part 1: dQ. Same work decomposition as the forward.
// ===================================================================================================
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qb = gridDim.x - 1 - blockIdx.x;
    const int h = blockIdx.y, b = blockIdx.z;
    const int kvh = h / (a.H / a.KVH);
    const int S = a.S;
    const long tok0 = (long)b * S;
    const int q0 = qb * 128, qw0 = q0 + wave * 32;
    const bf16* Kb = a.K + (long)kvh * D;
    const bf16* Vb = a.V + (long)kvh * D;

    bf16x8 qf[8], dof[8];
    load_own_rows(qf, a.Q + (long)h * D, a.sq, tok0, qw0, S, lane);
    load_own_rows(dof, a.dO + (long)h * D, a.sdo, tok0, qw0, S, lane);
    const int my_q = qw0 + (lane & 31);
    const int cq = min(my_q, S - 1);
    const float lse2 = a.LSE[((long)b * a.H + h) * S + cq] * LOG2E;
    const float delta = a.Delta[((long)b * a.H + h) * S + cq];
    const float sl2 = a.scale * LOG2E;

    f32x16 dq[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) zero16(dq[db]);

    const int last_key = min(q0 + 127, S - 1);
    const int ntiles = last_key / 64 + 1;

    tile_glds(Kb, a.sk, tok0, 0, S, smem, tid);
    tile_glds(Vb, a.sv, tok0, 0, S, smem + TILE, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    settle_frags(qf);
    settle_frags(dof);

    for (int j = 0; j < ntiles; ++j) {
        const int cur = j & 1;
        const int kv0 = j * 64;
        if (j + 1 < ntiles) {  // next tile into the other buffer (last read one iteration ago, before a barrier)
            char* nb = smem + (cur ^ 1) * 2 * TILE;
            tile_glds(Kb, a.sk, tok0, kv0 + 64, S, nb, tid);
            tile_glds(Vb, a.sv, tok0, kv0 + 64, S, nb + TILE, tid);
        }
        const char* Kt = smem + cur * 2 * TILE;
        const char* Vt = Kt + TILE;
        if (kv0 <= qw0 + 31) {
            const bool need_mask = kv0 + 63 > qw0;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x16 s, dp;
                zero16(s);
                zero16(dp);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) s = mfma32(frag_row_img256(Kt, 32 * t, ks, lane), qf[ks], s);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) dp = mfma32(frag_row_img256(Vt, 32 * t, ks, lane), dof[ks], dp);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float p = fast_exp2(__builtin_fmaf(s[r], sl2, -lse2));
                    if (need_mask) {
                        const int key = kv0 + 32 * t + acc_row(r, lane);
                        if (key > my_q) p = 0.f;
                    }
                    s[r] = p * (dp[r] - delta);  // dS^T[key][q]
                }
                const bf16x8 f0 = acc_to_frag<0>(s), f1 = acc_to_frag<1>(s);
                // dQ^T[d][q] += K^T[d][key] * dS^T[key][q]
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    dq[db] = mfma32(frag_col_img256<true>(Kt, 32 * t, 32 * db, lane), f0, dq[db]);
                    dq[db] = mfma32(frag_col_img256<true>(Kt, 32 * t + 16, 32 * db, lane), f1, dq[db]);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (my_q < S) store_rows_T(dq, a.dQ + (tok0 + my_q) * a.sdq + (long)h * D, a.scale, lane);
}

// ===================================================================================================
// backward, part 2: dK and dV. One block = 128 keys of one (batch, kv head); loops over the query heads of the
// group and over the query tiles at or after the block's first key.
// LDS per buffer: Q tile | dO tile | lse2[64] | delta[64]
// ===================================================================================================
constexpr int DKV_BUF = 2 * TILE + 512;

__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kb = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z;
    const int S = a.S;
    const long tok0 = (long)b * S;
    const int k0 = kb * 128, kw0 = k0 + wave * 32;
    const int my_k = kw0 + (lane & 31);
    const int group = a.H / a.KVH;
    const float sl2 = a.scale * LOG2E;

    bf16x8 kf[8], vf[8];
    load_own_rows(kf, a.K + (long)kvh * D, a.sk, tok0, kw0, S, lane);
    load_own_rows(vf, a.V + (long)kvh * D, a.sv, tok0, kw0, S, lane);

    f32x16 dk[4], dv[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) {
        zero16(dk[db]);
        zero16(dv[db]);
    }

    const int t_first = k0 / 64;
    const int t_end = (S + 63) / 64;
    const int per_head = t_end - t_first;
    const int nsteps = per_head * group;

    float s_lse = 0.f, s_delta = 0.f;

    auto issue = [&](int step, int buf) {
        const int hq = kvh * group + step / per_head;
        const int qt0 = (t_first + step % per_head) * 64;
        char* base = smem + buf * DKV_BUF;
        tile_glds(a.Q + (long)hq * D, a.sq, tok0, qt0, S, base, tid);
        tile_glds(a.dO + (long)hq * D, a.sdo, tok0, qt0, S, base + TILE, tid);
        if (tid < 64) {
            const int q = min(qt0 + tid, S - 1);
            const long o = ((long)b * a.H + hq) * S + q;
            s_lse = a.LSE[o] * LOG2E;
            s_delta = a.Delta[o];
        }
    };
    auto commit = [&](int buf) {
        char* base = smem + buf * DKV_BUF;
        if (tid < 64) {
            reinterpret_cast<float*>(base + 2 * TILE)[tid] = s_lse;
            reinterpret_cast<float*>(base + 2 * TILE + 256)[tid] = s_delta;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    issue(0, 0);
    commit(0);
    __syncthreads();
    settle_frags(kf);
    settle_frags(vf);

    for (int step = 0; step < nsteps; ++step) {
        const int cur = step & 1;
        if (step + 1 < nsteps) issue(step + 1, cur ^ 1);
        const int qt0 = (t_first + step % per_head) * 64;
        const char* Qt = smem + cur * DKV_BUF;
        const char* dOt = Qt + TILE;
        const float* lse_s = reinterpret_cast<const float*>(Qt + 2 * TILE);
        const float* del_s = lse_s + 64;
        if (qt0 + 63 >= kw0) {  // wave-uniform: some query of the tile can see some key of this wave
            auto tile = [&](auto masked) {
                constexpr bool MASKED = decltype(masked)::value;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    f32x16 s, dp;
                    zero16(s);
                    zero16(dp);
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) s = mfma32(frag_row_img256(Qt, 32 * u, ks, lane), kf[ks], s);
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) dp = mfma32(frag_row_img256(dOt, 32 * u, ks, lane), vf[ks], dp);
                    // s[reg]: query = qt0 + 32u + acc_row(reg), key = my_k
                    f32x16 ds;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int qr = 32 * u + 8 * g4 + 4 * (lane >> 5);
                        const f32x4 lv = *reinterpret_cast<const f32x4*>(lse_s + qr);
                        const f32x4 dl = *reinterpret_cast<const f32x4*>(del_s + qr);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int r = 4 * g4 + e;
                            float p = fast_exp2(__builtin_fmaf(s[r], sl2, -lv[e]));
                            if (MASKED) {
                                const int q = qt0 + qr + e;
                                if (my_k > q || q >= S) p = 0.f;
                            }
                            s[r] = p;
                            ds[r] = p * (dp[r] - dl[e]);
                        }
                    }
                    const bf16x8 p0 = acc_to_frag<0>(s), p1 = acc_to_frag<1>(s);
                    const bf16x8 d0 = acc_to_frag<0>(ds), d1 = acc_to_frag<1>(ds);
#pragma unroll
                    for (int db = 0; db < 4; ++db) {
                        // dV^T[d][key] += dO^T[d][q] * P[q][key] ; dK^T[d][key] += Q^T[d][q] * dS[q][key]
                        dv[db] = mfma32(frag_col_img256<true>(dOt, 32 * u, 32 * db, lane), p0, dv[db]);
                        dv[db] = mfma32(frag_col_img256<true>(dOt, 32 * u + 16, 32 * db, lane), p1, dv[db]);
                        dk[db] = mfma32(frag_col_img256<true>(Qt, 32 * u, 32 * db, lane), d0, dk[db]);
                        dk[db] = mfma32(frag_col_img256<true>(Qt, 32 * u + 16, 32 * db, lane), d1, dk[db]);
                    }
                }
            };
            if (qt0 < kw0 + 31 || qt0 + 63 >= S) tile(std::true_type{});
            else tile(std::false_type{});
        }
        if (step + 1 < nsteps) commit(cur ^ 1);
        __syncthreads();
    }
    if (my_k < S) {
        store_rows_T(dk, a.dK + (tok0 + my_k) * a.sdk + (long)kvh * D, a.scale, lane);
        store_rows_T(dv, a.dV + (tok0 + my_k) * a.sdv + (long)kvh * D, 1.f, lane);
    }
}

int check_common(const AttnArgs& a, int head_dim, const char* who) {
    MGEN_REQUIRE(head_dim == D, "%s: only head_dim 128 is built (got %d)", who, head_dim);
    MGEN_REQUIRE(a.B > 0 && a.S > 0 && a.H > 0 && a.KVH > 0 && a.H % a.KVH == 0, "%s: bad sizes", who);
    MGEN_REQUIRE(a.sq % 8 == 0 && a.sk % 8 == 0 && a.sv % 8 == 0, "%s: token strides must be multiples of 8", who);
    return MGEN_OK;
}

}  // namespace

extern "C" int mgen_attn_fwd(const void* q, const void* k, const void* v, void* out, float* lse, long sq, long sk,
                             long sv, long so, int B, int S, int H, int KVH, int head_dim, float scale,
                             hipStream_t stream) {
    MGEN_REQUIRE(q && k && v && out, "mgen_attn_fwd: null pointer");
    AttnArgs a = {};
    a.Q = (const bf16*)q; a.K = (const bf16*)k; a.V = (const bf16*)v; a.Out = (bf16*)out; a.LSE = lse;
    a.sq = sq; a.sk = sk; a.sv = sv; a.so = so; a.B = B; a.S = S; a.H = H; a.KVH = KVH; a.scale = scale;
    if (int e = check_common(a, head_dim, "mgen_attn_fwd")) return e;
    MGEN_REQUIRE(so % 4 == 0, "mgen_attn_fwd: output token stride must be a multiple of 4");
    const dim3 grid((S + 127) / 128, H, B), block(256);
    hipLaunchKernelGGL(attn_fwd_kernel, grid, block, 4 * TILE, stream, a);
    return mgen_check_launch("mgen_attn_fwd");
}

extern "C" int mgen_attn_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout,
                             const float* lse, float* delta, void* dq, void* dk, void* dv, long sq, long sk, long sv,
                             long so, long sdo, long sdq, long sdk, long sdv, int B, int S, int H, int KVH,
                             int head_dim, float scale, hipStream_t stream) {
    MGEN_REQUIRE(q && k && v && out && dout && lse && delta && dq && dk && dv, "mgen_attn_bwd: null pointer");
    AttnArgs a = {};
    a.Q = (const bf16*)q; a.K = (const bf16*)k; a.V = (const bf16*)v; a.O = (const bf16*)out; a.dO = (const bf16*)dout;
    a.dQ = (bf16*)dq; a.dK = (bf16*)dk; a.dV = (bf16*)dv; a.LSE = (float*)lse; a.Delta = delta;
    a.sq = sq; a.sk = sk; a.sv = sv; a.so = so; a.sdo = sdo; a.sdq = sdq; a.sdk = sdk; a.sdv = sdv;
    a.B = B; a.S = S; a.H = H; a.KVH = KVH; a.scale = scale;
    if (int e = check_common(a, head_dim, "mgen_attn_bwd")) return e;
    MGEN_REQUIRE(so % 8 == 0 && sdo % 8 == 0 && sdq % 4 == 0 && sdk % 4 == 0 && sdv % 4 == 0,
                 "mgen_attn_bwd: bad token stride alignment");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * DKV_BUF);
        if (e != hipSuccess) {
            mgen_set_error("mgen_attn_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return MGEN_ELAUNCH;
        }
        attr_set = true;
    }
    const long nrows = (long)B * S * H;
    long nb = (nrows + 15) / 16;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)nb), dim3(256), 0, stream, a.O, a.dO, so, sdo, delta, B, S, H);
    const dim3 block(256);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3((S + 127) / 128, KVH, B), block, 2 * DKV_BUF, stream, a);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((S + 127) / 128, H, B), block, 4 * TILE, stream, a);
    return mgen_check_launch("mgen_attn_bwd");
}

// Diagnostic: resident workgroups per CU the runtime predicts for the attention kernels (0 fwd, 1 dq, 2 dkv).
extern "C" int mgen_debug_attn_occupancy(int which) {
    int n = -1;
    hipError_t e;
    if (which == 0) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_fwd_kernel, 256, 4 * TILE);
    else if (which == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_bwd_dq_kernel, 256, 4 * TILE);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_bwd_dkv_kernel, 256, 2 * DKV_BUF);
    return e == hipSuccess ? n : -(int)e;
}
