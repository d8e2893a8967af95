// Causal self-attention on bf16 MFMA for gfx950, head_dim 128 or 64: forward (online softmax, saves LSE), and
// backward as two kernels (dK/dV per key block, dQ per query block) that recompute P from Q, K and LSE.
//
// Reference: the attention call in Qwen3Attention.forward — transformers integrations/sdpa_attention.py:154-163
// (is_causal, scale = head_dim^-0.5) and its eager twin models/qwen3/modeling_qwen3.py:185-207 (softmax in fp32) —
// plus its autograd backward. Operands are the bf16-rounded q, k (after q/k-norm + RoPE) and v.
//
// Layout: q/k/v/o/dq/dk/dv are token-major [B*S, heads, head_dim] views with an explicit token stride (so q, k, v
// may live inside one fused projection output). LSE and delta are [B, H, S] fp32.
//
// Structure shared by the three kernels: 256 threads = 4 waves; each wave owns 32 rows (queries in fwd/dQ, keys in
// dK/dV) and keeps its own operand rows in registers as MFMA B-fragments; the other side is streamed in 64-row
// tiles of 256 bytes per row through two LDS buffers by LDS-DMA (next tile issued before the current tile's MFMAs,
// one barrier per tile). With head_dim 128 a tile row is one head and a block covers 128 rows of one head; with
// head_dim 64 a tile row is a PAIR of adjacent heads, waves 0-1 work on the first head of the pair and waves 2-3 on
// the second, and a block covers 64 rows of both. Scores are computed transposed where that puts the softmax row
// on the lane (S^T = K Q^T in fwd/dQ), so row max / row sum need one cross-half exchange only, and the fp32 score
// tile is re-used in place as the bf16 operand of the next product (no LDS round trip for P or dS).
//
// This file is compiled twice: attn.hip (MGEN_ATTN_DROP 0: mgen_attn_fwd / mgen_attn_bwd) and attn_drop.hip
// (MGEN_ATTN_DROP 1: the same kernels with dropout on the probabilities, mgen_attn_{fwd,bwd}_dropout). Instances of
// one kernel template that are compiled together perturb each other's register allocation; separate translation
// units keep the plain kernels' code what it is without the dropout variants.
#include "common.h"
#include <type_traits>

#ifndef MGEN_ATTN_DROP
#error "include through attn.hip or attn_drop.hip"
#endif

namespace {

constexpr int ROWB = 256;            // bytes per LDS tile row (one head of 128, or two heads of 64)
constexpr int TILE = 64 * ROWB;      // bytes of one 64-row tile
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
    int gx, gy, gz;  // logical grid: row blocks, head (pair) index, batch; launched as one dimension
    // dropout on the attention probabilities (kernels instantiated with DROP): element (b*H + h, q, key) is kept iff
    // drop_fmix32(drop_rowkey(drop_key, b*H + h, q) ^ key) >= drop_thresh; kept probabilities count 1 / (1 - p)
    unsigned long long drop_key;
    unsigned drop_thresh;
    float drop_inv_keep;
};

// Logical workgroup coordinates from the 1-D launch index. Hardware deals consecutive workgroups round-robin to the
// 8 XCDs; the remap gives each XCD a contiguous range of logical ids, so all row blocks of one (batch, head) run on
// the same XCD at about the same time and share its L2 for the K/V (or Q/dO) tiles they all stream.
struct Coords { int x, y, z; };
__device__ __forceinline__ Coords block_coords(const AttnArgs& a) {
    const int L = xcd_remap(blockIdx.x, a.gx * a.gy * a.gz);
    Coords c;
    c.x = L % a.gx;
    c.y = (L / a.gx) % a.gy;
    c.z = L / (a.gx * a.gy);
    return c;
}

template <int HD> struct Geo {
    static constexpr int KS = HD / 16;                  // MFMA k-steps over the head dimension
    static constexpr int DB = HD / 32;                  // 32-wide output blocks over the head dimension
    static constexpr int ROWS = HD == 128 ? 128 : 64;   // rows (queries or keys) per workgroup
    static constexpr int HPB = HD == 128 ? 1 : 2;       // heads per workgroup
};

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// 64 rows x 256 bytes straight into the LDS image by LDS-DMA (no staging registers, no ds_write): a wave
// instruction fills one contiguous 1 KiB piece (4 rows), so the image's chunk swizzle goes on the source address.
// Rows are clamped to S-1 (their results are masked or never stored).
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
template <int KS>
__device__ __forceinline__ void load_own_rows(bf16x8 (&f)[KS], const bf16* __restrict__ base, long tok_stride,
                                              long tok0, int row0, int S, int lane) {
    const int gr = min(row0 + (lane & 31), S - 1);
    const bf16* p = base + (tok0 + gr) * tok_stride + 8 * (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) f[ks] = *reinterpret_cast<const bf16x8*>(p + 16 * ks);
}

// The register-resident operand rows are loaded once, before the main loop. Drain those loads here and pass the
// registers through an empty asm so that the compiler's wait-count bookkeeping does not tie the loop's MFMAs to
// vector-memory operations issued inside the loop.
template <int KS>
__device__ __forceinline__ void settle_frags(bf16x8 (&f)[KS]) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < KS; ++i) asm volatile("" : "+v"(f[i]));
}

__device__ __forceinline__ void zero16(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// An accumulation chain of KS MFMAs whose result lands in VGPRs. In a kernel that may use more than 256 registers the
// compiler puts every MFMA result into the accumulator half of the register file, so a result that vector ALU code
// consumes (scores, probabilities) costs one copy per register on the way out; written as asm the chain's destination
// is an ordinary vector register. Nothing inside an asm string is padded by the compiler: the leading s_nop covers a
// vector-ALU write of an operand just before the first MFMA, the trailing s_nop 11 the 12 wait states between the
// last MFMA's write and any reader of the result.
template <int KS>
__device__ __forceinline__ void mfma_chain_vgpr(f32x16& d, const bf16x8 (&a)[KS], const bf16x8 (&b)[KS]) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a[0]), "v"(b[0]));
#pragma unroll
    for (int ks = 1; ks < KS - 1; ++ks)
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a[ks]), "v"(b[ks]));
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 11"
                 : "+v"(d) : "v"(a[KS - 1]), "v"(b[KS - 1]));
}

// Store a transposed accumulator set acc[db] = X^T[d = 32db + acc_row(reg)][row = lane&31] as X[row][d] (bf16)
template <int DB>
__device__ __forceinline__ void store_rows_T(const f32x16 (&acc)[DB], bf16* __restrict__ dst, float mul, int lane) {
    const int hh = lane >> 5;
#pragma unroll
    for (int db = 0; db < DB; ++db)
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
template <int HD, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs a) {
    using G = Geo<HD>;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][K tile | V tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Coords bid = block_coords(a);
    const int qb = a.gx - 1 - bid.x;  // heaviest (latest) query blocks first
    const int b = bid.z;
    const int sub = HD == 128 ? 0 : (wave >> 1);             // which head of the pair (head_dim 64)
    const int wq = HD == 128 ? wave : (wave & 1);            // which 32-query slice of the block
    const int h = HD == 128 ? bid.y : 2 * bid.y + sub;
    const int kv_first = HD == 128 ? h / (a.H / a.KVH) : 2 * bid.y;  // first kv head of the tile row
    const int ks0 = 4 * sub, db0 = 2 * sub;                  // this head's k-steps / d-blocks inside the tile row
    const int S = a.S;
    const long tok0 = (long)b * S;
    const int q0 = qb * G::ROWS, qw0 = q0 + wq * 32;
    const bf16* Kb = a.K + (long)kv_first * HD;
    const bf16* Vb = a.V + (long)kv_first * HD;

    bf16x8 qf[G::KS];
    load_own_rows<G::KS>(qf, a.Q + (long)h * HD, a.sq, tok0, qw0, S, lane);

    f32x16 o[G::DB];
#pragma unroll
    for (int db = 0; db < G::DB; ++db) zero16(o[db]);
    float m = NEG_BIG, l = 0.f;
    const float sl2 = a.scale * LOG2E;
    const int my_q = qw0 + (lane & 31);
    // key = kv0 + 32t + 4*(lane>>5) + c_r with c_r = (r&3) + 8*(r>>2): the register part occupies bits the rest leaves
    // free, so the hash input is (rowkey ^ lane part) ^ c_r
    unsigned rk = 0;
    if constexpr (DROP) rk = drop_rowkey(a.drop_key, (unsigned)(b * a.H + h), (unsigned)my_q) ^ (unsigned)(4 * (lane >> 5));

    const int last_key = min(q0 + G::ROWS - 1, S - 1);
    const int ntiles = last_key / 64 + 1;

    tile_glds(Kb, a.sk, tok0, 0, S, smem, tid);
    tile_glds(Vb, a.sv, tok0, 0, S, smem + TILE, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    settle_frags<G::KS>(qf);

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
                    for (int ks = 0; ks < G::KS; ++ks)
                        s[t] = mfma32(frag_row_img256(Kt, 32 * t, ks0 + ks, lane), qf[ks], s[t]);
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
                    for (int db = 0; db < G::DB; ++db)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
                }
                const float nm = -m * sl2;
                float psum = 0.f;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float p = fast_exp2(__builtin_fmaf(s[t][r], sl2, nm));
                        psum += p;   // the softmax normaliser counts every visible key, dropped or not
                        if constexpr (DROP) {
                            const unsigned hsh = drop_fmix32(rk ^ (unsigned)(kv0 + 32 * t) ^ (unsigned)((r & 3) + 8 * (r >> 2)));
                            if (hsh < a.drop_thresh) p = 0.f;
                        }
                        s[t][r] = p;
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
                for (int db = 0; db < G::DB; ++db)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2)
                            o[db] = mfma32(frag_col_img256<true>(Vt, 32 * t + 16 * s2, 32 * (db0 + db), lane),
                                           pf[t][s2], o[db]);
            };
            if (kv0 + 63 > qw0) tile(std::true_type{});
            else tile(std::false_type{});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    const float lt = l + __shfl_xor(l, 32, 64);
    if (my_q < S) {
        const float inv = (DROP ? a.drop_inv_keep : 1.f) / lt;
        store_rows_T<G::DB>(o, a.Out + (tok0 + my_q) * a.so + (long)h * HD, inv, lane);
        if (lane < 32 && a.LSE) a.LSE[((long)b * a.H + h) * S + my_q] = (m * sl2 + __log2f(lt)) * LN2;
    }
}

// ===================================================================================================
// backward, part 0: delta[b,h,q] = sum_d dO * O
// ===================================================================================================
template <int HD>
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16* __restrict__ O, const bf16* __restrict__ dO,
                                                         long so, long sdo, float* __restrict__ delta, int B, int S,
                                                         int H) {
    constexpr int LPR = HD / 8;
    const long nrows = (long)B * S * H;
    const int sub = threadIdx.x % LPR;
    for (long row = (long)blockIdx.x * (256 / LPR) + threadIdx.x / LPR; row < nrows;
         row += (long)gridDim.x * (256 / LPR)) {
        const long tok = row / H;
        const int h = (int)(row % H);
        const bf16x8 ov = *reinterpret_cast<const bf16x8*>(O + tok * so + (long)h * HD + sub * 8);
        const bf16x8 dv = *reinterpret_cast<const bf16x8*>(dO + tok * sdo + (long)h * HD + sub * 8);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)ov[e] * (float)dv[e];
        s = group_sum<LPR>(s);
        if (sub == 0) {
            const long b = tok / S, q = tok % S;
            delta[(b * H + h) * S + q] = s;
        }
    }
}

// ===================================================================================================
// backward, part 1: dQ. Same work decomposition as the forward.
// ===================================================================================================
template <int HD, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnArgs a) {
    using G = Geo<HD>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Coords bid = block_coords(a);
    const int qb = a.gx - 1 - bid.x;
    const int b = bid.z;
    const int sub = HD == 128 ? 0 : (wave >> 1);
    const int wq = HD == 128 ? wave : (wave & 1);
    const int h = HD == 128 ? bid.y : 2 * bid.y + sub;
    const int kv_first = HD == 128 ? h / (a.H / a.KVH) : 2 * bid.y;
    const int ks0 = 4 * sub, db0 = 2 * sub;
    const int S = a.S;
    const long tok0 = (long)b * S;
    const int q0 = qb * G::ROWS, qw0 = q0 + wq * 32;
    const bf16* Kb = a.K + (long)kv_first * HD;
    const bf16* Vb = a.V + (long)kv_first * HD;

    bf16x8 qf[G::KS], dof[G::KS];
    load_own_rows<G::KS>(qf, a.Q + (long)h * HD, a.sq, tok0, qw0, S, lane);
    load_own_rows<G::KS>(dof, a.dO + (long)h * HD, a.sdo, tok0, qw0, S, lane);
    const int my_q = qw0 + (lane & 31);
    const int cq = min(my_q, S - 1);
    const float lse2 = a.LSE[((long)b * a.H + h) * S + cq] * LOG2E;
    const float delta = a.Delta[((long)b * a.H + h) * S + cq];
    const float sl2 = a.scale * LOG2E;
    unsigned rk = 0;   // as in the forward kernel
    if constexpr (DROP) rk = drop_rowkey(a.drop_key, (unsigned)(b * a.H + h), (unsigned)my_q) ^ (unsigned)(4 * (lane >> 5));

    f32x16 dq[G::DB];
#pragma unroll
    for (int db = 0; db < G::DB; ++db) zero16(dq[db]);

    const int last_key = min(q0 + G::ROWS - 1, S - 1);
    const int ntiles = last_key / 64 + 1;

    tile_glds(Kb, a.sk, tok0, 0, S, smem, tid);
    tile_glds(Vb, a.sv, tok0, 0, S, smem + TILE, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    settle_frags<G::KS>(qf);
    settle_frags<G::KS>(dof);

    for (int j = 0; j < ntiles; ++j) {
        const int cur = j & 1;
        const int kv0 = j * 64;
        if (j + 1 < ntiles) {
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
                for (int ks = 0; ks < G::KS; ++ks)
                    s = mfma32(frag_row_img256(Kt, 32 * t, ks0 + ks, lane), qf[ks], s);
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks)
                    dp = mfma32(frag_row_img256(Vt, 32 * t, ks0 + ks, lane), dof[ks], dp);
                // key = kv0 + 32t + acc_row(r, lane) is masked when key > my_q; acc_row = c_r + 4*(lane>>5) with c_r a
                // constant of the register, so one per-lane threshold serves all 16 compares (no mask: threshold
                // beyond every c_r)
                const int thr = need_mask ? my_q - kv0 - 32 * t - 4 * (lane >> 5) : 64;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float p = fast_exp2(__builtin_fmaf(s[r], sl2, -lse2));
                    if ((r & 3) + 8 * (r >> 2) > thr) p = 0.f;
                    float dpv = dp[r];
                    if constexpr (DROP) {
                        const unsigned hsh = drop_fmix32(rk ^ (unsigned)(kv0 + 32 * t) ^ (unsigned)((r & 3) + 8 * (r >> 2)));
                        dpv = hsh < a.drop_thresh ? 0.f : dpv * a.drop_inv_keep;
                    }
                    s[r] = p * (dpv - delta);  // dS^T[key][q]
                }
                const bf16x8 f0 = acc_to_frag<0>(s), f1 = acc_to_frag<1>(s);
                // dQ^T[d][q] += K^T[d][key] * dS^T[key][q]
#pragma unroll
                for (int db = 0; db < G::DB; ++db) {
                    dq[db] = mfma32(frag_col_img256<true>(Kt, 32 * t, 32 * (db0 + db), lane), f0, dq[db]);
                    dq[db] = mfma32(frag_col_img256<true>(Kt, 32 * t + 16, 32 * (db0 + db), lane), f1, dq[db]);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (my_q < S) store_rows_T<G::DB>(dq, a.dQ + (tok0 + my_q) * a.sdq + (long)h * HD, a.scale, lane);
}

// ===================================================================================================
// backward, part 2: dK and dV. One block = ROWS keys of one kv head (head_dim 128) or of a pair of kv heads
// (head_dim 64); loops over the query heads of the group and over the query tiles at or after the block's first
// key. LDS per buffer: Q tile | dO tile | lse2[2][64] | delta[2][64]
// ===================================================================================================
// (+ rowkey[2][64] of the dropout hash, + padding to a multiple of 1 KiB)
constexpr int DKV_BUF = 2 * TILE + 2048;

template <int HD, bool DROP>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_kernel(AttnArgs a) {
    using G = Geo<HD>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Coords bid = block_coords(a);
    const int kb = bid.x, b = bid.z;
    const int sub = HD == 128 ? 0 : (wave >> 1);
    const int wk = HD == 128 ? wave : (wave & 1);
    const int kvh = HD == 128 ? bid.y : 2 * bid.y + sub;
    const int ks0 = 4 * sub, db0 = 2 * sub;
    const int S = a.S;
    const long tok0 = (long)b * S;
    const int k0 = kb * G::ROWS, kw0 = k0 + wk * 32;
    const int my_k = kw0 + (lane & 31);
    const int group = HD == 128 ? a.H / a.KVH : 1;  // head_dim 64 requires H == KVH
    const float sl2 = a.scale * LOG2E;

    bf16x8 kf[G::KS], vf[G::KS];
    load_own_rows<G::KS>(kf, a.K + (long)kvh * HD, a.sk, tok0, kw0, S, lane);
    load_own_rows<G::KS>(vf, a.V + (long)kvh * HD, a.sv, tok0, kw0, S, lane);

    f32x16 dk[G::DB], dv[G::DB];
#pragma unroll
    for (int db = 0; db < G::DB; ++db) {
        zero16(dk[db]);
        zero16(dv[db]);
    }

    const int t_first = k0 / 64;
    const int t_end = (S + 63) / 64;
    const int per_head = t_end - t_first;
    const int nsteps = per_head * group;

    float s_lse = 0.f, s_delta = 0.f;
    unsigned s_rk = 0;

    auto issue = [&](int step, int buf) {
        // first query head of the tile row for this step
        const int hq0 = HD == 128 ? bid.y * group + step / per_head : 2 * bid.y;
        const int qt0 = (t_first + step % per_head) * 64;
        char* base = smem + buf * DKV_BUF;
        tile_glds(a.Q + (long)hq0 * HD, a.sq, tok0, qt0, S, base, tid);
        tile_glds(a.dO + (long)hq0 * HD, a.sdo, tok0, qt0, S, base + TILE, tid);
        if (tid < 64 * G::HPB) {
            const int q = min(qt0 + (tid & 63), S - 1);
            const long o = ((long)b * a.H + hq0 + (tid >> 6)) * S + q;
            s_lse = a.LSE[o] * LOG2E;
            s_delta = a.Delta[o];
            if constexpr (DROP) s_rk = drop_rowkey(a.drop_key, (unsigned)(b * a.H + hq0 + (tid >> 6)), (unsigned)q);
        }
    };
    auto commit = [&](int buf) {
        char* base = smem + buf * DKV_BUF;
        if (tid < 64 * G::HPB) {
            reinterpret_cast<float*>(base + 2 * TILE)[tid] = s_lse;
            reinterpret_cast<float*>(base + 2 * TILE + 512)[tid] = s_delta;
            if constexpr (DROP) reinterpret_cast<unsigned*>(base + 2 * TILE + 1024)[tid] = s_rk;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    issue(0, 0);
    commit(0);
    __syncthreads();
    settle_frags<G::KS>(kf);
    settle_frags<G::KS>(vf);

    // One step = one 64-query tile. The body exists in a masked and an unmasked form; each form runs in a loop of its
    // own, so that the loop bodies are straight-line code and the dK/dV accumulators stay in place from step to step
    // (with both forms behind a branch in one loop the compiler moved all 128 accumulator registers every step).
    auto run_step = [&](int step, auto masked) {
        constexpr bool MASKED = decltype(masked)::value;
        const int cur = step & 1;
        if (step + 1 < nsteps) issue(step + 1, cur ^ 1);
        const int qt0 = (t_first + step % per_head) * 64;
        const char* Qt = smem + cur * DKV_BUF;
        const char* dOt = Qt + TILE;
        const float* lse_s = reinterpret_cast<const float*>(Qt + 2 * TILE) + 64 * sub;
        const float* del_s = reinterpret_cast<const float*>(Qt + 2 * TILE + 512) + 64 * sub;
        const unsigned* rk_s = reinterpret_cast<const unsigned*>(Qt + 2 * TILE + 1024) + 64 * sub;
        // all four score / dP chains first: the matrix cores then work through the second half's chains while the
        // vector ALU does the first half's exponentials
        f32x16 s[2], dp[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            bf16x8 qa[G::KS], da[G::KS];
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
                qa[ks] = frag_row_img256(Qt, 32 * u, ks0 + ks, lane);
                da[ks] = frag_row_img256(dOt, 32 * u, ks0 + ks, lane);
            }
            mfma_chain_vgpr<G::KS>(s[u], qa, kf);
            mfma_chain_vgpr<G::KS>(dp[u], da, vf);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            // column fragments of both transposed products, issued ahead of the exponentials that hide their latency
            bf16x8 dot[G::DB][2], qtf[G::DB][2];
#pragma unroll
            for (int db = 0; db < G::DB; ++db) {
                const int c0 = 32 * (db0 + db);
                dot[db][0] = frag_col_img256<true>(dOt, 32 * u, c0, lane);
                dot[db][1] = frag_col_img256<true>(dOt, 32 * u + 16, c0, lane);
                qtf[db][0] = frag_col_img256<true>(Qt, 32 * u, c0, lane);
                qtf[db][1] = frag_col_img256<true>(Qt, 32 * u + 16, c0, lane);
            }
            // s[u][reg]: query = qt0 + 32u + acc_row(reg), key = my_k
            f32x16 ds;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int qr = 32 * u + 8 * g4 + 4 * (lane >> 5);
                const f32x4 lv = *reinterpret_cast<const f32x4*>(lse_s + qr);
                const f32x4 dl = *reinterpret_cast<const f32x4*>(del_s + qr);
                u32x4 rv = {0u, 0u, 0u, 0u};
                if constexpr (DROP) rv = *reinterpret_cast<const u32x4*>(rk_s + qr);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g4 + e;
                    float p = fast_exp2(__builtin_fmaf(s[u][r], sl2, -lv[e]));
                    if (MASKED) {
                        const int q = qt0 + qr + e;
                        if (my_k > q || q >= S) p = 0.f;
                    }
                    if constexpr (DROP) {   // dV sees the dropped probabilities, dS the dropped dP
                        const bool keep = drop_fmix32(rv[e] ^ (unsigned)my_k) >= a.drop_thresh;
                        s[u][r] = keep ? p : 0.f;
                        ds[r] = p * ((keep ? dp[u][r] * a.drop_inv_keep : 0.f) - dl[e]);
                    } else {
                        s[u][r] = p;
                        ds[r] = p * (dp[u][r] - dl[e]);
                    }
                }
            }
            const bf16x8 p0 = acc_to_frag<0>(s[u]), p1 = acc_to_frag<1>(s[u]);
            const bf16x8 d0 = acc_to_frag<0>(ds), d1 = acc_to_frag<1>(ds);
#pragma unroll
            for (int db = 0; db < G::DB; ++db) {
                // dV^T[d][key] += dO^T[d][q] * P[q][key] ; dK^T[d][key] += Q^T[d][q] * dS[q][key]
                dv[db] = mfma32(dot[db][0], p0, dv[db]);
                dv[db] = mfma32(dot[db][1], p1, dv[db]);
                dk[db] = mfma32(qtf[db][0], d0, dk[db]);
                dk[db] = mfma32(qtf[db][1], d1, dk[db]);
            }
        }
        if (step + 1 < nsteps) commit(cur ^ 1);
        __syncthreads();
    };
    // per head: the tiles that touch the block's diagonal (masked), the tiles below it (every pair visible), and a
    // last tile that runs past the sequence end (masked) when S is not a multiple of 64
    const int n_diag = min(per_head, G::ROWS / 64);
    const int n_tail = ((S & 63) != 0 && per_head > n_diag) ? 1 : 0;
    int step = 0;
    for (int hh = 0; hh < group; ++hh) {
        int j = 0;
        for (; j < n_diag; ++j, ++step) run_step(step, std::true_type{});
        for (; j < per_head - n_tail; ++j, ++step) run_step(step, std::false_type{});
        for (; j < per_head; ++j, ++step) run_step(step, std::true_type{});
    }
    if (my_k < S) {
        store_rows_T<G::DB>(dk, a.dK + (tok0 + my_k) * a.sdk + (long)kvh * HD, a.scale, lane);
        store_rows_T<G::DB>(dv, a.dV + (tok0 + my_k) * a.sdv + (long)kvh * HD, DROP ? a.drop_inv_keep : 1.f, lane);
    }
}

int check_common(const AttnArgs& a, int head_dim, const char* who) {
    MGEN_REQUIRE(head_dim == 128 || head_dim == 64, "%s: head_dim must be 64 or 128 (got %d)", who, head_dim);
    MGEN_REQUIRE(a.B > 0 && a.S > 0 && a.H > 0 && a.KVH > 0 && a.H % a.KVH == 0, "%s: bad sizes", who);
    if (head_dim == 64)
        MGEN_REQUIRE(a.H == a.KVH && a.H % 2 == 0,
                     "%s: head_dim 64 runs heads in pairs: needs an even head count and H == KVH", who);
    MGEN_REQUIRE(a.sq % 8 == 0 && a.sk % 8 == 0 && a.sv % 8 == 0, "%s: token strides must be multiples of 8", who);
    return MGEN_OK;
}

template <typename K>
int set_lds(K kernel, int bytes, const char* who) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        mgen_set_error("%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e));
        return MGEN_ELAUNCH;
    }
    return MGEN_OK;
}

int set_drop(AttnArgs& a, float p, unsigned long long key, const char* who) {
    MGEN_REQUIRE(p >= 0.f && p < 1.f, "%s: dropout probability must be in [0, 1) (got %g)", who, p);
    a.drop_key = key;
    a.drop_thresh = drop_threshold(p);
    a.drop_inv_keep = 1.f / (1.f - p);
    return MGEN_OK;
}

constexpr bool kDrop = MGEN_ATTN_DROP != 0;

int attn_fwd_impl(const void* q, const void* k, const void* v, void* out, float* lse, long sq, long sk, long sv, long so,
                  int B, int S, int H, int KVH, int head_dim, float scale, float p, unsigned long long key,
                  hipStream_t stream, const char* who) {
    MGEN_REQUIRE(q && k && v && out, "%s: null pointer", who);
    AttnArgs a = {};
    a.Q = (const bf16*)q; a.K = (const bf16*)k; a.V = (const bf16*)v; a.Out = (bf16*)out; a.LSE = lse;
    a.sq = sq; a.sk = sk; a.sv = sv; a.so = so; a.B = B; a.S = S; a.H = H; a.KVH = KVH; a.scale = scale;
    if (int e = check_common(a, head_dim, who)) return e;
    if (int e = set_drop(a, p, key, who)) return e;
    MGEN_REQUIRE(so % 4 == 0, "%s: output token stride must be a multiple of 4", who);
    const dim3 block(256);
    a.gx = head_dim == 128 ? (S + 127) / 128 : (S + 63) / 64;
    a.gy = head_dim == 128 ? H : H / 2;
    a.gz = B;
    MGEN_REQUIRE((long)a.gx * a.gy * a.gz < (1L << 31), "%s: grid too large", who);
    const dim3 grid((unsigned)(a.gx * a.gy * a.gz));
    if (head_dim == 128) hipLaunchKernelGGL((attn_fwd_kernel<128, kDrop>), grid, block, 4 * TILE, stream, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<64, kDrop>), grid, block, 4 * TILE, stream, a);
    return mgen_check_launch(who);
}

int attn_bwd_impl(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
                  float* delta, void* dq, void* dk, void* dv, long sq, long sk, long sv, long so, long sdo, long sdq,
                  long sdk, long sdv, int B, int S, int H, int KVH, int head_dim, float scale, float p,
                  unsigned long long key, hipStream_t stream, const char* who) {
    MGEN_REQUIRE(q && k && v && out && dout && lse && delta && dq && dk && dv, "%s: null pointer", who);
    AttnArgs a = {};
    a.Q = (const bf16*)q; a.K = (const bf16*)k; a.V = (const bf16*)v; a.O = (const bf16*)out; a.dO = (const bf16*)dout;
    a.dQ = (bf16*)dq; a.dK = (bf16*)dk; a.dV = (bf16*)dv; a.LSE = (float*)lse; a.Delta = delta;
    a.sq = sq; a.sk = sk; a.sv = sv; a.so = so; a.sdo = sdo; a.sdq = sdq; a.sdk = sdk; a.sdv = sdv;
    a.B = B; a.S = S; a.H = H; a.KVH = KVH; a.scale = scale;
    if (int e = check_common(a, head_dim, who)) return e;
    if (int e = set_drop(a, p, key, who)) return e;
    MGEN_REQUIRE(so % 8 == 0 && sdo % 8 == 0 && sdq % 4 == 0 && sdk % 4 == 0 && sdv % 4 == 0,
                 "%s: bad token stride alignment", who);
    // more than the default 64 KiB of dynamic LDS has to be asked for, once per kernel and device
    static bool asked[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    bool& attr_set = asked[dev];
    if (!attr_set) {
        if (int e = set_lds(attn_bwd_dkv_kernel<128, kDrop>, 2 * DKV_BUF, who)) return e;
        if (int e = set_lds(attn_bwd_dkv_kernel<64, kDrop>, 2 * DKV_BUF, who)) return e;
        attr_set = true;
    }
    const long nrows = (long)B * S * H;
    const int rpb = 256 / (head_dim / 8);
    long nb = (nrows + rpb - 1) / rpb;
    if (nb > 8192) nb = 8192;
    const dim3 block(256);
    const int nrb = head_dim == 128 ? (S + 127) / 128 : (S + 63) / 64;   // row blocks
    MGEN_REQUIRE((long)nrb * H * B < (1L << 31), "%s: grid too large", who);
    AttnArgs akv = a, aq = a;   // the two kernels walk different logical grids
    akv.gx = nrb; akv.gy = head_dim == 128 ? KVH : KVH / 2; akv.gz = B;
    aq.gx = nrb; aq.gy = head_dim == 128 ? H : H / 2; aq.gz = B;
    const dim3 gkv((unsigned)(akv.gx * akv.gy * akv.gz)), gq((unsigned)(aq.gx * aq.gy * aq.gz));
    // delta = rowsum(dO * O) holds with dropout too: sum_k P dP = sum_k P~ dP~ = dO . O
    if (head_dim == 128) {
        hipLaunchKernelGGL(attn_delta_kernel<128>, dim3((unsigned)nb), block, 0, stream, a.O, a.dO, so, sdo, delta, B, S, H);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<128, kDrop>), gkv, block, 2 * DKV_BUF, stream, akv);
        hipLaunchKernelGGL((attn_bwd_dq_kernel<128, kDrop>), gq, block, 4 * TILE, stream, aq);
    } else {
        hipLaunchKernelGGL(attn_delta_kernel<64>, dim3((unsigned)nb), block, 0, stream, a.O, a.dO, so, sdo, delta, B, S, H);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<64, kDrop>), gkv, block, 2 * DKV_BUF, stream, akv);
        hipLaunchKernelGGL((attn_bwd_dq_kernel<64, kDrop>), gq, block, 4 * TILE, stream, aq);
    }
    return mgen_check_launch(who);
}

}  // namespace

#if !MGEN_ATTN_DROP
extern "C" int mgen_attn_fwd(const void* q, const void* k, const void* v, void* out, float* lse, long sq, long sk,
                             long sv, long so, int B, int S, int H, int KVH, int head_dim, float scale,
                             hipStream_t stream) {
    return attn_fwd_impl(q, k, v, out, lse, sq, sk, sv, so, B, S, H, KVH, head_dim, scale, 0.f, 0ull, stream,
                         "mgen_attn_fwd");
}

extern "C" int mgen_attn_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout,
                             const float* lse, float* delta, void* dq, void* dk, void* dv, long sq, long sk, long sv,
                             long so, long sdo, long sdq, long sdk, long sdv, int B, int S, int H, int KVH,
                             int head_dim, float scale, hipStream_t stream) {
    return attn_bwd_impl(q, k, v, out, dout, lse, delta, dq, dk, dv, sq, sk, sv, so, sdo, sdq, sdk, sdv, B, S, H, KVH,
                         head_dim, scale, 0.f, 0ull, stream, "mgen_attn_bwd");
}

#else
// The same two with dropout on the attention probabilities (probability p in [0, 1), mask from `key`; see AttnArgs).
// The backward call must get the p and key of its forward call.
extern "C" int mgen_attn_fwd_dropout(const void* q, const void* k, const void* v, void* out, float* lse, long sq,
                                     long sk, long sv, long so, int B, int S, int H, int KVH, int head_dim,
                                     float scale, float p, unsigned long long key, hipStream_t stream) {
    return attn_fwd_impl(q, k, v, out, lse, sq, sk, sv, so, B, S, H, KVH, head_dim, scale, p, key, stream,
                         "mgen_attn_fwd_dropout");
}

extern "C" int mgen_attn_bwd_dropout(const void* q, const void* k, const void* v, const void* out, const void* dout,
                                     const float* lse, float* delta, void* dq, void* dk, void* dv, long sq, long sk,
                                     long sv, long so, long sdo, long sdq, long sdk, long sdv, int B, int S, int H,
                                     int KVH, int head_dim, float scale, float p, unsigned long long key,
                                     hipStream_t stream) {
    return attn_bwd_impl(q, k, v, out, dout, lse, delta, dq, dk, dv, sq, sk, sv, so, sdo, sdq, sdk, sdv, B, S, H, KVH,
                         head_dim, scale, p, key, stream, "mgen_attn_bwd_dropout");
}
#endif
