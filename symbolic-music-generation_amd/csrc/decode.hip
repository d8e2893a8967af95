// KV-cache decode kernels: cache append and single-query attention over the cache (HBM-bound).
//
// Reference: DynamicLayer.update (transformers cache_utils.py:127-146: torch.cat of the new K/V onto the cache, every
// layer, every token) followed by the attention call with q_len = 1 (models/qwen3/modeling_qwen3.py:259-279). Here the
// cache is pre-allocated once, head-major [B][KVH][max_len][D] bf16, so a stream's keys are one contiguous run of
// 2*D-byte rows and every decode step reads each cached element exactly once.
//
// Attention: one workgroup per (stream, kv head, key split). A wave takes 4 key rows per load instruction
// (16 lanes x 16 bytes per row), computes the 4 scores with a 16-lane reduction, and keeps one online-softmax state
// (m, l, 8 output columns per lane) per 16-lane group; groups, waves and key splits are merged at the end
// (the splits by a second small kernel). K and V go straight from HBM to registers: each element is used once, so LDS
// staging would only add a round trip.
#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float NEG_BIG = -1.0e30f;

// src rows: k_src/v_src [B*n_new tokens][ld] (token-major, heads at h*D); cache [B][KVH][max_len][D]
template <int D>
__global__ __launch_bounds__(256) void kv_append_kernel(const bf16* __restrict__ k_src, const bf16* __restrict__ v_src,
                                                        long ld_k, long ld_v, bf16* __restrict__ kc,
                                                        bf16* __restrict__ vc, const int* __restrict__ pos0, int B,
                                                        int n_new, int KVH, int max_len) {
    constexpr int LPR = D / 8;
    const long nrows = (long)B * n_new * KVH;
    const int sub = threadIdx.x % LPR;
    for (long row = (long)blockIdx.x * (256 / LPR) + threadIdx.x / LPR; row < nrows;
         row += (long)gridDim.x * (256 / LPR)) {
        const int h = (int)(row % KVH);
        const long tok = row / KVH;
        const int b = (int)(tok / n_new), i = (int)(tok % n_new);
        const int p = pos0[b] + i;
        if (p >= max_len) continue;
        const long dst = (((long)b * KVH + h) * max_len + p) * D + sub * 8;
        *reinterpret_cast<u32x4*>(kc + dst) = *reinterpret_cast<const u32x4*>(k_src + tok * ld_k + (long)h * D + sub * 8);
        *reinterpret_cast<u32x4*>(vc + dst) = *reinterpret_cast<const u32x4*>(v_src + tok * ld_v + (long)h * D + sub * 8);
    }
}

// One decode step's q/k/v rows (one token per stream): q heads get the per-head RMSNorm + rotation and go to qout,
// k heads get them and go straight into the key cache at the stream's position, v heads are copied into the value
// cache - what mgen_qknorm_rope_fwd followed by mgen_kv_append compute, in one launch (the arithmetic is the same,
// expression by expression, so the two routes give the same bits).
// D/16 lanes per (stream, head) row; a lane owns two groups of 8 elements so that both members of every rotation pair
// are in one lane (split-half: 8*sub and D/2 + 8*sub; adjacent pairs: 16*sub and 16*sub + 8).
template <int D, bool IL>
__global__ __launch_bounds__(256) void decode_qkv_prep_kernel(const bf16* __restrict__ qkv, long ld_in,
                                                              bf16* __restrict__ qout, long ld_q,
                                                              bf16* __restrict__ kc, bf16* __restrict__ vc,
                                                              const float* __restrict__ wq,
                                                              const float* __restrict__ wk,
                                                              const float* __restrict__ cs,
                                                              const int* __restrict__ pos, int B, int HQ, int HK,
                                                              int max_len, float eps) {
    constexpr int LPR = D / 16;
    const int NH = HQ + 2 * HK;
    const long row = (long)blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
    if (row >= (long)B * NH) return;          // the LPR lanes of a row leave together
    const int sub = threadIdx.x % LPR;
    const int b = (int)(row / NH), hh = (int)(row % NH);
    const int o1 = IL ? 16 * sub : 8 * sub;
    const int o2 = IL ? 16 * sub + 8 : D / 2 + 8 * sub;
    const int p = pos[b];
    const bf16* src = qkv + (long)b * ld_in + (long)hh * D;
    const bf16x8 r1 = *reinterpret_cast<const bf16x8*>(src + o1);
    const bf16x8 r2 = *reinterpret_cast<const bf16x8*>(src + o2);
    if (hh >= HQ + HK) {                      // value head: copy
        if (p < max_len) {
            bf16* dst = vc + (((long)b * HK + (hh - HQ - HK)) * max_len + p) * D;
            *reinterpret_cast<bf16x8*>(dst + o1) = r1;
            *reinterpret_cast<bf16x8*>(dst + o2) = r2;
        }
        return;
    }
    const bool isq = hh < HQ;
    float a[16], o[16], c[8], sn[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (float)r1[e]; a[8 + e] = (float)r2[e]; }
    {
        const float* cp = cs + ((long)p * (D / 2) + 8 * sub) * 2;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(cp + 2 * e);
            c[e] = v[0]; sn[e] = v[1]; c[e + 1] = v[2]; sn[e + 1] = v[3];
        }
    }
    if (wq != nullptr) {
        const float* g = isq ? wq : wk;
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) ss += a[e] * a[e];
        ss = group_sum<LPR>(ss);
        const float rs = rsqrtf(ss / (float)D + eps);
#pragma unroll
        for (int e = 0; e < 16; ++e) a[e] = g[e < 8 ? o1 + e : o2 + e - 8] * round_bf(a[e] * rs);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int pe = IL ? (e ^ 1) : (e ^ 8);
        const int ci = IL ? (e >> 1) : (e & 7);
        const bool second = IL ? (e & 1) : (e >= 8);
        o[e] = second ? (a[e] * c[ci] + a[pe] * sn[ci]) : (a[e] * c[ci] - a[pe] * sn[ci]);
    }
    bf16x8 w1, w2;
#pragma unroll
    for (int e = 0; e < 8; ++e) { w1[e] = (bf16)o[e]; w2[e] = (bf16)o[8 + e]; }
    bf16* dst;
    if (isq) dst = qout + (long)b * ld_q + (long)hh * D;
    else {
        if (p >= max_len) return;
        dst = kc + (((long)b * HK + (hh - HQ)) * max_len + p) * D;
    }
    *reinterpret_cast<bf16x8*>(dst + o1) = w1;
    *reinterpret_cast<bf16x8*>(dst + o2) = w2;
}

// partial results per (b, h, split): m, l, o[D] in fp32
// The row of head `hh` of a projection output, normalised (g != null) and rotated: the arithmetic of
// decode_qkv_prep_kernel, lane for lane (D/16 lanes, two groups of 8 elements each); the bf16 results go to dst[D].
template <int D, bool IL>
__device__ __forceinline__ void prep_row(const bf16* __restrict__ src, const float* __restrict__ g,
                                         const float* __restrict__ cs_row, float eps, int sub, bf16* dst) {
    constexpr int LPR = D / 16;
    const int o1 = IL ? 16 * sub : 8 * sub;
    const int o2 = IL ? 16 * sub + 8 : D / 2 + 8 * sub;
    const bf16x8 r1 = *reinterpret_cast<const bf16x8*>(src + o1);
    const bf16x8 r2 = *reinterpret_cast<const bf16x8*>(src + o2);
    float a[16], o[16], c[8], sn[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (float)r1[e]; a[8 + e] = (float)r2[e]; }
    const float* cp = cs_row + 16 * sub;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(cp + 2 * e);
        c[e] = v[0]; sn[e] = v[1]; c[e + 1] = v[2]; sn[e + 1] = v[3];
    }
    if (g != nullptr) {
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) ss += a[e] * a[e];
        ss = group_sum<LPR>(ss);
        const float rs = rsqrtf(ss / (float)D + eps);
#pragma unroll
        for (int e = 0; e < 16; ++e) a[e] = g[e < 8 ? o1 + e : o2 + e - 8] * round_bf(a[e] * rs);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int pe = IL ? (e ^ 1) : (e ^ 8);
        const int ci = IL ? (e >> 1) : (e & 7);
        const bool second = IL ? (e & 1) : (e >= 8);
        o[e] = second ? (a[e] * c[ci] + a[pe] * sn[ci]) : (a[e] * c[ci] - a[pe] * sn[ci]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { dst[o1 + e] = (bf16)o[e]; dst[o2 + e] = (bf16)o[8 + e]; }
}

// What the kernel needs to take the step's own token from the projection output instead of from prepared buffers
// (PREP instances): raw qkv rows, the norm gains, the rotary table.
struct DecodePrep {
    const bf16* qkv; long ld_in;
    const float* wq; const float* wk; const float* cs;
    const int* pos; float eps;
};

// PREP = 0: q is ready in `q`, the caches hold all kv_len[b] keys. PREP = 1 (split-half pairs) / 2 (adjacent pairs):
// q, the new key and the new value come from the raw projection rows in `pp` (per-head norm + rotation done here, by
// the lanes and in the order decode_qkv_prep_kernel uses, so both routes give the same bits); the caches hold
// kv_len[b] - 1 keys, the new key joins the softmax in the merge step, and the first query head of each kv head (in
// the last key split) writes the new key and value rows into the caches for the steps to come.
#ifndef MGEN_AD_OCC
#define MGEN_AD_OCC 5     // workgroups per CU the register budget is set for
#endif
#ifndef MGEN_AD_U
#define MGEN_AD_U 4       // load instructions in flight per tensor and wave
#endif
template <int D, int PREP>
__global__ __launch_bounds__(256, MGEN_AD_OCC) void attn_decode_kernel(const bf16* __restrict__ q, long ld_q,
                                                          bf16* __restrict__ kc, bf16* __restrict__ vc,
                                                          const int* __restrict__ kv_len, int H, int KVH, int max_len,
                                                          int nsplit, float scale, float* __restrict__ part_ml,
                                                          float* __restrict__ part_o, bf16* __restrict__ out,
                                                          long ld_o, DecodePrep pp) {
    constexpr int LPR = D / 8;        // lanes per key row
    constexpr int RPI = 64 / LPR;     // key rows per wave load instruction
    constexpr int U = MGEN_AD_U;      // load instructions in flight per tensor
    constexpr int NSLOT = 4 * RPI + (PREP ? 1 : 0);   // partial softmax states merged at the end (+ the new key)
    __shared__ float sm_ml[NSLOT * 2];
    __shared__ float sm_o[NSLOT * D];
    __shared__ __attribute__((aligned(16))) bf16 sm_new[PREP ? 3 * D : 8];   // q, new k, new v of this (stream, head)

    const int split = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int kvh = h / (H / KVH);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = lane / LPR, sub = lane % LPR;
    const int len_all = kv_len[b];
    const int len = PREP ? len_all - 1 : len_all;     // keys read from the cache
    const int per = (len_all + nsplit - 1) / nsplit;
    const int k_beg = split * per, k_end = min(len, k_beg + per);
    const bool owns_new = PREP && split == nsplit - 1;

    if (PREP) {
        // three groups of D/16 lanes of wave 0: q row, key row, value row
        constexpr int L16 = D / 16;
        if (tid < 3 * L16) {
            const int which = tid / L16, s16 = tid % L16;
            const int p = pp.pos[b];
            const bf16* row = pp.qkv + (long)b * pp.ld_in;
            const float* csr = pp.cs + (long)p * D;
            if (which == 0) prep_row<D, PREP == 2>(row + (long)h * D, pp.wq, csr, pp.eps, s16, sm_new);
            else if (which == 1) prep_row<D, PREP == 2>(row + (long)(H + kvh) * D, pp.wk, csr, pp.eps, s16, sm_new + D);
            else {
                const bf16* src = row + (long)(H + KVH + kvh) * D + 16 * s16;
                *reinterpret_cast<bf16x8*>(sm_new + 2 * D + 16 * s16) = *reinterpret_cast<const bf16x8*>(src);
                *reinterpret_cast<bf16x8*>(sm_new + 2 * D + 16 * s16 + 8) = *reinterpret_cast<const bf16x8*>(src + 8);
            }
        }
        __syncthreads();
        if (owns_new && h == kvh * (H / KVH) && tid < 2 * LPR) {   // one writer per (stream, kv head)
            const int p = pp.pos[b];
            if (p < max_len) {
                bf16* dst = (tid < LPR ? kc : vc) + (((long)b * KVH + kvh) * max_len + p) * D + (tid % LPR) * 8;
                *reinterpret_cast<bf16x8*>(dst) =
                    *reinterpret_cast<const bf16x8*>(sm_new + (tid < LPR ? D : 2 * D) + (tid % LPR) * 8);
            }
        }
    }

    float qv[8];
    {
        const bf16x8 v = PREP ? *reinterpret_cast<const bf16x8*>(sm_new + sub * 8)
                              : *reinterpret_cast<const bf16x8*>(q + (long)b * ld_q + (long)h * D + sub * 8);
        const float sl2 = scale * LOG2E;
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[e] = (float)v[e] * sl2;
    }
    const bf16* kb = kc + ((long)b * KVH + kvh) * max_len * D + sub * 8;
    const bf16* vb = vc + ((long)b * KVH + kvh) * max_len * D + sub * 8;

    float m = NEG_BIG, l = 0.f, acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;

    // wave w takes rows k_beg + (it*4 + w)*RPI*U + u*RPI + grp
    for (int base = k_beg + wave * RPI * U; base < k_end; base += 4 * RPI * U) {
        bf16x8 kk[U], vv[U];
        float s[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int key = base + u * RPI + grp;
            const int ck = min(key, k_end - 1);
            kk[u] = ld_stream(reinterpret_cast<const bf16x8*>(kb + (long)ck * D));
            vv[u] = ld_stream(reinterpret_cast<const bf16x8*>(vb + (long)ck * D));
        }
        float mx = m;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) d += qv[e] * (float)kk[u][e];
            d = group_sum<LPR>(d);
            const int key = base + u * RPI + grp;
            s[u] = key < k_end ? d : NEG_BIG;
            mx = fmaxf(mx, s[u]);
        }
        const float alpha = __builtin_amdgcn_exp2f(m - mx);
        m = mx;
        l *= alpha;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] *= alpha;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // a masked row must contribute nothing even while the running max is still the initial NEG_BIG
            const float p = s[u] > 0.5f * NEG_BIG ? __builtin_amdgcn_exp2f(s[u] - mx) : 0.f;
            l += p;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += p * (float)vv[u][e];
        }
    }
    // merge the RPI groups of the wave, then the 4 waves, through LDS
    const int slot = wave * RPI + grp;
    if (sub == 0) {
        sm_ml[slot * 2] = m;
        sm_ml[slot * 2 + 1] = l;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) sm_o[slot * D + sub * 8 + e] = acc[e];
    if (PREP && wave == 0 && grp == 0) {
        // the step's own key: score against the rotated q, weight 1 at its own maximum, output = its value row
        // (an empty state in the splits that do not own it)
        const bf16x8 kn = *reinterpret_cast<const bf16x8*>(sm_new + D + sub * 8);
        const bf16x8 vn = *reinterpret_cast<const bf16x8*>(sm_new + 2 * D + sub * 8);
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) d += qv[e] * (float)kn[e];
        d = group_sum<LPR>(d);
        if (sub == 0) {
            sm_ml[4 * RPI * 2] = owns_new ? d : NEG_BIG;
            sm_ml[4 * RPI * 2 + 1] = owns_new ? 1.f : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) sm_o[4 * RPI * D + sub * 8 + e] = owns_new ? (float)vn[e] : 0.f;
    }
    __syncthreads();
    if (tid < D) {
        float M = NEG_BIG;
        for (int sidx = 0; sidx < NSLOT; ++sidx) M = fmaxf(M, sm_ml[sidx * 2]);
        float L = 0.f, o = 0.f;
        for (int sidx = 0; sidx < NSLOT; ++sidx) {
            const float w = __builtin_amdgcn_exp2f(sm_ml[sidx * 2] - M);
            L += sm_ml[sidx * 2 + 1] * w;
            o += sm_o[sidx * D + tid] * w;
        }
        if (nsplit == 1) {
            out[(long)b * ld_o + (long)h * D + tid] = (bf16)(o / L);
        } else {
            const long pi = ((long)b * H + h) * nsplit + split;
            part_o[pi * D + tid] = o;
            if (tid == 0) {
                part_ml[pi * 2] = M;
                part_ml[pi * 2 + 1] = L;
            }
        }
    }
}

template <int D>
__global__ void attn_decode_merge_kernel(const float* __restrict__ part_ml, const float* __restrict__ part_o,
                                         int nsplit, int H, bf16* __restrict__ out, long ld_o) {
    const int h = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    const long p0 = ((long)b * H + h) * nsplit;
    float M = NEG_BIG;
    for (int s = 0; s < nsplit; ++s) M = fmaxf(M, part_ml[(p0 + s) * 2]);
    float L = 0.f, o = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float w = __builtin_amdgcn_exp2f(part_ml[(p0 + s) * 2] - M);
        L += part_ml[(p0 + s) * 2 + 1] * w;
        o += part_o[(p0 + s) * D + t] * w;
    }
    out[(long)b * ld_o + (long)h * D + t] = (bf16)(o / L);
}

}  // namespace

extern "C" int mgen_kv_append(const void* k_src, const void* v_src, long ld_k, long ld_v, void* k_cache, void* v_cache,
                              const int* pos0, int B, int n_new, int KVH, int D, int max_len, hipStream_t stream) {
    MGEN_REQUIRE(k_src && v_src && k_cache && v_cache && pos0, "mgen_kv_append: null pointer");
    MGEN_REQUIRE(B > 0 && n_new > 0 && KVH > 0 && max_len > 0 && (D == 64 || D == 128), "mgen_kv_append: bad sizes");
    MGEN_REQUIRE(ld_k % 8 == 0 && ld_v % 8 == 0, "mgen_kv_append: strides must be multiples of 8");
    const long nrows = (long)B * n_new * KVH;
    const int rpb = 256 / (D / 8);
    long nb = (nrows + rpb - 1) / rpb;
    if (nb > 8192) nb = 8192;
    if (D == 128)
        hipLaunchKernelGGL((kv_append_kernel<128>), dim3((unsigned)nb), dim3(256), 0, stream, (const bf16*)k_src,
                           (const bf16*)v_src, ld_k, ld_v, (bf16*)k_cache, (bf16*)v_cache, pos0, B, n_new, KVH, max_len);
    else
        hipLaunchKernelGGL((kv_append_kernel<64>), dim3((unsigned)nb), dim3(256), 0, stream, (const bf16*)k_src,
                           (const bf16*)v_src, ld_k, ld_v, (bf16*)k_cache, (bf16*)v_cache, pos0, B, n_new, KVH, max_len);
    return mgen_check_launch("mgen_kv_append");
}

// qkv [B][ld_in] bf16 = one new token per stream (HQ query heads, then HK key heads, then HK value heads);
// pos[b] = the token's position (rotary table row and cache slot); caches [B][HK][max_len][D]
extern "C" int mgen_decode_qkv_prep(const void* qkv, long ld_in, void* qout, long ld_q, void* k_cache, void* v_cache,
                                    const float* wq, const float* wk, const float* cos_sin, const int* pos, int B,
                                    int HQ, int HK, int D, int max_len, float eps, int interleaved,
                                    hipStream_t stream) {
    MGEN_REQUIRE(qkv && qout && k_cache && v_cache && cos_sin && pos, "mgen_decode_qkv_prep: null pointer");
    MGEN_REQUIRE((wq == nullptr) == (wk == nullptr), "mgen_decode_qkv_prep: wq and wk must both be given or both null");
    MGEN_REQUIRE(B > 0 && HQ > 0 && HK > 0 && max_len > 0 && (D == 64 || D == 128), "mgen_decode_qkv_prep: bad sizes");
    MGEN_REQUIRE(ld_in % 8 == 0 && ld_q % 8 == 0, "mgen_decode_qkv_prep: strides must be multiples of 8");
    const long rows = (long)B * (HQ + 2 * HK);
    const int rpb = 256 / (D / 16);
    const dim3 grid((unsigned)((rows + rpb - 1) / rpb)), block(256);
#define PREP(DD, ILV)                                                                                               \
    hipLaunchKernelGGL((decode_qkv_prep_kernel<DD, ILV>), grid, block, 0, stream, (const bf16*)qkv, ld_in,          \
                       (bf16*)qout, ld_q, (bf16*)k_cache, (bf16*)v_cache, wq, wk, cos_sin, pos, B, HQ, HK, max_len, eps)
    if (D == 128) { if (interleaved) PREP(128, true); else PREP(128, false); }
    else { if (interleaved) PREP(64, true); else PREP(64, false); }
#undef PREP
    return mgen_check_launch("mgen_decode_qkv_prep");
}

// workspace floats needed by mgen_attn_decode for a given launch shape
extern "C" long mgen_attn_decode_workspace(int B, int H, int D, int nsplit) {
    return nsplit <= 1 ? 0 : (long)B * H * nsplit * (D + 2);
}

template <int PREP>
static int attn_decode_launch(const void* q, long ld_q, void* k_cache, void* v_cache, const int* kv_len, void* out,
                              long ld_o, float* workspace, int B, int H, int KVH, int D, int max_len, int nsplit,
                              float scale, DecodePrep pp, hipStream_t stream, const char* what) {
    float* part_ml = workspace;
    float* part_o = workspace ? workspace + (long)B * H * nsplit * 2 : nullptr;
    const dim3 grid(nsplit, H, B), block(256);
    if (D == 128) {
        hipLaunchKernelGGL((attn_decode_kernel<128, PREP>), grid, block, 0, stream, (const bf16*)q, ld_q, (bf16*)k_cache,
                           (bf16*)v_cache, kv_len, H, KVH, max_len, nsplit, scale, part_ml, part_o, (bf16*)out, ld_o, pp);
        if (nsplit > 1)
            hipLaunchKernelGGL((attn_decode_merge_kernel<128>), dim3(H, B), dim3(128), 0, stream, part_ml, part_o, nsplit,
                               H, (bf16*)out, ld_o);
    } else {
        hipLaunchKernelGGL((attn_decode_kernel<64, PREP>), grid, block, 0, stream, (const bf16*)q, ld_q, (bf16*)k_cache,
                           (bf16*)v_cache, kv_len, H, KVH, max_len, nsplit, scale, part_ml, part_o, (bf16*)out, ld_o, pp);
        if (nsplit > 1)
            hipLaunchKernelGGL((attn_decode_merge_kernel<64>), dim3(H, B), dim3(64), 0, stream, part_ml, part_o, nsplit,
                               H, (bf16*)out, ld_o);
    }
    return mgen_check_launch(what);
}

// q [B][ld_q] (heads at h*D), out [B][ld_o]; kv_len[b] = number of cached keys of stream b (including the new one)
extern "C" int mgen_attn_decode(const void* q, long ld_q, const void* k_cache, const void* v_cache, const int* kv_len,
                                void* out, long ld_o, float* workspace, int B, int H, int KVH, int D, int max_len,
                                int nsplit, float scale, hipStream_t stream) {
    MGEN_REQUIRE(q && k_cache && v_cache && kv_len && out, "mgen_attn_decode: null pointer");
    MGEN_REQUIRE(B > 0 && H > 0 && KVH > 0 && H % KVH == 0 && (D == 64 || D == 128) && nsplit >= 1 && nsplit <= 64,
                 "mgen_attn_decode: bad sizes");
    MGEN_REQUIRE(nsplit == 1 || workspace != nullptr, "mgen_attn_decode: nsplit > 1 needs a workspace");
    MGEN_REQUIRE(ld_q % 8 == 0, "mgen_attn_decode: ld_q must be a multiple of 8");
    return attn_decode_launch<0>(q, ld_q, const_cast<void*>(k_cache), const_cast<void*>(v_cache), kv_len, out, ld_o,
                                 workspace, B, H, KVH, D, max_len, nsplit, scale, DecodePrep{}, stream,
                                 "mgen_attn_decode");
}

// The decode step's attention straight from the projection output: qkv [B][ld_in] bf16 holds one new token per
// stream (H query heads, KVH key heads, KVH value heads). Per-head RMSNorm (wq/wk, may both be null) and rotation by
// table row pos[b] of q and of the new key happen inside the kernel; the new key and value take part in the softmax
// and are written to cache slot pos[b]. kv_len[b] counts the new token (= pos[b] + 1 in a running stream). The same
// bits as mgen_decode_qkv_prep followed by mgen_attn_decode, one launch less.
extern "C" int mgen_attn_decode_prep(const void* qkv, long ld_in, const float* wq, const float* wk,
                                     const float* cos_sin, const int* pos, float eps, int interleaved, void* k_cache,
                                     void* v_cache, const int* kv_len, void* out, long ld_o, float* workspace, int B,
                                     int H, int KVH, int D, int max_len, int nsplit, float scale, hipStream_t stream) {
    MGEN_REQUIRE(qkv && cos_sin && pos && k_cache && v_cache && kv_len && out, "mgen_attn_decode_prep: null pointer");
    MGEN_REQUIRE((wq == nullptr) == (wk == nullptr), "mgen_attn_decode_prep: wq and wk must both be given or both null");
    MGEN_REQUIRE(B > 0 && H > 0 && KVH > 0 && H % KVH == 0 && (D == 64 || D == 128) && nsplit >= 1 && nsplit <= 64,
                 "mgen_attn_decode_prep: bad sizes");
    MGEN_REQUIRE(nsplit == 1 || workspace != nullptr, "mgen_attn_decode_prep: nsplit > 1 needs a workspace");
    MGEN_REQUIRE(ld_in % 8 == 0, "mgen_attn_decode_prep: ld_in must be a multiple of 8");
    DecodePrep pp{(const bf16*)qkv, ld_in, wq, wk, cos_sin, pos, eps};
    if (interleaved)
        return attn_decode_launch<2>(nullptr, 0, k_cache, v_cache, kv_len, out, ld_o, workspace, B, H, KVH, D, max_len,
                                     nsplit, scale, pp, stream, "mgen_attn_decode_prep");
    return attn_decode_launch<1>(nullptr, 0, k_cache, v_cache, kv_len, out, ld_o, workspace, B, H, KVH, D, max_len,
                                 nsplit, scale, pp, stream, "mgen_attn_decode_prep");
}
