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

// partial results per (b, h, split): m, l, o[D] in fp32
template <int D>
__global__ __launch_bounds__(256) void attn_decode_kernel(const bf16* __restrict__ q, long ld_q,
                                                          const bf16* __restrict__ kc, const bf16* __restrict__ vc,
                                                          const int* __restrict__ kv_len, int H, int KVH, int max_len,
                                                          int nsplit, float scale, float* __restrict__ part_ml,
                                                          float* __restrict__ part_o, bf16* __restrict__ out,
                                                          long ld_o) {
    constexpr int LPR = D / 8;        // lanes per key row
    constexpr int RPI = 64 / LPR;     // key rows per wave load instruction
    constexpr int U = 4;              // load instructions in flight per tensor
    __shared__ float sm_ml[4 * RPI * 2];
    __shared__ float sm_o[4 * RPI * D];

    const int split = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int kvh = h / (H / KVH);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = lane / LPR, sub = lane % LPR;
    const int len = kv_len[b];
    const int per = (len + nsplit - 1) / nsplit;
    const int k_beg = split * per, k_end = min(len, k_beg + per);

    float qv[8];
    {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(q + (long)b * ld_q + (long)h * D + sub * 8);
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
    __syncthreads();
    if (tid < D) {
        float M = NEG_BIG;
        for (int sidx = 0; sidx < 4 * RPI; ++sidx) M = fmaxf(M, sm_ml[sidx * 2]);
        float L = 0.f, o = 0.f;
        for (int sidx = 0; sidx < 4 * RPI; ++sidx) {
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

// workspace floats needed by mgen_attn_decode for a given launch shape
extern "C" long mgen_attn_decode_workspace(int B, int H, int D, int nsplit) {
    return nsplit <= 1 ? 0 : (long)B * H * nsplit * (D + 2);
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
    float* part_ml = workspace;
    float* part_o = workspace ? workspace + (long)B * H * nsplit * 2 : nullptr;
    const dim3 grid(nsplit, H, B), block(256);
    if (D == 128) {
        hipLaunchKernelGGL((attn_decode_kernel<128>), grid, block, 0, stream, (const bf16*)q, ld_q, (const bf16*)k_cache,
                           (const bf16*)v_cache, kv_len, H, KVH, max_len, nsplit, scale, part_ml, part_o, (bf16*)out, ld_o);
        if (nsplit > 1)
            hipLaunchKernelGGL((attn_decode_merge_kernel<128>), dim3(H, B), dim3(128), 0, stream, part_ml, part_o, nsplit,
                               H, (bf16*)out, ld_o);
    } else {
        hipLaunchKernelGGL((attn_decode_kernel<64>), grid, block, 0, stream, (const bf16*)q, ld_q, (const bf16*)k_cache,
                           (const bf16*)v_cache, kv_len, H, KVH, max_len, nsplit, scale, part_ml, part_o, (bf16*)out, ld_o);
        if (nsplit > 1)
            hipLaunchKernelGGL((attn_decode_merge_kernel<64>), dim3(H, B), dim3(64), 0, stream, part_ml, part_o, nsplit,
                               H, (bf16*)out, ld_o);
    }
    return mgen_check_launch("mgen_attn_decode");
}
