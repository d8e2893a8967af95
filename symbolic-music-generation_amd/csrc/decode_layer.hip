// One decoder layer of the one-token decode step as ONE call: the seven launches of a layer (QKV projection,
// attention over the cache with the new token's q/k norm + rotation + cache append, o_proj slabs, slab sum + residual
// + norm, gate/up projection, down projection slabs with the gated operand, slab sum + residual + next norm) are
// issued from here, so the host pays one call per layer instead of seven (at short contexts the decode step of a
// small model is otherwise bounded by the host's launch rate, not by the GPU).
//
// Reference: Qwen3DecoderLayer.forward with q_len = 1 and a cache (transformers models/qwen3/modeling_qwen3.py:
// 294-323, 252-279, 81-83; cache_utils.py:127-146) under GenerationMixin._sample (generation/utils.py:2876-2937).
#include "common.h"

extern "C" int mgen_gemm_skinny_fits(int M, int N, int K, long lda, long ldw, long ldc, int slabs);
extern "C" int mgen_gemm_skinny_slabs(int M, int N, int K);
extern "C" int mgen_gemm_bf16_skinny(const void* A, const void* W, void* C, int M, int N, int K, long lda, long ldw,
                                     long ldc, int slabs, long slab_stride, int a_swiglu, hipStream_t stream);
extern "C" int mgen_attn_decode_prep(const void* qkv, long ld_in, const float* wq, const float* wk,
                                     const float* cos_sin, const int* pos, float eps, int interleaved, void* k_cache,
                                     void* v_cache, const int* kv_len, void* out, long ld_o, float* workspace, int B,
                                     int H, int KVH, int D, int max_len, int nsplit, float scale, hipStream_t stream);
extern "C" int mgen_slab_resid_rmsnorm(const float* slabs, int nslabs, long slab_stride, const float* resid,
                                       float* h_out, const float* w, void* y, long T, int d, float eps, int round_bf16,
                                       hipStream_t stream);

// 1 when mgen_decode_layer takes these shapes (every projection on the few-row kernel)
extern "C" int mgen_decode_layer_fits(int B, int d, int H, int KVH, int D, int I) {
    const int NQ = (H + 2 * KVH) * D, HD = H * D;
    if (d < 64 || HD < 512 || I < 512) return 0;
    return mgen_gemm_skinny_fits(B, NQ, d, d, d, NQ, 0) && mgen_gemm_skinny_fits(B, 2 * I, d, d, d, 2 * I, 0) &&
           mgen_gemm_skinny_fits(B, d, HD, HD, HD, d, mgen_gemm_skinny_slabs(B, d, HD)) &&
           mgen_gemm_skinny_fits(B, d, I, 2L * I, I, d, mgen_gemm_skinny_slabs(B, d, I));
}

// x      bf16 [B][d]   in: this layer's normed input; out: the NEXT norm's output (next layer's input / final norm)
// h_in   fp32 [B][d]   residual stream entering the layer; h_mid, h_out fp32 [B][d] (h_out may be h_in's buffer of
//                      the layer before: three buffers rotate)
// qkv bf16 [B][(H+2KVH)D], att bf16 [B][H D], gu bf16 [B][2I]: scratch; slabs fp32 [n][B][d] with n = the larger of
// mgen_gemm_skinny_slabs(B, d, HD) and mgen_gemm_skinny_slabs(B, d, I): scratch
// weights: w_qkv [(H+2KVH)D][d], w_o [d][HD], w_gu [2I][d], w_down [d][I] bf16; gains fp32: w_qn, w_kn [D] (both may be
// null), w_ln2 [d] (post-attention norm), w_next [d] (the norm that follows the layer)
// pos[b] = position of the new token, kv_len[b] = pos[b] + 1; caches [B][KVH][max_len][D] bf16
extern "C" int mgen_decode_layer(void* x, const float* h_in, float* h_mid, float* h_out, void* qkv, void* att, void* gu,
                                 float* slabs, float* attn_workspace, const void* w_qkv, const float* w_qn,
                                 const float* w_kn, const void* w_o, const float* w_ln2, const void* w_gu,
                                 const void* w_down, const float* w_next, const float* cos_sin, const int* pos,
                                 const int* kv_len, void* k_cache, void* v_cache, int B, int d, int H, int KVH, int D,
                                 int I, int max_len, int nsplit, float eps, int interleaved, hipStream_t stream) {
    MGEN_REQUIRE(x && h_in && h_mid && h_out && qkv && att && gu && slabs && w_qkv && w_o && w_ln2 && w_gu && w_down &&
                     w_next && cos_sin && pos && kv_len && k_cache && v_cache,
                 "mgen_decode_layer: null pointer");
    MGEN_REQUIRE(mgen_decode_layer_fits(B, d, H, KVH, D, I), "mgen_decode_layer: shapes not taken (B=%d d=%d H=%d KVH=%d "
                 "D=%d I=%d): see mgen_decode_layer_fits", B, d, H, KVH, D, I);
    const int NQ = (H + 2 * KVH) * D, HD = H * D;
    const long bd = (long)B * d;
    const float scale = 1.0f / sqrtf((float)D);
    if (int e = mgen_gemm_bf16_skinny(x, w_qkv, qkv, B, NQ, d, d, d, NQ, 0, 0, 0, stream)) return e;
    if (int e = mgen_attn_decode_prep(qkv, NQ, w_qn, w_kn, cos_sin, pos, eps, interleaved, k_cache, v_cache, kv_len, att,
                                      HD, attn_workspace, B, H, KVH, D, max_len, nsplit, scale, stream))
        return e;
    const int ns_o = mgen_gemm_skinny_slabs(B, d, HD), ns_d = mgen_gemm_skinny_slabs(B, d, I);
    if (int e = mgen_gemm_bf16_skinny(att, w_o, slabs, B, d, HD, HD, HD, d, ns_o, bd, 0, stream)) return e;
    if (int e = mgen_slab_resid_rmsnorm(slabs, ns_o, bd, h_in, h_mid, w_ln2, x, B, d, eps, 1, stream)) return e;
    if (int e = mgen_gemm_bf16_skinny(x, w_gu, gu, B, 2 * I, d, d, d, 2 * I, 0, 0, 0, stream)) return e;
    if (int e = mgen_gemm_bf16_skinny(gu, w_down, slabs, B, d, I, 2L * I, I, d, ns_d, bd, 1, stream)) return e;
    return mgen_slab_resid_rmsnorm(slabs, ns_d, bd, h_mid, h_out, w_next, x, B, d, eps, 1, stream);
}

namespace {
__global__ void decode_advance_kernel(int* __restrict__ pos, int* __restrict__ kv_len, int n, int* __restrict__ step) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        pos[i] += 1;
        kv_len[i] += 1;
    }
    if (i == 0 && step) step[0] += 1;
}
}  // namespace

// End of a decode step whose launches keep the same arguments at every step (graph replay): every stream's position
// and key count move on by one, and so does the step counter mgen_sample_step reads.
extern "C" int mgen_decode_advance(int* pos, int* kv_len, int n, int* step, hipStream_t stream) {
    MGEN_REQUIRE(pos && kv_len && n > 0, "mgen_decode_advance: bad arguments");
    hipLaunchKernelGGL(decode_advance_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, pos, kv_len, n, step);
    return mgen_check_launch("mgen_decode_advance");
}
