// Element-wise dropout for gfx950: masks come from a stateless hash of (key, element index) (common.h), so a
// backward pass repeats the forward call on the gradient with the same key and no mask is ever stored.
//
// Reference: torch.nn.Dropout as used by transformers' GPT2Model / GPT2Attention / GPT2MLP (modeling_gpt2.py:
// `self.drop` on the summed embeddings, `resid_dropout` on the attention output projection, `dropout` on the MLP
// projection; configuration_gpt2.py:85-87, all 0.1) and by the low-rank adapters of
// /root/reference/src/model/model.py:251-271 (lora_dropout 0.05 on the adapter's input). Kept elements are scaled by
// 1 / (1 - p) and rounded to the tensor's type, as torch does.
#include "common.h"

namespace {

template <typename T, int VEC, bool ACC>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, long n,
                                                      unsigned long long key, unsigned thresh, float inv_keep) {
    typedef __attribute__((ext_vector_type(VEC))) T vec_t;
    const long nv = n / VEC;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        vec_t v = reinterpret_cast<const vec_t*>(x)[i];
        vec_t old;
        if (ACC) old = reinterpret_cast<const vec_t*>(y)[i];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const bool keep = drop_hash_flat(key, (unsigned long long)(i * VEC + e)) >= thresh;
            v[e] = (T)(keep ? (float)v[e] * inv_keep : 0.f);
            if (ACC) v[e] = (T)((float)old[e] + (float)v[e]);
        }
        reinterpret_cast<vec_t*>(y)[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - nv * VEC)) {   // tail
        const long i = nv * VEC + threadIdx.x;
        const bool keep = drop_hash_flat(key, (unsigned long long)i) >= thresh;
        const T d = (T)(keep ? (float)x[i] * inv_keep : 0.f);
        y[i] = ACC ? (T)((float)y[i] + (float)d) : d;
    }
}

// out[i] = resid[i] + float(bf16(dropout(y)[i])): the residual add after a dropped bf16 projection output
__global__ __launch_bounds__(256) void dropout_resid_kernel(const bf16* __restrict__ y, const float* __restrict__ resid,
                                                            float* __restrict__ out, long n, unsigned long long key,
                                                            unsigned thresh, float inv_keep) {
    const long nv = n / 8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        const bf16x8 v = reinterpret_cast<const bf16x8*>(y)[i];
        const f32x4 r0 = reinterpret_cast<const f32x4*>(resid)[2 * i], r1 = reinterpret_cast<const f32x4*>(resid)[2 * i + 1];
        f32x4 o0, o1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool keep = drop_hash_flat(key, (unsigned long long)(i * 8 + e)) >= thresh;
            const float d = (float)(bf16)(keep ? (float)v[e] * inv_keep : 0.f);
            if (e < 4) o0[e] = r0[e] + d;
            else o1[e - 4] = r1[e - 4] + d;
        }
        reinterpret_cast<f32x4*>(out)[2 * i] = o0;
        reinterpret_cast<f32x4*>(out)[2 * i + 1] = o1;
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - nv * 8)) {
        const long i = nv * 8 + threadIdx.x;
        const bool keep = drop_hash_flat(key, (unsigned long long)i) >= thresh;
        out[i] = resid[i] + (float)(bf16)(keep ? (float)y[i] * inv_keep : 0.f);
    }
}

unsigned grid_for(long nvec) {
    long nb = (nvec + 255) / 256;
    if (nb < 1) nb = 1;
    if (nb > 16384) nb = 16384;
    return (unsigned)nb;
}

}  // namespace

namespace {
template <bool ACC>
int dropout_launch(const void* x, void* y, long n, int is_bf16, float p, unsigned long long key, hipStream_t stream,
                   const char* who) {
    MGEN_REQUIRE(x && y && n > 0, "%s: bad arguments", who);
    MGEN_REQUIRE(p >= 0.f && p < 1.f, "%s: dropout probability must be in [0, 1) (got %g)", who, p);
    MGEN_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0, "%s: pointers must be 16-byte aligned", who);
    const unsigned thresh = drop_threshold(p);
    const float inv_keep = 1.f / (1.f - p);
    if (is_bf16)
        hipLaunchKernelGGL((dropout_kernel<bf16, 8, ACC>), dim3(grid_for(n / 8)), dim3(256), 0, stream, (const bf16*)x,
                           (bf16*)y, n, key, thresh, inv_keep);
    else
        hipLaunchKernelGGL((dropout_kernel<float, 4, ACC>), dim3(grid_for(n / 4)), dim3(256), 0, stream,
                           (const float*)x, (float*)y, n, key, thresh, inv_keep);
    return mgen_check_launch(who);
}
}  // namespace

// y[i] = keep(key, i) ? x[i] / (1 - p) : 0 for i < n; x and y contiguous, of type bf16 (is_bf16) or fp32, 16-byte
// aligned; y may be x. The gradient of the call is the same call on the upstream gradient.
extern "C" int mgen_dropout(const void* x, void* y, long n, int is_bf16, float p, unsigned long long key,
                            hipStream_t stream) {
    return dropout_launch<false>(x, y, n, is_bf16, p, key, stream, "mgen_dropout");
}

// y[i] += dropout(x)[i] (each term rounded to the tensors' type): a dropped branch's gradient added to the gradient
// the other branches have already left in y. y must not alias x.
extern "C" int mgen_dropout_add(const void* x, void* y, long n, int is_bf16, float p, unsigned long long key,
                                hipStream_t stream) {
    MGEN_REQUIRE(x != y, "mgen_dropout_add: y must not alias x");
    return dropout_launch<true>(x, y, n, is_bf16, p, key, stream, "mgen_dropout_add");
}

// out[i] = resid[i] + float(bf16(keep(key, i) ? y[i] / (1 - p) : 0)): y bf16 [n], resid / out fp32 [n] (out may be
// resid). The gradient with respect to y is mgen_dropout on the bf16 gradient of out with the same key.
extern "C" int mgen_dropout_resid(const void* y, const float* resid, float* out, long n, float p,
                                  unsigned long long key, hipStream_t stream) {
    MGEN_REQUIRE(y && resid && out && n > 0, "mgen_dropout_resid: bad arguments");
    MGEN_REQUIRE(p >= 0.f && p < 1.f, "mgen_dropout_resid: dropout probability must be in [0, 1) (got %g)", p);
    MGEN_REQUIRE(((uintptr_t)y % 16) == 0 && ((uintptr_t)resid % 16) == 0 && ((uintptr_t)out % 16) == 0,
                 "mgen_dropout_resid: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(dropout_resid_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, (const bf16*)y, resid, out, n,
                       key, drop_threshold(p), 1.f / (1.f - p));
    return mgen_check_launch("mgen_dropout_resid");
}
