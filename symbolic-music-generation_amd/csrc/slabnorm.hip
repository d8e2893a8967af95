// Consumer of the split-K slabs of a residual projection on the decode path (few token rows, all CUs needed for
// one small GEMM): sums the fp32 partial products in slice order, applies the bf16 rounding of the projection output,
// adds the residual stream, and applies the RMSNorm that follows in the same pass.
//
// Reference: the residual adds and the norm that follows them in Qwen3DecoderLayer.forward
// (transformers models/qwen3/modeling_qwen3.py:294-323) and Qwen3RMSNorm (/root/reference/src/model/modeling.py:60-65):
//   h_new = h + bf16(x @ W^T) ;  y = w * (h_new * rsqrt(mean(h_new^2) + eps)) rounded to bf16
#include "common.h"

namespace {

template <int VPL>
__global__ __launch_bounds__(64) void slab_resid_rmsnorm_kernel(const float* __restrict__ slabs, int nslabs,
                                                                 long slab_stride, const float* __restrict__ resid,
                                                                 float* __restrict__ h_out,
                                                                 const float* __restrict__ w, bf16* __restrict__ y,
                                                                 long T, int d, float eps, int round_bf16) {
    // one wave per row and one row per workgroup: the rows are few (decode) and should land on different CUs
    const int lane = threadIdx.x & 63;
    const long row = blockIdx.x;
    if (row >= T) return;
    f32x4 hv[VPL];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * 4;
        hv[i] = f32x4{0, 0, 0, 0};
        if (c < d) {
            f32x4 acc = *reinterpret_cast<const f32x4*>(slabs + row * d + c);
#pragma unroll 8
            for (int z = 1; z < nslabs; ++z) acc += *reinterpret_cast<const f32x4*>(slabs + z * slab_stride + row * d + c);
            if (round_bf16) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = round_bf(acc[e]);
            }
            hv[i] = acc + *reinterpret_cast<const f32x4*>(resid + row * d + c);
            *reinterpret_cast<f32x4*>(h_out + row * d + c) = hv[i];
            ss += hv[i][0] * hv[i][0] + hv[i][1] * hv[i][1] + hv[i][2] * hv[i][2] + hv[i][3] * hv[i][3];
        }
    }
    if (!w) return;
    ss = wave_sum(ss);
    const float r = rsqrtf(ss / (float)d + eps);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < d) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + c);
            const f32x4 o = wv * (hv[i] * r);
            bf16x4 ob = {(bf16)o[0], (bf16)o[1], (bf16)o[2], (bf16)o[3]};
            *reinterpret_cast<bf16x4*>(y + row * d + c) = ob;
        }
    }
}

}  // namespace

// slabs: nslabs fp32 partial products [T][d] (slab z at slabs + z * slab_stride); resid, h_out fp32 [T][d] (may be the
// same buffer); w (fp32 [d]) and y (bf16 [T][d]) may both be null (no norm). round_bf16: round the summed projection
// to bf16 before the residual add (what autocast does to the projection output).
extern "C" int mgen_slab_resid_rmsnorm(const float* slabs, int nslabs, long slab_stride, const float* resid,
                                       float* h_out, const float* w, void* y, long T, int d, float eps, int round_bf16,
                                       hipStream_t stream) {
    MGEN_REQUIRE(slabs && resid && h_out && nslabs >= 1 && T > 0, "mgen_slab_resid_rmsnorm: bad arguments");
    MGEN_REQUIRE((w == nullptr) == (y == nullptr), "mgen_slab_resid_rmsnorm: w and y go together");
    MGEN_REQUIRE(d > 0 && d % 4 == 0 && d <= 2048, "mgen_slab_resid_rmsnorm: need 0<d<=2048, d%%4==0 (d=%d)", d);
    MGEN_REQUIRE(T < (1L << 31), "mgen_slab_resid_rmsnorm: too many rows");
    const dim3 grid((unsigned)T), block(64);
    if (d <= 512)
        hipLaunchKernelGGL(slab_resid_rmsnorm_kernel<2>, grid, block, 0, stream, slabs, nslabs, slab_stride, resid,
                           h_out, w, (bf16*)y, T, d, eps, round_bf16);
    else
        hipLaunchKernelGGL(slab_resid_rmsnorm_kernel<8>, grid, block, 0, stream, slabs, nslabs, slab_stride, resid,
                           h_out, w, (bf16*)y, T, d, eps, round_bf16);
    return mgen_check_launch("mgen_slab_resid_rmsnorm");
}
