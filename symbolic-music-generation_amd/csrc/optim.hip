// Optimizer step over the flat fp32 parameter store: global gradient L2 norm, clip coefficient, and fused AdamW
// that also refreshes the bf16 shadow copy the GEMMs read. HBM-bound (16 B read + 12 B written per parameter in
// fp32, + 2 B for the shadow), 16-byte accesses, grid-stride.
//
// Reference: gradient_clip_val=1.0 (/root/reference/src/train.py:66 -> torch.nn.utils.clip_grad_norm_: total L2 norm,
// coef = min(1, max_norm / (norm + 1e-6))) and torch.optim.AdamW(lr, weight_decay=0.1, betas=(0.9, 0.95), eps=1e-5)
// (/root/reference/src/model/model.py:19-25): p *= 1 - lr*wd; m = lerp(m, g, 1-b1); v = b2*v + (1-b2)*g*g;
// p -= (lr / bc1) * m / (sqrt(v)/sqrt(bc2) + eps).
#include "common.h"
#include <mutex>

namespace {

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, double* __restrict__ partial) {
    __shared__ float red[4];
    float acc = 0.f;
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(g + 4 * i);
        acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0)
        for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) acc += g[i] * g[i];
    const float s = block_sum_256(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = (double)s;
}

// out[0] = total norm, out[1] = clip coefficient (1 when max_norm <= 0); fixed summation order
__global__ void clip_finalize_kernel(const double* __restrict__ partial, int nparts, float max_norm,
                                     float* __restrict__ out) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(red[0]);
        out[0] = norm;
        float coef = 1.f;
        if (max_norm > 0.f) coef = fminf(max_norm / (norm + 1e-6f), 1.f);
        out[1] = coef;
    }
}

struct AdamArgs {
    float lr, beta1, beta2, eps, wd, inv_bc1, inv_sqrt_bc2;
};

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    bf16* __restrict__ shadow, long n, const float* __restrict__ clip,
                                                    AdamArgs a) {
    const float coef = clip ? clip[1] : 1.f;
    const long n4 = n / 4;
    const float step = a.lr * a.inv_bc1;
    const float decay = 1.f - a.lr * a.wd;
    // a workgroup owns contiguous spans of 4 x 4 KiB per stream and requests all 16 vectors before the arithmetic
    constexpr int UN = 4;
    const long per = 256L * UN;
    for (long b = blockIdx.x; b * per < n4; b += gridDim.x) {
        f32x4 pv[UN], gv[UN], mv[UN], vv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long i = b * per + u * 256 + threadIdx.x;
            if (i < n4) {
                pv[u] = ld_stream(reinterpret_cast<const f32x4*>(p + 4 * i));
                gv[u] = ld_stream(reinterpret_cast<const f32x4*>(g + 4 * i));
                mv[u] = ld_stream(reinterpret_cast<const f32x4*>(m + 4 * i));
                vv[u] = ld_stream(reinterpret_cast<const f32x4*>(v + 4 * i));
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long i = b * per + u * 256 + threadIdx.x;
            if (i >= n4) continue;
            bf16x4 sh;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ge = gv[u][e] * coef;
                pv[u][e] *= decay;
                mv[u][e] = mv[u][e] + (ge - mv[u][e]) * (1.f - a.beta1);
                vv[u][e] = vv[u][e] * a.beta2 + (1.f - a.beta2) * ge * ge;
                const float denom = sqrtf(vv[u][e]) * a.inv_sqrt_bc2 + a.eps;
                pv[u][e] -= step * (mv[u][e] / denom);
                sh[e] = (bf16)pv[u][e];
            }
            st_stream(reinterpret_cast<f32x4*>(p + 4 * i), pv[u]);
            st_stream(reinterpret_cast<f32x4*>(m + 4 * i), mv[u]);
            st_stream(reinterpret_cast<f32x4*>(v + 4 * i), vv[u]);
            if (shadow) st_stream(reinterpret_cast<bf16x4*>(shadow + 4 * i), sh);
        }
    }
}

__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, long n, float s,
                                                    const float* __restrict__ a_dev) {
    if (a_dev) s *= *a_dev;
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
        *reinterpret_cast<f32x4*>(x + 4 * i) = v * s;
    }
}

__global__ __launch_bounds__(256) void scale_bf16_kernel(bf16* __restrict__ x, long n, float a,
                                                         const float* __restrict__ a_dev) {
    if (a_dev) a *= *a_dev;
    const long n8 = n / 8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        bf16x8 v = *reinterpret_cast<const bf16x8*>(x + 8 * i);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16)((float)v[e] * a);
        *reinterpret_cast<bf16x8*>(x + 8 * i) = v;
    }
}

__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, long n, float a,
                                                    const float* __restrict__ a_dev) {
    if (a_dev) a *= *a_dev;
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + 4 * i);
        f32x4 yv = *reinterpret_cast<const f32x4*>(y + 4 * i);
        *reinterpret_cast<f32x4*>(y + 4 * i) = yv + xv * a;
    }
}

}  // namespace

extern "C" int mgen_sumsq_partial_count(void) { return 1024; }

// workspace: mgen_sumsq_partial_count() doubles; out: 2 floats {norm, clip coefficient}
extern "C" int mgen_grad_norm_clip(const float* grad, long n, float max_norm, double* workspace, float* out,
                                   hipStream_t stream) {
    MGEN_REQUIRE(grad && workspace && out && n > 0, "mgen_grad_norm_clip: bad arguments");
    MGEN_REQUIRE(((uintptr_t)grad % 16) == 0, "mgen_grad_norm_clip: grad must be 16-byte aligned");
    long nb = (n / 4 + 255) / 256;
    if (nb < 1) nb = 1;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)nb), dim3(256), 0, stream, grad, n, workspace);
    hipLaunchKernelGGL(clip_finalize_kernel, dim3(1), dim3(256), 0, stream, workspace, (int)nb, max_norm, out);
    return mgen_check_launch("mgen_grad_norm_clip");
}

// ---------------------------------------------------------------------------------------------------
// The same reduction for callers that cannot allocate or bring a workspace (code under HIP graph capture, serving
// loops, harnesses that forbid hipMalloc): the per-workgroup partial sums go to an array that is part of the code
// object. Every GPU the library runs on holds its own copy (8 KiB) until the process ends.
// ---------------------------------------------------------------------------------------------------
static __device__ __align__(256) double scratch[1024];   // = mgen_sumsq_partial_count() partial sums

namespace {
constexpr int kMaxDevices = 64;
constexpr long kStaticMaxN = 1L << 36;   // elements (256 GiB of fp32 gradients): far beyond one GPU's memory
std::mutex g_scratch_mutex;
double* g_scratch_ptr[kMaxDevices] = {};   // device address of `scratch` on each device (the name itself is a
                                           // host-side stand-in: only hipGetSymbolAddress yields the address)
}  // namespace

// out: 2 floats {norm, clip coefficient}; n <= 2^36. No allocation, no workspace argument. Calls that overlap on one
// GPU (two streams) race on the partial sums: serialise them. Nothing has to be cleared between calls: every
// partial the second kernel reads was written by the first kernel of the same call.
extern "C" int mgen_grad_norm_clip_static(const float* grad, long n, float max_norm, float* out, hipStream_t stream) {
    MGEN_REQUIRE(grad && out && n > 0, "mgen_grad_norm_clip_static: bad arguments");
    MGEN_REQUIRE(n <= kStaticMaxN, "mgen_grad_norm_clip_static: n = %ld exceeds the supported maximum of %ld elements",
                 n, kStaticMaxN);
    MGEN_REQUIRE(((uintptr_t)grad % 16) == 0, "mgen_grad_norm_clip_static: grad must be 16-byte aligned");
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= kMaxDevices) {
        mgen_set_error("mgen_grad_norm_clip_static: hipGetDevice: %s (device %d)", hipGetErrorString(e), dev);
        return MGEN_ELAUNCH;
    }
    double* ws;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_ptr[dev]) {   // once per device, with that device current
            void* p = nullptr;
            e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
            if (e != hipSuccess || !p) {
                mgen_set_error("mgen_grad_norm_clip_static: hipGetSymbolAddress: %s", hipGetErrorString(e));
                return MGEN_ELAUNCH;
            }
            g_scratch_ptr[dev] = static_cast<double*>(p);
        }
        ws = g_scratch_ptr[dev];
    }
    long nb = (n / 4 + 255) / 256;
    if (nb < 1) nb = 1;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)nb), dim3(256), 0, stream, grad, n, ws);
    hipLaunchKernelGGL(clip_finalize_kernel, dim3(1), dim3(256), 0, stream, ws, (int)nb, max_norm, out);
    return mgen_check_launch("mgen_grad_norm_clip_static");
}

// n must be a multiple of 4 (the flat store pads every tensor to 64 elements). `step` is 1-based.
extern "C" int mgen_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                               long n, const float* clip, float lr, float beta1, float beta2, float eps, float wd,
                               long step, hipStream_t stream) {
    MGEN_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && n % 4 == 0 && step >= 1,
                 "mgen_adamw_step: bad arguments (n must be a multiple of 4, step >= 1)");
    AdamArgs a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = wd;
    a.inv_bc1 = (float)(1.0 / (1.0 - pow((double)beta1, (double)step)));
    a.inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)beta2, (double)step)));
    long nb = (n / 4 + 1023) / 1024;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)nb), dim3(256), 0, stream, param, grad, exp_avg, exp_avg_sq,
                       (bf16*)shadow_bf16, n, clip, a);
    return mgen_check_launch("mgen_adamw_step");
}

// x *= s * (*a_dev) over n fp32 elements (n % 4 == 0); a_dev may be null
extern "C" int mgen_scale_f32(float* x, long n, float s, const float* a_dev, hipStream_t stream) {
    MGEN_REQUIRE(x && n > 0 && n % 4 == 0, "mgen_scale_f32: bad arguments");
    long nb = (n / 4 + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)nb), dim3(256), 0, stream, x, n, s, a_dev);
    return mgen_check_launch("mgen_scale_f32");
}

// y += a * (*a_dev) * x over n fp32 elements (n % 4 == 0); a_dev may be null (factor 1)
extern "C" int mgen_axpy_f32(float* y, const float* x, long n, float a, const float* a_dev, hipStream_t stream) {
    MGEN_REQUIRE(x && y && n > 0 && n % 4 == 0, "mgen_axpy_f32: bad arguments");
    long nb = (n / 4 + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)nb), dim3(256), 0, stream, y, x, n, a, a_dev);
    return mgen_check_launch("mgen_axpy_f32");
}

// x *= a * (*a_dev) over n bf16 elements (n % 8 == 0); a_dev may be null
extern "C" int mgen_scale_bf16(void* x, long n, float a, const float* a_dev, hipStream_t stream) {
    MGEN_REQUIRE(x && n > 0 && n % 8 == 0, "mgen_scale_bf16: bad arguments");
    long nb = (n / 8 + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(scale_bf16_kernel, dim3((unsigned)nb), dim3(256), 0, stream, (bf16*)x, n, a, a_dev);
    return mgen_check_launch("mgen_scale_bf16");
}
