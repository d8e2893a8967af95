// RMSNorm over the fp32 residual stream, forward and backward (HBM-bound, one wave per row, 16-byte accesses).
//
// Reference: Qwen3RMSNorm.forward, /root/reference/src/model/modeling.py:60-65
//   y = w * (x.float() * rsqrt(mean(x^2) + eps)).to(x.dtype)
// The residual stream is fp32 under the reference's bf16-mixed precision, so x is fp32 here; y is written in the
// dtype the consuming GEMM reads (bf16: the rounding autocast applies at the GEMM input; fp32: parity mode).
//
// Backward (autograd of the above), with xh = x * rstd and g = dy * w:
//   dx = rstd * (g - xh * mean(g * xh)),   dw = sum_rows dy * xh
// dx is added into the fp32 gradient of the residual stream (the norm sits on a residual branch) and the updated
// gradient is also emitted in the GEMM input dtype for the next dgrad/wgrad.
#include "common.h"
#include "dw_rows.h"
#include <map>
#include <mutex>
#include <utility>

namespace {

template <typename T> struct Vec4;
template <> struct Vec4<float> {
    static __device__ __forceinline__ f32x4 load(const float* p) { return ld_stream(reinterpret_cast<const f32x4*>(p)); }
    static __device__ __forceinline__ void store(float* p, f32x4 v) { st_stream(reinterpret_cast<f32x4*>(p), v); }
};
template <> struct Vec4<bf16> {
    static __device__ __forceinline__ f32x4 load(const bf16* p) {
        bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
        return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    }
    static __device__ __forceinline__ void store(bf16* p, f32x4 v) {
        bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        *reinterpret_cast<bf16x4*>(p) = o;
    }
};

// VPL = float4 vectors per lane; a row has d <= 256*VPL elements, d % 4 == 0
template <int VPL, typename OutT>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          OutT* __restrict__ y, float* __restrict__ rstd_out,
                                                          long T, int d, float eps) {
    const int lane = threadIdx.x & 63;
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * 4;
    f32x4 wv[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * 4;
        wv[i] = c < d ? Vec4<float>::load(w + c) : f32x4{0, 0, 0, 0};
    }
    // Two rows per wave and iteration, both requested before the first row reduction. The residual stream was
    // written by the kernel before this one and is read again by the next residual projection: plain loads.
    for (long row0 = wave0; row0 < T; row0 += 2 * nwaves) {
        const long row1 = row0 + nwaves;
        f32x4 xa[VPL], xb[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * 64 + lane) * 4;
            xa[i] = c < d ? *reinterpret_cast<const f32x4*>(x + row0 * d + c) : f32x4{0, 0, 0, 0};
            xb[i] = (c < d && row1 < T) ? *reinterpret_cast<const f32x4*>(x + row1 * d + c) : f32x4{0, 0, 0, 0};
        }
        auto finish = [&](long row, const f32x4 (&xv)[VPL]) {
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i)
                ss += xv[i][0] * xv[i][0] + xv[i][1] * xv[i][1] + xv[i][2] * xv[i][2] + xv[i][3] * xv[i][3];
            ss = wave_sum(ss);
            const float r = rsqrtf(ss / (float)d + eps);
            if (lane == 0 && rstd_out) rstd_out[row] = r;
            OutT* yr = y + row * d;
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const int c = (i * 64 + lane) * 4;
                if (c < d) Vec4<OutT>::store(yr + c, wv[i] * (xv[i] * r));
            }
        };
        finish(row0, xa);
        if (row1 < T) finish(row1, xb);
    }
}

// h = resid + float(y) (y: the bf16 output of a residual branch's last projection), then the RMSNorm of h as above: the
// residual addition of the decoder layer (transformers models/qwen3/modeling_qwen3.py:316,322) and the norm that
// follows it (:59-64) in one pass over the row. Same arithmetic, in the same order, as the fp32 residual epilogue of
// the projection followed by rmsnorm_fwd_kernel: same bits.
template <int VPL>
__global__ __launch_bounds__(256) void resid_rmsnorm_fwd_kernel(const float* __restrict__ resid,
                                                                const bf16* __restrict__ y, long ldy,
                                                                const float* __restrict__ w, float* __restrict__ h_out,
                                                                bf16* __restrict__ x_out, float* __restrict__ rstd_out,
                                                                long T, int d, float eps) {
    const int lane = threadIdx.x & 63;
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * 4;
    f32x4 wv[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * 4;
        wv[i] = c < d ? Vec4<float>::load(w + c) : f32x4{0, 0, 0, 0};
    }
    for (long row0 = wave0; row0 < T; row0 += 2 * nwaves) {
        const long row1 = row0 + nwaves;
        f32x4 xa[VPL], xb[VPL];
        bf16x4 ya[VPL], yb[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * 64 + lane) * 4;
            const bool in0 = c < d, in1 = c < d && row1 < T;
            xa[i] = in0 ? ld_stream(reinterpret_cast<const f32x4*>(resid + row0 * d + c)) : f32x4{0, 0, 0, 0};
            ya[i] = in0 ? ld_stream(reinterpret_cast<const bf16x4*>(y + row0 * ldy + c)) : bf16x4{};
            xb[i] = in1 ? ld_stream(reinterpret_cast<const f32x4*>(resid + row1 * d + c)) : f32x4{0, 0, 0, 0};
            yb[i] = in1 ? ld_stream(reinterpret_cast<const bf16x4*>(y + row1 * ldy + c)) : bf16x4{};
        }
        auto finish = [&](long row, f32x4 (&xv)[VPL], const bf16x4 (&yv)[VPL]) {
            float ss = 0.f;
            float* hr = h_out + row * d;
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const int c = (i * 64 + lane) * 4;
                xv[i] += f32x4{(float)yv[i][0], (float)yv[i][1], (float)yv[i][2], (float)yv[i][3]};
                if (c < d) *reinterpret_cast<f32x4*>(hr + c) = xv[i];
            }
#pragma unroll
            for (int i = 0; i < VPL; ++i)
                ss += xv[i][0] * xv[i][0] + xv[i][1] * xv[i][1] + xv[i][2] * xv[i][2] + xv[i][3] * xv[i][3];
            ss = wave_sum(ss);
            const float r = rsqrtf(ss / (float)d + eps);
            if (lane == 0 && rstd_out) rstd_out[row] = r;
            bf16* xr = x_out + row * d;
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const int c = (i * 64 + lane) * 4;
                if (c < d) Vec4<bf16>::store(xr + c, wv[i] * (xv[i] * r));
            }
        };
        finish(row0, xa, ya);
        if (row1 < T) finish(row1, xb, yb);
    }
}

// Gain gradient of mgen_rmsnorm_bwd. Every workgroup of a launch leaves its partial gain-gradient row in a row of its
// own in a device array (plain stores), and a second small launch on the same stream adds the rows, column by column,
// into dw: no atomics, so the result does not depend on the order the workgroups finish in. (fp32 atomics on device
// memory run at 6-9 G/s on this GPU whatever the number of distinct addresses: one atomic per element and workgroup,
// even spread over 8 copies of the row, cost 19 us of a 104 us launch at 256 workgroups x 512 columns and 134 of 241 us
// at 768 x 1536.) The array holds NORM_DW_SLOTS such sets of rows: the entry point gives every (device, stream) pair a
// set of its own, so launches that overlap on one GPU (different streams) never share one, and launches on one stream
// are ordered. A stream that finds every set taken runs the DIRECT form (atomics straight into dw: no shared state).
constexpr int NORM_DW_MAXB = MGEN_DW_MAXB;   // workgroups per launch at most
constexpr int NORM_DW_SLOTS = 8;
static __device__ float g_dw_rows[(long)NORM_DW_SLOTS * NORM_DW_MAXB * MGEN_DW_ROW];

// Workgroups of 16 waves, one per CU (16 waves for rows of up to 512 elements; 4 for longer rows, whose per-lane
// state needs more than 128 VGPRs).
#ifndef MGEN_NB6_WAVES
#define MGEN_NB6_WAVES 4
#endif
#ifndef MGEN_NB6_CAP
#define MGEN_NB6_CAP 512
#endif
template <int VPL, typename DyT, bool DIRECT, int BWD_WAVES = (VPL <= 2 ? 16 : (VPL == 6 ? MGEN_NB6_WAVES : 4))>
__global__ __launch_bounds__(64 * BWD_WAVES) void rmsnorm_bwd_kernel(const DyT* __restrict__ dy,
                                                                     const float* __restrict__ x,
                                                                     const float* __restrict__ w,
                                                                     const float* __restrict__ rstd,
                                                                     float* __restrict__ dres,
                                                                     DyT* __restrict__ dres_lp, float* __restrict__ dw,
                                                                     long T, int d, int accumulate, int slot) {
    __shared__ float red[BWD_WAVES * 64 * 4];  // per-wave dw partials for one 256-element slab of the row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long wave0 = (long)blockIdx.x * BWD_WAVES + wave;
    const long nwaves = (long)gridDim.x * BWD_WAVES;
    f32x4 wv[VPL], dwv[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * 4;
        wv[i] = c < d ? Vec4<float>::load(w + c) : f32x4{0, 0, 0, 0};
        dwv[i] = f32x4{0, 0, 0, 0};
    }
    for (long row = wave0; row < T; row += nwaves) {
        const float r = rstd[row];
        f32x4 xh[VPL], gv[VPL], old[VPL];
        float dot = 0.f;
        // (the running gradient is requested together with the other two streams, not after the row reduction:
        // 120 -> 111 us at 65536 x 512; a second row per wave in flight added nothing)
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * 64 + lane) * 4;
            old[i] = (accumulate && c < d) ? Vec4<float>::load(dres + row * d + c) : f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < d) {
                xh[i] = Vec4<float>::load(x + row * d + c) * r;
                const f32x4 dyv = Vec4<DyT>::load(dy + row * d + c);
                gv[i] = dyv * wv[i];
                dwv[i] += dyv * xh[i];
                dot += gv[i][0] * xh[i][0] + gv[i][1] * xh[i][1] + gv[i][2] * xh[i][2] + gv[i][3] * xh[i][3];
            } else {
                xh[i] = f32x4{0, 0, 0, 0};
                gv[i] = f32x4{0, 0, 0, 0};
            }
        }
        dot = wave_sum(dot) / (float)d;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < d) {
                const f32x4 dx = (gv[i] - xh[i] * dot) * r + old[i];
                float* p = dres + row * d + c;
                Vec4<float>::store(p, dx);
                if (dres_lp) Vec4<DyT>::store(dres_lp + row * d + c, dx);
            }
        }
    }
    // dw: the waves of the block meet in LDS; the block's row goes to its place in the slot's set of rows
    float* myrow = g_dw_rows + ((long)slot * NORM_DW_MAXB + blockIdx.x) * 2048;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        __syncthreads();
        *reinterpret_cast<f32x4*>(&red[(wave * 64 + lane) * 4]) = dwv[i];
        __syncthreads();
        if (wave == 0) {
            f32x4 s = *reinterpret_cast<f32x4*>(&red[lane * 4]);
            for (int k = 1; k < BWD_WAVES; ++k) s += *reinterpret_cast<f32x4*>(&red[(k * 64 + lane) * 4]);
            const int c = (i * 64 + lane) * 4;
            if (c < d) {
                if (DIRECT)
                    for (int e = 0; e < 4; ++e) atomicAdd(dw + c + e, s[e]);
                else
                    *reinterpret_cast<f32x4*>(myrow + c) = s;
            }
        }
    }
}

// dw[col] += sum over the nblocks rows of the slot's set; one workgroup of 16 waves per 64 columns
__global__ __launch_bounds__(1024) void rmsnorm_dw_reduce_kernel(float* __restrict__ dw, int d, int nblocks, int slot,
                                                                 int col0) {
    __shared__ float part[16][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const float* rows = g_dw_rows + (long)slot * NORM_DW_MAXB * 2048 + col0;
    float s = 0.f;
    if (col < d) {
        // eight rows requested before the first is added (one load after the other is 32 cache round trips in a row)
        int b = g;
        for (; b + 7 * 16 < nblocks; b += 8 * 16) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = rows[(long)(b + 16 * k) * 2048 + col];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; b < nblocks; b += 16) s += rows[(long)b * 2048 + col];
    }
    part[g][lane] = s;
    __syncthreads();
    if (g == 0 && col < d) {
        float t = part[0][lane];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += part[k][lane];
        dw[col] += t;
    }
}

// (device, stream) -> set of gain-gradient copies; -1 when all sets of the device are taken by other streams
int norm_dw_slot(hipStream_t stream) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, int> slots;
    static std::map<int, int> used;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    std::lock_guard<std::mutex> lock(mu);
    auto it = slots.find({dev, stream});
    if (it != slots.end()) return it->second;
    int& n = used[dev];
    if (n >= NORM_DW_SLOTS) return -1;
    slots[{dev, stream}] = n;
    return n++;
}

template <int VPL, typename DyT>
void launch_rmsnorm_bwd(hipStream_t stream, int slot, const void* dy, const float* x, const float* w,
                        const float* rstd, float* dres, void* dres_lp, float* dw, long T, int d, int accumulate) {
    constexpr int NW = VPL <= 2 ? 16 : (VPL == 6 ? MGEN_NB6_WAVES : 4);
    const long b = (T + NW - 1) / NW;
    // one 16-wave block per CU / two 4-wave blocks (rows of up to 1536; three measured the same) / two 4-wave blocks
    static_assert(MGEN_NB6_CAP <= NORM_DW_MAXB && 512 <= NORM_DW_MAXB, "a launch's workgroups need a row each");
    const long cap = VPL <= 2 ? 256 : (VPL <= 6 ? MGEN_NB6_CAP : 512);
    const dim3 grid((int)(b < cap ? b : cap));
    const dim3 block(64 * NW);
    if (slot >= 0) {
        hipLaunchKernelGGL((rmsnorm_bwd_kernel<VPL, DyT, false>), grid, block, 0, stream, (const DyT*)dy, x, w, rstd,
                           dres, (DyT*)dres_lp, dw, T, d, accumulate, slot);
        hipLaunchKernelGGL(rmsnorm_dw_reduce_kernel, dim3((d + 63) / 64), dim3(1024), 0, stream, dw, d, (int)grid.x,
                           slot, 0);
    } else
        hipLaunchKernelGGL((rmsnorm_bwd_kernel<VPL, DyT, true>), grid, block, 0, stream, (const DyT*)dy, x, w, rstd,
                           dres, (DyT*)dres_lp, dw, T, d, accumulate, 0);
}

inline int norm_grid(long T) {
    long b = (T + 3) / 4;
    return (int)(b < 2048 ? (b < 1 ? 1 : b) : 2048);
}

}  // namespace

int mgen_dw_slot(hipStream_t stream) { return norm_dw_slot(stream); }

float* mgen_dw_rows(int slot) {
    static float* base[16] = {};
    int dev = 0;
    if (slot < 0 || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    if (!base[dev]) {   // (in host code the array's name is a stand-in: only hipGetSymbolAddress yields the address)
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_dw_rows)) != hipSuccess) return nullptr;
        base[dev] = reinterpret_cast<float*>(p);
    }
    return base[dev] + (long)slot * NORM_DW_MAXB * MGEN_DW_ROW;
}

void mgen_dw_reduce(float* dst, int n, int nblocks, int slot, int col0, hipStream_t stream) {
    hipLaunchKernelGGL(rmsnorm_dw_reduce_kernel, dim3((n + 63) / 64), dim3(1024), 0, stream, dst, n, nblocks, slot,
                       col0);
}


#define DISPATCH_VPL(d, CALL)                                                             \
    if ((d) <= 512) { constexpr int VPL = 2; CALL; }                                      \
    else { constexpr int VPL = 8; CALL; }

extern "C" int mgen_rmsnorm_fwd(const float* x, const float* w, void* y, float* rstd, long T, int d, float eps,
                                int out_bf16, hipStream_t stream) {
    MGEN_REQUIRE(x && w && y, "mgen_rmsnorm_fwd: null pointer");
    MGEN_REQUIRE(T > 0 && d > 0 && d % 4 == 0 && d <= 2048, "mgen_rmsnorm_fwd: need 0<d<=2048, d%%4==0 (d=%d)", d);
    const dim3 grid(norm_grid(T)), block(256);
    DISPATCH_VPL(d, {
        if (out_bf16)
            hipLaunchKernelGGL((rmsnorm_fwd_kernel<VPL, bf16>), grid, block, 0, stream, x, w, (bf16*)y, rstd, T, d, eps);
        else
            hipLaunchKernelGGL((rmsnorm_fwd_kernel<VPL, float>), grid, block, 0, stream, x, w, (float*)y, rstd, T, d, eps);
    });
    return mgen_check_launch("mgen_rmsnorm_fwd");
}

extern "C" int mgen_resid_rmsnorm_fwd(const float* resid, const void* y, long ldy, const float* w, float* h_out,
                                      void* x_out, float* rstd, long T, int d, float eps, hipStream_t stream) {
    MGEN_REQUIRE(resid && y && w && h_out && x_out, "mgen_resid_rmsnorm_fwd: null pointer");
    MGEN_REQUIRE(T > 0 && d > 0 && d % 4 == 0 && d <= 2048 && ldy >= d && ldy % 4 == 0,
                 "mgen_resid_rmsnorm_fwd: need 0<d<=2048, d%%4==0, ldy>=d, ldy%%4==0 (d=%d, ldy=%ld)", d, ldy);
    MGEN_REQUIRE(((uintptr_t)resid % 16) == 0 && ((uintptr_t)y % 8) == 0 && ((uintptr_t)h_out % 16) == 0 &&
                     ((uintptr_t)x_out % 8) == 0 && ((uintptr_t)w % 16) == 0,
                 "mgen_resid_rmsnorm_fwd: operands must be 16-byte (fp32) / 8-byte (bf16) aligned");
    const dim3 grid(norm_grid(T)), block(256);
    DISPATCH_VPL(d, {
        hipLaunchKernelGGL((resid_rmsnorm_fwd_kernel<VPL>), grid, block, 0, stream, resid, (const bf16*)y, ldy, w, h_out,
                           (bf16*)x_out, rstd, T, d, eps);
    });
    return mgen_check_launch("mgen_resid_rmsnorm_fwd");
}

extern "C" int mgen_rmsnorm_bwd(const void* dy, const float* x, const float* w, const float* rstd, float* dres,
                                void* dres_lp, float* dw, long T, int d, int accumulate, int dy_bf16,
                                hipStream_t stream) {
    MGEN_REQUIRE(dy && x && w && rstd && dres && dw, "mgen_rmsnorm_bwd: null pointer");
    MGEN_REQUIRE(T > 0 && d > 0 && d % 4 == 0 && d <= 2048, "mgen_rmsnorm_bwd: need 0<d<=2048, d%%4==0 (d=%d)", d);
    const int slot = norm_dw_slot(stream);
    if (d <= 512) {
        if (dy_bf16) launch_rmsnorm_bwd<2, bf16>(stream, slot, dy, x, w, rstd, dres, dres_lp, dw, T, d, accumulate);
        else launch_rmsnorm_bwd<2, float>(stream, slot, dy, x, w, rstd, dres, dres_lp, dw, T, d, accumulate);
    } else if (d <= 1536) {
        if (dy_bf16) launch_rmsnorm_bwd<6, bf16>(stream, slot, dy, x, w, rstd, dres, dres_lp, dw, T, d, accumulate);
        else launch_rmsnorm_bwd<6, float>(stream, slot, dy, x, w, rstd, dres, dres_lp, dw, T, d, accumulate);
    } else {
        if (dy_bf16) launch_rmsnorm_bwd<8, bf16>(stream, slot, dy, x, w, rstd, dres, dres_lp, dw, T, d, accumulate);
        else launch_rmsnorm_bwd<8, float>(stream, slot, dy, x, w, rstd, dres, dres_lp, dw, T, d, accumulate);
    }
    return mgen_check_launch("mgen_rmsnorm_bwd");
}
