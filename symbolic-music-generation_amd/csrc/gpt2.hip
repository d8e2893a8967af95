// Kernels only the GPT-2 style decoder needs: LayerNorm (with bias) over the fp32 residual stream, the tanh
// approximation of GELU, and the column sum that produces bias gradients. All HBM-bound, 16-byte accesses.
//
// Reference: GPT2Block / GPT2MLP in transformers models/gpt2/modeling_gpt2.py (ln_1, ln_2, ln_f = nn.LayerNorm with
// eps 1e-5 :252-254,497; act = NewGELUActivation :233-235, activations.py:
//   0.5 * x * (1 + tanh(sqrt(2/pi) * (x + 0.044715 * x^3))) ), built by the reference at
// /root/reference/src/model/model.py:54-65. Under autocast LayerNorm runs in fp32 on the fp32 residual stream and its
// output is rounded at the next GEMM input; the activation is evaluated in fp32 from the bf16 GEMM output (pow is an
// fp32 autocast op) and rounded at the next GEMM input.
#include "common.h"
#include "dw_rows.h"

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4(const bf16* p) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void st4(bf16* p, f32x4 v) {
    bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    *reinterpret_cast<bf16x4*>(p) = o;
}
__device__ __forceinline__ float hsum(f32x4 v) { return v[0] + v[1] + v[2] + v[3]; }

// One wave per row; VPL float4 vectors per lane (d <= 256*VPL, d % 4 == 0).
//   y = (x - mean) * rstd * w + b,   rstd = rsqrt(var + eps), var = mean((x - mean)^2)
template <int VPL, typename OutT>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, OutT* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            long T, int d, float eps) {
    const int lane = threadIdx.x & 63;
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * 4;
    f32x4 wv[VPL], bv[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * 4;
        wv[i] = c < d ? ld4(w + c) : f32x4{0, 0, 0, 0};
        bv[i] = c < d ? ld4(b + c) : f32x4{0, 0, 0, 0};
    }
    for (long row = wave0; row < T; row += nwaves) {
        const float* xr = x + row * d;
        f32x4 xv[VPL];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * 64 + lane) * 4;
            xv[i] = c < d ? ld4(xr + c) : f32x4{0, 0, 0, 0};
            s += hsum(xv[i]);
        }
        const float mu = wave_sum(s) / (float)d;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < d) {
                xv[i] = xv[i] - mu;
                ss += hsum(xv[i] * xv[i]);
            }
        }
        const float r = rsqrtf(wave_sum(ss) / (float)d + eps);
        if (lane == 0 && mean_out) mean_out[row] = mu;
        if (lane == 0 && rstd_out) rstd_out[row] = r;
        OutT* yr = y + row * d;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < d) st4(yr + c, xv[i] * r * wv[i] + bv[i]);
        }
    }
}

// With xh = (x - mean) * rstd and g = dy * w:
//   dx = rstd * (g - mean(g) - xh * mean(g * xh)),  dw = sum_rows dy * xh,  db = sum_rows dy
// dx is added into the fp32 gradient of the residual stream (`dres`, or overwrites it when !accumulate) and also
// emitted in the GEMM input dtype.
template <int VPL, typename DyT>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const DyT* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ dres,
                                                            DyT* __restrict__ dres_lp, float* __restrict__ dw,
                                                            float* __restrict__ db, long T, int d, int accumulate,
                                                            float* __restrict__ rows) {
    __shared__ float red[4 * 256 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long wave0 = (long)blockIdx.x * 4 + wave;
    const long nwaves = (long)gridDim.x * 4;
    f32x4 wv[VPL], dwv[VPL], dbv[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * 4;
        wv[i] = c < d ? ld4(w + c) : f32x4{0, 0, 0, 0};
        dwv[i] = f32x4{0, 0, 0, 0};
        dbv[i] = f32x4{0, 0, 0, 0};
    }
    for (long row = wave0; row < T; row += nwaves) {
        const float r = rstd[row], mu = mean[row];
        f32x4 xh[VPL], gv[VPL];
        float dot = 0.f, gs = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < d) {
                xh[i] = (ld4(x + row * d + c) - mu) * r;
                const f32x4 dyv = ld4(dy + row * d + c);
                gv[i] = dyv * wv[i];
                dwv[i] += dyv * xh[i];
                dbv[i] += dyv;
                dot += hsum(gv[i] * xh[i]);
                gs += hsum(gv[i]);
            } else {
                xh[i] = f32x4{0, 0, 0, 0};
                gv[i] = f32x4{0, 0, 0, 0};
            }
        }
        dot = wave_sum(dot) / (float)d;
        gs = wave_sum(gs) / (float)d;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < d) {
                f32x4 dx = (gv[i] - gs - xh[i] * dot) * r;
                float* p = dres + row * d + c;
                if (accumulate) dx += ld4(p);
                st4(p, dx);
                if (dres_lp) st4(dres_lp + row * d + c, dx);
            }
        }
    }
#pragma unroll
    for (int which = 0; which < 2; ++which)
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            __syncthreads();
            *reinterpret_cast<f32x4*>(&red[(wave * 64 + lane) * 4]) = which ? dbv[i] : dwv[i];
            __syncthreads();
            if (wave == 0) {
                f32x4 s = *reinterpret_cast<f32x4*>(&red[lane * 4]);
                for (int k = 1; k < 4; ++k) s += *reinterpret_cast<f32x4*>(&red[(k * 64 + lane) * 4]);
                const int c = (i * 64 + lane) * 4;
                if (c < d) {
                    // the block's partial row: [dw | db] in its row of the stream's set (summed by a second launch),
                    // or, without a set, atomics straight into the gradients
                    if (rows) *reinterpret_cast<f32x4*>(rows + (long)blockIdx.x * MGEN_DW_ROW + which * d + c) = s;
                    else
                        for (int e = 0; e < 4; ++e) atomicAdd((which ? db : dw) + c + e, s[e]);
                }
            }
        }
}

constexpr float kGeluA = 0.7978845608028654f;  // sqrt(2/pi)
constexpr float kGeluB = 0.044715f;

__device__ __forceinline__ float gelu_new(float x) {
    const float t = tanhf(kGeluA * (x + kGeluB * x * x * x));
    return 0.5f * x * (1.0f + t);
}
__device__ __forceinline__ float gelu_new_grad(float x) {
    const float u = kGeluA * (x + kGeluB * x * x * x);
    const float t = tanhf(u);
    const float du = kGeluA * (1.0f + 3.0f * kGeluB * x * x);
    return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * du;
}

// act = gelu_new(h) (bf16 in, bf16 out: the rounding the next GEMM input applies); dh = dact * gelu_new'(h)
template <bool BWD>
__global__ __launch_bounds__(256) void gelu_kernel(const bf16* __restrict__ h, const bf16* __restrict__ dact,
                                                   bf16* __restrict__ out, long n8) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const bf16x8 hv = reinterpret_cast<const bf16x8*>(h)[i];
        bf16x8 o;
        if (BWD) {
            const bf16x8 dv = reinterpret_cast<const bf16x8*>(dact)[i];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)((float)dv[e] * gelu_new_grad((float)hv[e]));
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)gelu_new((float)hv[e]);
        }
        reinterpret_cast<bf16x8*>(out)[i] = o;
    }
}

// out[n] (+)= sum_t x[t][n]  (x bf16 [T][ld], n < N, N % 8 == 0). Block = 256 threads = 32 column groups of 8 x 8 row
// lanes; each block reduces a strip of rows and leaves its partial in the strip's row of the stream's set of partial
// rows (or, without a set, adds it with atomics).
__global__ __launch_bounds__(256) void colsum_kernel(const bf16* __restrict__ x, long ld, float* __restrict__ out,
                                                     long T, int N, long rows_per_block, float* __restrict__ rows) {
    __shared__ float red[8][32][8];
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int n = (blockIdx.x * 32 + cg) * 8;
    const long t0 = (long)blockIdx.y * rows_per_block;
    const long t1 = t0 + rows_per_block < T ? t0 + rows_per_block : T;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (n < N)
        for (long t = t0 + rl; t < t1; t += 8) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + t * ld + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
        }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][cg][e] = s[e];
    __syncthreads();
    if (rl == 0 && n < N) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = 0.f;
            for (int k = 0; k < 8; ++k) a += red[k][cg][e];
            if (rows) rows[(long)blockIdx.y * MGEN_DW_ROW + n + e] = a;   // the strip's partial row (summed afterwards)
            else atomicAdd(out + n + e, a);
        }
    }
}

// out[t] = wte[ids[t]] + wpe[pos[t]]  (pos null: position = t % S); one wave per row
__global__ __launch_bounds__(256) void embed_pos_kernel(const long* __restrict__ ids, const float* __restrict__ wte,
                                                        const float* __restrict__ wpe, const int* __restrict__ pos,
                                                        float* __restrict__ out, long T, int S, int d, long V, int P,
                                                        int* __restrict__ err) {
    const int lane = threadIdx.x & 63;
    for (long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6); t < T; t += (long)gridDim.x * 4) {
        long id = ids[t];
        int p = pos ? pos[t] : (int)(t % S);
        if (id < 0 || id >= V || p < 0 || p >= P) {
            if (lane == 0 && err) atomicOr(err, 1);
            id = id < 0 ? 0 : (id >= V ? V - 1 : id);
            p = p < 0 ? 0 : (p >= P ? P - 1 : p);
        }
        for (int c = lane * 4; c < d; c += 256)
            st4(out + t * d + c, ld4(wte + id * d + c) + ld4(wpe + (long)p * d + c));
    }
}

inline int ln_grid(long T, int cap) {
    long b = (T + 3) / 4;
    return (int)(b < cap ? (b < 1 ? 1 : b) : cap);
}

}  // namespace

#define DISPATCH_VPL(d, CALL)                                                             \
    if ((d) <= 512) { constexpr int VPL = 2; CALL; }                                      \
    else { constexpr int VPL = 8; CALL; }

extern "C" int mgen_layernorm_fwd(const float* x, const float* w, const float* b, void* y, float* mean, float* rstd,
                                  long T, int d, float eps, int out_bf16, hipStream_t stream) {
    MGEN_REQUIRE(x && w && b && y, "mgen_layernorm_fwd: null pointer");
    MGEN_REQUIRE(T > 0 && d > 0 && d % 4 == 0 && d <= 2048, "mgen_layernorm_fwd: need 0<d<=2048, d%%4==0 (d=%d)", d);
    const dim3 grid(ln_grid(T, 2048)), block(256);
    DISPATCH_VPL(d, {
        if (out_bf16)
            hipLaunchKernelGGL((layernorm_fwd_kernel<VPL, bf16>), grid, block, 0, stream, x, w, b, (bf16*)y, mean, rstd,
                               T, d, eps);
        else
            hipLaunchKernelGGL((layernorm_fwd_kernel<VPL, float>), grid, block, 0, stream, x, w, b, (float*)y, mean,
                               rstd, T, d, eps);
    });
    return mgen_check_launch("mgen_layernorm_fwd");
}

extern "C" int mgen_layernorm_bwd(const void* dy, const float* x, const float* w, const float* mean,
                                  const float* rstd, float* dres, void* dres_lp, float* dw, float* db, long T, int d,
                                  int accumulate, int dy_bf16, hipStream_t stream) {
    MGEN_REQUIRE(dy && x && w && mean && rstd && dres && dw && db, "mgen_layernorm_bwd: null pointer");
    MGEN_REQUIRE(T > 0 && d > 0 && d % 4 == 0 && d <= 2048, "mgen_layernorm_bwd: need 0<d<=2048, d%%4==0 (d=%d)", d);
    const dim3 grid(ln_grid(T, 512)), block(256);
    static_assert(512 <= MGEN_DW_MAXB, "a launch's workgroups need a partial row each");
    const int slot = 2 * d <= MGEN_DW_ROW ? mgen_dw_slot(stream) : -1;
    float* rows = mgen_dw_rows(slot);
    DISPATCH_VPL(d, {
        if (dy_bf16)
            hipLaunchKernelGGL((layernorm_bwd_kernel<VPL, bf16>), grid, block, 0, stream, (const bf16*)dy, x, w, mean,
                               rstd, dres, (bf16*)dres_lp, dw, db, T, d, accumulate, rows);
        else
            hipLaunchKernelGGL((layernorm_bwd_kernel<VPL, float>), grid, block, 0, stream, (const float*)dy, x, w,
                               mean, rstd, dres, (float*)dres_lp, dw, db, T, d, accumulate, rows);
    });
    if (rows) {
        mgen_dw_reduce(dw, d, (int)grid.x, slot, 0, stream);
        mgen_dw_reduce(db, d, (int)grid.x, slot, d, stream);
    }
    return mgen_check_launch("mgen_layernorm_bwd");
}

extern "C" int mgen_gelu_new_fwd(const void* h, void* act, long n, hipStream_t stream) {
    MGEN_REQUIRE(h && act && n > 0 && n % 8 == 0, "mgen_gelu_new_fwd: bad arguments (n must be a multiple of 8)");
    long nb = (n / 8 + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL((gelu_kernel<false>), dim3((unsigned)nb), dim3(256), 0, stream, (const bf16*)h,
                       (const bf16*)nullptr, (bf16*)act, n / 8);
    return mgen_check_launch("mgen_gelu_new_fwd");
}

extern "C" int mgen_gelu_new_bwd(const void* h, const void* dact, void* dh, long n, hipStream_t stream) {
    MGEN_REQUIRE(h && dact && dh && n > 0 && n % 8 == 0, "mgen_gelu_new_bwd: bad arguments (n must be a multiple of 8)");
    long nb = (n / 8 + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL((gelu_kernel<true>), dim3((unsigned)nb), dim3(256), 0, stream, (const bf16*)h,
                       (const bf16*)dact, (bf16*)dh, n / 8);
    return mgen_check_launch("mgen_gelu_new_bwd");
}

// Token + learned position embedding (GPT2Model.forward: inputs_embeds + position_embeds, transformers
// models/gpt2/modeling_gpt2.py). ids int64 [T]; pos int32 [T] or null (position = t % S); tables fp32 [V][d], [P][d].
// An out-of-range id or position sets *err_flag and is clamped.
extern "C" int mgen_embed_pos_fwd(const long* ids, const float* wte, const float* wpe, const int* pos, float* out,
                                  long T, int S, int d, long V, int P, int* err_flag, hipStream_t stream) {
    MGEN_REQUIRE(ids && wte && wpe && out && T > 0 && S > 0 && d > 0 && d % 4 == 0 && V > 0 && P > 0,
                 "mgen_embed_pos_fwd: bad arguments");
    hipLaunchKernelGGL(embed_pos_kernel, dim3(ln_grid(T, 8192)), dim3(256), 0, stream, ids, wte, wpe, pos, out, T, S,
                       d, V, P, err_flag);
    return mgen_check_launch("mgen_embed_pos_fwd");
}

// out[n] += sum over the T rows of x[t][n]: the bias gradient of a linear layer from its bf16 output gradient.
extern "C" int mgen_colsum_bf16(const void* x, long ld, float* out, long T, int N, hipStream_t stream) {
    MGEN_REQUIRE(x && out && T > 0 && N > 0 && N % 8 == 0 && ld % 8 == 0 && ld >= N,
                 "mgen_colsum_bf16: bad arguments (N and ld must be multiples of 8)");
    const int gx = (N / 8 + 31) / 32;
    long gy = 1024 / gx;
    if (gy < 1) gy = 1;
    long rows = (T + gy - 1) / gy;
    if (rows < 64) rows = 64;
    gy = (T + rows - 1) / rows;
    const int slot = (N <= MGEN_DW_ROW && gy <= MGEN_DW_MAXB) ? mgen_dw_slot(stream) : -1;
    float* part = mgen_dw_rows(slot);
    hipLaunchKernelGGL(colsum_kernel, dim3(gx, (unsigned)gy), dim3(256), 0, stream, (const bf16*)x, ld, out, T, N, rows,
                       part);
    if (part) mgen_dw_reduce(out, N, (int)gy, slot, 0, stream);
    return mgen_check_launch("mgen_colsum_bf16");
}
