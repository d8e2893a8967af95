// Kernels of the fp32 parity mode: the same network evaluated without any bf16 rounding, to check the kernels
// against the fp32 reference at 1e-4. Not a speed path.
//
//  * mgen_split_bf16: x (fp32) -> hi = bf16(x), lo = bf16(x - hi). A product of two fp32 matrices is then evaluated on
//    the bf16 MFMA GEMM as hi*hi + hi*lo + lo*hi with fp32 accumulation (relative error about 2^-16 per term).
//  * SwiGLU in fp32 (transformers models/qwen3/modeling_qwen3.py:81-83 without autocast).
//  * Causal attention in fp32 with plain FMAs: softmax(q k^T * scale) v per head, the eager formula of
//    transformers models/qwen3/modeling_qwen3.py:185-207 (softmax in fp32), and its gradient
//    dS = P * (dP - rowsum(dP * P)) * scale, dq = dS k, dk = dS^T q, dv = P^T dO.
//    One wavefront per query row (forward, dq) or per key row (dk, dv).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ x, bf16* __restrict__ hi,
                                                    bf16* __restrict__ lo, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        bf16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            h[e] = (bf16)v[e];
            l[e] = (bf16)(v[e] - (float)h[e]);
        }
        reinterpret_cast<bf16x4*>(hi)[i] = h;
        reinterpret_cast<bf16x4*>(lo)[i] = l;
    }
}

__device__ __forceinline__ float sigmoid_acc(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void swiglu_f32_fwd_kernel(const float* __restrict__ gu, float* __restrict__ act,
                                                             long T, int I) {
    const long n = T * I;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long t = i / I;
        const int c = (int)(i % I);
        const float g = gu[t * 2 * I + c], u = gu[t * 2 * I + I + c];
        act[i] = g * sigmoid_acc(g) * u;
    }
}

__global__ __launch_bounds__(256) void swiglu_f32_bwd_kernel(const float* __restrict__ gu,
                                                             const float* __restrict__ dact,
                                                             float* __restrict__ dgu, long T, int I) {
    const long n = T * I;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long t = i / I;
        const int c = (int)(i % I);
        const float g = gu[t * 2 * I + c], u = gu[t * 2 * I + I + c], d = dact[i];
        const float sg = sigmoid_acc(g);
        dgu[t * 2 * I + I + c] = d * g * sg;
        dgu[t * 2 * I + c] = d * u * (sg * (1.f + g * (1.f - sg)));
    }
}

struct A32 {
    const float *q, *k, *v, *o, *dout;
    float *out, *lse, *delta, *dq, *dk, *dv;
    long sq, sk, sv, so, sdo, sdq, sdk, sdv;  // token strides in elements
    int B, S, H, KVH;
    float scale;
};

// dot of the row held in LDS (wave-private, D floats) with a global row
template <int D>
__device__ __forceinline__ float dot_lds(const float* __restrict__ lds_row, const float* __restrict__ g) {
    float s = 0.f;
#pragma unroll 8
    for (int d = 0; d < D; d += 4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(lds_row + d);
        const f32x4 b = *reinterpret_cast<const f32x4*>(g + d);
        s += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
    }
    return s;
}

// ---- forward: one wave per (b, h, i) ---------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void attn32_fwd_kernel(A32 a) {
    constexpr int DPL = D / 64;
    __shared__ float qs[4][D];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + w;
    const long nrows = (long)a.B * a.H * a.S;
    if (row >= nrows) return;
    const int i = (int)(row % a.S);
    const int h = (int)((row / a.S) % a.H);
    const int b = (int)(row / ((long)a.S * a.H));
    const int kvh = h / (a.H / a.KVH);
    const long tok0 = (long)b * a.S;
    const float* qrow = a.q + (tok0 + i) * a.sq + (long)h * D;
    for (int d = lane; d < D; d += 64) qs[w][d] = qrow[d];
    __builtin_amdgcn_wave_barrier();
    float m = -INFINITY, l = 0.f, acc[DPL];
#pragma unroll
    for (int e = 0; e < DPL; ++e) acc[e] = 0.f;
    for (int j0 = 0; j0 <= i; j0 += 64) {
        const int j = j0 + lane;
        float s = -INFINITY;
        if (j <= i) s = dot_lds<D>(qs[w], a.k + (tok0 + j) * a.sk + (long)kvh * D) * a.scale;
        const float mn = fmaxf(m, wave_max(s));
        const float p = j <= i ? expf(s - mn) : 0.f;
        const float corr = expf(m - mn);
        l = l * corr + wave_sum(p);
#pragma unroll
        for (int e = 0; e < DPL; ++e) acc[e] *= corr;
        const int cnt = min(64, i - j0 + 1);
        for (int jj = 0; jj < cnt; ++jj) {
            const float pj = __shfl(p, jj, 64);
            const float* vr = a.v + (tok0 + j0 + jj) * a.sv + (long)kvh * D;
#pragma unroll
            for (int e = 0; e < DPL; ++e) acc[e] += pj * vr[lane + 64 * e];
        }
        m = mn;
    }
    float* orow = a.out + (tok0 + i) * a.so + (long)h * D;
#pragma unroll
    for (int e = 0; e < DPL; ++e) orow[lane + 64 * e] = acc[e] / l;
    if (lane == 0) a.lse[((long)b * a.H + h) * a.S + i] = m + logf(l);
}

// ---- delta[b,h,i] = sum_d dO * O ---------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void attn32_delta_kernel(A32 a) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + w;
    if (row >= (long)a.B * a.H * a.S) return;
    const int i = (int)(row % a.S);
    const int h = (int)((row / a.S) % a.H);
    const int b = (int)(row / ((long)a.S * a.H));
    const float* o = a.o + ((long)b * a.S + i) * a.so + (long)h * D;
    const float* d = a.dout + ((long)b * a.S + i) * a.sdo + (long)h * D;
    float s = 0.f;
    for (int e = lane; e < D; e += 64) s += o[e] * d[e];
    s = wave_sum(s);
    if (lane == 0) a.delta[row] = s;
}

// ---- dq: one wave per (b, h, i) --------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void attn32_dq_kernel(A32 a) {
    constexpr int DPL = D / 64;
    __shared__ float qs[4][D], ds_[4][D];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + w;
    if (row >= (long)a.B * a.H * a.S) return;
    const int i = (int)(row % a.S);
    const int h = (int)((row / a.S) % a.H);
    const int b = (int)(row / ((long)a.S * a.H));
    const int kvh = h / (a.H / a.KVH);
    const long tok0 = (long)b * a.S;
    const float* qrow = a.q + (tok0 + i) * a.sq + (long)h * D;
    const float* drow = a.dout + (tok0 + i) * a.sdo + (long)h * D;
    for (int d = lane; d < D; d += 64) { qs[w][d] = qrow[d]; ds_[w][d] = drow[d]; }
    __builtin_amdgcn_wave_barrier();
    const float lse = a.lse[row], delta = a.delta[row];
    float acc[DPL];
#pragma unroll
    for (int e = 0; e < DPL; ++e) acc[e] = 0.f;
    for (int j0 = 0; j0 <= i; j0 += 64) {
        const int j = j0 + lane;
        float dS = 0.f;
        if (j <= i) {
            const float s = dot_lds<D>(qs[w], a.k + (tok0 + j) * a.sk + (long)kvh * D) * a.scale;
            const float dP = dot_lds<D>(ds_[w], a.v + (tok0 + j) * a.sv + (long)kvh * D);
            dS = expf(s - lse) * (dP - delta) * a.scale;
        }
        const int cnt = min(64, i - j0 + 1);
        for (int jj = 0; jj < cnt; ++jj) {
            const float g = __shfl(dS, jj, 64);
            const float* kr = a.k + (tok0 + j0 + jj) * a.sk + (long)kvh * D;
#pragma unroll
            for (int e = 0; e < DPL; ++e) acc[e] += g * kr[lane + 64 * e];
        }
    }
    float* o = a.dq + (tok0 + i) * a.sdq + (long)h * D;
#pragma unroll
    for (int e = 0; e < DPL; ++e) o[lane + 64 * e] = acc[e];
}

// ---- dk, dv: one wave per (b, kvh, j), summing over the query heads of the group ----------------------------------
template <int D>
__global__ __launch_bounds__(256) void attn32_dkv_kernel(A32 a) {
    constexpr int DPL = D / 64;
    __shared__ float ks[4][D], vs[4][D];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + w;
    if (row >= (long)a.B * a.KVH * a.S) return;
    const int j = (int)(row % a.S);
    const int kvh = (int)((row / a.S) % a.KVH);
    const int b = (int)(row / ((long)a.S * a.KVH));
    const long tok0 = (long)b * a.S;
    const float* krow = a.k + (tok0 + j) * a.sk + (long)kvh * D;
    const float* vrow = a.v + (tok0 + j) * a.sv + (long)kvh * D;
    for (int d = lane; d < D; d += 64) { ks[w][d] = krow[d]; vs[w][d] = vrow[d]; }
    __builtin_amdgcn_wave_barrier();
    float dk[DPL], dv[DPL];
#pragma unroll
    for (int e = 0; e < DPL; ++e) dk[e] = dv[e] = 0.f;
    const int rep = a.H / a.KVH;
    for (int r = 0; r < rep; ++r) {
        const int h = kvh * rep + r;
        const float* lse = a.lse + ((long)b * a.H + h) * a.S;
        const float* del = a.delta + ((long)b * a.H + h) * a.S;
        for (int i0 = j; i0 < a.S; i0 += 64) {
            const int i = i0 + lane;
            float p = 0.f, dS = 0.f;
            if (i < a.S) {
                const float s = dot_lds<D>(ks[w], a.q + (tok0 + i) * a.sq + (long)h * D) * a.scale;
                const float dP = dot_lds<D>(vs[w], a.dout + (tok0 + i) * a.sdo + (long)h * D);
                p = expf(s - lse[i]);
                dS = p * (dP - del[i]) * a.scale;
            }
            const int cnt = min(64, a.S - i0);
            for (int ii = 0; ii < cnt; ++ii) {
                const float pi = __shfl(p, ii, 64), gi = __shfl(dS, ii, 64);
                const float* dor = a.dout + (tok0 + i0 + ii) * a.sdo + (long)h * D;
                const float* qr = a.q + (tok0 + i0 + ii) * a.sq + (long)h * D;
#pragma unroll
                for (int e = 0; e < DPL; ++e) {
                    dv[e] += pi * dor[lane + 64 * e];
                    dk[e] += gi * qr[lane + 64 * e];
                }
            }
        }
    }
    float* ok = a.dk + (tok0 + j) * a.sdk + (long)kvh * D;
    float* ov = a.dv + (tok0 + j) * a.sdv + (long)kvh * D;
#pragma unroll
    for (int e = 0; e < DPL; ++e) { ok[lane + 64 * e] = dk[e]; ov[lane + 64 * e] = dv[e]; }
}

inline unsigned grid1d(long items, int per_block, long cap) {
    long b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}

}  // namespace

extern "C" int mgen_split_bf16(const float* x, void* hi, void* lo, long n, hipStream_t stream) {
    MGEN_REQUIRE(x && hi && lo && n > 0 && n % 4 == 0, "mgen_split_bf16: bad arguments (n must be a multiple of 4)");
    hipLaunchKernelGGL(split_kernel, dim3(grid1d(n / 4, 256, 8192)), dim3(256), 0, stream, x, (bf16*)hi, (bf16*)lo,
                       n / 4);
    return mgen_check_launch("mgen_split_bf16");
}

extern "C" int mgen_swiglu_f32_fwd(const float* gu, float* act, long T, int I, hipStream_t stream) {
    MGEN_REQUIRE(gu && act && T > 0 && I > 0, "mgen_swiglu_f32_fwd: bad arguments");
    hipLaunchKernelGGL(swiglu_f32_fwd_kernel, dim3(grid1d(T * I, 256, 8192)), dim3(256), 0, stream, gu, act, T, I);
    return mgen_check_launch("mgen_swiglu_f32_fwd");
}

extern "C" int mgen_swiglu_f32_bwd(const float* gu, const float* dact, float* dgu, long T, int I, hipStream_t stream) {
    MGEN_REQUIRE(gu && dact && dgu && T > 0 && I > 0, "mgen_swiglu_f32_bwd: bad arguments");
    hipLaunchKernelGGL(swiglu_f32_bwd_kernel, dim3(grid1d(T * I, 256, 8192)), dim3(256), 0, stream, gu, dact, dgu, T, I);
    return mgen_check_launch("mgen_swiglu_f32_bwd");
}

#define A32_COMMON_CHECKS(name)                                                                                  \
    MGEN_REQUIRE(B > 0 && S > 0 && H > 0 && KVH > 0 && H % KVH == 0, name ": bad sizes");                        \
    MGEN_REQUIRE(D == 128 || D == 64, name ": head_dim must be 64 or 128 (got %d)", D);                          \
    MGEN_REQUIRE((long)B * H * S < (1L << 33), name ": too many rows")

extern "C" int mgen_attn_f32_fwd(const float* q, const float* k, const float* v, float* out, float* lse, long sq,
                                 long sk, long sv, long so, int B, int S, int H, int KVH, int D, float scale,
                                 hipStream_t stream) {
    MGEN_REQUIRE(q && k && v && out && lse, "mgen_attn_f32_fwd: null pointer");
    A32_COMMON_CHECKS("mgen_attn_f32_fwd");
    MGEN_REQUIRE(sq % 4 == 0 && sk % 4 == 0 && sv % 4 == 0, "mgen_attn_f32_fwd: strides must be multiples of 4");
    A32 a{};
    a.q = q; a.k = k; a.v = v; a.out = out; a.lse = lse; a.sq = sq; a.sk = sk; a.sv = sv; a.so = so;
    a.B = B; a.S = S; a.H = H; a.KVH = KVH; a.scale = scale;
    const unsigned g = (unsigned)(((long)B * H * S + 3) / 4);
    if (D == 128) hipLaunchKernelGGL(attn32_fwd_kernel<128>, dim3(g), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(attn32_fwd_kernel<64>, dim3(g), dim3(256), 0, stream, a);
    return mgen_check_launch("mgen_attn_f32_fwd");
}

extern "C" int mgen_attn_f32_bwd(const float* q, const float* k, const float* v, const float* out, const float* dout,
                                 const float* lse, float* delta, float* dq, float* dk, float* dv, long sq, long sk,
                                 long sv, long so, long sdo, long sdq, long sdk, long sdv, int B, int S, int H, int KVH,
                                 int D, float scale, hipStream_t stream) {
    MGEN_REQUIRE(q && k && v && out && dout && lse && delta && dq && dk && dv, "mgen_attn_f32_bwd: null pointer");
    A32_COMMON_CHECKS("mgen_attn_f32_bwd");
    MGEN_REQUIRE(sq % 4 == 0 && sk % 4 == 0 && sv % 4 == 0 && sdo % 4 == 0,
                 "mgen_attn_f32_bwd: strides must be multiples of 4");
    A32 a{};
    a.q = q; a.k = k; a.v = v; a.o = out; a.dout = dout; a.lse = const_cast<float*>(lse); a.delta = delta;
    a.dq = dq; a.dk = dk; a.dv = dv;
    a.sq = sq; a.sk = sk; a.sv = sv; a.so = so; a.sdo = sdo; a.sdq = sdq; a.sdk = sdk; a.sdv = sdv;
    a.B = B; a.S = S; a.H = H; a.KVH = KVH; a.scale = scale;
    const unsigned gq = (unsigned)(((long)B * H * S + 3) / 4), gk = (unsigned)(((long)B * KVH * S + 3) / 4);
    if (D == 128) {
        hipLaunchKernelGGL(attn32_delta_kernel<128>, dim3(gq), dim3(256), 0, stream, a);
        hipLaunchKernelGGL(attn32_dq_kernel<128>, dim3(gq), dim3(256), 0, stream, a);
        hipLaunchKernelGGL(attn32_dkv_kernel<128>, dim3(gk), dim3(256), 0, stream, a);
    } else {
        hipLaunchKernelGGL(attn32_delta_kernel<64>, dim3(gq), dim3(256), 0, stream, a);
        hipLaunchKernelGGL(attn32_dq_kernel<64>, dim3(gq), dim3(256), 0, stream, a);
        hipLaunchKernelGGL(attn32_dkv_kernel<64>, dim3(gk), dim3(256), 0, stream, a);
    }
    return mgen_check_launch("mgen_attn_f32_bwd");
}

// ---------------------------------------------------------------------------------------------------
// GPT-2 blocks in the fp32 parity mode: gelu_new, the Conv1D bias and its gradient on fp32 activations
// (transformers activations.py NewGELUActivation; pytorch_utils.py Conv1D.forward = addmm(bias, x, W)).
// ---------------------------------------------------------------------------------------------------
namespace {

constexpr float kGeluA32 = 0.7978845608028654f;  // sqrt(2/pi)
constexpr float kGeluB32 = 0.044715f;

template <bool BWD>
__global__ __launch_bounds__(256) void gelu_f32_kernel(const float* __restrict__ h, const float* __restrict__ dact,
                                                       float* __restrict__ out, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 x = reinterpret_cast<const f32x4*>(h)[i];
        f32x4 o;
        f32x4 dv = {0.f, 0.f, 0.f, 0.f};
        if (BWD) dv = reinterpret_cast<const f32x4*>(dact)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float u = kGeluA32 * (x[e] + kGeluB32 * x[e] * x[e] * x[e]);
            const float t = tanhf(u);
            if (BWD) {
                const float du = kGeluA32 * (1.0f + 3.0f * kGeluB32 * x[e] * x[e]);
                o[e] = dv[e] * (0.5f * (1.0f + t) + 0.5f * x[e] * (1.0f - t * t) * du);
            } else {
                o[e] = 0.5f * x[e] * (1.0f + t);
            }
        }
        reinterpret_cast<f32x4*>(out)[i] = o;
    }
}

// y[t][n] += bias[n]
__global__ __launch_bounds__(256) void bias_add_f32_kernel(float* __restrict__ y, long ld, const float* __restrict__ bias,
                                                           long T, int N) {
    const int nv = N / 4;
    const long total = T * nv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long t = i / nv;
        const int c = (int)(i % nv) * 4;
        f32x4* p = reinterpret_cast<f32x4*>(y + t * ld + c);
        *p = *p + *reinterpret_cast<const f32x4*>(bias + c);
    }
}

// out[n] += sum_t x[t][n]: 256 threads = 64 column groups of 4 x 4 row lanes, a strip of rows per block
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ x, long ld, float* __restrict__ out,
                                                         long T, int N, long rows_per_block) {
    __shared__ float red[4][64][4];
    const int cg = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int n = (blockIdx.x * 64 + cg) * 4;
    const long t0 = (long)blockIdx.y * rows_per_block;
    const long t1 = t0 + rows_per_block < T ? t0 + rows_per_block : T;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (n < N)
        for (long t = t0 + rl; t < t1; t += 4) s += *reinterpret_cast<const f32x4*>(x + t * ld + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) red[rl][cg][e] = s[e];
    __syncthreads();
    if (rl == 0 && n < N) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            atomicAdd(out + n + e, red[0][cg][e] + red[1][cg][e] + red[2][cg][e] + red[3][cg][e]);
    }
}

}  // namespace

extern "C" int mgen_gelu_new_f32_fwd(const float* h, float* act, long n, hipStream_t stream) {
    MGEN_REQUIRE(h && act && n > 0 && n % 4 == 0, "mgen_gelu_new_f32_fwd: bad arguments (n must be a multiple of 4)");
    long nb = (n / 4 + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL((gelu_f32_kernel<false>), dim3((unsigned)nb), dim3(256), 0, stream, h, (const float*)nullptr,
                       act, n / 4);
    return mgen_check_launch("mgen_gelu_new_f32_fwd");
}

extern "C" int mgen_gelu_new_f32_bwd(const float* h, const float* dact, float* dh, long n, hipStream_t stream) {
    MGEN_REQUIRE(h && dact && dh && n > 0 && n % 4 == 0,
                 "mgen_gelu_new_f32_bwd: bad arguments (n must be a multiple of 4)");
    long nb = (n / 4 + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL((gelu_f32_kernel<true>), dim3((unsigned)nb), dim3(256), 0, stream, h, dact, dh, n / 4);
    return mgen_check_launch("mgen_gelu_new_f32_bwd");
}

// y [T][ld] fp32: y[t][n] += bias[n] for n < N (N % 4 == 0, ld % 4 == 0)
extern "C" int mgen_bias_add_f32(float* y, long ld, const float* bias, long T, int N, hipStream_t stream) {
    MGEN_REQUIRE(y && bias && T > 0 && N > 0 && N % 4 == 0 && ld % 4 == 0 && ld >= N,
                 "mgen_bias_add_f32: bad arguments (N and ld must be multiples of 4)");
    long nb = (T * (N / 4) + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(bias_add_f32_kernel, dim3((unsigned)nb), dim3(256), 0, stream, y, ld, bias, T, N);
    return mgen_check_launch("mgen_bias_add_f32");
}

// out[n] += sum over the T rows of x[t][n] (x fp32 [T][ld])
extern "C" int mgen_colsum_f32(const float* x, long ld, float* out, long T, int N, hipStream_t stream) {
    MGEN_REQUIRE(x && out && T > 0 && N > 0 && N % 4 == 0 && ld % 4 == 0 && ld >= N,
                 "mgen_colsum_f32: bad arguments (N and ld must be multiples of 4)");
    const int gx = (N / 4 + 63) / 64;
    long gy = 1024 / gx;
    if (gy < 1) gy = 1;
    long rows = (T + gy - 1) / gy;
    if (rows < 64) rows = 64;
    gy = (T + rows - 1) / rows;
    hipLaunchKernelGGL(colsum_f32_kernel, dim3(gx, (unsigned)gy), dim3(256), 0, stream, x, ld, out, T, N, rows);
    return mgen_check_launch("mgen_colsum_f32");
}
