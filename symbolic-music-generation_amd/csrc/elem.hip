// HBM-bound elementwise and gather/scatter kernels of the decoder: SwiGLU gate, token-embedding gather and its
// scatter-add backward, and dtype conversion. 16-byte accesses per lane, grid-stride.
//
// Reference: Qwen3MLP.forward, transformers models/qwen3/modeling_qwen3.py:81-83 (down(silu(gate(x)) * up(x)), silu and
// the product evaluated in bf16 under autocast); nn.Embedding at models/qwen3/modeling_qwen3.py:352,381 and its
// autograd backward (index_add into the fp32 table gradient).
#include "common.h"

namespace {

// (v_rcp_f32, 1 ulp: a correctly rounded fp32 division costs ten instructions per element, which the GEMM epilogue
// that shares this formula pays beside its MFMAs; the result is rounded to bf16 anyway)
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

// gu: [T, 2I] = [gate | up] per row; act: [T, I]
template <bool BF16ROUND>
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16* __restrict__ gu, bf16* __restrict__ act, long T,
                                                         int I) {
    const long nvec = T * (I / 8);
    const int vpr = I / 8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        const long t = i / vpr;
        const int c = (int)(i % vpr) * 8;
        const bf16x8 g = *reinterpret_cast<const bf16x8*>(gu + t * 2 * I + c);
        const bf16x8 u = *reinterpret_cast<const bf16x8*>(gu + t * 2 * I + I + c);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float gf = (float)g[e];
            float s = gf * sigmoidf_(gf);
            if (BF16ROUND) s = round_bf(s);  // the reference materialises silu(gate) in bf16 before the product
            o[e] = (bf16)(s * (float)u[e]);
        }
        *reinterpret_cast<bf16x8*>(act + t * I + c) = o;
    }
}

// dgu [T, 2I] = [dgate | dup] from dact [T, I] and the saved gu
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16* __restrict__ gu, const bf16* __restrict__ dact,
                                                         bf16* __restrict__ dgu, long T, int I, long ldgu, long lddact,
                                                         long lddgu) {
    const long nvec = T * (I / 8);
    const int vpr = I / 8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        const long t = i / vpr;
        const int c = (int)(i % vpr) * 8;
        const bf16x8 g = ld_stream(reinterpret_cast<const bf16x8*>(gu + t * ldgu + c));
        const bf16x8 u = ld_stream(reinterpret_cast<const bf16x8*>(gu + t * ldgu + I + c));
        const bf16x8 d = ld_stream(reinterpret_cast<const bf16x8*>(dact + t * lddact + c));
        bf16x8 dg, du;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float gf = (float)g[e], uf = (float)u[e], df = (float)d[e];
            const float sg = sigmoidf_(gf);
            const float silu = gf * sg;
            du[e] = (bf16)(df * silu);
            dg[e] = (bf16)(df * uf * (sg * (1.f + gf * (1.f - sg))));
        }
        st_stream(reinterpret_cast<bf16x8*>(dgu + t * lddgu + c), dg);
        st_stream(reinterpret_cast<bf16x8*>(dgu + t * lddgu + I + c), du);
    }
}

// out[t, :] = table[ids[t], :] (fp32 rows; one wave per row, float4 per lane per step)
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const long* __restrict__ ids, const float* __restrict__ table,
                                                            float* __restrict__ out, long T, int d, long V,
                                                            int* __restrict__ err) {
    const int lane = threadIdx.x & 63;
    for (long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6); t < T; t += (long)gridDim.x * 4) {
        long id = ids[t];
        if (id < 0 || id >= V) {
            if (lane == 0 && err) atomicOr(err, 1);
            id = 0;
        }
        const float* src = table + id * d;
        float* dst = out + t * d;
        for (int c = lane * 4; c < d; c += 256) *reinterpret_cast<f32x4*>(dst + c) = *reinterpret_cast<const f32x4*>(src + c);
    }
}

// dtable[ids[t], :] += dout[t, :]; each wave instruction adds 256 contiguous bytes of one row
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const long* __restrict__ ids, const float* __restrict__ dout,
                                                            float* __restrict__ dtable, long T, int d, long V) {
    const int lane = threadIdx.x & 63;
    for (long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6); t < T; t += (long)gridDim.x * 4) {
        const long id = ids[t];
        if (id < 0 || id >= V) continue;
        const float* src = dout + t * d;
        float* dst = dtable + id * d;
        for (int c = lane; c < d; c += 64) atomicAdd(dst + c, src[c]);
    }
}

template <typename A, typename Bt>
__global__ __launch_bounds__(256) void cast_kernel(const A* __restrict__ in, Bt* __restrict__ out, long n) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        if (i + 3 < n) {
#pragma unroll
            for (int e = 0; e < 4; ++e) out[i + e] = (Bt)(float)in[i + e];
        } else {
            for (long k = i; k < n; ++k) out[k] = (Bt)(float)in[k];
        }
    }
}

inline unsigned grid_for(long work_items, int per_block, int cap) {
    long b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}

}  // namespace

extern "C" int mgen_swiglu_fwd(const void* gu, void* act, long T, int I, int round_silu, hipStream_t stream) {
    MGEN_REQUIRE(gu && act && T > 0 && I > 0 && I % 8 == 0, "mgen_swiglu_fwd: bad arguments (I must be a multiple of 8)");
    const unsigned g = grid_for(T * (I / 8), 256, 8192);
    if (round_silu)
        hipLaunchKernelGGL((swiglu_fwd_kernel<true>), dim3(g), dim3(256), 0, stream, (const bf16*)gu, (bf16*)act, T, I);
    else
        hipLaunchKernelGGL((swiglu_fwd_kernel<false>), dim3(g), dim3(256), 0, stream, (const bf16*)gu, (bf16*)act, T, I);
    return mgen_check_launch("mgen_swiglu_fwd");
}

extern "C" int mgen_swiglu_bwd(const void* gu, const void* dact, void* dgu, long T, int I, hipStream_t stream) {
    MGEN_REQUIRE(gu && dact && dgu && T > 0 && I > 0 && I % 8 == 0, "mgen_swiglu_bwd: bad arguments");
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for(T * (I / 8), 256, 8192)), dim3(256), 0, stream,
                       (const bf16*)gu, (const bf16*)dact, (bf16*)dgu, T, I, 2L * I, (long)I, 2L * I);
    return mgen_check_launch("mgen_swiglu_bwd");
}

// the same with row pitches (elements) of gu [T, 2I], dact [T, I] and dgu [T, 2I]
extern "C" int mgen_swiglu_bwd_ld(const void* gu, long ldgu, const void* dact, long lddact, void* dgu, long lddgu, long T,
                                  int I, hipStream_t stream) {
    MGEN_REQUIRE(gu && dact && dgu && T > 0 && I > 0 && I % 8 == 0, "mgen_swiglu_bwd_ld: bad arguments");
    MGEN_REQUIRE(ldgu >= 2L * I && lddact >= I && lddgu >= 2L * I && ldgu % 8 == 0 && lddact % 8 == 0 && lddgu % 8 == 0,
                 "mgen_swiglu_bwd_ld: row pitches must cover the rows and be multiples of 8 elements");
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for(T * (I / 8), 256, 8192)), dim3(256), 0, stream,
                       (const bf16*)gu, (const bf16*)dact, (bf16*)dgu, T, I, ldgu, lddact, lddgu);
    return mgen_check_launch("mgen_swiglu_bwd_ld");
}

extern "C" int mgen_embedding_fwd(const long* ids, const float* table, float* out, long T, int d, long V, int* err_flag,
                                  hipStream_t stream) {
    MGEN_REQUIRE(ids && table && out && T > 0 && d > 0 && d % 4 == 0 && V > 0, "mgen_embedding_fwd: bad arguments");
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3(grid_for(T, 4, 8192)), dim3(256), 0, stream, ids, table, out, T, d, V,
                       err_flag);
    return mgen_check_launch("mgen_embedding_fwd");
}

extern "C" int mgen_embedding_bwd(const long* ids, const float* dout, float* dtable, long T, int d, long V,
                                  hipStream_t stream) {
    MGEN_REQUIRE(ids && dout && dtable && T > 0 && d > 0 && V > 0, "mgen_embedding_bwd: bad arguments");
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(grid_for(T, 4, 8192)), dim3(256), 0, stream, ids, dout, dtable, T, d, V);
    return mgen_check_launch("mgen_embedding_bwd");
}

// out[b][c][r] = in[b][r][c] for `nbatch` matrices of R x C 16-bit elements (64 x 64 tiles through LDS, 128-byte
// row segments on both sides; the 65-element pitch keeps the column reads on different banks)
__global__ __launch_bounds__(256) void transpose16_kernel(const unsigned short* __restrict__ in,
                                                          unsigned short* __restrict__ out, int R, int C, long ldin,
                                                          long ldout, long bs_in, long bs_out) {
    __shared__ unsigned short tile[64][65];
    in += (long)blockIdx.z * bs_in;
    out += (long)blockIdx.z * bs_out;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4)
        tile[i][tx] = (r0 + i < R && c0 + tx < C) ? in[(long)(r0 + i) * ldin + c0 + tx] : (unsigned short)0;
    __syncthreads();
    for (int i = ty; i < 64; i += 4)
        if (c0 + i < C && r0 + tx < R) out[(long)(c0 + i) * ldout + r0 + tx] = tile[tx][i];
}

// Transposed copies of bf16 matrices (weights whose input-gradient product wants the reduction index contiguous):
// out[b] [C][R] = in[b] [R][C]^T, b = 0 .. nbatch-1, matrices bs_in / bs_out elements apart.
extern "C" int mgen_transpose_bf16(const void* in, void* out, int R, int C, long ldin, long ldout, int nbatch,
                                   long bs_in, long bs_out, hipStream_t stream) {
    MGEN_REQUIRE(in && out && R > 0 && C > 0 && nbatch > 0 && nbatch < 65536 && ldin >= C && ldout >= R,
                 "mgen_transpose_bf16: bad arguments");
    hipLaunchKernelGGL(transpose16_kernel, dim3((C + 63) / 64, (R + 63) / 64, nbatch), dim3(256), 0, stream,
                       (const unsigned short*)in, (unsigned short*)out, R, C, ldin, ldout, bs_in, bs_out);
    return mgen_check_launch("mgen_transpose_bf16");
}

// kind: 0 = fp32 -> bf16, 1 = bf16 -> fp32
extern "C" int mgen_cast(const void* in, void* out, long n, int kind, hipStream_t stream) {
    MGEN_REQUIRE(in && out && n > 0, "mgen_cast: bad arguments");
    const unsigned g = grid_for(n, 1024, 8192);
    if (kind == 0)
        hipLaunchKernelGGL((cast_kernel<float, bf16>), dim3(g), dim3(256), 0, stream, (const float*)in, (bf16*)out, n);
    else if (kind == 1)
        hipLaunchKernelGGL((cast_kernel<bf16, float>), dim3(g), dim3(256), 0, stream, (const bf16*)in, (float*)out, n);
    else {
        mgen_set_error("mgen_cast: unknown kind %d", kind);
        return MGEN_EINVAL;
    }
    return mgen_check_launch("mgen_cast");
}
