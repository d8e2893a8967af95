// Projections of the one-token decode step: C[M, N] = A[M, K] W[N, K]^T with few token rows (M = number of streams,
// tens to a few hundred) and small weights (1-4 MB). Such a product moves almost nothing; what it costs is the number of
// dependent memory round trips between its launch and its last store, each of which stretches when the attention
// kernel of another batch part is saturating HBM beside it. So this kernel makes exactly one: a workgroup owns
// 64 token rows x 64 output columns x 512 reduction indices, every thread requests all 32 of its 16-byte pieces of
// the two operand tiles (128 KiB per workgroup) before it waits for the first, the tiles go to LDS (rows of 1 KiB,
// 16-byte chunks XOR-swizzled by the row so that the 32 rows an MFMA operand read touches fall on different banks),
// one barrier, 32 v_mfma_f32_32x32x16_bf16 per wave, store.
//
//   out bf16   (K == 512): C[m][n] rounded to bf16                                 QKV and gate/up projections
//   out slabs  (K = z * 512): fp32 partial products slabs[z][M][N], summed by mgen_slab_resid_rmsnorm
//                                                                                   o_proj and down_proj
//   a_swiglu: the A operand is silu(gate) * up formed from gu[M][2K] while the tile is loaded (gate rounded to bf16
//             after the silu, the product rounded to bf16: the arithmetic of mgen_swiglu_fwd), so the gated activation
//             of the MLP is never written
//
// Reference: the four nn.Linear calls of a decoder layer with q_len = 1 (transformers models/qwen3/modeling_qwen3.py:
// 252-254, 279, 81-83) under GenerationMixin._sample (generation/utils.py:2876-2937).
#include "common.h"

namespace {

constexpr int SK_KC = 512;            // reduction indices per workgroup
constexpr int SK_TM = 64, SK_TN = 64; // token rows and output columns per workgroup
constexpr int SK_ROWB = SK_KC * 2;    // bytes per LDS row
constexpr int SK_LDS = (SK_TM + SK_TN) * SK_ROWB;

__device__ __forceinline__ int sk_off(int row, int chunk) { return row * SK_ROWB + ((chunk ^ (row & 15)) << 4); }

template <bool SLABS, bool ASW>
__global__ __launch_bounds__(256, 1) void gemm_skinny_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W,
                                                             void* __restrict__ Cout, int M, int N, int K, long lda,
                                                             long ldw, long ldc, long slab_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                          // token tile   [64][512] bf16
    char* sW = smem + SK_TM * SK_ROWB;        // weight tile  [64][512] bf16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * SK_TM, n0 = blockIdx.y * SK_TN, z = blockIdx.z;
    const int k0 = z * SK_KC;
    const int ch = tid & 63;                  // 16-byte chunk of the row this thread moves
    const int r0 = tid >> 6;                  // first of its 16 rows (rows r0, r0 + 4, ...)

    // ---- every load of the workgroup in flight at once
    u32x4 wv[16], av[16], uv[ASW ? 16 : 1];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int r = r0 + 4 * j;
        wv[j] = *reinterpret_cast<const u32x4*>(W + (long)(n0 + r) * ldw + k0 + ch * 8);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int r = min(m0 + r0 + 4 * j, M - 1);        // rows past M repeat the last row (never stored)
        av[j] = *reinterpret_cast<const u32x4*>(A + (long)r * lda + k0 + ch * 8);
        if (ASW) uv[j] = *reinterpret_cast<const u32x4*>(A + (long)r * lda + K + k0 + ch * 8);
    }
    __builtin_amdgcn_sched_barrier(0);   // (the compiler would otherwise keep about 12 loads in flight and interleave)
#pragma unroll
    for (int j = 0; j < 16; ++j) *reinterpret_cast<u32x4*>(sW + sk_off(r0 + 4 * j, ch)) = wv[j];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        u32x4 v = av[j];
        if (ASW) {
            const bf16x8 g = __builtin_bit_cast(bf16x8, av[j]);
            const bf16x8 u = __builtin_bit_cast(bf16x8, uv[j]);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float gf = (float)g[e];
                const float s = round_bf(gf * __builtin_amdgcn_rcpf(1.f + __expf(-gf)));   // the formula of mgen_swiglu_fwd
                o[e] = (bf16)(s * (float)u[e]);
            }
            v = __builtin_bit_cast(u32x4, o);
        }
        *reinterpret_cast<u32x4*>(sA + sk_off(r0 + 4 * j, ch)) = v;
    }
    __syncthreads();

    // ---- wave (wn, wm): 32 weight rows x 32 token rows; D[n][m] so that a lane ends with 4 consecutive n of one token
    const int wn = wave & 1, wm = wave >> 1;
    const int li = lane & 31, g = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int rowW = wn * 32 + li, rowA = wm * 32 + li;
#pragma unroll 8
    for (int s = 0; s < SK_KC / 16; ++s) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(sW + sk_off(rowW, 2 * s + g));
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(sA + sk_off(rowA, 2 * s + g));
        acc = mfma32(a, b, acc);
    }
    const int m = m0 + wm * 32 + li;
    if (m >= M) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * 32 + q * 8 + g * 4;
        if (SLABS) {
            float* dst = reinterpret_cast<float*>(Cout) + (long)z * slab_stride + (long)m * ldc + n;
            *reinterpret_cast<f32x4*>(dst) = f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
        } else {
            bf16* dst = reinterpret_cast<bf16*>(Cout) + (long)m * ldc + n;
            const bf16x4 o = {(bf16)acc[4 * q], (bf16)acc[4 * q + 1], (bf16)acc[4 * q + 2], (bf16)acc[4 * q + 3]};
            *reinterpret_cast<bf16x4*>(dst) = o;
        }
    }
}

template <bool SLABS, bool ASW>
int sk_launch(const void* A, const void* W, void* C, int M, int N, int K, long lda, long ldw, long ldc,
              long slab_stride, hipStream_t stream) {
    static bool asked[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (!asked[dev]) {   // more than the default 64 KiB of dynamic LDS: asked for once per kernel and device
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_kernel<SLABS, ASW>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, SK_LDS);
        if (e != hipSuccess) {
            mgen_set_error("mgen_gemm_bf16_skinny: cannot reserve %d bytes of LDS: %s", SK_LDS, hipGetErrorString(e));
            return MGEN_ELAUNCH;
        }
        asked[dev] = true;
    }
    const dim3 grid((M + SK_TM - 1) / SK_TM, N / SK_TN, K / SK_KC), block(256);
    hipLaunchKernelGGL((gemm_skinny_kernel<SLABS, ASW>), grid, block, SK_LDS, stream, (const bf16*)A, (const bf16*)W, C,
                       M, N, K, lda, ldw, ldc, slab_stride);
    return mgen_check_launch("mgen_gemm_bf16_skinny");
}

}  // namespace

// 1 when mgen_gemm_bf16_skinny takes this product: whole 64-column and 512-index tiles, 16-byte aligned rows, few
// enough token rows that the operand re-reads of 64 x 64 tiles do not matter, and a bf16 output only without slabs.
extern "C" int mgen_gemm_skinny_fits(int M, int N, int K, long lda, long ldw, long ldc, int slabs) {
    if (M < 1 || M > 1024 || N < 64 || N % 64 || N > 16384 || K < 512 || K % 512) return 0;
    if (lda % 8 || ldw % 8 || ldc % 4) return 0;
    if (!slabs && K != 512) return 0;
    return 1;
}

// A [M][lda] bf16 (a_swiglu: A = gu [M][lda] with lda >= 2K, operand = silu(gu[:, k]) * gu[:, K + k]); W [N][ldw] bf16;
// slabs == 0: C bf16 [M][ldc] (K must be 512); slabs != 0: C fp32, K / 512 partial products [M][ldc] that lie
// slab_stride elements apart. Pointers 16-byte aligned.
extern "C" int mgen_gemm_bf16_skinny(const void* A, const void* W, void* C, int M, int N, int K, long lda, long ldw,
                                     long ldc, int slabs, long slab_stride, int a_swiglu, hipStream_t stream) {
    MGEN_REQUIRE(A && W && C, "mgen_gemm_bf16_skinny: null pointer");
    MGEN_REQUIRE(mgen_gemm_skinny_fits(M, N, K, lda, ldw, ldc, slabs),
                 "mgen_gemm_bf16_skinny: shape not taken (M=%d N=%d K=%d): see mgen_gemm_skinny_fits", M, N, K);
    MGEN_REQUIRE(lda >= (a_swiglu ? 2L * K : (long)K) && ldw >= K && ldc >= N, "mgen_gemm_bf16_skinny: row pitch too small");
    MGEN_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)C % 16) == 0,
                 "mgen_gemm_bf16_skinny: pointers must be 16-byte aligned");
    MGEN_REQUIRE(!slabs || slab_stride >= (long)M * ldc || K == 512, "mgen_gemm_bf16_skinny: slabs overlap");
    if (slabs) {
        if (a_swiglu) return sk_launch<true, true>(A, W, C, M, N, K, lda, ldw, ldc, slab_stride, stream);
        return sk_launch<true, false>(A, W, C, M, N, K, lda, ldw, ldc, slab_stride, stream);
    }
    if (a_swiglu) return sk_launch<false, true>(A, W, C, M, N, K, lda, ldw, ldc, slab_stride, stream);
    return sk_launch<false, false>(A, W, C, M, N, K, lda, ldw, ldc, slab_stride, stream);
}
