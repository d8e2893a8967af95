// Projections of the one-token decode step: C[M, N] = A[M, K] W[N, K]^T with few token rows (M = number of streams,
// tens to a few hundred) and small weights (1-4 MB). Such a product moves almost nothing; what it costs is (i) the
// number of dependent memory round trips between its launch and its last store and (ii) whether its workgroups find
// room on the CUs at all: when the batch is decoded in two parts on two streams, the attention kernel of the other part
// is resident on every CU (4 workgroups of 96 registers and 9.4 KiB of LDS each), and a workgroup that needs more
// than the 128 registers per lane and 121 KiB of LDS that remain waits until that kernel has ended - the projections
// of one part and the attention of the other then run one after the other instead of side by side.
//
// So: a workgroup owns 32 token rows x 64 output columns and takes its reduction range in chunks of 512 indices; both
// operand tiles of a chunk (96 KiB) are requested at once by LDS-DMA (24 one-KiB row pieces per wave, no data
// registers: one round trip), rows of 1 KiB with the 16-byte chunks XOR-swizzled by the row on the source address so
// that the 16 rows an MFMA operand read touches fall on different banks; one barrier; 32 v_mfma_f32_16x16x32_bf16 per
// wave (wave w: output columns 16 w .. 16 w + 15, reduction indices in ascending order: the same bits as the
// tile-loop kernel); the next chunk, if any, into the same LDS; store. With hidden size 512 every workgroup has one
// chunk; the 650M model's projections (reduction 1536 / 6144) take three or four in a row.
//
//   out bf16:  C[m][n] rounded to bf16, the workgroup walks the whole reduction      QKV and gate/up projections
//   out slabs: s fp32 partial products slabs[z][M][N] over K / s indices each, summed by mgen_slab_resid_rmsnorm
//              (s chosen by mgen_gemm_skinny_slabs so that the launch has a workgroup for every CU)
//                                                                                   o_proj and down_proj
//   a_swiglu: the A operand is silu(gate) * up formed from gu[M][2K] on its way to LDS (through registers; gate rounded
//             to bf16 after the silu, the product rounded to bf16: the arithmetic of mgen_swiglu_fwd), so the gated
//             activation of the MLP is never written
//
// Reference: the four nn.Linear calls of a decoder layer with q_len = 1 (transformers models/qwen3/modeling_qwen3.py:
// 252-254, 279, 81-83) under GenerationMixin._sample (generation/utils.py:2876-2937).
#include "lds_dma.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float sk_f32x4;

constexpr int SK_KC = 512;            // reduction indices per workgroup
constexpr int SK_TM = 32, SK_TN = 64; // token rows and output columns per workgroup
constexpr int SK_ROWB = SK_KC * 2;    // bytes per LDS row
constexpr int SK_LDS = (SK_TM + SK_TN) * SK_ROWB;   // 96 KiB

__device__ __forceinline__ int sk_off(int row, int chunk) { return row * SK_ROWB + ((chunk ^ (row & 15)) << 4); }

// LOOP: the workgroup takes cpw chunks in a row (else exactly one: the hidden-size-512 decoders, whose launches must
// stay as small in registers as they are)
template <bool SLABS, bool ASW, bool LOOP>
__global__ __launch_bounds__(256, 4) void gemm_skinny_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W,
                                                             void* __restrict__ Cout, int M, int N, int K, long lda,
                                                             long ldw, long ldc, long slab_stride, int cpw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                          // token tile   [32][512] bf16
    char* sW = smem + SK_TM * SK_ROWB;        // weight tile  [64][512] bf16
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the row tiles of one column tile read the same weight rows: they are neighbours in the order that gives the
    // workgroups of one XCD (one L2) a contiguous range
    const int mt = (M + SK_TM - 1) / SK_TM;
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (t % mt) * SK_TM, n0 = (t / mt) * SK_TN, z = blockIdx.y;
    const unsigned lds0 = lds_addr(smem);
    const int r16 = lane & 15, kg = lane >> 4;
    sk_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};

    const int nchunk = LOOP ? cpw : 1;
#pragma unroll 1
    for (int c = 0; c < nchunk; ++c) {   // the workgroup's chunks of 512 reduction indices (ascending)
    const int k0 = LOOP ? (z * cpw + c) * SK_KC : z * SK_KC;
    if (LOOP && c > 0) __syncthreads();  // every wave has read the chunk before out of LDS

    // ---- every load of the workgroup in flight at once: row r of a tile is one wave instruction (64 lanes x 16 bytes);
    // LDS position p of row r receives source chunk p ^ (r & 15)
    const u32x4 rsW = dma_rsrc(W + (long)n0 * ldw + k0, (unsigned)((63L * ldw + SK_KC) * 2));
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int r = wave + 4 * j;
        dma_piece(rsW, (lane ^ (r & 15)) << 4, lds0 + SK_TM * SK_ROWB + r * SK_ROWB, (int)(r * ldw * 2));
    }
    if (!ASW) {
        const u32x4 rsA = dma_rsrc(A + k0, (unsigned)(((long)(M - 1) * lda + SK_KC) * 2));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = wave + 4 * j;
            const int gr = min(m0 + r, M - 1);            // rows past M repeat the last row (never stored)
            dma_piece(rsA, (lane ^ (r & 15)) << 4, lds0 + r * SK_ROWB, (int)(gr * lda * 2));
        }
    } else {
        u32x4 gv[8], uv[8];
        if (LOOP) {
            const bf16* Ak = A + k0 + lane * 8;
            asm volatile("" : "+v"(Ak));   // (one pointer per chunk; the rows are scalar offsets from it)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const long go = (long)__builtin_amdgcn_readfirstlane(min(m0 + wave + 4 * j, M - 1)) * lda;
                gv[j] = *reinterpret_cast<const u32x4*>(Ak + go);
                uv[j] = *reinterpret_cast<const u32x4*>(Ak + go + K);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int gr = min(m0 + wave + 4 * j, M - 1);
                gv[j] = *reinterpret_cast<const u32x4*>(A + (long)gr * lda + k0 + lane * 8);
                uv[j] = *reinterpret_cast<const u32x4*>(A + (long)gr * lda + K + k0 + lane * 8);
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // (all 16 requests before the first use)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bf16x8 g = __builtin_bit_cast(bf16x8, gv[j]);
            const bf16x8 u = __builtin_bit_cast(bf16x8, uv[j]);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float gf = (float)g[e];
                const float sg = round_bf(gf * __builtin_amdgcn_rcpf(1.f + __expf(-gf)));   // the formula of mgen_swiglu_fwd
                o[e] = (bf16)(sg * (float)u[e]);
            }
            *reinterpret_cast<bf16x8*>(sA + sk_off(wave + 4 * j, lane)) = o;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- wave w: weight rows 16 w .. 16 w + 15 x the 32 token rows; D[n][m]: a lane ends with 4 consecutive n of one
    // token. A lane supplies 8 reduction indices (one 16-byte chunk) of row (lane & 15) per MFMA.
    const int rowW = wave * 16 + r16;
#pragma unroll 8
    for (int ks = 0; ks < SK_KC / 32; ++ks) {
        const int ch = 4 * ks + kg;
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(sW + sk_off(rowW, ch));
        const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(sA + sk_off(r16, ch));
        const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(sA + sk_off(16 + r16, ch));
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b1, acc1, 0, 0, 0);
    }
    }   // chunks
    const int n = n0 + wave * 16 + 4 * kg;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int m = m0 + 16 * mb + r16;
        const sk_f32x4 acc = mb ? acc1 : acc0;
        if (m < M) {
            if (SLABS) {
                float* dst = reinterpret_cast<float*>(Cout) + (long)z * slab_stride + (long)m * ldc + n;
                *reinterpret_cast<sk_f32x4*>(dst) = acc;
            } else {
                bf16* dst = reinterpret_cast<bf16*>(Cout) + (long)m * ldc + n;
                const bf16x4 o = {(bf16)acc[0], (bf16)acc[1], (bf16)acc[2], (bf16)acc[3]};
                *reinterpret_cast<bf16x4*>(dst) = o;
            }
        }
    }
}

template <bool SLABS, bool ASW, bool LOOP>
int sk_launch_(const void* A, const void* W, void* C, int M, int N, int K, long lda, long ldw, long ldc,
              long slab_stride, int nslabs, hipStream_t stream) {
    static bool asked[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (!asked[dev]) {   // more than the default 64 KiB of dynamic LDS: asked for once per kernel and device
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_kernel<SLABS, ASW, LOOP>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, SK_LDS);
        if (e != hipSuccess) {
            mgen_set_error("mgen_gemm_bf16_skinny: cannot reserve %d bytes of LDS: %s", SK_LDS, hipGetErrorString(e));
            return MGEN_ELAUNCH;
        }
        asked[dev] = true;
    }
    const int ny = SLABS ? nslabs : 1;   // bf16 output: one workgroup walks the whole reduction
    const dim3 grid(((M + SK_TM - 1) / SK_TM) * (N / SK_TN), ny), block(256);
    hipLaunchKernelGGL((gemm_skinny_kernel<SLABS, ASW, LOOP>), grid, block, SK_LDS, stream, (const bf16*)A, (const bf16*)W, C,
                       M, N, K, lda, ldw, ldc, slab_stride, K / SK_KC / ny);
    return mgen_check_launch("mgen_gemm_bf16_skinny");
}

template <bool SLABS, bool ASW>
int sk_launch(const void* A, const void* W, void* C, int M, int N, int K, long lda, long ldw, long ldc,
              long slab_stride, int nslabs, hipStream_t stream) {
    const int ny = SLABS ? nslabs : 1;
    if (K / SK_KC / ny > 1) return sk_launch_<SLABS, ASW, true>(A, W, C, M, N, K, lda, ldw, ldc, slab_stride, nslabs, stream);
    return sk_launch_<SLABS, ASW, false>(A, W, C, M, N, K, lda, ldw, ldc, slab_stride, nslabs, stream);
}

}  // namespace

// 1 when mgen_gemm_bf16_skinny takes this product: whole 64-column and 512-index tiles, 16-byte aligned rows, few
// enough token rows that the operand re-reads of 32 x 64 tiles do not matter. slabs: 0 = bf16 output, else any count
// that divides K / 512 (mgen_gemm_skinny_slabs gives the one the callers here use).
extern "C" int mgen_gemm_skinny_fits(int M, int N, int K, long lda, long ldw, long ldc, int slabs) {
    if (M < 1 || M > 1024 || N < 64 || N % 64 || N > 16384 || K < 512 || K % 512 || K > 16384) return 0;
    if (lda % 8 || ldw % 8 || ldc % 4) return 0;
    // (buffer offsets of the LDS-DMA pieces are 32-bit: a tile's last row must start below 2 GiB)
    if (63L * ldw * 2 >= (1L << 31) || (long)(M - 1) * lda * 2 >= (1L << 31)) return 0;
    if (slabs < 0 || (slabs > 0 && (K / SK_KC) % slabs != 0)) return 0;
    return 1;
}

// Number of fp32 slabs for a product summed by mgen_slab_resid_rmsnorm: the smallest divisor of K / 512 that gives the
// launch a workgroup for every CU of a 256-CU chip (more slabs are more bytes for the sum to read; fewer leave CUs
// idle and make every workgroup walk more chunks in a row), else one slab per chunk.
extern "C" int mgen_gemm_skinny_slabs(int M, int N, int K) {
    const int chunks = K / SK_KC > 0 ? K / SK_KC : 1;
    const long tiles = (long)((M + SK_TM - 1) / SK_TM) * (N / SK_TN > 0 ? N / SK_TN : 1);
    for (int s = 1; s <= chunks; ++s)
        if (chunks % s == 0 && tiles * s >= 256) return s;
    return chunks;
}

// A [M][lda] bf16 (a_swiglu: A = gu [M][lda] with lda >= 2K, operand = silu(gu[:, k]) * gu[:, K + k]); W [N][ldw] bf16;
// slabs == 0: C bf16 [M][ldc]; slabs = s > 0: C fp32, s partial products [M][ldc] over K / s reduction indices each
// that lie slab_stride elements apart. Pointers 16-byte aligned.
extern "C" int mgen_gemm_bf16_skinny(const void* A, const void* W, void* C, int M, int N, int K, long lda, long ldw,
                                     long ldc, int slabs, long slab_stride, int a_swiglu, hipStream_t stream) {
    MGEN_REQUIRE(A && W && C, "mgen_gemm_bf16_skinny: null pointer");
    MGEN_REQUIRE(mgen_gemm_skinny_fits(M, N, K, lda, ldw, ldc, slabs),
                 "mgen_gemm_bf16_skinny: shape not taken (M=%d N=%d K=%d): see mgen_gemm_skinny_fits", M, N, K);
    MGEN_REQUIRE(lda >= (a_swiglu ? 2L * K : (long)K) && ldw >= K && ldc >= N, "mgen_gemm_bf16_skinny: row pitch too small");
    MGEN_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)C % 16) == 0,
                 "mgen_gemm_bf16_skinny: pointers must be 16-byte aligned");
    MGEN_REQUIRE(slabs <= 1 || slab_stride >= (long)M * ldc, "mgen_gemm_bf16_skinny: slabs overlap");
    if (slabs) {
        if (a_swiglu) return sk_launch<true, true>(A, W, C, M, N, K, lda, ldw, ldc, slab_stride, slabs, stream);
        return sk_launch<true, false>(A, W, C, M, N, K, lda, ldw, ldc, slab_stride, slabs, stream);
    }
    if (a_swiglu) return sk_launch<false, true>(A, W, C, M, N, K, lda, ldw, ldc, slab_stride, 1, stream);
    return sk_launch<false, false>(A, W, C, M, N, K, lda, ldw, ldc, slab_stride, 1, stream);
}
