// bf16 GEMM for weight gradients: C[M,N] (fp32) (+)= A^T B with both operands stored with the reduction index as the
// row index (A = dY [tokens][M], B = X [tokens][N]; reference: the grad_weight product of nn.Linear's backward,
// grad_output^T @ input, as reached from transformers models/qwen3/modeling_qwen3.py:76-83,225-236).
//
// Output stationary: a workgroup of four waves (one per SIMD, 512 registers each) owns a 256 x 256 output tile, each
// wave a 128 x 128 quarter in 256 accumulator registers, and streams its slice of the reduction range in steps of
// 32 rows through a four-slot LDS ring filled by LDS-DMA: three steps of loads in flight, one barrier per step, the
// 8 DMA pieces of a wave spread between its MFMAs, the waves' pieces at staggered positions. Because the contiguous
// index of both operands is the OUTPUT index, a 32-row step still moves whole 256-byte row segments. Operand
// fragments are read with the transposing LDS read from the same swizzled image the other GEMM kernels use.
// Few output tiles and a long reduction (layer weights): the reduction range is cut into slices, one workgroup per
// tile and slice, added with fp32 atomics. Many tiles (lm_head): one workgroup per tile.
#include "lds_dma.h"
#include "gemm_os.h"
#include <type_traits>

namespace {

constexpr int OS_BK = 32;
constexpr int OS_IMG = OS_BK * 256;          // 32 reduction rows x 128 columns: 8 KiB
constexpr int OS_SLOT = 4 * OS_IMG;          // A columns 0..127 | A 128..255 | B 0..127 | B 128..255
constexpr int OS_NSLOT = 4;
constexpr int OS_LDS = OS_NSLOT * OS_SLOT;   // 128 KiB

struct OsArgs {
    const bf16* A;
    const bf16* B;
    float* C;
    long lda, ldb, ldc;
    int M, N, K;
    int nbm, nbn;
    int splitk, kper;   // reduction slices of kper rows (a multiple of OS_BK)
    int accumulate;
};

template <int WV>
__device__ __forceinline__ void os_body(const OsArgs& g, char* smem, int lane, int wave) {
    const int nt = g.nbm * g.nbn;
    const int wid = xcd_remap(blockIdx.x, nt * g.splitk);
    // slice-major: the tiles of one reduction slice are neighbours and share that slice of both operands in L2
    const int kslice = wid / nt, t = wid % nt;
    const int tm = t / g.nbn, tn = t % g.nbn;
    const int m0 = tm * 256, n0 = tn * 256;
    const int kbeg = kslice * g.kper;
    const int kend = min(g.K, kbeg + g.kper);
    const int nslot = (kend - kbeg) / OS_BK;
    if (nslot <= 0) return;
    const unsigned lds0 = lds_addr(smem);

    // ---- loader role: this wave fills one of the four images of every slot (8 pieces of 4 rows x 256 bytes)
    const bool isB = wave >= 2;
    const bf16* P = isB ? g.B : g.A;
    const int ld2 = (int)((isB ? g.ldb : g.lda) * 2);
    const int extent = isB ? g.N : g.M;
    const int col0 = isB ? n0 + 128 * (wave - 2) : m0 + 128 * wave;
    const u32x4 rs = dma_rsrc(P, (unsigned)(((long)(g.K - 1) * (ld2 / 2) + extent) * 2));
    int voff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int kr = 4 * q + (lane >> 4);
        voff[q] = kr * ld2 + col0 * 2 + 16 * ((lane & 15) ^ swz256(kr));
    }
    // piece pc (0..7) of the slot whose first row is k0: rows 4 pc .. 4 pc + 3 (voff[pc & 3], + 16 rows for pc >= 4)
    auto piece = [&](int sl, int pc) {
        const int sc = min(sl, nslot - 1);   // past the end: a harmless reload, so that the wait counts stay fixed
        const int soff = __builtin_amdgcn_readfirstlane((kbeg + sc * OS_BK + 16 * (pc >> 2)) * ld2);
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (sl & 3) * OS_SLOT + wave * OS_IMG + pc * 1024);
        dma_piece(rs, voff[pc & 3], dst, soff);
    };

    // ---- compute role: rows 128 wm.. of the tile from A image wm, columns 128 wn.. from B image wn
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#pragma unroll
    for (int sl = 0; sl < 3; ++sl)
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) piece(sl, pc);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    auto frags = [&](const char* slot, int ks, bf16x8 (&fa)[4], bf16x8 (&fb)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = frag_col_img256<false>(slot + wm * OS_IMG, 16 * ks, 32 * i, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = frag_col_img256<false>(slot + (2 + wn) * OS_IMG, 16 * ks, 32 * j, lane);
    };
    // 16 MFMAs of one 16-deep step; DMA piece pc0 + (idx >> 2) of slot `nsl` behind MFMA idx with idx & 3 == WV
    auto step = [&](const bf16x8 (&fa)[4], const bf16x8 (&fb)[4], int nsl, int pc0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = mfma32(fa[i], fb[j], acc[i][j]);   // D[row = m][col = n]: a register = 32 consecutive n
                if (((4 * i + j) & 3) == WV) piece(nsl, pc0 + ((4 * i + j) >> 2));
            }
    };

    bf16x8 fa0[4], fb0[4], fa1[4], fb1[4];
    frags(smem, 0, fa0, fb0);
    for (int s = 0; s < nslot; ++s) {
        // This wave's pieces of slot s + 1 have landed (younger: the 8 pieces of slot s + 2). Every wave's LDS reads of
        // slot s - 1 have returned (they fed MFMAs that are issued); past the barrier its ring slot takes slot s + 3,
        // and slot s + 1 is complete for every wave: its first fragments may be requested during this step.
        dma_wait<8>();
        asm volatile("s_barrier" ::: "memory");
        const char* cur = smem + (s & 3) * OS_SLOT;
        const char* nxt = smem + ((s + 1) & 3) * OS_SLOT;
        frags(cur, 1, fa1, fb1);
        step(fa0, fb0, s + 3, 0);
        frags(nxt, 0, fa0, fb0);
        step(fa1, fb1, s + 3, 4);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may be pending when the wave ends

    // ---- output: one accumulator register = 32 consecutive columns of two rows (two 128-byte segments)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + 128 * wn + 32 * j + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 128 * wm + 32 * i + acc_row(r, lane);
                if (m < g.M && n < g.N) {
                    float* c = g.C + (long)m * g.ldc + n;
                    if (g.accumulate) atomicAdd(c, acc[i][j][r]);
                    else *c = acc[i][j][r];
                }
            }
        }
}

__global__ __launch_bounds__(256, 1) void gemm_os_kernel(OsArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // four copies of the loop that differ only in where the wave's DMA pieces sit between its MFMAs
    switch (wave) {
        case 0: os_body<0>(g, smem, lane, wave); break;
        case 1: os_body<1>(g, smem, lane, wave); break;
        case 2: os_body<2>(g, smem, lane, wave); break;
        default: os_body<3>(g, smem, lane, wave); break;
    }
}

}  // namespace

bool mgen_gemm_os_fits(int M, int N, int K, long lda, long ldb, long ldc) {
    const char* e = getenv("MGEN_GEMM_OS");   // MGEN_GEMM_OS=0: never (same-process A/B runs)
    if (e && e[0] == '0') return false;
    if (K % OS_BK != 0 || K < 2048 || M < 256 || N < 256 || (M % 8) || (N % 8) || (lda % 8) || (ldb % 8)) return false;
    // byte offsets inside the buffer descriptors are 32-bit
    if ((long)K * lda * 2 >= (1L << 32) || (long)K * ldb * 2 >= (1L << 32)) return false;
    return true;
}

int mgen_gemm_os_launch(const void* A, const void* B, float* C, int M, int N, int K, long lda, long ldb, long ldc,
                        int accumulate, hipStream_t stream) {
    OsArgs g;
    g.A = (const bf16*)A; g.B = (const bf16*)B; g.C = C;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.nbm = (M + 255) / 256; g.nbn = (N + 255) / 256;
    g.accumulate = accumulate;
    // one round of workgroups on 256 CUs: cut the reduction range while the tiles alone do not fill the chip
    // (slices of at least 1024 rows; a product that does not accumulate cannot be cut)
    const int tiles = g.nbm * g.nbn;
    const int steps = K / OS_BK;
    int sk = 1;
    if (accumulate && tiles < 256) {
        sk = 256 / tiles;
        const int maxsk = steps / 32;
        if (sk > maxsk) sk = maxsk;
        if (sk < 1) sk = 1;
    }
    const int per = (steps + sk - 1) / sk;
    g.kper = per * OS_BK;
    g.splitk = (steps + per - 1) / per;
    static bool asked[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (!asked[dev]) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_os_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, OS_LDS);
        if (e != hipSuccess) {
            mgen_set_error("mgen_gemm_bf16: cannot reserve %d bytes of LDS: %s", OS_LDS, hipGetErrorString(e));
            return MGEN_ELAUNCH;
        }
        asked[dev] = true;
    }
    hipLaunchKernelGGL(gemm_os_kernel, dim3(tiles * g.splitk), dim3(256), OS_LDS, stream, g);
    return mgen_check_launch("mgen_gemm_bf16");
}
