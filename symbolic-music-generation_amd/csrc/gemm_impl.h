// bf16 MFMA GEMM for gfx950: C[M,N] = op(A)[M,K] * op(B)[K,N], fp32 accumulate.
//
// Serves the linear layers of the decoder on the training and prefill paths
// (reference: nn.Linear calls in transformers models/qwen3/modeling_qwen3.py:76-83,225-236 and their autograd
// backward): forward x*W^T (A k-contiguous, B k-contiguous), dgrad dY*W (B stored [K][N]), wgrad dY^T*X (both
// operands stored with the reduction index as the row index).
//
// Tiling: 256 threads = 4 waves, K step 64, v_mfma_f32_32x32x16_bf16. Workgroup tile 128x256 (each wave 64x128)
// when the product has enough row tiles, else 128x128 (each wave 64x64). When K is a multiple of 64 the operand
// tiles go global -> LDS by LDS-DMA into ONE stage (48 or 32 KiB), with 2-4 workgroups per CU covering each other's
// load latency (a second stage per workgroup costs more in occupancy than it hides: measured); otherwise a
// register-staged, zero-filling, double-buffered variant is used. Wide products (N >= 1024) with a tile for every
// CU run on 256x256 tiles with 8 waves and two 64 KiB LDS-DMA stages, one workgroup per CU (a persistent form of
// it that prefetches the next tile under the epilogue measured no faster: reverted). An operand whose reduction
// index is contiguous
// in memory lives in LDS as [128 rows][64 k] (row reads); one whose reduction index is the row index lives as
// [64 k][128 cols] and is read with the transposing LDS read, so no transposed copy of activations or weights is
// ever made in HBM. Workgroups are numbered so that each XCD (own L2) gets a contiguous range of tiles, grouped 8
// row tiles at a time; split-K slices are numbered slice-major.
//
// This file is compiled twice: gemm.hip (MGEN_GEMM_SEG 0: every entry point but one) and gemm_seg.hip
// (MGEN_GEMM_SEG 1: mgen_gemm_bf16_seg, whose reduction runs over two operand pairs back to back - a base product
// and a low-rank adapter product in one accumulator). Separate translation units keep the plain kernels' code free
// of the second pair's address selection.
#include "common.h"
#include "gemm_rs.h"
#include "gemm_os.h"

#ifndef MGEN_GEMM_SEG
#error "include through gemm.hip or gemm_seg.hip"
#endif

namespace {

constexpr bool kSeg = MGEN_GEMM_SEG != 0;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;  // 16 KiB per operand tile

enum OutMode { OUT_BF16 = 0, OUT_F32 = 1, OUT_F32_ACC = 2, OUT_F32_RESID_BF16ROUND = 3, OUT_F32_RESID = 4,
               OUT_F32_ATOMIC = 5 /* internal: split-K form of OUT_F32_ACC */,
               OUT_BF16_ACC = 6 /* C (bf16) = bf16(C + product): low-rank adapter paths, 128x128 tiles only */,
               OUT_BF16_SWIGLU = 7 /* internal: C = [gate | up] (N = 2 I) and act = silu(gate) * up, 256x256 tiles whose
                                      columns are 128 gate columns and the 128 up columns of the same indices */ };

struct GemmArgs {
    const bf16* A;
    const bf16* B;
    void* C;
    const float* R;  // residual for the OUT_F32_RESID* modes
    const bf16* bias;  // optional [N] added to the accumulator before any rounding (torch.addmm under autocast)
    int M, N, K;
    long lda, ldb, ldc, ldr;
    float alpha;
    int nbm, nbn;
    int splitk, kper;  // K is cut into `splitk` slices of `kper` elements (kper % BK == 0)
    long slab_stride;  // fp32 store modes: slice z writes its partial product at C + z * slab_stride (0: one output)
    // second operand pair (kSeg builds): reduction indices >= K1 come from A2 / B2 at index - K1 (K1 % 64 == 0)
    const bf16* A2;
    const bf16* B2;
    long lda2, ldb2;
    int K1;
    bf16* act;   // OUT_BF16_SWIGLU: [M][N / 2] output of the gated activation
    long ldact;
};

// ---- global -> register staging of one operand tile (4 x 16 bytes per thread) -----------------------------
// KC: operand stored [rows][K] (k contiguous): tile = 128 rows x 64 k
// MC: operand stored [K][cols] (reduction index is the row): tile = 64 k x 128 cols
template <bool MC>
__device__ __forceinline__ void load_tile(u32x4 (&st)[4], const bf16* __restrict__ P, long ld, int row0, int k0,
                                          int nrows, int K, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (!MC) {
            const int row = (tid >> 3) + 32 * i, ch = tid & 7;
            const int gr = row0 + row, gk = k0 + ch * 8;
            if (gr < nrows && gk < K) v = *reinterpret_cast<const u32x4*>(P + (long)gr * ld + gk);
        } else {
            const int kr = (tid >> 4) + 16 * i, ch = tid & 15;
            const int gk = k0 + kr, gc = row0 + ch * 8;
            if (gk < K && gc < nrows) v = *reinterpret_cast<const u32x4*>(P + (long)gk * ld + gc);
        }
        st[i] = v;
    }
}

template <bool MC>
__device__ __forceinline__ void store_tile(const u32x4 (&st)[4], char* lds, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (!MC) {
            const int row = (tid >> 3) + 32 * i, ch = tid & 7;
            lds_write_b128(lds, img128_off(row, ch), st[i]);
        } else {
            const int kr = (tid >> 4) + 16 * i, ch = tid & 15;
            lds_write_b128(lds, img256_off(kr, ch), st[i]);
        }
    }
}

// ---- global -> LDS directly (LDS-DMA, 16 bytes per lane): a wave instruction fills one contiguous 1 KiB piece of
// the tile image, so the XOR swizzle of the image is applied to each lane's SOURCE address. No zero fill: used when
// K is a multiple of the K step; rows/columns beyond the matrix edge are clamped (they feed outputs never stored).
template <bool MC, int NW = 4>
__device__ __forceinline__ void glds_tile(const bf16* __restrict__ P, long ld, int row0, int k0, int nrows,
                                          char* lds_tile, int tid) {
    const int wave = tid >> 6, l = tid & 63;
    constexpr int PPW = 16 / NW;   // 1 KiB pieces per wave
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int piece = wave * PPW + j;
        const bf16* src;
        if (!MC) {
            const int row = piece * 8 + (l >> 3);
            const int ch = (l & 7) ^ ((row >> 1) & 7);
            const int gr = min(row0 + row, nrows - 1);
            src = P + (long)gr * ld + k0 + ch * 8;
        } else {
            const int kr = piece * 4 + (l >> 4);
            const int ch = (l & 15) ^ swz256(kr);
            const int gc = min(row0 + ch * 8, ((nrows - 1) >> 3) << 3);
            src = P + (long)(k0 + kr) * ld + gc;
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds_tile + piece * 1024), 16, 0, 0);
    }
}

// fragment for output rows/cols [o0, o0+32) and k-step ks (16 reduction indices) of the current tile
template <bool MC>
__device__ __forceinline__ bf16x8 load_frag(const char* tile, int o0, int ks, int lane) {
    if (!MC) return frag_row_img128(tile, o0, ks, lane);
    return frag_col_img256<false>(tile, 16 * ks, o0, lane);
}

// GLDS = true: single 32 KiB LDS image filled by LDS-DMA, two barriers per K step, up to 4 workgroups per CU cover
// each other's load latency. GLDS = false: register-staged, zero-filling variant for K not a multiple of 64.
// NJ = 32-column MFMA tiles per wave: 2 -> 128-wide workgroup tile (wave 64x64), 4 -> 256-wide (wave 64x128, GLDS
// only). A k-step of a 64x64 wave tile reads 4 fragments from LDS for 4 MFMAs, which keeps the CU's LDS port as busy
// as its matrix cores; the 64x128 wave tile reads 6 fragments for 8 MFMAs.
// WM = 128-row A images per workgroup: 1 -> 4 waves; 2 -> 8 waves on a 256x256 tile (NJ = 4, GLDS), two 64 KiB LDS
// stages, one workgroup per CU: the LDS-DMA of K step kt+1 is issued before the MFMAs of step kt and one barrier per
// step both publishes a stage and frees the other. The 256x256 tile moves half the operand bytes per flop of the
// 128x128 one: what bounds the smaller tiles is the rate at which tiles can be brought from L2 into LDS.
template <bool TA, bool TB, int MODE, bool GLDS, int NJ, int WM = 1>
__global__ __launch_bounds__(256 * WM, WM == 2 ? 2 : (GLDS ? (NJ == 4 ? 2 : 3) : 2)) void gemm_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // buffer b: A tile at smem + b*2*TILE_BYTES, B tile right behind it (GLDS: A tile, then NJ/2 B tiles of 128 columns)
    constexpr int BNW = 64 * NJ;   // workgroup tile width
    constexpr int BMW = BM * WM;   // workgroup tile height
    static_assert(WM == 1 || (GLDS && NJ == 4), "256-row tiles: LDS-DMA variant with 256 columns only");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware, grouped tile order: 8 row-tiles share the sweep over the column tiles
    const int nwg = g.nbm * g.nbn;
    const int wid = xcd_remap(blockIdx.x, nwg * g.splitk);
    // K-slice-major: the output tiles of one K slice are neighbours (same XCD, same time) and share that slice of
    // both operands in L2; slices of the same output tile meet only in the fp32 atomics / slabs
    const int wg = wid % nwg;
    const int kslice = wid / nwg;
    const int kbeg = kslice * g.kper;
    const int kend = min(g.K, kbeg + g.kper);
    const long coff = (long)kslice * g.slab_stride;
    constexpr int GROUP = 8;
    const int width = GROUP * g.nbn;
    const int grp = wg / width;
    const int first_m = grp * GROUP;
    const int gsz = min(g.nbm - first_m, GROUP);
    const int tm = first_m + (wg % width) % gsz;
    const int tn = (wg % width) / gsz;
    const int m0 = tm * BMW, n0 = tn * BNW;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * (32 * NJ);

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (kend - kbeg + BK - 1) / BK;
    if (GLDS && WM == 2) {
        constexpr int STAGE = 4 * TILE_BYTES;   // [A rows 0..127][A rows 128..255][B cols 0..127][B cols 128..255]
        auto issue = [&](int kt, char* st) {
            int k0 = kbeg + kt * BK;
            const bf16* pa = g.A;
            const bf16* pb = g.B;
            long la = g.lda, lb = g.ldb;
            if (kSeg && k0 >= g.K1) { k0 -= g.K1; pa = g.A2; pb = g.B2; la = g.lda2; lb = g.ldb2; }
            glds_tile<TA, 8>(pa, la, m0, k0, g.M, st, tid);
            glds_tile<TA, 8>(pa, la, m0 + 128, k0, g.M, st + TILE_BYTES, tid);
            if constexpr (MODE == OUT_BF16_SWIGLU) {   // 128 gate rows of W and the up rows of the same indices
                glds_tile<TB, 8>(pb, lb, tn * 128, k0, g.N, st + 2 * TILE_BYTES, tid);
                glds_tile<TB, 8>(pb, lb, g.N / 2 + tn * 128, k0, g.N, st + 3 * TILE_BYTES, tid);
            } else {
                glds_tile<TB, 8>(pb, lb, n0, k0, g.N, st + 2 * TILE_BYTES, tid);
                glds_tile<TB, 8>(pb, lb, n0 + 128, k0, g.N, st + 3 * TILE_BYTES, tid);
            }
        };
        issue(0, smem);
        const int aoff = TILE_BYTES * (wm >> 7), arow = wm & 127;
        const int boff = TILE_BYTES * (2 + (wn >> 7));
        for (int kt = 0; kt < nk; ++kt) {
            const char* cur = smem + (kt & 1) * STAGE;
            // only the loads of step kt are outstanding here; past the barrier every wave has also finished the
            // MFMAs of step kt-1, so the other stage may be overwritten
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 1 < nk) issue(kt + 1, smem + ((kt + 1) & 1) * STAGE);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 fa[2], fb[NJ];
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = load_frag<TA>(cur + aoff, arow + 32 * i, ks, lane);
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[j] = load_frag<TB>(cur + boff, 32 * j, ks, lane);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = (MODE == OUT_F32_ATOMIC) ? mfma32(fa[i], fb[j], acc[i][j])
                                                             : mfma32(fb[j], fa[i], acc[i][j]);
            }
        }
        __syncthreads();   // the epilogues reuse the staging memory
    } else if (GLDS) {
        // the wave's B columns [wn, wn + 32*NJ) lie inside one 128-column B image
        const char* btile = smem + TILE_BYTES * (1 + (wn >> 7));
        const int bcol = wn & 127;
        for (int kt = 0; kt < nk; ++kt) {
            int k0 = kbeg + kt * BK;
            const bf16* pa = g.A;
            const bf16* pb = g.B;
            long la = g.lda, lb = g.ldb;
            if (kSeg && k0 >= g.K1) { k0 -= g.K1; pa = g.A2; pb = g.B2; la = g.lda2; lb = g.ldb2; }
            glds_tile<TA>(pa, la, m0, k0, g.M, smem, tid);
#pragma unroll
            for (int hb = 0; hb < NJ / 2; ++hb)
                glds_tile<TB>(pb, lb, n0 + 128 * hb, k0, g.N, smem + TILE_BYTES * (1 + hb), tid);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 fa[2], fb[NJ];
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = load_frag<TA>(smem, wm + 32 * i, ks, lane);
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[j] = load_frag<TB>(btile, bcol + 32 * j, ks, lane);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = (MODE == OUT_F32_ATOMIC) ? mfma32(fa[i], fb[j], acc[i][j])
                                                             : mfma32(fb[j], fa[i], acc[i][j]);
            }
            __syncthreads();
        }
    } else {
    u32x4 sa[4], sb[4];
    load_tile<TA>(sa, g.A, g.lda, m0, kbeg, g.M, kend, tid);
    load_tile<TB>(sb, g.B, g.ldb, n0, kbeg, g.N, kend, tid);
    store_tile<TA>(sa, smem, tid);
    store_tile<TB>(sb, smem + TILE_BYTES, tid);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            load_tile<TA>(sa, g.A, g.lda, m0, kbeg + (kt + 1) * BK, g.M, kend, tid);
            load_tile<TB>(sb, g.B, g.ldb, n0, kbeg + (kt + 1) * BK, g.N, kend, tid);
        }
        const char* ta = smem + cur * 2 * TILE_BYTES;
        const char* tb = ta + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = load_frag<TA>(ta, wm + 32 * i, ks, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = load_frag<TB>(tb, wn + 32 * j, ks, lane);
            // D[row = n][col = m]: the B-side fragment is the MFMA's A operand, so that each lane ends up with
            // 4 consecutive n of one output row m (vector stores along the contiguous dimension of C).
            // The atomic (split-K) epilogue wants the opposite: D[row = m][col = n] puts 32 consecutive n of one
            // row on 32 consecutive lanes, i.e. two 128-byte segments per atomic wave-instruction.
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = (MODE == OUT_F32_ATOMIC) ? mfma32(fa[i], fb[j], acc[i][j])
                                                         : mfma32(fb[j], fa[i], acc[i][j]);
        }
        if (kt + 1 < nk) {
            char* na = smem + (cur ^ 1) * 2 * TILE_BYTES;
            store_tile<TA>(sa, na, tid);
            store_tile<TB>(sb, na + TILE_BYTES, tid);
        }
        __syncthreads();
    }
    }

    if (MODE == OUT_F32_ATOMIC) {
        float* C = reinterpret_cast<float*>(g.C);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = n0 + wn + 32 * j + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm + 32 * i + acc_row(r, lane);
                    if (m < g.M && n < g.N) atomicAdd(C + (long)m * g.ldc + n, acc[i][j][r] * g.alpha);
                }
            }
        return;
    }
    const int h = lane >> 5;
    if (g.bias) {
        // the lane holds, per tile, columns 8*g4 + 4*h + e of the tile: add the bias to the fp32 accumulators
        // (alpha is 1 whenever a bias is given)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int n = n0 + wn + 32 * j + 8 * g4 + 4 * h;
                float b[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) b[e] = (n + e < g.N) ? (float)g.bias[n + e] : 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][4 * g4 + e] += b[e];
            }
    }
    if constexpr (MODE == OUT_BF16_SWIGLU && WM == 2 && NJ == 4) {
        // Waves 2p and 2p + 1 hold the gate and the up columns of the same 64 rows. Each parks a 64x64 half of its tile
        // (rounded to bf16: what the separate kernel would read back) in its 8 KiB, writes its own [gate | up] rows,
        // and computes the gated activation of 32 of the pair's rows from both parked halves. Arithmetic of
        // swiglu_fwd_kernel<true>: silu in fp32 from the bf16 gate, rounded to bf16, times up, rounded to bf16.
        char* st = smem + wave * 8192;
        const char* st_g = smem + (wave & ~1) * 8192;
        const char* st_u = smem + (wave | 1) * 8192;
        bf16* C = reinterpret_cast<bf16*>(g.C);
        const int I = g.N / 2;
        const int ncol0 = tn * 128 + ((wave & 1) ? I : 0);   // first [gate | up] column of this wave's tile
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half) __syncthreads();   // the pair's reads of the previous half are done
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int row = 32 * i + (lane & 31);
                        const int ch = 4 * j + g4;
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (bf16)(acc[i][2 * half + j][4 * g4 + e]);
                        *reinterpret_cast<bf16x4*>(st + row * 128 + ((ch ^ (row & 7)) << 4) + 8 * h) = o;
                    }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = it * 8 + (lane >> 3), ch = lane & 7;
                const u32x4 v = *reinterpret_cast<const u32x4*>(st + row * 128 + ((ch ^ (row & 7)) << 4));
                const int m = m0 + wm + row;
                if (m < g.M) *reinterpret_cast<u32x4*>(C + (long)m * g.ldc + ncol0 + 64 * half + ch * 8) = v;
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = 32 * (wave & 1) + it * 8 + (lane >> 3), ch = lane & 7;
                const int off = row * 128 + ((ch ^ (row & 7)) << 4);
                const bf16x8 gv = *reinterpret_cast<const bf16x8*>(st_g + off);
                const bf16x8 uv = *reinterpret_cast<const bf16x8*>(st_u + off);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float gf = (float)gv[e];
                    const float sl = round_bf(gf * __builtin_amdgcn_rcpf(1.f + __expf(-gf)));   // as sigmoidf_ in elem.hip
                    o[e] = (bf16)(sl * (float)uv[e]);
                }
                const int m = m0 + wm + row;
                if (m < g.M) *reinterpret_cast<bf16x8*>(g.act + (long)m * g.ldact + tn * 128 + 64 * half + ch * 8) = o;
            }
        }
        return;
    }
    if ((MODE == OUT_BF16 || MODE == OUT_BF16_ACC) && (g.ldc & 7) == 0) {
        // bf16 output through LDS: the wave parks its 64x64 sub-tile in its own 8 KiB of the (now idle) staging
        // area as [64 rows][128 bytes] with the 16-byte chunks XOR-swizzled by row, then writes whole 128-byte row
        // segments with 16 bytes per lane (8 store instructions per lane instead of 32 scattered 8-byte ones).
        // (a 64x128 wave tile goes through the same 8 KiB in two 64-column halves.)
        char* st = smem + wave * 8192;
        bf16* C = reinterpret_cast<bf16*>(g.C);
#pragma unroll
        for (int half = 0; half < NJ / 2; ++half) {
            if (half) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the previous half has been read out
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int row = 32 * i + (lane & 31);
                        const int ch = 4 * j + g4;
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (bf16)(acc[i][2 * half + j][4 * g4 + e] * g.alpha);
                        *reinterpret_cast<bf16x4*>(st + row * 128 + ((ch ^ (row & 7)) << 4) + 8 * h) = o;
                    }
            // same wave wrote and reads: no workgroup barrier needed, only the LDS counter
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = it * 8 + (lane >> 3), ch = lane & 7;
                const u32x4 v = *reinterpret_cast<const u32x4*>(st + row * 128 + ((ch ^ (row & 7)) << 4));
                const int m = m0 + wm + row, n = n0 + wn + 64 * half + ch * 8;
                if (m < g.M && n < g.N) {
                    bf16* c = C + (long)m * g.ldc + n;
                    if (n + 7 < g.N) {
                        if (MODE == OUT_BF16_ACC) {   // bf16(old + bf16(product)), 16 bytes at a time
                            const bf16x8 b = __builtin_bit_cast(bf16x8, v);
                            const bf16x8 old = *reinterpret_cast<const bf16x8*>(c);
                            bf16x8 sum;
#pragma unroll
                            for (int e = 0; e < 8; ++e) sum[e] = (bf16)((float)old[e] + (float)b[e]);
                            *reinterpret_cast<bf16x8*>(c) = sum;
                        } else {
                            *reinterpret_cast<u32x4*>(c) = v;
                        }
                    } else {
                        const bf16x8 b = __builtin_bit_cast(bf16x8, v);
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (n + e < g.N)
                                c[e] = MODE == OUT_BF16_ACC ? (bf16)((float)c[e] + (float)b[e]) : b[e];
                    }
                }
            }
        }
        return;
    }
    if (MODE != OUT_BF16 && MODE != OUT_BF16_ACC && (g.ldc & 3) == 0 && (g.N & 3) == 0 &&
        ((MODE != OUT_F32_RESID_BF16ROUND && MODE != OUT_F32_RESID) || (g.ldr & 3) == 0)) {
        // fp32 output through LDS, 32 rows at a time: [32 rows][256 bytes] per wave, 16-byte chunks XOR-swizzled by
        // row; then whole 256-byte row segments are combined with the old value / residual and stored, 16 B per lane.
        char* st = smem + wave * 8192;
        float* C = reinterpret_cast<float*>(g.C) + coff;
#pragma unroll
        for (int pass = 0; pass < NJ; ++pass) {   // (row half i, 64-column half) of the wave tile
            const int i = pass & 1, half = pass >> 1;
            if (pass) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int row = lane & 31;
                    const int ch = 8 * j + 2 * g4 + h;
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = acc[i][2 * half + j][4 * g4 + e] * g.alpha;
                    *reinterpret_cast<f32x4*>(st + row * 256 + ((ch ^ (row & 15)) << 4)) = o;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = it * 4 + (lane >> 4), ch = lane & 15;
                f32x4 v = *reinterpret_cast<const f32x4*>(st + row * 256 + ((ch ^ (row & 15)) << 4));
                const int m = m0 + wm + 32 * i + row, n = n0 + wn + 64 * half + ch * 4;
                if (m < g.M && n < g.N) {
                    float* c = C + (long)m * g.ldc + n;
                    if (MODE == OUT_F32_RESID_BF16ROUND || MODE == OUT_F32_RESID) {
                        const f32x4 r = *reinterpret_cast<const f32x4*>(g.R + (long)m * g.ldr + n);
                        if (MODE == OUT_F32_RESID_BF16ROUND) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = round_bf(v[e]);
                        }
                        v += r;
                    } else if (MODE == OUT_F32_ACC) {
                        v += *reinterpret_cast<const f32x4*>(c);
                    }
                    *reinterpret_cast<f32x4*>(c) = v;
                }
            }
        }
        return;
    }
    // ---- generic epilogue: lane holds, per tile, m = lane&31 and n = 8*g4 + 4*(lane>>5) + e (g4 = 0..3, e = 0..3)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm + 32 * i + (lane & 31);
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int n = n0 + wn + 32 * j + 8 * g4 + 4 * h;
                if (n >= g.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g4 + e] * g.alpha;
                const bool full = (n + 3 < g.N);
                if (MODE == OUT_BF16 || MODE == OUT_BF16_ACC) {
                    bf16* c = reinterpret_cast<bf16*>(g.C) + (long)m * g.ldc + n;
                    if (MODE == OUT_BF16_ACC) {
                        for (int e = 0; e < 4; ++e)
                            if (n + e < g.N) v[e] += (float)c[e];
                    }
                    if (full && ((g.ldc & 3) == 0)) {
                        bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                        *reinterpret_cast<bf16x4*>(c) = o;
                    } else {
                        for (int e = 0; e < 4 && n + e < g.N; ++e) c[e] = (bf16)v[e];
                    }
                } else {
                    float* c = reinterpret_cast<float*>(g.C) + coff + (long)m * g.ldc + n;
                    if (MODE == OUT_F32_RESID_BF16ROUND || MODE == OUT_F32_RESID) {
                        const float* r = g.R + (long)m * g.ldr + n;
                        for (int e = 0; e < 4; ++e)
                            if (n + e < g.N)
                                v[e] = r[e] + (MODE == OUT_F32_RESID_BF16ROUND ? round_bf(v[e]) : v[e]);
                    } else if (MODE == OUT_F32_ACC) {
                        for (int e = 0; e < 4; ++e)
                            if (n + e < g.N) v[e] += c[e];
                    }
                    if (full && ((g.ldc & 3) == 0)) {
                        f32x4 o = {v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(c) = o;
                    } else {
                        for (int e = 0; e < 4 && n + e < g.N; ++e) c[e] = v[e];
                    }
                }
            }
        }
    }
}

template <bool TA, bool TB, int MODE, bool GLDS, int NJ, int WM>
int launch_one(const GemmArgs& g, hipStream_t s) {
    const dim3 grid(g.nbm * g.nbn * g.splitk), block(256 * WM);
    const size_t lds = (WM == 2 ? 8 : (GLDS ? 1 + NJ / 2 : 4)) * TILE_BYTES;
    if (lds > 65536) {
        // more than the default 64 KiB of dynamic LDS has to be asked for, once per kernel and device
        static bool asked[16] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
        if (!asked[dev]) {
            const hipError_t e = hipFuncSetAttribute(
                reinterpret_cast<const void*>(&gemm_kernel<TA, TB, MODE, GLDS, NJ, WM>),
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) {
                mgen_set_error("mgen_gemm_bf16: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
                return MGEN_ELAUNCH;
            }
            asked[dev] = true;
        }
    }
    hipLaunchKernelGGL((gemm_kernel<TA, TB, MODE, GLDS, NJ, WM>), grid, block, lds, s, g);
    return mgen_check_launch("mgen_gemm_bf16");
}

template <bool TA, bool TB, bool GLDS, int NJ, int WM = 1>
int launch_mode(const GemmArgs& g, int mode, hipStream_t s) {
    switch (mode) {
        case OUT_BF16: return launch_one<TA, TB, OUT_BF16, GLDS, NJ, WM>(g, s);
        case OUT_F32: return launch_one<TA, TB, OUT_F32, GLDS, NJ, WM>(g, s);
        case OUT_F32_ACC: return launch_one<TA, TB, OUT_F32_ACC, GLDS, NJ, WM>(g, s);
        case OUT_F32_RESID_BF16ROUND: return launch_one<TA, TB, OUT_F32_RESID_BF16ROUND, GLDS, NJ, WM>(g, s);
        case OUT_F32_RESID: return launch_one<TA, TB, OUT_F32_RESID, GLDS, NJ, WM>(g, s);
        case OUT_F32_ATOMIC: return launch_one<TA, TB, OUT_F32_ATOMIC, GLDS, NJ, WM>(g, s);
        case OUT_BF16_ACC:
            if constexpr (NJ == 2 && WM == 1) return launch_one<TA, TB, OUT_BF16_ACC, GLDS, NJ, WM>(g, s);
            mgen_set_error("mgen_gemm_bf16: OUT_BF16_ACC runs on 128x128 tiles only");
            return MGEN_EINVAL;
        default: mgen_set_error("mgen_gemm_bf16: unknown out_mode %d", mode); return MGEN_EINVAL;
    }
}

}  // namespace

// second operand pair of a product whose reduction runs over [A | A2] and [B | B2] (K, then K2 indices)
struct SecondPair { const void* A2; const void* B2; long lda2, ldb2; int K2; };

static int gemm_impl(const void* A, const void* B, void* C, const void* resid, const void* bias, int M, int N, int K,
                     long lda, long ldb, long ldc, long ldr, int transA, int transB, int out_mode, float alpha,
                     hipStream_t stream, int slabs = 0, long slab_stride = 0, const SecondPair* second = nullptr) {
    MGEN_REQUIRE(A && B && C, "mgen_gemm_bf16: null operand");
    if (bias)
        MGEN_REQUIRE(alpha == 1.0f && out_mode != OUT_F32_ACC,
                     "mgen_gemm_bf16_bias: a bias goes with alpha = 1 and a non-accumulating output");
    MGEN_REQUIRE(M > 0 && N > 0 && K > 0, "mgen_gemm_bf16: bad shape M=%d N=%d K=%d", M, N, K);
    MGEN_REQUIRE((lda % 8) == 0 && (ldb % 8) == 0, "mgen_gemm_bf16: lda/ldb must be multiples of 8 elements");
    MGEN_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && ((uintptr_t)C % 16) == 0,
                 "mgen_gemm_bf16: operands must be 16-byte aligned");
    // A k-contiguous operand is read in 8-element vectors along K: when K is not a multiple of 8 the row must be
    // allocated up to the next multiple and the extra columns must hold zeros. A transposed operand is read in
    // 8-element vectors along its output index (rows beyond K are never touched).
    const int K8 = (K + 7) / 8 * 8;
    if (!transA) MGEN_REQUIRE(lda >= K8, "mgen_gemm_bf16: A [M,K] needs lda >= K rounded up to 8");
    else MGEN_REQUIRE(lda >= ((M + 7) / 8) * 8, "mgen_gemm_bf16: A [K,M] needs lda >= M rounded up to 8");
    if (!transB) MGEN_REQUIRE(ldb >= K8, "mgen_gemm_bf16: B [N,K] needs ldb >= K rounded up to 8");
    else MGEN_REQUIRE(ldb >= ((N + 7) / 8) * 8, "mgen_gemm_bf16: B [K,N] needs ldb >= N rounded up to 8");
    MGEN_REQUIRE(ldc >= N, "mgen_gemm_bf16: ldc < N");
    if (out_mode == OUT_F32_RESID_BF16ROUND || out_mode == OUT_F32_RESID)
        MGEN_REQUIRE(resid != nullptr && ldr >= N, "mgen_gemm_bf16: residual mode needs resid and ldr>=N");
    GemmArgs g;
    g.A2 = g.B2 = nullptr; g.lda2 = g.ldb2 = 0; g.K1 = 0x7fffffff;
    if (second) {
        MGEN_REQUIRE(kSeg, "mgen_gemm_bf16: this build has no second operand pair");
        const int K2 = second->K2;
        MGEN_REQUIRE(second->A2 && second->B2 && K2 > 0 && K % BK == 0 && K2 % BK == 0,
                     "mgen_gemm_bf16_seg: both reduction lengths must be multiples of %d (got %d, %d)", BK, K, K2);
        MGEN_REQUIRE(!transA && slabs <= 1 && bias == nullptr && out_mode != OUT_F32_ACC && out_mode != OUT_BF16_ACC,
                     "mgen_gemm_bf16_seg: A row-major, no bias, no accumulating output");
        MGEN_REQUIRE(second->lda2 % 8 == 0 && second->ldb2 % 8 == 0 && second->lda2 >= K2 &&
                         second->ldb2 >= (transB ? ((N + 7) / 8) * 8 : K2),
                     "mgen_gemm_bf16_seg: bad leading dimensions of the second pair");
        MGEN_REQUIRE(((uintptr_t)second->A2 % 16) == 0 && ((uintptr_t)second->B2 % 16) == 0,
                     "mgen_gemm_bf16_seg: operands must be 16-byte aligned");
        g.A2 = (const bf16*)second->A2; g.B2 = (const bf16*)second->B2;
        g.lda2 = second->lda2; g.ldb2 = second->ldb2; g.K1 = K;
        K += K2;   // from here on: the total reduction length
    }
#if !MGEN_GEMM_SEG
    // long reduction: weight gradients, and the input-gradient / residual-projection products with few output columns,
    // on the output-stationary kernel (gemm_os.hip)
    if (!bias && !second && alpha == 1.0f && slabs <= 1 &&
        mgen_gemm_os_fits(M, N, K, lda, ldb, ldc, ldr, transA, transB, out_mode))
        return mgen_gemm_os_launch(A, B, C, resid, M, N, K, lda, ldb, ldc, ldr, transA, transB, out_mode, stream);
    // short reduction, many rows, bf16 output: the kernel that keeps the row panel in registers (gemm_rs.hip)
    if (!transA && !transB && out_mode == OUT_BF16 && !bias && !second && alpha == 1.0f && slabs <= 1 &&
        mgen_gemm_rs_fits(M, N, K, lda, ldb, ldc))
        return mgen_gemm_rs_launch(A, B, C, nullptr, M, N, 0, lda, ldb, ldc, 0, stream);
#endif
    g.act = nullptr; g.ldact = 0;
    g.A = (const bf16*)A; g.B = (const bf16*)B; g.C = C; g.R = (const float*)resid; g.bias = (const bf16*)bias;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr; g.alpha = alpha;
    g.nbm = (M + BM - 1) / BM; g.nbn = (N + BN - 1) / BN;
    MGEN_REQUIRE((out_mode >= 0 && out_mode <= OUT_F32_RESID) || out_mode == OUT_BF16_ACC,
                 "mgen_gemm_bf16: unknown out_mode %d", out_mode);
    const bool glds = (K % BK) == 0;
    // 256-wide workgroup tiles (64x128 per wave) when they still give every CU at least two workgroups
    static const int wide_env = [] { const char* e = getenv("MGEN_GEMM_WIDE"); return e ? atoi(e) : -1; }();
    // (measured: +5..13 % on the layer products and the lm_head weight gradient; the short-K, very wide lm_head
    // forward is store-bound per tile and stays on 128-wide tiles)
    bool wide = glds && slabs <= 1 && N >= 256 && (long)g.nbm * ((N + 255) / 256) >= 512 &&
                !(K <= 512 && N >= 65536);
    if (wide_env == 0 || out_mode == OUT_BF16_ACC) wide = false;
    // weight gradients (few output tiles, long reduction, split along K): the 128x256 tiles too. Measured +8..13 %
    // on the QKV / gate-up weight gradients on the machines where those run slowest, no change elsewhere.
    if (!wide && wide_env != 0 && wide_env != 1 && glds && slabs <= 1 && out_mode == OUT_F32_ACC && N >= 256 &&
        K >= 8192)
        wide = true;
    // 256x256 tiles, one workgroup per CU, for products with at least four column tiles and a tile for every CU
    // (MGEN_GEMM_WIDE=1: never). Measured against the 128x256 tiles: +12 % on 8192^3, +3..6 % on the decoder's
    // N >= 1024 products, -5..10 % on its N = 512 ones (two tile columns leave no second workgroup per CU to
    // cover the first K step and the epilogue).
    const long tiles256 = (long)((M + 255) / 256) * ((N + 255) / 256);
    // (Measured and not kept for the lm_head input gradient, N = 512 with K = 151936, whose two tile columns read the
    // long operand twice: 256x256 tiles, 8 % slower, and 128x512 tiles that read it once, 15 % slower - one workgroup
    // per CU leaves nobody to cover the first K step and the barrier waits. Also not kept: the four images of a stage
    // requested in front of the four k-steps' MFMAs instead of together behind the barrier: +2..6 % on the forward
    // products, -14..19 % on the products that read an operand with the transposing LDS read.)
    const bool big = glds && slabs <= 1 && wide_env != 0 && wide_env != 1 && tiles256 >= 256 && N >= 1024 &&
                     out_mode != OUT_BF16_ACC &&
                     (out_mode != OUT_F32_ACC || tiles256 >= 512);
    if (big) { wide = false; g.nbm = (M + 255) / 256; }
    if (wide || big) g.nbn = (N + 255) / 256;
    g.splitk = 1;
    g.slab_stride = 0;
    g.kper = ((K + BK - 1) / BK) * BK;
    // An accumulating product with few output tiles and a long reduction (weight gradients: K = tokens) is cut
    // along K so that the chip is filled; the slices add their tiles with fp32 atomics.
    const long tiles = (long)g.nbm * g.nbn;
    if (out_mode == OUT_F32_ACC && tiles < 512) {
        const int ksteps = (K + BK - 1) / BK;
        // 4 (128x256 tiles: 2) workgroups per CU x 256 CUs resident at once: fill them in ONE round
        int want = (int)((wide ? 512 : 1024) / tiles);
        int maxsplit = ksteps / 8;  // at least 8 K steps (512 reduction indices) per slice
        int sk = want < maxsplit ? want : maxsplit;
        if (sk > 1) {
            const int steps_per = (ksteps + sk - 1) / sk;
            g.kper = steps_per * BK;
            g.splitk = (ksteps + steps_per - 1) / steps_per;
            out_mode = OUT_F32_ATOMIC;
        }
    }
    if (slabs > 1) {
        // deterministic split-K: slice z stores its partial product (fp32) at C + z * slab_stride
        const int ksteps = (K + BK - 1) / BK;
        const int steps_per = (ksteps + slabs - 1) / slabs;
        MGEN_REQUIRE((ksteps + steps_per - 1) / steps_per == slabs,
                     "mgen_gemm_bf16_slabs: %d slabs do not divide %d K steps evenly enough", slabs, ksteps);
        g.kper = steps_per * BK;
        g.splitk = slabs;
        g.slab_stride = slab_stride;
    }
    MGEN_REQUIRE(tiles * g.splitk < (1L << 31), "mgen_gemm_bf16: grid too large");
#define MGEN_GEMM_GO(TA_, TB_)                                                          \
    if (big) return launch_mode<TA_, TB_, true, 4, 2>(g, out_mode, stream);             \
    return wide ? launch_mode<TA_, TB_, true, 4>(g, out_mode, stream)                   \
                : (glds ? launch_mode<TA_, TB_, true, 2>(g, out_mode, stream)           \
                        : launch_mode<TA_, TB_, false, 2>(g, out_mode, stream))
    if (!transA && !transB) { MGEN_GEMM_GO(false, false); }
    if (!transA && transB) { MGEN_GEMM_GO(false, true); }
#if !MGEN_GEMM_SEG
    if (transA && !transB) { MGEN_GEMM_GO(true, false); }
    MGEN_GEMM_GO(true, true);
#else
    mgen_set_error("mgen_gemm_bf16_seg: A must be row-major");
    return MGEN_EINVAL;
#endif
#undef MGEN_GEMM_GO
}

#if !MGEN_GEMM_SEG
// gu[M, 2 I] = x[M, K] w[2 I, K]^T (w = [gate rows; up rows]) and act[M, I] = silu(gate) * up in the product's epilogue
// (reference: Qwen3MLP.forward, transformers models/qwen3/modeling_qwen3.py:81-83, evaluated in bf16 under autocast).
// 256 x 256 tiles: K a multiple of 64, I a multiple of 128.
extern "C" int mgen_gemm_bf16_swiglu(const void* x, const void* w, void* gu, void* act, int M, int I, int K, long ldx,
                                     long ldw, long ldgu, long ldact, hipStream_t stream) {
    MGEN_REQUIRE(x && w && gu && act && M > 0 && I > 0 && K > 0, "mgen_gemm_bf16_swiglu: bad arguments");
    MGEN_REQUIRE(K % BK == 0 && I % 128 == 0, "mgen_gemm_bf16_swiglu: needs K %% 64 == 0 and I %% 128 == 0 (K=%d I=%d)", K, I);
    MGEN_REQUIRE(ldx % 8 == 0 && ldw % 8 == 0 && ldgu % 8 == 0 && ldact % 8 == 0 && ldx >= K && ldw >= K &&
                     ldgu >= 2 * I && ldact >= I, "mgen_gemm_bf16_swiglu: bad leading dimensions");
    MGEN_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)gu % 16) == 0 &&
                     ((uintptr_t)act % 16) == 0, "mgen_gemm_bf16_swiglu: operands must be 16-byte aligned");
    // the SwiGLU form of the register-panel kernel (MGEN_GEMM_RS=2: its plain form only)
    const char* rs_env = getenv("MGEN_GEMM_RS");
    if (I % 32 == 0 && !(rs_env && rs_env[0] == '2') && mgen_gemm_rs_fits(M, 2 * I, K, ldx, ldw, ldgu))
        return mgen_gemm_rs_launch(x, w, gu, act, M, 2 * I, I, ldx, ldw, ldgu, ldact, stream);
    GemmArgs g;
    g.A2 = g.B2 = nullptr; g.lda2 = g.ldb2 = 0; g.K1 = 0x7fffffff;
    g.A = (const bf16*)x; g.B = (const bf16*)w; g.C = gu; g.R = nullptr; g.bias = nullptr;
    g.M = M; g.N = 2 * I; g.K = K; g.lda = ldx; g.ldb = ldw; g.ldc = ldgu; g.ldr = 0; g.alpha = 1.0f;
    g.nbm = (M + 255) / 256; g.nbn = I / 128;
    g.splitk = 1; g.slab_stride = 0; g.kper = K;
    g.act = (bf16*)act; g.ldact = ldact;
    MGEN_REQUIRE((long)g.nbm * g.nbn < (1L << 31), "mgen_gemm_bf16_swiglu: grid too large");
    return launch_one<false, false, OUT_BF16_SWIGLU, true, 4, 2>(g, stream);
}
#endif

#if MGEN_GEMM_SEG
// C = [A | A2] [B | B2]^T in one accumulator: the reduction runs over K indices of (A, B), then K2 of (A2, B2). What
// it is for: a projection with a low-rank adapter, y = x W^T + t B^T (A2 = t [M][K2], B2 = adapter B [N][K2]) and its
// input gradient dx = dy W + dt A (transB: B = W stored [K][N], B2 = adapter A stored [K2][N]). K and K2 multiples
// of 64 (pad the rank with zero columns / rows); out_mode 0, 1, 3 or 4 as in mgen_gemm_bf16.
extern "C" int mgen_gemm_bf16_seg(const void* A, const void* B, const void* A2, const void* B2, void* C,
                                  const void* resid, int M, int N, int K, int K2, long lda, long ldb, long lda2,
                                  long ldb2, long ldc, long ldr, int transB, int out_mode, hipStream_t stream) {
    const SecondPair sp = {A2, B2, lda2, ldb2, K2};
    return gemm_impl(A, B, C, resid, nullptr, M, N, K, lda, ldb, ldc, ldr, 0, transB, out_mode, 1.0f, stream, 0, 0, &sp);
}
#else

extern "C" int mgen_gemm_bf16(const void* A, const void* B, void* C, const void* resid, int M, int N, int K,
                              long lda, long ldb, long ldc, long ldr, int transA, int transB, int out_mode,
                              float alpha, hipStream_t stream) {
    return gemm_impl(A, B, C, resid, nullptr, M, N, K, lda, ldb, ldc, ldr, transA, transB, out_mode, alpha, stream);
}

// Split-K without atomics for products with few output tiles (decode: M = number of streams): the K range is cut
// into `nslabs` slices and slice z stores its fp32 partial product [M][N] (row stride N) at slabs + z * M * N.
// mgen_gemm_slab_count gives the number of slices to ask for; the consumer sums them in slice order.
extern "C" int mgen_gemm_slab_count(int K, int max_slabs) {
    const int ksteps = (K + BK - 1) / BK;
    int steps_per = 2;  // 128 reduction indices per slice
    while ((ksteps + steps_per - 1) / steps_per > max_slabs) ++steps_per;
    return (ksteps + steps_per - 1) / steps_per;
}

extern "C" int mgen_gemm_bf16_slabs(const void* A, const void* B, float* slabs, int M, int N, int K, long lda, long ldb,
                                    int transA, int transB, int nslabs, hipStream_t stream) {
    MGEN_REQUIRE(nslabs >= 1 && N % 4 == 0, "mgen_gemm_bf16_slabs: need nslabs >= 1 and N %% 4 == 0");
    return gemm_impl(A, B, slabs, nullptr, nullptr, M, N, K, lda, ldb, N, 0, transA, transB, OUT_F32, 1.0f, stream,
                     nslabs, (long)M * N);
}

// Same product with a bias row (bf16 [N]) added to the fp32 accumulators before the output rounding: the fused
// torch.addmm(bias, x, W) of GPT-2's Conv1D layers (transformers pytorch_utils.py Conv1D.forward) under autocast.
extern "C" int mgen_gemm_bf16_bias(const void* A, const void* B, void* C, const void* resid, const void* bias, int M,
                                   int N, int K, long lda, long ldb, long ldc, long ldr, int transA, int transB,
                                   int out_mode, hipStream_t stream) {
    MGEN_REQUIRE(bias != nullptr, "mgen_gemm_bf16_bias: null bias");
    return gemm_impl(A, B, C, resid, bias, M, N, K, lda, ldb, ldc, ldr, transA, transB, out_mode, 1.0f, stream);
}
#endif
