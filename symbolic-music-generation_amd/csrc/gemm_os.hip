// Output-stationary bf16 GEMM for products with a long reduction: C[M,N] = op(A) op(B), fp32 accumulate.
//
// Serves the backward products of the linear layers and the two residual projections (reference: nn.Linear forward
// and its autograd backward as reached from transformers models/qwen3/modeling_qwen3.py:76-83,225-236):
//   weight gradients   dW = dY^T X      both operands stored [reduction][output index], fp32 output added in place
//   input gradients    dX = dY W        A rows reduction-contiguous, B stored [reduction][output index], bf16 output
//   o_proj / down_proj y = x W^T + res  both operands reduction-contiguous, fp32 output = residual + bf16(product)
//
// A workgroup of four waves (one per SIMD, 512 registers each) owns a 256 x 256 output tile, each wave a 128 x 128
// quarter in 256 accumulator registers, and streams the reduction range in steps of 32 through a four-slot LDS ring
// filled by LDS-DMA: each wave fills one of a step's four 8 KiB images (A rows 0..127 | A 128..255 | B 0..127 |
// B 128..255), three steps of loads in flight, one barrier per step, a wave's 8 DMA pieces spread between its MFMAs
// and the four waves' pieces at staggered positions. A step is waited for one barrier early, so the first fragments
// of the next step are requested before the barrier that opens it.
//
// Images. An operand stored [reduction][output index] keeps whole 256-byte row segments per step (32 rows x 256
// bytes, the IMG256 swizzle of common.h) and is read with the transposing LDS read. A reduction-contiguous operand
// contributes 128 rows x 64 bytes per image; 16-byte chunk c of row r sits at position c ^ ((r >> 2) & 3), which
// spreads the sixteen lanes a ds_read_b128 serves together over all banks.
//
// Few output tiles and a long reduction (layer weight gradients): the reduction range is cut into slices, one
// workgroup per tile and slice, added with fp32 atomics. Otherwise one workgroup per tile.
#include "lds_dma.h"
#include "gemm_os.h"

namespace {

constexpr int OS_BK = 32;
constexpr int OS_IMG = 8192;                 // one image of one step
constexpr int OS_SLOT = 4 * OS_IMG;
constexpr int OS_NSLOT = 4;
constexpr int OS_LDS = OS_NSLOT * OS_SLOT;   // 128 KiB

enum { OS_OUT_BF16 = 0, OS_OUT_F32 = 1, OS_OUT_F32_ACC = 2, OS_OUT_RESID_ROUND = 3, OS_OUT_RESID = 4 };

struct OsArgs {
    const bf16* A;
    const bf16* B;
    void* C;
    const float* R;
    long lda, ldb, ldc, ldr;
    int M, N, K;
    int nbm, nbn;
    int splitk, kper;   // reduction slices of kper indices (a multiple of OS_BK)
    int kseg;           // a workgroup walks its slice in segments of kseg indices (a multiple of OS_BK), each with a
                        // buffer descriptor of its own: 32-bit buffer offsets reach 4 GiB, a slice may be longer
};

// AKM / BKM: the operand is stored [reduction][output index] (else [output index][reduction])
// t = tile of the product g, kslice = reduction slice (the caller maps workgroups to them)
template <bool AKM, bool BKM, int OUT, int WV>
__device__ __forceinline__ void os_body(const OsArgs& g, char* smem, int lane, int wave, int t, int kslice) {
    constexpr bool ROWMAJOR_OUT = OUT != OS_OUT_F32 && OUT != OS_OUT_F32_ACC;
    const int tm = t / g.nbn, tn = t % g.nbn;
    const int m0 = tm * 256, n0 = tn * 256;
    const int kbeg0 = kslice * g.kper;
    const int kend0 = min(g.K, kbeg0 + g.kper);
    if (kend0 - kbeg0 < OS_BK) return;
    const unsigned lds0 = lds_addr(smem);

    // ---- compute role: rows 128 wm.. of the tile from A image wm, columns 128 wn.. from B image wn
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    for (int kbeg = kbeg0; kbeg < kend0; kbeg += g.kseg) {   // one segment of the slice per pass (usually one pass)
    const int kend = min(kend0, kbeg + g.kseg);
    const int nslot = (kend - kbeg) / OS_BK;

    // ---- loader role: this wave fills one of the four images of every step with 8 one-KiB pieces. The buffer
    // descriptor starts at the image's first element of this workgroup's slice, so offsets stay far below 4 GiB.
    const bool isB = wave >= 2;
    const bool km = isB ? BKM : AKM;
    const int ld2 = (int)((isB ? g.ldb : g.lda) * 2);
    const int o0 = isB ? n0 + 128 * (wave - 2) : m0 + 128 * wave;    // first output index of the image
    const int extent = isB ? g.N : g.M;
    const bf16* base = (isB ? g.B : g.A) + (km ? (long)kbeg * (ld2 / 2) + o0 : (long)o0 * (ld2 / 2) + kbeg);
    // valid bytes from base on: whole rows up to the operand's last row, the last one only up to its last element
    const long rows_left = km ? (long)(g.K - kbeg) : (long)(extent - o0);
    const long last_len = km ? (long)(extent - o0) : (long)(g.K - kbeg);
    const long span = rows_left > 0 ? ((rows_left - 1) * (ld2 / 2) + last_len) * 2 : 0;
    const u32x4 rs = dma_rsrc(base, (unsigned)(span > 0xFFFFFFFFL ? 0xFFFFFFFFL : (span > 0 ? span : 0)));
    int voff[4];
    int pstep;   // buffer offset between a wave's consecutive pieces of the same voff
    if (km) {
        // piece pc: reduction rows 4 pc .. 4 pc + 3, 256 bytes each (IMG256 swizzle on the source chunk)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int kr = 4 * q + (lane >> 4);
            voff[q] = kr * ld2 + 16 * ((lane & 15) ^ swz256(kr));
        }
        pstep = 16 * ld2;
    } else {
        // piece pc: output rows 16 pc .. 16 pc + 15, 64 bytes each; the chunk swizzle does not depend on pc
        const int v = (lane >> 2) * ld2 + 16 * ((lane & 3) ^ ((lane >> 4) & 3));
#pragma unroll
        for (int q = 0; q < 4; ++q) voff[q] = v + 16 * q * ld2;
        pstep = 64 * ld2;
    }
    const int sstep = km ? OS_BK * ld2 : OS_BK * 2;   // buffer offset from one step to the next
    auto piece = [&](int sl, int pc) {
        const int sc = min(sl, nslot - 1);   // past the end: a harmless reload, so that the wait counts stay fixed
        const int soff = __builtin_amdgcn_readfirstlane(sc * sstep + (pc >> 2) * pstep);
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (sl & 3) * OS_SLOT + wave * OS_IMG + pc * 1024);
        dma_piece(rs, voff[pc & 3], dst, soff);
    };

#pragma unroll
    for (int sl = 0; sl < 3; ++sl)
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) piece(sl, pc);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    // fragment of 32 output indices x 16 reduction indices (k-step ks of the slot) from an image
    auto frag = [&](const char* img, bool kmaj, int ks, int o) -> bf16x8 {
        if (kmaj) return frag_col_img256<false>(img, 16 * ks, 32 * o, lane);
        const int row = 32 * o + (lane & 31);
        return lds_read_b128(img, row * 64 + 16 * ((2 * ks + (lane >> 5)) ^ ((row >> 2) & 3)));
    };
    auto frags = [&](const char* slot, int ks, bf16x8 (&fa)[4], bf16x8 (&fb)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = frag(slot + wm * OS_IMG, AKM, ks, i);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = frag(slot + (2 + wn) * OS_IMG, BKM, ks, j);
    };
    // 16 MFMAs of one 16-deep step; DMA piece pc0 + (idx >> 2) of step `nsl` behind MFMA idx with idx & 3 == WV.
    // Weight gradients: D[row = m][col = n], a register = 32 consecutive n of two rows (atomics by row segments).
    // Row-major outputs: D[row = n][col = m], a lane = one row m and groups of 4 consecutive n.
    auto step = [&](const bf16x8 (&fa)[4], const bf16x8 (&fb)[4], int nsl, int pc0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = ROWMAJOR_OUT ? mfma32(fb[j], fa[i], acc[i][j]) : mfma32(fa[i], fb[j], acc[i][j]);
                if (((4 * i + j) & 3) == WV) piece(nsl, pc0 + ((4 * i + j) >> 2));
            }
    };

    bf16x8 fa0[4], fb0[4], fa1[4], fb1[4];
    frags(smem, 0, fa0, fb0);
    for (int s = 0; s < nslot; ++s) {
        // This wave's pieces of step s + 1 have landed (younger: the 8 pieces of step s + 2). Every wave's LDS reads of
        // step s - 1 have returned (they fed MFMAs that are issued); past the barrier its ring slot takes step s + 3,
        // and step s + 1 is complete for every wave: its first fragments may be requested during this step.
        dma_wait<8>();
        asm volatile("s_barrier" ::: "memory");
        // the accumulators live in the accumulator half of the register file (some instances would otherwise be
        // given vector registers for them and spill everything else)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+a"(acc[i][j]));
        const char* cur = smem + (s & 3) * OS_SLOT;
        const char* nxt = smem + ((s + 1) & 3) * OS_SLOT;
        frags(cur, 1, fa1, fb1);
        step(fa0, fb0, s + 3, 0);
        frags(nxt, 0, fa0, fb0);
        step(fa1, fb1, s + 3, 4);
    }
    // no LDS-DMA may be pending when the wave ends (or refills the ring for the next segment); the row-major outputs
    // reuse the ring as a staging area once every wave has left the loop
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }   // segments

    // the output addresses are derived from a lane index taken here, after the loop: from the function's argument the
    // compiler would compute them in front of the loop and carry (or spill) them, or the index itself, through it
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    const int h = lane >> 5, r = lane & 31;
    if constexpr (!ROWMAJOR_OUT) {
        float* C = reinterpret_cast<float*>(g.C);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + 128 * wn + 32 * j + r;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + 128 * wm + 32 * i + acc_row(e, lane);
                    if (m < g.M && n < g.N) {
                        float* c = C + (long)m * g.ldc + n;
                        if (OUT == OS_OUT_F32_ACC) atomicAdd(c, acc[i][j][e]);
                        else *c = acc[i][j][e];
                    }
                }
            }
    } else {
        // Row-major outputs: the wave's quarter goes through its 32 KiB of the ring as fp32, 64 rows x 512 bytes at a
        // time (16-byte chunks swizzled by row), and is read back as whole row segments. bf16 output: rounded on the
        // way out, 16 bytes per lane; fp32 output = residual + (bf16-rounded) product.
        char* st = smem + wave * 32768;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int row = 32 * i2 + r, ch = 8 * j + 2 * g4 + h;   // 16-byte chunk = 4 floats
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = acc[2 * half + i2][j][4 * g4 + e];
                        *reinterpret_cast<f32x4*>(st + row * 512 + ((ch ^ (row & 31)) << 4)) = o;
                    }
            if constexpr (OUT == OS_OUT_BF16) {
                bf16* C = reinterpret_cast<bf16*>(g.C);
#pragma unroll 4
                for (int it = 0; it < 16; ++it) {
                    const int row = it * 4 + (lane >> 4), c8 = lane & 15;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(st + row * 512 + (((2 * c8) ^ (row & 31)) << 4));
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(st + row * 512 + (((2 * c8 + 1) ^ (row & 31)) << 4));
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[e] = (bf16)v0[e]; o[4 + e] = (bf16)v1[e]; }
                    const int m = m0 + 128 * wm + 64 * half + row, n = n0 + 128 * wn + c8 * 8;
                    if (m < g.M && n < g.N) *reinterpret_cast<bf16x8*>(C + (long)m * g.ldc + n) = o;
                }
            } else {
                float* C = reinterpret_cast<float*>(g.C);
#pragma unroll 8
                for (int it = 0; it < 32; ++it) {
                    const int row = it * 2 + (lane >> 5), ch = lane & 31;
                    f32x4 v = *reinterpret_cast<const f32x4*>(st + row * 512 + ((ch ^ (row & 31)) << 4));
                    const int m = m0 + 128 * wm + 64 * half + row, n = n0 + 128 * wn + ch * 4;
                    if (m < g.M && n < g.N) {
                        const f32x4 res = *reinterpret_cast<const f32x4*>(g.R + (long)m * g.ldr + n);
                        if (OUT == OS_OUT_RESID_ROUND) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = round_bf(v[e]);
                        }
                        v += res;
                        *reinterpret_cast<f32x4*>(C + (long)m * g.ldc + n) = v;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the first half has been read out
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Input-gradient form with whole cache lines on both sides: A rows reduction-contiguous, B stored [reduction][output
// index], bf16 output. A 32-deep step of a reduction-contiguous operand is only 64 bytes of every row - half a cache
// line per request. Here the A images cover TWO steps (128 rows x 128 bytes, the IMG128 layout of common.h) in a ring
// of three per row half, filled by waves 0 and 1 with 16 one-KiB pieces (8 rows x 128 bytes) during every even step;
// the B images stay one step deep in a ring of four, filled by waves 2 and 3 with 8 pieces per step. 160 KiB of LDS.
// ---------------------------------------------------------------------------------------------------------------
constexpr int KC_DIMG = 16384;                       // 128 rows x 64 reduction indices
constexpr int KC_AREGION = 2 * 3 * KC_DIMG;          // [row half][ring of 3]
constexpr int KC_BSLOT = 2 * OS_IMG;                 // B columns 0..127 | 128..255 of one step
constexpr int KC_LDS = KC_AREGION + 4 * KC_BSLOT;    // 160 KiB

template <int WV>
__device__ __forceinline__ void os_body_kc(const OsArgs& g, char* smem, int lane, int wave) {
    constexpr bool LOADS_A = WV < 2;
    const int nt = g.nbm * g.nbn;
    const int t = xcd_remap(blockIdx.x, nt);
    const int tm = t / g.nbn, tn = t % g.nbn;
    const int m0 = tm * 256, n0 = tn * 256;
    const int ndbl = g.K / 64;
    const unsigned lds0 = lds_addr(smem);

    // ---- loader role
    const int ld2 = (int)((LOADS_A ? g.lda : g.ldb) * 2);
    u32x4 rs;
    int voff[4];
    if (LOADS_A) {
        const int o0 = m0 + 128 * wave;
        rs = dma_rsrc(g.A + (long)o0 * g.lda, (unsigned)((127L * g.lda + g.K) * 2));
        // piece p (0..15): rows 8 p .. 8 p + 7; the chunk swizzle (row >> 1) & 7 = 4 (p & 1) + (lane >> 4)
#pragma unroll
        for (int q = 0; q < 2; ++q)
            voff[q] = (8 * q + (lane >> 3)) * ld2 + 16 * ((lane & 7) ^ (4 * q + (lane >> 4)));
        voff[2] = voff[3] = 0;
    } else {
        const int o0 = n0 + 128 * (wave - 2);
        rs = dma_rsrc(g.B + o0, (unsigned)(((long)(g.K - 1) * g.ldb + (g.N - o0)) * 2));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int kr = 4 * q + (lane >> 4);
            voff[q] = kr * ld2 + 16 * ((lane & 15) ^ swz256(kr));
        }
    }
    // A: piece p of the double step ds into ring buffer ds % 3 (the caller passes the buffer)
    auto piece_a = [&](int ds, int buf, int p) {
        const int dc = min(ds, ndbl - 1);   // past the end: a harmless reload, so that the wait counts stay fixed
        const int soff = __builtin_amdgcn_readfirstlane(dc * 128 + (p >> 1) * 16 * ld2);
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (wave * 3 + buf) * KC_DIMG + p * 1024);
        dma_piece(rs, voff[p & 1], dst, soff);
    };
    // B: piece pc (0..7) of step sl into ring slot sl & 3
    auto piece_b = [&](int sl, int pc) {
        const int sc = min(sl, 2 * ndbl - 1);
        const int soff = __builtin_amdgcn_readfirstlane((sc * OS_BK + 16 * (pc >> 2)) * ld2);
        const unsigned dst =
            __builtin_amdgcn_readfirstlane(lds0 + KC_AREGION + (sl & 3) * KC_BSLOT + (wave - 2) * OS_IMG + pc * 1024);
        dma_piece(rs, voff[pc & 3], dst, soff);
    };

    // ---- compute role
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (LOADS_A) {
#pragma unroll
        for (int ds = 0; ds < 2; ++ds)
#pragma unroll
            for (int p = 0; p < 16; ++p) piece_a(ds, ds, p);
    } else {
#pragma unroll
        for (int sl = 0; sl < 3; ++sl)
#pragma unroll
            for (int pc = 0; pc < 8; ++pc) piece_b(sl, pc);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    const char* abase = smem + wm * 3 * KC_DIMG;
    const char* bbase = smem + KC_AREGION + wn * OS_IMG;
    // fragments of k-step kq (0..3) of the double image in ring buffer buf / of k-step ks (0, 1) of B slot sl
    auto frags = [&](int buf, int kq, int sl, int ks, bf16x8 (&fa)[4], bf16x8 (&fb)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = frag_row_img128(abase + buf * KC_DIMG, 32 * i, kq, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = frag_col_img256<false>(bbase + (sl & 3) * KC_BSLOT, 16 * ks, 32 * j, lane);
    };
    // 16 MFMAs (D[row = n][col = m]); this wave's DMA pieces between them: an A loader issues 8 of the 16 pieces of
    // double step `ads` per call during even steps (behind every other MFMA), a B loader 4 of the 8 pieces of step
    // `bsl` per call in every step (behind every fourth)
    auto step = [&](const bf16x8 (&fa)[4], const bf16x8 (&fb)[4], bool even, int call, int ads, int abuf, int bsl) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = mfma32(fb[j], fa[i], acc[i][j]);
                const int idx = 4 * i + j;
                if (LOADS_A) {
                    if (even && (idx & 1) == (WV & 1)) piece_a(ads, abuf, 8 * call + (idx >> 1));
                } else {
                    if ((idx & 3) == WV) piece_b(bsl, 4 * call + (idx >> 2));
                }
            }
    };

    bf16x8 fa0[4], fb0[4], fa1[4], fb1[4];
    frags(0, 0, 0, 0, fa0, fb0);
    int bc = 0, bn = 1, bi = 2;   // ring buffers of the double steps d, d + 1, d + 2
    for (int d = 0; d < ndbl; ++d) {
        const int s = 2 * d;
        // Even step 2 d. A loaders: younger than what must have landed are the 16 pieces of double step d + 1; B
        // loaders: the 8 pieces of step s + 2. Everything a wave may read before the next barrier (the rest of this
        // step and the first fragments of step s + 1) was complete for every wave at the last barrier.
        if (LOADS_A) dma_wait<16>(); else dma_wait<8>();
        asm volatile("s_barrier" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+a"(acc[i][j]));
        frags(bc, 1, s, 1, fa1, fb1);
        step(fa0, fb0, true, 0, d + 2, bi, s + 3);
        frags(bc, 2, s + 1, 0, fa0, fb0);
        step(fa1, fb1, true, 1, d + 2, bi, s + 3);
        // Odd step 2 d + 1: double step d + 1 (issued during step 2 d - 2) must be complete before this barrier, the
        // 16 pieces of double step d + 2 are younger
        if (LOADS_A) dma_wait<16>(); else dma_wait<8>();
        asm volatile("s_barrier" ::: "memory");
        frags(bc, 3, s + 1, 1, fa1, fb1);
        step(fa0, fb0, false, 0, 0, 0, s + 4);
        frags(bn, 0, s + 2, 0, fa0, fb0);
        step(fa1, fb1, false, 1, 0, 0, s + 4);
        const int tmp = bc; bc = bn; bn = bi; bi = tmp;
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // ---- bf16 output through the wave's 32 KiB: fp32 quarter-tile halves of 64 rows x 512 bytes, rounded on the
    // way out, whole 256-byte row segments
    asm volatile("" : "+v"(lane));
    const int h = lane >> 5, r = lane & 31;
    char* st = smem + wave * 32768;
    bf16* C = reinterpret_cast<bf16*>(g.C);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int row = 32 * i2 + r, ch = 8 * j + 2 * g4 + h;
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = acc[2 * half + i2][j][4 * g4 + e];
                    *reinterpret_cast<f32x4*>(st + row * 512 + ((ch ^ (row & 31)) << 4)) = o;
                }
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int row = it * 4 + (lane >> 4), c8 = lane & 15;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(st + row * 512 + (((2 * c8) ^ (row & 31)) << 4));
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(st + row * 512 + (((2 * c8 + 1) ^ (row & 31)) << 4));
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] = (bf16)v0[e]; o[4 + e] = (bf16)v1[e]; }
            const int m = m0 + 128 * wm + 64 * half + row, n = n0 + 128 * wn + c8 * 8;
            if (m < g.M && n < g.N) *reinterpret_cast<bf16x8*>(C + (long)m * g.ldc + n) = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

__global__ __launch_bounds__(256, 1) void gemm_os_kc_kernel(OsArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    switch (wave) {
        case 0: os_body_kc<0>(g, smem, lane, wave); break;
        case 1: os_body_kc<1>(g, smem, lane, wave); break;
        case 2: os_body_kc<2>(g, smem, lane, wave); break;
        default: os_body_kc<3>(g, smem, lane, wave); break;
    }
}

template <bool AKM, bool BKM, int OUT>
__global__ __launch_bounds__(256, 1) void gemm_os_kernel(OsArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // slice-major: the tiles of one reduction slice are neighbours and share that slice of both operands in L2;
    // tiles of one tile row (same A rows) are neighbours too
    const int nt = g.nbm * g.nbn;
    const int wid = xcd_remap(blockIdx.x, nt * g.splitk);
    const int kslice = wid / nt, t = wid % nt;
    // four copies of the loop that differ only in where the wave's DMA pieces sit between its MFMAs
    switch (wave) {
        case 0: os_body<AKM, BKM, OUT, 0>(g, smem, lane, wave, t, kslice); break;
        case 1: os_body<AKM, BKM, OUT, 1>(g, smem, lane, wave, t, kslice); break;
        case 2: os_body<AKM, BKM, OUT, 2>(g, smem, lane, wave, t, kslice); break;
        default: os_body<AKM, BKM, OUT, 3>(g, smem, lane, wave, t, kslice); break;
    }
}

// Several weight gradients over the same reduction range (the four of a decoder layer) as ONE launch: their tiles are
// numbered through, so the reduction range is cut into as few slices as fill the chip once - a quarter of the slices,
// and of the fp32 atomics, that four separate launches need.
constexpr int OS_GROUP_MAX = 4;
struct OsGroupArgs {
    OsArgs p[OS_GROUP_MAX];
    int first_tile[OS_GROUP_MAX + 1];
    int n;
};

__global__ __launch_bounds__(256, 1) void gemm_os_group_kernel(OsGroupArgs gg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nt = gg.first_tile[gg.n];
    const int wid = xcd_remap(blockIdx.x, nt * gg.p[0].splitk);
    const int kslice = wid / nt;
    int t = wid % nt;
    int which = 0;
#pragma unroll
    for (int q = 1; q < OS_GROUP_MAX; ++q)
        if (q < gg.n && t >= gg.first_tile[q]) which = q;
    t -= gg.first_tile[which];
    const OsArgs& g = gg.p[which];
    switch (wave) {
        case 0: os_body<true, true, OS_OUT_F32_ACC, 0>(g, smem, lane, wave, t, kslice); break;
        case 1: os_body<true, true, OS_OUT_F32_ACC, 1>(g, smem, lane, wave, t, kslice); break;
        case 2: os_body<true, true, OS_OUT_F32_ACC, 2>(g, smem, lane, wave, t, kslice); break;
        default: os_body<true, true, OS_OUT_F32_ACC, 3>(g, smem, lane, wave, t, kslice); break;
    }
}

template <bool AKM, bool BKM, int OUT>
int os_launch(const OsArgs& g, hipStream_t stream) {
    static bool asked[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (!asked[dev]) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_os_kernel<AKM, BKM, OUT>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, OS_LDS);
        if (e != hipSuccess) {
            mgen_set_error("mgen_gemm_bf16: cannot reserve %d bytes of LDS: %s", OS_LDS, hipGetErrorString(e));
            return MGEN_ELAUNCH;
        }
        asked[dev] = true;
    }
    hipLaunchKernelGGL((gemm_os_kernel<AKM, BKM, OUT>), dim3(g.nbm * g.nbn * g.splitk), dim3(256), OS_LDS, stream, g);
    return mgen_check_launch("mgen_gemm_bf16");
}

}  // namespace

// Number of reduction slices of a product that is added to its output. Fewer tiles than CUs: as many slices as fill one
// round of workgroups. More: the count (at most 4) that leaves the least of the last round of workgroups empty - 576
// tiles (the four weight gradients of a layer of the 650M model) are 2.25 rounds on 256 CUs, i.e. 3; in 4 slices they
// are 9 full rounds. Slices keep at least 1024 reduction indices.
static int os_pick_slices(int tiles, int steps, bool many_ok) {
    const int maxsk = steps / 32 > 1 ? steps / 32 : 1;
    if (tiles < 256) {
        const int sk = 256 / tiles;
        return sk > maxsk ? maxsk : sk;
    }
    if (!many_ok) return 1;
    int best = 1;
    double best_eff = 0.0;
    const int maxmany = steps / 192;   // (slices of at least 6144 indices here: 4 slices of 4096 measured a loss)
    for (int sk = 1; sk <= 4 && sk <= maxsk && (sk == 1 || sk <= maxmany); ++sk) {
        const double r = (double)tiles * sk / 256.0;
        const double eff = r / (double)(long)(r + 0.999999);
        if (eff > best_eff + 0.03) {   // (more slices are more atomics: only for a real gain)
            best_eff = eff;
            best = sk;
        }
    }
    return best;
}

bool mgen_gemm_os_fits(int M, int N, int K, long lda, long ldb, long ldc, long ldr, int transA, int transB,
                       int out_mode) {
    const char* e = getenv("MGEN_GEMM_OS");   // MGEN_GEMM_OS=0: never (same-process A/B runs)
    if (e && e[0] == '0') return false;
    if (K % OS_BK != 0 || K < 1024 || M < 256 || N < 256 || (lda % 8) || (ldb % 8)) return false;
    const bool wgrad = transA && transB && (out_mode == OS_OUT_F32_ACC || out_mode == OS_OUT_F32);
    const bool dgrad = !transA && transB && out_mode == OS_OUT_BF16;
    const bool proj = !transA && !transB && (out_mode == OS_OUT_RESID_ROUND || out_mode == OS_OUT_RESID);
    // (the residual projections, both operands reduction-contiguous, measured 10-20 % slower here than on the
    // tile-loop kernel: their 64-byte row pieces double the number of cache-line requests; MGEN_GEMM_OS=2 routes them)
    // (MGEN_GEMM_OS=4: weight gradients only, for same-box comparisons)
    if (!(wgrad || (dgrad && !(e && e[0] == '4')) || (proj && e && e[0] == '2'))) return false;
    if (wgrad) {
        if (K < 2048 || (M % 8) || (N % 8)) return false;
        // buffer offsets are 32-bit: a workgroup walks its reduction range in segments below 4 GiB of either operand,
        // each with a descriptor of its own; a segment must hold at least 1024 rows
        const long ldmax = lda > ldb ? lda : ldb;
        if (1024L * ldmax * 2 >= (1L << 32)) return false;
        return true;
    }
    // the products with few output columns: whole 256-row tiles (rows beyond the operand would be read), at least as
    // many tiles as CUs
    // (input gradients up to MGEN_GEMM_OS_NMAX output columns, default 8192: the column tiles of a row panel are
    // neighbours in the XCD-aware order and share the panel's rows in L2; the 650M model's down-projection input
    // gradient, 32 768 x 6144 x 1536, runs 1.2 times faster here than on the tile loop)
    static const int nmax = [] { const char* v = getenv("MGEN_GEMM_OS_NMAX"); return v ? atoi(v) : 8192; }();
    if (M % 256 != 0 || N % 256 != 0 || N > (dgrad ? nmax : 1024) || (long)(M / 256) * (N / 256) < 256) return false;
    if ((ldc % 8) || (proj && ((ldc % 4) || (ldr % 4)))) return false;
    if (256L * lda * 2 >= (1L << 32)) return false;
    if (dgrad ? (long)K * ldb * 2 >= (1L << 32) : 256L * ldb * 2 >= (1L << 32)) return false;
    return true;
}

int mgen_gemm_os_launch(const void* A, const void* B, void* C, const void* resid, int M, int N, int K, long lda,
                        long ldb, long ldc, long ldr, int transA, int transB, int out_mode, hipStream_t stream) {
    OsArgs g;
    g.A = (const bf16*)A; g.B = (const bf16*)B; g.C = C; g.R = (const float*)resid;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr; g.M = M; g.N = N; g.K = K;
    g.nbm = (M + 255) / 256; g.nbn = (N + 255) / 256;
    // one round of workgroups on 256 CUs: cut the reduction range while the tiles alone do not fill the chip
    // (slices of at least 1024 indices; only a product that is added to its output can be cut)
    const int tiles = g.nbm * g.nbn;
    const int steps = K / OS_BK;
    int sk = 1;
    if (out_mode == OS_OUT_F32_ACC) {
        static const bool many = [] { const char* v = getenv("MGEN_GEMM_OS_SK_MANY"); return v && v[0] == '1'; }();
        sk = os_pick_slices(tiles, steps, many);
    }
    const int per = (steps + sk - 1) / sk;
    g.kper = per * OS_BK;
    g.splitk = (steps + per - 1) / per;
    g.kseg = g.kper;
    if (transA && transB) {   // both operands stored [reduction][...]: segments below 4 GiB (32-bit buffer offsets)
        const long ldmax = lda > ldb ? lda : ldb;
        const long max_rows = ((1L << 32) - 1) / (ldmax * 2) / OS_BK * OS_BK;
        if (g.kseg > max_rows) g.kseg = (int)max_rows;
    }
    if (transA && transB)
        return out_mode == OS_OUT_F32_ACC ? os_launch<true, true, OS_OUT_F32_ACC>(g, stream)
                                          : os_launch<true, true, OS_OUT_F32>(g, stream);
    if (transB) {
        const char* e = getenv("MGEN_GEMM_OS");   // 3: the one-step A images (measurements)
        if (K % 64 != 0 || (e && e[0] == '3')) return os_launch<false, true, OS_OUT_BF16>(g, stream);
        static bool asked[16] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
        if (!asked[dev]) {
            const hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_os_kc_kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, KC_LDS);
            if (err != hipSuccess) {
                mgen_set_error("mgen_gemm_bf16: cannot reserve %d bytes of LDS: %s", KC_LDS, hipGetErrorString(err));
                return MGEN_ELAUNCH;
            }
            asked[dev] = true;
        }
        hipLaunchKernelGGL(gemm_os_kc_kernel, dim3(g.nbm * g.nbn), dim3(256), KC_LDS, stream, g);
        return mgen_check_launch("mgen_gemm_bf16");
    }
    return out_mode == OS_OUT_RESID_ROUND ? os_launch<false, false, OS_OUT_RESID_ROUND>(g, stream)
                                          : os_launch<false, false, OS_OUT_RESID>(g, stream);
}

// dW_i (fp32, added in place) += dY_i^T X_i for n <= 4 products over the same K tokens, one launch
extern "C" int mgen_gemm_bf16_wgrad_group(int n, const void* const* A, const void* const* B, float* const* C,
                                          const int* M, const int* N, int K, const long* lda, const long* ldb,
                                          const long* ldc, hipStream_t stream) {
    MGEN_REQUIRE(n >= 1 && n <= OS_GROUP_MAX && A && B && C && M && N && lda && ldb && ldc,
                 "mgen_gemm_bf16_wgrad_group: bad arguments");
    OsGroupArgs gg;
    gg.n = n;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        MGEN_REQUIRE(A[i] && B[i] && C[i], "mgen_gemm_bf16_wgrad_group: null operand");
        MGEN_REQUIRE(((uintptr_t)A[i] % 16) == 0 && ((uintptr_t)B[i] % 16) == 0,
                     "mgen_gemm_bf16_wgrad_group: operands must be 16-byte aligned");
        MGEN_REQUIRE(mgen_gemm_os_fits(M[i], N[i], K, lda[i], ldb[i], ldc[i], 0, 1, 1, OS_OUT_F32_ACC) &&
                         lda[i] >= M[i] && ldb[i] >= N[i] && ldc[i] >= N[i],
                     "mgen_gemm_bf16_wgrad_group: product %d (M=%d N=%d K=%d) does not fit the output-stationary kernel",
                     i, M[i], N[i], K);
        OsArgs& g = gg.p[i];
        g.A = (const bf16*)A[i]; g.B = (const bf16*)B[i]; g.C = C[i]; g.R = nullptr;
        g.lda = lda[i]; g.ldb = ldb[i]; g.ldc = ldc[i]; g.ldr = 0; g.M = M[i]; g.N = N[i]; g.K = K;
        g.nbm = (M[i] + 255) / 256; g.nbn = (N[i] + 255) / 256;
        gg.first_tile[i] = tiles;
        tiles += g.nbm * g.nbn;
    }
    for (int i = n; i <= OS_GROUP_MAX; ++i) gg.first_tile[i] = tiles;
    const int steps = K / OS_BK;
    static const bool many = [] { const char* v = getenv("MGEN_GEMM_OS_SK_MANY"); return !(v && v[0] == '0'); }();
    int sk = os_pick_slices(tiles, steps, many);
    const int per = (steps + sk - 1) / sk;
    const int splitk = (steps + per - 1) / per;
    for (int i = 0; i < n; ++i) {
        gg.p[i].kper = per * OS_BK;
        gg.p[i].splitk = splitk;
        // segments of a slice below 4 GiB of either operand (32-bit buffer offsets)
        const long ldmax = lda[i] > ldb[i] ? lda[i] : ldb[i];
        const long max_rows = ((1L << 32) - 1) / (ldmax * 2) / OS_BK * OS_BK;
        gg.p[i].kseg = per * OS_BK > max_rows ? (int)max_rows : per * OS_BK;
    }
    static bool asked[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (!asked[dev]) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_os_group_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, OS_LDS);
        if (e != hipSuccess) {
            mgen_set_error("mgen_gemm_bf16_wgrad_group: cannot reserve %d bytes of LDS: %s", OS_LDS, hipGetErrorString(e));
            return MGEN_ELAUNCH;
        }
        asked[dev] = true;
    }
    hipLaunchKernelGGL(gemm_os_group_kernel, dim3(tiles * splitk), dim3(256), OS_LDS, stream, gg);
    return mgen_check_launch("mgen_gemm_bf16_wgrad_group");
}
