// bf16 GEMM for products with a SHORT reduction (K = 512) and many rows: C[M,N] = A[M,512] * B[N,512]^T.
//
// Serves the forward projections out of the residual stream of the d = 512 decoders (QKV, gate/up with the SwiGLU
// gate, the lm_head logits; reference: nn.Linear calls in transformers models/qwen3/modeling_qwen3.py:76-83,225-236
// and the lm_head of Qwen3ForCausalLM.forward).
//
// With K = 512 a classic tile loop spends a third of its time outside the K loop and moves every operand byte
// through LDS once per 256 output columns. Here the token operand never goes back to memory: a workgroup of four
// waves (one per SIMD, 512 registers each) owns 256 token rows, and each wave keeps ITS 64 rows x 512 reduction
// indices as 2 x 32 MFMA operand fragments in 256 registers for the whole kernel. The weight rows stream past in
// blocks of 32 rows (32 KiB) through a four-slot LDS ring filled by LDS-DMA; every fragment read from LDS feeds two
// MFMAs (the wave's two 32-row token tiles), the accumulators of a block are a single chain of 32 dependent
// v_mfma_f32_32x32x16_bf16 per tile. One barrier per block; three blocks of loads in flight; the output of two
// blocks (64 rows x 64 columns per wave) goes through a per-wave LDS image so that whole 128-byte row segments
// are stored. All workgroups stream the same weight rows at the same pace, so the stream is served by the L2.
//
// LDS image of a 32-row block (also used for the 128-row pieces of the token panel on their way into the
// registers): row r = 1024 bytes = 64 chunks of 16 bytes; chunk c sits at position (c & 48) | ((c & 15) ^ (r & 15)).
// The sixteen lanes that a ds_read_b128 serves together hold sixteen different values of r & 15, i.e. sixteen
// different 16-byte bank groups. The LDS-DMA writes a row as one linear KiB, so the permutation is applied to each
// lane's SOURCE address.
#include "lds_dma.h"
#include "gemm_rs.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(4))) float rs_f32x4;
__device__ __forceinline__ rs_f32x4 mfma16(bf16x8 a, bf16x8 b, rs_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

constexpr int RS_K = 512, RS_KS = RS_K / 16;
constexpr int RS_ROWS = 256;                       // token rows per workgroup
constexpr int RS_SLOT = 32 * RS_K * 2;             // one block of 32 rows: 32 KiB
constexpr int RS_NSLOT = 4;
#ifndef MGEN_RS_M32
#define MGEN_RS_M32 0
#endif
// The MFMA shape: v_mfma_f32_16x16x32_bf16 (default) or v_mfma_f32_32x32x16_bf16 (-DMGEN_RS_M32=1). Same flops per
// cycle, same LDS traffic, same registers; the chip holds a higher clock on the 16x16x32 shape under this load.
constexpr bool RS_M16 = !MGEN_RS_M32;
constexpr int RS_AHEAD = 3;                        // LDS fragment reads in flight ahead of their MFMAs
constexpr int RS_STAGE = 8192;                     // per-wave output image: 64 rows x 128 bytes
constexpr int RS_LDS = RS_NSLOT * RS_SLOT + 4 * RS_STAGE;   // 160 KiB

struct RsArgs {
    const bf16* A;
    const bf16* B;
    bf16* C;
    bf16* act;
    long lda, ldb, ldc, ldact;
    int M, N;             // N = output columns of C (SwiGLU form: 2 I)
    int I;                // SwiGLU form: gate rows [0, I), up rows [I, 2 I) of B
    int nrp;              // 256-row panels
    int parts;            // the column range is cut into `parts` runs of `ppp` 64-column pairs
    int ppp, npairs;
    int xcd_map;          // grid is a multiple of 8: workgroups of one XCD share a run
};

// The wave's 8 rows (wave + 4 q + 16 half) of a 32-row block whose first row starts `soff` bytes into the buffer,
// into the slot at LDS byte address `dst0` (+ 1024 * row). One asm statement: the compiler never learns that LDS
// writes are pending, all waiting is done by hand with counted s_waitcnt vmcnt.
__device__ __forceinline__ void rs_issue(const u32x4& rsrc, const int (&voff)[4], unsigned dst0, int soff, int step16) {
    asm volatile(
        "s_mov_b32 m0, %[dst]\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %[v0], %[rs], %[so] offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "buffer_load_dwordx4 %[v1], %[rs], %[so] offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "buffer_load_dwordx4 %[v2], %[rs], %[so] offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "buffer_load_dwordx4 %[v3], %[rs], %[so] offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\ts_add_u32 %[so], %[so], %[st]\n\t"
        "buffer_load_dwordx4 %[v0], %[rs], %[so] offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "buffer_load_dwordx4 %[v1], %[rs], %[so] offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "buffer_load_dwordx4 %[v2], %[rs], %[so] offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "buffer_load_dwordx4 %[v3], %[rs], %[so] offen lds"
        : [so] "+s"(soff)
        : [dst] "s"(dst0), [v0] "v"(voff[0]), [v1] "v"(voff[1]), [v2] "v"(voff[2]), [v3] "v"(voff[3]), [rs] "s"(rsrc),
          [st] "s"(step16)
        : "memory", "m0", "scc");
}

// SW: B = [gate rows; up rows] (2 I rows), C = [gate | up] and act = silu(gate) * up
// DBG (timing experiments only, results are wrong): 1 no global stores, 2 no output code at all, 4 no loads inside
// the loop, 8 no MFMAs
template <bool SW, int DBG = 0>
__global__ __launch_bounds__(256, 1) void gemm_rs_kernel(RsArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r = lane & 31;
    constexpr int S = (DBG & 3) ? 0 : (SW ? 12 : 8);   // global stores of a wave per 64-column pair

    int part, rp;
    if (g.xcd_map) {
        const int x = blockIdx.x & 7;
        part = x % g.parts;
        rp = (blockIdx.x >> 3) * (8 / g.parts) + x / g.parts;
    } else {
        part = blockIdx.x % g.parts;
        rp = blockIdx.x / g.parts;
    }
    const int p0 = part * g.ppp;
    const int p1 = min(g.npairs, p0 + g.ppp);
    const int nblk = 2 * (p1 - p0);
    if (nblk <= 0 || rp >= g.nrp) return;

    const unsigned lds0 = lds_addr(smem);
    const int ldb2 = (int)(g.ldb * 2);
    const int brows = SW ? 2 * g.I : g.N;
    const u32x4 rsB = dma_rsrc(g.B, (unsigned)(((long)(brows - 1) * g.ldb + RS_K) * 2));
    int voffB[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int rr = wave + 4 * q;
        const int pos = 16 * ((lane & 48) | ((lane & 15) ^ rr));
        voffB[q] = rr * ldb2 + pos;
    }
    // reader: row r of a block, k-step ks -> chunk 2 ks + h; ks & 7 picks one of 8 per-lane offsets, ks >> 3 adds 256
    int raddr[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) raddr[e] = r * 1024 + ((32 * e + 16 * h) ^ (16 * (r & 15)));
    // 16x16x32 shape: a fragment = 16 rows x 32 reduction indices, lane (i = lane & 15, kq = lane >> 4) takes row i,
    // indices 32 ks + 8 kq .. + 7 = chunk 4 ks + kq; sub-step u (0..31) of a 32-row block = (ks = u >> 1, rows
    // 16 (u & 1) ..). The sixteen lanes a ds_read_b128 serves together still hit sixteen different bank groups.
    const int li = lane & 15, kq = lane >> 4;
    int raddr16[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) raddr16[e] = li * 1024 + 16 * ((4 * e + kq) ^ li);
    // byte offset inside a 32-row image of the fragment of sub-step u
    auto frag_off = [&](int u) {
        return RS_M16 ? (u & 1) * 16384 + raddr16[(u >> 1) & 3] + 256 * (u >> 3) : raddr[u & 7] + 256 * (u >> 3);
    };

    // ---- the weight stream: blocks of one run in rotated order, so that workgroups that work in step do not all
    // store the same 128-byte column segment of their rows at the same time (one memory channel for the whole chip
    // whenever the row pitch is a multiple of the channel interleave)
    const int npp = p1 - p0;
    // (rotation over 4 KiB worth of column segments when the row pitch is a multiple of 4 KiB - the SwiGLU form's
    // 64-byte segments at 4 / 8 KiB pitch measured 1.3 ms per training step slower with a 1 KiB span)
    const int rspan = SW ? 64 : (((g.ldc * 2) & 4095) == 0 ? 32 : 16);
    const int rot = (rp & (rspan - 1)) % npp;
    auto pair_of = [&](int jp) { int q = jp + rot; if (q >= npp) q -= npp; return p0 + q; };
    auto issue_block = [&](int jl) {
        const int jc = min(jl, nblk - 1);
        const int gp = pair_of(jc >> 1);
        const int row = SW ? ((jc & 1) * g.I + 32 * gp) : (64 * gp + 32 * (jc & 1));
        rs_issue(rsB, voffB, lds0 + (jl & 3) * RS_SLOT + wave * 1024, row * ldb2, 16 * ldb2);
    };
    // the same block piece by piece (piece q behind k-step 4 q + 2 of the block that is being multiplied)
    auto block_soff = [&](int jl) {
        const int jc = min(jl, nblk - 1);
        const int gp = pair_of(jc >> 1);
        const int row = SW ? ((jc & 1) * g.I + 32 * gp) : (64 * gp + 32 * (jc & 1));
        return row * ldb2;
    };
    // ---- the token panel: 2 x 128 rows through the ring (whole 1 KiB rows, like the weight blocks) into registers:
    // fragment (mt, ks) of a lane = row 32 mt + r, reduction indices 16 ks + 8 h .. + 7
    const int mrow0 = rp * RS_ROWS + wave * 64;
    bf16x8 Af[64];   // 32x32x16: [mt (2)][ks (32)]; 16x16x32: [mt (4)][ks (16)]
    {
        const int lda2 = (int)(g.lda * 2);
        const u32x4 rsA = dma_rsrc(g.A, (unsigned)(((long)(g.M - 1) * g.lda + RS_K) * 2));
        int voffA[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) voffA[q] = voffB[q] + (wave + 4 * q) * (lda2 - ldb2);
#pragma unroll
        for (int f = 0; f < 2; ++f) {
#pragma unroll
            for (int sl = 0; sl < 4; ++sl)
                rs_issue(rsA, voffA, lds0 + sl * RS_SLOT + wave * 1024, (rp * RS_ROWS + 128 * f + 32 * sl) * lda2,
                         16 * lda2);
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if ((wave >> 1) == f) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const char* sb = smem + (2 * (wave & 1) + mt) * RS_SLOT;
                    // 32x32x16: fragment (mt, ks) = sub-step ks; 16x16x32: fragment (2 mt + half, ks) = sub-step 2 ks + half
#pragma unroll
                    for (int u = 0; u < 32; ++u)
                        Af[RS_M16 ? (2 * mt + (u & 1)) * 16 + (u >> 1) : mt * 32 + u] =
                            *reinterpret_cast<const bf16x8*>(sb + frag_off(u));
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    issue_block(0);
    issue_block(1);
    issue_block(2);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    // The output work rides behind the MFMAs of the following blocks: the two tiles a block has finished are rounded
    // and parked in the wave's LDS image behind the first 8 k-steps of the next block (which accumulates into the
    // other two tiles); once both halves of a pair are parked (8 k-steps into the next pair) the image is read back
    // as whole 128-byte row segments and stored, one slice per two k-steps.
    f32x16 acc[2][2];          // 32x32x16: [token tile (2)][block of the pair]
    rs_f32x4 acc16[4][2][2];   // 16x16x32: [token tile (4)][column tile (2)][block of the pair]
    char* st = smem + RS_NSLOT * RS_SLOT + wave * RS_STAGE;

    // slice t (0..7) of parking the tiles acc[.][j]: tile row block i = t >> 2, column group g4 -> 8 bytes per lane
    auto park = [&](int j, int t) {
        if (RS_M16) {
            // tile (mt = t >> 1, nt = t & 1): lane holds token row 16 mt + li, columns 16 nt + 4 kq .. + 3 of block j
            const int mt = t >> 1, nt = t & 1;
            const int row = 16 * mt + li;
            const int ch = 4 * j + 2 * nt + (kq >> 1);
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16)(acc16[mt][nt][j][e]);
            *reinterpret_cast<bf16x4*>(st + row * 128 + ((ch ^ (row & 7)) << 4) + 8 * (kq & 1)) = o;
            return;
        }
        const int i = t >> 2, g4 = t & 3;
        const int row = 32 * i + r;
        const int ch = 4 * j + g4;
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)(acc[i][j][4 * g4 + e]);
        *reinterpret_cast<bf16x4*>(st + row * 128 + ((ch ^ (row & 7)) << 4) + 8 * h) = o;
    };
    // plain form, slice it (0..7): 8 rows x 128 bytes of the image
    auto unpark = [&](int it) {
        const int row = it * 8 + (lane >> 3), ch = lane & 7;
        return *reinterpret_cast<const u32x4*>(st + row * 128 + ((ch ^ (row & 7)) << 4));
    };
    auto store_plain = [&](int it, int gp, const u32x4& v) {
        const int row = it * 8 + (lane >> 3), ch = lane & 7;
        u32x4* dst = reinterpret_cast<u32x4*>(g.C + (long)(mrow0 + row) * g.ldc + 64 * gp + ch * 8);
        // non-temporal: the output is a stream several times the size of the L2, whose capacity belongs to the weight
        // rows every workgroup re-reads (measured with plain stores: 2.3 TB/s of output, with these 4.4-4.7 TB/s)
        if (DBG & 1) asm volatile("" ::"v"(v));
        else if (DBG & 64) *dst = v;
        else __builtin_nontemporal_store(v, dst);
    };
    // SwiGLU form, slice it (0..3): 16 rows; chunks 0..3 of a row are 32 gate columns, chunks 4..7 the up columns of
    // the same indices. Arithmetic of the separate kernel: silu in fp32 from the bf16 gate, rounded to bf16, times up
    auto store_swiglu = [&](int it, int gp) {
        const int row = it * 16 + (lane >> 2), ch = lane & 3;
        const bf16x8 gv = *reinterpret_cast<const bf16x8*>(st + row * 128 + ((ch ^ (row & 7)) << 4));
        const bf16x8 uv = *reinterpret_cast<const bf16x8*>(st + row * 128 + (((4 + ch) ^ (row & 7)) << 4));
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float gf = (float)gv[e];
            const float sl = round_bf(gf * __builtin_amdgcn_rcpf(1.f + __expf(-gf)));   // as sigmoidf_ in elem.hip
            o[e] = (bf16)(sl * (float)uv[e]);
        }
        bf16* c = g.C + (long)(mrow0 + row) * g.ldc + 32 * gp + ch * 8;
        __builtin_nontemporal_store(gv, reinterpret_cast<bf16x8*>(c));
        __builtin_nontemporal_store(uv, reinterpret_cast<bf16x8*>(c + g.I));
        __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(g.act + (long)(mrow0 + row) * g.ldact + 32 * gp + ch * 8));
    };

    // Fragments of the first RS_AHEAD k-steps of the coming block, requested at the end of the block before it. That
    // is legal because a block is waited for one barrier early: the wave waits for ITS pieces of block jl + 1 in
    // front of the barrier that opens block jl, so behind that barrier block jl + 1 is complete for every wave.
    bf16x8 fpre[RS_AHEAD];
#pragma unroll
    for (int ks = 0; ks < RS_AHEAD; ++ks) fpre[ks] = *reinterpret_cast<const bf16x8*>(smem + frag_off(ks));

    // one pair of blocks; FIRST: there is no earlier block whose tiles wait to be parked and stored
    // WV = this wave's index as a constant: the four waves run four copies of the loop that differ only in WHEN
    // they issue their vector-memory instructions (LDS-DMA pieces behind k-step 4 q + WV, stores one k-step later in
    // the odd waves), so that the four do not queue at the CU's one memory pipeline at the same moments
    auto do_pair = [&](int jp, auto first_tag, auto wave_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int WV = decltype(wave_tag)::value;
        constexpr int WO = WV & 1;
        const int gpp = FIRST ? 0 : pair_of(jp - 1);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int jl = 2 * jp + nb;
            // this wave's pieces of block jl + 1 have landed: younger are the 8 pieces of block jl + 2 and, when the
            // block before this one carried a pair's stores (nb == 1), those stores. Every wave's LDS reads of block
            // jl - 1 have returned (they fed MFMAs that are issued); past the barrier its slot takes block jl + 3.
            if (!(DBG & 32)) {
                if (nb == 0) dma_wait<8>();
                else dma_wait<8 + S>();
            }
            asm volatile("s_barrier" ::: "memory");
            const int nsoff = __builtin_amdgcn_readfirstlane(block_soff(jl + 3));
            const unsigned ndst = __builtin_amdgcn_readfirstlane(lds0 + ((jl + 3) & 3) * RS_SLOT + wave * 1024);
            const char* sb = smem + (jl & 3) * RS_SLOT;
            const char* sbn = smem + ((jl + 1) & 3) * RS_SLOT;
            const bool out = !(FIRST && nb == 0) && !(DBG & 2);   // tiles of block jl - 1 to park
            bf16x8 fb[RS_KS];
            u32x4 ov[2] = {};
#pragma unroll
            for (int ks = 0; ks < RS_AHEAD; ++ks) fb[ks] = fpre[ks];
#pragma unroll
            for (int ks = 0; ks < RS_KS; ++ks) {
                // the fragment of k-step ks + RS_AHEAD is requested in front of the two MFMAs of step ks
                if (ks + RS_AHEAD < RS_KS)
                    fb[ks + RS_AHEAD] = *reinterpret_cast<const bf16x8*>(sb + frag_off(ks + RS_AHEAD));
                else
                    fpre[ks + RS_AHEAD - RS_KS] = *reinterpret_cast<const bf16x8*>(sbn + frag_off(ks + RS_AHEAD - RS_KS));
                if (DBG & 8) {
                    asm volatile("" ::"v"(fb[ks]));
                } else if (RS_M16) {
                    // sub-step ks = (k-step ks >> 1, column tile ks & 1): one weight fragment, the four token tiles
                    const int k16 = ks >> 1, nt = ks & 1;
                    const rs_f32x4 z = {};
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc16[mt][nt][nb] = mfma16(fb[ks], Af[mt * 16 + k16], k16 == 0 ? z : acc16[mt][nt][nb]);
                } else if (ks == 0) {
                    const f32x16 z = {};
                    acc[0][nb] = mfma32(fb[ks], Af[ks], z);
                    acc[1][nb] = mfma32(fb[ks], Af[32 + ks], z);
                } else {
                    acc[0][nb] = mfma32(fb[ks], Af[ks], acc[0][nb]);
                    acc[1][nb] = mfma32(fb[ks], Af[32 + ks], acc[1][nb]);
                }
                // piece q of block jl + 3 behind k-step 4 q + WV (the last one, 28 + WV, behind the block's last store)
                if (!(DBG & 4) && (ks & 3) == WV) {
                    const int q = ks >> 2;
                    dma_piece(rsB, voffB[q & 3], ndst + 0x1000 * q, nsoff + (q >> 2) * 16 * ldb2);
                }
                if (out) {
                    if (ks < 8) park(nb ^ 1, ks);
                    if (nb == 0 && !SW) {
                        // image slice it is requested behind step 8 + 2 it and stored behind step 11 + 2 it
                        const int kk = ks - WO;   // odd waves: everything one k-step later
                        if (kk >= 8 && kk < 24 && !(kk & 1)) ov[(kk >> 1) & 1] = unpark((kk - 8) >> 1);
                        if (kk >= 11 && kk < 27 && (kk & 1)) store_plain((kk - 11) >> 1, gpp, ov[((kk - 3) >> 1) & 1]);
                    }
                    if (nb == 0 && SW) {
                        const int kk = ks - 2 * WO;   // odd waves: two k-steps later
                        if (kk >= 8 && kk < 24 && (kk & 3) == 3) store_swiglu((kk - 8) >> 2, gpp);
                    }
                }
            }
        }
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;

    auto run = [&](auto wave_tag) {
        do_pair(0, yes{}, wave_tag);
        for (int jp = 1; jp < npp; ++jp) do_pair(jp, no{}, wave_tag);
    };
    switch (wave) {
        case 0: run(std::integral_constant<int, 0>{}); break;
        case 1: run(std::integral_constant<int, 1>{}); break;
        case 2: run(std::integral_constant<int, 2>{}); break;
        default: run(std::integral_constant<int, 3>{}); break;
    }
    // the last block's tiles and the last pair's stores
    if (!(DBG & 2)) {
        const int gpl = pair_of(npp - 1);
#pragma unroll
        for (int t = 0; t < 8; ++t) park(1, t);
        if (!SW) {
#pragma unroll
            for (int it = 0; it < 8; ++it) store_plain(it, gpl, unpark(it));
        } else {
#pragma unroll
            for (int it = 0; it < 4; ++it) store_swiglu(it, gpl);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may be pending when the wave ends
}

template <bool SW, int DBG = 0>
int rs_launch(const RsArgs& g, hipStream_t s) {
    static bool asked[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (!asked[dev]) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_rs_kernel<SW, DBG>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS);
        if (e != hipSuccess) {
            mgen_set_error("mgen_gemm_bf16: cannot reserve %d bytes of LDS: %s", RS_LDS, hipGetErrorString(e));
            return MGEN_ELAUNCH;
        }
        asked[dev] = true;
    }
    hipLaunchKernelGGL((gemm_rs_kernel<SW, DBG>), dim3(g.nrp * g.parts), dim3(256), RS_LDS, s, g);
    return mgen_check_launch("mgen_gemm_bf16");
}

}  // namespace

bool mgen_gemm_rs_fits(int M, int N, int K, long lda, long ldb, long ldc) {
    const char* e = getenv("MGEN_GEMM_RS");   // MGEN_GEMM_RS=0: never (same-process A/B runs)
    const bool off = e && e[0] == '0';
    // whole 256-row panels only: the counted waits of the kernel assume that every store instruction is issued
    if (off || K != RS_K || N % 64 != 0 || N < 1024 || M < 64 * RS_ROWS || M % RS_ROWS != 0) return false;
    if (lda < RS_K || ldb < RS_K || (lda % 8) || (ldb % 8) || (ldc % 8)) return false;
    // byte offsets inside the buffer descriptors are 32-bit
    if ((long)M * lda * 2 >= (1L << 32) || (long)N * ldb * 2 >= (1L << 32)) return false;
    return true;
}

// I > 0: SwiGLU form (B has 2 I rows, C = [gate | up] with 2 I columns, act [M][I]); else plain (B has N rows)
int mgen_gemm_rs_launch(const void* A, const void* B, void* C, void* act, int M, int N, int I, long lda, long ldb,
                        long ldc, long ldact, hipStream_t stream) {
    RsArgs g;
    g.A = (const bf16*)A; g.B = (const bf16*)B; g.C = (bf16*)C; g.act = (bf16*)act;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldact = ldact;
    g.M = M; g.N = N; g.I = I;
    g.nrp = (M + RS_ROWS - 1) / RS_ROWS;
    g.npairs = I > 0 ? I / 32 : N / 64;
    // fill the chip: cut the column range while there are fewer workgroups than CUs (runs of at least 8 pairs)
    int parts = 1;
    while (parts < 8 && g.nrp * parts < 256 && g.npairs / (parts * 2) >= 8) parts *= 2;
    g.parts = parts;
    g.ppp = (g.npairs + parts - 1) / parts;
    g.xcd_map = ((g.nrp * parts) % 8) == 0;
#ifdef MGEN_RS_EXPERIMENTS
    if (const char* e = getenv("MGEN_RS_DBG")) {
        switch (atoi(e)) {
            case 1: return rs_launch<false, 1>(g, stream);
            case 2: return rs_launch<false, 2>(g, stream);
            case 4: return rs_launch<false, 4>(g, stream);
            case 6: return rs_launch<false, 6>(g, stream);
            case 8: return rs_launch<false, 8>(g, stream);
            case 10: return rs_launch<false, 10>(g, stream);
            case 14: return rs_launch<false, 14>(g, stream);
            case 64: return rs_launch<false, 64>(g, stream);
            case 5: return rs_launch<false, 5>(g, stream);
            case 32: return rs_launch<false, 32>(g, stream);
            case 33: return rs_launch<false, 33>(g, stream);
            case 9: return rs_launch<false, 9>(g, stream);
            case 12: return rs_launch<false, 12>(g, stream);

        }
    }
#endif
    return I > 0 ? rs_launch<true>(g, stream) : rs_launch<false>(g, stream);
}
