// Causal self-attention on bf16 MFMA for gfx950, head_dim 128 or 64: forward (online softmax, saves LSE), and
// backward as two kernels (dK/dV per key block, dQ per query block) that recompute P from Q, K and LSE.
//
// Reference: the attention call in Qwen3Attention.forward — transformers integrations/sdpa_attention.py:154-163
// (is_causal, scale = head_dim^-0.5) and its eager twin models/qwen3/modeling_qwen3.py:185-207 (softmax in fp32) —
// plus its autograd backward. Operands are the bf16-rounded q, k (after q/k-norm + RoPE) and v.
//
// Layout: q/k/v/o/dq/dk/dv are token-major [B*S, heads, head_dim] views with an explicit token stride (so q, k, v
// may live inside one fused projection output). LSE and delta are [B, H, S] fp32.
//
// Structure shared by the three kernels: 256 threads = 4 waves; each wave owns 32 rows (queries in fwd/dQ, keys in
// dK/dV) and keeps its own operand rows in registers as MFMA B-fragments; the other side is streamed in 64-row
// tiles of 256 bytes per row through two LDS buffers by LDS-DMA (next tile issued before the current tile's MFMAs,
// one barrier per tile). With head_dim 128 a tile row is one head and a block covers 128 rows of one head; with
// head_dim 64 a tile row is a PAIR of adjacent heads, waves 0-1 work on the first head of the pair and waves 2-3 on
// the second, and a block covers 64 rows of both. Scores are computed transposed where that puts the softmax row
// on the lane (S^T = K Q^T in fwd/dQ), so row max / row sum need one cross-half exchange only, and the fp32 score
// tile is re-used in place as the bf16 operand of the next product (no LDS round trip for P or dS).
//
// This file is compiled twice: attn.hip (MGEN_ATTN_DROP 0: mgen_attn_fwd / mgen_attn_bwd) and attn_drop.hip
// (MGEN_ATTN_DROP 1: the same kernels with dropout on the probabilities, mgen_attn_{fwd,bwd}_dropout). Instances of
// one kernel template that are compiled together perturb each other's register allocation; separate translation
// units keep the plain kernels' code what it is without the dropout variants.
#include "common.h"
#include <type_traits>

#ifndef MGEN_ATTN_DROP
#error "include through attn.hip or attn_drop.hip"
#endif

namespace {

constexpr int ROWB = 256;            // bytes per LDS tile row (one head of 128, or two heads of 64)
constexpr int TILE = 64 * ROWB;      // bytes of one 64-row tile
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float NEG_BIG = -1.0e30f;

struct AttnArgs {
    const bf16* Q; const bf16* K; const bf16* V; const bf16* O; const bf16* dO;
    bf16* Out; bf16* dQ; bf16* dK; bf16* dV;
    float* LSE; const float* Delta;
    long sq, sk, sv, so, sdo, sdq, sdk, sdv;  // token strides in elements
    int B, S, H, KVH;
    float scale;
    int gx, gy, gz;  // logical grid: row blocks, head (pair) index, batch; launched as one dimension
    int nqb;         // forward: query blocks per sequence (a workgroup takes two of them, gx = ceil(nqb / 2))
    // dropout on the attention probabilities (kernels instantiated with DROP): element (b*H + h, q, key) is kept iff
    // drop_fmix32(drop_rowkey(drop_key, b*H + h, q) ^ key) >= drop_thresh; kept probabilities count 1 / (1 - p)
    unsigned long long drop_key;
    unsigned drop_thresh;
    float drop_inv_keep;
};

// Logical workgroup coordinates from the 1-D launch index. Hardware deals consecutive workgroups round-robin to the
// 8 XCDs; the remap gives each XCD a contiguous range of logical ids, so all row blocks of one (batch, head) run on
// the same XCD at about the same time and share its L2 for the K/V (or Q/dO) tiles they all stream.
struct Coords { int x, y, z; };
__device__ __forceinline__ Coords block_coords(const AttnArgs& a) {
    const int L = xcd_remap(blockIdx.x, a.gx * a.gy * a.gz);
    Coords c;
    c.x = L % a.gx;
    c.y = (L / a.gx) % a.gy;
    c.z = L / (a.gx * a.gy);
    return c;
}

template <int HD> struct Geo {
    static constexpr int KS = HD / 16;                  // MFMA k-steps over the head dimension
    static constexpr int DB = HD / 32;                  // 32-wide output blocks over the head dimension
    static constexpr int ROWS = HD == 128 ? 128 : 64;   // rows (queries or keys) per workgroup
    static constexpr int HPB = HD == 128 ? 1 : 2;       // heads per workgroup
};

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// 64 rows x 256 bytes straight into the LDS image by LDS-DMA (no staging registers, no ds_write): a wave
// instruction fills one contiguous 1 KiB piece (4 rows), so the image's chunk swizzle goes on the source address.
// Rows are clamped to S-1 (their results are masked or never stored).
__device__ __forceinline__ void tile_glds(const bf16* __restrict__ base, long tok_stride, long tok0, int row0, int S,
                                          char* lds, int tid) {
    const int wave = tid >> 6, l = tid & 63;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int piece = wave * 4 + j;
        const int row = piece * 4 + (l >> 4);
        const int ch = (l & 15) ^ swz256(row);
        const int gr = min(row0 + row, S - 1);
        const bf16* src = base + (tok0 + gr) * tok_stride + ch * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + piece * 1024), 16, 0, 0);
    }
}

// rows of this wave as B-operand fragments: lane (r, hh) holds row (row0 + r), columns 16ks + 8hh .. +7
template <int KS>
__device__ __forceinline__ void load_own_rows(bf16x8 (&f)[KS], const bf16* __restrict__ base, long tok_stride,
                                              long tok0, int row0, int S, int lane) {
    const int gr = min(row0 + (lane & 31), S - 1);
    const bf16* p = base + (tok0 + gr) * tok_stride + 8 * (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) f[ks] = *reinterpret_cast<const bf16x8*>(p + 16 * ks);
}

// The register-resident operand rows are loaded once, before the main loop. Drain those loads here and pass the
// registers through an empty asm so that the compiler's wait-count bookkeeping does not tie the loop's MFMAs to
// vector-memory operations issued inside the loop.
template <int KS>
__device__ __forceinline__ void settle_frags(bf16x8 (&f)[KS]) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < KS; ++i) asm volatile("" : "+v"(f[i]));
}

__device__ __forceinline__ void zero16(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// An accumulation chain of KS MFMAs whose result lands in VGPRs. In a kernel that may use more than 256 registers the
// compiler puts every MFMA result into the accumulator half of the register file, so a result that vector ALU code
// consumes (scores, probabilities) costs one copy per register on the way out; written as asm the chain's destination
// is an ordinary vector register. Nothing inside an asm string is padded by the compiler: the leading s_nop covers a
// vector-ALU write of an operand just before the first MFMA, the trailing s_nop 11 the 12 wait states between the
// last MFMA's write and any reader of the result.
template <int KS>
__device__ __forceinline__ void mfma_chain_vgpr(f32x16& d, const bf16x8 (&a)[KS], const bf16x8 (&b)[KS]) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a[0]), "v"(b[0]));
#pragma unroll
    for (int ks = 1; ks < KS - 1; ++ks)
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a[ks]), "v"(b[ks]));
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 11"
                 : "+v"(d) : "v"(a[KS - 1]), "v"(b[KS - 1]));
}

// Store a transposed accumulator set acc[db] = X^T[d = 32db + acc_row(reg)][row = lane&31] as X[row][d] (bf16)
template <int DB>
__device__ __forceinline__ void store_rows_T(const f32x16 (&acc)[DB], bf16* __restrict__ dst, float mul, int lane) {
    const int hh = lane >> 5;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16)(acc[db][4 * g4 + e] * mul);
            *reinterpret_cast<bf16x4*>(dst + 32 * db + 8 * g4 + 4 * hh) = o;
        }
}

// ===================================================================================================
// Tile streaming shared by the three kernels.
//
// A streamed operand (K, V in forward / dQ; Q, dO in dK/dV) is read through a buffer descriptor that covers the
// rows of ONE batch element from the first head of the tile row on: row r, 16-byte chunk c sits at byte offset
// r * stride_bytes + 16 c. Every lane keeps the offsets of its pieces of a 64-row tile in registers (computed once),
// the tile's first row enters as the scalar offset of the load, and rows at or past the sequence end are outside
// the descriptor's range and arrive as zeros: issuing a tile costs no vector arithmetic.
// ===================================================================================================
template <int NW>   // waves per workgroup sharing the tile: each issues 16 / NW one-KiB pieces (4 rows each)
struct TileStream {
    static constexpr int NP = 16 / NW;
    u32x4 rsrc;      // buffer descriptor (scalar registers)
    int voff;        // this lane's offset in the wave's first piece; the wave's pieces lie 4 NW rows apart, a multiple
                     // of 16 rows for NW >= 4, which leaves the chunk swizzle unchanged: they differ by a scalar
    int row_bytes;   // token stride in bytes

    // base, tok_stride and S must be the same in every lane (kernel arguments and workgroup / wave indices)
    // span_bytes = bytes of a token row, from base on, that the stream may touch (one tile row, or several heads)
    __device__ __forceinline__ void init(const bf16* base, long tok_stride, int S, int wave, int lane,
                                         int span_bytes = ROWB) {
        static_assert(NW == 4 || NW == 8, "pieces of one wave must be a multiple of 16 rows apart");
        row_bytes = (int)(tok_stride * 2);
        const unsigned long long p = (unsigned long long)base;
        rsrc[0] = __builtin_amdgcn_readfirstlane((unsigned)p);
        rsrc[1] = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32) & 0xFFFFu);   // stride 0: raw buffer
        rsrc[2] = __builtin_amdgcn_readfirstlane((unsigned)((S - 1) * row_bytes + span_bytes));
        rsrc[3] = 0x00020000u;
        const int row = wave * 4 + (lane >> 4);
        voff = row * row_bytes + 16 * ((lane & 15) ^ swz256(row));
    }
    // Piece j (0 .. NP-1) of this wave of rows [row0, row0 + 64) into the 16 KiB image at LDS byte address lds_base
    // (scalar); wave = this wave's (scalar) index. An asm statement: the compiler does not know that an LDS write is
    // pending (it would drain it with s_waitcnt vmcnt(0) in front of the next transposed LDS read, in the middle of
    // the tile the prefetch is meant to overlap). The caller waits (asm s_waitcnt vmcnt) and passes a barrier before
    // anyone reads the image. M0 (the instruction's LDS base) is saved and restored.
    // extra = a byte offset inside the token row (another head of the row the descriptor starts at)
    __device__ __forceinline__ void issue_piece(int j, int row0, unsigned lds_base, int wave, int extra = 0) const {
        unsigned keep;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(wave + NW * j) * 1024u);
        const int soff = __builtin_amdgcn_readfirstlane((row0 + 4 * NW * j) * row_bytes + extra);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    }
    __device__ __forceinline__ void issue(int row0, unsigned lds_base, int wave, int extra = 0) const {
#pragma unroll
        for (int j = 0; j < NP; ++j) issue_piece(j, row0, lds_base, wave, extra);
    }
    // the same piece with M0 declared clobbered instead of saved and restored (kernels that never rely on M0)
    __device__ __forceinline__ void issue_piece_m0(int j, int row0, unsigned lds_base, int wave, int extra = 0) const {
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(wave + NW * j) * 1024u);
        const int soff = __builtin_amdgcn_readfirstlane((row0 + 4 * NW * j) * row_bytes + extra);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     : : "s"(dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
    }
};

// per-lane byte offsets inside an IMG256 tile of the two kinds of operand read; rows and columns that are
// multiples of 16 rows / one whole tile enter as immediates of the DS instruction (+4096 per 16 rows)
// row read, k-step ks (absolute, 0..7), lane's row r = lane & 31 (+32 rows = +8192)
__device__ __forceinline__ int row_frag_off(int ks, int lane) {
    const int r = lane & 31;
    return 256 * r + 16 * ((2 * ks + (lane >> 5)) ^ swz256(r));
}
// transposed read in accumulator order (frag_col_img256<true>), 32-column block cb (absolute, 0..3), second = the
// read that supplies elements 4..7
__device__ __forceinline__ int col_frag_off(int cb, bool second, int lane) {
    const int g = lane >> 4, i = lane & 15, h = g >> 1, q = i >> 2, p = i & 3;
    const int row = 4 * h + q + (second ? 8 : 0);
    const int ch = 4 * cb + 2 * (g & 1) + (p >> 1);
    return 256 * row + 16 * (ch ^ swz256(row)) + 8 * (p & 1);
}
__device__ __forceinline__ bf16x8 col_frag(const char* smem, int off_a, int off_b) {
    s16x4 lo = lds_read_tr16(smem, off_a);
    s16x4 hi = lds_read_tr16(smem, off_b);
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, both);
}
// the other half-wave's value of a per-row quantity (lanes l and l ^ 32 hold the two halves of a row)
__device__ __forceinline__ float other_half(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const unsigned lo = r[0], hi = r[1];   // (named: a bit cast applied to a vector element reads element 0)
    const unsigned other = (threadIdx.x & 32) ? lo : hi;
    return __builtin_bit_cast(float, other);
}

// ===================================================================================================
// forward
//
// One wave per SIMD with the whole register file: a workgroup is 4 waves, a wave owns 64 query rows (two 32-row
// blocks A and B that share every K and V fragment it reads from LDS), so a workgroup covers 256 rows of one head
// (head_dim 128) or 128 rows of a pair of heads (head_dim 64). The output accumulators (O^T, 128 registers) and the
// query fragments live in the accumulator half of the register file; scores, probabilities and addresses in the
// vector half.
//
// The tile loop is software pipelined so that the matrix pipe always has work that does not depend on the vector
// ALU's current work:
//     phase X(j):  S(j+1) = K(j+1) Q^T   (MFMA)   beside   p = exp2(S(j) - m)  (first part, VALU)
//     phase Y(j):  O     += V(j) P(j)    (MFMA)   beside   rest of the exponentials, then rowmax(S(j+1))
// Each MFMA is followed by its share of the vector work and the LDS reads of a later MFMA, and a scheduling fence
// keeps that placement. The MFMAs are asm statements so that their register classes are fixed: scores land in
// vector registers (no copies out of the accumulator file), O stays in accumulator registers for the whole kernel.
// Nothing inside an asm string is padded by the compiler: operands written by the vector ALU are produced at least
// one MFMA slot before the MFMA that reads them, MFMA results are read by vector code a phase later or behind an
// explicit s_nop.
//
// K and V tiles travel through two-slot rings in LDS with one workgroup barrier per tile: after the barrier of
// iteration j, K(j+1) and V(j) are complete and K(j+2), V(j+1) are issued into the slots whose last readers
// (phase X(j-1), phase Y(j-1)) are behind that barrier.
//
// The running maximum of a row is raised only when a tile's maximum exceeds it by more than 2^RESCALE_LOG2 in the
// exponent's units: until then the probabilities are taken against the older reference (they stay below
// 2^RESCALE_LOG2: exact in fp32, bf16 keeps its relative precision), and the accumulators are not rescaled.
// ===================================================================================================
constexpr float RESCALE_LOG2 = 6.0f;

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

// Accumulator registers are addressed by NUMBER in these statements, so a tile of O or a query fragment stays where
// it is for the whole kernel and no statement carries a 16-register tied operand through the loop. The compiler does
// not know about it: the kernels keep to a64 and above, the source is compiled with vector-register spills to
// accumulator registers switched off, and the build audits the assembly (no compiler-generated v_accvgpr_* at or
// above a64, no scratch).
template <int ACC>   // a[ACC : ACC+15] += A(v) * B(v)
__device__ __forceinline__ void mfma_acc(const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" : : "v"(a), "v"(b), "i"(ACC), "i"(ACC + 15));
}
template <int QA>    // d(v) += A(v) * B(a[QA : QA+3])
__device__ __forceinline__ void mfma_vq(f32x16& d, const bf16x8& a) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(d) : "v"(a), "i"(QA), "i"(QA + 3));
}
template <int QA>    // d(v) = A(v) * B(a[QA : QA+3])
__device__ __forceinline__ void mfma_vq0(f32x16& d, const bf16x8& a) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], 0" : "=&v"(d) : "v"(a), "i"(QA), "i"(QA + 3));
}
template <int A> __device__ __forceinline__ void acc_write(float x) {
    asm volatile("v_accvgpr_write_b32 a%c1, %0" : : "v"(x), "i"(A));
}
template <int A> __device__ __forceinline__ float acc_read() {
    float x; asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(x) : "i"(A)); return x;
}
template <int A> __device__ __forceinline__ void acc_scale(float f) {
    float t; asm volatile("v_accvgpr_read_b32 %0, a%c2\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a%c2, %0"
                          : "=&v"(t) : "v"(f), "i"(A));
}
// Vector ALU steps of the pipelined loops as asm statements: their place between the MFMAs is part of the design,
// and asm statements are neither reordered nor merged into packed forms. A transcendental's result is read no
// earlier than two statements later (one wait state is required).
__device__ __forceinline__ float a_fma(float x, float y, float z) {
    float d; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z)); return d;
}
__device__ __forceinline__ float a_exp2(float x) { float d; asm volatile("v_exp_f32 %0, %1" : "=v"(d) : "v"(x)); return d; }
__device__ __forceinline__ float a_add(float x, float y) {
    float d; asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d;
}
__device__ __forceinline__ float a_max3(float x, float y, float z) {
    float d; asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z)); return d;
}
__device__ __forceinline__ unsigned a_cvt_pk_bf16(float lo, float hi) {
    unsigned d; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(lo), "v"(hi)); return d;
}

template <int HD, bool DROP>
__global__ __launch_bounds__(256, 1) void attn_fwd_kernel(AttnArgs a) {
    using G = Geo<HD>;
    constexpr int KS = G::KS, DB = G::DB;
    constexpr int QROWS = HD == 128 ? 256 : 128;     // query rows of one head per workgroup
    constexpr int GX = 4 * KS, GY = 8 * DB;          // MFMAs of phase X and of phase Y
    constexpr int KRING = 0, VRING = 2 * TILE;       // LDS: K slots at 0 / 16 KiB, V slots at 32 / 48 KiB
    constexpr int QREG = 4 * TILE;                   // the workgroup's query rows: QROWS / 64 tiles
    constexpr int STG = QREG + (QROWS / 64) * TILE;  // output staging: 8 KiB per wave
    // accumulator registers, addressed by number: O[blk][db] tiles, then Q[blk][ks]. The range starts at a64: when
    // the compiler runs out of vector registers it may place values of its own in accumulator registers, lowest
    // numbers first (the build's audit, tools/audit_attn_asm.py, fails if it ever reaches a64)
    constexpr int OACC = 64, QACC = 192;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    asm volatile("" ::: "a255");                     // makes the kernel allocate all 256 accumulator registers
#ifdef MGEN_ATTN_STAMPS   // diagnostic build: cycle stamps of the kernel's sections, written over part of LSE
    unsigned long long st0 = __builtin_amdgcn_s_memtime();
    unsigned long long st1 = 0, st2 = 0, st3 = 0;
#define STAMP(x) x = __builtin_amdgcn_s_memtime()
#else
#define STAMP(x)
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Coords bid = block_coords(a);
    // A workgroup handles the query blocks nqb - 1 - x and x one after the other (the heavy one first): every
    // workgroup then streams the same number of tiles, so the chip's workgroup slots (one per CU: the kernel uses the
    // whole register file) free up together and the in-order dispatcher never waits on one busy XCD.
    const int qb_hi = a.nqb - 1 - bid.x, qb_lo = bid.x;
    const int npass = qb_lo < qb_hi ? 2 : 1;
    const int b = bid.z;
    const int sub = HD == 128 ? 0 : (wave >> 1);             // which head of the pair (head_dim 64)
    const int wq = HD == 128 ? wave : (wave & 1);            // which 64-query slice of the block
    const int h = HD == 128 ? bid.y : 2 * bid.y + sub;
    const int kv_first = HD == 128 ? h / (a.H / a.KVH) : 2 * bid.y;  // first kv head of the tile row
    const int ks0 = 4 * sub, db0 = 2 * sub;                  // this head's k-steps / d-blocks inside the tile row
    const int S = a.S;
    const long tok0 = (long)b * S;
    int q0, qw0, J, jl, my_q;   // of the current pass
    auto set_block = [&](int qb) {
        q0 = qb * QROWS;
        qw0 = q0 + wq * 64;
        J = min(q0 + QROWS - 1, S - 1) / 64 + 1;      // tiles the workgroup streams
        jl = qw0 < S ? qw0 / 64 : -1;                 // this wave's last tile, the one on its diagonal
        my_q = qw0 + (lane & 31);                     // block A's row of this lane; block B's is my_q + 32
    };

    TileStream<4> Ks, Vs, Qs;
    Ks.init(a.K + tok0 * a.sk + (long)kv_first * HD, a.sk, S, wave, lane);
    Vs.init(a.V + tok0 * a.sv + (long)kv_first * HD, a.sv, S, wave, lane);
    Qs.init(a.Q + tok0 * a.sq + (long)(HD == 128 ? h : 2 * bid.y) * HD, a.sq, S, wave, lane);
    const unsigned lds0 = lds_addr(smem);
    auto issue_first = [&]() {   // what a pass needs before its tile loop: its query rows, K(0), K(1), V(0)
        Ks.issue(0, lds0 + KRING, wave);
#pragma unroll
        for (int t = 0; t < QROWS / 64; ++t) Qs.issue(q0 + 64 * t, lds0 + QREG + t * TILE, wave);
        if (J > 1) Ks.issue(64, lds0 + KRING + TILE, wave);
        Vs.issue(0, lds0 + VRING, wave);
    };
    set_block(qb_hi);
    issue_first();

    int koff[KS], voffa[DB], voffb[DB];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) koff[ks] = row_frag_off(ks0 + ks, lane);
#pragma unroll
    for (int db = 0; db < DB; ++db) {
        voffa[db] = col_frag_off(db0 + db, false, lane);
        voffb[db] = col_frag_off(db0 + db, true, lane);
    }

    float m[2], l[2], nm[2], mx[2];
    const float sl2 = a.scale * LOG2E;
    const int hh = lane >> 5;
    unsigned rk[2] = {0u, 0u};   // dropout: element hash input is (rowkey ^ lane part) ^ c_r ^ tile part

    // ---- pieces of the pipeline -------------------------------------------------------------------
    // row maximum of a score tile pair, one v_max3 per step n in [0, 32); on the diagonal tile the masked scores
    // are overwritten first (key = 64 jl + 32 t + 4 hh + c_r against query = 64 jl + 32 blk + lane row)
    auto max_step = [&](f32x16 (&sn)[2][2], auto nc, auto maskedc) {
        constexpr int n = decltype(nc)::value;
        constexpr bool MASKED = decltype(maskedc)::value;
        constexpr int blk = n >> 4, t = (n >> 3) & 1, r = 2 * (n & 7);
        if constexpr (MASKED) {
            const int thr = 32 * blk + (lane & 31) - 32 * t - 4 * hh;   // masked when c_r > thr
            // the empty asm pins these reads of MFMA results behind the asm statements before them: compiler-placed
            // vector code could otherwise read the registers inside the MFMA's write latency
            float v0 = sn[blk][t][r], v1 = sn[blk][t][r + 1];
            asm volatile("" : "+v"(v0), "+v"(v1));
            if ((r & 3) + 8 * (r >> 2) > thr) v0 = NEG_BIG;
            if (((r + 1) & 3) + 8 * ((r + 1) >> 2) > thr) v1 = NEG_BIG;
            sn[blk][t][r] = v0;
            sn[blk][t][r + 1] = v1;
            mx[blk] = a_max3(mx[blk], v0, v1);
        } else {
            mx[blk] = a_max3(mx[blk], sn[blk][t][r], sn[blk][t][r + 1]);
        }
    };
    // Element e (0..63) of the exponential pipeline: probabilities in the order phase Y consumes them, fragment
    // f = e / 8 = 2 (2 t + s2) + blk, register 8 s2 + e % 8 of S[blk][t]. Step n in [0, 67) runs the multiply-add of
    // element n, the exponential of element n - 1, the row-sum add of element n - 2 and the bf16 packing of a
    // finished pair.
    unsigned pw[32];          // packed probabilities: fragment f = words 4 f .. 4 f + 3
    float xs[64], ps[64];     // per-element temporaries (registers: every index is a compile-time constant)
    float psum[2];
    auto exp_step = [&](const f32x16 (&sc)[2][2], auto nc, int kv0) {
        constexpr int n = decltype(nc)::value;
        if constexpr (n < 64) xs[n] = a_fma(sc[(n >> 3) & 1][n >> 5][8 * ((n >> 4) & 1) + (n & 7)], sl2, nm[(n >> 3) & 1]);
        if constexpr (n >= 1 && n < 65) ps[n - 1] = a_exp2(xs[n - 1]);
        if constexpr (n >= 2 && n < 66) {
            constexpr int e = n - 2, f = e >> 3;
            psum[f & 1] = a_add(psum[f & 1], ps[e]);    // the normaliser counts every visible key, dropped or not
            if constexpr (DROP) {
                constexpr int k = f >> 1, r = 8 * (k & 1) + (e & 7);
                const unsigned hsh = drop_fmix32(rk[f & 1] ^ (unsigned)(kv0 + 32 * (k >> 1)) ^ (unsigned)((r & 3) + 8 * (r >> 2)));
                if (hsh < a.drop_thresh) ps[e] = 0.f;
            }
        }
        if constexpr (n >= 3 && (n & 1)) pw[(n - 3) >> 1] = a_cvt_pk_bf16(ps[n - 3], ps[n - 2]);   // a finished pair
    };
    auto p_frag = [&](auto fc) {
        constexpr int f = decltype(fc)::value;
        const u32x4 w = {pw[4 * f], pw[4 * f + 1], pw[4 * f + 2], pw[4 * f + 3]};
        return __builtin_bit_cast(bf16x8, w);
    };
    constexpr int NSTEP = 67;
    constexpr int NX = 43;                   // exponential steps run beside phase X (fragments 0..4 complete)
    constexpr int GE = HD == 128 ? 22 : 11;  // MFMA slots of phase Y that carry the remaining steps

    // the rare path: raise the reference maxima, rescale the row sums and O
    auto rescale = [&]() {
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // O may just have been written by MFMAs
        float alpha[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const float m_new = fmaxf(m[blk], mx[blk]);
            alpha[blk] = fast_exp2((m[blk] - m_new) * sl2);
            m[blk] = m_new;
            l[blk] *= alpha[blk];
        }
        static_for<2 * DB * 16>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            acc_scale<OACC + i>(alpha[i / (16 * DB)]);
        });
    };
    auto finish_max = [&]() {   // both halves of a row agree on its maximum; decide about the rescale
        mx[0] = fmaxf(mx[0], other_half(mx[0]));
        mx[1] = fmaxf(mx[1], other_half(mx[1]));
        if (__any(fmaxf(mx[0] - m[0], mx[1] - m[1]) * sl2 > RESCALE_LOG2)) rescale();
        nm[0] = -m[0] * sl2;
        nm[1] = -m[1] * sl2;
    };

    // KIND 0: tiles j and j + 1 below the diagonal; 1: tile j + 1 is the diagonal tile; 2: tile j is the wave's last
    auto iter = [&](f32x16 (&sc)[2][2], f32x16 (&sn)[2][2], auto parc, auto kindc, int j) {
        constexpr int PAR = decltype(parc)::value, KIND = decltype(kindc)::value;
        constexpr int KB = KRING + (PAR ^ 1) * TILE;    // K(j+1)
        constexpr int VB = VRING + PAR * TILE;          // V(j)
        const int kv0 = 64 * j;
        psum[0] = psum[1] = 0.f;
        if constexpr (KIND != 2) {
            // ---- phase X: 2 t x KS k-steps x 2 blocks; fragment u = t KS + ks serves both blocks
            bf16x8 kf[2 * KS];
            constexpr int AHEAD = 3;   // fragments read ahead of their MFMAs
            static_for<AHEAD>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                kf[u] = lds_read_b128(smem, koff[u % KS] + KB + 8192 * (u / KS));
            });
            SCHED_FENCE();
            static_for<GX>([&](auto gc) {
                constexpr int g = decltype(gc)::value, u = g >> 1, blk = g & 1, t = u / KS, ks = u % KS;
                if constexpr (ks == 0) mfma_vq0<QACC + (blk * KS + ks) * 4>(sn[blk][t], kf[u]);
                else mfma_vq<QACC + (blk * KS + ks) * 4>(sn[blk][t], kf[u]);
                if constexpr (blk == 1 && u + AHEAD < 2 * KS)
                    kf[u + AHEAD] = lds_read_b128(smem, koff[(u + AHEAD) % KS] + KB + 8192 * ((u + AHEAD) / KS));
                static_for<(g + 1) * NX / GX - g * NX / GX>([&](auto ic) {
                    exp_step(sc, std::integral_constant<int, g * NX / GX + decltype(ic)::value>{}, kv0);
                });
                SCHED_FENCE();
            });
        } else {
            static_for<NX>([&](auto nc) { exp_step(sc, nc, kv0); });
            SCHED_FENCE();
        }
        // ---- phase Y: 4 (t, s2) x DB d-blocks x 2 blocks; fragment u = (2 t + s2) DB + db serves both blocks
        {
            if constexpr (KIND != 2) mx[0] = mx[1] = NEG_BIG;
            bf16x8 vf[4 * DB];
            constexpr int AHEAD = 2;
            auto vread = [&](auto uc) {
                constexpr int u = decltype(uc)::value, k = u / DB, db = u % DB;
                constexpr int im = VB + 8192 * (k >> 1) + 4096 * (k & 1);
                vf[u] = col_frag(smem, voffa[db] + im, voffb[db] + im);
            };
            static_for<AHEAD>([&](auto uc) { vread(uc); });
            SCHED_FENCE();
            static_for<GY>([&](auto gc) {
                constexpr int g = decltype(gc)::value, u = g >> 1, blk = g & 1, k = u / DB, db = u % DB;
                mfma_acc<OACC + (blk * DB + db) * 16>(vf[u], p_frag(std::integral_constant<int, 2 * k + blk>{}));
                if constexpr (blk == 1 && u + AHEAD < 4 * DB) vread(std::integral_constant<int, u + AHEAD>{});
                if constexpr (g < GE) {
                    constexpr int lo = NX + g * (NSTEP - NX) / GE, hi = NX + (g + 1) * (NSTEP - NX) / GE;
                    static_for<hi - lo>([&](auto ic) {
                        exp_step(sc, std::integral_constant<int, lo + decltype(ic)::value>{}, kv0);
                    });
                }
                if constexpr (KIND != 2) {
                    constexpr int lo = g * 32 / GY, hi = (g + 1) * 32 / GY;
                    static_for<hi - lo>([&](auto ic) {
                        max_step(sn, std::integral_constant<int, lo + decltype(ic)::value>{},
                                 std::integral_constant<bool, KIND == 1>{});
                    });
                }
                SCHED_FENCE();
            });
        }
        l[0] += psum[0];
        l[1] += psum[1];
        if constexpr (KIND != 2) finish_max();
    };

    f32x16 sa[2][2], sb[2][2];
    for (int pass = 0; pass < npass; ++pass) {
    // ---- prologue: O = 0, the query rows, scores and maxima of tile 0
    static_for<2 * DB * 16>([&](auto ic) { acc_write<OACC + decltype(ic)::value>(0.f); });
    m[0] = m[1] = NEG_BIG;
    l[0] = l[1] = 0.f;
    nm[0] = nm[1] = 0.f;
    if constexpr (DROP) {
        rk[0] = drop_rowkey(a.drop_key, (unsigned)(b * a.H + h), (unsigned)my_q) ^ (unsigned)(4 * hh);
        rk[1] = drop_rowkey(a.drop_key, (unsigned)(b * a.H + h), (unsigned)(my_q + 32)) ^ (unsigned)(4 * hh);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    STAMP(st1);
    {   // query rows -> accumulator registers (B operands of the score MFMAs): tile wq of the Q region, the image
        // and the fragment reads are those of a K tile
        bf16x8 qf[2][KS];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) qf[blk][ks] = lds_read_b128(smem, koff[ks] + QREG + wq * TILE + 8192 * blk);
        static_for<2 * KS * 4>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const u32x4 w = __builtin_bit_cast(u32x4, qf[i / (4 * KS)][(i / 4) % KS]);
            const unsigned word = w[i % 4];   // (named: a bit cast applied to a vector element reads element 0)
            acc_write<QACC + i>(__builtin_bit_cast(float, word));
        });
    }
    if (jl >= 0) {
        static_for<2 * KS>([&](auto uc) {
            constexpr int u = decltype(uc)::value, t = u / KS, ks = u % KS;
            const bf16x8 kf = lds_read_b128(smem, koff[ks] + KRING + 8192 * t);
            if constexpr (ks == 0) {
                mfma_vq0<QACC + ks * 4>(sa[0][t], kf);
                mfma_vq0<QACC + (KS + ks) * 4>(sa[1][t], kf);
            } else {
                mfma_vq<QACC + ks * 4>(sa[0][t], kf);
                mfma_vq<QACC + (KS + ks) * 4>(sa[1][t], kf);
            }
        });
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        mx[0] = mx[1] = NEG_BIG;
        if (jl == 0) static_for<32>([&](auto nc) { max_step(sa, nc, std::true_type{}); });
        else static_for<32>([&](auto nc) { max_step(sa, nc, std::false_type{}); });
        // the first maxima are the reference (O and the row sums are zero: nothing to rescale); every row sees key 0
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            m[blk] = fmaxf(mx[blk], other_half(mx[blk]));
            nm[blk] = -m[blk] * sl2;
        }
    }

    auto body = [&](f32x16 (&sc)[2][2], f32x16 (&sn)[2][2], auto parc, int j) {
        constexpr int PAR = decltype(parc)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // K(j+1), V(j) complete; every wave is past its reads of K(j), V(j-1)
        if (j + 2 < J) Ks.issue(64 * (j + 2), lds0 + KRING + PAR * TILE, wave);
        if (j + 1 < J) Vs.issue(64 * (j + 1), lds0 + VRING + (PAR ^ 1) * TILE, wave);
        if (j + 1 < jl) iter(sc, sn, parc, std::integral_constant<int, 0>{}, j);
        else if (j + 1 == jl) iter(sc, sn, parc, std::integral_constant<int, 1>{}, j);
        else if (j == jl) iter(sc, sn, parc, std::integral_constant<int, 2>{}, j);
    };
    STAMP(st2);
    for (int j = 0; j < J; j += 2) {
        body(sa, sb, std::integral_constant<int, 0>{}, j);
        if (j + 1 < J) body(sb, sa, std::integral_constant<int, 1>{}, j + 1);
    }
    STAMP(st3);

    // ---- epilogue. The next pass's first tiles are requested first (they travel while this pass's rows are written
    // out); O^T goes through LDS (8 KiB of this wave's own, one 32-row block at a time) and leaves as whole rows.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // every wave is past its last K and V reads
    const int qw0_out = qw0;
    float lt[2], mm[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        lt[blk] = l[blk] + other_half(l[blk]);
        mm[blk] = m[blk];
    }
    if (pass + 1 < npass) {
        set_block(qb_lo);
        issue_first();
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    char* obuf = smem + STG + wave * 8192;
    static_for<2>([&](auto bc) {
        constexpr int blk = decltype(bc)::value;
        const float inv = (DROP ? a.drop_inv_keep : 1.f) / lt[blk];
        static_for<DB * 4>([&](auto ic) {   // O^T[d = 32 db + 8 g4 + 4 hh + e][q] -> row = lane row, columns d
            constexpr int db = decltype(ic)::value / 4, g4 = decltype(ic)::value % 4;
            bf16x4 ov;
            static_for<4>([&](auto ec) {
                constexpr int e = decltype(ec)::value;
                ov[e] = (bf16)(acc_read<OACC + (blk * DB + db) * 16 + 4 * g4 + e>() * inv);
            });
            *reinterpret_cast<bf16x4*>(obuf + img256_off(lane & 31, 4 * (db0 + db) + g4) + 8 * hh) = ov;
        });
        const int q = qw0_out + 32 * blk + (lane & 31);
        if (q < S && lane < 32 && a.LSE) a.LSE[((long)b * a.H + h) * S + q] = (mm[blk] * sl2 + __log2f(lt[blk])) * LN2;
        // (a wave reads back only what it wrote; with head_dim 64 a row holds its head in its half of the chunks)
        constexpr int CPR = HD / 8;                  // 16-byte chunks per output row
        constexpr int RPI = 64 / CPR;                // rows per store instruction
        const int c = lane % CPR, r0 = lane / CPR;
#pragma unroll
        for (int i = 0; i < 32 / RPI; ++i) {
            const int row = i * RPI + r0, qr = qw0_out + 32 * blk + row;
            const bf16x8 v = lds_read_b128(obuf, img256_off(row, 4 * db0 + c));
            if (qr < S) *reinterpret_cast<bf16x8*>(a.Out + (tok0 + qr) * a.so + (long)h * HD + 8 * c) = v;
        }
    });
#ifdef MGEN_ATTN_STAMPS
    {
        const unsigned long long st4 = __builtin_amdgcn_s_memtime();
        if (lane == 0) {   // per wave (raw 32-bit words): start, wait for the first tiles, prologue, loop, epilogue, J, jl, ids
            unsigned* d = reinterpret_cast<unsigned*>(a.LSE + ((long)b * a.H + h) * S + qw0_out);
            d[0] = (unsigned)st0; d[1] = (unsigned)(st1 - st0); d[2] = (unsigned)(st2 - st1); d[3] = (unsigned)(st3 - st2);
            d[4] = (unsigned)(st4 - st3); d[5] = (unsigned)pass; d[6] = (unsigned)npass;
            d[7] = __builtin_amdgcn_s_getreg((15 << 11) | 4);     // HW_ID bits 0..15: wave, simd, pipe, cu, sh, se
            d[8] = __builtin_amdgcn_s_getreg((3 << 11) | 20);     // XCC_ID
            d[9] = (unsigned)(st0 >> 32);
        }
        st0 = st4;
    }
#endif
    }   // pass
#undef STAMP
}

// ===================================================================================================
// backward, part 0: delta[b,h,q] = sum_d dO * O
// ===================================================================================================
template <int HD>
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16* __restrict__ O, const bf16* __restrict__ dO,
                                                         long so, long sdo, float* __restrict__ delta, int B, int S,
                                                         int H) {
    constexpr int LPR = HD / 8;
    const long nrows = (long)B * S * H;
    const int sub = threadIdx.x % LPR;
    // two rows per lane group and iteration, both requested before the first reduction
    const long stride = (long)gridDim.x * (256 / LPR);
    for (long row0 = (long)blockIdx.x * (256 / LPR) + threadIdx.x / LPR; row0 < nrows; row0 += 2 * stride) {
        const long row1 = row0 + stride;
        const bool two = row1 < nrows;
        const long tok0 = row0 / H, tok1 = two ? row1 / H : tok0;
        const int h0 = (int)(row0 % H), h1 = two ? (int)(row1 % H) : h0;
        const bf16x8 ov0 = ld_stream(reinterpret_cast<const bf16x8*>(O + tok0 * so + (long)h0 * HD + sub * 8));
        const bf16x8 dv0 = *reinterpret_cast<const bf16x8*>(dO + tok0 * sdo + (long)h0 * HD + sub * 8);
        const bf16x8 ov1 = ld_stream(reinterpret_cast<const bf16x8*>(O + tok1 * so + (long)h1 * HD + sub * 8));
        const bf16x8 dv1 = *reinterpret_cast<const bf16x8*>(dO + tok1 * sdo + (long)h1 * HD + sub * 8);
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s0 += (float)ov0[e] * (float)dv0[e];
            s1 += (float)ov1[e] * (float)dv1[e];
        }
        s0 = group_sum<LPR>(s0);
        s1 = group_sum<LPR>(s1);
        if (sub == 0) {
            delta[((tok0 / S) * H + h0) * S + tok0 % S] = s0;
            if (two) delta[((tok1 / S) * H + h1) * S + tok1 % S] = s1;
        }
    }
}

// ===================================================================================================
// backward, part 1: dQ. Same work decomposition as the forward.
// ===================================================================================================
template <int HD, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnArgs a) {
    using G = Geo<HD>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Coords bid = block_coords(a);
    // two query blocks per workgroup, nqb - 1 - x (heavy) then x (light): every workgroup streams the same number
    // of tiles (see the forward kernel)
    const int npass = bid.x < a.nqb - 1 - bid.x ? 2 : 1;
    const int b = bid.z;
    const int sub = HD == 128 ? 0 : (wave >> 1);
    const int wq = HD == 128 ? wave : (wave & 1);
    const int h = HD == 128 ? bid.y : 2 * bid.y + sub;
    const int kv_first = HD == 128 ? h / (a.H / a.KVH) : 2 * bid.y;
    const int ks0 = 4 * sub, db0 = 2 * sub;
    const int S = a.S;
    const long tok0 = (long)b * S;
    TileStream<4> Ks, Vs;
    Ks.init(a.K + tok0 * a.sk + (long)kv_first * HD, a.sk, S, wave, lane);
    Vs.init(a.V + tok0 * a.sv + (long)kv_first * HD, a.sv, S, wave, lane);
    const unsigned lds0 = lds_addr(smem);
    const float sl2 = a.scale * LOG2E;
    for (int pass = 0; pass < npass; ++pass) {
    const int qb = pass == 0 ? a.nqb - 1 - bid.x : bid.x;
    const int q0 = qb * G::ROWS, qw0 = q0 + wq * 32;
    Ks.issue(0, lds0, wave);
    Vs.issue(0, lds0 + TILE, wave);

    bf16x8 qf[G::KS], dof[G::KS];
    load_own_rows<G::KS>(qf, a.Q + (long)h * HD, a.sq, tok0, qw0, S, lane);
    load_own_rows<G::KS>(dof, a.dO + (long)h * HD, a.sdo, tok0, qw0, S, lane);
    const int my_q = qw0 + (lane & 31);
    const int cq = min(my_q, S - 1);
    const float lse2 = a.LSE[((long)b * a.H + h) * S + cq] * LOG2E;
    const float delta = a.Delta[((long)b * a.H + h) * S + cq];
    unsigned rk = 0;   // as in the forward kernel
    if constexpr (DROP) rk = drop_rowkey(a.drop_key, (unsigned)(b * a.H + h), (unsigned)my_q) ^ (unsigned)(4 * (lane >> 5));

    f32x16 dq[G::DB];
#pragma unroll
    for (int db = 0; db < G::DB; ++db) zero16(dq[db]);

    const int last_key = min(q0 + G::ROWS - 1, S - 1);
    const int ntiles = last_key / 64 + 1;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    settle_frags<G::KS>(qf);
    settle_frags<G::KS>(dof);

    for (int j = 0; j < ntiles; ++j) {
        const int cur = j & 1;
        const int kv0 = j * 64;
        if (j + 1 < ntiles) {
            const unsigned nb = lds0 + (cur ^ 1) * 2 * TILE;
            Ks.issue(kv0 + 64, nb, wave);
            Vs.issue(kv0 + 64, nb + TILE, wave);
        }
        const char* Kt = smem + cur * 2 * TILE;
        const char* Vt = Kt + TILE;
        if (kv0 <= qw0 + 31) {
            const bool need_mask = kv0 + 63 > qw0;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x16 s, dp;
                zero16(s);
                zero16(dp);
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks)
                    s = mfma32(frag_row_img256(Kt, 32 * t, ks0 + ks, lane), qf[ks], s);
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks)
                    dp = mfma32(frag_row_img256(Vt, 32 * t, ks0 + ks, lane), dof[ks], dp);
                // key = kv0 + 32t + acc_row(r, lane) is masked when key > my_q; acc_row = c_r + 4*(lane>>5) with c_r a
                // constant of the register, so one per-lane threshold serves all 16 compares (no mask: threshold
                // beyond every c_r)
                const int thr = need_mask ? my_q - kv0 - 32 * t - 4 * (lane >> 5) : 64;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float p = fast_exp2(__builtin_fmaf(s[r], sl2, -lse2));
                    if ((r & 3) + 8 * (r >> 2) > thr) p = 0.f;
                    float dpv = dp[r];
                    if constexpr (DROP) {
                        const unsigned hsh = drop_fmix32(rk ^ (unsigned)(kv0 + 32 * t) ^ (unsigned)((r & 3) + 8 * (r >> 2)));
                        dpv = hsh < a.drop_thresh ? 0.f : dpv * a.drop_inv_keep;
                    }
                    s[r] = p * (dpv - delta);  // dS^T[key][q]
                }
                const bf16x8 f0 = acc_to_frag<0>(s), f1 = acc_to_frag<1>(s);
                // dQ^T[d][q] += K^T[d][key] * dS^T[key][q]
#pragma unroll
                for (int db = 0; db < G::DB; ++db) {
                    dq[db] = mfma32(frag_col_img256<true>(Kt, 32 * t, 32 * (db0 + db), lane), f0, dq[db]);
                    dq[db] = mfma32(frag_col_img256<true>(Kt, 32 * t + 16, 32 * (db0 + db), lane), f1, dq[db]);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (my_q < S) store_rows_T<G::DB>(dq, a.dQ + (tok0 + my_q) * a.sdq + (long)h * HD, a.scale, lane);
    }   // pass (the last tile loop iteration ended with a barrier: both LDS buffers are free again)
}

// ===================================================================================================
// backward, part 2: dK and dV. One block = ROWS keys of one kv head (head_dim 128) or of a pair of kv heads
// (head_dim 64); loops over the query heads of the group and over the query tiles at or after the block's first
// key. LDS per buffer: Q tile | dO tile | lse2[2][64] | delta[2][64]
// ===================================================================================================
// (+ rowkey[2][64] of the dropout hash, + padding to a multiple of 1 KiB)
constexpr int DKV_BUF = 2 * TILE + 2048;

template <int HD, bool DROP>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_kernel(AttnArgs a) {
    using G = Geo<HD>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Coords bid = block_coords(a);
    // two key blocks per workgroup, x (heavy: every query tile from its first key on) then nqb - 1 - x (light):
    // every workgroup streams the same number of tiles (see the forward kernel)
    const int npass = bid.x < a.nqb - 1 - bid.x ? 2 : 1;
    const int b = bid.z;
    const int sub = HD == 128 ? 0 : (wave >> 1);
    const int wk = HD == 128 ? wave : (wave & 1);
    const int kvh = HD == 128 ? bid.y : 2 * bid.y + sub;
    const int ks0 = 4 * sub, db0 = 2 * sub;
    const int S = a.S;
    const long tok0 = (long)b * S;
    const int group = HD == 128 ? a.H / a.KVH : 1;  // head_dim 64 requires H == KVH
    const float sl2 = a.scale * LOG2E;
    // Q and dO tiles of every query head of the group through one descriptor each (the head enters as a byte offset)
    const int hq_first = HD == 128 ? bid.y * group : 2 * bid.y;
    TileStream<4> Qs, dOs;
    Qs.init(a.Q + tok0 * a.sq + (long)hq_first * HD, a.sq, S, wave, lane, HD == 128 ? group * 256 : 256);
    dOs.init(a.dO + tok0 * a.sdo + (long)hq_first * HD, a.sdo, S, wave, lane, HD == 128 ? group * 256 : 256);
    const unsigned lds0 = lds_addr(smem);
    for (int pass = 0; pass < npass; ++pass) {
    const int kb = pass == 0 ? bid.x : a.nqb - 1 - bid.x;
    const int k0 = kb * G::ROWS, kw0 = k0 + wk * 32;
    const int my_k = kw0 + (lane & 31);

    bf16x8 kf[G::KS], vf[G::KS];
    load_own_rows<G::KS>(kf, a.K + (long)kvh * HD, a.sk, tok0, kw0, S, lane);
    load_own_rows<G::KS>(vf, a.V + (long)kvh * HD, a.sv, tok0, kw0, S, lane);

    f32x16 dk[G::DB], dv[G::DB];
#pragma unroll
    for (int db = 0; db < G::DB; ++db) {
        zero16(dk[db]);
        zero16(dv[db]);
    }

    const int t_first = k0 / 64;
    const int t_end = (S + 63) / 64;
    const int per_head = t_end - t_first;
    const int nsteps = per_head * group;

    float s_lse = 0.f, s_delta = 0.f;
    unsigned s_rk = 0;

    auto issue = [&](int step, int buf) {
        // first query head of the tile row for this step
        const int hq0 = HD == 128 ? bid.y * group + step / per_head : 2 * bid.y;
        const int qt0 = (t_first + step % per_head) * 64;
        Qs.issue(qt0, lds0 + buf * DKV_BUF, wave, (hq0 - hq_first) * HD * 2);
        dOs.issue(qt0, lds0 + buf * DKV_BUF + TILE, wave, (hq0 - hq_first) * HD * 2);
        if (tid < 64 * G::HPB) {
            const int q = min(qt0 + (tid & 63), S - 1);
            const long o = ((long)b * a.H + hq0 + (tid >> 6)) * S + q;
            s_lse = a.LSE[o] * LOG2E;
            s_delta = a.Delta[o];
            if constexpr (DROP) s_rk = drop_rowkey(a.drop_key, (unsigned)(b * a.H + hq0 + (tid >> 6)), (unsigned)q);
        }
    };
    auto commit = [&](int buf) {
        char* base = smem + buf * DKV_BUF;
        if (tid < 64 * G::HPB) {
            reinterpret_cast<float*>(base + 2 * TILE)[tid] = s_lse;
            reinterpret_cast<float*>(base + 2 * TILE + 512)[tid] = s_delta;
            if constexpr (DROP) reinterpret_cast<unsigned*>(base + 2 * TILE + 1024)[tid] = s_rk;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    issue(0, 0);
    commit(0);
    __syncthreads();
    settle_frags<G::KS>(kf);
    settle_frags<G::KS>(vf);

    // One step = one 64-query tile. The body exists in a masked and an unmasked form; each form runs in a loop of its
    // own, so that the loop bodies are straight-line code and the dK/dV accumulators stay in place from step to step
    // (with both forms behind a branch in one loop the compiler moved all 128 accumulator registers every step).
    auto run_step = [&](int step, auto masked) {
        constexpr bool MASKED = decltype(masked)::value;
        const int cur = step & 1;
        if (step + 1 < nsteps) issue(step + 1, cur ^ 1);
        const int qt0 = (t_first + step % per_head) * 64;
        const char* Qt = smem + cur * DKV_BUF;
        const char* dOt = Qt + TILE;
        const float* lse_s = reinterpret_cast<const float*>(Qt + 2 * TILE) + 64 * sub;
        const float* del_s = reinterpret_cast<const float*>(Qt + 2 * TILE + 512) + 64 * sub;
        const unsigned* rk_s = reinterpret_cast<const unsigned*>(Qt + 2 * TILE + 1024) + 64 * sub;
        // all four score / dP chains first: the matrix cores then work through the second half's chains while the
        // vector ALU does the first half's exponentials
        f32x16 s[2], dp[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            bf16x8 qa[G::KS], da[G::KS];
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
                qa[ks] = frag_row_img256(Qt, 32 * u, ks0 + ks, lane);
                da[ks] = frag_row_img256(dOt, 32 * u, ks0 + ks, lane);
            }
            mfma_chain_vgpr<G::KS>(s[u], qa, kf);
            mfma_chain_vgpr<G::KS>(dp[u], da, vf);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            // column fragments of both transposed products, issued ahead of the exponentials that hide their latency
            bf16x8 dot[G::DB][2], qtf[G::DB][2];
#pragma unroll
            for (int db = 0; db < G::DB; ++db) {
                const int c0 = 32 * (db0 + db);
                dot[db][0] = frag_col_img256<true>(dOt, 32 * u, c0, lane);
                dot[db][1] = frag_col_img256<true>(dOt, 32 * u + 16, c0, lane);
                qtf[db][0] = frag_col_img256<true>(Qt, 32 * u, c0, lane);
                qtf[db][1] = frag_col_img256<true>(Qt, 32 * u + 16, c0, lane);
            }
            // s[u][reg]: query = qt0 + 32u + acc_row(reg), key = my_k
            f32x16 ds;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int qr = 32 * u + 8 * g4 + 4 * (lane >> 5);
                const f32x4 lv = *reinterpret_cast<const f32x4*>(lse_s + qr);
                const f32x4 dl = *reinterpret_cast<const f32x4*>(del_s + qr);
                u32x4 rv = {0u, 0u, 0u, 0u};
                if constexpr (DROP) rv = *reinterpret_cast<const u32x4*>(rk_s + qr);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g4 + e;
                    float p = fast_exp2(__builtin_fmaf(s[u][r], sl2, -lv[e]));
                    if (MASKED) {
                        const int q = qt0 + qr + e;
                        if (my_k > q || q >= S) p = 0.f;
                    }
                    if constexpr (DROP) {   // dV sees the dropped probabilities, dS the dropped dP
                        const bool keep = drop_fmix32(rv[e] ^ (unsigned)my_k) >= a.drop_thresh;
                        s[u][r] = keep ? p : 0.f;
                        ds[r] = p * ((keep ? dp[u][r] * a.drop_inv_keep : 0.f) - dl[e]);
                    } else {
                        s[u][r] = p;
                        ds[r] = p * (dp[u][r] - dl[e]);
                    }
                }
            }
            const bf16x8 p0 = acc_to_frag<0>(s[u]), p1 = acc_to_frag<1>(s[u]);
            const bf16x8 d0 = acc_to_frag<0>(ds), d1 = acc_to_frag<1>(ds);
#pragma unroll
            for (int db = 0; db < G::DB; ++db) {
                // dV^T[d][key] += dO^T[d][q] * P[q][key] ; dK^T[d][key] += Q^T[d][q] * dS[q][key]
                dv[db] = mfma32(dot[db][0], p0, dv[db]);
                dv[db] = mfma32(dot[db][1], p1, dv[db]);
                dk[db] = mfma32(qtf[db][0], d0, dk[db]);
                dk[db] = mfma32(qtf[db][1], d1, dk[db]);
            }
        }
        if (step + 1 < nsteps) commit(cur ^ 1);
        __syncthreads();
    };
    // per head: the tiles that touch the block's diagonal (masked), the tiles below it (every pair visible), and a
    // last tile that runs past the sequence end (masked) when S is not a multiple of 64
    const int n_diag = min(per_head, G::ROWS / 64);
    const int n_tail = ((S & 63) != 0 && per_head > n_diag) ? 1 : 0;
    int step = 0;
    for (int hh = 0; hh < group; ++hh) {
        int j = 0;
        for (; j < n_diag; ++j, ++step) run_step(step, std::true_type{});
        for (; j < per_head - n_tail; ++j, ++step) run_step(step, std::false_type{});
        for (; j < per_head; ++j, ++step) run_step(step, std::true_type{});
    }
    if (my_k < S) {
        store_rows_T<G::DB>(dk, a.dK + (tok0 + my_k) * a.sdk + (long)kvh * HD, a.scale, lane);
        store_rows_T<G::DB>(dv, a.dV + (tok0 + my_k) * a.sdv + (long)kvh * HD, DROP ? a.drop_inv_keep : 1.f, lane);
    }
    }   // pass (the last step ended with a barrier: both LDS buffers are free again)
}

// ===================================================================================================
// backward, part 2 without dropout: dK and dV with 64 keys per wave.
//
// In the kernel above every fragment read from LDS feeds one MFMA, which keeps the LDS port as busy as the matrix
// cores. Here a wave owns 64 keys (two 32-key blocks A and B that share every Q / dO fragment it reads), one wave per
// SIMD with the whole register file: the dK^T / dV^T accumulators fill the accumulator half (addressed by number, as
// in the forward kernel), the wave's K rows stay in vector registers as B operands, its V rows are read from a
// resident LDS image. A workgroup covers 256 keys of one kv head (head_dim 128) or 128 keys of a pair of heads.
//
// One unit = 32 queries x 64 keys. Its MFMAs run in the fixed order
//     P: dP = dO V^T   V: dV^T += dO^T P   K (first part): dK^T += Q^T dS   S: scores of the NEXT unit   K (rest)
// and the vector work sits between them: the exponentials of the next unit's scores beside K (rest) and the next
// unit's P, dS = P (dP - delta) beside V. Every MFMA, every vector step and every LDS read is an asm statement or
// sits between scheduling fences, so this placement is what runs.
//
// Q travels through a three-slot ring (the scores of a tile are taken one unit before the tile's other products),
// dO through a two-slot ring, LSE and delta beside them; one workgroup barrier per 64-query tile, placed between
// the V and K groups of the tile's second unit: behind it the slots of the tile before (Q) and of this tile (dO)
// are free and the tile after the next is requested.
//
// A wave skips the tiles whose queries all precede its keys (it only takes part in the barrier and the loads);
// its first tile is the one on its diagonal, computed with the causal mask, as is a last tile that runs past the
// sequence end.
// ===================================================================================================
struct VecStream {   // 64 consecutive floats of a [heads][S] array per wave instruction, by LDS-DMA
    u32x4 rsrc;
    int voff;
    __device__ __forceinline__ void init(const float* base, long nfloats, int lane) {
        const unsigned long long p = (unsigned long long)base;
        rsrc[0] = __builtin_amdgcn_readfirstlane((unsigned)p);
        rsrc[1] = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32) & 0xFFFFu);
        rsrc[2] = __builtin_amdgcn_readfirstlane((unsigned)(nfloats * 4));
        rsrc[3] = 0x00020000u;
        voff = lane * 4;
    }
    // floats [elem0, elem0 + 64) -> 256 bytes at LDS byte address lds_dst (past the array's end: zeros)
    __device__ __forceinline__ void issue(long elem0, unsigned lds_dst) const {
        unsigned keep;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
        const int soff = __builtin_amdgcn_readfirstlane((int)(elem0 * 4));
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "buffer_load_dword %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    }
};

// d(v) (+)= A(v) * B(v): score and dP chains
__device__ __forceinline__ void mfma_vv0(f32x16& d, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_vv(f32x16& d, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ float a_sub(float x, float y) {
    float d; asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d;
}
__device__ __forceinline__ float a_mul(float x, float y) {
    float d; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d;
}

// The lane index from an asm statement: what is derived from it is computed where it is used, not carried in
// registers from the kernel's start (through loops that have no registers to spare).
__device__ __forceinline__ int fresh_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

template <int HD> struct Dkv64 {
    static constexpr int KROWS = HD == 128 ? 256 : 128;   // keys of one head per workgroup
    static constexpr int NVT = KROWS / 64;                // 64-row tiles of the resident V image
    static constexpr int HPB = HD == 128 ? 1 : 2;
    static constexpr int VRES = 0, DORING = NVT * TILE, QRING = DORING + 2 * TILE, LSER = QRING + 3 * TILE;
    static constexpr int VECB = 64 * HPB * 4;             // bytes of one LSE / delta slot
    static constexpr int DELR = LSER + 3 * VECB;
    static constexpr int LDS_BYTES = DELR + 2 * VECB;
};

// LDS reads through an LDS byte address held in a register (the workgroup's LDS base is folded into the per-lane
// offsets once, so a read costs no address arithmetic beyond the XOR that selects its chunk)
__device__ __forceinline__ bf16x8 ldsa_read_b128(int addr) {
#ifdef MGEN_DBG_NOLDS
    return bf16x8{};
#endif
    return *(const __attribute__((address_space(3))) bf16x8*)(unsigned)addr;
}
__device__ __forceinline__ f32x4 ldsa_read_f32x4(int addr) {
    return *(const __attribute__((address_space(3))) f32x4*)(unsigned)addr;
}
__device__ __forceinline__ bf16x8 ldsa_col_frag(int addr_a, int addr_b) {
#ifdef MGEN_DBG_NOLDS
    return bf16x8{};
#endif
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(unsigned)addr_a);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(unsigned)addr_b);
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, both);
}
// One 64-row tile piece set of a wave (TileStream<4>: 4 pieces 16 rows apart) in ONE asm statement: M0 and the scalar
// offset advance inside it (M0 is declared clobbered instead of being saved and restored around every load).
__device__ __forceinline__ void tile_issue4(const TileStream<4>& t, int row0, unsigned lds_base, int wave, int extra = 0) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)wave * 1024u);
    int soff = __builtin_amdgcn_readfirstlane(row0 * t.row_bytes + extra);
    const int step = __builtin_amdgcn_readfirstlane(16 * t.row_bytes);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %2, %3, %0 offen lds\n\t"
                 "s_add_u32 m0, m0, 0x1000\n\ts_add_u32 %0, %0, %4\n\t"
                 "buffer_load_dwordx4 %2, %3, %0 offen lds\n\t"
                 "s_add_u32 m0, m0, 0x1000\n\ts_add_u32 %0, %0, %4\n\t"
                 "buffer_load_dwordx4 %2, %3, %0 offen lds\n\t"
                 "s_add_u32 m0, m0, 0x1000\n\ts_add_u32 %0, %0, %4\n\t"
                 "buffer_load_dwordx4 %2, %3, %0 offen lds"
                 : "+s"(soff) : "s"(dst), "v"(t.voff), "s"(t.rsrc), "s"(step) : "memory", "m0", "scc");
}
template <int HD>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv64_kernel(AttnArgs a) {
    using G = Geo<HD>;
    using L = Dkv64<HD>;
    constexpr int KS = G::KS, DB = G::DB;
    constexpr int DVACC = 0, DKACC = 2 * DB * 16;         // accumulator registers: dV^T[db][blk] tiles, then dK^T
    // MFMA slots of one unit
    constexpr int NP = 2 * KS, NV = 4 * DB, NS = 2 * KS, NK = 4 * DB, KH = NK / 4;
    constexpr int pV = NP, pK0 = pV + NV, pS = pK0 + KH, pK1 = pS + NS, NU = pK1 + NK - KH;
    constexpr int LEAD = 4, LEADS = KH < 4 ? KH : 4;      // slots between an LDS read and its MFMA
    constexpr int DSTEP = (NU - pK0) / 9;                 // slots between two pieces of the next loads
    // vector steps: exponentials (34 steps: multiply-add of element n, exponential of n - 1, packing of pairs). Packed
    // fp32 instructions (v_pk_*_f32) are not used: next to MFMAs they run at a fraction of their stand-alone rate
    constexpr int NV1 = 34, W1A = NU - (pS + NS + 1), W1B = NP - 1, N1A = NV1 * W1A / (W1A + W1B);
    constexpr int NV2 = 34, W2 = NV - 2;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    asm volatile("" ::: "a255");   // the kernel allocates all accumulator registers; the compiler uses none of them

    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Coords bid = block_coords(a);
    const int npass = bid.x < a.nqb - 1 - bid.x ? 2 : 1;
    const int b = bid.z;
    const int sub = HD == 128 ? 0 : (wave >> 1);
    const int wk = HD == 128 ? wave : (wave & 1);
    const int kvh = HD == 128 ? bid.y : 2 * bid.y + sub;
    const int ks0 = 4 * sub, db0 = 2 * sub;
    const int S = a.S;
    const long tok0 = (long)b * S;
    const int group = HD == 128 ? a.H / a.KVH : 1;
    const float sl2 = a.scale * LOG2E;
    const int hq_first = HD == 128 ? bid.y * group : 2 * bid.y;
    const int kv_row = HD == 128 ? bid.y : 2 * bid.y;    // first kv head of the V tile row
    TileStream<4> Qs, dOs;
    Qs.init(a.Q + tok0 * a.sq + (long)hq_first * HD, a.sq, S, wave, lane, HD == 128 ? group * 256 : 256);
    dOs.init(a.dO + tok0 * a.sdo + (long)hq_first * HD, a.sdo, S, wave, lane, HD == 128 ? group * 256 : 256);
    VecStream Ls, Ds;
    const int nheads_row = HD == 128 ? group : 2;
    Ls.init(a.LSE + ((long)b * a.H + hq_first) * S, (long)nheads_row * S, lane);
    Ds.init(a.Delta + ((long)b * a.H + hq_first) * S, (long)nheads_row * S, lane);
    const unsigned lds0 = lds_addr(smem);

    // per-lane LDS offsets; a ring slot's base is added and the k-step / d-block enters by XOR (see row_frag_off)
    // (LDS byte ADDRESSES: the workgroup's LDS base, a multiple of 1 KiB, is part of them)
    const int rrow = (int)lds0 + row_frag_off(ks0, lane);
    const int ccol = (int)lds0 + col_frag_off(db0, false, lane);
    const int aV = rrow + L::VRES + wk * TILE;
    const int vecoff = (int)lds0 + 16 * hh + 256 * sub;

    for (int pass = 0; pass < npass; ++pass) {
    const int kb = pass == 0 ? bid.x : a.nqb - 1 - bid.x;
    const int k0 = kb * L::KROWS, kw0 = k0 + 64 * wk;
    const int t_first = k0 / 64, t_end = (S + 63) / 64;
    const int per_head = t_end - t_first;
    const int nsteps = per_head * group;
    const int skipn = kw0 < S ? wk : per_head;           // tiles of a head sweep before this wave's diagonal tile

    // the loads of step s in 9 pieces (4 of the Q tile, 4 of the dO tile, this wave's part of LSE / delta): issued
    // back to back they hold the wave for ~60 cycles each (a CU takes 64 bytes per cycle from L2), spread over the
    // MFMA slots behind the barrier they cost a few cycles each
    // (the step's scalars are worked out once, in plan_step; a piece then costs two scalar adds and the load)
    int pl_qt0 = 0, pl_extra = 0, pl_hrel = 0;
    unsigned pl_q = 0, pl_d = 0, pl_l = 0, pl_e = 0;
    bool pl_on = false;
    auto plan_step = [&](int s) __attribute__((always_inline)) {
        pl_on = s < nsteps;
        pl_hrel = HD == 128 ? s / per_head : 0;
        pl_qt0 = (t_first + s % per_head) * 64;
        pl_extra = pl_hrel * HD * 2;
        const unsigned q3 = (unsigned)(s % 3), d2 = (unsigned)(s & 1);
        pl_q = lds0 + L::QRING + q3 * TILE;
        pl_d = lds0 + L::DORING + d2 * TILE;
        pl_l = lds0 + L::LSER + q3 * L::VECB;
        pl_e = lds0 + L::DELR + d2 * L::VECB;
    };
    auto issue_piece = [&](int i) __attribute__((always_inline)) {
        if (!pl_on) return;
        if (i < 4) Qs.issue_piece_m0(i, pl_qt0, pl_q, wave, pl_extra);
        else if (i < 8) dOs.issue_piece_m0(i - 4, pl_qt0, pl_d, wave, pl_extra);
        else if (wave < L::HPB) Ls.issue((long)(pl_hrel + wave) * S + pl_qt0, pl_l + wave * 256);
        else if (wave >= 2 && wave < 2 + L::HPB) Ds.issue((long)(pl_hrel + wave - 2) * S + pl_qt0, pl_e + (wave - 2) * 256);
    };
    auto issue_step = [&](int s) __attribute__((always_inline)) {
        plan_step(s);
#pragma unroll
        for (int i = 0; i < 9; ++i) issue_piece(i);
    };
    {
        TileStream<4> Vs;
        Vs.init(a.V + tok0 * a.sv + (long)kv_row * HD, a.sv, S, wave, fresh_lane());
#pragma unroll
        for (int t = 0; t < L::NVT; ++t) Vs.issue(k0 + 64 * t, lds0 + L::VRES + t * TILE, wave);
    }
    issue_step(0);
    if (nsteps > 1) issue_step(1);

    bf16x8 kf[2][KS];
    {
        const int lp = fresh_lane();
        int hk;   // (an asm copy: the pointers are formed here, not kept in registers from the kernel's start)
        asm volatile("s_mov_b32 %0, %1" : "=s"(hk) : "s"(kvh));
        load_own_rows<KS>(kf[0], a.K + (long)hk * HD, a.sk, tok0, kw0, S, lp);
        load_own_rows<KS>(kf[1], a.K + (long)hk * HD, a.sk, tok0, kw0 + 32, S, lp);
    }
    static_for<4 * DB * 16>([&](auto ic) { acc_write<decltype(ic)::value>(0.f); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    settle_frags<KS>(kf[0]);
    settle_frags<KS>(kf[1]);

    // ---- state of the pipeline
    f32x16 sc[2], dp[2];                 // scores of the unit whose exponentials are in progress; dP of the current unit
    float xs[32], ps[32], ts[32], dsv[32];
    unsigned pw[16], dw[16];             // packed P and dS fragments: fragment (blk, half) = words 4 (2 blk + half) ..
    f32x4 lv[4], dl[4];                  // LSE / delta of 4 query rows each
    float nlg[4];
    int thrA[2], thrS = 0;               // causal / sequence-end thresholds of the unit whose exponentials are in progress
    thrA[0] = thrA[1] = 0;
    bf16x8 fD[KS], fVA[KS], fVB[KS], fDT[2 * DB], fQn[KS], fQT[2 * DB];

    // element e of a unit, in the order both vector stages walk: row group g4 = e / 8, block (e / 4) % 2,
    // register 4 g4 + e % 4 of the block's tile; its packed word
    auto widx = [](int e) constexpr { return (((e >> 2) & 1) * 2 + (e >> 4)) * 4 + 2 * ((e >> 3) & 1) + ((e & 3) >> 1); };

    auto read_lse = [&](auto gc, int lseb, int un) {
        constexpr int g4 = decltype(gc)::value;
        lv[g4] = ldsa_read_f32x4(lseb + vecoff + 128 * un + 32 * g4);
    };
    auto read_delta = [&](auto gc, int delb, int u) {
        constexpr int g4 = decltype(gc)::value;
        dl[g4] = ldsa_read_f32x4(delb + vecoff + 128 * u + 32 * g4);
    };
    // exponential stage, step n in [0, NV1): p = exp2(s * scale * log2 e - lse * log2 e)
    auto v1_step = [&](auto nc, auto maskc, int lseb, int un) __attribute__((always_inline)) {
        constexpr int n = decltype(nc)::value;
#ifdef MGEN_DBG_NOVALU
        return;
#endif
        constexpr bool MASKED = decltype(maskc)::value;
        if constexpr (n < 32) {
            constexpr int e = n, g4 = e >> 3, blk = (e >> 2) & 1, i = e & 3, r = 4 * g4 + i;
            if constexpr ((e & 7) == 0 && g4 + 1 < 4) read_lse(std::integral_constant<int, g4 + 1>{}, lseb, un);
            if constexpr (blk == 0) nlg[i] = a_mul(lv[g4][i], -LOG2E);
            float x = a_fma(sc[blk][r], sl2, nlg[i]);
            if constexpr (MASKED) {   // hidden when c_r < thrA (key after query) or c_r >= thrS (query past the end)
                constexpr int c = (r & 3) + 8 * (r >> 2);
                float big;   // -1e30
                asm volatile("v_mov_b32 %1, 0xf149f2ca\n\t"
                             "v_cmp_lt_i32_e32 vcc, %2, %3\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\t"
                             "v_cmp_ge_i32_e32 vcc, %2, %4\n\tv_cndmask_b32_e32 %0, %0, %1, vcc"
                             : "+v"(x), "=&v"(big) : "n"(c), "v"(thrA[blk]), "v"(thrS) : "vcc");
            }
            xs[e] = x;
        }
        if constexpr (n >= 1 && n < 33) ps[n - 1] = a_exp2(xs[n - 1]);
        if constexpr (n >= 3 && (n & 1)) {
            if constexpr (n == 33) asm volatile("s_nop 0");   // the exponential right before this statement
            pw[widx(n - 3)] = a_cvt_pk_bf16(ps[n - 3], ps[n - 2]);
        }
    };
    // dS stage, step n in [0, NV2): dS = P (dP - delta)
    auto v2_step = [&](auto nc, int delb, int u) __attribute__((always_inline)) {
        constexpr int n = decltype(nc)::value;
#ifdef MGEN_DBG_NOVALU
        return;
#endif
        if constexpr (n < 32) {
            constexpr int e = n, g4 = e >> 3, blk = (e >> 2) & 1, i = e & 3, r = 4 * g4 + i;
            if constexpr ((e & 7) == 0 && g4 + 1 < 4) read_delta(std::integral_constant<int, g4 + 1>{}, delb, u);
            ts[e] = a_sub(dp[blk][r], dl[g4][i]);
        }
        if constexpr (n >= 1 && n < 33) dsv[n - 1] = a_mul(ps[n - 1], ts[n - 1]);
        if constexpr (n >= 2 && (n & 1) == 0) dw[widx(n - 2)] = a_cvt_pk_bf16(dsv[n - 2], dsv[n - 1]);
    };
    auto frag_of = [&](const unsigned (&w)[16], int f) {
        const u32x4 v = {w[4 * f], w[4 * f + 1], w[4 * f + 2], w[4 * f + 3]};
        return __builtin_bit_cast(bf16x8, v);
    };
    auto set_thr = [&](int qt0, int un) __attribute__((always_inline)) {   // (rare: the lane's part is computed here)
        const int lt = fresh_lane(), h4 = 4 * (lt >> 5);
        thrA[0] = kw0 + (lt & 31) - h4 - qt0 - 32 * un;   // block A's key of this lane, minus the lane half's row offset
        thrA[1] = thrA[0] + 32;
        thrS = S - qt0 - 32 * un - h4;
    };
    // fragments the P group of a unit needs before its first slots (read at the end of the unit before)
    auto read_p_head = [&](auto ic, int aD, int u) {
        constexpr int i = decltype(ic)::value;   // 0: dO[0], V_A[0]; 1: V_B[0]; 2: dO[1], V_A[1]
        if constexpr (i == 0) { fD[0] = ldsa_read_b128(aD + 8192 * u); fVA[0] = ldsa_read_b128(aV); }
        if constexpr (i == 1) fVB[0] = ldsa_read_b128(aV + 8192);
        if constexpr (i == 2) {
            fD[1] = ldsa_read_b128((aD ^ 32) + 8192 * u);
            fVA[1] = ldsa_read_b128(aV ^ 32);
        }
    };

    int s = 0;   // step = 64-query tile of one query head
#ifdef MGEN_DKV_STAMPS   // diagnostic build: cycles spent waiting at / issuing loads behind the tile barrier
    unsigned long long st_wait = 0, st_issue = 0, st_begin = __builtin_amdgcn_s_memtime();
#endif
    auto sync = [&](bool spread = false) __attribute__((always_inline)) {
#ifdef MGEN_DKV_STAMPS
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#ifdef MGEN_DKV_STAMPS
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
#ifndef MGEN_DBG_NODMA
        if (spread) plan_step(s + 2);
        else if (s + 2 < nsteps) issue_step(s + 2);
#endif
#ifdef MGEN_DKV_STAMPS
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        st_wait += t1 - t0;
        st_issue += t2 - t1;
#endif
    };

    // One unit. U: which half of the tile; M_CUR / M_NEXT: the unit's / the next unit's exponentials take the masks;
    // HAS_NEXT: there is a next unit in this head sweep. Ring bases are LDS byte offsets of this step's (and, for the
    // second unit, the next step's) slots.
    auto unit = [&](auto uc, auto mcurc, auto mnextc, auto hasc, int dob, int dob_n, int qb, int qb_n, int lseb,
                    int lseb_n, int delb, int qt0_c, int qt0_n) __attribute__((always_inline)) {
        constexpr int U = decltype(uc)::value, UN = U ^ 1;
        constexpr bool HAS_NEXT = decltype(hasc)::value;
        if constexpr (decltype(mcurc)::value) set_thr(qt0_c, U);
        const int aD = rrow + dob, cD = ccol + dob, cQ = ccol + qb, aQn = rrow + qb_n, aDn = rrow + dob_n;
        static_for<NU>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            // ---- the slot's MFMA
            if constexpr (g < pV) {
                constexpr int ks = g >> 1, blk = g & 1;
                if constexpr (ks == 0) mfma_vv0(dp[blk], fD[ks], blk ? fVB[ks] : fVA[ks]);
                else mfma_vv(dp[blk], fD[ks], blk ? fVB[ks] : fVA[ks]);
            } else if constexpr (g < pK0) {
                constexpr int v = g - pV, i = v >> 1, blk = v & 1, db = i >> 1, half = i & 1;
                mfma_acc<DVACC + (db * 2 + blk) * 16>(fDT[i], frag_of(pw, blk * 2 + half));
            } else if constexpr (g < pS || g >= pK1) {
                constexpr int k = g < pS ? g - pK0 : g - pK1 + KH, i = k >> 1, blk = k & 1, db = i >> 1, half = i & 1;
                mfma_acc<DKACC + (db * 2 + blk) * 16>(fQT[i], frag_of(dw, blk * 2 + half));
            } else if constexpr (HAS_NEXT) {
                constexpr int sidx = g - pS, ks = sidx >> 1, blk = sidx & 1;
                if constexpr (ks == 0) mfma_vv0(sc[blk], fQn[ks], kf[blk][ks]);
                else mfma_vv(sc[blk], fQn[ks], kf[blk][ks]);
            }
            // ---- LDS reads for later slots
            static_for<KS>([&](auto kc) {
                constexpr int ks = decltype(kc)::value;
                if constexpr (ks >= 2 && g == 2 * ks - 3) {
                    fD[ks] = ldsa_read_b128((aD ^ (32 * ks)) + 8192 * U);
                    fVA[ks] = ldsa_read_b128(aV ^ (32 * ks));
                }
                if constexpr (ks >= 1 && g == 2 * ks - 2) fVB[ks] = ldsa_read_b128((aV ^ (32 * ks)) + 8192);
                if constexpr (HAS_NEXT && g == pS + 2 * ks - LEADS)
                    fQn[ks] = ldsa_read_b128((aQn ^ (32 * ks)) + 8192 * UN);
            });
            static_for<2 * DB>([&](auto ic) {
                constexpr int i = decltype(ic)::value, db = i >> 1, half = i & 1;
                constexpr int im = 8192 * U + 4096 * half;
                if constexpr (g == pV + 2 * i - LEAD) {
                    const int x = cD ^ (64 * db);
                    fDT[i] = ldsa_col_frag(x + im, (x ^ 32) + 2048 + im);
                }
                constexpr int ck = 2 * i < KH ? pK0 + 2 * i : pK1 + 2 * i - KH;
                if constexpr (g == ck - LEAD) {
                    const int x = cQ ^ (64 * db);
                    fQT[i] = ldsa_col_frag(x + im, (x ^ 32) + 2048 + im);
                }
            });
            if constexpr (HAS_NEXT && g >= NU - 3) read_p_head(std::integral_constant<int, g - (NU - 3)>{}, aDn, UN);
            // ---- vector steps
            if constexpr (g < W1B) {          // rest of this unit's exponentials
                constexpr int lo = N1A + g * (NV1 - N1A) / W1B, hi = N1A + (g + 1) * (NV1 - N1A) / W1B;
                static_for<hi - lo>([&](auto ic) {
                    v1_step(std::integral_constant<int, lo + decltype(ic)::value>{}, mcurc, lseb, U);
                });
            }
            if constexpr (g == pV) read_delta(std::integral_constant<int, 0>{}, delb, U);
            if constexpr (g >= pV + 2 && g < pK0) {   // dS of this unit
                constexpr int w = g - (pV + 2), lo = w * NV2 / W2, hi = (w + 1) * NV2 / W2;
                static_for<hi - lo>([&](auto ic) {
                    v2_step(std::integral_constant<int, lo + decltype(ic)::value>{}, delb, U);
                });
            }
            if constexpr (HAS_NEXT && g == pS + NS - 1) {
                read_lse(std::integral_constant<int, 0>{}, lseb_n, UN);
                if constexpr (decltype(mnextc)::value) set_thr(qt0_n, UN);
            }
            if constexpr (HAS_NEXT && g > pS + NS) {   // first part of the next unit's exponentials
                constexpr int w = g - (pS + NS + 1), lo = w * N1A / W1A, hi = (w + 1) * N1A / W1A;
                static_for<hi - lo>([&](auto ic) {
                    v1_step(std::integral_constant<int, lo + decltype(ic)::value>{}, mnextc, lseb_n, UN);
                });
            }
            if constexpr (U == 1 && g == pK0 - 1) sync(true);
            if constexpr (U == 1 && g >= pK0 && (g - pK0) % DSTEP == 0 && (g - pK0) / DSTEP < 9) {
                issue_piece((g - pK0) / DSTEP);
            }
            SCHED_FENCE();
        });
    };
    // what the unit before would have left behind for a wave's first unit of a head sweep: the unit's scores, the first
    // part of its exponentials and the first P-group fragments
    auto prologue = [&](int dob, int qb, int lseb, int qt0) __attribute__((always_inline)) {
        const int aQ = rrow + qb;
        static_for<KS>([&](auto kc) {
            constexpr int ks = decltype(kc)::value;
            const bf16x8 f = ldsa_read_b128(aQ ^ (32 * ks));
            if constexpr (ks == 0) { mfma_vv0(sc[0], f, kf[0][ks]); mfma_vv0(sc[1], f, kf[1][ks]); }
            else { mfma_vv(sc[0], f, kf[0][ks]); mfma_vv(sc[1], f, kf[1][ks]); }
        });
        read_lse(std::integral_constant<int, 0>{}, lseb, 0);
        set_thr(qt0, 0);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        SCHED_FENCE();
        static_for<N1A>([&](auto nc) { v1_step(nc, std::true_type{}, lseb, 0); });
        static_for<3>([&](auto ic) { read_p_head(ic, rrow + dob, 0); });
        SCHED_FENCE();
    };

    const std::integral_constant<int, 0> U0;
    const std::integral_constant<int, 1> U1;
    const std::true_type T;
    const std::false_type F;
    // One step = the two units of a tile. A body that applies masks computes the thresholds of the unit it applies
    // them to, so masks may be on for a unit that needs none (slower, same result); a unit that needs them must have
    // them on in both bodies that carry its exponentials (the first part runs in the unit before).
    //   MASKED  masks on for this tile and for the start of the next one
    //   PLAIN   no masks
    //   LAST    masks on, last tile of the head sweep
    auto step = [&](auto kindc, int jt) __attribute__((always_inline)) {
        constexpr int KIND = decltype(kindc)::value;
        const int q3 = s % 3, q3n = (s + 1) % 3, d2 = s & 1;
        const int dob = L::DORING + d2 * TILE, dobn = L::DORING + (d2 ^ 1) * TILE;
        const int qb = L::QRING + q3 * TILE, qbn = L::QRING + q3n * TILE;
        const int lseb = L::LSER + q3 * L::VECB, lsebn = L::LSER + q3n * L::VECB;
        const int delb = L::DELR + d2 * L::VECB;
        const int qt0 = (t_first + jt) * 64;
        if constexpr (KIND == 0) {
            unit(U0, T, T, T, dob, dob, qb, qb, lseb, lseb, delb, qt0, qt0);
            unit(U1, T, T, T, dob, dobn, qb, qbn, lseb, lsebn, delb, qt0, qt0 + 64);
        } else if constexpr (KIND == 1) {
            unit(U0, F, F, T, dob, dob, qb, qb, lseb, lseb, delb, qt0, qt0);
            unit(U1, F, F, T, dob, dobn, qb, qbn, lseb, lsebn, delb, qt0, qt0 + 64);
        } else {
            unit(U0, T, T, T, dob, dob, qb, qb, lseb, lseb, delb, qt0, qt0);
            unit(U1, T, F, F, dob, dobn, qb, qbn, lseb, lsebn, delb, qt0, qt0 + 64);
        }
        ++s;
    };
    const std::integral_constant<int, 0> MASKED;
    const std::integral_constant<int, 1> PLAIN;
    const std::integral_constant<int, 2> LAST;
    // tiles of a head sweep for this wave: skipped ones, the diagonal tile (masked), plain tiles, a masked tile in front
    // of a last tile that runs past the sequence end, the last tile
    const int plain_end = per_head - 1 - ((S & 63) != 0 ? 1 : 0);
    for (int hq = 0; hq < group; ++hq) {
        for (int jt = 0; jt < skipn; ++jt, ++s) sync();
        if (skipn < per_head) {
            int jt = skipn;
            prologue(L::DORING + (s & 1) * TILE, L::QRING + (s % 3) * TILE, L::LSER + (s % 3) * L::VECB, (t_first + jt) * 64);
            if (jt < per_head - 1) {
                step(MASKED, jt);
                ++jt;
                for (; jt < plain_end; ++jt) step(PLAIN, jt);
                for (; jt < per_head - 1; ++jt) step(MASKED, jt);
            }
            step(LAST, jt);
        }
    }

    // ---- epilogue: dV^T and dK^T leave through LDS (8 KiB of the dO ring per wave, one 32-key block at a time) as
    // whole rows. The dO ring is free: every wave's last reads of it precede the last barrier.
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    char* obuf = smem + L::DORING + wave * 8192;
    const int le = fresh_lane(), he = le >> 5;
    int hk;
    asm volatile("s_mov_b32 %0, %1" : "=s"(hk) : "s"(kvh));
    static_for<4>([&](auto oc) {
        constexpr int which = decltype(oc)::value >> 1, blk = decltype(oc)::value & 1;   // 0: dV, 1: dK
        const float mul = which ? a.scale : 1.f;
        static_for<DB * 4>([&](auto ic) {
            constexpr int db = decltype(ic)::value / 4, g4 = decltype(ic)::value % 4;
            bf16x4 ov;
            static_for<4>([&](auto ec) {
                constexpr int e = decltype(ec)::value;
                ov[e] = (bf16)(acc_read<(which ? DKACC : DVACC) + (db * 2 + blk) * 16 + 4 * g4 + e>() * mul);
            });
            *reinterpret_cast<bf16x4*>(obuf + img256_off(le & 31, 4 * (db0 + db) + g4) + 8 * he) = ov;
        });
        constexpr int CPR = HD / 8, RPI = 64 / CPR;
        const int c = le % CPR, r0 = le / CPR;
        bf16* dst = which ? a.dK : a.dV;
        const long ld = which ? a.sdk : a.sdv;
#pragma unroll
        for (int i = 0; i < 32 / RPI; ++i) {
            const int row = i * RPI + r0, key = kw0 + 32 * blk + row;
            const bf16x8 v = lds_read_b128(obuf, img256_off(row, 4 * db0 + c));
            if (key < S) *reinterpret_cast<bf16x8*>(dst + (tok0 + key) * ld + (long)hk * HD + 8 * c) = v;
        }
    });
    __syncthreads();   // the rings and the V image are free for the next pass
#ifdef MGEN_DKV_STAMPS
    if (le == 0 && kw0 < S) {
        unsigned* d = reinterpret_cast<unsigned*>(a.dV + (tok0 + kw0) * a.sdv + (long)kvh * HD);
        const unsigned long long t3 = __builtin_amdgcn_s_memtime();
        d[0] = (unsigned)(t3 - st_begin); d[1] = (unsigned)st_wait; d[2] = (unsigned)st_issue; d[3] = (unsigned)nsteps;
        d[4] = (unsigned)skipn; d[5] = (unsigned)pass;
    }
#endif
    }   // pass
}

// ===================================================================================================
// backward, part 1 without dropout: dQ with 64 queries per wave (the forward kernel's decomposition).
//
// One wave per SIMD; the wave's 64 query rows (blocks A and B) stay in accumulator registers as the B operands of
// the score products (Q) and of the dP products (dO), next to the dQ^T accumulators, all addressed by number; K and V
// tiles are streamed. One unit = 32 keys x 64 queries:
//     P: dP^T = V dO^T        S: scores of the NEXT unit, K Q^T        Q: dQ^T += K^T dS^T
// with the exponentials of the next unit's scores beside Q and the next unit's P, and dS = P (dP - delta) beside S.
// K goes through a three-slot ring, V through a two-slot ring, one workgroup barrier per 64-key tile between the P and
// S groups of the tile's second unit. A wave's last tile is the one on its diagonal (masked); after it the wave only
// takes part in the barriers and the loads.
// ===================================================================================================
template <int HD> struct Dq64 {
    static constexpr int QROWS = HD == 128 ? 256 : 128;
    static constexpr int KRING = 0, VRING = 3 * TILE, STG = 5 * TILE;
    static constexpr int LDS_BYTES = STG + 4 * 8192;
};

template <int HD>
__global__ __launch_bounds__(256, 1) void attn_bwd_dq64_kernel(AttnArgs a) {
    using G = Geo<HD>;
    using L = Dq64<HD>;
    constexpr int KS = G::KS, DB = G::DB;
    constexpr int DQACC = 0, QACC = 2 * DB * 16, DOACC = QACC + 2 * KS * 4;
    constexpr int NP = 2 * KS, NS = 2 * KS, NQ = 4 * DB;
    constexpr int pS = NP, pQ = NP + NS, NU = pQ + NQ;
    constexpr int LEAD = 4;
    constexpr int NV1 = 33, W1A = NQ - 1, W1B = NP - 1, N1A = NV1 * W1A / (W1A + W1B);
    constexpr int NV2 = 34, W2 = NS - 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    asm volatile("" ::: "a255");   // the kernel allocates all accumulator registers; the compiler uses none of them

    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Coords bid = block_coords(a);
    const int npass = bid.x < a.nqb - 1 - bid.x ? 2 : 1;
    const int b = bid.z;
    const int sub = HD == 128 ? 0 : (wave >> 1);
    const int wq = HD == 128 ? wave : (wave & 1);
    const int h = HD == 128 ? bid.y : 2 * bid.y + sub;
    const int kv_first = HD == 128 ? h / (a.H / a.KVH) : 2 * bid.y;
    const int ks0 = 4 * sub, db0 = 2 * sub;
    const int S = a.S;
    const long tok0 = (long)b * S;
    const float sl2 = a.scale * LOG2E;
    TileStream<4> Ks, Vs;
    Ks.init(a.K + tok0 * a.sk + (long)kv_first * HD, a.sk, S, wave, lane);
    Vs.init(a.V + tok0 * a.sv + (long)kv_first * HD, a.sv, S, wave, lane);
    const unsigned lds0 = lds_addr(smem);
    const int rrow = row_frag_off(ks0, lane);
    const int ccol = col_frag_off(db0, false, lane);
    const float neg_big = NEG_BIG;

    for (int pass = 0; pass < npass; ++pass) {
    const int qb = pass == 0 ? a.nqb - 1 - bid.x : bid.x;
    const int q0 = qb * L::QROWS, qw0 = q0 + wq * 64;
    const int J = min(q0 + L::QROWS - 1, S - 1) / 64 + 1;    // tiles the workgroup streams
    const int jl = qw0 < S ? qw0 / 64 : -1;                  // this wave's last tile, the one on its diagonal
    const int qrel = qw0 + (lane & 31) - 4 * hh;              // block A's query of this lane minus the lane half's key offset

    auto issue_tile = [&](int j) __attribute__((always_inline)) {
        Ks.issue(64 * j, lds0 + L::KRING + (unsigned)(j % 3) * TILE, wave);
        Vs.issue(64 * j, lds0 + L::VRING + (unsigned)(j & 1) * TILE, wave);
    };
    issue_tile(0);
    if (J > 1) issue_tile(1);

    float nl[2], delta[2];
    {   // the wave's rows of Q and dO -> accumulator registers; LSE and delta of its two rows
        const int lp = fresh_lane();
        bf16x8 qf[KS];
        static_for<4>([&](auto ic) {
            constexpr int i = decltype(ic)::value, which = i >> 1, blk = i & 1;
            load_own_rows<KS>(qf, (which ? a.dO : a.Q) + (long)h * HD, which ? a.sdo : a.sq, tok0, qw0 + 32 * blk, S, lp);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            static_for<KS * 4>([&](auto wc) {
                constexpr int w = decltype(wc)::value;
                const u32x4 v = __builtin_bit_cast(u32x4, qf[w / 4]);
                const unsigned word = v[w % 4];   // (named: a bit cast applied to a vector element reads element 0)
                acc_write<(which ? DOACC : QACC) + blk * KS * 4 + w>(__builtin_bit_cast(float, word));
            });
        });
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int cq = min(qw0 + 32 * blk + (lp & 31), S - 1);
            nl[blk] = -a.LSE[((long)b * a.H + h) * S + cq] * LOG2E;
            delta[blk] = a.Delta[((long)b * a.H + h) * S + cq];
        }
    }
    static_for<2 * DB * 16>([&](auto ic) { acc_write<DQACC + decltype(ic)::value>(0.f); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    asm volatile("" : "+v"(nl[0]), "+v"(nl[1]), "+v"(delta[0]), "+v"(delta[1]));

    // ---- state of the pipeline
    f32x16 sc[2], dp[2];
    float xs[32], ps[32], ts[32], dsv[32];
    unsigned dw[16];
    int thr[2];
    thr[0] = thr[1] = 0;
    bf16x8 fV[KS], fK[KS], fKT[2 * DB];

    // element e: 16-key half e / 16, block (e / 8) % 2, register 8 half + e % 8 of the block's tile
    auto widx = [](int e) constexpr { return (((e >> 3) & 1) * 2 + (e >> 4)) * 4 + ((e & 7) >> 1); };
    auto v1_step = [&](auto nc, auto maskc) __attribute__((always_inline)) {
        constexpr int n = decltype(nc)::value;
#ifdef MGEN_DBG_NOVALU
        return;
#endif
        if constexpr (n < 32) {
            constexpr int e = n, blk = (e >> 3) & 1, r = 8 * (e >> 4) + (e & 7);
            float x = a_fma(sc[blk][r], sl2, nl[blk]);
            if constexpr (decltype(maskc)::value) {   // hidden when the key comes after the query: c_r > thr
                constexpr int c = (r & 3) + 8 * (r >> 2);
                asm volatile("v_cmp_gt_i32_e32 vcc, %2, %3\n\tv_cndmask_b32_e32 %0, %0, %1, vcc"
                             : "+v"(x) : "v"(neg_big), "n"(c), "v"(thr[blk]) : "vcc");
            }
            xs[e] = x;
        }
        if constexpr (n >= 1) ps[n - 1] = a_exp2(xs[n - 1]);
    };
    auto v2_step = [&](auto nc) __attribute__((always_inline)) {
        constexpr int n = decltype(nc)::value;
#ifdef MGEN_DBG_NOVALU
        return;
#endif
        if constexpr (n < 32) {
            constexpr int e = n, blk = (e >> 3) & 1, r = 8 * (e >> 4) + (e & 7);
            ts[e] = a_sub(dp[blk][r], delta[blk]);
        }
        if constexpr (n >= 1 && n < 33) dsv[n - 1] = a_mul(ps[n - 1], ts[n - 1]);
        if constexpr (n >= 2 && (n & 1) == 0) dw[widx(n - 2)] = a_cvt_pk_bf16(dsv[n - 2], dsv[n - 1]);
    };
    auto frag_of = [&](const unsigned (&w)[16], int f) __attribute__((always_inline)) {
        const u32x4 v = {w[4 * f], w[4 * f + 1], w[4 * f + 2], w[4 * f + 3]};
        return __builtin_bit_cast(bf16x8, v);
    };
    auto set_thr = [&](int j, int t) __attribute__((always_inline)) {
        thr[0] = qrel - 64 * j - 32 * t;
        thr[1] = thr[0] + 32;
    };
    // V fragments of a unit's first P slots (read at the end of the unit before)
    auto read_v_head = [&](auto ic, int aVt, int t) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        fV[i] = lds_read_b128(smem, (aVt ^ (32 * i)) + 8192 * t);
    };

    int j = 0;   // tile
    auto sync = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (j + 2 < J) issue_tile(j + 2);
    };

    // U: which 32-key half of tile j; M_CUR / M_NEXT: masks on this unit's / the next unit's exponentials
    auto unit = [&](auto uc, auto mcurc, auto mnextc, auto hasc, int kb, int kb_n, int vb, int vb_n, int j_n)
                    __attribute__((always_inline)) {
        constexpr int U = decltype(uc)::value, UN = U ^ 1;
        constexpr bool HAS_NEXT = decltype(hasc)::value;
        if constexpr (decltype(mcurc)::value) set_thr(j, U);
        const int aVc = rrow + vb, aVn = rrow + vb_n, aKn = rrow + kb_n, cK = ccol + kb;
        static_for<NU>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g < pS) {
                constexpr int ks = g >> 1, blk = g & 1;
                if constexpr (ks == 0) mfma_vq0<DOACC + (blk * KS + ks) * 4>(dp[blk], fV[ks]);
                else mfma_vq<DOACC + (blk * KS + ks) * 4>(dp[blk], fV[ks]);
            } else if constexpr (g < pQ) {
                if constexpr (HAS_NEXT) {
                    constexpr int sidx = g - pS, ks = sidx >> 1, blk = sidx & 1;
                    if constexpr (ks == 0) mfma_vq0<QACC + (blk * KS + ks) * 4>(sc[blk], fK[ks]);
                    else mfma_vq<QACC + (blk * KS + ks) * 4>(sc[blk], fK[ks]);
                }
            } else {
                constexpr int k = g - pQ, i = k >> 1, blk = k & 1, db = i >> 1, half = i & 1;
                mfma_acc<DQACC + (blk * DB + db) * 16>(fKT[i], frag_of(dw, blk * 2 + half));
            }
            // the tile's barrier: behind the last read of V(j), in front of the first read of K(j + 1)
            if constexpr (U == 1 && g == NP - 1) sync();
            // ---- LDS reads for later slots
            static_for<KS>([&](auto kc) {
                constexpr int ks = decltype(kc)::value;
                if constexpr (ks >= 2 && g == 2 * ks - LEAD) fV[ks] = lds_read_b128(smem, (aVc ^ (32 * ks)) + 8192 * U);
                constexpr int want = pS + 2 * ks - LEAD;
                if constexpr (HAS_NEXT && g == (want < NP - 1 ? NP - 1 : want))
                    fK[ks] = lds_read_b128(smem, (aKn ^ (32 * ks)) + 8192 * UN);
            });
            static_for<2 * DB>([&](auto ic) {
                constexpr int i = decltype(ic)::value, db = i >> 1, half = i & 1;
                constexpr int im = 8192 * U + 4096 * half;
                if constexpr (g == pQ + 2 * i - LEAD) {
                    const int x = cK ^ (64 * db);
                    fKT[i] = col_frag(smem, x + im, (x ^ 32) + 2048 + im);
                }
            });
            if constexpr (HAS_NEXT && g >= NU - 2) read_v_head(std::integral_constant<int, g - (NU - 2)>{}, aVn, UN);
            // ---- vector steps
            if constexpr (g < W1B) {
                constexpr int lo = N1A + g * (NV1 - N1A) / W1B, hi = N1A + (g + 1) * (NV1 - N1A) / W1B;
                static_for<hi - lo>([&](auto ic) {
                    v1_step(std::integral_constant<int, lo + decltype(ic)::value>{}, mcurc);
                });
            }
            if constexpr (g >= pS + 2 && g < pQ) {
                constexpr int w = g - (pS + 2), lo = w * NV2 / W2, hi = (w + 1) * NV2 / W2;
                static_for<hi - lo>([&](auto ic) { v2_step(std::integral_constant<int, lo + decltype(ic)::value>{}); });
            }
            if constexpr (HAS_NEXT && decltype(mnextc)::value && g == pQ) set_thr(j_n, UN);
            if constexpr (HAS_NEXT && g > pQ) {
                constexpr int w = g - (pQ + 1), lo = w * N1A / W1A, hi = (w + 1) * N1A / W1A;
                static_for<hi - lo>([&](auto ic) {
                    v1_step(std::integral_constant<int, lo + decltype(ic)::value>{}, mnextc);
                });
            }
            SCHED_FENCE();
        });
    };
    static_assert(LEAD == 4, "the K reads of the score group start in slot NP - 1, behind the barrier");

    const std::integral_constant<int, 0> U0;
    const std::integral_constant<int, 1> U1;
    const std::true_type T;
    const std::false_type F;
    // KIND 0: no masks; 1: masks on for this tile and the start of the next; 2: masks on, the wave's last tile
    auto step = [&](auto kindc) __attribute__((always_inline)) {
        constexpr int KIND = decltype(kindc)::value;
        const int kb = L::KRING + (j % 3) * TILE, kbn = L::KRING + ((j + 1) % 3) * TILE;
        const int vb = L::VRING + (j & 1) * TILE, vbn = L::VRING + ((j & 1) ^ 1) * TILE;
        if constexpr (KIND == 0) {
            unit(U0, F, F, T, kb, kb, vb, vb, j);
            unit(U1, F, F, T, kb, kbn, vb, vbn, j + 1);
        } else if constexpr (KIND == 1) {
            unit(U0, T, T, T, kb, kb, vb, vb, j);
            unit(U1, T, T, T, kb, kbn, vb, vbn, j + 1);
        } else {
            unit(U0, T, T, T, kb, kb, vb, vb, j);
            unit(U1, T, F, F, kb, kbn, vb, vbn, j + 1);
        }
        ++j;
    };
    if (jl >= 0) {
        {   // what a unit before the first would have left: scores of unit (0, 0), the first part of their
            // exponentials (masked when tile 0 is the diagonal tile), the first V fragments
            const int aK = rrow + L::KRING;
            static_for<KS>([&](auto kc) {
                constexpr int ks = decltype(kc)::value;
                const bf16x8 f = lds_read_b128(smem, aK ^ (32 * ks));
                if constexpr (ks == 0) { mfma_vq0<QACC + ks * 4>(sc[0], f); mfma_vq0<QACC + (KS + ks) * 4>(sc[1], f); }
                else { mfma_vq<QACC + ks * 4>(sc[0], f); mfma_vq<QACC + (KS + ks) * 4>(sc[1], f); }
            });
            set_thr(0, 0);
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
            SCHED_FENCE();
            if (jl == 0) static_for<N1A>([&](auto nc) { v1_step(nc, T); });
            else static_for<N1A>([&](auto nc) { v1_step(nc, F); });
            static_for<2>([&](auto ic) { read_v_head(ic, rrow + L::VRING, 0); });
            SCHED_FENCE();
        }
        const std::integral_constant<int, 0> PLAIN;
        const std::integral_constant<int, 1> MASKED;
        const std::integral_constant<int, 2> LAST;
        while (j < jl - 1) step(PLAIN);
        if (j < jl) step(MASKED);
        step(LAST);
    }
    while (j < J) { sync(); ++j; }

    // ---- epilogue: dQ^T leaves through LDS (8 KiB of this wave's own, one 32-row block at a time) as whole rows
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    char* obuf = smem + L::STG + wave * 8192;
    const int le = fresh_lane(), he = le >> 5;
    static_for<2>([&](auto bc) {
        constexpr int blk = decltype(bc)::value;
        static_for<DB * 4>([&](auto ic) {
            constexpr int db = decltype(ic)::value / 4, g4 = decltype(ic)::value % 4;
            bf16x4 ov;
            static_for<4>([&](auto ec) {
                constexpr int e = decltype(ec)::value;
                ov[e] = (bf16)(acc_read<DQACC + (blk * DB + db) * 16 + 4 * g4 + e>() * a.scale);
            });
            *reinterpret_cast<bf16x4*>(obuf + img256_off(le & 31, 4 * (db0 + db) + g4) + 8 * he) = ov;
        });
        constexpr int CPR = HD / 8, RPI = 64 / CPR;
        const int c = le % CPR, r0 = le / CPR;
#pragma unroll
        for (int i = 0; i < 32 / RPI; ++i) {
            const int row = i * RPI + r0, q = qw0 + 32 * blk + row;
            const bf16x8 v = lds_read_b128(obuf, img256_off(row, 4 * db0 + c));
            if (q < S) *reinterpret_cast<bf16x8*>(a.dQ + (tok0 + q) * a.sdq + (long)h * HD + 8 * c) = v;
        }
    });
    __syncthreads();   // the rings are free for the next pass
    }   // pass
}

int check_common(const AttnArgs& a, int head_dim, const char* who) {
    MGEN_REQUIRE(head_dim == 128 || head_dim == 64, "%s: head_dim must be 64 or 128 (got %d)", who, head_dim);
    MGEN_REQUIRE(a.B > 0 && a.S > 0 && a.H > 0 && a.KVH > 0 && a.H % a.KVH == 0, "%s: bad sizes", who);
    if (head_dim == 64)
        MGEN_REQUIRE(a.H == a.KVH && a.H % 2 == 0,
                     "%s: head_dim 64 runs heads in pairs: needs an even head count and H == KVH", who);
    MGEN_REQUIRE(a.sq % 8 == 0 && a.sk % 8 == 0 && a.sv % 8 == 0, "%s: token strides must be multiples of 8", who);
    return MGEN_OK;
}

template <typename K>
int set_lds(K kernel, int bytes, const char* who) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        mgen_set_error("%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e));
        return MGEN_ELAUNCH;
    }
    return MGEN_OK;
}

int set_drop(AttnArgs& a, float p, unsigned long long key, const char* who) {
    MGEN_REQUIRE(p >= 0.f && p < 1.f, "%s: dropout probability must be in [0, 1) (got %g)", who, p);
    a.drop_key = key;
    a.drop_thresh = drop_threshold(p);
    a.drop_inv_keep = 1.f / (1.f - p);
    return MGEN_OK;
}

constexpr bool kDrop = MGEN_ATTN_DROP != 0;

int attn_fwd_impl(const void* q, const void* k, const void* v, void* out, float* lse, long sq, long sk, long sv, long so,
                  int B, int S, int H, int KVH, int head_dim, float scale, float p, unsigned long long key,
                  hipStream_t stream, const char* who) {
    MGEN_REQUIRE(q && k && v && out, "%s: null pointer", who);
    AttnArgs a = {};
    a.Q = (const bf16*)q; a.K = (const bf16*)k; a.V = (const bf16*)v; a.Out = (bf16*)out; a.LSE = lse;
    a.sq = sq; a.sk = sk; a.sv = sv; a.so = so; a.B = B; a.S = S; a.H = H; a.KVH = KVH; a.scale = scale;
    if (int e = check_common(a, head_dim, who)) return e;
    if (int e = set_drop(a, p, key, who)) return e;
    const dim3 block(256);
    a.nqb = head_dim == 128 ? (S + 255) / 256 : (S + 127) / 128;
    a.gx = (a.nqb + 1) / 2;
    a.gy = head_dim == 128 ? H : H / 2;
    a.gz = B;
    MGEN_REQUIRE((long)a.gx * a.gy * a.gz < (1L << 31), "%s: grid too large", who);
    MGEN_REQUIRE((long)S * sk < (1L << 30) && (long)S * sv < (1L << 30), "%s: one sequence's K or V exceeds 2 GiB", who);
    const dim3 grid((unsigned)(a.gx * a.gy * a.gz));
    MGEN_REQUIRE(so % 8 == 0, "%s: output token stride must be a multiple of 8", who);
    MGEN_REQUIRE((long)S * sq < (1L << 30), "%s: one sequence's Q exceeds 2 GiB", who);
    // more than the default 64 KiB of dynamic LDS has to be asked for, once per kernel and device
    static bool asked_fwd[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (!asked_fwd[dev]) {
        if (int e = set_lds(attn_fwd_kernel<128, kDrop>, 10 * TILE, who)) return e;
        if (int e = set_lds(attn_fwd_kernel<64, kDrop>, 8 * TILE, who)) return e;
        asked_fwd[dev] = true;
    }
    if (head_dim == 128) hipLaunchKernelGGL((attn_fwd_kernel<128, kDrop>), grid, block, 10 * TILE, stream, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<64, kDrop>), grid, block, 8 * TILE, stream, a);
    return mgen_check_launch(who);
}

int attn_bwd_impl(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
                  float* delta, void* dq, void* dk, void* dv, long sq, long sk, long sv, long so, long sdo, long sdq,
                  long sdk, long sdv, int B, int S, int H, int KVH, int head_dim, float scale, float p,
                  unsigned long long key, hipStream_t stream, const char* who) {
    MGEN_REQUIRE(q && k && v && out && dout && lse && delta && dq && dk && dv, "%s: null pointer", who);
    AttnArgs a = {};
    a.Q = (const bf16*)q; a.K = (const bf16*)k; a.V = (const bf16*)v; a.O = (const bf16*)out; a.dO = (const bf16*)dout;
    a.dQ = (bf16*)dq; a.dK = (bf16*)dk; a.dV = (bf16*)dv; a.LSE = (float*)lse; a.Delta = delta;
    a.sq = sq; a.sk = sk; a.sv = sv; a.so = so; a.sdo = sdo; a.sdq = sdq; a.sdk = sdk; a.sdv = sdv;
    a.B = B; a.S = S; a.H = H; a.KVH = KVH; a.scale = scale;
    if (int e = check_common(a, head_dim, who)) return e;
    if (int e = set_drop(a, p, key, who)) return e;
    MGEN_REQUIRE(so % 8 == 0 && sdo % 8 == 0 && sdq % 4 == 0 && sdk % 4 == 0 && sdv % 4 == 0,
                 "%s: bad token stride alignment", who);
    // more than the default 64 KiB of dynamic LDS has to be asked for, once per kernel and device
    static bool asked[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    bool& attr_set = asked[dev];
    if (!attr_set) {
        if (int e = set_lds(attn_bwd_dkv_kernel<128, kDrop>, 2 * DKV_BUF, who)) return e;
        if (int e = set_lds(attn_bwd_dkv_kernel<64, kDrop>, 2 * DKV_BUF, who)) return e;
        attr_set = true;
    }
    const long nrows = (long)B * S * H;
    const int rpb = 256 / (head_dim / 8);
    long nb = (nrows + rpb - 1) / rpb;
    if (nb > 8192) nb = 8192;
    const dim3 block(256);
    const int nrb = head_dim == 128 ? (S + 127) / 128 : (S + 63) / 64;   // row blocks
    MGEN_REQUIRE((long)nrb * H * B < (1L << 31), "%s: grid too large", who);
    AttnArgs akv = a, aq = a;   // the two kernels walk different logical grids
    akv.nqb = nrb; akv.gx = (nrb + 1) / 2; akv.gy = head_dim == 128 ? KVH : KVH / 2; akv.gz = B;
    aq.nqb = nrb; aq.gx = (nrb + 1) / 2; aq.gy = head_dim == 128 ? H : H / 2; aq.gz = B;
    MGEN_REQUIRE((long)S * sk < (1L << 30) && (long)S * sv < (1L << 30) && (long)S * sq < (1L << 30) &&
                 (long)S * sdo < (1L << 30), "%s: one sequence's Q, K, V or dO exceeds 2 GiB", who);
    if (!kDrop) {   // 64 keys per wave: key blocks of 256 (head_dim 128) or 128 rows
        akv.nqb = head_dim == 128 ? (S + 255) / 256 : (S + 127) / 128;
        akv.gx = (akv.nqb + 1) / 2;
        aq.nqb = akv.nqb;   // the dQ kernel's query blocks have the same size
        aq.gx = akv.gx;
        static bool asked64[16] = {};
        if (!asked64[dev]) {
            if (int e = set_lds(attn_bwd_dkv64_kernel<128>, Dkv64<128>::LDS_BYTES, who)) return e;
            if (int e = set_lds(attn_bwd_dkv64_kernel<64>, Dkv64<64>::LDS_BYTES, who)) return e;
            if (int e = set_lds(attn_bwd_dq64_kernel<128>, Dq64<128>::LDS_BYTES, who)) return e;
            if (int e = set_lds(attn_bwd_dq64_kernel<64>, Dq64<64>::LDS_BYTES, who)) return e;
            asked64[dev] = true;
        }
    }
    const dim3 gkv((unsigned)(akv.gx * akv.gy * akv.gz)), gq((unsigned)(aq.gx * aq.gy * aq.gz));
    // delta = rowsum(dO * O) holds with dropout too: sum_k P dP = sum_k P~ dP~ = dO . O
    if (head_dim == 128) {
        hipLaunchKernelGGL(attn_delta_kernel<128>, dim3((unsigned)nb), block, 0, stream, a.O, a.dO, so, sdo, delta, B, S, H);
        if (kDrop) hipLaunchKernelGGL((attn_bwd_dkv_kernel<128, kDrop>), gkv, block, 2 * DKV_BUF, stream, akv);
        else hipLaunchKernelGGL(attn_bwd_dkv64_kernel<128>, gkv, block, Dkv64<128>::LDS_BYTES, stream, akv);
        if (kDrop) hipLaunchKernelGGL((attn_bwd_dq_kernel<128, kDrop>), gq, block, 4 * TILE, stream, aq);
        else hipLaunchKernelGGL(attn_bwd_dq64_kernel<128>, gq, block, Dq64<128>::LDS_BYTES, stream, aq);
    } else {
        hipLaunchKernelGGL(attn_delta_kernel<64>, dim3((unsigned)nb), block, 0, stream, a.O, a.dO, so, sdo, delta, B, S, H);
        if (kDrop) hipLaunchKernelGGL((attn_bwd_dkv_kernel<64, kDrop>), gkv, block, 2 * DKV_BUF, stream, akv);
        else hipLaunchKernelGGL(attn_bwd_dkv64_kernel<64>, gkv, block, Dkv64<64>::LDS_BYTES, stream, akv);
        if (kDrop) hipLaunchKernelGGL((attn_bwd_dq_kernel<64, kDrop>), gq, block, 4 * TILE, stream, aq);
        else hipLaunchKernelGGL(attn_bwd_dq64_kernel<64>, gq, block, Dq64<64>::LDS_BYTES, stream, aq);
    }
    return mgen_check_launch(who);
}

}  // namespace

#if !MGEN_ATTN_DROP
extern "C" int mgen_attn_fwd(const void* q, const void* k, const void* v, void* out, float* lse, long sq, long sk,
                             long sv, long so, int B, int S, int H, int KVH, int head_dim, float scale,
                             hipStream_t stream) {
    return attn_fwd_impl(q, k, v, out, lse, sq, sk, sv, so, B, S, H, KVH, head_dim, scale, 0.f, 0ull, stream,
                         "mgen_attn_fwd");
}

extern "C" int mgen_attn_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout,
                             const float* lse, float* delta, void* dq, void* dk, void* dv, long sq, long sk, long sv,
                             long so, long sdo, long sdq, long sdk, long sdv, int B, int S, int H, int KVH,
                             int head_dim, float scale, hipStream_t stream) {
    return attn_bwd_impl(q, k, v, out, dout, lse, delta, dq, dk, dv, sq, sk, sv, so, sdo, sdq, sdk, sdv, B, S, H, KVH,
                         head_dim, scale, 0.f, 0ull, stream, "mgen_attn_bwd");
}

#else
// The same two with dropout on the attention probabilities (probability p in [0, 1), mask from `key`; see AttnArgs).
// The backward call must get the p and key of its forward call.
extern "C" int mgen_attn_fwd_dropout(const void* q, const void* k, const void* v, void* out, float* lse, long sq,
                                     long sk, long sv, long so, int B, int S, int H, int KVH, int head_dim,
                                     float scale, float p, unsigned long long key, hipStream_t stream) {
    return attn_fwd_impl(q, k, v, out, lse, sq, sk, sv, so, B, S, H, KVH, head_dim, scale, p, key, stream,
                         "mgen_attn_fwd_dropout");
}

extern "C" int mgen_attn_bwd_dropout(const void* q, const void* k, const void* v, const void* out, const void* dout,
                                     const float* lse, float* delta, void* dq, void* dk, void* dv, long sq, long sk,
                                     long sv, long so, long sdo, long sdq, long sdk, long sdv, int B, int S, int H,
                                     int KVH, int head_dim, float scale, float p, unsigned long long key,
                                     hipStream_t stream) {
    return attn_bwd_impl(q, k, v, out, dout, lse, delta, dq, dk, dv, sq, sk, sv, so, sdo, sdq, sdk, sdv, B, S, H, KVH,
                         head_dim, scale, p, key, stream, "mgen_attn_bwd_dropout");
}
#endif
