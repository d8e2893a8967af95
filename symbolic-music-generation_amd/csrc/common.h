// Shared device helpers for the gfx950 (MI355X, CDNA4) kernels: vector types, wave64 reductions,
// bf16 packing, and the LDS tile images used by every MFMA kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// Streaming accesses of the element-wise kernels: data that is read once or written for a later kernel carries the
// non-temporal hint, so it does not displace what other kernels keep in L2. MGEN_STREAM_NT=0 builds plain accesses
// (same-machine comparisons).
#ifndef MGEN_STREAM_NT
#define MGEN_STREAM_NT 1
#endif
template <typename V> __device__ __forceinline__ V ld_stream(const V* p) {
#if MGEN_STREAM_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
template <typename V> __device__ __forceinline__ void st_stream(V* p, V v) {
#if MGEN_STREAM_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

#define MGEN_WAVE 64

// error codes returned by the C-ABI
#define MGEN_OK 0
#define MGEN_EINVAL (-1)
#define MGEN_ELAUNCH (-2)

void mgen_set_error(const char* fmt, ...);
int mgen_check_launch(const char* what);

#define MGEN_REQUIRE(cond, ...)              \
    do {                                     \
        if (!(cond)) {                       \
            mgen_set_error(__VA_ARGS__);     \
            return MGEN_EINVAL;              \
        }                                    \
    } while (0)

// ---------------------------------------------------------------------------------------------------
// wave64 reductions (all 64 lanes end with the result)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// reduction over an aligned group of W consecutive lanes (W a power of two <= 64)
template <int W>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = W / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum for blockDim.x == 256 (4 waves); `red` is a __shared__ float[4+]
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max_256(float v, float* red) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ---------------------------------------------------------------------------------------------------
// Dropout masks: a stateless 32-bit hash of (key, element index) compared with a threshold; an element is KEPT
// iff hash >= thresh, thresh = floor(p * 2^32). Backward passes recompute the mask from the same key, nothing is
// stored. Two index forms: flat (any tensor, 64-bit element index) and attention (batch*heads + head, query, key).
// ---------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ unsigned drop_fmix32(unsigned h) {
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}
__host__ __device__ __forceinline__ unsigned drop_hash_flat(unsigned long long key, unsigned long long idx) {
    const unsigned r = drop_fmix32((unsigned)idx ^ (unsigned)key);
    return drop_fmix32(r + (unsigned)(key >> 32) + (unsigned)(idx >> 32) * 0x9E3779B1u);
}
// per (head, query) part of the attention hash; the element hash is drop_fmix32(rowkey ^ key_index)
__host__ __device__ __forceinline__ unsigned drop_rowkey(unsigned long long key, unsigned bh, unsigned q) {
    return drop_fmix32((bh * 0x9E3779B1u + q) ^ (unsigned)key) + (unsigned)(key >> 32);
}
__host__ __forceinline__ unsigned drop_threshold(float p) {
    const double t = (double)p * 4294967296.0;
    return t >= 4294967295.0 ? 0xFFFFFFFFu : (unsigned)t;
}

// ---------------------------------------------------------------------------------------------------
// bf16 helpers
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }  // round-to-nearest-even, NaN kept
__device__ __forceinline__ float round_bf(float x) { return (float)((bf16)x); }

// ---------------------------------------------------------------------------------------------------
// LDS tile images.
//
// IMG256: a tile of R rows x 128 bf16 (256-byte rows). 16-byte chunk `ch` (0..15) of row `row` lives at
//   256*row + 16*(ch ^ swz256(row)),  swz256(row) = ((row&3)<<2) | ((row>>2)&3).
// One image serves both the row read (ds_read_b128: a lane takes 8 consecutive columns of its row) and the
// transposed read (ds_read_b64_tr_b16: a lane ends up with 4 consecutive ROWS of its column) without bank
// conflicts for the 32x32x16 MFMA operand shapes.
//
// IMG128: a tile of R rows x 64 bf16 (128-byte rows), row reads only:
//   128*row + 16*(ch ^ ((row>>1)&7)), ch in 0..7.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int swz256(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int img256_off(int row, int ch) { return 256 * row + 16 * (ch ^ swz256(row)); }
__device__ __forceinline__ int img128_off(int row, int ch) { return 128 * row + 16 * (ch ^ ((row >> 1) & 7)); }

__device__ __forceinline__ bf16x8 lds_read_b128(const char* lds, int byte_off) {
    return *reinterpret_cast<const bf16x8*>(lds + byte_off);
}
__device__ __forceinline__ void lds_write_b128(char* lds, int byte_off, u32x4 v) {
    *reinterpret_cast<u32x4*>(lds + byte_off) = v;
}

// Transposed 4x16 block read: within each group of 16 consecutive lanes, lane 4q+p passes the address of row q,
// columns 4p..4p+3 of a 4-row x 16-column block of 16-bit elements; lane i of the group receives column i of
// the 4 rows (row q in element q). All 64 lanes must be active.
__device__ __forceinline__ s16x4 lds_read_tr16(const char* lds, int byte_off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3)))*)(lds + byte_off));
}

// Operand fragment for v_mfma_f32_32x32x16_bf16 taken by ROW from an IMG256 tile:
// lane (r = lane&31, h = lane>>5) gets tile[row0 + r][16*ks + 8h .. +7].
__device__ __forceinline__ bf16x8 frag_row_img256(const char* tile, int row0, int ks, int lane) {
    const int row = row0 + (lane & 31);
    return lds_read_b128(tile, img256_off(row, 2 * ks + (lane >> 5)));
}
__device__ __forceinline__ bf16x8 frag_row_img128(const char* tile, int row0, int ks, int lane) {
    const int row = row0 + (lane & 31);
    return lds_read_b128(tile, img128_off(row, 2 * ks + (lane >> 5)));
}

// Operand fragment taken by COLUMN from an IMG256 tile: lane (r, h) gets, for column col0 + r, the 8 rows
//   natural order  (PERM = false): krow0 + 8h + j                       (j = 0..7)
//   accumulator order (PERM = true): krow0 + 8(j>>2) + 4h + (j&3)
// The second form matches the k order of a 32x32 fp32 accumulator tile re-used as the other operand
// (registers 8s..8s+7 of the tile, krow0 = 16s).
template <bool PERM>
__device__ __forceinline__ bf16x8 frag_col_img256(const char* tile, int krow0, int col0, int lane) {
    const int g = lane >> 4, i = lane & 15, h = g >> 1;
    const int q = i >> 2, p = i & 3;
    const int col = col0 + 16 * (g & 1) + 4 * p;  // first of this lane's 4 address columns
    const int ch = col >> 3, half = (col >> 2) & 1;
    const int ra = krow0 + (PERM ? 4 * h : 8 * h) + q;
    const int rb = ra + (PERM ? 8 : 4);
    s16x4 lo = lds_read_tr16(tile, img256_off(ra, ch) + 8 * half);
    s16x4 hi = lds_read_tr16(tile, img256_off(rb, ch) + 8 * half);
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, both);
}

// Converts registers [8s, 8s+8) of a 32x32 fp32 accumulator tile into the bf16 fragment of k-step s.
template <int S>
__device__ __forceinline__ bf16x8 acc_to_frag(const f32x16& a) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (bf16)a[8 * S + j];
    return r;
}

// accumulator register -> row index inside a 32x32 tile (column = lane & 31)
__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// LDS byte address (what M0 and the DS instructions take) of a pointer into the workgroup's LDS
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(__UINTPTR_TYPE__)(const __attribute__((address_space(3))) char*)p;
}

// Bijective remap of a 1-D block id so that the blocks that share an XCD (ids congruent mod 8 under round-robin
// dispatch) receive one contiguous range of work items. Speed only; any placement is correct.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}
