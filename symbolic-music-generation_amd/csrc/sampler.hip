// Next-token selection from one row of logits per stream: greedy argmax, or temperature -> top-k -> top-p -> one
// multinomial draw, without sorting the vocabulary.
//
// Reference: GenerationMixin._sample (transformers generation/utils.py:2894-2925) with the warpers
// TemperatureLogitsWarper (generation/logits_process.py:301-303: scores / T), TopKLogitsWarper (:590-595: mask
// scores < k-th largest), TopPLogitsWarper (:527-539: ascending sort, softmax, cumsum, remove while
// cumulative <= 1 - top_p, keep >= 1), then softmax and torch.multinomial(probs, 1), which draws
// argmax_i(probs_i / q_i) with q_i ~ Exp(1). The draw is made here from EXPLICIT noise: either a caller-supplied
// q[B][V], or q = -ln(u) from a counter-based Philox4x32-10 stream that mgen_philox_exponential can also write out.
//
// Thresholds instead of a sort: scores are mapped to order-preserving integer keys; the k-th largest key and the
// top-p cut (the ascending cumulative sum of the reference, evaluated at a key) are found from histograms over the
// keys - counts, and probability mass as fixed-point integers - in the register-resident kernel, and by 4-ary
// searches on counts and masses in the generic one. Elements with equal scores are kept or dropped together (the
// reference's unstable sort leaves that case unspecified).
//
// Two kernels: a register-resident one for bf16 rows of up to 155 648 entries (the row is read from HBM once), and
// a generic one (fp32 logits, longer or unaligned rows) that streams the row from cache in every pass.
#include "common.h"

namespace {

constexpr int NT = 1024;
constexpr float LOG2E = 1.4426950408889634f;

template <typename LT> __device__ __forceinline__ float ld_logit(const LT* row, int i) { return (float)row[i]; }

__device__ __forceinline__ unsigned fkey(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// ---- Philox4x32-10 -----------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
// the 4 words of block `blk` (= elements 4*blk .. 4*blk+3) of row `row` at stream position `offset`
__device__ __forceinline__ void philox4(unsigned long long seed, unsigned long long offset, unsigned row, unsigned blk,
                                        unsigned (&c)[4]) {
    c[0] = blk; c[1] = row; c[2] = (unsigned)offset; c[3] = (unsigned)(offset >> 32);
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ float word_to_exp(unsigned w) {
    const float u = ((float)(w >> 8) + 0.5f) * (1.0f / 16777216.0f);  // 24-bit uniform in (0, 1)
    return -0.6931471805599453f * __builtin_amdgcn_logf(u);           // hardware log2 (u is a normal number)
}
// Exp(1) sample for element i of row `row` at stream position `offset`
__device__ __forceinline__ float philox_exp(unsigned long long seed, unsigned long long offset, unsigned row, int i) {
    unsigned c[4];
    philox4(seed, offset, row, (unsigned)(i >> 2), c);
    return word_to_exp(c[i & 3]);
}

// ---- block reductions over NW wavefronts --------------------------------------------------------------------
struct Red {
    float f[16][3];
    unsigned long long u[16];
};

template <int NW = 16>
__device__ __forceinline__ void block_sum3(float (&v)[3], Red& r) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = wave_sum(v[k]);
    __syncthreads();
    if (lane == 0)
        for (int k = 0; k < 3; ++k) r.f[w][k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float s = 0.f;
        for (int i = 0; i < NW; ++i) s += r.f[i][k];
        v[k] = s;
    }
}
// max over packed 64-bit (key << 32 | payload) values
template <int NW = 16>
__device__ __forceinline__ unsigned long long block_max_u64(unsigned long long v, Red& r) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned long long t = __shfl_xor(v, o, 64);
        v = t > v ? t : v;
    }
    __syncthreads();
    if (lane == 0) r.u[w] = v;
    __syncthreads();
    unsigned long long m = r.u[0];
    for (int i = 1; i < NW; ++i) m = r.u[i] > m ? r.u[i] : m;
    return m;
}

struct SampleArgs {
    const void* logits; long ld; int V;
    float temperature; int top_k; float top_p; int do_sample;
    const float* noise; long ld_noise; unsigned long long seed, offset;
    long* out;
    long* out2; long out2_stride;  // optional second destination: out2[row * out2_stride]
    unsigned row0;                 // Philox row index of logits row 0 (a call on rows [row0, row0 + B) of a larger batch)
    // decode loops whose launch arguments must not change from step to step (graph replay):
    const int* step;               // device counter s (or null = 0): Philox offset + s, out2 column + s
    long* unfinished;              // [B] 1 = stream still generating (or null): finished streams receive pad_id, a stream
    long eos_id, pad_id;           //     that draws eos_id becomes finished (GenerationMixin._sample's bookkeeping)
};

// the step counter folded into the arguments (every thread, at the start of a kernel)
__device__ __forceinline__ void sample_begin(SampleArgs& a) {
    if (a.step) {
        const int s = *a.step;
        a.offset += (unsigned long long)s;
        if (a.out2) a.out2 += s;
    }
}
// one thread per row: EOS / pad bookkeeping and the stores
__device__ __forceinline__ void sample_emit(const SampleArgs& a, int row_i, long tok) {
    if (a.unfinished) {
        if (a.unfinished[row_i] == 0) tok = a.pad_id;
        else if (tok == a.eos_id) a.unfinished[row_i] = 0;
    }
    a.out[row_i] = tok;
    if (a.out2) a.out2[(long)row_i * a.out2_stride] = tok;
}

// ---- generic kernel: one 1024-thread workgroup per row, every pass streams the row from cache ----------------
template <typename LT>
__global__ __launch_bounds__(NT) void sample_kernel(SampleArgs a) {
    __shared__ Red red;
    sample_begin(a);
    const int row_i = blockIdx.x, tid = threadIdx.x, V = a.V;
    const LT* row = reinterpret_cast<const LT*>(a.logits) + (long)row_i * a.ld;

    if (!a.do_sample) {
        // argmax of the raw logits, lowest index on ties: pack (key, ~index)
        unsigned long long best = 0;
        for (int i = tid; i < V; i += NT) {
            const unsigned long long p = ((unsigned long long)fkey(ld_logit(row, i)) << 32) | (unsigned)(~i);
            best = p > best ? p : best;
        }
        best = block_max_u64(best, red);
        if (tid == 0) {
            const long tok = (long)(~(unsigned)best);
            sample_emit(a, row_i, tok);
        }
        return;
    }

    const float T = a.temperature;
    // pass 1: key range
    unsigned kmax = 0, kmin = 0xFFFFFFFFu;
    for (int i = tid; i < V; i += NT) {
        const unsigned k = fkey(ld_logit(row, i) / T);
        kmax = max(kmax, k);
        kmin = min(kmin, k);
    }
    kmax = (unsigned)(block_max_u64((unsigned long long)kmax << 32, red) >> 32);
    kmin = ~(unsigned)(block_max_u64((unsigned long long)(~kmin) << 32, red) >> 32);
    // the largest score as a float (keys are invertible)
    const unsigned ub = (kmax & 0x80000000u) ? (kmax & 0x7FFFFFFFu) : ~kmax;
    const float smax = __uint_as_float(ub);

    // ---- top-k: largest key x with count{key >= x} >= k  <=> (smallest x with count{key >= x} < k) - 1
    unsigned kcut = kmin;
    if (a.top_k > 0 && a.top_k < V) {
        unsigned long long lo = kmin, hi = (unsigned long long)kmax + 1;  // answer in [lo, hi]; G(hi) true by definition
        const float kf = (float)a.top_k;
        while (lo < hi) {
            const unsigned long long span = hi - lo;
            const unsigned long long m1 = lo + span / 4, m2 = lo + span / 2, m3 = lo + span / 2 + span / 4;
            float c[3] = {0.f, 0.f, 0.f};
            for (int i = tid; i < V; i += NT) {
                const unsigned long long k = fkey(ld_logit(row, i) / T);
                c[0] += k >= m1 ? 1.f : 0.f;
                c[1] += k >= m2 ? 1.f : 0.f;
                c[2] += k >= m3 ? 1.f : 0.f;
            }
            block_sum3(c, red);  // counts < 2^24 are exact in fp32
            if (c[0] < kf) hi = m1;
            else if (c[1] < kf) { lo = m1 + 1; hi = m2; }
            else if (c[2] < kf) { lo = m2 + 1; hi = m3; }
            else lo = m3 + 1;
        }
        kcut = (unsigned)(lo - 1);
    }

    // ---- normaliser over the top-k survivors
    float z[3] = {0.f, 0.f, 0.f};
    const float c2 = LOG2E;
    for (int i = tid; i < V; i += NT) {
        const float s = ld_logit(row, i) / T;
        if (fkey(s) >= kcut) z[0] += __builtin_amdgcn_exp2f((s - smax) * c2);
    }
    block_sum3(z, red);
    const float Z = z[0];

    // ---- top-p: smallest key x with mass{kcut <= key <= x} > (1 - top_p) * Z ; keep key >= x
    if (a.top_p < 1.0f) {
        const float thr = (1.0f - a.top_p) * Z;
        unsigned long long lo = kcut, hi = kmax;  // F(kmax) = Z > thr
        while (lo < hi) {
            const unsigned long long span = hi - lo;
            const unsigned long long m1 = lo + span / 4, m2 = lo + span / 2, m3 = lo + span / 2 + span / 4;
            float f[3] = {0.f, 0.f, 0.f};
            for (int i = tid; i < V; i += NT) {
                const float s = ld_logit(row, i) / T;
                const unsigned long long k = fkey(s);
                if (k >= kcut && k <= m3) {
                    const float e = __builtin_amdgcn_exp2f((s - smax) * c2);
                    f[2] += e;
                    if (k <= m2) f[1] += e;
                    if (k <= m1) f[0] += e;
                }
            }
            block_sum3(f, red);
            if (f[0] > thr) hi = m1;
            else if (f[1] > thr) { lo = m1 + 1; hi = m2; }
            else if (f[2] > thr) { lo = m2 + 1; hi = m3; }
            else lo = m3 + 1;
        }
        kcut = (unsigned)lo;
    }

    // ---- normaliser over the final survivors, then the draw: argmax_i (e_i / Zk) / q_i
    float zk[3] = {0.f, 0.f, 0.f};
    for (int i = tid; i < V; i += NT) {
        const float s = ld_logit(row, i) / T;
        if (fkey(s) >= kcut) zk[0] += __builtin_amdgcn_exp2f((s - smax) * c2);
    }
    block_sum3(zk, red);
    const float Zk = zk[0];
    unsigned long long best = 0;
    for (int i = tid; i < V; i += NT) {
        const float s = ld_logit(row, i) / T;
        if (fkey(s) >= kcut) {
            const float q = a.noise ? a.noise[(long)row_i * a.ld_noise + i] : philox_exp(a.seed, a.offset, a.row0 + row_i, i);
            const float sc = (__builtin_amdgcn_exp2f((s - smax) * c2) / Zk) / q;  // >= 0: its bits order like the value
            const unsigned long long p = ((unsigned long long)__float_as_uint(sc) << 32) | (unsigned)(~i);
            best = p > best ? p : best;
        }
    }
    best = block_max_u64(best, red);
    if (tid == 0) {
        const long tok = (long)(~(unsigned)best);
        sample_emit(a, row_i, tok);
    }
}

// ---- register-resident kernel for bf16 rows -------------------------------------------------------------------
// A bf16 row of up to NCH * FNT * 8 logits (155 648 >= the 151 936-entry vocabulary) is read from HBM once, 16 bytes
// per lane per chunk, converted to 16-bit order-preserving keys with packed 16-bit arithmetic, and stays in 76 VGPRs
// per thread (1024 threads: 4 wavefronts per SIMD, 128 VGPRs each; with 512 threads and 152 key registers a launch of
// 128 rows took 158 us against 129); every later pass works on registers. bf16 has only 2^16 distinct values and the
// order of logit / T equals the order of the logit for T > 0, so thresholds are searched over those keys. Counts are
// taken wave-wide on the scalar unit (ballot + popcount). Padding (beyond V) has key 0, below the key of every
// number, and is never counted, summed or drawn.
constexpr int FNT = 1024;
constexpr int NCH = 19;
constexpr int FAST_MAX_V = NCH * FNT * 8;

typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;
typedef __attribute__((ext_vector_type(2))) short s16x2;

__device__ __forceinline__ unsigned keys_of_pair(unsigned raw) {
    const u16x2 b = __builtin_bit_cast(u16x2, raw);
    const s16x2 sgn = __builtin_bit_cast(s16x2, raw) >> (short)15;                 // 0xFFFF for negatives
    const u16x2 flip = __builtin_bit_cast(u16x2, sgn) | (unsigned short)0x8000;    // negatives: ~b, others: b | 0x8000
    return __builtin_bit_cast(unsigned, (u16x2)(b ^ flip));
}
// the bf16 value behind a key, as a float
__device__ __forceinline__ float key_val(unsigned k) {
    const unsigned b = (k & 0x8000u) ? (k & 0x7FFFu) : (~k & 0xFFFFu);
    return __uint_as_float(b << 16);
}

// (every pass xors the packed keys with a zero the compiler cannot see through: without it the 304 extracted keys
// are hoisted out of the search loops as loop invariants and spill)
// CHUNK_END runs after the 8 keys of a chunk: accumulating passes pin their partial sums there with an empty asm so
// that the compiler does not re-associate a 304-term sum into a tree (which keeps every term alive and spills).
#define MGEN_FOR_EACH_KEY2(BODY, CHUNK_END)                                            \
    unsigned opaque_zero;                                                              \
    asm volatile("s_mov_b32 %0, 0" : "=s"(opaque_zero));                               \
    _Pragma("unroll") for (int c = 0; c < NCH; ++c) {                                  \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                \
            const unsigned dw = pk[c][e >> 1] ^ opaque_zero;                           \
            const unsigned k = (e & 1) ? (dw >> 16) : (dw & 0xFFFFu);                  \
            BODY                                                                       \
        }                                                                              \
        CHUNK_END                                                                      \
    }
#define MGEN_FOR_EACH_KEY(BODY) MGEN_FOR_EACH_KEY2(BODY, {})

constexpr int SAMPLE_HB = 8192;          // histogram bins of the top-p search (64 KiB of dynamic LDS)

__global__ __launch_bounds__(FNT) void sample_bf16_kernel(SampleArgs a) {
    constexpr int NW = FNT / 64;
    __shared__ Red red;
    extern __shared__ __attribute__((aligned(16))) unsigned long long hist[];
    sample_begin(a);
    const int row_i = blockIdx.x, tid = threadIdx.x, V = a.V;
    const unsigned short* row = reinterpret_cast<const unsigned short*>(a.logits) + (long)row_i * a.ld;

    u32x4 pk[NCH];
    unsigned kmax = 0, kmin1 = 0xFFFFu;              // kmin1 = min over (key - 1) of the real entries
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i0 = (c * FNT + tid) * 8;
        u32x4 kk = {0u, 0u, 0u, 0u};
        if (i0 < V) {                                 // ld is a multiple of 8 and >= V: the 16 bytes are readable
            const u32x4 raw = *reinterpret_cast<const u32x4*>(row + i0);
#pragma unroll
            for (int j = 0; j < 4; ++j) kk[j] = keys_of_pair(raw[j]);
            if (i0 + 8 > V) {                         // the one lane that straddles V
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (i0 + e >= V) kk[e >> 1] &= (e & 1) ? 0x0000FFFFu : 0xFFFF0000u;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned klo = kk[j] & 0xFFFFu, khi = kk[j] >> 16;
            kmax = max(kmax, max(klo, khi));
            kmin1 = min(kmin1, min(klo - 1u, khi - 1u));   // padding (key 0) wraps to 0xFFFFFFFF
        }
        pk[c] = kk;
    }

    if (!a.do_sample) {
        unsigned long long best = 0;
        MGEN_FOR_EACH_KEY({
            const int i = (c * FNT + tid) * 8 + e;
            const unsigned long long p = ((unsigned long long)k << 32) | (unsigned)(~i);
            best = p > best ? p : best;
        })
        best = block_max_u64<NW>(best, red);
        if (tid == 0) {
            const long tok = (long)(~(unsigned)best);
            sample_emit(a, row_i, tok);
        }
        return;
    }

    kmax =(unsigned)(block_max_u64<NW>((unsigned long long)kmax << 32, red) >> 32);
    const unsigned kmin = 0x10000u - (unsigned)(block_max_u64<NW>((unsigned long long)(0xFFFFu - kmin1) << 32, red) >> 32);
    const float T = a.temperature;
    // score = logit / T, correctly rounded, from a refined reciprocal and one residual step (3 FMAs per element)
    const bool unit_T = T == 1.0f;
    float rT = __builtin_amdgcn_rcpf(T);
    rT = fmaf(fmaf(-T, rT, 1.0f), rT, rT);
    auto score = [&](unsigned k) -> float {
        const float l = key_val(k);
        if (unit_T) return l;
        const float q0 = l * rT;
        const float q1 = fmaf(fmaf(-T, q0, l), rT, q0);
        return (k == 0x007Fu) ? l : q1;              // -inf stays -inf (the residual would be inf - inf)
    };
    const float smax = score(kmax);

    // ---- top-k: the largest key x with count{key >= x} >= k. One pass counts the keys into a histogram in LDS, a scan
    // from the top finds the bin where the running count reaches k; key ranges wider than the histogram use bins of
    // 2^shift keys and a second pass over the keys of the bin that was found.
    unsigned kcut = kmin;
    if (a.top_k > 0 && a.top_k < V) {
        unsigned* cnt = reinterpret_cast<unsigned*>(hist);
        const unsigned want = (unsigned)a.top_k;
        unsigned above = 0;                              // count{key >= base + nkeys}
        unsigned base = kmin, nkeys = kmax - kmin + 1;   // the answer lies in [base, base + nkeys)
        for (;;) {
            int sh = 0;
            while (((nkeys - 1) >> sh) >= (unsigned)SAMPLE_HB) ++sh;
            const int nb = (int)((nkeys - 1) >> sh) + 1;
            for (int i = tid; i < nb; i += FNT) cnt[i] = 0u;
            __syncthreads();
            MGEN_FOR_EACH_KEY({
                const unsigned rel = k - base;           // (padding has key 0 < kmin: wraps far above nkeys)
                if (rel < nkeys) atomicAdd(&cnt[rel >> sh], 1u);
            })
            __syncthreads();
            // thread t owns bins [t * per, t * per + per); counts above a thread's bins = total - inclusive prefix
            const int per = (nb + FNT - 1) / FNT;
            unsigned mine = 0;
            for (int j = 0; j < per; ++j) {
                const int bi = tid * per + j;
                if (bi < nb) mine += cnt[bi];
            }
            unsigned incl = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned up = __shfl_up(incl, o, 64);
                if ((tid & 63) >= o) incl += up;
            }
            if ((tid & 63) == 63) red.u[tid >> 6] = incl;
            __syncthreads();
            unsigned before = 0, total = 0;
            for (int w = 0; w < NW; ++w) {
                const unsigned tw = (unsigned)red.u[w];
                if (w < (tid >> 6)) before += tw;
                total += tw;
            }
            __syncthreads();                             // red.u is free again
            const unsigned over = above + (total - (before + incl));   // count of keys above this thread's bins
            if (over < want && over + mine >= want) {    // exactly one thread (the row has more than k entries)
                unsigned run = over;
                for (int j = per - 1; j >= 0; --j) {
                    const int bi = tid * per + j;
                    const unsigned h = bi < nb ? cnt[bi] : 0u;
                    if (run + h >= want) {
                        red.u[0] = run;
                        red.u[1] = (unsigned long long)bi;
                        break;
                    }
                    run += h;
                }
            }
            __syncthreads();
            above = (unsigned)red.u[0];
            const unsigned bin = (unsigned)red.u[1];
            __syncthreads();
            base += bin << sh;
            if (sh == 0) break;
            nkeys = min(1u << sh, kmax - base + 1);
        }
        kcut = base;
    }

    // ---- normaliser Z of the top-k survivors and the top-p cut: the smallest key x with mass{kcut <= key <= x} >
    // (1 - p) Z. One pass over the row sums Z and adds every survivor's e_i, as a 2^-40 fixed-point integer, into a
    // histogram over the keys in LDS (integer adds: the bins do not depend on the order the lanes arrive in); a scan of
    // the bins finds the one where the running mass passes the threshold. Key ranges wider than the histogram use
    // bins of 2^shift keys and a second pass over the keys of the bin that was found.
    const float c2 = LOG2E;
    const bool want_p = a.top_p < 1.0f;
    float Z = 0.f;
    unsigned long long below_fx = 0, thr_fx = 0;    // mass{kcut <= key < base}, threshold (both x 2^40)
    unsigned base = kcut, nkeys = kmax - kcut + 1;  // the cut lies in [base, base + nkeys)
    for (int level = 0;; ++level) {
        int sh = 0;
        while (((nkeys - 1) >> sh) >= (unsigned)SAMPLE_HB) ++sh;
        const int nb = (int)((nkeys - 1) >> sh) + 1;
        if (want_p) {
            for (int i = tid; i < nb; i += FNT) hist[i] = 0ull;
            __syncthreads();
        }
        float f[3] = {0.f, 0.f, 0.f};
        MGEN_FOR_EACH_KEY({
            const unsigned rel = k - base;           // (padding has key 0 < kcut: wraps far above nkeys)
            const bool in = rel < nkeys;
            if (__builtin_amdgcn_ballot_w64(in) != 0ull) {
                if (in) {
                    const float ee = __builtin_amdgcn_exp2f((score(k) - smax) * c2);
                    f[2] += ee;
                    if (want_p) {
                        // ee in [0, 1] as (hi, lo) = ee * 2^40 split at 2^32: both factors are powers of two (exact)
                        const float e8 = ee * 256.0f;
                        const float hi = __builtin_floorf(e8);
                        const unsigned long long fx =
                            ((unsigned long long)(unsigned)hi << 32) | (unsigned)((e8 - hi) * 4294967296.0f);
                        atomicAdd(&hist[rel >> sh], fx);
                    }
                }
            }
        })
        if (level == 0) {
            block_sum3<NW>(f, red);                  // (its barriers also end the histogram adds)
            Z = f[2];
            if (!want_p) break;
            const double t = (double)((1.0f - a.top_p) * Z) * 1099511627776.0;
            thr_fx = (unsigned long long)t;
        } else {
            __syncthreads();
        }
        // thread t owns bins [t * per, t * per + per): block-exclusive prefix of the owners' sums, then the owner of
        // the crossing walks its bins. The threshold is kept below the total, so a crossing exists (keep >= 1: the
        // largest key always survives).
        const int per = (nb + FNT - 1) / FNT;
        unsigned long long mine = 0;
        for (int j = 0; j < per; ++j) {
            const int bi = tid * per + j;
            if (bi < nb) mine += hist[bi];
        }
        unsigned long long incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long up = __shfl_up(incl, o, 64);
            if ((tid & 63) >= o) incl += up;
        }
        if ((tid & 63) == 63) red.u[tid >> 6] = incl;
        __syncthreads();
        unsigned long long before = 0, total = 0;
        for (int w = 0; w < NW; ++w) {
            const unsigned long long tw = red.u[w];
            if (w < (tid >> 6)) before += tw;
            total += tw;
        }
        __syncthreads();                             // red.u is free again
        if (level == 0 && thr_fx >= total) thr_fx = total - 1;   // (total >= 2^40: the largest key has e = 1)
        const unsigned long long excl = below_fx + before + (incl - mine);
        if (excl <= thr_fx && excl + mine > thr_fx) {   // exactly one thread
            unsigned long long run = excl;
            for (int j = 0; j < per; ++j) {
                const int bi = tid * per + j;
                const unsigned long long h = bi < nb ? hist[bi] : 0ull;
                if (run + h > thr_fx) {
                    red.u[0] = run;
                    red.u[1] = (unsigned long long)bi;
                    break;
                }
                run += h;
            }
        }
        __syncthreads();
        below_fx = red.u[0];
        const unsigned bin = (unsigned)red.u[1];
        __syncthreads();
        base += bin << sh;
        if (sh == 0) break;
        nkeys = min(1u << sh, kmax - base + 1);
    }
    if (want_p) kcut = base;
    const float below = (float)((double)below_fx * (1.0 / 1099511627776.0));
    const float Zk = Z - below;              // mass of the final survivors

    // ---- the draw: argmax over the survivors of (e_i / Zk) / q_i; one Philox block serves 4 neighbouring elements
    unsigned long long best = 0;
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        const u32x4 kk = pk[c];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            unsigned k4[4] = {kk[2 * h] & 0xFFFFu, kk[2 * h] >> 16, kk[2 * h + 1] & 0xFFFFu, kk[2 * h + 1] >> 16};
            const bool any4 = k4[0] >= kcut || k4[1] >= kcut || k4[2] >= kcut || k4[3] >= kcut;
            if (__builtin_amdgcn_ballot_w64(any4) == 0ull) continue;
            if (any4) {
                const int i4 = (c * FNT + tid) * 8 + 4 * h;
                unsigned w[4];
                if (!a.noise) philox4(a.seed, a.offset, a.row0 + row_i, (unsigned)(i4 >> 2), w);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (k4[e] >= kcut) {
                        const float q = a.noise ? a.noise[(long)row_i * a.ld_noise + i4 + e] : word_to_exp(w[e]);
                        const float sc = (__builtin_amdgcn_exp2f((score(k4[e]) - smax) * c2) / Zk) / q;
                        const unsigned long long p =
                            ((unsigned long long)__float_as_uint(sc) << 32) | (unsigned)(~(i4 + e));
                        best = p > best ? p : best;
                    }
                }
            }
        }
    }
    best = block_max_u64<NW>(best, red);
    if (tid == 0) {
        const long tok = (long)(~(unsigned)best);
        sample_emit(a, row_i, tok);
    }
}

__global__ void philox_exp_kernel(float* __restrict__ out, long ld, int B, int V, unsigned long long seed,
                                  unsigned long long offset) {
    const long n = (long)B * V;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
        const int r = (int)(idx / V), i = (int)(idx % V);
        out[(long)r * ld + i] = philox_exp(seed, offset, r, i);
    }
}

}  // namespace

// mgen_sample_rows for a decode loop: step (device int, may be null) is added to the Philox offset and to the out2
// column at run time, so the launch arguments are the same at every step; unfinished ([B] int64, may be null) carries
// the EOS bookkeeping of GenerationMixin._sample (generation/utils.py:2919-2931): a finished stream receives pad_id,
// a stream that draws eos_id is finished from the next step on.
extern "C" int mgen_sample_step(const void* logits, long ld, int B, int V, int is_bf16, int do_sample,
                                float temperature, int top_k, float top_p, const float* noise, long ld_noise,
                                unsigned long long seed, unsigned long long offset, int row0, long* out, long* out2,
                                long out2_stride, const int* step, long* unfinished, long eos_id, long pad_id,
                                hipStream_t stream) {
    MGEN_REQUIRE(logits && out && B > 0 && V > 0 && ld >= V && row0 >= 0, "mgen_sample: bad arguments");
    if (do_sample) {
        MGEN_REQUIRE(temperature > 0.f, "mgen_sample: `temperature` (=%g) has to be a strictly positive float", temperature);
        MGEN_REQUIRE(top_p > 0.f && top_p <= 1.f, "mgen_sample: `top_p` has to be a float > 0 and <= 1, but is %g", top_p);
        MGEN_REQUIRE(top_k >= 0, "mgen_sample: `top_k` has to be a non-negative integer, but is %d", top_k);
        MGEN_REQUIRE(noise == nullptr || ld_noise >= V, "mgen_sample: ld_noise < V");
    }
    SampleArgs a;
    a.logits = logits; a.ld = ld; a.V = V; a.temperature = temperature; a.top_k = top_k; a.top_p = top_p;
    a.do_sample = do_sample; a.noise = noise; a.ld_noise = ld_noise; a.seed = seed; a.offset = offset; a.out = out;
    a.out2 = out2; a.out2_stride = out2_stride; a.row0 = (unsigned)row0;
    a.step = step; a.unfinished = unfinished; a.eos_id = eos_id; a.pad_id = pad_id;
    const bool fast = is_bf16 && V <= FAST_MAX_V && ld % 8 == 0 && ((uintptr_t)logits % 16) == 0;
    if (fast) {
        constexpr int lds = SAMPLE_HB * 8;
        static bool asked[16] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
        if (!asked[dev]) {   // with the static part more than the default 64 KiB of LDS: asked for once per device
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sample_bf16_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) {
                mgen_set_error("mgen_sample: cannot reserve %d bytes of LDS: %s", lds, hipGetErrorString(e));
                return MGEN_ELAUNCH;
            }
            asked[dev] = true;
        }
        hipLaunchKernelGGL(sample_bf16_kernel, dim3(B), dim3(FNT), lds, stream, a);
    }
    else if (is_bf16) hipLaunchKernelGGL((sample_kernel<bf16>), dim3(B), dim3(NT), 0, stream, a);
    else hipLaunchKernelGGL((sample_kernel<float>), dim3(B), dim3(NT), 0, stream, a);
    return mgen_check_launch("mgen_sample");
}

// logits [B][ld] (bf16 or fp32) -> out[B] int64. do_sample = 0: argmax. Otherwise temperature > 0, top_k (0 = off),
// top_p in (0, 1] (1 = off); noise [B][ld_noise] fp32 Exp(1) draws, or null to use Philox(seed, offset).
// out2 (may be null) receives the same ids at out2[b * out2_stride] (a column of the generated-sequence matrix).
// row0: the rows are rows [row0, row0 + B) of a larger batch; the Philox stream of row i is that of row row0 + i of
// one call on the whole batch (explicit noise is indexed from the pointer given).
extern "C" int mgen_sample_rows(const void* logits, long ld, int B, int V, int is_bf16, int do_sample,
                                float temperature, int top_k, float top_p, const float* noise, long ld_noise,
                                unsigned long long seed, unsigned long long offset, int row0, long* out, long* out2,
                                long out2_stride, hipStream_t stream) {
    return mgen_sample_step(logits, ld, B, V, is_bf16, do_sample, temperature, top_k, top_p, noise, ld_noise, seed,
                            offset, row0, out, out2, out2_stride, nullptr, nullptr, 0, 0, stream);
}

extern "C" int mgen_sample(const void* logits, long ld, int B, int V, int is_bf16, int do_sample, float temperature,
                           int top_k, float top_p, const float* noise, long ld_noise, unsigned long long seed,
                           unsigned long long offset, long* out, long* out2, long out2_stride, hipStream_t stream) {
    return mgen_sample_rows(logits, ld, B, V, is_bf16, do_sample, temperature, top_k, top_p, noise, ld_noise, seed,
                            offset, 0, out, out2, out2_stride, stream);
}

// writes the Exp(1) draws mgen_sample would use for (seed, offset): out [B][ld]
extern "C" int mgen_philox_exponential(float* out, long ld, int B, int V, unsigned long long seed,
                                       unsigned long long offset, hipStream_t stream) {
    MGEN_REQUIRE(out && B > 0 && V > 0 && ld >= V, "mgen_philox_exponential: bad arguments");
    long nb = ((long)B * V + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(philox_exp_kernel, dim3((unsigned)nb), dim3(256), 0, stream, out, ld, B, V, seed, offset);
    return mgen_check_launch("mgen_philox_exponential");
}
