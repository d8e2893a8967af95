// Next-token selection from one row of logits per stream: greedy argmax, or temperature -> top-k -> top-p -> one
// multinomial draw, without sorting the vocabulary. One 1024-thread workgroup per row; every pass streams the row
// (L2 / Infinity-Cache resident after the first) and ends in a wavefront + LDS reduction.
//
// Reference: GenerationMixin._sample (transformers generation/utils.py:2894-2925) with the warpers
// TemperatureLogitsWarper (generation/logits_process.py:301-303: scores / T), TopKLogitsWarper (:590-595: mask
// scores < k-th largest), TopPLogitsWarper (:527-539: ascending sort, softmax, cumsum, remove while
// cumulative <= 1 - top_p, keep >= 1), then softmax and torch.multinomial(probs, 1), which draws
// argmax_i(probs_i / q_i) with q_i ~ Exp(1). The draw is made here from EXPLICIT noise: either a caller-supplied
// q[B][V], or q = -ln(u) from a counter-based Philox4x32-10 stream that mgen_philox_exponential can also write out.
//
// Thresholds instead of a sort: floats are mapped to order-preserving 32-bit keys; the k-th largest key is found by
// a 4-ary search on counts, and the top-p cut by a 4-ary search on the probability mass at or below a key (the
// ascending cumulative sum of the reference, evaluated at a key). Elements with equal scores are kept or dropped
// together (the reference's unstable sort leaves that case unspecified).
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
// Exp(1) sample for element i of row `row` at stream position `offset`
__device__ __forceinline__ float philox_exp(unsigned long long seed, unsigned long long offset, unsigned row, int i) {
    unsigned c[4] = {(unsigned)(i >> 2), row, (unsigned)offset, (unsigned)(offset >> 32)};
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const unsigned w = c[i & 3];
    const float u = ((float)(w >> 8) + 0.5f) * (1.0f / 16777216.0f);  // 24-bit uniform in (0, 1)
    return -logf(u);
}

// ---- block reductions over 1024 threads ------------------------------------------------------------------
struct Red {
    float f[16][3];
    unsigned long long u[16];
};

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
        for (int i = 0; i < 16; ++i) s += r.f[i][k];
        v[k] = s;
    }
}
// max over packed 64-bit (key << 32 | payload) values
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
    for (int i = 1; i < 16; ++i) m = r.u[i] > m ? r.u[i] : m;
    return m;
}

struct SampleArgs {
    const void* logits; long ld; int V;
    float temperature; int top_k; float top_p; int do_sample;
    const float* noise; long ld_noise; unsigned long long seed, offset;
    long* out;
    long* out2; long out2_stride;  // optional second destination: out2[row * out2_stride]
};

template <typename LT>
__global__ __launch_bounds__(NT) void sample_kernel(SampleArgs a) {
    __shared__ Red red;
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
            a.out[row_i] = tok;
            if (a.out2) a.out2[(long)row_i * a.out2_stride] = tok;
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
            const float q = a.noise ? a.noise[(long)row_i * a.ld_noise + i] : philox_exp(a.seed, a.offset, row_i, i);
            const float sc = (__builtin_amdgcn_exp2f((s - smax) * c2) / Zk) / q;  // >= 0: its bits order like the value
            const unsigned long long p = ((unsigned long long)__float_as_uint(sc) << 32) | (unsigned)(~i);
            best = p > best ? p : best;
        }
    }
    best = block_max_u64(best, red);
    if (tid == 0) {
        const long tok = (long)(~(unsigned)best);
        a.out[row_i] = tok;
        if (a.out2) a.out2[(long)row_i * a.out2_stride] = tok;
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

// logits [B][ld] (bf16 or fp32) -> out[B] int64. do_sample = 0: argmax. Otherwise temperature > 0, top_k (0 = off),
// top_p in (0, 1] (1 = off); noise [B][ld_noise] fp32 Exp(1) draws, or null to use Philox(seed, offset).
// out2 (may be null) receives the same ids at out2[b * out2_stride] (a column of the generated-sequence matrix).
extern "C" int mgen_sample(const void* logits, long ld, int B, int V, int is_bf16, int do_sample, float temperature,
                           int top_k, float top_p, const float* noise, long ld_noise, unsigned long long seed,
                           unsigned long long offset, long* out, long* out2, long out2_stride, hipStream_t stream) {
    MGEN_REQUIRE(logits && out && B > 0 && V > 0 && ld >= V, "mgen_sample: bad arguments");
    if (do_sample) {
        MGEN_REQUIRE(temperature > 0.f, "mgen_sample: `temperature` (=%g) has to be a strictly positive float", temperature);
        MGEN_REQUIRE(top_p > 0.f && top_p <= 1.f, "mgen_sample: `top_p` has to be a float > 0 and <= 1, but is %g", top_p);
        MGEN_REQUIRE(top_k >= 0, "mgen_sample: `top_k` has to be a non-negative integer, but is %d", top_k);
        MGEN_REQUIRE(noise == nullptr || ld_noise >= V, "mgen_sample: ld_noise < V");
    }
    SampleArgs a;
    a.logits = logits; a.ld = ld; a.V = V; a.temperature = temperature; a.top_k = top_k; a.top_p = top_p;
    a.do_sample = do_sample; a.noise = noise; a.ld_noise = ld_noise; a.seed = seed; a.offset = offset; a.out = out;
    a.out2 = out2; a.out2_stride = out2_stride;
    if (is_bf16) hipLaunchKernelGGL((sample_kernel<bf16>), dim3(B), dim3(NT), 0, stream, a);
    else hipLaunchKernelGGL((sample_kernel<float>), dim3(B), dim3(NT), 0, stream, a);
    return mgen_check_launch("mgen_sample");
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
