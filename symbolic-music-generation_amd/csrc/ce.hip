// Shifted-label cross entropy over a chunk of logits rows, forward and backward in one pass (HBM-bound).
//
// Reference: ForCausalLMLoss, transformers loss/loss_utils.py:49-71 — logits.float(); labels padded right with -100 and
// shifted by one; cross_entropy(mean over labels != ignore_index). Row t = (b, s) is scored against labels[b, s+1]
// (ignored when s is the last position or the label equals ignore_index).
//
// One 256-thread block per row: pass 1 computes the row's max and sum of exponentials with 16-byte loads, pass 2
// overwrites the logits in place with d(loss)/d(logits) = (softmax - onehot) * grad_scale / n_valid in the same dtype
// (what the lm_head dgrad/wgrad GEMMs then read), and the row's loss term goes to row_loss[t] (0 for ignored rows).
// The mean is taken by mgen_ce_finalize in a fixed order, so the loss is bitwise reproducible.
#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

template <typename LT> struct Row8;
template <> struct Row8<bf16> {
    static __device__ __forceinline__ void load(const bf16* p, float (&x)[8]) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = (float)v[e];
    }
    static __device__ __forceinline__ void store(bf16* p, const float (&x)[8]) {
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16)x[e];
        *reinterpret_cast<bf16x8*>(p) = v;
    }
};
template <> struct Row8<float> {
    static __device__ __forceinline__ void load(const float* p, float (&x)[8]) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[e] = a[e]; x[4 + e] = b[e]; }
    }
    static __device__ __forceinline__ void store(float* p, const float (&x)[8]) {
        *reinterpret_cast<f32x4*>(p) = f32x4{x[0], x[1], x[2], x[3]};
        *reinterpret_cast<f32x4*>(p + 4) = f32x4{x[4], x[5], x[6], x[7]};
    }
};

// One 1024-thread block (the count feeds 1/n on the device, so it stays a single block), four independent loads in
// flight per thread.
__global__ __launch_bounds__(1024) void ce_count_kernel(const long* __restrict__ labels, long B, int S,
                                                        long ignore_index, float* __restrict__ inv_count,
                                                        int* __restrict__ count, long V, int* __restrict__ err) {
    __shared__ int red[16];
    int c = 0;
    const long n = B * S;
    for (long i0 = threadIdx.x; i0 < n; i0 += 4096) {
        long lab[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = i0 + 1024 * u;
            lab[u] = (i + 1 < n) ? labels[i + 1] : ignore_index;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = i0 + 1024 * u;
            if (i < n && (int)(i % S) + 1 < S && lab[u] != ignore_index) {
                // the same validity test as the loss kernels: a label outside [0, V) scores nothing and is not counted
                // (the reference asserts on it: bit 2 of the error flag lets the host raise)
                if (V > 0 && (lab[u] < 0 || lab[u] >= V)) {
                    if (err) atomicOr(err, 4);
                } else {
                    ++c;
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < 16; ++w) tot += red[w];
        *inv_count = 1.f / (float)tot;  // n_valid == 0 gives inf and the mean becomes NaN, as in the reference
        if (count) *count = tot;
    }
}

template <typename LT, bool WRITE_GRAD>
__global__ __launch_bounds__(256) void ce_kernel(LT* __restrict__ logits, const long* __restrict__ labels, long tok0,
                                                 int V, long ld, int S, long ignore_index,
                                                 const float* __restrict__ inv_count, float grad_scale,
                                                 float* __restrict__ row_loss, float* __restrict__ lse_out) {
    __shared__ float red[4];
    const long r = blockIdx.x;
    const long t = tok0 + r;
    const int s = (int)(t % S);
    long label = ignore_index;
    if (s + 1 < S) label = labels[t + 1];
    const bool valid = label != ignore_index && label >= 0 && label < V;
    LT* row = logits + r * ld;
    const int nvec = (V + 7) / 8;

    float m = -1.0e30f, sum = 0.f;
    for (int i = threadIdx.x; i < nvec; i += 256) {
        float x[8];
        Row8<LT>::load(row + i * 8, x);
        float mx = m;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            x[e] = (i * 8 + e < V) ? x[e] * LOG2E : -1.0e30f;
            mx = fmaxf(mx, x[e]);
        }
        float acc = sum * __builtin_amdgcn_exp2f(m - mx);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += __builtin_amdgcn_exp2f(x[e] - mx);
        sum = acc;
        m = mx;
    }
    const float M = block_max_256(m, red);
    sum = block_sum_256(sum * __builtin_amdgcn_exp2f(m - M), red);
    const float lse = (M + __log2f(sum)) * LN2;
    if (threadIdx.x == 0) {
        if (lse_out) lse_out[t] = lse;
        row_loss[t] = valid ? (lse - (float)row[label]) : 0.f;
    }
    if (!WRITE_GRAD) return;
    __syncthreads();  // row[label] is read above before anyone overwrites the row
    const float g = valid ? grad_scale * (*inv_count) : 0.f;
    const float inv_sum = 1.f / sum;
    for (int i = threadIdx.x; i < nvec; i += 256) {
        float x[8];
        Row8<LT>::load(row + i * 8, x);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int idx = i * 8 + e;
            float p = __builtin_amdgcn_exp2f(x[e] * LOG2E - M) * inv_sum;
            if (idx == label) p -= 1.f;
            x[e] = idx < V ? p * g : 0.f;
        }
        Row8<LT>::store(row + i * 8, x);
    }
}

// bf16 rows of up to NT * 8 * NV logits with the whole row held in registers (16 bytes x NV per thread, NT
// threads): the logits are read from memory once and the gradient written once, instead of read twice.
// One block per CU holds a row, so the vector work of a row is not hidden behind another row's loads: it is kept
// short. Three passes over the registers: maximum (one v_max3 per pair), sum of exp2(x log2e - M), and the gradient
// exp2(x log2e - M - log2(sum) + log2(g)) with the label's element and the columns past V patched afterwards.
__device__ __forceinline__ float bf_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

template <int NT, int NV, bool WRITE_GRAD>
__global__ __launch_bounds__(NT) void ce_row_kernel(bf16* __restrict__ logits, const long* __restrict__ labels,
                                                      long tok0, long rows, int V, long ld, int S, long ignore_index,
                                                      const float* __restrict__ inv_count, float grad_scale,
                                                      float* __restrict__ row_loss, float* __restrict__ lse_out) {
    __shared__ float red[2][NT / 64];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int nvec = (V + 7) / 8;
    const int vlast = V >> 3, vrem = V & 7;   // vector vlast holds the first column past V (when V is no multiple of 8)

    // A block walks rows blockIdx.x, + gridDim.x, ...; the next row's vectors are requested into the registers the
    // current row's gradient has just left, so a CU's loads travel while it computes and stores.
    u32x4 v[NV];
    auto load_vec = [&](const bf16* row, int j) {
        const int i = tid + j * NT;
        if (i < nvec) v[j] = ld_stream(reinterpret_cast<const u32x4*>(row + (long)i * 8));
        else v[j] = u32x4{0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u};   // -inf: no weight in max or sum
    };
    long r = blockIdx.x;
    if (r < rows) {
#pragma unroll
        for (int j = 0; j < NV; ++j) load_vec(logits + r * ld, j);
    }
    for (; r < rows; r += gridDim.x) {
        const long t = tok0 + r;
        const int s = (int)(t % S);
        long label = ignore_index;
        if (s + 1 < S) label = labels[t + 1];
        const bool valid = label != ignore_index && label >= 0 && label < V;
        bf16* row = logits + r * ld;
        const long rn = r + gridDim.x;
        const bool has_next = rn < rows;
        const bf16* row_next = logits + (has_next ? rn : r) * ld;
        if (vrem) {   // columns past V inside the last vector
#pragma unroll
            for (int j = 0; j < NV; ++j)
                if (tid + j * NT == vlast) {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (e >= vrem) v[j][e >> 1] = (e & 1) ? ((v[j][e >> 1] & 0xffffu) | 0xff800000u)
                                                              : ((v[j][e >> 1] & 0xffff0000u) | 0xff80u);
                }
        }
        float m = -3.0e38f;
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int w = 0; w < 4; ++w) m = __builtin_fmaxf(m, __builtin_fmaxf(bf_lo(v[j][w]), bf_hi(v[j][w])));
        const float wm = wave_max(m);
        __syncthreads();   // the previous row's readers of red[] are done
        if ((tid & 63) == 0) red[0][wave] = wm;
        __syncthreads();
        // (the row's registers pass through an empty asm between the passes: the unpacked values of one pass are not
        // kept for the next, which would need twice the register file)
#pragma unroll
        for (int j = 0; j < NV; ++j) asm volatile("" : "+v"(v[j]));
        float M = red[0][0];
#pragma unroll
        for (int w = 1; w < NT / 64; ++w) M = fmaxf(M, red[0][w]);
        M *= LOG2E;   // the row maximum in the exponent's units (finite: every row has V >= 1 real columns)
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                sum += __builtin_amdgcn_exp2f(__builtin_fmaf(bf_lo(v[j][w]), LOG2E, -M));
                sum += __builtin_amdgcn_exp2f(__builtin_fmaf(bf_hi(v[j][w]), LOG2E, -M));
            }
        const float ws = wave_sum(sum);
        if ((tid & 63) == 0) red[1][wave] = ws;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NV; ++j) asm volatile("" : "+v"(v[j]));
        sum = 0.f;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) sum += red[1][w];
        const float l2s = __log2f(sum);
        const float lse = (M + l2s) * LN2;
        if (tid == 0) {
            if (lse_out) lse_out[t] = lse;
            row_loss[t] = valid ? (lse - (float)row[label]) : 0.f;
        }
        if (!WRITE_GRAD) {
            if (has_next) {
#pragma unroll
                for (int j = 0; j < NV; ++j) load_vec(row_next, j);
            }
            continue;
        }
        __syncthreads();  // row[label] is read above before anyone overwrites the row
        const float g = valid ? grad_scale * (*inv_count) : 0.f;
        // p g = exp2(x log2e - M - log2(sum) + log2(g)); g = 0 (ignored row) gives exp2(-inf) = 0
        const float shift = -M - l2s + __log2f(g);
        const int lvec = valid ? (int)(label >> 3) : -1, le = (int)(label & 7);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = tid + j * NT;
            if (i < nvec) {
                u32x4 o;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(bf_lo(v[j][w]), LOG2E, shift));
                    const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(bf_hi(v[j][w]), LOG2E, shift));
                    const bf16x2 pk = {(bf16)p0, (bf16)p1};
                    o[w] = __builtin_bit_cast(unsigned, pk);
                }
                if (i == lvec) {   // the label's column: (p - 1) g, from the same p the other columns are scaled from
#pragma unroll
                    for (int w = 0; w < 4; ++w)
                        if (w == (le >> 1)) {
                            const float x = (le & 1) ? bf_hi(v[j][w]) : bf_lo(v[j][w]);
                            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(x, LOG2E, -M - l2s));
                            const bf16 q = (bf16)((p - 1.f) * g);
                            const unsigned qb = (unsigned)__builtin_bit_cast(unsigned short, q);
                            o[w] = (le & 1) ? ((o[w] & 0xffffu) | (qb << 16)) : ((o[w] & 0xffff0000u) | qb);
                        }
                }
                // (columns past V inside the last vector were set to -inf: their gradient is exp2(-inf) = 0)
                st_stream(reinterpret_cast<u32x4*>(row + (long)i * 8), o);
            }
            if (has_next) load_vec(row_next, j);
        }
    }
}

// Fixed summation order (thread t adds rows t, t + 1024, ... in double; then a fixed tree): the loss is bitwise
// reproducible from run to run.
__global__ __launch_bounds__(1024) void ce_finalize_kernel(const float* __restrict__ row_loss, long T,
                                                           const float* __restrict__ inv_count,
                                                           float* __restrict__ loss) {
    __shared__ double red[1024];
    double acc = 0.0;
    for (long i0 = threadIdx.x; i0 < T; i0 += 4096) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = i0 + 1024 * u;
            v[u] = i < T ? row_loss[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += (double)v[u];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 512; o >= 1; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = (float)(red[0] * (double)(*inv_count));
}

}  // namespace

extern "C" int mgen_ce_count(const long* labels, long B, int S, long ignore_index, float* inv_count, int* count,
                             long V, int* err_flag, hipStream_t stream) {
    MGEN_REQUIRE(labels && inv_count && B > 0 && S > 0, "mgen_ce_count: bad arguments");
    hipLaunchKernelGGL(ce_count_kernel, dim3(1), dim3(1024), 0, stream, labels, B, S, ignore_index, inv_count, count, V,
                       err_flag);
    return mgen_check_launch("mgen_ce_count");
}

extern "C" int mgen_ce_fwd_bwd(void* logits, const long* labels, long tok0, long rows, int V, long ld, int S,
                               long ignore_index, const float* inv_count, float grad_scale, float* row_loss,
                               float* lse_out, int write_grad, int is_bf16, hipStream_t stream) {
    MGEN_REQUIRE(logits && labels && inv_count && row_loss, "mgen_ce_fwd_bwd: null pointer");
    MGEN_REQUIRE(rows > 0 && V > 0 && S > 0 && ld >= ((V + 7) / 8) * 8 && ld % 8 == 0,
                 "mgen_ce_fwd_bwd: need ld %% 8 == 0 and ld >= V rounded up to 8 (V=%d ld=%ld)", V, ld);
    const dim3 grid((unsigned)rows), block(256);
    const int nvec = (V + 7) / 8;
    static const int row_env = [] { const char* e = getenv("MGEN_CE_ROW"); return e ? atoi(e) : 1; }();
    // (the row kernel folds the gradient scale into its exponent as log2(g): a negative scale takes the other kernel)
    if (is_bf16 && row_env && grad_scale >= 0.f && nvec > 2048 && nvec <= 512 * 38) {
        // long bf16 rows (the 151,936-entry vocabulary): one 512-thread block keeps the row in registers
        // one resident block per CU (the row fills the register file), each walking rows grid-stride
        static const int ncu = [] {
            hipDeviceProp_t pr; int d = 0;
            if (hipGetDevice(&d) != hipSuccess || hipGetDeviceProperties(&pr, d) != hipSuccess) return 256;
            return pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
        }();
        const dim3 rgrid((unsigned)(rows < ncu ? rows : ncu));
#define CE_ROW(NV_, WG)                                                                                          \
    hipLaunchKernelGGL((ce_row_kernel<512, NV_, WG>), rgrid, dim3(512), 0, stream, (bf16*)logits, labels, tok0, rows, V, \
                       ld, S, ignore_index, inv_count, grad_scale, row_loss, lse_out)
        if (nvec <= 512 * 16) { if (write_grad) CE_ROW(16, true); else CE_ROW(16, false); }
        else { if (write_grad) CE_ROW(38, true); else CE_ROW(38, false); }
#undef CE_ROW
        return mgen_check_launch("mgen_ce_fwd_bwd");
    }
#define CE_LAUNCH(TT, WG)                                                                                         \
    hipLaunchKernelGGL((ce_kernel<TT, WG>), grid, block, 0, stream, (TT*)logits, labels, tok0, V, ld, S, ignore_index, \
                       inv_count, grad_scale, row_loss, lse_out)
    if (is_bf16) { if (write_grad) CE_LAUNCH(bf16, true); else CE_LAUNCH(bf16, false); }
    else { if (write_grad) CE_LAUNCH(float, true); else CE_LAUNCH(float, false); }
#undef CE_LAUNCH
    return mgen_check_launch("mgen_ce_fwd_bwd");
}

extern "C" int mgen_ce_finalize(const float* row_loss, long T, const float* inv_count, float* loss, hipStream_t stream) {
    MGEN_REQUIRE(row_loss && inv_count && loss && T > 0, "mgen_ce_finalize: bad arguments");
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(1024), 0, stream, row_loss, T, inv_count, loss);
    return mgen_check_launch("mgen_ce_finalize");
}
