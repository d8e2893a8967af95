// Head-side kernels of the 4-D position model: constant input rows, unit-normalised angle pairs with the cosine
// loss (forward and gradient in one pass), and the decoding of a predicted row into the next note.
//
// Reference: /root/reference/src/model/modeling.py
//   :389-394,426-428  every position's input is the same vector [1,0,1,0,...]
//   :620-629          head output [T,16] viewed as 8 pairs, pred = pair / (|pair| + 1e-8)
//   :105-141          targets: per axis a fine and a coarse angle, (start, duration): pi*frac(x), pi*floor(x)/32;
//                     (pitch, velocity): pi*(x mod 12)/12, pi*(12*floor(x/12))/128; target pair = (cos, sin)
//   :632-660          rows count when all 4 label columns differ from -100; loss = mean over rows and pairs of
//                     1 - <pred, target>; no counted row -> loss 0
// and /root/reference/generate_music.py:110-142 (process_last_logits) and :173-174 (start-time accumulation).
#include "common.h"

namespace {

constexpr float kPi = 3.14159265358979323846f;

__global__ void fill_rows_kernel(const float* __restrict__ vec, float* __restrict__ out, long T, int d) {
    const long n4 = T * (d / 4);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % (d / 4));
        reinterpret_cast<f32x4*>(out)[i] = reinterpret_cast<const f32x4*>(vec)[c];
    }
}

__device__ __forceinline__ bool row_counts(const float* lab) {
    return lab[0] != -100.f && lab[1] != -100.f && lab[2] != -100.f && lab[3] != -100.f;
}

// inv_count = 1 / (8 * number of counted rows), or 0 when there is none
__global__ void cos4d_count_kernel(const float* __restrict__ labels, long T, float* __restrict__ inv_count,
                                   int* __restrict__ count) {
    int c = 0;
    for (long t = threadIdx.x; t < T; t += blockDim.x) c += row_counts(labels + 4 * t) ? 1 : 0;
    __shared__ int red[256];
    red[threadIdx.x] = c;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *inv_count = red[0] > 0 ? 1.f / (8.f * (float)red[0]) : 0.f;
        if (count) *count = red[0];
    }
}

// torch's remainder and floor-division for floats (python semantics), as used by `x % c` and `x // c * c`
__device__ __forceinline__ float py_mod(float x, float c) {
    float m = fmodf(x, c);
    if (m != 0.f && ((m < 0.f) != (c < 0.f))) m += c;
    return m;
}
__device__ __forceinline__ float py_floordiv(float x, float c) {
    float m = fmodf(x, c);
    float div = (x - m) / c;
    if (m != 0.f && ((c < 0.f) != (m < 0.f))) div -= 1.f;
    if (div == 0.f) return copysignf(0.f, x / c);
    float fl = floorf(div);
    if (div - fl > 0.5f) fl += 1.f;
    return fl;
}

// target angle of pair j (0..7) for a label row
__device__ __forceinline__ float target_angle(const float* lab, int j) {
    const float x = lab[j >> 1];
    const bool coarse = j & 1;
    if (j < 4) return coarse ? kPi * (py_floordiv(x, 1.f) * 1.f / 32.f) : kPi * (py_mod(x, 1.f) / 1.f);
    return coarse ? kPi * (py_floordiv(x, 12.f) * 12.f / 128.f) : kPi * (py_mod(x, 12.f) / 12.f);
}

// One thread per (row, pair): 8 consecutive lanes share a row.
//  pairs [T,16] (bf16 or f32), pred_out [T,16] f32 (optional), row_loss [T] = sum over the 8 pairs of 1 - cos_sim
//  (0 for rows that do not count), dpairs [T,16] (same type as pairs, optional) = d(loss)/d(pairs) * grad_scale.
template <typename IoT>
__global__ __launch_bounds__(256) void cos4d_kernel(const IoT* __restrict__ pairs, const float* __restrict__ labels,
                                                    const float* __restrict__ inv_count, float grad_scale,
                                                    float* __restrict__ pred_out, float* __restrict__ row_loss,
                                                    IoT* __restrict__ dpairs, long T) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long t = i >> 3;
    const int j = (int)(i & 7);
    if (t >= T) return;   // T*8 threads: the 8 lanes of a row leave together
    const float a = (float)pairs[t * 16 + 2 * j], b = (float)pairs[t * 16 + 2 * j + 1];
    const float n = sqrtf(a * a + b * b);
    const float den = n + 1e-8f;
    const float pa = a / den, pb = b / den;
    if (pred_out) {
        pred_out[t * 16 + 2 * j] = pa;
        pred_out[t * 16 + 2 * j + 1] = pb;
    }
    if (!labels) return;
    const float* lab = labels + 4 * t;
    const bool on = row_counts(lab);
    float l = 0.f, da = 0.f, db = 0.f;
    if (on) {
        const float ang = target_angle(lab, j);
        const float ct = cosf(ang), st = sinf(ang);
        l = 1.f - (pa * ct + pb * st);
        if (dpairs) {
            // g = dL/dpred = -target * inv_count ; d pred/d pair = I/den - pair pair^T / (n den^2)
            const float k = -(*inv_count) * grad_scale;
            const float ga = k * ct, gb = k * st;
            const float gp = n > 0.f ? (ga * a + gb * b) / (n * den * den) : 0.f;
            da = ga / den - gp * a;
            db = gb / den - gp * b;
        }
    }
    if (dpairs) {
        dpairs[t * 16 + 2 * j] = (IoT)da;
        dpairs[t * 16 + 2 * j + 1] = (IoT)db;
    }
    if (row_loss) {
        l = group_sum<8>(l);
        if (j == 0) row_loss[t] = l;
    }
}

// out8[j] = mean over the rows that count of 1 - cos_sim of pair j (pred holds unit pairs), 0 when no row counts.
// One workgroup; thread sums in double in a fixed order, so the eight values are reproducible.
__global__ __launch_bounds__(1024) void cos4d_components_kernel(const float* __restrict__ pred,
                                                                  const float* __restrict__ labels, long T,
                                                                  float* __restrict__ out8) {
    __shared__ double red[1024];
    __shared__ int cnt[128];
    const int j = threadIdx.x & 7;
    const int lane_row = threadIdx.x >> 3;   // 128 rows per sweep
    double acc = 0.0;
    int c = 0;
    for (long t = lane_row; t < T; t += 128) {
        const float* lab = labels + 4 * t;
        if (!row_counts(lab)) continue;
        const float ang = target_angle(lab, j);
        acc += (double)(1.f - (pred[t * 16 + 2 * j] * cosf(ang) + pred[t * 16 + 2 * j + 1] * sinf(ang)));
        ++c;
    }
    red[threadIdx.x] = acc;
    if (j == 0) cnt[lane_row] = c;
    __syncthreads();
    for (int s = 64; s > 0; s >>= 1) {
        if (lane_row < s) {
            red[threadIdx.x] += red[threadIdx.x + 8 * s];
            if (j == 0) cnt[lane_row] += cnt[lane_row + s];
        }
        __syncthreads();
    }
    if (threadIdx.x < 8) out8[threadIdx.x] = cnt[0] > 0 ? (float)(red[threadIdx.x] / (double)cnt[0]) : 0.f;
}

// next[b] = decoded note from pred[b] (8 unit pairs), with the start time of prev[b] added to the start delta.
__global__ void pos4d_next_kernel(const float* __restrict__ pred, long ld_pred, const float* __restrict__ prev,
                                  long ld_prev, float* __restrict__ next, long ld_next, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* p = pred + (long)b * ld_pred;
    const float scale[8] = {1.f, 32.f, 1.f, 32.f, 12.f, 128.f, 12.f, 128.f};
    float pos[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float ang = atan2f(p[2 * j + 1], p[2 * j]) / kPi;
        ang = fminf(fmaxf(ang, 0.f), 1.f);
        pos[j] = ang * scale[j];
    }
    const float f0 = rintf(pos[1] / 1.f) * 1.f + pos[0];
    const float f1 = rintf(pos[3] / 1.f) * 1.f + pos[2];
    const float f2 = rintf(pos[5] / 12.f) * 12.f + pos[4];
    const float f3 = rintf(pos[7] / 12.f) * 12.f + pos[6];
    float* o = next + (long)b * ld_next;
    o[0] = f0 + prev[(long)b * ld_prev];
    o[1] = f1;
    o[2] = rintf(f2);
    o[3] = rintf(f3);
}

}  // namespace

extern "C" int mgen_fill_rows(const float* vec, float* out, long T, int d, hipStream_t stream) {
    MGEN_REQUIRE(vec && out && T > 0 && d > 0 && d % 4 == 0, "mgen_fill_rows: bad arguments (d must be a multiple of 4)");
    long nb = (T * (d / 4) + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(fill_rows_kernel, dim3((unsigned)nb), dim3(256), 0, stream, vec, out, T, d);
    return mgen_check_launch("mgen_fill_rows");
}

extern "C" int mgen_cos4d_count(const float* labels, long T, float* inv_count, int* count, hipStream_t stream) {
    MGEN_REQUIRE(labels && inv_count && T > 0, "mgen_cos4d_count: bad arguments");
    hipLaunchKernelGGL(cos4d_count_kernel, dim3(1), dim3(256), 0, stream, labels, T, inv_count, count);
    return mgen_check_launch("mgen_cos4d_count");
}

extern "C" int mgen_cos4d_fwd_bwd(const void* pairs, const float* labels, const float* inv_count, float grad_scale,
                                  float* pred_out, float* row_loss, void* dpairs, long T, int is_bf16,
                                  hipStream_t stream) {
    MGEN_REQUIRE(pairs && T > 0, "mgen_cos4d_fwd_bwd: bad arguments");
    MGEN_REQUIRE(!labels || inv_count, "mgen_cos4d_fwd_bwd: labels need inv_count (mgen_cos4d_count)");
    MGEN_REQUIRE(labels || (!row_loss && !dpairs), "mgen_cos4d_fwd_bwd: loss outputs need labels");
    const unsigned nb = (unsigned)((T * 8 + 255) / 256);
    if (is_bf16)
        hipLaunchKernelGGL(cos4d_kernel<bf16>, dim3(nb), dim3(256), 0, stream, (const bf16*)pairs, labels, inv_count,
                           grad_scale, pred_out, row_loss, (bf16*)dpairs, T);
    else
        hipLaunchKernelGGL(cos4d_kernel<float>, dim3(nb), dim3(256), 0, stream, (const float*)pairs, labels,
                           inv_count, grad_scale, pred_out, row_loss, (float*)dpairs, T);
    return mgen_check_launch("mgen_cos4d_fwd_bwd");
}

extern "C" int mgen_cos4d_components(const float* pred, const float* labels, long T, float* out8, hipStream_t stream) {
    MGEN_REQUIRE(pred && labels && out8 && T > 0, "mgen_cos4d_components: bad arguments");
    hipLaunchKernelGGL(cos4d_components_kernel, dim3(1), dim3(1024), 0, stream, pred, labels, T, out8);
    return mgen_check_launch("mgen_cos4d_components");
}

extern "C" int mgen_pos4d_next(const float* pred, long ld_pred, const float* prev, long ld_prev, float* next,
                               long ld_next, int B, hipStream_t stream) {
    MGEN_REQUIRE(pred && prev && next && B > 0 && ld_pred >= 16 && ld_prev >= 1 && ld_next >= 4,
                 "mgen_pos4d_next: bad arguments");
    hipLaunchKernelGGL(pos4d_next_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, stream, pred, ld_pred, prev,
                       ld_prev, next, ld_next, B);
    return mgen_check_launch("mgen_pos4d_next");
}
