// HBM streaming-rate probe: what a read+write element-wise kernel can reach on this chip, by access recipe.
// Build: hipcc -O3 --offload-arch=gfx950 tools/lab/stream_rate.hip -o gpurun_out/stream_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// MODE bit 0: non-temporal stores, bit 1: non-temporal loads. UN = 16-byte vectors in flight per thread.
template <int MODE, int UN>
__global__ __launch_bounds__(256) void copy_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, long n) {
    const long stride = (long)gridDim.x * 256;
    for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += stride * UN) {
        u32x4 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long i = i0 + u * stride;
            if (i < n) v[u] = (MODE & 2) ? __builtin_nontemporal_load(in + i) : in[i];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long i = i0 + u * stride;
            if (i < n) {
                if (MODE & 1) __builtin_nontemporal_store(v[u], out + i);
                else out[i] = v[u];
            }
        }
    }
}

// block-contiguous variant: a workgroup owns a contiguous span of UN * 4 KiB per iteration
template <int MODE, int UN>
__global__ __launch_bounds__(256) void copy_block_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, long n) {
    const long per = 256L * UN;
    for (long b = blockIdx.x; b * per < n; b += gridDim.x) {
        const long base = b * per + threadIdx.x;
        u32x4 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long i = base + u * 256;
            if (i < n) v[u] = (MODE & 2) ? __builtin_nontemporal_load(in + i) : in[i];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long i = base + u * 256;
            if (i < n) {
                if (MODE & 1) __builtin_nontemporal_store(v[u], out + i);
                else out[i] = v[u];
            }
        }
    }
}

template <int MODE, int UN>
__global__ __launch_bounds__(256) void read_kernel(const u32x4* __restrict__ in, unsigned* __restrict__ out, long n) {
    const long stride = (long)gridDim.x * 256;
    unsigned acc = 0;
    for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += stride * UN) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long i = i0 + u * stride;
            if (i < n) {
                const u32x4 v = (MODE & 2) ? __builtin_nontemporal_load(in + i) : in[i];
                acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
            }
        }
    }
    if (acc == 0x12345u) out[0] = acc;
}

template <int MODE>
__global__ __launch_bounds__(256) void write_kernel(u32x4* __restrict__ out, long n) {
    const long stride = (long)gridDim.x * 256;
    const u32x4 v = {1u, 2u, 3u, (unsigned)threadIdx.x};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        if (MODE & 1) __builtin_nontemporal_store(v, out + i);
        else out[i] = v;
    }
}

template <typename F> static float time_ms(F f, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    f(); f();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) f();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    const long bytes = 1L << 30;   // per buffer
    const long n = bytes / 16;
    u32x4 *a, *b;
    unsigned* o;
    CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMalloc(&o, 64));
    CHECK(hipMemset(a, 1, bytes)); CHECK(hipMemset(b, 2, bytes));
    const int grids[] = {256 * 4, 256 * 8, 256 * 16, 256 * 64, (int)(n / 256 / 4)};
#define RUN(name, traffic, ...)                                                                   \
    for (int g : grids) {                                                                         \
        const float ms = time_ms([&] { __VA_ARGS__; }, 10);                                             \
        printf("%-34s grid %7d  %7.1f us  %6.2f TB/s\n", name, g, ms * 1e3, (traffic) / ms / 1e9); \
    }
    RUN("copy plain un4", 2.0 * bytes, copy_kernel<0, 4><<<g, 256>>>(a, b, n));
    RUN("copy nt-store un4", 2.0 * bytes, copy_kernel<1, 4><<<g, 256>>>(a, b, n));
    RUN("copy nt-load nt-store un4", 2.0 * bytes, copy_kernel<3, 4><<<g, 256>>>(a, b, n));
    RUN("copy nt-load un4", 2.0 * bytes, copy_kernel<2, 4><<<g, 256>>>(a, b, n));
    RUN("copy plain un8", 2.0 * bytes, copy_kernel<0, 8><<<g, 256>>>(a, b, n));
    RUN("copy nt-store un8", 2.0 * bytes, copy_kernel<1, 8><<<g, 256>>>(a, b, n));
    RUN("copy nt both un1", 2.0 * bytes, copy_kernel<3, 1><<<g, 256>>>(a, b, n));
    RUN("copy plain un1", 2.0 * bytes, copy_kernel<0, 1><<<g, 256>>>(a, b, n));
    RUN("blockcopy plain un4", 2.0 * bytes, copy_block_kernel<0, 4><<<g, 256>>>(a, b, n));
    RUN("blockcopy nt both un4", 2.0 * bytes, copy_block_kernel<3, 4><<<g, 256>>>(a, b, n));
    RUN("blockcopy nt both un8", 2.0 * bytes, copy_block_kernel<3, 8><<<g, 256>>>(a, b, n));
    RUN("blockcopy in-place plain un4", 2.0 * bytes, copy_block_kernel<0, 4><<<g, 256>>>(a, a, n));
    RUN("blockcopy in-place nt both un4", 2.0 * bytes, copy_block_kernel<3, 4><<<g, 256>>>(a, a, n));
    RUN("in-place plain un4", 2.0 * bytes, copy_kernel<0, 4><<<g, 256>>>(a, a, n));
    RUN("in-place nt both un4", 2.0 * bytes, copy_kernel<3, 4><<<g, 256>>>(a, a, n));
    RUN("read plain un4", 1.0 * bytes, read_kernel<0, 4><<<g, 256>>>(a, o, n));
    RUN("read nt un4", 1.0 * bytes, read_kernel<2, 4><<<g, 256>>>(a, o, n));
    RUN("read plain un8", 1.0 * bytes, read_kernel<0, 8><<<g, 256>>>(a, o, n));
    RUN("write plain", 1.0 * bytes, write_kernel<0><<<g, 256>>>(b, n));
    RUN("write nt", 1.0 * bytes, write_kernel<1><<<g, 256>>>(b, n));
    return 0;
}
