// Issue-rate probe for one wave per SIMD on gfx950: cycles (s_memtime) per instruction of several instruction mixes.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int T>
__global__ __launch_bounds__(256, 1) void probe(unsigned* out, float seed, const char* gbuf) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    f32x2 p0 = {seed, seed}, p1 = p0, p2 = p0, p3 = p0;
    f32x16 acc0 = {}, acc1 = {};
    bf16x8 fa = {}, fb = {};
    int addr = threadIdx.x * 16;
    const char* gptr = gbuf + (blockIdx.x * 256 + threadIdx.x) * 16;
    const unsigned ldsb = __builtin_amdgcn_readfirstlane((unsigned)(__UINTPTR_TYPE__)(const __attribute__((address_space(3))) char*)lds + (threadIdx.x >> 6) * 4096);
    bf16x8 l0 = {}, l1 = {};
    f32x2 q0 = {}, q1 = {};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 64; ++it) {
        if (T == 1) { REP8(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (T == 2) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));) }
        if (T == 3) { REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (T == 4) { REP8(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0" : "+v"(a0));) }   // dependent chain
        if (T == 5) { REP8(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1" : "+v"(acc0), "+v"(acc1) : "v"(fa), "v"(fb));) }
        if (T == 6) { REP8(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n v_fma_f32 %8, %8, %8, %8\n v_fma_f32 %9, %9, %9, %9\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n v_fma_f32 %8, %8, %8, %8\n v_fma_f32 %9, %9, %9, %9" : "+v"(acc0), "+v"(acc1) : "v"(fa), "v"(fb), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5));) }   // 2 MFMA + 12 fma
        if (T == 7) { REP8(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(acc0), "+v"(acc1) : "v"(fa), "v"(fb), "v"(a0), "v"(a1), "v"(a2), "v"(a3));) }   // 2 MFMA + 4 exp
        if (T == 8) { REP8(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_pk_fma_f32 %4, %4, %4, %4\n v_pk_fma_f32 %5, %5, %5, %5\n v_pk_fma_f32 %6, %6, %6, %6\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n v_pk_fma_f32 %7, %7, %7, %7\n v_pk_fma_f32 %4, %4, %4, %4\n v_pk_fma_f32 %5, %5, %5, %5" : "+v"(acc0), "+v"(acc1) : "v"(fa), "v"(fb), "v"(p0), "v"(p1), "v"(p2), "v"(p3));) }   // 2 MFMA + 6 pk_fma
        if (T == 9) { REP8(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n ds_read_b128 %4, %6\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n ds_read_b128 %5, %6 offset:4096\n s_waitcnt lgkmcnt(0)" : "+v"(acc0), "+v"(acc1) : "v"(fa), "v"(fb), "v"(l0), "v"(l1), "v"(addr));) }   // 2 MFMA + 2 LDS reads + wait
        if (T == 10) { REP8(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_exp_f32 %7, %7\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_exp_f32 %8, %8" : "+v"(acc0), "+v"(acc1) : "v"(fa), "v"(fb), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4));) }   // per MFMA: 3 fma + 1 exp
        if (T == 11) { REP8(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0" : "+v"(acc0), "+v"(acc1) : "v"(fa), "v"(fb));) }   // 2 MFMA + 12 s_nop
        if (T == 12) { REP8(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n v_fma_f32 %8, %8, %8, %8\n v_fma_f32 %9, %9, %9, %9\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n v_fma_f32 %8, %8, %8, %8\n v_fma_f32 %9, %9, %9, %9\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5" : "+v"(acc0), "+v"(acc1) : "v"(fa), "v"(fb), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5));) }   // 2 MFMA + 16 fma

        if (T == 13) { REP8(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n ds_read_b128 %4, %6\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n ds_read_b128 %5, %6 offset:4096\n s_waitcnt lgkmcnt(2)" : "+v"(acc0), "+v"(acc1) : "v"(fa), "v"(fb), "v"(l0), "v"(l1), "v"(addr));) }
        if (T == 14) { REP8(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n ds_read_b128 %4, %6\n ds_read_b128 %5, %6 offset:1024\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n ds_read_b128 %4, %6 offset:2048\n ds_read_b128 %5, %6 offset:4096\n s_waitcnt lgkmcnt(4)" : "+v"(acc0), "+v"(acc1) : "v"(fa), "v"(fb), "v"(l0), "v"(l1), "v"(addr));) }
        if (T == 15) { REP8(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n ds_read_b64_tr_b16 %4, %6\n ds_read_b64_tr_b16 %5, %6 offset:1024\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n ds_read_b64_tr_b16 %4, %6 offset:2048\n ds_read_b64_tr_b16 %5, %6 offset:4096\n s_waitcnt lgkmcnt(4)" : "+v"(acc0), "+v"(acc1) : "v"(fa), "v"(fb), "v"(q0), "v"(q1), "v"(addr));) }
        if (T == 16) { REP8(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_cvt_pk_bf16_f32 %4, %5, %6\n v_cvt_pk_bf16_f32 %4, %5, %6\n v_cvt_pk_bf16_f32 %4, %5, %6\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n v_xor_b32 %7, 32, %7\n v_xor_b32 %7, 64, %7\n v_xor_b32 %7, 96, %7" : "+v"(acc0), "+v"(acc1) : "v"(fa), "v"(fb), "v"(a0), "v"(a1), "v"(a2), "v"(addr));) }
        if (T == 17) { REP8(asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:1024\n ds_read_b128 %0, %2 offset:2048\n ds_read_b128 %1, %2 offset:3072\n ds_read_b128 %0, %2 offset:4096\n ds_read_b128 %1, %2 offset:5120\n ds_read_b128 %0, %2 offset:6144\n ds_read_b128 %1, %2 offset:7168\n s_waitcnt lgkmcnt(8)" : "+v"(l0), "+v"(l1) : "v"(addr));) }
        if (T == 18) { REP8(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n ds_read_b128 %4, %6\n v_fma_f32 %7, %7, %7, %7\n v_fma_f32 %8, %8, %8, %8\n v_fma_f32 %9, %9, %9, %9\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n ds_read_b128 %5, %6 offset:4096\n v_fma_f32 %7, %7, %7, %7\n v_fma_f32 %8, %8, %8, %8\n v_exp_f32 %9, %9\n s_waitcnt lgkmcnt(2)" : "+v"(acc0), "+v"(acc1) : "v"(fa), "v"(fb), "v"(l0), "v"(l1), "v"(addr), "v"(a0), "v"(a1), "v"(a2));) }
        if (T == 19) { REP8(asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]\n v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]\n v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]\n v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]\n v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]\n v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]\n v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]\n v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]" : : "v"(fa), "v"(fb) : "a0","a31");) }
        if (T == 20) { REP8(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(fa), "v"(fb));) }

        if (T == 21) { REP8(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n s_mov_b32 m0, %6\n s_nop 0\n global_load_lds_dwordx4 %4, off\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1" : "+v"(acc0), "+v"(acc1) : "v"(fa), "v"(fb), "v"(gptr), "v"(addr), "s"(ldsb) : "memory", "m0");) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        if (T == 22) { REP8(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n s_mov_b32 m0, %6\n s_nop 0\n global_load_lds_dwordx4 %4, off\n global_load_lds_dwordx4 %4, off offset:1024\n global_load_lds_dwordx4 %4, off offset:2048\n global_load_lds_dwordx4 %4, off offset:3072\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1" : "+v"(acc0), "+v"(acc1) : "v"(fa), "v"(fb), "v"(gptr), "v"(addr), "s"(ldsb) : "memory", "m0");) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        if (T == 23) { REP8(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n s_mov_b32 m0, %6\n s_nop 0\n global_load_lds_dwordx4 %4, off\n v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1" : "+v"(acc0), "+v"(acc1) : "v"(fa), "v"(fb), "v"(gptr), "v"(addr), "s"(ldsb) : "memory", "m0");) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        if (T == 24) { REP8(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n global_load_dwordx4 %5, %4, off\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n s_waitcnt vmcnt(4)" : "+v"(acc0), "+v"(acc1) : "v"(fa), "v"(fb), "v"(gptr), "v"(l0) : "memory");) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p1[0] + p2[1] + p3[1] + acc0[0] + acc1[3] + (float)l0[0] + (float)l1[1] + q0[0] + q1[1];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (unsigned)(t1 - t0); }
    if (sink == 12345.678f) out[1] = 1;
    if (threadIdx.x == 1000) lds[addr] = 1;
}

static char* gb;
template <int T> void run(const char* what, int ninstr, unsigned* d) {
    unsigned h[2];
    for (int r = 0; r < 2; ++r) {
        hipLaunchKernelGGL(probe<T>, dim3(256), dim3(256), 0, 0, d, 0.5f, gb);
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    }
    printf("%-44s %8u cycles  %6.2f per group of %d\n", what, h[0], h[0] / 512.0, ninstr);
}
int main() {
    unsigned* d; hipMalloc(&d, 64); hipMalloc(&gb, 64 << 20);
    run<1>("8 independent v_fma_f32", 8, d);
    run<2>("8 v_pk_fma_f32 (4 registers pairs)", 8, d);
    run<3>("8 independent v_exp_f32", 8, d);
    run<4>("8 dependent v_fma_f32", 8, d);
    run<5>("8 mfma 32x32x16 (2 accumulators)", 8, d);
    run<6>("2 mfma + 12 v_fma", 14, d);
    run<12>("2 mfma + 16 v_fma", 18, d);
    run<7>("2 mfma + 4 v_exp", 6, d);
    run<8>("2 mfma + 6 v_pk_fma", 8, d);
    run<9>("2 mfma + 2 ds_read_b128 + wait", 5, d);
    run<10>("2 mfma + 6 v_fma + 2 v_exp", 10, d);
    run<11>("2 mfma + 12 s_nop", 14, d);
    run<13>("2 mfma + 2 ds_read_b128, wait one group back", 4, d);
    run<14>("2 mfma + 4 ds_read_b128, wait one group back", 6, d);
    run<15>("2 mfma + 4 ds_read_b64_tr_b16, one group back", 6, d);
    run<16>("2 mfma + 3 cvt_pk + 3 xor", 8, d);
    run<17>("8 ds_read_b128 (4 waves/CU), one group back", 8, d);
    run<18>("2 mfma + 2 ds_read + 4 fma + 1 exp", 9, d);
    run<19>("8 mfma on 2 AGPR accumulators", 8, d);
    run<20>("8 mfma on ONE accumulator (dependent)", 8, d);
    run<21>("2 mfma + 1 global_load_lds x4 (L2 hits)", 3, d);
    run<22>("2 mfma + 4 global_load_lds x4", 6, d);
    run<23>("8 mfma + 1 global_load_lds x4", 9, d);
    run<24>("2 mfma + 1 global_load_dwordx4 to VGPR", 3, d);
    return 0;
}
