// LDS-DMA (buffer_load ... lds) issued from asm statements, shared by the ring-fed GEMM kernels (gemm_rs.hip,
// gemm_os.hip). The compiler never learns that LDS writes are pending, so it inserts no waits of its own: the
// kernels wait by hand with counted s_waitcnt vmcnt and publish a landed slot to the other waves with s_barrier.
#pragma once
#include "common.h"

// raw buffer descriptor over `bytes` bytes from p on (reads past the end return zeros)
__device__ __forceinline__ u32x4 dma_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}

// one wave instruction: lane l moves 16 bytes from buffer offset voff(l) + soff to LDS byte address dst + 16 l
__device__ __forceinline__ void dma_piece(const u32x4& rsrc, int voff, unsigned dst, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 : : "s"(dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}

template <int N>
__device__ __forceinline__ void dma_wait() {   // all but this wave's N youngest vector-memory operations are done
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
