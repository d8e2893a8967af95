// Row-panel-in-registers GEMM for K = 512 products (gemm_rs.hip); called from the bf16 GEMM entry points.
#pragma once
#include <hip/hip_runtime.h>

// true when C[M,N] = A[M,512] * B[N,512]^T with a bf16 output can run on the register-panel kernel
bool mgen_gemm_rs_fits(int M, int N, int K, long lda, long ldb, long ldc);
// I > 0: SwiGLU form (B = [gate rows; up rows] of I rows each, C = [gate | up], act = silu(gate) * up [M][I]; N = 2 I)
int mgen_gemm_rs_launch(const void* A, const void* B, void* C, void* act, int M, int N, int I, long lda, long ldb,
                        long ldc, long ldact, hipStream_t stream);
