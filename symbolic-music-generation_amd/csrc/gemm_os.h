// Output-stationary GEMM for products whose two operands are stored with the reduction index as the row index
// (weight gradients), gemm_os.hip; called from the bf16 GEMM entry points.
#pragma once
#include <hip/hip_runtime.h>

bool mgen_gemm_os_fits(int M, int N, int K, long lda, long ldb, long ldc);
// C[M,N] (fp32) = or += A^T B, A stored [K][M], B stored [K][N]; accumulate != 0: added to C with fp32 atomics
int mgen_gemm_os_launch(const void* A, const void* B, float* C, int M, int N, int K, long lda, long ldb, long ldc,
                        int accumulate, hipStream_t stream);
