// Output-stationary GEMM (gemm_os.hip) for products with a long reduction: weight gradients (both operands stored
// with the reduction index as the row index) and the products with 512 output columns (input gradients, residual
// projections); called from the bf16 GEMM entry points.
#pragma once
#include <hip/hip_runtime.h>

// out modes: the OutMode values of gemm_impl.h (0 bf16, 1 fp32, 2 fp32 accumulate, 3 fp32 = resid + bf16-rounded
// product, 4 fp32 = resid + product)
bool mgen_gemm_os_fits(int M, int N, int K, long lda, long ldb, long ldc, long ldr, int transA, int transB,
                       int out_mode);
int mgen_gemm_os_launch(const void* A, const void* B, void* C, const void* resid, int M, int N, int K, long lda,
                        long ldb, long ldc, long ldr, int transA, int transB, int out_mode, hipStream_t stream);
