// bf16 MFMA GEMM whose reduction runs over two operand pairs: see gemm_impl.h
#define MGEN_GEMM_SEG 1
#include "gemm_impl.h"
