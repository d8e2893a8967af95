// bf16 MFMA GEMM, plain instances: see gemm_impl.h
#define MGEN_GEMM_SEG 0
#include "gemm_impl.h"
