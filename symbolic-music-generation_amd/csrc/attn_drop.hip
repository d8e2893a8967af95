// Causal self-attention with dropout on the probabilities: see attn_impl.h
#define MGEN_ATTN_DROP 1
#include "attn_impl.h"
