// Causal self-attention, plain instances: see attn_impl.h
#define MGEN_ATTN_DROP 0
#include "attn_impl.h"
