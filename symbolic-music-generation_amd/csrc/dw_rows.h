// Column sums over many rows (gain / bias gradients) without contended atomics: every workgroup of a launch leaves its
// partial row in a row of its own in a device array owned by the library, and a small second launch on the same stream
// adds the rows into the destination. (fp32 atomics on a few thousand addresses run at 6-9 G/s on this GPU whatever
// their spread: half the time of the kernels that used them.) The array holds one set of rows per (device, stream)
// pair, so launches that overlap on one GPU never share a set; a stream that finds every set taken gets slot -1 and
// its kernels fall back to atomics. Defined in norm.hip.
#pragma once
#include "common.h"

constexpr int MGEN_DW_ROW = 2048;    // floats per partial row
constexpr int MGEN_DW_MAXB = 512;    // partial rows (workgroups) per launch at most

int mgen_dw_slot(hipStream_t stream);          // the stream's set of rows on the current device, or -1
float* mgen_dw_rows(int slot);                 // device address of the set's rows [MGEN_DW_MAXB][MGEN_DW_ROW], or null
// dst[c] += sum over b < nblocks of rows[b][col0 + c] for c < n
void mgen_dw_reduce(float* dst, int n, int nblocks, int slot, int col0, hipStream_t stream);
