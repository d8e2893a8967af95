// Note rows -> MIDI events: the step after the 4-D model's sampler, before the bytes go to disk.
//
// Reference: notes_to_midi, /root/reference/generate_music.py:62-105. Each row [start s, duration s, pitch, velocity]
// (fp32) becomes a note-on at int(start * k) and a note-off at int((start + duration) * k) ticks, k = tempo / 60 *
// ticks_per_beat; pitch and velocity are clipped to 0..127 and truncated. The reference iterates a float32 numpy
// array and multiplies by a Python float, which numpy >= 2 (the reference pins 2.3.1) evaluates in float32: the
// products and the sum below are single fp32 operations in that order, int() truncates toward zero.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void notes_to_events_kernel(const float* __restrict__ rows, long n, float k,
                                                              int* __restrict__ ticks, int* __restrict__ msgs) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f32x4 r = *reinterpret_cast<const f32x4*>(rows + 4 * i);
    const float on = __fmul_rn(r[0], k);
    const float off = __fmul_rn(__fadd_rn(r[0], r[1]), k);
    const int pitch = (int)fminf(fmaxf(r[2], 0.f), 127.f);
    const int vel = (int)fminf(fmaxf(r[3], 0.f), 127.f);
    ticks[2 * i] = (int)on;
    ticks[2 * i + 1] = (int)off;
    msgs[2 * i] = 0x90 | (pitch << 8) | (vel << 16);   // status | data1 << 8 | data2 << 16
    msgs[2 * i + 1] = 0x80 | (pitch << 8);
}

}  // namespace

// rows: fp32 [n][4] on the device (16-byte aligned). ticks, msgs: int32 [2n]; entry 2i is note i's note-on, 2i+1 its
// note-off, in input order (the caller sorts by tick, stably, as the reference does).
extern "C" int mgen_notes_to_events(const float* rows, long n, float ticks_per_second, int* ticks, int* msgs,
                                    hipStream_t stream) {
    MGEN_REQUIRE(rows && ticks && msgs && n > 0, "mgen_notes_to_events: bad arguments");
    MGEN_REQUIRE(((uintptr_t)rows % 16) == 0, "mgen_notes_to_events: rows must be 16-byte aligned");
    hipLaunchKernelGGL(notes_to_events_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rows, n,
                       ticks_per_second, ticks, msgs);
    return mgen_check_launch("mgen_notes_to_events");
}
