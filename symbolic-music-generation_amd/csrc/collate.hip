// Batch assembly on the device: the step either side of the forward pass that the reference runs on the host for
// every batch. Inputs are the concatenated examples already resident in HBM; outputs are the tensors the model call
// takes. Integer / copy work, HBM-bound, one thread per output element.
//
// Reference:
//  * token models, miditok DataCollator(pad_token_id, copy_inputs_as_labels=True) (/root/reference/src/train.py:33-34):
//    right-pad to the longest example, attention_mask = 1 on real tokens, labels = inputs with -100 on the padding;
//  * melody-harmonisation collate_fn (/root/reference/src/train_aria.py:118-145): right-pad, labels = inputs with the
//    first melody_length positions set to -100, padding NOT masked, no attention mask;
//  * 4-D position model, MidiDataset4D._create_chunks (/root/reference/src/train_model_new.py:114-158): chunk c of
//    the concatenated note rows, padded with [last start, 0, 2, 0]; labels = next row (the last label repeats the
//    last row) with the start replaced by the start difference to the current row; -100 on the padded tail;
//    attention_mask = 1 on real rows; input_ids = 0.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void collate_tokens_kernel(const long* __restrict__ flat,
                                                             const long* __restrict__ offsets,
                                                             const int* __restrict__ prompt_len,
                                                             long* __restrict__ ids, long* __restrict__ mask,
                                                             long* __restrict__ labels, int B, int S, long pad_id,
                                                             int mask_pads) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * S) return;
    const int b = (int)(i / S), s = (int)(i % S);
    const long beg = offsets[b];
    const long len = offsets[b + 1] - beg;
    const bool real = s < len;
    const long tok = real ? flat[beg + s] : pad_id;
    ids[i] = tok;
    if (mask) mask[i] = real ? 1 : 0;
    if (labels) {
        long lab = tok;
        if (mask_pads && !real) lab = -100;
        if (prompt_len && s < prompt_len[b]) lab = -100;
        labels[i] = lab;
    }
}

__global__ __launch_bounds__(256) void collate_pos4d_kernel(const float* __restrict__ rows, long total,
                                                            const long* __restrict__ chunk_ids,
                                                            float* __restrict__ pos, float* __restrict__ labels,
                                                            long* __restrict__ mask, long* __restrict__ ids, int B,
                                                            int S) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * S) return;
    const int b = (int)(i / S), s = (int)(i % S);
    const long beg = chunk_ids[b] * S;
    long len = total - beg;
    if (len > S) len = S;
    if (len < 0) len = 0;
    const float pad_t = len > 0 ? rows[(beg + len - 1) * 4] : 0.f;
    auto row_at = [&](int t, float (&r)[4]) {   // row t of the padded chunk
        if (t < len) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(rows + (beg + t) * 4);
            r[0] = v[0]; r[1] = v[1]; r[2] = v[2]; r[3] = v[3];
        } else {
            r[0] = pad_t; r[1] = 0.f; r[2] = 2.f; r[3] = 0.f;
        }
    };
    float cur[4], nxt[4];
    row_at(s, cur);
    row_at(s + 1 < S ? s + 1 : S - 1, nxt);
    f32x4 lab = {nxt[0] - cur[0], nxt[1], nxt[2], nxt[3]};
    if (s >= len) lab = f32x4{-100.f, -100.f, -100.f, -100.f};
    *reinterpret_cast<f32x4*>(pos + i * 4) = f32x4{cur[0], cur[1], cur[2], cur[3]};
    *reinterpret_cast<f32x4*>(labels + i * 4) = lab;
    if (mask) mask[i] = s < len ? 1 : 0;
    if (ids) ids[i] = 0;
}

__global__ __launch_bounds__(256) void right_padding_kernel(const long* __restrict__ mask, int B, int S,
                                                            int* __restrict__ err) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * S) return;
    const int s = (int)(i % S);
    if (s + 1 < S && mask[i] == 0 && mask[i + 1] != 0) atomicOr(err, 2);
}

}  // namespace

// Sets bit 1 (value 2) of *err_flag if some row of mask [B][S] has a 0 followed by a non-zero (not right padding).
extern "C" int mgen_check_right_padding(const long* mask, int B, int S, int* err_flag, hipStream_t stream) {
    MGEN_REQUIRE(mask && err_flag && B > 0 && S > 0, "mgen_check_right_padding: bad arguments");
    const long n = (long)B * S;
    hipLaunchKernelGGL(right_padding_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, mask, B, S,
                       err_flag);
    return mgen_check_launch("mgen_check_right_padding");
}

// flat: the B examples back to back (int64); offsets[B+1]: where each starts. Outputs [B][S] int64 (S >= the longest
// example): ids, attention mask (may be null), labels (may be null). mask_pads: labels = -100 on the padding.
// prompt_len (int32 [B], may be null): labels = -100 on the first prompt_len[b] positions.
extern "C" int mgen_collate_tokens(const long* flat, const long* offsets, const int* prompt_len, long* ids, long* mask,
                                   long* labels, int B, int S, long pad_id, int mask_pads, hipStream_t stream) {
    MGEN_REQUIRE(flat && offsets && ids && B > 0 && S > 0, "mgen_collate_tokens: bad arguments");
    const long n = (long)B * S;
    hipLaunchKernelGGL(collate_tokens_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, flat, offsets,
                       prompt_len, ids, mask, labels, B, S, pad_id, mask_pads);
    return mgen_check_launch("mgen_collate_tokens");
}

// rows: all note rows back to back, fp32 [total][4]; chunk_ids[B]: which S-row chunk each batch entry takes.
// Outputs: pos, labels fp32 [B][S][4]; mask, ids int64 [B][S] (may be null).
extern "C" int mgen_collate_pos4d(const float* rows, long total, const long* chunk_ids, float* pos, float* labels,
                                  long* mask, long* ids, int B, int S, hipStream_t stream) {
    MGEN_REQUIRE(rows && chunk_ids && pos && labels && total > 0 && B > 0 && S > 0, "mgen_collate_pos4d: bad arguments");
    MGEN_REQUIRE(((uintptr_t)rows % 16) == 0 && ((uintptr_t)pos % 16) == 0 && ((uintptr_t)labels % 16) == 0,
                 "mgen_collate_pos4d: rows, pos and labels must be 16-byte aligned");
    const long n = (long)B * S;
    hipLaunchKernelGGL(collate_pos4d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rows, total,
                       chunk_ids, pos, labels, mask, ids, B, S);
    return mgen_check_launch("mgen_collate_pos4d");
}
