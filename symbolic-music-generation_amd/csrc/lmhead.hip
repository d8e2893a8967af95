// lm_head + shifted cross entropy, forward and backward, over chunks of tokens: the [T, V] logits never exist at once.
//
// Reference: logits = lm_head(hidden) followed by ForCausalLMLoss (transformers models/qwen3/modeling_qwen3.py:495-499,
// loss/loss_utils.py:49-71: labels padded with -100 on the right and shifted, mean cross entropy over the labels that
// are not ignore_index) and autograd through both, as reached from /root/reference/src/model/model.py:142-143.
//
// Per chunk of at most chunk_tokens rows: logits = x W^T (bf16) -> cross entropy rewrites them in place as
// d(loss)/d(logits) -> dx = dlogits W -> dW += dlogits^T x. The first chunk of the call stores dW (no zero fill by
// the caller), every later chunk is added: one launch per chunk, as few atomics as possible.
// The caller owns every buffer; mgen_lmhead_ce_workspace_bytes says how large the workspace must be.
#include "common.h"

extern "C" int mgen_gemm_bf16(const void* A, const void* B, void* C, const void* resid, int M, int N, int K, long lda,
                              long ldb, long ldc, long ldr, int transA, int transB, int out_mode, float alpha,
                              hipStream_t stream);
extern "C" int mgen_ce_count(const long* labels, long B, int S, long ignore_index, float* inv_count, int* count, long V,
                             int* err_flag, hipStream_t stream);
extern "C" int mgen_ce_fwd_bwd(void* logits, const long* labels, long tok0, long rows, int V, long ld, int S,
                               long ignore_index, const float* inv_count, float grad_scale, float* row_loss,
                               float* lse_out, int write_grad, int is_bf16, hipStream_t stream);
extern "C" int mgen_ce_finalize(const float* row_loss, long T, const float* inv_count, float* loss, hipStream_t stream);

namespace {
inline long align256(long x) { return (x + 255) / 256 * 256; }
inline long ld_of(int V) { return ((long)V + 7) / 8 * 8; }
}  // namespace

// bytes of workspace for T = B * S tokens: one chunk of bf16 logits, the per-row losses, the label count
extern "C" long mgen_lmhead_ce_workspace_bytes(long T, int V, long chunk_tokens) {
    if (T <= 0 || V <= 0 || chunk_tokens <= 0) return -1;
    const long chunk = chunk_tokens < T ? chunk_tokens : T;
    return align256(chunk * ld_of(V) * 2) + align256(T * 4) + 256;
}

// x [T][ldx] bf16 (final-normed hidden rows, T = B * S), W [V][ldw] bf16 (the head's weight, [out, in]), labels int64
// [B][S]. loss: fp32 scalar on the device. with_grad != 0: dx [T][lddx] bf16 = d(loss)/dx and, unless dW is null
// (frozen head), dW [V][d] fp32 = d(loss)/dW, both for an upstream gradient of 1. err_flag (may be null): bit 4 is set
// when a label is outside [0, V) and is not ignore_index.
extern "C" int mgen_lmhead_ce_fwd_bwd(const void* x, long ldx, const void* W, long ldw, const long* labels, long B,
                                      int S, int V, int d, long ignore_index, long chunk_tokens, void* workspace,
                                      long workspace_bytes, float* loss, void* dx, long lddx, float* dW, int with_grad,
                                      int* err_flag, hipStream_t stream) {
    MGEN_REQUIRE(x && W && labels && workspace && loss, "mgen_lmhead_ce_fwd_bwd: null pointer");
    MGEN_REQUIRE(B > 0 && S > 0 && V > 0 && d > 0 && chunk_tokens > 0, "mgen_lmhead_ce_fwd_bwd: bad sizes");
    MGEN_REQUIRE(!with_grad || dx, "mgen_lmhead_ce_fwd_bwd: with_grad needs dx");
    const long T = B * S;
    const long need = mgen_lmhead_ce_workspace_bytes(T, V, chunk_tokens);
    MGEN_REQUIRE(workspace_bytes >= need, "mgen_lmhead_ce_fwd_bwd: workspace of %ld bytes, %ld needed", workspace_bytes, need);
    MGEN_REQUIRE(((uintptr_t)workspace % 256) == 0, "mgen_lmhead_ce_fwd_bwd: workspace must be 256-byte aligned");
    const long chunk = chunk_tokens < T ? chunk_tokens : T;
    const long ld = ld_of(V);
    char* ws = reinterpret_cast<char*>(workspace);
    bf16* logits = reinterpret_cast<bf16*>(ws);
    float* row_loss = reinterpret_cast<float*>(ws + align256(chunk * ld * 2));
    float* inv_count = reinterpret_cast<float*>(ws + align256(chunk * ld * 2) + align256(T * 4));
    const bf16* xb = reinterpret_cast<const bf16*>(x);
    bf16* dxb = reinterpret_cast<bf16*>(dx);
    if (int e = mgen_ce_count(labels, B, S, ignore_index, inv_count, nullptr, V, err_flag, stream)) return e;
    bool first = true;
    for (long t0 = 0; t0 < T; t0 += chunk) {
        const long n = chunk < T - t0 ? chunk : T - t0;
        const bf16* xc = xb + t0 * ldx;
        if (int e = mgen_gemm_bf16(xc, W, logits, nullptr, (int)n, V, d, ldx, ldw, ld, 0, 0, 0, /*bf16*/ 0, 1.0f, stream))
            return e;
        if (int e = mgen_ce_fwd_bwd(logits, labels, t0, n, V, ld, S, ignore_index, inv_count, 1.0f, row_loss, nullptr,
                                    with_grad ? 1 : 0, 1, stream))
            return e;
        if (!with_grad) continue;
        // dx[c] = dlogits[c] W
        if (int e = mgen_gemm_bf16(logits, W, dxb + t0 * lddx, nullptr, (int)n, d, V, ld, ldw, lddx, 0, 0, 1, 0, 1.0f, stream))
            return e;
        if (!dW) continue;
        // dW (+)= dlogits[c]^T x[c]: a plain store for the first chunk of the call, accumulating launches after it
        if (int e = mgen_gemm_bf16(logits, xc, dW, nullptr, V, d, (int)n, ld, ldx, d, 0, 1, 1,
                                   first ? /*fp32 store*/ 1 : /*fp32 accumulate*/ 2, 1.0f, stream))
            return e;
        first = false;
    }
    return mgen_ce_finalize(row_loss, T, inv_count, loss, stream);
}
