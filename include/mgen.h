/* libmgen_hip.so — C-ABI of the gfx950 (MI355X) kernels for the decoder-only MIDI language model hot path.
 *
 * The reference (xingjianll/symbolic-music-generation) has no FFI boundary of its own: its hot path is the Python
 * call `model(input_ids, attention_mask, labels)` / `model.generate(...)` (src/model/model.py:142-149,
 * src/test_tokenizer.py:129-136), whose arithmetic is executed by torch/transformers operators. Each entry point
 * below replaces one such operator (or a fused group of them) and cites the reference-side call it stands for
 * (TF = the `transformers` package the reference pins, files under transformers/).
 *
 * Conventions: every pointer is a DEVICE pointer unless said otherwise; sizes are element counts; "ld"/strides are in
 * elements; `stream` is a hipStream_t. Every function returns 0 on success, a negative code otherwise
 * (-1 invalid argument, -2 launch failure) and leaves a message for mgen_last_error() (thread local). No function
 * allocates or frees device memory or synchronises the device. bf16 = IEEE bfloat16 stored as 16-bit words.
 */
#ifndef MGEN_H
#define MGEN_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mgen_stream_t; /* hipStream_t */

/* ---- runtime ------------------------------------------------------------------------------------------------ */
const char* mgen_last_error(void);
int mgen_version(void);
/* identity of the kernel sources the library was built from (16 hex digits: a SHA-256 prefix over csrc/); the Python
 * loader refuses an in-tree library whose identity differs from the sources beside it. */
const char* mgen_source_id(void);
/* name of the device's gfx architecture into name[name_len]; returns the CU count. */
int mgen_device_info(int device, char* name, int name_len);

/* ---- linear layers -------------------------------------------------------------------------------------------
 * C[M,N] = alpha * op(A)[M,K] * op(B)[K,N], bf16 operands, fp32 accumulate on MFMA.
 * Replaces nn.Linear forward (TF models/qwen3/modeling_qwen3.py:76-83,225-236: x @ W^T with W [out,in]:
 * transA=0, transB=0), its input gradient (dY @ W: transB=1) and its weight gradient (dY^T @ X: transA=1, transB=1).
 * transA=0: A stored [M][lda>=K8]; transA=1: A stored [K][lda>=M8] (reduction index is the row).
 * transB=0: B stored [N][ldb>=K8]; transB=1: B stored [K][ldb>=N8]. (X8 = X rounded up to 8; padding columns of a
 * k-contiguous operand must hold zeros.)
 * out_mode: 0 bf16 store, 1 fp32 store, 2 fp32 accumulate (C += ...), 3 fp32 C = resid + float(bf16(acc))
 * (residual add of a bf16 projection output, TF models/qwen3/modeling_qwen3.py:316,322), 4 fp32 C = resid + acc,
 * 6 bf16 C = bf16(float(C) + acc) (a low-rank adapter's product added to its base layer's output: the LoRA mode of
 * the reference, src/model/model.py:251-271). */
int mgen_gemm_bf16(const void* A, const void* B, void* C, const void* resid, int M, int N, int K, long lda, long ldb,
                   long ldc, long ldr, int transA, int transB, int out_mode, float alpha, mgen_stream_t stream);

/* C = [A | A2] [B | B2]^T in one accumulator: the reduction runs over K indices of (A, B), then K2 of (A2, B2). A
 * projection with a low-rank adapter in one launch - peft's Linear.forward, base(x) + lora_B(lora_A(x)) * scaling
 * (reference src/model/model.py:251-271): y = x W^T + t B^T with t = scaling * x A^T, and its input gradient
 * dx = dy W + dt A (transB = 1: B = W stored [K][N], B2 = adapter A stored [K2][N]). A, A2 row-major; K, K2 multiples
 * of 64 (the rank is padded with zero columns / rows); out_mode 0, 1, 3 or 4. */
int mgen_gemm_bf16_seg(const void* A, const void* B, const void* A2, const void* B2, void* C, const void* resid, int M,
                       int N, int K, int K2, long lda, long ldb, long lda2, long ldb2, long ldc, long ldr, int transB,
                       int out_mode, mgen_stream_t stream);
/* Split-K without atomics for products with few output tiles (the decode path: M = number of streams). The K range
 * is cut into nslabs slices (mgen_gemm_slab_count(K, max_slabs) gives a valid count: 128 reduction indices per slice,
 * more when that would exceed max_slabs); slice z stores its fp32 partial product [M][N] at slabs + z*M*N.
 * mgen_slab_resid_rmsnorm is the consumer for the two residual projections of a decoder layer
 * (TF models/qwen3/modeling_qwen3.py:294-323): h_out = resid + sum_z slab_z (the sum rounded to bf16 first when
 * round_bf16), then, when w and y are given, y = bf16(w * h_out * rsqrt(mean(h_out^2) + eps)) — the norm that follows. */
int mgen_gemm_slab_count(int K, int max_slabs);
int mgen_gemm_bf16_slabs(const void* A, const void* B, float* slabs, int M, int N, int K, long lda, long ldb,
                         int transA, int transB, int nslabs, mgen_stream_t stream);
int mgen_slab_resid_rmsnorm(const float* slabs, int nslabs, long slab_stride, const float* resid, float* h_out,
                            const float* w, void* y, long T, int d, float eps, int round_bf16, mgen_stream_t stream);

/* ---- RMSNorm over the fp32 residual stream -------------------------------------------------------------------
 * Qwen3RMSNorm.forward, reference src/model/modeling.py:60-65. y is bf16 (out_bf16=1) or fp32; rstd[T] is saved for
 * the backward. d % 4 == 0, d <= 2048. */
int mgen_rmsnorm_fwd(const float* x, const float* w, void* y, float* rstd, long T, int d, float eps, int out_bf16,
                     mgen_stream_t stream);
/* The residual addition that ends a decoder layer's attention or MLP branch (TF models/qwen3/modeling_qwen3.py:316,322)
 * and the RMSNorm that follows it, in one pass: h_out[T,d] (fp32) = resid + float(y), y = the branch's last projection
 * in bf16 [T][ldy]; x_out (bf16 [T,d]) = w * (h_out * rstd); rstd[T] saved for the backward (may be null). Same bits as
 * mgen_gemm_bf16 with out_mode 3 followed by mgen_rmsnorm_fwd; lets that projection run as a bf16-output product. */
int mgen_resid_rmsnorm_fwd(const float* resid, const void* y, long ldy, const float* w, float* h_out, void* x_out,
                           float* rstd, long T, int d, float eps, mgen_stream_t stream);
/* autograd backward of the above: dres (fp32 [T,d]) = (accumulate ? dres : 0) + dx; optional copy of the updated dres
 * in the dy dtype (dres_lp, may be null); dw[d] += sum_rows dy * x * rstd. The blocks of a launch meet in a small
 * device array that the library owns (one per GPU): calls on one stream are ordered, calls that overlap on one GPU
 * from different streams would race on it. */
int mgen_rmsnorm_bwd(const void* dy, const float* x, const float* w, const float* rstd, float* dres, void* dres_lp,
                     float* dw, long T, int d, int accumulate, int dy_bf16, mgen_stream_t stream);

/* ---- rotary embedding + per-head q/k RMSNorm -----------------------------------------------------------------
 * cos_sin[npos][half_dim][2] = {cos, sin}(pos * inv_freq[i]) — Qwen3RotaryEmbedding.forward,
 * TF models/qwen3/modeling_qwen3.py:126-137 (inv_freq computed by the caller as at :120-123). */
int mgen_rope_table(const float* inv_freq, float* cos_sin, int npos, int half_dim, mgen_stream_t stream);
/* q_norm/k_norm + apply_rotary_pos_emb (TF models/qwen3/modeling_qwen3.py:252-257, :140-170, split-half pairing) for
 * the HQ query heads and HK key heads stored at the start of each row of qkv [T][ld_in]. wq/wk null = no norm
 * (Llama-style block). pos[T] int32 or null (position = t % S). D = 64 or 128. qout/kout: [T][ld_q], [T][ld_k]. */
int mgen_qknorm_rope_fwd(const void* qkv, long ld_in, void* qout, void* kout, long ld_q, long ld_k, const float* wq,
                         const float* wk, const float* cos_sin, const int* pos, int S, long T, int HQ, int HK, int D,
                         float eps, int is_bf16, mgen_stream_t stream);
/* backward of the above; dqk holds d(out) for the q and k heads ([T][ld_in]); results go to dq_raw/dk_raw (may alias
 * dqk); raw_qkv is the forward input; dwq/dwk[D] += gain gradients. */
int mgen_qknorm_rope_bwd(const void* dqk, long ld_in, void* dq_raw, void* dk_raw, long ld_q, long ld_k,
                         const void* raw_qkv, long ld_raw, const float* wq, const float* wk, float* dwq, float* dwk,
                         const float* cos_sin, const int* pos, int S, long T, int HQ, int HK, int D, float eps,
                         int is_bf16, mgen_stream_t stream);
/* The same two with adjacent pairs (elements 2i and 2i+1 of a head rotate by the angle of frequency i; the pairing of
 * GPT-J-style blocks, which the Aria checkpoint the reference loads at src/model/model.py:191 is recalled to use) and
 * the same position-indexed table. Equal to the split-half form on q/k projections whose output rows are permuted
 * [0, 2, 4, ..., 1, 3, 5, ...] within each head. */
int mgen_qknorm_rope_il_fwd(const void* qkv, long ld_in, void* qout, void* kout, long ld_q, long ld_k, const float* wq,
                            const float* wk, const float* cos_sin, const int* pos, int S, long T, int HQ, int HK, int D,
                            float eps, int is_bf16, mgen_stream_t stream);
int mgen_qknorm_rope_il_bwd(const void* dqk, long ld_in, void* dq_raw, void* dk_raw, long ld_q, long ld_k,
                            const void* raw_qkv, long ld_raw, const float* wq, const float* wk, float* dwq, float* dwk,
                            const float* cos_sin, const int* pos, int S, long T, int HQ, int HK, int D, float eps,
                            int is_bf16, mgen_stream_t stream);

/* ---- 4-D position model: adjacent-pair rotary embedding from continuous note coordinates ------------------------
 * Replaces Qwen3Model.rot_pos_emb (reference src/model/modeling.py:498-545): cos_sin[t][a*(D/8)+j] =
 * (cos, sin)(pos4[t][a] * freqs[j]) for the 4 axes a and D/8 frequencies j (the caller passes the reference's
 * frequency vector: D/8-4 values theta^(-2j/D) followed by 4 zeros). pos4 [T][4] fp32, cos_sin [T][D/2][2] fp32. */
int mgen_rope4d_table(const float* pos4, const float* freqs, float* cos_sin, long T, int D, mgen_stream_t stream);
/* Replaces q_norm/k_norm + apply_rotary_pos_emb on q,k and apply_rotary_pos_emb_1 on v with the adjacent pairing
 * (x0,x1)->(-x1,x0) (reference src/model/modeling.py:87-98,164-174,264-270). qkv [T][ld_in] holds HQ query heads,
 * HK key heads, HV value heads of width D; q/k rows are RMS-normalised with gains wq/wk (null = no norm), v rows are
 * only rotated. Any head count may be 0: HQ = HK = 0 is a plain rotation of HV heads by +angle. */
int mgen_qkvnorm_rope4d_fwd(const void* qkv, long ld_in, void* qout, void* kout, void* vout, long ld_q, long ld_k,
                            long ld_v, const float* wq, const float* wk, const float* cos_sin, long T, int HQ, int HK,
                            int HV, int D, float eps, int is_bf16, mgen_stream_t stream);
/* Backward of the above (outputs may alias dqkv). With HQ = HK = 0 it is the rotation by -angle, which is also the
 * forward of apply_rotary_pos_emb_T on the attention output (reference src/model/modeling.py:176-180,300-302). */
int mgen_qkvnorm_rope4d_bwd(const void* dqkv, long ld_in, void* dq_raw, void* dk_raw, void* dv_raw, long ld_q,
                            long ld_k, long ld_v, const void* raw_qkv, long ld_raw, const float* wq, const float* wk,
                            float* dwq, float* dwk, const float* cos_sin, long T, int HQ, int HK, int HV, int D,
                            float eps, int is_bf16, mgen_stream_t stream);
/* out[t][:] = vec[:] for every t: the constant input vector of the 4-D model (reference src/model/modeling.py:
 * 389-394,426-428). */
int mgen_fill_rows(const float* vec, float* out, long T, int d, mgen_stream_t stream);
/* Cosine loss of the 4-D model (reference src/model/modeling.py:620-660 with the targets of :105-141).
 * labels [T][4] fp32; a row counts when none of its 4 values is -100. inv_count = 1/(8*rows that count) or 0. */
int mgen_cos4d_count(const float* labels, long T, float* inv_count, int* count, mgen_stream_t stream);
/* pairs [T][16] (bf16 or fp32) = head output; pred_out [T][16] fp32 = unit-normalised pairs (optional);
 * with labels: row_loss[T] = sum over the 8 pairs of 1 - <pred, target> (0 for rows that do not count), and
 * dpairs [T][16] (type of pairs, optional) = grad_scale * d(mean loss)/d(pairs). */
int mgen_cos4d_fwd_bwd(const void* pairs, const float* labels, const float* inv_count, float grad_scale,
                       float* pred_out, float* row_loss, void* dpairs, long T, int is_bf16, mgen_stream_t stream);
/* The eight per-pair means the reference logs beside the loss ("loss/s_0" ... "loss/v_1", reference
 * src/model/modeling.py:646-655): out8[j] = mean over the rows that count of 1 - <pred pair j, target pair j>
 * (pred [T][16] fp32 unit pairs from mgen_cos4d_fwd_bwd); zeros when no row counts. Their mean is the loss. */
int mgen_cos4d_components(const float* pred, const float* labels, long T, float* out8, mgen_stream_t stream);
/* process_last_logits + the start-time accumulation of generate_music (reference generate_music.py:110-142,173-174):
 * next[b][0..3] = note decoded from the 8 unit pairs pred[b][0..15], start = decoded delta + prev[b][0]. */
int mgen_pos4d_next(const float* pred, long ld_pred, const float* prev, long ld_prev, float* next, long ld_next, int B,
                    mgen_stream_t stream);

/* ---- causal self-attention (bf16 MFMA, head_dim 128) ---------------------------------------------------------
 * scaled_dot_product_attention(is_causal=True, scale) — TF integrations/sdpa_attention.py:154-163, eager twin
 * TF models/qwen3/modeling_qwen3.py:185-207. q/k/v/out are token-major [B*S][heads][128] views with token strides
 * sq/sk/sv/so; lse [B][H][S] fp32 (natural log) is saved for the backward. */
int mgen_attn_fwd(const void* q, const void* k, const void* v, void* out, float* lse, long sq, long sk, long sv,
                  long so, int B, int S, int H, int KVH, int head_dim, float scale, mgen_stream_t stream);
/* autograd backward: dq/dk/dv (bf16) from dout; delta [B][H][S] fp32 is caller-provided scratch. */
int mgen_attn_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
                  float* delta, void* dq, void* dk, void* dv, long sq, long sk, long sv, long so, long sdo, long sdq,
                  long sdk, long sdv, int B, int S, int H, int KVH, int head_dim, float scale, mgen_stream_t stream);
/* The same two with dropout on the attention probabilities — GPT2Attention's attn_dropout, TF
 * models/gpt2/modeling_gpt2.py (eager_attention_forward: softmax, dropout, then the product with v; attn_pdrop 0.1,
 * configuration_gpt2.py:87). Probability (b*H + h, q, key) is kept iff fmix32(rowkey ^ key) >= floor(p * 2^32) with
 * rowkey = fmix32(((b*H + h) * 0x9E3779B1 + q) ^ low32(key64)) + high32(key64) and fmix32 the 32-bit murmur3
 * finaliser; kept probabilities are scaled by 1/(1-p), the softmax normaliser counts all of them. The backward call
 * takes the p and key of its forward call (the mask is recomputed, never stored). p in [0, 1). */
int mgen_attn_fwd_dropout(const void* q, const void* k, const void* v, void* out, float* lse, long sq, long sk, long sv,
                          long so, int B, int S, int H, int KVH, int head_dim, float scale, float p,
                          unsigned long long key, mgen_stream_t stream);
int mgen_attn_bwd_dropout(const void* q, const void* k, const void* v, const void* out, const void* dout,
                          const float* lse, float* delta, void* dq, void* dk, void* dv, long sq, long sk, long sv,
                          long so, long sdo, long sdq, long sdk, long sdv, int B, int S, int H, int KVH, int head_dim,
                          float scale, float p, unsigned long long key, mgen_stream_t stream);

/* ---- element-wise dropout ------------------------------------------------------------------------------------
 * torch.nn.Dropout in training mode as GPT-2 uses it (TF models/gpt2/modeling_gpt2.py: `drop` on the summed
 * embeddings, `resid_dropout` after attn.c_proj, `dropout` after mlp.c_proj; configuration_gpt2.py:85-86) and the
 * adapters' lora_dropout (reference src/model/model.py:255). Element i is kept iff
 * fmix32(fmix32(low32(i) ^ low32(key)) + high32(key) + high32(i) * 0x9E3779B1) >= floor(p * 2^32); kept elements are
 * scaled by 1/(1-p) and rounded to the tensor's type. y may alias x; the gradient is the same call on the upstream
 * gradient. x, y: n contiguous bf16 (is_bf16) or fp32 elements, 16-byte aligned. */
int mgen_dropout(const void* x, void* y, long n, int is_bf16, float p, unsigned long long key, mgen_stream_t stream);
/* y += dropout(x), each term rounded to the tensors' type (a dropped branch's gradient joining the others'). */
int mgen_dropout_add(const void* x, void* y, long n, int is_bf16, float p, unsigned long long key,
                     mgen_stream_t stream);
/* out = resid + float(bf16(dropout(y))): the residual add after a dropped bf16 projection (y bf16, resid/out fp32). */
int mgen_dropout_resid(const void* y, const float* resid, float* out, long n, float p, unsigned long long key,
                       mgen_stream_t stream);

/* ---- SwiGLU gate ---------------------------------------------------------------------------------------------
 * act = silu(gate) * up with gu = [gate | up] per row ([T][2I] bf16) — Qwen3MLP.forward,
 * TF models/qwen3/modeling_qwen3.py:81-83. round_silu=1 rounds silu(gate) to bf16 first (autocast behaviour). */
int mgen_swiglu_fwd(const void* gu, void* act, long T, int I, int round_silu, mgen_stream_t stream);
int mgen_swiglu_bwd(const void* gu, const void* dact, void* dgu, long T, int I, mgen_stream_t stream);
/* The same with row pitches in elements (activation rows padded so that their pitch is not a multiple of the memory
 * channel period: a kernel whose workgroups read the same column segment of many rows in step otherwise queues on one
 * channel). */
int mgen_swiglu_bwd_ld(const void* gu, long ldgu, const void* dact, long lddact, void* dgu, long lddgu, long T, int I,
                       mgen_stream_t stream);
/* The gate/up projection and the gate in one launch: gu[M][2I] = x[M][K] * w[2I][K]^T (w = gate rows, then up rows) and
 * act[M][I] = silu(gate) * up from the product's epilogue, same rounding as mgen_swiglu_fwd(round_silu = 1) on the stored
 * gu — gate_proj, up_proj and act_fn of Qwen3MLP.forward, TF models/qwen3/modeling_qwen3.py:81-83. K % 64 == 0,
 * I % 128 == 0. */
int mgen_gemm_bf16_swiglu(const void* x, const void* w, void* gu, void* act, int M, int I, int K, long ldx, long ldw,
                          long ldgu, long ldact, mgen_stream_t stream);

/* ---- token embedding -----------------------------------------------------------------------------------------
 * nn.Embedding gather (TF models/qwen3/modeling_qwen3.py:352,381) and its index_add backward. ids int64 [T].
 * *err_flag (int, may be null) is set to 1 when an id is outside [0, V). */
int mgen_embedding_fwd(const long* ids, const float* table, float* out, long T, int d, long V, int* err_flag,
                       mgen_stream_t stream);
int mgen_embedding_bwd(const long* ids, const float* dout, float* dtable, long T, int d, long V, mgen_stream_t stream);

/* ---- GPT-2 style decoder (reference src/model/model.py:54-65 builds transformers' GPT2LMHeadModel) ---------------
 * Conv1D layers = torch.addmm(bias, x, W) with W stored [in][out] (TF pytorch_utils.py Conv1D.forward): the product
 * of mgen_gemm_bf16 with transB=1 plus a bf16 bias row [N] added to the fp32 accumulators before the output rounding
 * (alpha = 1; out_mode 0, 1, 3 or 4). */
int mgen_gemm_bf16_bias(const void* A, const void* B, void* C, const void* resid, const void* bias, int M, int N,
                        int K, long lda, long ldb, long ldc, long ldr, int transA, int transB, int out_mode,
                        mgen_stream_t stream);
/* nn.LayerNorm over the fp32 residual stream (TF models/gpt2/modeling_gpt2.py ln_1/ln_2/ln_f, eps 1e-5):
 * y = (x - mean) * rstd * w + b; mean[T], rstd[T] saved for backward (may be null). */
int mgen_layernorm_fwd(const float* x, const float* w, const float* b, void* y, float* mean, float* rstd, long T,
                       int d, float eps, int out_bf16, mgen_stream_t stream);
/* dres (+)= d(loss)/dx (fp32), dres_lp = the same in the dtype of dy (may be null), dw[d] += , db[d] += . */
int mgen_layernorm_bwd(const void* dy, const float* x, const float* w, const float* mean, const float* rstd,
                       float* dres, void* dres_lp, float* dw, float* db, long T, int d, int accumulate, int dy_bf16,
                       mgen_stream_t stream);
/* NewGELUActivation (TF activations.py: 0.5 x (1 + tanh(sqrt(2/pi)(x + 0.044715 x^3)))) on bf16 [n], n % 8 == 0,
 * evaluated in fp32; backward dh = dact * gelu'(h). */
int mgen_gelu_new_fwd(const void* h, void* act, long n, mgen_stream_t stream);
int mgen_gelu_new_bwd(const void* h, const void* dact, void* dh, long n, mgen_stream_t stream);
/* out[n] += sum_t x[t][n] for bf16 x [T][ld]: the bias gradient of a Conv1D from its output gradient. */
int mgen_colsum_bf16(const void* x, long ld, float* out, long T, int N, mgen_stream_t stream);
/* wte[ids[t]] + wpe[pos[t]] (GPT2Model.forward: inputs_embeds + position_embeds); pos int32 [T] or null
 * (position = t % S); *err_flag is set when an id or position is out of range. */
int mgen_embed_pos_fwd(const long* ids, const float* wte, const float* wpe, const int* pos, float* out, long T, int S,
                       int d, long V, int P, int* err_flag, mgen_stream_t stream);

/* ---- fp32 parity mode ------------------------------------------------------------------------------------------
 * The network without autocast (reference src/train_aria.py:323-331 trains in fp32; the fp32 outputs of the stock
 * modules are what the parity fixtures hold). Products run on the bf16 MFMA GEMM from split operands:
 * hi = bf16(x), lo = bf16(x - hi), x*w ~= hi*hi + hi*lo + lo*hi with fp32 accumulation. n % 4 == 0. */
int mgen_split_bf16(const float* x, void* hi, void* lo, long n, mgen_stream_t stream);
/* Qwen3MLP gate in fp32 (TF models/qwen3/modeling_qwen3.py:81-83): gu [T][2I] = [gate | up]. */
int mgen_swiglu_f32_fwd(const float* gu, float* act, long T, int I, mgen_stream_t stream);
int mgen_swiglu_f32_bwd(const float* gu, const float* dact, float* dgu, long T, int I, mgen_stream_t stream);
/* Causal attention in fp32 (eager formula, TF models/qwen3/modeling_qwen3.py:185-207) on token-major operands with
 * the strides of mgen_attn_fwd/bwd (in elements); lse, delta [B][H][S]. */
int mgen_attn_f32_fwd(const float* q, const float* k, const float* v, float* out, float* lse, long sq, long sk, long sv,
                      long so, int B, int S, int H, int KVH, int D, float scale, mgen_stream_t stream);
int mgen_attn_f32_bwd(const float* q, const float* k, const float* v, const float* out, const float* dout,
                      const float* lse, float* delta, float* dq, float* dk, float* dv, long sq, long sk, long sv, long so,
                      long sdo, long sdq, long sdk, long sdv, int B, int S, int H, int KVH, int D, float scale,
                      mgen_stream_t stream);
/* GPT-2 blocks in the fp32 parity mode (fp32 activations): gelu_new (TF activations.py NewGELUActivation) and the
 * Conv1D bias (TF pytorch_utils.py Conv1D.forward, addmm(bias, x, W)) with its gradient. n, N, ld multiples of 4. */
int mgen_gelu_new_f32_fwd(const float* h, float* act, long n, mgen_stream_t stream);
int mgen_gelu_new_f32_bwd(const float* h, const float* dact, float* dh, long n, mgen_stream_t stream);
int mgen_bias_add_f32(float* y, long ld, const float* bias, long T, int N, mgen_stream_t stream);
int mgen_colsum_f32(const float* x, long ld, float* out, long T, int N, mgen_stream_t stream);


/* ---- batch assembly on the device ---------------------------------------------------------------------------------
 * Token batches: flat = the B examples back to back, offsets[B+1]; outputs [B][S] int64 right-padded with pad_id.
 * mask_pads=1, prompt_len=null: miditok DataCollator(pad, copy_inputs_as_labels=True) (reference src/train.py:33-34):
 * labels = ids with -100 on padding, attention mask 1 on real tokens. mask_pads=0 with prompt_len[B]: the melody
 * collate_fn (reference src/train_aria.py:118-145): labels = ids with -100 on the first prompt_len[b] positions,
 * padding not masked. mask / labels may be null. */
int mgen_collate_tokens(const long* flat, const long* offsets, const int* prompt_len, long* ids, long* mask,
                        long* labels, int B, int S, long pad_id, int mask_pads, mgen_stream_t stream);
/* 4-D position batches, MidiDataset4D._create_chunks (reference src/train_model_new.py:114-158): rows fp32 [total][4];
 * batch entry b takes chunk chunk_ids[b] of S rows, padded with [last start, 0, 2, 0]; labels = next row with the
 * start difference in column 0 (last label: the last row, difference 0), -100 on the padded tail. */
/* The attention_mask of the forward call surface (reference src/model/model.py:142-143; the collator of
 * src/train.py:33-35 pads on the right): sets bit 1 (value 2) of *err_flag if a row of mask [B][S] (int64) has a 0
 * followed by a non-zero, i.e. is anything but right padding, the one form under which causal attention needs no
 * mask for the real tokens. */
int mgen_check_right_padding(const long* mask, int B, int S, int* err_flag, mgen_stream_t stream);

int mgen_collate_pos4d(const float* rows, long total, const long* chunk_ids, float* pos, float* labels, long* mask,
                       long* ids, int B, int S, mgen_stream_t stream);

/* ---- generated note rows -> MIDI events (notes_to_midi, reference generate_music.py:62-105) -----------------------
 * rows fp32 [n][4] = (start s, duration s, pitch, velocity); ticks_per_second = tempo / 60 * ticks_per_beat.
 * ticks[2i] = int(start * tps), ticks[2i+1] = int((start + duration) * tps) (fp32 arithmetic, truncation);
 * msgs[2i] = 0x90 | pitch << 8 | velocity << 16, msgs[2i+1] = 0x80 | pitch << 8, pitch/velocity clipped to 0..127. */
int mgen_notes_to_events(const float* rows, long n, float ticks_per_second, int* ticks, int* msgs,
                         mgen_stream_t stream);

/* dtype conversion: kind 0 fp32 -> bf16, 1 bf16 -> fp32. */
int mgen_cast(const void* in, void* out, long n, int kind, mgen_stream_t stream);
/* Weight gradients of up to four linear layers that share the token range, as ONE launch: C_i [M_i][N_i] (fp32, added in
 * place) += A_i^T B_i with A_i = dY_i stored [K][M_i], B_i = X_i stored [K][N_i] (host arrays of n pointers / sizes).
 * Replaces n grad_weight products of autograd's mm backward (grad_output^T @ input of nn.Linear, reached from
 * models/qwen3/modeling_qwen3.py:76-83,225-236); numbering the tiles of all products through means a quarter of the
 * reduction slices (and fp32 atomics) that n separate calls need. Each product must satisfy what the
 * output-stationary kernel needs (K a multiple of 32 and >= 2048, M_i, N_i >= 256 and multiples of 8). */
int mgen_gemm_bf16_wgrad_group(int n, const void* const* A, const void* const* B, float* const* C, const int* M,
                               const int* N, int K, const long* lda, const long* ldb, const long* ldc,
                               mgen_stream_t stream);

/* out[b] [C][R] = in[b] [R][C]^T for nbatch bf16 matrices bs_in / bs_out elements apart. Replaces the transposed
 * operand read inside autograd's mm backward (grad_input = grad_output @ weight, torch/csrc/autograd FunctionsManual
 * mm_mat1_backward as reached from nn.Linear in models/qwen3/modeling_qwen3.py:76-83): the host keeps transposed bf16
 * copies of the weights whose input-gradient product has a 512-long reduction, refreshed once per optimizer step. */
int mgen_transpose_bf16(const void* in, void* out, int R, int C, long ldin, long ldout, int nbatch, long bs_in,
                        long bs_out, mgen_stream_t stream);

/* ---- shifted cross entropy (ForCausalLMLoss, TF loss/loss_utils.py:49-71) -----------------------------------
 * labels int64 [B][S]; row (b,s) is scored against labels[b][s+1]. */
/* V > 0: labels outside [0, V) that are not ignore_index are left out of the count, as the loss kernels leave them
 * out of the sum, and set bit 2 (value 4) of *err_flag (may be null): torch's cross_entropy asserts on them. */
int mgen_ce_count(const long* labels, long B, int S, long ignore_index, float* inv_count, int* count, long V,
                  int* err_flag, mgen_stream_t stream);
/* rows `tok0 .. tok0+rows` of the flattened [B*S] token axis; logits [rows][ld] (bf16 or fp32) are replaced in place
 * by d(loss)/d(logits) when write_grad; row_loss[B*S] and lse_out[B*S] (may be null) are indexed by token. */
int mgen_ce_fwd_bwd(void* logits, const long* labels, long tok0, long rows, int V, long ld, int S, long ignore_index,
                    const float* inv_count, float grad_scale, float* row_loss, float* lse_out, int write_grad,
                    int is_bf16, mgen_stream_t stream);
int mgen_ce_finalize(const float* row_loss, long T, const float* inv_count, float* loss, mgen_stream_t stream);

/* lm_head + shifted cross entropy, forward and backward, with the chunk loop inside the library: per chunk of at most
 * chunk_tokens rows, logits = x W^T -> cross entropy in place -> dx = dlogits W -> dW (+)= dlogits^T x (a plain store
 * for the first chunk of the call, accumulating launches after it). Replaces `logits = self.lm_head(hidden)` +
 * `self.loss_function(...)` of Qwen3ForCausalLM.forward (TF models/qwen3/modeling_qwen3.py:495-499 ->
 * TF loss/loss_utils.py:49-71) and autograd through both, as called from /root/reference/src/model/model.py:142-143.
 * x [B*S][ldx] bf16, W [V][ldw] bf16, labels int64 [B][S]; loss fp32 scalar (device); with_grad: dx [B*S][lddx] bf16 and
 * (unless null) dW [V][d] fp32, for an upstream gradient of 1. workspace: mgen_lmhead_ce_workspace_bytes(B*S, V,
 * chunk_tokens) bytes, 256-byte aligned (one chunk of logits, the per-row losses, the label count). */
long mgen_lmhead_ce_workspace_bytes(long T, int V, long chunk_tokens);
int mgen_lmhead_ce_fwd_bwd(const void* x, long ldx, const void* W, long ldw, const long* labels, long B, int S, int V,
                           int d, long ignore_index, long chunk_tokens, void* workspace, long workspace_bytes,
                           float* loss, void* dx, long lddx, float* dW, int with_grad, int* err_flag,
                           mgen_stream_t stream);

/* ---- optimizer -----------------------------------------------------------------------------------------------
 * clip_grad_norm_ (reference src/train.py:66): out[0] = ||grad||_2, out[1] = min(1, max_norm/(norm+1e-6)).
 * workspace: mgen_sumsq_partial_count() doubles. */
int mgen_sumsq_partial_count(void);
int mgen_grad_norm_clip(const float* grad, long n, float max_norm, double* workspace, float* out,
                        mgen_stream_t stream);
/* The same result with NO workspace argument and no allocation (no counterpart in the reference): for callers that
 * cannot allocate device memory, i.e. code under HIP graph capture, tight serving loops and harnesses that forbid
 * hipMalloc. The partial sums live in an 8 KiB array inside the code object, one copy per GPU for the life of the
 * process. n <= 2^36 elements (larger n is rejected). Calls that overlap on one GPU race: serialise them. */
int mgen_grad_norm_clip_static(const float* grad, long n, float max_norm, float* out, mgen_stream_t stream);
/* torch.optim.AdamW step (reference src/model/model.py:19-25) over n flat fp32 elements; grad is scaled by clip[1]
 * when clip is given; shadow_bf16 (may be null) receives the rounded new parameters. step is 1-based. */
int mgen_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow_bf16, long n,
                    const float* clip, float lr, float beta1, float beta2, float eps, float wd, long step,
                    mgen_stream_t stream);
int mgen_scale_f32(float* x, long n, float s, const float* a_dev, mgen_stream_t stream);
/* y += a * (*a_dev) * x ; x *= a * (*a_dev)  (a_dev may be null): the upstream loss gradient stays on the device. */
int mgen_axpy_f32(float* y, const float* x, long n, float a, const float* a_dev, mgen_stream_t stream);
int mgen_scale_bf16(void* x, long n, float a, const float* a_dev, mgen_stream_t stream);

/* Projections of the one-token decode step (few token rows M, small weights): C[M, N] = A[M, K] W[N, K]^T with one
 * memory round trip per 512 reduction indices and workgroup (32 x 64 x 512 tiles, all operand loads of a chunk
 * requested up front; a workgroup small enough to share a CU with the attention kernel of another batch part).
 * slabs == 0: bf16 C [M][ldc], the workgroup walks the whole reduction. slabs = s > 0 (s divides K / 512): s fp32
 * partial products over K / s reduction indices each, slab_stride elements apart (summed by mgen_slab_resid_rmsnorm);
 * mgen_gemm_skinny_slabs(M, N, K) is the count mgen_decode_layer uses (a workgroup for every CU, no more slabs than
 * that needs). a_swiglu != 0: A is gu [M][lda >= 2K] and the operand is silu(gate) * up (Qwen3MLP.forward, TF
 * models/qwen3/modeling_qwen3.py:81-83), formed while the tile is loaded.
 * mgen_gemm_skinny_fits returns 1 for the shapes the kernel takes (N % 64 == 0, K % 512 == 0, M <= 1024, ...).
 * Replaces the nn.Linear calls of a decoder layer at q_len = 1 (TF models/qwen3/modeling_qwen3.py:252-254, 279). */
int mgen_gemm_skinny_fits(int M, int N, int K, long lda, long ldw, long ldc, int slabs);
int mgen_gemm_skinny_slabs(int M, int N, int K);
int mgen_gemm_bf16_skinny(const void* A, const void* W, void* C, int M, int N, int K, long lda, long ldw, long ldc,
                          int slabs, long slab_stride, int a_swiglu, mgen_stream_t stream);

/* ---- KV-cache decode ----------------------------------------------------------------------------------------
 * The cache is pre-allocated [B][KVH][max_len][D] bf16 per layer (the reference grows it with torch.cat per token:
 * DynamicLayer.update, TF cache_utils.py:127-146). mgen_kv_append copies n_new tokens per stream from token-major
 * projection outputs (k_src/v_src [B*n_new][ld], heads at h*D) to positions pos0[b] .. pos0[b]+n_new-1. */
int mgen_kv_append(const void* k_src, const void* v_src, long ld_k, long ld_v, void* k_cache, void* v_cache,
                   const int* pos0, int B, int n_new, int KVH, int D, int max_len, mgen_stream_t stream);
/* One decode step's projection rows qkv [B][ld_in] (one new token per stream: HQ query heads, HK key heads, HK value
 * heads) in one launch: q/k per-head RMSNorm (wq/wk, may both be null) + rotation by the table row pos[b]
 * (TF models/qwen3/modeling_qwen3.py:252-257), q to qout [B][ld_q], k and v into cache slot pos[b] of their stream
 * (TF cache_utils.py:127-146). Same bits as mgen_qknorm_rope[_il]_fwd followed by mgen_kv_append. */
int mgen_decode_qkv_prep(const void* qkv, long ld_in, void* qout, long ld_q, void* k_cache, void* v_cache,
                         const float* wq, const float* wk, const float* cos_sin, const int* pos, int B, int HQ, int HK,
                         int D, int max_len, float eps, int interleaved, mgen_stream_t stream);
/* single-query attention over the cache (TF models/qwen3/modeling_qwen3.py:259-279 with q_len = 1): q [B][ld_q],
 * out [B][ld_o] bf16, kv_len[b] keys per stream; nsplit > 1 splits the keys over workgroups and needs
 * mgen_attn_decode_workspace(B, H, D, nsplit) floats of workspace. */
long mgen_attn_decode_workspace(int B, int H, int D, int nsplit);
int mgen_attn_decode(const void* q, long ld_q, const void* k_cache, const void* v_cache, const int* kv_len, void* out,
                     long ld_o, float* workspace, int B, int H, int KVH, int D, int max_len, int nsplit, float scale,
                     mgen_stream_t stream);

/* One decoder layer of the one-token decode step in one call (seven launches issued by the library): QKV projection,
 * mgen_attn_decode_prep, o_proj slabs, slab sum + residual + post-attention norm, gate/up projection, down projection
 * with the gated operand, slab sum + residual + the norm that follows the layer. x (bf16 [B][d]) holds the layer's
 * normed input on entry and the next norm's output on return; h_in / h_mid / h_out are the fp32 residual stream before,
 * inside and after the layer; qkv, att, gu, slabs ([max(H*D, I)/512][B][d] fp32) are scratch. For RMSNorm blocks whose
 * four projections fit mgen_gemm_bf16_skinny (mgen_decode_layer_fits). Replaces Qwen3DecoderLayer.forward at
 * q_len = 1 with a cache (TF models/qwen3/modeling_qwen3.py:294-323, cache_utils.py:127-146). */
int mgen_decode_layer_fits(int B, int d, int H, int KVH, int D, int I);
int mgen_decode_layer(void* x, const float* h_in, float* h_mid, float* h_out, void* qkv, void* att, void* gu,
                      float* slabs, float* attn_workspace, const void* w_qkv, const float* w_qn, const float* w_kn,
                      const void* w_o, const float* w_ln2, const void* w_gu, const void* w_down, const float* w_next,
                      const float* cos_sin, const int* pos, const int* kv_len, void* k_cache, void* v_cache, int B,
                      int d, int H, int KVH, int D, int I, int max_len, int nsplit, float eps, int interleaved,
                      mgen_stream_t stream);
/* mgen_decode_qkv_prep + mgen_attn_decode in one launch: q, the new key and the new value are taken from the raw
 * projection rows qkv [B][ld_in] inside the attention kernel (same bits as the two calls). kv_len[b] counts the new
 * token; the caches are read up to kv_len[b] - 1 keys and receive the new rows at slot pos[b]. */
int mgen_attn_decode_prep(const void* qkv, long ld_in, const float* wq, const float* wk, const float* cos_sin,
                          const int* pos, float eps, int interleaved, void* k_cache, void* v_cache, const int* kv_len,
                          void* out, long ld_o, float* workspace, int B, int H, int KVH, int D, int max_len, int nsplit,
                          float scale, mgen_stream_t stream);

/* ---- sampler -------------------------------------------------------------------------------------------------
 * GenerationMixin._sample token selection (TF generation/utils.py:2894-2925) with the warpers
 * TF generation/logits_process.py:301-303 (temperature), :590-595 (top-k), :527-539 (top-p), in that order.
 * logits [B][ld] bf16 or fp32; do_sample = 0 -> argmax. The multinomial draw is argmax(p_i / q_i) over explicit
 * Exp(1) noise: noise [B][ld_noise] fp32, or null = Philox4x32-10(seed, offset), which mgen_philox_exponential
 * writes out for checking. out[B] int64; out2 (may be null) gets the same ids at out2[b * out2_stride]. */
int mgen_sample(const void* logits, long ld, int B, int V, int is_bf16, int do_sample, float temperature, int top_k,
                float top_p, const float* noise, long ld_noise, unsigned long long seed, unsigned long long offset,
                long* out, long* out2, long out2_stride, mgen_stream_t stream);
/* mgen_sample on rows [row0, row0 + B) of a larger batch: the Philox stream of logits row i is the one row row0 + i
 * of a call on the whole batch would use (decoding a batch in independent parts, TF generation/utils.py:2921-2923). */
int mgen_sample_rows(const void* logits, long ld, int B, int V, int is_bf16, int do_sample, float temperature,
                     int top_k, float top_p, const float* noise, long ld_noise, unsigned long long seed,
                     unsigned long long offset, int row0, long* out, long* out2, long out2_stride,
                     mgen_stream_t stream);
/* mgen_sample_rows for a decode loop whose launch arguments stay the same from step to step (HIP graph replay):
 * step (device int, may be null = 0) is added to the Philox offset and to the out2 column when the kernel runs;
 * unfinished ([B] int64, may be null) carries the EOS bookkeeping of GenerationMixin._sample (TF
 * generation/utils.py:2919-2931): a finished stream receives pad_id, a stream that draws eos_id is finished from the
 * next step on. mgen_decode_advance ends such a step: pos[i] += 1, kv_len[i] += 1 for the n streams, step[0] += 1
 * (the cache positions mgen_attn_decode_prep reads; TF cache_utils.py:127-146 advances them on the host). */
int mgen_sample_step(const void* logits, long ld, int B, int V, int is_bf16, int do_sample, float temperature,
                     int top_k, float top_p, const float* noise, long ld_noise, unsigned long long seed,
                     unsigned long long offset, int row0, long* out, long* out2, long out2_stride, const int* step,
                     long* unfinished, long eos_id, long pad_id, mgen_stream_t stream);
int mgen_decode_advance(int* pos, int* kv_len, int n, int* step, mgen_stream_t stream);
int mgen_philox_exponential(float* out, long ld, int B, int V, unsigned long long seed, unsigned long long offset,
                            mgen_stream_t stream);

/* ---- data-parallel gradient exchange over RCCL -----------------------------------------------------------------
 * What lightning's DDP strategy does behind lightning.Trainer(...) (/root/reference/src/train.py:63-72,
 * src/train_aria.py:322-336: torch DistributedDataParallel all-reduces gradient buckets during backward and divides by
 * the world size). One communicator per process (one process per GPU). RCCL is opened with dlopen on first use.
 *   mgen_comm_unique_id     rank 0 makes the 128-byte id; every rank needs the same bytes (hand them over out of band)
 *   mgen_comm_init          joins the communicator of `world` ranks on `device`
 *   mgen_comm_allreduce_bucket  in-place fp32 mean (average != 0) or sum of buf[0..n) over the ranks, enqueued on the
 *                           communicator's own stream behind what compute_stream holds now; returns at once
 *   mgen_comm_wait          makes `stream` wait for every bucket enqueued so far; optional outputs: bytes and buckets
 *                           reduced since the previous wait
 *                           reduced since the previous wait. Both calls first ask RCCL for an asynchronous error of the
 *                           communicator (a rank that died, a link failure): then the communicator is aborted, which
 *                           unblocks the collectives still queued, and the call fails instead of hanging
 *   mgen_comm_abort         gives up on the communicator (ncclCommAbort); later calls on it fail
 *   mgen_comm_rccl_version  the RCCL version code (e.g. 22205) */
typedef struct mgen_comm mgen_comm_t;
int mgen_comm_unique_id(void* id128);
int mgen_comm_rccl_version(int* version);
int mgen_comm_init(mgen_comm_t** comm, const void* id128, int rank, int world, int device);
int mgen_comm_allreduce_bucket(mgen_comm_t* comm, float* buf, long n, int average, mgen_stream_t compute_stream);
int mgen_comm_wait(mgen_comm_t* comm, mgen_stream_t stream, long* bytes, int* buckets);
int mgen_comm_abort(mgen_comm_t* comm);
int mgen_comm_destroy(mgen_comm_t* comm);

#ifdef __cplusplus
}
#endif
#endif /* MGEN_H */
