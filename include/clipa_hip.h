/* clipa_hip.h - C ABI of libclipa_hip.so, the MI355X (gfx950) compute engine behind the open_clip
 * model/loss API of UCSC-VLAA/CLIPA (clipa_torch).
 *
 * The reference has no FFI: its hot path is torch ops called from open_clip/{transformer,model,loss}.py.
 * Each entry point below replaces the torch op(s) at the cited reference lines (paths relative to
 * /root/reference/clipa_torch).  Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch-allocated); no hidden
 *     allocations, no implicit synchronisation; `stream` is a hipStream_t;
 *   - "bf16" buffers are raw uint16 bfloat16; matrices are row-major with element strides `ld*`;
 *   - return 0 on success, negative on error (never throws); clipa_last_error() gives the message;
 *   - re-entrant: may be called from PyTorch's autograd worker thread.
 */
#ifndef CLIPA_HIP_H
#define CLIPA_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* epilogues of clipa_gemm_nt */
#define CLIPA_EPI_NONE 0 /* C = bf16(alpha*acc + bias)                                        */
#define CLIPA_EPI_ACT 1  /* C = act(v), optional C2 = v (pre-activation, for the backward)    */
#define CLIPA_EPI_ADD 2  /* C = v + aux            (residual add, transformer.py:248-249)     */
#define CLIPA_EPI_DACT 3 /* C = v * act'(aux)      (activation backward)                      */
/* MI355X engine extensions (no reference counterpart): the pre-activation kept as OCP e4m3 bytes (saturating, no scale) - half
 * the HBM of the bf16 copy.  Whole-tile shapes only (M, N multiples of 256, K multiple of 128, K >= 256): other shapes return
 * CLIPA_ERR_ARG and the caller composes CLIPA_EPI_ACT + clipa_cast_bf16_to_e4m3 / clipa_cast_e4m3_to_bf16 + CLIPA_EPI_DACT. */
#define CLIPA_EPI_ACT_PRE8 4 /* C = act(v), C2 = e4m3(v): uint8 [M, ldc]                      */
#define CLIPA_EPI_DACT8 5    /* C = v * act'(aux), aux = e4m3 bytes, uint8 [M, ldaux]; optional C2 = act(aux), bf16 [M, ldc]:
                              * what clipa_activation_fwd_e4m3 writes for the same bytes (the c_proj weight gradient's operand) */
/* activations: nn.GELU(approximate='none'|'tanh') (model.py:128-129), QuickGELU (transformer.py:37-40) */
#define CLIPA_ACT_GELU_ERF 0
#define CLIPA_ACT_GELU_TANH 1
#define CLIPA_ACT_QUICK_GELU 2
/* input dtypes of clipa_patchify */
#define CLIPA_DT_U8 0
#define CLIPA_DT_BF16 1
#define CLIPA_DT_F32 2
/* pooling modes (transformer.py:472-478, model.py:254-260) */
#define CLIPA_POOL_FIRST 0      /* x[:,0]  (cls token / big_vision_tok)          */
#define CLIPA_POOL_LAST 1       /* x[:,-1] (big_vision_last)                      */
#define CLIPA_POOL_INDEX 2      /* x[b, idx[b]] (EOT row, text.argmax(-1))        */
#define CLIPA_POOL_MEAN_ALL 3   /* x.mean(1) incl. cls (open_clip GAP)            */
#define CLIPA_POOL_MEAN_PATCH 4 /* x[:,1:].mean(1) (big_vision_gap)               */

const char* clipa_last_error(void);
int clipa_version(void);

/* C[M,N] = epi(alpha * A[M,K] . B[N,K]^T + bias[N]); A,B bf16; C bf16 (or f32 when out_f32, epi NONE).
 * Replaces nn.Linear / packed in-proj / out-proj / conv1-as-GEMM / `@ proj` forward and, with B = W^T,
 * their input gradients: transformer.py:209,217-219,234,371,491,528-529; model.py:254; loss.py:135-142.
 * K % 8 == 0; N, lda, ldb, ldc, ldaux % 8 == 0. */
int clipa_gemm_nt(const void* A, const void* B, void* C, void* C2, const float* bias, const void* aux,
                  int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldaux,
                  float alpha, int epi, int act, int out_f32, void* stream);

/* e4m3 pre-activations ("light8" keep tier): bf16 -> saturating OCP e4m3 bytes and back (exact), and act(e4m3 x) -> bf16
 * (the re-materialisation of the MLP activation in backward: transformer.py:217-219 `gelu`). */
int clipa_cast_bf16_to_e4m3(const void* in, void* out, int64_t n, void* stream);
int clipa_cast_e4m3_to_bf16(const void* in, void* out, int64_t n, void* stream);
int clipa_activation_fwd_e4m3(const void* x8, void* out, int64_t n, int act, void* stream);

/* out[R,C] = sum_m P[m,R] * Q[m,C]  (weight gradients dW = dY^T . X of the same layers; also the
 * gathered-feature gradients of loss.py:135-139).  out is f32 or bf16. workspace: split-M partial slabs. */
int64_t clipa_gemm_tn_workspace(int64_t M, int64_t R, int64_t C, int64_t* nslices);
/* colsum_out (optional, f32 [R]): column sums of P = the bias gradient of the same layer, fused into the pass. */
int clipa_gemm_tn(const void* P, const void* Q, void* out, float* colsum_out, int64_t M, int64_t R, int64_t C,
                  int64_t ldp, int64_t ldq, int out_bf16, void* workspace, int64_t workspace_bytes, void* stream);

/* fp8 path (BASELINE.json configs[3]: "fp8 MFMA weights/activations"; the reference's precision menu,
 * training/params.py:195-200, stops at bf16, so these have no reference counterpart - same call sites as clipa_gemm_nt:
 * the linear layers of a residual block, transformer.py:209,217-219,234, forward and input gradient).
 * clipa_quantize_rows: q[r,:] = fp8(x[r,:] * FMAX / max|x[r,:]|) (x bf16, q bytes; fmt 0 = OCP e4m3, FMAX 448; 1 = e5m2,
 * FMAX 57344), dq[r] = max|x[r,:]| / FMAX (0 for an all-zero row).  K % 8 == 0, K <= 8192.
 * clipa_gemm_nt_f8: C = epi(alpha * scale_a[m] * scale_b[n] * A8 . B8^T + bias[n]) on v_mfma_f32_16x16x128_f8f6f4, fp32
 * accumulation, bf16 C / C2 / aux and the epilogues of clipa_gemm_nt; scale_a [M], scale_b [N] may be NULL (= 1).  Whole-tile
 * shapes (M, N % 256 == 0, K % 256 == 0, K >= 512, e4m3 weights) run on the four-wave kernel of gemm_f8a.hip, bit-identical.
 * K, lda, ldb % 16 == 0 (bytes); N, ldc, ldaux % 8 == 0.  CLIPA_EPI_ACT_PRE8 / CLIPA_EPI_DACT8 (e4m3 pre-activation copy / operand, as in
 * clipa_gemm_nt; round 6) exist on the four-wave kernel only: whole-tile shapes, else CLIPA_ERR_ARG (compose GEMM + cast).
 * clipa_layernorm_fwd_q8: LayerNorm of bf16 rows emitting the e4m3 operand of the next GEMM (q, dq as quantize_rows of
 * the bf16-rounded output) and, when y != NULL, the bf16 output itself. */
int clipa_quantize_rows(const void* x, void* q, float* dq, int64_t rows, int64_t K, int64_t ldx, int64_t ldq, int fmt,
                        void* stream);
int clipa_gemm_nt_f8(const void* A8, const void* B8, const float* scale_a, const float* scale_b, void* C, void* C2,
                     const float* bias, const void* aux, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                     int64_t ldc, int64_t ldaux, float alpha, int epi, int act, int fmt_a, int fmt_b, void* stream);
int clipa_layernorm_fwd_q8(const void* x, const float* gamma, const float* beta, void* y, void* q, float* dq,
                           int64_t rows, int64_t D, float eps, void* stream);
/* Producer-fused quantisation (round 6): the fp8 operand of the next GEMM written by the kernel that produces it, with a row scale
 * the caller PREDICTS (the row maximum is spread over a row's N tiles): |sum_k a[m,k] w[n,k]| <= ||a[m,:]|| * max_n ||w[n,:]||.
 * clipa_layernorm_fwd_q8n: clipa_layernorm_fwd_q8 that also returns rownorm[r] = ||LayerNorm(x)[r,:]||_2 (bf16-rounded values).
 * clipa_gemm_nt_f8q: clipa_gemm_nt_f8 whose output C8 is e4m3 bytes (row stride ldc BYTES): row m = the bf16-rounded result times
 * scale_out[m], saturating at +-448.  epi: CLIPA_EPI_ACT (C2 NULL or the bf16 pre-activation copy), CLIPA_EPI_ACT_PRE8 (C2 = e4m3
 * pre-activation copy), CLIPA_EPI_DACT8 (aux = e4m3 bytes; colsum_partial [M / 128][N] f32 receives per-128-row column sums of the
 * UNSCALED outputs: the bias gradient after clipa_reduce_partial_rows).  Whole-tile shapes, e4m3 weights; bit-identical to
 * clipa_gemm_nt_f8 followed by clipa_scale_quantize_rows(out, scale_out, t = 1).
 * clipa_reduce_partial_rows: out[k] = sum_r partial[r][k] in a fixed order (K % 4 == 0). */
int clipa_layernorm_fwd_q8n(const void* x, const float* gamma, const float* beta, void* y, void* q, float* dq, float* rownorm,
                            int64_t rows, int64_t D, float eps, void* stream);
int clipa_gemm_nt_f8q(const void* A8, const void* B8, const float* scale_a, const float* scale_b, void* C8, void* C2,
                      const float* bias, const void* aux, const float* scale_out, float* colsum_partial, int64_t M, int64_t N,
                      int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldaux, float alpha, int epi, int act, int fmt_a,
                      void* stream);
/* CLIPA_EPI_DACT8 (C = bf16((A8 . B8^T) scale_a scale_b * act'(aux8))) that also writes the activation operand of the same layer's fp8
 * weight gradient, X8[m, n] = e4m3(act(aux8[m, n]) * scale_a[m] / t_dev[0]) (uint8 [M, ldc]) - bit for bit what
 * clipa_scale_quantize_rows_e4m3(aux8, scale_a, t_dev, ..., act) writes, from the epilogue that reads aux8 anyway.  Whole tiles only. */
int clipa_gemm_nt_f8_emit(const void* A8, const void* B8, const float* scale_a, const float* scale_b, void* C, void* X8,
                          const void* aux8, const float* t_dev, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                          int64_t ldc, int64_t ldaux, int act, int fmt_a, void* stream);
int clipa_reduce_partial_rows(const float* partial, float* out, int64_t nrows, int64_t K, void* stream);
/* The predicted row scales themselves: clipa_rownorm_max: out[0] = max_r ||w[r,:]||_2 (w bf16 [rows, K], row stride ld);
 * clipa_absmax_f32: out[0] = max |v[i]|; clipa_row_bound: bound = factor * rownorm[m] * wnorm[0] + bmax[0] (bmax NULL = 0) ->
 * scale[m] = bound / 448 (the row's de-quantisation scale), inv[m] = 448 / bound (its clipa_gemm_nt_f8q scale_out); both 0 for a
 * zero row.  wnorm / bmax are device scalars, computed once per optimizer step. */
int clipa_rownorm_max(const void* w, int64_t rows, int64_t K, int64_t ld, float* out, void* stream);
int clipa_absmax_f32(const float* v, int64_t n, float* out, void* stream);
int clipa_row_bound(const float* rownorm, const float* wnorm_dev, const float* bmax_dev, float factor, float* scale, float* inv,
                    int64_t n, void* stream);

/* fp8 WEIGHT gradients (round 6; same call sites as clipa_gemm_tn: the autograd transposes of nn.Linear / in-proj / out-proj,
 * transformer.py:209,217-219,234).  The reduction of dW = dY^T . X runs over tokens, so a per-token scale cannot be factored out:
 * P8 is the row-quantised gradient the input-gradient GEMM already consumes (clipa_quantize_rows: dY[m,:] ~ ds[m] * P8[m,:]),
 * the activation operand absorbs ds before it is quantised, Q8[m,:] = e4m3(ds[m] * X[m,:] / t), t = max_m ds[m] * sx[m] with
 * sx the activation's own row scale of the forward pass (|Q8| <= 448, nothing saturates), and dW = t * P8^T . Q8.
 * clipa_rowscale_max: out[0] = max_m a[m] * b[m] (b NULL = 1; a, b >= 0) - the scalar t, kept on the device.
 * clipa_scale_quantize_rows: q[m,:] = e4m3(act(x[m,:]) * rowscale[m] / t[0]) (act -1 = none, else the MLP activation applied and
 * rounded to bf16 first: the re-materialised gelu(h) of transformer.py:217-219); x bf16, t device scalar (0 -> zeros);
 * clipa_scale_quantize_rows_e4m3: the same with x as saturating e4m3 bytes (the kept pre-activation of CLIPA_EPI_ACT_PRE8).
 * clipa_layernorm_fwd_q8s: the same for x -> LayerNorm(x) (transformer.py:19-34; bf16-rounded as clipa_layernorm_fwd writes it).
 * clipa_gemm_tn_f8: out[R,C] = alpha * alpha_dev[0] * sum_m P8[m,R] * Q8[m,C] on v_mfma_f32_16x16x128_f8f6f4, fp32 accumulation
 * in split-M slabs + a fixed-order reduce; fmt_p 0 = e4m3 / 1 = e5m2 gradient bytes, Q8 e4m3; alpha_dev may be NULL (= 1);
 * out f32 or bf16.  R, C % 8 == 0.  Whole 256 x 256 tiles of 16-byte-aligned operands (base, ldp, ldq in bytes) run on the
 * four-wave kernel of gemm_tn8.hip; rows beyond its slices, and every other shape, on a byte-gather kernel. */
/* clipa_quantize_rows_colsum: clipa_quantize_rows that also returns colsum[k] = sum_r x[r,k] (f32 [K]) - the bias gradient of the
 * layer (the column sums of dY: nn.Linear's bias, transformer.py:209,217-219) from the pass that quantises dY for the input-gradient
 * and weight-gradient GEMMs; fixed summation order; rownorm (optional, f32 [rows]) = ||x[r,:]||_2.  workspace: per-block partial rows. */
int64_t clipa_quantize_rows_colsum_workspace(int64_t rows, int64_t K);
int clipa_quantize_rows_colsum(const void* x, void* q, float* dq, float* colsum, float* rownorm, int64_t rows, int64_t K, int64_t ldx,
                               int64_t ldq, int fmt, void* workspace, int64_t workspace_bytes, void* stream);
int clipa_rowscale_max(const float* a, const float* b, int64_t n, float* out, void* stream);
int clipa_scale_quantize_rows(const void* x, const float* rowscale, const float* t_dev, void* q, int64_t rows, int64_t K,
                              int64_t ldx, int64_t ldq, int act, void* stream);
int clipa_scale_quantize_rows_e4m3(const void* x8, const float* rowscale, const float* t_dev, void* q, int64_t rows, int64_t K,
                                   int64_t ldx, int64_t ldq, int act, void* stream);
int clipa_layernorm_fwd_q8s(const void* x, const float* gamma, const float* beta, const float* rowscale, const float* t_dev,
                            void* q, int64_t rows, int64_t D, float eps, void* stream);
int64_t clipa_gemm_tn_f8_workspace(int64_t M, int64_t R, int64_t C);
int clipa_gemm_tn_f8(const void* P8, const void* Q8, void* out, int64_t M, int64_t R, int64_t C, int64_t ldp, int64_t ldq,
                     float alpha, const float* alpha_dev, int fmt_p, int out_bf16, void* workspace, int64_t workspace_bytes,
                     void* stream);

/* F.layer_norm over the last dim, eps inside sqrt, affine (transformer.py:19-34). x/dx share a dtype
 * (x_f32), y/dy share a dtype (y_f32). bwd: dx = LN'(dy) [+ dres]; dgamma, dbeta f32 [D].
 * clipa_layernorm_bwd_y: the same backward that also writes y = LayerNorm(x) (dtype of dy), bit for bit what
 * clipa_layernorm_fwd writes - the operand of the following layer's weight gradient when a block recomputes its LayerNorm
 * outputs in backward (the reference's checkpointed block re-runs ln_1 / ln_2 there, transformer.py:238-250,320-325). */
int clipa_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, int64_t rows,
                        int64_t D, float eps, int x_f32, int y_f32, void* stream);
int64_t clipa_layernorm_bwd_workspace(int64_t rows, int64_t D);
int clipa_layernorm_bwd(const void* x, const float* gamma, const void* dy, const void* dres, void* dx,
                        float* dgamma, float* dbeta, int64_t rows, int64_t D, float eps, int x_f32,
                        int y_f32, void* workspace, int64_t workspace_bytes, void* stream);
int clipa_layernorm_bwd_y(const void* x, const float* gamma, const float* beta, const void* dy, const void* dres, void* dx,
                          void* y, float* dgamma, float* dbeta, int64_t rows, int64_t D, float eps, int x_f32,
                          int y_f32, void* workspace, int64_t workspace_bytes, void* stream);
/* fp8 engine (no reference counterpart, as the clipa_quantize_rows family above): the bf16 LayerNorm backward that also hands
 * the linear layer in front of it its incoming gradient as fp8 operand - q uint8 [rows, D] (fmt 0 = e4m3, 1 = e5m2), dq f32 [rows]
 * (bit for bit what clipa_quantize_rows makes of the bf16 dx this call writes), colsum f32 [D] = sum over rows of dx (that
 * layer's bias gradient), rownorm f32 [rows] (optional: ||dx[r,:]||_2).  beta / y: NULL, or both given as in clipa_layernorm_bwd_y. */
int64_t clipa_layernorm_bwd_q8_workspace(int64_t rows, int64_t D);
int clipa_layernorm_bwd_q8(const void* x, const float* gamma, const float* beta, const void* dy, const void* dres, void* dx, void* y,
                           void* q, float* dq, float* colsum, float* rownorm, float* dgamma, float* dbeta, int64_t rows, int64_t D,
                           float eps, int fmt, void* workspace, int64_t workspace_bytes, void* stream);

/* softmax(q.k^T * scale + mask).v per (batch, head); q/k/v are column blocks of the packed projection
 * output (row stride ld_qkv), out is [B*L, H*dh] (row stride ld_o). causal = additive triu(1)*-inf mask.
 * Replaces F.scaled_dot_product_attention inside nn.MultiheadAttention (transformer.py:223-236). */
/* stats (optional in fwd, required in bwd): f32 [B*H*L][2] = (scale*log2(e)*rowmax, 1/rowsum) of the softmax. */
int clipa_attention_fwd(const void* q, const void* k, const void* v, void* out, float* stats, int64_t B,
                        int64_t H, int64_t L, int64_t dh, int64_t ld_qkv, int64_t ld_o, float scale,
                        int causal, void* stream);
int clipa_attention_bwd(const void* q, const void* k, const void* v, const void* out, const void* d_out,
                        const float* stats, void* dq, void* dk, void* dv, int64_t B, int64_t H, int64_t L,
                        int64_t dh, int64_t ld_qkv, int64_t ld_o, int64_t ld_dqkv, float scale, int causal,
                        void* stream);
/* The same kernels on PACKED variable-length sequences (MI355X engine extension: the text tower on the tokens up to each
 * caption's EOT - under the causal mask of transformer.py:618-624 no later position can reach the pooled output of
 * model.py:251-254 or receive gradient, so features, loss and gradients are unchanged).  q / k / v / out rows are token rows of
 * the packed matrix; sequence s occupies rows [seq_start[s], seq_start[s] + seq_len[s]).  One call processes the nseq
 * sequences named by seq_ids (int32 device array; NULL = sequences 0 .. nseq-1), all of length <= 32 * tiles (tiles <= 9);
 * stats: T * H (max, 1/sum) pairs.  Head dims 64 / 80. */
int clipa_attention_fwd_varlen(const void* q, const void* k, const void* v, void* out, float* stats,
                               const int32_t* seq_start, const int32_t* seq_len, const int32_t* seq_ids, int64_t nseq,
                               int64_t tiles, int64_t H, int64_t dh, int64_t ld_qkv, int64_t ld_o, float scale, int causal,
                               void* stream);
int clipa_attention_bwd_varlen(const void* q, const void* k, const void* v, const void* out, const void* d_out,
                               const float* stats, void* dq, void* dk, void* dv, const int32_t* seq_start,
                               const int32_t* seq_len, const int32_t* seq_ids, int64_t nseq, int64_t tiles, int64_t H, int64_t dh,
                               int64_t ld_qkv, int64_t ld_o, int64_t ld_dqkv, float scale, int causal, void* stream);

/* image [B,3,S,S] (or NHWC) u8/bf16/f32 -> bf16 patch matrix [B*(S/P)^2, Kp], elements in (ph,pw,c)
 * order, optional (x/255 - mean)/std (train.py:191-197 + conv1 im2col, transformer.py:371,491-493). */
int clipa_patchify(const void* img, void* out, int64_t B, int64_t S, int64_t P, int64_t Kp, int in_dtype,
                   int nhwc, int normalize, const float* mean3, const float* std3, void* stream);
/* Device-side train transform on uint8 NHWC batches (replaces the per-sample CPU work of
 * open_clip/transform.py:152-168 after the H2D copy of training/train.py:187-189): out[b] = Grayscale?(resize(crop(src[b],
 * box[b]), S x S, BICUBIC)) bit-exact with Pillow's ImagingResample / rgb2l (what torchvision's RandomResizedCrop and
 * Grayscale(3) run on PIL images).  src [B,Hs,Ws,3], out [B,S,S,3]; boxes int32 [B][4] = (top, left, height, width), sampled
 * by the caller; gray_flags uint8 [B] or NULL; err_count (device int32, may be NULL) counts rejected samples (box outside the
 * image, or a down-scale above 11x) whose output is zeros. */
int64_t clipa_resized_crop_workspace(int64_t B, int64_t Hs, int64_t S);
int clipa_resized_crop_u8(const void* src, const int32_t* boxes, const uint8_t* gray_flags, void* out, int64_t B, int64_t Hs,
                          int64_t Ws, int64_t S, void* workspace, int64_t workspace_bytes, int32_t* err_count, void* stream);
/* torchvision ColorJitter (per-sample op order[b][0..3] over {0 brightness, 1 contrast, 2 saturation, 3 hue}, factors[b][op];
 * samples with apply[b] == 0 are left alone; order == NULL skips the jitter) followed by Grayscale(3) of the samples flagged in
 * gray_flags (NULL: none), IN PLACE on uint8 [B,S,S,3] - the color_jitter / gray_scale stages of open_clip/transform.py:61-84,
 * 160-168, bit-exact with the Pillow code they run on PIL images.  workspace: 8 bytes per sample. */
int clipa_color_jitter_u8(void* img, const uint8_t* apply, const int32_t* order, const float* factors, const uint8_t* gray_flags,
                          int64_t B, int64_t S, void* workspace, int64_t workspace_bytes, void* stream);
/* cat(class_embedding) + positional_embedding (transformer.py:496-499) and its gradient. */
int clipa_assemble_tokens(const void* patch, const float* cls, const float* pos, void* tokens, int64_t B,
                          int64_t L, int64_t D, void* stream);
/* the batch sums (dcls, dpos) go through per-chunk partials in `workspace` and a fixed-order reduce: bit-reproducible */
int64_t clipa_assemble_tokens_bwd_workspace(int64_t B, int64_t L, int64_t D);
int clipa_assemble_tokens_bwd(const void* dtokens, void* dpatch, float* dcls, float* dpos, int64_t B,
                              int64_t L, int64_t D, void* workspace, int64_t workspace_bytes, void* stream);
/* token_embedding(text) + positional_embedding (model.py:245-247) and gradients (dense f32 table grad). */
int clipa_embed_tokens(const int64_t* ids, const void* table, int table_bf16, const float* pos, void* out,
                       int64_t B, int64_t T, int64_t D, int64_t vocab, int32_t* oob_count, void* stream);
/* oob_count (device int32, may be NULL): incremented once per token id outside [0, vocab) - nn.Embedding raises on
 * those; the forward substitutes row 0, the backward skips the row, the caller turns a non-zero count into an error
 * (clipa_amd.ops checks it asynchronously). */
/* The table gradient is a scatter-add on 64-bit fixed-point accumulators in `workspace` (integer atomics commute: the
 * result is bit-reproducible), converted to f32 at the end; dpos goes through clipa_assemble_tokens_bwd's ordered sums. */
int64_t clipa_embed_tokens_bwd_workspace(int64_t B, int64_t T, int64_t D, int64_t vocab, int need_table, int need_pos);
int clipa_embed_tokens_bwd(const int64_t* ids, const void* dx, float* dtable, float* dpos, int64_t B,
                           int64_t T, int64_t D, int64_t vocab, int32_t* oob_count, void* workspace,
                           int64_t workspace_bytes, void* stream);
/* text.argmax(dim=-1) (model.py:254) */
int clipa_argmax_tokens(const int64_t* ids, int32_t* out, int64_t B, int64_t T, void* stream);
/* pooling [B,L,D] bf16 -> [B,D] f32 and its gradient (writes all of dx) */
int clipa_pool_fwd(const void* x, const int32_t* idx, float* out, int64_t B, int64_t L, int64_t D, int mode,
                   void* stream);
int clipa_pool_bwd(const float* dout, const int32_t* idx, void* dx, int64_t B, int64_t L, int64_t D,
                   int mode, void* stream);
/* PatchDropout (transformer.py:53-83,501-502): the kept tokens as a row selection of the bf16 token matrix.
 * gather: out[r,:] = x[rows[r],:] (r < n_out; rows outside [0, n_src) read zeros); scatter = its backward:
 * dx[n_dst, D] = 0, dx[rows[r],:] = dy[r,:] (rows distinct).  rows: int64 on the device.  D % 8 == 0. */
int clipa_gather_rows(const void* x, const int64_t* rows, void* out, int64_t n_out, int64_t n_src, int64_t D,
                      void* stream);
int clipa_scatter_rows(const void* dy, const int64_t* rows, void* dx, int64_t n_src, int64_t n_dst, int64_t D,
                       void* stream);
/* F.normalize(x, dim=-1) (model.py:240,263); y_bf16 optional second output */
int clipa_l2norm_fwd(const float* x, float* y, void* y_bf16, float* inv_norm, int64_t rows, int64_t E,
                     float eps, void* stream);
int clipa_l2norm_bwd(const float* y, const float* inv_norm, const float* dy, float* dx, int64_t rows,
                     int64_t E, void* stream);
/* bias gradients: out[n] = sum_m dY[m,n] */
int64_t clipa_colsum_workspace(int64_t M, int64_t N);
int clipa_colsum(const void* dy, float* out, int64_t M, int64_t N, int64_t ld, void* workspace,
                 int64_t workspace_bytes, void* stream);
/* dtype / layout plumbing for weights */
int clipa_cast_to_bf16(const void* in, int in_f32, void* out, int64_t n, void* stream);
int clipa_cast_bf16_to_f32(const void* in, float* out, int64_t n, void* stream);
int clipa_transpose_to_bf16(const void* in, int in_f32, void* out, int64_t R, int64_t C, int64_t ldi,
                            int64_t ldo, void* stream);
/* out = act(in) elementwise, bf16 (MLP activation re-materialised from the kept pre-activation) */
int clipa_activation_fwd(const void* in, void* out, int64_t n, int act, void* stream);
/* Fused similarity + cross-entropy of the InfoNCE loss (loss.py:128-155): logits = s * rows . cols^T (rows [R,E],
 * cols [N,E] bf16, s = exp(logit_scale) read from DEVICE memory, NULL = 1), labels label0 + r.  The [R,N] fp32 logits are
 * never written: the forward GEMM's epilogue keeps per-tile (max, sum exp) partials, merged into lse[R] and
 * loss_rows[R] = lse - logit[label]; the backward re-runs the GEMM and writes the bf16 gradient
 * d loss / d (rows.cols^T) = s * gscale * (softmax - onehot) ([R, ldd], ldd >= N rounded up to 8, pad columns zero) plus
 * dscale_rows[R] = per-row d loss / d s.  Workspace: clipa_simce_workspace(R, N) bytes for either call. */
int64_t clipa_simce_workspace(int64_t R, int64_t N);
int clipa_simce_fwd(const void* rows, const void* cols, int64_t R, int64_t N, int64_t E, int64_t lda, int64_t ldb,
                    const float* scale, int64_t label0, float* lse, float* loss_rows, void* workspace,
                    int64_t workspace_bytes, void* stream);
int clipa_simce_bwd(const void* rows, const void* cols, int64_t R, int64_t N, int64_t E, int64_t lda, int64_t ldb,
                    const float* scale, int64_t label0, float gscale, const float* lse, void* dlogits_bf16,
                    int64_t ldd, float* dscale_rows, void* workspace, int64_t workspace_bytes, void* stream);
int clipa_sum_scale(const float* in, float* out, int64_t n, float scale, int accumulate, void* stream);

/* AdamW over one flat tensor (training/main.py:318-326 torch.optim.AdamW + train.py:285-286 clamp is
 * done by the caller): p -= lr*(m_hat/(sqrt(v_hat)+eps) + wd*p). param/grad bf16 or f32; m, v f32. */
int clipa_adamw(void* param, const void* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int param_f32,
                int grad_f32, float lr, float beta1, float beta2, float eps, float weight_decay,
                int64_t step, float grad_scale, void* stream);
/* The same update over `count` tensors that share dtypes, hyper-parameters and step (one parameter group of
 * main.py:311-326): HOST arrays of device pointers / element counts; a few launches instead of one per tensor.
 * Fused optimizer tail (train.py:270-286): grad_scale_dev (optional DEVICE float, e.g. the clip coefficient of
 * clipa_clip_coef) multiplies every gradient; tensor `clamp_index` (-1: none) is clamped to [clamp_lo, clamp_hi]
 * after its update (logit_scale.clamp_(0, ln 100)). */
int clipa_adamw_multi(void* const* params, const void* const* grads, float* const* exp_avg,
                      float* const* exp_avg_sq, const int64_t* numel, int count, int param_f32, int grad_f32,
                      float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                      float grad_scale, const float* grad_scale_dev, int clamp_index, float clamp_lo, float clamp_hi,
                      void* stream);
/* torch.nn.utils.clip_grad_norm_ (train.py:270-277) without a host round trip: acc[0] += sum of squares of the listed
 * gradients (zero it first; call once per dtype bucket), then coef = min(1, max_norm / (sqrt(acc) + 1e-6)) and the
 * norm itself land in device memory for clipa_adamw_multi's grad_scale_dev.  `partials` is caller-provided scratch of at
 * least sum_i ceil(numel[i] / 4096) floats: block partials are summed in a fixed order (no float atomics - the coefficient is
 * bit-reproducible run to run). */
int clipa_grad_sqnorm_multi(const void* const* grads, const int64_t* numel, int count, int grad_f32, float* acc,
                            float* partials, int64_t partials_cap, void* stream);
int clipa_clip_coef(const float* acc, float max_norm, float* norm_out, float* coef_out, void* stream);
/* Sharded gradient exchange (replaces the ring all-reduce of the DDP wrapper, training/main.py:292-299):
 * out[i] = scale * sum over w < W of in[w*n + i] - the local sum of the W shard pieces a rank receives in the one-hop
 * all-to-all form of a reduce-scatter (scale = 1/W: DDP's gradient average).  bf16 or f32 in / out, fp32 sums in rank
 * order; n % 8 == 0. */
int clipa_reduce_shards(const void* in, void* out, int64_t n, int W, int in_f32, int out_f32, float scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif
