/*
 * halo.h -- C ABI of libhalo.so, the MI355X (gfx950) engine for haloop's acoustic hot path.
 *
 * The reference (proger/haloop, /root/reference) has no FFI: its "boundary" is the Python
 * surface of ha.rnn / ha.recognizer / ha.ctc / ha.beam, under which it calls stock torch
 * operators (cuDNN LSTM, ATen ctc_loss, ...).  Each entry point below replaces one of those
 * operator call sites; the citation after "replaces:" is the reference line that makes the call.
 * The modules under haloop_amd/ are the binding (ctypes) that keeps the reference's Python signatures.
 *
 * Conventions
 *   - plain pointers are DEVICE pointers unless a parameter is documented "host";
 *   - every buffer (outputs, reserve, workspace) is caller-allocated; *_bytes() queries sizes;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); nothing here allocates,
 *     frees, synchronises or throws, so calls are capturable in a hipGraph;
 *   - return value: HALO_OK (0) or a negative HALO_E* code; halo_strerror() names it;
 *   - threads: every switch a halo_set_* entry changes lives in a settings record ("Contexts" below); calls from different threads are
 *     independent when each thread has selected its own context (and its own scratch buffer) and works on its own stream; calls that
 *     share a record -- the process-wide default one included -- must be serialised by the caller.  The persistent LSTM recurrences need
 *     every CU of the device while they run (~0.1 ms): other work on the GPU delays them, it cannot deadlock them (every wait is bounded);
 *   - dense fp32 math (arithmetic mode f32) runs on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32 / 16x16x4_f32), i.e. k-ordered fmaf
 *     chains, no reduced precision; modes bf16x3 / bf16: halo_set_math_mode.
 */
#ifndef HALO_H
#define HALO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HALO_ABI_VERSION 19

#define HALO_OK 0
#define HALO_EINVAL (-22)    /* bad argument (null pointer, non-positive size, unsupported shape) */
#define HALO_ENOTSUP (-95)   /* shape outside what the kernels were built for */
#define HALO_ELAUNCH (-5)    /* hipLaunchKernel reported an error */

typedef void *halo_stream_t; /* hipStream_t */

int halo_abi_version(void);
/* Measurement aid (tools/pmc_calibrate.py): one launch that READS exactly `bytes` bytes of src once with a chosen access shape, so that
 * rocprofv3's FETCH_SIZE counter can be calibrated per load width (MI355X_MICROARCH.md: "other access widths are uncalibrated").
 * pattern 0: 16 B per lane (global_load_dwordx4); 1: 4 B per lane, 256 B contiguous per wave instruction; 2: 4 B per lane in 64-byte
 * segments of four rows row_bytes apart (the saved-activation loads of the persistent recurrences); 3: 16 B per lane buffer loads
 * with sc1 (their fragment loads).  sink: one float nobody reads. */
int halo_debug_read(const void *src, size_t bytes, int pattern, size_t row_bytes, float *sink, void *stream);
const char *halo_strerror(int code);
/* Device sanity for the loader: returns HALO_OK and fills arch (e.g. "gfx950") and CU count. */
int halo_device_info(int device, char *arch, int arch_len, int *cu_count);

/* Arithmetic of the dense products (large GEMMs, the recurrent LSTM GEMM, attention Q.K^T / P.V and their gradients):
 *   HALO_MATH_F32     exact-f32 MFMA (default)
 *   HALO_MATH_BF16X3  operands split into bf16 hi+lo, three bf16 MFMAs per product, fp32 accumulate
 *                     (~2^-16 relative error per product; 5.3x the f32 matrix rate): fp32-grade results
 *   HALO_MATH_BF16    operands rounded to bf16, one MFMA per product, fp32 accumulate -- the arithmetic of the reference's
 *                     `torch.autocast(dtype=bfloat16)` runs (ha/attention_loop.py:87); GEMMs and attention only, the
 *                     recurrent LSTM step keeps the split form.  State, softmax, LayerNorm, losses stay fp32 in every mode.
 * Process-wide; set it before the first call / graph capture. */
#define HALO_MATH_F32 0
#define HALO_MATH_BF16X3 1
#define HALO_MATH_BF16 2
int halo_set_math_mode(int mode);
int halo_get_math_mode(void);

/* Multi-layer LSTM schedule (HALO_MATH_BF16X3, H % 64 == 0, 2..4 layers):
 *   0 (default)  per-layer: each layer's T steps in turn, input projections as batched GEMMs;
 *   1            layer-diagonal fusion: one launch per diagonal runs step d-l of every layer l, the
 *                upper layers' input projection (forward) and input gradient (backward) folded into
 *                the step as extra contraction depth.  Same results; measured 2.7 % slower at
 *                LC-2x1024/B=64 (the batched GEMMs move W_ih more efficiently than 21 folded steps),
 *                so it is off by default. */
int halo_set_lstm_fusion(int on);

/* Optional scratch for split-K partial sums.  The library never allocates: the caller may lend it
 * one persistent device buffer (16-byte aligned; 64 MiB is plenty for this model).  With it, GEMMs
 * whose output has too few tiles to fill 256 CUs are split along K into slabs that a second launch
 * sums in a fixed order (bitwise reproducible); without it they run unsplit.  Calls that use the
 * scratch must be ordered on one stream.  Pass NULL to withdraw it. */
int halo_set_scratch(void *device_ptr, size_t bytes);

/* ------------------------------------------------------------------------------------------
 * Dropout stream.  Philox4x32-10, counter = (lo32(e>>2), hi32(e>>2), stream_id, offset),
 * key = seed, value = out[e&3], keep iff value >= uint32(p*2^32), scale 1/(1-p).
 * e is the flat element index of the tensor the mask is applied to.  Every entry point that
 * takes `offset` also takes `offset_dev`, an optional DEVICE uint32 that is added to it when the
 * kernel runs, so a captured hipGraph draws fresh masks on every replay.
 * replaces: torch's nn.Dropout RNG at ha/rnn.py:8,24, nn.LSTM(dropout=) rnn.py:11 and
 * ha/recognizer.py:41,44 (the reference stream itself is not reproducible).
 * ------------------------------------------------------------------------------------------ */
#define HALO_STREAM_SUBSAMPLE 1u
#define HALO_STREAM_CLASSIFIER 2u
#define HALO_STREAM_LSTM_LAYER0 16u /* + layer index */

/* y[i] = x[i] * mask(i);  replaces: self.dropout(features) ha/recognizer.py:44 */
int halo_dropout_fwd(const float *x, float *y, size_t n, float p, uint64_t seed, uint32_t stream_id,
                     uint32_t offset, const uint32_t *offset_dev, halo_stream_t stream);
/* *counter += 1 (the device-side step counter that offset_dev points at) */
int halo_counter_inc(uint32_t *counter, halo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Dense GEMM on the exact-f32 MFMA.  C[M,N] = opA(A)[M,K] * opB(B)[K,N]  (+ epilogue)
 *   a_kcontig = 1: A stored [M,K] (row stride lda), 0: A stored [K,M]
 *   b_kcontig = 1: B stored [N,K] (row stride ldb), 0: B stored [K,N]
 * epilogue, in this order: + bias1[n] + bias2[n] (either may be NULL); relu / tanh-GELU per flags;
 *   * dropout mask(e = m*ldc + n) if p_drop > 0; + previous C if HALO_GEMM_ACCUM.
 * replaces: the ATen/oneDNN/cuBLAS matmuls under nn.Conv1d rnn.py:22, nn.LSTM input
 * projections rnn.py:25 and nn.Linear recognizer.py:45, and their autograd backward GEMMs.
 * ------------------------------------------------------------------------------------------ */
#define HALO_GEMM_RELU 1
#define HALO_GEMM_GELU 2   /* tanh-GELU ("new_gelu", ha/attention.py:12-17), applied after the bias */
#define HALO_GEMM_ACCUM 4  /* C += result (residual connections, ha/attention.py:178-179) */
#define HALO_GEMM_GELU_ERF 8 /* exact GELU (nn.GELU() ha/transformer.py:456, F.gelu ha/conv.py:46), after the bias */
int halo_gemm_f32(int a_kcontig, int b_kcontig, int M, int N, int K, const float *A, int lda,
                  const float *B, int ldb, float *C, int ldc, const float *bias1, const float *bias2,
                  int flags, float p_drop, uint64_t seed, uint32_t stream_id, uint32_t offset,
                  const uint32_t *offset_dev, halo_stream_t stream);

/* Split-bf16 GEMM on pre-tiled operand images (the HALO_MATH_BF16X3 path, usable on its own).
 *   halo_split_image: logical X[rows][k] (fp32; memory [rows][k], or [k][rows] when src_transposed,
 *     leading dimension ld) -> image of halo_split_image_bytes(rows, k) bytes holding bf16 hi/lo
 *     parts in 128x32 tiles laid out as the MFMA fragment reads want them (zero padded).
 *   halo_gemm_split: C[M,N] = A[M,K] * B[N,K]^T from two images (fp32 accumulate, three bf16 MFMAs
 *     per product), epilogue as halo_gemm_f32.  An operand used by several GEMMs is split once.
 * replaces: the same matmuls as halo_gemm_f32. */
size_t halo_split_image_bytes(int rows, int k);
int halo_split_image(const float *src, int rows, int k, int ld, int src_transposed, void *image,
                     halo_stream_t stream);
/* F.layer_norm(x, (C,), weight, bias|NULL, eps) written straight into the split image of its rows (the A operand of the Linear
 * that follows: ln_1 -> c_attn, ln_2 -> c_fc, ha/attention.py:175-179; ln_time / ln_chan, ha/transformer.py:476,495), and
 * optionally as fp32 rows y (NULL to skip).  C % 32 == 0. */
int halo_layernorm_image(const float *x, const float *weight, const float *bias, float *y, void *image, int rows, int C,
                         float eps, halo_stream_t stream);
/* The same LayerNorm with the normalised rows as ROW-MAJOR bf16 [rows][C] (C % 8 == 0) -- the operand form halo_gemm_split_io (a_hi) and
 * halo_gemm_tn_bf16 read; y (fp32 rows) and image (the tiled image as well, C % 32 == 0) are optional: a product that stages its A operand
 * from row-major rows fetches half cache lines and runs 10-15 % slower from cold caches than from the image, so the forward
 * product of a training step takes the image and the weight-gradient product the rows, both from this one launch. */
int halo_layernorm_bf16(const float *x, const float *weight, const float *bias, float *y, void *y_bf16, void *image, int rows, int C,
                        float eps, halo_stream_t stream);
/* n fp32 values -> bf16 (n % 8 == 0, 16-byte aligned): row-major bf16 operands from fp32 tensors no launch produced as bf16 */
int halo_cast_bf16(const float *x, void *y, size_t n, halo_stream_t stream);
/* The GPT MLP's two elementwise passes with a bf16 result (n % 8 == 0, 16-byte aligned): y = gelu(a) -- the c_proj input, never needed in
 * fp32 -- and da = dy * gelu'(a) (exact: 0 tanh form new_gelu ha/attention.py:12-17, 1 erf form).  Measured against the same math in the
 * producing GEMM's epilogue (GPT training direction): the separate pass is hidden under its HBM stream, the epilogue is not. */
int halo_gelu_bf16(const float *a, void *y_bf16, size_t n, int exact, halo_stream_t stream);
int halo_gelu_bwd_bf16(const float *dy, const float *a, void *da_bf16, size_t n, int exact, halo_stream_t stream);
/* ... and with bf16 on BOTH sides (round 5: the c_fc product and the c_proj input-gradient product leave their results as row-major
 * bf16, halo_gemm_rows -- what the reference's autocast path holds there, ha/attention_loop.py:164): fp32 arithmetic on the bf16 values. */
int halo_gelu_b16(const void *a_bf16, void *y_bf16, size_t n, int exact, halo_stream_t stream);
int halo_gelu_bwd_b16(const void *dy_bf16, const void *a_bf16, void *da_bf16, size_t n, int exact, halo_stream_t stream);
/* One read of an fp32 matrix src [rows][cols] (leading dimension ld), optionally through an elementwise operator, written as
 * BOTH split images a Linear's backward consumes: image_rows = halo_split_image(value [rows][cols]) (the A operand of
 * dx = dy W, ha/attention.py:117-129,141) and image_cols = halo_split_image(value^T [cols][rows]) (the operand of dW = dy^T x);
 * either may be NULL.  The fp32 value itself is never written.  op:
 *   HALO_PAIR_COPY            value = src
 *   HALO_PAIR_GELU_TANH/_ERF  value = gelu(src)                (forward of new_gelu ha/attention.py:12-17 / nn.GELU())
 *   HALO_PAIR_GELU_*_BWD      value = src * gelu'(src2)        (src = dy, src2 = the pre-activation [rows][cols], ld2)
 * halo_cross_entropy_bwd_images: value = d loss / d logits as halo_cross_entropy_bwd writes it (same arguments), as images of
 *   [rows][V] and [V][rows] for the lm_head's two backward products (ha/attention.py:266-269); logits are left untouched. */
#define HALO_PAIR_COPY 0
#define HALO_PAIR_GELU_TANH 1
#define HALO_PAIR_GELU_ERF 2
#define HALO_PAIR_GELU_TANH_BWD 3
#define HALO_PAIR_GELU_ERF_BWD 4
int halo_image_pair(const float *src, const float *src2, int rows, int cols, long ld, long ld2, int op, void *image_rows,
                    void *image_cols, halo_stream_t stream);
int halo_cross_entropy_bwd_images(const float *logits, const int64_t *targets, const float *lse, const float *grad,
                                  long grad_stride, int rows, int V, long ld, long ignore_index, void *image_rows,
                                  void *image_cols, halo_stream_t stream);
/* halo_image_pairs: HALO_PAIR_COPY image pairs of n matrices (src[i] [rows[i]][cols[i]], row stride ld[i]) in ceil(n / 6) launches
 * instead of n: the weights of a transformer block, whose forward and input-gradient products read one image each
 * (ha/attention.py:117-129,141-143), are split once per optimizer step. */
int halo_image_pairs(int n, const float *const *src, const int *rows, const int *cols, const long *ld, void *const *image_rows,
                     void *const *image_cols, halo_stream_t stream);
int halo_gemm_split(const void *a_image, const void *b_image, int M, int N, int K, float *C, int ldc,
                    const float *bias1, const float *bias2, int flags, float p_drop, uint64_t seed,
                    uint32_t stream_id, uint32_t offset, const uint32_t *offset_dev, halo_stream_t stream);

/* halo_gemm_split with the addend of the residual connection read from its own buffer: C = epilogue(A B^T) + residual (row stride
 * ldr), C need not hold x beforehand -- x1 = x0 + c_proj(y) (ha/attention.py:178-179) without first copying x0 into the output. */
int halo_gemm_split_residual(const void *a_image, const void *b_image, int M, int N, int K, float *C, int ldc, const float *residual,
                             int ldr, const float *bias1, const float *bias2, int flags, float p_drop, uint64_t seed,
                             uint32_t stream_id, uint32_t offset, const uint32_t *offset_dev, halo_stream_t stream);

/* C [M][N] (+)= A^T B with A [K][M] (leading dimension lda) and B [K][N] (ldb) ROW-MAJOR bf16, the contraction index as their row:
 * the weight gradient dW = dy^T x of a Linear (ha/attention.py's nn.Linear layers under autograd) straight from the row-major bf16
 * activations and output gradients the neighbouring launches wrote -- the operands are transposed on their way from LDS into the MFMA
 * (ds_read_b64_tr_b16), no transposed operand image is built.  Single-pass bf16 operands, fp32 accumulation; K % 32 == 0, M % 8 == 0,
 * N % 8 == 0, lda % 8 == 0, ldb % 8 == 0, 16-byte aligned operands.  flags: 0 or HALO_GEMM_ACCUM (C += result).  Split-K slices go
 * through the scratch lent by halo_set_scratch. */
int halo_gemm_tn_bf16(const void *a, long lda, const void *b, long ldb, int M, int N, int K, float *C, int ldc, int flags,
                      halo_stream_t stream);
/* n <= 4 such products over the SAME contraction length K in ONE launch, each on whole-K tiles (no K-slices, no scratch, no reduce
 * launches): the four weight gradients of a GPT block -- c_attn, attention c_proj, c_fc, MLP c_proj -- all contract over the B T token
 * rows and have 36 .. 144 output tiles each; together they fill the chip.  Arrays of n entries; flags: 0 or HALO_GEMM_ACCUM for all. */
int halo_gemm_tn_bf16_group(int n, const void *const *a, const long *lda, const void *const *b, const long *ldb, const int *M, const int *N, int K,
                            float *const *C, const int *ldc, int flags, halo_stream_t stream);

/* The same grouped weight-gradient products (C_i [M_i][N_i] = A_i^T B_i, A_i [K][M_i] and B_i [K][N_i] row-major bf16, fp32 results
 * overwritten; replaces: the autograd of a GPT block's Linear layers and of the tied lm_head, ha/attention.py:117-143,228-231) on 256-row
 * tiles of whole K -- 256 x 128 or 256 x 256, picked per group so that the tiles fill whole rounds of the CUs (csrc/gemm_tn_rows.hip:
 * LDS-DMA of whole operand rows, transposing LDS reads into the MFMA; twice the arithmetic per operand byte of the 128 x 128 tiles).
 * Single-pass bf16 arithmetic only (HALO_ENOTSUP otherwise); K % 32 == 0, M_i % 8 == 0, N_i % 8 == 0, lda / ldb % 8 == 0, ldc % 4 == 0,
 * 16-byte aligned operands and results; n <= 4. */
int halo_gemm_tn_rows_supported(int n, const int *M, const int *N, int K);
/* 1 where these tiles measured faster than halo_gemm_tn_bf16_group's (at least two rounds of 256 x 256 tiles over the CUs: the lm_head) */
int halo_gemm_tn_rows_preferred(int n, const int *M, const int *N, int K);
int halo_gemm_tn_rows_group(int n, const void *const *a, const long *lda, const void *const *b, const long *ldb, const int *M, const int *N, int K,
                            float *const *C, const long *ldc, halo_stream_t stream);

/* halo_gemm_split with row-major bf16 activations on either side, so that consecutive Linears hand their activations on without an
 * operand-image pass (ha/attention.py:136-143: c_fc -> gelu -> c_proj).  A comes either from an image (a_image) or from a row-major
 * bf16 matrix a_hi [M][lda] (and a_lo, its split remainder, in bf16x3 mode; K % 32 == 0, lda % 8 == 0, 16-byte aligned): the kernel's
 * LDS staging fetches each 16-byte chunk from where the swizzled image would hold it.  The result goes to fp32 C (may be NULL when
 * out_hi is given) and / or to row-major bf16 out_hi [M][ldo] = bf16(v) (and out_lo = bf16(v - out_hi) in bf16x3 mode).
 * flags: HALO_GEMM_RELU / GELU* / ACCUM (adds residual [M][ldr], or C itself when residual is NULL).  No dropout, no split-K.
 * Combinations built: image -> bf16 (with or without activation), bf16 -> fp32 (with or without ACCUM); others HALO_ENOTSUP.  */
int halo_gemm_split_io(const void *a_image, const void *a_hi, const void *a_lo, long lda, const void *b_image, int M, int N, int K, float *C,
                       int ldc, void *out_hi, void *out_lo, long ldo, const float *residual, int ldr, const float *bias1,
                       const float *bias2, int flags, halo_stream_t stream);

/* Round 5: softmax attention forward from ROW-MAJOR bf16 q / k / v (the c_attn product's bf16 result, heads packed inside a row: head h at
 * columns [64 h, 64 h + 64)), single-pass bf16 arithmetic, head_dim 64, causal or full, no key lengths, no dropout (csrc/attn_b16.hip: K / V
 * tiles straight from bf16 rows into the LDS images, two tiles in flight).  y (fp32) and y_bf16 are both optional outputs (at least one);
 * lse optional.  HALO_ENOTSUP outside bf16 mode / head_dim 64: the callers keep halo_attention_fwd_bf16.
 * replaces: F.scaled_dot_product_attention in MonitoredSelfAttention.forward (ha/attention.py:96-129) under the reference's autocast. */
int halo_attention_fwd_b16(const void *q, long q_row_stride, long q_batch_stride, const void *k, const void *v, long kv_row_stride,
                           long kv_batch_stride, float *y, long y_row_stride, long y_batch_stride, void *y_bf16, long yb_row_stride,
                           long yb_batch_stride, float *lse, int N, int heads, int head_dim, int Tq, int Tk, int causal, halo_stream_t stream);
/* ... and its backward: q / k / v, the forward's bf16 output y and the output gradient dy all as row-major bf16 rows (y and dy with one pair
 * of strides), lse from the forward; delta [N * heads * Tq] is scratch the first sweep writes (rowsum(dy * y)) and the second reads; dq / dk /
 * dv leave as row-major bf16 (one pair of strides), the operands of the c_attn Linear's two gradient products.  Same limits as the forward. */
int halo_attention_bwd_b16(const void *q, long q_row_stride, long q_batch_stride, const void *k, const void *v, long kv_row_stride,
                           long kv_batch_stride, const void *y_bf16, const void *dy_bf16, long y_row_stride, long y_batch_stride, const float *lse,
                           float *delta, void *dq_bf16, void *dk_bf16, void *dv_bf16, long d_row_stride, long d_batch_stride, int N, int heads,
                           int head_dim, int Tq, int Tk, int causal, halo_stream_t stream);

/* Round 5: the activation-by-weight products of the GPT path on 256-row x 96 / 192 / 288-column workgroup tiles (csrc/gemm_rows.h), single-pass
 * bf16 arithmetic only (HALO_ENOTSUP in the other modes; halo_gemm_rows_supported says so up front).  C [M][N] = A [M][K] x B [N][K]^T with
 * A either a tiled image (a_image) or ROW-MAJOR bf16 a_bf16 [M][lda] (lda % 8 == 0), B a tiled image (halo_split_image of the weight, or
 * of its transpose for an input gradient), K % 32 == 0, N % 8 == 0, and exactly one result: fp32 C [M][ldc] (+ residual [M][ldr] when
 * given: x1 = x0 + c_proj(y), ha/attention.py:178-179, no copy of x0 first), or row-major bf16 out_bf16 [M][ldo] (what the next launch's
 * operand staging, the GELU passes and the attention kernels read).  The tile width is chosen per shape so that the tiles fill whole
 * rounds of the device's CUs: N = 768 -> 96 columns, 2304 -> 288, 3072 -> 192 at M = 8192.
 * replaces: F.linear in nn.Linear.forward of c_attn / c_proj / c_fc (ha/attention.py:96-144) and its input-gradient matmul under autograd. */
int halo_gemm_rows_supported(int M, int N, int K);
int halo_gemm_rows(const void *a_image, const void *a_bf16, long lda, const void *b_image, int M, int N, int K, float *C, long ldc,
                   const float *residual, long ldr, void *out_bf16, long ldo, halo_stream_t stream);
/* ... with new_gelu / F.gelu in the epilogue (exact: 0 tanh form ha/attention.py:12-17, 1 erf form): out_bf16 = gelu(bf16(A B^T)), and, when
 * pre_bf16 is given, pre_bf16 = bf16(A B^T) as well (the pre-activation a training step keeps for the backward) -- bit for bit what
 * halo_gemm_rows (bf16 result) followed by halo_gelu_b16 writes, without the pass over the [M][N] rows between them.
 * replaces: self.c_fc + new_gelu in MLP.forward (ha/attention.py:136-140). */
int halo_gemm_rows_gelu(const void *a_image, const void *a_bf16, long lda, const void *b_image, int M, int N, int K, void *out_bf16,
                        void *pre_bf16, long ldo, int exact, halo_stream_t stream);
/* ... and the lm_head product with the cross-entropy statistics in its epilogue (as halo_gemm_split_ce: loss[m] = logsumexp(logits[m, :]) -
 * logits[m, targets[m]], 0 on ignored rows; lse optional) and the logits kept as ROW-MAJOR bf16 logits_bf16 [M][ldo] (NULL: scoring,
 * nothing of size M x N is written) -- the reference's autocast lm_head output, ha/attention.py:228-231.  The statistics and the target's
 * logit are taken from the fp32 accumulators, so the loss does not see the rounding of the stored logits.
 * halo_cross_entropy_bwd_bf16 then turns the stored logits IN PLACE into (softmax - onehot(target)) * grad[m * grad_stride] (0 on
 * ignored rows; V % 8 == 0): the row-major bf16 operand of the lm_head's two gradient products (halo_gemm_tn_bf16, halo_gemm_rows).
 * replaces: F.cross_entropy(lm_head(x), targets, ignore_index=0) and its backward to the logits. */
size_t halo_gemm_rows_ce_workspace_bytes(int M, int N);
int halo_gemm_rows_ce(const void *a_image, const void *a_bf16, long lda, const void *b_image, int M, int N, int K, const int64_t *targets,
                      long ignore_index, void *workspace, float *loss, float *lse, void *logits_bf16, long ldo, halo_stream_t stream);
int halo_cross_entropy_bwd_bf16(void *logits_bf16, const int64_t *targets, const float *lse, const float *grad, long grad_stride, int rows,
                                int V, long ld, long ignore_index, halo_stream_t stream);

/* lm_head + cross-entropy without materialising the logits (ha/attention.py:228-231): the split GEMM
 * logits[M,N] = A B^T (+ bias[N]) whose epilogue reduces each 64-column strip of a row to (max, sum exp) and picks out the
 * target's logit; a second small kernel merges the strips into loss[m] = logsumexp(logits[m,:]) - logits[m, targets[m]]
 * (0 where targets[m] == ignore_index) and, when lse != NULL, the row log-sum-exp for the backward.  logits may be NULL
 * (scoring: nothing of size M x N is written) or a [M, ldc] buffer that also receives them (training keeps them for
 * halo_cross_entropy_bwd_images).  workspace: halo_gemm_split_ce_workspace_bytes(M, N).  Split modes only (HALO_EINVAL in f32). */
size_t halo_gemm_split_ce_workspace_bytes(int M, int N);
int halo_gemm_split_ce(const void *a_image, const void *b_image, int M, int N, int K, float *logits, int ldc, const float *bias,
                       const int64_t *targets, long ignore_index, void *workspace, float *loss, float *lse, halo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Conv1d stride-s subsample + relu + dropout.   replaces: ha/rnn.py:22-24
 *   x [B,T,F] (N,T,C contiguous, as the caller holds it before .mT), w [C,F,ks], bias [C]
 *   y [T',B,C] TIME-MAJOR, T' = floor((T + 2*pad - ks)/stride) + 1
 *   col: workspace [T'*B, F*ks] floats (im2col image, also the saved input for backward)
 * ------------------------------------------------------------------------------------------ */
size_t halo_subsample_col_bytes(int B, int T, int F, int ks, int stride, int pad);
int halo_subsample_fwd(const float *x, const float *w, const float *bias, float *y, float *col, int B,
                       int T, int F, int C, int ks, int stride, int pad, float p_drop, uint64_t seed,
                       uint32_t offset, const uint32_t *offset_dev, halo_stream_t stream);
/* dy [T',B,C] is the gradient w.r.t. y; y is the forward output (relu/dropout mask is read off it);
 * dpre: workspace [T'*B, C]; dw [C,F,ks], dbias [C] are overwritten.  Input gradient is not
 * produced (mel features are leaves without grad in ha/loop.py:116). */
int halo_subsample_bwd(const float *dy, const float *y, const float *col, float *dpre, float *dw,
                       float *dbias, int B, int T, int F, int C, int ks, int stride, int pad, float p_drop,
                       halo_stream_t stream);
/* The same with dy handed over as `slabs` K-slices of the product that formed it -- [slabs][T'*B][C] floats, dy = their sum in order --
 * which the kernel adds while it reads them (halo_set_lstm_dx_slabs below: the LSTM backward's input-gradient product then runs without
 * its split-K reduce launch).  slabs = 1 is halo_subsample_bwd.  The unfused fallback sums the slices into slice 0 first. */
int halo_subsample_bwd_slabs(float *dy, int slabs, const float *y, const float *col, float *dpre, float *dw,
                             float *dbias, int B, int T, int F, int C, int ks, int stride, int pad, float p_drop,
                             halo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Multi-layer LSTM, gate order i,f,g,o, two bias vectors per layer.
 * replaces: nn.LSTM forward/backward (cuDNN RNN / oneDNN) at ha/rnn.py:25 (batch_first encoder)
 * and ha/rnn.py:50,63 (time-major Decoder with carried state).
 *   x [T,B,in0] time-major.  w_ih[l] [4H,in_l], w_hh[l] [4H,H], b_ih[l], b_hh[l] [4H]: HOST arrays
 *   of L device pointers.  h0/c0 [L,B,H] or NULL (zeros).  hn/cn [L,B,H] or NULL.
 *   y: last layer's output h_t written at y + t*y_stride_t + b*y_stride_b + j (so the caller picks
 *   batch-first or time-major); relu applied when y_relu (the encoder's x.relu(), rnn.py:26).
 *   Inter-layer dropout p_drop (train only) uses stream HALO_STREAM_LSTM_LAYER0 + l on the
 *   time-major [T,B,H] index.
 *   reserve: halo_lstm_reserve_bytes(); holds per layer h[T+1,B,H], c[T+1,B,H], gates[T,B,4H],
 *   dropped output [T,B,H]; consumed (overwritten with gate gradients) by halo_lstm_bwd.
 * ------------------------------------------------------------------------------------------ */
size_t halo_lstm_reserve_bytes(int T, int B, int in0, int H, int L);
size_t halo_lstm_bwd_workspace_bytes(int T, int B, int in0, int H, int L);
int halo_lstm_fwd(const float *x, const float *const *w_ih, const float *const *w_hh,
                  const float *const *b_ih, const float *const *b_hh, const float *h0, const float *c0,
                  float *y, long y_stride_t, long y_stride_b, int y_relu, float *hn, float *cn,
                  float *reserve, int T, int B, int in0, int H, int L, float p_drop, uint64_t seed,
                  uint32_t offset, const uint32_t *offset_dev, halo_stream_t stream);
/* dy has the strides of y; dhn/dcn [L,B,H] or NULL.  dx [T,B,in0] or NULL.
 * dw_ih/dw_hh/db_ih/db_hh: HOST arrays of L device pointers, overwritten.
 * Layers [layer_begin, layer_end) are processed, top down.  A whole backward is (0, L); a caller
 * that wants the upper layers' gradients early (to start their all-reduce) calls (k, L) and later
 * (0, k) with the SAME workspace, which carries the gradient between the two calls. */
int halo_lstm_bwd(const float *x, const float *const *w_ih, const float *const *w_hh, const float *dy,
                  long y_stride_t, long y_stride_b, int y_relu, const float *dhn, const float *dcn,
                  float *reserve, float *workspace, float *dx, float *const *dw_ih, float *const *dw_hh,
                  float *const *db_ih, float *const *db_hh, int T, int B, int in0, int H, int L,
                  int layer_begin, int layer_end, float p_drop, uint64_t seed, uint32_t offset,
                  const uint32_t *offset_dev, halo_stream_t stream);

/* Contexts.  Every halo_set_* switch (arithmetic mode, scratch, LSTM schedule switches, measurement hooks, the status word, the beam
 * search's reference ISA) lives in a settings record.  There is one process-wide default record; halo_ctx_create() makes a caller-owned
 * copy of the calling thread's current record, and halo_ctx_use(ctx) makes it the record that THIS thread's later halo_* calls read and
 * halo_set_* calls write (NULL: back to the default).  Two trainers, or a trainer and a recognizer, on different threads therefore do
 * not see each other's switches; within one thread, select the context before a group of calls.  Work launched on distinct streams from
 * distinct threads with distinct contexts (and distinct scratch buffers) is independent; calls that share one record, or the default
 * one, must be serialised by the caller. */
typedef struct halo_ctx halo_ctx;
halo_ctx *halo_ctx_create(void);
void halo_ctx_destroy(halo_ctx *ctx);
int halo_ctx_use(halo_ctx *ctx);
/* A caller-owned device uint32 that the persistent recurrences (csrc/lstm_persist*.hip) set to 1 when one of their bounded waits times
 * out (co-residency lost: a CU mask, another tenant, a profiler).  It is sticky -- the library only ever sets it -- and
 * halo_clip_coef_step treats a raised word like a non-finite gradient norm (no update is applied).  NULL (default): no word. */
int halo_set_status_word(uint32_t *device_word);
/* Test hook: the workgroup with this blockIdx of every persistent LSTM forward launched afterwards never publishes its epoch, so its
 * peers run into their bounded waits (0.2 s), raise the abort and status words and leave; -1 (default): none. */
int halo_debug_mute_workgroup(int block);
/* Diagnostic: `blocks` workgroups of 4 waves run `iters` rounds of a bare bf16 MFMA loop on random register operands (shape 0: four
 * 32x32x16 per round and wave, 1: eight 16x16x32 -- 131072 FLOP per wave and round either way) and report per workgroup
 * ticks[2 b] = shader clocks (s_memtime), ticks[2 b + 1] = 100 MHz ticks (s_memrealtime): the clock the chip holds under matrix load
 * (tools/mfma_clock.py).  sink: one float of device memory, never written in practice. */
int halo_debug_mfma_clock(unsigned long long *ticks, float *sink, int blocks, int iters, int shape, unsigned seed, halo_stream_t stream);

/* Weight-resident persistent recurrence (csrc/lstm_persist.hip): when the shape is eligible (split-bf16 arithmetic modes,
 * H in {256, 512, 768, 1024}, (H/16) * ceil(B/16) <= number of CUs) a layer's T dependent step launches become ONE launch
 * whose workgroups keep their W_hh slice in registers and hand h_t (forward) / the gate gradients (backward) to each other
 * through write-through stores and per-workgroup epoch words.  On by default; halo_set_lstm_persistent(0) selects the
 * per-step launch chain (also HALO_LSTM_PERSIST=0 in the environment).  The kernel needs every workgroup resident at once, so
 * nothing else should occupy the GPU's CUs for long while it runs; all its waits are bounded (0.2 s): on a timeout the u32 at
 * byte offset halo_lstm_status_offset() of the reserve (backward != 0: of the backward workspace) is set to 1 and the results
 * of that call are invalid.  It reads 0 after a good call. */
int halo_set_lstm_persistent(int on);
/* The persistent backward writes the gate gradients' split-bf16 GEMM operand images itself (B % 32 == 0, split-bf16 and bf16 modes) instead of
 * leaving them to the operand-image launch behind it (bf16 mode: the hi parts only): same bits, one 4H-wide fp32 re-read less per layer.  On by default
 * (HALO_PERSIST_EMIT=0 / halo_set_lstm_persistent_images(0): off). */
int halo_set_lstm_persistent_images(int on);
int halo_lstm_persistent_eligible(int B, int H);
/* Two-layer persistent recurrence (csrc/lstm_persist2.hip): for L >= 2 in the single-pass bf16 arithmetic mode (HALO_MATH_BF16) and a
 * shape the persistent recurrence takes, halo_lstm_fwd runs the stack's TOP TWO layers' T steps in one launch of T + 2 combined steps
 * (the lower layer at time s beside the upper at time s - 2: one hand-off per combined step, the upper layer's input projection inside
 * its step; the layers below them one launch each), and halo_lstm_bwd called with layer_end = L and layer_begin <= L - 2 likewise (the
 * upper layer's input gradient formed inside the launch).  Same reserve contents as the per-layer path, so either backward follows
 * either forward.  Any batch: the batch rows are independent chains, so more 16-row tiles than the chip has CUs for (H/16 workgroups per
 * tile) run as consecutive launches over the same buffers.  On by default (HALO_LSTM_PERSIST2=0 / halo_set_lstm_persistent2(0): off). */
int halo_set_lstm_persistent2(int on);
/* Two batch tiles per workgroup in the two-layer launches (csrc/lstm_persist2x.hip): a batch of more 16-row tiles than one launch holds
 * workgroups for (more than 64 rows at H = 1024) runs two tiles in each workgroup, interleaved -- while one tile's hand-off is in flight
 * the workgroup computes the other tile's step on the same register-resident weights -- instead of as consecutive launches.  Same
 * buffers, images and results as those launches but for the order in which the two tiles' bias-gradient rows are added.  On by
 * default (HALO_LSTM_INTERLEAVE=0 / halo_set_lstm_interleave(0): consecutive launches).  Per context. */
int halo_set_lstm_interleave(int on);
/* The two-layer launches' weight-gradient products on 256 x 256 workgroup tiles (csrc/gemm256.h): in single-pass `bf16` arithmetic both
 * layers' dW_hh | dW_ih and the K-slices of the caller's input gradient run as ONE launch of one workgroup per CU instead of two launches
 * of the 128 x 128-tile kernel.  Same operand images, same sums up to the order of the k-blocks' additions inside a tile.  On by default
 * (HALO_GEMM256=0 / halo_set_gemm256(0): the 128 x 128-tile launches).  Process-wide. */
int halo_set_gemm256(int on);
/* Data-parallel overlap hook.  event (a hipEvent_t, or NULL: off): halo_lstm_bwd records it on its stream right behind the launch that
 * stores the TOP layer's weight gradients (dw_ih[L-1], dw_hh[L-1]) when the call covers the top two layers as one two-layer launch --
 * the point from which a caller's side stream may start reducing those gradients over the ranks (ha/attention_loop.py:154:
 * DistributedDataParallel overlaps its buckets with backward) while the lower layer's products and the front end's backward still run.
 * Not recorded by any other path: halo_lstm_bwd_mid_event_recorded() tells (the caller then waits for the whole call).  Not for use inside a stream capture.  Per context. */
int halo_set_lstm_bwd_mid_event(void *event);
int halo_lstm_bwd_mid_event_recorded(void);     /* how many times the event has been recorded since it was set (0: this path has no such point) */
/* Inference with static weights.  stamp != 0 is the caller's promise that the LSTM weights change only when the stamp does: a forward-only
 * call (halo_set_lstm_expect_backward(0)) of the two-layer launch then KEEPS the packed weight images that the previous call with the same
 * reserve buffer, weight pointers, shape and stamp left in that reserve (24 MB read + 12 MB written per call at H = 1024 otherwise), so the
 * caller must own the reserve between the calls.  0 (default): every call packs.  Per context. */
int halo_set_lstm_weights_stamp(uint64_t stamp);
/* The two-layer forward packs its three weight images; when a backward of the same step will follow (the default) it writes the
 * backward's three transposed images from the same read of the weights, into the reserve, and halo_lstm_bwd called with that reserve
 * and those weight pointers packs nothing.  Inference callers switch it off (0): the forward then packs its own three only. */
int halo_set_lstm_expect_backward(int on);
/* n > 1: the caller's dx buffer of the NEXT halo_lstm_bwd calls has room for n matrices [T*B][in0] and its consumer can add K-slices
 * (halo_subsample_bwd_slabs): the two-layer launch's input-gradient product (layer_begin = 0) may then leave up to n unreduced slices there
 * instead of running a reduce launch.  halo_lstm_dx_slabs_left() after the call says how many it left (1: dx is the gradient itself, as
 * always on every other path).  1 <= n <= 64; default 1.  Per context. */
int halo_set_lstm_dx_slabs(int n);
int halo_lstm_dx_slabs_left(void);
/* Small reductions that ride in another launch.  on = 1: halo_ctc_head_bwd's second launch (the fixed-order sum of its per-utterance
 * partials) and the two-layer halo_lstm_bwd's bias-gradient sums are QUEUED on the context (at most 4; a full queue launches as before)
 * instead of launched, and the next halo_subsample_bwd[_slabs] runs them from the tail blocks of its reduce launch -- same arithmetic and
 * order, one ~5 us launch each less.  halo_flush_small_jobs launches whatever is queued (nothing: no launch).  The caller keeps the
 * workspaces those reductions read alive until then, calls everything on one stream, and reads the results (dweight / dbias, db_ih /
 * db_hh) only after the carrying launch.  Switching on clears the queue.  Default 0.  Per context. */
int halo_set_defer_small_jobs(int on);
int halo_flush_small_jobs(halo_stream_t stream);
/* The squared gradient norm without a pass over the gradients.  partials != NULL: the launches of the NEXT backward calls that store
 * clipped gradients also write the sums of their squares, one float per workgroup, into partials[0 .. count) -- the two-layer
 * halo_lstm_bwd's two weight-gradient launches (from the accumulators they store), its bias sums (when deferred: small jobs above), and
 * halo_subsample_bwd[_slabs]'s reduce launch.  halo_grad_sumsq_state reports how many slots were written and which producers
 * contributed: bit 0 / 1 the upper / lower layer's W_ih + W_hh, bit 2 / 3 their b_ih + b_hh, bit 4 the conv's weight + bias.  A caller whose
 * clipped parameters are exactly those (a 2-layer stack behind the conv front end) passes the partials to halo_clip_coef[_step] with that
 * count when all five bits are set, and runs halo_sumsq otherwise.  capacity: floats available (a producer without room does not
 * contribute).  NULL: off (default).  Setting it (to anything) resets count and bits.  Host bookkeeping, per context. */
int halo_set_grad_sumsq(float *partials, int capacity);
int halo_grad_sumsq_state(int *count, unsigned *covered);
int halo_lstm_persistent2_eligible(int T, int B, int H, int L);
size_t halo_lstm_status_offset(int backward, int T, int B, int in0, int H, int L);

/* Diagnostic: device buffer of uint64 [blocks][T][16] that the persistent forward fills with 100 MHz time stamps of its phases
 * (tools/persist_stamps.py reads them); NULL (default) turns the stamps off. */
int halo_lstm_persist_stamps(void *device_buffer);

/* Measurement hook (bench.py): while set, every layer's recurrent chain inside halo_lstm_fwd / halo_lstm_bwd is bracketed by
 * hipEventRecord(ev_begin) / hipEventRecord(ev_end) on the call's stream (hipEvent_t handles; NULL, NULL clears).  The
 * batched GEMMs and operand preparation of the same call stay outside the bracket.  Not for use under stream capture.
 * halo_lstm_chain_info: how the last forward (backward != 0: backward) chain ran: number of kernel launches
 * (1 = the persistent weight-resident kernel, T = one launch per time step) and the kernel's name. */
int halo_lstm_chain_events(void *ev_begin, void *ev_end);
int halo_lstm_chain_info(int backward, int *launches, char *kernel, int kernel_len);

/* ------------------------------------------------------------------------------------------
 * Row-wise log-softmax.   replaces: features.log_softmax(dim=-1) ha/recognizer.py:46
 * ------------------------------------------------------------------------------------------ */
int halo_log_softmax_fwd(const float *x, float *y, int rows, int cols, halo_stream_t stream);
int halo_log_softmax_bwd(const float *dy, const float *y, float *dx, int rows, int cols,
                         halo_stream_t stream);
/* column sums: out[n] = sum_m x[m*ld + n]  (bias gradients) */
int halo_colsum(const float *x, int rows, int cols, int ld, float *out, halo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * CTC lattice, blank = 0, one workgroup per utterance.
 * replaces: F.ctc_loss (ATen ctc_loss_gpu alpha/beta kernels) at ha/recognizer.py:71 and the
 * pure-torch recursion ha/ctc.py:110-174 (ctc_forward_score3), ctc.py:4-107 (score1/score2).
 *   lp: log-probabilities, element (t,n,c) at lp + t*stride_t + n*stride_n + c
 *   targets [N,S] int64 (row stride tg_stride), input_lengths/target_lengths [N] int64
 *   alpha [N,T,2S+1] out;  nll [N] out (positive negative-log-likelihood)
 *   flags:
 *     HALO_CTC_FULL_LATTICE  run all T frames over all 2S+1 states and read out at
 *                            (il-1, 2*tl), (il-1, 2*tl-1) like ctc.py:131-174; otherwise restrict
 *                            to the utterance's own lengths like F.ctc_loss.
 *     HALO_CTC_FINITE_MIN    "log 0" is -FLT_MAX (ctc.py:135) instead of -inf
 *     HALO_CTC_NO_LEAD_BLANK_LOOP  state 0 is never updated after t=0 (ctc.py:26, :103)
 *     HALO_CTC_WRAP_SKIP     state 1 also receives alpha[t-1, last] (python index -1, ctc.py:29)
 * ------------------------------------------------------------------------------------------ */
#define HALO_CTC_FULL_LATTICE 1
#define HALO_CTC_FINITE_MIN 2
#define HALO_CTC_NO_LEAD_BLANK_LOOP 4
#define HALO_CTC_WRAP_SKIP 8
int halo_ctc_fwd(const float *lp, long stride_t, long stride_n, int T, int N, int C,
                 const int64_t *targets, long tg_stride, int S, const int64_t *input_lengths,
                 const int64_t *target_lengths, int flags, float *alpha, float *nll,
                 halo_stream_t stream);
/* Gradient in F.ctc_loss's convention (ATen ctc_loss_backward): for t < il
 *   grad(t,n,c) = grad_out[n] * (exp(lp) - exp(logsum_{s: l'_s=c}(alpha+beta) + nll - lp)), else 0.
 * beta: workspace [N,T,2S+1].  grad has lp's strides (gstride_t, gstride_n). */
int halo_ctc_bwd(const float *lp, long stride_t, long stride_n, int T, int N, int C,
                 const int64_t *targets, long tg_stride, int S, const int64_t *input_lengths,
                 const int64_t *target_lengths, const float *alpha, const float *nll,
                 const float *grad_out, float *beta, float *grad, long gstride_t, long gstride_n,
                 halo_stream_t stream);

/* Two scalar-glue launches of the training step (ha/rnn.py:13-18, ha/recognizer.py:71 reduction='mean'):
 *   halo_ctc_prepare: feature_lengths[n] = floor((il[n] + 2*pad - ks)/stride + 1) computed in float like
 *     the reference, grad_out[n] = 1 / (max(tl[n],1) * N) = d(mean loss)/d(nll[n]);
 *   halo_ctc_mean_loss: *loss = mean_n(nll[n] / max(tl[n],1)), fixed summation order. */
int halo_ctc_prepare(const int64_t *input_lengths, const int64_t *target_lengths, int n, int ks, int stride,
                     int pad, int64_t *feature_lengths, float *grad_out, halo_stream_t stream);
int halo_ctc_mean_loss(const float *nll, const int64_t *target_lengths, int n, float *loss,
                       halo_stream_t stream);

/* TemporalClassifier.decode (ha/recognizer.py:48-59) in ONE launch, one workgroup per utterance: Linear -> log_softmax -> per frame the
 * best class and its log-prob (alignments, scores [B][T]) -> unique_consecutive, blanks dropped (hyp [B][T] zero padded, hyp_len [B]);
 * lp (optional) receives the log-probs [B][T][V].  Same shape limits as halo_ctc_head_fwd (T <= 32, V <= 32). */
int halo_ctc_head_greedy(const float *features, const float *weight, const float *bias, float *lp, int64_t *alignments, float *scores,
                         int64_t *hyp, int64_t *hyp_len, int B, int T, int H, int V, halo_stream_t stream);
/* ------------------------------------------------------------------------------------------
 * The whole CTC head of a training step (TemporalClassifier.forward and its backward) in three launches.
 * replaces: dropout -> nn.Linear -> log_softmax -> F.ctc_loss(mean) ha/recognizer.py:43-46,61-73 plus the feature-length arithmetic of
 *           ha/rnn.py:13-18, i.e. halo_dropout_fwd + halo_gemm_f32 + halo_log_softmax_fwd + halo_ctc_prepare + halo_ctc_fwd +
 *           halo_ctc_mean_loss, and in the backward halo_ctc_bwd + halo_log_softmax_bwd + 2 x halo_gemm_f32 + halo_colsum.
 * Fast path for T <= 32 frames, V <= 32 classes, 2S+1 <= 64 lattice states, H % 64 == 0 (halo_ctc_head_supported; otherwise
 * HALO_ENOTSUP and the caller uses the separate operators).  One workgroup per utterance, products on the exact-f32 MFMA.
 *   features [B,T,H] (the encoder's output), weight [V,H], bias [V]; the classifier dropout (p_drop, stream_id) is applied to the
 *   features as halo_dropout_fwd would (flat index of [B,T,H]) and again, from the same stream, to d features in the backward.
 *   input_lengths: frames BEFORE the subsample conv (ks, stride, pad); feature_lengths [B] int64 out.
 *   lp [B,T,V], alpha [B,T,2S+1], nll [B], grad_out [B] (= 1 / (max(tl,1) * B)), loss (scalar: reduction='mean') out;
 *   ticket: device uint32, 0 before the first call (the kernel leaves it 0).
 *   backward: dfeatures [B,T,H], dweight [V,H], dbias [V] out; workspace halo_ctc_head_workspace_bytes(). */
int halo_ctc_head_supported(int T, int H, int V, int S);
size_t halo_ctc_head_workspace_bytes(int B, int H, int V);
int halo_ctc_head_fwd(const float *features, const float *weight, const float *bias, float p_drop, uint64_t seed,
                      uint32_t stream_id, uint32_t offset, const uint32_t *offset_dev, const int64_t *input_lengths, int ks,
                      int stride, int pad, const int64_t *targets, long tg_stride, int S, const int64_t *target_lengths,
                      float *lp, float *alpha, float *nll, int64_t *feature_lengths, float *grad_out, float *loss,
                      uint32_t *ticket, int B, int T, int H, int V, halo_stream_t stream);
int halo_ctc_head_bwd(const float *features, const float *weight, float p_drop, uint64_t seed, uint32_t stream_id,
                      uint32_t offset, const uint32_t *offset_dev, const int64_t *feature_lengths, const int64_t *targets,
                      long tg_stride, int S, const int64_t *target_lengths, const float *lp, const float *alpha,
                      const float *nll, const float *grad_out, float *dfeatures, float *dweight, float *dbias,
                      void *workspace, int B, int T, int H, int V, halo_stream_t stream);

/* The same head, forward AND backward, in ONE launch -- for a caller that runs both every step (a training step: ha/loop.py:176-196
 * calls the loss and loss.backward() back to back).  replaces: halo_ctc_head_fwd + halo_ctc_head_bwd, i.e. ha/recognizer.py:43-46,61-73
 * and their autograd backward.  Same shape limits (halo_ctc_head_supported); products on split-bf16 MFMA (fp32-grade: the `bf16x3`
 * arithmetic), so HALO_ENOTSUP in the exact-f32 mode -- the caller then uses the two launches above.  grid = B x slices workgroups,
 * the slices of an utterance (H/slices feature columns each) exchanging their partial logits once; alpha and beta run side by side.
 *   lp [B,T,V]: optional (NULL: not written); nll [B], feature_lengths [B], loss out as halo_ctc_head_fwd;
 *   dfeatures [B,T,H], dweight [V,H], dbias [V] out as halo_ctc_head_bwd (the partials' sum may be deferred the same way);
 *   workspace: halo_ctc_head_train_workspace_bytes(); ticket: halo_ctc_head_train_ticket_words(B, H) device uint32 words (8-byte
 *   aligned), all 0 before the first call, the caller's to keep between calls: the loss ticket, the count of launches, and the
 *   slices' partial logits, each stored with that count as its tag.
 * A slice that never arrives (0.2 s) makes the loss NaN instead of hanging the launch. */
size_t halo_ctc_head_train_workspace_bytes(int B, int H, int V);
size_t halo_ctc_head_train_ticket_words(int B, int H);
int halo_ctc_head_train(const float *features, const float *weight, const float *bias, float p_drop, uint64_t seed,
                        uint32_t stream_id, uint32_t offset, const uint32_t *offset_dev, const int64_t *input_lengths, int ks,
                        int stride, int pad, const int64_t *targets, long tg_stride, int S, const int64_t *target_lengths,
                        float *lp, float *nll, int64_t *feature_lengths, float *loss, uint32_t *ticket, float *dfeatures,
                        float *dweight, float *dbias, void *workspace, int B, int T, int H, int V, halo_stream_t stream);

/* Greedy decode.   replaces: logits.max(-1) + unique_consecutive + drop-0 loop, recognizer.py:51-55
 *   lp [N,T,C] contiguous; alignments [N,T] int64, scores [N,T] f32, hyp [N,T] int64 (first
 *   hyp_len[n] entries valid), hyp_len [N] int64.  Input lengths are ignored, as in the reference. */
int halo_ctc_greedy(const float *lp, int N, int T, int C, int64_t *alignments, float *scores,
                    int64_t *hyp, int64_t *hyp_len, halo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * 80-mel log filterbank front-end on the device (ha/data.py:136-140: torchaudio.compliance.kaldi.fbank(wav, num_mel_bins=80) with that
 * function's defaults; haloop_amd/fbank.py chains these with two halo_gemm_f32 products -- the real DFT and the mel filters).
 *   halo_fbank_frames  waveform [n_samples] -> frames [n_frames][padded]: frame f = samples [f*shift, f*shift + frame_len) (snip edges),
 *                      minus its mean (remove_dc), pre-emphasised x[i] - c*x[i-1] (x[0] - c*x[0]), times window[i], zero padded
 *   halo_fbank_power   spectrum [n_frames][2*bins] (real parts, then imaginary parts) -> power [n_frames][ld] (columns >= bins zero)
 *   halo_fbank_log     x = log(max(x, eps)) in place */
int halo_fbank_frames(const float *wav, long n_samples, int frame_len, int shift, int padded, float preemphasis, int remove_dc,
                      const float *window, float *frames, int n_frames, halo_stream_t stream);
int halo_fbank_power(const float *spectrum, int n_frames, int bins, float *power, int ld, halo_stream_t stream);
int halo_fbank_log(float *x, long n, float eps, halo_stream_t stream);
/* The rest of the front-end in ONE launch and in float64: the N-point real DFT of every frame against a table of the N twiddles
 * (twiddle[2 j] = cos(2 pi j / N), twiddle[2 j + 1] = sin(2 pi j / N), N = padded, a power of two <= 2048), |X|^2, the num_bins
 * triangular filters (banks [num_bins][N / 2 + 1], float64) and log(max(., eps)) -> out [n_frames][num_bins] float32.  Replaces
 * the two exact-f32 products + halo_fbank_power + halo_fbank_log where the log-mels must hold to 1e-4 far below a frame's peak. */
int halo_fbank_spectrum_mel(const float *frames, int n_frames, int padded, const double *twiddle, const double *banks, int num_bins,
                            float eps, float *out, halo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The reference's two further lattices, one workgroup per utterance, forward score and gradient.
 *   halo_star_ctc_fwd     ha/star.py:65-166 star_ctc_forward_score(emissions [T, N, C] log-probabilities, targets [N, S], lengths,
 *                         star_penalty): losses [N] = -log of the 4S+3-state star lattice (stars are evaluated from the C symbols in
 *                         the kernel: the 2C-wide emissions of intersperse_stars :9-49 are never built).  workspace (optional, needed
 *                         for the backward): halo_star_ctc_workspace_bytes(T, N, S) bytes, keeps the alpha lattice.
 *   halo_star_ctc_bwd     d sum_n grad_losses[n] * losses[n] / d emissions (what autograd gives the reference), written to
 *                         grad_log_probs with the strides of log_probs; rows at or past emission_lengths[n] are zero.
 *   halo_transducer_fwd   ha/transducer.py:175-207 transducer_forward_score(joint [N, T, U+1, K] log-probabilities, targets [N, U],
 *                         joint_lengths, target_lengths) -> losses [N]; any T (the reference's scan needs T in (2^(k-1/2), 2^k]).
 *   halo_transducer_bwd   its gradient w.r.t. joint (dense [N, T, U+1, K], zero outside the two symbols of every lattice cell). */
size_t halo_star_ctc_workspace_bytes(int T, int N, int S);
int halo_star_ctc_fwd(const float *log_probs, long stride_t, long stride_n, int T, int N, int C, const int64_t *targets, int S,
                      const int64_t *emission_lengths, const int64_t *target_lengths, float star_penalty, void *workspace,
                      float *losses, halo_stream_t stream);
int halo_star_ctc_bwd(const float *log_probs, long stride_t, long stride_n, int T, int N, int C, const int64_t *targets, int S,
                      const int64_t *emission_lengths, const int64_t *target_lengths, float star_penalty,
                      const void *workspace, const float *losses, const float *grad_losses, float *grad_log_probs,
                      halo_stream_t stream);
size_t halo_transducer_workspace_bytes(int N, int T, int U1);
int halo_transducer_fwd(const float *joint, int N, int T, int U1, int K, const int64_t *targets, const int *joint_lengths,
                        const int *target_lengths, void *workspace, float *losses, halo_stream_t stream);
int halo_transducer_bwd(const float *joint, int N, int T, int U1, int K, const int64_t *targets, const int *joint_lengths,
                        const int *target_lengths, const void *workspace, const float *losses, const float *grad_losses,
                        float *grad_joint, halo_stream_t stream);

/* Beam search with the reference's exact (quirky) semantics, one workgroup per utterance.
 * replaces: ha/beam.py:71-137 (logits) and ha/beam.py:5-68 (probs, log_domain = 0).
 *   em [N,T,V]; seqs [N,beam,T] int64, lens [N,beam] int32, scores [N,beam] f32, ranked best first.
 *   workspace: halo_ctc_beam_workspace_bytes().  Requires beam <= 1+V (reference: topk raises).
 *   Scores are formed with torch.logaddexp's CPU arithmetic reproduced bit for bit (ATen's vector loop on Sleef's expf / log1pf
 *   for whole vector chunks of the candidate array, the scalar loop on glibc's expf / log1pf for the remainder and for the
 *   per-prefix update), because on flat emissions score-tied hypotheses are separated by its last ulp.
 *   halo_set_beam_vector_chunk: 2 * Vectorized<float>::size() of the machine whose reference run is to be reproduced:
 *   32 (AVX-512, default: the fixtures under tests/golden/), 16 (AVX2), 0 (no vector ISA: everything through the scalar loop).
 *   halo_logaddexp_aten: that function on its own, element i of n taking the vector path iff i < n - n % chunk. */
int halo_set_beam_vector_chunk(int elements);
int halo_logaddexp_aten(const float *a, const float *b, float *out, size_t n, halo_stream_t stream);
size_t halo_ctc_beam_workspace_bytes(int N, int T, int V, int beam);
int halo_ctc_beam(const float *em, int N, int T, int V, int beam, int log_domain, int64_t *seqs,
                  int32_t *lens, float *scores, void *workspace, halo_stream_t stream);

/* Row-wise top-k with the order torch.topk(largest=True, sorted=True) produces on CPU, equal keys
 * included (ATen TopKImpl.h: partial_sort when k*64 <= n, else nth_element + sort; libstdc++).
 * replaces: seq_logits.topk(beam_size) ha/beam.py:129 (and beam.py:60).
 *   values [rows,n]; out_values [rows,k] f32; out_indices [rows,k] int64; workspace rows*n*8 bytes. */
int halo_topk_f32(const float *values, int rows, int n, int k, float *out_values, int64_t *out_indices,
                  void *workspace, halo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * GPT forward (scoring) operators.  replaces, in ha/attention.py:
 *   halo_embed_fwd            wte(input_ids) + wpe(pos)                      :222-224
 *   halo_layernorm_fwd        F.layer_norm(x, (C,), weight, bias|NULL, eps)  :29  (also StableEmbedding's norm :61)
 *   halo_attention_causal_fwd F.scaled_dot_product_attention(q,k,v,is_causal=True) on the packed c_attn
 *                             output qkv [B,T,3C] (q|k|v, heads side by side) -> y [B,T,C]  :113-123,90
 *   halo_cross_entropy_fwd    F.cross_entropy(logits, targets, ignore_index, reduction='none')  :231
 * Linear layers, tanh-GELU and the residual adds are halo_gemm_f32 / halo_gemm_split epilogues.
 * halo_attention_causal_fwd is halo_attention_fwd on the packed qkv rows; head_dim in {16, 32, 64} (GPT-2 small: 64). */
int halo_embed_fwd(const int64_t *ids, const float *wte, const float *wpe, float *x, int n_tokens, int T, int C,
                   int pos0, int vocab, halo_stream_t stream);
int halo_layernorm_fwd(const float *x, const float *weight, const float *bias, float *y, int rows, int C,
                       float eps, halo_stream_t stream);
int halo_attention_causal_fwd(const float *qkv, float *y, int B, int T, int n_head, int C, halo_stream_t stream);
int halo_cross_entropy_fwd(const float *logits, const int64_t *targets, float *loss, int rows, int V, long ld,
                           long ignore_index, halo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Attention operators shared by the GPT path and the encoder-decoder ASR path (ha/transformer.py).
 *   halo_attention_fwd     F.scaled_dot_product_attention(q, k, v, attn_mask|is_causal)   transformer.py:356, attention.py:90
 *                          q rows [N, Tq], k/v rows [N, Tk] given by row/batch strides (elements); head h occupies
 *                          columns [h*head_dim, (h+1)*head_dim) of a row, so the packed q|k|v output of one fused GEMM
 *                          is consumed in place.  Key j is visible to query i iff j < min(Tk, key_lengths[n]) (the
 *                          ~memory_mask of transformer.py:476) and, when causal, j <= i + Tk - Tq.
 *                          lse (optional) [N, heads, Tq] = log-sum-exp of the scaled scores (saved for the backward);
 *                          entropy (optional) [N, heads, Tq] = -sum_j att*log(att + 1e-8), the monitor of attend()
 *                          transformer.py:413-430 (its mean over all rows is att_entropy).  head_dim in {16, 32, 64}.
 *   halo_rope_table /      rotate_interleaved transformer.py:16-31: cos/sin tables [T, head_dim/2] of t * base^(-2i/head_dim);
 *   halo_rope_interleaved  in-place rotation of the pairs (2i, 2i+1) of every head of x rows (row r sits at position
 *                          t0 + r % T); inverse != 0 rotates back (the backward of the rotation).
 *   halo_kv_cache_store    kv_cache[layer, 0|1, alive, :, t0:t0+S, :] = k|v  transformer.py:318-319,333-334: fp32 rows
 *                          (k at column 0, v at column v_offset of each source row) -> float16 caches [N, heads, cache_len, head_dim]
 *   halo_attention_decode  one query token per (n, head) against the first n_keys cached keys: the T == 1 branch of
 *                          MultiHeadAttention.forward transformer.py:313-356; cos/sin tables (optional) rotate the cached
 *                          keys at positions 0..n_keys-1 (transformer.py:343), q must already be rotated.
 *   halo_logprob_max       log_softmax(dim=-1).max(dim=-1) per row (+ sum p*logp/log 2)  transformer.py:175-179,116
 *   halo_greedy_update     the per-step bookkeeping of Decoder.decode transformer.py:179-192 (alive rows only; the entropy
 *                          increment is the sum over ALL alive rows, as the reference's .sum(dim=(-2,-1)) makes it). */
int halo_attention_fwd(const float *q, long q_row_stride, long q_batch_stride, const float *k, const float *v,
                       long kv_row_stride, long kv_batch_stride, float *y, long y_row_stride, long y_batch_stride,
                       float *lse, float *entropy, int N, int heads, int head_dim, int Tq, int Tk, int causal,
                       const int *key_lengths, halo_stream_t stream);
/* halo_attention_fwd with explicit head strides: q_head_stride / kv_head_stride = head_dim for packed rows, or
 * cache_len * head_dim (with row stride head_dim, batch stride heads * cache_len * head_dim) to read K/V straight from a
 * [N, heads, cache_len, head_dim] fp32 cache -- the `past` / `present` layout of ha/attention.py:64-69,232 (attend_cached).
 * p_drop > 0: inverted dropout on the attention probabilities (the dropout_p of SDPA, ha/transformer.py:356,
 * ha/attention.py:90) from the Philox stream (seed, stream_id, offset [+ *offset_dev]); probability (n, h, i, j) uses element
 *   e = ((((n*heads + h)*Tq + i) * ceil(Tk/64) + j/64) * 64 + 4*(j%16) + (j%64)/16
 * of the stream; halo_attention_bwd must be given the same five values. */
int halo_attention_fwd_strided(const float *q, long q_row_stride, long q_batch_stride, long q_head_stride,
                               const float *k, const float *v, long kv_row_stride, long kv_batch_stride,
                               long kv_head_stride, float *y, long y_row_stride, long y_batch_stride, float *lse,
                               float *entropy, int N, int heads, int head_dim, int Tq, int Tk, int causal,
                               const int *key_lengths, float p_drop, uint64_t seed, uint32_t stream_id,
                               uint32_t offset, const uint32_t *offset_dev, halo_stream_t stream);
int halo_rope_table(float *cos_table, float *sin_table, int T, int head_dim, float base, halo_stream_t stream);
int halo_rope_interleaved(float *x, long row_stride, int n_rows, int T, int heads, int head_dim, int t0,
                          const float *cos_table, const float *sin_table, int table_rows, int inverse,
                          halo_stream_t stream);
int halo_kv_cache_store(const float *src, long src_row_stride, long v_offset, void *cache_k, void *cache_v, int N,
                        int S, int heads, int head_dim, int cache_len, int t0, halo_stream_t stream);
/* fp32 twin of halo_kv_cache_store: present[layer, 0|1, :, :, t0:t0+S, :] of ha/attention.py:66-67,129 */
int halo_kv_cache_store_f32(const float *src, long src_row_stride, long v_offset, float *cache_k, float *cache_v,
                            int N, int S, int heads, int head_dim, int cache_len, int t0, halo_stream_t stream);
int halo_attention_decode(const float *q, long q_row_stride, const void *cache_k, const void *cache_v, float *y,
                          long y_row_stride, int N, int heads, int head_dim, int cache_len, int n_keys,
                          const int *key_lengths, const float *cos_table, const float *sin_table,
                          halo_stream_t stream);
/* One self-attention decode step in ONE launch: k_new / v_new (rows like q, same row stride) are rounded to float16 and stored
 * at cache position n_keys - 1, q is rotated by that position (when tables are given), the cached keys by theirs, and the
 * token attends to all n_keys positions: halo_kv_cache_store + halo_rope_interleaved + halo_attention_decode fused. */
int halo_attention_decode_step(const float *q, const float *k_new, const float *v_new, long row_stride, void *cache_k,
                               void *cache_v, float *y, long y_row_stride, int N, int heads, int head_dim, int cache_len,
                               int n_keys, const float *cos_table, const float *sin_table, halo_stream_t stream);
int halo_logprob_max(const float *logits, long ld, int rows, int V, float *values, int64_t *indices,
                     float *neg_entropy_bits, halo_stream_t stream);
int halo_greedy_update(const float *values, const int64_t *indices, const float *neg_entropy_bits, int64_t *tokens,
                       long tokens_ld, int t, int plen, int etx, uint8_t *alive, int *output_lengths,
                       float *log_probs, float *sum_entropies, int N, halo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused launches of one batched greedy decode step (ha/transformer.py:160-195; Block.forward :476-494 with kv caches, T == 1):
 * 5 launches per decoder layer + 2 per token.  M = N utterances is one or a few 16-row MFMA tiles, so these do not go through the
 * tiled GEMM: activations are read as fp32 rows, weights from "decode images" (split-bf16 fragments in MFMA operand order).
 *   halo_decode_image        image of a weight matrix W [n_out][k] (nn.Linear layout); halo_decode_image_bytes() bytes, 16-aligned
 *   halo_decode_linear       out (+)= act(layer_norm?(x) W^T): x [rows][k] fp32; ln_weight != NULL applies F.layer_norm(x, weight, no
 *                            bias, eps) first (k in {512, 768, 1024}); flags: HALO_GEMM_ACCUM (residual add into out) and / or
 *                            HALO_GEMM_GELU_ERF; without LayerNorm k % 512 == 0.  Split-bf16 arithmetic (three MFMAs per product, fp32 accumulate).
 *   halo_decode_attention_pair  both attentions of a step from the packed projection rows a [N][4C] = cross query | self q | k | v:
 *                            cross-attention over the fp16 memory caches [N][heads][S][head_dim] (keys < memory_lengths[n]) into
 *                            y[:, 0:C]; halo_attention_decode_step on the time caches (store at n_keys - 1, rotary) into y[:, C:2C]
 *   halo_decode_memory_caches  the float16 cross-attention caches [layers][2][N][heads][S][head_dim] of every layer (transformer.py:324-334)
 *                            from ONE product: kv rows [N*S], layer l's keys at columns [l*2C, l*2C + C), values at [l*2C + C, (l+1)*2C)
 *   halo_decode_token        halo_logprob_max + halo_greedy_update on logits [N][V], then y_next[n] = wte[tokens[n, t + 1]] (the next
 *                            step's embedding; y_next may be NULL).  alive is [2][N], double-buffered: step t reads plane t & 1 and
 *                            writes the other (several workgroups share the step).  N <= 1024. */
int halo_decode_memory_caches(const float *kv, long row_stride, int layers, void *caches, int N, int S, int heads,
                              int head_dim, halo_stream_t stream);
size_t halo_decode_image_bytes(int n_out, int k);
int halo_decode_image(const float *weight, int n_out, int k, long ld, void *image, halo_stream_t stream);
int halo_decode_linear_supported(int k, int layernorm);
int halo_decode_linear(const float *x, long ldx, int rows, int k, const float *ln_weight, float eps, const void *w_image,
                       int n_out, float *out, long ldo, int flags, halo_stream_t stream);
/* The same launch with the residual stream as a PAIR (main, side), x = main + side -- what lets the two accumulating products of a decoder
 * layer (x += proj(...), x += mix_chan[2](...): ha/transformer.py:476-494) run as TWO K-slices on twice the workgroups with every sum in a
 * fixed order: x_side (LayerNorm variants) is added to the input rows before the statistics; side_out != NULL (no LayerNorm, flags ==
 * HALO_GEMM_ACCUM, k % 256 == 0): slice 0 writes out = (out + side_in) + its half of the product (side_in may be NULL), slice 1 its half
 * alone to side_out [rows][ldo]; the next launch reads out + side_out.  side_out must differ from out and side_in. */
int halo_decode_linear_pair(const float *x, const float *x_side, long ldx, int rows, int k, const float *ln_weight, float eps, const void *w_image,
                            int n_out, float *out, const float *side_in, float *side_out, long ldo, int flags, halo_stream_t stream);
int halo_decode_attention_pair(const float *a, long a_row_stride, int N, int heads, int head_dim, const void *mem_k,
                               const void *mem_v, int S, const int *memory_lengths, void *time_k, void *time_v,
                               int cache_len, int n_keys, const float *cos_table, const float *sin_table, float *y,
                               long y_row_stride, halo_stream_t stream);
int halo_decode_token(const float *logits, long ld, int N, int V, int64_t *tokens, long tokens_ld, int t, int plen, int etx,
                      uint8_t *alive, int *output_lengths, float *log_probs, float *sum_entropies, const float *wte,
                      int vocab, int C, float *y_next, halo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Backward operators of the GPT / transformer training step (the autograd graph of ha/attention.py:205-232 as
 * `hal` runs it, ha/attention_loop.py:196-215: loss.backward()).
 *   halo_attention_bwd         gradient of halo_attention_fwd: dq, dk, dv (same row layouts as q, k, v; written, not
 *                              accumulated) from dy, the saved y and lse; delta [N, heads, Tq] is scratch.
 *                              Two sweeps (per query tile, per key tile), no atomics: bitwise reproducible.
 *   halo_layernorm_bwd         dx = dres + d(layer_norm)/dx . dy (dres: the residual-stream gradient, may be NULL),
 *                              dweight, dbias (NULL when the LayerNorm has no bias); workspace per *_workspace_bytes.
 *   halo_gelu_fwd / _bwd       y = gelu(a) keeping the pre-activation; da = dy * gelu'(a); exact != 0 -> erf form
 *   halo_cross_entropy_fwd_lse halo_cross_entropy_fwd that also returns the row log-sum-exp
 *   halo_cross_entropy_bwd     logits <- (softmax(logits) - onehot(target)) * grad[n * grad_stride] in place
 *                              (grad_stride 0: one scalar for all rows; 1: per row); ignored rows -> 0
 *   halo_embed_bwd             dwte[ids] += dx (float atomics; dwte is the tied lm_head gradient buffer),
 *                              dwpe[pos0 + t] (+)= sum_b dx  (either of dwte / dwpe may be NULL) */
int halo_attention_bwd(const float *q, long q_row_stride, long q_batch_stride, const float *k, const float *v,
                       long kv_row_stride, long kv_batch_stride, const float *y, const float *dy, long y_row_stride,
                       long y_batch_stride, const float *lse, float *delta, float *dq, long dq_row_stride,
                       long dq_batch_stride, float *dk, float *dv, long dkv_row_stride, long dkv_batch_stride, int N,
                       int heads, int head_dim, int Tq, int Tk, int causal, const int *key_lengths, float p_drop,
                       uint64_t seed, uint32_t stream_id, uint32_t offset, const uint32_t *offset_dev,
                       halo_stream_t stream);
/* ha/transformer.py:413-430 `attend(q, k, v, mask)` for ANY boolean mask and head dimension (the reference's tests/test_attention.py: head
 * dimension 7, a (T, S) triangle): q [N][heads][T][hd], k / v [N][heads][S][hd] contiguous fp32; mask bytes (non-zero = the key is hidden
 * from the query) addressed as mask[n * stride_n + h * stride_h + t * stride_t + s] (stride 0 = broadcast), NULL = none; y like q;
 * entropy [N][heads][T] = -sum att log(att + 1e-8) (optional).  One wave per query row, exact fp32: the slow, general path; the models run
 * the tiled kernels above through key lengths and the causal flag.  HALO_ENOTSUP when 4 (hd + S) floats exceed the LDS. */
int halo_attention_masked(const float *q, const float *k, const float *v, const unsigned char *mask, long mask_stride_n, long mask_stride_h,
                          long mask_stride_t, float *y, float *entropy, int N, int heads, int T, int S, int head_dim, halo_stream_t stream);
/* halo_attention_fwd_strided (packed rows) / halo_attention_bwd on the matrix-core kernels with row-major bf16 outputs for the Linear
 * layers around the attention (bf16 / bf16x3 modes, head_dim 32 or 64; HALO_ENOTSUP otherwise): the forward writes y in fp32 (the
 * backward's delta needs it) AND as bf16 [rows][ybf_row_stride]; the backward writes dq, dk, dv as bf16 ONLY (one row / batch stride,
 * e.g. the three column blocks of a packed [rows][3C] buffer) -- they are operands of c_attn's two gradient products and nothing else. */
int halo_attention_fwd_bf16(const float *q, long q_row_stride, long q_batch_stride, const float *k, const float *v, long kv_row_stride,
                            long kv_batch_stride, float *y, long y_row_stride, long y_batch_stride, void *y_bf16, long ybf_row_stride,
                            long ybf_batch_stride, float *lse, int N, int heads, int head_dim, int Tq, int Tk, int causal,
                            const int *key_lengths, float p_drop, uint64_t seed, uint32_t stream_id, uint32_t offset,
                            const uint32_t *offset_dev, halo_stream_t stream);
int halo_attention_bwd_bf16(const float *q, long q_row_stride, long q_batch_stride, const float *k, const float *v, long kv_row_stride,
                            long kv_batch_stride, const float *y, const float *dy, long y_row_stride, long y_batch_stride, const float *lse,
                            float *delta, void *dq_bf16, void *dk_bf16, void *dv_bf16, long d_row_stride, long d_batch_stride, int N,
                            int heads, int head_dim, int Tq, int Tk, int causal, const int *key_lengths, float p_drop, uint64_t seed,
                            uint32_t stream_id, uint32_t offset, const uint32_t *offset_dev, halo_stream_t stream);
size_t halo_layernorm_bwd_workspace_bytes(int rows, int C);
int halo_layernorm_bwd(const float *dy, const float *x, const float *weight, const float *dres, float *dx,
                       float *dweight, float *dbias, void *workspace, int rows, int C, float eps,
                       halo_stream_t stream);
/* halo_layernorm_bwd that also writes dx as row-major bf16 (dx_bf16 [rows][C]; C % 4 == 0, C <= 2048, 16-byte aligned operands): dx is the
 * output gradient of the Linear below, whose two gradient products read it as bf16 */
int halo_layernorm_bwd_bf16(const float *dy, const float *x, const float *weight, const float *dres, float *dx, void *dx_bf16,
                            float *dweight, float *dbias, void *workspace, int rows, int C, float eps, halo_stream_t stream);
/* ... with the incoming gradient dy as ROW-MAJOR bf16 (round 5: the input-gradient product's bf16 result, halo_gemm_rows); dx_bf16 optional.
 * C % 4 == 0, C <= 2048. */
int halo_layernorm_bwd_b16(const void *dy_bf16, const float *x, const float *weight, const float *dres, float *dx, void *dx_bf16, float *dweight,
                           float *dbias, void *workspace, int rows, int C, float eps, halo_stream_t stream);
int halo_gelu_fwd(const float *a, float *y, size_t n, int exact, halo_stream_t stream);
int halo_gelu_bwd(const float *dy, const float *a, float *da, size_t n, int exact, halo_stream_t stream);
int halo_cross_entropy_fwd_lse(const float *logits, const int64_t *targets, float *loss, float *lse, int rows, int V,
                               long ld, long ignore_index, halo_stream_t stream);
int halo_cross_entropy_bwd(float *logits, const int64_t *targets, const float *lse, const float *grad,
                           long grad_stride, int rows, int V, long ld, long ignore_index, halo_stream_t stream);
int halo_embed_bwd(const int64_t *ids, const float *dx, float *dwte, float *dwpe, int B, int T, int C, int pos0,
                   int vocab, int accumulate_wpe, halo_stream_t stream);
/* x[n, :] += p[n % T, :]: tok_emb + pos_emb when both went through StableEmbedding's LayerNorm first (ha/attention.py:30-61,224) */
int halo_add_rows_bcast(float *x, const float *p, int rows, int T, int C, halo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Convolutional front-end of AudioEncoder, channels-last.  replaces, in ha/conv.py:
 *   halo_im2col_cl    the unfold of nn.Conv1d(input_dim, hidden_dim, 3, stride, padding=1) :29; the conv itself is a
 *                     GEMM of col [N*T', Cin*ks] with weight [Cout, Cin*ks] + bias + HALO_GEMM_GELU_ERF (:46)
 *   halo_dwconv1d_cl  DWConv1d.depthwise (groups = channels) :15-18; .pointwise :19 is a GEMM over the rows
 * x [N, T, C] -> [N, T', C] with T' = (T + 2*pad - ks)/stride + 1; no length masking (the reference has none). */
int halo_im2col_cl(const float *x, float *col, int N, int T, int Cin, int ks, int stride, int pad,
                   halo_stream_t stream);
/* gradient of the unfold (the fold): dx [N,T,Cin] from dcol [N*T', Cin*ks]; with it a dense Conv1d whose input is an activation
 * (ha/attention_audio.py:71 conv_subsample) back-propagates as GEMM + halo_col2im_cl */
int halo_col2im_cl(const float *dcol, float *dx, int N, int T, int Cin, int ks, int stride, int pad,
                   halo_stream_t stream);
int halo_dwconv1d_cl(const float *x, const float *weight, const float *bias, float *y, int N, int T, int C, int ks,
                     int stride, int pad, halo_stream_t stream);
/* gradient of halo_dwconv1d_cl: dx [N,T,C] (NULL to skip), dweight [C,ks], dbias [C] (NULL when bias-free); ks <= 8;
 * fixed-order partial sums through the workspace (bitwise reproducible) */
size_t halo_dwconv1d_cl_bwd_workspace_bytes(int C, int ks);
int halo_dwconv1d_cl_bwd(const float *dy, const float *x, const float *weight, float *dx, float *dweight,
                         float *dbias, void *workspace, int N, int T, int C, int ks, int stride, int pad,
                         halo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Token-tape batching (the data formats either side of the LM paths).  Integer gathers on
 * HBM-resident tapes, bit-exact.
 *   halo_tape_batch    SymbolTapeNoPad.__getitem__(i)  ha/symbol_tape.py:239-279: out [rows, batch_size] (row-major, same
 *                      element type as the tape: elem_bytes 1/2/4/8), out[t, b] = data[b*(tape_len-1) + i*bptt_len + t] with
 *                      tape_len = ceil(n_tokens / batch_size); positions past the tape get pad_value.  rows = bptt_len, or
 *                      the trailing token count for the last part.
 *   halo_lm_batch_u16  get_batch of ha/attention_loop.py:98-125 on a uint16 tape (np.memmap dtype uint16 :90): x[b, :] =
 *                      data[offsets[b] : offsets[b] + T] as int64, y = x shifted left with a 0 in the last column
 *                      (objective "lm"); objective_cond != 0 keeps only column (#non-zero x) - 2 of y ("cond"). */
int halo_tape_batch(const void *data, int elem_bytes, long n_tokens, int batch_size, int bptt_len, long part_index,
                    int rows, long pad_value, void *out, halo_stream_t stream);
int halo_lm_batch_u16(const uint16_t *data, long n_tokens, const int64_t *offsets, int B, int T, int objective_cond,
                      int64_t *x, int64_t *y, halo_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Optimizer step on flat buffers.
 * replaces: clip_grad_norm_ ha/loop.py:184 and torch.optim.AdamW(fused) ha/optim.py:137-139
 *   halo_sumsq: partials[0..HALO_SUMSQ_PARTS) = per-workgroup sums of x^2 (fixed order, so the
 *   norm is bitwise reproducible); halo_clip_coef: norm = sqrt(sum partials[0..count)),
 *   coef[0] = min(1, max_norm / (norm + 1e-6)) as clip_grad_norm_ computes it, coef[1] = 1; when
 *   the norm is not finite both are NaN, and halo_adamw skips the whole update when *grad_scale
 *   is NaN -- the on-device form of the reference's "skip batch" on NaN/Inf loss or gradient norm
 *   (ha/loop.py:167-174,185-189).  coef is a device float[2], norm_out a device float (or NULL).
 *   halo_adamw: torch.optim.AdamW single-tensor arithmetic; grad is multiplied by *grad_scale
 *   (device scalar, may be NULL) before use.  step >= 1 is the 1-based update count.
 * ------------------------------------------------------------------------------------------ */
#define HALO_SUMSQ_PARTS 1024
/* y = alpha*y + beta*x on flat buffers: accumulates micro-batch gradients (loss / accumulate; ha/loop.py:176-181) */
int halo_scale_add(float *y, const float *x, float alpha, float beta, size_t n, halo_stream_t stream);
/* the same with a device guard: when *guard (e.g. this micro-batch's loss) is not finite, x contributes nothing -- the reference
 * skips a micro-batch whose loss is NaN/Inf (ha/loop.py:167-174).  alpha == 0 never reads y (a poisoned sum cannot survive). */
int halo_scale_add_guarded(float *y, const float *x, float alpha, float beta, size_t n, const float *guard,
                           halo_stream_t stream);
/* wire format of the data-parallel gradient all-reduce (optional bf16 buckets, haloop_amd/dp.py): y_bf16[i] = bf16(x[i]);
 * y[i] = float(x_bf16[i]) * scale (scale = 1 / world after an all-reduce SUM) */
int halo_cast_f32_bf16(const float *x, void *y_bf16, size_t n, halo_stream_t stream);
int halo_cast_bf16_f32(const void *x_bf16, float *y, float scale, size_t n, halo_stream_t stream);
int halo_sumsq(const float *x, size_t n, float *partials, halo_stream_t stream);
/* The sharded data-parallel update (ha/attention_loop.py:154 semantics, ZeRO-1 shape; haloop_amd/dp.py): a rank owns a chunk of each
 * reduce-scattered span of the flat buffers.  n <= 8 ranges, each beginning and ending on a multiple of 4 elements.
 *   halo_sumsq_ranges:       partials[0..HALO_SUMSQ_PARTS) of the concatenation of x[begin[k], end[k]) (fixed order).
 *   halo_pack_ranges_bf16:   dst (bf16, contiguous) <- the concatenation of src[begin[k], end[k]) rounded to nearest-even: what a rank
 *                            sends in the all-gather of matrix parameters whose consumers only ever multiply by their bf16 values.
 *   halo_expand_ranges_bf16: the inverse on the gathered buffer.  stage holds `world` records of sum(chunk[k]) bf16 values, rank-major;
 *                            span k of dst begins at begin[k] and consists of `world` chunks of chunk[k] elements; every chunk but
 *                            skip_rank's (the caller's own fp32 master values; -1: none) is overwritten with the gathered values. */
/* Round 5: direct peer exchange for the data-parallel step (csrc/dp_direct.hip; replaces the NCCL collectives
 * DistributedDataParallel issues, ha/attention_loop.py:154,203, with writes over the point-to-point xGMI links).
 *   halo_dx_alloc / open / close / free: an ARENA of device memory peers can write (uncached where the runtime allows) and its 64-byte HIP
 *       IPC handle; a peer maps it with halo_dx_open.  The arena starts zeroed; its first 4096 bytes are flag blocks (64 bytes each).
 *   halo_dx_push:   piece p of src (piece_bytes each, piece_stride_bytes apart; same != 0: src itself for every peer) -> peer p's arena at
 *       dst_offset + rank * piece_bytes, every peer p != rank, one launch.  16-byte granules.
 *   halo_dx_signal: epoch -> word `rank` of the flag block at flag_offset of every peer's arena (system-scope release).
 *   halo_dx_wait:   one workgroup waits until every peer's word of own_flags carries epoch (or later); BOUNDED (5 s): a timeout raises the
 *       status word of halo_set_status_word and returns.
 *   halo_dx_reduce: own [elems] <- scale * (sum in RANK ORDER of own (at position rank) and the inbox pieces p != rank, elems apart).
 * peer_bases: `world` (<= 16) mapped arena pointers, own rank's the local one. */
int halo_dx_alloc(size_t bytes, void **ptr, void *handle64);
int halo_dx_open(const void *handle64, void **ptr);
int halo_dx_close(void *ptr);
int halo_dx_free(void *ptr);
int halo_dx_push(const void *src, size_t piece_bytes, size_t piece_stride_bytes, int same, void *const *peer_bases, size_t dst_offset, int world,
                 int rank, halo_stream_t stream);
int halo_dx_signal(void *const *peer_bases, size_t flag_offset, int world, int rank, uint32_t epoch, halo_stream_t stream);
int halo_dx_wait(const void *own_flags, int world, int rank, uint32_t epoch, halo_stream_t stream);
int halo_dx_reduce(float *own, const float *inbox, size_t elems, int world, int rank, float scale, halo_stream_t stream);

int halo_sumsq_ranges(const float *x, int n, const size_t *begin, const size_t *end, float *partials, halo_stream_t stream);
int halo_pack_ranges_bf16(const float *src, int n, const size_t *begin, const size_t *end, void *dst, halo_stream_t stream);
int halo_expand_ranges_bf16(const void *stage, int n, const size_t *begin, const size_t *chunk, int world, int skip_rank, float *dst,
                            halo_stream_t stream);
int halo_clip_coef(const float *partials, int count, float max_norm, float *coef, float *norm_out,
                   halo_stream_t stream);
/* halo_clip_coef that also advances a device-side update counter (uint32) when the norm is finite: torch's AdamW step count
 * advances only with an applied update (the reference skips optimizer.step() on a non-finite norm, ha/loop.py:185-189).
 * A raised status word (halo_set_status_word) counts as a non-finite norm: the update is skipped on the device. */
int halo_clip_coef_step(const float *partials, int count, float max_norm, float *coef, float *norm_out,
                        uint32_t *applied_steps, halo_stream_t stream);
int halo_adamw(float *p, const float *g, float *m, float *v, size_t n, float lr, float beta1, float beta2,
               float eps, float weight_decay, int step, const float *grad_scale, halo_stream_t stream);
/* halo_adamw over n_ranges (<= 8) contiguous element ranges [begin[r], end[r]) of the same flat buffers in ONE launch (host
 * arrays; offsets multiples of 4): range r has its own weight_decay[r] and device gradient scale grad_scale[r] (NULL: 1).
 * counter (optional, device): incremented once -- the dropout step counter of halo_counter_inc rides along. */
int halo_adamw_ranges(float *p, const float *g, float *m, float *v, int n_ranges, const size_t *begin, const size_t *end,
                      const float *weight_decay, const float *const *grad_scale, float lr, float beta1, float beta2,
                      float eps, int step, uint32_t *counter, halo_stream_t stream);

/* halo_adamw_ranges with the 1-based update count read from the device (the counter halo_clip_coef_step advances), so that a
 * whole training step -- optimizer included -- has no host-side scalar and replays from one hipGraph. */
/* lr_dev (optional, device float): the learning rate is read from there instead of ``lr`` -- a schedule (the reference applies
 * lr.apply_lr_ every step, ha/loop.py:191) then reaches a captured launch: the caller writes the device scalar between replays. */
int halo_adamw_ranges_dev(float *p, const float *g, float *m, float *v, int n_ranges, const size_t *begin, const size_t *end,
                          const float *weight_decay, const float *const *grad_scale, float lr, const float *lr_dev, float beta1,
                          float beta2, float eps, const uint32_t *step_dev, uint32_t *counter, halo_stream_t stream);

/* halo_adamw over MANY tensors in one launch (what torch.optim.AdamW(fused=True) does for a parameter list, ha/attention_loop.py:
 * 141-147).  tensor_table (device, built once): n_tensors records of halo_adamw_multi_tensor_bytes() bytes each:
 *   { float *p; float *m; float *v; uint64 n; float weight_decay; int32 pad }
 * chunk_table (device, built once): n_chunks pairs { uint32 tensor, uint32 chunk } -- every tensor cut into
 * ceil(n / halo_adamw_multi_chunk()) chunks, one workgroup each.  grads: HOST array of the n_tensors gradient pointers of this
 * step (autograd allocates new ones every backward); they are copied into the kernel arguments, so the caller may reuse the
 * array at once.  n_tensors <= halo_adamw_multi_max_tensors() per call.  lr may change from call to call (LR schedules,
 * ha/optim.py:68-72): the decay factor 1 - lr*weight_decay is formed in the kernel.  Bit-identical to one halo_adamw call per
 * tensor. */
size_t halo_adamw_multi_tensor_bytes(void);
unsigned halo_adamw_multi_chunk(void);
int halo_adamw_multi_max_tensors(void);
int halo_adamw_multi(const void *tensor_table, const void *chunk_table, int n_chunks, const float *const *grads, int n_tensors,
                     float lr, float beta1, float beta2, float eps, int step, const float *grad_scale, halo_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HALO_H */
