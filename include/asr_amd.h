/*
 * asr_amd.h -- C ABI of libasr_amd.so: the MI355X (gfx950) kernels behind the
 * AttModel training hot path of syoamakase/ASR.
 *
 * The reference has no FFI of its own: its seam is the nn.Module / autograd
 * contract.  Every entry point below therefore cites
 * the reference *Python* lines whose arithmetic it replaces; the Python
 * mirror in asr_amd/Models and asr_amd/Loss binds them through ctypes
 * (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain pointers + sizes only; all pointers are DEVICE pointers unless
 *     the name ends in _host.  float = IEEE fp32, lengths/targets = int64.
 *   - the library never allocates or frees device memory; callers pass
 *     outputs, saved activations and workspaces.  Kernels are enqueued on
 *     `stream` (a hipStream_t passed as void*) and the call returns without
 *     synchronising.
 *   - return 0 on success, a negative code on error; asr_last_error() gives
 *     the message (thread-local).  Nothing throws across the ABI.
 */
#ifndef ASR_AMD_H
#define ASR_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASR_OK 0
#define ASR_ERR_ARG (-1)
#define ASR_ERR_HIP (-2)
#define ASR_ERR_WORKSPACE (-3)

int asr_version(void);
/* timing experiments only: non-zero flags make the persistent kernels skip their
 * in-launch waits (results are then wrong).  The product path never calls it. */
void asr_debug_set(int flags);
const char *asr_last_error(void);

/* ------------------------------------------------------------------ loss --
 * Replaces Loss/label_smoothing.py:14-28 (log_softmax + dense smoothed
 * one-hot + per-utterance python loop) and its autograd backward.
 *   logits   (B,L,V) f32        targets (B,L) i64       text_lengths (B) i64
 *   row_loss (B*L)   f32 scratch: weighted per-row loss
 *   loss     (1)     f32 out
 *   dlogits  (B,L,V) f32 out: d loss / d logits (rows past text_length = 0);
 *            may be NULL (forward only)
 */
int asr_ls_ce_fwd_bwd(const float *logits, const int64_t *targets, const int64_t *text_lengths,
                      int B, int L, int V, float eps, int T_norm, int B_norm,
                      float *row_loss, float *loss, float *dlogits, void *stream);

/* ------------------------------------------------------------------- ctc --
 * The CTC head's loss: replaces train.py:56-57,
 *   predict_ts = F.log_softmax(predict_ts, dim=2).transpose(0, 1)
 *   loss = F.ctc_loss(predict_ts, text, mel_lengths, text_lengths, blank=0)
 * (reduction 'mean': mean over the batch of nll_b / max(target_length_b, 1); zero_infinity
 * False) and its backward through the log_softmax, in one call.
 *   logits (B,T,V) f32, the output of CTCModel.forward (Models/CTCModel.py:19-23)
 *   targets (B,L) i64 (may be NULL when L == 0); input_lengths, target_lengths (B) i32, all on the device
 *   ws      asr_ctc_ws_floats(B,T,L) floats of scratch
 *   nll     (B) f32 out: per-utterance negative log likelihood (reduction 'none')
 *   loss    (1) f32 out
 *   dlogits (B,T,V) f32 out: d loss / d logits, 0 for frames t >= input_lengths[b]
 * L <= 511.  An utterance whose labels do not fit its frames has nll = +inf, as in torch.
 */
int64_t asr_ctc_ws_floats(int B, int T, int L);
int asr_ctc_loss_fwd_bwd(const float *logits, const int64_t *targets, const int32_t *input_lengths,
                         const int32_t *target_lengths, int B, int T, int V, int L, int blank,
                         float *ws, float *nll, float *loss, float *dlogits, void *stream);

/* ------------------------------------------------------------- features --
 * Device-side input pipeline: replaces, for a whole padded batch,
 * datasets.py:110-112 (x = (x - mean) / sqrt(var) per channel), datasets.py:52-81,118
 * (SpecAugment: nF frequency masks [f0,f1) filled with the mean of the utterance as masked so
 * far -- or 0 with fill_zero -- then nT time masks [t0,t1) filled with 0), the zero padding of
 * collate_fn (datasets.py:225-247) and frame_stacking (utils/utils.py:47-65).
 *   raw (B,Tm,F) f32 padded batch; mel_lengths (B) i32; mean, var (F) f32 or both NULL
 *   fmask (B,nF,2), tmask (B,nT,2) i32 device lists of [begin,end) (empty ranges allowed);
 *   the positions are drawn on the host exactly as the reference draws them
 *   ws  asr_featurize_ws_floats(B,Tm,F,nF) floats;  out (B, Tm/stack, F*stack) f32
 */
int64_t asr_featurize_ws_floats(int B, int Tm, int F, int nF);
int asr_featurize(const float *raw, const int32_t *mel_lengths, const float *mean, const float *var,
                  const int32_t *fmask, int nF, int fill_zero, const int32_t *tmask, int nT, int stack,
                  int B, int Tm, int F, float *ws, float *out, void *stream);

/* ------------------------------------------------------------------ gemm --
 * fp32 MFMA GEMM used for every hoisted matrix product of the path:
 * the per-layer LSTM input projections and their gradients (the GEMMs inside
 * torch.nn.LSTM, Models/encoder.py:25-33), L_he(hbatch) (Models/attention.py:30,
 * hoisted out of the decoder loop), L_yy (Models/decoder.py:55) and the
 * weight gradients of all decoder Linear layers.
 *
 *   C[m,n] (+)= alpha * sum_k A(m,k) * B(n,k)  (+ bias1[n] + bias2[n])
 *
 * a_mode 0: A(m,k) at A[rowoff_a(m) + k]      (k contiguous)
 * a_mode 1: A(m,k) at A[rowoff_a(k) + m]      (m contiguous, "A transposed")
 * b_mode 0: B(n,k) at B[rowoff_b(n) + k]      (k contiguous, weight[n][k])
 * b_mode 1: B(n,k) at B[rowoff_b(k) + n]      (n contiguous)
 * C(m,n)    at C[rowoff_c(m) + n]
 * rowoff_x(i) = (i / x_d) * x_s1 + (i % x_d) * x_s0   (two-level row stride, in
 * elements; lets time-major and batch-major activations share one kernel).
 * nbatch > 1 runs independent problems at pointer strides batch_{a,b,c}.
 * splitk > 1 splits K over blocks and accumulates with float atomics: C must
 * then hold the initial value (accumulate is implied).
 * Row lists (device int32, may be NULL) let a ragged batch skip its padded frames the way
 * pack_padded_sequence does for nn.LSTM (Models/encoder.py:32): the products then run over the
 * listed rows only.
 *   m_idx (a_mode 0 only): M counts list entries; entry m names the row of A and of C:
 *          A(m,k) at A[rowoff_a(m_idx[m]) + k], C(m,n) at C[rowoff_c(m_idx[m]) + n].
 *          Rows of C that are not listed are left untouched.
 *   k_idx (a_mode 1 and b_mode 1 only): K counts list entries; entry k names the row of both
 *          operands: A(m,k) at A[rowoff_a(k_idx[k]) + m], B(n,k) at B[rowoff_b(k_idx[k]) + n].
 * Every entry must be a row inside the operand; the library cannot check device data.
 */
typedef struct {
    const float *A, *B;
    float *C;
    int M, N, K;
    int a_mode, b_mode;
    int64_t a_s0, a_s1; int a_d;
    int64_t b_s0, b_s1; int b_d;
    int64_t c_s0, c_s1; int c_d;
    const float *bias1, *bias2;
    float alpha;
    int accumulate;     /* C += result (read-modify-write) */
    int act;            /* 0 none, 1 tanh, 2 relu applied to the stored value (splitk must be 1) */
    int splitk;
    int nbatch;
    int64_t batch_a, batch_b, batch_c;
    const int32_t *m_idx, *k_idx;   /* row lists, see above (nbatch must be 1 with either) */
    /* bias_d > 0: the bias vectors are two-level like the rows, column n reads
     * bias1[(n / bias_d) * bias1_s1 + n % bias_d] (resp. bias2, bias2_s1): the forward and the
     * reverse direction's biases of an nn.LSTM layer are separate tensors, one launch serves both */
    int bias_d;
    int64_t bias1_s1, bias2_s1;
    /* c_keep != NULL (splitk 1): the stored value becomes c_keep[off] ? v * c_keep_scale : 0 with
     * off the element offset of C(m,n): the backward of the inverted dropout that produced the
     * GEMM's consumer (nn.LSTM(dropout=p), Models/encoder.py:26), fused into the dX epilogue */
    const uint8_t *c_keep;
    float c_keep_scale;
    /* batch_d > 0: two-level batch index; batch n addresses operand X at
     * X + (n % batch_d) * batch_x + (n / batch_d) * batch_x1 -- the (utterance, head) pairs of the
     * relative-position attention (Models/Transformer_encoder.py:97-98,136), whose heads are
     * interleaved inside the (B, T, heads * d_k) projections */
    int batch_d;
    int64_t batch_a1, batch_b1, batch_c1;
    /* bit 0 (A) / bit 1 (B), mode 1 operands only: every row is readable up to the next multiple of
     * 4 elements past M (resp. N) -- e.g. attention maps stored with a padded row stride -- so the
     * 16-byte loads may be used although M (N) is not a multiple of 4.
     * bit 2, A in mode 0 (k contiguous): every row of A is readable AND ZERO from K up to the next multiple of 4 -- the
     * attention maps again, as the left operand of attn v / d ac k / d bd p (Models/Transformer_encoder.py:136), whose
     * padding columns the softmax kernels keep at zero -- so K need not be a multiple of 4 for the 16-byte path (the
     * rows of a mode-1 B operand are taken one k at a time and stop at K exactly) */
    int pad_ok;
    /* optional: bytes readable from A (resp. B), counted from the pointer in this descriptor (0 = unknown).  When both
     * are given and below 2 GiB the bf16 kernels fetch the operands with range-checked buffer loads (no per-load
     * address arithmetic or masks); results are identical either way. */
    int64_t a_span, b_span;
    /* != 0: the operand holds pre-split elements, one 32-bit word {bf16 hi : bf16 lo} each (asr_pack_hilo) instead of
     * fp32 -- the bf16x3 split done once by a producer instead of by every tile that reads the element.  Only with
     * asr_gemm_precision 1 and 128 x 128 tiles (>= 320 of them, or >= 256 with splitk > 1), and either a_mode 0 with known
     * spans and b_packed alone or both (a_packed alone has no kernel), or a_mode 1 / b_mode 1 (a weight gradient) with
     * b_packed and a k list or known spans: anything else is rejected, ask asr_amd/ops.py::packed_ok / packed_dw_ok.
     * Bit-identical results. */
    int a_packed, b_packed;
} asr_gemm_desc;

int asr_sgemm(const asr_gemm_desc *d, void *stream);
/* Arithmetic of asr_sgemm's products (the reference multiplies in fp32 everywhere, Models/encoder.py:25-26):
 *   0  exact fp32 products on v_mfma_f32_32x32x2_f32 (bit for bit an fmaf chain);
 *   1  bf16x3: every fp32 operand split into two bf16 (hi + lo), three products per term on the bf16 matrix
 *      pipe, fp32 accumulate -- ~16 mantissa bits per product, inside the fp32 parity bar (1e-4 / 1e-3);
 *   2  plain bf16 products (8 mantissa bits), fp32 accumulate: the opt-in reduced-precision path, with a
 *      looser, stated tolerance (tests/test_gpu_parity.py::test_attmodel_gemm_precision_modes).
 * Returns the previous mode; a mode outside 0..2 only queries.  Start-up value: environment variable
 * ASR_GEMM_PRECISION = f32 | bf16x3 | bf16 (default bf16x3).  Shapes outside the fast path (unaligned
 * operands, K < 32) use mode 0 whatever the setting. */
int asr_gemm_precision(int mode);
/* dst[i] = {bf16 hi : bf16 lo} of src0[i] for i < n, of src1[i - n] for n <= i < 2n (src1 may be NULL: n words only):
 * the pre-split form asr_gemm_desc.a_packed / b_packed reads; two sources so that the forward and the reverse
 * direction's weight_ih of an nn.LSTM layer (Models/encoder.py:25) land in one contiguous [2][4H][D] operand. */
int asr_pack_hilo(const float *src0, const float *src1, int64_t n, float *dst, void *stream);
/* asr_dropout_fwd (below: the same mask and values for the same p and seed) whose output is written as {hi:lo} words:
 * the dropout between two nn.LSTM layers (Models/encoder.py:26) feeds only matrix products -- the next layer's input
 * projection (a_packed) and, in the backward, its dW_ih (b_packed) -- which then read it without splitting again. */
int asr_dropout_fwd_packed(const float *x, float *y_packed, uint8_t *mask, int64_t n, float p, uint64_t seed,
                           void *stream);

/* small helpers used by the backward passes (bias gradients, layouts) */
/* out[n] (+)= sum_m A[rowoff(m) + n]; accumulate != 0 keeps the old value.
 * part: asr_colsum_ws_floats(M,N) floats of scratch (may be NULL for small M);
 * two-stage and atomic-free, so the sum is deterministic. */
int asr_colsum(const float *A, int M, int N, int64_t s0, int64_t s1, int d,
               float *out, int accumulate, float *part, void *stream);
int64_t asr_colsum_ws_floats(int M, int N);
/* The same sum for up to ASR_COLSUM_LIST_MAX contiguous row-major matrices of equal height M
 * (src[i] is M x n[i], out[i] gets n[i] sums; off is filled in by the library) in two launches: the
 * bias gradients of one block of the Conformer backward (Models/Transformer_encoder.py, every
 * nn.Linear / nn.Conv1d bias and pos_bias_u / pos_bias_v).  part: asr_colsum_multi_ws_floats floats. */
#define ASR_COLSUM_LIST_MAX 8
typedef struct {
    int count;
    const float *src[ASR_COLSUM_LIST_MAX];
    float *out[ASR_COLSUM_LIST_MAX];
    int n[ASR_COLSUM_LIST_MAX];
    int64_t off[ASR_COLSUM_LIST_MAX];
} asr_colsum_list;
int64_t asr_colsum_multi_ws_floats(const asr_colsum_list *l, int M);
int asr_colsum_multi(const asr_colsum_list *l, int M, float *part, void *stream);
/* out[c*R + r] = in[r*C + c], nbatch matrices back to back */
int asr_transpose(const float *in, float *out, int R, int C, int nbatch, void *stream);
/* table[v][:] = sum_{i: idx[i]==v} src[i][:]  for every v < n_table_rows (embedding
 * gradient of L_ys, Models/decoder.py:57); deterministic, overwrites table. */
int asr_scatter_rows(const float *src, const int64_t *idx, int n_rows, int width,
                     int64_t src_stride, float *table, int n_table_rows, void *stream);
/* out = dU * (1 - U*U): gradient through u = tanh(.) of Models/decoder.py:55 */
int asr_mul_dtanh(const float *dU, const float *U, float *out, int64_t n, void *stream);
/* x *= s (1/world_size after the gradient all-reduce: DDP averages, train.py:141) */
int asr_scale(float *x, int64_t n, float s, void *stream);
/* *value = NaN when *flag_word != 0 (one 4-byte word on the device, e.g. word ASR_BLSTM_ERR_WORD of a
 * recurrence workspace): surfaces a timed-out in-launch wait through the loss the training loop
 * already reads every step -- the reference ends the job on a NaN loss, train.py:66-68. */
int asr_poison_if_flag(const void *flag_word, float *value, void *stream);
/* A HIP stream with a hardware queue of its own at normal priority (created with a compute-unit
 * mask: bit i of cu_mask = compute unit i; NULL / 0 words = every CU), for work that overlaps the
 * backward's critical path -- the reference leaves stream handling to DDP's reducer (train.py:141).
 * Optional: asr_amd/streams.py uses it only under ASR_SIDE_STREAM_KIND=cumask; its default helper
 * streams are plain normal-priority streams created before the process group (streams.prepare).
 * Destroy it with asr_stream_destroy once its work has completed. */
int asr_stream_create(void **stream, const uint32_t *cu_mask, int n_words);
int asr_stream_destroy(void *stream);
/* Tests and timing experiments only (no counterpart in the reference, never on the product path): n_wg workgroups
 * of `threads` lanes that hold lds_bytes (<= 64 KiB) of LDS each and do nothing for usec (<= 100000) microseconds of
 * wall time -- what a collective's kernel (one resident workgroup per RCCL channel) looks like to the kernels that
 * share the GPU with it.  tests/test_gpu_parity.py::test_step_under_competing_kernels runs a training step under it. */
int asr_debug_occupy(int n_wg, int threads, int lds_bytes, int usec, void *stream);
/* out[r] = argmax_c x[r*ld + c], first maximum (the frame-wise labels of CTCModel.decode,
 * Models/CTCModel.py:33: decoder_output[b].argmax(dim=1)) */
int asr_argmax_rows(const float *x, int64_t rows, int cols, int64_t ld, int64_t *out, void *stream);

/* ------------------------------------------------- optimizer step (next) --
 * The caller-side tail of the step, train.py:60-63 with the optimizer built at train.py:146:
 *   torch.nn.utils.clip_grad_norm_(model.parameters(), hp.clip)     -> asr_grad_sqnorm
 *   torch.optim.Adam(lr, weight_decay=1e-5).step()                   -> asr_clip_adam_step
 * A tensor list travels by value (device pointers + element counts, at most
 * ASR_TENSOR_LIST_MAX per call; longer lists are passed in several calls, the squared norm
 * accumulated with accumulate != 0).  Nothing synchronises with the host: the clip
 * coefficient min(1, max_norm / (sqrt(norm2) + 1e-6)) is computed on the device. */
#define ASR_TENSOR_LIST_MAX 48
typedef struct asr_tensor_list {
    int count;
    float *p[ASR_TENSOR_LIST_MAX];
    int64_t n[ASR_TENSOR_LIST_MAX];
} asr_tensor_list;
/* norm2[0] (+)= sum over all listed tensors of the sum of squares (deterministic two-stage
 * sum; ws: asr_sqnorm_ws_floats() floats of scratch). */
int64_t asr_sqnorm_ws_floats(void);
int asr_grad_sqnorm(const asr_tensor_list *grads, float *norm2, int accumulate, float *ws, void *stream);
/* One Adam step (torch.optim.Adam semantics: coupled L2 weight decay, bias correction, no
 * amsgrad) on every listed tensor, with the gradients first scaled by the clip coefficient
 * (max_norm <= 0: no clipping, norm2 may be NULL) and written back scaled, as clip_grad_norm_
 * does.  step counts from 1.  Hyper-parameters are doubles (Python floats in torch): 1 - beta
 * and the bias corrections are formed in double and rounded to fp32 once, as torch does. */
int asr_clip_adam_step(const asr_tensor_list *params, const asr_tensor_list *grads,
                       const asr_tensor_list *exp_avg, const asr_tensor_list *exp_avg_sq,
                       const float *norm2, double max_norm, double lr, double beta1, double beta2,
                       double eps, double weight_decay, int64_t step, void *stream);

/* ----------------------------------------------------------------- BLSTM --
 * One bidirectional LSTM layer over a ragged batch WITHOUT packing: the
 * recurrence of torch.nn.LSTM as called by Models/encoder.py:32-34
 * (pack_padded_sequence(enforce_sorted=False) -> LSTM -> pad_packed_sequence).
 * Gate order i,f,g,o.  Sequence b runs over lengths[b] frames; the reverse
 * direction starts at frame lengths[b]-1 from a zero state; output rows at
 * t >= lengths[b] are exactly 0.
 *
 * Layouts (time-major inside the encoder):
 *   G     [T][B][2][4H] in : x_t W_ih^T + b_ih + b_hh for dir 0 / dir 1
 *                      out: activated gates (saved for backward)
 *   Cs    [T][B][2][H]  out: cell state c_t (0 at padded positions)
 *   Hout  element (t,b,dir*H+j) at t*ho_t + b*ho_b + dir*H + j   (2H wide rows)
 *   W_hh  [2][4H][H]   (dir 0 = weight_hh_l{k}, dir 1 = weight_hh_l{k}_reverse)
 *   lengths (B) int32 on device
 *   ws    asr_blstm_ws_floats(B,H) floats, 16-byte aligned, ZERO on first use: flag
 *         words + the h exchange of the persistent (one launch per layer) kernels;
 *         word 1024 is a sticky error word a kernel sets if an in-launch wait
 *         timed out.  ws == NULL selects the one-launch-per-time-step kernels.
 */
int64_t asr_blstm_ws_floats(int B, int H);
/* Arithmetic of the recurrent products inside the persistent kernels (the reference: fp32, torch.nn.LSTM):
 * 0 exact fp32 products, 1 the bf16x3 split on the bf16 matrix pipe (fp32 accumulate; H % 32 == 0 in the forward,
 * other shapes stay fp32).  Returns the previous mode; other values only query.  Start-up value: environment
 * variable ASR_BLSTM_PRECISION = f32 | bf16x3. */
int asr_blstm_precision(int mode);
/* Grid of the persistent recurrence kernels at 17..32 batch rows (no counterpart in the reference: a placement choice).
 * 1 (default): one (direction, batch tile) exchange group per XCD, hand-off through that XCD's L2 with plain stores
 * once the group has verified from HW_REG_XCC_ID that it shares an XCD (write-through stores otherwise); 0: the
 * three-dimensional grid, write-through stores only.  Bit-identical results.  Returns the previous setting; other
 * values only query.  Start-up value: environment variable ASR_BLSTM_XCD = 0 | 1. */
int asr_blstm_xcd_groups(int mode);
/* With the XCD-grouped grid, the workgroups of the unused XCD labels touch the saved activations (gates, cell states,
 * upstream gradient) `dist` time steps ahead of the recurrence, so that its own loads find them in the memory-side
 * cache instead of HBM (no counterpart in the reference; results are unaffected).  dist = 0 switches it off; returns
 * the previous value, a negative dist only queries.  Start-up value: environment variable ASR_BLSTM_PREFETCH (16). */
int asr_blstm_prefetch(int dist);
int asr_blstm_fwd(const float *W_hh, float *G, float *Cs, float *Hout,
                  int64_t ho_t, int64_t ho_b, const int32_t *lengths,
                  int T, int B, int H, float *ws, void *stream);
/* The same recurrence (Models/encoder.py:33), steps [s0, s1) only: step s is frame s of the
 * forward direction and frame T-1-s of the reverse direction.  s0 = 0 starts a layer; a call
 * with s0 > 0 continues the previous call on the same G/Cs/Hout/ws (nothing else may use ws
 * in between).  Lets the host overlap the next layer's input projection of the frames that
 * are already final with the rest of the recurrence. */
int asr_blstm_fwd_range(const float *W_hh, float *G, float *Cs, float *Hout,
                        int64_t ho_t, int64_t ho_b, const int32_t *lengths,
                        int T, int B, int H, int s0, int s1, float *ws, void *stream);
/* asr_blstm_fwd without a workspace argument and without any allocation.  NO COUNTERPART in the
 * reference: it exists for callers that cannot allocate device memory -- code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  The flag words and the h exchange
 * of the persistent kernel live in a file-scope device array of asr_blstm_fwd_noalloc_max_floats()
 * floats (264.25 KiB, one copy per GPU the library runs on, held until the process ends); shapes
 * beyond it (B > 32, H > 512, H % 4 != 0) are rejected with ASR_ERR_ARG.  Every call on a GPU
 * shares that GPU's array: calls that overlap on one GPU (two streams) race.  Same results as
 * asr_blstm_fwd, bit for bit.  The stream must belong to the current device (checked).
 * asr_blstm_fwd_noalloc_err_word(): device address (current device) of the sticky error word inside the
 * static array -- the counterpart of word 1024 of a workspace: non-zero once a launch's in-kernel wait has
 * timed out, after which that call's Hout / Cs are garbage although it returned ASR_OK (launches are
 * asynchronous).  Pass it to asr_poison_if_flag on the same stream, or copy the 4 bytes back; the library
 * never clears it.  NULL on failure (asr_last_error). */
int64_t asr_blstm_fwd_noalloc_max_floats(void);
const void *asr_blstm_fwd_noalloc_err_word(void);
int asr_blstm_fwd_noalloc(const float *W_hh, float *G, float *Cs, float *Hout,
                          int64_t ho_t, int64_t ho_b, const int32_t *lengths,
                          int T, int B, int H, void *stream);

/* Backward of the recurrence.
 *   W_hh  [2][4H][H]   as in the forward (used by the persistent kernel; may be NULL)
 *   W_hhT [2][H][4H]   transposed recurrent weights (asr_transpose)
 *   G     in: activated gates; out: d loss / d pre-activation gates (0 at padding)
 *   dHout upstream gradient, element (t,b,dir*H+j) at t*dho_t + b*dho_b + dir*H + j
 *   dc_ws [2][B][H] scratch
 * The weight/input gradients are plain GEMMs over G afterwards (asr_sgemm).
 */
/* 1 if asr_blstm_bwd / asr_blstm_bwd_range read W_hhT for this shape, 0 if only W_hh is read (the reduce-scatter
 * form): the caller may then pass W_hh in both places and skip the transposition (a 36 us kernel on the critical
 * path of every layer at the config-2 size).  A call that does need W_hhT rejects W_hhT == W_hh. */
int asr_blstm_bwd_needs_transposed(int T, int B, int H, int64_t dho_t, int64_t dho_b, int have_ws);
int asr_blstm_bwd(const float *W_hh, const float *W_hhT, float *G, const float *Cs, const float *dHout,
                  int64_t dho_t, int64_t dho_b, const int32_t *lengths,
                  float *dc_ws, int T, int B, int H, float *ws, void *stream);
/* Steps [s0, s1) of the same backward (autograd of Models/encoder.py:33): step s is frame
 * T-1-s of the forward direction and frame s of the reverse direction; after the call the
 * rows of G for those frames hold the final pre-activation gradients.  s0 > 0 continues the
 * previous call on the same G/dc_ws/ws. */
int asr_blstm_bwd_range(const float *W_hh, const float *W_hhT, float *G, const float *Cs, const float *dHout,
                        int64_t dho_t, int64_t dho_b, const int32_t *lengths,
                        float *dc_ws, int T, int B, int H, int s0, int s1, float *ws, void *stream);

/* inverted dropout between encoder layers (nn.LSTM dropout=, Models/encoder.py:26):
 * y = x * mask / (1-p), mask ~ Bernoulli(1-p) from a counter-based hash of
 * (seed, element index); mask (uint8) is saved.  Backward: dx = dy * mask/(1-p). */
int asr_dropout_fwd(const float *x, float *y, uint8_t *mask, int64_t n, float p,
                    uint64_t seed, void *stream);
int asr_dropout_bwd(const float *dy, const uint8_t *mask, float *dx, int64_t n, float p,
                    void *stream);
/* The same dropout (Models/encoder.py:26) on a rows x cols block of a larger row-major tensor
 * with leading dimension ld: x/y/mask point at the block's first element, `first` is that
 * element's index in the whole tensor, so the block gets exactly the mask asr_dropout_fwd
 * would give it. */
int asr_dropout_fwd_block(const float *x, float *y, uint8_t *mask, int64_t rows, int64_t cols, int64_t ld,
                          int64_t first, float p, uint64_t seed, void *stream);

/* ------------------------------------------ Conformer encoder (next, 8f rank 3) --
 * Models/Transformer_encoder.py:441-451 (TransformerEncoder = CNN_embedding + ConformerEncoder),
 * entered at Models/AttModel.py:17,24.  Activations are (B, T, C) row-major throughout.  The matrix
 * products (every nn.Linear, the 1x1 Conv1d, the attention products, the Conv2d as col @ W^T) are
 * asr_sgemm calls; these entry points are the rest. */
/* nn.LayerNorm(d) (:161,194,218,250), d <= 1024: y, and per row mean / rstd for the backward */
int asr_layernorm_fwd(const float *x, const float *gamma, const float *beta, float *y, float *mean,
                      float *rstd, int64_t rows, int d, float eps, void *stream);
/* dx = its gradient (+ res when res != NULL: the gradient arriving over the residual connection);
 * part[asr_layernorm_bwd_blocks(rows)][2][d] = per-block sums of dy*xhat (d gamma) and dy (d beta),
 * to be summed with asr_colsum (deterministic) */
int64_t asr_layernorm_bwd_blocks(int64_t rows);
int asr_layernorm_bwd(const float *dy, const float *x, const float *gamma, const float *mean,
                      const float *rstd, float *dx, float *part, int64_t rows, int d, const float *res,
                      void *stream);
/* Swish (:175-177), GLU over the halves of a (rows, 2d) tensor (:232-233), ReLU gradient (:391-394),
 * out = a*x + b*y (residuals :259,263-265; running statistics), qu = q + u / qv = q + v (:94-95) */
/* (mask != NULL: followed by the inverted dropout of :205 in the same pass, mask as in asr_dropout_fwd) */
int asr_swish_fwd(const float *x, float *y, uint8_t *mask, int64_t n, float p, uint64_t seed, void *stream);
int asr_swish_bwd(const float *dy, const float *x, const uint8_t *mask, float *dx, int64_t n, float p,
                  void *stream);
/* y = x + factor * keep * o, keep = AND of ndrop dropout draws (seeds seed, seed+1, ...), written to mask:
 * a block's output dropouts (:207 and :263-265) and residual sum in one pass (factor carries the
 * 1/(1-p)^ndrop and the 0.5 of :259); mask == NULL: no dropout.  asr_scale_mask is its gradient
 * w.r.t. o: out = mask ? dy * factor : 0. */
int asr_residual_dropout(const float *x, const float *o, float *y, uint8_t *mask, int64_t n, float factor,
                         float p, uint64_t seed, int ndrop, void *stream);
int asr_scale_mask(const float *dy, const uint8_t *mask, float *out, int64_t n, float factor, void *stream);
int asr_glu_fwd(const float *x, float *y, int64_t rows, int d, void *stream);
int asr_glu_bwd(const float *dy, const float *x, float *dx, int64_t rows, int d, void *stream);
int asr_relu_bwd(const float *dy, const float *y, float *dx, int64_t n, void *stream);
int asr_axpby(const float *x, const float *y, float *out, int64_t n, float a, float b, void *stream);
int asr_add_two_bias(const float *q, const float *u, const float *v, float *qu, float *qv, int64_t rows,
                     int d, void *stream);
/* DepthwiseConv.conv (:183,187-188): y[b,t,c] = bias[c] + sum_k w[c,k] x[b,t+k-15,c], K = 31;
 * flip != 0 (bias NULL) gives the data gradient.  asr_dwconv_bwd_w writes
 * part[asr_dwconv_bwd_w_chunks(B,T)][K][C], to be summed with asr_colsum into d w as (K, C). */
int asr_dwconv(const float *x, const float *w, const float *bias, float *y, int B, int T, int C, int K,
               int flip, void *stream);
int64_t asr_dwconv_bwd_w_chunks(int B, int T);
int asr_dwconv_bwd_w(const float *dy, const float *x, float *part, int B, int T, int C, int K, void *stream);
/* nn.BatchNorm1d (:223,235) + Swish (:236) over rows = B*T frames.  msum / vsum: per-channel sum of
 * x and of (x - mean)^2 with multipliers ms / vs (training: 1/rows; eval: the running mean / var with
 * multipliers 1).  asr_bn_var_part and asr_bn_swish_bwd_part write asr_bn_blocks(rows) partial rows
 * (C resp. 2C wide: sum du, sum du*xhat = d beta, d gamma) for asr_colsum.  sums == NULL: eval mode. */
int64_t asr_bn_blocks(int64_t rows);
int asr_bn_var_part(const float *x, const float *msum, float ms, float *part, int64_t rows, int C, void *stream);
int asr_bn_swish_fwd(const float *x, const float *msum, const float *vsum, float ms, float vs,
                     const float *gamma, const float *beta, float *y, int64_t rows, int C, float eps,
                     void *stream);
int asr_bn_swish_bwd_part(const float *dy, const float *x, const float *msum, const float *vsum, float ms,
                          float vs, const float *gamma, const float *beta, float *part, int64_t rows, int C,
                          float eps, void *stream);
int asr_bn_swish_bwd(const float *dy, const float *x, const float *msum, const float *vsum, float ms, float vs,
                     const float *gamma, const float *beta, const float *sums, float *dx, int64_t rows,
                     int C, float eps, void *stream);
/* RelativeMultiHeadAttention (:99-102,110-137): ac (B,H,T,T) becomes
 * softmax_j(keep[b,j] ? (ac + rel_shift(bd)) * scale : -1e4) in place, T <= 1024.  Backward: dattn
 * becomes d ac (= the shifted score gradient) in place and dbd its un-shifted copy.  All maps have
 * row stride ld >= T (a multiple of 4 lets the products that read them use 16-byte loads,
 * asr_gemm_desc.pad_ok); the kernels keep columns T..ld-1 at zero.
 * attn_drop / drop_mask != NULL: the dropout on the map (:133-134) in the same pass -- attn_drop gets the
 * dropped copy the value product reads, drop_mask its mask; the backward then takes the gradient w.r.t. the
 * dropped map in dattn and applies the mask itself. */
int asr_relattn_softmax_fwd(float *ac, const float *bd, const uint8_t *keep, float scale, int B, int H, int T,
                            int ld, float *attn_drop, uint8_t *drop_mask, float p, uint64_t seed, void *stream);
int asr_relattn_softmax_bwd(const float *attn, float *dattn, float *dbd, const uint8_t *keep, float scale,
                            int B, int H, int T, int ld, const uint8_t *drop_mask, float p, void *stream);
/* CNN_embedding (:386-407): patches of nn.Conv2d(Cin, Cout, 3, 2) on a channels-last map
 * x (B,Ti,Fi,Cin): col (B*To*Fo, ld) in (ki,kj,cin) order, To = (Ti-3)/2+1, Fo = (Fi-3)/2+1, columns
 * 9*Cin..ld-1 zero (ld = 12 gives the first convolution, Cin = 1, 16-byte rows);
 * asr_col2im_3x3s2 is the data gradient (a gather over rows of length 9*Cin, deterministic), times
 * the gradient of the ReLU that produced the map y when y != NULL. */
int asr_im2col_3x3s2(const float *x, float *col, int B, int Ti, int Fi, int Cin, int ld, void *stream);
int asr_col2im_3x3s2(const float *dcol, const float *y, float *dx, int B, int Ti, int Fi, int Cin,
                     void *stream);
/* asr_col2im_3x3s2 for the map y that the FIRST convolution produced (one input channel, the features get no
 * gradient): instead of writing relu'(y) * col2im(dcol), the kernel reduces it straight into the first
 * convolution's weight and bias gradients -- part[blk][c][0..8] = sum_pos d pre[pos][c] * taps[pos][k],
 * part[blk][c][9] = sum_pos d pre[pos][c] over the positions block blk visited; taps = the first convolution's
 * patches (B*Ti*Fi rows of ldt >= 9 floats, asr_im2col_3x3s2's output).  asr_col2im_dw_blocks rows of Cin*10
 * floats; asr_colsum over the rows gives [dW1 | db1].  Cin % 4 == 0 and (Cin / 4) must divide 256. */
int asr_col2im_dw_blocks(int B, int Ti, int Fi, int Cin);
int asr_col2im_3x3s2_dw(const float *dcol, const float *y, const float *taps, int ldt, float *part, int B, int Ti,
                        int Fi, int Cin, void *stream);

/* ------------------------------------------------- attention decoder loop --
 * Replaces the L-step teacher-forced loop Models/decoder.py:52-60 with
 * Models/attention.py:24-40 and Models/decoder.py:269-279 inside it, and the
 * autograd graph PyTorch would record for it.
 *
 * Dimensions: B batch, T frames, E encoder width (2*H_enc), Hd decoder width,
 * A = 2*Hd attention width, L label steps.  NC=10 conv channels, KW=100 taps.
 *
 * Hoisted by the caller (asr_sgemm): PH = L_he(hbatch) before the call;
 * U = tanh(L_gy g + b_gy + L_sy S[0..L-1]) and youtput = L_yy U after it
 * (neither is on the recurrent chain).
 */
typedef struct {
    int B, T, E, Hd, L, V;
    /* inputs */
    const float *hbatch;      /* (B,T,E) */
    const float *PH;          /* (B,T,A) = L_he(hbatch), hoisted (asr_sgemm) */
    const int32_t *lengths;   /* (B) */
    const int64_t *targets;   /* (B,L) */
    /* parameters, reference layouts */
    const float *W_se;        /* (A,Hd)   att.L_se.weight */
    const float *w_ee;        /* (A)      att.L_ee.weight */
    const float *W_fe;        /* (A,10)   att.L_fe.weight */
    const float *conv_w;      /* (10,100) att.F_conv1d.weight */
    const float *Emb;         /* (V,4Hd)  L_ys.weight */
    const float *W_ss;        /* (4Hd,Hd) */
    const float *W_gs;        /* (4Hd,E)  */
    const float *b_gs;        /* (4Hd)    */
    /* saved activations (outputs of fwd, inputs of bwd) */
    float *S;                 /* (L+1,B,Hd) decoder state, S[0]=0 */
    float *C;                 /* (L+1,B,Hd) cell state,   C[0]=0 */
    float *alpha;             /* (L,B,T)   alignments (0 at t >= lengths[b]) */
    float *sproj;             /* (L,B,A)   L_se(s) per step */
    float *g;                 /* (L,B,E)   contexts */
    float *gates;             /* (L,B,4Hd) activated cell gates */
    float *Fc;                /* (L,B,T,10) conv features of alpha_{s-1} */
    /* workspace */
    float *ws;                /* asr_attdec_ws_floats() floats */
} asr_attdec_fwd_args;

int64_t asr_attdec_ws_floats(int B, int T, int E, int Hd, int L);
int asr_attdec_fwd(const asr_attdec_fwd_args *a, void *stream);
/* asr_attdec_fwd runs the whole loop as ONE persistent launch when the (B x ceil(T/64)) workgroups each get a
 * compute unit and the shape fits its register-resident layout (2*Hd <= 640, E <= 640, Hd <= 384, the
 * cell tile jobs and the state-projection tile jobs each <= workgroups; csrc/attdec.hip), else four launches per
 * label step.
 * asr_attdec_persist(0 / 1) switches the one-launch form off / on (returns the previous setting; any other
 * value only queries); environment variable ASR_ATTDEC_PERSIST=1 / 0 sets it for the process (default: on).
 * asr_attdec_err_offset: index (in floats) inside a->ws of a 4-byte word that is non-zero after the call iff an
 * in-launch wait timed out (outputs invalid) -- fold it into the loss with asr_poison_if_flag. */
int asr_attdec_persist(int on);
/* Which form the two calls take for a shape on the current device when every tensor is 16-byte aligned: bit 0 = the
 * forward runs as one persistent launch, bit 1 = the backward does; 0 after asr_attdec_persist(0).  A caller whose
 * batch is too large for one launch (B * ceil(T/64) > compute units: T = 800 at B = 32, the CSJ shape) can split it
 * into parts that fit -- utterances are independent inside Models/decoder.py:52-60 -- as asr_amd/Models/decoder.py does. */
int asr_attdec_plan(int B, int T, int E, int Hd, int L);
/* diagnostics (tools/dec_stamps.py): device buffer for per-phase time stamps of the persistent launches, NULL = off */
void asr_attdec_debug_stamps(void *buf);
/* Pauses in front of the loop kernels' in-launch waits, in units of 256 cycles (a wave that starts polling long before
 * its data can be there only adds fabric traffic in front of the stores it waits for): which = 0..4 forward (state
 * projection, energies, context, state for the projection job, state for the cell job; the context pause is a start
 * value that every wave adapts to its own failed first attempts unless entry 5 is non-zero), 8.. backward.  Sets one and
 * returns the previous value; units < 0 only reads.  The defaults are tuned at the BASELINE shapes
 * (tools/dec_pace.py); results do not depend on them.  No counterpart in the reference. */
int asr_attdec_pace(int which, int units);
int64_t asr_attdec_err_offset(int B, int T, int E, int Hd, int L);

/* Greedy search, batch 1: the beam_width == 1 path of Models/decoder.py:63-130.
 * Runs a->L (= hp.max_decoder_seq_len) steps with the argmax of step s fed back
 * as the label of step s; tokens (1,L) int64 receives the argmax sequence and
 * must be the buffer a->targets points at.  The caller cuts at hp.eos_id
 * (the reference returns [] when eos never appears). */
int asr_attdec_greedy(const asr_attdec_fwd_args *a, const float *W_sy, const float *W_gy,
                      const float *b_gy, const float *W_yy, const float *b_yy, int64_t *tokens,
                      void *stream);
/* One decoder step in two halves, for callers that choose the fed-back token themselves: the
 * beam search of Models/decoder.py:79-123.  `a` is read as a single step (L ignored):
 *   asr_attdec_step_att : attention + context (Models/attention.py:24-40) from the state a->S
 *       ([B][Hd]) and the previous alignment alpha_prev ([B][T], NULL = zeros: first step) ->
 *       a->alpha [B][T], a->g [B][E]  (a->sproj, a->Fc, a->ws are scratch of one step's size)
 *   asr_attdec_step_cell: rec = Emb[targets[b]] + L_ss s + L_gs g, LSTM cell
 *       (Models/decoder.py:118-119,269-279): a->S, a->C in -> a->S + B*Hd, a->C + B*Hd out
 *       (targets: [B] int64, a->gates [B][4Hd] scratch). */
int asr_attdec_step_att(const asr_attdec_fwd_args *a, const float *alpha_prev, void *stream);
int asr_attdec_step_cell(const asr_attdec_fwd_args *a, void *stream);

typedef struct {
    asr_attdec_fwd_args f;    /* same tensors as the forward call */
    /* transposed parameter copies made by the caller (asr_transpose) */
    const float *W_seT;       /* (Hd,A)   */
    const float *W_ssT;       /* (Hd,4Hd) */
    const float *W_syT;       /* (Hd,Hd)  */
    const float *W_gsT;       /* (E,4Hd)  */
    const float *W_gyT;       /* (E,Hd)   */
    /* upstream */
    const float *dyp;         /* (L,B,Hd) d loss / d (L_gy g + L_sy s) = dU * (1 - U^2) */
    /* per-step gradients saved for the hoisted weight-gradient GEMMs */
    float *drec;              /* (L,B,4Hd) d loss / d rec_input */
    float *dsp;               /* (L,B,A)   d loss / d L_se(s)  (dsp[0] = 0: S[0] = 0) */
    float *dg;                /* (L,B,E)   d loss / d context */
    float *de;                /* (L,B,T)   d loss / d energy  (valid at t < lengths[b]) */
    float *dF;                /* (L,B,T,10) d loss / d conv features (valid at t < lengths[b]) */
    /* outputs */
    float *dPH;               /* (B,T,A)  */
    float *dw_ee;             /* (A)      */
    float *dW_fe;             /* (A,10)   */
    float *dconv_w;           /* (10,100) */
    float *ws;                /* asr_attdec_bwd_ws_floats() floats */
    /* != 0: asr_attdec_bwd leaves dw_ee, dW_fe and dconv_w (gradients of attention.py's own small weights; nothing in
     * the backward pass reads them) to asr_attdec_bwd_tail, which the caller may run on another stream */
    int defer_tail;
} asr_attdec_bwd_args;

int64_t asr_attdec_bwd_ws_floats(int B, int T, int E, int Hd, int L);
int asr_attdec_bwd(const asr_attdec_bwd_args *a, void *stream);
/* Like asr_attdec_fwd, asr_attdec_bwd runs its L-step loop as ONE persistent launch when the shape fits (same
 * switch: asr_attdec_persist / ASR_ATTDEC_PERSIST), else four launches per label step.
 * asr_attdec_bwd_err_offset: index (in floats) inside a->ws of a 4-byte word that is non-zero after the call iff an
 * in-launch wait timed out (gradients invalid) -- fold it into a gradient with asr_poison_if_flag. */
int64_t asr_attdec_bwd_err_offset(int B, int T, int E, int Hd, int L);
/* The weight-gradient tail of asr_attdec_bwd when a->defer_tail was set (autograd of Models/attention.py:22-28's
 * L_fe, L_ee, F_conv1d): column sums of the partials the main call left in a->ws -> dW_fe, dw_ee; dconv_w from the
 * saved dF and alpha.  Same args as the main call; must be ordered after it (same stream, or an event). */
int asr_attdec_bwd_tail(const asr_attdec_bwd_args *a, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ASR_AMD_H */
