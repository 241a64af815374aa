/* gdkvm.h -- C ABI of the MI355X-native GDKVM memory path (libgdkvm_hip.so, gfx950 only).
 *
 * The reference snapshot exposes NO plugin / operator / FFI interface for this path: /root/reference is the
 * project website and holds no model code (the code lives in the repo named at
 * /root/reference/README.md:1 and is excluded at /root/reference/.gitignore:73-76).  There is therefore no
 * reference interface file:line for these entry points to replace; each one is instead tied to the
 * operation the reference *names* (/root/reference/README.md:20, .../website/src/content/homepage/en.json:20)
 * and to the row of SURVEY.md §8(a) that specifies it.  INTEGRATION.md shows the binding a maintainer of the
 * real model code would add (ctypes stub + nn.Module seam).
 *
 * Conventions (all entry points)
 *   - every pointer is a DEVICE pointer owned by the caller (hipMalloc / torch CUDA tensor); nothing is
 *     allocated, retained or freed by the callee; 16-byte alignment is required for every buffer;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls are asynchronous with
 *     respect to the host and stream-ordered; no host synchronisation happens inside (graph-capturable);
 *   - return value: 0 on success, <0 on error; never throws, never exits:
 *       GDKVM_ERR_SHAPE (-1) bad / unsupported shape    GDKVM_ERR_DTYPE (-2) unsupported io_dtype
 *       GDKVM_ERR_ARCH  (-3) device is not gfx950       GDKVM_ERR_LAUNCH(-4) HIP launch failure
 *       GDKVM_ERR_WORKSPACE (-5) workspace too small    GDKVM_ERR_ARG (-6) null / misaligned pointer
 *     (a return code speaks about the ARGUMENTS of an asynchronous call, never about its data: see gdkvm_scan_status, GDKVM_ERR_RANGE)
 *     gdkvm_last_error() returns a thread-local message for the most recent failure on this thread;
 *   - re-entrant; concurrent calls are safe iff their output / workspace buffers are distinct.
 *
 * Tensor layouts are token-major with the channel innermost (what an NHWC 1x1 convolution emits):
 *   q, k   [B, T, N, Hh, Dk]     v, r   [B, T, N, Hh, Dv]     (io_dtype: f32 or bf16)
 *   alpha  [B, T, Hh]  fp32      beta   [B, T, N, Hh]  fp32
 *   state  [B, Hh, Dk, Dv] fp32  (always fp32: the recurrent memory is never stored narrow)
 */
#ifndef GDKVM_H
#define GDKVM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GDKVM_ABI_VERSION 1

enum { GDKVM_OK = 0, GDKVM_ERR_SHAPE = -1, GDKVM_ERR_DTYPE = -2, GDKVM_ERR_ARCH = -3,
       GDKVM_ERR_LAUNCH = -4, GDKVM_ERR_WORKSPACE = -5, GDKVM_ERR_ARG = -6,
       GDKVM_ERR_RANGE = -7 /* gdkvm_scan_status only: the data left the numeric range of the default operand format */ };

enum { GDKVM_F32 = 0, GDKVM_BF16 = 1 };                       /* io_dtype */

/* GDR write rule */
enum { GDKVM_RULE_GATED_LINEAR = 0,      /* S <- a S + sum_i b_i k_i v_i^T  (literal BASELINE.json formula) */
       GDKVM_RULE_DELTA_PARALLEL = 1,    /* all tokens of a frame erase against the decayed state          */
       GDKVM_RULE_DELTA_SEQUENTIAL = 2   /* token-sequential gated delta rule (default)                    */ };

/* prologue flags */
enum { GDKVM_FLAG_NORMALIZE_QK = 1,      /* q,k <- x * rsqrt(sum x^2 + 1e-12)   */
       GDKVM_FLAG_GATE_LOGITS = 2,       /* alpha, beta are logits -> sigmoid   */
       GDKVM_FLAG_TRAIN = 4,             /* prep also emits the operand layouts gdkvm_scan_bwd reads (set by
                                            gdkvm_scan_fwd itself whenever s_hist != NULL) */
       GDKVM_FLAG_WIDE_RANGE = 8         /* operands of the state recurrence as three bf16 terms (the whole fp32 range, twice
                                            the MFMAs) instead of the default fp16 pairs (22 bits).  The default serves
                                            values and carried states of any magnitude for rules GATED_LINEAR and
                                            DELTA_SEQUENTIAL PROVIDED every frame's map is a contraction -- keys of unit norm,
                                            alpha and beta in [0, 1]: guaranteed by the kernels themselves when both
                                            GDKVM_FLAG_NORMALIZE_QK and GDKVM_FLAG_GATE_LOGITS are set, and the caller's
                                            promise when it passes keys / gates it normalised itself (raw gates above 1 or
                                            un-normalised keys without the flags void the bound below: pass this flag then).
                                            The serial kernel carries the state at 2^-e and sizes e per call and 16-column
                                            slice from that bound, 8 (max|s_in| + sum_t max|G_t|) -- e = 4 (the
                                            format's default, hence bit-identical chunked calls) for every ordinary input,
                                            larger exactly when the state needs it.  One corner is refused loudly rather
                                            than served: frames of more than 64 tokens whose chunk composition leaves the
                                            pair's range (|values| around 1e5 times the usual) come back as NaNs -- pass
                                            this flag for such inputs.  gdkvm_scan_fwd sets it itself for rule
                                            DELTA_PARALLEL, the one rule without a bound (not contractive); a caller of the
                                            split prep / apply / transition entry points passes the same flags to each. */ };

int gdkvm_abi_version(void);
const char* gdkvm_last_error(void);

/* Bytes of scratch gdkvm_scan_fwd needs for this shape (per-frame WY factors). */
size_t gdkvm_scan_workspace_bytes(int B, int T, int Hh, int N, int Dk, int Dv);

/* Rows a1+a2+a3+a5: fused LKVA read + GDR write over T frames with state carry.
 *   for t in 0..T-1:  R_t = Q_t S_{t-1};   S_t = GDR(S_{t-1}, K_t, V_t, alpha_t, beta_t)
 * s_in may be NULL (zero state); s_out may be NULL.  T == 1 is the per-frame `step` entry point.  A clip
 * processed as consecutive calls with the state carried is bit-identical to one call.
 * Names: LKVA / GDR at /root/reference/README.md:20; "state transition matrix" at
 * /root/reference/website/src/content/homepage/en.json:20.
 * Supported: Dk == 64 (the measured kernels); 8 <= Dk < 64 in multiples of 8 (gdkvm_scan_fwd only, not with s_hist: the call zero-extends
 * q, k and the state to 64 channels inside the workspace, which is exact); 64 < Dk <= 256 in multiples of 8 (gdkvm_scan_fwd only,
 * inference: the definitional recurrence in fp32 on one workgroup per 16-column slice of the state, csrc/gdr_general.hip -- same
 * results, same chunk bit-identity, no workspace, far slower than the Dk = 64 path; gdkvm_scan_status has nothing to report for it);
 * Dv % 16 == 0; 0 <= N <= 4096 (a 1024x1024 frame at stride 16).
 * s_hist (training): if non-NULL, [B,T,Hh,Dk,Dv] fp32 receives the state BEFORE every frame; gdkvm_scan_bwd needs
 * it together with the untouched workspace of this call. */
int gdkvm_scan_fwd(const void* q, const void* k, const void* v, const float* alpha, const float* beta,
                   const float* s_in, void* r_out, float* s_out, float* s_hist, void* workspace, size_t workspace_bytes,
                   int B, int T, int Hh, int N, int Dk, int Dv,
                   int io_dtype, int rule, int flags, void* stream);

/* Did the last gdkvm_scan_fwd / gdkvm_scan_fwd_normed / gdkvm_scan_apply on `workspace` (same shape arguments and flags) stay inside the
 * range of the default fp16-pair operands?  Those calls are asynchronous and return before a kernel has seen the data; where the bound
 * on the state is not finite -- or a chunk composition of a frame of more than 64 tokens overflowed on the way -- the results are NaNs
 * (never saturated numbers), and this call says so: it copies the per-slice exponents the serial kernel left in the workspace back on
 * `stream`, WAITS for the stream (the one synchronising entry point of the library), and returns GDKVM_ERR_RANGE or GDKVM_OK.
 * GDKVM_FLAG_WIDE_RANGE serves such inputs.  Optional: nothing else in the library depends on it being called. */
int gdkvm_scan_status(const void* workspace, size_t workspace_bytes, int B, int T, int Hh, int N, int Dk, int Dv, int flags, void* stream);

/* The two stages of gdkvm_scan_fwd as separate entry points (gdkvm_scan_fwd == prep then apply on one stream).
 * `prep` is state-independent and parallel over frames: the a5 prologue incl. the query norms, and every frame folded
 * into the affine map S' = alpha (P S) + G of SURVEY.md A.3 (P = I - Kn^T Wt, G = Kn^T Ut) in the workspace; with
 * GDKVM_FLAG_TRAIN also the WY factors the backward needs.  `apply` is the serial-in-time read/write recurrence.
 * Exposed so a caller can run the prep of chunk c+1 on a second stream while chunk c is being applied, and so each
 * kernel can be timed on its own. */
int gdkvm_scan_prep(const void* q, const void* k, const void* v, const float* beta, void* workspace, size_t workspace_bytes,
                    int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype, int rule, int flags, void* stream);
int gdkvm_scan_apply(const void* q, const float* alpha, const float* s_in, void* r_out, float* s_out,
                     float* s_hist, const void* workspace, size_t workspace_bytes,
                     int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype, int flags, void* stream);

/* the same two calls for a caller that already holds the inverse L2 norms of the key and query rows --
 * gdkvm_proj_gates, which produces q, k, v, the gate logits AND these norms in one launch.  norms [B*T*N, Hh, 2] fp32:
 * 1 / sqrt(sum k^2 + 1e-12) and the same for q, of the rows AS STORED (io_dtype).  The frame-parallel kernel then neither reads q
 * nor reduces anything in its first phase.  Requires GDKVM_FLAG_NORMALIZE_QK, Dk == 64, no s_hist (inference); everything else as
 * gdkvm_scan_prep / gdkvm_scan_fwd, and the results agree with them to the last few ulp of the norms' summation order. */
int gdkvm_scan_prep_normed(const void* q, const void* k, const void* v, const float* beta, const float* norms,
                           void* workspace, size_t workspace_bytes,
                           int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype, int rule, int flags, void* stream);
int gdkvm_scan_fwd_normed(const void* q, const void* k, const void* v, const float* alpha, const float* beta, const float* norms,
                          const float* s_in, void* r_out, float* s_out, void* workspace, size_t workspace_bytes,
                          int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype, int rule, int flags, void* stream);

/* SURVEY.md §8(f) row n3.  The effect of a block of frames on the state is affine, S_out = Phi S_in + S_loc, with the
 * state-transition matrix Phi = prod_t alpha_t (I - Kn_t^T Wt_t) ("Linear Key-Value Association defines frame-to-frame
 * causal relations as the state transition matrix", /root/reference/website/src/content/homepage/en.json:20).
 * gdkvm_scan_transition returns Phi [B,Hh,Dk,Dk] (fp32) of the T frames a preceding gdkvm_scan_prep call prepared in
 * `workspace` (the same recurrence kernel, started from S = I with a zero write term); S_loc is gdkvm_scan_apply with
 * s_in = NULL (r_out may be NULL when only states are wanted).  Segments of a long clip -- or of a clip sharded over GPUs
 * in time -- can then be scanned independently and stitched with one small exchange. */
int gdkvm_scan_transition(const void* q, const float* alpha, float* phi_out, const void* workspace, size_t workspace_bytes,
                          int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype, int flags, void* stream);
/* Row n3, the sequential stitch of a time-segmented scan: phi [B,S,Hh,Dk,Dk] and s_loc [B,S,Hh,Dk,Dv] are the transition matrix
 * and the zero-start end state of each of S consecutive segments (gdkvm_scan_transition / gdkvm_scan_apply on the clip viewed as
 * B*S clips of T/S frames); starts [B,S,Hh,Dk,Dv] receives the state every segment really starts from,
 *   starts[:,0] = s_in (zero if NULL),  starts[:,c+1] = phi[:,c] starts[:,c] + s_loc[:,c],
 * and s_end (optional) the state after the last segment.  All fp32; exact fp32 arithmetic. */
int gdkvm_scan_stitch(const float* phi, const float* s_loc, const float* s_in, float* starts, float* s_end,
                      int B, int S, int Hh, int Dk, int Dv, void* stream);

/* Row n3 as one call: gdkvm_scan_fwd with the time axis cut into `segments` equal pieces that run concurrently (prep of the clip
 * viewed as B*segments clips, gdkvm_scan_transition + a read-out-free gdkvm_scan_apply per segment, gdkvm_scan_stitch, then
 * gdkvm_scan_apply from the true start states).  For long clips on few clips / heads / columns, where the serial recurrence leaves
 * CUs idle: 2x512 frames, Dv = 256 (32 serial workgroups) 269 us with 16 segments against 346 us (round 3).  segments must divide T;
 * segments == 0 chooses by shape (gdkvm_scan_segments returns the choice: a power of two >= 4, or 1 = plain gdkvm_scan_fwd).
 * Equal to gdkvm_scan_fwd up to fp32 re-association through Phi -- NOT bit-identical, which is why gdkvm_scan_fwd, whose contract
 * is bit-identity under chunked calls, never takes this path by itself.  No s_hist (inference).  workspace:
 * gdkvm_scan_segmented_workspace_bytes with the same `segments` argument. */
int gdkvm_scan_segments(int B, int T, int Hh, int Dv, int segments);
size_t gdkvm_scan_segmented_workspace_bytes(int B, int T, int Hh, int N, int Dk, int Dv, int segments);
int gdkvm_scan_fwd_segmented(const void* q, const void* k, const void* v, const float* alpha, const float* beta,
                             const float* s_in, void* r_out, float* s_out, void* workspace, size_t workspace_bytes,
                             int B, int T, int Hh, int N, int Dk, int Dv, int segments,
                             int io_dtype, int rule, int flags, void* stream);

/* SURVEY.md A.1, the `normalizer` flag of the LKVA read: besides S the memory carries z [B, Hh, Dk] (fp32) -- "the same recurrence on
 * v == 1", i.e. the state column of a value channel that is 1 for every token -- and the read-out of token n of frame t is
 *     R_t[n, :] = (Qn_t[n] S_{t-1}) / (|Qn_t[n] . z_{t-1}| + eps).
 * Everything else as gdkvm_scan_fwd (Dk == 64, inference only: no s_hist): z_in / s_in may be NULL (zeros), z_out / s_out may be NULL;
 * a clip processed as consecutive calls with (S, z) carried is bit-identical to one call.  Implemented as the library's own scan on
 * Dv + 16 value channels (channel Dv holds the ones) with the division on the way out: csrc/gdr_normalizer.hip.  0 < eps < 1
 * (SPEC-v0 default 1e-6).  Names: "Linear Key-Value Association" at /root/reference/README.md:20. */
size_t gdkvm_scan_normalizer_workspace_bytes(int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype);
int gdkvm_scan_fwd_normalizer(const void* q, const void* k, const void* v, const float* alpha, const float* beta,
                              const float* s_in, const float* z_in, void* r_out, float* s_out, float* z_out,
                              void* workspace, size_t workspace_bytes,
                              int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype, int rule, int flags, float eps, void* stream);

/* The per-frame `step` mode with mask feedback: when the value written for frame t depends on the mask
 * predicted for frame t, the frame loop is read -> KPFF -> decoder -> mask -> write, one frame of every clip at a time.
 *   gdkvm_lkva_read       row a1 on its own: r_out [B, N, Hh, Dv] = Qn S for ONE frame per clip, q [B, N, Hh, Dk] (io_dtype), the state
 *                         s [B, Hh, Dk, Dv] fp32 given as a tensor.  flags: GDKVM_FLAG_NORMALIZE_QK as in gdkvm_scan_fwd; norms
 *                         (may be NULL) = gdkvm_proj_gates' [B*N, Hh, 2] inverse norms (entry 1 is the query's).  Plain fp32 fmaf
 *                         chains in key-channel order: deterministic.  Dk == 64.
 *   gdkvm_mask_embed_add  v [BT, h*w, C] (io_dtype) += w_embed[c] * m, m = the mean over the token's adaptive-average-pool cell
 *                         (PyTorch's adaptive_avg_pool2d cells) of the foreground indicator of mask [BT, H, W] (uint8: class != 0;
 *                         255 = unlabelled counts as background): the first-frame-mask embedding of the module applied to a predicted
 *                         mask.  The write itself is gdkvm_scan_fwd with T == 1. */
int gdkvm_lkva_read(const void* q, const float* norms, const float* s, void* r_out,
                    int B, int N, int Hh, int Dk, int Dv, int io_dtype, int flags, void* stream);
int gdkvm_mask_embed_add(const uint8_t* mask, const float* w_embed, void* v, int BT, int H, int W, int h, int w, int C,
                         int io_dtype, void* stream);

/* Row a7: backward of gdkvm_scan_fwd.  Inputs: the forward's inputs, its s_hist, its workspace exactly as the
 * forward left it, the gradients d_r [B,T,N,Hh,Dv] (io_dtype) and d_s_out [B,Hh,Dk,Dv] (fp32, may be NULL = 0).
 * Outputs: d_q, d_k [B,T,N,Hh,Dk], d_v [B,T,N,Hh,Dv] (io_dtype), d_alpha [B,T,Hh], d_beta [B,T,N,Hh] (fp32; with
 * GDKVM_FLAG_GATE_LOGITS they are gradients w.r.t. the logits), d_s_in [B,Hh,Dk,Dv] (fp32, may be NULL).
 * bwd_workspace (gdkvm_scan_bwd_workspace_bytes) holds the per-frame state gradients.  Supported: N <= 64 (any N: gdkvm_scan_train_fwd /
 * gdkvm_scan_train_bwd below). */
size_t gdkvm_scan_bwd_workspace_bytes(int B, int T, int Hh, int N, int Dk, int Dv);
int gdkvm_scan_bwd(const void* q, const void* k, const void* v, const float* alpha, const float* beta,
                   const float* s_hist, const void* fwd_workspace, size_t fwd_workspace_bytes,
                   const void* d_r, const float* d_s_out,
                   void* d_q, void* d_k, void* d_v, float* d_alpha, float* d_beta, float* d_s_in,
                   void* bwd_workspace, size_t bwd_workspace_bytes,
                   int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype, int rule, int flags, void* stream);

/* Row a7 for frames of ANY token count in two calls.  gdkvm_scan_train_fwd is gdkvm_scan_fwd that keeps what the backward needs
 * in ONE opaque workspace (gdkvm_scan_train_workspace_bytes: state history, WY factors, and for N > 64 the frame re-cut into
 * ceil(N/64) pseudo-frames of 64 tokens -- the first with the frame's gate, the others with gate 1, padding tokens with beta = 0 --
 * on which the state recurrence runs; the read-out of all N tokens uses the state before the frame).  gdkvm_scan_train_bwd takes
 * the same inputs, that workspace untouched, d_r and d_s_out (may be NULL = 0), and returns the gradients of gdkvm_scan_bwd.
 * N <= 64 is exactly gdkvm_scan_fwd (s_hist inside the workspace) + gdkvm_scan_bwd; N > 64 composes gdkvm_scan_fwd without a
 * read-out, gdkvm_readout_fwd / _bwd and gdkvm_scan_state_bwd below.  GDKVM_RULE_DELTA_PARALLEL: N <= 64 only (its chunks
 * combine additively, not in sequence).  Supported: Dk == 64, Dv % 16 == 0, 1 <= N <= 4096. */
size_t gdkvm_scan_train_workspace_bytes(int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype);
int gdkvm_scan_train_fwd(const void* q, const void* k, const void* v, const float* alpha, const float* beta, const float* s_in,
                         void* r_out, float* s_out, void* train_workspace, size_t train_workspace_bytes,
                         int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype, int rule, int flags, void* stream);
int gdkvm_scan_train_bwd(const void* q, const void* k, const void* v, const float* alpha, const float* beta,
                         const void* d_r, const float* d_s_out,
                         void* d_q, void* d_k, void* d_v, float* d_alpha, float* d_beta, float* d_s_in,
                         void* train_workspace, size_t train_workspace_bytes,
                         int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype, int rule, int flags, void* stream);

/* Row a7 for frames of more than 64 tokens (and any caller that does its own read-out): the backward of the STATE recurrence
 * alone.  No q / d_r / d_q; instead d_hist [B,T,Hh,Dk,Dv] fp32 (may be NULL) is the gradient with respect to the state before
 * every frame (what a read-out R_t = f(S_{t-1}) done outside sends back), which enters the reverse recurrence as its additive
 * term.  gdkvm_amd/ops.py::scan uses it with every 64-token chunk of a frame as a pseudo-frame (gate 1 after the first, padding
 * tokens with beta = 0) and the read-out as a batched matmul on the saved states.  Same workspaces as gdkvm_scan_bwd. */
int gdkvm_scan_state_bwd(const void* k, const void* v, const float* alpha, const float* beta,
                         const float* s_hist, const void* fwd_workspace, size_t fwd_workspace_bytes,
                         const float* d_hist, const float* d_s_out,
                         void* d_k, void* d_v, float* d_alpha, float* d_beta, float* d_s_in,
                         void* bwd_workspace, size_t bwd_workspace_bytes,
                         int B, int T, int Hh, int N, int Dk, int Dv, int io_dtype, int rule, int flags, void* stream);

/* Rows a1 / a7 for training on frames of more than 64 tokens: the LKVA read-out R_t = diag(qinv) Q_t S_{t-1} of all N tokens of
 * a frame from a saved state history, and its backward.  s_hist [B, T*hist_stride, Hh, Dk, Dv] fp32 is the history
 * gdkvm_scan_fwd saves over the 64-token pseudo-frames of gdkvm_amd/ops.py::_scan_chunked (hist_stride of them per frame): frame
 * t reads the state at index t*hist_stride.  q, r_out, d_r, d_q [B,T,N,Hh,*] in io_dtype.  gdkvm_readout_bwd writes
 * d_q and the state gradients dS = Qn^T dR into d_hist at the same indices t*hist_stride (the caller zero-fills the rest and
 * hands d_hist to gdkvm_scan_state_bwd).  GDKVM_FLAG_NORMALIZE_QK as in gdkvm_scan_fwd.  Any N >= 0; Dk == 64; exact fp32. */
int gdkvm_readout_fwd(const void* q, const float* s_hist, void* r_out, int B, int T, int Hh, int N, int Dk, int Dv,
                      int hist_stride, int io_dtype, int flags, void* stream);
int gdkvm_readout_bwd(const void* q, const float* s_hist, const void* d_r, void* d_q, float* d_hist,
                      int B, int T, int Hh, int N, int Dk, int Dv, int hist_stride, int io_dtype, int flags, void* stream);

/* Row a7: the plain products of the training backward (KPFF's dX / dW products, the 1x1 projections and their gradients),
 * hand-written like the rest of the path.  Row-major operands in io_dtype, fp32 accumulation.
 *   gdkvm_gemm_nt:  C[M,N] = A[M,K] B[N,K]^T (+ bias[N], fp32, may be NULL), C in io_dtype.  K % 32 == 0 (bf16) / % 16 (fp32).
 *   gdkvm_gemm_tn:  C[K1,N] (fp32) = A[M,K1]^T B[M,N]: the reduction over the M rows (the tokens of a batch) is split over
 *                   workgroups into fp32 partial tiles in `workspace` (gdkvm_gemm_tn_workspace_bytes) and summed in a fixed
 *                   order (deterministic).  K1, N multiples of 8 (bf16) / 4 (fp32); any M >= 0. */
int gdkvm_gemm_nt(const void* a, const void* b, const float* bias, void* c, int M, int N, int K, int io_dtype, void* stream);
size_t gdkvm_gemm_tn_workspace_bytes(int M, int K1, int N);
int gdkvm_gemm_tn(const void* a, const void* b, float* c, void* workspace, size_t workspace_bytes,
                  int M, int K1, int N, int io_dtype, void* stream);
/* The same product with the column sums of A as a second result: colsum[K1] (fp32) = sum over the M rows of A[m][k1] -- the bias
 * gradient of y = x W^T + b when A = dY -- from one more MFMA per step against a fragment of ones, summed over the row splits in the same
 * fixed order (workspace: gdkvm_gemm_tn_colsum_workspace_bytes). */
size_t gdkvm_gemm_tn_colsum_workspace_bytes(int M, int K1, int N);
int gdkvm_gemm_tn_colsum(const void* a, const void* b, float* c, float* colsum, void* workspace, size_t workspace_bytes,
                         int M, int K1, int N, int io_dtype, void* stream);

/* Row a4: Key-Pixel Feature Fusion ("fuses the local key feature, the global key feature with the pixel
 * feature", /root/reference/website/src/content/homepage/en.json:20; "multiple scales", README.md:20).
 *   local [BT,N,Ck]  global [BT,N,Cv]  pixel [BT,N,Cp]  out [BT,N,Cp]   (io_dtype), N = h*w
 *   wa [2Cp, Cp+Ck+Cv]  ba [2Cp]  wl [Cp,Ck]  wg [Cp,Cv]                 (fp32)
 *   Gms = mean_{s in 1,2,4} cellmean_s(global);  g = sigmoid([P;L;Gms] wa^T + ba) = (g_l | g_g)
 *   out = P + g_l * (L wl^T) + g_g * (Gms wg^T)
 * Arithmetic: io_dtype f32 with channel counts that are multiples of 32 and a workspace -> operands as two bf16 terms
 * (16 significant bits, the fp32 exponent range), three bf16 MFMAs per product, fp32 accumulation / pooling / residual /
 * epilogue: about 2^-16 relative per product (3e-5 absolute against the fp64 oracle at unit-variance features).  io_dtype f32
 * otherwise -- other channel counts, or workspace == NULL, which is how a caller asks for it -> exact fp32 MFMA (3x slower).
 * io_dtype bf16 with channel counts that are multiples of 32 -> bf16 MFMA with fp32 accumulation (weights and the pooled
 * feature are rounded to bf16, i.e. bf16-autocast accuracy); other bf16 shapes use the exact fp32 arm.  `workspace`
 * (gdkvm_kpff_workspace_bytes) holds the bf16 copies of the weights (high and low terms for f32), rebuilt on every call.
 * Supported: Ck, Cv, Cp multiples of 16; h*w <= 4096 (grids wider than 16 columns are processed as 4-row x 16-column tiles). */
size_t gdkvm_kpff_workspace_bytes(int Ck, int Cv, int Cp, int io_dtype);
int gdkvm_kpff_fwd(const void* local, const void* global, const void* pixel,
                   const float* wa, const float* ba, const float* wl, const float* wg, void* out,
                   void* workspace, size_t workspace_bytes,
                   int BT, int Ck, int Cv, int Cp, int h, int w, int io_dtype, void* stream);

/* gdkvm_kpff_fwd with the weight re-pack skipped: `packed_workspace` must be the workspace a previous gdkvm_kpff_fwd
 * call filled from the SAME (unchanged) wa / wl / wg -- the inference case. */
int gdkvm_kpff_fwd_packed(const void* local, const void* global, const void* pixel,
                          const float* wa, const float* ba, const float* wl, const float* wg, void* out,
                          void* packed_workspace, size_t workspace_bytes,
                          int BT, int Ck, int Cv, int Cp, int h, int w, int io_dtype, void* stream);

/* Row a7 for KPFF.  The training forward also stores what the backward needs (all [M = BT*N, .] in io_dtype): the
 * gates after the sigmoid [M,2Cp], L wl^T [M,Cp], Gms wg^T [M,Cp] and the pooled feature Gms [M,Cv].  The backward is
 *   gdkvm_kpff_bwd_pre   d_z = (d_out*Lp*g_l(1-g_l) | d_out*Gp*g_g(1-g_g)), d_lp = d_out*g_l, d_gp = d_out*g_g
 *   six plain GEMMs      d_x = d_z wa, d_l_add = d_lp wl, d_g_add = d_gp wg, d_wa = d_z^T [P;L;Gms], d_wl = d_lp^T L,
 *                        d_wg = d_gp^T Gms  (and d_ba = column sums of d_z) -- on the caller's side: gdkvm_gemm_nt for the
 *                        products against weights, gdkvm_gemm_tn (split over the token rows, deterministic) for the weight gradients
 *   gdkvm_kpff_bwd_post  d_pixel = d_out + d_x[:, :Cp], d_local = d_x[:, Cp:Cp+Ck] + d_l_add,
 *                        d_global = pool(d_x[:, Cp+Ck:] + d_g_add)   (the multi-scale pooling operator is symmetric) */
int gdkvm_kpff_fwd_train(const void* local, const void* global, const void* pixel,
                         const float* wa, const float* ba, const float* wl, const float* wg, void* out,
                         void* save_gates, void* save_lp, void* save_gp, void* save_gms,
                         void* workspace, size_t workspace_bytes,
                         int BT, int Ck, int Cv, int Cp, int h, int w, int io_dtype, void* stream);
int gdkvm_kpff_bwd_pre(const void* d_out, const void* gates, const void* lp, const void* gp,
                       void* d_z, void* d_lp, void* d_gp, int BT, int N, int Cp, int io_dtype, void* stream);
int gdkvm_kpff_bwd_post(const void* d_out, const void* d_x, const void* d_l_add, const void* d_g_add,
                        void* d_pixel, void* d_local, void* d_global,
                        int BT, int Ck, int Cv, int Cp, int h, int w, int io_dtype, void* stream);

/* Row a6: mask = argmax_c logits (ties -> lowest class index); optional integer Dice counts.
 *   logits [BT, ncls, H, W] (io_dtype)   target [BT,H,W] u8 or NULL   mask [BT,H,W] u8
 *   counts [BT, ncls, 3] int32 = { |mask==c & target==c|, |mask==c|, |target==c| }  (zeroed by the callee;
 *   required iff target != NULL). */
int gdkvm_argmax_dice(const void* logits, const uint8_t* target, uint8_t* mask, int32_t* counts,
                      int BT, int ncls, int H, int W, int io_dtype, void* stream);

/* Row a6, fused with the decoder's last step: logits [BT, ncls, hl, wl] at low resolution are upsampled bilinearly
 * (align_corners = false, PyTorch's formula, fp32 without fused multiply-add) to H x W inside the kernel and reduced to
 * the argmax mask / Dice counts directly; the full-resolution logits are never materialised. */
int gdkvm_upsample_argmax_dice(const void* logits, const uint8_t* target, uint8_t* mask, int32_t* counts,
                               int BT, int ncls, int hl, int wl, int H, int W, int io_dtype, void* stream);

/* ... and with the decoder's 1x1 head folded in as well: x [BT, hl, wl, C] is the stride-4 feature (NHWC, io_dtype), w [ncls, C] and
 * b [ncls] (fp32) the head; every block computes the class planes of the low-resolution rows it needs into LDS, with
 * gdkvm_head_logits' arithmetic and rounding -- masks and counts are bit-identical to gdkvm_head_logits followed by
 * gdkvm_upsample_argmax_dice, and the class planes never reach memory.  C/8 (bf16) or C/4 (f32) a power of two <= 64, ncls <= it. */
int gdkvm_head_upsample_argmax_dice(const void* x, const float* w, const float* b, const uint8_t* target, uint8_t* mask,
                                    int32_t* counts, int BT, int C, int ncls, int hl, int wl, int H, int W, int io_dtype, void* stream);

/* SURVEY.md §8(f) row n1 (inference build only): fused pointwise epilogue for the unchanged MIOpen convolutions,
 * y[r, c] = act(x[r, c] + bias[c] (+ residual[r, c])) over an NHWC tensor viewed as [rows, C]; relu != 0 applies ReLU.
 * x, residual, y in io_dtype (y may alias x), bias fp32.  C must be a multiple of 4 (f32) / 8 (bf16). */
int gdkvm_bias_act(const void* x, const float* bias, const void* residual, void* y,
                   size_t rows, int C, int relu, int io_dtype, void* stream);

/* Row n1, the decoder's head: 1x1 convolution + bias from an NHWC feature x [N, H, W, C] to NCHW class planes out [N, classes, H, W]
 * (io_dtype; weight fp32 [classes, C], bias fp32 [classes]; classes <= C/8 (bf16) / C/4 (f32)) -- the layout and dtype
 * gdkvm_upsample_argmax_dice reads; one pass instead of convolution + bias add + layout copy. */
int gdkvm_head_logits(const void* x, const float* w, const float* b, void* out, int N, int H, int W, int C, int classes,
                      int io_dtype, void* stream);
/* Training: the head's backward in one pass.  dz [N, classes, H, W] (io dtype, NCHW planes as gdkvm_seg_loss_bwd writes them), x the feature
 * gdkvm_head_logits read, w fp32 [classes, C] -> dx [N, H, W, C] (io dtype), dw fp32 [classes, C], db fp32 [classes]; per-workgroup partial
 * sums in `workspace` (gdkvm_head_bwd_workspace_bytes) added in index order: deterministic.  classes <= min(C/8 (C/4 for fp32), 8). */
size_t gdkvm_head_bwd_workspace_bytes(int C, int ncls);
int gdkvm_head_bwd(const void* x, const void* dz, const float* w, void* dx, float* dw, float* db, void* workspace, size_t workspace_bytes,
                   int N, int H, int W, int C, int ncls, int io_dtype, void* stream);

/* Row n4, the key / query / value projections in one pass over the token rows x [rows, K] (bf16):
 *   [out0 | out1 | out2][row, :] = x[row, :] W^T + bias,  W [w0 + w1 + w2, K] -- three contiguous outputs [rows, w0], [rows, w1],
 *   [rows, w2] (w1, w2 may be 0), widths multiples of 16, K a multiple of 32 up to 512, fp32 bias [w0 + w1 + w2].
 * wpack is W in MFMA fragment order, bf16: element ((ot * K/32 + ks) * 64 + lane) * 8 + j = W[16 ot + (lane & 15)][32 ks + 8 (lane >> 4) + j]
 * (gdkvm_amd/ops.py::pack_rows_weight builds it).  fp32 accumulation, one rounding. */
int gdkvm_proj_rows(const void* x, const void* wpack, const float* bias, void* out0, void* out1, void* out2,
                    long long rows, int K, int w0, int w1, int w2, int io_dtype, void* stream);

/* Row n4, the gates of the memory path in one pass over the stride-16 pixel feature p [frames, N, Cp] (io_dtype):
 *   beta_logit [frames, N, Hh]  = <p[f, n, :], w_gate[h, :]> + b_gate[h]           (per token; what the gate 1x1 projection computes)
 *   alpha_logit [frames, Hh]    = <mean_n p[f, n, :], w_decay[h, :]> + b_decay[h]  (per frame; token mean -> decay projection)
 * both fp32 (the dtype gdkvm_scan_prep / gdkvm_scan_apply read), weights fp32 [Hh, Cp], fp32 accumulation throughout. */
int gdkvm_gate_logits(const void* p, const float* w_gate, const float* b_gate, const float* w_decay, const float* b_decay,
                      float* beta, float* alpha, int frames, int N, int Cp, int Hh, int io_dtype, void* stream);

/* Row n4: gdkvm_proj_rows + gdkvm_gate_logits + the inverse norms of the key / query rows in ONE launch over the stride-16 pixel
 * feature x [frames * N, K] (bf16): out_k, out_q [rows, Hh*Dk], out_v [rows, Hh*Dv] (bf16; wpack / bias as for gdkvm_proj_rows with
 * the key, query and value weights stacked in that order), beta [rows, Hh] and alpha [frames, Hh] (fp32 logits, fp32 gate weights
 * [Hh, K]), norms [rows, Hh, 2] (fp32: what gdkvm_scan_fwd_normed takes).  The tokens are read from HBM once; deterministic (no
 * atomics).  K a power of two times 8 up to 512. */
int gdkvm_proj_gates(const void* x, const void* wpack, const float* bias, void* out_k, void* out_q, void* out_v,
                     const float* w_gate, const float* b_gate, const float* w_decay, const float* b_decay,
                     float* beta, float* alpha, float* norms,
                     int frames, int N, int K, int Hh, int Dk, int Dv, int io_dtype, void* stream);

/* Row n1 (inference build): 3x3 / stride 1 / pad 1 convolution with the folded-BatchNorm bias, the residual add and the ReLU
 * in its epilogue,  y = act(conv(x, w) + bias[k] (+ residual)),  x [N, H, W, C] (NHWC), w [K, 3, 3, C] (channels_last weights),
 * y / residual [N, H, W, K], bias fp32 [K]; bf16 only.  The fp32 accumulator is rounded ONCE (after the epilogue).  Both kernels
 * are hand-written: kernel 4 = 64 -> 64 channels (conv3x3_c64.hip: LDS halo band, weights resident in registers), kernel 5 = C a
 * multiple of 64, K of 16, rows of <= 64 pixels (conv3x3_tile.hip: 64-channel LDS chunks, weights streamed); kernel 0 picks by
 * shape; 6, 7, 8 pin kernel 5's wave grid (4 channel groups x 1 tile, 4 x 2, 2 x 2: tuning; 5 picks by K), 10 / 11 = the 4 x 2 / 2 x 2
 * grids on four waves with half-size pixel tiles, two workgroups per CU (what 0 / 5 pick: 10 for K a multiple of 128, else 11; same
 * sums, same bits).
 * kernel 9 | GDKVM_CONV_PACKED_WEIGHTS = the general implicit-GEMM kernel (conv_igemm.hip): any R x S, stride and padding, rows of
 * any width -- the strided 3x3 layers, the 1x1 downsamples, 3x3 / 1 / 1 on maps wider than 64 pixels; C a multiple of 32, K of 128,
 * w the gdkvm_conv_igemm_pack_weights copy of the [K, R, S, C] weights.  Shapes none of the three serves (odd channel counts)
 * return GDKVM_ERR_SHAPE: those stay on the framework convolution followed by gdkvm_bias_act.
 * kernel | GDKVM_CONV_PACKED_WEIGHTS: w is the fragment-ordered copy gdkvm_conv3x3_pack_weights made of the
 * [K, 3, 3, C] weights (same size) -- each 1 KiB weight fragment is then one contiguous read; same result bit for bit. */
enum { GDKVM_CONV_PACKED_WEIGHTS = 32 };
int gdkvm_conv3x3_pack_weights(const void* w, void* packed, int K, int C, int io_dtype, void* stream);
/* kernel 9's pack: w [K, R, S, C] bf16 -> the same bytes in MFMA-fragment order (K a multiple of 16, C of 32). */
int gdkvm_conv_igemm_pack_weights(const void* w, void* packed, int K, int C, int R, int S, int io_dtype, void* stream);
/* The pack of the same layer's DATA-GRADIENT convolution, from the forward weights w [K, 3, 3, C]:  dx = conv3x3(dy, w'),
 * w'[c][ty][tx][k] = w[k][2 - ty][2 - tx][c] -- K input channels (a multiple of 64), C output channels (of 16).  Use it as
 * gdkvm_conv_bias_act(dy, packed, zero bias, NULL, dx, N, K, H, W, C, 3, 3, 1, 1, 0, kernel | GDKVM_CONV_PACKED_WEIGHTS, ...). */
int gdkvm_conv3x3_pack_weights_dgrad(const void* w, void* packed, int K, int C, int io_dtype, void* stream);
/* Training: both packs of up to GDKVM_PACK_MAX_LAYERS 3x3 layers in ONE launch, straight from the fp32 master weights (the per-layer
 * sequence "cast to bf16, gdkvm_conv3x3_pack_weights, gdkvm_conv3x3_pack_weights_dgrad" is three launches of ~5 us per layer and step).
 * w[i]: fp32 [K_i, C_i, 3, 3] with element strides strides[4i .. 4i+3] = (k, c, r, s) -- any memory format; packed_fwd[i] /
 * packed_dgrad[i]: 18 K_i C_i bytes each, the copies the two calls above produce from the bf16-rounded weights, bit for bit.
 * K_i, C_i multiples of 64.  The five arrays are HOST arrays. */
#define GDKVM_PACK_MAX_LAYERS 24
int gdkvm_conv3x3_pack_weights_train(int nlayers, const void* const* w, void* const* packed_fwd, void* const* packed_dgrad,
                                     const int* K, const int* C, const long long* strides, void* stream);
/* Weight gradient of the same layers (training):  dw [K, C, 3, 3] fp32 = sum over pixels dy[n,y,x,k] * x[n,y+ty-1,x+tx-1,c]  for
 * NHWC bf16 x [N,H,W,C] and dy [N,H,W,K]; C and K multiples of 64, rows of <= 64 pixels.  Deterministic (per-workgroup partial
 * blocks in the workspace, added in a fixed order).  workspace: gdkvm_conv3x3_wgrad_workspace_bytes. */
size_t gdkvm_conv3x3_wgrad_workspace_bytes(int N, int C, int H, int W, int K);
int gdkvm_conv3x3_wgrad(const void* x, const void* dy, float* dw, void* workspace, size_t workspace_bytes,
                        int N, int C, int H, int W, int K, int io_dtype, void* stream);
/* The same gradient written as [K, 3, 3, C] -- the memory order of a channels_last [K, C, 3, 3] parameter (the framework re-lays a
 * contiguous gradient out with one copy per layer and step). */
int gdkvm_conv3x3_wgrad_krsc(const void* x, const void* dy, float* dw, void* workspace, size_t workspace_bytes,
                        int N, int C, int H, int W, int K, int io_dtype, void* stream);
int gdkvm_conv_bias_act(const void* x, const void* w, const float* bias, const void* residual, void* y,
                        int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int relu, int kernel,
                        int io_dtype, void* stream);
/* A residual block's first convolution and its 1x1 downsample branch in ONE launch of kernel 9:  y = act(conv_RxS(x, w) + bias)  and
 * y_down = conv_1x1(x, w_down) (+ bias_down, may be NULL), both with the same stride and K output channels -- the 1x1's input pixel
 * is the centre of the R x S window (R = S odd, pad = R / 2).  w and w_down are gdkvm_conv_igemm_pack_weights copies of the
 * [K, R, S, C] and [K, 1, 1, C] weights; C a multiple of 32, K of 128; bf16. */
int gdkvm_conv_down_bias_act(const void* x, const void* w, const float* bias, void* y, int relu,
                             const void* w_down, const float* bias_down, void* y_down,
                             int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int io_dtype, void* stream);
/* Training form of the same pair (a residual block's 3x3 / stride-2 / pad-1 convolution w [K, C, 3, 3] and its 1x1 / stride-2 branch
 * w_down [K, C, 1, 1], both without bias), forward and backward on hand-written kernels, deterministic:
 *   gdkvm_conv_s2_pack_train   ONE launch from the fp32 master weights (element strides (k, c, r, s) resp. (k, c): any memory format) to
 *                              packed_fwd / packed_fwd_down (what gdkvm_conv_igemm_pack_weights makes of the bf16-rounded weights: feed
 *                              them to gdkvm_conv_down_bias_act, stride 2, pad 1) and packed_dgrad (gdkvm_conv_s2_dgrad_pack_bytes).
 *                              w_down may be NULL (no branch).  K a multiple of 128, C of 64.
 *   gdkvm_conv_s2_dgrad        dx [N, H, W, C] = the gradient of x from dy [N, Ho, Wo, K] (and dy_down, same shape, or NULL), Ho = (H-1)/2+1:
 *                              every input pixel sums exactly the taps that reach it (four parity classes, no products with zeros), the
 *                              branch's gradient inside the same accumulation, one rounding to bf16.  with_down: was packed_dgrad made WITH the
 *                              1x1 branch (gdkvm_conv_s2_pack_train with w_down != NULL)?  The class offsets inside the pack follow
 *                              this, not dy_down: with_down = 1 and dy_down = NULL is a zero branch gradient; dy_down with
 *                              with_down = 0 is GDKVM_ERR_ARG.
 *   gdkvm_conv_wgrad_strided   dw[k sk + c sc + r sr + s ss] (fp32, element strides: the parameter's own memory format) =
 *                              sum_{n, yo, xo} dy[n, yo, xo, k] x[n, yo stride - pad + r, xo stride - pad + s, c]  for ANY R x S / stride /
 *                              pad; C a multiple of 64, K of 8; the rows of the batch are summed in fixed-order splits (no atomics).
 * (The reference's own recipe is a multi-process DDP launch, /root/reference/website/src/pages/[lang]/reprod/index.astro:238-249: with
 * every gradient of the step deterministic, the DDP-wrapped step at world size 1 is bit-equal to the bare one.) */
size_t gdkvm_conv_s2_dgrad_pack_bytes(int C, int K, int with_down);
int gdkvm_conv_s2_pack_train(const float* w, const long long* w_strides, const float* w_down, const long long* w_down_strides,
                             void* packed_fwd, void* packed_fwd_down, void* packed_dgrad, int K, int C, void* stream);
int gdkvm_conv_s2_dgrad(const void* dy, const void* dy_down, const void* packed_dgrad, void* dx,
                        int N, int C, int H, int W, int K, int with_down, int io_dtype, void* stream);
size_t gdkvm_conv_wgrad_strided_workspace_bytes(int N, int C, int H, int W, int K, int R, int S, int stride, int pad);
int gdkvm_conv_wgrad_strided(const void* x, const void* dy, float* dw, long long sk, long long sc, long long sr, long long ss,
                             void* workspace, size_t workspace_bytes,
                             int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int io_dtype, void* stream);
/* The same 3x3 / 1 / 1 convolution over the channel concatenation [x1 (C1 channels) ; x2 (C2)] of two NHWC tensors, which is
 * never materialised -- the decoder's  conv(cat(upsample(feature), skip))  without the concatenated copy.  C1, C2 multiples of
 * 64, K of 16, rows of <= 64 pixels; w [K, 3, 3, C1 + C2] or its packed copy (kernel | GDKVM_CONV_PACKED_WEIGHTS); kernel 0 or
 * 5..8, 10, 11.  Bit-identical to gdkvm_conv_bias_act on the concatenated tensor. */
int gdkvm_conv_cat_bias_act(const void* x1, const void* x2, const void* w, const float* bias, const void* residual, void* y,
                            int N, int C1, int C2, int H, int W, int K, int relu, int kernel, int io_dtype, void* stream);

/* Row n1, the stem: bias + ReLU + 3x3 / stride 2 / pad 1 max-pool in one pass over the NHWC conv output
 * x [N, H, W, C] -> y [N, (H-1)/2+1, (W-1)/2+1, C]:  y = relu(max_window(x) + bias)  (== max_window(relu(x + bias)), the
 * rounding of x + b being monotonic).  The full-resolution activation is read once and never written back.  x may be the
 * top-left H x W corner of a larger stored image of x_rows x x_cols pixels (x_rows >= H, x_cols >= W). */
int gdkvm_bias_relu_maxpool(const void* x, const float* bias, void* y, int N, int H, int W, int C,
                            int x_rows, int x_cols, int io_dtype, void* stream);

/* Row n1, the inference stem in one kernel: y = maxpool3x3/2/pad1(relu(conv4x4/1/pad2(xs, w) + bias)) cropped to the Hs x Ws
 * convolution outputs -- xs [N, Hs, Ws, 16] is the space-to-depth image of gdkvm_stem_s2d, w [64, 4, 4, 16] the 4x4 kernel
 * model.FusedConvPool.enable_s2d builds from a 7x7 / stride 2 stem, y [N, (Hs-1)/2+1, (Ws-1)/2+1, 64]; bf16, 64 output
 * channels.  The full-resolution activation exists only as an LDS tile (csrc/stem_conv_pool.hip). */
int gdkvm_stem_conv_pool(const void* xs, const void* w, const float* bias, void* y, int N, int Hs, int Ws,
                         int io_dtype, void* stream);
/* The same kernel reading the NCHW frames x [N, C <= 4, H, W] themselves (H, W even; Hs = H/2, Ws = W/2): the workgroup builds its band of
 * the space-to-depth image on the way into LDS, so gdkvm_stem_s2d and its 16-channel copy of the input are not needed.  Bit-identical to
 * gdkvm_stem_s2d followed by gdkvm_stem_conv_pool. */
int gdkvm_stem_conv_pool_nchw(const void* x, const void* w, const float* bias, void* y, int N, int C, int H, int W,
                              int io_dtype, void* stream);
/* Training: the stem's raw convolution (7x7 / stride 2 / pad 3, no bias) on the same kernel, y [N, H/2, W/2, 64] bf16 written to memory
 * (BatchNorm follows), from NCHW frames and the kernel in the space-to-depth form w4 [64, 4, 4, 16] bf16 -- which gdkvm_stem_pack_s2d
 * builds from the fp32 [64, C, 7, 7] weight (element strides sk, sc, sr, ss) in one small launch per step. */
int gdkvm_stem_conv_nchw(const void* x, const void* w4, void* y, int N, int C, int H, int W, int io_dtype, void* stream);
int gdkvm_stem_pack_s2d(const float* w7, void* w4, int C, long long sk, long long sc, long long sr, long long ss, void* stream);
/* Training: the stem convolution's weight gradient dw7 [64, C, 7, 7] fp32 (element strides sk, sc, sr, ss; every element written) from
 * the NCHW frames x [N, C <= 4, H, W] bf16 and dy [N, H/2, W/2, 64] bf16 (NHWC), as a pixel-axis product in the space-to-depth form with a
 * fixed-order reduction over workgroup partials held in `workspace` (gdkvm_stem_wgrad_workspace_bytes): deterministic, unlike the framework
 * convolution's atomically accumulated gradient it replaces. */
size_t gdkvm_stem_wgrad_workspace_bytes(int N, int H, int W);
int gdkvm_stem_wgrad_nchw(const void* x, const void* dy, float* dw, long long sk, long long sc, long long sr, long long ss,
                          void* workspace, size_t workspace_bytes, int N, int C, int H, int W, int io_dtype, void* stream);

/* Row n1, the training stem: 3x3 / stride 2 / pad 1 max-pool of an NHWC tensor x [N, H, W, C] -> y [N, Ho, Wo, C]
 * (Ho = (H-1)/2+1) recording the winning tap of every output element in idx (one byte each, [N, Ho, Wo, C]: 3*dy + dx in
 * window coordinates; first maximum in scan order, NaN wins -- PyTorch's rule), and its backward: dx [N, H, W, C] gathers dy
 * from the at most four windows that contain a pixel (deterministic, dx written once). */
int gdkvm_maxpool_fwd(const void* x, void* y, void* idx, int N, int H, int W, int C, int io_dtype, void* stream);
int gdkvm_maxpool_bwd(const void* dy, const void* idx, void* dx, int N, int H, int W, int C, int io_dtype, void* stream);

/* Row n1, the stem input: NCHW frames x [N, C, H, W] (H, W even) -> the space-to-depth image out [N, H/2, W/2, Cp] (NHWC),
 * out[n, i, j, (c*2 + p)*2 + q] = x[n, c, 2i + p, 2j + q], channels 4C .. Cp-1 zero.  A k x k / stride 2 convolution on x is
 * a ceil(k/2)+... x stride 1 convolution on out (gdkvm_amd/model.py::FusedConvPool builds the 4x4 kernel of a 7x7 stem). */
int gdkvm_stem_s2d(const void* x, void* out, int N, int C, int H, int W, int Cp, int io_dtype, void* stream);

/* Row n1, the training side: BatchNorm in batch-statistics mode fused with the residual add and the ReLU that follow it
 * (what torch.nn.BatchNorm2d(train) -> (+ skip) -> ReLU computes on an NHWC conv output), forward and backward.
 * x, residual, y, dy, dx, dres: [rows, C] in io_dtype (rows = N*H*W of an NHWC tensor, C a multiple of 8 (bf16) / 4 (f32));
 * gamma, beta, running_*, dgamma, dbeta: fp32 [C];  save_stats: fp32 [4][C] = mean, 1/sqrt(var+eps), scale, shift.
 *   fwd:  mean, var (biased) over the rows;  y = act(x*scale + shift (+ residual)),  scale = gamma rstd, shift = beta - mean scale;
 *         running = (1 - momentum) running + momentum stat (variance unbiased; either pointer may be NULL).
 *   bwd:  g = dy masked by the ReLU;  dbeta = sum g,  dgamma = sum g xhat,  dx = gamma rstd (g - dbeta/n - xhat dgamma/n);
 *         dres (optional) receives g, the gradient of the residual branch.
 *         relu: 0 = none, 1 = mask from the saved output (y > 0), 2 = mask recomputed from x (x*scale + shift > 0, the forward's
 *         own expression: valid when the forward had NO residual; y is then not read and may be NULL).
 * Deterministic (no atomics).  ws: gdkvm_bn_workspace_bytes(C) bytes of scratch per call. */
size_t gdkvm_bn_workspace_bytes(int C);
int gdkvm_bn_fwd_train(const void* x, const void* residual, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, void* y, float* save_stats,
                       void* ws, size_t ws_bytes, long long rows, int C, float eps, float momentum, int relu,
                       int io_dtype, void* stream);
int gdkvm_bn_bwd(const void* x, const void* y, const void* dy, const float* gamma, const float* save_stats,
                 void* dx, void* dres, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                 long long rows, int C, int relu, int io_dtype, void* stream);
/* The training stem's tail as ONE op: BatchNorm(batch statistics) -> ReLU -> 3x3 / stride 2 / pad 1 max-pool of the raw convolution
 * x [N, H, W, C] (bf16, NHWC): y_pool [N, Ho, Wo, C] (Ho = (H-1)/2+1) and the winning taps idx (one byte per element, gdkvm_maxpool_fwd's
 * format and tie rule) -- the normalised full-resolution activation is never written (every pooled element evaluates its window's
 * relu(x*scale + shift), rounded as gdkvm_bn_fwd_train would have stored it: the same bits as the two ops in sequence).  Backward:
 * dx, dgamma, dbeta from dy_pool, idx and x -- the pre-pool gradient is gathered inside the two BatchNorm backward passes
 * (gdkvm_maxpool_bwd's rule and rounding) and never written either.  Fewer than 2^22 pixels per call; bf16 only; deterministic. */
int gdkvm_bn_pool_fwd_train(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                            void* y_pool, void* idx, float* save_stats, void* ws, size_t ws_bytes,
                            int N, int H, int W, int C, float eps, float momentum, int io_dtype, void* stream);
int gdkvm_bn_pool_bwd(const void* x, const void* dy_pool, const void* idx, const float* gamma, const float* save_stats,
                      void* dx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                      int N, int H, int W, int C, int io_dtype, void* stream);

/* decoder glue, out = concat(bilinear_upsample(lo -> H x W), skip) over
 * NHWC tensors  lo [N,hl,wl,C1], skip [N,H,W,C2], out [N,H,W,C1+C2]; align_corners = false; bf16 only.  skip == NULL with
 * C2 == 0: the enlargement alone (its consumer, gdkvm_conv_cat_bias_act, reads the skip tensor where it lies). */
int gdkvm_upsample_cat(const void* lo, const void* skip, void* out,
                       int Nimg, int hl, int wl, int H, int W, int C1, int C2, int io_dtype, void* stream);
/* Backward of gdkvm_upsample_cat (training): dlo [Nimg, hl, wl, C1] = transpose of the bilinear enlargement applied to
 * dout[..., :C1] (a deterministic gather: no atomics), dskip [Nimg, H, W, C2] = dout[..., C1:].  bf16 only. */
int gdkvm_upsample_cat_bwd(const void* dout, void* dlo, void* dskip,
                           int Nimg, int hl, int wl, int H, int W, int C1, int C2, int io_dtype, void* stream);

/* Rows n1/n2, the training objective on the decoder's stride-4 logits z [images, C, h, w] (NCHW, io_dtype) and integer labels
 * target [images, H, W] (target_bytes = 1: uint8, 8: int64), 2 <= C <= 8:
 *   l = bilinear_upsample(z -> H x W, align_corners = false) in fp32;  p = softmax(l);
 *   loss = mean CE(l, target) + dice_weight * (1 - mean_c (2 I_c + eps) / (P_c + O_c + eps)),  sums over the whole batch.
 * A label outside [0, C) marks an unlabelled pixel (EchoNet-Dynamic: every frame but the two traced ones): it adds to no sum at all -- the CE
 * mean runs over the labelled pixels, and neither I_c, O_c nor P_c see it (round 5; before, its softmax still counted in P_c, which pushed
 * the predictions on unlabelled frames towards "nothing").
 * fwd: out[0] = loss, out[1] = CE, out[2] = Dice term; ws keeps the per-class coefficients the backward needs.
 * bwd: dz [images, C, h, w] = (*grad_out, or 1 if NULL) * dloss/dz -- a gather per stride-4 pixel, deterministic.
 * Full-resolution logits are never materialised.  ws: gdkvm_seg_loss_workspace_bytes(C), the SAME buffer for fwd and bwd. */
size_t gdkvm_seg_loss_workspace_bytes(int C);
int gdkvm_seg_loss_fwd(const void* z, const void* target, float* out, void* ws, size_t ws_bytes,
                       int images, int C, int h, int w, int H, int W, float dice_weight, float eps,
                       int io_dtype, int target_bytes, void* stream);
int gdkvm_seg_loss_bwd(const void* z, const void* target, const void* ws, size_t ws_bytes, const float* grad_out, void* dz,
                       int images, int C, int h, int w, int H, int W, int io_dtype, int target_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GDKVM_H */
