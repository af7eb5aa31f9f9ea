/*
 * stv.h - C ABI of libstv_hip.so: the MI355X (gfx950) kernels behind the
 * per-step optimisation path of style_transfer_visualizer.
 *
 * The reference has no FFI; its boundary for this path is the Python API of
 * core_model.py / optimization.py, whose arithmetic is delegated to torch.
 * Each entry point below replaces the torch op(s) issued at the cited
 * reference line (paths relative to
 * /root/reference/src/style_transfer_visualizer/).  See INTEGRATION.md for the
 * ctypes binding a maintainer would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (PyTorch allocates);
 *    the library never frees or retains a pointer past the call, except the
 *    command buffer of stv_exec_*, which is copied at creation;
 *  - activations are NHWC ([H][W][C], batch 1) in `dtype` storage
 *    (STV_F32 | STV_BF16); the image and its gradient are NCHW fp32, exactly
 *    the tensors the reference optimises (core_model.py:66-100);
 *  - `stream` is a hipStream_t; all work is enqueued on it, nothing syncs;
 *  - return value: 0 = ok, otherwise an STV_ERR_* code (Python raises
 *    RuntimeError).  No CPU fallback exists.
 */
#ifndef STV_H_
#define STV_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { STV_F32 = 0, STV_BF16 = 1 };

enum {
  STV_OK = 0,
  STV_ERR_ARG = 1,      /* unsupported shape / null pointer / bad dtype */
  STV_ERR_LAUNCH = 2,   /* hipGetLastError() after a launch */
  STV_ERR_ALLOC = 3,
  STV_ERR_GRAPH = 4
};

/* flags */
enum {
  STV_RELU_IN = 1,   /* apply max(0,.) to the input while staging it */
  STV_RELU_OUT = 2,  /* apply max(0,.) before storing (conv+ReLU fusion) */
  STV_MASK = 4,      /* multiply result by (ref > 0): ReLU backward */
  STV_ACCUM = 8,     /* out += result instead of out = result */
  STV_W_BLOCKED = 16, /* conv weights are K-blocked: [taps][cin/CK][cout][CK], CK = 32 bytes of `dtype` */
  STV_POOL_IDX = 32,  /* stv_maxpool_bwd: `x` is the arg-max byte map of stv_conv_igemm_pool, not the activation */
  STV_POOL_ROUTE = 64, /* stv_op_t only: the CONV op is stv_conv_igemm_route (p2 = arg-max map, q1 = routed output) */
  STV_POOL_ONLY = 128  /* stv_conv_igemm_pool: do not store the full-resolution map `y` (only y_pool / pool_idx are wanted) */
};

int stv_version(void);
/* Algorithmic workspace sizing helpers (bytes). */
size_t stv_gram_partials_bytes(int n_pixels, int channels);
int stv_gram_ksplit(int n_pixels, int channels);

/* ---- VGG conv3x3 stacks: replaces F.conv2d / relu / max_pool2d issued by
 *      `x = block(x)` (core_model.py:316) and their autograd backward
 *      (`loss.backward()`, optimization.py:313). ---------------------------- */

/* (The two unpacked entries below read `wf` as it is, with the shape-generic first-layer
 * kernels; the library holds no scratch of its own.  The 3 -> 64 layer's fast kernels need the
 * kernel-side weight order: stv_conv_first_pack + the *_packed entries further down, which is
 * what the step path and ops.py use.) */
/* First conv: NCHW fp32 image [cin][H][W] -> NHWC `dtype` [H][W][cout].
 * wf is [9][cout][cin] fp32 (tap = ky*3+kx), bias fp32[cout]. */
int stv_conv_first_fwd(const float* x_nchw, const float* wf, const float* bias,
                       void* y, int H, int W, int cin, int cout, int dtype,
                       void* stream);
/* Its input-gradient: dy NHWC `dtype` [H][W][cout] -> dx NCHW fp32 [cin][H][W].
 * wf is the same forward-packed fp32 weight. */
int stv_conv_first_dgrad(const void* dy, const float* wf, float* dx_nchw,
                         int H, int W, int cin, int cout, int dtype, void* stream);

/* The same two calls for frozen weights: stv_conv_first_pack() writes the
 * kernel-side packing of `wf` (forward order, then flipped taps for dgrad) into
 * a caller-owned buffer of stv_conv_first_packed_bytes(cin, cout) bytes, once;
 * the *_packed entries then skip the per-call repack. */
size_t stv_conv_first_packed_bytes(int cin, int cout);
int stv_conv_first_pack(const float* wf, float* packed, int cin, int cout, void* stream);
int stv_conv_first_fwd_packed(const float* x_nchw, const float* packed, const float* bias,
                              void* y, int H, int W, int cin, int cout, int dtype,
                              void* stream);
int stv_conv_first_dgrad_packed(const void* dy, const float* packed, float* dx_nchw,
                                int H, int W, int cin, int cout, int dtype, void* stream);
/* Forward of a first layer that is a STYLE TAP (conv1_1 in the default layer set, core_model.py:213-226 /
 * 297-314): besides y it leaves the split-K slabs of R = F^T F exactly as stv_gram_partial(y) would
 * (stv_gram_ksplit(H*W, cout) slabs of [cout][cout] fp32 in `gram_partials`, to be reduced by
 * stv_gram_finish / a stv_gram_multi tap with F == NULL), computed from the rounded values it stores, so the
 * map is not read back for the Gram matrix.  bf16, cin = 3, cout = 64 only (stv_conv_first_gram_supported). */
int stv_conv_first_gram_supported(int H, int W, int cin, int cout, int dtype);
int stv_conv_first_fwd_gram(const float* x_nchw, const float* packed, const float* bias, void* y,
                            float* gram_partials, int H, int W, int cin, int cout, int dtype, void* stream);

/* Implicit-GEMM 3x3 conv, pad 1, stride 1, NHWC.  `w` is [taps][cout][cin] in
 * `dtype` (K-contiguous rows); bias fp32[cout] or NULL.  taps = 9 (3x3) or 1
 * (1x1, used for the Gram backward product).  flags: RELU_IN, RELU_OUT,
 * MASK (needs `ref`, NHWC [H][W][cout]), ACCUM, W_BLOCKED.  The same entry
 * computes the input gradient when given the flipped/transposed weights (dgrad).
 * With W_BLOCKED `w` is [taps][cin/CK][cout][CK] (CK = 16 bf16 / 8 fp32 channels,
 * cin % CK == 0): the 32-byte K slices a workgroup stages for its output channels
 * are then contiguous, so every weight load is a fully used cache line. */
/* Optional, once per shape (e.g. when a schedule is built): time every tile
 * configuration of stv_conv_igemm for this shape on scratch data and remember the
 * fastest for later calls (process-wide table keyed by shape; the analytic choice
 * is kept unless another tile is > 3 % faster).  Returns the configuration index
 * (>= 0; the value stv_conv_config reports afterwards), -1 when the shape runs on
 * the direct kernel (nothing to tune), or -(100 + STV_ERR_*) on failure.
 * Synchronises `stream` when it measures: only with STV_CONV_TUNE=1 / 2 in the environment (see the table
 * functions below); STV_CONV_TUNE=0 ignores the table and pins the analytic choice.
 * Different tiles sum K in different orders: results agree to fp32 rounding, not
 * bit for bit, across configurations.
 * taps = STV_TUNE_ROUTE (bf16): the shape is measured as stv_conv_igemm_route runs it (dgrad + pooling backward in
 * the epilogue, 2H x 2W output) and remembered under its own key - the routed epilogue prefers smaller tiles. */
#define STV_TUNE_ROUTE 109
int stv_conv_tune(int H, int W, int cin, int cout, int taps, int dtype, void* stream);
/* The tile table as data (7 ints per entry: H, W, cin, cout, taps - 9, 1 or STV_TUNE_ROUTE -, element bytes, tile
 * index).  Measuring only happens with STV_CONV_TUNE=1 (or 2) in the environment; otherwise stv_conv_tune and every
 * conv launch look a shape up in this table and fall back to the analytic choice.  The host persists measured
 * tables (conv_tiles_gfx950.json) and imports them at load time, so that a shape runs on the same tile - same
 * summation order, same kernel name in a profile - in every process.  stv_conv_tune_export returns the number of
 * entries (and writes up to max_entries of them); stv_conv_tune_import adds / overwrites entries (0 entries: clears
 * the table), STV_ERR_ARG on a malformed entry. */
int stv_conv_tune_export(int* out7, int max_entries);
int stv_conv_tune_import(const int* in7, int n_entries);

/* Scratch for convolutions that split K across workgroups (small-grid 3x3 bf16 layers: two workgroups share an
 * output tile, each walks half of K, the later one adds the other's fp32 partial sums - a + b in fp32, the same
 * whichever came first - and runs the epilogue).  Caller-owned device memory of at least
 * stv_conv_workspace_bytes() bytes, ZEROED once by the caller (the kernels leave it zeroed); applies to the conv
 * launches this host thread makes until it is changed; NULL / 0 clears it (then no launch splits K).  One
 * workspace must not serve two launches that run at the same time.  No reference counterpart: an execution
 * detail under F.conv2d (core_model.py:316). */
void stv_conv_workspace(void* ws, size_t bytes);
size_t stv_conv_workspace_bytes(void);
/* Hint for the NEXT stv_conv_igemm* launch issued from this thread: `bytes` of weights that the conv launched AFTER
 * it will read.  That next launch touches them (one 128-byte line per lane, between its main loop and its epilogue) so
 * they are on chip when their own launch starts.  One shot: the launch that follows on this thread consumes the hint
 * (whichever kernel serves it), so a hint can never reach a later, unrelated launch; the tile tuner runs without one.  No
 * counterpart in the reference (cuDNN / oneDNN own their weights' residency). */
void stv_conv_next_weights(const void* w, size_t bytes);

/* Number of tile configurations stv_conv_config() can return (0 .. n-1): tools and the bench that name the kernel
 * instantiation of a tile check their tables against it. */
int stv_conv_num_configs(void);
int stv_conv_igemm(const void* x, const void* w, const float* bias, const void* ref,
                   void* y, int H, int W, int cin, int cout, int taps, int flags,
                   int dtype, void* stream);

/* Forward 3x3 conv that also emits MaxPool2d(2,2) of its output (`y_pool`,
 * NHWC [H/2][W/2][cout]) from the same epilogue: replaces the conv2d + relu +
 * max_pool2d run of `x = block(x)` (core_model.py:316) where a pool follows the
 * conv.  Matrix-core shapes only (stv_conv_config >= 0), else STV_ERR_ARG and the
 * caller pools with stv_maxpool_fwd.  flags: RELU_IN, RELU_OUT, W_BLOCKED.
 * `pool_idx` (optional, [H/2][W/2][cout] bytes) receives what max_pool2d's backward needs
 * instead of the full-resolution activation: bits 0-1 = window position (2*dy + dx) of the
 * first maximum in scan order, bit 2 = that maximum is > 0 (the ReLU mask of the winner);
 * decided on the values as stored in `y`.  stv_maxpool_bwd takes it with STV_POOL_IDX.
 * STV_POOL_ONLY: `y` is not written (it may be NULL) - for a conv whose full-resolution output nobody reads again:
 * the forward pass continues from `y_pool`, the backward pass routes through `pool_idx` (stv_conv_igemm_route /
 * stv_maxpool_bwd with STV_POOL_IDX), so the pre-pool map - four times the pooled one - never has to reach HBM.
 * The pooled map and the arg-max map are bit-identical to the ones written without the flag. */
int stv_conv_igemm_pool(const void* x, const void* w, const float* bias, void* y, void* y_pool,
                        void* pool_idx, int H, int W, int cin, int cout, int flags, int dtype,
                        void* stream);

/* Two-term gradient in one launch:
 *   y = [accumulate onto y +] mask(ref > 0) * conv3x3(x, w)  +  x2 . w2^T
 * x2 is NHWC [H][W][cin2], w2 plain [cout][cin2].  This is the backward of a layer whose
 * output feeds both the next conv (first term: that conv's dgrad, masked by this layer's
 * ReLU) and a Gram loss tap (second term: dF = F.S, `core_model.py:56-63` backward) - the
 * second term no longer needs its own launch nor a read-modify-write of y.
 * flags: MASK (applies to the first term only), ACCUM, W_BLOCKED (for w).  Matrix-core
 * shapes only (stv_conv_config(H,W,cin,cout,9,dtype) >= 0 and cin2 a multiple of the K
 * stage), else STV_ERR_ARG and the caller issues the two launches. */
int stv_conv_igemm_dual(const void* x, const void* w, const void* x2, const void* w2, const void* ref,
                        void* y, int H, int W, int cin, int cin2, int cout, int flags, int dtype,
                        void* stream);

/* Input gradient of the conv BEHIND a MaxPool2d(2,2), with the pooling backward folded into its epilogue:
 *   y_full[2H][2W][cout] <- route( conv3x3(x, w) )   (x = dy of that conv, w = its flipped/transposed weights)
 * every computed element goes to the position of its window's first maximum, as recorded by
 * stv_conv_igemm_pool in `pool_idx` ([H][W][cout] bytes), and - with STV_MASK - only where that maximum was
 * positive (the ReLU mask of the pre-pool activation, bit 2 of the byte); the other three positions of the
 * window receive zeros.  Replaces convolution_backward + max_pool2d_backward + threshold_backward of
 * `loss.backward()` (optimization.py:313) for that pair of layers, without the pooled-resolution gradient
 * ever being stored.  bf16 only, matrix-core shapes only, even 2H x 2W (the caller falls back to
 * stv_conv_igemm + stv_maxpool_bwd otherwise).  flags: MASK, W_BLOCKED. */
int stv_conv_igemm_route(const void* x, const void* w, const void* pool_idx, void* y_full, int H, int W, int cin,
                         int cout, int flags, int dtype, void* stream);

/* Which tile the dispatcher picks for a shape: -1 = scalar fallback (channel counts not a
 * multiple of the MFMA K-slice), else 0..3 = {8x128, 8x64, 4x128, 4x64} (rows x couts) and
 * 4 = 4x64 with K split over two wave groups, 5 = 8x64 and 6 = 4x64 with a two-deep LDS ring,
 * 7 = 2x64 and 8 = 1x64 (four waves) with the K split. */
int stv_conv_config(int H, int W, int cin, int cout, int taps, int dtype);

/* 1 when stv_conv_igemm / _pool / _dual run this launch on the weight-stationary persistent kernel
 * (csrc/conv_ws.hip: 3x3, cin = 64, bf16, cout a multiple of 64; with a ReLU-mask / fused 1x1 term
 * only for cout = 64 and ref == x2), 0 when the general implicit-GEMM kernel takes it.  has_ref: the
 * launch carries `ref` (STV_MASK) and/or the fused term's x2; has_pool: it emits the pooled map. */
int stv_conv_uses_ws(int H, int W, int cin, int cout, int taps, int dtype, int flags, int has_ref, int has_pool);

/* MaxPool2d(2,2) forward / backward (first-max-wins like torch); backward
 * optionally applies the ReLU mask of the stored pre-pool activation (STV_MASK) and
 * accumulates (STV_ACCUM).  With STV_POOL_IDX `x` is the [H/2][W/2][C] byte map written by
 * stv_conv_igemm_pool: the pass then reads 1 byte instead of 4 activations per window. */
int stv_maxpool_fwd(const void* x, void* y, int H, int W, int C, int dtype, void* stream);
int stv_maxpool_bwd(const void* x, const void* dy, void* dx, int H, int W, int C,
                    int flags, int dtype, void* stream);
int stv_relu_fwd(const void* x, void* y, size_t n, int dtype, void* stream);
/* dx (=|+=) (x > 0) * dy ; dx may alias dy */
int stv_relu_bwd(const void* x, const void* dy, void* dx, size_t n, int flags,
                 int dtype, void* stream);

/* ---- Gram / style loss: replaces gram_matrix (core_model.py:29-63:
 *      mm(F,F^T).clamp(max).div(b*c*h*w)) + mse_loss (core_model.py:264) and
 *      their backward. ----------------------------------------------------- */

/* Split-K partial sums of F^T F over pixels.  F is NHWC [n_pixels][C];
 * partials is fp32 [ksplit][C][C] (upper tile triangle valid). */
int stv_gram_partial(const void* F, float* partials, int n_pixels, int C,
                     int dtype, void* stream);
/* Reduce partials -> raw Gram R; G = min(R, clamp_max) / norm.
 *  gram_out  (optional) fp32 [C][C]  <- G            (target capture)
 *  target    (optional) fp32 [C][C]
 *  loss_part (optional) fp32 [stv_gram_loss_parts(C)] <- partial sums of (G-T)^2
 *  sgrad     (optional) `dtype` [C][C] <- coef*(*coef_dev)*4/(C*C*norm) * (R<=clamp) * (G-T)
 */
int stv_gram_loss_parts(int channels);
int stv_gram_finish(const float* partials, const float* target, float* gram_out,
                    float* loss_part, void* sgrad, int n_pixels, int C, float clamp_max,
                    float norm, float coef, const float* coef_dev, int dtype, void* stream);

/* The Gram chain of several taps (layers) in one call: one batched stv_gram_partial launch per
 * tile size present plus one batched stv_gram_finish launch per slab-depth class, instead of two launches per tap.
 * The chain of a step is five small latency-bound problems; side by side in one grid they cost
 * the slowest, not the sum.  Fields as the arguments of the two calls above; at most 8 taps.
 * Same arithmetic, same fixed summation order as the per-tap calls. */
typedef struct {
  const void* F;          /* features, NHWC [n_pixels][channels] in `dtype`; NULL: `partials` is already filled
                             (stv_conv_first_fwd_gram) and only the finish pass runs for this tap */
  float* partials;        /* stv_gram_partials_bytes(n_pixels, channels) */
  const float* target;    /* [C][C] or NULL */
  float* gram_out;        /* [C][C] or NULL */
  float* loss_part;       /* stv_gram_loss_parts(channels) floats, or NULL */
  void* sgrad;            /* backward seed [C][C] in `dtype`, or NULL */
  const float* coef_dev;  /* optional device scalar multiplying the seed */
  int n_pixels, channels;
  float clamp_max, norm, coef;
} stv_gram_tap_t;
int stv_gram_multi(const stv_gram_tap_t* taps, int n_taps, int dtype, void* stream);

/* ---- Content loss: mse_loss(features, target) (core_model.py:295). -------- */
#define STV_CONTENT_LOSS_PARTS 256
int stv_content_loss(const void* F, const void* target, float* loss_part, size_t n,
                     int dtype, void* stream);
/* Both at once for a step whose coefficient is known up front (the fused step: reference optimization.py:298-313
 * with content_w folded in): the loss partials exactly as stv_content_loss leaves them, and dF = coef*(2/n)*(F - target)
 * WRITTEN - the dgrad that later produces this layer's gradient accumulates onto it (STV_ACCUM). */
int stv_content_loss_grad(const void* F, const void* target, float* loss_part, void* dF, size_t n,
                          float coef, int dtype, void* stream);
/* dF (=|+=) coef*(*coef_dev)*(2/n)*(F - target) */
int stv_content_grad(const void* F, const void* target, void* dF, size_t n, float coef,
                     const float* coef_dev, int flags, int dtype, void* stream);

/* ---- Frame / PNG export: prepare_image_for_output (image_io.py:129-152: denormalise,
 *      nan_to_num(nan=0, posinf=1, neginf=0), clamp 0..1) fused with the uint8 conversion, on the
 *      device, so only H*W*3 bytes cross to the host.  x NCHW fp32 [3][H][W] -> out HWC uint8.
 *      mean3/std3: HOST arrays of 3 floats (ImageNet statistics, constants.py:11-12) or both NULL
 *      for an un-normalised image.  round = 0: (uint8)(v*255), the truncating frame conversion of
 *      optimization.py:445-451; round = 1: (uint8)clamp(v*255 + 0.5, 0, 255), what
 *      torchvision.utils.save_image does for the final PNG (runtime/output.py:101). */
int stv_image_to_u8(const float* x_nchw, uint8_t* out_hwc, int H, int W, const float* mean3,
                    const float* std3, int round, void* stream);

/* ---- Score combine: torch.stack(losses).sum() and
 *      loss = style_w*style + content_w*content (optimization.py:298-312).
 *  table: int32 [n_terms][3] = {offset into parts, count, kind(0 style,1 content)}
 *  scale: fp32 [n_terms] (1/C^2 for style, 1/n for content)
 *  losses: fp32 [n_terms]; scores: fp32 [4] = {style, content, total, finite_flag}
 *  n_terms <= 64 (STV_ERR_ARG beyond). */
int stv_loss_combine(const float* parts, const int32_t* table, const float* scale,
                     int n_terms, float style_w, float content_w, float* losses,
                     float* scores, void* stream);
/* The same, and the producer also keeps the per-step history the reference's LossAccumulator keeps on the
 * device (loss_accumulator.py:97-118: one ring slot per step): log_ring fp32 [3][log_capacity] (style, content,
 * total), log_count = evaluations so far (device counter, advanced here), slot = count % capacity.  Saves the
 * per-step copy kernel between two replays of the captured step.  log_ring and log_count both NULL: as above.
 * log_seq (optional): the ring is HOST memory from stv_host_mailbox_alloc and *log_seq, in the same allocation, receives
 * the record count after the record (system-scope release): the host reads a step's scores as soon as the combine kernel
 * has run - no copy, no stream synchronisation, the rest of the step still in flight (the reference's logging point,
 * optimization.py:375-391, blocks on .item() there). */
int stv_loss_combine_log(const float* parts, const int32_t* table, const float* scale,
                         int n_terms, float style_w, float content_w, float* losses,
                         float* scores, float* log_ring, int log_capacity, uint32_t* log_count,
                         uint32_t* log_seq, void* stream);
/* Pinned host memory, mapped under the same pointer on the device, fine-grained coherent, zeroed. */
int stv_host_mailbox_alloc(size_t bytes, void** out);
void stv_host_mailbox_free(void* p);

/* ---- Optimizer updates (torch.optim.LBFGS.step / Adam.step driven from
 *      optimization.py:175), device resident: no host synchronisation. ----- */

typedef struct stv_lbfgs_state stv_lbfgs_state;  /* opaque, lives in device memory */
size_t stv_lbfgs_state_bytes(int history);
size_t stv_lbfgs_workspace_bytes(size_t n, int history);
/* state/workspace zero-initialised by the caller before the first step. */
int stv_lbfgs_step(float* x, const float* grad, void* state, void* workspace, size_t n,
                   int history, int m_max, float lr, float tol_grad, float tol_change,
                   void* stream);
/* Same update with the history read twice per step instead of through 2m dependent
 * passes (inner products of {s_i},{y_i},g kept in S x S tables; see lbfgs_compact.hip).
 * Own state/workspace layout; both zero-initialised by the caller. */
size_t stv_lbfgsc_state_bytes(int history);
size_t stv_lbfgsc_workspace_bytes(size_t n, int history);
int stv_lbfgsc_step(float* x, const float* grad, void* state, void* workspace, size_t n,
                    int history, int m_max, float lr, float tol_grad, float tol_change,
                    void* stream);
/* The same step in two halves, for an image sharded over several processes (one 4K image as row strips,
 * DESIGN.md §6): stv_lbfgsc_dots leaves the step's inner products - partial sums over THIS shard - as
 * doubles in the workspace at stv_lbfgsc_dots_offset(); the caller all-reduces them (SUM, except entry
 * *max_index = max|g|: MAX) and stv_lbfgsc_apply then runs the identical scalar recursion on every shard
 * and updates its own elements.  stv_lbfgsc_step == dots + apply. */
int stv_lbfgsc_dots(const float* grad, void* state, void* workspace, size_t n, int history, int m_max, void* stream);
int stv_lbfgsc_apply(float* x, const float* grad, void* state, void* workspace, size_t n, int history, float lr,
                     float tol_grad, float tol_change, void* stream);
size_t stv_lbfgsc_dots_offset(size_t n, int history, int* count, int* max_index);
/* scalars are computed in double on the host exactly as torch does
 * (1-beta1, 1-beta2, 1-beta1**t, sqrt(1-beta2**t)) and passed rounded to fp32 */
int stv_adam_step(float* x, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                  float lr, float one_minus_beta1, float beta2, float one_minus_beta2, float eps,
                  float bias_c1, float bias_c2_sqrt, void* stream);

/* ---- Command-buffer executor: one call runs a whole forward(+backward)
 *      schedule built by the host (style_transfer_visualizer_amd/plan.py);
 *      optionally captured into a hipGraph and replayed. ------------------- */
enum {
  STV_OP_CONV_FIRST_FWD = 1, STV_OP_CONV_FIRST_DGRAD, STV_OP_CONV, STV_OP_POOL_FWD,
  STV_OP_POOL_BWD, STV_OP_RELU_FWD, STV_OP_RELU_BWD, STV_OP_GRAM_PARTIAL,
  STV_OP_GRAM_FINISH, STV_OP_CONTENT_LOSS, STV_OP_CONTENT_GRAD, STV_OP_LOSS_COMBINE,
  STV_OP_MEMSET, STV_OP_GRAM_MULTI, STV_OP_LBFGS_STEP
};
/* Scheduling hints in stv_op_t.flags (masked off before the kernel sees them):
 * an op with STV_LANE_SIDE may run concurrently with the ops after it: it reads only
 * what earlier ops produced, and what it writes is first read by an op flagged
 * STV_LANE_JOIN (or after the program).  The executor runs such ops on a second
 * stream forked from / joined to the caller's stream with events. */
enum { STV_LANE_SIDE = 1 << 29, STV_LANE_JOIN = 1 << 30 };
/* Operands follow the direct entry points' argument order (inputs p0.., outputs q0..).
 * CONV_FIRST_FWD takes the optional stv_conv_first_pack buffer in p3 (and, with q1 set, runs
 * stv_conv_first_fwd_gram with q1 = gram_partials), CONV_FIRST_DGRAD in p2;
 * CONV with q1 set runs stv_conv_igemm_pool (q1 = pooled output, q2 = optional arg-max map); CONV with q2 AND q3 set runs
 * stv_conv_igemm_dual (q2 = x2, q3 = w2, n = cin2: inputs, despite the slot names).
 * GRAM_MULTI: p0 = HOST pointer to an array of stv_gram_tap_t, n = its length; the array is
 * copied into the program when it is created.
 * CONTENT_LOSS with q1 set runs stv_content_loss_grad (q1 = dF, f0 = coef).
 * LOSS_COMBINE with q2 AND q3 set runs stv_loss_combine_log (q2 = log_ring, q3 = log_count, n = log_capacity, p3 = log_seq).
 * LBFGS_STEP runs stv_lbfgsc_step as the LAST op of a step's schedule (p0 = grad, q0 = x, q1 = state, q2 = workspace,
 * n = elements, cin = history, cout = m_max, f0 = lr, f1 = tol_grad, f2 = tol_change): closure and optimizer update are
 * then one replayed hipGraph - the reference's optimizer.step(closure), optimization.py:186, without a graph boundary
 * between the two. */
typedef struct {
  int32_t op, dtype, flags, taps;
  int32_t H, W, cin, cout;
  int64_t n;
  float f0, f1, f2, f3;
  const void* p0; const void* p1; const void* p2; const void* p3;
  void* q0; void* q1; void* q2; void* q3;
} stv_op_t;

typedef struct stv_program stv_program;  /* host object */
int stv_program_create(const stv_op_t* ops, int n_ops, stv_program** out);
/* use_graph != 0: capture on first run, replay afterwards. */
int stv_program_run(stv_program* prog, int use_graph, void* stream);
/* Measurement helper: eager run with a hipEvent pair around every op (recorded on
 * `stream`); ms_out[i] = device milliseconds of op i.  Synchronises. */
int stv_program_profile(stv_program* prog, void* stream, float* ms_out, int n_out);
/* The same with each op launched `reps` times back to back inside its event pair (time divided by
 * reps): amortises the event pair's own few microseconds.  Leaves the buffers meaningless. */
int stv_program_profile_reps(stv_program* prog, void* stream, int reps, float* ms_out, int n_out);
int stv_program_op_count(const stv_program* prog);
void stv_program_destroy(stv_program* prog);

#ifdef __cplusplus
}
#endif
#endif /* STV_H_ */
