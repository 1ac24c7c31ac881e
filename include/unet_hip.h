/* libunet_hip.so -- C ABI of the MI355X (gfx950) U-Net hot path.
 *
 * The reference (abin102/unet) has no native/FFI boundary: its hot path is reached through
 * torch.nn modules (models/unet.py, models/reparm_unet.py, losses/segmentation.py).  Each entry
 * point below names the reference call site whose ATen/cuDNN work it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; the library never allocates, frees or
 *     retains device memory (workspaces are arguments);
 *   - `stream` is a hipStream_t; every call only enqueues work on it;
 *   - dtype: UNET_F32 (0) or UNET_BF16 (1) = storage type of activations and packed weights;
 *     accumulation is always fp32 (fp64 for cross-workgroup sums);
 *   - activations are NHWC; a tensor argument is (pointer, ld = elements per pixel, off = first
 *     channel) so kernels read and write channel slices of wider buffers (the skip-concat buffer);
 *   - return 0 on success, negative on error with text from unet_last_error() (thread local).
 */
#ifndef UNET_HIP_H
#define UNET_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define UNET_F32 0
#define UNET_BF16 1

int unet_version(void);
const char* unet_last_error(void);

/* ---- MFMA implicit-GEMM convolution ---------------------------------------------------------
 * nn.Conv2d(k=3,pad=1) forward and dgrad (models/unet.py:10,13), the RepUNet kx1/1xk/kxk
 * branches and deployed conv (models/reparm_unet.py:55-61,139), nn.ConvTranspose2d(2,2) forward
 * (epi=1: 1x1 GEMM + 2x2 pixel shuffle straight into the concat buffer, models/unet.py:35,53,59)
 * and its dgrad (kh=2,kw=1,stride_h=2 on the [N,2H,W,2C] view of the output gradient).
 * w: packed [kh*kw][Cout][Cin] (unet_pack_weights).  bias/stats may be NULL.  stats receives
 * unet_conv_tiles() rows of [Cout][2] = partial (sum, sum of squares) of the fp32 outputs; only the
 * sum over the rows is defined (a row may carry several tiles' sums and others zeros; the row count depends on the
 * kernel that serves the launch, hence the launch's Cin / Cout / flags / csplit in unet_conv_tiles) -- unless flags
 * bit 2 asks for per-tile rows grouped per image ([N][tiles per image], what unet_gn_finalize reads). */
int unet_conv_tiles(int dtype, int kh, int kw, int stride_h, int epi, int N, int Hout, int Wout,
                    int Cin, int Cout, int flags, int csplit);
int unet_conv_fwd(int dtype, int kh, int kw, int stride_h, int epi,
                  const void* x, const void* w, const float* bias, void* y, float* stats,
                  int N, int Hin, int Win, int Cin, int ldx, int xoff,
                  int Hout, int Wout, int Cout, int ldy, int yoff,
                  int pad_h, int pad_w,
                  void* y2, int ldy2, int yoff2, int csplit, int flags, void* stream);
/* flags bit 0: y += conv(x) (the RepUNet branch dgrads sum into one input gradient); bit 1: ReLU
 * on the stored result (deployed RepUNet: conv + bias + ReLU, models/reparm_unet.py:162-167); bit 2: statistics rows
 * per tile and per image (GroupNorm, models/reparm_unet.py:17-25).
 * csplit > 0: output channels >= csplit are written to y2 (channel co - csplit) instead of y; the
 * dgrad of the first decoder conv uses it to hand the skip half and the upsampled half of the
 * concat gradient to two dense buffers (torch.cat backward, models/unet.py:59). */

/* weight gradient of the same convolutions (autograd of models/unet.py:10,13,35).
 * slab: workspace of unet_conv_wgrad_splits() * kh*kw*Cout*Cin floats.
 * dw[o][i][tap] for tap in [dw_tap0, dw_tap0+kh*kw) of dw_taps taps per (o,i). */
int unet_conv_wgrad_splits(int dtype, int kh, int kw, int stride_h, int N, int Hout, int Wout, int Cin, int Cout);
int unet_conv_wgrad(int dtype, int kh, int kw, int stride_h,
                    const void* dz, const void* x, float* slab, float* dw, int accumulate,
                    int dw_taps, int dw_tap0,
                    int N, int Hin, int Win, int Cin, int ldx, int xoff,
                    int Hout, int Wout, int Cout, int lddz, int dzoff,
                    int pad_h, int pad_w, void* stream);
/* ConvTranspose2d weight grad: call unet_conv_wgrad(kh=2,kw=1,stride_h=2, dw=NULL) with dz := the
 * layer input [N,H,W,Ci] and x := the [N,2H,W,2*Co] view of the output gradient (the slab sum is
 * then skipped), then this on the slab [nsplit][2][Ci][2*Co] -> dw (Ci,Co,2,2). */
int unet_convT_wgrad_reduce(const float* slab, float* dw, int nsplit, int Ci, int Co, void* stream);

/* ---- first conv (Cin<=4, fp32 image in) and final 1x1 conv (fp32 NCHW logits out) -----------
 * models/unet.py:30 (in_channels=1) and models/unet.py:39,62. */
int unet_conv_first_rows(int dtype, int N, int H, int W, int Cout);
/* z may be NULL for the Cin = 1 3x3 case: statistics rows only (see unet_first_bn_act). */
int unet_conv_first_fwd(int dtype, int kh, int kw, const float* x, const float* w, const float* bias, void* z, float* stats,
                        int N, int H, int W, int Cin, int Cout, void* stream);
int unet_conv_first_wgrad(int dtype, int kh, int kw, const void* dz, const float* x, float* dw,
                          int N, int H, int W, int Cin, int Cout, double* acc_ws, void* stream);
int unet_conv_out_fwd(int dtype, const void* x, int ldx, int xoff, const float* w, const float* bias, float* logits,
                      long long M, long long HW, int C, int K, void* stream);
int unet_conv_out_bwd(int dtype, const float* dlogits, const void* x, int ldx, int xoff, const float* w,
                      void* dx, float* dw, float* db, long long M, long long HW, int C, int K,
                      double* acc_ws, void* stream);

/* ---- the two ends of the U-Net fused into their BatchNorm passes (edge_fused.hip) --------------
 * First layer, in_channels = 1, 3x3 (models/unet.py:30 + :11-12).  Forward: unet_conv_first_fwd with z = NULL writes only the
 * statistics rows; after unet_bn_finalize, unet_first_bn_act computes y = act(scale*conv(x)+shift) in one pass: the pre-norm
 * tensor is never stored.  Backward: unet_first_bn_relu_bwd = unet_bn_relu_bwd with that tensor recomputed from x, plus the
 * conv's weight gradient dw (Cout,1,3,3) (dz is not stored: the first conv has no input gradient).  ONE pass over dy: dz is
 * affine in (dy', z) with per-channel coefficients, so dw follows from sum dy'*patch (accumulated next to the BN sums) and
 * the 3x3 patch moments of x (a small kernel of their own).  relu bits as unet_bn_relu_bwd.  acc_ws: 12*Cout + 64 doubles.
 * unet_first_stats_moments: the same patch moments give the BatchNorm batch statistics of the first conv without computing its
 * output (sum z = w.s + n b, sum z^2 = w^T R w + ...): `stats` receives two rows [2][Cout][2] for unet_bn_finalize (value +
 * rounding remainder), `mom` the 64 doubles that unet_first_bn_relu_bwd takes as `moments` (NULL there: computed again). */
int unet_first_stats_moments(const float* x, const float* w, const float* bias, int N, int H, int W, int Cout,
                             double* mom, float* stats, void* stream);
int unet_first_bn_act(int dtype, int relu, const float* x, const float* w, const float* bias,
                      const float* scale, const float* shift, void* y, int ldy, int yoff,
                      int N, int H, int W, int Cout, void* stream);
int unet_first_bn_relu_bwd(int dtype, int relu, const void* dy, int ldd, int doff,
                           const float* x, const float* w, const float* bias,
                           const float* scale, const float* shift, const float* mean, const float* invstd,
                           float* dgamma, float* dbeta, float* dbias, float* dw,
                           float* c1c2_ws, double* acc_ws, const double* moments, int N, int H, int W, int Cout, void* stream);
/* Last layer (models/unet.py:14-15 of the last DoubleConv + final_conv :39,62), n_classes <= 4.  Forward:
 * logits = W act(scale*z+shift) + b in one pass (y optional, may be NULL).  Backward: unet_bn_relu_bwd with its incoming
 * gradient recomputed per element as W^T dlogits, and final_conv's dw (K,C) / db (K) accumulated from the recomputed
 * activation.  acc_ws: 3*C + K*C + K doubles. */
int unet_affine_act_out(int dtype, int relu, const void* z, int ldz, int zoff, const float* scale, const float* shift,
                        const float* w, const float* bias, float* logits, void* y, int ldy, int yoff,
                        long long M, long long HW, int C, int K, void* stream);
int unet_bn_relu_bwd_out(int dtype, int relu, const float* dlogits, const float* w, int K, long long HW,
                         const void* z, int ldz, int zoff, void* dz, int ldo, int ooff,
                         const float* scale, const float* shift, const float* mean, const float* invstd,
                         float* dgamma, float* dbeta, float* dbias, float* dw, float* db,
                         float* c1c2_ws, double* acc_ws, long long M, int C, void* stream);

/* fp32 master weights -> packed compute-dtype images.  mode 0 conv fwd, 1 conv dgrad (flipped,
 * transposed), 2 convT fwd, 3 convT dgrad.  O,I = dims 0,1 of the source tensor. */
int unet_pack_weights(int dtype, int mode, const float* src, void* dst, int O, int I, int ntaps, void* stream);
/* The same for n tensors in one launch per 40 tensors (HOST arrays; they travel as kernel arguments).  Conv images
 * (modes 0/1) go through an LDS tile that reads each source row segment once; entries that share `src` (the forward
 * and dgrad image of one weight) are served by the same pass. */
int unet_pack_weights_multi(int dtype, int nt, const int* modes, const void* const* src, void* const* dst,
                            const int* O, const int* I, const int* ntaps, void* stream);

/* Inference: fold BatchNorm2d running statistics into the preceding conv of DoubleConv (models/unet.py:9-16):
 * w_out = w * s, b_out = beta + (bias - mean) * s, s = gamma / sqrt(var + eps) -- mergeBN's algebra
 * (models/reparm_unet.py:28-31) for the plain UNet.  O = Cout, per = Cin*kh*kw.  bias may be NULL. */
int unet_fold_bn_conv(const float* w, const float* bias, const float* gamma, const float* beta,
                      const float* running_mean, const float* running_var, float eps, int O, int per,
                      float* w_out, float* b_out, void* stream);

/* ---- optimizer tail of step_batch ------------------------------------------------------------
 * utils/train_step.py:56-63 (scaler.unscale_ -> clip_grad_norm_ -> optimizer.step) with the AdamW of
 * utils/builders.py:143-146, as multi-tensor kernels over fp32 tensors.  The tensor lists are HOST
 * arrays of device pointers.  unet_grad_sumsq writes one fp64 partial per 16384-element chunk
 * (unet_mt_blocks of them); unet_grad_norm_finalize sums them in a fixed order and writes
 * norm_out[0] = ||grad_scale*g||_2 and norm_out[1] = grad_scale * min(1, max_norm / (norm + 1e-6)) (clip_grad_norm_'s
 * coefficient) on the device; grad_scale = 1 normally, 1/world when the tensors hold the all-reduced SUM of the ranks'
 * gradients (train.py:143 DDP averages them): the averaging then costs no pass of its own.  unet_adamw_step multiplies the gradients by norm_out[1] (NULL: no
 * clipping; write_grads: store the clipped gradients back as clip_grad_norm_ does) and applies
 * torch.optim.AdamW's update for step number `step` (>= 1) to every tensor.  veto (may be NULL) points at a device float:
 * if it is > 0 when the kernel runs, nothing is touched.  step_batch passes the loss kernel's illegal-label counter, so a
 * batch on which the reference raises inside loss_fn (F.one_hot, losses/segmentation.py:9; before backward and the
 * optimizer, utils/train_step.py:52-63) does not change weights or optimizer state here either, with no extra host sync. */
int unet_mt_blocks(int nt, const long long* numel);
int unet_grad_sumsq(int nt, const void* const* grads, const long long* numel, double* partials, void* stream);
int unet_grad_norm_finalize(const double* partials, int nblocks, float max_norm, float grad_scale, float* norm_out, void* stream);
int unet_adamw_step(int nt, void* const* params, void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                    const long long* numel, const float* norm_out, int write_grads, double lr, double beta1,
                    double beta2, double eps, double weight_decay, long long step, const float* veto, void* stream);
int unet_cast(int dtype, int to_f32, const void* src, void* dst, long long n, void* stream);

/* ---- BatchNorm2d (+ReLU) -- models/unet.py:11-12,14-15 --------------------------------------
 * train: batch mean / biased variance normalise, running stats use unbiased variance, momentum 0.1 */
int unet_bn_finalize(const float* stat_rows, int nrows, int C, long long count,
                     const float* gamma, const float* beta,
                     float* running_mean, float* running_var, long long* num_batches_tracked,
                     float momentum, float eps,
                     float* scale, float* shift, float* mean, float* invstd,
                     double* acc_ws, void* stream);
/* eval mode (running statistics): scale = gamma/sqrt(rv+eps), shift = beta - rm*scale; mean / invstd (nullable) receive
 * rm and 1/sqrt(rv+eps) for a backward pass through the frozen statistics (relu bit 1 of unet_bn_relu_bwd). */
int unet_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* running_mean,
                        const float* running_var, float eps, float* scale, float* shift,
                        float* mean, float* invstd, void* stream);
/* y = act(z*scale + shift) */
int unet_affine_act(int dtype, int relu, const void* z, int ldz, int zoff, void* y, int ldy, int yoff,
                    const float* scale, const float* shift, long long M, int C, void* stream);
/* affine_act and MaxPool2d(2,2) of its result in one pass (models/unet.py:45-47): y, pooled [N,H/2,W/2,C] and
 * code are bit-identical to unet_affine_act followed by unet_maxpool2x2_fwd; y is not read back. */
int unet_affine_act_pool(int dtype, int relu, const void* z, int ldz, int zoff, void* y, int ldy, int yoff,
                         const float* scale, const float* shift, void* pooled, uint8_t* code,
                         int N, int H, int W, int C, void* stream);
/* dz, dgamma, dbeta (and the conv-bias grad = sum dz) from dy and the saved pre-norm z.
 * relu: bit 0 = ReLU after the norm; bit 1 = frozen statistics (model.eval() with gradients enabled: mean / invstd are
 * the running statistics and carry no gradient): dz = scale*dy', dgamma = sum dy'*(z-mean)*invstd, dbeta = sum dy'.
 * bit 2 = acc_ws is already zero (a caller that clears one arena for many calls; otherwise the call clears it itself).
 * acc_ws: 3*C + 2 doubles (the sums and the second pass's completion counter: its last workgroup writes dbias, there is no
 * conversion launch).  c1c2_ws is no longer used (the finalize between the two passes now runs inside the second) and may
 * be NULL; the argument stays so that existing callers keep working. */
int unet_bn_relu_bwd(int dtype, int relu, const void* dy, int ldd, int doff, const void* z, int ldz, int zoff,
                     void* dz, int ldo, int ooff,
                     const float* scale, const float* shift, const float* mean, const float* invstd,
                     float* dgamma, float* dbeta, float* dbias,
                     float* c1c2_ws, double* acc_ws, long long M, int C, void* stream);
/* The first pass of unet_bn_relu_bwd inside the kernel that PRODUCES dy: the 3x3 dgrad of a DoubleConv's second conv
 * (models/unet.py:13 under autograd; w_dgrad = unet_pack_weights mode 1) writes dy and, per 16x32 pixel tile, the rows
 * (sum dy', sum dy'*xhat) of the BatchNorm + ReLU in front of it (models/unet.py:11-12; z = its saved pre-norm tensor).
 * unet_bn_relu_bwd_rows then adds the rows up and runs the second pass only.  bf16 and whole tiles / 64-channel blocks
 * (unet_conv_dgrad_bnred_ok); sum_rows: N*(H/16)*(W/32) rows of [Cout][2] floats. */
int unet_conv_dgrad_bnred_ok(int dtype, int H, int W, int Cin, int Cout);
int unet_conv_dgrad_bnred(int dtype, const void* dz, int ldx, int xoff, const void* w_dgrad, void* dy, int ldy, int yoff,
                          float* sum_rows, const void* z, int ldz, int zoff,
                          const float* scale, const float* shift, const float* mean, const float* invstd,
                          int N, int H, int W, int Cin, int Cout, void* stream);
int unet_bn_relu_bwd_rows(int dtype, int relu, const void* dy, int ldd, int doff, const void* z, int ldz, int zoff,
                          void* dz, int ldo, int ooff,
                          const float* scale, const float* shift, const float* mean, const float* invstd,
                          float* dgamma, float* dbeta, float* dbias,
                          const float* sum_rows, int nrows, double* acc_ws, long long M, int C, void* stream);
/* unet_bn_relu_bwd without its two workspace arguments.  NO COUNTERPART IN THE REFERENCE: it exists for callers that
 * cannot allocate device memory -- code under HIP graph capture, tight serving / fine-tuning loops, harnesses that
 * forbid hipMalloc.  The 3*C + 2 fp64 accumulators and 2*C fp32 coefficients live in a 64 KiB device array inside the library
 * (one copy per GPU, held until the process ends), so C <= 2048 (larger C is rejected with an error; use unet_bn_relu_bwd).
 * The call allocates and frees nothing.  Every call on a GPU shares that GPU's array: calls that overlap on one GPU
 * (two streams, two threads) race -- serialise them on one stream.  The first call on a device fetches the array's
 * address (hipGetSymbolAddress), so make one warm-up call per device before capturing a graph.  Same results as
 * unet_bn_relu_bwd (same kernels). */
int unet_bn_relu_bwd_noalloc(int dtype, int relu, const void* dy, int ldd, int doff, const void* z, int ldz, int zoff,
                             void* dz, int ldo, int ooff,
                             const float* scale, const float* shift, const float* mean, const float* invstd,
                             float* dgamma, float* dbeta, float* dbias,
                             long long M, int C, void* stream);
/* The same with dy = dskip + unpool(dpool, code): the backward of MaxPool2d(2,2) (models/unet.py:47) and the
 * skip-connection add evaluated inside both passes, so the gradient of an encoder level's output is never stored.
 * dpool [N,H/2,W/2,C] dense, code from unet_maxpool2x2_fwd, dskip a channel slice (lds, soff) of [N,H,W,*]. */
int unet_bn_relu_bwd_pool(int dtype, int relu, const void* dpool, const uint8_t* code, const void* dskip, int lds, int soff,
                          const void* z, int ldz, int zoff, void* dz, int ldo, int ooff,
                          const float* scale, const float* shift, const float* mean, const float* invstd,
                          float* dgamma, float* dbeta, float* dbias, float* c1c2_ws, double* acc_ws,
                          int N, int H, int W, int C, void* stream);
int unet_relu_bwd(int dtype, const void* dy, int ldd, int doff, const void* y, int ldy, int yoff,
                  void* dz, int ldo, int ooff, long long M, int C, void* stream);
int unet_channel_sum(int dtype, const void* x, int ldx, int xoff, long long M, int C, float* out, int accumulate,
                     double* acc_ws, void* stream);
/* out[c] = sum over r of rows[r*ld + off + c*stride]: column sums of the fp32 partial rows a conv epilogue wrote.  The bias
 * gradient of nn.ConvTranspose2d (models/unet.py:35) is taken this way from the statistics rows of the split dgrad that
 * produced its output gradient.  No workspace, no atomics, fixed summation order. */
int unet_rows_colsum(const float* rows, int nrows, int ld, int off, int C, int stride, float* out, void* stream);

/* ---- MaxPool2d(2,2) with argmax -- models/unet.py:26,47 --------------------------------------
 * code[n,ho,wo,c] = 2*dh+dw of the winner (ATen tie rule: first in row-major order, NaN wins);
 * int64 index of torch = (2*ho+dh)*W + 2*wo+dw.  bwd adds the skip-connection gradient. */
int unet_maxpool2x2_fwd(int dtype, const void* x, int ldx, int xoff, void* y, uint8_t* code,
                        int N, int H, int W, int C, void* stream);
int unet_maxpool2x2_bwd(int dtype, const void* dy, const uint8_t* code, const void* dskip, int lds, int soff,
                        void* dx, int N, int H, int W, int C, void* stream);

/* ---- Dice + CE -- losses/segmentation.py:5-27 ------------------------------------------------
 * logits NCHW fp32, target int64.  acc_ws: N*C*3+3 doubles, shared by fwd and bwd.
 * out4 = {loss, ce, dice, count of labels the reference would have raised on}. */
int unet_dice_ce_fwd(const float* logits, const long long* target, int N, int C, long long HW,
                     float ce_weight, float dice_weight, long long ignore_index, int ce_only, double eps,
                     double* acc_ws, float* out4, void* stream);
int unet_dice_ce_bwd(const float* logits, const long long* target, int N, int C, long long HW,
                     float ce_weight, float dice_weight, long long ignore_index, int ce_only, double eps,
                     const double* acc_ws, const float* grad_out, float* dlogits, void* stream);
/* ---- device input pipeline --------------------------------------------------------------------
 * CT_NpyDataset.__getitem__ (data/ct_dataset.py:25-84) for a batch of N <= 64 raw slices of one size:
 * lung window (:32-33), bilinear / nearest safety resize to SxS (:40-47), hflip / vflip (:52-59),
 * rotation (:62-66, torchvision TF.rotate: affine grid + grid_sample, zeros outside), mask > 0 (:82).
 * image fp32 [N,H0,W0], mask uint8 [N,H0,W0] -> out_image fp32 [N,1,S,S], out_mask int64 [N,S,S].
 * flags[n] (HOST): 1 hflip | 2 vflip | 4 rotate; theta (HOST, N x 6 floats): the rescaled inverse affine
 * matrix of torchvision's _gen_affine_grid as [T00 T01 T10 T11 T20 T21], grid = [x y 1] * T.
 * tmp_image / tmp_mask (N*S*S each) are needed only when (H0,W0) != (S,S). */
int unet_ct_preprocess(const float* image, const uint8_t* mask, int N, int H0, int W0, int S,
                       const int* flags, const float* theta, float* tmp_image, uint8_t* tmp_mask,
                       float* out_image, long long* out_mask, void* stream);

/* logits.argmax(1) -- trainer.py:119, utils/eval_utils.py:103 */
int unet_argmax(const float* logits, long long* out, int N, int C, long long HW, void* stream);
/* entropy-weighted two-model fusion per pixel -- eval_ensemble.py:9-11,77-99 */
int unet_ensemble_fuse(const float* logits1, const float* logits2, long long* out, int N, int C, long long HW,
                       float entropy_thresh, void* stream);
/* confusion counts for calculate_segmentation_metrics -- utils/eval_utils.py:16-40; counts[66] */
int unet_seg_counts(const long long* pred, const long long* target, long long total, int K,
                    unsigned long long* counts, void* stream);

/* ---- RepUNet training path: GroupNorm + multi-branch sum -- models/reparm_unet.py:17-25,67-72,114-122
 * stat_rows: the conv epilogue's rows, grouped per image ([N][rows_per_image][C][2]). */
int unet_gn_finalize(const float* stat_rows, int N, int rows_per_image, int C, int groups, long long HW,
                     const float* gamma, const float* beta, float eps,
                     float* scale, float* shift, float* mean, float* invstd, void* stream);
/* y = act(sum_b z_b*scale_b + shift_b); z/scale/shift are HOST arrays of nb device pointers;
 * coef_nstride = C (per-sample coefficients, GroupNorm) or 0 (per-channel, BatchNorm variant) */
int unet_branch_sum_act(int dtype, int relu, int nb, const void* const* z, const float* const* scale, const float* const* shift,
                        int coef_nstride, void* y, int ldy, int yoff, int N, long long HW, int C, void* stream);
/* GroupNorm backward of one branch; u = ReLU-masked upstream gradient.  acc_ws: N*C*2+C*2 doubles,
 * coef_ws: 3*N*C floats. */
int unet_gn_bwd(int dtype, const void* u, const void* z, void* dz, int N, long long HW, int C, int groups,
                const float* gamma, const float* mean, const float* invstd,
                float* dgamma, float* dbeta, double* acc_ws, float* coef_ws, void* stream);
/* GroupNorm backward of the nb <= 8 branches of one MultiConvBlock (models/reparm_unet.py:114-122), which share u: u is
 * read once per pass instead of once per branch.  z/dz/gamma/mean/invstd/dgamma/dbeta are HOST arrays of nb device
 * pointers; acc_ws: nb*(N*C*2+C*2) doubles, coef_ws: nb*3*N*C floats.  Same arithmetic per branch as unet_gn_bwd. */
int unet_gn_bwd_multi(int dtype, int nb, const void* u, const void* const* z, void* const* dz, int N, long long HW, int C, int groups,
                      const float* const* gamma, const float* const* mean, const float* const* invstd,
                      float* const* dgamma, float* const* dbeta, double* acc_ws, float* coef_ws, void* stream);

/* ---- RepUNet fold (switch_to_deploy) -- models/reparm_unet.py:28-37,74-82,124-135; bit-exact fp32 */
int unet_fold_asymm(int k, int O, int I, const float* w_sq, const float* w_ve, const float* w_ho,
                    const float* sq_w, const float* sq_b, const float* sq_rm, const float* sq_rv,
                    const float* ve_w, const float* ve_b, const float* ve_rm, const float* ve_rv,
                    const float* ho_w, const float* ho_b, const float* ho_rm, const float* ho_rv,
                    float eps, float* out_w, float* out_b, void* stream);
/* unet_fold_asymm with std = sqrt(running_var + eps) supplied per branch (length O each).  The sqrt is the only step of
 * mergeBN (models/reparm_unet.py:29) that torch's CPU backend does not round correctly (MKL VML), so the IEEE fold above
 * differs from a torch-CPU fold in ~1 % of the elements by one ulp; with the reference host's own sqrt handed in, the
 * result equals the reference's mergedConv weights and biases bit for bit. */
int unet_fold_asymm_std(int k, int O, int I, const float* w_sq, const float* w_ve, const float* w_ho,
                        const float* sq_w, const float* sq_b, const float* sq_rm, const float* sq_rv, const float* sq_std,
                        const float* ve_w, const float* ve_b, const float* ve_rm, const float* ve_rv, const float* ve_std,
                        const float* ho_w, const float* ho_b, const float* ho_rm, const float* ho_rv, const float* ho_std,
                        float* out_w, float* out_b, void* stream);
int unet_fold_embed_add(int K, int k, int O, int I, float* target_w, const float* src_w, float* target_b,
                        const float* src_b, void* stream);

/* ---- bilinear resize -- models/unet.py:56-57 ------------------------------------------------------------------
 * F.interpolate(x, size=skip.shape[2:], mode="bilinear", align_corners=True): the branch UNet.forward takes when H or W
 * is not divisible by 2^levels and an upsampled tensor misses its skip connection by a pixel.  NHWC channel slices
 * (ptr, ld, off) as everywhere; x is [N,h,w,C], y is [N,H,W,C].  _bwd is the adjoint (dx = J^T dy), evaluated as a gather
 * (deterministic).  unet_maxpool2x2_fwd/_bwd accept odd H, W with nn.MaxPool2d's floor rule for the same case. */
int unet_resize_bilinear_fwd(int dtype, const void* x, int ldx, int xoff, int h, int w,
                             void* y, int ldy, int yoff, int H, int W, int N, int C, void* stream);
int unet_resize_bilinear_bwd(int dtype, const void* dy, int ldy, int yoff, int H, int W,
                             void* dx, int ldx, int xoff, int h, int w, int N, int C, void* stream);

/* ---- data parallel over RCCL -- train.py:33 (init_process_group "nccl"), :139-144 (DDP wrap) ---------------------
 * For hosts without torch.distributed (the Python mirror, unet_amd/parallel.py, uses torch's RCCL backend instead).
 * One process per GPU, that GPU current.  unet_comm_unique_id fills 128 bytes on ONE rank; the host hands them to every
 * rank (its own channel: file, socket, MPI); unet_comm_init joins the communicator and returns an opaque handle the
 * CALLER owns (unet_comm_destroy).  unet_allreduce_bucket sums `count` elements of dtype UNET_F32 / UNET_BF16 in place
 * across the ranks on `stream` (DDP's per-bucket ncclAllReduce; the 1/world of the average is folded into
 * unet_grad_norm_finalize's grad_scale, so there is no divide pass); unet_broadcast copies root's buffer to every rank
 * (parameters at wrap time, BatchNorm buffers before each training forward: DDP's broadcast_buffers=True).
 * RCCL is dlopen'ed on first use; the calls fail with a message if it is absent. */
int unet_comm_unique_id(void* id128);
int unet_comm_init(void** comm, int world, int rank, const void* id128);
int unet_comm_destroy(void* comm);
int unet_allreduce_bucket(void* comm, void* buf, long long count, int dtype, void* stream);
int unet_broadcast(void* comm, void* buf, long long count, int dtype, int root, void* stream);

/* Tile scheduling of the persistent bf16 conv / weight-gradient kernels.  No counterpart in the reference (DDP there shares
 * the GPU between cuDNN's kernels and NCCL's through the hardware scheduler).  These kernels own every CU and by default
 * split their tiles statically, so a collective's workgroups that hold a few CUs while one is launched delay it by their
 * whole hold time.  level 1: launches with at least three rounds of tiles draw them from per-XCD atomic
 * queues instead (a late CU takes fewer tiles; a dry queue steals); which workgroup computes which tile is then timing
 * dependent, so split-K partial sums are grouped differently from run to run (last-bit differences in dW, none in
 * activations).  level 2: also launches with few tiles per workgroup (tests).  0: static (default; also the environment
 * variable UNET_AMD_DYNAMIC_TILES).  Process-wide; takes effect at the next launch. */
int unet_set_dynamic_tiles(int level);

#ifdef __cplusplus
}
#endif
#endif
