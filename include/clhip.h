/*
 * clhip.h -- C ABI of libclhip.so, the MI355X (gfx950) hot-path library of libcontinual_amd.
 *
 * The reference (RL-VIG/LibContinual) is 100 % Python on torch and has no FFI of its own
 * (SURVEY.md section 2.6): every FLOP of its hot path is a torch op called from
 *   core/trainer.py:585-612            (the batch loop)
 *   core/model/{finetune,ewc,lwf,icarl,lucir}.py  (observe / inference / Fisher)
 *   core/model/backbone/resnet.py      (ResNet forward)
 * The entry points below are what those call sites bind to in the drop-in (INTEGRATION.md shows
 * the ctypes stubs); each declaration cites the reference lines whose arithmetic it replaces.
 *
 * Conventions
 *   - plain C, no torch types; all pointers are DEVICE pointers borrowed for the call (the caller
 *     -- torch's caching allocator -- keeps ownership), except where marked "host".
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).
 *   - return 0 on success, a negative CLHIP_E* code otherwise; never throws.  clhip_last_error()
 *     gives a thread-local message.  Functions are re-entrant per stream; no global mutable state.
 *   - activations are NHWC in `dtype` (CLHIP_BF16 = bf16 storage / bf16 MFMA / fp32 accumulate,
 *     CLHIP_F32 = fp32 storage / fp32 MFMA: the exact-arithmetic parity mode); parameters,
 *     gradients, optimizer state, Fisher, BN statistics and losses are always fp32.
 *   - conv weights (fp32 master) are stored K,R,S,C ("channels_last" strides of a [K,C,R,S] tensor).
 */
#ifndef CLHIP_H
#define CLHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLHIP_OK 0
#define CLHIP_EINVAL (-1)   /* bad argument / unsupported shape */
#define CLHIP_EHIP (-2)     /* a HIP runtime call failed */
#define CLHIP_ENOMEM (-3)

#define CLHIP_BF16 0
#define CLHIP_F32 1

const char* clhip_last_error(void);
int clhip_version(void);

/* ------------------------------------------------------------------------------------------------
 * Layout / precision conversion.
 * x: fp32 NCHW [N,C,H,W] (the batch dict's "image", core/data/dataset.py:266)  ->  y: NHWC `dtype`
 * with channels zero-padded to Cpad (multiple of 8).                                               */
int clhip_nchw_to_nhwc(const float* x, void* y, int N, int C, int H, int W, int Cpad, int dtype, void* stream);
/* y: NHWC `dtype` [N,H,W,C] -> x: fp32 NCHW (for 'fmaps', backbone/resnet.py:392-395) */
int clhip_nhwc_to_nchw(const void* y, float* x, int N, int C, int H, int W, int dtype, void* stream);

/* fp32 master conv weight [K][taps][Creal] -> `dtype` shadows:
 *   w_fwd [K][taps][Cpad]   (forward / wgrad operand)
 *   w_dg  [Cpad][taps][K]   (dgrad operand), may be NULL                                           */
int clhip_conv_weight_prep(const float* w, void* w_fwd, void* w_dg, int K, int taps, int Creal, int Cpad,
                           int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Convolution as implicit GEMM on MFMA (replaces nn.Conv2d fwd / dgrad / wgrad, reference
 * backbone/resnet.py:17-24, 295-298, 337, 367).  ksize in {1,3}; C, K multiples of 8 (pad the
 * 3-channel stem to 8); H,W arbitrary.  Ho = (H + 2*pad - ksize)/stride + 1.
 *
 * fwd : z[N,Ho,Wo,K] = conv(x[N,H,W,C], w_fwd).  If stat_partials != NULL it receives per-tile
 *       partial sums for train-mode BatchNorm: float[tiles][2][K] (sum, sum of squares of the fp32
 *       accumulators), tiles = clhip_conv_fwd_tiles(); *not* atomics -> deterministic.
 * dgrad: dx[N,H,W,C] (+)= conv^T(dz[N,Ho,Wo,K], w_dg);  accumulate!=0 adds to the existing dx.
 * wgrad: dw[K][taps][Creal] += sum_pixels dz * x   (fp32).  `ws` (nullable) is scratch of
 *        clhip_conv_wgrad_ws_bytes() bytes: with it the 3x3/s1 kernel writes per-split partial blocks and reduces
 *        them in a fixed order (bitwise reproducible); without it partial sums are combined with fp32 atomics. */
int clhip_conv_fwd_tiles(int N, int H, int W, int C, int K, int ksize, int stride, int pad);
int clhip_conv_fwd(const void* x, const void* w_fwd, void* z, float* stat_partials, int N, int H, int W, int C, int K,
                   int ksize, int stride, int pad, int dtype, void* stream);
/* same forward, but the per-channel sums of z and z^2 are ADDED into stat_acc[replicas][2][K] (fp64, hardware atomics; zeroed
 * by the caller; a workgroup uses replica blockIdx & (replicas-1), replicas = power of two <= 64) instead of written as per-tile
 * partial rows: the consumer (clhip_bn_apply_train) sums the replicas itself and needs no finalize launch */
int clhip_conv_fwd_acc(const void* x, const void* w_fwd, void* z, double* stat_acc, int replicas, int N, int H, int W, int C, int K,
                       int ksize, int stride, int pad, int dtype, void* stream);
int clhip_conv_dgrad(const void* dz, const void* w_dg, void* dx, int accumulate, int N, int H, int W, int C, int K,
                     int ksize, int stride, int pad, int dtype, void* stream);
/* clhip_conv_dgrad whose epilogue also reduces the BatchNorm backward of the layer that PRODUCED the tensor x (the two per-channel
 * sums of autograd's batch-norm backward, resnet.py:296-316: sum g and sum g * xhat with g = dx * (y_prod > 0), xhat =
 * (z_prod - mean) * invstd), from the fp32 results before they are rounded: the separate reduction pass over dx and z_prod
 * (first launch of clhip_bn_bwd_acc) disappears.  Valid when this launch COMPLETES dx (it is the last writer / accumulator).
 * z_prod, y_prod: [N,H,W,C] pre- / post-activation outputs of the producing layer (y_prod NULL: no ReLU); acc[replicas][2][C] fp64,
 * zeroed by the caller.  3x3 / stride 1 / pad 1 bf16 layers the fourth-generation kernel covers:
 * clhip_conv_dgrad_bn_reduce_supported() != 0. */
int clhip_conv_dgrad_bn_reduce_supported(int N, int H, int W, int C, int K, int ksize, int stride, int pad, int dtype);
/* != 0: the layer runs on a kernel that hides this epilogue (conv8.hip: two workgroups per CU, one multiplies while the other reduces), so the
 * fused form pays on large maps as well -- the plan fuses such layers whatever their pixel count */
int clhip_conv_dgrad_bn_reduce_overlapped(int N, int H, int W, int C, int K, int ksize, int stride, int pad, int dtype);
int clhip_conv_dgrad_bn_reduce(const void* dz, const void* w_dg, void* dx, int accumulate, const void* z_prod, const void* y_prod /*nullable*/,
                               const float* mean, const float* invstd, double* acc, int replicas, int N, int H, int W, int C, int K,
                               int ksize, int stride, int pad, int dtype, void* stream);
/* ... with cheaper sources for the producer's ReLU mask (both nullable; kernels that cannot use them fall back to y_prod): mask_prod = its packed mask
 * [N*H*W][C/8] (bit e of byte c/8 = element c%8 > 0, what clhip_bn_apply_train_mask writes), gamma_prod / beta_prod = its weight / bias for a ReLU straight
 * behind the BatchNorm (mask = scale z + shift > 0, scale = gamma * invstd, shift = beta - mean * scale as in the forward) */
int clhip_conv_dgrad_bn_reduce_ex(const void* dz, const void* w_dg, void* dx, int accumulate, const void* z_prod, const void* y_prod /*nullable*/,
                                  const void* mask_prod /*nullable*/, const float* gamma_prod /*nullable*/, const float* beta_prod, const float* mean, const float* invstd, double* acc,
                                  int replicas, int N, int H, int W, int C, int K, int ksize, int stride, int pad, int dtype, void* stream);
size_t clhip_conv_wgrad_ws_bytes(int N, int H, int W, int C, int Creal, int K, int ksize, int stride, int pad, int dtype);
int clhip_conv_wgrad(const void* x, const void* dz, float* dw, void* ws, int N, int H, int W, int C, int Creal, int K, int ksize,
                     int stride, int pad, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * BatchNorm2d (+ residual add + ReLU), reference backbone/resnet.py:296-316 (nn.BatchNorm2d,
 * momentum 0.1, eps 1e-5; train = batch statistics + running-stat update with unbiased variance).
 *
 * bn_stats_finalize: reduce the conv's stat partials -> mean/invstd (saved for backward) and the
 *   fused affine scale = gamma*invstd, shift = beta - mean*scale; updates running_mean/var.
 * bn_eval_affine: scale/shift from the running statistics (eval mode).
 * bn_apply: y = [relu]( z*scale[c] + shift[c] [+ res] ).
 * bn_bwd: given dy (grad wrt y), y (ReLU mask), z:  dgamma,dbeta (+= into fp32 grads), dz, and the
 *   residual-branch gradient dres (= masked dy; written or accumulated).  `ws` = float scratch of
 *   clhip_bn_bwd_ws_floats(M, C) floats.                                                            */
int clhip_bn_stats_finalize(const float* stat_partials, int tiles, int64_t M, int C, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float momentum, float eps, float* mean,
                            float* invstd, float* scale, float* shift, void* stream);
int clhip_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                         float eps, int C, float* scale, float* shift, void* stream);
int clhip_bn_apply(const void* z, const float* scale, const float* shift, const void* res, void* y, int64_t M, int C,
                   int relu, int dtype, void* stream);
/* eval-mode BatchNorm (running statistics) [+ residual] [+ ReLU] in ONE launch: clhip_bn_eval_affine + clhip_bn_apply with the per-channel
 * scale / shift derived per workgroup (same expressions, same values).  nn.BatchNorm2d.eval() of the reference backbones (resnet.py:296-316). */
int clhip_bn_apply_eval(const void* z, const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                        const void* res /*nullable*/, void* y, int64_t M, int C, int relu, int dtype, void* stream);
/* training-mode BatchNorm straight from the fp64 sums of clhip_conv_fwd_acc: y = relu?(bn(z) + res?), batch mean / invstd saved
 * for the backward, running statistics updated (momentum, unbiased variance) -- bn_stats_finalize + bn_apply in ONE launch.
 * C must be a power of two. */
int clhip_bn_apply_train(const void* z, const double* stat_acc, int replicas, int64_t M, int C, const float* gamma, const float* beta, float* rm,
                         float* rv, float momentum, float eps, float* mean, float* invstd, const void* res /*nullable*/, void* y,
                         int relu, int dtype, void* stream);
/* clhip_bn_apply_train with ReLU that also writes the packed ReLU mask the backward can read instead of y: relu_mask[M * C / 8] bytes,
 * bit j of byte i = (stored y[8 i + j] > 0).  The mask of `out = F.relu(out + residual)` (resnet.py:316) costs the backward 1/16 of the
 * bytes of the activation. */
int clhip_bn_apply_train_mask(const void* z, const double* stat_acc, int replicas, int64_t M, int C, const float* gamma, const float* beta, float* rm,
                              float* rv, float momentum, float eps, float* mean, float* invstd, const void* res /*nullable*/, void* y,
                              void* relu_mask, int dtype, void* stream);
/* clhip_bn_bwd with the two per-channel sums accumulated into acc[replicas][2][C] (fp64 atomics, zeroed by the caller) and
 * consumed directly by the apply pass: no partial buffer, no finalize launch.  relu: 0 none, 1 mask = (y > 0), 3: `y` points to the
 * packed mask of clhip_bn_apply_train_mask. */
int clhip_bn_bwd_acc(const void* dy, const void* y, const void* z, const float* mean, const float* invstd, const float* gamma,
                     float* dgamma, float* dbeta, void* dz, void* dres, int dres_accumulate, int64_t M, int C, int relu, double* acc,
                     int replicas, int dtype, void* stream);
/* the same for a unit whose ReLU follows the BatchNorm directly (no residual): the ReLU mask is recomputed from z with the forward's
 * own scale / shift expressions (gamma * invstd, beta - mean * scale), so the activation tensor is not read                          */
int clhip_bn_bwd_acc_zmask(const void* dy, const void* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                           float* dgamma, float* dbeta, void* dz, int64_t M, int C, double* acc, int replicas, int dtype, void* stream);
/* The apply half alone: acc[replicas][2][C] already holds sum g and sum g * xhat -- accumulated by the launch that produced dy
 * (clhip_conv_dgrad_bn_reduce).  relu: 0 none, 1 mask = (y > 0), 2 mask recomputed from z, gamma, beta (then beta is required and
 * dres must be NULL), 3: `y` points to the packed mask of clhip_bn_apply_train_mask.  Same arithmetic as the second launch of clhip_bn_bwd_acc (autograd of nn.BatchNorm2d + ReLU + residual add,
 * core/model/backbone/resnet.py:296-316). */
int clhip_bn_bwd_apply_acc(const void* dy, const void* y /*nullable*/, const void* z, const float* mean, const float* invstd, const float* gamma,
                           const float* beta /*nullable*/, float* dgamma, float* dbeta, void* dz, void* dres /*nullable*/, int dres_accumulate,
                           int64_t M, int C, int relu, const double* acc, int replicas, int dtype, void* stream);
/* number of reduction workgroups clhip_bn_bwd / clhip_bn_bwd_acc launch (= producers adding into the accumulator) */
int clhip_bn_bwd_blocks(int64_t M, int C);
size_t clhip_bn_bwd_ws_floats(int64_t M, int C);
int clhip_bn_bwd(const void* dy, const void* y, const void* z, const float* mean, const float* invstd, const float* gamma,
                 float* dgamma, float* dbeta, void* dz, void* dres, int dres_accumulate, int64_t M, int C, int relu,
                 float* ws, int dtype, void* stream);

/* global average pool: feat[N,C] (fp32) = mean_hw a[N,HW,C]; backward broadcasts dfeat/HW           */
int clhip_avgpool_fwd(const void* a, float* feat, int N, int HW, int C, int dtype, void* stream);
int clhip_avgpool_bwd(const float* dfeat, void* da, int N, int HW, int C, int dtype, void* stream);
/* clhip_avgpool_bwd that also reduces the BatchNorm backward of the layer that PRODUCED the pooled activation (the backbone's last unit; the pooling
 * is that activation's only reader): z_prod, y_prod (nullable: no ReLU), mean, invstd, acc [replicas][2][C] fp64 as clhip_conv_dgrad_bn_reduce.
 * C a power of two in [8, 2048].  _supported: 1 / 0. */
int clhip_avgpool_bwd_bn_reduce_supported(int N, int HW, int C, int dtype);
int clhip_avgpool_bwd_bn_reduce(const float* dfeat, void* da, const void* z_prod, const void* y_prod /*nullable*/, const float* mean, const float* invstd,
                                double* acc, int replicas, int N, int HW, int C, int dtype, void* stream);

/* AvgPool2d(win) + NCHW flatten: feat[n][(c*Ph + ph)*Pw + pw] (fp32), Ph = H/win, Pw = W/win (resnet.py:643, 675-676) */
int clhip_avgpool_win_fwd(const void* a, float* feat, int N, int H, int W, int C, int win, int dtype, void* stream);
int clhip_avgpool_win_bwd(const float* dfeat, void* da, int N, int H, int W, int C, int win, int dtype, void* stream);

/* z[M,C] += r[M,C] in place and, when stat_acc != NULL, the per-channel sum / sum of squares of the result added into the fp64
 * accumulator [replicas][2][C] that clhip_bn_apply_train reads: `out += residual` of BasicBlock2 (resnet.py:615) followed by the
 * next block's bn1 (:602).  clhip_add_stats_blocks = number of producer workgroups (for sizing `replicas`).                         */
int clhip_add_stats_blocks(int64_t M, int C);
int clhip_add_stats(void* z, const void* r, double* stat_acc, int replicas, int64_t M, int C, int dtype, void* stream);
/* a[n] += b[n] (n % 8 == 0): joins the two gradient paths of a raw sum (autograd's accumulation at `out += residual`) */
int clhip_add_inplace(void* a, const void* b, int64_t n, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Whole-backbone plan: a static list of (conv -> BN -> +res -> ReLU) units + global avg-pool, run
 * with ONE call per direction (replaces the per-op dispatch of CifarResNet.forward / ResNet._forward_impl
 * / modified_ResNet.forward, backbone/resnet.py:381-395, 215-223, 549-560, and autograd's backward).  */
/* bits of clhip_unit_desc.relu.  The last three re-associate pre-activation blocks (ResNet_BIC / BasicBlock2, backbone/resnet.py:589-680:
 * BN -> ReLU -> conv, the shortcut added to the RAW conv output, BN-free 1x1 shortcut convs) onto the same unit list: a unit is then
 * conv -> (+ raw sum of unit res-1) -> BatchNorm of the FOLLOWING block -> ReLU.                                                      */
enum {
    CLHIP_UNIT_RELU = 1,     /* ReLU after the BatchNorm (+ residual)                                                                 */
    CLHIP_UNIT_PRE_RES = 2,  /* `res` names a unit whose raw (pre-BatchNorm) sum is added to this conv's output BEFORE the BatchNorm  */
    CLHIP_UNIT_RAW_SRC = 4,  /* the conv reads the raw sum of activation `src` instead of its normalised output                       */
    CLHIP_UNIT_NO_BN = 8     /* conv only (gamma/beta/rm/rv offsets ignored); the result is consumed through PRE_RES / RAW_SRC         */
};

typedef struct {
    int32_t cin, cout, ksize, stride, pad;
    int32_t src;      /* activation index consumed: 0 = network input, i+1 = output of unit i */
    int32_t res;      /* activation index added before the ReLU, or -1 */
    int32_t relu;     /* CLHIP_UNIT_* bits (1 = plain ReLU, 0 = none: the values older callers pass) */
    int64_t w_off;    /* element offsets into the flat fp32 parameter / gradient buffers */
    int64_t gamma_off;
    int64_t beta_off;
    int64_t rm_off;   /* element offsets into the flat fp32 BN-statistics buffer */
    int64_t rv_off;
} clhip_unit_desc;

typedef struct clhip_plan clhip_plan;

clhip_plan* clhip_plan_create(const clhip_unit_desc* units /*host*/, int n_units, int N, int H, int W, int Cin, int dtype);
/* pool_win > 0: nn.AvgPool2d(pool_win) + flatten of the NCHW result instead of the global pool (ResNet_BIC.forward, resnet.py:675-676;
 * feat_dim = C * (H/win) * (W/win)) */
clhip_plan* clhip_plan_create_ex(const clhip_unit_desc* units /*host*/, int n_units, int N, int H, int W, int Cin, int dtype, int pool_win);
void clhip_plan_destroy(clhip_plan*);
size_t clhip_plan_workspace_bytes(const clhip_plan*);      /* activations + saved tensors + grads + scratch */
size_t clhip_plan_shadow_bytes(const clhip_plan*);         /* `dtype` copies of the conv weights */
int clhip_plan_feat_dim(const clhip_plan*);
/* (round 6) Runs of BasicBlocks of the CIFAR ResNet-32s (core/model/backbone/resnet.py:289-316, 381-392) execute as ONE launch per direction in training mode
 * when the batch fits the device one image per compute unit (bf16 plans, N <= number of CUs; clhip_config("STAGE_TRAIN", "0") keeps one launch per unit).
 * Those launches need every workgroup resident at once, so: (1) a plan's training passes must not run on two streams concurrently, nor beside the training
 * passes of another plan of the same device on another stream (the library serialises stream switches outside a capture; inside a capture the caller must);
 * (2) several PROCESSES sharing one GPU must switch STAGE_TRAIN off.  Every in-launch wait is bounded: a violation ends in wrong results plus a sticky error
 * word, not a hung device.  clhip_plan_stage_status returns that word (0 = clean; synchronises the device -- call it at epoch / task boundaries).
 * The batch statistics of grids of 64+ workgroups are exchanged XCD-first (xch.h: plain stores inside an XCD's L2, the XCDs' partial sums across): that form is used
 * only on a device where a probe at plan creation found workgroup w on the XCD of workgroup w % 8 for every grid size it tries; clhip_config("STAGE_XCH3", "0")
 * keeps the two-hop form (another fixed summation order: the two forms agree to fp64 rounding, not bit for bit). */
int clhip_plan_stage_status(clhip_plan*);
/* what: 0 = units of the plan that run inside stage-level training launches, 1 / 2 = such forward / backward launches made so far (tests, diagnostics) */
long long clhip_plan_stage_info(const clhip_plan*, int what);
/* diagnostic (clhip_config("STAGE_TRACE", "<channels>:<convolution>")): 24 phase stamps (s_memtime ticks: shader cycles on this part) of workgroup 0's last traced forward [0..7] / backward [8..23] unit */
int clhip_plan_stage_trace(clhip_plan*, unsigned long long* out24);
/* refresh the `dtype` weight shadows from the fp32 masters (call after every optimizer step) */
int clhip_plan_prep_weights(clhip_plan*, const float* params, void* shadow, void* stream);
/* x: fp32 NCHW input; feat: fp32 [N, feat_dim].  training!=0: batch statistics, running stats updated,
 * activations saved in `workspace` for clhip_plan_backward.  training == 2: the same forward with the promise that NO backward (and no clhip_plan_read_act)
 * follows -- a frozen teacher that runs on batch statistics: the stage-level launches then skip the stores of z and of the activations inside a run
 * (same features, same statistics); launches that do not know the hint ignore it.                      */
int clhip_plan_forward(clhip_plan*, const float* x, const float* params, float* bn_stats, const void* shadow,
                       void* workspace, float* feat, int training, void* stream);
/* the same with nn.BatchNorm2d's `num_batches_tracked += 1` (torch/nn/modules/batchnorm.py, one int64 counter per unit, slot i = unit i;
 * nullable) folded into the forward's first launch when training != 0 -- no separate host-side add per step */
int clhip_plan_forward_ex(clhip_plan*, const float* x, const float* params, float* bn_stats, const void* shadow,
                          void* workspace, float* feat, int training, int64_t* num_batches_tracked, void* stream);
/* dfeat: fp32 [N, feat_dim]; grads: flat fp32 buffer laid out like params, accumulated into (+=).      */
int clhip_plan_backward(clhip_plan*, const float* dfeat, const float* params, const void* shadow, void* workspace,
                        float* grads, void* stream);
/* the same backward in pieces: units [unit_lo, unit_hi) in reverse order (unit_hi == number of units starts from dfeat).  Lets
 * the data-parallel host start the all-reduce of the deepest layers' gradients (the tail of the flat gradient buffer: the
 * parameters are laid out in unit order) while the backward of the shallower layers is still running. */
int clhip_plan_num_units(const clhip_plan* p);
int clhip_plan_backward_range(clhip_plan* p, const float* dfeat, const float* params, const void* shadow, void* workspace,
                              float* grads, int unit_hi, int unit_lo, void* stream);
/* debugging / tests: copy activation `idx` (0=input) as fp32 NCHW; which: 0 = y, 1 = pre-BN z (idx>=1), 2 = dy.
 * Not every intermediate is written by the fast paths.  An activation whose BatchNorm the consumer applies on its operand load (training: the first convolution
 * of a block, also inside a stage-level run of stage_train.hip; eval: EVAL_LAZY) is REBUILT here from its z and coefficients by the apply launch the eager path
 * would have made -- which is why `workspace` is written to despite the const, and why the call must not be made between a forward and its backward on another
 * stream.  An activation INSIDE a run the eval forward executed as one launch (stage.hip) was never produced: CLHIP_EINVAL (clhip_config("STAGE_EVAL", "0") keeps
 * one launch per unit).  dy (which = 2) exists for the per-unit backward only: inside a stage-level training run the gradient of an activation never leaves LDS
 * (clhip_config("STAGE_TRAIN", "0") before the forward keeps the per-unit backward). */
int clhip_plan_read_act(clhip_plan*, const void* workspace, int idx, int which, float* out_nchw, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Heads and losses (fp32).
 * linear: out[B,O] = x[B,D] W[O,D]^T + b   (nn.Linear heads: ewc.py:50, lwf.py:29-40, icarl.py:31)      */
int clhip_linear_fwd(const float* x, const float* w, const float* b /*nullable*/, float* out, int B, int D, int O, void* stream);
int clhip_linear_bwd(const float* x, const float* w, const float* dout, float* dx /*nullable*/, float* dw, float* db /*nullable*/,
                     int B, int D, int O, int accumulate, void* stream);
/* cross entropy over the column slice [lo,hi) with labels offset by lo (F.cross_entropy(logit[:, old:], y-old),
 * ewc.py:99, lwf.py:62, icarl.py:209) fused with argmax over [0,pred_hi) and the correct-count
 * (ewc.py:102-108).  loss_out[0] (+)= weight*mean CE ; dlogits[B,O] (+)= weight*dCE (zero outside the slice);
 * pred[B] int64 ; correct[0] int32.                                                                   */
int clhip_ce_slice(const float* logits, const int64_t* labels, int B, int O, int lo, int hi, int pred_hi, float weight,
                   float* loss_out, int loss_accumulate, float* dlogits /*nullable*/, int grad_accumulate,
                   int64_t* pred /*nullable*/, int32_t* correct /*nullable*/, void* stream);
/* same with the argmax restricted to [pred_lo, pred_hi): L2P's logits masked to -inf outside the current task's classes
 * (l2p.py:92-106) are CE over [lo,hi) and argmax over [lo,hi) */
int clhip_ce_window(const float* logits, const int64_t* labels, int B, int O, int lo, int hi, int pred_lo, int pred_hi, float weight,
                    float* loss_out, int loss_accumulate, float* dlogits /*nullable*/, int grad_accumulate,
                    int64_t* pred /*nullable*/, int32_t* correct /*nullable*/, void* stream);
/* distillation: -(softmax(soft/T) * log_softmax(pred/T)).sum()/B over the first k columns
 * (lwf.py:75-78, icarl.py:198-206); strides are the row pitches (O) of the two logit matrices.         */
int clhip_kd_loss(const float* pred, int pred_stride, const float* soft, int soft_stride, int B, int k, float T,
                  float weight, float* loss_out, int loss_accumulate, float* dpred /*nullable*/, int grad_accumulate,
                  void* stream);

/* LUCIR heads/losses (backbone/resnet.py:418-463, lucir.py:175-210) */
int clhip_cosine_linear_fwd(const float* x, const float* w, float* out, float* xnorm, float* wnorm, int B, int D, int O, void* stream);
int clhip_cosine_linear_bwd(const float* x, const float* w, const float* out, const float* xnorm, const float* wnorm,
                            const float* dout, float* dx, float* dw, int B, int D, int O, int accumulate, void* stream);
/* mean(1 - cos(a,b)) * weight ; da (+)= grad ; b is the detached teacher feature */
int clhip_cos_embed_loss(const float* a, const float* b, int B, int D, float weight, float* loss_out, int loss_accumulate,
                         float* da, int grad_accumulate, void* stream);
/* margin ranking on old-class samples: for rows with label < num_old, over the top-K novel scores:
 * mean(max(0, margin - (gt - novel))) * weight; scores[B,O] are the pre-sigma cosine scores.            */
int clhip_margin_rank_loss(const float* scores, const int64_t* labels, int B, int O, int num_old, int K, float margin,
                           float weight, float* loss_out, int loss_accumulate, float* dscores, int grad_accumulate,
                           int32_t* hard_count, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Flat-buffer (multi-tensor) elementwise family, fp32.
 * ewc_penalty: loss_out (+)= weight * sum_i F_i (p_i - ref_i)^2 / 2       (ewc.py:221-225)
 * ewc_grad   : g_i += weight * F_i (p_i - ref_i)                          (its autograd gradient)
 * fisher_accum: fisher_i += g_i^2 * scale                                 (ewc.py:171-174)
 * fisher_merge: new_i = alpha*old_i + (1-alpha)*new_i                     (ewc.py:128-131)
 * sgd_step : torch.optim.SGD semantics (trainer.py:159-166): d = g*grad_scale + wd*p [+ ewc term];
 *            m = momentum*m + d; p -= lr*m.  `mom` may be NULL when momentum == 0.
 * adam_step: torch.optim.Adam (no amsgrad); step = 1-based step count.
 * sq_norm  : out[0] (+)= sum g_i^2   (clip_grad_norm_, l2p.py:104); scale: g *= s                       */
int clhip_ewc_penalty(const float* p, const float* ref, const float* fisher, int64_t n, float weight, float* loss_out,
                      int loss_accumulate, void* stream);
/* dev_scale (nullable): device scalar multiplied into `weight` (the upstream autograd gradient, no host sync) */
int clhip_ewc_grad(const float* p, const float* ref, const float* fisher, float* g, int64_t n, float weight,
                   const float* dev_scale, void* stream);
/* the same over up to 4 (p, ref, fisher[, g]) segments in ONE launch (host arrays of device pointers / lengths): EWC's parameters are the backbone's
 * flat buffer plus the head's weight and bias prefixes (ewc.py:207-225 loops over named_parameters) */
int clhip_ewc_penalty_multi(int count, const float* const* p, const float* const* ref, const float* const* fisher, const int64_t* n, float weight,
                            float* loss_out, int loss_accumulate, void* stream);
int clhip_ewc_grad_multi(int count, const float* const* p, const float* const* ref, const float* const* fisher, float* const* g, const int64_t* n,
                         float weight, const float* dev_scale, void* stream);
int clhip_fisher_accum(float* fisher, const float* g, int64_t n, float scale, void* stream);
int clhip_fisher_merge(float* new_f, const float* old_f, int64_t n, float alpha, void* stream);
int clhip_sgd_step(float* p, const float* g, float* mom, int64_t n, float lr, float momentum, float weight_decay,
                   float grad_scale, const float* ewc_ref, const float* ewc_fisher, float ewc_weight, void* stream);
/* the same update (without the EWC term) over `count` <= 8 tensors in ONE launch: p[k], g[k], mom[k] (mom NULL when momentum == 0), n[k] elements each.
 * Element for element the arithmetic of clhip_sgd_step. */
int clhip_sgd_step_multi(int count, float* const* p, const float* const* g, float* const* mom /*nullable*/, const int64_t* n, float lr, float momentum,
                         float weight_decay, float grad_scale, void* stream);
/* ... and, for the tensors whose bit is set in zero_grad_mask, g <- 0 once it has been consumed: the flat gradient buffer of a backbone then
 * enters the next backward already zeroed (every writer of it accumulates) and the per-step 44.7-MB fill launch of ResNet-18 disappears
 * (trainer.train_steps opts in: between optimizer.step() and the next zero_grad() nothing reads the gradients there). */
int clhip_sgd_step_multi_zero(int count, float* const* p, float* const* g, float* const* mom /*nullable*/, const int64_t* n, float lr, float momentum,
                              float weight_decay, float grad_scale, unsigned zero_grad_mask, void* stream);
int clhip_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                    float weight_decay, float grad_scale, int step, void* stream);
int clhip_sq_norm(const float* g, int64_t n, float* out, int accumulate, void* stream);
int clhip_scale(float* g, int64_t n, float s, void* stream);
/* out_i = g_i * s * (*dev_scale)  (out may alias g); chains a precomputed loss gradient with the upstream grad */
int clhip_scale_dev(const float* g, float* out, int64_t n, float s, const float* dev_scale, void* stream);
/* LUCIR sigma: logits = sigma*scores ; backward: dscores = sigma*dlogits, dsigma (+)= sum dlogits*scores
 * (backbone/resnet.py:439-441, 459-463) */
int clhip_sigma_scale_fwd(const float* scores, const float* sigma_dev, float* logits, int64_t n, void* stream);
int clhip_sigma_scale_bwd(const float* scores, const float* sigma_dev, const float* dlogits, float* dscores, float* dsigma,
                          int dsigma_accumulate, int64_t n, void* stream);

/* iCaRL nearest-class-mean: dist[B,M] = sum_d (f[b,d]-means[m,d])^2 ; pred = argmin (icarl.py:122-152);
 * herding step: idx = argmin_i || mu - (S + f_i)/(k+1) ||_2 over rows with taken[i]==0
 * (buffer/linearherdingbuffer.py:140-161)                                                              */
/* out[r] = x[r] / ||x[r]||_2  (feats / feats.norm(dim=1), linearherdingbuffer.py:133, icarl.py:256) */
int clhip_l2_normalize_rows(const float* x, float* out, int R, int D, void* stream);
int clhip_ncm_classify(const float* feats, const float* means, int B, int M, int D, int64_t* pred, void* stream);
int clhip_herding_select(const float* feats /*[n,D], L2-normalised*/, int n, int D, int m, int32_t* chosen /*[m]*/,
                         float* ws /*[2*D + n]*/, void* stream);
/* the same for every class of a task in ONE launch (one block per class): class c = rows [offsets[c], offsets[c+1]) of feats (offsets: n_classes + 1
 * int32 on the device, max_rows = the largest class), picks are class-local indices in chosen[c * m ...] (-1 beyond a class's row count);
 * ws: 2 * D * n_classes + offsets[n_classes] floats.  Same arithmetic, same picks as n_classes calls of clhip_herding_select. */
int clhip_herding_select_batched(const float* feats, const int32_t* offsets, int n_classes, int max_rows, int D, int m, int32_t* chosen, float* ws,
                                 void* stream);

/* ------------------------------------------------------------------------------------------------
 * ViT path: frozen ViT-B/16 backbone with L2P prompt tokens or InfLoRA's
 * LoRA branch.  Token activations are [B*N, D] row-major in the compute dtype (batch-first: the reference's
 * seq-first [N,B,D] layout and its two permutes per block, transformer.py:1322-1329, do not exist here).
 *
 * clhip_gemm_nt: C[M,N] = epi(A[M,K] . B[N,K]^T); replaces every F.linear on the path (transformer.py:172,194,
 *   255,1267-1271) and, with the [in,out] copy of a frozen weight as B, its input gradient.  Epilogues:
 *   0 none | 1 +bias | 2 +bias +R (residual add, :1333-1334) | 3 +bias, then C <- GELU(.) (:1268) and H <- GELU'(.)
 *   (nullable; saved for the backward) | 4 C <- (.) * H (backward of 3).  bias fp32; R, H in the compute dtype.
 *   K % 64 == 0, N % 4 == 0.  Products with fewer than 256 tiles of 128 x 128 and K >= 3072 (bf16) run as 2-4 K slices + one fixed-order
 *   reduce / epilogue pass through a library-owned fp32 scratch (per stream, allocated at the first such call outside stream capture, never
 *   freed while a graph may hold it); results are bitwise reproducible either way.                                                          */
int clhip_gemm_nt(const void* A, const void* B, void* C, const float* bias, const void* R, void* H, int M, int N, int K,
                  int lda, int ldb, int ldc, int ldr, int ldh, int epilogue, int dtype, void* stream);
/* The input gradient AND the weight gradient of one layer in ONE launch (both consume dz and nothing of each other): the layers whose two
 * backward convolutions are launches at their latency floor -- 3x3/s1/p1 with 16 -> 16 channels on 32-wide or 32 -> 32 channels on 16-wide
 * images, bf16 (CifarResNet-32 stages 1 and 2, core/model/backbone/resnet.py:289-316 under autograd).  Same arguments and results (bit for
 * bit) as clhip_conv_dgrad [or clhip_conv_dgrad_bn_reduce when z_prod != NULL: then y_prod (nullable), mean, invstd, acc, replicas as there]
 * followed by clhip_conv_wgrad with scratch `ws` (clhip_conv_wgrad_ws_bytes; required: the weight gradient is the deterministic form). */
int clhip_conv_dgrad_wgrad_supported(int N, int H, int W, int C, int Creal, int K, int ksize, int stride, int pad, int dtype);
int clhip_conv_dgrad_wgrad(const void* x, const void* dz, const void* w_dg, void* dx, int accumulate, float* dw, void* ws,
                           const void* z_prod, const void* y_prod, const float* mean, const float* invstd, double* acc, int replicas,
                           int N, int H, int W, int C, int Creal, int K, int ksize, int stride, int pad, int dtype, void* stream);

/* "Lazy" BatchNorm input (core/model/backbone/resnet.py:289-316: conv -> bn -> relu -> conv inside a basic block).  The consumer
 * convolution reads its producer's PRE-BatchNorm output z' and applies relu(scale * z' + shift) while it stages the operand: the
 * producer's BatchNorm-apply launch and its activation tensor do not exist (the values are bit for bit the ones clhip_bn_apply_train would
 * have stored).  Every workgroup derives scale / shift from the producer's statistics accumulators as clhip_bn_apply_train does;
 * the launch also leaves `mean`, `invstd` (for clhip_bn_bwd_*) and `coef` = [2][C] scale, shift, and updates the running statistics.
 * Served by the kernels that stage their operand through registers: 3x3/s1/p1 with 16 -> 16 or 32 -> 32 channels, bf16 (CifarResNet-32
 * stages 1 and 2).  The backward of that consumer takes the same tensor and `coef`: clhip_conv_dgrad_wgrad_bn_input = clhip_conv_dgrad_wgrad
 * with x = relu(coef[0] * x_z + coef[1]) and, when acc != NULL, the producer's BatchNorm-backward sums reduced in the dgrad epilogue
 * with its ReLU mask taken from x_z (z_prod == x_z). */
typedef struct clhip_bn_input {
    const double* stat_acc; int replicas;     /* the producer's [replicas][2][C] fp64 sums (clhip_conv_fwd_acc); NULL = the producer's BatchNorm is in EVAL mode:
                                                 scale / shift come from running_mean / running_var (required, read only), mean / invstd / coef are not written
                                                 (nullable), and the launch takes stat_acc == NULL for its own output too (no statistics in eval mode) */
    const float* gamma; const float* beta;
    float* running_mean; float* running_var;  /* nullable pair */
    float momentum, eps;
    float* mean; float* invstd;               /* out [C] */
    float* coef;                              /* out [2][C] */
} clhip_bn_input;
/* The same in the backward for the layer's OWN BatchNorm: clhip_conv_dgrad_wgrad_bn_grad = clhip_bn_bwd_apply_acc (ReLU mask from z, no residual)
 * followed by clhip_conv_dgrad_wgrad, with the BatchNorm-backward result dz = scale * (g - mean(g) - xhat * mean(g xhat)) computed from dy and
 * z on both bodies' operand loads instead of being written and re-read; `sums` are the [replicas][2][C] accumulators a consumer's dgrad
 * epilogue filled (clhip_conv_dgrad_bn_reduce / clhip_conv_dgrad_wgrad*), one workgroup adds them to dgamma / dbeta.  Same domain. */
typedef struct clhip_bn_grad {
    const void* dy; const void* z;            /* [N,H,W,K] gradient of the layer's activation, its pre-BatchNorm output */
    const double* sums; int replicas;
    const float* mean; const float* invstd; const float* gamma; const float* beta;
    float* dgamma; float* dbeta;              /* accumulated into */
    const void* relu_mask;                    /* NULL: ReLU straight after the BatchNorm, mask from z; else the packed mask of a conv -> BN -> +res -> ReLU layer
                                                 (clhip_bn_apply_train_mask) and ... */
    void* dres; int dres_accumulate;          /* ... its residual gradient g = dy * mask, written / accumulated by the launch (NULL: none) */
} clhip_bn_grad;
int clhip_conv_dgrad_wgrad_bn_grad(const void* x, const float* x_coef /*nullable: x is a lazy BatchNorm input, z_prod == x, y_prod unused*/,
                                   const clhip_bn_grad* bn /*host*/, const void* w_dg, void* dx, int accumulate, float* dw, void* ws,
                                   const void* z_prod, const void* y_prod, const float* mean, const float* invstd, double* acc, int replicas, int N,
                                   int H, int W, int C, int Creal, int K, int ksize, int stride, int pad, int dtype, void* stream);
int clhip_conv_bn_input_supported(int N, int H, int W, int C, int K, int ksize, int stride, int pad, int dtype);
int clhip_conv_fwd_acc_bn_input(const void* z_in, const clhip_bn_input* bn /*host*/, const void* w_fwd, void* z, double* stat_acc, int replicas,
                                int N, int H, int W, int C, int K, int ksize, int stride, int pad, int dtype, void* stream);
/* The same when the producer is a conv -> BN -> +res -> ReLU layer (the last unit of a basic block, resnet.py:312-316): the operand is
 * relu(scale * z_in + shift + res), and because that activation has later readers (the next block's residual add, the backward) the
 * launch also WRITES it and its packed ReLU mask -- each pixel by the workgroup that owns it -- exactly as clhip_bn_apply_train_mask would
 * have: the apply launch disappears, the convolution's read of the activation becomes the read of z_in and res.  Same domain. */
typedef struct clhip_bn_res_input {
    const void* res;                          /* [N,H,W,C] the residual */
    void* y;                                  /* out [N,H,W,C] relu(bn(z_in) + res) */
    void* relu_mask;                          /* out [N*H*W*C/8] bytes, bit e = (stored element e > 0); nullable when the producer is in eval mode (bn->stat_acc == NULL) */
} clhip_bn_res_input;
int clhip_conv_fwd_acc_bn_res_input(const void* z_in, const clhip_bn_input* bn /*host*/, const clhip_bn_res_input* rs /*host*/, const void* w_fwd, void* z,
                                    double* stat_acc, int replicas, int N, int H, int W, int C, int K, int ksize, int stride, int pad, int dtype,
                                    void* stream);
/* The write-through form for the LDS-DMA kernels of the wide layers (64 ... 512 channels, 3x3 / stride 1: ResNet-18's BasicBlocks,
 * resnet.py:289-316; conv4.hip / conv5.hip): LDS-DMA has no arithmetic on the way, so the patch lands RAW (z_in) and the wave that issued a
 * DMA piece rewrites its slots in LDS with relu(scale * z_in + shift [+ res]) before the patch is published; the workgroup that owns a pixel
 * writes rs->y (always: the consumer's weight gradient reads it) and, for a conv -> BN -> +res -> ReLU producer, rs->relu_mask.
 * rs->res / rs->relu_mask NULL: plain relu(bn(z_in)).  Values bit for bit those of clhip_bn_apply_train[_mask]; bf16 only. */
int clhip_conv_bn_input_wt_supported(int N, int H, int W, int C, int K, int ksize, int stride, int pad, int dtype);
int clhip_conv_fwd_acc_bn_input_wt(const void* z_in, const clhip_bn_input* bn /*host*/, const clhip_bn_res_input* rs /*host*/, const void* w_fwd, void* z,
                                   double* stat_acc, int replicas, int N, int H, int W, int C, int K, int ksize, int stride, int pad, int dtype,
                                   void* stream);
int clhip_conv_dgrad_wgrad_bn_input(const void* x_z, const float* x_coef, const void* dz, const void* w_dg, void* dx, int accumulate, float* dw,
                                    void* ws, const float* mean, const float* invstd, double* acc /*nullable*/, int replicas, int N, int H, int W,
                                    int C, int Creal, int K, int ksize, int stride, int pad, int dtype, void* stream);

/* The input gradient of a DOWN-SAMPLING block entry in one launch (core/model/backbone/resnet.py:226-234: `conv1` 3x3/s2/p1 and
 * `downsample[0]` 1x1/s2/p0 both read the block input x [N,H,W,C]):  dx (+)= dgrad3x3s2(dz, W) + dgrad1x1s2(dz_sc, W_sc).
 * dz, dz_sc [N,H/2,W/2,K]; dz_sc NULL: the 3x3 convolution alone.  The weights come as ONE packed buffer of
 * clhip_conv_dgrad_pair_packed_bytes(C, K) bytes made by clhip_conv_dgrad_pair_pack from the dgrad copies w_dg [C][9][K] and
 * w_sc_dg [C][1][K] (nullable) of clhip_conv_weight_prep -- per (64-channel tile, 16-wide K chunk) the LDS image of that chunk's ten
 * taps (csrc/conv6.hip); a plan writes this layout in its own weight preparation.  Same result as clhip_conv_dgrad of the shortcut
 * followed by clhip_conv_dgrad(accumulate) of the 3x3 layer up to the bf16 rounding of the intermediate (here the two are summed
 * in fp32).  bf16, C % 64 == 0, K % 32 == 0, H/2 and W/2 powers of two, W/2 <= 16 (ResNet-18's three entries; csrc/conv6.hip), or
 * C in {16, 32} with K = 2 C (CifarResNet-32's two; csrc/conv7.hip, packed [C][10][K]).  _supported: 1 / 0. */
int clhip_conv_dgrad_pair_supported(int N, int H, int W, int C, int K, int dtype);
size_t clhip_conv_dgrad_pair_packed_bytes(int C, int K);
int clhip_conv_dgrad_pair_pack(const void* w_dg, const void* w_sc_dg /*nullable*/, void* packed, int C, int K, int dtype, void* stream);
int clhip_conv_dgrad_pair(const void* dz, const void* w_packed, const void* dz_sc /*nullable*/, void* dx, int accumulate, int N, int H,
                          int W, int C, int K, int dtype, void* stream);

/* clhip_conv_dgrad_pair whose epilogue also reduces the BatchNorm backward of the layer that PRODUCED the block input (z_prod, y_prod, mean, invstd,
 * acc, replicas exactly as clhip_conv_dgrad_bn_reduce; valid when this launch completes dx, i.e. the two layers are the activation's only readers).
 * The small-channel entries (C in {16, 32}, K = 2 C).  _supported: 1 / 0. */
int clhip_conv_dgrad_pair_bn_reduce_supported(int N, int H, int W, int C, int K, int dtype);
int clhip_conv_dgrad_pair_bn_reduce(const void* dz, const void* w_packed, const void* dz_sc /*nullable*/, void* dx, int accumulate, const void* z_prod,
                                    const void* y_prod /*nullable*/, const float* mean, const float* invstd, double* acc, int replicas, int N, int H,
                                    int W, int C, int K, int dtype, void* stream);
/* The WEIGHT gradients of the same two layers in one launch (csrc/conv7.hip): dw [K][9][C] += the 3x3/s2 layer's, dw_sc [K][1][C] += the
 * shortcut's, both over the block input x [N,H,W,C]; dz, dz_sc [N,H/2,W/2,K].  ws / ws_sc: scratch of clhip_conv_wgrad_ws_bytes() bytes of
 * the 3x3 / the 1x1 layer (partial blocks per image group, summed in a fixed order: bitwise reproducible).  bf16; CifarResNet-32's two
 * entries (32 x 32 x 16 -> 32 channels, 16 x 16 x 32 -> 64).  _supported: 1 / 0. */
int clhip_conv_wgrad_pair_supported(int N, int H, int W, int C, int K, int dtype);
/* ... and their two FORWARD convolutions in one launch: z [N,H/2,W/2,K] = conv3x3/s2/p1(x, w_fwd [K][9][C]), z_sc = conv1x1/s2(x, w_sc_fwd [K][1][C]),
 * each with its BatchNorm statistics added into its own fp64 accumulators exactly as clhip_conv_fwd_acc does.  Same domain as the small-channel
 * clhip_conv_dgrad_pair (C in {16, 32}, K = 2 C). */
int clhip_conv_fwd_acc_pair_supported(int N, int H, int W, int C, int K, int dtype);
int clhip_conv_fwd_acc_pair(const void* x, const void* w_fwd, const void* w_sc_fwd, void* z, void* z_sc, double* stat_acc, int replicas,
                            double* stat_acc_sc, int replicas_sc, int N, int H, int W, int C, int K, int dtype, void* stream);
int clhip_conv_wgrad_pair(const void* x, const void* dz, const void* dz_sc, float* dw, float* dw_sc, void* ws, void* ws_sc, int N, int H, int W,
                          int C, int K, int dtype, void* stream);

/* ---- run-time configuration ------------------------------------------------------------------------------------------------
 * ONE entry point for every dispatch switch, tuning value and micro-benchmark hook of the library (there are no other steering
 * exports).  `key` is a name from the list below (a leading "CLHIP_" is accepted), `value` its new value as text; value == NULL
 * returns the switch to its default, which is the environment variable CLHIP_<key> if that is set and the built-in default
 * otherwise.  Unknown keys fail with CLHIP_EINVAL.  Most switches are read once, at the first launch or plan creation that
 * consults them: configure before that.  The product never needs a call: the defaults ARE the product; tests use the switches to
 * pin a code path, tools/ to sweep.
 *   dispatch (0 / 1 unless noted):
 *     CONV4 (0: 3x3/s1 layers stay on conv3.hip), CONV5 (0: the 64 -> 64-channel 3x3/s1 layers stay on conv4.hip), CONV64 (0: 64 -> 64 channels on small maps stay on conv4 / wgrad4; CONV64_FWD 0: only their backward; CONV64_BM 64 | 128), CONV6 (0: no fused stride-2 dgrad pair kernel), CONV6_PAIR (0: plans keep the two separate input-gradient launches), CONV7 (0: no small-channel entry kernels; FWD7 0 / WGRAD7 0: not their forward / weight-gradient pairs; CONV7_TPW n: tiles per wave), BN_INPUT (0: no lazy BatchNorm inputs), BN_INPUT_WT (0: not on the LDS-DMA kernels of the wide layers), BN_RES_INPUT (0: block outputs keep their own apply launch), EVAL_LAZY (0: eval-mode forwards keep one BatchNorm apply launch per unit instead of the consumer-side forms), BN_GRAD (0: no BatchNorm backward on the operand loads; BN_GRAD_MINC n: only for layers of >= n channels, BN_GRAD_RES 0: not for the +res layers), CONV_V1,
 *     NO_CONV3, NO_CONV16, NO_STEM, NO_SHORTCUT, NO_PARITY_DGRAD, CONV3G, WGRAD4 (0 off, 2 stride-1 layers only), WGRAD5, WGRAD32,
 *     WGRAD_NO_TR, WGRAD2_ATOMIC (1: the generic weight-gradient kernel keeps fp32 atomics even when scratch is handed in), BWD_FUSED (0: dgrad and weight gradient of the 16 / 32-channel layers as two launches), WGRAD_DEFER_SIDE (n > 0: plans WITH a weight-gradient stream reduce in groups of n launches), WGRAD_DEFER (0: plans without a weight-gradient stream reduce their partial blocks per layer instead of once per backward), GEMM_NO_SPLIT, GEMM_TAIL, GEMM_SPLITK (0: no split-K for the few-tile / long-K products; n > 1: the minimum K that splits, default 3072), ATTN_GENERIC, CE_ROWS,
 *     BN_PARTIALS (partial rows + finalize launches instead of the fp64 accumulators), BN_FUSE (0 never, 1 everywhere; default: small
 *     activations), BN_FUSE_MAX_M, BN_MASK_BITS, BN_MASK_FROM_Y, BN_ONEPASS, PREP_NARROW,
 *     WGRAD_STREAM (0: weight gradients on the caller's stream), BRANCH_STREAM (shortcut branches on a third stream: 0 never, 1 forward and backward, 2 forward only = default, 3 backward only), WGRAD_ALWAYS_QUEUE, SIDE_PRIO, EVENT_FLAGS, EVENT_RECORD
 *     round 5: CONV8 (0: the 64 -> 64-channel 3x3/s1 layers with >= CONV8_MIN_TILES [1024] tiles of 128 pixels stay on conv5.hip), CONV8_BNR (1: conv8's dgrad epilogue
 *     reduces the producer's BatchNorm backward -- measured slower, off), CONV9 (1: 128 / 256-channel 3x3/s1 layers on conv9.hip -- equal in the step, off),
 *     STREAM_PROBE (0: the executors' extra stream is the first one created, not the first one MEASURED to run beside the caller's stream)
 *   tuning values:
 *     DZ_BUFFERS (2..4 rotating gradient buffers of the two-stream backward, default 4), CONV8_GRID, CONV8_OPT, CONV64_MAX_W, CONV64_MAX_M,
 *     WGRAD_TARGET (workgroups of the weight-gradient kernels), WGRAD_NET_GFLOP, WGRAD4_MIN_STEPS, WGRAD4_MIN_TOTAL, CONV3_CFG "wm,wn",
 *     CONV4_CFG "wm,wn,kg,ck", CONV4_GRID, CONV5_MIN_TILES, CONV5_GRID, PLAN_SKIP (timing ablations: 1 no forward BatchNorm apply, 2 no BatchNorm backward, 4 no weight gradients -- results invalid), IGEMM_TILE "bm,bn", GEMM_MT, GEMM_GROUP_M, STEM_GRID,
 *     STEM_WGRAD_GRID, SHORTCUT_MIN_PIXELS, BN_ACC_CPT, BN_BWD_ITERS
 *   micro-benchmark / ablation hooks (tools/ubench; take effect at once): CONV4_FORCE_CFG "wm,wn,kg,ck", CONV4_ENABLE, CONV4_DEBUG, CONV6_DEBUG (read once),
 *     CONV3_DEBUG, WGRAD_DEBUG (bit masks of phases to skip), CONV4_TRACE, WGRAD4_TRACE, CONV8_TRACE, CONV9_TRACE (device address of
 *     a stamp buffer as a number; ablation builds only) */
int clhip_config(const char* key, const char* value);
/* the value clhip_config() last set for `key` (NULL: never set or erased -- the environment's value applies).  For callers that flip a
 * switch around a region and must put back what was there (ops.TeacherPass).  The pointer stays valid for the process' life. */
const char* clhip_config_get(const char* key);
/* the 256 x 256 eight-phase kernel with 64-deep K tiles (gemm8.hip, round 5; bf16, N % 256 == 0, K % 128 == 0, K >= 256): 0 never, 1 (the default) the
 * row panels that fill whole rounds of its 256 persistent workgroups -- the remaining rows go to the register-staged kernel --, 2 wherever it is
 * supported (tests); -1 = from $CLHIP_GEMM8.  Same epilogues as clhip_gemm_nt's other kernels
 * (core/model/backbone/transformer.py:172, 194, 1259-1271). */
void clhip_gemm8_config(int mode);
/* workgroups the LDS-DMA weight-gradient kernel (wgrad4.hip) aims for: 0 = the default (160 -- 128 until round 4 --, chosen for the training step, where the
 * launch shares the chip with the dgrad / BatchNorm chain of the caller's stream; $CLHIP_WGRAD_TARGET), 256 = one per CU (the kernel
 * alone: bench.py's `full_chip` figures).  The scratch size (clhip_conv_wgrad_ws_bytes) follows the setting. */
void clhip_wgrad4_config(int target_workgroups);
/* softmax(q k^T / sqrt(d)) v per (batch, head) on the packed qkv [B*N, 3D] (column = which*D + head*d + i), out [B*N, D],
 * lse [B,H,N] (nullable in forward-only use); MultiHeadAttention.forward, transformer.py:169-197.  N <= 256, d <= 64. */
int clhip_attn_fwd(const void* qkv, void* out, float* lse, int B, int N, int H, int D, int dtype, void* stream);
/* dqkv [B*N, 3D] from dout [B*N, D]; dsum_ws: [B,H,N] floats (only used by the generic fp32 path) */
int clhip_attn_bwd(const void* qkv, const void* out, const float* lse, const void* dout, void* dqkv, float* dsum_ws, int B, int N,
                   int H, int D, int dtype, void* stream);
/* nn.LayerNorm over the last dim (transformer.py:1331-1336): y = (x-mean)*rstd*gamma+beta; mean/rstd [M] saved when given */
int clhip_ln_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int M, int D, float eps,
                 int dtype, void* stream);
/* g += dLN/dx^T dy  (gamma/beta are frozen on this path: input gradient only) */
int clhip_ln_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* g, int M, int D,
                 int dtype, void* stream);
/* final LN (eps 1e-6) + mean over the first P tokens of each sample -> feat [B,D] fp32 (P = 1: cls token;
 * P = top_k*length: L2P's prompt-token mean, transformer.py:2254-2261) and its backward into g [B*N, D] (rows >= P zeroed) */
int clhip_ln_pool_fwd(const void* x, const float* gamma, const float* beta, float* feat, int B, int N, int D, int P, float eps,
                      int dtype, void* stream);
int clhip_ln_pool_bwd(const float* dfeat, const void* x, const float* gamma, void* g, int B, int N, int D, int P, float eps,
                      int dtype, void* stream);
/* images fp32 NCHW [B,3,img,img] -> patch rows [B*(img/patch)^2, 3*patch^2] (timm PatchEmbed's Conv2d as a GEMM operand) */
int clhip_patchify(const float* images, void* patches, int B, int img, int patch, int dtype, void* stream);
/* x [B, P+1+np, D]: prompt tokens (no pos-embed) | cls + pos[0] | patch_emb + pos[1..]   (transformer.py:2239-2243, 2010-2014) */
int clhip_vit_assemble(const void* patch_emb, const float* cls_token, const float* pos_embed, const float* prompt_tokens, void* x,
                       int B, int n_patches, int n_prompt, int D, int dtype, void* stream);
/* dprompt_tokens[t] = sum_b g[b, t]  (t < n_prompt) */
int clhip_vit_prompt_grad(const void* g, float* dprompt, int B, int N, int n_prompt, int D, int dtype, void* stream);
/* fp32 master [rows, cols] (+ LoRA: rows [rows/3, 2rows/3) += B_k A_k, [2rows/3, rows) += B_v A_v, transformer.py:249-255)
 * -> compute-dtype copies wt [rows, cols] and wt_t [cols, rows] (either nullable) */
int clhip_weight_prep2(const float* w, void* wt, void* wt_t, int rows, int cols, const float* lora_a_k, const float* lora_b_k,
                       const float* lora_a_v, const float* lora_b_v, int rank, int dtype, void* stream);
/* per-step refresh of the k / v rows of the effective qkv copies (W + B A) of `layers` layers in ONE launch: arrays of `layers`
 * device pointers (fp32 masters and LoRA factors; wt [3D, D] and wt_t [D, 3D] in the compute dtype); q rows are left untouched */
int clhip_lora_qkv_refresh(int layers, const float* const* qkv_w, const float* const* lora_a_k, const float* const* lora_b_k,
                           const float* const* lora_a_v, const float* const* lora_b_v, void* const* wt, void* const* wt_t, int D, int rank,
                           int dtype, void* stream);
/* merge_weight (transformer.py:228-234) on the fp32 master qkv weight [3D, D] */
int clhip_lora_merge(float* qkv_w, const float* lora_a_k, const float* lora_b_k, const float* lora_a_v, const float* lora_b_v, int D,
                     int rank, void* stream);
/* d lora_B_k [D, r] += dK^T (X A_k^T), d lora_B_v likewise, dK/dV = columns [D,2D) / [2D,3D) of dqkv; x = the attention
 * input [M, D].  ws: clhip_lora_grad_ws_bytes(M, D, rank) bytes.  Deterministic (slab partials + ordered reduce). */
size_t clhip_lora_grad_ws_bytes(int M, int D, int rank);
/* a_cat (nullable): [32, D] compute-dtype matrix [A_k; A_v; 0] from clhip_lora_acat -- enables the bf16 MFMA path */
int clhip_lora_acat(const float* lora_a_k, const float* lora_a_v, void* a_cat, int D, int rank, int dtype, void* stream);
int clhip_lora_grad(const void* x, const void* dqkv, const float* lora_a_k, const float* lora_a_v, const void* a_cat, float* d_b_k,
                    float* d_b_v, void* ws, int M, int D, int rank, int dtype, void* stream);
/* G [D, D] fp32 += X^T X  (MultiHeadAttention_LoRA get_input_matrix, transformer.py:241-244; the running mean is the caller's) */
int clhip_gram_accum(const void* x, float* G, int M, int D, int dtype, void* stream);
/* the same for n_layers inputs of one pass in ONE launch: layer l reads x + l * layer_stride_elems and adds into G + l * D * D.  bf16: an
 * MFMA "TN" product, one workgroup per (layer, 128 x 128 tile) over all rows -- no atomics, bitwise reproducible; fp32: the scalar kernel per layer */
int clhip_gram_accum_batched(const void* x, size_t layer_stride_elems, int n_layers, float* G, int M, int D, int dtype, void* stream);
/* prompt.L2P.forward (prompt.py:369-406): cosine top-k per sample, batch-majority top-k ids (ties: lowest id), gathered
 * prompt tokens [top_k*length, D], reduce_sim (scalar) and d reduce_sim / d prompt_key [pool, D].  scratch: B+pool+D+B*pool floats. */
int clhip_l2p_select(const float* cls_feat, const float* prompt_key, const float* prompt, int B, int D, int pool, int top_k, int length,
                     int* ids, float* prompt_tokens, float* reduce_sim, float* dkey, float* scratch, void* stream);
int clhip_l2p_scatter(const float* dtokens, const int* ids, float* dprompt_pool, int pool, int top_k, int length, int D, void* stream);

/* Whole-backbone executor: one C call per forward / backward (VisionTransformer.forward, transformer.py:2222-2294, and
 * the autograd backward of L2P.observe / the trainer's loss.backward()). */
typedef struct clhip_vit_desc {
    int32_t img, patch, dim, depth, heads, mlp, lora_rank;
    float block_ln_eps;   /* eps of the per-block LayerNorms: 0 -> 1e-5 (transformer.py nn.LayerNorm default); timm-style trees
                           * (vit_inflora.py:375) use 1e-6 everywhere.  The final norm is 1e-6 in both. */
} clhip_vit_desc;
typedef struct clhip_vit_layer_params {
    const float *qkv_w, *qkv_b, *proj_w, *proj_b, *ln1_w, *ln1_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b, *ln2_w, *ln2_b;
    const float *lora_a_k, *lora_b_k, *lora_a_v, *lora_b_v;          /* NULL without LoRA */
} clhip_vit_layer_params;
typedef struct clhip_vit_params {
    const float *cls_token, *pos_embed, *pe_w, *pe_b, *norm_w, *norm_b;
    const clhip_vit_layer_params* layers;                            /* [depth] */
} clhip_vit_params;
typedef struct clhip_vit clhip_vit;
clhip_vit* clhip_vit_create(const clhip_vit_desc* desc, int dtype);
void clhip_vit_destroy(clhip_vit* v);
size_t clhip_vit_shadow_bytes(const clhip_vit* v);
size_t clhip_vit_workspace_bytes(const clhip_vit* v, int B, int n_prompt, int save_for_backward);
/* fp32 masters -> compute-dtype weight copies in `shadow`; apply_lora folds B A into the k/v rows; qkv_only != 0 refreshes
 * just the qkv copies (what changes between InfLoRA steps) */
int clhip_vit_prep_weights(clhip_vit* v, const clhip_vit_params* P, void* shadow, int apply_lora, int qkv_only, void* stream);
/* images fp32 [B,3,img,img]; prompt_tokens fp32 [n_prompt, D] (NULL / 0: none); feat [B, D] fp32 = final LN pooled over the
 * prompt tokens (n_prompt > 0) or at the cls token; gram (nullable) [depth, D, D] += X^T X of every attention input */
int clhip_vit_forward(clhip_vit* v, const clhip_vit_params* P, const void* shadow, void* workspace, const float* images, int B,
                      const float* prompt_tokens, int n_prompt, int save_for_backward, float* gram, float* feat, void* stream);
/* after a forward with save_for_backward: dprompt_tokens [n_prompt, D] (nullable); d_lora_b (nullable): [2*depth] pointers
 * (k, v per layer) accumulated into */
int clhip_vit_backward(clhip_vit* v, const clhip_vit_params* P, const void* shadow, void* workspace, const float* dfeat,
                       float* dprompt_tokens, float* const* d_lora_b, void* stream);
/* debug/test: copy one saved activation of layer l (0 x_in, 1 qkv, 2 attn out, 3 x_mid, 4 GELU derivative of the mlp) to fp32 */
int clhip_vit_read_act(clhip_vit* v, void* workspace, int layer, int which, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Input pipeline on the GPU.  store: uint8 [Nimg, H, W, 3] resident in HBM; index [B]: rows
 * of the store that form the batch; out: fp32 NCHW [B, 3, S, S].  mean3 / std3 are HOST pointers (3 floats each).
 * crop_flip = RandomCrop(S, padding=pad) + RandomHorizontalFlip + ColorJitter(brightness) + ToTensor + Normalize of the
 *   reference's CIFAR ResNet pipeline (core/data/data.py:4-19); params [B,3] = (dy, dx in [0, H+2pad-S], flip 0/1) drawn by the
 *   caller; brightness [B] factors or NULL.  (dy = dx = pad, flip = 0, brightness NULL = the test transform.)
 * rrc_flip  = RandomResizedCrop(S) (bilinear) + RandomHorizontalFlip + ToTensor [+ Normalize] of the ViT configs
 *   (config/l2p-...yaml train_trfms); params [B,5] = (y0, x0, h, w, flip): the crop box in source pixels. */
int clhip_augment_crop_flip(const uint8_t* store, const int64_t* index, const int32_t* params, const float* brightness /*nullable*/,
                            float* out, int B, int H, int W, int S, int pad, const float* mean3, const float* std3, void* stream);
int clhip_augment_rrc_flip(const uint8_t* store, const int64_t* index, const int32_t* params, float* out, int B, int H, int W, int S,
                           const float* mean3, const float* std3, void* stream);
/* rrc_aa = the same pipeline for stores whose images are LARGER than S (ImageNet-R, config/InfLoRA_opt-vit-imagenetr-b20-20-10.yaml
 *   :28-35), where torchvision's resize on PIL images anti-aliases: Pillow's two fixed-point passes (triangle filter of support
 *   max(scale, 1), 22-bit weights, uint8 intermediate image), reproduced bit for bit.  The store may be RAGGED: image i = uint8
 *   [hw[2i], hw[2i+1], 3] at store + offsets[i] (offsets / hw NULL: a uniform [Nimg, H, W, 3] store).  max_box >= the largest
 *   crop-box side of the batch (sizes the coefficient table: ws of clhip_augment_rrc_aa_ws_bytes(B, S, max_box) bytes). */
size_t clhip_augment_rrc_aa_ws_bytes(int B, int S, int max_box);
int clhip_augment_rrc_aa(const uint8_t* store, const int64_t* offsets /*nullable*/, const int32_t* hw /*nullable*/, const int64_t* index,
                         const int32_t* params, float* out, void* ws, int B, int H, int W, int S, int max_box, const float* mean3,
                         const float* std3, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CLHIP_H */
