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
int clhip_conv_dgrad(const void* dz, const void* w_dg, void* dx, int accumulate, int N, int H, int W, int C, int K,
                     int ksize, int stride, int pad, int dtype, void* stream);
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
size_t clhip_bn_bwd_ws_floats(int64_t M, int C);
int clhip_bn_bwd(const void* dy, const void* y, const void* z, const float* mean, const float* invstd, const float* gamma,
                 float* dgamma, float* dbeta, void* dz, void* dres, int dres_accumulate, int64_t M, int C, int relu,
                 float* ws, int dtype, void* stream);

/* global average pool: feat[N,C] (fp32) = mean_hw a[N,HW,C]; backward broadcasts dfeat/HW           */
int clhip_avgpool_fwd(const void* a, float* feat, int N, int HW, int C, int dtype, void* stream);
int clhip_avgpool_bwd(const float* dfeat, void* da, int N, int HW, int C, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Whole-backbone plan: a static list of (conv -> BN -> +res -> ReLU) units + global avg-pool, run
 * with ONE call per direction (replaces the per-op dispatch of CifarResNet.forward / ResNet._forward_impl
 * / modified_ResNet.forward, backbone/resnet.py:381-395, 215-223, 549-560, and autograd's backward).  */
typedef struct {
    int32_t cin, cout, ksize, stride, pad;
    int32_t src;      /* activation index consumed: 0 = network input, i+1 = output of unit i */
    int32_t res;      /* activation index added before the ReLU, or -1 */
    int32_t relu;
    int64_t w_off;    /* element offsets into the flat fp32 parameter / gradient buffers */
    int64_t gamma_off;
    int64_t beta_off;
    int64_t rm_off;   /* element offsets into the flat fp32 BN-statistics buffer */
    int64_t rv_off;
} clhip_unit_desc;

typedef struct clhip_plan clhip_plan;

clhip_plan* clhip_plan_create(const clhip_unit_desc* units /*host*/, int n_units, int N, int H, int W, int Cin, int dtype);
void clhip_plan_destroy(clhip_plan*);
size_t clhip_plan_workspace_bytes(const clhip_plan*);      /* activations + saved tensors + grads + scratch */
size_t clhip_plan_shadow_bytes(const clhip_plan*);         /* `dtype` copies of the conv weights */
int clhip_plan_feat_dim(const clhip_plan*);
/* refresh the `dtype` weight shadows from the fp32 masters (call after every optimizer step) */
int clhip_plan_prep_weights(clhip_plan*, const float* params, void* shadow, void* stream);
/* x: fp32 NCHW input; feat: fp32 [N, feat_dim].  training!=0: batch statistics, running stats updated,
 * activations saved in `workspace` for clhip_plan_backward.                                           */
int clhip_plan_forward(clhip_plan*, const float* x, const float* params, float* bn_stats, const void* shadow,
                       void* workspace, float* feat, int training, void* stream);
/* dfeat: fp32 [N, feat_dim]; grads: flat fp32 buffer laid out like params, accumulated into (+=).      */
int clhip_plan_backward(clhip_plan*, const float* dfeat, const float* params, const void* shadow, void* workspace,
                        float* grads, void* stream);
/* debugging / tests: copy activation `idx` (0=input) as fp32 NCHW; which: 0 = y, 1 = pre-BN z (idx>=1), 2 = dy */
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
int clhip_fisher_accum(float* fisher, const float* g, int64_t n, float scale, void* stream);
int clhip_fisher_merge(float* new_f, const float* old_f, int64_t n, float alpha, void* stream);
int clhip_sgd_step(float* p, const float* g, float* mom, int64_t n, float lr, float momentum, float weight_decay,
                   float grad_scale, const float* ewc_ref, const float* ewc_fisher, float ewc_weight, void* stream);
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

#ifdef __cplusplus
}
#endif
#endif /* CLHIP_H */
