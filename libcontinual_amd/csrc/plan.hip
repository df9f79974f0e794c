// plan.hip -- static execution plan for a ResNet backbone: a list of (conv -> BatchNorm -> +residual ->
// ReLU) units followed by a global average pool.  One C call runs the whole forward (or backward):
// no per-op host dispatch, no autograd graph, fixed workspace offsets (hipGraph-capturable: nothing
// here allocates or synchronises).  Replaces CifarResNet.forward / ResNet._forward_impl /
// modified_ResNet.forward (core/model/backbone/resnet.py:381-395, 215-223, 549-560) and the autograd
// backward the reference trainer triggers with loss.backward() (core/trainer.py:604).
//
// Pre-activation networks (ResNet_BIC, resnet.py:589-680: BN -> ReLU -> conv, shortcut added to the RAW conv output, BN-free 1x1
// shortcut convs) use the same unit list re-associated as  conv -> (+ raw sum of another unit) -> BN of the NEXT block -> ReLU,
// with three flag bits in `relu` (CLHIP_UNIT_*): PRE_RES adds `res`'s raw sum before the BatchNorm, RAW_SRC convolves the raw
// sum of `src` instead of its normalised output, NO_BN stops after the conv.  A raw sum consumed that way receives the consumer's
// gradient through a persistent buffer (`dzr`), added to its own BatchNorm-backward result.
#include <stdlib.h>
#include <vector>
#include <new>

#include "common.h"

void clhip_bn_set_stop_event(hipEvent_t ev);          // bn.hip: one-shot completion event of the next accumulator-path backward apply launch
hipEvent_t clhip_bn_pending_stop_event();
void clhip_bn_set_fwd_stop_event(hipEvent_t ev);      // ... of the next accumulator-path forward apply launch
void clhip_wgrad_defer_begin();                        // conv3.hip: collect the partial-block reduces of the weight-gradient launches ...
int clhip_wgrad_defer_flush(hipStream_t st, bool end);  // ... and run them as one launch
void clhip_wgrad_defer_abort();
void clhip_wgrad_defer_pause(bool paused);

namespace {
constexpr float kBnMomentum = 0.1f;   // nn.BatchNorm2d defaults used by every reference ResNet
constexpr float kBnEps = 1e-5f;

size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

struct Act {
    int H, W, C;
    size_t y_off;    // activation (bytes into workspace)
    size_t dy_off;   // its gradient
    size_t bytes;
};

}  // namespace
bool clhip_stage_eval_supported(int H, int W, int C, int nconv, int dtype);                                                   // stage.hip
int clhip_stage_eval_launch(const void* x, void* y, int N, int H, int W, int C, int nconv, const void* const* w, const float* const* gamma, const float* const* beta,
                            const float* const* mean, const float* const* var, float eps, int dtype, hipStream_t st);
bool clhip_stage_train_supported(int N, int H, int W, int C, int nconv, int dtype);                                            // stage_train.hip
size_t clhip_stage_train_xch_bytes(int N);
bool clhip_stage_train_xcd_rule(bool probe);
int clhip_stage_train_fwd_launch(const void* x, int N, int H, int W, int C, int nconv, const void* const* w, const float* const* gamma, const float* const* beta,
                                 float* const* rm, float* const* rv, float* const* mean, float* const* invstd, float* const* coef, void* const* z, void* const* y,
                                 void* const* mask, float momentum, float eps, void* xch, int trace, int entry, float* feat, int dtype, hipStream_t st);
int clhip_stage_train_bwd_launch(const void* x, const void* dy, void* dx, int dx_accumulate, int N, int H, int W, int C, int nconv, const void* const* wd,
                                 const float* const* gamma, const float* const* beta, const float* const* mean, const float* const* invstd, const void* const* z,
                                 const void* const* y, float* const* dgamma, float* const* dbeta, float* const* slab, void* const* dzg, void* xch, int trace, int entry, const float* dfeat, int dtype, hipStream_t st);
int clhip_stage_train_slab_blocks(int N, int C);
int clhip_stage_train_trace(void* xch, unsigned long long* out24);
int clhip_stage_train_status(void* xch);
int clhip_wgrad_reduce_launch(const float* slab, float* dw, int64_t n4, int splits, hipStream_t st);                                // conv3.hip
namespace {
struct Unit {
    clhip_unit_desc d;
    int cin_pad;
    int H, W, Ho, Wo;
    int64_t M;
    int tiles;
    size_t z_off;
    size_t sh_fwd, sh_dg;          // byte offsets in the shadow buffer
    size_t sh_pk;                  // packed dgrad weights of a paired down-sampling entry (3x3/s2 unit; see `pair`)
    int pair;                      // >= 0: this 3x3/s2 unit and the 1x1/s2 shortcut unit `pair` read the same activation: ONE launch writes its
                                   // gradient (clhip_conv_dgrad_pair); on the shortcut unit: the index of the 3x3 unit
    int lazy_to;                   // >= 0: this unit's activation relu(bn(z)) is consumed by unit `lazy_to` alone, through a convolution kernel that applies it
                                   // while it stages its operand (clhip_conv_fwd_acc_bn_input): the training forward skips the apply launch
    int lazy_from;                 // >= 0: the producer of this unit's input is such a unit
    int res_lazy_to;               // >= 0: a conv -> BN -> +res -> ReLU unit whose first consumer, convolution `res_lazy_to`, applies the BatchNorm, the
                                   // residual add and the ReLU on its operand load AND writes the activation + packed mask for the later readers
                                   // (clhip_conv_fwd_acc_bn_res_input): the training forward skips the apply launch
    int res_lazy_from;             // >= 0: this unit's input comes from such a unit
    int wt_to;                     // >= 0 (round 4, the LDS-DMA kernels of the wide layers): this unit's BatchNorm [+ residual] + ReLU is applied by convolution
                                   // `wt_to` -- the FIRST reader of its activation in unit order -- on the patch it has landed in LDS, and that launch WRITES the
                                   // activation [+ packed mask] for every later reader (clhip_conv_fwd_acc_bn_input_wt): the training forward skips the apply launch
    int wt_from;                   // >= 0: this unit's input comes from such a unit
    bool pair_fuse_bn;             // ... that one launch also reduces the BatchNorm backward of the unit that produced the activation (its only readers are the pair)
    bool fpair;                    // ... their two forward convolutions are one launch (clhip_conv_fwd_acc_pair, conv7.hip; set on both units)
    bool wpair;                    // ... and their two weight gradients are one launch too (clhip_conv_wgrad_pair, conv7.hip; set on both units)
    bool pair_acc;                 // the accumulate flag of that one launch (= the shortcut dgrad's, the first writer of the two)
    size_t f_mean, f_invstd, f_scale, f_shift;   // float offsets in the fp32 region
    size_t a_fwd, a_bwd;                         // double offsets in the accumulator region ([rep][2][cout] each)
    int rep_fwd, rep_bwd;                        // accumulator replicas (power of two): ~64 producer workgroups per replica
    int dx_acc, dres_acc;
    bool relu, pre_res, raw_src, no_bn;          // decoded CLHIP_UNIT_* bits of d.relu
    bool has_dzr;                                // this unit's raw sum is consumed raw (PRE_RES / RAW_SRC) by a later unit
    size_t dzr_off;                              // ... whose gradient contribution lands here (bytes into the workspace)
    size_t mask_off;                             // packed ReLU mask of a conv -> BN -> +res -> ReLU unit (bytes into the workspace; 0: none)
    int branch;                                  // >= 0: a shortcut unit (1x1 conv -> BN, consumed only as the residual of a later unit) that runs on
                                                 // the plan's branch stream beside the block's main path; value = its slot in the event arrays
    int forks;                                   // >= 0: this unit's forward BatchNorm launch completes ev_fork[forks] (its activation feeds a branch unit)
    int joins;                                   // >= 0: this unit adds the output of branch unit slot `joins` as its residual
    size_t wg_own;                               // this unit's own weight-gradient scratch (plans that defer the reduces), else the shared one
    int stage_len;                               // > 0 (round 5): this unit opens a run of `stage_len` units = stage_len / 2 BasicBlocks of C -> C 3x3 / stride-1 convolutions that
                                                 // the EVAL forward runs as ONE launch with the image resident in LDS (stage.hip); the activations inside the run are not written
    int entry_first;                             // >= 0 (on the unit that opens a training run): the run's launch can start three units earlier, with the stage's down-sampling
                                                 // block -- units entry_first (3x3 / s2), + 1 (1x1 / s2 shortcut), + 2 (second convolution) -- in front of it (STAGE_ENTRY)
    int run_first;                               // >= 0: this unit lies inside the training run that unit `run_first` opens (stage_train)
    size_t st_slab;                              // ... its weight-gradient slab of that launch: N blocks of cout * 9 * cin floats (bytes into the workspace)
    bool stage_train;                            // (round 6) ... and the TRAINING forward / backward can run it as one launch per direction (stage_train.hip): every
                                                 // first convolution of a block hands its activation to the second one lazily, every block output keeps a packed mask
    bool fuse_src_bn;                            // this unit's dgrad completes the gradient of its input activation AND can reduce the
                                                 // BatchNorm backward of the unit that produced it in its epilogue (clhip_conv_dgrad_bn_reduce)
};
}  // namespace

struct clhip_plan {
    int N, H, W, Cin, Cin_pad, dtype, esize;
    std::vector<Unit> units;
    std::vector<Act> acts;
    size_t ws_bytes, shadow_bytes;
    size_t dz_off;           // scratch for the pre-BN gradient
    static constexpr int kDz = 4;   // dz buffers of the two-stream backward (round 5: four instead of two -- the caller's chain may run up to three units ahead
                             // of the weight-gradient stream, which is the longer of the two on the 8x8 / 4x4 stages; DZ_BUFFERS = 2 restores the twin)
    size_t dz_offs[kDz];
    int n_dz;
    size_t dz_off2;          // its twin: units alternate, so unit i's weight gradient (side stream) may still read one while unit i-1's
                             // BatchNorm backward (main stream) fills the other
    hipStream_t side;        // weight-gradient stream (created on first use), with the events that order it against the caller's stream
    hipEvent_t ev_dz[kDz], ev_wg[kDz], ev_end;
    bool wg_pending[kDz];
    // branch stream: the 1x1 shortcut convolution + BatchNorm of a down-sampling block is independent of the block's first 3x3 unit in
    // the forward, and its BatchNorm backward / input gradient / weight gradient are independent of that unit's in the backward: small
    // launches that do not fill the chip, run beside the main path instead of in front of it
    static constexpr int kMaxBranch = 8;
    hipStream_t br;
    hipEvent_t ev_fork[kMaxBranch], ev_join[kMaxBranch], ev_bfork[kMaxBranch], ev_bjoin[kMaxBranch];
    int n_branch;
    int br_act;              // backward: activation whose gradient the branch stream is writing (-1: none); ev_bjoin[br_slot] orders it
    int br_slot;
    hipEvent_t ev_br_end;
    hipEvent_t bfork_ev[kMaxBranch];   // backward: the event that completes when the consumer's BatchNorm backward has written this branch's dy
    size_t wg_off2;          // weight-gradient scratch of the branch stream
    bool defer_reduce;       // every unit has its own scratch and the "dw += slab" reduces of a backward run as ONE launch at its end
    int defer_side;          // > 0 (plans WITH a weight-gradient stream): the reduces of that stream's launches run in groups of this many
    size_t f_base;           // byte offset of the fp32 region
    size_t f_part, f_bnws;   // float offsets: conv stat partials, bn backward scratch
    size_t wg_off;           // byte offset of the weight-gradient partial-block scratch (0 bytes if unused)
    size_t acc_off, acc_bytes;   // fp64 BN accumulators of all units (forward sums, backward sums): zeroed once per training forward
    bool use_acc;            // some unit takes its BN statistics through the fp64 accumulators (see Unit::acc_fwd)
    std::vector<char> res_pending;      // per unit: its forward skipped the (+res) apply launch and its consumer has not run yet
    std::vector<char> wt_pending;       // ... likewise for the write-through form of the wide layers
    const float* params_dev = nullptr;  // the parameter / running-statistics pointers of the last forward (clhip_plan_read_act rebuilds an eval-lazy activation from them)
    const float* bn_stats_dev = nullptr;
    std::vector<char> stage_skipped;    // per unit: the last EVAL forward ran it inside a fused stage launch: its activation does not exist (clhip_plan_read_act refuses)
    std::vector<char> eval_unwritten;   // per unit: the last EVAL forward consumed its BatchNorm on the next convolution's operand load and never wrote the activation
    std::vector<char> lazy_live;        // per unit: the last training forward left its activation unwritten (its z, mean / invstd and coefficients are there)
    std::vector<char> bwd_sums_ready;   // per unit: its BatchNorm-backward sums were accumulated by a consumer's dgrad (since the last forward)
    std::vector<char> mask_stale;       // per unit: the last training forward ran it inside a stage-level launch, which writes the block outputs but NOT their packed ReLU
                                        // masks (byte stores from an MFMA-layout epilogue cost more than the rest of the epilogue): a per-unit backward reads the activation
    void* xch = nullptr;     // exchange buffer of the stage-level training launches (xch.h): owned by the plan, zeroed at creation
    hipStream_t xch_stream = nullptr;   // the stream of the last such launch (two of them must never be in flight on two streams: see stage_train_serialize)
    bool xch_used = false;
    long long st_fwd_launches = 0, st_bwd_launches = 0;      // stage-level training launches so far (clhip_plan_stage_info)
    int feat_dim;
    bool side_ok;            // some unit's weight gradient is big enough for the side stream to pay (see clhip_plan_backward_range)
    int pool_win;            // 0: global average pool, else nn.AvgPool2d(pool_win) + NCHW flatten
};

extern "C" clhip_plan* clhip_plan_create(const clhip_unit_desc* units, int n_units, int N, int H, int W, int Cin, int dtype) {
    return clhip_plan_create_ex(units, n_units, N, H, W, Cin, dtype, 0);
}

extern "C" clhip_plan* clhip_plan_create_ex(const clhip_unit_desc* units, int n_units, int N, int H, int W, int Cin, int dtype, int pool_win) {
    if (!units || n_units <= 0 || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || pool_win < 0 || (dtype != CLHIP_BF16 && dtype != CLHIP_F32)) {
        clhip_set_error("clhip_plan_create: invalid argument");
        return nullptr;
    }
    clhip_plan* p = new (std::nothrow) clhip_plan();
    if (!p) { clhip_set_error("out of host memory"); return nullptr; }
    p->N = N; p->H = H; p->W = W; p->Cin = Cin; p->dtype = dtype; p->esize = dtype == CLHIP_BF16 ? 2 : 4;
    p->pool_win = pool_win;
    p->Cin_pad = (Cin + 7) / 8 * 8;
    if (ilog2_exact(p->Cin_pad) < 0) { int c = 8; while (c < Cin) c <<= 1; p->Cin_pad = c; }
    size_t off = 0, sh = 0;
    Act a0{H, W, p->Cin_pad, 0, 0, (size_t)N * H * W * p->Cin_pad * p->esize};
    a0.y_off = off; off = align_up(off + a0.bytes);
    p->acts.push_back(a0);
    size_t max_z = 0, max_part = 0, max_bnws = 0, nfloat = 0, max_wg = 0, ndouble = 0;
    // BN statistics go through fp64 atomic accumulators (conv epilogue / backward-reduce workgroups ADD their per-channel sums,
    // the apply kernels finalise on the fly): no finalize launches, no per-tile partial rows.  Same-address atomic traffic is
    // what it costs (measured: +5..9 us per kernel with 1024 producers on ONE accumulator), so a layer gets one replica per
    // ~64 producer workgroups (blockIdx & (rep-1)); the consumers sum the replicas in a block-cooperative prologue.
    // CLHIP_BN_PARTIALS=1 restores the partial-row + finalize path.
    const bool want_acc = clhip_cfg("BN_PARTIALS") == nullptr;
    auto replicas = [](int producers) { int r = 1; while (r < 32 && producers > 64 * r) r <<= 1; return r; };
    p->use_acc = false;
    for (int i = 0; i < n_units; ++i) {
        Unit u{};
        u.d = units[i];
        u.relu = (u.d.relu & CLHIP_UNIT_RELU) != 0; u.pre_res = (u.d.relu & CLHIP_UNIT_PRE_RES) != 0;
        u.raw_src = (u.d.relu & CLHIP_UNIT_RAW_SRC) != 0; u.no_bn = (u.d.relu & CLHIP_UNIT_NO_BN) != 0;
        if (u.d.src < 0 || u.d.src > i || u.d.res > i || (u.d.ksize != 1 && u.d.ksize != 3) || u.d.stride < 1 || u.d.cout % 16 ||
            (u.raw_src && u.d.src < 1) || (u.pre_res && u.d.res < 1) || (u.no_bn && (u.pre_res || u.relu || u.d.res >= 0)) ||
            (u.d.relu & ~(CLHIP_UNIT_RELU | CLHIP_UNIT_PRE_RES | CLHIP_UNIT_RAW_SRC | CLHIP_UNIT_NO_BN))) {
            clhip_set_error("clhip_plan_create: bad unit %d", i);
            delete p;
            return nullptr;
        }
        const Act& s = p->acts[u.d.src];
        u.cin_pad = s.C;
        if ((u.d.src == 0 && u.d.cin != Cin) || (u.d.src != 0 && u.d.cin != s.C)) {
            clhip_set_error("clhip_plan_create: unit %d channel mismatch", i);
            delete p;
            return nullptr;
        }
        u.H = s.H; u.W = s.W;
        u.Ho = (s.H + 2 * u.d.pad - u.d.ksize) / u.d.stride + 1;
        u.Wo = (s.W + 2 * u.d.pad - u.d.ksize) / u.d.stride + 1;
        u.M = (int64_t)N * u.Ho * u.Wo;
        u.tiles = clhip_conv_fwd_tiles(N, u.H, u.W, u.cin_pad, u.d.cout, u.d.ksize, u.d.stride, u.d.pad);
        if (u.d.res >= 0) {
            const Act& r = p->acts[u.d.res];
            if (r.H != u.Ho || r.W != u.Wo || r.C != u.d.cout) {
                clhip_set_error("clhip_plan_create: unit %d residual shape mismatch", i);
                delete p;
                return nullptr;
            }
        }
        size_t bytes = (size_t)u.M * u.d.cout * p->esize;
        u.z_off = off; off = align_up(off + bytes);
        Act a{u.Ho, u.Wo, u.d.cout, off, 0, bytes};
        off = align_up(off + bytes);
        p->acts.push_back(a);
        size_t wbytes = (size_t)u.d.cout * u.d.ksize * u.d.ksize * u.cin_pad * p->esize;
        u.sh_fwd = sh; sh = align_up(sh + wbytes);
        u.sh_dg = sh; sh = align_up(sh + wbytes);
        u.f_mean = nfloat; u.f_invstd = nfloat + u.d.cout; u.f_scale = nfloat + 2 * (size_t)u.d.cout; u.f_shift = nfloat + 3 * (size_t)u.d.cout;
        nfloat += 4 * (size_t)u.d.cout;
        const bool pow2 = (u.d.cout & (u.d.cout - 1)) == 0;
        u.rep_fwd = (want_acc && pow2) ? replicas(u.tiles) : 0;
        if (u.pre_res) {            // statistics of the SUM come from clhip_add_stats, always through the accumulator
            if (!pow2) { clhip_set_error("clhip_plan_create: unit %d: a pre-BatchNorm residual needs a power-of-two channel count", i); delete p; return nullptr; }
            u.rep_fwd = replicas(clhip_add_stats_blocks(u.M, u.d.cout));
        }
        u.rep_bwd = (want_acc && pow2) ? replicas(clhip_bn_bwd_blocks(u.M, u.d.cout)) : 0;
        // conv -> BN -> +residual -> ReLU units: the forward apply also writes one mask bit per element, and the two backward passes read
        // that instead of the activation (1/16 of its bytes; CLHIP_BN_MASK_BITS=0: read y)
        static const bool mask_bits = !(clhip_cfg("BN_MASK_BITS") && atoi(clhip_cfg("BN_MASK_BITS")) == 0);
        u.mask_off = 0;
        if (mask_bits && u.relu && u.d.res >= 0 && !u.pre_res && !u.no_bn && !u.raw_src && u.rep_fwd > 0 && u.rep_bwd > 0) {
            u.mask_off = off;
            off = align_up(off + (size_t)u.M * u.d.cout / 8);
        }
        if (u.no_bn) u.rep_fwd = u.rep_bwd = 0;
        u.a_fwd = ndouble; ndouble += 2 * (size_t)u.d.cout * (u.rep_fwd > 0 ? u.rep_fwd : 1);
        u.a_bwd = ndouble; ndouble += 2 * (size_t)u.d.cout * (u.rep_bwd > 0 ? u.rep_bwd : 1);
        if (u.rep_fwd || u.rep_bwd) p->use_acc = true;
        if (bytes > max_z) max_z = bytes;
        size_t part = (size_t)u.tiles * 2 * u.d.cout;
        if (part > max_part) max_part = part;
        size_t bw = clhip_bn_bwd_ws_floats(u.M, u.d.cout);
        if (bw > max_bnws) max_bnws = bw;
        size_t wg = clhip_conv_wgrad_ws_bytes(N, u.H, u.W, u.cin_pad, u.d.cin, u.d.cout, u.d.ksize, u.d.stride, u.d.pad, dtype);
        if (wg > max_wg) max_wg = wg;
        p->units.push_back(u);
    }
    // shortcut branches: unit b is a branch when it is a plain 1x1 conv -> BN without ReLU on the accumulator path whose activation has exactly
    // one consumer, which takes it as a (post-BatchNorm) residual, and whose input is the activation of an earlier plain unit
    p->n_branch = 0; p->br = nullptr; p->br_act = -1; p->br_slot = -1;
    size_t max_wg2 = 0;
    for (auto& u : p->units) { u.branch = u.forks = u.joins = -1; }
    static const bool branch_off = clhip_cfg("BRANCH_STREAM") != nullptr && atoi(clhip_cfg("BRANCH_STREAM")) == 0;
    for (int b = 0; b < n_units && !branch_off; ++b) {
        Unit& u = p->units[b];
        if (u.d.ksize != 1 || u.relu || u.pre_res || u.raw_src || u.no_bn || u.d.res >= 0 || u.rep_fwd <= 0 || u.rep_bwd <= 0 || u.d.src < 1) continue;
        if (p->n_branch >= clhip_plan::kMaxBranch) break;
        int consumer = -1, n_cons = 0;
        for (int k = 0; k < n_units; ++k) {
            const Unit& o = p->units[k];
            if (o.d.src == b + 1) n_cons += 2;                                   // consumed through a convolution: not a pure shortcut
            if (o.d.res == b + 1) { ++n_cons; consumer = k; }
        }
        if (n_cons != 1 || consumer <= b || p->units[consumer].pre_res || p->units[consumer].rep_fwd <= 0 || p->units[consumer].rep_bwd <= 0) continue;
        Unit& prod = p->units[u.d.src - 1];
        if (prod.no_bn || prod.rep_fwd <= 0 || prod.forks >= 0 || p->units[consumer].joins >= 0) continue;
        u.branch = p->n_branch; prod.forks = p->n_branch; p->units[consumer].joins = p->n_branch;
        ++p->n_branch;
        size_t wg = clhip_conv_wgrad_ws_bytes(N, u.H, u.W, u.cin_pad, u.d.cin, u.d.cout, u.d.ksize, u.d.stride, u.d.pad, dtype);
        if (wg > max_wg2) max_wg2 = wg;
    }
    for (size_t i = 1; i < p->acts.size(); ++i) { p->acts[i].dy_off = off; off = align_up(off + p->acts[i].bytes); }
    // raw sums consumed raw: exactly one such consumer each (true of every pre-activation ResNet), whose gradient contribution gets
    // a buffer of its own
    for (int i = 0; i < n_units; ++i) {
        const Unit& c = p->units[i];
        const int targets[2] = {c.pre_res ? c.d.res : 0, c.raw_src ? c.d.src : 0};
        for (int t : targets) {
            if (t < 1) continue;
            Unit& prod = p->units[t - 1];
            if (prod.has_dzr) { clhip_set_error("clhip_plan_create: the raw output of unit %d has two raw consumers", t - 1); delete p; return nullptr; }
            prod.has_dzr = true;
            prod.dzr_off = off; off = align_up(off + p->acts[t].bytes);
        }
    }
    for (int i = 0; i < n_units; ++i)
        if (p->units[i].no_bn && !p->units[i].has_dzr) { clhip_set_error("clhip_plan_create: unit %d has no BatchNorm and no consumer", i); delete p; return nullptr; }
    {
        const double net_flops = 1.0e9 * (clhip_cfg("WGRAD_NET_GFLOP") ? atof(clhip_cfg("WGRAD_NET_GFLOP")) : 4.0);
        p->side_ok = false;
        for (const Unit& u : p->units)
            if (2.0 * (double)u.M * u.d.ksize * u.d.ksize * u.d.cin * u.d.cout >= net_flops) p->side_ok = true;
    }
    p->dz_off = off; off = align_up(off + max_z);
    p->dz_off2 = off; off = align_up(off + max_z);
    {
        static const int ndz_cfg = clhip_cfg("DZ_BUFFERS") ? atoi(clhip_cfg("DZ_BUFFERS")) : clhip_plan::kDz;
        p->n_dz = (p->side_ok && ndz_cfg >= 2 && ndz_cfg <= clhip_plan::kDz) ? ndz_cfg : 2;
        p->dz_offs[0] = p->dz_off; p->dz_offs[1] = p->dz_off2;
        for (int q = 2; q < clhip_plan::kDz; ++q) {
            p->dz_offs[q] = p->dz_off;
            if (q < p->n_dz) { p->dz_offs[q] = off; off = align_up(off + max_z); }
        }
    }
    p->wg_off = off; off = align_up(off + max_wg);
    // networks whose weight gradients stay on the caller's stream (no layer big enough for the side stream: the CIFAR ResNet-32s): one
    // scratch region PER UNIT, so that the partial-block reduces can wait for the end of the backward and run as one launch
    // (33 launches of ~5 us per CifarResNet-32 step otherwise); WGRAD_DEFER=0 keeps the per-layer reduces
    static const bool defer_off = clhip_cfg("WGRAD_DEFER") != nullptr && atoi(clhip_cfg("WGRAD_DEFER")) == 0;
    p->defer_reduce = !p->side_ok && !defer_off;
    static const int defer_side = clhip_cfg("WGRAD_DEFER_SIDE") ? atoi(clhip_cfg("WGRAD_DEFER_SIDE")) : 0;
    p->defer_side = p->side_ok ? defer_side : 0;
    for (Unit& u : p->units) {
        u.wg_own = p->wg_off;
        if (p->defer_reduce || p->defer_side > 0) {
            u.wg_own = off;
            off = align_up(off + clhip_conv_wgrad_ws_bytes(N, u.H, u.W, u.cin_pad, u.d.cin, u.d.cout, u.d.ksize, u.d.stride, u.d.pad, dtype));
        }
    }
    p->wg_off2 = off; off = align_up(off + max_wg2);
    p->f_base = off;
    nfloat = (nfloat + 63) / 64 * 64;
    p->f_part = nfloat; nfloat += (max_part + 63) / 64 * 64;
    p->f_bnws = nfloat; nfloat += (max_bnws + 63) / 64 * 64;
    off = align_up(off + nfloat * sizeof(float));
    p->acc_off = off; p->acc_bytes = ndouble * sizeof(double);
    off = align_up(off + p->acc_bytes);
    p->ws_bytes = off;
    p->shadow_bytes = sh;
    {
        const Act& last = p->acts.back();
        if (p->units.back().no_bn || (pool_win > 0 && (last.H < pool_win || last.W < pool_win))) {
            clhip_set_error("clhip_plan_create: the last unit cannot feed the pool");
            delete p;
            return nullptr;
        }
        p->feat_dim = pool_win > 0 ? last.C * (last.H / pool_win) * (last.W / pool_win) : last.C;
    }
    // gradient write/accumulate flags: simulate the reverse sweep
    std::vector<char> written(p->acts.size(), 0);
    written.back() = 1;   // pool backward writes the last activation's gradient
    for (int i = n_units - 1; i >= 0; --i) {
        Unit& u = p->units[i];
        if (u.d.res >= 0 && !u.pre_res) { u.dres_acc = written[u.d.res]; written[u.d.res] = 1; }
        if (u.d.src != 0 && !u.raw_src) { u.dx_acc = written[u.d.src]; written[u.d.src] = 1; }
    }
    // BatchNorm-backward reduction fused into a dgrad epilogue: unit i's dgrad may reduce for the producer of its input activation
    // when it is the LAST writer of that activation's gradient in the reverse sweep (= the lowest-index consumer, and it consumes
    // through its convolution, not through a residual add), the producer is a plain conv -> BN [-> +res] [-> ReLU] unit on the
    // accumulator path, and the fourth-generation dgrad kernel covers the layer.  On the 32x32 / 16x16 stages of the ResNet-18 step at batch
    // 256 it is time-neutral to slightly slower (2.617 ms all fused vs 2.592 ms with the 13 separate reduce passes, profiles/r02_conv4_notes.md):
    // the epilogue's 8-byte gathers of z / y and its 16-lane DPP reductions cost what the streaming reduce pass costs there.
    // Default: fused for activations of at most 16384 pixels (the 8x8 / 4x4 stages at batch 256, everything at batch 32), where the reduce
    // pass is a launch at its latency floor (7-9 us) and the epilogue's extra work is small: 2.321 -> 2.305 ms per step.  CLHIP_BN_FUSE=0:
    // never; =1: every qualifying layer (CLHIP_BN_FUSE_MAX_M bounds the pixels).  Read per plan: tests build both variants.
    const char* fe = clhip_cfg("BN_FUSE");
    const bool fuse_on = fe == nullptr || atoi(fe) != 0;
    // (plans without a weight-gradient stream -- the CIFAR ResNet-32s, every launch of which sits at its latency floor -- fuse wherever a
    //  kernel supports it: one launch less per unit, measured in profiles/r03_step_notes.md)
    const long long fuse_max_m = clhip_cfg("BN_FUSE_MAX_M") ? atoll(clhip_cfg("BN_FUSE_MAX_M")) : ((fe == nullptr && p->side_ok) ? 16384 : (1ll << 62));
    p->bwd_sums_ready.assign(p->units.size(), 0);
    p->mask_stale.assign(p->units.size(), 0);
    for (int i = 0; i < n_units; ++i) {
        Unit& u = p->units[i];
        u.fuse_src_bn = false;
        if (!fuse_on || u.d.src < 1 || u.raw_src || u.pre_res || u.no_bn) continue;
        if ((long long)u.M > fuse_max_m && !clhip_conv_dgrad_bn_reduce_overlapped(N, u.H, u.W, u.cin_pad, u.d.cout, u.d.ksize, u.d.stride, u.d.pad, dtype)) continue;
        const int a = u.d.src;                      // the activation; produced by unit a - 1
        const Unit& prod = p->units[a - 1];
        if (prod.no_bn || prod.pre_res || prod.raw_src || prod.has_dzr || prod.rep_bwd <= 0) continue;
        bool lowest = true;
        for (int k = 0; k < n_units; ++k) {
            if (k == i) continue;
            const Unit& o = p->units[k];
            const bool consumes = (o.d.src == a && !o.raw_src) || (o.d.res == a && !o.pre_res);
            if (consumes && k < i) lowest = false;
        }
        if (u.d.res == a) lowest = false;            // the unit's own residual add writes after its dgrad?  (never in these nets; stay safe)
        if (!lowest) continue;
        if (!clhip_conv_dgrad_bn_reduce_supported(N, u.H, u.W, u.cin_pad, u.d.cout, u.d.ksize, u.d.stride, u.d.pad, dtype)) continue;
        u.fuse_src_bn = true;
    }
    // down-sampling entries: a 3x3/s2/p1 unit a and a 1x1/s2/p0 unit b > a with the same source activation and channel counts, the only two
    // consumers of that activation -> their two input gradients are one launch at unit a (conv6.hip); the packed weights of both live at a.sh_pk
    for (auto& u : p->units) { u.pair = -1; u.pair_acc = false; u.sh_pk = 0; u.wpair = u.fpair = u.pair_fuse_bn = false; }
    const char* pair_cfg = clhip_cfg("CONV6_PAIR");             // (per plan: the tests build one with and one without)
    const bool pair_off = pair_cfg != nullptr && atoi(pair_cfg) == 0;
    for (int b = 0; b < n_units && !pair_off; ++b) {
        Unit& ub = p->units[b];
        if (ub.d.ksize != 1 || ub.d.stride != 2 || ub.d.pad != 0 || ub.pre_res || ub.raw_src || ub.no_bn || ub.has_dzr || ub.d.src < 1) continue;
        int a = -1, users = 0;
        for (int k = 0; k < n_units; ++k) {
            const Unit& o = p->units[k];
            if (o.d.src == ub.d.src || o.d.res == ub.d.src) ++users;
            if (k < b && o.d.src == ub.d.src && o.d.ksize == 3 && o.d.stride == 2 && o.d.pad == 1 && !o.raw_src && !o.pre_res && !o.no_bn && !o.has_dzr &&
                o.d.cout == ub.d.cout && o.cin_pad == ub.cin_pad && !o.fuse_src_bn)
                a = k;
        }
        if (a < 0 || users != 2 || p->units[a].pair >= 0) continue;
        Unit& ua = p->units[a];
        if (!clhip_conv_dgrad_pair_supported(N, ua.H, ua.W, ua.cin_pad, ua.d.cout, dtype)) continue;
        ua.pair = b; ub.pair = a;
        ua.pair_acc = ub.dx_acc;
        ua.sh_pk = p->shadow_bytes; p->shadow_bytes = align_up(p->shadow_bytes + clhip_conv_dgrad_pair_packed_bytes(ua.cin_pad, ua.d.cout));
        // (own scratch regions per unit -- the deferred-reduce plans -- so that both partial-block slabs survive until the one reduce launch)
        {
            const Unit& prod = p->units[ua.d.src - 1];       // users == 2: the pair's launch is the only writer of that activation's gradient
            ua.pair_fuse_bn = fuse_on && !prod.no_bn && !prod.pre_res && !prod.raw_src && !prod.has_dzr && prod.rep_bwd > 0 && (long long)prod.M <= fuse_max_m &&
                              clhip_conv_dgrad_pair_bn_reduce_supported(N, ua.H, ua.W, ua.cin_pad, ua.d.cout, dtype) != 0;
        }
        ua.fpair = ub.fpair = ua.rep_fwd > 0 && ub.rep_fwd > 0 && ua.cin_pad == ua.d.cin &&
                              clhip_conv_fwd_acc_pair_supported(N, ua.H, ua.W, ua.cin_pad, ua.d.cout, dtype) != 0;
        ua.wpair = ub.wpair = p->defer_reduce && ua.cin_pad == ua.d.cin && ub.cin_pad == ub.d.cin &&
                              clhip_conv_wgrad_pair_supported(N, ua.H, ua.W, ua.cin_pad, ua.d.cout, dtype) != 0;
    }
    // lazy activations: unit a = conv -> BN -> ReLU whose activation has exactly one consumer b, a convolution that can apply the BatchNorm + ReLU
    // on its operand load (forward) and in its fused dgrad + weight-gradient launch (backward)
    for (auto& u : p->units) { u.lazy_to = u.lazy_from = -1; }
    p->lazy_live.assign(p->units.size(), 0);
    p->eval_unwritten.assign(p->units.size(), 0);
    p->stage_skipped.assign(p->units.size(), 0);
    static const bool mask_y = clhip_cfg("BN_MASK_FROM_Y") != nullptr;
    for (int a = 0; a + 1 < n_units && want_acc && !mask_y; ++a) {
        Unit& ua = p->units[a];
        if (!ua.relu || ua.d.res >= 0 || ua.pre_res || ua.no_bn || ua.has_dzr || ua.rep_fwd <= 0 || ua.rep_bwd <= 0 || ua.mask_off != 0 || ua.branch >= 0 || ua.forks >= 0) continue;
        int b = -1, users = 0;
        for (int k = 0; k < n_units; ++k) {
            const Unit& o = p->units[k];
            if (o.d.src == a + 1) { ++users; b = k; if (o.raw_src) users += 2; }
            if (o.d.res == a + 1) users += 2;
        }
        if (users != 1 || b <= a) continue;
        Unit& ub = p->units[b];
        if (ub.no_bn || ub.pre_res || ub.rep_fwd <= 0 || ub.cin_pad != ub.d.cin || ub.branch >= 0 || ub.pair >= 0) continue;
        if (!clhip_conv_bn_input_supported(N, ub.H, ub.W, ub.cin_pad, ub.d.cout, ub.d.ksize, ub.d.stride, ub.d.pad, dtype)) continue;
        ua.lazy_to = b; ub.lazy_from = a;
    }
    // the same for the LAST unit of a basic block (conv -> BN -> +res -> ReLU with a packed mask): its first consumer in unit order -- the next
    // block's first convolution, ahead of that block's residual add -- applies it while staging and writes what the apply launch would have
    for (auto& u : p->units) { u.res_lazy_to = u.res_lazy_from = -1; }
    p->res_pending.assign(p->units.size(), 0);
    for (int a = 0; a + 1 < n_units && want_acc && !mask_y; ++a) {
        Unit& ua = p->units[a];
        if (!ua.relu || ua.d.res < 0 || ua.pre_res || ua.no_bn || ua.has_dzr || ua.raw_src || ua.rep_fwd <= 0 || ua.mask_off == 0 || ua.branch >= 0 || ua.forks >= 0 ||
            ua.lazy_to >= 0) continue;
        int b = -1;
        bool ok = true;
        for (int k = 0; k < n_units; ++k) {
            const Unit& o = p->units[k];
            const bool reads = o.d.src == a + 1 || o.d.res == a + 1;
            if (!reads) continue;
            if (k <= a) { ok = false; break; }
            if (b < 0) { if (o.d.src != a + 1 || o.raw_src) { ok = false; break; } b = k; }      // the first reader must be a convolution of the activation
            else if (o.d.src == a + 1 && o.raw_src) { ok = false; break; }
        }
        if (!ok || b < 0) continue;
        Unit& ub = p->units[b];
        if (ub.no_bn || ub.pre_res || ub.rep_fwd <= 0 || ub.cin_pad != ub.d.cin || ub.branch >= 0 || ub.pair >= 0 || ub.lazy_from >= 0) continue;
        if (!clhip_conv_bn_input_supported(N, ub.H, ub.W, ub.cin_pad, ub.d.cout, ub.d.ksize, ub.d.stride, ub.d.pad, dtype)) continue;
        ua.res_lazy_to = b; ub.res_lazy_from = a;
    }
    // the write-through form on the LDS-DMA kernels (conv4 / conv5): unit a = conv -> BN [-> +res] -> ReLU whose FIRST reader in unit order is a
    // convolution b that can transform its landed patch in LDS; every other reader (a later residual add, b's weight gradient, a's backward) finds the
    // activation b's launch wrote.  Producers that start a branch stream with their apply launch keep it.
    for (auto& u : p->units) { u.wt_to = u.wt_from = -1; }
    p->wt_pending.assign(p->units.size(), 0);
    for (int a = 0; a + 1 < n_units && want_acc && !mask_y; ++a) {
        Unit& ua = p->units[a];
        if (!ua.relu || ua.pre_res || ua.no_bn || ua.has_dzr || ua.raw_src || ua.rep_fwd <= 0 || ua.branch >= 0 || ua.forks >= 0 || ua.lazy_to >= 0 || ua.res_lazy_to >= 0) continue;
        if (ua.d.res >= 0 && ua.mask_off == 0) continue;
        int b = -1;
        bool ok = true;
        for (int k = 0; k < n_units; ++k) {
            const Unit& o = p->units[k];
            const bool reads = o.d.src == a + 1 || o.d.res == a + 1;
            if (!reads) continue;
            if (k <= a) { ok = false; break; }
            if (b < 0) { if (o.d.src != a + 1 || o.raw_src) { ok = false; break; } b = k; }
            else if (o.d.src == a + 1 && o.raw_src) { ok = false; break; }
        }
        if (!ok || b < 0) continue;
        Unit& ub = p->units[b];
        if (ub.no_bn || ub.pre_res || ub.rep_fwd <= 0 || ub.cin_pad != ub.d.cin || ub.branch >= 0 || ub.pair >= 0 || ub.lazy_from >= 0 || ub.res_lazy_from >= 0) continue;
        if (!clhip_conv_bn_input_wt_supported(N, ub.H, ub.W, ub.cin_pad, ub.d.cout, ub.d.ksize, ub.d.stride, ub.d.pad, dtype)) continue;
        ua.wt_to = b; ub.wt_from = a;
    }
    // runs of BasicBlocks the eval forward can take as one launch (stage.hip): pairs (a, b) of 3x3 / stride-1 C -> C units, a: conv -> BN -> ReLU on the block input,
    // b: conv -> BN -> + block input -> ReLU on a's output, the next pair on b's output
    for (Unit& u : p->units) u.stage_len = 0;
    for (int i = 0; i + 1 < n_units;) {
        int len = 0;
        const int C = p->units[i].d.cout, Hs = p->units[i].H, Ws = p->units[i].W;
        int s_in = p->units[i].d.src;
        while (i + len + 1 < n_units && len + 2 <= 16) {
            const Unit& a = p->units[i + len];
            const Unit& b = p->units[i + len + 1];
            auto plain3 = [&](const Unit& q) {
                return q.d.ksize == 3 && q.d.stride == 1 && q.d.pad == 1 && q.d.cin == C && q.d.cout == C && q.cin_pad == C && q.H == Hs && q.W == Ws && q.relu && !q.pre_res &&
                       !q.raw_src && !q.no_bn && q.branch < 0;
            };
            if (!(plain3(a) && plain3(b) && a.d.res < 0 && a.d.src == s_in && b.d.src == i + len + 1 && b.d.res == s_in)) break;
            len += 2;
            s_in = i + len;                                   // b's output activation
        }
        if (len >= 2 && clhip_stage_eval_supported(Hs, Ws, C, len, dtype)) { p->units[i].stage_len = len; i += len; }
        else ++i;
    }
    // ... and the runs the TRAINING passes can take as one launch per direction (stage_train.hip): every workgroup (= image) resident at once, the lazy hand-over
    // inside each block and the packed mask of each block output in place (what that forward leaves is what the per-unit backward expects, and vice versa)
    bool any_train = false;
    for (int i = 0; i < n_units; ++i) {
        Unit& u = p->units[i];
        u.stage_train = false;
        if (u.stage_len <= 0 || !clhip_stage_train_supported(N, u.H, u.W, u.d.cout, u.stage_len, dtype)) continue;
        bool ok = want_acc;
        for (int k = 0; k < u.stage_len && ok; ++k) {
            const Unit& q = p->units[i + k];
            // (a unit whose apply launch would start a shortcut-branch stream keeps it: only plans with layers big enough for extra streams have one -- branch_stream_on)
            if (q.rep_fwd <= 0 || q.rep_bwd <= 0 || q.has_dzr || (q.forks >= 0 && p->side_ok) || q.joins >= 0 || q.pair >= 0) ok = false;
            if ((k & 1) == 0 && q.lazy_to != i + k + 1) ok = false;
            if ((k & 1) == 1 && q.mask_off == 0) ok = false;
        }
        u.stage_train = ok;
        any_train = any_train || ok;
        if (getenv("CLHIP_PLAN_DEBUG")) {
            fprintf(stderr, "plan: run at unit %d len %d C %d H %d: stage_train %d\n", i, u.stage_len, u.d.cout, u.H, (int)ok);
            for (int k = 0; k < u.stage_len; ++k) {
                const Unit& q = p->units[i + k];
                fprintf(stderr, "   unit %d: rep %d/%d dzr %d forks %d joins %d pair %d lazy_to %d res_lazy_to %d mask %d\n", i + k, q.rep_fwd, q.rep_bwd, (int)q.has_dzr, q.forks, q.joins,
                        q.pair, q.lazy_to, q.res_lazy_to, (int)(q.mask_off != 0));
            }
        }
    }
    if (getenv("CLHIP_PLAN_DEBUG"))
        for (int i = 0; i < n_units; ++i) fprintf(stderr, "plan: unit %d stage_len %d src %d res %d k %d s %d C %d->%d H %d\n", i, p->units[i].stage_len, p->units[i].d.src, p->units[i].d.res,
                                                  p->units[i].d.ksize, p->units[i].d.stride, p->units[i].d.cin, p->units[i].d.cout, p->units[i].H);
    for (Unit& u : p->units) { u.run_first = -1; u.st_slab = 0; u.entry_first = -1; }
    // the down-sampling block in front of a run: a = 3x3 / s2 (C / 2 -> C) on activation s, b = 1x1 / s2 shortcut on the same activation (conv -> BN, no ReLU), c = the
    // block's second convolution (src a, res b), and the run opens on c's output
    for (int f = 3; f < n_units; ++f) {
        Unit& uf = p->units[f];
        if (!uf.stage_train || uf.d.cout < 32) continue;
        const Unit &a = p->units[f - 3], &b = p->units[f - 2], &c = p->units[f - 1];
        const int C = uf.d.cout;
        // (branch-stream roles -- forks / joins / branch -- only matter in plans that can have extra streams: branch_stream_on)
        auto plain = [&](const Unit& q) { return !q.pre_res && !q.raw_src && !q.no_bn && !q.has_dzr && q.rep_fwd > 0 && q.rep_bwd > 0 && ((q.joins < 0 && q.forks < 0 && q.branch < 0) || !p->side_ok); };
        if (!(plain(a) && plain(b) && plain(c))) continue;
        if (!(a.d.ksize == 3 && a.d.stride == 2 && a.d.pad == 1 && a.d.cin == C / 2 && a.cin_pad == C / 2 && a.d.cout == C && a.relu && a.d.res < 0 && a.H == 2 * uf.H && a.W == 2 * uf.W)) continue;
        if (!(b.d.ksize == 1 && b.d.stride == 2 && b.d.pad == 0 && b.d.src == a.d.src && b.d.cin == C / 2 && b.cin_pad == C / 2 && b.d.cout == C && !b.relu && b.d.res < 0)) continue;
        if (!(c.d.ksize == 3 && c.d.stride == 1 && c.d.pad == 1 && c.d.cin == C && c.d.cout == C && c.relu && c.d.src == f - 2 && c.d.res == f - 1 && c.mask_off != 0 && uf.d.src == f)) continue;
        if (a.d.src < 1 || uf.stage_len + 1 > 16) continue;
        // a's activation must reach c the way the per-unit backward expects it: lazily (its only consumer) -- otherwise the launch writes it
        uf.entry_first = f - 3;
    }
    if (any_train) {
        size_t off2 = p->ws_bytes;
        for (int i = 0; i < n_units; ++i) {
            if (!p->units[i].stage_train) continue;
            for (int k = 0; k < p->units[i].stage_len; ++k) {
                Unit& q = p->units[i + k];
                q.run_first = i;
                q.st_slab = off2;
                off2 = align_up(off2 + (size_t)N * q.d.cout * 9 * q.cin_pad * sizeof(float));
            }
        }
        for (int f = 0; f < n_units; ++f) {                          // ... and of the down-sampling block a launch can take along (one block per image each)
            if (p->units[f].entry_first < 0) continue;
            for (int k = 0; k < 3; ++k) {                           // (3x3 / s2, shortcut, and the block's second convolution)
                Unit& q = p->units[p->units[f].entry_first + k];
                q.st_slab = off2;
                off2 = align_up(off2 + (size_t)N * q.d.cout * q.d.ksize * q.d.ksize * q.cin_pad * sizeof(float));
            }
        }
        p->ws_bytes = off2;
    }
    if (any_train) {
        (void)clhip_stage_train_xcd_rule(true);                      // (once per device: may the exchange take its three-level form?  stage_train.hip)
        const size_t xb = clhip_stage_train_xch_bytes(N);
        if (hipMalloc(&p->xch, xb) != hipSuccess || hipMemset(p->xch, 0, xb) != hipSuccess) {
            (void)hipGetLastError();
            if (p->xch) { (void)hipFree(p->xch); p->xch = nullptr; }
            for (Unit& u : p->units) { u.stage_train = false; u.run_first = -1; }          // (no exchange buffer: the per-unit launches)
        }
    }
    return p;
}

extern "C" void clhip_plan_destroy(clhip_plan* p) {
    if (p && p->side) {
        (void)hipStreamSynchronize(p->side);
        for (int k = 0; k < clhip_plan::kDz; ++k) { (void)hipEventDestroy(p->ev_dz[k]); (void)hipEventDestroy(p->ev_wg[k]); }
        (void)hipEventDestroy(p->ev_end);              // (the stream itself is the process-wide one: clhip_shared_stream)
    }
    if (p && p->br) {
        (void)hipStreamSynchronize(p->br);
        for (int k = 0; k < clhip_plan::kMaxBranch; ++k) { (void)hipEventDestroy(p->ev_fork[k]); (void)hipEventDestroy(p->ev_join[k]); (void)hipEventDestroy(p->ev_bfork[k]); (void)hipEventDestroy(p->ev_bjoin[k]); }
        (void)hipEventDestroy(p->ev_br_end);
    }
    if (p && p->xch) {
        if (p->xch_used) (void)hipDeviceSynchronize();
        (void)hipFree(p->xch);
    }
    delete p;
}
// 0: no stage-level training launch of this plan has run out of its bounded waits (or the plan has none); synchronises the device
// what: 0 = units the plan can run inside stage-level training launches, 1 / 2 = such forward / backward launches so far
extern "C" long long clhip_plan_stage_info(const clhip_plan* p, int what) {
    if (!p) return 0;
    if (what == 1) return p->st_fwd_launches;
    if (what == 2) return p->st_bwd_launches;
    long long n = 0;
    for (const Unit& u : p->units) if (u.stage_train) n += u.stage_len;
    return n;
}
extern "C" int clhip_plan_stage_status(clhip_plan* p) { return (p && p->xch && p->xch_used) ? clhip_stage_train_status(p->xch) : 0; }
extern "C" size_t clhip_plan_workspace_bytes(const clhip_plan* p) { return p ? p->ws_bytes : 0; }
extern "C" size_t clhip_plan_shadow_bytes(const clhip_plan* p) { return p ? p->shadow_bytes : 0; }
extern "C" int clhip_plan_feat_dim(const clhip_plan* p) { return p ? p->feat_dim : 0; }

// PLAN_SKIP (timing ablations only, results invalid): 1 no forward BatchNorm apply, 2 no BatchNorm backward, 4 no weight gradients
static int plan_skip() { static const int v = clhip_cfg("PLAN_SKIP") ? atoi(clhip_cfg("PLAN_SKIP")) : 0; return v; }

#define TRY(call)            \
    do {                     \
        int e_ = (call);     \
        if (e_) return e_;   \
    } while (0)

// the branch stream and its events, created on first use; false inside a stream capture or when extra streams are switched off
static bool branch_stream_on(clhip_plan* p, hipStream_t main_s) {
    // like the weight-gradient stream, only networks with layers big enough for the overlap to beat the host cost of the event calls
    if (p->n_branch == 0 || !p->side_ok) return false;
    static const bool streams_off = clhip_cfg("WGRAD_STREAM") && atoi(clhip_cfg("WGRAD_STREAM")) == 0;
    if (streams_off) return false;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(main_s, &cap);
    if (cap != hipStreamCaptureStatusNone) return false;
    hipStream_t shared_br = clhip_shared_stream(1, main_s, false);        // (looked up per call: the caller's stream may change between calls)
    if (shared_br == nullptr) { p->n_branch = 0; return false; }
    const bool first_br = p->br == nullptr;
    p->br = shared_br;
    if (first_br) {
        static const unsigned ev_flags = clhip_cfg("EVENT_FLAGS") ? (unsigned)strtoul(clhip_cfg("EVENT_FLAGS"), nullptr, 0)
                                                                  : (hipEventDisableTiming | hipEventDisableSystemFence);
        for (int k = 0; k < clhip_plan::kMaxBranch; ++k) {
            (void)hipEventCreateWithFlags(&p->ev_fork[k], ev_flags); (void)hipEventCreateWithFlags(&p->ev_join[k], ev_flags);
            (void)hipEventCreateWithFlags(&p->ev_bfork[k], ev_flags); (void)hipEventCreateWithFlags(&p->ev_bjoin[k], ev_flags);
            p->bfork_ev[k] = nullptr;
        }
        (void)hipEventCreateWithFlags(&p->ev_br_end, ev_flags);
    }
    return true;
}

// the STAGE_TRAIN switch: "0" / "1"; unset = on for batches above 64.  Measured on the EWC CifarResNet-32 step (ms per step, stage-level / per-unit launches,
// profiles/r06_stage_train_notes.md): batch 32 0.76 / 0.75, 64 0.81 / 0.81, 96 0.83 / 0.94, 128 0.85 / 0.91, 192 0.94 / 1.03, 256 0.99 / 1.19 -- a stage-level
// launch costs the same whatever the batch (one image per compute unit: its time is the per-image work plus the in-launch exchanges), the per-unit launches of a
// small batch sit at their latency floor and replay from a HIP graph.
static bool stage_train_default(const char* cfg, int N) { return cfg != nullptr ? atoi(cfg) != 0 : N > 64; }
// STAGE_TRACE = "<channels>:<convolution>" (diagnostic): workgroup 0 of the runs with that channel count stamps the phases of that convolution
static int stage_trace_cfg(int C) {
    const char* v = clhip_cfg("STAGE_TRACE");
    if (v == nullptr) return 0;
    int c = 0, cv = 0;
    if (sscanf(v, "%d:%d", &c, &cv) != 2 || c != C) return 0;
    return cv;
}
extern "C" int clhip_plan_stage_trace(clhip_plan* p, unsigned long long* out24) { return p && p->xch ? clhip_stage_train_trace(p->xch, out24) : CLHIP_EINVAL; }

// Two stage-level training launches must never be in flight on two streams at once: each needs ALL its workgroups resident (they wait for one another), and two
// half-resident grids would wait for each other's compute units until their bounded spins run out.  One plan's launches normally share one stream; when the
// stream changes (outside a capture) the host waits for the previous one first.  (Callers that run a second network beside this one -- ops.TeacherPass --
// switch STAGE_TRAIN off for the side pass; several PROCESSES sharing one GPU must switch it off too: parallel.attach does.)
static int stage_train_serialize(clhip_plan* p, hipStream_t st) {
    static hipStream_t g_last[16] = {};
    static bool g_any[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cap);
    if (dev >= 0 && dev < 16 && cap == hipStreamCaptureStatusNone) {
        if (g_any[dev] && g_last[dev] != st) (void)hipStreamSynchronize(g_last[dev]);
        g_last[dev] = st; g_any[dev] = true;
    }
    p->xch_stream = st; p->xch_used = true;
    return CLHIP_OK;
}

// All conv weights of the backbone in ONE launch (the per-conv launches were 20 x 6.7 us of a 3.5 ms ResNet-18 step):
// the per-conv descriptors travel by value in the kernel arguments, a block finds its conv by a short uniform scan.
namespace {
constexpr int kPrepMax = 48;
struct PrepEntry { int64_t w_off; int64_t wf_off, wd_off; int64_t wp_off; int wp_tap0; int K, taps, Creal, Cpad; unsigned first_block; };
// third copy (wp_off >= 0): the packed dgrad layout of conv6.hip -- per (64-channel tile of C, 16-wide chunk of K) 64 rows x 21 sixteen-byte
// slots, slot 2 * tap + (k % 16) / 8; the shortcut unit of a pair writes its single tap as tap 9 of its partner's buffer
// (fewer than 64 channels: conv7.hip's [C][10][K])
__device__ __forceinline__ size_t packed_index(int c, int tap, int k, int K, int Cpad) {
    if (Cpad < 64) return ((size_t)c * 10 + tap) * K + k;
    return ((((size_t)(c >> 6) * (K >> 4) + (k >> 4)) * 64 + (c & 63)) * 21 + tap * 2 + ((k >> 3) & 1)) * 8 + (k & 7);
}
struct PrepTable { int n; unsigned first[kPrepMax]; PrepEntry e[kPrepMax]; };      // first[u] = e[u].first_block (a block finds its conv with ONE 64-lane load + ballot)
// the entry whose block range holds blockIdx.x.  (The first version walked e[1..].first_block with one dependent scalar load per entry: up to 31 round
// trips before a block's first real load -- 13.8 us for CifarResNet-32's 1.8 MB of weights, most of it this scan.)
__device__ __forceinline__ int prep_entry_of_block(const PrepTable& t) {
    static_assert(kPrepMax <= 64, "one lane per entry");
    const int l = threadIdx.x & 63;
    const bool le = l < t.n && t.first[l < kPrepMax ? l : 0] <= blockIdx.x;
    return __popcll(__ballot(le)) - 1;
}

// block = one (tap, 32 out-channels x 32 in-channels) tile: coalesced fp32 reads along C, coalesced writes of the forward copy
// along C and -- through a 32 x 33 LDS transpose -- of the dgrad copy along K (the first version wrote the dgrad copy with a
// K*taps element stride per lane: 88 us for ResNet-18's 11.2 M weights, ~1 TB/s)
template <typename T>
__global__ __launch_bounds__(256) void weight_prep_multi_kernel(const float* __restrict__ params, char* __restrict__ shadow, PrepTable t) {
    __shared__ float tile[32][33];
    const int u = __builtin_amdgcn_readfirstlane(prep_entry_of_block(t));
    const PrepEntry& d = t.e[u];
    const int kb = (d.K + 31) / 32, cb = (d.Cpad + 31) / 32;
    int r = blockIdx.x - d.first_block;
    const int c0 = (r % cb) * 32; r /= cb;
    const int k0 = (r % kb) * 32;
    const int tap = r / kb;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    T* wf = reinterpret_cast<T*>(shadow + d.wf_off);
    for (int i = ty; i < 32; i += 8) {
        const int k = k0 + i, c = c0 + tx;
        float v = 0.f;
        if (k < d.K && c < d.Cpad) {
            if (c < d.Creal) v = params[d.w_off + ((size_t)k * d.taps + tap) * d.Creal + c];
            Elem<T>::st(wf + ((size_t)k * d.taps + tap) * d.Cpad + c, v);
        }
        tile[i][tx] = v;
    }
    if (d.wd_off < 0) return;
    __syncthreads();
    T* wd = reinterpret_cast<T*>(shadow + d.wd_off);
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, k = k0 + tx;
        if (c < d.Cpad && k < d.K) {
            Elem<T>::st(wd + ((size_t)c * d.taps + tap) * d.K + k, tile[tx][i]);
            if (d.wp_off >= 0) Elem<T>::st(reinterpret_cast<T*>(shadow + d.wp_off) + packed_index(c, d.wp_tap0 + tap, k, d.K, d.Cpad), tile[tx][i]);
        }
    }
}

// bf16 plans with K % 4 == 0 and Cpad % 4 == 0 (every ResNet here): block = one (tap, 64 out-channels x 64 in-channels) tile, four
// consecutive elements per lane -- 8-byte stores of both copies instead of 2-byte ones (the 32 x 32 version above moved ResNet-18's
// 11.2 M weights in 37 us = 2.4 TB/s per training step; this one: profiles/r02_conv4_notes.md)
__global__ __launch_bounds__(256) void weight_prep_multi64_kernel(const float* __restrict__ params, char* __restrict__ shadow, PrepTable t) {
    __shared__ float tile[64][65];
    const int u = __builtin_amdgcn_readfirstlane(prep_entry_of_block(t));
    const PrepEntry& d = t.e[u];
    const int kb = (d.K + 63) / 64, cb = (d.Cpad + 63) / 64;
    int r = blockIdx.x - d.first_block;
    const int c0 = (r % cb) * 64; r /= cb;
    const int k0 = (r % kb) * 64;
    const int tap = r / kb;
    const int tx = (threadIdx.x & 15) * 4, ty = threadIdx.x >> 4;
    bf16_t* wf = reinterpret_cast<bf16_t*>(shadow + d.wf_off);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int kk = ty + 16 * i, k = k0 + kk, c = c0 + tx;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (k < d.K && c < d.Cpad) {
            const float* src = params + d.w_off + ((size_t)k * d.taps + tap) * d.Creal + c;
#pragma unroll
            for (int e = 0; e < 4; ++e) if (c + e < d.Creal) v[e] = src[e];
            *reinterpret_cast<uint2*>(wf + ((size_t)k * d.taps + tap) * d.Cpad + c) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[kk][tx + e] = v[e];
    }
    if (d.wd_off < 0) return;
    __syncthreads();
    bf16_t* wd = reinterpret_cast<bf16_t*>(shadow + d.wd_off);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cc = ty + 16 * i, c = c0 + cc, k = k0 + tx;
        if (c < d.Cpad && k < d.K) {
            const uint2 v = make_uint2(pack_bf16x2(tile[tx][cc], tile[tx + 1][cc]), pack_bf16x2(tile[tx + 2][cc], tile[tx + 3][cc]));
            *reinterpret_cast<uint2*>(wd + ((size_t)c * d.taps + tap) * d.K + k) = v;
            if (d.wp_off >= 0) *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(shadow + d.wp_off) + packed_index(c, d.wp_tap0 + tap, k, d.K, d.Cpad)) = v;
        }
    }
}
}  // namespace

extern "C" int clhip_plan_prep_weights(clhip_plan* p, const float* params, void* shadow, void* stream) {
    CLHIP_CHECK_ARG(p && params && shadow);
    char* sh = static_cast<char*>(shadow);
    size_t i = 0;
    while (i < p->units.size()) {
        PrepTable t;
        t.n = 0;
        unsigned blocks = 0;
        bool wide = p->dtype == CLHIP_BF16;
        static const bool no_wide = clhip_cfg("PREP_NARROW") != nullptr;
        for (size_t j = i; j < p->units.size() && j < i + kPrepMax; ++j)
            wide = wide && !no_wide && p->units[j].d.cout % 4 == 0 && p->units[j].cin_pad % 4 == 0 && p->units[j].sh_fwd % 8 == 0 && p->units[j].sh_dg % 8 == 0;
        const int tb = wide ? 64 : 32;
        for (; i < p->units.size() && t.n < kPrepMax; ++i) {
            const Unit& u = p->units[i];
            PrepEntry& e = t.e[t.n++];
            e.w_off = u.d.w_off; e.wf_off = (int64_t)u.sh_fwd; e.wd_off = u.d.src != 0 ? (int64_t)u.sh_dg : -1;
            e.wp_off = -1; e.wp_tap0 = 0;
            if (u.pair >= 0) {                        // the 3x3 unit owns the packed buffer, the shortcut unit adds its tap
                const bool owner = u.d.ksize == 3;
                e.wp_off = (int64_t)(owner ? u.sh_pk : p->units[u.pair].sh_pk);
                e.wp_tap0 = owner ? 0 : 9;
            }
            e.K = u.d.cout; e.taps = u.d.ksize * u.d.ksize; e.Creal = u.d.cin; e.Cpad = u.cin_pad;
            e.first_block = blocks; t.first[t.n - 1] = blocks;
            blocks += (unsigned)(e.taps * ((e.K + tb - 1) / tb) * ((e.Cpad + tb - 1) / tb));
        }
        if (wide) hipLaunchKernelGGL(weight_prep_multi64_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, sh, t);
        else if (p->dtype == CLHIP_BF16) hipLaunchKernelGGL(weight_prep_multi_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, sh, t);
        else hipLaunchKernelGGL(weight_prep_multi_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, sh, t);
        CLHIP_LAUNCH_CHECK();
    }
    return CLHIP_OK;
}

// The first launch of a forward: NCHW fp32 -> NHWC compute dtype (channels zero-padded) and, in the same grid, the two pieces of
// per-step housekeeping that used to be launches of their own -- zeroing the fp64 BatchNorm accumulators (was a hipMemsetAsync =
// a fillBufferAligned kernel per step) and `num_batches_tracked += 1` of every BatchNorm (was a torch add<long> kernel per step).
namespace {
template <typename T>
__global__ __launch_bounds__(256) void plan_prologue_kernel(const float* __restrict__ x, T* __restrict__ y, int N, int C, int HW, int Cpad, unsigned layout_blocks,
                                                            uint4* __restrict__ acc, size_t acc_n16, long long* __restrict__ nbt, unsigned long long nbt_mask_lo,
                                                            unsigned long long nbt_mask_hi, int n_units) {
    if (blockIdx.x >= layout_blocks) {
        const size_t i0 = (size_t)(blockIdx.x - layout_blocks) * 256 * 4 + threadIdx.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const size_t i = i0 + (size_t)k * 256; if (i < acc_n16) acc[i] = make_uint4(0, 0, 0, 0); }
        return;
    }
    if (nbt != nullptr && blockIdx.x == 0)
        for (int u = threadIdx.x; u < n_units; u += 256)
            if ((u < 64 ? (nbt_mask_lo >> u) : (nbt_mask_hi >> (u - 64))) & 1ull) nbt[u] += 1;
    const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pix >= (int64_t)N * HW) return;
    const int n = (int)(pix / HW), hw = (int)(pix - (int64_t)n * HW);
    for (int c0 = 0; c0 < Cpad; c0 += 8) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (c0 + e < C) ? x[((size_t)n * C + c0 + e) * HW + hw] : 0.f;
        store8<T>(y + pix * Cpad + c0, v);
    }
}
}  // namespace

extern "C" int clhip_plan_forward(clhip_plan* p, const float* x, const float* params, float* bn_stats, const void* shadow,
                                  void* workspace, float* feat, int training, void* stream) {
    return clhip_plan_forward_ex(p, x, params, bn_stats, shadow, workspace, feat, training, nullptr, stream);
}

extern "C" int clhip_plan_forward_ex(clhip_plan* p, const float* x, const float* params, float* bn_stats, const void* shadow,
                                     void* workspace, float* feat, int training, int64_t* num_batches_tracked, void* stream) {
    CLHIP_CHECK_ARG(p && x && params && bn_stats && shadow && workspace && feat);
    CLHIP_CHECK_ARG(num_batches_tracked == nullptr || p->units.size() <= 128);
    char* ws = static_cast<char*>(workspace);
    const char* sh = static_cast<const char*>(shadow);
    float* fr = reinterpret_cast<float*>(ws + p->f_base);
    double* acc = reinterpret_cast<double*>(ws + p->acc_off);
    const bool use_acc = training && p->use_acc;
    std::fill(p->bwd_sums_ready.begin(), p->bwd_sums_ready.end(), 0);
    {
        const int64_t npix = (int64_t)p->N * p->H * p->W;
        const unsigned lb = (unsigned)((npix + 255) / 256);
        const size_t n16 = use_acc ? p->acc_bytes / 16 : 0;                       // acc_bytes is a multiple of 16 (2 * cout doubles per replica, cout % 16 == 0)
        const unsigned ab = (unsigned)((n16 + 1023) / 1024);
        unsigned long long mlo = 0, mhi = 0;                                      // units that own a BatchNorm (the counter array has one slot per unit)
        for (size_t i = 0; i < p->units.size() && i < 128; ++i)
            if (!p->units[i].no_bn) { if (i < 64) mlo |= 1ull << i; else mhi |= 1ull << (i - 64); }
        long long* nbt = training ? reinterpret_cast<long long*>(num_batches_tracked) : nullptr;
        uint4* a16 = reinterpret_cast<uint4*>(acc);
        if (p->dtype == CLHIP_BF16)
            hipLaunchKernelGGL(plan_prologue_kernel<bf16_t>, dim3(lb + ab), dim3(256), 0, (hipStream_t)stream, x, reinterpret_cast<bf16_t*>(ws + p->acts[0].y_off), p->N, p->Cin,
                               p->H * p->W, p->Cin_pad, lb, a16, n16, nbt, mlo, mhi, (int)p->units.size());
        else
            hipLaunchKernelGGL(plan_prologue_kernel<float>, dim3(lb + ab), dim3(256), 0, (hipStream_t)stream, x, reinterpret_cast<float*>(ws + p->acts[0].y_off), p->N, p->Cin,
                               p->H * p->W, p->Cin_pad, lb, a16, n16, nbt, mlo, mhi, (int)p->units.size());
        CLHIP_LAUNCH_CHECK();
    }
    // shortcut branches run on their own stream in the training forward (accumulator path only)
    // BRANCH_STREAM: 0 off, 1 both directions, 2 forward only (default), 3 backward only.  Measured on the ResNet-18 step at batch 256 (two
    // alternating runs each on one box, ms per step): off 2.157 / 2.153, forward only 2.148 / 2.144, both 2.160 / 2.163, backward only 2.197 /
    // 2.205 -- in the backward the chip is already shared with the weight-gradient stream and a third stream only adds contention and
    // barrier packets; in the forward nothing else runs beside the chain (profiles/r03_step_notes.md)
    // (looked up per call: a caller that runs a second network beside this one -- ops.TeacherPass: the frozen teacher on a stream of its own -- switches
    //  the branch streams of both off for that step; five streams in one step made the LwF ResNet-18 task >= 1 step 6.36 ms instead of 2.60, profiles/r03_step_notes.md)
    const char* br_cfg_f = clhip_cfg("BRANCH_STREAM");
    const int br_mode_f = br_cfg_f ? atoi(br_cfg_f) : 2;
    const bool br_on = use_acc && (br_mode_f == 1 || br_mode_f == 2) && branch_stream_on(p, (hipStream_t)stream);
    struct FwdStopGuard { ~FwdStopGuard() { clhip_bn_set_fwd_stop_event(nullptr); } } fwd_stop_guard;
    const char* lazy_cfg = clhip_cfg("BN_INPUT");            // (looked up per call: the tests flip it between two models of one process)
    const bool lazy_env = !(lazy_cfg != nullptr && atoi(lazy_cfg) == 0);
    const bool lazy_on = lazy_env && training && use_acc;
    const char* rlazy_cfg = clhip_cfg("BN_RES_INPUT");
    const bool rlazy_on = lazy_on && !(rlazy_cfg != nullptr && atoi(rlazy_cfg) == 0);
    const char* wt_cfg = clhip_cfg("BN_INPUT_WT");
    const bool wt_on = training && use_acc && wt_cfg != nullptr && atoi(wt_cfg) != 0;          // (off by default: conv.hip clhip_conv_bn_input_wt_supported)
    // EVAL_LAZY (default on; looked up per call): the eval-mode forward of a bf16 plan uses the consumer-side BatchNorm forms too
    const char* elazy_cfg = clhip_cfg("EVAL_LAZY");
    const bool eval_lazy = !training && p->use_acc && p->dtype == CLHIP_BF16 && lazy_env && !(rlazy_cfg != nullptr && atoi(rlazy_cfg) == 0) &&
                           !(elazy_cfg != nullptr && atoi(elazy_cfg) == 0);
    const char* stage_cfg = clhip_cfg("STAGE_EVAL");                 // (looked up per call, like EVAL_LAZY: the tests compare the two forms in one process)
    const bool stage_on = eval_lazy && !(stage_cfg != nullptr && atoi(stage_cfg) == 0);
    // STAGE_TRAIN (default on; looked up per call): runs of BasicBlocks as ONE training launch (stage_train.hip) where the plan found them and the lazy forms are on
    const char* strain_cfg = clhip_cfg("STAGE_TRAIN");
    const bool strain_on = rlazy_on && p->xch != nullptr && stage_train_default(strain_cfg, p->N);
    // training == 2: batch-statistics forward that no backward will follow (a teacher under torch.no_grad()): the stage-level launches skip the z / activation
    // stores of everything inside a run (per convolution 32 KB per image written for nobody); the per-unit launches ignore the hint
    const bool nosave = training == 2;
    // ... and the launch of the LAST run also averages its output (the global pooling behind it): STAGE_POOL=0 keeps the pooling launch
    const char* spool_cfg = clhip_cfg("STAGE_POOL");
    const bool spool_on = !(spool_cfg != nullptr && atoi(spool_cfg) == 0) && p->pool_win == 0;
    auto pool_in_run = [&](size_t end, int C) { return spool_on && end == p->units.size() && p->acts.back().C == C; };
    bool pooled = false;
    for (size_t i = 0; i < p->units.size(); ++i) p->lazy_live[i] = p->res_pending[i] = p->wt_pending[i] = p->eval_unwritten[i] = p->stage_skipped[i] = 0;
    if (training) std::fill(p->mask_stale.begin(), p->mask_stale.end(), 0);
    p->params_dev = params; p->bn_stats_dev = bn_stats;
    int fwd_pair_done = -1;                                  // the 3x3/s2 unit whose launch also ran its shortcut partner's convolution
    for (size_t i = 0; i < p->units.size(); ++i) {
        const Unit& u = p->units[i];
        const Act& src = p->acts[u.d.src];
        const Act& dst = p->acts[i + 1];
        const char* in = u.raw_src ? ws + p->units[u.d.src - 1].z_off : ws + src.y_off;
        if (u.no_bn || u.pre_res) {
            TRY(clhip_conv_fwd(in, sh + u.sh_fwd, ws + u.z_off, nullptr, p->N, u.H, u.W, u.cin_pad, u.d.cout, u.d.ksize, u.d.stride, u.d.pad, p->dtype, stream));
            if (u.no_bn) continue;
            TRY(clhip_add_stats(ws + u.z_off, ws + p->units[u.d.res - 1].z_off, training ? acc + u.a_fwd : nullptr, u.rep_fwd, u.M, u.d.cout, p->dtype, stream));
            if (training) {
                TRY(clhip_bn_apply_train(ws + u.z_off, acc + u.a_fwd, u.rep_fwd, u.M, u.d.cout, params + u.d.gamma_off, params + u.d.beta_off,
                                         bn_stats + u.d.rm_off, bn_stats + u.d.rv_off, kBnMomentum, kBnEps, fr + u.f_mean, fr + u.f_invstd, nullptr,
                                         ws + dst.y_off, u.relu, p->dtype, stream));
            } else {
                TRY(clhip_bn_apply_eval(ws + u.z_off, params + u.d.gamma_off, params + u.d.beta_off, bn_stats + u.d.rm_off, bn_stats + u.d.rv_off, kBnEps, nullptr,
                                        ws + dst.y_off, u.M, u.d.cout, u.relu, p->dtype, stream));
            }
            continue;
        }
        if (use_acc && u.rep_fwd > 0) {
            // conv epilogue adds the per-channel sums into the fp64 accumulator; BN-apply derives scale / shift on the fly
            static const bool entry_off = clhip_cfg("STAGE_ENTRY") != nullptr && atoi(clhip_cfg("STAGE_ENTRY")) == 0;
            if (strain_on && !entry_off && i + 3 < p->units.size() && p->units[i + 3].stage_train && p->units[i + 3].entry_first == (int)i) {
                // the stage's down-sampling block AND the run behind it as one launch (stage_train.hip, ENTRY): units i (3x3 / s2), i + 1 (shortcut), i + 2, then the run
                const Unit& uf = p->units[i + 3];
                const int len = uf.stage_len + 3;
                const void* wv[20]; const float *gv[20], *bv[20]; float *rmv[20], *rvv[20], *mev[20], *isv[20], *cov[20]; void *zv[20], *yv[20], *mkv[20];
                for (int k = 0; k < len; ++k) {
                    const Unit& q = p->units[i + k];
                    wv[k] = sh + q.sh_fwd; gv[k] = params + q.d.gamma_off; bv[k] = params + q.d.beta_off; rmv[k] = bn_stats + q.d.rm_off; rvv[k] = bn_stats + q.d.rv_off;
                    mev[k] = fr + q.f_mean; isv[k] = fr + q.f_invstd; cov[k] = fr + q.f_scale; zv[k] = ws + q.z_off; mkv[k] = nullptr;
                    // block outputs and the shortcut's BatchNorm output are written; a first convolution's activation only where its consumer does not take it lazily
                    const bool first = k == 0 || (k >= 3 && ((k - 3) & 1) == 0);
                    const bool lazy = first && q.lazy_to == (int)i + k + (k == 0 ? 2 : 1);
                    yv[k] = lazy ? nullptr : ws + p->acts[i + k + 1].y_off;
                    // (no-save: only what a LATER launch of this forward reads survives -- the launch's output, unless it is the network's and the pooling rides along)
                    if (nosave) { zv[k] = nullptr; if (!(k == len - 1 && !pool_in_run(i + len, uf.d.cout))) yv[k] = nullptr; }
                    p->lazy_live[i + k] = lazy ? 1 : 0;
                    p->mask_stale[i + k] = (k == 2 || (k >= 3 && ((k - 3) & 1) == 1)) ? 1 : 0;
                }
                TRY(stage_train_serialize(p, (hipStream_t)stream));
                TRY(clhip_stage_train_fwd_launch(ws + src.y_off, p->N, uf.H, uf.W, uf.d.cout, uf.stage_len + 1, wv, gv, bv, rmv, rvv, mev, isv, cov, zv, yv, mkv, kBnMomentum, kBnEps,
                                                 p->xch, stage_trace_cfg(uf.d.cout), 1, pool_in_run(i + len, uf.d.cout) ? feat : nullptr, p->dtype, (hipStream_t)stream));
                ++p->st_fwd_launches;
                if (pool_in_run(i + len, uf.d.cout)) pooled = true;
                i += len - 1;
                continue;
            }
            if (strain_on && u.stage_train) {
                // a run of BasicBlocks as ONE launch: one workgroup per image, the activation resident in LDS, the batch statistics of every convolution through the
                // in-launch all-reduce (stage_train.hip).  Leaves what the per-unit launches leave: z, saved + running statistics, the block outputs with their masks;
                // the activation between a block's two convolutions stays lazy (the backward recomputes it from z)
                const void* wv[16]; const float *gv[16], *bv[16]; float *rmv[16], *rvv[16], *mev[16], *isv[16], *cov[16]; void *zv[16], *yv[16], *mkv[16];
                for (int k = 0; k < u.stage_len; ++k) {
                    const Unit& q = p->units[i + k];
                    wv[k] = sh + q.sh_fwd; gv[k] = params + q.d.gamma_off; bv[k] = params + q.d.beta_off; rmv[k] = bn_stats + q.d.rm_off; rvv[k] = bn_stats + q.d.rv_off;
                    mev[k] = fr + q.f_mean; isv[k] = fr + q.f_invstd; cov[k] = fr + q.f_scale; zv[k] = ws + q.z_off;
                    yv[k] = (k & 1) ? ws + p->acts[i + k + 1].y_off : nullptr; mkv[k] = nullptr;
                    // (no-save: only what a LATER launch of this forward reads survives -- the run's output, unless the pooling rides in this launch too)
                    if (nosave) { zv[k] = nullptr; if (!(k == u.stage_len - 1 && !pool_in_run(i + u.stage_len, u.d.cout))) yv[k] = nullptr; }
                    p->lazy_live[i + k] = (k & 1) ? 0 : 1;
                    p->mask_stale[i + k] = (k & 1) ? 1 : 0;
                }
                TRY(stage_train_serialize(p, (hipStream_t)stream));
                TRY(clhip_stage_train_fwd_launch(ws + src.y_off, p->N, u.H, u.W, u.d.cout, u.stage_len, wv, gv, bv, rmv, rvv, mev, isv, cov, zv, yv, mkv, kBnMomentum, kBnEps,
                                                 p->xch, stage_trace_cfg(u.d.cout), 0, pool_in_run(i + u.stage_len, u.d.cout) ? feat : nullptr, p->dtype, (hipStream_t)stream));
                ++p->st_fwd_launches;
                if (pool_in_run(i + u.stage_len, u.d.cout)) pooled = true;
                i += u.stage_len - 1;
                continue;
            }
            void* us = stream;                                   // the stream this unit's two launches go to
            const bool on_br = br_on && u.branch >= 0;
            if (on_br) {                                         // shortcut branch: starts when its input activation is complete (ev_fork)
                (void)hipStreamWaitEvent(p->br, p->ev_fork[u.branch], 0);
                us = p->br;
            }
            if (u.wt_from >= 0 && p->wt_pending[u.wt_from]) {
                // the producer's BatchNorm [+ residual] + ReLU happen on this convolution's landed patch (in LDS), and this launch writes the activation
                // [+ packed mask] for the readers that follow; mean / invstd / coefficients and the running statistics are its by-products
                const Unit& a = p->units[u.wt_from];
                clhip_bn_input bi;
                bi.stat_acc = acc + a.a_fwd; bi.replicas = a.rep_fwd; bi.gamma = params + a.d.gamma_off; bi.beta = params + a.d.beta_off;
                bi.running_mean = bn_stats + a.d.rm_off; bi.running_var = bn_stats + a.d.rv_off; bi.momentum = kBnMomentum; bi.eps = kBnEps;
                bi.mean = fr + a.f_mean; bi.invstd = fr + a.f_invstd; bi.coef = fr + a.f_scale;
                clhip_bn_res_input rs;
                rs.res = a.d.res >= 0 ? ws + p->acts[a.d.res].y_off : nullptr; rs.y = ws + src.y_off; rs.relu_mask = a.mask_off != 0 ? ws + a.mask_off : nullptr;
                if (br_on && a.joins >= 0) (void)hipStreamWaitEvent((hipStream_t)us, p->ev_join[a.joins], 0);      // the residual comes from the branch stream
                TRY(clhip_conv_fwd_acc_bn_input_wt(ws + a.z_off, &bi, &rs, sh + u.sh_fwd, ws + u.z_off, acc + u.a_fwd, u.rep_fwd, p->N, u.H, u.W, u.cin_pad, u.d.cout,
                                                   u.d.ksize, u.d.stride, u.d.pad, p->dtype, us));
                p->wt_pending[u.wt_from] = 0;
            } else if (u.res_lazy_from >= 0 && p->res_pending[u.res_lazy_from]) {
                // the producer is a block's last unit: its BatchNorm + residual add + ReLU happen on this convolution's operand load, and this launch
                // writes the activation and the packed mask for the readers that follow
                const Unit& a = p->units[u.res_lazy_from];
                clhip_bn_input bi;
                bi.stat_acc = acc + a.a_fwd; bi.replicas = a.rep_fwd; bi.gamma = params + a.d.gamma_off; bi.beta = params + a.d.beta_off;
                bi.running_mean = bn_stats + a.d.rm_off; bi.running_var = bn_stats + a.d.rv_off; bi.momentum = kBnMomentum; bi.eps = kBnEps;
                bi.mean = fr + a.f_mean; bi.invstd = fr + a.f_invstd; bi.coef = fr + a.f_scale;
                clhip_bn_res_input rs;
                rs.res = ws + p->acts[a.d.res].y_off; rs.y = ws + src.y_off; rs.relu_mask = ws + a.mask_off;
                if (br_on && a.joins >= 0) (void)hipStreamWaitEvent((hipStream_t)us, p->ev_join[a.joins], 0);      // the residual comes from the branch stream
                TRY(clhip_conv_fwd_acc_bn_res_input(ws + a.z_off, &bi, &rs, sh + u.sh_fwd, ws + u.z_off, acc + u.a_fwd, u.rep_fwd, p->N, u.H, u.W, u.cin_pad,
                                                    u.d.cout, u.d.ksize, u.d.stride, u.d.pad, p->dtype, us));
                p->res_pending[u.res_lazy_from] = 0;
            } else if (u.fpair && u.d.ksize == 1 && fwd_pair_done == u.pair) {
                // z and the statistics of this shortcut came out of its 3x3 partner's launch
            } else if (u.fpair && u.d.ksize == 3 && !br_on && !on_br && !u.raw_src) {
                const Unit& sc = p->units[u.pair];
                TRY(clhip_conv_fwd_acc_pair(in, sh + u.sh_fwd, sh + sc.sh_fwd, ws + u.z_off, ws + sc.z_off, acc + u.a_fwd, u.rep_fwd, acc + sc.a_fwd, sc.rep_fwd, p->N, u.H,
                                            u.W, u.cin_pad, u.d.cout, p->dtype, us));
                fwd_pair_done = (int)i;
            } else if (u.lazy_from >= 0 && p->lazy_live[u.lazy_from]) {
                // the producer's BatchNorm + ReLU happen on this convolution's operand load: scale / shift, the saved statistics and the running
                // statistics of the producer are this launch's by-products
                const Unit& a = p->units[u.lazy_from];
                clhip_bn_input bi;
                bi.stat_acc = acc + a.a_fwd; bi.replicas = a.rep_fwd; bi.gamma = params + a.d.gamma_off; bi.beta = params + a.d.beta_off;
                bi.running_mean = bn_stats + a.d.rm_off; bi.running_var = bn_stats + a.d.rv_off; bi.momentum = kBnMomentum; bi.eps = kBnEps;
                bi.mean = fr + a.f_mean; bi.invstd = fr + a.f_invstd; bi.coef = fr + a.f_scale;
                TRY(clhip_conv_fwd_acc_bn_input(ws + a.z_off, &bi, sh + u.sh_fwd, ws + u.z_off, acc + u.a_fwd, u.rep_fwd, p->N, u.H, u.W, u.cin_pad, u.d.cout,
                                                u.d.ksize, u.d.stride, u.d.pad, p->dtype, us));
            } else {
                TRY(clhip_conv_fwd_acc(in, sh + u.sh_fwd, ws + u.z_off, acc + u.a_fwd, u.rep_fwd, p->N, u.H, u.W, u.cin_pad, u.d.cout, u.d.ksize,
                                       u.d.stride, u.d.pad, p->dtype, us));
            }
            p->lazy_live[i] = 0;
            // (a fused training run reads a WRITTEN activation: its producer keeps the apply launch)
            auto opens_run = [&](int k) { return strain_on && p->units[k].stage_train; };
            if (wt_on && u.wt_to >= 0 && !opens_run(u.wt_to)) { p->wt_pending[i] = 1; continue; }              // its first reader applies it in LDS and writes the activation [+ mask]
            if (lazy_on && u.lazy_to >= 0 && !opens_run(u.lazy_to)) { p->lazy_live[i] = 1; continue; }          // its one consumer applies the BatchNorm: no apply launch, no activation
            if (rlazy_on && u.res_lazy_to >= 0 && !opens_run(u.res_lazy_to)) { p->res_pending[i] = 1; continue; }   // its first consumer applies it and writes the activation + mask
            const void* res_ = u.d.res >= 0 ? ws + p->acts[u.d.res].y_off : nullptr;
            if (plan_skip() & 1) continue;                       // timing ablation: no forward BatchNorm apply (results invalid)
            if (br_on && u.joins >= 0) (void)hipStreamWaitEvent((hipStream_t)stream, p->ev_join[u.joins], 0);      // the residual comes from the branch stream
            if (br_on && u.forks >= 0) clhip_bn_set_fwd_stop_event(p->ev_fork[u.forks]);                           // this launch's completion starts the branch
            if (u.mask_off != 0)
                TRY(clhip_bn_apply_train_mask(ws + u.z_off, acc + u.a_fwd, u.rep_fwd, u.M, u.d.cout, params + u.d.gamma_off, params + u.d.beta_off,
                                              bn_stats + u.d.rm_off, bn_stats + u.d.rv_off, kBnMomentum, kBnEps, fr + u.f_mean, fr + u.f_invstd, res_,
                                              ws + dst.y_off, ws + u.mask_off, p->dtype, us));
            else
                TRY(clhip_bn_apply_train(ws + u.z_off, acc + u.a_fwd, u.rep_fwd, u.M, u.d.cout, params + u.d.gamma_off, params + u.d.beta_off,
                                         bn_stats + u.d.rm_off, bn_stats + u.d.rv_off, kBnMomentum, kBnEps, fr + u.f_mean, fr + u.f_invstd, res_,
                                         ws + dst.y_off, u.relu, p->dtype, us));
            if (on_br) (void)hipEventRecord(p->ev_join[u.branch], p->br);
            continue;
        }
        float* part = training ? fr + p->f_part : nullptr;
        if (eval_lazy) {
            // eval-mode forward (frozen teacher of LwF / iCaRL / LUCIR, validation, feature extraction for herding / NCM): the SAME consumer-side forms as the
            // training forward, with scale / shift of the running statistics -- relu(bn(z)) [+ res] is applied by the next convolution's operand staging and the
            // apply launch of every paired unit disappears (CifarResNet-32: 62 launches -> 33).  Bit-identical to the apply launches (same fp32 expressions).
            auto eval_bi = [&](const Unit& a) {
                clhip_bn_input bi;
                bi.stat_acc = nullptr; bi.replicas = 1; bi.gamma = params + a.d.gamma_off; bi.beta = params + a.d.beta_off;
                bi.running_mean = bn_stats + a.d.rm_off; bi.running_var = bn_stats + a.d.rv_off; bi.momentum = 0.f; bi.eps = kBnEps;
                bi.mean = nullptr; bi.invstd = nullptr; bi.coef = nullptr;
                return bi;
            };
            if (stage_on && u.stage_len > 0) {
                // a run of BasicBlocks as ONE launch, the image resident in LDS (stage.hip); its input activation was written (the producers' lazy forms are
                // switched off below for units that open a run), its inner activations are not
                const void* wv[16]; const float *gv[16], *bv[16], *mv[16], *vv[16];
                for (int k = 0; k < u.stage_len; ++k) {
                    const Unit& q = p->units[i + k];
                    wv[k] = sh + q.sh_fwd; gv[k] = params + q.d.gamma_off; bv[k] = params + q.d.beta_off; mv[k] = bn_stats + q.d.rm_off; vv[k] = bn_stats + q.d.rv_off;
                    if (k + 1 < u.stage_len) p->stage_skipped[i + k] = 1;
                }
                TRY(clhip_stage_eval_launch(ws + src.y_off, ws + p->acts[i + u.stage_len].y_off, p->N, u.H, u.W, u.d.cout, u.stage_len, wv, gv, bv, mv, vv, kBnEps, p->dtype,
                                            (hipStream_t)stream));
                i += u.stage_len - 1;
                continue;
            }
            if (u.res_lazy_from >= 0 && p->res_pending[u.res_lazy_from]) {
                const Unit& a = p->units[u.res_lazy_from];
                const clhip_bn_input bi = eval_bi(a);
                clhip_bn_res_input rs;
                rs.res = ws + p->acts[a.d.res].y_off; rs.y = ws + src.y_off; rs.relu_mask = nullptr;
                TRY(clhip_conv_fwd_acc_bn_res_input(ws + a.z_off, &bi, &rs, sh + u.sh_fwd, ws + u.z_off, nullptr, 1, p->N, u.H, u.W, u.cin_pad, u.d.cout, u.d.ksize,
                                                    u.d.stride, u.d.pad, p->dtype, stream));
                p->res_pending[u.res_lazy_from] = 0;
            } else if (u.lazy_from >= 0 && p->lazy_live[u.lazy_from]) {
                const Unit& a = p->units[u.lazy_from];
                const clhip_bn_input bi = eval_bi(a);
                TRY(clhip_conv_fwd_acc_bn_input(ws + a.z_off, &bi, sh + u.sh_fwd, ws + u.z_off, nullptr, 1, p->N, u.H, u.W, u.cin_pad, u.d.cout, u.d.ksize, u.d.stride,
                                                u.d.pad, p->dtype, stream));
                p->lazy_live[u.lazy_from] = 0;
                p->eval_unwritten[u.lazy_from] = 1;       // clhip_plan_read_act materialises it on demand
            } else {
                TRY(clhip_conv_fwd(in, sh + u.sh_fwd, ws + u.z_off, nullptr, p->N, u.H, u.W, u.cin_pad, u.d.cout, u.d.ksize, u.d.stride, u.d.pad, p->dtype, stream));
            }
            if (u.lazy_to >= 0 && !(stage_on && p->units[u.lazy_to].stage_len > 0)) { p->lazy_live[i] = 1; continue; }          // (a fused run reads a WRITTEN activation)
            if (u.res_lazy_to >= 0 && !(stage_on && p->units[u.res_lazy_to].stage_len > 0)) { p->res_pending[i] = 1; continue; }
            const void* res_e = u.d.res >= 0 ? ws + p->acts[u.d.res].y_off : nullptr;
            TRY(clhip_bn_apply_eval(ws + u.z_off, params + u.d.gamma_off, params + u.d.beta_off, bn_stats + u.d.rm_off, bn_stats + u.d.rv_off, kBnEps, res_e, ws + dst.y_off,
                                    u.M, u.d.cout, u.relu, p->dtype, stream));
            continue;
        }
        TRY(clhip_conv_fwd(in, sh + u.sh_fwd, ws + u.z_off, part, p->N, u.H, u.W, u.cin_pad, u.d.cout, u.d.ksize,
                           u.d.stride, u.d.pad, p->dtype, stream));
        if (training) {
            TRY(clhip_bn_stats_finalize(part, u.tiles, u.M, u.d.cout, params + u.d.gamma_off, params + u.d.beta_off,
                                        bn_stats + u.d.rm_off, bn_stats + u.d.rv_off, kBnMomentum, kBnEps, fr + u.f_mean,
                                        fr + u.f_invstd, fr + u.f_scale, fr + u.f_shift, stream));
        }
        const void* res = u.d.res >= 0 ? ws + p->acts[u.d.res].y_off : nullptr;
        if (!training) {
            // eval mode: scale / shift of the running statistics are derived inside the apply launch (one launch per unit instead of two)
            TRY(clhip_bn_apply_eval(ws + u.z_off, params + u.d.gamma_off, params + u.d.beta_off, bn_stats + u.d.rm_off, bn_stats + u.d.rv_off, kBnEps, res, ws + dst.y_off,
                                    u.M, u.d.cout, u.relu, p->dtype, stream));
            continue;
        }
        TRY(clhip_bn_apply(ws + u.z_off, fr + u.f_scale, fr + u.f_shift, res, ws + dst.y_off, u.M, u.d.cout, u.relu, p->dtype, stream));
    }
    const Act& last = p->acts.back();
    if (pooled) return CLHIP_OK;                              // (the last run's launch averaged its output on the way out)
    if (p->pool_win > 0) TRY(clhip_avgpool_win_fwd(ws + last.y_off, feat, p->N, last.H, last.W, last.C, p->pool_win, p->dtype, stream));
    else TRY(clhip_avgpool_fwd(ws + last.y_off, feat, p->N, last.H * last.W, last.C, p->dtype, stream));
    return CLHIP_OK;
}

extern "C" int clhip_plan_num_units(const clhip_plan* p) { return p ? (int)p->units.size() : 0; }

extern "C" int clhip_plan_backward(clhip_plan* p, const float* dfeat, const float* params, const void* shadow, void* workspace,
                                   float* grads, void* stream) {
    CLHIP_CHECK_ARG(p);
    return clhip_plan_backward_range(p, dfeat, params, shadow, workspace, grads, (int)p->units.size(), 0, stream);
}

extern "C" int clhip_plan_backward_range(clhip_plan* p, const float* dfeat, const float* params, const void* shadow, void* workspace,
                                         float* grads, int unit_hi, int unit_lo, void* stream) {
    CLHIP_CHECK_ARG(p && dfeat && params && shadow && workspace && grads);
    CLHIP_CHECK_ARG(0 <= unit_lo && unit_lo < unit_hi && unit_hi <= (int)p->units.size());
    char* ws = static_cast<char*>(workspace);
    const char* sh = static_cast<const char*>(shadow);
    float* fr = reinterpret_cast<float*>(ws + p->f_base);
    // the pooling's backward inside the last run's stage-level launch (stage_train.hip: the gradient of the run's output is dfeat / (H W), formed in registers)
    bool pool_in_stage = false;
    {
        const char* c1 = clhip_cfg("STAGE_TRAIN_BWD");
        const char* c2 = clhip_cfg("STAGE_TRAIN");
        const char* c3 = clhip_cfg("STAGE_POOL");
        const Unit& lu = p->units.back();
        pool_in_stage = unit_hi == (int)p->units.size() && p->pool_win == 0 && p->xch != nullptr && !(c1 != nullptr && atoi(c1) == 0) && stage_train_default(c2, p->N) &&
                        !(c3 != nullptr && atoi(c3) == 0) && !(plan_skip() & 6) && lu.run_first >= 0 && lu.run_first >= unit_lo &&
                        lu.run_first + p->units[lu.run_first].stage_len == (int)p->units.size() && p->acts.back().C == lu.d.cout;
    }
    if (unit_hi == (int)p->units.size() && !pool_in_stage) {
        const Act& last = p->acts.back();
        if (p->pool_win > 0) TRY(clhip_avgpool_win_bwd(dfeat, ws + last.dy_off, p->N, last.H, last.W, last.C, p->pool_win, p->dtype, stream));
        else {
            // the pooling is the last activation's only reader: its backward completes that gradient and reduces the last unit's BatchNorm backward on the way
            const Unit& lu = p->units.back();
            static const bool pool_fuse_off = clhip_cfg("POOL_BN_FUSE") != nullptr && atoi(clhip_cfg("POOL_BN_FUSE")) == 0;
            const char* fe = clhip_cfg("BN_FUSE");
            bool sole = true;
            for (const Unit& o : p->units) if (o.d.src == (int)p->units.size() || o.d.res == (int)p->units.size()) sole = false;
            const bool fuse = !pool_fuse_off && p->use_acc && (fe == nullptr || atoi(fe) != 0) && sole && !lu.no_bn && !lu.pre_res && !lu.raw_src && !lu.has_dzr && lu.rep_bwd > 0 &&
                              lu.d.cout == last.C && last.C <= 256 /* wider: one fp64 atomic per channel and workgroup is 262 k atomics on ResNet-18's 512-channel map -- measured 0.5 % slower */ &&
                              clhip_avgpool_bwd_bn_reduce_supported(p->N, last.H * last.W, last.C, p->dtype) != 0;
            if (fuse) {
                TRY(clhip_avgpool_bwd_bn_reduce(dfeat, ws + last.dy_off, ws + lu.z_off, lu.relu ? ws + last.y_off : nullptr, fr + lu.f_mean, fr + lu.f_invstd,
                                                reinterpret_cast<double*>(ws + p->acc_off) + lu.a_bwd, lu.rep_bwd, p->N, last.H * last.W, last.C, p->dtype, stream));
                p->bwd_sums_ready[p->units.size() - 1] = 1;
            } else TRY(clhip_avgpool_bwd(dfeat, ws + last.dy_off, p->N, last.H * last.W, last.C, p->dtype, stream));
        }
    }
    // The weight gradients hang off the backward chain (BN backward -> dgrad -> next unit) as leaves: they run on a second stream,
    // so their kernels fill the load / store phases of the chain's kernels instead of queueing behind them.  dz is double-buffered;
    // events order  BN backward(i) -> wgrad(i)  and  wgrad(i) -> BN backward(i-2) (same dz buffer).  CLHIP_WGRAD_STREAM=0: one stream.
    static const bool two_streams_env = !(clhip_cfg("WGRAD_STREAM") && atoi(clhip_cfg("WGRAD_STREAM")) == 0);
    hipStream_t main_s = static_cast<hipStream_t>(stream);
    // inside a stream capture (trainer.GraphedStep: small, host-bound batches) everything stays on the captured stream: the fork /
    // join of a second stream gains nothing at those sizes, and ROCm 7.2 crashed in hipStreamEndCapture on it
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(main_s, &cap);
    const bool two_streams = two_streams_env && cap == hipStreamCaptureStatusNone;
    hipStream_t shared_side = nullptr;
    if (two_streams) {
        static const bool flat_prio = !(clhip_cfg("SIDE_PRIO") != nullptr && atoi(clhip_cfg("SIDE_PRIO")) != 0);
        shared_side = clhip_shared_stream(0, main_s, !flat_prio);          // (looked up per call: the caller's stream may change between calls)
        if (shared_side == nullptr) { clhip_set_error("clhip_plan_backward: cannot create the weight-gradient stream"); return CLHIP_EHIP; }
    }
    const bool first_side = two_streams && !p->side;
    if (two_streams) p->side = shared_side;
    if (first_side) {
        // lowest priority: the weight gradients are off the critical path (nothing waits for them before the optimizer step); when both
        // queues have workgroups ready the dispatcher should serve the caller's stream (dgrad, BatchNorm backward) first
        // round 4: a NORMAL-priority side stream is the default (SIDE_PRIO=1 restores the lowest priority).  A priority is a property of the hardware queue the
        // stream maps to; with the three-queue cap of the package (libcontinual_amd/__init__.py) the low-priority stream no longer gets a queue class of its own
        // and the step measured 1.2-1.7 % faster flat (2.0547 vs 2.0936 / 2.0834 on one box, 2.115 / 2.120 vs 2.149 / 2.141 on a slower one)
        // the events order two streams of ONE device: no timing, and no system-scope fence (the default flags make every record a cache
        // write-back + invalidate in the middle of the caller's stream; CLHIP_EVENT_FLAGS overrides the flag word)
        static const unsigned ev_flags = clhip_cfg("EVENT_FLAGS") ? (unsigned)strtoul(clhip_cfg("EVENT_FLAGS"), nullptr, 0)
                                                                      : (hipEventDisableTiming | hipEventDisableSystemFence);
        for (int k = 0; k < clhip_plan::kDz; ++k) {
            (void)hipEventCreateWithFlags(&p->ev_dz[k], ev_flags);
            (void)hipEventCreateWithFlags(&p->ev_wg[k], ev_flags);
            p->wg_pending[k] = false;
        }
        (void)hipEventCreateWithFlags(&p->ev_end, ev_flags);
    }
    // whatever path leaves this function (an error return included), no armed one-shot event survives it
    struct StopEventGuard { ~StopEventGuard() { clhip_bn_set_stop_event(nullptr); } } stop_event_guard;
    const bool defer_side = p->defer_side > 0 && two_streams;
    struct DeferGuard { bool on; ~DeferGuard() { if (on) clhip_wgrad_defer_abort(); } } defer_guard{p->defer_reduce || defer_side};
    if (p->defer_reduce || defer_side) clhip_wgrad_defer_begin();
    int side_deferred = 0;
    const char* br_cfg_b = clhip_cfg("BRANCH_STREAM");
    const int br_mode_b = br_cfg_b ? atoi(br_cfg_b) : 2;
    const bool br_on = two_streams && (br_mode_b == 1 || br_mode_b == 3) && branch_stream_on(p, main_s);
    bool br_used = false;
    p->br_act = -1;
    auto join_branch = [&]() {          // the caller's stream is about to touch the activation gradient the branch stream is writing
        (void)hipStreamWaitEvent(main_s, p->ev_bjoin[p->br_slot], 0);
        p->br_act = -1;
    };
    // down-sampling entries: the shortcut unit leaves its dz where it is and skips its own input gradient; the block's 3x3/s2 unit,
    // two iterations of this loop at most later and on the other dz buffer, produces the gradient of their common input in ONE launch
    // (clhip_conv_dgrad_pair).  Needs both units inside this call's range; on one stream (graph capture), where every unit uses the first
    // dz buffer, the shortcut unit takes the twin -- the arithmetic of a step does not depend on the stream layout.
    const bool pair_on = !br_on;
    const void* pair_dz = nullptr;
    const char* lg_cfg = clhip_cfg("BN_GRAD");
    const bool lazy_grad_on = !(lg_cfg != nullptr && atoi(lg_cfg) == 0);
    const char* lgc_cfg = clhip_cfg("BN_GRAD_MINC");
    const int lazy_grad_minc = lgc_cfg != nullptr ? atoi(lgc_cfg) : 0;
    const char* lgr_cfg = clhip_cfg("BN_GRAD_RES");
    const bool lazy_res_on = !(lgr_cfg != nullptr && atoi(lgr_cfg) == 0);
    // a lazy activation some launch of this sweep has to READ as a tensor after all: write it now (same values the forward would have stored)
    auto materialise = [&](int a) -> int {
        const Unit& ua = p->units[a];
        if (!p->lazy_live[a]) return CLHIP_OK;
        p->lazy_live[a] = 0;
        return clhip_bn_apply(ws + ua.z_off, fr + ua.f_scale, fr + ua.f_shift, nullptr, ws + p->acts[a + 1].y_off, ua.M, ua.d.cout, 1, p->dtype, stream);
    };
    int k = 0;
    const int n_dz = two_streams ? p->n_dz : 2;
    auto any_pending = [&]() { for (int q = 0; q < clhip_plan::kDz; ++q) if (p->wg_pending[q]) return true; return false; };
    auto clear_pending = [&]() { for (int q = 0; q < clhip_plan::kDz; ++q) p->wg_pending[q] = false; };
    // STAGE_TRAIN_BWD (default on; looked up per call): a run of BasicBlocks that lies inside this call's range goes as ONE launch (stage_train.hip), whichever forward ran
    const char* stb_cfg = clhip_cfg("STAGE_TRAIN_BWD");
    const char* stb_cfg2 = clhip_cfg("STAGE_TRAIN");
    const bool stb_on = p->xch != nullptr && !(stb_cfg != nullptr && atoi(stb_cfg) == 0) && stage_train_default(stb_cfg2, p->N) && !br_on && !(plan_skip() & 6);
    if (pool_in_stage && !stb_on) {                          // (decided before the branch stream's state was known: the pooling's backward as its own launch after all)
        const Act& last = p->acts.back();
        TRY(clhip_avgpool_bwd(dfeat, ws + last.dy_off, p->N, last.H * last.W, last.C, p->dtype, stream));
        pool_in_stage = false;
    }
    for (int i = unit_hi - 1; i >= unit_lo; --i, k = (k + 1) % n_dz) {
        if (stb_on && p->units[i].run_first >= 0) {
            const int f = p->units[i].run_first;
            const Unit& uf = p->units[f];
            if (i == f + uf.stage_len - 1 && f >= unit_lo) {
                const int len = uf.stage_len;
                static const bool entry_off = clhip_cfg("STAGE_ENTRY") != nullptr && atoi(clhip_cfg("STAGE_ENTRY")) == 0;
                const bool entry_bwd_off = clhip_cfg("STAGE_ENTRY_BWD") != nullptr && atoi(clhip_cfg("STAGE_ENTRY_BWD")) == 0;
                const int entry_sel = clhip_cfg("STAGE_ENTRY_BWD") != nullptr ? atoi(clhip_cfg("STAGE_ENTRY_BWD")) : 1;      // (2 / 3: only the 32- / 64-channel stage: debugging)
                if (!entry_off && !entry_bwd_off && uf.entry_first >= 0 && uf.entry_first >= unit_lo && (entry_sel == 1 || (entry_sel == 2 && uf.d.cout == 32) || (entry_sel == 3 && uf.d.cout == 64))) {
                    // the launch takes the stage's down-sampling block along: units e (3x3 / s2), e + 1 (shortcut), e + 2 (second convolution), then the run
                    const int e0 = uf.entry_first, n = len + 3;
                    const Unit& ua = p->units[e0];
                    const void* wdv[20]; const float *gv[20], *bv[20], *mev[20], *isv[20]; const void *zv[20], *yv[20]; float *dgv[20], *dbv[20], *slv[20]; void* dzv[20];
                    for (int q = 0; q < n; ++q) {
                        const Unit& uq = p->units[e0 + q];
                        wdv[q] = sh + uq.sh_dg; gv[q] = params + uq.d.gamma_off; bv[q] = params + uq.d.beta_off; mev[q] = fr + uq.f_mean; isv[q] = fr + uq.f_invstd;
                        zv[q] = ws + uq.z_off; yv[q] = (q == 2 || (q >= 3 && ((q - 3) & 1))) ? ws + p->acts[e0 + q + 1].y_off : nullptr;
                        dgv[q] = grads + uq.d.gamma_off; dbv[q] = grads + uq.d.beta_off; slv[q] = reinterpret_cast<float*>(ws + uq.st_slab);
                        dzv[q] = ws + p->acts[e0 + q + 1].dy_off;
                    }
                    if (p->br_act >= 0) join_branch();
                    TRY(stage_train_serialize(p, main_s));
                    TRY(clhip_stage_train_bwd_launch(ws + p->acts[ua.d.src].y_off, ws + p->acts[f + len].dy_off, ws + p->acts[ua.d.src].dy_off, p->units[e0 + 1].dx_acc, p->N, uf.H,
                                                     uf.W, uf.d.cout, len + 1, wdv, gv, bv, mev, isv, zv, yv, dgv, dbv, slv, dzv, p->xch, stage_trace_cfg(uf.d.cout), 1,
                                                     pool_in_stage && f + len == (int)p->units.size() ? dfeat : nullptr, p->dtype, main_s));
                    ++p->st_bwd_launches;
                    for (int q = n - 1; q >= 0; --q) {
                        const Unit& uq = p->units[e0 + q];
                        const int blocks = q < 2 ? p->N : clhip_stage_train_slab_blocks(p->N, uq.d.cout);
                        TRY(clhip_wgrad_reduce_launch(reinterpret_cast<const float*>(ws + uq.st_slab), grads + uq.d.w_off, (int64_t)uq.d.cout * uq.d.ksize * uq.d.ksize * uq.cin_pad / 4,
                                                      blocks, main_s));
                    }
                    i = e0;
                    continue;
                }
                const void* wdv[16]; const float *gv[16], *bv[16], *mev[16], *isv[16]; const void *zv[16], *yv[16]; float *dgv[16], *dbv[16], *slv[16]; void* dzv[16];
                for (int q = 0; q < len; ++q) {
                    const Unit& uq = p->units[f + q];
                    wdv[q] = sh + uq.sh_dg; gv[q] = params + uq.d.gamma_off; bv[q] = params + uq.d.beta_off; mev[q] = fr + uq.f_mean; isv[q] = fr + uq.f_invstd;
                    zv[q] = ws + uq.z_off; yv[q] = (q & 1) ? ws + p->acts[f + q + 1].y_off : nullptr;
                    dgv[q] = grads + uq.d.gamma_off; dbv[q] = grads + uq.d.beta_off; slv[q] = reinterpret_cast<float*>(ws + uq.st_slab);
                    dzv[q] = ws + p->acts[f + q + 1].dy_off;         // (the gradient buffer of the unit's output: the launch keeps that gradient in registers and parks its dz there)
                }
                if (p->br_act >= 0) join_branch();
                TRY(stage_train_serialize(p, main_s));
                TRY(clhip_stage_train_bwd_launch(ws + p->acts[uf.d.src].y_off, ws + p->acts[f + len].dy_off, ws + p->acts[uf.d.src].dy_off, p->units[f + 1].dres_acc, p->N, uf.H,
                                                 uf.W, uf.d.cout, len, wdv, gv, bv, mev, isv, zv, yv, dgv, dbv, slv, dzv, p->xch, stage_trace_cfg(uf.d.cout), 0,
                                                 pool_in_stage && f + len == (int)p->units.size() ? dfeat : nullptr, p->dtype, main_s));
                ++p->st_bwd_launches;
                for (int q = len - 1; q >= 0; --q) {
                    const Unit& uq = p->units[f + q];
                    TRY(clhip_wgrad_reduce_launch(reinterpret_cast<const float*>(ws + uq.st_slab), grads + uq.d.w_off, (int64_t)uq.d.cout * 9 * uq.cin_pad / 4, clhip_stage_train_slab_blocks(p->N, uq.d.cout), main_s));
                }
                i = f;                                               // (the loop's own decrement moves past the run)
                continue;
            }
        }
        const Unit& u = p->units[i];
        const Act& src = p->acts[u.d.src];
        const Act& dst = p->acts[i + 1];
        void* dres = (u.d.res >= 0 && !u.pre_res) ? ws + p->acts[u.d.res].dy_off : nullptr;
        const bool pair_b = pair_on && u.pair >= 0 && u.d.ksize == 1 && u.pair == i - 1 && u.pair >= unit_lo;
        char* dz = ws + (two_streams ? p->dz_offs[k] : (pair_b ? p->dz_off2 : p->dz_off));
        if (br_on && u.branch >= 0 && p->bfork_ev[u.branch] != nullptr) {
            // ---- shortcut branch (1x1 conv -> BN, its activation consumed as a residual only): BatchNorm backward, input gradient and weight
            //      gradient on the branch stream, beside the main path's next unit (which shares nothing with it but the gradient of their
            //      common input activation: this branch writes it first, the main path joins before it accumulates)
            const int slot = u.branch;
            (void)hipStreamWaitEvent(p->br, p->bfork_ev[slot], 0);            // its dy = the residual gradient written by the consumer's BatchNorm backward
            p->bfork_ev[slot] = nullptr;
            if (p->wg_pending[k]) { (void)hipStreamWaitEvent(p->br, p->ev_wg[k], 0); p->wg_pending[k] = false; }      // the last reader of this dz buffer
            if (p->br_act >= 0) join_branch();                                 // (two branches never overlap: keep the bookkeeping single-slot)
            TRY(clhip_bn_bwd_acc(ws + dst.dy_off, ws + dst.y_off, ws + u.z_off, fr + u.f_mean, fr + u.f_invstd, params + u.d.gamma_off,
                                 grads + u.d.gamma_off, grads + u.d.beta_off, dz, nullptr, 0, u.M, u.d.cout, 0,
                                 reinterpret_cast<double*>(ws + p->acc_off) + u.a_bwd, u.rep_bwd, p->dtype, p->br));
            TRY(clhip_conv_dgrad(dz, sh + u.sh_dg, ws + src.dy_off, u.dx_acc, p->N, u.H, u.W, u.cin_pad, u.d.cout, u.d.ksize, u.d.stride, u.d.pad, p->dtype, p->br));
            (void)hipEventRecord(p->ev_bjoin[slot], p->br);
            p->br_act = u.d.src; p->br_slot = slot;
            TRY(clhip_conv_wgrad(ws + src.y_off, dz, grads + u.d.w_off, ws + p->wg_off2, p->N, u.H, u.W, u.cin_pad, u.d.cin, u.d.cout,
                                 u.d.ksize, u.d.stride, u.d.pad, p->dtype, p->br));
            (void)hipEventRecord(p->ev_wg[k], p->br);                          // ... of this dz buffer is now on the branch stream
            p->wg_pending[k] = true;
            br_used = true;
            continue;
        }
        if (p->br_act >= 0 && i + 1 == p->br_act) join_branch();             // this unit's BatchNorm backward reads the gradient the branch is writing
        // pre-activation wiring: the gradient of this unit's sum IS the gradient of the raw sum it added (written straight into that
        // producer's buffer); a BN-less unit's gradient is whatever its raw consumer left in its own buffer
        if (u.pre_res) dz = ws + p->units[u.d.res - 1].dzr_off;
        if (u.no_bn) dz = ws + u.dzr_off;
        const char* in = u.raw_src ? ws + p->units[u.d.src - 1].z_off : ws + src.y_off;
        if (two_streams && p->wg_pending[k]) {
            (void)hipStreamWaitEvent(main_s, p->ev_wg[k], 0);         // the weight gradient of two units ago has finished reading this buffer
            p->wg_pending[k] = false;
        }
        static const bool mask_from_y = clhip_cfg("BN_MASK_FROM_Y") != nullptr;       // ablation: always read the activation
        // the weight gradient goes to the side stream (see below): let the BatchNorm apply launch -- the last writer of dz -- complete the
        // event itself instead of recording one behind it
        const double wg_flops = 2.0 * (double)u.M * u.d.ksize * u.d.ksize * u.d.cin * u.d.cout;
        const bool on_side = two_streams && p->side_ok && wg_flops >= 1.0e9;
        static const bool ev_in_launch = clhip_cfg("EVENT_RECORD") == nullptr;
        const bool hook = on_side && ev_in_launch && !u.no_bn && u.rep_bwd > 0 && !u.has_dzr;
        if (hook) clhip_bn_set_stop_event(p->ev_dz[k]);
        // layers whose dgrad and weight gradient are both launches at their latency floor on this one stream: ONE launch for the two
        // (clhip_conv_dgrad_wgrad, conv3.hip: CifarResNet-32 stages 1 and 2) ...
        const bool both = !on_side && u.d.src != 0 && !u.raw_src && !(p->br_act >= 0 && u.d.src == p->br_act) &&
                          clhip_conv_dgrad_wgrad_supported(p->N, u.H, u.W, u.cin_pad, u.d.cin, u.d.cout, u.d.ksize, u.d.stride, u.d.pad, p->dtype) &&
                          clhip_conv_wgrad_ws_bytes(p->N, u.H, u.W, u.cin_pad, u.d.cin, u.d.cout, u.d.ksize, u.d.stride, u.d.pad, p->dtype) > 0;
        // ... and, where the unit's own BatchNorm backward is an apply pass with the ReLU mask from z (sums already reduced by the consumer's dgrad
        // epilogue, no residual gradient to write), that pass happens on the operand loads of the same launch: dz is never written
        // (two forms: ReLU straight after the BatchNorm, mask from z; or conv -> BN -> +res -> ReLU with the packed mask of the forward, the residual
        //  gradient written by the same launch)
        const bool lazy_in = u.lazy_from >= 0 && p->lazy_live[u.lazy_from];
        const bool bn_grad_ok = both && lazy_grad_on && u.d.cout >= lazy_grad_minc && !u.no_bn && !u.pre_res && !u.has_dzr && u.rep_bwd > 0 && p->bwd_sums_ready[i] &&
                                u.relu && !mask_from_y && u.cin_pad == u.d.cin &&                                 clhip_conv_bn_input_supported(p->N, u.H, u.W, u.cin_pad, u.d.cout, u.d.ksize, u.d.stride, u.d.pad, p->dtype);
        const bool stale = p->mask_stale[i] != 0;                  // its packed mask was not written (stage-level forward): the activation instead
        const bool bn_grad = bn_grad_ok && (dres == nullptr ? true : (u.mask_off != 0 && lazy_res_on && !stale));
        if (u.no_bn || bn_grad) {
        } else if (((plan_skip() & 2) || ((plan_skip() & 8) && u.d.ksize == 1 && !u.relu)) && u.rep_bwd > 0) {         // timing ablation: no BatchNorm backward [8: of the shortcut units] (results invalid)
        } else if (u.rep_bwd > 0 && p->bwd_sums_ready[i]) {
            // the two channel sums came out of the epilogue of the dgrad that completed dy (see fuse_src_bn): apply pass only
            const bool zmask = u.relu && dres == nullptr && !mask_from_y;
            const bool bits = !zmask && u.relu && u.mask_off != 0 && !mask_from_y && !stale;
            TRY(clhip_bn_bwd_apply_acc(ws + dst.dy_off, zmask ? nullptr : (bits ? ws + u.mask_off : ws + dst.y_off), ws + u.z_off, fr + u.f_mean, fr + u.f_invstd,
                                       params + u.d.gamma_off, params + u.d.beta_off, grads + u.d.gamma_off, grads + u.d.beta_off, dz, dres, u.dres_acc, u.M,
                                       u.d.cout, zmask ? 2 : (bits ? 3 : (u.relu ? 1 : 0)), reinterpret_cast<double*>(ws + p->acc_off) + u.a_bwd, u.rep_bwd,
                                       p->dtype, stream));
        } else if (u.rep_bwd > 0 && u.relu && dres == nullptr && !mask_from_y) {
            // ReLU straight after the BatchNorm (no residual in between): the mask is recomputed from z, y is not read
            TRY(clhip_bn_bwd_acc_zmask(ws + dst.dy_off, ws + u.z_off, fr + u.f_mean, fr + u.f_invstd, params + u.d.gamma_off, params + u.d.beta_off,
                                       grads + u.d.gamma_off, grads + u.d.beta_off, dz, u.M, u.d.cout,
                                       reinterpret_cast<double*>(ws + p->acc_off) + u.a_bwd, u.rep_bwd, p->dtype, stream));
        } else if (u.rep_bwd > 0) {
            const bool bits = u.relu && u.mask_off != 0 && !mask_from_y && !stale;
            TRY(clhip_bn_bwd_acc(ws + dst.dy_off, bits ? ws + u.mask_off : ws + dst.y_off, ws + u.z_off, fr + u.f_mean, fr + u.f_invstd, params + u.d.gamma_off,
                                 grads + u.d.gamma_off, grads + u.d.beta_off, dz, dres, u.dres_acc, u.M, u.d.cout, bits ? 3 : (int)u.relu,
                                 reinterpret_cast<double*>(ws + p->acc_off) + u.a_bwd, u.rep_bwd, p->dtype, stream));
        } else {
            TRY(clhip_bn_bwd(ws + dst.dy_off, ws + dst.y_off, ws + u.z_off, fr + u.f_mean, fr + u.f_invstd, params + u.d.gamma_off,
                             grads + u.d.gamma_off, grads + u.d.beta_off, dz, dres, u.dres_acc, u.M, u.d.cout, u.relu,
                             fr + p->f_bnws, p->dtype, stream));
        }
        if (u.has_dzr && !u.no_bn) TRY(clhip_add_inplace(dz, ws + u.dzr_off, u.M * u.d.cout, p->dtype, stream));
        bool dz_event_recorded = false;
        if (br_on && u.joins >= 0 && !u.pre_res && dres != nullptr) {
            // the residual gradient this BatchNorm backward wrote is the dy of a shortcut branch: the branch may start as soon as THIS launch
            // is complete -- the launch's own completion event when it carries one, an event recorded behind it otherwise
            if (hook && clhip_bn_pending_stop_event() == nullptr) p->bfork_ev[u.joins] = p->ev_dz[k];
            else {
                if (on_side) { clhip_bn_set_stop_event(nullptr); (void)hipEventRecord(p->ev_dz[k], main_s); dz_event_recorded = true; p->bfork_ev[u.joins] = p->ev_dz[k]; }
                else { (void)hipEventRecord(p->ev_bfork[u.joins], main_s); p->bfork_ev[u.joins] = p->ev_bfork[u.joins]; }
            }
        }
        // small layers stay on the caller's stream: below ~1 GFLOP the three event calls cost more host time than the overlap wins
        // (ResNet-32 at batch <= 128 is host-bound: 60.5 k img/s on one stream vs 57.1 k on two; ResNet-18 gains from batch 64 up)
        // A network made of such layers only gets no side stream at all: ResNet-32 at batch 256 (1.2 GFLOP per layer, 7-19 us kernels) ran
        // 2.08-2.35 ms per EWC step from run to run with its weight gradients on the side stream, 2.200 +- 0.003 ms without -- the same mean,
        // whereas ResNet-18 (10-19 GFLOP layers; its 1-GFLOP shortcut convs included) gains 10 % reproducibly.  `side_ok` = the plan has a
        // layer of CLHIP_WGRAD_NET_GFLOP (default 4) GFLOP or more.
        void* wg_stream = stream;
        if (on_side) {
            if (!dz_event_recorded && (!hook || clhip_bn_pending_stop_event() != nullptr)) {          // (not taken: a BatchNorm path without the hook)
                clhip_bn_set_stop_event(nullptr);
                (void)hipEventRecord(p->ev_dz[k], main_s);
            }
            (void)hipStreamWaitEvent(p->side, p->ev_dz[k], 0);
            wg_stream = p->side;
        } else if (two_streams && any_pending() &&
                   (clhip_cfg("WGRAD_ALWAYS_QUEUE") != nullptr ||
                    clhip_conv_wgrad_ws_bytes(p->N, u.H, u.W, u.cin_pad, u.d.cin, u.d.cout, u.d.ksize, u.d.stride, u.d.pad, p->dtype) > 0)) {
            // this unit's weight gradient shares the partial-sum scratch with the ones in flight on the side stream: queue behind them
            // (the atomic kernels -- the stem -- use no scratch and run beside the side stream's tail instead of behind it)
            (void)hipEventRecord(p->ev_end, p->side);
            (void)hipStreamWaitEvent(main_s, p->ev_end, 0);
            clear_pending();
        }
        if (defer_side) clhip_wgrad_defer_pause(!on_side);             // only the side stream's launches are collected
        // layers whose dgrad and weight gradient are both launches at their latency floor on this one stream: ONE launch for the two
        // (clhip_conv_dgrad_wgrad, conv3.hip: CifarResNet-32 stages 1 and 2)
        if (bn_grad) {
            const Unit* prod = u.fuse_src_bn ? &p->units[u.d.src - 1] : nullptr;
            clhip_bn_grad bg;
            bg.dy = ws + dst.dy_off; bg.z = ws + u.z_off; bg.sums = reinterpret_cast<double*>(ws + p->acc_off) + u.a_bwd; bg.replicas = u.rep_bwd;
            bg.mean = fr + u.f_mean; bg.invstd = fr + u.f_invstd; bg.gamma = params + u.d.gamma_off; bg.beta = params + u.d.beta_off;
            bg.dgamma = grads + u.d.gamma_off; bg.dbeta = grads + u.d.beta_off;
            bg.relu_mask = dres != nullptr ? ws + u.mask_off : nullptr; bg.dres = dres; bg.dres_accumulate = u.dres_acc;
            const Unit* la = lazy_in ? &p->units[u.lazy_from] : nullptr;       // its input is a lazy activation as well: x = z of that unit + its coefficients
            TRY(clhip_conv_dgrad_wgrad_bn_grad(la ? ws + la->z_off : in, la ? fr + la->f_scale : nullptr, &bg, sh + u.sh_dg, ws + src.dy_off, u.dx_acc,
                                               grads + u.d.w_off, ws + u.wg_own, prod ? ws + prod->z_off : nullptr, (prod && prod->relu) ? ws + src.y_off : nullptr,
                                               prod ? fr + prod->f_mean : nullptr, prod ? fr + prod->f_invstd : nullptr,
                                               prod ? reinterpret_cast<double*>(ws + p->acc_off) + prod->a_bwd : nullptr, prod ? prod->rep_bwd : 1, p->N, u.H, u.W,
                                               u.cin_pad, u.d.cin, u.d.cout, u.d.ksize, u.d.stride, u.d.pad, p->dtype, stream));
            if (prod) p->bwd_sums_ready[u.d.src - 1] = 1;
            continue;
        }
        if (both && u.lazy_from >= 0 && p->lazy_live[u.lazy_from]) {
            const Unit& a = p->units[u.lazy_from];                // x = relu(bn(z_a)) on the weight gradient's operand load, the producer's ReLU mask from z_a
            const bool red = u.fuse_src_bn;
            TRY(clhip_conv_dgrad_wgrad_bn_input(ws + a.z_off, fr + a.f_scale, dz, sh + u.sh_dg, ws + src.dy_off, u.dx_acc, grads + u.d.w_off, ws + u.wg_own,
                                                red ? fr + a.f_mean : nullptr, red ? fr + a.f_invstd : nullptr,
                                                red ? reinterpret_cast<double*>(ws + p->acc_off) + a.a_bwd : nullptr, red ? a.rep_bwd : 1, p->N, u.H, u.W,
                                                u.cin_pad, u.d.cin, u.d.cout, u.d.ksize, u.d.stride, u.d.pad, p->dtype, stream));
            if (red) p->bwd_sums_ready[u.lazy_from] = 1;
            continue;
        }
        if (u.lazy_from >= 0) TRY(materialise(u.lazy_from));
        if (both) {
            const Unit* prod = u.fuse_src_bn ? &p->units[u.d.src - 1] : nullptr;
            TRY(clhip_conv_dgrad_wgrad(in, dz, sh + u.sh_dg, ws + src.dy_off, u.dx_acc, grads + u.d.w_off, ws + u.wg_own,
                                       prod ? ws + prod->z_off : nullptr, (prod && prod->relu) ? ws + src.y_off : nullptr, prod ? fr + prod->f_mean : nullptr,
                                       prod ? fr + prod->f_invstd : nullptr, prod ? reinterpret_cast<double*>(ws + p->acc_off) + prod->a_bwd : nullptr,
                                       prod ? prod->rep_bwd : 1, p->N, u.H, u.W, u.cin_pad, u.d.cin, u.d.cout, u.d.ksize, u.d.stride, u.d.pad, p->dtype, stream));
            if (prod) p->bwd_sums_ready[u.d.src - 1] = 1;
            continue;
        }
        if (pair_b && u.wpair) {
            // its weight gradient comes out of its 3x3 partner's launch (the next unit of this sweep)
        } else if (pair_on && u.pair >= 0 && u.wpair && u.d.ksize == 3 && pair_dz != nullptr) {
            const Unit& sc = p->units[u.pair];
            TRY(clhip_conv_wgrad_pair(in, dz, pair_dz, grads + u.d.w_off, grads + sc.d.w_off, ws + u.wg_own, ws + sc.wg_own, p->N, u.H, u.W, u.cin_pad, u.d.cout,
                                      p->dtype, wg_stream));
        } else if (!(plan_skip() & 4))
        TRY(clhip_conv_wgrad(in, dz, grads + u.d.w_off, ws + u.wg_own, p->N, u.H, u.W, u.cin_pad, u.d.cin, u.d.cout,
                             u.d.ksize, u.d.stride, u.d.pad, p->dtype, wg_stream));
        if (defer_side && on_side && ++side_deferred % p->defer_side == 0) TRY(clhip_wgrad_defer_flush(p->side, false));
        if (on_side) {
            (void)hipEventRecord(p->ev_wg[k], p->side);
            p->wg_pending[k] = true;
        }
        if (p->br_act >= 0 && u.d.src == p->br_act && !u.raw_src) join_branch();      // this unit's dgrad accumulates into the gradient the branch wrote first
        if (u.raw_src) {
            TRY(clhip_conv_dgrad(dz, sh + u.sh_dg, ws + p->units[u.d.src - 1].dzr_off, 0, p->N, u.H, u.W, u.cin_pad, u.d.cout,
                                 u.d.ksize, u.d.stride, u.d.pad, p->dtype, stream));
        } else if (u.d.src != 0 && u.fuse_src_bn) {
            const Unit& prod = p->units[u.d.src - 1];
            // (the producer's ReLU mask: its packed bits if it keeps them, its scale / shift if the ReLU follows the BatchNorm directly, else its activation)
            const bool pz = prod.relu && prod.d.res < 0 && !mask_from_y, pb = prod.relu && prod.mask_off != 0 && !mask_from_y && !p->mask_stale[u.d.src - 1];
            TRY(clhip_conv_dgrad_bn_reduce_ex(dz, sh + u.sh_dg, ws + src.dy_off, u.dx_acc, ws + prod.z_off, prod.relu ? ws + src.y_off : nullptr,
                                              pb ? ws + prod.mask_off : nullptr, pz ? params + prod.d.gamma_off : nullptr, pz ? params + prod.d.beta_off : nullptr,
                                              fr + prod.f_mean, fr + prod.f_invstd, reinterpret_cast<double*>(ws + p->acc_off) + prod.a_bwd, prod.rep_bwd,
                                              p->N, u.H, u.W, u.cin_pad, u.d.cout, u.d.ksize, u.d.stride, u.d.pad, p->dtype, stream));
            p->bwd_sums_ready[u.d.src - 1] = 1;
        } else if (pair_b) {
            pair_dz = dz;                                      // its partner is the next unit of this sweep
        } else if (pair_on && u.pair >= 0 && u.d.ksize == 3 && pair_dz != nullptr) {
            static const bool pfb_off = clhip_cfg("PAIR_BN_FUSE") != nullptr && atoi(clhip_cfg("PAIR_BN_FUSE")) == 0;
            if (u.pair_fuse_bn && !pfb_off) {
                const Unit& prod = p->units[u.d.src - 1];
                TRY(clhip_conv_dgrad_pair_bn_reduce(dz, sh + u.sh_pk, pair_dz, ws + src.dy_off, u.pair_acc, ws + prod.z_off, prod.relu ? ws + src.y_off : nullptr,
                                                    fr + prod.f_mean, fr + prod.f_invstd, reinterpret_cast<double*>(ws + p->acc_off) + prod.a_bwd, prod.rep_bwd, p->N, u.H, u.W,
                                                    u.cin_pad, u.d.cout, p->dtype, stream));
                p->bwd_sums_ready[u.d.src - 1] = 1;
            } else
            TRY(clhip_conv_dgrad_pair(dz, sh + u.sh_pk, pair_dz, ws + src.dy_off, u.pair_acc, p->N, u.H, u.W, u.cin_pad, u.d.cout, p->dtype, stream));
            pair_dz = nullptr;
        } else if (u.d.src != 0) {
            TRY(clhip_conv_dgrad(dz, sh + u.sh_dg, ws + src.dy_off, u.dx_acc, p->N, u.H, u.W, u.cin_pad, u.d.cout,
                                 u.d.ksize, u.d.stride, u.d.pad, p->dtype, stream));
        }
    }
    if (p->defer_reduce) { TRY(clhip_wgrad_defer_flush(main_s, true)); defer_guard.on = false; }      // all "dw += slab" of this range, one launch
    if (defer_side) { clhip_wgrad_defer_pause(false); TRY(clhip_wgrad_defer_flush(p->side, true)); defer_guard.on = false; }
    if (br_used) {                                                     // ... and everything the branch stream wrote
        (void)hipEventRecord(p->ev_br_end, p->br);
        (void)hipStreamWaitEvent(main_s, p->ev_br_end, 0);
        p->br_act = -1;
    }
    for (int q = 0; q < clhip_plan::kMaxBranch && br_on; ++q) p->bfork_ev[q] = nullptr;      // (a consumer whose branch unit lies outside this range)
    if (two_streams && any_pending()) {      // the caller's stream owns the gradients again when this call returns
        (void)hipEventRecord(p->ev_end, p->side);
        (void)hipStreamWaitEvent(main_s, p->ev_end, 0);
        clear_pending();
    }
    return CLHIP_OK;
}

extern "C" int clhip_plan_read_act(clhip_plan* p, const void* workspace, int idx, int which, float* out_nchw, void* stream) {
    CLHIP_CHECK_ARG(p && workspace && out_nchw && idx >= 0 && idx < (int)p->acts.size() && which >= 0 && which <= 2);
    CLHIP_CHECK_ARG(!(idx == 0 && which != 0));
    const char* ws = static_cast<const char*>(workspace);
    const Act& a = p->acts[idx];
    if (which == 0 && idx >= 1 && p->lazy_live[idx - 1]) {
        // a lazy activation (its consumer applied the BatchNorm on its operand load): write it on demand -- the buffer is there, the values
        // are the ones the apply launch would have stored
        const Unit& ua = p->units[idx - 1];
        char* wsm = static_cast<char*>(const_cast<void*>(workspace));
        float* fr = reinterpret_cast<float*>(wsm + p->f_base);
        if (int e = clhip_bn_apply(wsm + ua.z_off, fr + ua.f_scale, fr + ua.f_shift, nullptr, wsm + a.y_off, ua.M, ua.d.cout, 1, p->dtype, stream)) return e;
        p->lazy_live[idx - 1] = 0;
    }
    if (which <= 1 && idx >= 1 && p->stage_skipped[idx - 1]) {
        clhip_set_error("clhip_plan_read_act: activation %d lies inside a run of blocks the eval forward executed as one launch (stage.hip) and was never written; "
                        "clhip_config(\"STAGE_EVAL\", \"0\") keeps one launch per unit", idx);
        return CLHIP_EINVAL;
    }
    if (which == 0 && idx >= 1 && p->eval_unwritten[idx - 1]) {
        // the eval forward applied this unit's BatchNorm (running statistics) on its consumer's operand load: the buffer holds whatever an
        // earlier pass left there.  Its z is intact, so the apply launch the eager eval path would have made produces the same values now.
        const Unit& ua = p->units[idx - 1];
        if (p->params_dev == nullptr || p->bn_stats_dev == nullptr) { clhip_set_error("clhip_plan_read_act: activation %d was not written by the eval forward (EVAL_LAZY) and the plan has no parameter pointers to rebuild it from", idx); return CLHIP_EINVAL; }
        char* wsm = static_cast<char*>(const_cast<void*>(workspace));
        const float* params = p->params_dev; const float* bn_stats = p->bn_stats_dev;
        const void* res_e = ua.d.res >= 0 ? wsm + p->acts[ua.d.res].y_off : nullptr;
        if (int e = clhip_bn_apply_eval(wsm + ua.z_off, params + ua.d.gamma_off, params + ua.d.beta_off, bn_stats + ua.d.rm_off, bn_stats + ua.d.rv_off, kBnEps, res_e,
                                        wsm + a.y_off, ua.M, ua.d.cout, ua.relu, p->dtype, stream)) return e;
        p->eval_unwritten[idx - 1] = 0;
    }
    const char* src = which == 0 ? ws + a.y_off : (which == 1 ? ws + p->units[idx - 1].z_off : ws + a.dy_off);
    return clhip_nhwc_to_nchw(src, out_nchw, p->N, a.C, a.H, a.W, p->dtype, stream);
}
