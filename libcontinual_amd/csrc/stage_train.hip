// stage_train.hip -- TRAINING forward / backward of a RUN of BasicBlocks of the CIFAR ResNet-32s as ONE launch per direction (round 6).
//
// Replaces, for runs of  conv3x3(C -> C, stride 1) -> BN -> ReLU -> conv3x3 -> BN -> (+ x) -> ReLU  blocks in training mode
// (core/model/backbone/resnet.py:289-316; the stages built at :381-392 and :549-557; autograd's backward of the same, core/trainer.py:604), the 2 x blocks launches
// per direction of the per-unit path (9-23 us each on activations of at most 8 MB: launch prologues and epilogues, not bytes) by one:
//  * ONE workgroup (four waves) per IMAGE, all of them co-resident (N <= number of CUs; the plan falls back to the per-unit launches otherwise); the image's
//    activation stays in LDS for the whole run, zero-haloed, exactly as in the eval-mode kernel (stage.hip): a convolution is an implicit GEMM on
//    v_mfma_f32_16x16x32_bf16 with the filters as the A operand, filters in registers with the next convolution's set in flight;
//  * BatchNorm in training mode needs the batch statistics of every convolution's output before anything downstream can start: the per-image channel sums
//    (fp32, from the fp32 accumulators as the per-unit kernels take them) go through the in-launch all-reduce of xch.h -- two hops of tagged 8-byte
//    granules, fp64 totals in ONE fixed order, identical in every workgroup -- so a run is bit-reproducible and independent of workgroup placement;
//  * the forward leaves what the per-unit forward leaves: z of every convolution, mean / invstd / scale / shift and the running statistics of every
//    BatchNorm, the activation + packed ReLU mask of every block output; the activation between the two convolutions of a block is NOT written (the
//    backward recomputes it from z, as the per-unit "lazy" path does).  Either backward can therefore follow either forward.
#include <stdlib.h>

#include "common.h"
#include "xch.h"

namespace {

constexpr int kMaxConvT = 16;

// phase stamps (100-MHz s_memtime ticks) of ONE convolution of workgroup 0 into ctl[8 + 2 i ..] of the exchange buffer: a diagnostic, off unless STAGE_TRACE is set
__device__ __forceinline__ void st_stamp(const XchBuf& b, bool on, int slot) {
    if (on && threadIdx.x == 0) reinterpret_cast<unsigned long long*>(b.ctl + 8)[slot] = __builtin_amdgcn_s_memtime();
}

struct StConv {
    const bf16_t* w;          // [C][9][C] forward copy
    const bf16_t* wd;         // [C][9][C] dgrad copy (backward)
    const float* gamma;
    const float* beta;
    float* rm;                // running statistics (forward, workgroup 0)
    float* rv;
    float* mean;              // [C]   saved statistics: written by the forward, read by the backward
    float* invstd;            // [C]
    float* coef;              // [2][C] scale, shift
    bf16_t* z;                // [N][HW][HW][C] pre-BatchNorm output
    bf16_t* y;                // block outputs (second convolution of a block): the activation; nullptr for the first convolution of a block
    unsigned char* mask;      // ... and its packed ReLU mask (bit e of byte i = element 8 i + e > 0), or nullptr
};

struct StFwdParams {
    const bf16_t* x;          // [N][HW][HW][C] the run's input activation
    int N, nconv;
    float eps, momentum;
    double invM, unbias;      // 1 / (N HW HW), M / (M - 1)
    int trace;                // > 0: workgroup 0 stamps the phases of convolution `trace` into the exchange buffer's spare words (clhip_stage_train_trace)
    float* feat;              // nullable: [N][C] the global average of the run's output (the pooling behind the LAST run of a backbone)
    XchBuf xb;
    StConv ea, ed;            // (entry block) its 3x3 / stride-2 convolution and its 1x1 / stride-2 shortcut convolution; x is then the block's INPUT [N][2 HW][2 HW][C / 2]
    StConv c[kMaxConvT];
};

template <int C, int HW>
struct StGeo {
    static constexpr int P = HW + 2;                       // padded width
    static constexpr int PB = C == 16 ? 32 : 2 * C + 16;   // bytes per pixel in LDS: 16 consecutive pixels of a fragment read fall into disjoint banks (16 channels: the bare 32
                                                           // bytes already do, and at 37 KB per image the forward of a run and a frozen teacher's eval launch -- stage.hip --
                                                           // fit one compute unit together: 79 + 76 of 160 KB)
    static constexpr int BUF = P * P * PB;                 // one activation buffer
    static constexpr int NPT = HW * HW / 16;               // pixel tiles of 16
    static constexpr int KT = C / 16;                      // output-channel tiles of 16
    static constexpr int WK = C == 64 ? 4 : 1;             // waves along the output channels
    static constexpr int WP = 4 / WK;                      // waves along the pixel tiles
    static constexpr int PTW = NPT / WP;                   // pixel tiles per wave
    static constexpr int KTW = KT / WK;                    // channel tiles per wave
    static constexpr int KS = (9 * C + 31) / 32;           // MFMA K steps
    static constexpr int AUX = 2 * C * 4 + 8 * C * 4 + 2 * C * 4 + 2 * C * 8 + kXchScratchDoubles * 8;      // tab, red, vals, tot, scratch
    static constexpr int LDS_FWD = 2 * BUF + AUX;
};

// ENTRY: the launch starts with the stage's DOWN-SAMPLING block (resnet.py:289-316 with stride 2 and the 1x1 / stride-2 shortcut of :299-305, built at :383 / :386):
// the input image (2 HW x 2 HW pixels, C / 2 channels) lands in a third LDS buffer, the block's first convolution (3x3 / stride 2) and the shortcut convolution read it
// at strided pixel positions with the output tiling of every other convolution of the run, each with its own statistics exchange; the block's second convolution is
// then an ordinary "second" convolution.  p.c[] = [second convolution of the entry block, first / second convolutions of the following blocks ...] (nconv odd).
template <int C, int HW, bool ENTRY>
__global__ __launch_bounds__(256) void stage_train_fwd_kernel(const StFwdParams p) {
    using G = StGeo<C, HW>;
    constexpr int P = G::P, PB = G::PB, BUF = G::BUF, PTW = G::PTW, KTW = G::KTW, KS = G::KS, WK = G::WK;
    constexpr int LOGC = C == 16 ? 4 : (C == 32 ? 5 : 6);
    constexpr int CI = C / 2, HWI = 2 * HW, PI = HWI + 2, PBI = CI == 16 ? 32 : 2 * CI + 16, INB = ENTRY ? PI * PI * PBI : 0;      // the entry block's input image
    constexpr int KSA = (9 * CI + 31) / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* IN = smem;
    char* X = smem + INB;
    char* Y = X + BUF;
    float* tab = reinterpret_cast<float*>(Y + BUF);               // [C][2]: scale, shift of the convolution in flight
    float* red = tab + 2 * C;                                      // [4 waves][2][C]
    float* vals = red + 8 * C;                                     // [2][C]: this image's sums
    double* tot = reinterpret_cast<double*>(vals + 2 * C);         // [2][C]: the batch's sums
    double* scratch = tot + 2 * C;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int wp = wave / WK, wk = wave % WK;
    const int img = blockIdx.x;
    const unsigned base = xch_base(p.xb);

    // ---- zero the buffers (the halo rings stay zero for the whole run), land the input image
    for (int o = tid * 16; o < INB + 2 * BUF; o += 256 * 16) *reinterpret_cast<uint4*>(smem + o) = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    if constexpr (ENTRY) {
        constexpr int CPP = CI / 8;
        const uint4* src = reinterpret_cast<const uint4*>(p.x + (size_t)img * HWI * HWI * CI);
        for (int i = tid; i < HWI * HWI * CPP; i += 256) {
            const int q = i / CPP, cc = i - q * CPP;
            const int yy = q / HWI, xx = q - yy * HWI;
            *reinterpret_cast<uint4*>(IN + ((yy + 1) * PI + xx + 1) * PBI + cc * 16) = src[i];
        }
    } else {
        constexpr int CPP = C / 8;                                  // 16-byte chunks per pixel
        const uint4* src = reinterpret_cast<const uint4*>(p.x + (size_t)img * HW * HW * C);
        for (int i = tid; i < HW * HW * CPP; i += 256) {
            const int q = i / CPP, cc = i - q * CPP;
            const int yy = q / HW, xx = q - yy * HW;
            *reinterpret_cast<uint4*>(X + ((yy + 1) * P + xx + 1) * PB + cc * 16) = src[i];
        }
    }
    __syncthreads();

    int pbase[PTW];                                                 // byte offset of pixel (y, x) shifted to tap (0, 0) = padded (y, x)
    int pixq[PTW];                                                  // its index in the image
#pragma unroll
    for (int t = 0; t < PTW; ++t) {
        const int q = (wp * PTW + t) * 16 + l15;
        const int yy = q / HW, xx = q - yy * HW;
        pbase[t] = (yy * P + xx) * PB;
        pixq[t] = q;
    }
    int koff[KS];                                                   // byte offset of this lane's 8 K elements of step s inside the patch; -1: beyond 9 C (zero)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int kk = 32 * s + 8 * g;
        const int tap = kk >> LOGC, c0 = kk & (C - 1);
        const int dy = (tap * 11) >> 5, dx = tap - 3 * dy;           // tap / 3 for tap < 9
        koff[s] = kk < 9 * C ? (dy * P + dx) * PB + c0 * 2 : -1;
    }

    auto load_filters = [&](int cv, bf16x8_t (&wf)[KTW][KS]) {
#pragma unroll
        for (int kt = 0; kt < KTW; ++kt) {
            const bf16_t* wr = p.c[cv].w + (size_t)((wk * KTW + kt) * 16 + l15) * 9 * C;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int kk = 32 * s + 8 * g;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (kk < 9 * C) v = *reinterpret_cast<const uint4*>(wr + kk);
                wf[kt][s] = __builtin_bit_cast(bf16x8_t, v);
            }
        }
    };
    // everything behind a convolution's MFMA loop: z to global memory, the image's channel sums, the batch's sums (exchange `xi` of the launch), scale / shift, then
    // the BatchNorm [+ block input] [+ ReLU] of the accumulators into LDS buffer D.  mode 0: first convolution of a block (ReLU); 1: second (adds what D holds -- the
    // block input -- and writes the block output to global memory too); 2: the shortcut convolution of an entry block (no ReLU, into the block-input buffer)
    auto finish = [&](const StConv& cc, f32x4 (&acc)[PTW][KTW], float c_gamma, float c_beta, float c_rm, float c_rv, int xi, int mode, char* D, bool tr) {
        st_stamp(p.xb, tr, 1);
        unsigned zp[PTW][KTW][2];
        float sv[KTW * 8];
#pragma unroll
        for (int kt = 0; kt < KTW; ++kt) {
            const int ch = (wk * KTW + kt) * 16 + 4 * g;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int t = 0; t < PTW; ++t) { const float v = acc[t][kt][e]; s1 += v; s2 = fmaf(v, v, s2); }
                sv[kt * 8 + e] = s1; sv[kt * 8 + 4 + e] = s2;
            }
#pragma unroll
            for (int t = 0; t < PTW; ++t) {
                zp[t][kt][0] = pack_bf16x2(acc[t][kt][0], acc[t][kt][1]);
                zp[t][kt][1] = pack_bf16x2(acc[t][kt][2], acc[t][kt][3]);
                if (cc.z != nullptr) *reinterpret_cast<uint2*>(cc.z + ((size_t)img * HW * HW + pixq[t]) * C + ch) = make_uint2(zp[t][kt][0], zp[t][kt][1]);      // (nullptr: a forward no backward follows)
            }
        }
        row16_sum_n(sv);
        if (l15 == 0) {
#pragma unroll
            for (int kt = 0; kt < KTW; ++kt) {
                const int ch = (wk * KTW + kt) * 16 + 4 * g;
                *reinterpret_cast<float4*>(red + (wave * 2 + 0) * C + ch) = make_float4(sv[kt * 8], sv[kt * 8 + 1], sv[kt * 8 + 2], sv[kt * 8 + 3]);
                *reinterpret_cast<float4*>(red + (wave * 2 + 1) * C + ch) = make_float4(sv[kt * 8 + 4], sv[kt * 8 + 5], sv[kt * 8 + 6], sv[kt * 8 + 7]);
            }
        }
        __syncthreads();
        if (tid < 2 * C) {
            const int stat = tid / C, ch = tid - stat * C;
            float v;
            if (WK == 1) v = ((red[(0 * 2 + stat) * C + ch] + red[(1 * 2 + stat) * C + ch]) + red[(2 * 2 + stat) * C + ch]) + red[(3 * 2 + stat) * C + ch];
            else v = red[((ch >> 4) * 2 + stat) * C + ch];
            vals[tid] = v;
        }
        __syncthreads();
        st_stamp(p.xb, tr, 2);
        // ---- the batch's sums: identical fp64 totals in every workgroup
        const unsigned tag = base + (unsigned)xi + 1u;
        xch_begin(p.xb, img, p.N, 2 * C, tag, vals, scratch);
        st_stamp(p.xb, tr, 3);
        xch_end(p.xb, p.N, 2 * C, tag, tot, scratch);
        st_stamp(p.xb, tr, 4);
        if (tid < C) {
            // (the expressions of bn_apply_train_kernel's prologue, bn.hip / lazy_input_coefs, conv3.hip)
            const double mean = tot[tid] * p.invM;
            double var = tot[C + tid] * p.invM - mean * mean;
            if (var < 0.0) var = 0.0;
            const float istd = rsqrtf((float)var + p.eps);
            const float sc = c_gamma * istd;
            const float sh = c_beta - (float)mean * sc;
            tab[tid * 2] = sc;
            tab[tid * 2 + 1] = sh;
            if (img == 0) {
                cc.mean[tid] = (float)mean;
                cc.invstd[tid] = istd;
                cc.coef[tid] = sc;
                cc.coef[C + tid] = sh;
                cc.rm[tid] = (1.f - p.momentum) * c_rm + p.momentum * (float)mean;
                cc.rv[tid] = (1.f - p.momentum) * c_rv + p.momentum * (float)(var * p.unbias);
            }
        }
        __syncthreads();
        st_stamp(p.xb, tr, 5);
#pragma unroll
        for (int kt = 0; kt < KTW; ++kt) {
            const int ch = (wk * KTW + kt) * 16 + 4 * g;
            const float4 s01 = *reinterpret_cast<const float4*>(tab + ch * 2);
            const float4 s23 = *reinterpret_cast<const float4*>(tab + ch * 2 + 4);
#pragma unroll
            for (int t = 0; t < PTW; ++t) {
                const int o = pbase[t] + (P + 1) * PB;              // the pixel itself (interior position)
                float v[4];
                v[0] = fmaf(__uint_as_float(zp[t][kt][0] << 16), s01.x, s01.y);            // (fmaf, then + res, then max: bn_apply_train_kernel's order)
                v[1] = fmaf(__uint_as_float(zp[t][kt][0] & 0xffff0000u), s01.z, s01.w);
                v[2] = fmaf(__uint_as_float(zp[t][kt][1] << 16), s23.x, s23.y);
                v[3] = fmaf(__uint_as_float(zp[t][kt][1] & 0xffff0000u), s23.z, s23.w);
                if (mode == 1) {
                    const uint2 r = *reinterpret_cast<const uint2*>(D + o + ch * 2);
                    v[0] += __uint_as_float(r.x << 16); v[1] += __uint_as_float(r.x & 0xffff0000u);
                    v[2] += __uint_as_float(r.y << 16); v[3] += __uint_as_float(r.y & 0xffff0000u);
                }
                if (mode != 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                const uint2 out = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                *reinterpret_cast<uint2*>(D + o + ch * 2) = out;
                if (cc.y != nullptr) *reinterpret_cast<uint2*>(cc.y + ((size_t)img * HW * HW + pixq[t]) * C + ch) = out;
            }
        }
        __syncthreads();
        st_stamp(p.xb, tr, 6);
    };
    auto conv = [&](int cv, const bf16x8_t (&wf)[KTW][KS]) {
        const bool second = ENTRY ? !(cv & 1) : (cv & 1);           // second convolution of a block: reads Y, adds the block input from X, writes X
        const char* S = second ? Y : X;
        char* D = second ? X : Y;
        const StConv& cc = p.c[cv];
        const bool tr = p.trace > 0 && cv == p.trace && img == 0;
        st_stamp(p.xb, tr, 0);
        // (this channel's parameters -- workgroup 0: the running statistics too -- are requested before the MFMA loop: behind the exchange they would be an L2
        //  round trip on every workgroup's critical path, and workgroup 0's read-modify-write one on everybody's)
        float c_gamma = 0.f, c_beta = 0.f, c_rm = 0.f, c_rv = 0.f;
        if (tid < C) {
            c_gamma = cc.gamma[tid]; c_beta = cc.beta[tid];
            if (img == 0) { c_rm = cc.rm[tid]; c_rv = cc.rv[tid]; }
        }
        f32x4 acc[PTW][KTW];
#pragma unroll
        for (int t = 0; t < PTW; ++t)
#pragma unroll
            for (int kt = 0; kt < KTW; ++kt) acc[t][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        constexpr int SLOTS = KS * PTW;
        auto xread = [&](int n) {
            const int s_ = n / PTW, t_ = n - s_ * PTW;
            const int a = koff[s_] >= 0 ? pbase[t_] + koff[s_] : 0;
            return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(S + a));
        };
        // (pixel fragments RING - 1 slots ahead of their MFMAs: with one wave per SIMD nothing else hides the LDS latency)
        constexpr int RING = 8;
        bf16x8_t xr[RING];
#pragma unroll
        for (int n = 0; n < RING - 1; ++n) if (n < SLOTS) xr[n] = xread(n);
#pragma unroll
        for (int n = 0; n < SLOTS; ++n) {
            const int s_ = n / PTW, t_ = n - s_ * PTW;
            if (n + RING - 1 < SLOTS) xr[(n + RING - 1) % RING] = xread(n + RING - 1);
#pragma unroll
            for (int kt = 0; kt < KTW; ++kt) acc[t_][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kt][s_], xr[n % RING], acc[t_][kt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        finish(cc, acc, c_gamma, c_beta, c_rm, c_rv, (ENTRY ? 2 : 0) + cv, second ? 1 : 0, D, tr);
    };
    // the two strided convolutions of an entry block: output pixel (y, x) reads the input at (2 y + dy - 1, 2 x + dx - 1) = padded (2 y + dy, 2 x + dx) [3x3, pad 1] or
    // at (2 y, 2 x) = padded (2 y + 1, 2 x + 1) [1x1, pad 0]; K = tap * CI + channel
    auto entry_conv = [&](const StConv& cc, bool shortcut, int xi, char* D) {
        constexpr int KSE = KSA;                                    // (the 1x1 uses its first K step only)
        float c_gamma = 0.f, c_beta = 0.f, c_rm = 0.f, c_rv = 0.f;
        if (tid < C) {
            c_gamma = cc.gamma[tid]; c_beta = cc.beta[tid];
            if (img == 0) { c_rm = cc.rm[tid]; c_rv = cc.rv[tid]; }
        }
        const int ks = shortcut ? 1 : KSE, kmax = shortcut ? CI : 9 * CI;
        f32x4 acc[PTW][KTW];
#pragma unroll
        for (int t = 0; t < PTW; ++t)
#pragma unroll
            for (int kt = 0; kt < KTW; ++kt) acc[t][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < ks; ++s) {
            const int kk = 32 * s + 8 * g;
            const int tap = kk / CI, c0 = kk - tap * CI;
            const int dy = shortcut ? 1 : tap / 3, dx = shortcut ? 1 : tap - 3 * (tap / 3);
            const bool live = kk < kmax;
            bf16x8_t wf[KTW];
#pragma unroll
            for (int kt = 0; kt < KTW; ++kt) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (live) v = *reinterpret_cast<const uint4*>(cc.w + (size_t)((wk * KTW + kt) * 16 + l15) * (shortcut ? CI : 9 * CI) + kk);
                wf[kt] = __builtin_bit_cast(bf16x8_t, v);
            }
            const int ko = (dy * PI + dx) * PBI + c0 * 2;
#pragma unroll
            for (int t = 0; t < PTW; ++t) {
                const int q = pixq[t], yy = q / HW, xx = q - yy * HW;
                uint4 xv = make_uint4(0u, 0u, 0u, 0u);
                if (live) xv = *reinterpret_cast<const uint4*>(IN + (2 * yy * PI + 2 * xx) * PBI + ko);
#pragma unroll
                for (int kt = 0; kt < KTW; ++kt) acc[t][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kt], __builtin_bit_cast(bf16x8_t, xv), acc[t][kt], 0, 0, 0);
            }
        }
        finish(cc, acc, c_gamma, c_beta, c_rm, c_rv, xi, shortcut ? 2 : 0, D, false);
    };
    bf16x8_t wa[KTW][KS], wb[KTW][KS];
    if constexpr (ENTRY) {
        load_filters(0, wa);
        entry_conv(p.ea, false, 0, Y);                              // conv -> BN -> ReLU into Y
        entry_conv(p.ed, true, 1, X);                               // shortcut conv -> BN into X: the "block input" the second convolution adds
        conv(0, wa);
        for (int cv = 1; cv + 1 < p.nconv; cv += 2) {               // (nconv is odd: the entry block's second convolution, then pairs)
            load_filters(cv, wb);
            load_filters(cv + 1, wa);
            conv(cv, wb);
            conv(cv + 1, wa);
        }
        if (img == 0 && tid == 0) xch_advance(p.xb, base, (unsigned)p.nconv + 2u);
    } else {
        load_filters(0, wa);
        for (int cv = 0; cv < p.nconv; cv += 2) {                   // (nconv is even: pairs of convolutions = BasicBlocks)
            load_filters(cv + 1, wb);
            conv(cv, wa);
            if (cv + 2 < p.nconv) load_filters(cv + 2, wa);
            conv(cv + 1, wb);
        }
        if (img == 0 && tid == 0) xch_advance(p.xb, base, (unsigned)p.nconv);
    }
    if (p.feat != nullptr && tid < C) {
        // the global average pooling behind the run: its output sits in X (every run ends on a block's second convolution); summed in pixel order like
        // avgpool_fwd_kernel (bn.hip), so the features are bit-identical to the pooling launch's
        float s = 0.f;
#pragma unroll 8
        for (int px = 0; px < HW * HW; ++px) {
            const int yy = px / HW, xx = px - yy * HW;
            s += __uint_as_float((unsigned)*reinterpret_cast<const unsigned short*>(X + ((yy + 1) * P + xx + 1) * PB + tid * 2) << 16);
        }
        p.feat[(size_t)img * C + tid] = s / (float)(HW * HW);
    }
}

template <int C, int HW, bool ENTRY>
int launch_fwd(const StFwdParams& p, hipStream_t st) {
    constexpr int CI = C / 2, PI = 2 * HW + 2, PBI = CI == 16 ? 32 : 2 * CI + 16;
    constexpr int lds = StGeo<C, HW>::LDS_FWD + (ENTRY ? PI * PI * PBI : 0);
    int dev = 0;
    (void)hipGetDevice(&dev);
    static bool attr[16] = {};
    if (dev < 0 || dev >= 16 || !attr[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(stage_train_fwd_kernel<C, HW, ENTRY>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            clhip_set_error("stage_train_fwd: cannot reserve %d bytes of LDS", lds);
            return CLHIP_EHIP;
        }
        if (dev >= 0 && dev < 16) attr[dev] = true;
    }
    hipLaunchKernelGGL((stage_train_fwd_kernel<C, HW, ENTRY>), dim3(p.N), dim3(256), lds, st, p);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

// =============================================================================================== backward
// One workgroup per image again.  Per unit cv (last to first), with `acc` = the gradient of the unit's output in registers (the MFMA result layout: lane (l15, g) holds
// channels 4 g .. 4 g + 3 of pixel l15 of each of its tiles -- the layout the forward's epilogue and the previous dgrad's result share, so the residual gradient of a
// block simply stays in registers):
//   A  g = dy * ReLU mask (block outputs: y > 0; first convolution of a block: sign of scale z + shift), per-image sums of g and g * xhat
//   B  the batch's sums through the in-launch all-reduce (xch.h) -- and WHILE they travel:
//   C  the weight gradient of unit cv + 1 (its dz and its input image are still in LDS), per-image partial sums to a slab that the plan's fixed-order reduce adds up
//   D  dz = gamma invstd (g - mean(g) - xhat mean(g xhat)) into the zero-haloed LDS buffer
//   E  dgrad = the same implicit GEMM as the forward with the [C][9][K] weight copy and mirrored taps; + the residual gradient at the top of a block
// Rounding points follow the per-unit path: dy and dz pass through bf16, sums and accumulators are fp32 / fp64.
struct StConvB {
    const bf16_t* wd;         // [C][9][C] dgrad copy
    const float* gamma;
    const float* beta;
    const float* mean;        // saved by the forward
    const float* invstd;
    const bf16_t* z;
    const bf16_t* y;          // block outputs (odd positions of the run): the activation; nullptr at even positions
    float* dgamma;
    float* dbeta;
    float* slab;              // weight-gradient partial sums: [N][C * 9 * C], one block per image (16 channels); [groups][C * 9 * C] (wider: StGrp)
    bf16_t* dzg;              // (group weight gradient) [C / 16][N][HW][HW][16]: this unit's dz in global memory, channel-tile-major, for other workgroups' work items
};

struct StBwdParams {
    const bf16_t* x;          // the run's input activation
    const bf16_t* dy;         // gradient of the run's output activation
    bf16_t* dx;               // gradient of the run's input activation (written, or accumulated into when dx_acc)
    int dx_acc;
    int N, nconv;
    double invM;
    int trace;
    const float* dfeat;       // nullable: [N][C] gradient of the pooled features -- the run's output gradient is then dfeat / (HW HW) at every pixel and dy is not read
    XchBuf xb;
    StConvB ea, ed;           // (entry block) its 3x3 / stride-2 convolution and its shortcut convolution: wd = [C / 2][9][C] / [C / 2][1][C], slab = [N][C][9][C / 2] / [N][C][C / 2];
                              // x is then the block's INPUT [N][2 HW][2 HW][C / 2] and dx its gradient
    StConvB c[kMaxConvT];
};

// the GROUP weight gradient of the 32- / 64-channel stages: a per-image partial block would be 36 / 147 KB (more than the image's operands: 302 MB of partial sums per
// stage-3 run at batch 256), so there a workgroup takes one (output-channel tile, input-channel tile) pair of a GROUP of IPG images -- 16 x 16 x 9 sums = 9 KB whatever
// the stage -- and fetches the 16-channel slices of the group's dz and input images from global memory into LDS.
template <int C, int HW>
struct StGrp {
    static constexpr int IPG = (C / 16) * (C / 16);                  // images per group = number of (out tile, in tile) pairs: 4 / 16
    static constexpr int P2 = HW + 2;
    static constexpr int XS = P2 * P2 * 32, ZS = HW * HW * 32;       // one image's padded input slice / gradient slice (16 channels = 32 bytes per pixel)
    static constexpr int IMG = XS + ZS;
    static constexpr int KPI = HW * HW / 32;                         // K steps (32 pixels) per image: 8 / 2
    static constexpr int BYTES = IPG * IMG + 2 * 5 * 64 * 16;        // staging + the K halves' hand-over
};

template <int C, int HW>
struct StEnt {                                                     // the input image of an entry block and the shortcut's dz, behind everything else in LDS
    static constexpr int CI = C / 2, HWI = 2 * HW, PI = HWI + 2, PBI = CI == 16 ? 32 : 2 * CI + 16;
    static constexpr int INB = PI * PI * PBI;
    static constexpr int BYTES = INB + StGeo<C, HW>::BUF;
};

template <int C, int HW, bool GRP>
struct StGeoB {
    using G = StGeo<C, HW>;
    static constexpr int AUX = 8 * C * 4 + 8 * C * 4 + 2 * C * 4 + 2 * C * 8 + kXchScratchDoubles * 8;      // ctab, red, vals, tot, scratch
    // own-image weight gradient: dz + the input image of the unit whose weight gradient is pending (+ the cross-wave sum of the 16-channel form); group form: dz +
    // the staging area
    static constexpr int LDS = GRP ? G::BUF + AUX + StGrp<C, HW>::BYTES : 2 * G::BUF + AUX + (C == 16 ? 4 * 2304 * 4 + HW * HW * 32 : 0);      // (16 channels: + the residual gradient of the block in flight, parked in LDS)
};

__device__ __forceinline__ uint4 st_tr8(const char* base, int addr, int second) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + addr));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + addr + second));
    uint2 l = __builtin_bit_cast(uint2, lo), h = __builtin_bit_cast(uint2, hi);
    return make_uint4(l.x, l.y, h.x, h.y);
}

// dw[k][tap][c] of ONE 16-channel image (stage 1): A = dz^T (rows = output channels, reduction = 32 pixels = one image row), B = the input image shifted by the tap,
// both fetched from the pixel-major padded LDS images with transposing reads (ds_read_b64_tr_b16: 16 lanes fetch a [4 pixels][16 channels] block, lane i keeps
// channel i).  The reduction index is a free permutation as long as both operands use the same one: lane group fg takes pixels 4 fg .. 4 fg + 3 and 16 + 4 fg .. of
// the row, so the four groups of a read cover 512 contiguous bytes (32-byte pixels: every bank once).  Wave w takes HW / 4 consecutive
// rows and the four partial tiles are summed through LDS.  `slab`: this image's block.
template <int HW>
__device__ __forceinline__ void st_wgrad16(const char* D, const char* XA, float* red, float* slab) {
    using G = StGeo<16, HW>;
    constexpr int P = G::P, PB = G::PB, RPW = HW / 4;               // rows per wave: CONSECUTIVE rows, so that an input row fetched for tap row r serves r - 1 and r - 2 too
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int seg = (fr & 3) * 8;
    const int col = fg * 4 + (fr >> 2);
    const int row0 = wave * RPW;
    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // input rows in a ring of four slots (local row j in slot (j + 1) & 3), fetched two rows ahead of their first MFMA; the gradient row one ahead
    uint4 X[4][3], zf[2];
    auto xrow = [&](int j, int slot) {                              // image row row0 + j = padded row row0 + j + 1 (rows -1 and HW are the zero halo)
        const int pb = ((row0 + j + 1) * P + col + 1) * PB + seg;
#pragma unroll
        for (int sx = 0; sx < 3; ++sx) X[slot][sx] = st_tr8(XA, pb + (sx - 1) * PB, 16 * PB);
    };
    auto zrow = [&](int j, int b) { zf[b] = st_tr8(D, ((row0 + j + 1) * P + col + 1) * PB + seg, 16 * PB); };
    xrow(-1, 0); xrow(0, 1); xrow(1, 2);
    zrow(0, 0);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        if (i + 2 <= RPW) xrow(i + 2, (i + 3) & 3);
        if (i + 1 < RPW) zrow(i + 1, (i + 1) & 1);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int sx = 0; sx < 3; ++sx)
                acc[r * 3 + sx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, zf[i & 1]), __builtin_bit_cast(bf16x8_t, X[(i + r) & 3][sx]), acc[r * 3 + sx], 0, 0, 0);
    }
    // D[row = out channel fg * 4 + e][col = in channel fr]
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave * 2304 + ((fg * 4 + e) * 9 + t) * 16 + fr] = acc[t][e];
    __syncthreads();
    for (int i = tid; i < 2304; i += 256) slab[i] = ((red[i] + red[2304 + i]) + red[4608 + i]) + red[6912 + i];
}

// the same for the wider stages (one image, everything from this workgroup's LDS; the partial block is 36 / 147 KB per image -- fine for small batches, see StGrp):
//   32 channels: a K step = two image rows of 16 pixels; wave = (input-channel tile, tap half) over all K steps and both output-channel tiles: no cross-wave sum
//   64 channels: a K step = four image rows of 8 pixels; wave = input-channel tile, nine taps, the four output-channel tiles in two passes of two
template <int C, int HW>
__device__ __forceinline__ void st_wgrad_wide(const char* D, const char* XA, float* slab) {
    using G = StGeo<C, HW>;
    constexpr int P = G::P, PB = G::PB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int seg = (fr & 3) * 8;
    if constexpr (C == 32) {
        const int it = wave & 1, th = wave >> 1, t0 = th * 5, nt = th == 0 ? 5 : 4;
        const int prow = fg >> 1, pcol = (fg & 1) * 8 + (fr >> 2);
        f32x4 acc[2][5];
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int q = 0; q < 5; ++q) acc[o][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int h0 = 0; h0 < HW; h0 += 2) {
            const int pb = ((h0 + prow + 1) * P + pcol + 1) * PB + seg;
            uint4 zf[2];
#pragma unroll
            for (int o = 0; o < 2; ++o) zf[o] = st_tr8(D, pb + o * 32, 4 * PB);
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                if (q < nt) {
                    const int t = t0 + q, r = t / 3, sx = t - 3 * r;
                    const uint4 xf = st_tr8(XA, pb + ((r - 1) * P + (sx - 1)) * PB + it * 32, 4 * PB);
#pragma unroll
                    for (int o = 0; o < 2; ++o)
                        acc[o][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, zf[o]), __builtin_bit_cast(bf16x8_t, xf), acc[o][q], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int q = 0; q < 5; ++q)
                if (q < nt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) slab[((o * 16 + fg * 4 + e) * 9 + t0 + q) * 32 + it * 16 + fr] = acc[o][q][e];
    } else {
        const int it = wave, prow = fg, pcol = fr >> 2;
#pragma unroll 1
        for (int op = 0; op < 2; ++op) {
            f32x4 acc[2][9];
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[o][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int h0 = 0; h0 < HW; h0 += 4) {
                const int pb = ((h0 + prow + 1) * P + pcol + 1) * PB + seg;
                uint4 zf[2];
#pragma unroll
                for (int o = 0; o < 2; ++o) zf[o] = st_tr8(D, pb + (op * 2 + o) * 32, 4 * PB);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int r = t / 3, sx = t - 3 * r;
                    const uint4 xf = st_tr8(XA, pb + ((r - 1) * P + (sx - 1)) * PB + it * 32, 4 * PB);
#pragma unroll
                    for (int o = 0; o < 2; ++o)
                        acc[o][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, zf[o]), __builtin_bit_cast(bf16x8_t, xf), acc[o][t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) slab[(((op * 2 + o) * 16 + fg * 4 + e) * 9 + t) * 64 + it * 16 + fr] = acc[o][t][e];
        }
    }
}

// the operands of one work item of the group weight gradient, requested into registers: chunk q = tid + 256 k of the item's 4096 16-byte chunks (per image: the
// input slice, then the gradient slice).  `xsrc`: the input tensor (an activation, or the z of the block's first convolution whose relu(scale z + shift) is taken on
// the way into LDS); `dzrs`: buffer descriptor of the unit's dz tensor, written by OTHER workgroups of this launch with write-through stores: sc1 loads.
template <int C, int HW>
struct StGrpRegs { uint4 v[16]; };

template <int C, int HW>
__device__ __forceinline__ void st_grp_fetch(StGrpRegs<C, HW>& r, const bf16_t* xsrc, __amdgpu_buffer_rsrc_t dzrs, int item, int N) {
    using Q = StGrp<C, HW>;
    constexpr int CPI = HW * HW * 4;                                 // chunks per image: 2 per pixel of either slice
    const int combo = item % Q::IPG, grp = item / Q::IPG;
    const int ot = combo / (C / 16), it = combo % (C / 16);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int q = threadIdx.x + 256 * k;
        const int j = q / CPI, w = q - j * CPI;                      // image of the group, chunk inside the image's pair of slices
        const int n = grp * Q::IPG + j;
        const bool isz = w >= HW * HW * 2;
        const int pc = isz ? w - HW * HW * 2 : w;
        const int pix = pc >> 1, half = pc & 1;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (n < N) {
            // (dz is kept channel-tile-major -- [out tile][image][pixel][16] -- so that a slice is contiguous; the input tensors are the forward's pixel-major ones)
            if (isz) v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(dzrs, (int)((((size_t)ot * N + n) * HW * HW + pix) * 32 + half * 16), 0, 16));
            else v = *reinterpret_cast<const uint4*>(xsrc + ((size_t)n * HW * HW + pix) * C + it * 16 + half * 8);
        }
        r.v[k] = v;
    }
}

// ... and the item itself: registers -> LDS (x slices zero-haloed: the halo of the staging area is zeroed once per launch and never written), then 32 K steps of 32
// pixels x nine taps, the K STEPS split over the four waves (splitting the taps instead makes every wave read every gradient fragment: the loop is LDS-bandwidth
// bound and ran 4 x longer), the four partial tile sets summed through `red` (its own 36 KB, 16-byte writes: lane (fr, fg) holds four consecutive output channels).
// lazy: x = relu(sc z + sh), coef[0..7] / coef[8..15] = scale / shift of this thread's eight channels (thread-constant: chunk parity and the item's input tile fix them).
template <int C, int HW>
__device__ __forceinline__ void st_grp_run(const StGrpRegs<C, HW>& r, char* stg, float* red, const float* coef, bool lazy, int item, float* slab_unit, const XchBuf& tb, bool tr) {
    using Q = StGrp<C, HW>;
    constexpr int CPI = HW * HW * 4, P2 = Q::P2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int combo = item % Q::IPG, grp = item / Q::IPG;
    const int ot = combo / (C / 16), it = combo % (C / 16);
    st_stamp(tb, tr, 17);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int q = tid + 256 * k;
        const int j = q / CPI, w = q - j * CPI;
        const bool isz = w >= HW * HW * 2;
        const int pc = isz ? w - HW * HW * 2 : w;
        const int pix = pc >> 1, half = pc & 1;
        const int yy = pix / HW, xx = pix - yy * HW;
        uint4 v = r.v[k];
        if (!isz && lazy) v = bn_relu8_bf16(v, coef, coef + 8);
        char* dst = stg + j * Q::IMG + (isz ? Q::XS + pix * 32 : ((yy + 1) * P2 + xx + 1) * 32) + half * 16;
        *reinterpret_cast<uint4*>(dst) = v;
    }
    __syncthreads();
    st_stamp(tb, tr, 18);
    // wave = (K half kh, tap group tg): 16 K steps x 5 (taps 0..4) or 4 (taps 5..8) MFMAs; the two K halves of a tap group meet through LDS in the lane layout they
    // already have (one 16-byte write / read per tile, no index arithmetic), and the kh = 0 wave stores the sums
    const int kh = wave & 1, tg = wave >> 1, t0 = tg * 5, nt = tg == 0 ? 5 : 4;
    f32x4 acc[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int seg = (fr & 3) * 8;
    // reduction index of a K step (a free permutation, the same for both operands; chosen so that the four lane groups of a read touch disjoint banks):
    //   16-pixel rows: rows 2 s, 2 s + 1; group fg = columns 4 fg .. 4 fg + 3 of the first (first read) and of the second row (second read)
    //    8-pixel rows: rows 4 s .. 4 s + 3; group fg = columns 4 (fg & 1) .. of row (fg >> 1) (first read) and of row (fg >> 1) + 2 (second read)
    const int row0 = HW == 16 ? 0 : (fg >> 1), col0 = HW == 16 ? 4 * fg + (fr >> 2) : 4 * (fg & 1) + (fr >> 2);
    constexpr int RSTEP = HW == 16 ? 2 : 4, SECOND = HW == 16 ? 1 : 2;      // rows per K step; row distance of the second read
    int toff[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int t = t0 + (q < nt ? q : 0), rr = t / 3, sx = t - 3 * rr;
        toff[q] = ((rr - 1) * P2 + (sx - 1)) * 32;
    }
    uint4 zf[2], xf[2][5];
    auto fetch = [&](int i, int b) {                                 // the wave's i-th K step = step kh * 16 + i of the item (consecutive steps: image by image)
        const int ks = kh * 16 + i;
        const int j = ks / Q::KPI, ls = ks - j * Q::KPI;
        const char* xs = stg + j * Q::IMG;
        const int row = ls * RSTEP + row0;
        zf[b] = st_tr8(xs + Q::XS, (row * HW + col0) * 32 + seg, SECOND * HW * 32);
        const int xb = ((row + 1) * P2 + col0 + 1) * 32 + seg;
#pragma unroll
        for (int q = 0; q < 5; ++q) xf[b][q] = st_tr8(xs, xb + toff[q], SECOND * P2 * 32);
    };
    fetch(0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int b = i & 1;
        if (i + 1 < 16) fetch(i + 1, b ^ 1);
#pragma unroll
        for (int q = 0; q < 5; ++q)
            if (q < nt) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, zf[b]), __builtin_bit_cast(bf16x8_t, xf[b][q]), acc[q], 0, 0, 0);
    }
    st_stamp(tb, tr, 19);
    float4* rq = reinterpret_cast<float4*>(red) + (tg * 5) * 64 + lane;
    if (kh == 1) {
#pragma unroll
        for (int q = 0; q < 5; ++q) rq[q * 64] = make_float4(acc[q][0], acc[q][1], acc[q][2], acc[q][3]);
    }
    __syncthreads();
    if (kh == 0) {
        // D[row = out channel fg * 4 + e][col = in channel fr] -> the group's block [K][9][C]
        float* out = slab_unit + (size_t)grp * (9 * C * C);
#pragma unroll
        for (int q = 0; q < 5; ++q)
            if (q < nt) {
                const float4 o = rq[q * 64];
                const float v[4] = {acc[q][0] + o.x, acc[q][1] + o.y, acc[q][2] + o.z, acc[q][3] + o.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) out[((ot * 16 + fg * 4 + e) * 9 + t0 + q) * C + it * 16 + fr] = v[e];
            }
    }
    st_stamp(tb, tr, 20);
    // (the next item's staging stores come after its own barrier-free prologue: the barrier at the top of the next call's MFMA phase orders them -- but this call's
    //  readers must be done before ANY restaging: one more barrier here)
    __syncthreads();
}

// ---- the backward of an entry block's two strided convolutions (ENTRY), everything from this workgroup's LDS: D / Dds = dz of the 3x3 / stride-2 convolution and of
// the 1x1 / stride-2 shortcut (HW x HW x C, zero-haloed), IN = the block's input image (2 HW x 2 HW x C / 2, zero-haloed).
// Weight gradients: dw[k][tap][ci] = sum over output pixels (y, x) of dz[y][x][k] * in[2 y + dy - 1][2 x + dx - 1][ci]  (padded coordinates (2 y + dy, 2 x + dx)); the
// shortcut's single tap reads in[2 y][2 x].  Transposing reads as in the other weight gradients; a lane's four-pixel stride in the input image is eight pixels.
template <int C, int HW>
__device__ __forceinline__ void st_wgrad_entry(const char* D, const char* Dds, const char* IN, float* slab_a, float* slab_d) {
    using G = StGeo<C, HW>;
    using E = StEnt<C, HW>;
    constexpr int P = G::P, PB = G::PB, CI = E::CI, PI = E::PI, PBI = E::PBI;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int seg = (fr & 3) * 8;
    if constexpr (C == 32) {
        // K step = two output rows of 16 pixels; wave = (output-channel tile, tap half); one input-channel tile
        const int ot = wave & 1, th = wave >> 1, t0 = th * 5, nt = th == 0 ? 5 : 4;
        const int prow = fg >> 1, pcol = (fg & 1) * 8 + (fr >> 2);
        f32x4 acc[5], accd = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 5; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int h0 = 0; h0 < HW; h0 += 2) {
            const int pd = ((h0 + prow + 1) * P + pcol + 1) * PB + ot * 32 + seg;
            const int pi = (2 * (h0 + prow) * PI + 2 * pcol) * PBI + seg;
            const uint4 zf = st_tr8(D, pd, 4 * PB);
#pragma unroll
            for (int q = 0; q < 5; ++q)
                if (q < nt) {
                    const int t = t0 + q, dy = t / 3, dx = t - 3 * dy;
                    const uint4 xf = st_tr8(IN, pi + (dy * PI + dx) * PBI, 8 * PBI);
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, zf), __builtin_bit_cast(bf16x8_t, xf), acc[q], 0, 0, 0);
                }
            if (th == 0) {                                          // (the shortcut's weight gradient rides on the tap-half-0 waves)
                const uint4 zd = st_tr8(Dds, pd, 4 * PB);
                const uint4 xf = st_tr8(IN, pi + (PI + 1) * PBI, 8 * PBI);
                accd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, zd), __builtin_bit_cast(bf16x8_t, xf), accd, 0, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < 5; ++q)
            if (q < nt)
#pragma unroll
                for (int e = 0; e < 4; ++e) slab_a[((ot * 16 + fg * 4 + e) * 9 + t0 + q) * CI + fr] = acc[q][e];
        if (th == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) slab_d[(ot * 16 + fg * 4 + e) * CI + fr] = accd[e];
        }
    } else {
        // K step = four output rows of 8 pixels (two per image); wave = output-channel tile, both input-channel tiles, nine taps + the shortcut's
        const int ot = wave, prow = fg, pcol = fr >> 2;
        f32x4 acc[2][9], accd[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            accd[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int h0 = 0; h0 < HW; h0 += 4) {
            const int pd = ((h0 + prow + 1) * P + pcol + 1) * PB + ot * 32 + seg;
            const int pi = (2 * (h0 + prow) * PI + 2 * pcol) * PBI + seg;
            const uint4 zf = st_tr8(D, pd, 4 * PB), zd = st_tr8(Dds, pd, 4 * PB);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int dy = t / 3, dx = t - 3 * dy;
                    const uint4 xf = st_tr8(IN, pi + (dy * PI + dx) * PBI + i * 32, 8 * PBI);
                    acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, zf), __builtin_bit_cast(bf16x8_t, xf), acc[i][t], 0, 0, 0);
                    if (t == 4) accd[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, zd), __builtin_bit_cast(bf16x8_t, xf), accd[i], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) slab_a[((ot * 16 + fg * 4 + e) * 9 + t) * CI + i * 16 + fr] = acc[i][t][e];
#pragma unroll
            for (int e = 0; e < 4; ++e) slab_d[(ot * 16 + fg * 4 + e) * CI + i * 16 + fr] = accd[i][e];
        }
    }
}

// Input gradient of the entry block: in[i][j] receives, from the 3x3 / stride-2 convolution, the taps (dy, dx) with i + 1 - dy and j + 1 - dx even,
//     dz[(i + 1 - dy) / 2][(j + 1 - dx) / 2] . w[dy][dx],
// and at even (i, j) the shortcut's dz_ds[i / 2][j / 2] . w_ds.  Sorted by the parity of (i, j) that is four small convolutions over the HW x HW grids of D / Dds (1 + 1,
// 2, 2 and 4 taps -- a zero-stuffed 3x3 would run 9 taps on four times the pixels), each writing one pixel of every 2 x 2 cell of dx.  wa = [C / 2][9][C], wd = [C / 2][C].
template <int C, int HW>
__device__ __forceinline__ void st_dgrad_entry(const char* D, const char* Dds, const bf16_t* wa, const bf16_t* wd, bf16_t* dx_img, int dx_acc) {
    using G = StGeo<C, HW>;
    using E = StEnt<C, HW>;
    constexpr int P = G::P, PB = G::PB, CI = E::CI, HWI = E::HWI;
    constexpr int KTC = CI / 16;                                    // output (input-channel) tiles: 1 / 2
    constexpr int PTC = (HW * HW / 16) / 4;                         // pixel tiles per wave: 4 / 1
    constexpr int KSC = C / 32;                                     // K steps per tap: 1 / 2
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int cls = 0; cls < 4; ++cls) {
        const int pi = cls >> 1, pj = cls & 1;
        f32x4 acc[PTC][KTC];
#pragma unroll
        for (int t = 0; t < PTC; ++t)
#pragma unroll
            for (int kt = 0; kt < KTC; ++kt) acc[t][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // taps of this class: dy in {1} (pi = 0) or {0, 2} (pi = 1), likewise dx; source pixel offset (pi + 1 - dy) / 2
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            if (pi == 0 && a == 1) continue;
            const int dy = pi == 0 ? 1 : 2 * a, oy = (pi + 1 - dy) / 2;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                if (pj == 0 && b == 1) continue;
                const int dx = pj == 0 ? 1 : 2 * b, ox = (pj + 1 - dx) / 2;
                const int tap = dy * 3 + dx;
#pragma unroll
                for (int s = 0; s < KSC; ++s) {
                    const int kk = 32 * s + 8 * g;
                    bf16x8_t wf[KTC];
#pragma unroll
                    for (int kt = 0; kt < KTC; ++kt) wf[kt] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wa + ((size_t)(kt * 16 + l15) * 9 + tap) * C + kk));
#pragma unroll
                    for (int t = 0; t < PTC; ++t) {
                        const int q = (wave * PTC + t) * 16 + l15, yy = q / HW, xx = q - yy * HW;
                        const uint4 xv = *reinterpret_cast<const uint4*>(D + ((yy + oy + 1) * P + xx + ox + 1) * PB + kk * 2);
#pragma unroll
                        for (int kt = 0; kt < KTC; ++kt) acc[t][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kt], __builtin_bit_cast(bf16x8_t, xv), acc[t][kt], 0, 0, 0);
                    }
                }
            }
        }
        if (cls == 0) {                                             // the shortcut's input gradient lands on the even pixels
#pragma unroll
            for (int s = 0; s < KSC; ++s) {
                const int kk = 32 * s + 8 * g;
                bf16x8_t wf[KTC];
#pragma unroll
                for (int kt = 0; kt < KTC; ++kt) wf[kt] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wd + (size_t)(kt * 16 + l15) * C + kk));
#pragma unroll
                for (int t = 0; t < PTC; ++t) {
                    const int q = (wave * PTC + t) * 16 + l15, yy = q / HW, xx = q - yy * HW;
                    const uint4 xv = *reinterpret_cast<const uint4*>(Dds + ((yy + 1) * P + xx + 1) * PB + kk * 2);
#pragma unroll
                    for (int kt = 0; kt < KTC; ++kt) acc[t][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kt], __builtin_bit_cast(bf16x8_t, xv), acc[t][kt], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < PTC; ++t) {
            const int q = (wave * PTC + t) * 16 + l15, yy = q / HW, xx = q - yy * HW;
#pragma unroll
            for (int kt = 0; kt < KTC; ++kt) {
                bf16_t* o = dx_img + ((size_t)(2 * yy + pi) * HWI + 2 * xx + pj) * CI + kt * 16 + 4 * g;
                float v0 = acc[t][kt][0], v1 = acc[t][kt][1], v2 = acc[t][kt][2], v3 = acc[t][kt][3];
                if (dx_acc) {
                    const uint2 old = *reinterpret_cast<const uint2*>(o);
                    v0 += __uint_as_float(old.x << 16); v1 += __uint_as_float(old.x & 0xffff0000u);
                    v2 += __uint_as_float(old.y << 16); v3 += __uint_as_float(old.y & 0xffff0000u);
                }
                *reinterpret_cast<uint2*>(o) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
            }
        }
    }
}

// ENTRY: p.c[] = [second convolution of the stage's down-sampling block, first / second convolutions of the blocks behind it ...] (nconv odd), and behind unit 0 the
// launch runs that block's 3x3 / stride-2 convolution (p.ea) and shortcut (p.ed) as well: BatchNorm backward of both (two more exchanges, the pending weight gradients in
// their shadow), the input gradient by pixel parity (st_dgrad_entry) and the two strided weight gradients (st_wgrad_entry).
template <int C, int HW, bool GRP, bool ENTRY>
__global__ __launch_bounds__(256) void stage_train_bwd_kernel(const StBwdParams p) {
    using G = StGeo<C, HW>;
    using GB = StGeoB<C, HW, GRP>;
    constexpr int LDS_ALL = GB::LDS + (ENTRY ? StEnt<C, HW>::BYTES : 0);
    auto is_second = [](int cv) { return ENTRY ? !(cv & 1) : (cv & 1) != 0; };      // second convolution of a block (block output, residual add)
    constexpr int P = G::P, PB = G::PB, BUF = G::BUF, PTW = G::PTW, KTW = G::KTW, KS = G::KS, WK = G::WK;
    constexpr int LOGC = C == 16 ? 4 : (C == 32 ? 5 : 6);
    constexpr int CPP = C / 8;
    constexpr int NCH = HW * HW * CPP / 256;                       // 16-byte chunks of an image per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NBUF = GRP ? 1 : 2;                               // (GRP: the group weight gradient -- StGrp -- instead of the own-image one)
    char* D = smem;                                                // dz of the unit in flight, zero-haloed
    char* XA = smem + BUF;                                         // (16 channels) input image of the unit whose weight gradient is pending, zero-haloed
    float* ctab = reinterpret_cast<float*>(smem + NBUF * BUF);     // [5][C]: mean(g), mean(g xhat), -, -, gamma invstd
    float* red = ctab + 8 * C;                                     // [4 waves][2][C]
    float* vals = red + 8 * C;                                     // [2][C]
    double* tot = reinterpret_cast<double*>(vals + 2 * C);
    double* scratch = tot + 2 * C;
    char* stg = smem + NBUF * BUF + GB::AUX;                       // 16 channels: the cross-wave sum of the weight gradient; wider: the group staging area
    float* wred = reinterpret_cast<float*>(stg);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4;
    const int wp = wave / WK, wk = wave % WK;
    const int img = blockIdx.x;
    const unsigned base = xch_base(p.xb);
    const size_t ibase = (size_t)img * HW * HW;

    for (int o = tid * 16; o < LDS_ALL; o += 256 * 16) *reinterpret_cast<uint4*>(smem + o) = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    // the convolution whose output feeds unit cv (its z is the lazy input of a block's second convolution; its y the input of a block's first one)
    auto prevc = [&](int cv) -> const StConvB& { return (ENTRY && cv == 0) ? p.ea : p.c[cv > 0 ? cv - 1 : 0]; };

    // the group weight gradient of unit u (32 / 64 channels): this workgroup's work items -- (out tile, in tile) pair x group of images -- from global memory.  The first
    // item's operands are REQUESTED (grp_issue) a stage before they are used (grp_finish): a batch of sixteen 16-byte loads per thread, a quarter of them write-through
    // data of other workgroups (sc1: served past the L2), is 2-3 us of latency with every workgroup asking at once
    StGrpRegs<C, HW> grp_regs;
    float4 grp_raw[8];                                              // gamma, invstd, beta, mean of this thread's eight input channels (lazy input), requested AHEAD of the big batch:
                                                                    // waiting for them must not mean waiting for the sixteen 16-byte loads behind them
    auto grp_issue = [&](int u, int item) __attribute__((always_inline)) {
        if constexpr (GRP) {
            using Q = StGrp<C, HW>;
            const bool lazy = is_second(u);
            const bf16_t* xsrc = lazy ? prevc(u).z : (u == 0 ? p.x : p.c[u - 1].y);
            const __amdgpu_buffer_rsrc_t dzrs = __builtin_amdgcn_make_buffer_rsrc(p.c[u].dzg, 0, p.N * HW * HW * C * 2, 0x00020000);
            if (lazy) {
                const StConvB& a = prevc(u);
                const int c0 = ((item % Q::IPG) % (C / 16)) * 16 + (tid & 1) * 8;
                grp_raw[0] = *reinterpret_cast<const float4*>(a.gamma + c0); grp_raw[1] = *reinterpret_cast<const float4*>(a.gamma + c0 + 4);
                grp_raw[2] = *reinterpret_cast<const float4*>(a.invstd + c0); grp_raw[3] = *reinterpret_cast<const float4*>(a.invstd + c0 + 4);
                grp_raw[4] = *reinterpret_cast<const float4*>(a.beta + c0); grp_raw[5] = *reinterpret_cast<const float4*>(a.beta + c0 + 4);
                grp_raw[6] = *reinterpret_cast<const float4*>(a.mean + c0); grp_raw[7] = *reinterpret_cast<const float4*>(a.mean + c0 + 4);
            }
            st_grp_fetch<C, HW>(grp_regs, xsrc, dzrs, item, p.N);
        }
    };
    auto grp_finish = [&](int u, bool tr) __attribute__((always_inline)) {
        if constexpr (GRP) {
            using Q = StGrp<C, HW>;
            const int nitems = Q::IPG * ((p.N + Q::IPG - 1) / Q::IPG);
            for (int item = img; item < nitems; item += p.N) {
                if (item != img) grp_issue(u, item);                 // (a second item: only when the batch is not a multiple of the group size)
                float coef[16];
                {
                    const float ga[8] = {grp_raw[0].x, grp_raw[0].y, grp_raw[0].z, grp_raw[0].w, grp_raw[1].x, grp_raw[1].y, grp_raw[1].z, grp_raw[1].w};
                    const float is[8] = {grp_raw[2].x, grp_raw[2].y, grp_raw[2].z, grp_raw[2].w, grp_raw[3].x, grp_raw[3].y, grp_raw[3].z, grp_raw[3].w};
                    const float be[8] = {grp_raw[4].x, grp_raw[4].y, grp_raw[4].z, grp_raw[4].w, grp_raw[5].x, grp_raw[5].y, grp_raw[5].z, grp_raw[5].w};
                    const float mu[8] = {grp_raw[6].x, grp_raw[6].y, grp_raw[6].z, grp_raw[6].w, grp_raw[7].x, grp_raw[7].y, grp_raw[7].z, grp_raw[7].w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) { coef[e] = ga[e] * is[e]; coef[8 + e] = be[e] - mu[e] * coef[e]; }
                }
                st_grp_run<C, HW>(grp_regs, stg, reinterpret_cast<float*>(stg + Q::IPG * Q::IMG), coef, is_second(u), item, p.c[u].slab, p.xb, tr);
            }
        }
    };

    // ---- state that lives across units.  Everything a unit needs from global memory is requested one stage ahead of its use:
    //   zq / yq / mu4 .. : z, the block output and the saved statistics of the NEXT unit to process at this lane's positions -- requested before the dgrad of the
    //                      unit above it multiplies;  xin: the unit's input image -- requested before the weight gradient of the unit above runs
    f32x4 acc[PTW][KTW];
    // the residual gradient of the block in flight (bf16 pairs): registers -- except at 16 channels, where those 32 registers are the difference between a kernel that
    // fits its 256 + 256 and one that shuffles ~1 300 accumulator-register copies and 127 scratch accesses per pair of units: there it is parked in LDS (GR, same
    // lane -> pixel mapping on the way in and out: 512 contiguous bytes per wave and store)
    constexpr bool GRES_LDS = C == 16 && !ENTRY && !GRP;
    unsigned gres[GRES_LDS ? 1 : PTW][KTW][2];
    char* GR = stg + 4 * 2304 * 4;
    unsigned zq[PTW][KTW][2], yq[PTW][KTW][2];
    float4 mu4[KTW], is4[KTW], ga4[KTW], be4[KTW];
    float c_gamma = 0.f, c_invstd = 0.f, c_dg = 0.f, c_db = 0.f;    // thread c < C: its channel's parameters; workgroup 0: the old dgamma / dbeta
    float4 md4[KTW], id4[KTW];                                      // (ENTRY) the shortcut BatchNorm's saved statistics at this lane's channels
    float d_gamma = 0.f, d_invstd = 0.f, d_dg = 0.f, d_db = 0.f;
    // (ENTRY) behind unit 0: z of the block's 3x3 / stride-2 convolution into zq, z of its shortcut into yq, the statistics of both
    auto prefetch_entry = [&](int l15) __attribute__((always_inline)) {
        if constexpr (ENTRY) {
#pragma unroll
            for (int kt = 0; kt < KTW; ++kt) {
                const int ch = (wk * KTW + kt) * 16 + 4 * g;
                mu4[kt] = *reinterpret_cast<const float4*>(p.ea.mean + ch);
                is4[kt] = *reinterpret_cast<const float4*>(p.ea.invstd + ch);
                ga4[kt] = *reinterpret_cast<const float4*>(p.ea.gamma + ch);
                be4[kt] = *reinterpret_cast<const float4*>(p.ea.beta + ch);
                md4[kt] = *reinterpret_cast<const float4*>(p.ed.mean + ch);
                id4[kt] = *reinterpret_cast<const float4*>(p.ed.invstd + ch);
#pragma unroll
                for (int t = 0; t < PTW; ++t) {
                    const size_t at = (ibase + (wp * PTW + t) * 16 + l15) * C + ch;
                    const uint2 za = *reinterpret_cast<const uint2*>(p.ea.z + at), zd = *reinterpret_cast<const uint2*>(p.ed.z + at);
                    zq[t][kt][0] = za.x; zq[t][kt][1] = za.y; yq[t][kt][0] = zd.x; yq[t][kt][1] = zd.y;
                }
            }
            if (tid < C) {
                c_gamma = p.ea.gamma[tid]; c_invstd = p.ea.invstd[tid]; d_gamma = p.ed.gamma[tid]; d_invstd = p.ed.invstd[tid];
                if (img == 0) { c_dg = p.ea.dgamma[tid]; c_db = p.ea.dbeta[tid]; d_dg = p.ed.dgamma[tid]; d_db = p.ed.dbeta[tid]; }
            }
        }
    };

    auto prefetch = [&](int cv, int l15) __attribute__((always_inline)) {
        const StConvB& cc = p.c[cv];
#pragma unroll
        for (int kt = 0; kt < KTW; ++kt) {
            const int ch = (wk * KTW + kt) * 16 + 4 * g;
            mu4[kt] = *reinterpret_cast<const float4*>(cc.mean + ch);
            is4[kt] = *reinterpret_cast<const float4*>(cc.invstd + ch);
            ga4[kt] = *reinterpret_cast<const float4*>(cc.gamma + ch);
            be4[kt] = *reinterpret_cast<const float4*>(cc.beta + ch);
#pragma unroll
            for (int t = 0; t < PTW; ++t) {
                const size_t at = (ibase + (wp * PTW + t) * 16 + l15) * C + ch;
                const uint2 zz = *reinterpret_cast<const uint2*>(cc.z + at);
                zq[t][kt][0] = zz.x; zq[t][kt][1] = zz.y;
                if (is_second(cv)) { const uint2 yy = *reinterpret_cast<const uint2*>(cc.y + at); yq[t][kt][0] = yy.x; yq[t][kt][1] = yy.y; }
            }
        }
        if (tid < C) {
            c_gamma = cc.gamma[tid]; c_invstd = cc.invstd[tid];
            if (img == 0) { c_dg = cc.dgamma[tid]; c_db = cc.dbeta[tid]; }
        }
    };

    auto unit = [&](int cv) __attribute__((always_inline)) {
        const bool second = is_second(cv);
        const StConvB& cc = p.c[cv];
        const bool tr = p.trace > 0 && cv == p.trace && img == 0;
        st_stamp(p.xb, tr, 8);
        st_stamp(p.xb, tr, 21);
        if (GRP && cv + 2 < p.nconv) grp_issue(cv + 2, img);        // (the weight gradient of the unit two above: its operands are complete in memory since the last exchange)
        st_stamp(p.xb, tr, 22);
        // (an opaque copy of the lane index: everything derived from it -- LDS addresses, global offsets -- is recomputed per unit instead of being hoisted out of
        //  the unit loop into ~100 loop-invariant registers)
        int l15 = lane & 15;
        asm volatile("" : "+v"(l15));
        int pbase[PTW];
#pragma unroll
        for (int t = 0; t < PTW; ++t) {
            const int q = (wp * PTW + t) * 16 + l15;
            const int yy = q / HW, xx = q - yy * HW;
            pbase[t] = (yy * P + xx) * PB;
        }
        // ---- A: g = dy * mask, per-image sums of g and g xhat
        unsigned gq[PTW][KTW][2];                                   // the masked gradient g of this unit at this lane's positions (bf16 pairs: g IS a bf16 value)
        float sv[KTW * 8];
#pragma unroll
        for (int q = 0; q < KTW * 8; ++q) sv[q] = 0.f;
#pragma unroll
        for (int kt = 0; kt < KTW; ++kt) {
            const float mu[4] = {mu4[kt].x, mu4[kt].y, mu4[kt].z, mu4[kt].w}, is[4] = {is4[kt].x, is4[kt].y, is4[kt].z, is4[kt].w};
            float sc[4], sh[4];
            {
                const float ga[4] = {ga4[kt].x, ga4[kt].y, ga4[kt].z, ga4[kt].w}, be[4] = {be4[kt].x, be4[kt].y, be4[kt].z, be4[kt].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { sc[e] = ga[e] * is[e]; sh[e] = be[e] - mu[e] * sc[e]; }      // (the forward's scale / shift expressions)
            }
#pragma unroll
            for (int t = 0; t < PTW; ++t) {
                const unsigned z0 = zq[t][kt][0], z1 = zq[t][kt][1];
                const float zf[4] = {__uint_as_float(z0 << 16), __uint_as_float(z0 & 0xffff0000u), __uint_as_float(z1 << 16), __uint_as_float(z1 & 0xffff0000u)};
                bool on[4];
                if (second) {
                    const unsigned y0 = yq[t][kt][0], y1 = yq[t][kt][1];
                    on[0] = (y0 & 0x7fffu) != 0u && (y0 & 0x8000u) == 0u; on[1] = (y0 & 0x7fff0000u) != 0u && (y0 & 0x80000000u) == 0u;
                    on[2] = (y1 & 0x7fffu) != 0u && (y1 & 0x8000u) == 0u; on[3] = (y1 & 0x7fff0000u) != 0u && (y1 & 0x80000000u) == 0u;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) on[e] = fmaf(zf[e], sc[e], sh[e]) > 0.f;
                }
                // dy passes through bf16 (the per-unit path stores it), then the mask
                const unsigned d0 = pack_bf16x2(acc[t][kt][0], acc[t][kt][1]), d1 = pack_bf16x2(acc[t][kt][2], acc[t][kt][3]);
                float gg[4] = {__uint_as_float(d0 << 16), __uint_as_float(d0 & 0xffff0000u), __uint_as_float(d1 << 16), __uint_as_float(d1 & 0xffff0000u)};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    gg[e] = on[e] ? gg[e] : 0.f;
                    const float xh = (zf[e] - mu[e]) * is[e];
                    sv[kt * 8 + e] += gg[e];
                    sv[kt * 8 + 4 + e] = fmaf(gg[e], xh, sv[kt * 8 + 4 + e]);
                }
                gq[t][kt][0] = pack_bf16x2(gg[0], gg[1]); gq[t][kt][1] = pack_bf16x2(gg[2], gg[3]);      // (exact)
                if (second) {
                    if constexpr (GRES_LDS) *reinterpret_cast<uint2*>(GR + (((wp * PTW + t) * 16 + l15) * C + (wk * KTW + kt) * 16 + 4 * g) * 2) = make_uint2(gq[t][kt][0], gq[t][kt][1]);
                    else { gres[t][kt][0] = gq[t][kt][0]; gres[t][kt][1] = gq[t][kt][1]; }
                }
            }
        }
        st_stamp(p.xb, tr, 9);
        row16_sum_n(sv);
        if (l15 == 0) {
#pragma unroll
            for (int kt = 0; kt < KTW; ++kt) {
                const int ch = (wk * KTW + kt) * 16 + 4 * g;
                *reinterpret_cast<float4*>(red + (wave * 2 + 0) * C + ch) = make_float4(sv[kt * 8], sv[kt * 8 + 1], sv[kt * 8 + 2], sv[kt * 8 + 3]);
                *reinterpret_cast<float4*>(red + (wave * 2 + 1) * C + ch) = make_float4(sv[kt * 8 + 4], sv[kt * 8 + 5], sv[kt * 8 + 6], sv[kt * 8 + 7]);
            }
        }
        __syncthreads();
        if (tid < 2 * C) {
            const int stat = tid / C, ch = tid - stat * C;
            float v;
            if (WK == 1) v = ((red[(0 * 2 + stat) * C + ch] + red[(1 * 2 + stat) * C + ch]) + red[(2 * 2 + stat) * C + ch]) + red[(3 * 2 + stat) * C + ch];
            else v = red[((ch >> 4) * 2 + stat) * C + ch];
            vals[tid] = v;
        }
        // (group weight gradient: what this workgroup publishes next tells the others that its dz of the unit above is in memory -- every wave drains its stores first)
        if constexpr (GRP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        st_stamp(p.xb, tr, 10);
        // ---- B: the batch's sums are on their way ...
        const unsigned tag = base + (unsigned)(p.nconv - 1 - cv) + 1u;
        xch_begin<4>(p.xb, img, p.N, 2 * C, tag, vals, scratch);
        st_stamp(p.xb, tr, 11);
        // ---- C: ... while this unit's input image is requested (the block input -- a written activation -- for the first convolution of a block; z of the first
        //      convolution for the second one, turned into relu(scale z + shift) with the forward's expressions on its way into LDS) and the weight gradient of
        //      the unit above runs (its dz in D, its input image in XA)
        if constexpr (!GRP) {
            uint4 xin[NCH];
            float xsc[8], xsh[8];
            {
                // (the lazy input's coefficients are requested -- and consumed -- BEFORE the image: vector loads retire in order, so waiting for coefficients that were
                //  issued behind the image meant waiting for the image, ~2 us in front of the weight gradient that was to hide it)
                if (second) {
                    const StConvB& a = prevc(cv);
                    const int c0 = (tid & (CPP - 1)) * 8;
                    const float4 g0 = *reinterpret_cast<const float4*>(a.gamma + c0), g1 = *reinterpret_cast<const float4*>(a.gamma + c0 + 4);
                    const float4 i0 = *reinterpret_cast<const float4*>(a.invstd + c0), i1 = *reinterpret_cast<const float4*>(a.invstd + c0 + 4);
                    const float4 b0 = *reinterpret_cast<const float4*>(a.beta + c0), b1 = *reinterpret_cast<const float4*>(a.beta + c0 + 4);
                    const float4 m0 = *reinterpret_cast<const float4*>(a.mean + c0), m1 = *reinterpret_cast<const float4*>(a.mean + c0 + 4);
                    const float ga[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, is[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
                    const float be[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w}, mu[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) { xsc[e] = ga[e] * is[e]; xsh[e] = be[e] - mu[e] * xsc[e]; }
                }
                __builtin_amdgcn_sched_barrier(0);
                const bf16_t* src = second ? prevc(cv).z : (cv == 0 ? p.x : p.c[cv - 1].y);
                const uint4* s4 = reinterpret_cast<const uint4*>(src + ibase * C);
#pragma unroll
                for (int k = 0; k < NCH; ++k) xin[k] = s4[tid + k * 256];
            }
            if (cv + 1 < p.nconv) {
                if constexpr (C == 16) st_wgrad16<HW>(D, XA, wred, p.c[cv + 1].slab + (size_t)img * (9 * C * C));
                else st_wgrad_wide<C, HW>(D, XA, p.c[cv + 1].slab + (size_t)img * (9 * C * C));
                __syncthreads();
            }
            st_stamp(p.xb, tr, 12);
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int i = tid + k * 256;
                const int q = i / CPP, c8 = i - q * CPP;
                const int yy = q / HW, xx = q - yy * HW;
                *reinterpret_cast<uint4*>(XA + ((yy + 1) * P + xx + 1) * PB + c8 * 16) = second ? bn_relu8_bf16(xin[k], xsc, xsh) : xin[k];
            }
        } else {
            // the group weight gradient of the unit TWO above: every workgroup's dz of that unit was in memory before it published the exchange that has since completed
            if (cv + 2 < p.nconv) grp_finish(cv + 2, tr);
            st_stamp(p.xb, tr, 12);
        }
        bf16x8_t wf[KTW][KS];
#pragma unroll
        for (int kt = 0; kt < KTW; ++kt) {                          // (the dgrad filters: requested before the wait for the totals)
            const bf16_t* wr = cc.wd + (size_t)((wk * KTW + kt) * 16 + l15) * 9 * C;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int kk = 32 * s + 8 * g;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (kk < 9 * C) v = *reinterpret_cast<const uint4*>(wr + kk);
                wf[kt][s] = __builtin_bit_cast(bf16x8_t, v);
            }
        }
        st_stamp(p.xb, tr, 13);
        // ---- D: dz of this unit into D
        [[maybe_unused]] const __amdgpu_buffer_rsrc_t dzrs_own = __builtin_amdgcn_make_buffer_rsrc(GRP ? cc.dzg : const_cast<bf16_t*>(cc.z), 0, p.N * HW * HW * C * 2, 0x00020000);
        xch_end(p.xb, p.N, 2 * C, tag, tot, scratch);
        st_stamp(p.xb, tr, 14);
        if (tid < C) {
            const double s1 = tot[tid], s2 = tot[C + tid];
            ctab[tid] = (float)(s1 * p.invM);
            ctab[C + tid] = (float)(s2 * p.invM);
            ctab[4 * C + tid] = c_gamma * c_invstd;
            if (img == 0) { cc.dbeta[tid] = c_db + (float)s1; cc.dgamma[tid] = c_dg + (float)s2; }
        }
        __syncthreads();
#pragma unroll
        for (int kt = 0; kt < KTW; ++kt) {
            const int ch = (wk * KTW + kt) * 16 + 4 * g;
            const float4 k04 = *reinterpret_cast<const float4*>(ctab + ch), k14 = *reinterpret_cast<const float4*>(ctab + C + ch);
            const float4 gi4 = *reinterpret_cast<const float4*>(ctab + 4 * C + ch);
            const float k0[4] = {k04.x, k04.y, k04.z, k04.w}, k1[4] = {k14.x, k14.y, k14.z, k14.w}, gi[4] = {gi4.x, gi4.y, gi4.z, gi4.w};
            const float mu[4] = {mu4[kt].x, mu4[kt].y, mu4[kt].z, mu4[kt].w}, is[4] = {is4[kt].x, is4[kt].y, is4[kt].z, is4[kt].w};
#pragma unroll
            for (int t = 0; t < PTW; ++t) {
                const float zf[4] = {__uint_as_float(zq[t][kt][0] << 16), __uint_as_float(zq[t][kt][0] & 0xffff0000u), __uint_as_float(zq[t][kt][1] << 16),
                                     __uint_as_float(zq[t][kt][1] & 0xffff0000u)};
                const float gg[4] = {__uint_as_float(gq[t][kt][0] << 16), __uint_as_float(gq[t][kt][0] & 0xffff0000u), __uint_as_float(gq[t][kt][1] << 16),
                                     __uint_as_float(gq[t][kt][1] & 0xffff0000u)};
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = (zf[e] - mu[e]) * is[e];
                    o[e] = gi[e] * (gg[e] - k0[e] - xh * k1[e]);          // (bn_bwd_apply_acc_kernel's expression, bn.hip / lazy_dz8, conv3.hip)
                }
                const uint2 dzv = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
                *reinterpret_cast<uint2*>(D + pbase[t] + (P + 1) * PB + ch * 2) = dzv;
                if constexpr (GRP) {                                 // write-through (sc1): other workgroups read it inside this launch
                    typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
                    __builtin_amdgcn_raw_buffer_store_b64((u32x2_t){dzv.x, dzv.y}, dzrs_own,
                                                          (int)((((size_t)(ch >> 4) * p.N + img) * HW * HW + (wp * PTW + t) * 16 + l15) * 32 + (ch & 15) * 2), 0, 16);
                }
            }
        }
        __syncthreads();
        st_stamp(p.xb, tr, 15);
        // ---- E: the gradient of this unit's input; the operands of the unit below are requested first
        if (cv > 0) prefetch(cv - 1, l15);
        else prefetch_entry(l15);
        int koff[KS];                                               // mirrored taps: filter tap (r, s) reads the gradient at (y + 1 - r, x + 1 - s) = padded (y + 2 - r, x + 2 - s)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int kk = 32 * s + 8 * g;
            const int tap = kk >> LOGC, c0 = kk & (C - 1);
            const int dy = (tap * 11) >> 5, dx = tap - 3 * dy;
            koff[s] = kk < 9 * C ? ((2 - dy) * P + (2 - dx)) * PB + c0 * 2 : -1;
        }
#pragma unroll
        for (int t = 0; t < PTW; ++t)
#pragma unroll
            for (int kt = 0; kt < KTW; ++kt) acc[t][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        constexpr int SLOTS = KS * PTW;
        auto xread = [&](int n) {
            const int s_ = n / PTW, t_ = n - s_ * PTW;
            const int a = koff[s_] >= 0 ? pbase[t_] + koff[s_] : 0;
            return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(D + a));
        };
        // (pixel fragments RING - 1 slots ahead of their MFMAs: with one wave per SIMD nothing else hides the LDS latency)
        constexpr int RING = 8;
        bf16x8_t xr[RING];
#pragma unroll
        for (int n = 0; n < RING - 1; ++n) if (n < SLOTS) xr[n] = xread(n);
#pragma unroll
        for (int n = 0; n < SLOTS; ++n) {
            const int s_ = n / PTW, t_ = n - s_ * PTW;
            if (n + RING - 1 < SLOTS) xr[(n + RING - 1) % RING] = xread(n + RING - 1);
#pragma unroll
            for (int kt = 0; kt < KTW; ++kt) acc[t_][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kt][s_], xr[n % RING], acc[t_][kt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        st_stamp(p.xb, tr, 16);
        if (!second) {
            // the top of a block: + the gradient that went around it (what the block's last BatchNorm backward left: g of its output)
#pragma unroll
            for (int t = 0; t < PTW; ++t)
#pragma unroll
                for (int kt = 0; kt < KTW; ++kt) {
                    // (the per-unit path rounds the dgrad result to bf16 before the residual gradient is added to it)
                    const unsigned d0 = pack_bf16x2(acc[t][kt][0], acc[t][kt][1]), d1 = pack_bf16x2(acc[t][kt][2], acc[t][kt][3]);
                    unsigned r0, r1;
                    if constexpr (GRES_LDS) {
                        const uint2 rr = *reinterpret_cast<const uint2*>(GR + (((wp * PTW + t) * 16 + l15) * C + (wk * KTW + kt) * 16 + 4 * g) * 2);
                        r0 = rr.x; r1 = rr.y;
                    } else { r0 = gres[t][kt][0]; r1 = gres[t][kt][1]; }
                    acc[t][kt][0] = __uint_as_float(d0 << 16) + __uint_as_float(r0 << 16);
                    acc[t][kt][1] = __uint_as_float(d0 & 0xffff0000u) + __uint_as_float(r0 & 0xffff0000u);
                    acc[t][kt][2] = __uint_as_float(d1 << 16) + __uint_as_float(r1 << 16);
                    acc[t][kt][3] = __uint_as_float(d1 & 0xffff0000u) + __uint_as_float(r1 & 0xffff0000u);
                }
        }
    };

    // ---- the gradient of the run's output, the operands of its last unit
    {
        const int l15 = lane & 15;
        prefetch(p.nconv - 1, l15);
#pragma unroll
        for (int kt = 0; kt < KTW; ++kt) {
            const int ch = (wk * KTW + kt) * 16 + 4 * g;
#pragma unroll
            for (int t = 0; t < PTW; ++t) {
                uint2 v;
                if (p.dfeat != nullptr) {
                    // the global pooling behind the run: every pixel's gradient is dfeat / (H W), rounded to bf16 like the tensor avgpool_bwd_kernel (bn.hip) stores
                    const float4 d = *reinterpret_cast<const float4*>(p.dfeat + (size_t)img * C + ch);
                    constexpr float inv = 1.f / (float)(HW * HW);
                    v = make_uint2(pack_bf16x2(d.x * inv, d.y * inv), pack_bf16x2(d.z * inv, d.w * inv));
                } else v = *reinterpret_cast<const uint2*>(p.dy + (ibase + (wp * PTW + t) * 16 + l15) * C + ch);
                acc[t][kt] = (f32x4){__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u)};
                if constexpr (!GRES_LDS) gres[t][kt][0] = gres[t][kt][1] = 0u;
            }
        }
    }
#pragma unroll 1
    for (int cv = p.nconv - 1; cv > 0; cv -= 2) {                  // (pairs from the top; ENTRY: nconv is odd and unit 0 -- the entry block's second convolution -- is left)
        unit(cv);
        unit(cv - 1);
    }
    if constexpr (ENTRY) {
        unit(0);
        // ================= the entry block's two strided convolutions (acc = gradient of the 3x3 / stride-2 convolution's activation, gres = gradient of the
        // shortcut's BatchNorm output -- what went around the block -- zq / yq = their z)
        using E = StEnt<C, HW>;
        constexpr int CI = E::CI, HWI = E::HWI, PI = E::PI, PBI = E::PBI;
        char* INB = smem + GB::LDS;
        char* Dds = INB + E::INB;
        float* vals2 = ctab + 5 * C;                                // the shortcut's sums (ctab uses [0, 5 C) of its 8 C floats)
        int l15 = lane & 15;
        asm volatile("" : "+v"(l15));
        int pbase[PTW];
#pragma unroll
        for (int t = 0; t < PTW; ++t) {
            const int q = (wp * PTW + t) * 16 + l15;
            const int yy = q / HW, xx = q - yy * HW;
            pbase[t] = (yy * P + xx) * PB;
        }
        unsigned gqa[PTW][KTW][2];
        float sva[KTW * 8], svd[KTW * 8];
#pragma unroll
        for (int q = 0; q < KTW * 8; ++q) { sva[q] = 0.f; svd[q] = 0.f; }
#pragma unroll
        for (int kt = 0; kt < KTW; ++kt) {
            const float mu[4] = {mu4[kt].x, mu4[kt].y, mu4[kt].z, mu4[kt].w}, is[4] = {is4[kt].x, is4[kt].y, is4[kt].z, is4[kt].w};
            const float mud[4] = {md4[kt].x, md4[kt].y, md4[kt].z, md4[kt].w}, isd[4] = {id4[kt].x, id4[kt].y, id4[kt].z, id4[kt].w};
            float sc[4], sh[4];
            {
                const float ga[4] = {ga4[kt].x, ga4[kt].y, ga4[kt].z, ga4[kt].w}, be[4] = {be4[kt].x, be4[kt].y, be4[kt].z, be4[kt].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { sc[e] = ga[e] * is[e]; sh[e] = be[e] - mu[e] * sc[e]; }
            }
#pragma unroll
            for (int t = 0; t < PTW; ++t) {
                const unsigned z0 = zq[t][kt][0], z1 = zq[t][kt][1], y0 = yq[t][kt][0], y1 = yq[t][kt][1];
                const float zf[4] = {__uint_as_float(z0 << 16), __uint_as_float(z0 & 0xffff0000u), __uint_as_float(z1 << 16), __uint_as_float(z1 & 0xffff0000u)};
                const float zd[4] = {__uint_as_float(y0 << 16), __uint_as_float(y0 & 0xffff0000u), __uint_as_float(y1 << 16), __uint_as_float(y1 & 0xffff0000u)};
                const unsigned d0 = pack_bf16x2(acc[t][kt][0], acc[t][kt][1]), d1 = pack_bf16x2(acc[t][kt][2], acc[t][kt][3]);
                float gg[4] = {__uint_as_float(d0 << 16), __uint_as_float(d0 & 0xffff0000u), __uint_as_float(d1 << 16), __uint_as_float(d1 & 0xffff0000u)};
                const float gd[4] = {__uint_as_float(gres[t][kt][0] << 16), __uint_as_float(gres[t][kt][0] & 0xffff0000u), __uint_as_float(gres[t][kt][1] << 16),
                                     __uint_as_float(gres[t][kt][1] & 0xffff0000u)};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    gg[e] = fmaf(zf[e], sc[e], sh[e]) > 0.f ? gg[e] : 0.f;
                    sva[kt * 8 + e] += gg[e];
                    sva[kt * 8 + 4 + e] = fmaf(gg[e], (zf[e] - mu[e]) * is[e], sva[kt * 8 + 4 + e]);
                    svd[kt * 8 + e] += gd[e];                      // (no ReLU behind the shortcut's BatchNorm: its gradient is what the block's last BatchNorm backward left)
                    svd[kt * 8 + 4 + e] = fmaf(gd[e], (zd[e] - mud[e]) * isd[e], svd[kt * 8 + 4 + e]);
                }
                gqa[t][kt][0] = pack_bf16x2(gg[0], gg[1]); gqa[t][kt][1] = pack_bf16x2(gg[2], gg[3]);
            }
        }
        auto reduce_to = [&](float (&sv)[KTW * 8], float* out) {
            row16_sum_n(sv);
            if (l15 == 0) {
#pragma unroll
                for (int kt = 0; kt < KTW; ++kt) {
                    const int ch = (wk * KTW + kt) * 16 + 4 * g;
                    *reinterpret_cast<float4*>(red + (wave * 2 + 0) * C + ch) = make_float4(sv[kt * 8], sv[kt * 8 + 1], sv[kt * 8 + 2], sv[kt * 8 + 3]);
                    *reinterpret_cast<float4*>(red + (wave * 2 + 1) * C + ch) = make_float4(sv[kt * 8 + 4], sv[kt * 8 + 5], sv[kt * 8 + 6], sv[kt * 8 + 7]);
                }
            }
            __syncthreads();
            if (tid < 2 * C) {
                const int stat = tid / C, ch = tid - stat * C;
                float v;
                if (WK == 1) v = ((red[(0 * 2 + stat) * C + ch] + red[(1 * 2 + stat) * C + ch]) + red[(2 * 2 + stat) * C + ch]) + red[(3 * 2 + stat) * C + ch];
                else v = red[((ch >> 4) * 2 + stat) * C + ch];
                out[tid] = v;
            }
            __syncthreads();
        };
        reduce_to(sva, vals);
        reduce_to(svd, vals2);
        if constexpr (GRP) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
        const unsigned tagA = base + (unsigned)p.nconv + 1u, tagD = tagA + 1u;
        // ---- exchange A (the 3x3 / stride-2 convolution's sums), the pending weight gradient of unit 0 (own-image) / unit 1 (group) in its shadow
        if (GRP && p.nconv > 1) grp_issue(1, img);
        xch_begin<4>(p.xb, img, p.N, 2 * C, tagA, vals, scratch);
        if constexpr (!GRP) { st_wgrad_wide<C, HW>(D, XA, p.c[0].slab + (size_t)img * (9 * C * C)); __syncthreads(); }
        else if (p.nconv > 1) grp_finish(1, false);
        xch_end(p.xb, p.N, 2 * C, tagA, tot, scratch);
        if (tid < C) {
            const double s1 = tot[tid], s2 = tot[C + tid];
            ctab[tid] = (float)(s1 * p.invM);
            ctab[C + tid] = (float)(s2 * p.invM);
            ctab[4 * C + tid] = c_gamma * c_invstd;
            if (img == 0) { p.ea.dbeta[tid] = c_db + (float)s1; p.ea.dgamma[tid] = c_dg + (float)s2; }
        }
        __syncthreads();
        auto write_dz = [&](char* dst, const unsigned (&gq)[PTW][KTW][2], const unsigned (&zz)[PTW][KTW][2], const float4 (&m4)[KTW], const float4 (&i4)[KTW]) {
#pragma unroll
            for (int kt = 0; kt < KTW; ++kt) {
                const int ch = (wk * KTW + kt) * 16 + 4 * g;
                const float4 k04 = *reinterpret_cast<const float4*>(ctab + ch), k14 = *reinterpret_cast<const float4*>(ctab + C + ch);
                const float4 gi4 = *reinterpret_cast<const float4*>(ctab + 4 * C + ch);
                const float k0[4] = {k04.x, k04.y, k04.z, k04.w}, k1[4] = {k14.x, k14.y, k14.z, k14.w}, gi[4] = {gi4.x, gi4.y, gi4.z, gi4.w};
                const float mu[4] = {m4[kt].x, m4[kt].y, m4[kt].z, m4[kt].w}, is[4] = {i4[kt].x, i4[kt].y, i4[kt].z, i4[kt].w};
#pragma unroll
                for (int t = 0; t < PTW; ++t) {
                    const float zf[4] = {__uint_as_float(zz[t][kt][0] << 16), __uint_as_float(zz[t][kt][0] & 0xffff0000u), __uint_as_float(zz[t][kt][1] << 16),
                                         __uint_as_float(zz[t][kt][1] & 0xffff0000u)};
                    const float gg[4] = {__uint_as_float(gq[t][kt][0] << 16), __uint_as_float(gq[t][kt][0] & 0xffff0000u), __uint_as_float(gq[t][kt][1] << 16),
                                         __uint_as_float(gq[t][kt][1] & 0xffff0000u)};
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = gi[e] * (gg[e] - k0[e] - (zf[e] - mu[e]) * is[e] * k1[e]);
                    *reinterpret_cast<uint2*>(dst + pbase[t] + (P + 1) * PB + ch * 2) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
                }
            }
        };
        write_dz(D, gqa, zq, mu4, is4);
        // ---- exchange D (the shortcut's sums); the block's input image comes into LDS -- and the group weight gradient of unit 0 runs -- in its shadow
        if constexpr (GRP) grp_issue(0, img);                       // (every workgroup's dz of unit 0 was in memory before it published exchange A)
        xch_begin<4>(p.xb, img, p.N, 2 * C, tagD, vals2, scratch);
        {
            constexpr int CPPI = CI / 8;
            const uint4* src = reinterpret_cast<const uint4*>(p.x + (size_t)img * HWI * HWI * CI);
            for (int i = tid; i < HWI * HWI * CPPI; i += 256) {
                const int q = i / CPPI, c8 = i - q * CPPI;
                const int yy = q / HWI, xx = q - yy * HWI;
                *reinterpret_cast<uint4*>(INB + ((yy + 1) * PI + xx + 1) * PBI + c8 * 16) = src[i];
            }
        }
        if constexpr (GRP) grp_finish(0, false);
        xch_end(p.xb, p.N, 2 * C, tagD, tot, scratch);
        if (tid < C) {
            const double s1 = tot[tid], s2 = tot[C + tid];
            ctab[tid] = (float)(s1 * p.invM);
            ctab[C + tid] = (float)(s2 * p.invM);
            ctab[4 * C + tid] = d_gamma * d_invstd;
            if (img == 0) { p.ed.dbeta[tid] = d_db + (float)s1; p.ed.dgamma[tid] = d_dg + (float)s2; }
        }
        __syncthreads();
        write_dz(Dds, gres, yq, md4, id4);
        __syncthreads();
        st_dgrad_entry<C, HW>(D, Dds, p.ea.wd, p.ed.wd, p.dx + (size_t)img * HWI * HWI * CI, p.dx_acc);
        st_wgrad_entry<C, HW>(D, Dds, INB, p.ea.slab + (size_t)img * (9 * C * CI), p.ed.slab + (size_t)img * (C * CI));
        if (img == 0 && tid == 0) xch_advance(p.xb, base, (unsigned)p.nconv + 2u);
    } else {
        // ---- the gradient of the run's input
        {
            const int l15 = lane & 15;
#pragma unroll
            for (int kt = 0; kt < KTW; ++kt) {
                const int ch = (wk * KTW + kt) * 16 + 4 * g;
#pragma unroll
                for (int t = 0; t < PTW; ++t) {
                    bf16_t* o = p.dx + (ibase + (wp * PTW + t) * 16 + l15) * C + ch;
                    float v0 = acc[t][kt][0], v1 = acc[t][kt][1], v2 = acc[t][kt][2], v3 = acc[t][kt][3];
                    if (p.dx_acc) {
                        const uint2 old = *reinterpret_cast<const uint2*>(o);
                        v0 += __uint_as_float(old.x << 16); v1 += __uint_as_float(old.x & 0xffff0000u);
                        v2 += __uint_as_float(old.y << 16); v3 += __uint_as_float(old.y & 0xffff0000u);
                    }
                    *reinterpret_cast<uint2*>(o) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
                }
            }
        }
        // ---- the weight gradients still owed
        if constexpr (!GRP) {
            if constexpr (C == 16) st_wgrad16<HW>(D, XA, wred, p.c[0].slab + (size_t)img * (9 * C * C));
            else st_wgrad_wide<C, HW>(D, XA, p.c[0].slab + (size_t)img * (9 * C * C));
            if (img == 0 && tid == 0) xch_advance(p.xb, base, (unsigned)p.nconv);
        } else {
            // unit 1's dz is in memory everywhere (the exchange of unit 0 has completed); unit 0's needs one more grid-wide hand-shake, taken around unit 1's work
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const unsigned tag = base + (unsigned)p.nconv + 1u;
            if (p.nconv > 1) grp_issue(1, img);
            xch_begin<4>(p.xb, img, p.N, 2 * C, tag, vals, scratch);
            if (p.nconv > 1) grp_finish(1, false);
            xch_end(p.xb, p.N, 2 * C, tag, tot, scratch);
            grp_issue(0, img);
            grp_finish(0, false);
            if (img == 0 && tid == 0) xch_advance(p.xb, base, (unsigned)p.nconv + 1u);
        }
    }
}

template <int C, int HW, bool GRP, bool ENTRY>
int launch_bwd(const StBwdParams& p, hipStream_t st) {
    constexpr int lds = StGeoB<C, HW, GRP>::LDS + (ENTRY ? StEnt<C, HW>::BYTES : 0);
    int dev = 0;
    (void)hipGetDevice(&dev);
    static bool attr[16] = {};
    if (dev < 0 || dev >= 16 || !attr[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(stage_train_bwd_kernel<C, HW, GRP, ENTRY>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            clhip_set_error("stage_train_bwd: cannot reserve %d bytes of LDS", lds);
            return CLHIP_EHIP;
        }
        if (dev >= 0 && dev < 16) attr[dev] = true;
    }
    hipLaunchKernelGGL((stage_train_bwd_kernel<C, HW, GRP, ENTRY>), dim3(p.N), dim3(256), lds, st, p);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

}  // namespace

// the largest batch a stage-level training launch takes: every workgroup (= image) must be resident at once
int clhip_stage_train_max_batch() {
    static int cus = -1;
    if (cus < 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        else cus = 0;
        if (cus > 256) cus = 256;                                     // (xch.h: at most 256 contributors)
    }
    return cus;
}

bool clhip_stage_train_supported(int N, int H, int W, int C, int nconv, int dtype) {
    if (dtype != CLHIP_BF16 || H != W || nconv < 2 || nconv > kMaxConvT || (nconv & 1) || N < 1 || N > clhip_stage_train_max_batch()) return false;
    return (C == 16 && H == 32) || (C == 32 && H == 16) || (C == 64 && H == 8);
}

size_t clhip_stage_train_xch_bytes(int N) { return xch_bytes(N, 128); }

// The three-level exchange of xch.h is correct only where every workgroup w of a grid runs on XCD w % 8 -- more precisely, where workgroups with equal w % 8 share an
// L2.  Checked ONCE per device (a probe launch per grid size + a read-back: plan.hip calls this when it creates a plan, never inside a stream capture); launches
// consult the cached answer and keep the two-hop form where the rule does not hold, was never checked, or STAGE_XCH3=0.
namespace {
__global__ void xcd_probe_kernel(unsigned* out) {
    if (threadIdx.x == 0) {
        unsigned v;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
        out[blockIdx.x] = v & 0xfu;
    }
}
int g_xcd_rule[16] = {};          // per device: 0 unknown, 1 holds, -1 does not
}  // namespace
bool clhip_stage_train_xcd_rule(bool probe) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return false;
    if (g_xcd_rule[dev] == 0 && probe) {
        int rule = -1;
        unsigned* d = nullptr;
        if (hipMalloc(&d, 256 * sizeof(unsigned)) == hipSuccess) {
            bool ok = true;
            for (int G : {256, 200, 128, 100, 72, 64}) {
                unsigned h[256];
                hipLaunchKernelGGL(xcd_probe_kernel, dim3(G), dim3(64), 0, nullptr, d);
                if (hipMemcpy(h, d, G * sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) { ok = false; break; }
                for (int i = kXchXcds; i < G && ok; ++i) ok = h[i] == h[i & (kXchXcds - 1)];
                if (!ok) break;
            }
            (void)hipFree(d);
            if (ok) rule = 1;
        }
        (void)hipGetLastError();
        g_xcd_rule[dev] = rule;
    }
    const char* sw = clhip_cfg("STAGE_XCH3");
    return g_xcd_rule[dev] == 1 && !(sw != nullptr && atoi(sw) == 0);
}
// The weight gradient of a run's backward: per image out of the workgroup's own LDS (partial blocks of 9 / 36 / 147 KB per image and convolution), or -- 64 channels at
// 128 images or more, where those blocks would be 300 MB per run -- per (channel-tile pair, group of 16 images) from global memory (StGrp).  Measured at batch 256 /
// 32 (profiles/r06_stage_train_notes.md): the gather of 16-channel slices out of 64- / 128-byte pixels costs 2.4 / 5.8 us of load issue per unit at 64 / 32 channels,
// more than the larger slabs cost the 32-channel stage or any stage at a small batch.
static bool stage_train_group(int N, int C) { return C == 64 && N >= 128; }
// partial blocks (of C * 9 * C floats) per convolution that a backward launch leaves in its slab
int clhip_stage_train_slab_blocks(int N, int C) { const int ipg = (C / 16) * (C / 16); return stage_train_group(N, C) ? (N + ipg - 1) / ipg : N; }

// One entry per convolution of the run (C ABI of the plan: plain arrays of pointers).  x: the run's input activation [N][H][W][C] bf16.
// entry != 0: the launch starts with the stage's down-sampling block -- the arrays then hold nconv + 2 entries, [0] = that block's 3x3 / stride-2 convolution,
// [1] = its 1x1 / stride-2 shortcut convolution (w: [C][9][C / 2] and [C][1][C / 2]), [2] = its second convolution, [3 ...] = the blocks behind it (nconv odd), and x
// is the block's input [N][2 H][2 W][C / 2].
int clhip_stage_train_fwd_launch(const void* x, int N, int H, int W, int C, int nconv, const void* const* w, const float* const* gamma, const float* const* beta,
                                 float* const* rm, float* const* rv, float* const* mean, float* const* invstd, float* const* coef, void* const* z, void* const* y,
                                 void* const* mask, float momentum, float eps, void* xch, int trace, int entry, float* feat, int dtype, hipStream_t st) {
    const bool ok = entry ? (clhip_stage_train_supported(N, H, W, C, nconv + 1, dtype) && (nconv & 1) && C >= 32) : clhip_stage_train_supported(N, H, W, C, nconv, dtype);
    if (!ok || xch == nullptr) { clhip_set_error("stage_train_fwd: unsupported geometry"); return CLHIP_EINVAL; }
    StFwdParams p;
    p.x = static_cast<const bf16_t*>(x); p.N = N; p.nconv = nconv; p.eps = eps; p.momentum = momentum; p.trace = trace; p.feat = feat;
    const double M = (double)N * H * W;
    p.invM = 1.0 / M; p.unbias = M > 1.0 ? M / (M - 1.0) : 1.0;
    p.xb = xch_carve(xch, N, 128);
    if (!clhip_stage_train_xcd_rule(false)) p.xb.gx = nullptr;       // (two hops instead of three levels)
    auto fill = [&](StConv& c, int i) {
        c.w = static_cast<const bf16_t*>(w[i]); c.wd = nullptr; c.gamma = gamma[i]; c.beta = beta[i]; c.rm = rm[i]; c.rv = rv[i];
        c.mean = mean[i]; c.invstd = invstd[i]; c.coef = coef[i]; c.z = static_cast<bf16_t*>(z[i]);
        c.y = static_cast<bf16_t*>(y[i]); c.mask = static_cast<unsigned char*>(mask[i]);
    };
    const int o = entry ? 2 : 0;
    if (entry) { fill(p.ea, 0); fill(p.ed, 1); }
    for (int i = 0; i < nconv; ++i) fill(p.c[i], o + i);
    if (entry) {
        if (C == 32) return launch_fwd<32, 16, true>(p, st);
        return launch_fwd<64, 8, true>(p, st);
    }
    if (C == 16) return launch_fwd<16, 32, false>(p, st);
    if (C == 32) return launch_fwd<32, 16, false>(p, st);
    return launch_fwd<64, 8, false>(p, st);
}

// The backward of the same run.  dy: gradient of the run's output activation, dx: gradient of its input activation (accumulated into when dx_accumulate), slab[i]:
// scratch for the weight gradient's partial blocks -- the caller adds clhip_stage_train_slab_blocks() of them to the weight gradient in a fixed order
// (clhip_wgrad_reduce_launch).  entry != 0: as in the forward -- nconv + 2 array entries, [0] = the 3x3 / stride-2 convolution (wd [C / 2][9][C], slab N x C x 9 x C / 2
// floats), [1] = the shortcut (wd [C / 2][1][C], slab N x C x C / 2), x / dx = the block's input [N][2 H][2 W][C / 2] and its gradient.
int clhip_stage_train_bwd_launch(const void* x, const void* dy, void* dx, int dx_accumulate, int N, int H, int W, int C, int nconv, const void* const* wd,
                                 const float* const* gamma, const float* const* beta, const float* const* mean, const float* const* invstd, const void* const* z,
                                 const void* const* y, float* const* dgamma, float* const* dbeta, float* const* slab, void* const* dzg, void* xch, int trace, int entry, const float* dfeat, int dtype, hipStream_t st) {
    const bool ok = entry ? (clhip_stage_train_supported(N, H, W, C, nconv + 1, dtype) && (nconv & 1) && C >= 32) : clhip_stage_train_supported(N, H, W, C, nconv, dtype);
    if (!ok || xch == nullptr) { clhip_set_error("stage_train_bwd: unsupported geometry"); return CLHIP_EINVAL; }
    StBwdParams p;
    p.x = static_cast<const bf16_t*>(x); p.dy = static_cast<const bf16_t*>(dy); p.dx = static_cast<bf16_t*>(dx); p.dx_acc = dx_accumulate;
    p.N = N; p.nconv = nconv; p.invM = 1.0 / ((double)N * H * W); p.trace = trace; p.dfeat = dfeat;
    p.xb = xch_carve(xch, N, 128);
    if (!clhip_stage_train_xcd_rule(false)) p.xb.gx = nullptr;       // (two hops instead of three levels)
    auto fill = [&](StConvB& c, int i) {
        c.wd = static_cast<const bf16_t*>(wd[i]); c.gamma = gamma[i]; c.beta = beta[i]; c.mean = mean[i]; c.invstd = invstd[i];
        c.z = static_cast<const bf16_t*>(z[i]); c.y = static_cast<const bf16_t*>(y[i]); c.dgamma = dgamma[i]; c.dbeta = dbeta[i]; c.slab = slab[i];
        c.dzg = static_cast<bf16_t*>(dzg[i]);
    };
    const int o = entry ? 2 : 0;
    if (entry) { fill(p.ea, 0); fill(p.ed, 1); }
    for (int i = 0; i < nconv; ++i) fill(p.c[i], o + i);
    if (entry) {
        if (C == 32) return launch_bwd<32, 16, false, true>(p, st);
        if (stage_train_group(N, C)) return launch_bwd<64, 8, true, true>(p, st);
        return launch_bwd<64, 8, false, true>(p, st);
    }
    if (C == 16) return launch_bwd<16, 32, false, false>(p, st);
    if (C == 32) return launch_bwd<32, 16, false, false>(p, st);
    if (stage_train_group(N, C)) return launch_bwd<64, 8, true, false>(p, st);
    return launch_bwd<64, 8, false, false>(p, st);
}

// the sticky error word of an exchange buffer (non-zero: a bounded spin ran out -- the grid was not co-resident); synchronises the device
int clhip_stage_train_status(void* xch) {
    unsigned ctl[4] = {0, 0, 0, 0};
    if (xch == nullptr) return 0;
    if (hipMemcpy(ctl, xch, sizeof(ctl), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int)ctl[1];
}

// the phase stamps of the last traced launches (STAGE_TRACE): 24 64-bit tick counts (100 MHz); synchronises the device
int clhip_stage_train_trace(void* xch, unsigned long long* out24) {
    if (xch == nullptr || out24 == nullptr) return CLHIP_EINVAL;
    if (hipMemcpy(out24, static_cast<char*>(xch) + 32, 24 * 8, hipMemcpyDeviceToHost) != hipSuccess) return CLHIP_EHIP;
    return CLHIP_OK;
}
