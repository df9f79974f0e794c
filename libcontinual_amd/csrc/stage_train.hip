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
    XchBuf xb;
    StConv c[kMaxConvT];
};

template <int C, int HW>
struct StGeo {
    static constexpr int P = HW + 2;                       // padded width
    static constexpr int PB = 2 * C + 16;                  // bytes per pixel in LDS (16 consecutive pixels of a fragment read fall into disjoint banks)
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

template <int C, int HW>
__global__ __launch_bounds__(256) void stage_train_fwd_kernel(const StFwdParams p) {
    using G = StGeo<C, HW>;
    constexpr int P = G::P, PB = G::PB, BUF = G::BUF, PTW = G::PTW, KTW = G::KTW, KS = G::KS, WK = G::WK;
    constexpr int LOGC = C == 16 ? 4 : (C == 32 ? 5 : 6);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* X = smem;
    char* Y = smem + BUF;
    float* tab = reinterpret_cast<float*>(smem + 2 * BUF);        // [C][2]: scale, shift of the convolution in flight
    float* red = tab + 2 * C;                                      // [4 waves][2][C]
    float* vals = red + 8 * C;                                     // [2][C]: this image's sums
    double* tot = reinterpret_cast<double*>(vals + 2 * C);         // [2][C]: the batch's sums
    double* scratch = tot + 2 * C;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int wp = wave / WK, wk = wave % WK;
    const int img = blockIdx.x;
    const unsigned base = xch_base(p.xb);

    // ---- zero both buffers (the halo rings stay zero for the whole run), land the image in X
    for (int o = tid * 16; o < 2 * BUF; o += 256 * 16) *reinterpret_cast<uint4*>(smem + o) = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    {
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
    auto conv = [&](int cv, const bf16x8_t (&wf)[KTW][KS]) {
        const bool second = cv & 1;                                 // second convolution of a block: reads Y, adds the block input from X, writes X
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
        bf16x8_t xr[3];
        xr[0] = xread(0);
        if (SLOTS > 1) xr[1] = xread(1);
#pragma unroll
        for (int n = 0; n < SLOTS; ++n) {
            const int s_ = n / PTW, t_ = n - s_ * PTW;
            if (n + 2 < SLOTS) xr[(n + 2) % 3] = xread(n + 2);
#pragma unroll
            for (int kt = 0; kt < KTW; ++kt) acc[t_][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kt][s_], xr[n % 3], acc[t_][kt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        st_stamp(p.xb, tr, 1);
        // ---- z (bf16) to global memory for the backward; per-image channel sums of the fp32 accumulators
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
                *reinterpret_cast<uint2*>(cc.z + ((size_t)img * HW * HW + pixq[t]) * C + ch) = make_uint2(zp[t][kt][0], zp[t][kt][1]);
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
        const unsigned tag = base + (unsigned)cv + 1u;
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
        // ---- BatchNorm (+ block input) + ReLU on the bf16-rounded z, into the other LDS buffer; block outputs also go to global memory with their packed mask
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
                if (second) {
                    const uint2 r = *reinterpret_cast<const uint2*>(D + o + ch * 2);
                    v[0] += __uint_as_float(r.x << 16); v[1] += __uint_as_float(r.x & 0xffff0000u);
                    v[2] += __uint_as_float(r.y << 16); v[3] += __uint_as_float(r.y & 0xffff0000u);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                const uint2 out = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                *reinterpret_cast<uint2*>(D + o + ch * 2) = out;
                if (cc.y != nullptr) {
                    const size_t at = ((size_t)img * HW * HW + pixq[t]) * C + ch;
                    *reinterpret_cast<uint2*>(cc.y + at) = out;
                    if (cc.mask != nullptr) {
                        unsigned nib = 0;
                        nib |= ((out.x & 0x7fffu) != 0u && (out.x & 0x8000u) == 0u) ? 1u : 0u;
                        nib |= ((out.x & 0x7fff0000u) != 0u && (out.x & 0x80000000u) == 0u) ? 2u : 0u;
                        nib |= ((out.y & 0x7fffu) != 0u && (out.y & 0x8000u) == 0u) ? 4u : 0u;
                        nib |= ((out.y & 0x7fff0000u) != 0u && (out.y & 0x80000000u) == 0u) ? 8u : 0u;
                        const unsigned other = (unsigned)__shfl_xor((int)nib, 16, 64);      // the lane with the neighbouring four channels
                        if ((g & 1) == 0) cc.mask[at >> 3] = (unsigned char)(nib | (other << 4));
                    }
                }
            }
        }
        __syncthreads();
        st_stamp(p.xb, tr, 6);
    };
    bf16x8_t wa[KTW][KS], wb[KTW][KS];
    load_filters(0, wa);
    for (int cv = 0; cv < p.nconv; cv += 2) {                       // (nconv is even: pairs of convolutions = BasicBlocks)
        load_filters(cv + 1, wb);
        conv(cv, wa);
        if (cv + 2 < p.nconv) load_filters(cv + 2, wa);
        conv(cv + 1, wb);
    }
    if (img == 0 && tid == 0) xch_advance(p.xb, base, (unsigned)p.nconv);
}

template <int C, int HW>
int launch_fwd(const StFwdParams& p, hipStream_t st) {
    constexpr int lds = StGeo<C, HW>::LDS_FWD;
    int dev = 0;
    (void)hipGetDevice(&dev);
    static bool attr[16] = {};
    if (dev < 0 || dev >= 16 || !attr[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(stage_train_fwd_kernel<C, HW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            clhip_set_error("stage_train_fwd: cannot reserve %d bytes of LDS", lds);
            return CLHIP_EHIP;
        }
        if (dev >= 0 && dev < 16) attr[dev] = true;
    }
    hipLaunchKernelGGL((stage_train_fwd_kernel<C, HW>), dim3(p.N), dim3(256), lds, st, p);
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
    float* slab;              // [N][C * 9 * C] weight-gradient partial sums, one block per image
};

struct StBwdParams {
    const bf16_t* x;          // the run's input activation
    const bf16_t* dy;         // gradient of the run's output activation
    bf16_t* dx;               // gradient of the run's input activation (written, or accumulated into when dx_acc)
    int dx_acc;
    int N, nconv;
    double invM;
    int trace;
    XchBuf xb;
    StConvB c[kMaxConvT];
};

template <int C, int HW>
struct StGeoB {
    using G = StGeo<C, HW>;
    static constexpr int RED = C == 16 ? 4 * 2304 * 4 : 0;          // cross-wave sum of the 16-channel weight gradient
    static constexpr int AUX = 8 * C * 4 + 8 * C * 4 + 2 * C * 4 + 2 * C * 8 + kXchScratchDoubles * 8;      // ctab, red, vals, tot, scratch
    static constexpr int LDS = 2 * G::BUF + AUX + RED;
};

__device__ __forceinline__ uint4 st_tr8(const char* base, int addr, int second) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + addr));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + addr + second));
    uint2 l = __builtin_bit_cast(uint2, lo), h = __builtin_bit_cast(uint2, hi);
    return make_uint4(l.x, l.y, h.x, h.y);
}

// dw[k][tap][c] of ONE image: A = dz^T (rows = output channels, reduction = 32 pixels), B = the input image shifted by the tap, both fetched from the pixel-major
// padded LDS images with transposing reads (ds_read_b64_tr_b16: 16 lanes fetch a [4 pixels][16 channels] block, lane i keeps channel i).  `slab`: this image's block.
template <int C, int HW>
__device__ __forceinline__ void st_wgrad(const char* D, const char* XA, float* red, float* slab) {
    using G = StGeo<C, HW>;
    constexpr int P = G::P, PB = G::PB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int seg = (fr & 3) * 8;
    if constexpr (C == 16) {
        // a K step = one image row of 32 pixels; wave w takes rows w, w + 4, ...; the four partial tiles are summed through LDS
        f32x4 acc[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int col = fg * 8 + (fr >> 2);
        for (int h = wave; h < HW; h += 4) {
            const int pb = ((h + 1) * P + col + 1) * PB + seg;
            const uint4 zf = st_tr8(D, pb, 4 * PB);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int r = t / 3, sx = t - 3 * r;
                const uint4 xf = st_tr8(XA, pb + ((r - 1) * P + (sx - 1)) * PB, 4 * PB);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, zf), __builtin_bit_cast(bf16x8_t, xf), acc[t], 0, 0, 0);
            }
        }
        // D[row = out channel fg * 4 + e][col = in channel fr]
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wave * 2304 + ((fg * 4 + e) * 9 + t) * 16 + fr] = acc[t][e];
        __syncthreads();
        for (int i = tid; i < 2304; i += 256) slab[i] = ((red[i] + red[2304 + i]) + red[4608 + i]) + red[6912 + i];
    } else if constexpr (C == 32) {
        // a K step = two image rows of 16 pixels; wave = (input-channel tile, tap half) over all K steps and both output-channel tiles: no cross-wave sum
        const int it = wave & 1, th = wave >> 1, t0 = th * 5, nt = th == 0 ? 5 : 4;
        const int prow = fg >> 1, pcol = (fg & 1) * 8 + (fr >> 2);
        f32x4 acc[2][5];
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int q = 0; q < 5; ++q) acc[o][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
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
        // a K step = four image rows of 8 pixels; wave = input-channel tile, nine taps, the four output-channel tiles in two passes of two (72 accumulator
        // registers instead of 144: the backward keeps its gradient tile, the residual gradient and the dgrad filters live across this call)
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

template <int C, int HW>
__global__ __launch_bounds__(256) void stage_train_bwd_kernel(const StBwdParams p) {
    using G = StGeo<C, HW>;
    using GB = StGeoB<C, HW>;
    constexpr int P = G::P, PB = G::PB, BUF = G::BUF, PTW = G::PTW, KTW = G::KTW, KS = G::KS, WK = G::WK;
    constexpr int LOGC = C == 16 ? 4 : (C == 32 ? 5 : 6);
    constexpr int CPP = C / 8;
    constexpr int NCH = HW * HW * CPP / 256;                       // 16-byte chunks of an image per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* D = smem;                                                // dz of the unit in flight, zero-haloed
    char* XA = smem + BUF;                                         // input image of the unit whose weight gradient is pending, zero-haloed
    float* ctab = reinterpret_cast<float*>(smem + 2 * BUF);        // [5][C]: mean(g), mean(g xhat), -, -, gamma invstd
    float* red = ctab + 8 * C;                                     // [4 waves][2][C]
    float* vals = red + 8 * C;                                     // [2][C]
    double* tot = reinterpret_cast<double*>(vals + 2 * C);
    double* scratch = tot + 2 * C;
    float* wred = reinterpret_cast<float*>(smem + 2 * BUF + GB::AUX);      // (C == 16) cross-wave sum of the weight gradient
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4;
    const int wp = wave / WK, wk = wave % WK;
    const int img = blockIdx.x;
    const unsigned base = xch_base(p.xb);
    const size_t ibase = (size_t)img * HW * HW;

    for (int o = tid * 16; o < 2 * BUF; o += 256 * 16) *reinterpret_cast<uint4*>(smem + o) = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();

    // ---- state that lives across units.  Everything a unit needs from global memory is requested one stage ahead of its use:
    //   zq / yq / mu4 .. : z, the block output and the saved statistics of the NEXT unit to process at this lane's positions -- requested before the dgrad of the
    //                      unit above it multiplies;  xin: the unit's input image -- requested before the weight gradient of the unit above runs
    f32x4 acc[PTW][KTW];
    unsigned gres[PTW][KTW][2];                                     // the residual gradient of the block in flight (bf16 pairs)
    unsigned zq[PTW][KTW][2], yq[PTW][KTW][2];
    float4 mu4[KTW], is4[KTW], ga4[KTW], be4[KTW];
    float c_gamma = 0.f, c_invstd = 0.f, c_dg = 0.f, c_db = 0.f;    // thread c < C: its channel's parameters; workgroup 0: the old dgamma / dbeta

    auto prefetch = [&](int cv, int l15) {
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
                if (cv & 1) { const uint2 yy = *reinterpret_cast<const uint2*>(cc.y + at); yq[t][kt][0] = yy.x; yq[t][kt][1] = yy.y; }
            }
        }
        if (tid < C) {
            c_gamma = cc.gamma[tid]; c_invstd = cc.invstd[tid];
            if (img == 0) { c_dg = cc.dgamma[tid]; c_db = cc.dbeta[tid]; }
        }
    };

    auto unit = [&](int cv) {
        const bool second = cv & 1;
        const StConvB& cc = p.c[cv];
        const bool tr = p.trace > 0 && cv == p.trace && img == 0;
        st_stamp(p.xb, tr, 8);
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
                if (second) { gres[t][kt][0] = gq[t][kt][0]; gres[t][kt][1] = gq[t][kt][1]; }
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
        __syncthreads();
        st_stamp(p.xb, tr, 10);
        // ---- B: the batch's sums are on their way ...
        const unsigned tag = base + (unsigned)(p.nconv - 1 - cv) + 1u;
        xch_begin(p.xb, img, p.N, 2 * C, tag, vals, scratch);
        st_stamp(p.xb, tr, 11);
        // ---- C: ... while this unit's input image is requested (the block input -- a written activation -- for the first convolution of a block; z of the first
        //      convolution for the second one, turned into relu(scale z + shift) with the forward's expressions on its way into LDS) and the weight gradient of
        //      the unit above runs (its dz in D, its input image in XA)
        uint4 xin[NCH];
        float xsc[8], xsh[8];
        {
            const bf16_t* src = second ? p.c[cv - 1].z : (cv == 0 ? p.x : p.c[cv - 1].y);
            const uint4* s4 = reinterpret_cast<const uint4*>(src + ibase * C);
#pragma unroll
            for (int k = 0; k < NCH; ++k) xin[k] = s4[tid + k * 256];
            if (second) {
                const StConvB& a = p.c[cv - 1];
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
        }
        if (cv + 1 < p.nconv) {
            st_wgrad<C, HW>(D, XA, wred, p.c[cv + 1].slab + (size_t)img * (9 * C * C));
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
                *reinterpret_cast<uint2*>(D + pbase[t] + (P + 1) * PB + ch * 2) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
            }
        }
        __syncthreads();
        st_stamp(p.xb, tr, 15);
        // ---- E: the gradient of this unit's input; the operands of the unit below are requested first
        if (cv > 0) prefetch(cv - 1, l15);
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
        bf16x8_t xr[3];
        xr[0] = xread(0);
        if (SLOTS > 1) xr[1] = xread(1);
#pragma unroll
        for (int n = 0; n < SLOTS; ++n) {
            const int s_ = n / PTW, t_ = n - s_ * PTW;
            if (n + 2 < SLOTS) xr[(n + 2) % 3] = xread(n + 2);
#pragma unroll
            for (int kt = 0; kt < KTW; ++kt) acc[t_][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kt][s_], xr[n % 3], acc[t_][kt], 0, 0, 0);
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
                    acc[t][kt][0] = __uint_as_float(d0 << 16) + __uint_as_float(gres[t][kt][0] << 16);
                    acc[t][kt][1] = __uint_as_float(d0 & 0xffff0000u) + __uint_as_float(gres[t][kt][0] & 0xffff0000u);
                    acc[t][kt][2] = __uint_as_float(d1 << 16) + __uint_as_float(gres[t][kt][1] << 16);
                    acc[t][kt][3] = __uint_as_float(d1 & 0xffff0000u) + __uint_as_float(gres[t][kt][1] & 0xffff0000u);
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
                const uint2 v = *reinterpret_cast<const uint2*>(p.dy + (ibase + (wp * PTW + t) * 16 + l15) * C + ch);
                acc[t][kt] = (f32x4){__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u)};
                gres[t][kt][0] = gres[t][kt][1] = 0u;
            }
        }
    }
#pragma unroll 1
    for (int cv = p.nconv - 1; cv > 0; cv -= 2) {                  // (nconv is even: odd positions = second convolution of a block)
        unit(cv);
        unit(cv - 1);
    }
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
    // ---- the weight gradient of the first unit
    st_wgrad<C, HW>(D, XA, wred, p.c[0].slab + (size_t)img * (9 * C * C));
    if (img == 0 && tid == 0) xch_advance(p.xb, base, (unsigned)p.nconv);
}

template <int C, int HW>
int launch_bwd(const StBwdParams& p, hipStream_t st) {
    constexpr int lds = StGeoB<C, HW>::LDS;
    int dev = 0;
    (void)hipGetDevice(&dev);
    static bool attr[16] = {};
    if (dev < 0 || dev >= 16 || !attr[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(stage_train_bwd_kernel<C, HW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            clhip_set_error("stage_train_bwd: cannot reserve %d bytes of LDS", lds);
            return CLHIP_EHIP;
        }
        if (dev >= 0 && dev < 16) attr[dev] = true;
    }
    hipLaunchKernelGGL((stage_train_bwd_kernel<C, HW>), dim3(p.N), dim3(256), lds, st, p);
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

// One entry per convolution of the run (C ABI of the plan: plain arrays of pointers).  x: the run's input activation [N][H][W][C] bf16.
int clhip_stage_train_fwd_launch(const void* x, int N, int H, int W, int C, int nconv, const void* const* w, const float* const* gamma, const float* const* beta,
                                 float* const* rm, float* const* rv, float* const* mean, float* const* invstd, float* const* coef, void* const* z, void* const* y,
                                 void* const* mask, float momentum, float eps, void* xch, int trace, int dtype, hipStream_t st) {
    if (!clhip_stage_train_supported(N, H, W, C, nconv, dtype) || xch == nullptr) { clhip_set_error("stage_train_fwd: unsupported geometry"); return CLHIP_EINVAL; }
    StFwdParams p;
    p.x = static_cast<const bf16_t*>(x); p.N = N; p.nconv = nconv; p.eps = eps; p.momentum = momentum; p.trace = trace;
    const double M = (double)N * H * W;
    p.invM = 1.0 / M; p.unbias = M > 1.0 ? M / (M - 1.0) : 1.0;
    p.xb = xch_carve(xch, N, 128);
    for (int i = 0; i < nconv; ++i) {
        StConv& c = p.c[i];
        c.w = static_cast<const bf16_t*>(w[i]); c.wd = nullptr; c.gamma = gamma[i]; c.beta = beta[i]; c.rm = rm[i]; c.rv = rv[i];
        c.mean = mean[i]; c.invstd = invstd[i]; c.coef = coef[i]; c.z = static_cast<bf16_t*>(z[i]);
        c.y = static_cast<bf16_t*>(y[i]); c.mask = static_cast<unsigned char*>(mask[i]);
    }
    if (C == 16) return launch_fwd<16, 32>(p, st);
    if (C == 32) return launch_fwd<32, 16>(p, st);
    return launch_fwd<64, 8>(p, st);
}

// The backward of the same run.  dy: gradient of the run's output activation, dx: gradient of its input activation (accumulated into when dx_accumulate), slab[i]:
// N x (C * 9 * C) floats of scratch per convolution -- the caller adds the N blocks to the weight gradient in a fixed order (clhip_wgrad_reduce_launch).
int clhip_stage_train_bwd_launch(const void* x, const void* dy, void* dx, int dx_accumulate, int N, int H, int W, int C, int nconv, const void* const* wd,
                                 const float* const* gamma, const float* const* beta, const float* const* mean, const float* const* invstd, const void* const* z,
                                 const void* const* y, float* const* dgamma, float* const* dbeta, float* const* slab, void* xch, int trace, int dtype, hipStream_t st) {
    if (!clhip_stage_train_supported(N, H, W, C, nconv, dtype) || xch == nullptr) { clhip_set_error("stage_train_bwd: unsupported geometry"); return CLHIP_EINVAL; }
    StBwdParams p;
    p.x = static_cast<const bf16_t*>(x); p.dy = static_cast<const bf16_t*>(dy); p.dx = static_cast<bf16_t*>(dx); p.dx_acc = dx_accumulate;
    p.N = N; p.nconv = nconv; p.invM = 1.0 / ((double)N * H * W); p.trace = trace;
    p.xb = xch_carve(xch, N, 128);
    for (int i = 0; i < nconv; ++i) {
        StConvB& c = p.c[i];
        c.wd = static_cast<const bf16_t*>(wd[i]); c.gamma = gamma[i]; c.beta = beta[i]; c.mean = mean[i]; c.invstd = invstd[i];
        c.z = static_cast<const bf16_t*>(z[i]); c.y = static_cast<const bf16_t*>(y[i]); c.dgamma = dgamma[i]; c.dbeta = dbeta[i]; c.slab = slab[i];
    }
    if (C == 16) return launch_bwd<16, 32>(p, st);
    if (C == 32) return launch_bwd<32, 16>(p, st);
    return launch_bwd<64, 8>(p, st);
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
