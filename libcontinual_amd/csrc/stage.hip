// stage.hip -- eval-mode forward of a RUN of BasicBlocks of the CIFAR ResNet-32s as ONE launch (round 5; VERDICT r2-r4 "stage-level kernels", the part that
// has no statistics barrier): frozen teachers (LwF / iCaRL / LUCIR / WA / BiC / DER steps), validation, herding / NCM feature extraction.
//
// Replaces, for blocks of conv3x3(C -> C, stride 1) -> BN -> ReLU -> conv3x3 -> BN -> (+ x) -> ReLU in eval mode
// (core/model/backbone/resnet.py:289-316, the stages built at :381-392), 2 x blocks launches of 6-8 us on activations of at most 8 MB by one:
//  * one workgroup (four waves) per IMAGE; the image's activation lives in LDS for the whole run -- two zero-haloed buffers X / Y of (HW + 2)^2 pixels,
//    pixel pitch 2 C + 16 bytes (16 consecutive pixels of a fragment read fall into disjoint banks; 32 bytes for 16 channels): 32 x 32 x 16: 2 x 37 KB, 16 x 16 x 32: 2 x 26 KB, 8 x 8 x 64: 2 x 14 KB;
//  * a convolution is an implicit GEMM on v_mfma_f32_16x16x32_bf16 with the FILTERS as the A operand (rows = output channels) and the pixels as B
//    (columns), K = tap * C + c: a lane then holds four consecutive output channels of one pixel and writes them with one 8-byte LDS store.  The filter
//    fragments of a convolution sit in registers (20 / 72 / 72 per wave; the next convolution's set is requested while this one multiplies), loaded from the
//    plan's bf16 [K][9][C] copy (L2 hits: every workgroup reads the same);
//  * BatchNorm (running statistics), the residual add and the ReLU run on the accumulators; conv1 of a block writes Y, conv2 adds the block input
//    from X and writes X in place (the lane that reads a residual element is the one that overwrites it), one workgroup barrier per convolution;
//  * the pre-BatchNorm value is rounded to bf16 first, as the unfused path stores it, so the two paths differ by summation order only.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int kMaxConv = 16;

struct StageConv {
    const bf16_t* w;          // [C][9][C]
    const float* gamma;
    const float* beta;
    const float* mean;
    const float* var;
};

struct StageParams {
    const bf16_t* x;          // [N][HW][HW][C]
    bf16_t* y;
    int N, nconv;
    float eps;
    StageConv c[kMaxConv];
};

template <int C, int HW>
struct StageGeo {
    static constexpr int P = HW + 2;                       // padded width
    static constexpr int PB = C == 16 ? 32 : 2 * C + 16;   // bytes per pixel in LDS (16 channels: the bare 32 bytes are conflict-free already -- conv16 -- and leave room for a training launch beside this one, stage_train.hip)
    static constexpr int BUF = P * P * PB;                 // one activation buffer
    static constexpr int NPT = HW * HW / 16;               // pixel tiles of 16
    static constexpr int KT = C / 16;                      // output-channel tiles of 16
    static constexpr int WK = C == 64 ? 4 : 1;             // waves along the output channels (64 channels: one 16-channel tile per wave, so that TWO filter sets fit its registers)
    static constexpr int WP = 4 / WK;                      // waves along the pixel tiles
    static constexpr int PTW = NPT / WP;                   // pixel tiles per wave
    static constexpr int KTW = KT / WK;                    // channel tiles per wave
    static constexpr int KS = (9 * C + 31) / 32;           // MFMA K steps
    static constexpr int TAB = kMaxConv * C * 2 * 4;       // scale / shift table
    static constexpr int LDS = 2 * BUF + TAB;
};

template <int C, int HW>
__global__ __launch_bounds__(256) void stage_eval_kernel(const StageParams p) {
    using G = StageGeo<C, HW>;
    constexpr int P = G::P, PB = G::PB, BUF = G::BUF, PTW = G::PTW, KTW = G::KTW, KS = G::KS;
    constexpr int LOGC = C == 16 ? 4 : (C == 32 ? 5 : 6);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* X = smem;
    char* Y = smem + BUF;
    float* tab = reinterpret_cast<float*>(smem + 2 * BUF);      // [conv][C][2]: scale, shift
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int wp = wave / G::WK, wk = wave % G::WK;
    const int img = blockIdx.x;

    // ---- zero both buffers (the halo rings stay zero for the whole run), build the scale / shift table, land the image in X
    for (int o = tid * 16; o < 2 * BUF; o += 256 * 16) *reinterpret_cast<uint4*>(smem + o) = make_uint4(0u, 0u, 0u, 0u);
    for (int i = tid; i < p.nconv * C; i += 256) {
        const int cv = i / C, ch = i - cv * C;
        // the expressions of bn_apply_eval_kernel (bn.hip): invstd = 1 / sqrt(var + eps), scale = gamma * invstd, shift = beta - mean * scale
        const float invstd = 1.0f / sqrtf(p.c[cv].var[ch] + p.eps);
        const float sc = p.c[cv].gamma[ch] * invstd;
        tab[i * 2] = sc;
        tab[i * 2 + 1] = p.c[cv].beta[ch] - p.c[cv].mean[ch] * sc;
    }
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

    // ---- per-lane constants: the B-operand (pixel) base of each of the wave's pixel tiles, the K-step offsets
    int pbase[PTW];                                                 // byte offset of pixel (y, x) shifted to tap (0, 0) = padded (y, x)
#pragma unroll
    for (int t = 0; t < PTW; ++t) {
        const int q = (wp * PTW + t) * 16 + l15;
        const int yy = q / HW, xx = q - yy * HW;
        pbase[t] = (yy * P + xx) * PB;
    }
    int koff[KS];                                                   // byte offset of this lane's 8 K elements of step s inside the patch; -1: beyond 9 C (zero)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int kk = 32 * s + 8 * g;
        const int tap = kk >> LOGC, c0 = kk & (C - 1);
        const int dy = (tap * 11) >> 5, dx = tap - 3 * dy;           // tap / 3 for tap < 9
        koff[s] = kk < 9 * C ? (dy * P + dx) * PB + c0 * 2 : -1;
    }

    // the filters of this wave's output-channel tiles: A operand, row = channel (wk KTW + kt) 16 + l15, 8 K elements at 32 s + 8 g.  The set of convolution cv + 1 is
    // requested while convolution cv multiplies (two register sets): a convolution is ~1 us of MFMAs, an L2 round trip in front of each would double it
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
        const bool second = cv & 1;                                 // conv2 of a block: reads Y, adds the block input from X, writes X
        const char* S = second ? Y : X;
        char* D = second ? X : Y;
        f32x4 acc[PTW][KTW];
#pragma unroll
        for (int t = 0; t < PTW; ++t)
#pragma unroll
            for (int kt = 0; kt < KTW; ++kt) acc[t][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // pixel fragments two (K step, pixel tile) slots ahead of their MFMAs (a ring of three); K elements beyond 9 C read the zero halo pixel at offset 0
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
        // ---- epilogue: lane = pixel l15 of the tile, channels (wk KTW + kt) 16 + 4 g .. + 3
        const float* tb = tab + cv * C * 2;
#pragma unroll
        for (int kt = 0; kt < KTW; ++kt) {
            const int ch = (wk * KTW + kt) * 16 + 4 * g;
            const float4 s01 = *reinterpret_cast<const float4*>(tb + ch * 2);
            const float4 s23 = *reinterpret_cast<const float4*>(tb + ch * 2 + 4);
#pragma unroll
            for (int t = 0; t < PTW; ++t) {
                const int o = pbase[t] + (P + 1) * PB;              // the pixel itself (interior position)
                float v[4];
                // z is stored as bf16 by the unfused path before its BatchNorm launch reads it: the same rounding here
                v[0] = fmaf(bf16_to_f32(f32_to_bf16(acc[t][kt][0])), s01.x, s01.y);      // (fmaf, then + res, then max: bn_apply_eval_kernel's order)
                v[1] = fmaf(bf16_to_f32(f32_to_bf16(acc[t][kt][1])), s01.z, s01.w);
                v[2] = fmaf(bf16_to_f32(f32_to_bf16(acc[t][kt][2])), s23.x, s23.y);
                v[3] = fmaf(bf16_to_f32(f32_to_bf16(acc[t][kt][3])), s23.z, s23.w);
                if (second) {
                    const uint2 r = *reinterpret_cast<const uint2*>(D + o + ch * 2);
                    v[0] += __uint_as_float(r.x << 16); v[1] += __uint_as_float(r.x & 0xffff0000u);
                    v[2] += __uint_as_float(r.y << 16); v[3] += __uint_as_float(r.y & 0xffff0000u);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                *reinterpret_cast<uint2*>(D + o + ch * 2) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            }
        }
        __syncthreads();
    };
    bf16x8_t wa[KTW][KS], wb[KTW][KS];
    load_filters(0, wa);
    for (int cv = 0; cv < p.nconv; cv += 2) {                       // (nconv is even: pairs of convolutions = BasicBlocks)
        load_filters(cv + 1, wb);
        conv(cv, wa);
        if (cv + 2 < p.nconv) load_filters(cv + 2, wa);
        conv(cv + 1, wb);
    }

    // ---- the run's output: X (an even number of convolutions) back to global memory
    {
        constexpr int CPP = C / 8;
        const char* R = (p.nconv & 1) ? Y : X;
        uint4* dst = reinterpret_cast<uint4*>(p.y + (size_t)img * HW * HW * C);
        for (int i = tid; i < HW * HW * CPP; i += 256) {
            const int q = i / CPP, cc = i - q * CPP;
            const int yy = q / HW, xx = q - yy * HW;
            dst[i] = *reinterpret_cast<const uint4*>(R + ((yy + 1) * P + xx + 1) * PB + cc * 16);
        }
    }
}

template <int C, int HW>
int launch_stage(const StageParams& p, hipStream_t st) {
    constexpr int lds = StageGeo<C, HW>::LDS;
    int dev = 0;
    (void)hipGetDevice(&dev);
    static bool attr[16] = {};
    if (dev < 0 || dev >= 16 || !attr[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(stage_eval_kernel<C, HW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            clhip_set_error("stage_eval: cannot reserve %d bytes of LDS", lds);
            return CLHIP_EHIP;
        }
        if (dev >= 0 && dev < 16) attr[dev] = true;
    }
    hipLaunchKernelGGL((stage_eval_kernel<C, HW>), dim3(p.N), dim3(256), lds, st, p);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

}  // namespace

bool clhip_stage_eval_supported(int H, int W, int C, int nconv, int dtype) {
    if (dtype != CLHIP_BF16 || H != W || nconv < 2 || nconv > kMaxConv || (nconv & 1)) return false;
    return (C == 16 && H == 32) || (C == 32 && H == 16) || (C == 64 && H == 8);
}

// x, y: [N][H][W][C] bf16 activations; w[i]: the [C][9][C] bf16 forward copy of convolution i; gamma / beta / mean / var[i]: its BatchNorm (fp32, eval mode)
int clhip_stage_eval_launch(const void* x, void* y, int N, int H, int W, int C, int nconv, const void* const* w, const float* const* gamma, const float* const* beta,
                            const float* const* mean, const float* const* var, float eps, int dtype, hipStream_t st) {
    if (!clhip_stage_eval_supported(H, W, C, nconv, dtype) || N < 1) { clhip_set_error("stage_eval: unsupported geometry"); return CLHIP_EINVAL; }
    StageParams p;
    p.x = static_cast<const bf16_t*>(x); p.y = static_cast<bf16_t*>(y); p.N = N; p.nconv = nconv; p.eps = eps;
    for (int i = 0; i < nconv; ++i) {
        p.c[i].w = static_cast<const bf16_t*>(w[i]); p.c[i].gamma = gamma[i]; p.c[i].beta = beta[i]; p.c[i].mean = mean[i]; p.c[i].var = var[i];
    }
    if (C == 16) return launch_stage<16, 32>(p, st);
    if (C == 32) return launch_stage<32, 16>(p, st);
    return launch_stage<64, 8>(p, st);
}
