// shortcut.hip -- the input gradient of the 1x1 / stride-2 convolutions of the ResNet-18 shortcuts (`downsample` of
// core/model/backbone/resnet.py:226-234), bf16, gfx950.
//
// dx[pin(p)][c] += sum_o dz[p][o] w[o][c] with pin(n, ho, wo) = (n, 2 ho, 2 wo): a tiny memory-bound GEMM scattered over a quarter of the
// rows of dx.  The generic implicit-GEMM kernel walks ALL of dx for it (31.8 / 18.2 us for layer2.0 / layer3.0 at batch 256); here only
// the touched rows are read and written (10.6 / 13.2 us).  Nothing is staged through LDS: with v_mfma_f32_32x32x16_bf16, A = a weight
// fragment (32 input channels x 16 reduction elements: 16 bytes of one row of the dgrad shadow per lane, L1 / L2 resident), B = a pixel
// fragment (16 bytes of one dz row per lane).  A wave owns 32 pixels x 128 (or 64) channels; D[row = channel][col = pixel] gives every lane 4
// consecutive channels per group and the half-waves are paired with v_permlane32_swap into 16-byte accesses, as in conv4.hip.  The three
// quarters of dx this layer does not touch keep what the main branch wrote (accumulate) or get zeros from the lane that owns their 2 x 2
// cell (first writer: in the reverse sweep of a residual block the shortcut comes before the main branch).  The forward of these layers and
// layer4.0's gradient (4096 pixels x 512 features: each wave would re-read a 64 KB weight panel for 32 pixels) measured slower this way
// than through the generic kernel and stay there (profiles/r02_layer_roofline.md).
#include <stdlib.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct ShortParams {
    const bf16_t* src;   // dz [N,Ho,Wo,R]
    const bf16_t* wt;    // [F][R]: the dgrad shadow [C][1][K]
    bf16_t* dst;         // dx [N,H,W,F], accumulated
    int N, H, W, Ho, Wo, R, F, M;        // R reduction length, F features, M = N*Ho*Wo pixels of the small grid
};

template <int NT, bool ACC>    // NT 32-channel tiles per workgroup (2 or 4); ACC: dx += (else dx = , zeros on the three sibling pixels of every 2x2 cell)
__global__ __launch_bounds__(256) void shortcut_dgrad_kernel(const ShortParams p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int f0 = blockIdx.y * (NT * 32);
    const int px = blockIdx.x * 128 + wave * 32 + l31;
    const bool valid = px < p.M;
    const int hw = p.Ho * p.Wo;
    const int n = px / hw, rem = px - n * hw, ho = rem / p.Wo, wo = rem - ho * p.Wo;
    const size_t pin = ((size_t)n * p.H + 2 * ho) * p.W + 2 * wo;          // the pixel of the large grid
    const bf16_t* prow = p.src + (size_t)px * p.R + kh * 8;
    const bf16_t* wrow[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wrow[t] = p.wt + (size_t)(f0 + t * 32 + l31) * p.R + kh * 8;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int k = 0; k < p.R; k += 32) {                       // two MFMA K steps per iteration: four independent loads per operand row in flight
        uint4 b0 = make_uint4(0, 0, 0, 0), b1 = b0;
        if (valid) { b0 = *reinterpret_cast<const uint4*>(prow + k); b1 = *reinterpret_cast<const uint4*>(prow + k + 16); }
        uint4 a0[NT], a1[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { a0[t] = *reinterpret_cast<const uint4*>(wrow[t] + k); a1[t] = *reinterpret_cast<const uint4*>(wrow[t] + k + 16); }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a0[t]), __builtin_bit_cast(bf16x8_t, b0), acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a1[t]), __builtin_bit_cast(bf16x8_t, b1), acc[t], 0, 0, 0);
        }
    }
    // D[row = feature (r & 3) + 8 (r >> 2) + 4 kh][col = pixel l31]
    bf16_t* drow = p.dst + pin * p.F + f0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (ACC && valid) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const uint2 old = *reinterpret_cast<const uint2*>(drow + t * 32 + g4 * 8 + kh * 4);
                acc[t][4 * g4 + 0] += __uint_as_float(old.x << 16); acc[t][4 * g4 + 1] += __uint_as_float(old.x & 0xffff0000u);
                acc[t][4 * g4 + 2] += __uint_as_float(old.y << 16); acc[t][4 * g4 + 3] += __uint_as_float(old.y & 0xffff0000u);
            }
        }
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            unsigned ax = pack_bf16x2(acc[t][8 * pr + 0], acc[t][8 * pr + 1]), ay = pack_bf16x2(acc[t][8 * pr + 2], acc[t][8 * pr + 3]);
            unsigned bx = pack_bf16x2(acc[t][8 * pr + 4], acc[t][8 * pr + 5]), by = pack_bf16x2(acc[t][8 * pr + 6], acc[t][8 * pr + 7]);
            auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
            auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
            if (valid) {
                bf16_t* q = drow + t * 32 + pr * 16 + kh * 8;
                *reinterpret_cast<u32x4*>(q) = u32x4{rx[0], ry[0], rx[1], ry[1]};
                if (!ACC) {
                    const u32x4 zero{0u, 0u, 0u, 0u};
                    *reinterpret_cast<u32x4*>(q + p.F) = zero;
                    *reinterpret_cast<u32x4*>(q + (size_t)p.W * p.F) = zero;
                    *reinterpret_cast<u32x4*>(q + (size_t)(p.W + 1) * p.F) = zero;
                }
            }
        }
    }
}

int launch_short(const ShortParams& p, bool accumulate, hipStream_t st) {
    const int gx = (p.M + 127) / 128;
    if (p.F % 128 == 0) {
        if (accumulate) hipLaunchKernelGGL((shortcut_dgrad_kernel<4, true>), dim3(gx, p.F / 128), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((shortcut_dgrad_kernel<4, false>), dim3(gx, p.F / 128), dim3(256), 0, st, p);
    } else {
        if (accumulate) hipLaunchKernelGGL((shortcut_dgrad_kernel<2, true>), dim3(gx, p.F / 64), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((shortcut_dgrad_kernel<2, false>), dim3(gx, p.F / 64), dim3(256), 0, st, p);
    }
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

}  // namespace

// 1x1, stride 2, pad 0, even image, bf16, channel counts multiples of 64, at least 8192 output pixels (CLHIP_SHORTCUT_MIN_PIXELS)
bool clhip_shortcut_supported(int N, int H, int W, int C, int K, int ksize, int stride, int pad, int dtype) {
    static const bool off = clhip_cfg("NO_SHORTCUT") != nullptr;
    static const long long min_px = clhip_cfg("SHORTCUT_MIN_PIXELS") ? atoll(clhip_cfg("SHORTCUT_MIN_PIXELS")) : 8192;
    return !off && dtype == CLHIP_BF16 && ksize == 1 && stride == 2 && pad == 0 && H % 2 == 0 && W % 2 == 0 && C % 64 == 0 && K % 64 == 0 && N >= 1 &&
           (long long)N * (H / 2) * (W / 2) >= min_px && (long long)N * H * W * (C > K ? C : K) * 2 < (1ll << 31);
}

int clhip_shortcut_dgrad(const void* dz, const void* w_dg, void* dx, int accumulate, int N, int H, int W, int C, int K, hipStream_t st) {
    ShortParams p{static_cast<const bf16_t*>(dz), static_cast<const bf16_t*>(w_dg), static_cast<bf16_t*>(dx), N, H, W, H / 2, W / 2, K, C, N * (H / 2) * (W / 2)};
    return launch_short(p, accumulate != 0, st);
}
