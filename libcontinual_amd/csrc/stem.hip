// stem.hip -- 3x3 / stride 1 / pad 1 convolution of an image with at most 8 (padded) input channels, bf16, gfx950: the ResNet stems
// (`conv1` of core/model/backbone/resnet.py:215-217, 381-383: 3 -> 16 / 3 -> 64 channels on 32 x 32 images).
//
// The generic implicit-GEMM kernel spent 33 us on ResNet-18's stem at batch 256 (0.9 GFLOP, 38 MB of traffic: a 5-us problem): its
// K dimension is 9 taps x 8 channels = 72, staged through LDS in 64-deep steps for 64 x 64 wave tiles.  Here nothing goes through LDS:
//  * v_mfma_f32_16x16x32_bf16 with the reduction index = (tap, channel): one K step is four taps x 8 padded channels, so the B operand
//    of a lane (pixel = lane & 15, tap = 4 ks + (lane >> 4)) is ONE 16-byte global load of that pixel's neighbour -- out-of-image taps
//    and the three padding taps (9, 10, 11) load nothing and multiply zeros; neighbouring pixels share their loads through L1 / L2;
//  * the weights (K x 72 bf16, at most 9 KB) live in registers for the whole kernel: 3 fragments per 16 output channels;
//  * D[row = output channel 4 g + e][col = pixel]: a lane stores 4 consecutive channels (8 bytes) per 16-channel tile; the BatchNorm
//    sums come from the fp32 accumulators (16-lane DPP sums, one LDS exchange per workgroup, fp64 atomics into the replicated
//    accumulators like conv4.hip).
#include <stdlib.h>

#include "common.h"

namespace {

struct StemParams {
    const bf16_t* x;     // [N,H,W,8]
    const bf16_t* w;     // [K][9][8]
    bf16_t* z;           // [N,H,W,K]
    double* acc;         // [rep][2][K] or nullptr
    int rep;
    int N, H, W, K, M;
};

template <int KT>        // 16-channel output tiles
__global__ __launch_bounds__(256) void conv_stem_kernel(const StemParams p) {
    __shared__ float red[4][2][KT * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int H = p.H, W = p.W;
    bf16x8_t a[KT][3];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const int tap = 4 * ks + g;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (tap < 9) v = *reinterpret_cast<const uint4*>(p.w + ((size_t)(kt * 16 + l15) * 9 + tap) * 8);
            a[kt][ks] = __builtin_bit_cast(bf16x8_t, v);
        }
    int dr[3], ds[3];
    bool tv[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) { const int tap = 4 * ks + g; tv[ks] = tap < 9; dr[ks] = tap / 3 - 1; ds[ks] = tap % 3 - 1; }
    float s1[KT][4], s2[KT][4];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1[kt][e] = 0.f; s2[kt][e] = 0.f; }

    const int ntile = (p.M + 63) / 64;                       // 64 pixels per wave and iteration
    for (int tile = blockIdx.x * 4 + wave; tile < ntile; tile += gridDim.x * 4) {
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const int px = tile * 64 + pt * 16 + l15;
            const bool valid = px < p.M;
            const int w0 = px % W, hn = px / W, h0 = hn % H;
            bf16x8_t b[3];
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                const int hh = h0 + dr[ks], ww = w0 + ds[ks];
                uint4 v = make_uint4(0, 0, 0, 0);
                if (valid && tv[ks] && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W)
                    v = *reinterpret_cast<const uint4*>(p.x + ((size_t)px + dr[ks] * W + ds[ks]) * 8);
                b[ks] = __builtin_bit_cast(bf16x8_t, v);
            }
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kt][ks], b[ks], c, 0, 0, 0);
                if (valid)
                    *reinterpret_cast<uint2*>(p.z + (size_t)px * p.K + kt * 16 + 4 * g) = make_uint2(pack_bf16x2(c[0], c[1]), pack_bf16x2(c[2], c[3]));
#pragma unroll
                for (int e = 0; e < 4; ++e) { s1[kt][e] += c[e]; s2[kt][e] = fmaf(c[e], c[e], s2[kt][e]); }     // pixels behind M multiplied zeros
            }
        }
    }
    if (p.acc == nullptr) return;
    // per-channel sums: over the 16 pixels of a row of lanes (DPP), over the workgroup's four waves (LDS), then one fp64 atomic each
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        float v[8] = {s1[kt][0], s1[kt][1], s1[kt][2], s1[kt][3], s2[kt][0], s2[kt][1], s2[kt][2], s2[kt][3]};
        row16_sum_n(v);
        if (l15 == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { red[wave][0][kt * 16 + 4 * g + e] = v[e]; red[wave][1][kt * 16 + 4 * g + e] = v[4 + e]; }
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * KT * 16; i += 256) {
        const int which = i / (KT * 16), ch = i - which * (KT * 16);
        const float t = red[0][which][ch] + red[1][which][ch] + red[2][which][ch] + red[3][which][ch];
        atomicAdd(p.acc + ((size_t)(blockIdx.x & (p.rep - 1)) * 2 + which) * p.K + ch, (double)t);
    }
}

}  // namespace

bool clhip_stem_supported(int N, int H, int W, int C, int K, int ksize, int stride, int pad, int dtype) {
    static const bool off = getenv("CLHIP_NO_STEM") != nullptr;
    return !off && dtype == CLHIP_BF16 && ksize == 3 && stride == 1 && pad == 1 && C == 8 && (K == 16 || K == 32 || K == 64) && N >= 1 && H >= 1 && W >= 1;
}

int clhip_stem_launch(const void* x, const void* w, void* z, double* acc, int rep, int N, int H, int W, int K, hipStream_t st) {
    StemParams p{static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(w), static_cast<bf16_t*>(z), acc, rep > 0 ? rep : 1, N, H, W, K, N * H * W};
    const int ntile = (p.M + 63) / 64;
    int grid = (ntile + 3) / 4;
    if (grid > 1024) grid = 1024;
    if (K == 16) hipLaunchKernelGGL(conv_stem_kernel<1>, dim3(grid), dim3(256), 0, st, p);
    else if (K == 32) hipLaunchKernelGGL(conv_stem_kernel<2>, dim3(grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(conv_stem_kernel<4>, dim3(grid), dim3(256), 0, st, p);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}
