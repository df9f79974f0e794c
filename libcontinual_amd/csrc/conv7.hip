// conv7.hip -- input gradient of a down-sampling block entry with FEW channels, bf16, gfx950: the CifarResNet-32 stage entries
// (core/model/backbone/resnet.py:226-234 with 16 -> 32 and 32 -> 64 channels; `conv1` 3x3 / s2 / p1 and `downsample[0]` 1x1 / s2 / p0 read the
// same block input), dx (+)= dgrad3x3s2(dz, W) + dgrad1x1s2(dz_sc, W_sc) in ONE launch -- conv6.hip's job for channel counts below its 64-wide tiles.
//
// On the generic implicit-GEMM kernel the two launches cost 26.7 + 13.8 us (16 -> 32 at batch 256) and 10.5 + 5.9 us (32 -> 64) for 17 / 8 MB of
// traffic: a 16-wide output tile, a 288-deep gather per pixel.  Here the stride is resolved by PARITY CLASS as in conv6.hip: the four input
// pixels (2a + ph, 2b + pw) of a 2 x 2 quad take 1 / 2 / 2 / 4 of the nine taps, all from the four gradient pixels (a, b), (a, b + 1), (a + 1, b),
// (a + 1, b + 1); the shortcut's single tap lands on class (0, 0).  A 16-quad tile is ten 16 x 16 x 32 MFMAs per 16 input channels and 32 gradient
// channels, the weight fragments (ten taps) live in registers for the life of a wave, the gradient fragments are 16-byte loads straight from global
// memory (each gradient pixel is fetched by four lanes' roles: L1 / L2 hits), no LDS, no barrier.  A lane ends up with four consecutive channels
// of one input pixel per class: 8-byte stores.
//
// Weights come packed as [C][10][K] (taps 0..8 = the 3x3 layer's dgrad copy [C][9][K], tap 9 = the shortcut's [C][1][K]); a plan writes that
// layout in its own weight preparation, clhip_conv_dgrad_pair_pack makes it from the two copies.
#include <stdlib.h>

#include "common.h"

namespace {

struct Dgrad7Params {
    const bf16_t* dz;    // [N,Ho,Wo,K]
    const bf16_t* dzs;   // [N,Ho,Wo,K] gradient of the shortcut's output, or nullptr
    const bf16_t* wpk;   // [C][10][K]
    bf16_t* dx;          // [N,2Ho,2Wo,C]
    int Ho, Wo, lgHo, lgWo, Mq, accumulate, ntiles, tpw;      // Mq = N*Ho*Wo quads, tiles of 16 quads, tiles per wave
};

template <int C, bool SC>
__global__ __launch_bounds__(256) void dgrad7_kernel(const Dgrad7Params p) {
    constexpr int K = 2 * C, KS = K / 32, NTAP = SC ? 10 : 9;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int ct = blockIdx.y;                               // 16-channel tile of the input gradient

    uint4 w[NTAP][KS];                                       // A operand: row = input channel ct * 16 + fr, 8 gradient channels per lane and K step
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) w[t][ks] = *reinterpret_cast<const uint4*>(p.wpk + ((size_t)(ct * 16 + fr) * 10 + t) * K + ks * 32 + fg * 8);

    const int Wo = p.Wo, Ho = p.Ho, W2 = 2 * Wo;
    const int t0 = (blockIdx.x * 4 + wave) * p.tpw;
    for (int i = 0; i < p.tpw; ++i) {
        const int tile = t0 + i;
        if (tile >= p.ntiles) break;                         // uniform over the wave
        const int q = tile * 16 + fr;
        const bool qv = q < p.Mq;
        const int b = q & (Wo - 1), a = (q >> p.lgWo) & (Ho - 1), n = q >> (p.lgWo + p.lgHo);
        const bool rv = qv && b + 1 < Wo, dv = qv && a + 1 < Ho;
        // B operand: column = quad fr, 8 gradient channels per lane and K step, of the four neighbours (and the shortcut's gradient)
        uint4 z00[KS], z01[KS], z10[KS], z11[KS], zs[KS];
        const uint4 zero = make_uint4(0, 0, 0, 0);
        const bf16_t* zp = p.dz + (size_t)q * K + fg * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            z00[ks] = qv ? *reinterpret_cast<const uint4*>(zp + ks * 32) : zero;
            z01[ks] = rv ? *reinterpret_cast<const uint4*>(zp + K + ks * 32) : zero;
            z10[ks] = dv ? *reinterpret_cast<const uint4*>(zp + (size_t)Wo * K + ks * 32) : zero;
            z11[ks] = (rv && dv) ? *reinterpret_cast<const uint4*>(zp + (size_t)(Wo + 1) * K + ks * 32) : zero;
            if (SC) zs[ks] = qv ? *reinterpret_cast<const uint4*>(p.dzs + (size_t)q * K + fg * 8 + ks * 32) : zero;
        }
        f32x4 acc[4];                                        // class 2 ph + pw: D[row = channel fg * 4 + e][col = quad fr]
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#define MM(cls, tap, zf) acc[cls] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w[tap][ks]), __builtin_bit_cast(bf16x8_t, zf[ks]), acc[cls], 0, 0, 0)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            // dx[h, w] = sum over taps (r, s) with (h + 1 - r, w + 1 - s) even of dz[(h + 1 - r) / 2, (w + 1 - s) / 2] . W[r][s]
            MM(0, 4, z00);                                   // (even, even): tap (1, 1)
            if (SC) MM(0, 9, zs);                            //               + the shortcut
            MM(1, 3, z01); MM(1, 5, z00);                    // (even, odd):  (1, 0) @ (a, b + 1), (1, 2) @ (a, b)
            MM(2, 1, z10); MM(2, 7, z00);                    // (odd, even):  (0, 1) @ (a + 1, b), (2, 1) @ (a, b)
            MM(3, 0, z11); MM(3, 2, z10); MM(3, 6, z01); MM(3, 8, z00);      // (odd, odd)
        }
#undef MM
        if (qv) {
#pragma unroll
            for (int cls = 0; cls < 4; ++cls) {
                bf16_t* o = p.dx + ((size_t)(n * 2 * Ho + 2 * a + (cls >> 1)) * W2 + 2 * b + (cls & 1)) * C + ct * 16 + fg * 4;
                float v0 = acc[cls][0], v1 = acc[cls][1], v2 = acc[cls][2], v3 = acc[cls][3];
                if (p.accumulate) {
                    const uint2 old = *reinterpret_cast<const uint2*>(o);
                    v0 += __uint_as_float(old.x << 16); v1 += __uint_as_float(old.x & 0xffff0000u);
                    v2 += __uint_as_float(old.y << 16); v3 += __uint_as_float(old.y & 0xffff0000u);
                }
                *reinterpret_cast<uint2*>(o) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
            }
        }
    }
}

// [C][9][K] and (nullable) [C][1][K] -> [C][10][K]
__global__ __launch_bounds__(256) void pack7_kernel(const bf16_t* __restrict__ w, const bf16_t* __restrict__ wsc, bf16_t* __restrict__ out, int C, int K) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= C * 10 * K) return;
    const int k = i % K, t = (i / K) % 10, c = i / (10 * K);
    out[i] = t < 9 ? w[((size_t)c * 9 + t) * K + k] : (wsc != nullptr ? wsc[(size_t)c * K + k] : (bf16_t)0);
}

int ilog2_7(int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; }

template <int C, bool SC>
int launch7(const Dgrad7Params& p, hipStream_t st) {
    const int waves = (p.ntiles + p.tpw - 1) / p.tpw;
    hipLaunchKernelGGL((dgrad7_kernel<C, SC>), dim3((waves + 3) / 4, C / 16), dim3(256), 0, st, p);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

}  // namespace

// (N, H, W, C) = the block input whose gradient is produced; K = 2 C channels of the two convolutions' outputs
bool clhip_dgrad7_supported(int N, int H, int W, int C, int K, int dtype) {
    static const bool off = clhip_cfg("CONV7") != nullptr && atoi(clhip_cfg("CONV7")) == 0;
    if (off || dtype != CLHIP_BF16 || (H & 1) || (W & 1) || !(C == 16 || C == 32) || K != 2 * C) return false;
    const int Ho = H / 2, Wo = W / 2;
    if (ilog2_7(Ho) < 0 || ilog2_7(Wo) < 0 || Wo < 2 || Ho < 2) return false;
    return (int64_t)N * H * W * C < ((int64_t)1 << 31);
}

size_t clhip_dgrad7_packed_bytes(int C, int K) { return (size_t)C * 10 * K * sizeof(bf16_t); }

int clhip_dgrad7_pack(const void* w_dg, const void* w_sc_dg, void* packed, int C, int K, hipStream_t st) {
    hipLaunchKernelGGL(pack7_kernel, dim3((C * 10 * K + 255) / 256), dim3(256), 0, st, static_cast<const bf16_t*>(w_dg), static_cast<const bf16_t*>(w_sc_dg),
                       static_cast<bf16_t*>(packed), C, K);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

int clhip_dgrad7_launch(const void* dz, const void* w_packed, const void* dz_sc, void* dx, int accumulate, int N, int H, int W, int C, int K, hipStream_t st) {
    Dgrad7Params p;
    p.dz = static_cast<const bf16_t*>(dz); p.dzs = static_cast<const bf16_t*>(dz_sc); p.wpk = static_cast<const bf16_t*>(w_packed); p.dx = static_cast<bf16_t*>(dx);
    p.Ho = H / 2; p.Wo = W / 2; p.lgHo = ilog2_7(p.Ho); p.lgWo = ilog2_7(p.Wo);
    p.Mq = N * p.Ho * p.Wo; p.accumulate = accumulate; p.ntiles = (p.Mq + 15) / 16;
    // tiles per wave: the weight fragments are loaded once per wave (10 - 20 KB out of L2), so a wave walks up to four tiles while that still
    // leaves every CU a workgroup
    static const int tpw_cfg = clhip_cfg("CONV7_TPW") ? atoi(clhip_cfg("CONV7_TPW")) : 0;
    int tpw = tpw_cfg > 0 ? tpw_cfg : 4;
    while (tpw > 1 && tpw_cfg <= 0 && (p.ntiles / tpw / 4) * (C / 16) < 256) tpw >>= 1;
    p.tpw = tpw;
    const bool sc = dz_sc != nullptr;
    if (C == 16) return sc ? launch7<16, true>(p, st) : launch7<16, false>(p, st);
    return sc ? launch7<32, true>(p, st) : launch7<32, false>(p, st);
}
