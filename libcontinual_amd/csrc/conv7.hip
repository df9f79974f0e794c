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
    // BN: the BatchNorm-backward sums of the layer that PRODUCED the block input (sum g, sum g * xhat; g = dx masked by that layer's ReLU), from the
    // fp32 results before their rounding -- clhip_conv_dgrad_bn_reduce's epilogue (conv4.hip / conv3.hip) for this launch
    const bf16_t* bn_z = nullptr;    // [N,2Ho,2Wo,C] the producer's pre-BatchNorm output
    const bf16_t* bn_y = nullptr;    // its activation (mask y > 0), or nullptr: no ReLU
    const float* bn_mean = nullptr; const float* bn_invstd = nullptr;
    double* bn_acc = nullptr;        // [bn_rep][2][C]
    int bn_rep = 1;
};

template <int C, bool SC, bool BN = false>
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
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};      // BN: sum g, sum g z' of channels ct * 16 + fg * 4 + e over this lane's pixels
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
                if constexpr (BN) {
                    const size_t at = o - p.dx;
                    const uint2 zz = *reinterpret_cast<const uint2*>(p.bn_z + at);
                    uint2 yy = make_uint2(0x3f803f80u, 0x3f803f80u);
                    if (p.bn_y != nullptr) yy = *reinterpret_cast<const uint2*>(p.bn_y + at);
                    const float z4[4] = {__uint_as_float(zz.x << 16), __uint_as_float(zz.x & 0xffff0000u), __uint_as_float(zz.y << 16), __uint_as_float(zz.y & 0xffff0000u)};
                    const float y4[4] = {__uint_as_float(yy.x << 16), __uint_as_float(yy.x & 0xffff0000u), __uint_as_float(yy.y << 16), __uint_as_float(yy.y & 0xffff0000u)};
                    const float v4[4] = {v0, v1, v2, v3};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float g = y4[e] > 0.f ? v4[e] : 0.f;
                        s1[e] += g; s2[e] = fmaf(g, z4[e], s2[e]);
                    }
                }
            }
        }
    }
    if constexpr (BN) {
        // 16-lane sums, the four waves through LDS, one fp64 atomic per channel and sum per workgroup; the centred form
        // sum g xhat = invstd (sum g z' - mean sum g) is taken once per channel
        __shared__ float red[4][4][8];                       // [wave][fg][sum e, sum-z e]
        float sv[8] = {s1[0], s1[1], s1[2], s1[3], s2[0], s2[1], s2[2], s2[3]};
        row16_sum_n(sv);
        if (fr == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[wave][fg][e] = sv[e];
        }
        __syncthreads();
        if (tid < 16) {
            const int g = tid >> 2, e = tid & 3, c = ct * 16 + g * 4 + e;
            const float a1 = red[0][g][e] + red[1][g][e] + red[2][g][e] + red[3][g][e];
            const float a2 = red[0][g][4 + e] + red[1][g][4 + e] + red[2][g][4 + e] + red[3][g][4 + e];
            double* acc = p.bn_acc + (size_t)(blockIdx.x & (p.bn_rep - 1)) * 2 * C;
            atomicAdd(acc + c, (double)a1);
            atomicAdd(acc + C + c, (double)(p.bn_invstd[c] * (a2 - p.bn_mean[c] * a1)));
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
    if (p.bn_z != nullptr) hipLaunchKernelGGL((dgrad7_kernel<C, SC, true>), dim3((waves + 3) / 4, C / 16), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((dgrad7_kernel<C, SC>), dim3((waves + 3) / 4, C / 16), dim3(256), 0, st, p);
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

int clhip_dgrad7_launch(const void* dz, const void* w_packed, const void* dz_sc, void* dx, int accumulate, int N, int H, int W, int C, int K, hipStream_t st,
                        const void* bn_z, const void* bn_y, const float* bn_mean, const float* bn_invstd, double* bn_acc, int bn_rep) {
    Dgrad7Params p;
    p.bn_z = static_cast<const bf16_t*>(bn_z); p.bn_y = static_cast<const bf16_t*>(bn_y); p.bn_mean = bn_mean; p.bn_invstd = bn_invstd; p.bn_acc = bn_acc;
    p.bn_rep = bn_rep > 0 ? bn_rep : 1;
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

// =========================================================================================================================================
// The WEIGHT gradients of the same two layers in one launch:  dw3[k][tap][c] += sum_p dz[p][k] x[p @ tap][c]  (3x3 / s2 / p1) and
// dwsc[k][c] += sum_p dz_sc[p][k] x[2 ho, 2 wo][c]  (1x1 / s2) -- both read the block input x.  On the generic kernel's deterministic form they
// are 21.0 + 9.6 us (16 -> 32) and 21.0 + 9.6 us launches at batch 256 (profiles/r03_step_notes.md).  wgrad16 / wgrad32's scheme with a stride:
// a workgroup walks a group of images; the zero-padded input image and the two gradient images sit in LDS; 32 output pixels are one MFMA K step;
// both operands are pixel-major columns fetched with transposing reads (every lane of such a read has its own address, so the input pixels of a
// step are simply two apart); the ten taps (nine + the shortcut, which shares tap (1, 1)'s input fragment) are constant offsets.  Every wave owns
// its own output tiles over ALL pixels -- 16 -> 32: gradient-channel tile (wave & 1) x taps {0..4} / {5..9}; 32 -> 64: gradient-channel tile
// `wave` x both input-channel tiles x ten taps -- so there is no cross-wave sum; a group's partial blocks go to two slabs that the fixed-order
// reduce adds to the two gradients (bitwise reproducible).
namespace {

struct Wgrad7Params { const bf16_t* x; const bf16_t* dz; const bf16_t* dzs; float* slab3; float* slabsc; int N, ipg; };

__device__ __forceinline__ uint4 tr8_7(const char* base, int addr, int second) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + addr));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + addr + second));
    uint2 l = __builtin_bit_cast(uint2, lo), h = __builtin_bit_cast(uint2, hi);
    return make_uint4(l.x, l.y, h.x, h.y);
}

template <int C>
__global__ __launch_bounds__(256) void wgrad7_kernel(const Wgrad7Params p) {
    constexpr int K = 2 * C, W = C == 16 ? 32 : 16, H = W, Wo = W / 2, Ho = H / 2, PW = W + 2;
    constexpr int PXB = C * 2, PZB = K * 2;                  // bytes per input / gradient pixel
    constexpr int XS = (H + 2) * PW * PXB, ZS = Ho * Wo * PZB;
    constexpr int NCT = C / 16, NTW = C == 16 ? 5 : 10;      // input-channel tiles and taps per wave
    constexpr int STEPS = Ho * Wo / 32, RS = 32 / Wo;        // K steps per image, output rows per step
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs = smem;
    char* zs = smem + XS;
    char* zss = zs + ZS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int kt = C == 16 ? (wave & 1) : wave;
    const int t_lo = C == 16 ? (wave >> 1) * 5 : 0;
    const int grp = blockIdx.x;
    const int n_beg = grp * p.ipg, n_end = min(p.N, n_beg + p.ipg);

    for (int i = tid; i < XS / 16; i += 256) *reinterpret_cast<uint4*>(xs + i * 16) = make_uint4(0, 0, 0, 0);      // the border stays zero

    f32x4 acc[NCT][NTW];
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // lane (fr, fg) of a transposing read addresses reduction pixel k0 = 8 fg + (fr >> 2) of the step (+ 4 for the second read), 4-channel segment fr & 3
    const int k0 = fg * 8 + (fr >> 2), seg = (fr & 3) * 8;
    const int ho_l = k0 / Wo, wo = k0 % Wo;
    const int zaddr = k0 * PZB + kt * 32 + seg;
    const int xaddr = ((2 * ho_l) * PW + 2 * wo) * PXB + seg;           // padded coordinates of tap (0, 0) = input pixel (2 ho - 1, 2 wo - 1)
    constexpr int CPX = C / 8, CPZ = K / 8;                  // 16-byte chunks per pixel

    for (int n = n_beg; n < n_end; ++n) {
        __syncthreads();                                     // the previous image has been multiplied (first trip: the zero fill is complete)
        const bf16_t* xi = p.x + (size_t)n * H * W * C;
        for (int i = tid; i < H * W * CPX; i += 256) {
            const int px = i / CPX, ch = i - px * CPX, row = px / W, col = px - row * W;
            *reinterpret_cast<uint4*>(xs + ((row + 1) * PW + col + 1) * PXB + ch * 16) = *reinterpret_cast<const uint4*>(xi + (size_t)px * C + ch * 8);
        }
        const bf16_t* zi = p.dz + (size_t)n * Ho * Wo * K;
        const bf16_t* zsi = p.dzs + (size_t)n * Ho * Wo * K;
        for (int i = tid; i < Ho * Wo * CPZ; i += 256) {
            *reinterpret_cast<uint4*>(zs + i * 16) = *reinterpret_cast<const uint4*>(zi + (size_t)i * 8);
            *reinterpret_cast<uint4*>(zss + i * 16) = *reinterpret_cast<const uint4*>(zsi + (size_t)i * 8);
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const uint4 zf = tr8_7(zs, zaddr + s * 32 * PZB, 4 * PZB);
            uint4 zsf = make_uint4(0, 0, 0, 0);
            if (C == 32 || t_lo == 5) zsf = tr8_7(zss, zaddr + s * 32 * PZB, 4 * PZB);
            const int xb = xaddr + s * RS * 2 * PW * PXB;
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
#pragma unroll
                for (int t = 0; t < NTW; ++t) {
                    const int tap = t_lo + t;                // 9 = the shortcut: tap (1, 1)'s input fragment, the shortcut's gradient
                    const int tt = tap == 9 ? 4 : tap;
                    const int r = tt / 3, sx = tt - 3 * r;
                    const uint4 xf = tr8_7(xs, xb + (r * PW + sx) * PXB + c * 32, 8 * PXB);
                    acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, tap == 9 ? zsf : zf), __builtin_bit_cast(bf16x8_t, xf), acc[c][t], 0, 0, 0);
                }
            }
        }
    }
    // D[row = gradient channel kt * 16 + fg * 4 + e][col = input channel c * 16 + fr]
    float* o3 = p.slab3 + (size_t)grp * K * 9 * C;
    float* osc = p.slabsc + (size_t)grp * K * C;
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int tap = t_lo + t;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = kt * 16 + fg * 4 + e;
                if (tap == 9) osc[(size_t)k * C + c * 16 + fr] = acc[c][t][e];
                else o3[((size_t)k * 9 + tap) * C + c * 16 + fr] = acc[c][t][e];
            }
        }
}

template <int C> int wgrad7_groups(int N) { const int ipg = C == 16 ? (N >= 256 ? N / 128 : 1) : (N >= 128 ? N / 64 : 1); return (N + ipg - 1) / ipg; }
template <int C> int wgrad7_ipg(int N) { return C == 16 ? (N >= 256 ? N / 128 : 1) : (N >= 128 ? N / 64 : 1); }

}  // namespace

int clhip_wgrad_reduce_launch(const float* slab, float* dw, int64_t n4, int splits, hipStream_t st);      // conv3.hip

// (N, H, W, C) = the block input; K = 2 C.  CifarResNet-32's two entries: 32 x 32 x 16 and 16 x 16 x 32
bool clhip_wgrad7_supported(int N, int H, int W, int C, int K, int dtype) {
    static const bool off = clhip_cfg("CONV7") != nullptr && atoi(clhip_cfg("CONV7")) == 0;
    static const bool woff = clhip_cfg("WGRAD7") != nullptr && atoi(clhip_cfg("WGRAD7")) == 0;
    if (off || woff || dtype != CLHIP_BF16 || N < 1) return false;
    return (C == 16 && K == 32 && H == 32 && W == 32) || (C == 32 && K == 64 && H == 16 && W == 16);
}
// scratch of the 3x3 layer's partial blocks (which = 0) and of the shortcut's (which = 1)
size_t clhip_wgrad7_ws_bytes(int N, int C, int K, int which) {
    const int groups = C == 16 ? wgrad7_groups<16>(N) : wgrad7_groups<32>(N);
    return (size_t)groups * K * (which == 0 ? 9 : 1) * C * sizeof(float);
}

int clhip_wgrad7_launch(const void* x, const void* dz, const void* dz_sc, float* dw, float* dw_sc, float* ws3, float* ws_sc, int N, int C, hipStream_t st) {
    Wgrad7Params p{static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(dz), static_cast<const bf16_t*>(dz_sc), ws3, ws_sc, N, 1};
    const int K = 2 * C;
    int groups;
    if (C == 16) {
        constexpr int lds = 34 * 34 * 32 + 2 * 256 * 64;
        static bool attr = false;
        if (!attr) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad7_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) { clhip_set_error("wgrad7: cannot reserve %d bytes of LDS", lds); return CLHIP_EHIP; }
            attr = true;
        }
        p.ipg = wgrad7_ipg<16>(N); groups = wgrad7_groups<16>(N);
        hipLaunchKernelGGL(wgrad7_kernel<16>, dim3(groups), dim3(256), lds, st, p);
    } else {
        constexpr int lds = 18 * 18 * 64 + 2 * 64 * 128;
        p.ipg = wgrad7_ipg<32>(N); groups = wgrad7_groups<32>(N);
        hipLaunchKernelGGL(wgrad7_kernel<32>, dim3(groups), dim3(256), lds, st, p);
    }
    CLHIP_LAUNCH_CHECK();
    if (int e = clhip_wgrad_reduce_launch(ws3, dw, (int64_t)K * 9 * C / 4, groups, st)) return e;
    return clhip_wgrad_reduce_launch(ws_sc, dw_sc, (int64_t)K * C / 4, groups, st);
}

// =========================================================================================================================================
// The FORWARD of the same two layers in one launch: z3 = conv3x3/s2/p1(x, W3), zsc = conv1x1/s2(x, Wsc), both with their BatchNorm statistics
// (sum z, sum z^2 of the fp32 accumulators into the fp64 accumulators), from one pass over the block input.  Generic kernel: 9.6 + 5.9 us
// (16 -> 32) and 7.9 + 4.6 us (32 -> 64) at batch 256.  dgrad7's scheme: no LDS -- a 16-pixel output tile, the weight fragments of two
// 16-channel output tiles x ten taps in registers, the input fragments 16-byte global loads (16 channels: two taps per 32-deep MFMA K step, as in
// conv16; 32 channels: one tap per step); out-of-image taps are zero fragments; the shortcut reads tap (1, 1)'s pixel.
namespace {

struct Fwd7Params {
    const bf16_t* x; const bf16_t* w3; const bf16_t* wsc; bf16_t* z3; bf16_t* zsc;
    double* acc3; double* accsc; int rep3, repsc;
    int Ho, Wo, lgHo, lgWo, Mq, ntiles, tpw;
};

template <int C>
__global__ __launch_bounds__(256) void fwd7_kernel(const Fwd7Params p) {
    constexpr int K = 2 * C, NS = C == 16 ? 5 : 9;          // K steps of the 3x3 layer (16 channels: tap pairs)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int k0 = blockIdx.y * 32;                          // this wave's two 16-channel output tiles
    const int cb = C == 16 ? (fg & 1) * 8 : fg * 8;          // the 8 input channels this lane feeds
    const int tsel = fg >> 1;                                // 16 channels: which tap of the pair

    uint4 w[2][NS], ws[2];
    const uint4 zero = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = k0 + j * 16 + fr;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int tap = C == 16 ? 2 * s + tsel : s;
            w[j][s] = tap < 9 ? *reinterpret_cast<const uint4*>(p.w3 + ((size_t)k * 9 + tap) * C + cb) : zero;
        }
        ws[j] = (C == 32 || tsel == 0) ? *reinterpret_cast<const uint4*>(p.wsc + (size_t)k * C + cb) : zero;
    }
    const int Wo = p.Wo, Ho = p.Ho, W = 2 * Wo, H = 2 * Ho;
    float s1[2][2][4], s2[2][2][4];                          // [layer][tile][channel fg*4+e]
#pragma unroll
    for (int l = 0; l < 2; ++l)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) s1[l][j][e] = s2[l][j][e] = 0.f;

    const int t0 = (blockIdx.x * 4 + wave) * p.tpw;
    for (int i = 0; i < p.tpw; ++i) {
        const int tile = t0 + i;
        if (tile >= p.ntiles) break;
        const int q = tile * 16 + fr;
        const bool qv = q < p.Mq;
        const int wo = q & (Wo - 1), ho = (q >> p.lgWo) & (Ho - 1), n = q >> (p.lgWo + p.lgHo);
        const bf16_t* xi = p.x + (size_t)n * H * W * C + cb;
        uint4 xf[NS], xc;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int tap = C == 16 ? 2 * s + tsel : s;
            const int r = tap / 3, sx = tap - 3 * r;
            const int hi = 2 * ho - 1 + r, wi = 2 * wo - 1 + sx;
            const bool ok = qv && tap < 9 && hi >= 0 && wi >= 0 && hi < H && wi < W;
            xf[s] = ok ? *reinterpret_cast<const uint4*>(xi + ((size_t)hi * W + wi) * C) : zero;
        }
        xc = (qv && (C == 32 || tsel == 0)) ? *reinterpret_cast<const uint4*>(xi + ((size_t)(2 * ho) * W + 2 * wo) * C) : zero;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x4 a3 = f32x4{0.f, 0.f, 0.f, 0.f}, as = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NS; ++s) a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w[j][s]), __builtin_bit_cast(bf16x8_t, xf[s]), a3, 0, 0, 0);
            as = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ws[j]), __builtin_bit_cast(bf16x8_t, xc), as, 0, 0, 0);
            // D[row = output channel k0 + j * 16 + fg * 4 + e][col = pixel fr]
            if (qv) {
                const size_t at = (size_t)q * K + k0 + j * 16 + fg * 4;
                *reinterpret_cast<uint2*>(p.z3 + at) = make_uint2(pack_bf16x2(a3[0], a3[1]), pack_bf16x2(a3[2], a3[3]));
                *reinterpret_cast<uint2*>(p.zsc + at) = make_uint2(pack_bf16x2(as[0], as[1]), pack_bf16x2(as[2], as[3]));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s1[0][j][e] += a3[e]; s2[0][j][e] = fmaf(a3[e], a3[e], s2[0][j][e]);
                    s1[1][j][e] += as[e]; s2[1][j][e] = fmaf(as[e], as[e], s2[1][j][e]);
                }
            }
        }
    }
    // per-channel sums over the wave's pixels (16-lane DPP sums), the four waves through LDS, then ONE fp64 atomic per channel and sum per workgroup
    // (replica by workgroup, as the other kernels; a wave-level atomic each was 26 us of contention on the 16 -> 32 layer)
    __shared__ float red[4][2][2][4][8];                     // [wave][layer][tile][fg][sum e, sum-of-squares e]
#pragma unroll
    for (int l = 0; l < 2; ++l)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float sv[8] = {s1[l][j][0], s1[l][j][1], s1[l][j][2], s1[l][j][3], s2[l][j][0], s2[l][j][1], s2[l][j][2], s2[l][j][3]};
            row16_sum_n(sv);
            if (fr == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) red[wave][l][j][fg][e] = sv[e];
            }
        }
    __syncthreads();
    if (tid < 128) {
        const int e = tid & 7, g = (tid >> 3) & 3, j = (tid >> 5) & 1, l = tid >> 6;
        const float v = red[0][l][j][g][e] + red[1][l][j][g][e] + red[2][l][j][g][e] + red[3][l][j][g][e];
        double* base = l == 0 ? p.acc3 + (size_t)(blockIdx.x & (p.rep3 - 1)) * 2 * K : p.accsc + (size_t)(blockIdx.x & (p.repsc - 1)) * 2 * K;
        const int k = k0 + j * 16 + g * 4 + (e & 3);
        atomicAdd(base + (e >> 2) * K + k, (double)v);
    }
}

}  // namespace

bool clhip_fwd7_supported(int N, int H, int W, int C, int K, int dtype) {
    static const bool foff = clhip_cfg("FWD7") != nullptr && atoi(clhip_cfg("FWD7")) == 0;
    return !foff && clhip_dgrad7_supported(N, H, W, C, K, dtype);
}

int clhip_fwd7_launch(const void* x, const void* w3, const void* wsc, void* z3, void* zsc, double* acc3, int rep3, double* accsc, int repsc, int N, int H, int W, int C,
                      hipStream_t st) {
    Fwd7Params p;
    p.x = static_cast<const bf16_t*>(x); p.w3 = static_cast<const bf16_t*>(w3); p.wsc = static_cast<const bf16_t*>(wsc);
    p.z3 = static_cast<bf16_t*>(z3); p.zsc = static_cast<bf16_t*>(zsc); p.acc3 = acc3; p.accsc = accsc; p.rep3 = rep3 > 0 ? rep3 : 1; p.repsc = repsc > 0 ? repsc : 1;
    p.Ho = H / 2; p.Wo = W / 2; p.lgHo = ilog2_7(p.Ho); p.lgWo = ilog2_7(p.Wo);
    p.Mq = N * p.Ho * p.Wo; p.ntiles = (p.Mq + 15) / 16;
    const int ny = 2 * C / 32;
    int tpw = 4;
    while (tpw > 1 && (p.ntiles / tpw / 4) * ny < 256) tpw >>= 1;
    p.tpw = tpw;
    const int waves = (p.ntiles + tpw - 1) / tpw;
    if (C == 16) hipLaunchKernelGGL(fwd7_kernel<16>, dim3((waves + 3) / 4, ny), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(fwd7_kernel<32>, dim3((waves + 3) / 4, ny), dim3(256), 0, st, p);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}
