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


// ------------------------------------------------------------------------------------------------------------ weight gradient
// dw[o][tap][c] += sum_p dz[p][o] * x[p @ tap][c] for the same layers.  The atomic generic kernel took 53 us on ResNet-18's stem -- on
// the caller's stream, at the very end of the backward, where nothing hides it.  Here every wave walks 32-pixel steps on its own: the
// step's gradient rows and its im2col rows (9 taps x 8 padded channels, 16 bytes per tap straight from the neighbour pixel) go into
// the wave's private LDS tiles, both MFMA operands come back through transposing reads (the reduction index is the pixel), and a
// 16 x 16 x 32 MFMA per (16 output channels, 2 taps) accumulates D[row = output channel][col = (tap, channel)].  Workgroup partials
// go to a [workgroup][K][9][Creal] slab that wgrad3_reduce_kernel sums in a fixed order: deterministic, like every other weight
// gradient of the ResNets.
struct StemWParams { const bf16_t* x; const bf16_t* dz; float* slab; int N, H, W, K, M, Creal, nstep; };

__device__ __forceinline__ bf16x8_t tr8s(const char* base, int addr, int second) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + addr));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + addr + second));
    uint2 l = __builtin_bit_cast(uint2, lo), h = __builtin_bit_cast(uint2, hi);
    return __builtin_bit_cast(bf16x8_t, make_uint4(l.x, l.y, h.x, h.y));
}

template <int KT>
__global__ __launch_bounds__(256) void wgrad_stem_kernel(const StemWParams p) {
    constexpr int PZ = KT * 32 + 16, PX = 176;               // LDS pitches: gradient row (16 KT channels), im2col row (10 x 16 B + pad)
    constexpr int WAVE_LDS = 32 * PZ + 32 * PX;
    __shared__ __attribute__((aligned(16))) char smem[4 * WAVE_LDS > KT * 16 * 80 * 4 ? 4 * WAVE_LDS : KT * 16 * 80 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int H = p.H, W = p.W, K = p.K;
    char* zs = smem + wave * WAVE_LDS;
    char* xs = zs + 32 * PZ;
    // the tenth tap slot of every im2col row stays zero
    if (lane < 32) *reinterpret_cast<uint4*>(xs + lane * PX + 144) = make_uint4(0, 0, 0, 0);
    f32x4 acc[KT][5];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[kt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int col = fg * 8 + (fr >> 2), seg = (fr & 3) * 8;
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    // One step = 32 pixels.  The global loads of step s + 1 are issued (into registers) as soon as those of step s have been stored to
    // LDS, and land while step s is read back and multiplied: a wave no longer pays a full L2 / HBM round trip per step with nothing
    // else to do (the serial version: 8 steps x ~2 us of the kernel's 37 us at batch 256).
    uint4 zr[KT], xr[5];
    auto fetch = [&](int s) {
        const int p0 = s * 32;
        // gradient rows: 32 pixels x 2 KT chunks of 16 bytes
#pragma unroll
        for (int i = 0; i < KT; ++i) {
            const int id = lane + 64 * i, px = id / (2 * KT), part = id - px * (2 * KT);
            zr[i] = make_uint4(0, 0, 0, 0);
            if (p0 + px < p.M) zr[i] = *reinterpret_cast<const uint4*>(p.dz + (size_t)(p0 + px) * K + part * 8);
        }
        // im2col rows: 32 pixels x 9 taps
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int id = lane + 64 * i;
            xr[i] = make_uint4(0, 0, 0, 0);
            if (id < 288) {
                const int px = id / 9, tap = id - px * 9, g = p0 + px;
                const int w0 = g % W, h0 = (g / W) % H, dr = tap / 3 - 1, ds = tap % 3 - 1;
                if (g < p.M && (unsigned)(h0 + dr) < (unsigned)H && (unsigned)(w0 + ds) < (unsigned)W)
                    xr[i] = *reinterpret_cast<const uint4*>(p.x + ((size_t)g + dr * W + ds) * 8);
            }
        }
    };
    if (gw < p.nstep) fetch(gw);
    for (int s = gw; s < p.nstep; s += nw) {
#pragma unroll
        for (int i = 0; i < KT; ++i) {
            const int id = lane + 64 * i, px = id / (2 * KT), part = id - px * (2 * KT);
            *reinterpret_cast<uint4*>(zs + px * PZ + part * 16) = zr[i];
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int id = lane + 64 * i;
            if (id < 288) { const int px = id / 9, tap = id - px * 9; *reinterpret_cast<uint4*>(xs + px * PX + tap * 16) = xr[i]; }
        }
        if (s + nw < p.nstep) fetch(s + nw);
        bf16x8_t zf[KT], xf[5];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) zf[kt] = tr8s(zs, col * PZ + kt * 32 + seg, 4 * PZ);
#pragma unroll
        for (int j = 0; j < 5; ++j) xf[j] = tr8s(xs, col * PX + j * 32 + seg, 4 * PX);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int j = 0; j < 5; ++j) acc[kt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zf[kt], xf[j], acc[kt][j], 0, 0, 0);
        __builtin_amdgcn_s_waitcnt(0xC07F);                  // the tiles are overwritten by the next step's stores
    }
    // D[row = out channel 4 fg + e][col = 16 j + fr = (tap 2 j + fr / 8, channel fr % 8)] -> red[o][80], the four waves in a fixed order
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int j = 0; j < 5; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float* q = red + (kt * 16 + fg * 4 + e) * 80 + j * 16 + fr;
                        *q = w == 0 ? acc[kt][j][e] : *q + acc[kt][j][e];
                    }
        }
        __syncthreads();
    }
    float* out = p.slab + (size_t)blockIdx.x * K * 9 * p.Creal;
    for (int i = tid; i < K * 9 * p.Creal; i += 256) {
        const int c = i % p.Creal, t = (i / p.Creal) % 9, o = i / (p.Creal * 9);
        out[i] = red[o * 80 + t * 8 + c];
    }
}

int stem_wgrad_grid(int M) {
    const int nstep = (M + 31) / 32;
    int grid = (nstep + 15) / 16;                            // >= 4 steps per wave
    static const int cap = clhip_cfg("STEM_WGRAD_GRID") ? atoi(clhip_cfg("STEM_WGRAD_GRID")) : 256;
    if (grid > cap) grid = cap;
    if (grid < 1) grid = 1;
    return grid;
}

}  // namespace

int clhip_wgrad_reduce_launch(const float* slab, float* dw, int64_t n4, int splits, hipStream_t st);      // conv3.hip

bool clhip_stem_wgrad_supported(int N, int H, int W, int C, int Creal, int K, int ksize, int stride, int pad, int dtype) {
    static const bool off = clhip_cfg("NO_STEM") != nullptr;
    return !off && dtype == CLHIP_BF16 && ksize == 3 && stride == 1 && pad == 1 && C == 8 && Creal >= 1 && Creal <= 8 && (K == 16 || K == 32 || K == 64) &&
           (K * 9 * Creal) % 4 == 0 && N >= 1 && (long long)N * H * W >= 2048;
}

size_t clhip_stem_wgrad_ws_bytes(int N, int H, int W, int Creal, int K) { return (size_t)stem_wgrad_grid(N * H * W) * K * 9 * Creal * sizeof(float); }

int clhip_stem_wgrad_launch(const void* x, const void* dz, float* dw, float* ws, int N, int H, int W, int Creal, int K, hipStream_t st) {
    StemWParams p{static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(dz), ws, N, H, W, K, N * H * W, Creal, (N * H * W + 31) / 32};
    const int grid = stem_wgrad_grid(p.M);
    if (K == 16) hipLaunchKernelGGL(wgrad_stem_kernel<1>, dim3(grid), dim3(256), 0, st, p);
    else if (K == 32) hipLaunchKernelGGL(wgrad_stem_kernel<2>, dim3(grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(wgrad_stem_kernel<4>, dim3(grid), dim3(256), 0, st, p);
    CLHIP_LAUNCH_CHECK();
    return clhip_wgrad_reduce_launch(ws, dw, (int64_t)K * 9 * Creal / 4, grid, st);
}

bool clhip_stem_supported(int N, int H, int W, int C, int K, int ksize, int stride, int pad, int dtype) {
    static const bool off = clhip_cfg("NO_STEM") != nullptr;
    return !off && dtype == CLHIP_BF16 && ksize == 3 && stride == 1 && pad == 1 && C == 8 && (K == 16 || K == 32 || K == 64) && N >= 1 && H >= 1 && W >= 1;
}

int clhip_stem_launch(const void* x, const void* w, void* z, double* acc, int rep, int N, int H, int W, int K, hipStream_t st) {
    StemParams p{static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(w), static_cast<bf16_t*>(z), acc, rep > 0 ? rep : 1, N, H, W, K, N * H * W};
    const int ntile = (p.M + 63) / 64;
    int grid = (ntile + 3) / 4;
    static const int cap = clhip_cfg("STEM_GRID") ? atoi(clhip_cfg("STEM_GRID")) : 512;      // two tiles per wave at batch 256: 17.9 -> 15.9 us (64 features)
    if (grid > cap) grid = cap;
    if (K == 16) hipLaunchKernelGGL(conv_stem_kernel<1>, dim3(grid), dim3(256), 0, st, p);
    else if (K == 32) hipLaunchKernelGGL(conv_stem_kernel<2>, dim3(grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(conv_stem_kernel<4>, dim3(grid), dim3(256), 0, st, p);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}
