// wgrad4.hip -- weight gradient of the 3x3 / stride 1 / pad 1 layers, bf16, gfx950:  dw[o][tap][c] += sum_p dz[p][o] * x[p @ tap][c].
//
// conv_wgrad3_kernel (conv3.hip) keeps a zero-padded input patch in LDS and reads both MFMA operands with transposing LDS reads; what
// the round-2 measurements said about it (tools/ubench/wgrad_bench, profiles/r02_wgrad4_notes.md): 1.5 us per 64-pixel step even on
// an idle chip -- the step's global loads are issued at its start, parked in registers and stored to LDS at its end, so every step
// pays one L2 / HBM round trip, with one workgroup per CU and nothing else to hide it -- and 0.61 fragment reads per MFMA (a wave owns
// 32 x 16 x 9 outputs), which makes LDS the next limit.  This generation:
//
//  * both operands are streamed by LDS-DMA through buffer descriptors (`buffer_load ... lds`, 16 bytes per lane, no staging
//    registers) into a 3-stage ring TWO 128-pixel steps ahead; waits are counted (s_waitcnt vmcnt(6)), the barrier is the raw
//    s_barrier.  Zero padding costs nothing: pad slots, halo rows outside the image and pixels behind the tensor's end get an offset
//    the hardware range check rejects, and the DMA writes zeros there.
//  * the patch has ONE pad column per row (it is the right neighbour of the row's last pixel and the left neighbour of the next
//    row's first) and one zero row between images, so a 128-pixel step stages 171-206 patch pixels (wgrad3: 136 per 64).
//  * a wave owns 64 (out) x 16 (in) x 9 outputs = 36 accumulator tiles, 144 registers: 4 + 9 fragment reads per 36 MFMAs.  Waves
//    0-3 and 4-7 split the step's four 32-pixel MFMA K-steps; the two halves are summed through LDS at the end (each keeps two
//    of the four out-channel tiles, so all eight waves store).
//  * A = the input fragment, B = the gradient fragment: the accumulator then holds four consecutive INPUT channels per lane and the
//    partial block is written with 16-byte stores.
//  * workgroups that share a pixel range (the (C/64) x (K/64) tiles of one split) are placed on one XCD, so the second read of a
//    tensor is an L2 hit.
//
// The partial blocks go to the same [split][K][9][C] fp32 workspace as wgrad3's and are summed in a fixed order by
// wgrad3_reduce_kernel (bitwise reproducible weight gradients).  Replaces the weight-gradient half of nn.Conv2d's backward for the
// reference ResNets' 3x3 stride-1 layers (core/model/backbone/resnet.py:17-24, 295-298).
#include <stdlib.h>
#include <type_traits>

#include "common.h"

int clhip_wgrad_reduce_launch(const float* slab, float* dw, int64_t n4, int splits, hipStream_t st);      // conv3.hip

namespace {

typedef __attribute__((address_space(3))) void lvoid_t;

struct Wgrad4Params {
    const bf16_t* x;     // [N,H,W,C]
    const bf16_t* dz;    // [N,Ho,Wo,K]
    float* slab;         // [splits][K][taps][C]
    int N, H, C, K, M;   // M = N*Ho*Wo output pixels
    int Ho;
    int R, nimg;         // a step of SP output pixels = nimg images x R output rows x Wo columns (nimg == 1: R rows of one image; nimg > 1: whole images)
    int npatch;          // patch pixels: 1 + rows * (W + 1)  (3x3); SP (1x1)
    int steps_per_split, total_steps;
    int tiles_c, tiles, splits, xcd_map;
    unsigned long long* trace;   // CLHIP_ABLATION builds: s_memtime stamps of waves 0 and 4 of workgroup 0 ([2][64])
};

unsigned long long* g_trace_w4 = nullptr;
int g_target_w4 = 0;                      // clhip_wgrad4_config(): workgroup target of the next launches (0: the in-step default)
#ifdef CLHIP_ABLATION
#define STAMP4() do { if (p.trace && blockIdx.x == 0 && lane == 0 && (wave & 3) == 0 && nstamp < 64) p.trace[(wave >> 2) * 64 + nstamp++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP4() do { } while (0)
#endif

constexpr int P4 = 144;                     // LDS bytes per pixel: 64 channels + one pad slot (4 consecutive pixels fall in disjoint bank ranges)
constexpr int OOB = 0x40000000;

// Step geometry of a (stride S, KS x KS) layer.  3x3 / stride 1: 128 output pixels, a patch of up to 213 input pixels.  3x3 /
// stride 2: 64 output pixels -- their patch spans 2 R + 1 input rows (every input pixel is used: odd and even rows / columns serve
// different taps), up to 334 pixels.  1x1 / stride 2: 128 output pixels and exactly their 128 input pixels (each DMA lane has its
// own source address, so only x[2 ho, 2 wo] is fetched).
template <int S, int KS> struct Geo {
    static constexpr int SP = (KS == 3 && S == 2) ? 64 : 128;              // output pixels per step
    static constexpr int NINST = KS == 1 ? 40 : (S == 2 ? 56 : 48);        // DMA instructions per stage (1 KB each)
    static constexpr int WINST = NINST / 8;                                 // ... per wave
    static constexpr int ZINST = SP * 9 / 64;                               // ... of the gradient tile
    static constexpr int STAGE = NINST * 1024;
    static constexpr int ZBYTES = SP * P4;                                  // the patch follows the gradient tile
    static constexpr int NT = KS * KS;                                      // taps
    static constexpr int UK = SP / 64;                                      // 32-pixel MFMA K-steps per wave and step
    static constexpr int XCH = 8 * 9 * 64 * 16;                             // one exchange round of the epilogue (73728 bytes)
    static constexpr int LDS = 2 * STAGE > XCH ? 2 * STAGE : XCH;
};

__device__ __forceinline__ void wg_barrier4() {
    __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): this wave's LDS reads of the stage that is about to be overwritten
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// two transposing 8-byte reads -> the 8 reduction elements (pixels) of one MFMA operand row; `second` = byte distance of pixels k+4..k+7
__device__ __forceinline__ bf16x8_t tr8x(const char* base, int addr, int second) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + addr));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + addr + second));
    uint2 l = __builtin_bit_cast(uint2, lo), h = __builtin_bit_cast(uint2, hi);
    return __builtin_bit_cast(bf16x8_t, make_uint4(l.x, l.y, h.x, h.y));
}

// The LDS-DMA is issued from inline asm ON PURPOSE.  hipcc (ROCm 7.2) knows that a `buffer_load ... lds` builtin writes LDS and puts
// `s_waitcnt vmcnt(0)` in front of the first ds_read_tr intrinsic that follows one (the intrinsic carries an LDS memory operand which
// every DMA in flight may alias) -- the ring would be drained every step.  Hidden in asm, the DMA is invisible to that bookkeeping;
// its completion is waited for by hand (dma_wait<N>: loads return in order) and published by the workgroup barrier.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int IMM>
__device__ __forceinline__ void dma16(u32x4 rsrc, unsigned lds_addr, int voffset) {
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr + IMM), "v"(voffset), "s"(rsrc) : "memory");      // m0 is a reserved register: hipcc re-materialises it before each of its own uses
}
template <int N> __device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int W, int S, int KS>
__global__ __launch_bounds__(512) void conv_wgrad4_kernel(const Wgrad4Params p) {
    using G = Geo<S, KS>;
    constexpr int PW = W + 1, Wo = W / S, SP = G::SP, WINST = G::WINST, ZINST = G::ZINST, STAGE = G::STAGE, ZBYTES = G::ZBYTES, NT = G::NT, UK = G::UK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave >> 2, ct = wave & 3;               // K-step half, 16-channel input column
    int nstamp = 0; (void)nstamp;
    STAMP4();
    const int fr = lane & 15, fg = lane >> 4;
    const int H = p.H, Ho = p.Ho, R = p.R;

    // ---- workgroup -> (tile, split): the tiles of one split sit on one XCD (block b runs on XCD b % 8)
    int tile, split;
    if (p.xcd_map) { const int b = blockIdx.x, xcd = b & 7, j = b >> 3; tile = j % p.tiles; split = (j / p.tiles) * 8 + xcd; }
    else { tile = blockIdx.x % p.tiles; split = blockIdx.x / p.tiles; }
    const int c0 = (tile % p.tiles_c) * 64, o0 = (tile / p.tiles_c) * 64;
    const int s_beg = split * p.steps_per_split;
    const int s_end = min(p.total_steps, s_beg + p.steps_per_split);
    const int nst = s_end - s_beg;

    // ---- fragment addresses.  Reduction element k of a step is output pixel (img, row, col) = (k / (R*Wo), (k % (R*Wo)) / Wo, k % Wo);
    //      lane (fr, fg) feeds k = ks*32 + fg*8 + {0..7}; a transposing read fetches 4 consecutive k of one output row (their input
    //      pixels are S apart: every lane of a transposing read has its own address).
    int zaddr[UK], xaddr[UK];
#pragma unroll
    for (int u = 0; u < UK; ++u) {
        const int k = (kg * UK + u) * 32 + fg * 8 + (fr >> 2);
        zaddr[u] = k * P4 + (fr & 3) * 8;
        const int per = R * Wo;
        const int img = k / per, rem = k - img * per, rr = rem / Wo, cc = rem - rr * Wo;
        if (KS == 3) {
            // patch row of input row S*ho - 1 (tap row 0); patch row 0 = the input row above the step's first (or an image's zero row)
            const int j = p.nimg == 1 ? S * rr : img * (H + 1) + S * rr;
            // biased to tap (0, 0) = input pixel (S*ho - 1, S*wo - 1): the nine taps are the non-negative immediates (r * PW + s) * P4
            xaddr[u] = ZBYTES + (1 + j * PW + S * cc - 1) * P4 + (ct * 16 + (fr & 3) * 4) * 2;
        } else {
            xaddr[u] = ZBYTES + k * P4 + (ct * 16 + (fr & 3) * 4) * 2;
        }
    }
    constexpr int xsecond = KS == 1 ? 4 * P4 : (Wo >= 8 ? 4 * S : S * PW) * P4;       // pixels k+4..k+7: same output row, or (4-wide rows) the next one

    // ---- DMA lanes.  Instruction I = wave * WINST + i of a stage fills LDS bytes [I * 1024, +1024): slot n = I * 64 + lane is
    //      (pixel n / 9, 16-byte column n % 9); instructions [0, ZINST) carry the gradient tile, the rest the patch.
    const unsigned lds0 = (unsigned)(uintptr_t)smem;        // LDS byte address of the ring (low half of the flat address)
    const uintptr_t tz = reinterpret_cast<uintptr_t>(p.dz), tx = reinterpret_cast<uintptr_t>(p.x);
    u32x4 rsz, rsx;                                          // raw buffer descriptors: base, stride 0, bytes, DATA_FORMAT = 32 bits
    rsz.x = __builtin_amdgcn_readfirstlane((unsigned)tz); rsz.y = __builtin_amdgcn_readfirstlane((unsigned)(tz >> 32) & 0xffffu);
    rsz.z = __builtin_amdgcn_readfirstlane((unsigned)(p.M * p.K * 2)); rsz.w = 0x00020000u;
    rsx.x = __builtin_amdgcn_readfirstlane((unsigned)tx); rsx.y = __builtin_amdgcn_readfirstlane((unsigned)(tx >> 32) & 0xffffu);
    rsx.z = __builtin_amdgcn_readfirstlane((unsigned)(p.N * H * W * p.C * 2)); rsx.w = 0x00020000u;
    int prel[WINST];
    unsigned topm = 0, botm = 0, zmask = 0;
#pragma unroll
    for (int i = 0; i < WINST; ++i) {
        const int I = wave * WINST + i;
        const int n = I * 64 + lane;
        if (I < ZINST) {
            zmask |= 1u << i;
            const int q = n / 9, sub = n - q * 9;
            prel[i] = sub < 8 ? q * p.K * 2 + sub * 16 : OOB;
        } else {
            const int n2 = n - ZINST * 64;
            const int q = n2 / 9, sub = n2 - q * 9;
            int v = OOB;
            if (KS == 3) {
                if (q >= 1 && q < p.npatch && sub < 8) {
                    const int j = (q - 1) / PW, w = (q - 1) - j * PW;
                    if (w < W) {
                        if (p.nimg == 1) {
                            v = ((j - 1) * W + w) * p.C * 2 + sub * 16;           // relative to input row S*ho0 of the step
                            if (j == 0) topm |= 1u << i;
                            if (S == 1 && j == R + 1) botm |= 1u << i;
                        } else {
                            const int img = j / (H + 1), jj = j - img * (H + 1);
                            if (jj != 0) v = ((img * H + jj - 1) * W + w) * p.C * 2 + sub * 16;
                        }
                    }
                }
            } else if (q < SP && sub < 8) {
                const int per = R * Wo;
                const int img = q / per, rem = q - img * per, rr = rem / Wo, cc = rem - rr * Wo;
                v = ((img * H + S * rr) * W + S * cc) * p.C * 2 + sub * 16;
            }
            prel[i] = v;
        }
    }
    zmask = __builtin_amdgcn_readfirstlane(zmask);
    auto dma = [&](int s, int stage) {
        const int p0 = s * SP;                                   // first output pixel of the step
        const int n0 = p0 / (Ho * Wo), ho0 = (p0 / Wo) % Ho;
        const int zbase = (p0 * p.K + o0) * 2;
        const int xbase = (((n0 * H + S * ho0) * W) * p.C + c0) * 2;
        const bool top_ok = p.nimg != 1 || ho0 > 0, bot_ok = p.nimg != 1 || ho0 + R < Ho;
        const unsigned l = lds0 + stage * STAGE + wave * (WINST * 1024);
#if defined(__HIP_DEVICE_COMPILE__)
        auto one = [&](auto IC) {
            constexpr int i = decltype(IC)::value;
            if constexpr (i < WINST) {
                const bool isz = (zmask >> i) & 1u;
                int v = prel[i] + (isz ? zbase : xbase);
                if (((topm >> i) & 1u) && !top_ok) v = OOB;
                if (((botm >> i) & 1u) && !bot_ok) v = OOB;
                if (isz) dma16<i * 1024>(rsz, l, v); else dma16<i * 1024>(rsx, l, v);
            }
        };
        one(std::integral_constant<int, 0>{}); one(std::integral_constant<int, 1>{}); one(std::integral_constant<int, 2>{});
        one(std::integral_constant<int, 3>{}); one(std::integral_constant<int, 4>{}); one(std::integral_constant<int, 5>{});
        one(std::integral_constant<int, 6>{});
#else
        (void)l; (void)zbase; (void)xbase; (void)top_ok; (void)bot_ok;
#endif
    };

    f32x4 acc[4][NT];                                        // [out-channel tile][tap]: D[row = in channel 4 fg + e][col = out channel fr]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    STAMP4();
    if (nst > 0) dma(s_beg, 0);
    STAMP4();
    int stage = 0;
    for (int i = 0; i < nst; ++i) {
        dma_wait<0>();                                               // this wave's part of step i has landed
        STAMP4();
        wg_barrier4();                                               // ... everybody's; and the other stage is no longer read
        STAMP4();
        if (i + 1 < nst) dma(s_beg + i + 1, stage ^ 1);              // one step (1-1.5 us of MFMAs) ahead
        // Round-5 measurements of this loop (profiles/r05_conv_notes.md section 5; 160 workgroups, cycles per 128-pixel step of workgroup 0):
        // as it stands 4 450 (2 304 of matrix pipe); DMA alone 2 100 (= 7.7 TB/s over the 160 CUs: the memory side is at its roofline);
        // MFMAs + fragment reads alone 3 100; conflict-free fragment addresses: no change; fragment reads two slots ahead of their
        // MFMAs (hipcc keeps ONE input fragment: read -> wait -> 4 MFMAs): 3 900 per step and the SAME 42 us per launch, 0.3 % slower
        // inside the step; the six DMA instructions spread over the step's MFMAs instead of behind the barrier: 52 us (they land late).
        // With no MFMAs at all the launch + its reduce still take 31-35 of the 42 us: prologue 3.3, 13 DMA steps 13, exchange +
        // partial block 5.4, reduce launch 7.4.
        const char* sb = smem + stage * STAGE;
#pragma unroll
        for (int u = 0; u < UK; ++u) {
            bf16x8_t zf[4];
#pragma unroll
            for (int oi = 0; oi < 4; ++oi) zf[oi] = tr8x(sb, zaddr[u] + oi * 32, 4 * P4);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int r = t / 3, s2 = t - 3 * r;
                const bf16x8_t xf = tr8x(sb, xaddr[u] + (r * PW + s2) * P4, xsecond);
#pragma unroll
                for (int oi = 0; oi < 4; ++oi) acc[oi][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, zf[oi], acc[oi][t], 0, 0, 0);
            }
        }
        stage ^= 1;
    }

    // ---- the two K-step halves: waves 0-3 keep out-channel tiles 0-1 and hand over 2-3, waves 4-7 the other way round
    STAMP4();
    wg_barrier4();
    STAMP4();
    f32x4* ex = reinterpret_cast<f32x4*>(smem);              // [wave][NT][64], one out-channel tile per round
    float* out = p.slab + (size_t)split * p.K * NT * p.C;
    const int c = c0 + ct * 16 + fg * 4;
    const int partner = wave ^ 4;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (kg == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t) ex[(wave * NT + t) * 64 + lane] = acc[2 + h][t];
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) ex[(wave * NT + t) * 64 + lane] = acc[h][t];
        }
        wg_barrier4();
        if (kg == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4 v = acc[h][t] + ex[(partner * NT + t) * 64 + lane];
                *reinterpret_cast<f32x4*>(out + ((size_t)(o0 + h * 16 + fr) * NT + t) * p.C + c) = v;
            }
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4 v = acc[2 + h][t] + ex[(partner * NT + t) * 64 + lane];
                *reinterpret_cast<f32x4*>(out + ((size_t)(o0 + (2 + h) * 16 + fr) * NT + t) * p.C + c) = v;
            }
        }
        if (h == 0) wg_barrier4();
    }
    STAMP4();
#ifdef CLHIP_ABLATION
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    STAMP4();
}

template <int S, int KS>
bool geometry_sk(int N, int H, int W, int C, int K, Wgrad4Params& p) {
    using G = Geo<S, KS>;
    constexpr int SP = G::SP;
    if (H % S || W % S) return false;
    const int Ho = H / S, Wo = W / S;
    p.N = N; p.H = H; p.C = C; p.K = K; p.Ho = Ho; p.M = N * Ho * Wo;
    const int hwo = Ho * Wo;
    int rows;
    if (hwo >= SP) {
        if (SP % Wo || Ho % (SP / Wo)) return false;
        p.nimg = 1; p.R = SP / Wo; rows = S == 1 ? p.R + 2 : 2 * p.R + 1;
    } else {
        if (SP % hwo) return false;
        p.nimg = SP / hwo; p.R = Ho; rows = p.nimg * (H + 1) + 1;
    }
    p.npatch = KS == 3 ? 1 + rows * (W + 1) : SP;
    if ((SP + p.npatch) * 9 > G::NINST * 64) return false;
    p.tiles_c = C / 64; p.tiles = (C / 64) * (K / 64);
    p.total_steps = (p.M + SP - 1) / SP;
    // ~160 workgroups (128 until round 4), not 256: inside a training step this kernel runs on the weight-gradient stream beside the dgrad / BatchNorm chain
    // of the caller's stream, which is the critical path -- a launch that fills every CU (one workgroup each: 96 KB of LDS, 196 VGPRs)
    // costs the step more than its own 2 us (ResNet-18, batch 256: 2.66 ms per step at 256, 2.55 at 128-160; re-swept after the BatchNorm
    // kernels got shorter: 2.39 at 256, 2.28 at 160, 2.23 at 128-136, 2.26 at 104-112; profiles/r02_wgrad4_notes.md)
    // (round 4, with the normal-priority weight-gradient stream and the queue cap: 128: 2.005 / 2.049, 160: 1.996 / 2.037, 192: 1.998 / 2.048, 224: - / 2.114 on two boxes, two or three
    //  alternating runs each -> 160)
    static const int target_env = clhip_cfg("WGRAD_TARGET") ? atoi(clhip_cfg("WGRAD_TARGET")) : 160;
    const int target = g_target_w4 > 0 ? g_target_w4 : target_env;
    static const int min_steps = clhip_cfg("WGRAD4_MIN_STEPS") ? atoi(clhip_cfg("WGRAD4_MIN_STEPS")) : 2;
    int splits = (target + p.tiles - 1) / p.tiles;
    const int max_splits = (p.total_steps + min_steps - 1) / min_steps;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    p.steps_per_split = (p.total_steps + splits - 1) / splits;
    p.splits = (p.total_steps + p.steps_per_split - 1) / p.steps_per_split;
    p.xcd_map = (p.splits % 8 == 0) ? 1 : 0;
    return true;
}

bool geometry4(int N, int H, int W, int C, int K, int ksize, int stride, Wgrad4Params& p) {
    if (ksize == 3 && stride == 1) return geometry_sk<1, 3>(N, H, W, C, K, p);
    if (ksize == 3 && stride == 2) return geometry_sk<2, 3>(N, H, W, C, K, p);
    if (ksize == 1 && stride == 2) return geometry_sk<2, 1>(N, H, W, C, K, p);
    return false;
}

template <int W, int S, int KS>
int launch4(const Wgrad4Params& p, hipStream_t st) {
    constexpr int lds = Geo<S, KS>::LDS;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad4_kernel<W, S, KS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            clhip_set_error("wgrad4: cannot reserve %d bytes of LDS", lds);
            return CLHIP_EHIP;
        }
        attr = true;
    }
    hipLaunchKernelGGL((conv_wgrad4_kernel<W, S, KS>), dim3(p.tiles * p.splits), dim3(512), lds, st, p);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

}  // namespace

bool clhip_wgrad4_supported(int N, int H, int W, int C, int Creal, int K, int ksize, int stride, int pad, int dtype) {
    static const bool off = clhip_cfg("WGRAD4") != nullptr && atoi(clhip_cfg("WGRAD4")) == 0;
    static const bool s1_only = clhip_cfg("WGRAD4") != nullptr && atoi(clhip_cfg("WGRAD4")) == 2;       // A/B switch: the stride-1 layers only
    if (off) return false;
    if (!(dtype == CLHIP_BF16 && C % 64 == 0 && K % 64 == 0 && Creal == C && N >= 1 && H >= 1)) return false;
    if (!((ksize == 3 && pad == 1 && (stride == 1 || stride == 2)) || (ksize == 1 && pad == 0 && stride == 2))) return false;
    if (s1_only && stride != 1) return false;
    if (stride == 1 ? !(W == 4 || W == 8 || W == 16 || W == 32) : !(W == 8 || W == 16 || W == 32)) return false;
    if ((long long)N * H * W * C * 2 >= (1ll << 30) || (long long)N * (H / stride) * (W / stride) * K * 2 >= (1ll << 30)) return false;      // descriptor offsets + the out-of-range marker stay below 2^31
    Wgrad4Params p;
    if (!geometry4(N, H, W, C, K, ksize, stride, p)) return false;
    // small stride-2 problems stay on the one-launch atomic kernel (5 x 32 x 32, 20 steps: 8.7 us there, 14.4 us here with the reduce
    // launch; 256 x 8 x 8 1x1, 32 steps: 16.6 us there, 12.7 us here)
    static const int min_total = clhip_cfg("WGRAD4_MIN_TOTAL") ? atoi(clhip_cfg("WGRAD4_MIN_TOTAL")) : 32;
    return stride == 1 || p.total_steps >= min_total;
}

size_t clhip_wgrad4_ws_bytes(int N, int H, int W, int C, int K, int ksize, int stride) {
    Wgrad4Params p;
    if (!geometry4(N, H, W, C, K, ksize, stride, p)) return 0;
    return (size_t)p.splits * K * ksize * ksize * C * sizeof(float);
}

int clhip_wgrad4_launch(const void* x, const void* dz, float* dw, float* ws, int N, int H, int W, int C, int K, int ksize, int stride, hipStream_t st) {
    Wgrad4Params p;
    if (!geometry4(N, H, W, C, K, ksize, stride, p) || ws == nullptr) { clhip_set_error("wgrad4: unsupported geometry or no workspace"); return CLHIP_EINVAL; }
    p.x = static_cast<const bf16_t*>(x); p.dz = static_cast<const bf16_t*>(dz); p.slab = ws; p.trace = g_trace_w4;
    int rc = CLHIP_EINVAL;
    if (ksize == 3 && stride == 1) {
        switch (W) {
            case 4: rc = launch4<4, 1, 3>(p, st); break;
            case 8: rc = launch4<8, 1, 3>(p, st); break;
            case 16: rc = launch4<16, 1, 3>(p, st); break;
            case 32: rc = launch4<32, 1, 3>(p, st); break;
        }
    } else if (ksize == 3) {
        switch (W) {
            case 8: rc = launch4<8, 2, 3>(p, st); break;
            case 16: rc = launch4<16, 2, 3>(p, st); break;
            case 32: rc = launch4<32, 2, 3>(p, st); break;
        }
    } else {
        switch (W) {
            case 8: rc = launch4<8, 2, 1>(p, st); break;
            case 16: rc = launch4<16, 2, 1>(p, st); break;
            case 32: rc = launch4<32, 2, 1>(p, st); break;
        }
    }
    if (rc != CLHIP_OK) return rc;
    return clhip_wgrad_reduce_launch(ws, dw, (int64_t)K * ksize * ksize * C / 4, p.splits, st);
}

// phase stamps of workgroup 0 (ablation build only; tools/ubench/wgrad_bench trace)
extern "C" void clhip_wgrad4_config(int target_workgroups) { g_target_w4 = target_workgroups > 0 ? target_workgroups : 0; }
void clhip_wgrad4_set_trace(unsigned long long* dev_buf) { g_trace_w4 = dev_buf; }
