// conv5.hip -- WEIGHT-STATIONARY 3x3 / stride 1 / pad 1 convolution (forward and dgrad) for the 64 -> 64-channel layers, bf16, gfx950.
//
// What round 2 measured on conv4.hip (profiles/r02_conv4_notes.md, r02_gemm5_notes.md): a wave completes one ds_read_b128 per ~30
// cycles, an MFMA 32x32x16 takes 32, and a 64 x 64 wave tile needs one fragment read per MFMA -- half reader, half multiplier by
// construction, plus a weight ring with two barriers per tap.  For 64 -> 64 channels the whole filter bank is 64 x 9 x 64 bf16 =
// 73.7 KB = 288 registers per lane in MFMA A-operand layout: it fits the 512-entry unified register file of a wave that has a SIMD
// to itself.  So here
//   * every wave loads ALL weights once (72 sixteen-byte loads per lane out of L2) and keeps them for the life of the workgroup:
//     no weight ring, no weight DMA, no per-tap barrier -- the only LDS reads left are the pixel fragments, 0.5 per MFMA;
//   * a workgroup is 4 waves (one per SIMD, launch bound 1 workgroup per CU), a tile is 256 pixels x 64 channels, a wave owns
//     64 pixels x 64 channels (2 x 2 MFMA tiles, 64 accumulator registers) and runs the 9 taps x 4 K-steps = 144 MFMAs of a tile as
//     straight-line code with the fragment reads of step t+1 in flight under the MFMAs of step t;
//   * the input patch  pixels [m0 - W - 1, m0 + 256 + W + 1) x 64 channels  is staged by LDS-DMA through a buffer descriptor exactly
//     as in conv4.hip (out-of-tensor pixels and pad slots zero-filled by the range check, 144-byte pixel pitch, masked taps
//     redirected to a zero area at the same bank offset), double-buffered: the patch of the workgroup's next tile lands while the
//     current one is multiplied; ONE workgroup barrier per tile;
//   * the epilogue is conv4's: v_permlane32_swap pairs the half-waves into 16-byte stores straight from the accumulators, BatchNorm
//     sums from the fp32 values into the fp64 accumulators.
// MODE 0 = forward, MODE 1 = dgrad (mirrored taps, the dgrad weight copy [C][9][K] has the forward copy's layout).  Replaces
// nn.Conv2d forward / input gradient of the 64-channel 3x3 layers (core/model/backbone/resnet.py:17-24, 295-298: ResNet-18 layer1).
#include <stdlib.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void lvoid_t;

struct Conv5Params {
    const bf16_t* src;   // [N,H,W,64]
    const bf16_t* wt;    // [64][9][64]
    bf16_t* dst;         // [N,H,W,64]
    double* stat_acc;    // forward: [stat_rep][2][64] fp64 accumulators, or nullptr
    int stat_rep;
    int H, W, M, accumulate;
    int wshift, hshift;
    int np;              // patch pixels = 256 + 2W + 2
    int patch_bytes;
    int zoff;
    int n_tiles;
    LazyIn in;           // XF != 0 (forward only): src is the producer's pre-BatchNorm output, see common.h
    int in_debug;        // WT_DEBUG (timing experiments, results invalid): 1 no activation / mask stores, 2 no transform at all
};

int g_enable5 = -1, g_min_tiles5 = -1;      // clhip_config("CONV5" / "CONV5_MIN_TILES"): take effect at once (tests and tools/ubench flip them between launches)

constexpr int C5 = 64;                 // channels in = channels out
constexpr int PP5 = C5 * 2 + 16;       // patch pitch: 9 sixteen-byte slots, the last one a pad (odd slot count: conflict-free at any tap shift)
constexpr int SPP5 = PP5 / 16;
constexpr int BM5 = 256;

template <int MODE>
__device__ __forceinline__ unsigned tap_mask5(int g, const Conv5Params& p) {
    int w, h;
    if (p.wshift >= 0 && p.hshift >= 0) { w = g & (p.W - 1); h = (g >> p.wshift) & (p.H - 1); }
    else { w = g % p.W; h = (g / p.W) % p.H; }
    constexpr unsigned UP = MODE == 0 ? 0x007u : 0x1c0u, DOWN = MODE == 0 ? 0x1c0u : 0x007u;
    constexpr unsigned LEFT = MODE == 0 ? 0x049u : 0x124u, RIGHT = MODE == 0 ? 0x124u : 0x049u;
    unsigned m = 0x1ffu;
    if (h == 0) m &= ~UP;
    if (h == p.H - 1) m &= ~DOWN;
    if (w == 0) m &= ~LEFT;
    if (w == p.W - 1) m &= ~RIGHT;
    return m;
}

__device__ __forceinline__ void wait_vm0() { __builtin_amdgcn_s_waitcnt(0x0070 | 0xF00); }          // vmcnt(0), lgkmcnt / expcnt untouched
__device__ __forceinline__ void wait_lds5() { __builtin_amdgcn_s_waitcnt(0xC07F); }                // lgkmcnt(0)

// XF: 0 = the source is an activation; 1 = lazy input relu(bn(z')); 2 = lazy input relu(bn(z') + r) with the packed ReLU mask written as well
// (common.h LazyIn; both forms write the activation for the pixels the workgroup owns).  LDS with XF: | patch 0 | patch 1 | statistics |
// coefficient table [2][64] | fp64 scratch of the coefficient prologue / (XF == 2) the landed patch of r |
template <int MODE, int PINST, int XF = 0>
__global__ __launch_bounds__(256, 1) void conv5_kernel(const Conv5Params p) {
    static_assert(XF == 0 || MODE == 0, "lazy inputs exist in the forward only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int W = p.W, halo = W + 1;

    // ---- the filter bank -> registers, MFMA A-operand layout: row o = j*32 + l31, K-step ks covers channels ks*16 + kh*8 .. +8.
    //      Read straight from global memory a fragment load touches 32 cache lines (row stride 1152 bytes): 72 such loads per wave made
    //      the prologue 8-9 us (first version, profiles/r03_conv5_notes.md).  The rows go through LDS instead: 32 rows (37 KB) at a
    //      time are DMA'd lane-linearly -- fully coalesced -- into patch buffer 0 at a row pitch of 73 sixteen-byte slots (odd:
    //      16 consecutive rows start in 16 different bank groups), and every wave picks its fragments with 36 conflict-free
    //      ds_read_b128.  The first tile's patch lands in buffer 1 meanwhile.
    bf16x8_t wr[2][9][4];
    constexpr int WPITCH = 73 * 16;                              // bytes per staged filter row (72 data slots + 1 pad)
    constexpr int WINST5 = (32 * 73 + 63) / 64;                  // 37 DMA instructions per 32 rows
    constexpr int WPW = (WINST5 + 3) / 4;                        // ... per wave (10)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.wt), 0, C5 * 9 * C5 * 2, 0x00020000);
    auto wdma = [&](int j) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            const int I = wv * WPW + i;
            const int n = I * 64 + lane;
            const int row = n / 73, sub = n - row * 73;
            const int off = (row < 32 && sub < 72) ? ((j * 32 + row) * 72 + sub) * 16 : 0x40000000;
            if (I < WINST5) __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lvoid_t*)(smem + I * 1024), 16, off, 0, 0, 0);
        }
#else
        (void)j;
#endif
    };
    auto wread = [&](int j) {
        const char* wl = smem + l31 * WPITCH + kh * 16;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wr[j][t][ks] = *reinterpret_cast<const bf16x8_t*>(wl + t * 128 + ks * 32);
    };

    // ---- LDS map: two patch buffers | statistics scratch (two parities)
    char* patch = smem;
    float* red0 = reinterpret_cast<float*>(smem + 2 * p.patch_bytes);         // [2 parities][8 half-wave rows][2][64]

    // ---- patch DMA lanes (see conv4.hip): slot n of a buffer = (pixel n / 9, 16-byte column n % 9); pad slots, the zero area and
    //      pixels outside the tensor get an offset the range check rejects -> zeros
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.src), 0, p.M * C5 * 2, 0x00020000);
    constexpr int OOB = 0x40000000;
    int prel[PINST];
#pragma unroll
    for (int i = 0; i < PINST; ++i) {
        const int slot = (wv * PINST + i) * 64 + lane;
        const int q = slot / SPP5, sub = slot - q * SPP5;
        prel[i] = (q < p.np && sub < SPP5 - 1) ? q * (C5 * 2) + sub * 16 : OOB;
    }
    auto pdma = [&](int tile, int buf) {
        const int base = (tile * BM5 - halo) * (C5 * 2);
        char* l = patch + buf * p.patch_bytes + wv * (PINST * 1024);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int i = 0; i < PINST; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (lvoid_t*)(l + i * 1024), 16, prel[i] + base, 0, 0, 0);
#else
        (void)base; (void)l;
#endif
    };

    // ---- lazy input: coefficient table, the residual's patch (same slots, same lanes as the z' patch), the in-place transform of the slots
    //      THIS wave's DMA pieces landed (legal right after the wave's own vmcnt wait; the tile barrier publishes the result)
    float* coefs = reinterpret_cast<float*>(smem + 2 * p.patch_bytes + 2 * 8 * 2 * C5 * sizeof(float));
    char* rbuf = reinterpret_cast<char*>(coefs + 2 * C5);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(XF == 2 ? p.in.res : p.src), 0, p.M * C5 * 2, 0x00020000);
    auto rdma = [&](int tile) {
        if constexpr (XF == 2) {
            const int base = (tile * BM5 - halo) * (C5 * 2);
            char* l = rbuf + wv * (PINST * 1024);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
            for (int i = 0; i < PINST; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rrs, (lvoid_t*)(l + i * 1024), 16, prel[i] + base, 0, 0, 0);
#else
            (void)base; (void)l;
#endif
        }
    };
    auto transform = [&](int tile, int buf) {
        if constexpr (XF != 0) {
            if (p.in_debug & 2) return;
            const int pix0 = tile * BM5 - halo;                       // global pixel of patch pixel 0
            char* l = patch + buf * p.patch_bytes + (wv * PINST) * 1024 + lane * 16;
            const char* lr = rbuf + (wv * PINST) * 1024 + lane * 16;
#pragma unroll 2
            for (int i = 0; i < PINST; ++i) {
                const int pr = prel[i];
                if (pr == OOB) continue;                              // pad slot / zero area / behind the patch: stays zero
                const int q = pr >> 7, sub = (pr >> 4) & 7;
                const int g = pix0 + q;
                if (g < 0 || g >= p.M) continue;                      // outside the tensor: the DMA wrote zeros, and zeros they stay
                const uint4 v = *reinterpret_cast<const uint4*>(l + i * 1024);
                float sc[8], sh[8];
                *reinterpret_cast<f32x4*>(sc) = *reinterpret_cast<const f32x4*>(coefs + sub * 8);
                *reinterpret_cast<f32x4*>(sc + 4) = *reinterpret_cast<const f32x4*>(coefs + sub * 8 + 4);
                *reinterpret_cast<f32x4*>(sh) = *reinterpret_cast<const f32x4*>(coefs + C5 + sub * 8);
                *reinterpret_cast<f32x4*>(sh + 4) = *reinterpret_cast<const f32x4*>(coefs + C5 + sub * 8 + 4);
                uint4 o;
                unsigned mk = 0;
                if constexpr (XF == 2) o = bn_res_relu8_bf16(v, *reinterpret_cast<const uint4*>(lr + i * 1024), sc, sh, mk);
                else o = bn_relu8_bf16(v, sc, sh);
                *reinterpret_cast<uint4*>(l + i * 1024) = o;
                if (q >= halo && q < halo + BM5 && !(p.in_debug & 1)) {      // a pixel of this workgroup's tile: the activation's one writer
                    *reinterpret_cast<uint4*>(p.in.y + (size_t)g * C5 + sub * 8) = o;
                    if (XF == 2 && p.in.mask != nullptr) p.in.mask[(size_t)g * (C5 / 8) + sub] = (unsigned char)mk;
                }
            }
        }
    };

    // ---- pixel fragment addresses: tile i of this wave = pixels wv*64 + i*32 + l31, K half kh
    int xaddr[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) xaddr[i] = (wv * 64 + i * 32 + l31 + halo) * PP5 + kh * 16;
    const int zoff = p.zoff;

    // workgroup -> tiles.  Block b runs on XCD b % 8 (observed placement, used for speed only): every XCD walks a CONTIGUOUS range of
    // tiles, its 32 workgroups side by side, so the 2W + 2 halo pixels a tile shares with its neighbours are hits in that XCD's L2
    // instead of a second trip to the fabric (the round-robin deal put adjacent tiles on different XCDs)
    const int G = gridDim.x;
    int t_first, t_step, nmy;
    if ((G & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = (p.n_tiles + 7) >> 3;
        const int lo = xcd * per, hi = min(p.n_tiles, lo + per);
        t_step = G >> 3;
        t_first = lo + j;
        nmy = t_first < hi ? (hi - t_first + t_step - 1) / t_step : 0;
    } else {
        t_first = blockIdx.x; t_step = G;
        nmy = (p.n_tiles - t_first + G - 1) / G;
    }
    if (nmy <= 0) {
        // (a lazy input's by-products -- saved statistics, running-statistics update -- are workgroup 0's, which always has a tile)
        return;
    }
    // prologue: patch of the first tile -> buffer 1, filter rows 0-31 -> buffer 0 -> registers, rows 32-63 likewise
    pdma(t_first, 1);
    rdma(t_first);
    wdma(0);
    if constexpr (XF != 0) {
        // scale / shift of the producer's BatchNorm from its fp64 sums while the first DMAs fly (the fp64 scratch is the residual buffer's tail:
        // nothing lands there -- the DMA pieces cover PINST * 4 KB from its start, the scratch sits behind the zero area of that image)
        lazy_in_coefs(p.in, C5, coefs, reinterpret_cast<double*>(coefs + 2 * C5 + (XF == 2 ? p.patch_bytes / 4 : 0)), blockIdx.x == 0);
    }
    wait_vm0();
    transform(t_first, 1);
    wait_lds5();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    wread(0);
    wait_lds5();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    wdma(1);
    wait_vm0();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    wread(1);
    wait_lds5();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    for (int k = 0; k < nmy; ++k) {
        const int tile = t_first + k * t_step;
        const int m0 = tile * BM5;
        const char* pb = patch + ((k + 1) & 1) * p.patch_bytes;         // the first tile sits in buffer 1 (buffer 0 staged the filter rows)
        if (k + 1 < nmy) { pdma(tile + t_step, k & 1); rdma(tile + t_step); }     // lands under this tile's 144 MFMAs
        unsigned tmask[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int g = m0 + wv * 64 + i * 32 + l31;
            tmask[i] = g < p.M ? tap_mask5<MODE>(g, p) : 0u;
        }
        f32x16 acc[2][2];                                       // [channel tile j][pixel tile i]
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

        // 36 steps (tap t, K-step ks); the two fragments of step n+1 are requested before the four MFMAs of step n are issued
        bf16x8_t xf[2][2];
        auto frag = [&](int n, bf16x8_t (&f)[2]) {
            const int t = n >> 2, ks = n & 3;
            const int r = t / 3, s = t - 3 * r;
            const int shift = (MODE == 0 ? (r - 1) * W + (s - 1) : (1 - r) * W + (1 - s)) * PP5;
            const unsigned bit = 1u << t;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int a = xaddr[i] + shift;
                const int xa = (tmask[i] & bit) ? a : zoff + (a & 255);
                f[i] = *reinterpret_cast<const bf16x8_t*>(pb + xa + ks * 32);
            }
        };
        frag(0, xf[0]);
#pragma unroll
        for (int n = 0; n < 36; ++n) {
            const int t = n >> 2, ks = n & 3;
            if (n + 1 < 36) frag(n + 1, xf[(n + 1) & 1]);
            // pin the order: the reads of step n + 1 are ISSUED before the MFMAs of step n (left alone, hipcc sinks each read to its
            // use -- one fragment register set, every LDS round trip exposed between two MFMAs)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[j][t][ks], xf[n & 1][i], acc[j][i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        wait_vm0();                                             // this wave's part of the next patch has landed (issued 144 MFMAs ago)

        // ---- epilogue (conv4.hip's): D[row = channel (r&3) + 8*(r>>2) + 4*kh][col = pixel l31]
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pix = m0 + wv * 64 + i * 32 + l31;
            const bool pv = pix < p.M;
            bf16_t* drow = p.dst + (size_t)pix * C5;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (MODE == 1 && p.accumulate && pv) {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const uint2 old = *reinterpret_cast<const uint2*>(drow + j * 32 + g4 * 8 + kh * 4);
                        acc[j][i][4 * g4 + 0] += __uint_as_float(old.x << 16); acc[j][i][4 * g4 + 1] += __uint_as_float(old.x & 0xffff0000u);
                        acc[j][i][4 * g4 + 2] += __uint_as_float(old.y << 16); acc[j][i][4 * g4 + 3] += __uint_as_float(old.y & 0xffff0000u);
                    }
                }
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    unsigned ax = pack_bf16x2(acc[j][i][8 * pr + 0], acc[j][i][8 * pr + 1]), ay = pack_bf16x2(acc[j][i][8 * pr + 2], acc[j][i][8 * pr + 3]);
                    unsigned bx = pack_bf16x2(acc[j][i][8 * pr + 4], acc[j][i][8 * pr + 5]), by = pack_bf16x2(acc[j][i][8 * pr + 6], acc[j][i][8 * pr + 7]);
                    auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                    auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                    if (pv) *reinterpret_cast<u32x4*>(drow + j * 32 + pr * 16 + kh * 8) = u32x4{rx[0], ry[0], rx[1], ry[1]};
                }
            }
        }
        float* red = red0 + (k & 1) * (8 * 2 * C5);
        const bool stats = MODE == 0 && p.stat_acc != nullptr;
        if (stats) {
            // per-channel sum z, sum z^2 over this wave's 64 pixels from the fp32 accumulators (pixels beyond M hold exact zeros)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float sv[32];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float a = acc[j][0][r], b = acc[j][1][r];
                    sv[r] = a + b;
                    sv[16 + r] = fmaf(a, a, b * b);
                }
                row16_sum_n(sv);
                if ((lane & 15) == 0) {
                    const int rh = (lane >> 4) & 1;
#pragma unroll
                    for (int which = 0; which < 2; ++which)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            const int cc2 = j * 32 + g4 * 8 + kh * 4;
                            const float* v = sv + which * 16 + g4 * 4;
                            *reinterpret_cast<f32x4*>(red + ((wv * 2 + rh) * 2 + which) * C5 + cc2) = f32x4{v[0], v[1], v[2], v[3]};
                        }
                }
            }
        }
        if (k + 1 < nmy) transform(tile + t_step, k & 1);       // this wave's slots of the next patch: raw z' -> the activation, in place
        // one barrier per tile: every wave is done reading this tile's patch buffer (the DMA of tile k + 2 may overwrite it), every wave's
        // part of the next patch has landed (waited for above), and the statistics rows of this tile are visible
        wait_lds5();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (stats && tid < 2 * C5) {
            const int which = tid / C5, c2 = tid - which * C5;
            float t = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < 8; ++w2) t += red[(w2 * 2 + which) * C5 + c2];
            atomicAdd(p.stat_acc + ((size_t)(tile & (p.stat_rep - 1)) * 2 + which) * C5 + c2, (double)t);
        }
    }
}

template <int MODE, int PINST, int XF = 0>
int launch5(Conv5Params& p, hipStream_t st) {
    size_t lds = (size_t)2 * p.patch_bytes + 2 * 8 * 2 * C5 * sizeof(float);
    if (XF != 0) lds += 2 * C5 * sizeof(float) + (XF == 2 ? p.patch_bytes : 0) + 256 * sizeof(double);
    if (lds > 160 * 1024) { clhip_set_error("conv5: %zu bytes of LDS", lds); return CLHIP_EINVAL; }
    auto kern = conv5_kernel<MODE, PINST, XF>;
    static size_t attr = 0;
    if (lds > attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            clhip_set_error("conv5: cannot reserve %zu bytes of LDS", lds);
            return CLHIP_EHIP;
        }
        attr = lds;
    }
    static const int force_grid = clhip_cfg("CONV5_GRID") ? atoi(clhip_cfg("CONV5_GRID")) : 0;
    int grid = force_grid > 0 ? force_grid : 256;
    if (grid > p.n_tiles) grid = p.n_tiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, p);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

int pinst5(int W) { return (((BM5 + 2 * W + 2) * PP5 + 255) / 256 * 256 + 512 + 1023) / 1024 / 4 + 1; }

}  // namespace

bool clhip_conv5_supported(int N, int H, int W, int Cs, int Cd, int ksize, int stride, int pad, int dtype) {
    static const bool on_env = clhip_cfg("CONV5") ? atoi(clhip_cfg("CONV5")) != 0 : true;
    if (g_enable5 >= 0 ? g_enable5 == 0 : !on_env) return false;
    if (!(dtype == CLHIP_BF16 && ksize == 3 && stride == 1 && pad == 1 && Cs == C5 && Cd == C5 && W <= 32 && W >= 2 && H >= 1)) return false;
    const int64_t M = (int64_t)N * H * W;
    if (M * C5 * 2 >= ((int64_t)1 << 29)) return false;          // the out-of-range marker of the patch DMA is a 1 GiB offset
    // at least ~3 tiles of 256 pixels per CU-resident workgroup: below that the one-time filter load (73.7 KB per wave) does not pay
    static const int min_tiles_env = clhip_cfg("CONV5_MIN_TILES") ? atoi(clhip_cfg("CONV5_MIN_TILES")) : 512;
    return (M + BM5 - 1) / BM5 >= (g_min_tiles5 >= 0 ? g_min_tiles5 : min_tiles_env);
}

void clhip_conv5_enable(int on) { g_enable5 = on; }
void clhip_conv5_min_tiles(int n) { g_min_tiles5 = n; }

int clhip_conv5_tiles_m(int M) { return (M + BM5 - 1) / BM5; }

int clhip_conv5_launch_in(const void* src, const void* wt, void* dst, double* stat_acc, int stat_rep, int N, int H, int W, int accumulate, int mode, const LazyIn* in,
                          hipStream_t st);

int clhip_conv5_launch(const void* src, const void* wt, void* dst, double* stat_acc, int stat_rep, int N, int H, int W, int accumulate, int mode, hipStream_t st) {
    return clhip_conv5_launch_in(src, wt, dst, stat_acc, stat_rep, N, H, W, accumulate, mode, nullptr, st);
}

// in != nullptr (forward only): src is the producer's pre-BatchNorm output, the operand relu(bn(src) [+ in->res]) is formed in LDS and written to in->y
int clhip_conv5_launch_in(const void* src, const void* wt, void* dst, double* stat_acc, int stat_rep, int N, int H, int W, int accumulate, int mode, const LazyIn* in,
                          hipStream_t st) {
    Conv5Params p;
    if (in != nullptr) {
        if (mode != 0 || in->acc == nullptr || in->y == nullptr) { clhip_set_error("conv5: a lazy input needs the forward mode, the producer's sums and an output activation"); return CLHIP_EINVAL; }
        p.in = *in;
    }
    static const int wt_debug = clhip_cfg("WT_DEBUG") ? atoi(clhip_cfg("WT_DEBUG")) : 0;
    p.in_debug = wt_debug;
    p.src = static_cast<const bf16_t*>(src); p.wt = static_cast<const bf16_t*>(wt); p.dst = static_cast<bf16_t*>(dst);
    p.stat_acc = stat_acc; p.stat_rep = stat_rep > 0 ? stat_rep : 1;
    p.H = H; p.W = W; p.M = N * H * W; p.accumulate = accumulate;
    p.wshift = ilog2_exact(W); p.hshift = ilog2_exact(H);
    p.np = BM5 + 2 * W + 2;
    p.zoff = (p.np * PP5 + 255) / 256 * 256;
    const int pinst = pinst5(W);
    p.patch_bytes = pinst * 4 * 1024;
    p.n_tiles = (p.M + BM5 - 1) / BM5;
    if (p.zoff + 512 > p.patch_bytes) { clhip_set_error("conv5: patch geometry"); return CLHIP_EINVAL; }
#define L5(PI) (mode == 0 ? (in == nullptr ? launch5<0, PI>(p, st) : (in->res != nullptr ? launch5<0, PI, 2>(p, st) : launch5<0, PI, 1>(p, st))) : launch5<1, PI>(p, st))
    switch (pinst) {
        case 10: return L5(10);
        case 11: return L5(11);
        case 12: return L5(12);
        case 13: return L5(13);
        default: break;
    }
#undef L5
    clhip_set_error("conv5: no kernel for W = %d (%d DMA instructions per wave)", W, pinst);
    return CLHIP_EINVAL;
}
