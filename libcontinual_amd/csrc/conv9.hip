// conv9.hip -- 3x3 / stride 1 / pad 1 convolution (forward and dgrad) of the 128 -> 128 and 256 -> 256-channel layers, bf16, gfx950: the input patch
// of a tile stays in LDS for the whole reduction, the filters stream through a two-stage ring of 32-KB slabs with ONE barrier per slab.
//
// What the round-5 phase traces said about conv4.hip on these layers (gpurun_out/r05_c4t, profiles/r05_conv_notes.md): its reduction is cut into
// (tap, 32 / 64-channel) slabs of 8-16 MFMAs per wave with a read phase, a DMA wait and two barriers around each -- per 512 cycles of matrix pipe a
// SIMD spends ~650 cycles waiting for the slab that was requested one slab earlier, ~500 for its fragment reads and ~800 at barriers -- and every
// fragment address goes through a per-tap mask select.  Here
//   * a workgroup (8 waves, two per SIMD, one workgroup per CU) owns 256 pixels x 128 output channels (128 channels: one 16 x 16 image, or 8 rows of
//     a 32-wide one) or 128 pixels x 128 of the 256 output channels (256 channels, the 8 x 8 maps: two images; the two halves of the eight K-steps of
//     a slab go to two wave groups whose partial tiles are summed through LDS at the end).  Its patch -- ALL input channels, 77-87 KB -- lands once;
//   * the patch is ZERO-PADDED as in conv8.hip (one pad column per row, a zero row between / around images: no tap masks, no address selects) and
//     XOR-swizzled at 2 C bytes per pixel (swizzle on the DMA's source address and on the read address; the LDS image is lane-linear);
//   * a slab = (tap, 128 input channels) x 128 output channels = 32 KB: 32 MFMAs per wave (16 with two K groups) between two barriers, the next slab
//     requested a whole slab ahead (1000-2000 cycles of MFMAs: an L2 round trip) -- 9 / 18 barriers per launch instead of ~150;
//   * a wave owns 64 pixels x 64 channels (2 x 2 MFMA 32x32x16 tiles), reads the fragments of K-step n + 1 under the MFMAs of step n; weight
//     fragment addresses are eight precomputed registers, pixel fragment addresses one XOR per read.
// XF (forward): the producer's BatchNorm [+ residual] + ReLU applied to the landed patch in place, once per launch, the activation [+ packed ReLU mask]
// written by the workgroup that owns the pixel (common.h LazyIn; bit for bit what bn_apply_train_kernel stores).
// MODE 0 = forward, 1 = dgrad (mirrored taps; the dgrad weight copy [C][9][K] has the forward copy's layout), with conv4.hip's epilogues (BatchNorm
// statistics from the fp32 accumulators; dgrad: accumulate, BatchNorm-backward sums of the producing layer).
// Replaces nn.Conv2d forward / input gradient of ResNet-18's layer2 / layer3 3x3 stride-1 convolutions (core/model/backbone/resnet.py:17-24, 295-298)
// and the BatchNorm + ReLU (+ residual) in front of them (resnet.py:37-63).
#include <stdlib.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void lvoid_t;

constexpr int OOB9 = 0x40000000;
constexpr int PINST9 = 11;              // patch DMA pieces (1 KB) per wave: up to 88 per patch
constexpr int RING9 = 2 * 32768;

struct Conv9Params {
    const bf16_t* src;   // [N,H,W,C]
    const bf16_t* wt;    // [C][9][C]
    bf16_t* dst;         // [N,H,W,C]
    double* stat_acc;    // forward: [stat_rep][2][C] fp64 accumulators, or nullptr
    int stat_rep;
    const bf16_t* bn_z;  // dgrad: BatchNorm-backward sums of the producer (conv4.hip)
    const bf16_t* bn_y;
    const float* bn_mean;
    const float* bn_invstd;
    double* bn_acc;
    int bn_rep;
    int H, W, M, accumulate;
    int TP;              // pixels per tile (256 / KG)
    int R;               // single-image tiles: image rows per tile; multi-image tiles: 0
    int nimg;            // images per tile (multi-image tiles)
    int tiles_per_img;
    int np, npieces, patch_bytes;
    int n_mtiles, n_ntiles;
    int inv_pw, inv_h1, inv_w, inv_hw;   // ceil(65536 / d): floor(x / d) = (x * inv) >> 16 for the small x of the prologue (x * (d * inv - 65536) < 65536)
    LazyIn in;
    unsigned long long* trace;
};

int g_enable9 = -1;
unsigned long long* g_trace9 = nullptr;
#ifdef CLHIP_ABLATION
#define STAMP9() do { if (p.trace && blockIdx.x == 0 && lane == 0 && nstamp < 64) p.trace[wave * 64 + nstamp++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP9() do { } while (0)
#endif

__device__ __forceinline__ void wait_vm0_9() { __builtin_amdgcn_s_waitcnt(0x0070 | 0xF00); }
__device__ __forceinline__ void wait_lds9() { __builtin_amdgcn_s_waitcnt(0xC07F); }
__device__ __forceinline__ void wg_barrier9() {
    wait_lds9();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ unsigned or8_9(unsigned v) {
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);
    return v;
}

template <int C, int MODE, int XF, bool BNR>
__global__ __launch_bounds__(512, 2) void conv9_kernel(const Conv9Params p) {
    static_assert(C == 128 || C == 256, "channel counts");
    static_assert(XF == 0 || MODE == 0, "lazy inputs exist in the forward only");
    static_assert(!BNR || MODE == 1, "BatchNorm-backward sums belong to the dgrad");
    constexpr int KG = C / 128;             // K groups: wave groups that split the eight K-steps of a slab
    constexpr int NCH = C / 128;            // 128-channel halves of the reduction per tap
    constexpr int NS = 9 * NCH;             // slabs
    constexpr int KSW = 8 / KG;             // K-steps of a slab per wave
    constexpr int SPP = C / 8;              // sixteen-byte slots per patch pixel
    constexpr int LSPP = C == 128 ? 4 : 5;
    constexpr int PITCH = 2 * C;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = KG == 1 ? 0 : wave >> 2;
    const int pq = KG == 1 ? wave >> 1 : (wave >> 1) & 1;
    const int jh = wave & 1;
    const int l31 = lane & 31, kh = lane >> 5;
    const int W = p.W, H = p.H, PW = W + 1;
    const int mt = blockIdx.x / p.n_ntiles, nt = blockIdx.x - mt * p.n_ntiles;
    const int n0 = nt * 128;
    const int g0 = mt * p.TP;                                   // first pixel of the tile
    int nstamp = 0; (void)nstamp;
    STAMP9();

    // ---- LDS map: the patch | two ring stages of 32 KB (stage 1 also hosts the coefficient table of a lazy input before slab 1 is requested; the
    //      whole ring hosts the partial tiles of the second K group and the statistics rows after the last slab)
    char* ring = smem + p.patch_bytes;

    // ---- patch DMA: piece I = 8 i + wave fills LDS bytes [I * 1024, +1024): slot n = I * 64 + lane is patch pixel lp = n / SPP, swizzled slot
    //      n % SPP; its channel chunk = slot ^ (lp & 15) on the low four bits.  prel = byte offset relative to the tile's first pixel (negative in the
    //      halo row above a single-image tile), OOB9 for pad slots, zero rows and halo rows outside the image
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.src), 0, p.M * C * 2, 0x00020000);
    int prel[PINST9];
    {
        const int ti = p.R > 0 ? mt % p.tiles_per_img : 0;
        const bool kill_top = p.R > 0 && ti == 0, kill_bot = p.R > 0 && ti == p.tiles_per_img - 1;
#pragma unroll
        for (int i = 0; i < PINST9; ++i) {
            const int I = i * 8 + wave;
            const int n = I * 64 + lane;
            const int lp = n >> LSPP, sp = n & (SPP - 1);
            const int r = (lp * p.inv_pw) >> 16, c = lp - r * PW;
            const int chunk = (sp & ~15) | ((sp ^ lp) & 15);
            int v = OOB9;
            if (I < p.npieces && lp < p.np - 1 && c != 0) {
                if (p.R > 0) {
                    if (!(r == 0 && kill_top) && !(r == p.R + 1 && kill_bot)) v = ((r - 1) * W + (c - 1)) * PITCH + chunk * 16;
                } else {
                    const int img = (r * p.inv_h1) >> 16, rb = r - img * (H + 1);
                    if (rb != 0 && img < p.nimg) v = ((img * H + rb - 1) * W + (c - 1)) * PITCH + chunk * 16;
                }
            }
            prel[i] = v;
        }
    }
    {
        const int base = g0 * PITCH;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int i = 0; i < PINST9; ++i) {
            const int I = i * 8 + wave;
            if (I < p.npieces) __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (lvoid_t*)(smem + I * 1024), 16, prel[i] == OOB9 ? OOB9 : prel[i] + base, 0, 0, 0);
        }
#else
        (void)base;
#endif
    }
    // the residual of a lazy input: plain 16-byte loads (coalesced: consecutive lanes = consecutive chunks of a pixel), consumed by the transform
    uint4 rr[XF == 2 ? PINST9 : 1];
    if constexpr (XF == 2) {
#pragma unroll
        for (int i = 0; i < PINST9; ++i) {
            rr[i] = make_uint4(0, 0, 0, 0);
            if (i * 8 + wave < p.npieces && prel[i] != OOB9)
                rr[i] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.in.res) + (size_t)g0 * PITCH + prel[i]);
        }
    }

    // ---- weight slabs: slab s = (tap, 128-channel half ch) = rows n0 .. n0 + 127 of the filter bank, 256 bytes each, into ring stage s & 1; the
    //      sixteen slots of a row are XOR-swizzled by the row (conflict-free fragment reads of 16 consecutive rows)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.wt), 0, C * 9 * C * 2, 0x00020000);
    // (from slab 1 on the four waves 4 .. 7 -- the second wave of every SIMD -- request the slabs, eight pieces each, while waves 0 .. 3 start their
    //  MFMAs at once: the two waves of a SIMD fall out of step and the matrix pipe is fed by one while the other issues DMA / waits at the barrier)
    int wrel[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int n = (i * 4 + (wave & 3)) * 64 + lane;
        const int row = n >> 4, sl = n & 15;
        wrel[i] = (n0 + row) * (9 * C * 2) + ((sl ^ row) & 15) * 16;
    }
    auto wdma = [&](int tap, int ch, int stage, int i0, int i1) {
        const int uni = (tap * C + ch * 128) * 2;
        char* l = ring + stage * 32768 + (wave & 3) * 1024;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i >= i0 && i < i1) __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lvoid_t*)(l + i * 4096), 16, wrel[i] + uni, 0, 0, 0);
#else
        (void)uni; (void)l; (void)i0; (void)i1;
#endif
    };
    if (wave < 4) wdma(0, 0, 0, 0, 4); else wdma(0, 0, 0, 4, 8);        // slab 0: everybody, four pieces each
    STAMP9();

    // ---- lazy input: coefficient table, then every wave rewrites the slots its own DMA pieces landed
    if constexpr (XF != 0) {
        float* coefs = reinterpret_cast<float*>(ring + 32768);
        lazy_in_coefs(p.in, C, coefs, reinterpret_cast<double*>(ring + 32768 + 2 * C * sizeof(float)), blockIdx.x == 0);
        wait_vm0_9();
#pragma unroll
        for (int i = 0; i < PINST9; ++i) {
            const int I = i * 8 + wave;
            if (I >= p.npieces) continue;
            const int pr = prel[i];
            if (pr == OOB9) continue;                                // pad slot / zero row / outside the image: stays zero
            char* l = smem + I * 1024 + lane * 16;
            const int rel = pr >> 4;                                 // (pixel relative to the tile) * SPP + chunk
            const int sub = rel & (SPP - 1), pix = rel >> LSPP;
            const uint4 v = *reinterpret_cast<const uint4*>(l);
            float sc[8], sh[8];
            *reinterpret_cast<f32x4*>(sc) = *reinterpret_cast<const f32x4*>(coefs + sub * 8);
            *reinterpret_cast<f32x4*>(sc + 4) = *reinterpret_cast<const f32x4*>(coefs + sub * 8 + 4);
            *reinterpret_cast<f32x4*>(sh) = *reinterpret_cast<const f32x4*>(coefs + C + sub * 8);
            *reinterpret_cast<f32x4*>(sh + 4) = *reinterpret_cast<const f32x4*>(coefs + C + sub * 8 + 4);
            uint4 o;
            unsigned mk = 0;
            if constexpr (XF == 2) o = bn_res_relu8_bf16(v, rr[i], sc, sh, mk);
            else o = bn_relu8_bf16(v, sc, sh);
            *reinterpret_cast<uint4*>(l) = o;
            const bool own = pix >= 0 && pix < p.TP && nt == 0;      // the tile's pixels: the activation's one writer (the first channel tile's workgroup)
            if (own) *reinterpret_cast<uint4*>(p.in.y + ((size_t)g0 + pix) * C + sub * 8) = o;
            if (XF == 2 && p.in.mask != nullptr) {
                // the 8 lanes of an aligned group hold 8 consecutive mask bytes of one pixel: one 8-byte store
                const unsigned b = sub & 7;
                const unsigned lo = or8_9(b < 4 ? mk << (8 * b) : 0u), hi = or8_9(b >= 4 ? mk << (8 * (b - 4)) : 0u);
                if (own && (lane & 7) == 0) *reinterpret_cast<uint2*>(p.in.mask + ((size_t)g0 + pix) * SPP + (sub & ~7)) = make_uint2(lo, hi);
            }
        }
    }

    // ---- fragment bases.  Output pixel q of the tile sits at patch pixel lp(q); biased to tap (0, 0) = (row - 1, col - 1)
    int lptl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = pq * 64 + i * 32 + l31;
        int lp;
        if (p.R > 0) { const int r = (q * p.inv_w) >> 16, c = q - r * W; lp = (r + 1) * PW + c + 1; }
        else { const int hw = H * W, img = (q * p.inv_hw) >> 16, rem = q - img * hw, r = (rem * p.inv_w) >> 16, c = rem - r * W; lp = (img * (H + 1) + r + 1) * PW + c + 1; }
        lptl[i] = lp - PW - 1;
    }
    int wad[2][KSW];                                            // weight fragment addresses inside a stage: row o = jh*64 + j*32 + l31, chunk 2 ks + kh
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int o = jh * 64 + j * 32 + l31;
        const int wb = o * 256 + (((o & 15) ^ kh) << 4);
#pragma unroll
        for (int q = 0; q < KSW; ++q) wad[j][q] = wb ^ ((kg * KSW + q) << 5);
    }

    f32x16 acc[2][2];                                           // [channel tile j][pixel tile i]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int tr = tap / 3, ts = tap - 3 * tr;
        const int shift = MODE == 0 ? tr * PW + ts : (2 - tr) * PW + (2 - ts);
        int xb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int lp = lptl[i] + shift;
            xb[i] = lp * PITCH + (((lp & 15) ^ kh) << 4);
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int s = tap * NCH + ch;
            wait_vm0_9();                                       // this wave's pieces of slab s (and, at s = 0, of the patch) have landed
            STAMP9();
            wg_barrier9();                                      // ... everybody's; and nobody reads the other stage any more
            STAMP9();
            if (s + 1 < NS && wave >= 4) wdma((s + 1) / NCH, (s + 1) % NCH, (s + 1) & 1, 0, 8);      // lands under this slab's MFMAs
            STAMP9();
            const char* ws = ring + (s & 1) * 32768;
            const char* xs = smem + ch * 256;
            // The fragments of K-step q + 2 are requested BETWEEN the four MFMAs of step q, one read behind each: an in-order wave that issues its
            // reads and address arithmetic in a block leaves the pipe idle meanwhile (r05 trace: a wave that had its SIMD to itself multiplied at
            // 43 % of the pipe rate with the reads of a step issued ahead of its four MFMAs)
            bf16x8_t xf[3][2], wf[3][2];
            auto xread = [&](int q, int i) { return *reinterpret_cast<const bf16x8_t*>(xs + (xb[i] ^ ((kg * KSW + q) << 5))); };
            auto wread = [&](int q, int j) { return *reinterpret_cast<const bf16x8_t*>(ws + wad[j][q]); };
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int i = 0; i < 2; ++i) xf[q][i] = xread(q, i);
#pragma unroll
                for (int j = 0; j < 2; ++j) wf[q][j] = wread(q, j);
            }
#pragma unroll
            for (int q = 0; q < KSW; ++q) {
                const int c = q % 3, n = (q + 2) % 3;
                const bool more = q + 2 < KSW;
                __builtin_amdgcn_sched_barrier(0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c][0], xf[c][0], acc[0][0], 0, 0, 0);
                if (more) xf[n][0] = xread(q + 2, 0);
                __builtin_amdgcn_sched_barrier(0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c][0], xf[c][1], acc[0][1], 0, 0, 0);
                if (more) xf[n][1] = xread(q + 2, 1);
                __builtin_amdgcn_sched_barrier(0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c][1], xf[c][0], acc[1][0], 0, 0, 0);
                if (more) wf[n][0] = wread(q + 2, 0);
                __builtin_amdgcn_sched_barrier(0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c][1], xf[c][1], acc[1][1], 0, 0, 0);
                if (more) wf[n][1] = wread(q + 2, 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            STAMP9();
        }
    }

    // ---- K groups: the partial tiles of group 1 are summed into group 0 through LDS (the ring is free: one barrier after its last reads)
    STAMP9();
    wg_barrier9();
    STAMP9();
    if constexpr (KG > 1) {
        float* xr = reinterpret_cast<float*>(ring);
        if (kg > 0) {
            float* q = xr + (size_t)(wave & 3) * 64 * 64;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4)
                        *reinterpret_cast<f32x4*>(q + ((j * 2 + i) * 4 + r4) * 256 + lane * 4) =
                            f32x4{acc[j][i][4 * r4], acc[j][i][4 * r4 + 1], acc[j][i][4 * r4 + 2], acc[j][i][4 * r4 + 3]};
        }
        wg_barrier9();
        if (kg > 0) return;
        const float* q = xr + (size_t)(wave & 3) * 64 * 64;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(q + ((j * 2 + i) * 4 + r4) * 256 + lane * 4);
                    acc[j][i][4 * r4] += v[0]; acc[j][i][4 * r4 + 1] += v[1]; acc[j][i][4 * r4 + 2] += v[2]; acc[j][i][4 * r4 + 3] += v[3];
                }
        wg_barrier9();                                          // (the statistics rows below reuse the exchange area)
    }

    // ---- epilogue (conv4.hip's): D[row = channel (r&3) + 8*(r>>2) + 4*kh][col = pixel l31]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const size_t pix = (size_t)g0 + pq * 64 + i * 32 + l31;
        bf16_t* drow = p.dst + pix * C + n0 + jh * 64;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (MODE == 1 && p.accumulate) {
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
                *reinterpret_cast<u32x4*>(drow + j * 32 + pr * 16 + kh * 8) = u32x4{rx[0], ry[0], rx[1], ry[1]};
            }
        }
    }
    STAMP9();
    const bool fwd_stats = MODE == 0 && p.stat_acc != nullptr;
    constexpr int NROW = (KG == 1 ? 4 : 2) * 2;                 // half-wave rows of 16 pixels per channel: pixel groups x 2
    float* red = reinterpret_cast<float*>(ring);                // [NROW][2][128]
    if (fwd_stats || BNR) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float sv[32];
            if constexpr (MODE == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float a = acc[j][0][r], b = acc[j][1][r];
                    sv[r] = a + b;
                    sv[16 + r] = fmaf(a, a, b * b);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 32; ++r) sv[r] = 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const size_t pix = (size_t)g0 + pq * 64 + i * 32 + l31;
                    const size_t row = pix * C + n0 + jh * 64 + j * 32 + kh * 4;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const uint2 zz = *reinterpret_cast<const uint2*>(p.bn_z + row + g4 * 8);
                        uint2 yy = make_uint2(0x3f803f80u, 0x3f803f80u);
                        if (p.bn_y != nullptr) yy = *reinterpret_cast<const uint2*>(p.bn_y + row + g4 * 8);
                        const float z4[4] = {__uint_as_float(zz.x << 16), __uint_as_float(zz.x & 0xffff0000u), __uint_as_float(zz.y << 16), __uint_as_float(zz.y & 0xffff0000u)};
                        const float y4[4] = {__uint_as_float(yy.x << 16), __uint_as_float(yy.x & 0xffff0000u), __uint_as_float(yy.y << 16), __uint_as_float(yy.y & 0xffff0000u)};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float g = y4[e] > 0.f ? acc[j][i][4 * g4 + e] : 0.f;
                            sv[4 * g4 + e] += g;
                            sv[16 + 4 * g4 + e] = fmaf(g, z4[e], sv[16 + 4 * g4 + e]);
                        }
                    }
                }
            }
            row16_sum_n(sv);
            if ((lane & 15) == 0) {
                const int rh = (lane >> 4) & 1;
#pragma unroll
                for (int which = 0; which < 2; ++which)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int cc2 = jh * 64 + j * 32 + g4 * 8 + kh * 4;
                        const float* v = sv + which * 16 + g4 * 4;
                        *reinterpret_cast<f32x4*>(red + ((pq * 2 + rh) * 2 + which) * 128 + cc2) = f32x4{v[0], v[1], v[2], v[3]};
                    }
            }
        }
        STAMP9();
        wg_barrier9();
        STAMP9();
        if (tid < 256) {
            const int which = tid >> 7, c2 = tid & 127;
            float t = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < NROW; ++w2) t += red[(w2 * 2 + which) * 128 + c2];
            if constexpr (MODE == 0) {
                atomicAdd(p.stat_acc + ((size_t)(mt & (p.stat_rep - 1)) * 2 + which) * C + n0 + c2, (double)t);
            } else {
                if (which == 1) {                               // sum g * xhat from sum g z' and sum g
                    float sg = 0.f;
#pragma unroll
                    for (int w2 = 0; w2 < NROW; ++w2) sg += red[(w2 * 2 + 0) * 128 + c2];
                    t = p.bn_invstd[n0 + c2] * (t - p.bn_mean[n0 + c2] * sg);
                }
                atomicAdd(p.bn_acc + ((size_t)(mt & (p.bn_rep - 1)) * 2 + which) * C + n0 + c2, (double)t);
            }
        }
    }
    STAMP9();
}

bool geometry9(int N, int H, int W, int C, Conv9Params& p) {
    if (!(C == 128 || C == 256) || W < 4 || W > 32 || H < 1) return false;
    p.TP = 256 / (C / 128);
    p.H = H; p.W = W; p.M = N * H * W;
    const int hw = H * W;
    if (hw >= p.TP) {
        if (p.TP % W != 0) return false;
        p.R = p.TP / W; p.nimg = 1;
        if (H % p.R != 0) return false;
        p.tiles_per_img = H / p.R;
        p.np = (p.R + 2) * (W + 1) + 1;
    } else {
        if (p.TP % hw != 0) return false;
        p.R = 0; p.nimg = p.TP / hw; p.tiles_per_img = 1;
        if (N % p.nimg != 0) return false;
        p.np = (p.nimg * (H + 1) + 1) * (W + 1) + 1;
    }
    p.npieces = (p.np * 2 * C + 1023) / 1024;
    if (p.npieces > 8 * PINST9) return false;
    p.patch_bytes = p.npieces * 1024;
    if (p.patch_bytes + RING9 > 160 * 1024) return false;
    p.n_mtiles = p.M / p.TP;
    p.n_ntiles = C / 128;
    auto inv = [](int d) { return (65536 + d - 1) / d; };
    p.inv_pw = inv(W + 1); p.inv_h1 = inv(H + 1); p.inv_w = inv(W); p.inv_hw = inv(hw);
    return true;
}

template <int C, int MODE, int XF, bool BNR>
int launch9(Conv9Params& p, hipStream_t st) {
    const size_t lds = (size_t)p.patch_bytes + RING9;
    auto kern = conv9_kernel<C, MODE, XF, BNR>;
    int dev = 0;
    (void)hipGetDevice(&dev);
    static size_t attr[16] = {0};
    if (dev < 0 || dev >= 16 || lds > attr[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            clhip_set_error("conv9: cannot reserve %zu bytes of LDS", lds);
            return CLHIP_EHIP;
        }
        if (dev >= 0 && dev < 16) attr[dev] = lds;
    }
    hipLaunchKernelGGL(kern, dim3(p.n_mtiles * p.n_ntiles), dim3(512), lds, st, p);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

}  // namespace

bool clhip_conv9_supported(int N, int H, int W, int Cs, int Cd, int ksize, int stride, int pad, int dtype) {
    // OFF by default (CONV9=1 enables it): stand-alone it matches conv4.hip on the 128-channel layer (24.3 / 22.4 vs 24.2 / 22.9 us) and beats it on the
    // 256-channel one (23.0 / 21.6 vs 26.2 / 25.6 us), inside the ResNet-18 step every launch takes what conv4's takes (28 us) and the step is 1 % slower
    // (2.014 vs 1.995 ms, r05 A/B) -- profiles/r05_conv_notes.md
    static const bool on_env = clhip_cfg("CONV9") ? atoi(clhip_cfg("CONV9")) != 0 : false;
    if (g_enable9 >= 0 ? g_enable9 == 0 : !on_env) return false;
    if (!(dtype == CLHIP_BF16 && ksize == 3 && stride == 1 && pad == 1 && Cs == Cd && N >= 1)) return false;
    Conv9Params p;
    if (!geometry9(N, H, W, Cs, p)) return false;
    if ((int64_t)p.M * Cs * 2 >= ((int64_t)1 << 29)) return false;          // the out-of-range marker of the patch DMA is a 1 GiB offset
    return true;
}

void clhip_conv9_enable(int on) { g_enable9 = on; }
void clhip_conv9_set_trace(unsigned long long* dev_buf) { g_trace9 = dev_buf; }
int clhip_conv9_tiles_m(int N, int H, int W, int C) { Conv9Params p; return geometry9(N, H, W, C, p) ? p.n_mtiles : 0; }

// mode 0: forward (stat_acc may be nullptr); mode 1: dgrad, with the producer's BatchNorm-backward sums when bn_z != nullptr.
// in != nullptr (forward only): src is the producer's pre-BatchNorm output, the operand relu(bn(src) [+ in->res]) is formed in LDS and written to in->y
int clhip_conv9_launch(const void* src, const void* wt, void* dst, double* stat_acc, int stat_rep, int N, int H, int W, int C, int accumulate, int mode, const LazyIn* in,
                       const void* bn_z, const void* bn_y, const float* bn_mean, const float* bn_invstd, double* bn_acc, int bn_rep, hipStream_t st) {
    Conv9Params p;
    if (!geometry9(N, H, W, C, p)) { clhip_set_error("conv9: unsupported geometry %d x %d x %d x %d", N, H, W, C); return CLHIP_EINVAL; }
    if (in != nullptr) {
        if (mode != 0 || in->acc == nullptr || in->y == nullptr) { clhip_set_error("conv9: a lazy input needs the forward mode, the producer's sums and an output activation"); return CLHIP_EINVAL; }
        p.in = *in;
    }
    p.src = static_cast<const bf16_t*>(src); p.wt = static_cast<const bf16_t*>(wt); p.dst = static_cast<bf16_t*>(dst);
    p.stat_acc = stat_acc; p.stat_rep = stat_rep > 0 ? stat_rep : 1; p.accumulate = accumulate;
    p.bn_z = static_cast<const bf16_t*>(bn_z); p.bn_y = static_cast<const bf16_t*>(bn_y);
    p.bn_mean = bn_mean; p.bn_invstd = bn_invstd; p.bn_acc = bn_acc; p.bn_rep = bn_rep > 0 ? bn_rep : 1;
    p.trace = g_trace9;
#define L9(CC)                                                                                                      \
    do {                                                                                                            \
        if (mode == 0) {                                                                                            \
            if (in == nullptr) return launch9<CC, 0, 0, false>(p, st);                                              \
            return in->res != nullptr ? launch9<CC, 0, 2, false>(p, st) : launch9<CC, 0, 1, false>(p, st);          \
        }                                                                                                           \
        return bn_z != nullptr ? launch9<CC, 1, 0, true>(p, st) : launch9<CC, 1, 0, false>(p, st);                  \
    } while (0)
    if (C == 128) L9(128);
    L9(256);
#undef L9
}
