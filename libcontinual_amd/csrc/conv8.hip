// conv8.hip -- 3x3 / stride 1 / pad 1 convolution (forward and dgrad) of the 64 -> 64-channel layers on LARGE maps, bf16, gfx950:
// TWO independent four-wave workgroups per CU, so that on every SIMD one wave multiplies while the other one waits for memory.
//
// What rounds 3 and 4 measured on conv5.hip (profiles/r03_conv5_notes.md, r04_wt_notes.md): its 144 MFMAs per 256-pixel tile take 1.9 us, a
// tile costs 6.9 us -- the one 512-register wave a SIMD has issues its patch DMA (every `buffer_load ... lds` piece blocks the issuing wave for
// 100-320 cycles while the memory pipeline is full), multiplies, converts and stores IN SEQUENCE, and anything added to that wave (the
// in-LDS BatchNorm transform of round 4) is added to the launch.  Here
//   * a wave keeps the filters of 32 OUTPUT channels only (wave = (pixel half, channel half) of a 128-pixel x 64-channel tile): 144
//     registers of A operands, 32 accumulators -- a 256-register wave, two workgroups (<= 80 KB of LDS each) per CU.  The two waves of a SIMD
//     belong to different workgroups, share nothing but the matrix pipe and drift apart by themselves: DMA issue, epilogue stores, BatchNorm
//     sums and the operand transform of one run under the MFMAs of the other;
//   * a tile is R = 128 / W whole image rows.  The patch in LDS is ZERO-PADDED: (R + 2) rows of W + 1 pixels (one pad column serves as the
//     right neighbour of a row's last pixel and the left neighbour of the next row's first), the rows above / below an image get an offset
//     the buffer range check rejects -> no tap masks, no zero area, no per-tap address selects in the MFMA loop;
//   * 128 bytes per patch pixel (no pad slot: three buffers fit 80 KB) with the sixteen-byte slots XOR-swizzled by the pixel index -- applied
//     to the DMA's SOURCE address and to the fragment read address, the LDS image itself is lane-linear -- conflict-free for ds_read_b128;
//   * the filter rows reach the registers through LDS (coalesced DMA, conflict-free fragment reads) as in conv5.hip; BatchNorm statistics
//     (forward) / the producer's BatchNorm-backward sums (dgrad) come from the fp32 accumulators and leave the workgroup ONCE, after its last tile.
// XF (forward): 0 the source is an activation; 1 the source is the producer's pre-BatchNorm output z', the patch is rewritten in LDS with
// relu(scale z' + shift) by the wave that landed it and the workgroup that owns a pixel writes the activation; 2 relu(scale z' + shift + r)
// with the packed ReLU mask written as well (common.h LazyIn).  Bit for bit what bn_apply_train_kernel stores.
// MODE 0 = forward, 1 = dgrad (mirrored taps; the dgrad weight copy [C][9][K] has the forward copy's layout).  Replaces nn.Conv2d forward /
// input gradient of ResNet-18's layer1 (core/model/backbone/resnet.py:17-24, 295-298) and the BatchNorm + ReLU (+ residual) in front of it
// (resnet.py:37-63).
#include <stdlib.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void lvoid_t;

constexpr int C8 = 64;                  // channels in = channels out
constexpr int BM8 = 128;                // pixels per tile
constexpr int PINST8 = 7;               // patch DMA pieces (1 KB) per wave: up to 28 per buffer
constexpr int OOB8 = 0x40000000;

struct Conv8Params {
    const bf16_t* src;   // [N,H,W,64]
    const bf16_t* wt;    // [64][9][64]
    bf16_t* dst;         // [N,H,W,64]
    double* stat_acc;    // forward: [stat_rep][2][64] fp64 accumulators, or nullptr
    int stat_rep;
    // dgrad: BatchNorm-backward sums of the layer that PRODUCED the tensor whose gradient this launch completes (see conv4.hip)
    const bf16_t* bn_z;
    const bf16_t* bn_y;          // its activation (ReLU mask y > 0), or nullptr
    const unsigned char* bn_mask; // ... or its packed mask [M][8]
    const float* bn_gamma;       // ... or its BatchNorm weight / bias (ReLU straight behind the BatchNorm: mask = scale z' + shift > 0 with the forward's
    const float* bn_beta;        //     scale = gamma * invstd, shift = beta - mean * scale); none of the three: no ReLU
    const float* bn_mean;
    const float* bn_invstd;
    double* bn_acc;
    int bn_rep;
    int H, W, M, accumulate;
    int R;               // image rows per tile = 128 / W
    int tiles_per_img;   // H / R
    int np;              // patch pixels (R + 2) * (W + 1) + 1
    int npieces;         // DMA pieces per patch buffer
    int patch_bytes;     // npieces * 1024
    int n_tiles;
    LazyIn in;
    int opt;             // CONV8_OPT (experiments): 1 no raised priority for the second workgroup, 2 the second workgroup starts half a tile late
    unsigned long long* trace;
};

int g_enable8 = -1, g_min_tiles8 = -1;
unsigned long long* g_trace8 = nullptr;

#ifdef CLHIP_ABLATION
#define STAMP8() do { if (p.trace && (blockIdx.x == 0 || blockIdx.x == 256) && lane == 0 && nstamp < 62) p.trace[((blockIdx.x ? 4 : 0) + wv) * 64 + nstamp++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP8() do { } while (0)
#endif

__device__ __forceinline__ void wait_vm0_8() { __builtin_amdgcn_s_waitcnt(0x0070 | 0xF00); }          // vmcnt(0), lgkmcnt / expcnt untouched
template <int N> __device__ __forceinline__ void wait_vm8() { __builtin_amdgcn_s_waitcnt((N & 15) | 0x70 | 0xF00 | ((N >> 4) << 14)); }
__device__ __forceinline__ void wait_lds8() { __builtin_amdgcn_s_waitcnt(0xC07F); }                  // lgkmcnt(0)
__device__ __forceinline__ void wg_barrier8() {
    wait_lds8();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// OR over the 8 lanes of an aligned lane group (DPP: quad xor 1, quad xor 2, half-row mirror)
__device__ __forceinline__ unsigned or8(unsigned v) {
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);
    return v;
}

template <int MODE, int XF, int BNR>      // BNR (dgrad): 0 no sums; the producer's ReLU mask comes 1 from its packed bits, 2 from z' (scale / shift), 3 from its activation, 4 there is no ReLU
__global__ __launch_bounds__(256, 2) void conv8_kernel(const Conv8Params p) {
    static_assert(XF == 0 || MODE == 0, "lazy inputs exist in the forward only");
    static_assert(BNR == 0 || MODE == 1, "BatchNorm-backward sums belong to the dgrad");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ph = wv >> 1, jj = wv & 1;                    // pixel half (64 pixels), output-channel half (32 channels)
    const int l31 = lane & 31, kh = lane >> 5;
    const int W = p.W, PW = W + 1;
    const int PB = p.patch_bytes;
    int nstamp = 0; (void)nstamp;
    STAMP8();
#ifdef CLHIP_ABLATION
    if (p.trace && (blockIdx.x == 0 || blockIdx.x == 256) && lane == 0)
        p.trace[((blockIdx.x ? 4 : 0) + wv) * 64 + 63] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492);   // XCC_ID, HW_ID
#endif

    // ---- LDS map: three patch buffers | statistics rows [2 parities][4 half-wave rows][2][64] | coefficient table [2][64]
    float* red0 = reinterpret_cast<float*>(smem + 3 * PB);
    float* coefs = red0 + 2 * 4 * 2 * C8;

    // workgroup -> tiles: every XCD walks a contiguous range (block b runs on XCD b % 8: halo rows shared with the neighbours hit that XCD's L2)
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
    if (nmy <= 0) return;               // (a lazy input's by-products are workgroup 0's, which always has a tile)
    // the second workgroup of a CU (dispatch order: usually b and b + 256) runs at raised priority: on a SIMD shared by two MFMA loops the
    // pipe alternates anyway, with a priority the two workgroups fall out of step at once instead of marching through the same phases together
    if (((blockIdx.x / 256) & 1) && !(p.opt & 1)) __builtin_amdgcn_s_setprio(1);

    // ---- the filter rows of this wave's 32 output channels -> registers (MFMA A operand: row o = jj*32 + l31, K-step ks = channels ks*16 + kh*8 ..)
    bf16x8_t wr[9][4];
    constexpr int WPITCH = 73 * 16;                              // bytes per staged filter row (72 data slots + 1 pad: conflict-free)
    constexpr int WINST8 = (32 * 73 + 63) / 64;                  // 37 DMA instructions per 32 rows
    constexpr int WPW = (WINST8 + 3) / 4;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.wt), 0, C8 * 9 * C8 * 2, 0x00020000);
    auto wdma = [&](int j) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            const int I = wv * WPW + i;
            const int n = I * 64 + lane;
            const int row = n / 73, sub = n - row * 73;
            const int off = (row < 32 && sub < 72) ? ((j * 32 + row) * 72 + sub) * 16 : OOB8;
            if (I < WINST8) __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lvoid_t*)(smem + I * 1024), 16, off, 0, 0, 0);
        }
#else
        (void)j;
#endif
    };
    auto wread = [&]() {
        const char* wl = smem + l31 * WPITCH + kh * 16;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wr[t][ks] = *reinterpret_cast<const bf16x8_t*>(wl + t * 128 + ks * 32);
    };

    // ---- patch DMA lanes.  Piece I = 4 i + wave fills LDS bytes [I * 1024, +1024) of a buffer: slot n = I * 64 + lane is patch pixel lp = n / 8
    //      (row lp / PW, column lp % PW; column 0 = the pad), swizzled slot n % 8 = channel chunk ^ ((lp >> 1) & 7).  prel = byte offset relative
    //      to the tile's first pixel (negative in the halo row above), low bits: 1 = top halo row, 2 = bottom halo row
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.src), 0, p.M * C8 * 2, 0x00020000);
    int prel[PINST8];
#pragma unroll
    for (int i = 0; i < PINST8; ++i) {
        const int I = i * 4 + wv;
        const int n = I * 64 + lane;
        const int lp = n >> 3, sp = n & 7;
        const int r = lp / PW, c = lp - r * PW;
        const int s = sp ^ ((lp >> 1) & 7);
        int v = OOB8;
        if (I < p.npieces && lp < p.np - 1 && c != 0) v = (((r - 1) * W + (c - 1)) * C8 * 2 + s * 16) | (r == 0 ? 1 : 0) | (r == p.R + 1 ? 2 : 0);
        prel[i] = v;
    }
    auto dma_from = [&](const __amdgpu_buffer_rsrc_t& rs, int tile, char* b) {
        const int base = tile * BM8 * (C8 * 2);
        const int ti = tile % p.tiles_per_img;
        const int kill = (ti == 0 ? 1 : 0) | (ti == p.tiles_per_img - 1 ? 2 : 0);       // halo rows outside the image: zeros
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int i = 0; i < PINST8; ++i) {
            const int I = i * 4 + wv;
            const int off = (prel[i] == OOB8 || (prel[i] & kill)) ? OOB8 : (prel[i] & ~3) + base;
            if (I < p.npieces) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lvoid_t*)(b + I * 1024), 16, off, 0, 0, 0);
        }
#else
        (void)base; (void)kill; (void)b; (void)rs;
#endif
    };
    auto pdma = [&](int tile, char* b) { dma_from(srs, tile, b); };
    // dgrad with the producer's BatchNorm-backward sums: z' of THIS wave's 64 pixels x 32 channels (64 bytes per pixel) -> buffer 0, four pieces per
    // wave, slot n = 64 i + lane = (pixel n / 4, swizzled 16-byte slot n % 4 = chunk ^ ((pixel >> 2) & 3)).  Nobody else reads them: the wave's own
    // vmcnt wait is all the synchronisation they need.  (The first version gathered z' / y' from global memory in 8-byte pieces, 32 cache lines per
    // load instruction: the texture path became the limit and the ResNet-18 step went from 2.02 to 2.19 ms.)
    const __amdgpu_buffer_rsrc_t zrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(BNR != 0 ? p.bn_z : p.src), 0, p.M * C8 * 2, 0x00020000);
    int zrel[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = i * 64 + lane, pl = n >> 2, sl = n & 3;
        zrel[i] = (ph * 64 + pl) * (C8 * 2) + (jj * 4 + (sl ^ ((pl >> 2) & 3))) * 16;
    }
    auto zdma = [&](int tile) {
        if constexpr (BNR != 0) {
            const int base = tile * BM8 * (C8 * 2);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
            for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(zrs, (lvoid_t*)(smem + wv * 4096 + i * 1024), 16, zrel[i] + base, 0, 0, 0);
#else
            (void)base;
#endif
        }
    };
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(XF == 2 ? p.in.res : p.src), 0, p.M * C8 * 2, 0x00020000);
    auto rdma = [&](int tile) { if constexpr (XF == 2) dma_from(rrs, tile, smem); };         // the residual's patch: always buffer 0

    // ---- lazy input: the slots THIS wave's DMA pieces landed are rewritten in place (legal right after the wave's own vmcnt wait; the tile
    //      barrier publishes the result); a lane's pieces hold two channel chunks only (4 i + wave: (lp >> 1) & 7 alternates with i)
    auto transform = [&](int tile, char* b) {
        if constexpr (XF != 0) {
            const int ti = tile % p.tiles_per_img;
            const int kill = (ti == 0 ? 1 : 0) | (ti == p.tiles_per_img - 1 ? 2 : 0);
            const size_t base = (size_t)tile * BM8 * C8;
#pragma unroll
            for (int i = 0; i < PINST8; ++i) {
                const int I = i * 4 + wv;
                if (I >= p.npieces) continue;
                const int pr = prel[i];
                if (pr == OOB8 || (pr & kill)) continue;             // pad slot / outside the image: the DMA wrote zeros, and zeros they stay
                char* l = b + I * 1024 + lane * 16;
                const int rel = pr >> 4;                              // (pixel relative to the tile) * 8 + chunk, floor division keeps the halo row's sign
                const int sub = rel & 7, pix = rel >> 3;
                const uint4 v = *reinterpret_cast<const uint4*>(l);
                float sc[8], sh[8];
                *reinterpret_cast<f32x4*>(sc) = *reinterpret_cast<const f32x4*>(coefs + sub * 8);
                *reinterpret_cast<f32x4*>(sc + 4) = *reinterpret_cast<const f32x4*>(coefs + sub * 8 + 4);
                *reinterpret_cast<f32x4*>(sh) = *reinterpret_cast<const f32x4*>(coefs + C8 + sub * 8);
                *reinterpret_cast<f32x4*>(sh + 4) = *reinterpret_cast<const f32x4*>(coefs + C8 + sub * 8 + 4);
                uint4 o;
                unsigned mk = 0;
                if constexpr (XF == 2) o = bn_res_relu8_bf16(v, *reinterpret_cast<const uint4*>(smem + I * 1024 + lane * 16), sc, sh, mk);
                else o = bn_relu8_bf16(v, sc, sh);
                *reinterpret_cast<uint4*>(l) = o;
                const bool own = pix >= 0 && pix < BM8;              // a pixel of this workgroup's tile: the activation's one writer
                if (own) *reinterpret_cast<uint4*>(p.in.y + base + (size_t)pix * C8 + sub * 8) = o;
                if (XF == 2 && p.in.mask != nullptr) {
                    // the 8 lanes of a pixel hold its 8 mask bytes: one 8-byte store per pixel (byte `sub` of the word)
                    const unsigned lo = or8(sub < 4 ? mk << (8 * sub) : 0u), hi = or8(sub >= 4 ? mk << (8 * (sub - 4)) : 0u);
                    if (own && (lane & 7) == 0) *reinterpret_cast<uint2*>(p.in.mask + (base >> 3) + (size_t)pix * 8) = make_uint2(lo, hi);
                }
            }
        }
    };

    // ---- prologue: patch of the first tile -> buffer 2, filter rows -> buffers 0-1 -> registers, one channel half at a time
    // (r05 trace: ~5000 cycles until these 17 pieces are issued -- the L1 -> LDS path moves ~50 bytes per cycle and CU, and a CU stages the 74 KB of
    //  filters twice -- then ~1000 per wait / fragment-read round; counted waits that keep the patch in flight under the rounds changed nothing)
    wdma(0);
    pdma(t_first, smem + 2 * PB);
    STAMP8();
    if constexpr (XF != 0) {
        // scale / shift of the producer's BatchNorm from its fp64 sums while the first DMAs fly (fp64 scratch: the statistics rows, idle until the first epilogue)
        lazy_in_coefs(p.in, C8, coefs, reinterpret_cast<double*>(red0), blockIdx.x == 0);
    }
    if constexpr (BNR == 2) {
        if (tid < C8) {              // the producer's scale / shift, as its forward formed them (published by the barriers below)
            const float sc_c = p.bn_gamma[tid] * p.bn_invstd[tid];
            coefs[tid] = sc_c;
            coefs[C8 + tid] = p.bn_beta[tid] - p.bn_mean[tid] * sc_c;
        }
    }
    wait_vm0_8();
    STAMP8();
    wg_barrier8();
    if (jj == 0) wread();
    wg_barrier8();
    STAMP8();
    wdma(1);
    wait_vm0_8();
    STAMP8();
    wg_barrier8();
    if (jj == 1) wread();
    wg_barrier8();
    STAMP8();
    if constexpr (XF == 2) { rdma(t_first); wait_vm0_8(); }
    transform(t_first, smem + 2 * PB);
    wg_barrier8();
    STAMP8();

    // ---- fragment bases: output pixel i*32 + l31 of this wave's half sits at patch pixel (row + 1) * PW + col + 1
    int lpo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = ph * 64 + i * 32 + l31;
        const int r = q / W, c = q - r * W;
        lpo[i] = (r + 1) * PW + c + 1;
    }
    const int kh2 = kh << 1;
    // per-lane running sums over this workgroup's tiles (forward: z, z^2; dgrad + BNR: g, g z'), reduced over pixels ONCE after the last tile: the
    // 128 DPP adds of the 16-lane reduction were most of a tile's epilogue (r05 phase trace: 2700 cycles per tile with them, 900 without)
    float sv[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) sv[r] = 0.f;
    if (((blockIdx.x / 256) & 1) && (p.opt & 2)) __builtin_amdgcn_s_sleep(40);

    for (int k = 0; k < nmy; ++k) {
        const int tile = t_first + k * t_step;
        const char* pb = smem + ((k & 1) ? 1 : 2) * PB;                  // the first tile sits in buffer 2
        char* nb = smem + ((k & 1) ? 2 : 1) * PB;
        if (k + 1 < nmy) { pdma(tile + t_step, nb); rdma(tile + t_step); }       // lands under this tile's MFMAs
        zdma(tile);

        STAMP8();
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        // (the fragment addresses do not depend on the tile: left alone, hipcc hoists all 72 of them out of this loop and spills)
        asm volatile("" : "+v"(lpo[0]), "+v"(lpo[1]));

        // 36 steps (tap t, K-step ks); the two fragments of step n + 1 are requested before the two MFMAs of step n are issued
        bf16x8_t xf[2][2];
        int fb[2];                                              // swizzled fragment base of the current tap
        auto frag = [&](int n, bf16x8_t (&f)[2]) {
            const int t = n >> 2, ks = n & 3;
            if (ks == 0) {
                const int r = t / 3, s = t - 3 * r;
                const int shift = MODE == 0 ? (r - 1) * PW + (s - 1) : (1 - r) * PW + (1 - s);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int lp = lpo[i] + shift;
                    fb[i] = (lp << 7) | ((((lp ^ kh2) >> 1) & 7) << 4);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) f[i] = *reinterpret_cast<const bf16x8_t*>(pb + (fb[i] ^ (ks << 5)));
        };
        frag(0, xf[0]);
#pragma unroll
        for (int n = 0; n < 36; ++n) {
            const int t = n >> 2, ks = n & 3;
            if (n + 1 < 36) frag(n + 1, xf[(n + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[t][ks], xf[n & 1][i], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        STAMP8();
        wait_vm0_8();                                           // this wave's part of the next patch has landed
        STAMP8();

        // ---- epilogue: D[row = channel (r&3) + 8*(r>>2) + 4*kh][col = pixel l31]
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const size_t pix = (size_t)tile * BM8 + ph * 64 + i * 32 + l31;
            bf16_t* drow = p.dst + pix * C8 + jj * 32;
            if (MODE == 1 && p.accumulate) {
                // the old gradient in the STORE pattern (16 bytes = 8 consecutive channels per lane, 32 contiguous bytes per pixel and instruction; the
                // 8-byte form touched 32 cache lines per instruction for 16 bytes each), the half-waves exchange the halves they hold for each other
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const u32x4 old = *reinterpret_cast<const u32x4*>(drow + pr * 16 + kh * 8);
                    auto s0 = __builtin_amdgcn_permlane32_swap(old[0], old[2], false, false);
                    auto s1 = __builtin_amdgcn_permlane32_swap(old[1], old[3], false, false);
                    // lanes 0-31: {own d0, partner's d0}, {own d1, partner's d1} = channels 16 pr + {0..3} and 16 pr + 8 + {0..3};
                    // lanes 32-63: {partner's d2, own d2}, ... = channels 16 pr + 4 + {0..3} and 16 pr + 12 + {0..3}
                    const unsigned a0 = s0[0], a1 = s1[0], b0 = s0[1], b1 = s1[1];
                    acc[i][8 * pr + 0] += __uint_as_float(a0 << 16); acc[i][8 * pr + 1] += __uint_as_float(a0 & 0xffff0000u);
                    acc[i][8 * pr + 2] += __uint_as_float(a1 << 16); acc[i][8 * pr + 3] += __uint_as_float(a1 & 0xffff0000u);
                    acc[i][8 * pr + 4] += __uint_as_float(b0 << 16); acc[i][8 * pr + 5] += __uint_as_float(b0 & 0xffff0000u);
                    acc[i][8 * pr + 6] += __uint_as_float(b1 << 16); acc[i][8 * pr + 7] += __uint_as_float(b1 & 0xffff0000u);
                }
            }
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                unsigned ax = pack_bf16x2(acc[i][8 * pr + 0], acc[i][8 * pr + 1]), ay = pack_bf16x2(acc[i][8 * pr + 2], acc[i][8 * pr + 3]);
                unsigned bx = pack_bf16x2(acc[i][8 * pr + 4], acc[i][8 * pr + 5]), by = pack_bf16x2(acc[i][8 * pr + 6], acc[i][8 * pr + 7]);
                auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                *reinterpret_cast<u32x4*>(drow + pr * 16 + kh * 8) = u32x4{rx[0], ry[0], rx[1], ry[1]};
            }
            if constexpr (MODE == 1 && BNR) {
                // sum g, sum g * z' of the producer with g = dy * (y' > 0), from the fp32 results; z' out of this wave's LDS tile
                const int pl = i * 32 + l31;
                const char* zl = smem + wv * 4096 + pl * 64 + kh * 8;
                int cofs = 0;                                    // (opaque: the table reads below are tile-invariant, hoisted they cost 32 registers and spill)
                asm volatile("" : "+v"(cofs));
                unsigned mbits = 0xffffffffu;                    // packed ReLU mask of this lane's pixel: the four chunks of this wave's channels
                if constexpr (BNR == 1) mbits = *reinterpret_cast<const unsigned*>(p.bn_mask + pix * 8 + jj * 4) >> (4 * kh);
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const uint2 zz = *reinterpret_cast<const uint2*>(zl + ((g4 ^ ((pl >> 2) & 3)) << 4));
                    const float z4[4] = {__uint_as_float(zz.x << 16), __uint_as_float(zz.x & 0xffff0000u), __uint_as_float(zz.y << 16), __uint_as_float(zz.y & 0xffff0000u)};
                    float y4[4] = {1.f, 1.f, 1.f, 1.f};
                    if constexpr (BNR == 2) {
                        const int cc = jj * 32 + g4 * 8 + kh * 4;
                        const f32x4 sc = *reinterpret_cast<const f32x4*>(coefs + cofs + cc), sh = *reinterpret_cast<const f32x4*>(coefs + cofs + C8 + cc);
#pragma unroll
                        for (int e = 0; e < 4; ++e) y4[e] = fmaf(z4[e], sc[e], sh[e]);
                    } else if constexpr (BNR == 3) {
                        const uint2 yy = *reinterpret_cast<const uint2*>(p.bn_y + pix * C8 + jj * 32 + kh * 4 + g4 * 8);
                        y4[0] = __uint_as_float(yy.x << 16); y4[1] = __uint_as_float(yy.x & 0xffff0000u); y4[2] = __uint_as_float(yy.y << 16); y4[3] = __uint_as_float(yy.y & 0xffff0000u);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool on = BNR == 1 ? ((mbits >> (8 * g4 + e)) & 1u) != 0u : y4[e] > 0.f;
                        const float g = on ? acc[i][4 * g4 + e] : 0.f;
                        sv[4 * g4 + e] += g;
                        sv[16 + 4 * g4 + e] = fmaf(g, z4[e], sv[16 + 4 * g4 + e]);
                    }
                }
            }
        }
        if constexpr (MODE == 0) {
            if (p.stat_acc != nullptr) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float a = acc[0][r], b = acc[1][r];
                    sv[r] += a + b;
                    sv[16 + r] = fmaf(a, a, fmaf(b, b, sv[16 + r]));
                }
            }
        }
        STAMP8();
        if (k + 1 < nmy) transform(tile + t_step, nb);          // this wave's slots of the next patch: raw z' -> the activation, in place
        // one barrier per tile: every wave is done reading this tile's patch buffer (the DMA of tile k + 2 overwrites it), every wave's part of the
        // next patch has landed (waited for above) and is transformed
        wg_barrier8();
        STAMP8();
    }
    // ---- the sums: 16-lane reduction, the four half-wave rows through LDS, one fp64 atomic per (which, channel) and workgroup
    const bool stats = (MODE == 0 && p.stat_acc != nullptr) || (MODE == 1 && BNR != 0);
    double run = 0.0;
    if (stats) {
        row16_sum_n(sv);
        if ((lane & 15) == 0) {
            const int rh = (lane >> 4) & 1;
#pragma unroll
            for (int which = 0; which < 2; ++which)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int cc2 = jj * 32 + g4 * 8 + kh * 4;
                    const float* v = sv + which * 16 + g4 * 4;
                    *reinterpret_cast<f32x4*>(red0 + ((ph * 2 + rh) * 2 + which) * C8 + cc2) = f32x4{v[0], v[1], v[2], v[3]};
                }
        }
        wg_barrier8();
        if (tid < 2 * C8) {
            const int which = tid >> 6, c2 = tid & 63;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) run += (double)red0[(w2 * 2 + which) * C8 + c2];
        }
    }
    // ---- the sums leave the workgroup once
    if constexpr (MODE == 0) {
        if (p.stat_acc != nullptr && tid < 2 * C8)
            atomicAdd(p.stat_acc + ((size_t)(blockIdx.x & (p.stat_rep - 1)) * 2 + (tid >> 6)) * C8 + (tid & 63), run);
    } else if constexpr (BNR != 0) {
        // sum g * xhat = invstd * (sum g z' - mean * sum g), in fp64
        double* ex = reinterpret_cast<double*>(red0 + 4 * 2 * C8);      // (the second half of the statistics area)
        if (tid < C8) ex[tid] = run;
        wg_barrier8();
        if (tid < 2 * C8) {
            const int which = tid >> 6, c2 = tid & 63;
            double v = run;
            if (which == 1) v = (double)p.bn_invstd[c2] * (run - (double)p.bn_mean[c2] * ex[c2]);
            atomicAdd(p.bn_acc + ((size_t)(blockIdx.x & (p.bn_rep - 1)) * 2 + which) * C8 + c2, v);
        }
    }
    STAMP8();
}

bool geometry8(int N, int H, int W, Conv8Params& p) {
    if (W < 8 || W > 32 || BM8 % W != 0) return false;
    p.R = BM8 / W;
    if (H % p.R != 0) return false;
    p.H = H; p.W = W; p.M = N * H * W;
    p.tiles_per_img = H / p.R;
    p.np = (p.R + 2) * (W + 1) + 1;
    p.npieces = (p.np * 128 + 1023) / 1024;
    if (p.npieces > 4 * PINST8) return false;
    p.patch_bytes = p.npieces * 1024;
    if (2 * p.patch_bytes < 37 * 1024) return false;            // the filter staging area of the prologue
    p.n_tiles = p.M / BM8;
    return true;
}

size_t lds8(const Conv8Params& p) { return (size_t)3 * p.patch_bytes + 2 * 4 * 2 * C8 * sizeof(float) + 2 * C8 * sizeof(float); }

template <int MODE, int XF, int BNR>
int launch8(Conv8Params& p, hipStream_t st) {
    const size_t lds = lds8(p);
    if (lds > 80 * 1024) { clhip_set_error("conv8: %zu bytes of LDS", lds); return CLHIP_EINVAL; }
    auto kern = conv8_kernel<MODE, XF, BNR>;
    int dev = 0;
    (void)hipGetDevice(&dev);
    static size_t attr[16] = {0};
    if (dev < 0 || dev >= 16 || lds > attr[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            clhip_set_error("conv8: cannot reserve %zu bytes of LDS", lds);
            return CLHIP_EHIP;
        }
        if (dev >= 0 && dev < 16) attr[dev] = lds;
    }
    p.opt = clhip_cfg("CONV8_OPT") ? atoi(clhip_cfg("CONV8_OPT")) : 0;
    static const int force_grid = clhip_cfg("CONV8_GRID") ? atoi(clhip_cfg("CONV8_GRID")) : 0;
    int grid = force_grid > 0 ? force_grid : 512;
    if (grid > p.n_tiles) grid = p.n_tiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, p);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

}  // namespace

bool clhip_conv8_supported(int N, int H, int W, int Cs, int Cd, int ksize, int stride, int pad, int dtype) {
    static const bool on_env = clhip_cfg("CONV8") ? atoi(clhip_cfg("CONV8")) != 0 : true;
    if (g_enable8 >= 0 ? g_enable8 == 0 : !on_env) return false;
    if (!(dtype == CLHIP_BF16 && ksize == 3 && stride == 1 && pad == 1 && Cs == C8 && Cd == C8 && N >= 1)) return false;
    Conv8Params p;
    if (!geometry8(N, H, W, p)) return false;
    if ((int64_t)p.M * C8 * 2 >= ((int64_t)1 << 29)) return false;          // the out-of-range marker of the patch DMA is a 1 GiB offset
    // two tiles per resident workgroup at least: below that the filter staging of the prologue does not pay (conv5 / conv4 / conv64 take over)
    static const int min_tiles_env = clhip_cfg("CONV8_MIN_TILES") ? atoi(clhip_cfg("CONV8_MIN_TILES")) : 1024;
    return p.n_tiles >= (g_min_tiles8 >= 0 ? g_min_tiles8 : min_tiles_env);
}

void clhip_conv8_enable(int on) { g_enable8 = on; }
void clhip_conv8_min_tiles(int n) { g_min_tiles8 = n; }
void clhip_conv8_set_trace(unsigned long long* dev_buf) { g_trace8 = dev_buf; }
int clhip_conv8_tiles_m(int M) { return (M + BM8 - 1) / BM8; }

// mode 0: forward (stat_acc may be nullptr); mode 1: dgrad, with the producer's BatchNorm-backward sums when bn_z != nullptr.
// in != nullptr (forward only): src is the producer's pre-BatchNorm output, the operand relu(bn(src) [+ in->res]) is formed in LDS and written to in->y
int clhip_conv8_launch(const void* src, const void* wt, void* dst, double* stat_acc, int stat_rep, int N, int H, int W, int accumulate, int mode, const LazyIn* in,
                       const void* bn_z, const void* bn_y, const void* bn_mask, const float* bn_gamma, const float* bn_beta, const float* bn_mean, const float* bn_invstd,
                       double* bn_acc, int bn_rep, hipStream_t st) {
    Conv8Params p;
    p.bn_gamma = bn_gamma; p.bn_beta = bn_gamma != nullptr ? bn_beta : nullptr;
    if (!geometry8(N, H, W, p)) { clhip_set_error("conv8: unsupported geometry %d x %d x %d", N, H, W); return CLHIP_EINVAL; }
    if (in != nullptr) {
        if (mode != 0 || in->acc == nullptr || in->y == nullptr) { clhip_set_error("conv8: a lazy input needs the forward mode, the producer's sums and an output activation"); return CLHIP_EINVAL; }
        p.in = *in;
    }
    p.src = static_cast<const bf16_t*>(src); p.wt = static_cast<const bf16_t*>(wt); p.dst = static_cast<bf16_t*>(dst);
    p.stat_acc = stat_acc; p.stat_rep = stat_rep > 0 ? stat_rep : 1; p.accumulate = accumulate;
    p.bn_z = static_cast<const bf16_t*>(bn_z); p.bn_y = static_cast<const bf16_t*>(bn_y); p.bn_mask = static_cast<const unsigned char*>(bn_mask);
    p.bn_mean = bn_mean; p.bn_invstd = bn_invstd; p.bn_acc = bn_acc; p.bn_rep = bn_rep > 0 ? bn_rep : 1;
    p.trace = g_trace8;
    if (mode == 0) {
        if (in == nullptr) return launch8<0, 0, 0>(p, st);
        return in->res != nullptr ? launch8<0, 2, 0>(p, st) : launch8<0, 1, 0>(p, st);
    }
    if (bn_z == nullptr) return launch8<1, 0, 0>(p, st);
    if (bn_mask != nullptr) return launch8<1, 0, 1>(p, st);
    if (bn_gamma != nullptr) return launch8<1, 0, 2>(p, st);
    return bn_y != nullptr ? launch8<1, 0, 3>(p, st) : launch8<1, 0, 4>(p, st);
}
