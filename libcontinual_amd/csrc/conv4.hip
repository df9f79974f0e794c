// conv4.hip -- fourth-generation 3x3 / stride 1 / pad 1 convolution (forward and dgrad), bf16, gfx950.
//
// What the round-1 profiles said about conv3.hip (profiles/r01_conv3_ablation_and_pmc.txt): 2500 VALU + 1000 SALU
// instructions per wave against 288 MFMAs, a weight prefetch distance of one tap (~0.25 us of MFMAs against a 1-2 us L2 /
// HBM round trip), an LDS-staged epilogue behind two workgroup barriers, and load -> compute -> store phases that add up
// because every workgroup of a launch walks through them together.  This kernel is built the other way round:
//
//  * v_mfma_f32_32x32x16_bf16: half the issue slots per FLOP of the 16x16x32 form and the higher measured peak; a wave
//    owns a 64 pixel x 64 channel tile = 2 x 2 MFMA tiles, 64 accumulator registers.
//  * the weights of a (tap, CK-channel chunk) slab are streamed by LDS-DMA (global_load_lds_dwordx4: no staging registers,
//    no VALU) into a 3-stage ring TWO slabs ahead of the MFMAs; waits are counted (s_waitcnt vmcnt(N)), the barrier is the
//    raw s_barrier (a __syncthreads() would drain the DMA queue).  The DMA writes lane-linearly, so the ring rows keep
//    their natural 2*CK-byte pitch and the XOR swizzle that makes the 32-row ds_read_b128 fragment reads conflict-free is
//    applied to the per-lane SOURCE address (guide rule 21) and, precomputed, to the read address.
//  * the input patch  pixels [m0 - W - 1, m0 + BM + W + 1) x CK channels  is staged through registers with buffer loads --
//    out-of-tensor pixels are zero-filled by the hardware range check, no per-lane compares -- into a pitch of 2*CK + 16
//    bytes (an odd number of 16-byte slots: conflict-free for 32 consecutive pixels at ANY tap shift), double-buffered:
//    chunk c+1 (or the first chunk of the workgroup's next tile) is in flight while chunk c is multiplied.
//  * workgroups are persistent: each walks a static list of (pixel tile, channel tile) items, the weight ring and the patch
//    prefetch run across item boundaries, so only the first patch load of a workgroup is exposed.
//  * the epilogue needs no LDS and no barrier: v_permlane32_swap pairs the two half-waves' 8-byte channel groups into
//    16-byte stores straight from the accumulators (guide T21).  BatchNorm statistics come from the fp32 accumulators.
//  * small-M layers (8x8, 4x4 images) split the reduction over KG wave groups inside the workgroup (each group owns a
//    range of channel chunks, its own patch buffers and ring slots; the partial tiles are summed through LDS at the end),
//    so that 256+ workgroups exist without shrinking the wave tile.
//
// MODE 0 = forward (tap (r,s) reads pixel (h+r-1, w+s-1), weights [Cd][9][Cs]); MODE 1 = dgrad (mirrored taps, the dgrad
// weight copy [C][9][K] has the same layout).  Replaces nn.Conv2d forward / input gradient of the reference ResNets
// (core/model/backbone/resnet.py:17-24, 295-298, 337, 367).
#include <stdlib.h>

#include "common.h"

namespace {

int g_force4[4] = {0, 0, 0, 0};   // tuning hooks, see the end of the file
int g_enable4 = -1, g_debug4 = 0;
unsigned long long* g_trace4 = nullptr;

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

struct Conv4Params {
    const bf16_t* src;   // [N,H,W,Cs]
    const bf16_t* wt;    // [Cd][9][Cs]
    bf16_t* dst;         // [N,H,W,Cd]
    float* stats;        // per-tile partial rows [tile][2][Cd] or nullptr
    double* stat_acc;    // [stat_rep][2][Cd] fp64 accumulators or nullptr
    int stat_rep;
    // dgrad only: BatchNorm-backward sums of the layer that PRODUCED the tensor whose gradient this launch completes
    // (sum g and sum g * xhat per channel, g = dy masked by the ReLU), accumulated from the fp32 results in the epilogue
    const bf16_t* bn_z;      // [N,H,W,Cd] pre-BatchNorm output of that layer (xhat = (z - mean) * invstd), or nullptr: no reduction
    const bf16_t* bn_y;      // its post-activation output for the ReLU mask (y > 0), or nullptr: no ReLU
    const float* bn_mean;
    const float* bn_invstd;
    double* bn_acc;          // [bn_rep][2][Cd]
    int bn_rep;
    int H, W, Cs, Cd, M, accumulate;
    int wshift, hshift;
    int np;              // patch pixels = BM + 2W + 2
    int patch_bytes;     // one patch buffer: whole DMA instructions covering (np + 1) pixels (the extra row is the zero row)
    int n_ntiles, nt_shift, n_items;
    int cpg;             // channel chunks per K group
    unsigned long long* trace;   // CLHIP_ABLATION builds: s_memtime stamps of waves 0 and 4 of workgroup 0 ([2][256])
    int debug;           // ablation switches (tools/ubench/conv_bench): 1 no MFMA, 2 no weight DMA, 4 no patch loads, 8 no stores, 16 no statistics, 32 no barriers, 64 no fragment reads
};

template <int MODE>
__device__ __forceinline__ unsigned tap_mask4(int g, const Conv4Params& p) {
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

#ifdef CLHIP_ABLATION
#define DBG(p) ((p).debug)
#define STAMP() do { if (p.trace && blockIdx.x == 0 && lane == 0 && (wave & 3) == 0 && nstamp < 256) p.trace[(wave >> 2) * 256 + nstamp++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP() do { } while (0)
#define DBG(p) 0          // the production build carries no ablation branches (they split the software pipeline's basic blocks)
#endif

// Waits go through __builtin_amdgcn_s_waitcnt, which hipcc's own wait-count bookkeeping understands: an inline-asm s_waitcnt is
// invisible to it, and it then re-waits lgkmcnt(0) in front of the first MFMA that follows a batch of LDS reads -- exactly the
// round trip the software pipeline exists to hide (profiles/r02_conv4_notes.md).  gfx9 encoding of the immediate:
// vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14.
template <int N> __device__ __forceinline__ void wait_vm() { __builtin_amdgcn_s_waitcnt((N & 15) | 0x70 | 0xF00 | ((N >> 4) << 14)); }
__device__ __forceinline__ void wait_lds() { __builtin_amdgcn_s_waitcnt(0xC07F); }     // lgkmcnt(0), vmcnt / expcnt untouched
__device__ __forceinline__ void wg_barrier(bool skip = false) {
    wait_lds();
    if (!skip) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// sum over the 32 lanes that share lane >> 5 is left to the caller: this is the 16-lane DPP part (see common.h)
template <int WM, int WN, int KG, int CK, int WMAX, int MODE>
__global__ __launch_bounds__(WM * WN * KG * 64, 2) void conv4_kernel(const Conv4Params p) {
    constexpr int BM = WM * 64, BN = WN * 64, NWG = WM * WN, NTG = NWG * 64;
    constexpr int NST = 3;                        // ring stages: 9 taps per chunk -> stage = tap % 3 at compile time
    constexpr int PP = CK * 2 + 16;               // patch pitch (bytes per pixel)
    constexpr int KS = CK / 16;                   // MFMA K steps per slab
    constexpr int WROW = CK * 2;                  // ring row bytes
    constexpr int SPR = WROW / 16;                // 16-byte slots per ring row (8 / 4)
    constexpr int RPI = 64 / SPR;                 // ring rows per DMA instruction
    constexpr int SLAB = BN * WROW;               // bytes per (stage, group)
    constexpr int WINST = (BN / RPI) / NWG;       // DMA instructions per wave per slab
    static_assert((BN / RPI) % NWG == 0 && WINST >= 1, "ring rows must divide among the waves of a group");
    constexpr int SPP = PP / 16;                  // 16-byte slots per patch pixel, pad slot included (9 / 5)
    constexpr int ZBYTES = 512;                   // zero area behind the patch (see zoff)
    constexpr int PINST = ((((BM + 2 * WMAX + 2) * PP + 255) / 256 * 256 + ZBYTES) / 1024 + 1 + NWG - 1) / NWG;   // patch DMA instructions per wave (W <= WMAX, zero area included)
    constexpr int PL_TAP = 2;                     // tap at which the next patch's DMA is issued
    constexpr bool SETPRIO = false;               // raised MFMA priority starves the partner wave's read phase (VALU address math): 800 -> ? cycles
    constexpr bool STAG = NWG * KG == 8;          // two waves per SIMD inside one workgroup: staggered read / MFMA phases

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave / NWG, wv = wave % NWG;             // K group, wave within the group
    int nstamp = 0; (void)nstamp;
    const int wm = wv / WN, wn = wv % WN;
    const int l31 = lane & 31, kh = lane >> 5;
    const int W = p.W, Cs = p.Cs, halo = W + 1;
    const int K9 = 9 * Cs;

    // ---- LDS map: [group][2] patch buffers | [stage][group] ring | statistics scratch
    char* patch_g = smem + grp * 2 * p.patch_bytes;
    char* ring0 = smem + KG * 2 * p.patch_bytes;
    char* ring_g = ring0 + grp * SLAB;                       // + stage * KG * SLAB
    // statistics scratch [WM*2][2][BN]: behind the ring and, with K groups, behind the partial-tile exchange area as well
    constexpr int XR_BYTES = KG > 1 ? (KG - 1) * NWG * 64 * 64 * 4 : 0;
    const int ring_end = KG * 2 * p.patch_bytes + NST * KG * SLAB;
    float* red = reinterpret_cast<float*>(smem + (ring_end > XR_BYTES ? ring_end : XR_BYTES));

    // ---- B (pixel) fragment addresses: tile i of this wave = pixels wm*64 + i*32 + l31; k half = lane >> 5
    int xaddr[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) xaddr[i] = (wm * 64 + i * 32 + l31 + halo) * PP + kh * 16;
    // Out-of-image taps read zeros.  A masked lane is redirected to the zero area at the SAME offset within the 256-byte bank row as
    // its regular address, so a fragment read stays conflict-free whatever the mask (one shared zero row cost 22 % extra LDS cycles).
    const int zoff = (p.np * PP + 255) / 256 * 256;
    // ---- A (weight) fragment addresses in a ring slab: row o = wn*64 + j*32 + l31, chunk 2*ks + kh, XOR-swizzled
    int waddr[2][KS];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int o = wn * 64 + j * 32 + l31;
        const int swz = CK == 64 ? ((o >> 1) & 7) : ((o >> 2) & 3);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) waddr[j][ks] = o * WROW + (((2 * ks + kh) ^ swz) << 4);
    }
    // ---- DMA: instruction i of this wave fills ring rows [(wv*WINST + i)*RPI, +RPI); lane -> (row, slot) -> source chunk
    unsigned dma_lane[WINST];
#pragma unroll
    for (int i = 0; i < WINST; ++i) {
        const int row = (wv * WINST + i) * RPI + lane / SPR;
        const int slot = lane % SPR;
        const int swz = CK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3);
        dma_lane[i] = (unsigned)((row * K9 + (slot ^ swz) * 8) * 2);
    }
    const char* wbase = reinterpret_cast<const char*>(p.wt);
    auto dma = [&](int item, int cc, int tap, int stage) {
        const int n0 = (item & (p.n_ntiles - 1)) * BN;
        const size_t uni = ((size_t)n0 * K9 + (size_t)tap * Cs + (size_t)(grp * p.cpg + cc) * CK) * 2;
        const char* g = wbase + uni;
        char* l = ring_g + stage * (KG * SLAB) + wv * (WINST * 1024);
        if (DBG(p) & 2) return;
        // (the host pass of hipcc rejects the gfx950-only 16-byte DMA width inside a template and then silently drops the kernel's
        //  host stub, hence the device-pass guard around the two DMA builtins)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int i = 0; i < WINST; ++i)
            __builtin_amdgcn_global_load_lds((gvoid_t*)(g + dma_lane[i]), (lvoid_t*)(l + i * 1024), 16, 0, 0);
#else
        (void)g; (void)l;
#endif
    };

    // ---- patch staging by LDS-DMA through a buffer descriptor: no data registers, no LDS store pass.  The DMA writes 64
    //      consecutive 16-byte slots per instruction; slot n of a patch buffer is (pixel n / SPP, 16-byte column n % SPP), the
    //      last column of a pixel being the pad.  Pad slots, the zero row and pixels outside the tensor get an offset the
    //      hardware range check rejects: the DMA writes zeros there (so the zero row is refreshed with every patch).
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.src), 0, p.M * Cs * 2, 0x00020000);
    constexpr int OOB = 0x40000000;
    int prel[PINST];
#pragma unroll
    for (int i = 0; i < PINST; ++i) {
        const int slot = (wv * PINST + i) * 64 + lane;
        const int q = slot / SPP, sub = slot - q * SPP;
        prel[i] = (q < p.np && sub < SPP - 1) ? q * (Cs * 2) + sub * 16 : OOB;
    }
    auto pdma = [&](int item, int cc, int buf) {
        const int m0 = (item >> p.nt_shift) * BM;
        const int base = (m0 - halo) * (Cs * 2) + (grp * p.cpg + cc) * CK * 2;
        char* l = patch_g + buf * p.patch_bytes + wv * (PINST * 1024);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int i = 0; i < PINST; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (lvoid_t*)(l + i * 1024), 16, prel[i] + base, 0, 0, 0);
#else
        (void)base; (void)l;
#endif
    };

    // ---- this workgroup's items: blockIdx.x, blockIdx.x + gridDim.x, ...
    const int G = gridDim.x;
    const int nmy = (p.n_items - (int)blockIdx.x + G - 1) / G;
    const int cpg = p.cpg;

    // prologue: patch of (item 0, chunk 0), ring slabs of steps 0, 1 and 2 (a chunk always has 9)
    const bool pl_on = !(DBG(p) & 4);
    if (pl_on) pdma(blockIdx.x, 0, 0);
    dma(blockIdx.x, 0, 0, 0);
    dma(blockIdx.x, 0, 1, 1);
    if constexpr (STAG) {
        wait_vm<WINST>();                                    // patch and slab 0 have landed (in-order return); slab 2 is issued by tap 0
    } else {
        dma(blockIdx.x, 0, 2, 2);
        wait_vm<2 * WINST>();
    }
    wg_barrier();

    // Software pipeline.  A slab is consumed in two units of UK MFMA K-steps; the fragments of unit u+1 are read from LDS into
    // the other register set while the MFMAs of unit u run.  The barrier that publishes slab t+1 sits between "MFMAs of
    // (t, unit 0) issued" and "reads of (t+1, unit 0)", i.e. behind 4*UK MFMAs of cover:
    //     read B <- (t, u1) | mfma A | wait DMA(t+1), barrier | DMA(t+3) | read A <- (t+1, u0) | mfma B
    // Ring: slab t lives in stage t % 3; DMA(t+3) overwrites stage t % 3, whose last reads are complete at that barrier.
    constexpr int UK = STAG ? KS : KS / 2;                    // MFMA K-steps per unit
    bf16x8_t ax[UK][2], aw[UK][2], bx[UK][2], bw[UK][2];
    int cur = 0;                                             // patch buffer holding the chunk being multiplied
    for (int k = 0; k < nmy; ++k) {
        const int item = blockIdx.x + k * G;
        const int m0 = (item >> p.nt_shift) * BM, n0 = (item & (p.n_ntiles - 1)) * BN;
        unsigned tmask[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int g = m0 + wm * 64 + i * 32 + l31;
            tmask[i] = g < p.M ? tap_mask4<MODE>(g, p) : 0u;
        }
        // tap = 3 r + s: the filter row r is a run-time loop index, the column s is unrolled, so the ring stage (tap % 3 = s) stays a
        // compile-time address offset while the tap body exists three times, not nine (instruction fetch: the straight-line nine-tap
        // version spent more time fetching code it executes once than multiplying -- profiles/r02_conv4_ablation.txt)
        auto frags = [&](bf16x8_t (&xf)[UK][2], bf16x8_t (&wf)[UK][2], const char* pb, int r, int s, int u) {
            if (DBG(p) & 64) return;
            const char* ws = ring_g + s * (KG * SLAB);
            const int shift = (MODE == 0 ? (r - 1) * W + (s - 1) : (1 - r) * W + (1 - s)) * PP;
            const unsigned bit = 1u << (3 * r + s);
            int xa[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { const int a = xaddr[i] + shift; xa[i] = (tmask[i] & bit) ? a : zoff + (a & 255); }
#pragma unroll
            for (int q = 0; q < UK; ++q) {
#pragma unroll
                for (int i = 0; i < 2; ++i) xf[q][i] = *reinterpret_cast<const bf16x8_t*>(pb + xa[i] + (u * UK + q) * 32);
#pragma unroll
                for (int j = 0; j < 2; ++j) wf[q][j] = *reinterpret_cast<const bf16x8_t*>(ws + waddr[j][u * UK + q]);
            }
        };
        f32x16 acc[2][2];                                    // [channel tile j][pixel tile i]
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
        auto mma = [&](const bf16x8_t (&xf)[UK][2], const bf16x8_t (&wf)[UK][2]) {
            if (DBG(p) & 1) {
#pragma unroll
                for (int q = 0; q < UK; ++q)
#pragma unroll
                    for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(xf[q][j]), "v"(wf[q][j]));
                return;
            }
#pragma unroll
            for (int q = 0; q < UK; ++q)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[q][j], xf[q][i], acc[j][i], 0, 0, 0);
        };

        if constexpr (STAG) {
            // ---- 8-wave workgroups: two waves share every SIMD (w and w + 4).  Waves 4..7 run ONE barrier behind waves 0..3 and
            //      the K loop alternates a read phase (LDS fragment reads of one unit, DMA issue, DMA wait) with an MFMA phase
            //      (the unit's 8 MFMAs at raised priority), a barrier after each: on every SIMD one wave multiplies while its
            //      partner loads, instead of both loading and then both queueing on the matrix pipe (guide: 8-phase template).
            //      Ring hazards with the lag: slab t is fully read once the lagging half has passed its last read phase of t,
            //      i.e. at the barrier before the leading half's first read phase of t+1 -- where DMA(t+2 -> stage (t-1) % 3)
            //      is issued; the wait for DMA(t+1) sits at the end of the LAST read phase of slab t for both halves, so that
            //      it precedes, for both, the barrier in front of the leading half's first read of slab t+1.
            constexpr int NU = KS / UK;                          // units (2 MFMA K-steps, 8 MFMAs) per slab
            const bool lag = wave >= 4;
            if (lag) __builtin_amdgcn_s_barrier();
            for (int cc = 0; cc < cpg; ++cc, cur ^= 1) {
                int ncc = cc + 1, nk = k;
                if (ncc == cpg) { ncc = 0; nk = k + 1; }
                const bool nvalid = nk < nmy;
                const int nitem = blockIdx.x + nk * G;
                const char* pb = patch_g + cur * p.patch_bytes;
                const bool pl = nvalid && pl_on;
#pragma unroll 1
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
#pragma unroll
                        for (int u = 0; u < NU; ++u) {
                            // ---- read phase
                            frags(ax, aw, pb, r, s, u);
                            if (u == 0) {
                                // slab t + 2 into the stage of slab t - 1 (tap + 2: same filter row for s = 0, next row / next chunk otherwise)
                                if (s == 0) dma(item, cc, 3 * r + 2, 2);
                                else if (r < 2) dma(item, cc, 3 * (r + 1) + s - 1, s - 1);
                                else if (nvalid) dma(nitem, ncc, s - 1, s - 1);
                                if (s == PL_TAP && r == 0 && pl) pdma(nitem, ncc, cur ^ 1);
                            }
                            STAMP();
                            if (u == NU - 1) {
                                const bool d2 = (s == 0 || r < 2) ? true : nvalid;     // was DMA(t+2) issued?
                                if (d2) {
                                    if (((s == PL_TAP && r == 0) || (s == 0 && r == 1)) && pl) wait_vm<WINST + PINST>();
                                    else wait_vm<WINST>();
                                } else {
                                    wait_vm<0>();
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            STAMP();
                            wait_lds();                      // the fragments have landed BEFORE the barrier: when the other half passes it and
                            STAMP();
                            __builtin_amdgcn_s_barrier();    // issues DMA into the stage this half has just read, those reads are complete
                            STAMP();
                            // ---- MFMA phase
                            __builtin_amdgcn_sched_barrier(0);
                            if (SETPRIO) __builtin_amdgcn_s_setprio(1);
                            mma(ax, aw);
                            if (SETPRIO) __builtin_amdgcn_s_setprio(0);
                            __builtin_amdgcn_sched_barrier(0);
                            STAMP();
                            __builtin_amdgcn_s_barrier();
                            STAMP();
                        }
                    }
                }
            }
            if (!lag) __builtin_amdgcn_s_barrier();              // both halves aligned again for the epilogue
        } else {
        frags(ax, aw, patch_g + cur * p.patch_bytes, 0, 0, 0);  // the one exposed LDS round trip of the item
        for (int cc = 0; cc < cpg; ++cc, cur ^= 1) {
            // the chunk after this one (possibly the first chunk of the next item)
            int ncc = cc + 1, nk = k;
            if (ncc == cpg) { ncc = 0; nk = k + 1; }
            const bool nvalid = nk < nmy;
            const bool last_chunk = ncc == 0;
            const int nitem = blockIdx.x + nk * G;
            const char* pb = patch_g + cur * p.patch_bytes;
            const char* pbn = patch_g + (cur ^ 1) * p.patch_bytes;
#pragma unroll 1
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    wait_lds();                                   // set A (issued before the previous unit's MFMAs) has landed: free
                    frags(bx, bw, pb, r, s, 1);
                    __builtin_amdgcn_sched_barrier(0);            // keep the two register sets apart: reads first, then the other set's MFMAs
                    mma(ax, aw);
                    __builtin_amdgcn_sched_barrier(0);
                    const bool pl = nvalid && pl_on;
                    if (s == PL_TAP && r == 0 && pl) pdma(nitem, ncc, cur ^ 1);      // tap 2: next patch, published by the barriers of taps 4..8
                    // ---- slab t+1 must have landed; only younger LOADS may stay in flight (in-order return)
                    const bool d2 = (r < 2 || s == 0) ? true : nvalid;               // was DMA(t+2) issued?  (taps 7, 8 reach into the next chunk)
                    if (d2) {
                        if (((s == PL_TAP && r == 0) || (s == 0 && r == 1)) && pl) wait_vm<WINST + PINST>();   // taps 2 and 3: the patch DMA is younger
                        else wait_vm<WINST>();
                    } else {
                        wait_vm<0>();
                    }
                    wg_barrier(DBG(p) & 32);
                    if (r < 2) dma(item, cc, 3 * (r + 1) + s, s);                    // tap + 3, same ring stage as this tap
                    else if (nvalid) dma(nitem, ncc, s, s);
                    if (s < 2) frags(ax, aw, pb, r, s + 1, 0);
                    else if (r < 2) frags(ax, aw, pb, r + 1, 0, 0);
                    else if (!last_chunk) frags(ax, aw, pbn, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    mma(bx, bw);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }

        }   // !STAG

        // ---- K groups: partial tiles of groups 1.. are summed into group 0 through LDS (single-item workgroups only)
        if constexpr (KG > 1) {
            wait_vm<0>();
            float* xr = reinterpret_cast<float*>(smem);
            if (grp > 0) {
                float* q = xr + ((size_t)((grp - 1) * NWG + wv) * 64) * 64;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4)
                            *reinterpret_cast<f32x4*>(q + ((j * 2 + i) * 4 + r4) * 256 + lane * 4) =
                                f32x4{acc[j][i][4 * r4], acc[j][i][4 * r4 + 1], acc[j][i][4 * r4 + 2], acc[j][i][4 * r4 + 3]};
            }
            wg_barrier();
            if (grp > 0) return;
#pragma unroll
            for (int g2 = 1; g2 < KG; ++g2) {
                const float* q = xr + ((size_t)((g2 - 1) * NWG + wv) * 64) * 64;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) {
                            const f32x4 v = *reinterpret_cast<const f32x4*>(q + ((j * 2 + i) * 4 + r4) * 256 + lane * 4);
                            acc[j][i][4 * r4] += v[0]; acc[j][i][4 * r4 + 1] += v[1]; acc[j][i][4 * r4 + 2] += v[2]; acc[j][i][4 * r4 + 3] += v[3];
                        }
            }
        }

        // ---- epilogue.  D[row = channel (r&3) + 8*(r>>2) + 4*kh][col = pixel l31]: a lane holds 4 groups of 4 consecutive
        //      channels per tile; the half-waves are paired with v_permlane32_swap into 16-byte row segments.
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pix = m0 + wm * 64 + i * 32 + l31;
            const bool pv = pix < p.M;
            bf16_t* drow = p.dst + (size_t)pix * p.Cd + n0 + wn * 64;
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
                    if (pv && !(DBG(p) & 8)) *reinterpret_cast<u32x4*>(drow + j * 32 + pr * 16 + kh * 8) = u32x4{rx[0], ry[0], rx[1], ry[1]};
                }
            }
        }
        const bool fwd_stats = MODE == 0 && (p.stats != nullptr || p.stat_acc != nullptr) && !(DBG(p) & 16);
        const bool bwd_sums = MODE == 1 && p.bn_z != nullptr;
        if (fwd_stats || bwd_sums) {
            // Two per-channel sums over this wave's 64 pixels, one channel tile at a time (pixels beyond M hold exact zeros):
            //   forward: sum z, sum z^2 (BatchNorm batch statistics of THIS layer, from the fp32 accumulators);
            //   dgrad  : sum g, sum g*z' with g = dy * (y' > 0) -- the BatchNorm-backward reduction of the layer that produced the
            //            tensor whose gradient dy this launch completes (z', y' = that layer's pre- / post-activation outputs); the
            //            centred form  sum g * xhat = invstd * (sum g z' - mean * sum g)  is taken once per channel below, so
            //            no per-channel constant is needed per element.  Replaces one full bn_bwd_reduce pass (2 tensor reads).
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float sv[32];
                if (MODE == 0) {
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
                        const int pix = m0 + wm * 64 + i * 32 + l31;
                        if (pix < p.M) {
                            const size_t row = (size_t)pix * p.Cd + n0 + wn * 64 + j * 32 + kh * 4;
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
                }
                row16_sum_n(sv);
                if ((lane & 15) == 0) {
                    const int rh = (lane >> 4) & 1;               // which 16-lane row of the half-wave
#pragma unroll
                    for (int which = 0; which < 2; ++which)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            const int cc2 = wn * 64 + j * 32 + g4 * 8 + kh * 4;
                            const float* v = sv + which * 16 + g4 * 4;
                            *reinterpret_cast<f32x4*>(red + ((wm * 2 + rh) * 2 + which) * BN + cc2) = f32x4{v[0], v[1], v[2], v[3]};
                        }
                }
            }
            wg_barrier();
            for (int idx = tid; idx < 2 * BN; idx += NTG) {
                const int which = idx / BN, c2 = idx - which * BN;
                float t = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < WM * 2; ++w2) t += red[(w2 * 2 + which) * BN + c2];
                const int mt = item >> p.nt_shift;
                if (MODE == 0) {
                    if (p.stat_acc != nullptr) atomicAdd(p.stat_acc + ((size_t)(mt & (p.stat_rep - 1)) * 2 + which) * p.Cd + n0 + c2, (double)t);
                    else p.stats[((size_t)mt * 2 + which) * p.Cd + n0 + c2] = t;
                } else {
                    if (which == 1) {                             // sum g * xhat from sum g z' and sum g
                        float sg = 0.f;
#pragma unroll
                        for (int w2 = 0; w2 < WM * 2; ++w2) sg += red[(w2 * 2 + 0) * BN + c2];
                        t = p.bn_invstd[n0 + c2] * (t - p.bn_mean[n0 + c2] * sg);
                    }
                    atomicAdd(p.bn_acc + ((size_t)(mt & (p.bn_rep - 1)) * 2 + which) * p.Cd + n0 + c2, (double)t);
                }
            }
            // `red` is rewritten one item later, many barriers from here
        }
    }
}

struct Cfg4 { int wm, wn, kg, ck; };

template <int WM, int WN, int KG, int CK, int WMAX, int MODE>
int launch4(Conv4Params& p, hipStream_t st) {
    constexpr int BM = WM * 64, BN = WN * 64, PP = CK * 2 + 16, NWG = WM * WN;
    p.np = BM + 2 * p.W + 2;
    p.patch_bytes = (((((BM + 2 * WMAX + 2) * PP + 255) / 256 * 256 + 512) / 1024 + 1 + NWG - 1) / NWG) * NWG * 1024;   // PINST * NWG DMA instructions of 1 KB
    p.n_ntiles = p.Cd / BN;
    p.nt_shift = ilog2_exact(p.n_ntiles);
    const int n_mtiles = (p.M + BM - 1) / BM;
    p.n_items = n_mtiles * p.n_ntiles;
    p.cpg = (p.Cs / CK) / KG;
    size_t lds = (size_t)KG * 2 * p.patch_bytes + (size_t)3 * KG * BN * CK * 2;
    if (KG > 1) { size_t x = (size_t)(KG - 1) * NWG * 64 * 64 * sizeof(float); if (x > lds) lds = x; }
    lds += (size_t)WM * 2 * 2 * BN * sizeof(float);
    if (lds > 160 * 1024) { clhip_set_error("conv4: %zu bytes of LDS", lds); return CLHIP_EINVAL; }
    auto kern = conv4_kernel<WM, WN, KG, CK, WMAX, MODE>;
    static size_t attr = 0;
    if (lds > attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            clhip_set_error("conv4: cannot reserve %zu bytes of LDS", lds);
            return CLHIP_EHIP;
        }
        attr = lds;
    }
    int per_cu = (int)((160 * 1024) / lds);
    const int wave_cap = 8 / (NWG * KG) > 0 ? 8 / (NWG * KG) : 1;           // <= 2 waves per SIMD
    if (per_cu > wave_cap) per_cu = wave_cap;
    if (per_cu < 1) per_cu = 1;
    static const int force_grid = clhip_cfg("CONV4_GRID") ? atoi(clhip_cfg("CONV4_GRID")) : 0;
    int grid = 256 * per_cu;
    if (force_grid > 0) grid = force_grid;
    if (KG > 1 || grid > p.n_items) grid = p.n_items;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NWG * KG * 64), lds, st, p);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

// Tile choice (measured on 256 x {32x32x64, 16x16x128, 8x8x256, 4x4x512} and the 8x8x64 stage of CifarResNet-32,
// profiles/r02_conv4_sweep.txt): fill the 256 CUs first, then prefer 8-wave workgroups (staggered read / MFMA phases) and
// 64-channel slabs (16 MFMAs per wave between two ring barriers).
Cfg4 pick4(int M, int Cs, int Cd, int W) {
    static const char* ov = clhip_cfg("CONV4_CFG");       // tuning override "wm,wn,kg,ck"
    if (g_force4[0] > 0) return Cfg4{g_force4[0], g_force4[1], g_force4[2], g_force4[3]};
    if (ov) { Cfg4 c{0, 0, 0, 0}; if (sscanf(ov, "%d,%d,%d,%d", &c.wm, &c.wn, &c.kg, &c.ck) == 4) return c; }
    auto tiles = [&](int bm, int bn) { return (int64_t)((M + bm - 1) / bm) * (Cd / bn); };
    if (Cd % 128 == 0 && tiles(256, 128) >= 200) return Cfg4{4, 2, 1, 64};
    if (Cd % 128 == 0 && tiles(128, 128) >= 200 && (Cs / 32) % 2 == 0) return Cfg4{2, 2, 2, 32};
    if (tiles(256, 64) >= 256) return Cfg4{4, 1, 1, 32};
    // few pixels (batch <= 128 on the 4x4 stage, <= 32 on the 8x8 one): 64-pixel tiles, the reduction split over four wave groups --
    // these launches are one item per workgroup long and weight-streaming bound (4x4x512: 30 -> 20 us at batch 32 / 64; sweep of
    // every configuration at batch 32 ... 256 in profiles/r02_small_batch_notes.md)
    if (tiles(64, 64) <= 256 && (Cs / 32) % 4 == 0 && W <= 16) return Cfg4{1, 1, 4, 32};
    if (W <= 8 && tiles(128, 64) >= 128 && (Cs / 64) % 2 == 0) return Cfg4{2, 1, 2, 64};
    if ((Cs / 32) % 2 == 0) return Cfg4{2, 1, 2, 32};
    return Cfg4{2, 1, 1, 32};
}

bool cfg_ok(const Cfg4& c, int Cs, int Cd) {
    if (!(c.ck == 32 || c.ck == 64)) return false;
    if (Cs % c.ck || Cd % (c.wn * 64)) return false;
    if ((Cs / c.ck) % c.kg) return false;
    const int nt = Cd / (c.wn * 64);
    return (nt & (nt - 1)) == 0;
}

}  // namespace

bool clhip_conv4_supported(int N, int H, int W, int Cs, int Cd, int ksize, int stride, int pad, int dtype) {
    static const bool on = clhip_cfg("CONV4") ? atoi(clhip_cfg("CONV4")) != 0 : true;
    if (g_enable4 >= 0 ? g_enable4 == 0 : !on) return false;
    if (!(dtype == CLHIP_BF16 && ksize == 3 && stride == 1 && pad == 1 && (Cs % 64) == 0 && (Cd % 64) == 0 && W <= 32 && W >= 2 && H >= 1)) return false;
    // the patch DMA marks rejected slots with a 1 GiB offset: the gathered tensor has to stay well below that
    if ((int64_t)N * H * W * (Cs > Cd ? Cs : Cd) * 2 >= ((int64_t)1 << 29)) return false;
    const int nt = Cd / 64;
    return (nt & (nt - 1)) == 0;
}

int clhip_conv4_tiles_m(int M, int Cs, int Cd, int W) { return (M + pick4(M, Cs, Cd, W).wm * 64 - 1) / (pick4(M, Cs, Cd, W).wm * 64); }

int clhip_conv4_launch_bn(const void* src, const void* wt, void* dst, float* stats, double* stat_acc, int stat_rep, int N, int H, int W, int Cs, int Cd,
                          int accumulate, int mode, const void* bn_z, const void* bn_y, const float* bn_mean, const float* bn_invstd, double* bn_acc, int bn_rep,
                          hipStream_t st);

int clhip_conv4_launch(const void* src, const void* wt, void* dst, float* stats, double* stat_acc, int stat_rep, int N, int H, int W, int Cs, int Cd,
                       int accumulate, int mode, hipStream_t st) {
    return clhip_conv4_launch_bn(src, wt, dst, stats, stat_acc, stat_rep, N, H, W, Cs, Cd, accumulate, mode, nullptr, nullptr, nullptr, nullptr, nullptr, 1, st);
}

int clhip_conv4_launch_bn(const void* src, const void* wt, void* dst, float* stats, double* stat_acc, int stat_rep, int N, int H, int W, int Cs, int Cd,
                          int accumulate, int mode, const void* bn_z, const void* bn_y, const float* bn_mean, const float* bn_invstd, double* bn_acc, int bn_rep,
                          hipStream_t st) {
    Conv4Params p;
    p.bn_z = static_cast<const bf16_t*>(bn_z); p.bn_y = static_cast<const bf16_t*>(bn_y); p.bn_mean = bn_mean; p.bn_invstd = bn_invstd;
    p.bn_acc = bn_acc; p.bn_rep = bn_rep > 0 ? bn_rep : 1;
    p.src = static_cast<const bf16_t*>(src); p.wt = static_cast<const bf16_t*>(wt); p.dst = static_cast<bf16_t*>(dst);
    p.stats = stats; p.stat_acc = stat_acc; p.stat_rep = stat_rep > 0 ? stat_rep : 1;
    p.H = H; p.W = W; p.wshift = ilog2_exact(W); p.hshift = ilog2_exact(H); p.Cs = Cs; p.Cd = Cd; p.accumulate = accumulate; p.M = N * H * W;
    p.debug = g_debug4; p.trace = g_trace4;
    Cfg4 c = pick4(p.M, Cs, Cd, W);
    if (!cfg_ok(c, Cs, Cd)) { clhip_set_error("conv4: configuration %d,%d,%d,%d does not fit Cs=%d Cd=%d", c.wm, c.wn, c.kg, c.ck, Cs, Cd); return CLHIP_EINVAL; }
#define L4(a, b, g, k) (W <= 8 ? (mode == 0 ? launch4<a, b, g, k, 8, 0>(p, st) : launch4<a, b, g, k, 8, 1>(p, st)) \
                               : (mode == 0 ? launch4<a, b, g, k, 32, 0>(p, st) : launch4<a, b, g, k, 32, 1>(p, st)))
    const int key = c.wm * 1000 + c.wn * 100 + c.kg * 10 + (c.ck == 64 ? 1 : 0);
    switch (key) {
        case 4110: return L4(4, 1, 1, 32);
        case 4111: return L4(4, 1, 1, 64);
        case 2110: return L4(2, 1, 1, 32);
        case 2111: return L4(2, 1, 1, 64);
        case 2210: return L4(2, 2, 1, 32);
        case 2211: return L4(2, 2, 1, 64);
        case 4210: return L4(4, 2, 1, 32);
        case 4211: return L4(4, 2, 1, 64);
        case 2120: return L4(2, 1, 2, 32);
        case 2121: return L4(2, 1, 2, 64);
        case 1220: return L4(1, 2, 2, 32);
        case 2220: return L4(2, 2, 2, 32);
        case 1140: return L4(1, 1, 4, 32);
        case 1141: return L4(1, 1, 4, 64);
        case 2140: return L4(2, 1, 4, 32);
        case 2141: return L4(2, 1, 4, 64);
        case 1240: return L4(1, 2, 4, 32);
        default: break;
    }
#undef L4
    clhip_set_error("conv4: no kernel for configuration %d,%d,%d,%d", c.wm, c.wn, c.kg, c.ck);
    return CLHIP_EINVAL;
}

// ---- tuning hooks (tools/ubench/conv_bench.cpp; not part of include/clhip.h)
void clhip_conv4_set_cfg(int wm, int wn, int kg, int ck) { g_force4[0] = wm; g_force4[1] = wn; g_force4[2] = kg; g_force4[3] = ck; }
void clhip_conv4_enable(int on) { g_enable4 = on; }
void clhip_conv4_set_debug(int bits) { g_debug4 = bits; }
void clhip_conv4_set_trace(unsigned long long* dev_buf) { g_trace4 = dev_buf; }

// ---- lazy BatchNorm input (common.h LazyIn): not built for this kernel family yet
bool clhip_conv4_in_supported(int N, int H, int W, int Cs, int Cd) { (void)N; (void)H; (void)W; (void)Cs; (void)Cd; return false; }
int clhip_conv4_launch_in(const void* src, const void* wt, void* dst, double* stat_acc, int stat_rep, int N, int H, int W, int Cs, int Cd, const LazyIn* in, hipStream_t st) {
    (void)src; (void)wt; (void)dst; (void)stat_acc; (void)stat_rep; (void)N; (void)H; (void)W; (void)Cs; (void)Cd; (void)in; (void)st;
    clhip_set_error("conv4: lazy inputs are not supported");
    return CLHIP_EINVAL;
}
