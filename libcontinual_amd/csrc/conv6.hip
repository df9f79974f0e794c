// conv6.hip -- input gradient of a DOWN-SAMPLING block entry in one launch: the 3x3 / stride 2 / pad 1 convolution's dgrad and, in the
// same accumulators, the dgrad of the 1x1 / stride 2 shortcut convolution that reads the same block input (bf16, gfx950).
//
//   dx[n, 2i+pr, 2j+pc, c] = sum_k sum_{taps (r,s) of the class}  dz[n, i+dr, j+dc, k] * W[k][r][s][c]      (+ for pr = pc = 0:
//                            sum_k dzs[n, i, j, k] * Ws[k][c])
//
// The generic kernel (conv2.hip, parity-class tiles, 16x16 MFMA, register-staged operands) ran these three ResNet-18 launches at
// 0.09-0.14 of their roofline (45 / 33 / 43 us) and the shortcut's dgrad was a second launch that read-modified-wrote the whole
// gradient tensor for a quarter of its pixels (20 / 17 / 14 us) -- profiles/r03_layer_roofline.md.  Here:
//   * a pixel of dz feeds the FOUR parity classes of dx through nine taps, and those nine taps read only four shifted copies of the
//     dz fragment (shift 0 / +1 column / +1 row / both): a wave holds the accumulators of all four classes (64 channels x PXF*32
//     pixels x 4 classes = 128 / 256 registers of the 512-entry file, one wave per SIMD), so per 16-wide K step it reads 4 PXF pixel
//     fragments + 18 weight fragments for 18 PXF MFMAs 32x32x16 -- 0.72 LDS reads per MFMA at PXF = 2 (conv4: 1.0);
//   * the shortcut is a tenth tap into the (even, even) class with its own dz and weights: no second launch, no read-modify-write;
//   * operands are staged by LDS-DMA through buffer descriptors (pad slots, pixels beyond the tensor and the zero area are
//     rejected by the range check and arrive as zeros) in K chunks of 16, NST stages deep, ONE workgroup barrier per chunk placed
//     in the middle of the chunk's MFMAs: the fragments of chunk c+1 are requested under the second half of chunk c;
//   * pitches are odd numbers of 16-byte slots (19 per weight row, 3 per pixel): every ds_read_b128 is conflict-free; lanes whose
//     shifted pixel lies outside the image read a zero area at the same bank offset (conv4.hip's trick);
//   * the epilogue is conv4's (v_permlane32_swap pairs half-waves into 16-byte stores) with the class's strided pixel address.
// Replaces the autograd input gradients of `conv1` + `downsample[0]` of a BasicBlock (core/model/backbone/resnet.py:226-234, 17-24).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void lvoid_t;

struct Dgrad6Params {
    const bf16_t* dz;    // [N,Ho,Wo,K]   gradient of the 3x3 convolution's output
    const bf16_t* wpk;   // packed dgrad weights of both convolutions (layout below)
    const bf16_t* dzs;   // [N,Ho,Wo,K]   gradient of the shortcut convolution's output
    bf16_t* dx;          // [N,2Ho,2Wo,C]
    int Ho, Wo, lgHo, lgWo, C, K, Mz, accumulate, n_ctiles;
    unsigned long long* trace;   // CONV6_TRACE (measurement only): s_memtime stamps of wave 0 of workgroup `trace_wg`
    int trace_wg;
    int debug;           // CONV6_DEBUG (ABL=1 builds; bit 1 everywhere): 1 force 128-pixel tiles, 2 no stores, 8 no DMA waits, 64 no barriers, 128 no K loop, 256 no epilogue
};

int g_enable6 = -1;
unsigned long long* g_trace6 = nullptr;
int g_trace_wg6 = 0;
// measurement switches and s_memtime stamps exist in ABL=1 builds only (libclhip_abl.so): a run-time test around an MFMA or a DMA splits
// the software pipeline's basic blocks -- with them compiled in, a step of four MFMAs took 296 cycles instead of 128
#ifdef CLHIP_ABLATION
#define DBG6(p) ((p).debug)
#define STAMP6() do { if (p.trace && (int)blockIdx.x == p.trace_wg && tid == 0 && nstamp < 255) p.trace[1 + nstamp++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DBG6(p) 0
#define STAMP6() do { } while (0)
#endif

// Packed weights: for every (64-channel tile ct, 16-wide K chunk kc) one contiguous block of 64 rows x 21 sixteen-byte slots --
// slot 2 t + h of row c = W[k = kc*16 + h*8 .. +8][tap t][c] for the nine taps t of the 3x3 filter, slots 18 / 19 the shortcut's 1x1
// filter, slot 20 a pad (odd row pitch: conflict-free fragment reads).  A block IS the LDS image of a chunk's weights, so its DMA
// reads 1 KB of consecutive addresses per instruction.  (The first version gathered 32-byte pieces of the [C][9][K] copy: every
// 128-byte line crossed the L2 -> L1 path for a quarter of its bytes, and the kernel ran at the speed of that path -- 25 us with the
// MFMAs and the stores switched off.)
constexpr int WSL6 = 21;                 // slots per packed row
constexpr int WBLK6 = 64 * WSL6 * 16;    // bytes per (ct, kc) block = 21 DMA instructions
constexpr int XSL6 = 5;                  // slots per staged pixel: 32 K values (two chunks) + pad
constexpr int OOB6 = 0x40000000;

__device__ __forceinline__ void wait_lds6() { __builtin_amdgcn_s_waitcnt(0xC07F); }
template <int N> __device__ __forceinline__ void wait_vm6() { __builtin_amdgcn_s_waitcnt((N & 15) | 0x70 | 0xF00 | ((N >> 4) << 14)); }

// step order: (tap, pixel-fragment copy, parity class).  Copies: 0 = shift 0, 1 = +1 column, 2 = +1 row, 3 = both, 4 = the shortcut's
// dz.  Class = pr * 2 + pc.  Tap (r, s) belongs to class (r != 1, s != 1) and reads dz row i + (r == 0), column j + (s == 0).  The
// four-tap class (1, 1) takes every other step so that two MFMAs into one accumulator are never adjacent.
constexpr int ORD6[10][3] = {{4, 0, 0}, {8, 0, 3}, {5, 0, 1}, {6, 1, 3}, {7, 0, 2}, {2, 2, 3}, {3, 1, 1}, {0, 3, 3}, {1, 2, 2}, {9, 4, 0}};

// one thread per 16-byte slot of the packed copy
__global__ __launch_bounds__(256) void pack6_kernel(const bf16_t* __restrict__ wdg, const bf16_t* __restrict__ wsc, uint4* __restrict__ out, int C, int K) {
    const int nkc = K / 16;
    const int64_t total = (int64_t)(C / 64) * nkc * 64 * WSL6;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int slot = (int)(t % WSL6), row = (int)((t / WSL6) % 64), kc = (int)((t / (WSL6 * 64)) % nkc), ct = (int)(t / ((int64_t)WSL6 * 64 * nkc));
    const int c = ct * 64 + row;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (slot < 18) v = *reinterpret_cast<const uint4*>(wdg + ((size_t)c * 9 + (slot >> 1)) * K + kc * 16 + (slot & 1) * 8);
    else if (slot < 20 && wsc != nullptr) v = *reinterpret_cast<const uint4*>(wsc + (size_t)c * K + kc * 16 + (slot - 18) * 8);
    out[t] = v;
}

template <int PXF, bool SC>
__global__ __launch_bounds__(256, 1) void dgrad6_kernel(const Dgrad6Params p) {
    constexpr int BMPX = 4 * PXF * 32;                             // dz pixels per workgroup
    constexpr int NPATCH = BMPX + 16 + 2;                          // staged pixels (Wo <= 16): the tile + one row + one pixel (+1 spare)
    constexpr int I_W = 21, I_P = (NPATCH * XSL6 + 63) / 64, I_Q = SC ? BMPX * XSL6 / 64 : 0;      // DMA instructions (1 KB) per region
    constexpr int NSTW = PXF == 2 ? 3 : 4;                         // weight ring stages (one 16-wide K chunk each)
    constexpr int PBUF = (I_P + I_Q) * 1024;                       // one patch buffer: 32 K values (two chunks) of the pixels
    constexpr int OFF_P = NSTW * WBLK6, OFF_Z = OFF_P + 2 * PBUF;
    // every wave issues the same instruction list (the vmcnt waits are compile-time counts): instructions wv, wv+4, ... of a region;
    // list entries beyond a region's end fetch nothing (offset rejected by the range check) into the zero area, which stays zero
    constexpr int N_W = (I_W + 3) / 4, N_P = (I_P + 3) / 4, N_Q = (I_Q + 3) / 4;
    constexpr int MID = 5;                                         // the step in front of which a chunk's barrier sits
    static_assert(OFF_Z + 1024 <= 160 * 1024, "LDS");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int K = p.K, Wo = p.Wo;
    const int ct = blockIdx.x % p.n_ctiles, pt = blockIdx.x / p.n_ctiles;
    const int m0 = pt * BMPX, c0 = ct * 64;
    const int nchunks = (DBG6(p) & 128) ? 0 : K / 16;
    int nstamp = 0; (void)nstamp;
    STAMP6();

    reinterpret_cast<float*>(smem + OFF_Z)[tid] = 0.f;             // the zero area, once

    // ---- DMA set-up
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.wpk) + (size_t)ct * nchunks * (WBLK6 / 2), 0, nchunks * WBLK6, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.dz), 0, p.Mz * K * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(SC ? p.dzs : p.dz), 0, p.Mz * K * 2, 0x00020000);
    int w_rel[N_W], w_dst[N_W], p_rel[N_P], p_dst[N_P], q_rel[N_Q > 0 ? N_Q : 1], q_dst[N_Q > 0 ? N_Q : 1];
    const int np = BMPX + Wo + 2;
#pragma unroll
    for (int t = 0; t < N_W; ++t) {
        const int k = wv + 4 * t;
        w_rel[t] = k < I_W ? k * 1024 + lane * 16 : OOB6;
        w_dst[t] = k < I_W ? k * 1024 : OFF_Z;                     // (ring stage offset added per chunk; the zero area is absolute)
    }
#pragma unroll
    for (int t = 0; t < N_P; ++t) {
        const int k = wv + 4 * t, n = k * 64 + lane, q = n / XSL6, sub = n - q * XSL6;
        p_rel[t] = (k < I_P && sub < 4 && q < np) ? (m0 + q) * K * 2 + sub * 16 : OOB6;
        p_dst[t] = k < I_P ? k * 1024 : -1;
    }
#pragma unroll
    for (int t = 0; t < N_Q; ++t) {
        const int k = wv + 4 * t, n = k * 64 + lane, q = n / XSL6, sub = n - q * XSL6;
        q_rel[t] = (k < I_Q && sub < 4) ? (m0 + q) * K * 2 + sub * 16 : OOB6;
        q_dst[t] = k < I_Q ? (I_P + k) * 1024 : -1;
    }
    // one DMA instruction each (the per-chunk lists are spread over the MFMA steps that follow the barrier: issued in a burst, the
    // 17 instructions of an even chunk stalled every wave for ~1900 cycles -- the CU's load path takes one per ~28 cycles)
    auto wdma1 = [&](int c, int t) {                               // entry t of the weights of chunk c -> ring stage c % NSTW
#if defined(__HIP_DEVICE_COMPILE__)
        char* l = smem + (c % NSTW) * WBLK6;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lvoid_t*)(w_dst[t] == OFF_Z ? smem + OFF_Z : l + w_dst[t]), 16, w_rel[t] + c * WBLK6, 0, 0, 0);
#else
        (void)c; (void)t;
#endif
    };
    auto pdma1 = [&](int u, int t) {                               // entry t of patch unit u (K values [32 u, 32 u + 32)) -> patch buffer u & 1
#if defined(__HIP_DEVICE_COMPILE__)
        char* l = smem + OFF_P + (u & 1) * PBUF;
        if (t < N_P) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_p, (lvoid_t*)(p_dst[t] < 0 ? smem + OFF_Z : l + p_dst[t]), 16, p_rel[t] + u * 64, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_q, (lvoid_t*)(q_dst[t - N_P] < 0 ? smem + OFF_Z : l + q_dst[t - N_P]), 16, q_rel[t - N_P] + u * 64, 0, 0, 0);
#else
        (void)u; (void)t;
#endif
    };
    auto wdma = [&](int c) {
#pragma unroll
        for (int t = 0; t < N_W; ++t) wdma1(c, t);
    };
    auto pdma = [&](int u) {
#pragma unroll
        for (int t = 0; t < N_P + N_Q; ++t) pdma1(u, t);
    };

    // ---- fragment addresses
    const int waddr = l31 * (WSL6 * 16) + kh * 16;                        // in a ring stage: + j * 32 rows, + tap * 32 bytes
    int xaddr[5][PXF];                                                     // in a patch buffer, per copy; masked lanes -> zero area at the same bank offset
#pragma unroll
    for (int i = 0; i < PXF; ++i) {
        const int q = (wv * PXF + i) * 32 + l31, g = m0 + q;
        const int jj = g & (Wo - 1), ii2 = (g >> p.lgWo) & (p.Ho - 1);
        const bool right = jj < Wo - 1, down = ii2 < p.Ho - 1;
        const int a0 = q * (XSL6 * 16) + kh * 16;
        const int a1 = a0 + XSL6 * 16, a2 = a0 + Wo * (XSL6 * 16), a3 = a2 + XSL6 * 16;
        xaddr[0][i] = a0;
        xaddr[1][i] = right ? a1 : -1 - (a1 & 255);                        // negative: -1 - (offset in the zero area)
        xaddr[2][i] = down ? a2 : -1 - (a2 & 255);
        xaddr[3][i] = (right && down) ? a3 : -1 - (a3 & 255);
        xaddr[4][i] = I_P * 1024 + q * (XSL6 * 16) + kh * 16;
    }

    // prologue: patch unit 0 and the weights of chunks 0 .. NSTW-2 in flight; patch 0 and chunk 0 landed and published
    pdma(0);
#pragma unroll
    for (int c = 0; c < NSTW - 1; ++c)
        if (c < nchunks) wdma(c);
    f32x16 acc[4][2][PXF];                                                 // [class][channel tile j][pixel tile i]
#pragma unroll
    for (int cl = 0; cl < 4; ++cl)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < PXF; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[cl][j][i][r] = 0.f;

    STAMP6();
    if (nchunks >= NSTW - 1) wait_vm6<(NSTW - 2) * N_W>(); else wait_vm6<0>();
    wait_lds6();                                                   // (the zero area's store)
    STAMP6();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    STAMP6();

    // Software pipeline: ten steps per chunk (a tap each; without the shortcut the tenth is empty), the weight fragments of step n+2
    // are requested before the MFMAs of step n -- four register sets, and 10 = 2 mod 4 keeps the set of a step a compile-time function
    // of (chunk parity, step); the pixel fragments of chunk c+1 are requested right after the barrier in the middle of chunk c.
    constexpr int NSTEP = 10;
    bf16x8_t xr[2][SC ? 5 : 4][PXF], wf[4][2];
    auto xread = [&](bf16x8_t (&x)[SC ? 5 : 4][PXF], int c) {             // pixel fragments of chunk c
        const char* pb = smem + OFF_P + ((c >> 1) & 1) * PBUF + (c & 1) * 32;
        const char* zb = smem + OFF_Z;
#pragma unroll
        for (int cp = 0; cp < (SC ? 5 : 4); ++cp)
#pragma unroll
            for (int i = 0; i < PXF; ++i) {
                const int a = xaddr[cp][i];
                x[cp][i] = *reinterpret_cast<const bf16x8_t*>(a >= 0 ? pb + a : zb + (-1 - a));
            }
    };
    auto wread = [&](bf16x8_t (&w)[2], const char* sb, int tap) {
#pragma unroll
        for (int j = 0; j < 2; ++j) w[j] = *reinterpret_cast<const bf16x8_t*>(sb + waddr + j * (32 * WSL6 * 16) + tap * 32);
    };
    xread(xr[0], 0);
    wread(wf[0], smem, ORD6[0][0]);
    wread(wf[1], smem, ORD6[1][0]);

    auto chunk = [&](auto par_c, int c) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr int WB = 2 * PAR;                                        // weight set of step 0
        constexpr int NE = (PAR == 0 ? N_P + N_Q : 0) + N_W;               // DMA instructions issued in this chunk ...
        constexpr int PER = (NE + (NSTEP - MID) - 1) / (NSTEP - MID);      // ... per step, from the barrier on
        const char* sb = smem + (c % NSTW) * WBLK6;
        const char* sbn = smem + ((c + 1) % NSTW) * WBLK6;
        const bool more = c + 1 < nchunks;
        const bool pd = PAR == 0 && c + 2 < nchunks, wd = c + NSTW - 1 < nchunks;
#pragma unroll
        for (int n = 0; n < NSTEP; ++n) {
            if (n == 0) STAMP6();
            if (n == MID) {
                STAMP6();
                // the weights of chunk c+1 (and, before an even chunk, its patch unit) have landed -- this wave's part; younger ring DMAs
                // may stay in flight -- then the barrier: they are visible, and every wave is past chunk c-1, whose ring stage and (even c)
                // patch buffer the DMAs issued from here on overwrite.  Issue order patch, then ring: what is needed first returns first.
                if (DBG6(p) & 8) {
                } else if (NSTW > 3 && c + NSTW - 2 < nchunks) wait_vm6<(NSTW - 3) * N_W>();
                else wait_vm6<0>();
                STAMP6();
                if (!(DBG6(p) & 64)) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                STAMP6();
                if (more) xread(xr[PAR ^ 1], c + 1);
            }
            if (n >= MID) {
#pragma unroll
                for (int e = (n - MID) * PER; e < (n - MID + 1) * PER && e < NE; ++e) {
                    if (PAR == 0 && e < N_P + N_Q) { if (pd) pdma1((c >> 1) + 1, e); }
                    else if (wd) wdma1(c + NSTW - 1, e - (PAR == 0 ? N_P + N_Q : 0));
                }
                if (n == NSTEP - 1) STAMP6();
            }
            // weight fragments two steps ahead (tap 9 without a shortcut: no step)
            if (n + 2 < NSTEP) { if (SC || ORD6[n + 2][0] < 9) wread(wf[(WB + n + 2) & 3], sb, ORD6[n + 2][0]); }
            else if (more) wread(wf[(WB + n + 2) & 3], sbn, ORD6[n + 2 - NSTEP][0]);
            __builtin_amdgcn_sched_barrier(0);
            if (SC || ORD6[n][0] < 9) {
                const int cp = ORD6[n][1], cl = ORD6[n][2];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < PXF; ++i) {
                        acc[cl][j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[(WB + n) & 3][j], xr[PAR][SC ? cp : (cp < 4 ? cp : 0)][i], acc[cl][j][i], 0, 0, 0);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int c = 0; c < nchunks; c += 2) {
        chunk(std::integral_constant<int, 0>{}, c);
        chunk(std::integral_constant<int, 1>{}, c + 1);
    }

    STAMP6();
    if (DBG6(p) & 256) return;
    // ---- epilogue: D[row = channel (r&3) + 8*(r>>2) + 4*kh][col = pixel l31]; class (pr, pc) of dz pixel (n, i, j) is dx pixel (n, 2i+pr, 2j+pc)
    const int W2 = 2 * Wo, H2 = 2 * p.Ho, C = p.C;
#pragma unroll
    for (int i = 0; i < PXF; ++i) {
        const int g = m0 + (wv * PXF + i) * 32 + l31;
        const bool pv = g < p.Mz;
        const int jj = g & (Wo - 1), ii2 = (g >> p.lgWo) & (p.Ho - 1), n = g >> (p.lgWo + p.lgHo);
#pragma unroll
        for (int cl = 0; cl < 4; ++cl) {
            bf16_t* drow = p.dx + ((size_t)(n * H2 + 2 * ii2 + (cl >> 1)) * W2 + 2 * jj + (cl & 1)) * C + c0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (p.accumulate && pv) {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const uint2 old = *reinterpret_cast<const uint2*>(drow + j * 32 + g4 * 8 + kh * 4);
                        acc[cl][j][i][4 * g4 + 0] += __uint_as_float(old.x << 16); acc[cl][j][i][4 * g4 + 1] += __uint_as_float(old.x & 0xffff0000u);
                        acc[cl][j][i][4 * g4 + 2] += __uint_as_float(old.y << 16); acc[cl][j][i][4 * g4 + 3] += __uint_as_float(old.y & 0xffff0000u);
                    }
                }
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    unsigned ax = pack_bf16x2(acc[cl][j][i][8 * pr + 0], acc[cl][j][i][8 * pr + 1]), ay = pack_bf16x2(acc[cl][j][i][8 * pr + 2], acc[cl][j][i][8 * pr + 3]);
                    unsigned bx = pack_bf16x2(acc[cl][j][i][8 * pr + 4], acc[cl][j][i][8 * pr + 5]), by = pack_bf16x2(acc[cl][j][i][8 * pr + 6], acc[cl][j][i][8 * pr + 7]);
                    auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                    auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                    if (pv && !(DBG6(p) & 2)) *reinterpret_cast<u32x4*>(drow + j * 32 + pr * 16 + kh * 8) = u32x4{rx[0], ry[0], rx[1], ry[1]};
                }
            }
        }
    }
    STAMP6();
#ifdef CLHIP_ABLATION
    if (p.trace && (int)blockIdx.x == p.trace_wg && tid == 0) p.trace[0] = (unsigned long long)nstamp;
#endif
}

template <int PXF, bool SC>
int launch6(const Dgrad6Params& p, hipStream_t st) {
    constexpr int BMPX = 4 * PXF * 32, NPATCH = BMPX + 18;
    constexpr int I_P = (NPATCH * XSL6 + 63) / 64, I_Q = SC ? BMPX * XSL6 / 64 : 0, NSTW = PXF == 2 ? 3 : 4;
    constexpr size_t lds = (size_t)NSTW * WBLK6 + 2 * (I_P + I_Q) * 1024 + 1024;
    auto kern = dgrad6_kernel<PXF, SC>;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            clhip_set_error("conv6: cannot reserve %zu bytes of LDS", lds);
            return CLHIP_EHIP;
        }
        attr = true;
    }
    const int grid = ((p.Mz + BMPX - 1) / BMPX) * p.n_ctiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, p);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

int ilog2_6(int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; }

}  // namespace

// (N, H, W, C) = the block input whose gradient is produced; K = channels of the two convolutions' outputs
bool clhip_dgrad6_supported(int N, int H, int W, int C, int K, int dtype) {
    static const bool on_env = clhip_cfg("CONV6") ? atoi(clhip_cfg("CONV6")) != 0 : true;
    if (g_enable6 >= 0 ? g_enable6 == 0 : !on_env) return false;
    if (dtype != CLHIP_BF16 || (H & 1) || (W & 1) || C % 64 != 0 || K % 32 != 0 || K < 32) return false;
    const int Ho = H / 2, Wo = W / 2;
    if (ilog2_6(Ho) < 0 || ilog2_6(Wo) < 0 || Wo > 16 || Wo < 2 || Ho < 1) return false;
    const int64_t Mz = (int64_t)N * Ho * Wo;
    if (Mz * K * 2 >= ((int64_t)1 << 29) || (int64_t)(K / 16) * WBLK6 >= ((int64_t)1 << 29)) return false;      // the DMA's out-of-range marker is a 1 GiB offset
    return Mz >= 128;
}

void clhip_conv6_enable(int on) { g_enable6 = on; }
void clhip_conv6_set_trace(unsigned long long* buf, int wg) { g_trace6 = buf; g_trace_wg6 = wg; }

size_t clhip_dgrad6_packed_bytes(int C, int K) { return (size_t)(C / 64) * (K / 16) * WBLK6; }

// packed copy of the two dgrad weight tensors ([C][9][K] and, nullable, [C][1][K]); the plan's weight preparation writes the same
// layout directly (plan.hip)
int clhip_dgrad6_pack(const void* w_dg, const void* w_sc_dg, void* packed, int C, int K, hipStream_t st) {
    const int64_t total = (int64_t)(C / 64) * (K / 16) * 64 * WSL6;
    hipLaunchKernelGGL(pack6_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, static_cast<const bf16_t*>(w_dg), static_cast<const bf16_t*>(w_sc_dg),
                       static_cast<uint4*>(packed), C, K);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

int clhip_dgrad6_launch(const void* dz, const void* w_packed, const void* dz_sc, void* dx, int accumulate, int N, int H, int W, int C, int K, hipStream_t st) {
    Dgrad6Params p;
    p.dz = static_cast<const bf16_t*>(dz); p.wpk = static_cast<const bf16_t*>(w_packed);
    p.dzs = static_cast<const bf16_t*>(dz_sc);
    p.dx = static_cast<bf16_t*>(dx);
    p.Ho = H / 2; p.Wo = W / 2; p.lgHo = ilog2_6(p.Ho); p.lgWo = ilog2_6(p.Wo);
    p.C = C; p.K = K; p.Mz = N * p.Ho * p.Wo; p.accumulate = accumulate; p.n_ctiles = C / 64;
    static const int dbg = clhip_cfg("CONV6_DEBUG") ? atoi(clhip_cfg("CONV6_DEBUG")) : 0;
    p.debug = dbg;
    p.trace = g_trace6; p.trace_wg = g_trace_wg6;
    // 256 dz pixels per workgroup while that fills the chip, else 128
    const bool big = ((p.Mz + 255) / 256) * p.n_ctiles >= 256 && !(dbg & 1);
    const bool sc = dz_sc != nullptr;
    if (big) return sc ? launch6<2, true>(p, st) : launch6<2, false>(p, st);
    return sc ? launch6<1, true>(p, st) : launch6<1, false>(p, st);
}
