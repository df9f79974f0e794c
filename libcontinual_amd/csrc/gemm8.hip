// gemm8.hip -- C[M,N] = epi(A[M,K] . B[N,K]^T), bf16, gfx950: 256 x 256 tile, 64-deep K tiles, eight-phase schedule (round 5).
//
// gemm5.hip (256 x 256, 32-deep slabs, 4-stage ring) runs its K loop at 2 080 cycles per 32-deep slab against 1 024 of matrix pipe
// (profiles/r02_gemm5_notes.md): what is exposed there is the READ phase of the staggered halves -- twelve fragment reads whose data comes back
// ~600 cycles after issue, behind a barrier, against 512 cycles of MFMAs in the partner wave -- and 64-byte DMA rows.  This kernel is the
// schedule cdna_hip_programming.md section 5 describes for gfx950 ("the 256^2 8-phase template"), re-derived for this library's operand
// roles (weight fragment = MFMA A operand, so a lane owns consecutive n of one output row) and its persistent tile loop:
//
//  * a 64-deep K tile is FOUR 16-KB half tiles of 128 rows x 128 bytes, split by QUADRANT of the wave tile, not by wave: W0 / W1 = the weight
//    rows every wave uses for the n half j = 0 / 1 of its 128 (m) x 64 (n) tile, X0 / X1 = the activation rows of its m half i = 0 / 1.  A
//    K tile is four phases of 8 MFMAs (32x32x16: 256 cycles) over the quadrants (i,j) = (0,0) (0,1) (1,1) (1,0); phase 0 reads W0 (4
//    ds_read_b128) and X0 (8), phase 1 reads W1 (4), phase 2 reads X1 (8), phase 3 reads nothing (W0 stays in registers).  So every half tile
//    is dead early -- W0 after phase 0, X0 after phase 0, W1 after phase 1, X1 after phase 2 -- and can be restaged for the K tile AFTER the
//    next one two phases later: phase q issues, by LDS-DMA (2 x buffer_load ... lds per lane, 128-byte rows), the half tile SEVEN phases
//    ahead of its first use.  Two K-tile buffers = 128 KB of LDS.
//  * ONE counted wait per K tile (s_waitcnt vmcnt(6) in phase 3: everything but the three newest half tiles has landed = the whole next K
//    tile), raw s_barrier twice per phase; the DMA queue is never drained inside the loop, and runs on across tile boundaries of the
//    persistent workgroup, so a tile's epilogue covers the next tile's first loads.  Past the last half tile the cursor issues
//    out-of-range (zero-fill, no traffic) requests so that the count stays uniform.
//  * waves 4-7 (m half of the tile = the younger half of the workgroup) run ONE barrier behind waves 0-3: on every SIMD one wave multiplies
//    while its partner reads and issues; s_setprio 1 around the MFMAs.
//  * rows of a half tile are XOR-swizzled in 16-byte chunks (chunk ^ (row >> 1) & 7) on the DMA's per-lane SOURCE address and on the
//    read address (the LDS image of a DMA is lane-linear), which makes the 32-row ds_read_b128 of a fragment conflict-free.
//
// Safety of the restaging (the argument behind the phase table; intervals = barrier-to-barrier, waves 0-3 read in interval 2P of phase P,
// waves 4-7 in 2P + 1): a half tile read in phase P is retired by the reader's lgkmcnt wait no later than interval 2P + 2; its successor is
// issued in phase P + 1 (W0: its reads are retired by lgkmcnt(8) BEFORE the reading phase's first barrier) or P + 2.  Data waited for in
// phase 3 of K tile T (both halves, each before its own barrier) is first read in phase 0 of T + 1, two barriers later.
//
// Epilogues as gemm.hip / gemm5.hip.  Replaces F.linear on the ViT path (core/model/backbone/transformer.py:172, 194, 1259-1271).
#include <stdlib.h>
#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void lvoid_t;
typedef __attribute__((ext_vector_type(2))) float f2;

enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_RES = 2, EPI_BIAS_GELU = 3, EPI_MUL = 4 };

struct Gemm8Params {
    const bf16_t* A; const bf16_t* B; bf16_t* C;
    const float* bias; const bf16_t* R; bf16_t* H;
    int M, N, K, lda, ldb, ldc, ldr, ldh;
    int nt, items, ipx;          // n tiles, tiles, tiles per XCD
    int group_m, panels;         // rasterisation: tiles are numbered column-major inside groups of group_m row panels (1: row-major)
    int max_nmy, shift;          // tiles of the busiest workgroup; start delay (shader cycles) of the workgroups that walk fewer tiles (see the kernel)
    int opt;                     // experiments (CLHIP_GEMM8_OPT): bit 0 = the two wave halves realign at a tile's end and store at the same time (measured: no gain, qkv 101 -> 107 us)
};

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF = 128 * 128;               // one half tile: 128 rows x 128 bytes
constexpr int KTILE = 4 * HALF;               // W0, X0, W1, X1
constexpr int LDS8 = 2 * KTILE;               // 131072
constexpr int SUB_W0 = 0, SUB_X0 = 1, SUB_W1 = 2, SUB_X1 = 3;
constexpr int OOB8 = 0x7ffffff0;

template <int N> __device__ __forceinline__ void wait_vm8() { __builtin_amdgcn_s_waitcnt((N & 15) | 0x70 | 0xF00 | ((N >> 4) << 14)); }
template <int N> __device__ __forceinline__ void wait_lgkm8() { __builtin_amdgcn_s_waitcnt(0x3F | 0x70 | (N << 8) | 0xC000); }

__device__ __forceinline__ void gelu_both8(f2 x, f2& y, f2& dy) {
    const f2 ax = {fabsf(x.x), fabsf(x.y)};
    const f2 z = ax * 0.70710678118654752f;
    const f2 den = z * 0.3275911f + 1.0f;
    const f2 t = {__frcp_rn(den.x), __frcp_rn(den.y)};
    const f2 nz2 = -(z * z);
    const f2 e = {__expf(nz2.x), __expf(nz2.y)};
    f2 pl = t * 1.061405429f + (-1.453152027f);
    pl = t * pl + 1.421413741f;
    pl = t * pl + (-0.284496736f);
    pl = t * pl + 0.254829592f;
    const f2 erf_abs = 1.0f - pl * t * e;
    const f2 se = {copysignf(erf_abs.x, x.x), copysignf(erf_abs.y, x.y)};
    const f2 phi = se * 0.5f + 0.5f;
    y = x * phi;
    dy = x * 0.3989422804014327f * e + phi;
}

__device__ __forceinline__ void unpack4_8(uint2 v, float (&f)[4]) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
}

__device__ __forceinline__ bf16x8_t ldsf(const char* p) { return *reinterpret_cast<const bf16x8_t*>(p); }

template <int EPI>
__global__ __launch_bounds__(512) void gemm8_kernel(const Gemm8Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                 // wave tile: rows wm*128.., columns wn*64..; waves w and w+4 share a SIMD
    const bool lag = wave >= 4;
    const int l31 = lane & 31, kh = lane >> 5;

    // ---- fragment addresses inside a half tile: row * 128 + ((2 ks + kh) ^ swz(row)) * 16, swz(row) = (row >> 1) & 7
    int xaddr[2][4], waddr[4];
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2) {
        const int row = wm * 64 + i2 * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) xaddr[i2][ks] = row * 128 + (((2 * ks + kh) ^ ((row >> 1) & 7)) << 4);
    }
    {
        const int row = wn * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) waddr[ks] = row * 128 + (((2 * ks + kh) ^ ((row >> 1) & 7)) << 4);
    }

    // ---- DMA lanes: instruction q of this wave fills half-tile rows [(wave*2 + q) * 8, +8); lane -> (row, slot); the slot holds source chunk slot ^ swz(row).
    //      Half-tile row lr of X<i> is activation row wm' * 128 + i * 64 + (lr & 63), wm' = lr >> 6; of W<j> weight row wn' * 64 + j * 32 + (lr & 31), wn' = lr >> 5.
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A), 0, (int)(((size_t)(p.M - 1) * p.lda + p.K) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.B), 0, (int)(((size_t)(p.N - 1) * p.ldb + p.K) * 2), 0x00020000);
    int xrel[2][2], wrel[2][2];                              // [quadrant][instruction]
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int lr = (wave * 2 + q) * 8 + (lane >> 3), slot = lane & 7;
        const int chunk = slot ^ ((lr >> 1) & 7);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            xrel[h][q] = ((lr >> 6) * 128 + h * 64 + (lr & 63)) * p.lda * 2 + chunk * 16;
            wrel[h][q] = ((lr >> 5) * 64 + h * 32 + (lr & 31)) * p.ldb * 2 + chunk * 16;
        }
    }
    const int nkt = p.K / BK;                                // even (K % 128 == 0)

    // ---- this workgroup's tiles: XCD x = blockIdx % 8 owns tiles [x * ipx, (x + 1) * ipx), its workgroups take them round-robin
    const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int t_lo = xcd * p.ipx, t_hi = min(p.items, t_lo + p.ipx);
    const int ndp = t_lo + slot0 < t_hi ? (t_hi - t_lo - slot0 + per_xcd - 1) / per_xcd : 0;      // whole tiles of the data-parallel rounds
    const int nmy = ndp;
    if (nmy == 0) return;
    // All workgroups start together and stay in lockstep from tile to tile: the chip alternates between "every CU multiplies" and "every CU stores its tile"
    // (fc1 + GELU + GELU': 62 MB per round, the HBM write rate fully exposed).  A workgroup that walks FEWER tiles than the busiest one has the time of a tile to
    // spare: it starts half a tile late, so that its epilogues fall into the others' K loops -- free, whenever the last round is not full.
    if (p.shift > 0 && (nmy < p.max_nmy || ((p.opt & 2) && (slot0 & 1)))) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)p.shift) __builtin_amdgcn_s_sleep(16);
    }
    auto tile_of = [&](int k, int& m0, int& n0) {
        const int t = t_lo + slot0 + k * per_xcd;
        if (p.group_m <= 1) {
            const int mt = t / p.nt;
            m0 = mt * BM; n0 = (t - mt * p.nt) * BN;
        } else {
            // groups of group_m row panels, column-major inside a group: the 32 tiles an XCD's workgroups hold at a time share
            // group_m activation panels and 32 / group_m weight panels instead of ~3 and all of them
            const int per_group = p.group_m * p.nt;
            const int gi = t / per_group, r = t - gi * per_group;
            const int gm = min(p.group_m, p.panels - gi * p.group_m);
            const int col = r / gm, mt = gi * p.group_m + (r - col * gm);
            m0 = mt * BM; n0 = col * BN;
        }
    };

    // segment k of this workgroup: a whole tile (all of K).  (Round 5's opt-in stream-K last round cut the left-over tiles into K spans here; it measured slower than
    // whole rounds + a tail launch -- profiles/r05_gemm8_notes.md section 7 -- and was removed in round 6.)
    auto seg_of = [&](int k, int& m0, int& n0, int& kb, int& kc) { tile_of(k, m0, n0); kb = 0; kc = nkt; };

    // ---- DMA cursor: half tiles are issued in the order W0 X0 W1 X1 of K tile 0, 1, ... of tile 0, 1, ...; it advances behind every X1
    // (at most two K tiles ahead of the multiplying phases and K >= 256, so it enters tile k + 1 while tile k is being multiplied: the bases of the
    // NEXT tile are computed once per tile, outside the phase code)
    int d_s = 0, d_kb, d_kc, d_abase, d_bbase, n_abase = 0, n_bbase = 0, n_kb = 0, n_kc = 2;
    bool d_live = true, n_live = false;
    {
        int m0, n0;
        seg_of(0, m0, n0, d_kb, d_kc);
        d_abase = m0 * p.lda * 2; d_bbase = n0 * p.ldb * 2;
    }
    auto dma = [&](auto sub_c, auto par_c) {
        constexpr int SUB = decltype(sub_c)::value, PAR = decltype(par_c)::value;
        char* l = smem + PAR * KTILE + SUB * HALF + wave * 2048;
        const int koff = (d_kb + d_s) * (BK * 2);
        constexpr int h = SUB >> 1;
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (SUB == SUB_W0 || SUB == SUB_W1) {
            const int v0 = d_live ? wrel[h][0] + d_bbase : OOB8, v1 = d_live ? wrel[h][1] + d_bbase : OOB8;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (lvoid_t*)(l), 16, v0, koff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (lvoid_t*)(l + 1024), 16, v1, koff, 0, 0);
        } else {
            const int v0 = d_live ? xrel[h][0] + d_abase : OOB8, v1 = d_live ? xrel[h][1] + d_abase : OOB8;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lvoid_t*)(l), 16, v0, koff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lvoid_t*)(l + 1024), 16, v1, koff, 0, 0);
        }
#else
        (void)l; (void)koff; (void)h;
#endif
        if constexpr (SUB == SUB_X1) {                      // branch-free: a branch here splits the K-tile body into basic blocks and hipcc sinks MFMAs across the barriers
            const int s1 = d_s + 1;
            const bool wrap = s1 == d_kc;
            d_s = wrap ? 0 : s1;
            d_abase = wrap ? n_abase : d_abase;
            d_bbase = wrap ? n_bbase : d_bbase;
            d_kb = wrap ? n_kb : d_kb;
            d_kc = wrap ? n_kc : d_kc;
            d_live = wrap ? n_live : d_live;
        }
    };
#define IC(v) std::integral_constant<int, (v)>{}

    // ---- epilogue of ONE 32 x 32 accumulator block (m fragment i, n half j) of a tile at (m0, n0).  D[row = n: (r & 3) + 8 (r >> 2) + 4 kh][col = m: l31]: a lane
    //      holds 4 groups of 4 consecutive n
    auto epi_block = [&](f32x16& a, int m0, int n0, int i, int j) {
        const int m = m0 + wm * 128 + i * 32 + l31;
        const bool mv = m < p.M;
        const int nb = n0 + wn * 64 + j * 32;                // this lane's groups: nb + 8 g4 + 4 kh
        if (mv && EPI != EPI_NONE) {
            // residual / multiplier operands: ONE 16-byte load per 8 outputs at the address the lane will store to (crow + pr * 16 + kh * 8), brought into
            // the accumulator layout by the inverse of the store's half-wave exchange (v_permlane32_swap is its own inverse)
            unsigned ex[2][4];                               // [pr][ax ay bx by] packed bf16 pairs of the lane's groups g4 = 2 pr (a) and 2 pr + 1 (b)
            if constexpr (EPI == EPI_BIAS_RES || EPI == EPI_MUL) {
                const bf16_t* erow = (EPI == EPI_BIAS_RES ? p.R + (size_t)m * p.ldr : p.H + (size_t)m * p.ldh) + nb;
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const u32x4 q = *reinterpret_cast<const u32x4*>(erow + pr * 16 + kh * 8);
                    auto sx = __builtin_amdgcn_permlane32_swap(q[0], q[2], false, false);
                    auto sy = __builtin_amdgcn_permlane32_swap(q[1], q[3], false, false);
                    ex[pr][0] = sx[0]; ex[pr][1] = sy[0]; ex[pr][2] = sx[1]; ex[pr][3] = sy[1];
                }
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int n = nb + 8 * g4 + 4 * kh;
                float v[4] = {a[4 * g4], a[4 * g4 + 1], a[4 * g4 + 2], a[4 * g4 + 3]};
                if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_RES || EPI == EPI_BIAS_GELU) {
                    const float4 bb = *reinterpret_cast<const float4*>(p.bias + n);
                    v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
                }
                if constexpr (EPI == EPI_BIAS_RES || EPI == EPI_MUL) {
                    float e4[4];
                    unpack4_8(make_uint2(ex[g4 >> 1][2 * (g4 & 1)], ex[g4 >> 1][2 * (g4 & 1) + 1]), e4);
                    if constexpr (EPI == EPI_BIAS_RES) { v[0] += e4[0]; v[1] += e4[1]; v[2] += e4[2]; v[3] += e4[3]; }
                    else { v[0] *= e4[0]; v[1] *= e4[1]; v[2] *= e4[2]; v[3] *= e4[3]; }
                }
                a[4 * g4] = v[0]; a[4 * g4 + 1] = v[1]; a[4 * g4 + 2] = v[2]; a[4 * g4 + 3] = v[3];
            }
        }
        float dv[16];
        if constexpr (EPI == EPI_BIAS_GELU) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f2 y, d;
                gelu_both8((f2){a[r], a[r + 1]}, y, d);
                a[r] = y.x; a[r + 1] = y.y; dv[r] = d.x; dv[r + 1] = d.y;
            }
        }
        bf16_t* crow = p.C + (size_t)m * p.ldc + nb;
        bf16_t* hrow = p.H + (size_t)m * p.ldh + nb;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            {
                unsigned ax = pack_bf16x2(a[8 * pr + 0], a[8 * pr + 1]), ay = pack_bf16x2(a[8 * pr + 2], a[8 * pr + 3]);
                unsigned bx = pack_bf16x2(a[8 * pr + 4], a[8 * pr + 5]), by = pack_bf16x2(a[8 * pr + 6], a[8 * pr + 7]);
                auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                if (mv) *reinterpret_cast<u32x4*>(crow + pr * 16 + kh * 8) = u32x4{rx[0], ry[0], rx[1], ry[1]};
            }
            if constexpr (EPI == EPI_BIAS_GELU) {
                unsigned ax = pack_bf16x2(dv[8 * pr + 0], dv[8 * pr + 1]), ay = pack_bf16x2(dv[8 * pr + 2], dv[8 * pr + 3]);
                unsigned bx = pack_bf16x2(dv[8 * pr + 4], dv[8 * pr + 5]), by = pack_bf16x2(dv[8 * pr + 6], dv[8 * pr + 7]);
                auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                if (mv && p.H != nullptr) *reinterpret_cast<u32x4*>(hrow + pr * 16 + kh * 8) = u32x4{rx[0], ry[0], rx[1], ry[1]};
            }
        }
    };

    // ---- prologue: seven half tiles in flight, K tile 0 landed and published
    dma(IC(SUB_W0), IC(0)); dma(IC(SUB_X0), IC(0)); dma(IC(SUB_W1), IC(0)); dma(IC(SUB_X1), IC(0));
    dma(IC(SUB_W0), IC(1)); dma(IC(SUB_X0), IC(1)); dma(IC(SUB_W1), IC(1));
    wait_vm8<6>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    bool aligned = true;                                     // both wave halves at the same barrier count (false while waves 4-7 run one behind)
    for (int k = 0; k < nmy; ++k) {
        int m0, n0, kb, kc;
        seg_of(k, m0, n0, kb, kc);
        (void)kb;
        n_live = k + 1 < nmy;
        if (n_live) {
            int m1, n1;
            seg_of(k + 1, m1, n1, n_kb, n_kc);
            n_abase = m1 * p.lda * 2; n_bbase = n1 * p.ldb * 2;
        }
        f32x16 acc[2][4];                                    // [n half j][m fragment 2 i + i2]
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

        bf16x8_t xf[2][4], wf0[4], wf1[4];
        if (aligned) { if (lag) __builtin_amdgcn_s_barrier(); aligned = false; }         // waves 4-7 run one barrier behind from here on
        // one K tile out of buffer PAR: four phases
#define MFMA8(J, IB, WF)                                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                                      \
        __builtin_amdgcn_s_setprio(1);                                                                                          \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                                      \
            acc[J][IB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WF[ks], xf[0][ks], acc[J][IB], 0, 0, 0);                       \
            acc[J][IB + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WF[ks], xf[1][ks], acc[J][IB + 1], 0, 0, 0);               \
        }                                                                                                                       \
        __builtin_amdgcn_s_setprio(0);                                                                                          \
        asm volatile("" : "+v"(acc[J][IB]), "+v"(acc[J][IB + 1]));                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                                      \
        __builtin_amdgcn_s_barrier();                                                                                           \
        asm volatile("" ::: "memory");
#define KTILE8(PAR)                                                                                                             \
        {                                                                                                                       \
            const char* sb = smem + (PAR) * KTILE;                                                                              \
            /* phase 0: W0 (retired before the barrier: its region is restaged in phase 1), X0; X1 of the next K tile */        \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) wf0[ks] = ldsf(sb + SUB_W0 * HALF + waddr[ks]);                    \
            __builtin_amdgcn_sched_barrier(0);                                                                                  \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                                  \
                xf[0][ks] = ldsf(sb + SUB_X0 * HALF + xaddr[0][ks]);                                                            \
                xf[1][ks] = ldsf(sb + SUB_X0 * HALF + xaddr[1][ks]);                                                            \
            }                                                                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                                                  \
            dma(IC(SUB_X1), IC((PAR) ^ 1));                                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                                  \
            wait_lgkm8<8>();                                                                                                    \
            __builtin_amdgcn_s_barrier();                                                                                       \
            wait_lgkm8<0>();                                                                                                    \
            MFMA8(0, 0, wf0)                                                                                                    \
            /* phase 1: W1; W0 of the K tile after the next */                                                                  \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) wf1[ks] = ldsf(sb + SUB_W1 * HALF + waddr[ks]);                    \
            __builtin_amdgcn_sched_barrier(0);                                                                                  \
            dma(IC(SUB_W0), IC(PAR));                                                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                                  \
            __builtin_amdgcn_s_barrier();                                                                                       \
            wait_lgkm8<0>();                                                                                                    \
            MFMA8(1, 0, wf1)                                                                                                    \
            /* phase 2: X1; X0 of the K tile after the next */                                                                  \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                                  \
                xf[0][ks] = ldsf(sb + SUB_X1 * HALF + xaddr[0][ks]);                                                            \
                xf[1][ks] = ldsf(sb + SUB_X1 * HALF + xaddr[1][ks]);                                                            \
            }                                                                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                                                  \
            dma(IC(SUB_X0), IC(PAR));                                                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                                  \
            __builtin_amdgcn_s_barrier();                                                                                       \
            wait_lgkm8<0>();                                                                                                    \
            MFMA8(1, 2, wf1)                                                                                                    \
            /* phase 3: no reads; W1 of the K tile after the next; the whole next K tile has landed */                          \
            dma(IC(SUB_W1), IC(PAR));                                                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                                  \
            wait_vm8<6>();                                                                                                      \
            __builtin_amdgcn_s_barrier();                                                                                       \
            MFMA8(0, 2, wf0)                                                                                                    \
        }

        for (int s = 0; s < kc; s += 2) {
            KTILE8(0)
            KTILE8(1)
        }
#undef KTILE8
#undef MFMA8
        if (k == nmy - 1 || (p.opt & 1)) { if (!lag) __builtin_amdgcn_s_barrier(); aligned = true; }      // the barrier waves 4-7 still owe

        // ---- epilogue of a whole tile
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) epi_block(acc[j][i], m0, n0, i, j);
    }

    wait_vm8<0>();                                           // the zero-fill requests behind the last tile write LDS too: nothing may be in flight when the wave ends
#undef IC
}

template <int EPI>
int launch8(const Gemm8Params& p, hipStream_t st) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    static bool attr[16] = {};
    if (dev < 0 || dev >= 16 || !attr[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS8) != hipSuccess) {
            clhip_set_error("gemm8: cannot reserve %d bytes of LDS", LDS8);
            return CLHIP_EHIP;
        }
        if (dev >= 0 && dev < 16) attr[dev] = true;
    }
    static const int force_grid = clhip_cfg("GEMM8_GRID") ? atoi(clhip_cfg("GEMM8_GRID")) : 0;
    int grid = force_grid > 0 ? force_grid : 256;
    if (grid > (p.items + 7) / 8 * 8) grid = (p.items + 7) / 8 * 8;
    grid = (grid + 7) / 8 * 8;
    hipLaunchKernelGGL(gemm8_kernel<EPI>, dim3(grid), dim3(512), LDS8, st, p);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

}  // namespace

// bf16, N % 256 == 0, K % 128 == 0, operands below 2 GB.  Mode 0 never, 1 (DEFAULT) the shapes it is picked for (see below), 2 whenever legal (tests,
// micro-benchmarks).  CLHIP_GEMM8 / clhip_gemm8_config.
// Measured (profiles/r05_gemm8_notes.md): stand-alone at M = 25216 qkv 126 -> 101-108 us, fc1 152 -> 126-130, fc2 dX 145 -> 123, 4096^3 887 -> 1255 TF/s;
// inside the steps InfLoRA_OPT batch 128 15.72 -> 15.40 ms, batch 256 30.30 -> 29.19, L2P batch 256 48.37 -> 45.60 (same-box alternating runs) -- less
// than the launches' own gain: with this kernel in the step EVERY other kernel of the step runs 7-11 % longer (the chip holds a lower clock).
// Returns the number of leading rows of the product this kernel should compute (0: none; M: all).  256 persistent workgroups walk the
// 256 x 256 tiles in rounds, so a partial last round idles most of the chip (M = 25216, N = 768: 297 tiles = 2 rounds at 58 %): in mode 1
// the kernel takes the row panels that fill WHOLE rounds and the caller hands the remaining rows to the register-staged kernel, whose
// small tiles fill the chip with what is left (CLHIP_GEMM8_SPLIT=0: all rows or none, by the fill of the last round).
static int g_mode8 = -1;
int clhip_gemm8_rows(int M, int N, int K, int lda, int ldb, int ldc, int ldr, int ldh, int dtype) {
    if (g_mode8 < 0) g_mode8 = clhip_cfg("GEMM8") ? atoi(clhip_cfg("GEMM8")) : 1;
    if (g_mode8 == 0 || dtype != CLHIP_BF16) return 0;
    if (N % 256 != 0 || K % 128 != 0 || K < 256 || lda % 8 != 0 || ldb % 8 != 0 || ldc % 8 != 0 || ldr % 8 != 0 || ldh % 8 != 0) return 0;
    if ((long long)M * lda * 2 >= (1ll << 31) - (1 << 20) || (long long)N * ldb * 2 >= (1ll << 31) - (1 << 20)) return 0;
    if (g_mode8 == 2) return M;
    static const int min_n = clhip_cfg("GEMM8_MINN") ? atoi(clhip_cfg("GEMM8_MINN")) : 0;
    if (N < min_n) return 0;
    static const int split = clhip_cfg("GEMM8_SPLIT") ? atoi(clhip_cfg("GEMM8_SPLIT")) : 1;
    static const int min_fill = clhip_cfg("GEMM8_FILL") ? atoi(clhip_cfg("GEMM8_FILL")) : 90;      // per cent of the last round
    const int nt = N / 256, panels = (M + 255) / 256;
    const long tiles = (long)panels * nt;
    const long rounds = (tiles + 255) / 256;
    // a single round of 12 K tiles (proj at batch 128: 24 us of tile latency + the tail's launch) does not beat the register-staged kernel's 47 us
    static const int min_work = clhip_cfg("GEMM8_MINWORK") ? atoi(clhip_cfg("GEMM8_MINWORK")) : 24;      // whole rounds x K tiles
    if (tiles * 100 >= rounds * 256 * min_fill) return rounds * (K / 64) >= min_work ? M : 0;           // the last round is (nearly) full
    if (!split || tiles < 256) return 0;
    if ((tiles / 256) * (K / 64) < min_work) return 0;
    const int full_panels = (int)((tiles / 256) * 256 / nt);               // whole rounds (the last panel of a round may leave a few tiles unused)
    return full_panels * 256;
}
extern "C" void clhip_gemm8_config(int mode) { g_mode8 = mode; }

int clhip_gemm8_launch(const void* A, const void* B, void* C, const float* bias, const void* R, void* H, int M, int N, int K,
                       int lda, int ldb, int ldc, int ldr, int ldh, int epilogue, hipStream_t st) {
    Gemm8Params p{static_cast<const bf16_t*>(A), static_cast<const bf16_t*>(B), static_cast<bf16_t*>(C), bias, static_cast<const bf16_t*>(R),
                  static_cast<bf16_t*>(H), M, N, K, lda, ldb, ldc, ldr, ldh, 0, 0, 0, 1, 0, 0, 0, 0};
    static const int opt = clhip_cfg("GEMM8_OPT") ? atoi(clhip_cfg("GEMM8_OPT")) : 0;
    p.opt = opt;
    p.nt = N / 256;
    p.items = ((M + 255) / 256) * p.nt;
    p.ipx = (p.items + 7) / 8;
    p.panels = (M + 255) / 256;
    static const int group_m = clhip_cfg("GEMM8_GROUP") ? atoi(clhip_cfg("GEMM8_GROUP")) : 1;
    p.group_m = group_m;
    {
        int grid = 256;
        if (grid > (p.items + 7) / 8 * 8) grid = (p.items + 7) / 8 * 8;
        const int per_xcd = grid / 8;
        p.max_nmy = (p.ipx + per_xcd - 1) / per_xcd;
        static const int shift_kt = clhip_cfg("GEMM8_SHIFT") ? atoi(clhip_cfg("GEMM8_SHIFT")) : 2000;      // shader cycles per K tile: about half of what a K tile takes
        p.shift = p.max_nmy > 1 ? (K / 64) * shift_kt : 0;
    }
    switch (epilogue) {
        case EPI_NONE: return launch8<EPI_NONE>(p, st);
        case EPI_BIAS: return launch8<EPI_BIAS>(p, st);
        case EPI_BIAS_RES: return launch8<EPI_BIAS_RES>(p, st);
        case EPI_BIAS_GELU: return launch8<EPI_BIAS_GELU>(p, st);
        case EPI_MUL: return launch8<EPI_MUL>(p, st);
    }
    clhip_set_error("gemm8: unknown epilogue %d", epilogue);
    return CLHIP_EINVAL;
}
