// gemm5.hip -- C[M,N] = epi(A[M,K] . B[N,K]^T), bf16, gfx950: the 256 x 256 tile kernel of the ViT path's large GEMMs.
//
// What round 1's gemm_nt_kernel (gemm.hip) leaves on the table against the vendor library on the ViT-B/16 shapes
// (tools/gemm_vs_blas.py, profiles/r02_gemm5_notes.md): 1.04-1.35x at M = 25216, 1.7x at 4096^3.  Its profile
// (profiles/r01_gemm_ablation.txt) says the operand path global -> registers -> LDS -> registers bounds the K loop, not the matrix
// pipe.  This kernel is conv4.hip's machinery with both operands streamed:
//
//  * v_mfma_f32_32x32x16_bf16; a wave owns 128 (m) x 64 (n) = 4 x 2 MFMA tiles, 128 accumulator registers: 6 fragment reads per
//    8 MFMAs.  The weight fragment is the A operand, so a lane holds 4 consecutive n per accumulator group and the half-waves are
//    paired with v_permlane32_swap into 16-byte row-segment stores (no LDS in the epilogue).
//  * both operands of a 32-deep K slab (256 rows x 64 bytes each) are streamed by LDS-DMA through buffer descriptors (rows behind
//    M / N are zero-filled by the range check) into a 4-stage ring THREE slabs ahead; counted s_waitcnt vmcnt(8), raw s_barrier.
//    The DMA writes lane-linearly; the XOR swizzle that makes the 32-row ds_read_b128 conflict-free is applied to the per-lane
//    SOURCE address and to the read address.
//  * eight waves, two per SIMD: waves 4-7 run one barrier behind waves 0-3 and the K loop alternates a read phase (12 fragment
//    reads, 4 DMA instructions, the DMA wait) with an MFMA phase (16 MFMAs), a barrier after each -- on every SIMD one wave
//    multiplies while its partner loads (conv4.hip's staggered halves).
//  * workgroups are persistent (one per CU) and walk a list of tiles; the ring runs across tile boundaries, so only a workgroup's
//    first slabs are exposed.  Tiles are dealt so that the workgroups of one XCD work on neighbouring tiles of the same row panels.
//
// Epilogues as gemm.hip: none / + bias / + bias + residual / + bias, exact GELU and its derivative / * H.  Replaces F.linear on the
// ViT path (core/model/backbone/transformer.py:172, 194, 1259 ...) where M, N are large; everything else stays on gemm_nt_kernel.
#include <stdlib.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void lvoid_t;
typedef __attribute__((ext_vector_type(2))) float f2;

enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_RES = 2, EPI_BIAS_GELU = 3, EPI_MUL = 4 };

struct Gemm5Params {
    const bf16_t* A; const bf16_t* B; bf16_t* C;
    const float* bias; const bf16_t* R; bf16_t* H;
    int M, N, K, lda, ldb, ldc, ldr, ldh;
    int nt, items, ipx;          // n tiles, tiles, tiles per XCD
    unsigned long long* trace;   // CLHIP_ABLATION builds: s_memtime stamps of waves 0 and 4 of workgroup 0 ([2][256])
    int debug;                   // CLHIP_ABLATION builds: 1 no MFMA, 2 no DMA in the loop, 4 no fragment reads, 8 no C stores
};

unsigned long long* g_trace5 = nullptr;
int g_debug5 = 0;
#ifdef CLHIP_ABLATION
#define DBG5(p) ((p).debug)
#else
#define DBG5(p) 0
#endif
#ifdef CLHIP_ABLATION
#define STAMP5() do { if (p.trace && blockIdx.x == 0 && lane == 0 && (wave & 3) == 0 && nstamp < 256) p.trace[(wave >> 2) * 256 + nstamp++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP5() do { } while (0)
#endif

constexpr int BM = 256, BN = 256, BK = 32, NST = 4;
constexpr int PART = 256 * BK * 2;            // one operand of a slab: 256 rows x 64 bytes
constexpr int STAGE = 2 * PART;
constexpr int WINST = 4;                      // DMA instructions per wave per slab (2 of A, 2 of B; 16 rows each)
constexpr int LDS5 = NST * STAGE;             // 131072

template <int N> __device__ __forceinline__ void wait_vm5() { __builtin_amdgcn_s_waitcnt((N & 15) | 0x70 | 0xF00 | ((N >> 4) << 14)); }
__device__ __forceinline__ void wait_lds5() { __builtin_amdgcn_s_waitcnt(0xC07F); }

// exact (erf) GELU and its derivative, two elements at a time (see gemm.hip: one exp serves both)
__device__ __forceinline__ void gelu_both5(f2 x, f2& y, f2& dy) {
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

__device__ __forceinline__ void unpack4(uint2 v, float (&f)[4]) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm5_kernel(const Gemm5Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                 // wave tile: rows wm*128.., columns wn*64..; waves w and w+4 share a SIMD
    const bool lag = wave >= 4;
    const int l31 = lane & 31, kh = lane >> 5;
    int nstamp = 0; (void)nstamp;
    STAMP5();

    // ---- fragment addresses inside a stage: row * 64 + ((2 ks + kh) ^ swz(row)) * 16, swz(row) = (row >> 2) & 3
    int xaddr[4][2], waddr[2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wm * 128 + i * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) xaddr[i][ks] = row * 64 + (((2 * ks + kh) ^ ((row >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = wn * 64 + j * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) waddr[j][ks] = PART + row * 64 + (((2 * ks + kh) ^ ((row >> 2) & 3)) << 4);
    }

    // ---- DMA lanes: instruction q of this wave fills rows [(wave*2 + (q&1)) * 16, +16) of A (q < 2) or B (q >= 2); lane -> (row, slot);
    //      the slot holds source chunk slot ^ swz(row)
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A), 0, (int)(((size_t)(p.M - 1) * p.lda + p.K) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.B), 0, (int)(((size_t)(p.N - 1) * p.ldb + p.K) * 2), 0x00020000);
    int arel[2], brel[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = (wave * 2 + q) * 16 + (lane >> 2), slot = lane & 3;
        const int chunk = slot ^ ((row >> 2) & 3);
        arel[q] = row * p.lda * 2 + chunk * 16;
        brel[q] = row * p.ldb * 2 + chunk * 16;
    }
    const int nslab = p.K / BK;

    // ---- this workgroup's tiles: XCD x = blockIdx % 8 owns tiles [x * ipx, (x + 1) * ipx), its workgroups take them round-robin
    const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int t_lo = xcd * p.ipx, t_hi = min(p.items, t_lo + p.ipx);
    const int nmy = t_lo + slot0 < t_hi ? (t_hi - t_lo - slot0 + per_xcd - 1) / per_xcd : 0;
    auto tile_of = [&](int k, int& m0, int& n0) {
        const int t = t_lo + slot0 + k * per_xcd;
        const int mt = t / p.nt;
        m0 = mt * BM; n0 = (t - mt * p.nt) * BN;
    };
    // global slab counter g = k * nslab + s of this workgroup; stage = g % 4.  The DMA cursor walks the slabs in order (no divisions
    // in the loop: the tile bases are recomputed once per tile)
    int d_s = 0, d_k = 0, d_abase = 0, d_bbase = 0;
    {
        int m0, n0;
        tile_of(0, m0, n0);
        d_abase = m0 * p.lda * 2; d_bbase = n0 * p.ldb * 2;
    }
    auto dma = [&](int g) {                                  // called with g = 0, 1, 2, ... exactly once each
        const int koff = d_s * (BK * 2);
        char* l = smem + (g & (NST - 1)) * STAGE + wave * 2048;
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lvoid_t*)(l), 16, arel[0] + d_abase, koff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lvoid_t*)(l + 1024), 16, arel[1] + d_abase, koff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (lvoid_t*)(l + PART), 16, brel[0] + d_bbase, koff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (lvoid_t*)(l + PART + 1024), 16, brel[1] + d_bbase, koff, 0, 0);
#else
        (void)koff; (void)l;
#endif
        if (++d_s == nslab) {
            d_s = 0; ++d_k;
            if (d_k < nmy) {
                int m0, n0;
                tile_of(d_k, m0, n0);
                d_abase = m0 * p.lda * 2; d_bbase = n0 * p.ldb * 2;
            }
        }
    };

    const int gtot = nmy * nslab;
    if (gtot == 0) return;
    // prologue: slabs 0, 1, 2 in flight; slab 0 landed and published
    dma(0);
    if (gtot > 1) dma(1);
    if (gtot > 2) dma(2);
    if (gtot > 2) wait_vm5<2 * WINST>(); else if (gtot > 1) wait_vm5<WINST>(); else wait_vm5<0>();
    wait_lds5();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int g = 0;
    for (int k = 0; k < nmy; ++k) {
        int m0, n0;
        tile_of(k, m0, n0);
        f32x16 acc[2][4];                                    // [n tile j][m tile i]
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

        if (lag) __builtin_amdgcn_s_barrier();
        for (int s = 0; s < nslab; ++s, ++g) {            // ---- read phase: the 12 fragments of slab g, DMA of slab g + 3 into the stage slab g - 1 has left, wait for slab g + 1
            const char* sb = smem + (g & (NST - 1)) * STAGE;
            bf16x8_t xf[2][4], wf[2][2];
            if (!(DBG5(p) & 4))
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int j = 0; j < 2; ++j) wf[ks][j] = *reinterpret_cast<const bf16x8_t*>(sb + waddr[j][ks]);
#pragma unroll
                for (int i = 0; i < 4; ++i) xf[ks][i] = *reinterpret_cast<const bf16x8_t*>(sb + xaddr[i][ks]);
            }
            STAMP5();
            if (DBG5(p) & 2) { }
            else if (g + 3 < gtot) { dma(g + 3); wait_vm5<2 * WINST>(); }
            else if (g + 2 < gtot) wait_vm5<WINST>();
            else wait_vm5<0>();
            __builtin_amdgcn_sched_barrier(0);
            STAMP5();
            wait_lds5();                                     // the fragments have landed before the barrier: the other half may then
            STAMP5();
            __builtin_amdgcn_s_barrier();                    // overwrite the stage this half has just read
            STAMP5();
            // ---- MFMA phase
            __builtin_amdgcn_sched_barrier(0);
            if (DBG5(p) & 1) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(wf[ks][j]));
#pragma unroll
                    for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(xf[ks][i]));
                }
            } else
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][j], xf[ks][i], acc[j][i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            STAMP5();
            __builtin_amdgcn_s_barrier();
            STAMP5();
        }
        if (!lag) __builtin_amdgcn_s_barrier();              // both halves aligned again

        // ---- epilogue.  D[row = n: (r & 3) + 8 (r >> 2) + 4 kh][col = m: l31]; per (j, i) a lane holds 4 groups of 4 consecutive n
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 128 + i * 32 + l31;
            const bool mv = m < p.M;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nb = n0 + wn * 64 + j * 32;        // this lane's groups: nb + 8 g4 + 4 kh
                if (mv && EPI != EPI_NONE) {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int n = nb + 8 * g4 + 4 * kh;
                        float v[4] = {acc[j][i][4 * g4], acc[j][i][4 * g4 + 1], acc[j][i][4 * g4 + 2], acc[j][i][4 * g4 + 3]};
                        if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_RES || EPI == EPI_BIAS_GELU) {
                            const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
                            v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                        }
                        if constexpr (EPI == EPI_BIAS_RES) {
                            float r4[4];
                            unpack4(*reinterpret_cast<const uint2*>(p.R + (size_t)m * p.ldr + n), r4);
                            v[0] += r4[0]; v[1] += r4[1]; v[2] += r4[2]; v[3] += r4[3];
                        }
                        if constexpr (EPI == EPI_MUL) {
                            float h4[4];
                            unpack4(*reinterpret_cast<const uint2*>(p.H + (size_t)m * p.ldh + n), h4);
                            v[0] *= h4[0]; v[1] *= h4[1]; v[2] *= h4[2]; v[3] *= h4[3];
                        }
                        acc[j][i][4 * g4] = v[0]; acc[j][i][4 * g4 + 1] = v[1]; acc[j][i][4 * g4 + 2] = v[2]; acc[j][i][4 * g4 + 3] = v[3];
                    }
                }
                float dv[16];
                if constexpr (EPI == EPI_BIAS_GELU) {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        f2 y, d;
                        gelu_both5((f2){acc[j][i][r], acc[j][i][r + 1]}, y, d);
                        acc[j][i][r] = y.x; acc[j][i][r + 1] = y.y; dv[r] = d.x; dv[r + 1] = d.y;
                    }
                }
                bf16_t* crow = p.C + (size_t)m * p.ldc + nb;
                bf16_t* hrow = p.H + (size_t)m * p.ldh + nb;
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    {
                        unsigned ax = pack_bf16x2(acc[j][i][8 * pr + 0], acc[j][i][8 * pr + 1]), ay = pack_bf16x2(acc[j][i][8 * pr + 2], acc[j][i][8 * pr + 3]);
                        unsigned bx = pack_bf16x2(acc[j][i][8 * pr + 4], acc[j][i][8 * pr + 5]), by = pack_bf16x2(acc[j][i][8 * pr + 6], acc[j][i][8 * pr + 7]);
                        auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                        auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                        if (mv && !(DBG5(p) & 8)) *reinterpret_cast<u32x4*>(crow + pr * 16 + kh * 8) = u32x4{rx[0], ry[0], rx[1], ry[1]};
                    }
                    if constexpr (EPI == EPI_BIAS_GELU) {
                        unsigned ax = pack_bf16x2(dv[8 * pr + 0], dv[8 * pr + 1]), ay = pack_bf16x2(dv[8 * pr + 2], dv[8 * pr + 3]);
                        unsigned bx = pack_bf16x2(dv[8 * pr + 4], dv[8 * pr + 5]), by = pack_bf16x2(dv[8 * pr + 6], dv[8 * pr + 7]);
                        auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                        auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                        if (mv && p.H != nullptr) *reinterpret_cast<u32x4*>(hrow + pr * 16 + kh * 8) = u32x4{rx[0], ry[0], rx[1], ry[1]};
                    }
                }
            }
        }
    }
}

template <int EPI>
int launch5(const Gemm5Params& p, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm5_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS5) != hipSuccess) {
            clhip_set_error("gemm5: cannot reserve %d bytes of LDS", LDS5);
            return CLHIP_EHIP;
        }
        attr = true;
    }
    static const int force_grid = clhip_cfg("GEMM5_GRID") ? atoi(clhip_cfg("GEMM5_GRID")) : 0;
    int grid = force_grid > 0 ? force_grid : 256;
    if (grid > (p.items + 7) / 8 * 8) grid = (p.items + 7) / 8 * 8;
    grid = (grid + 7) / 8 * 8;
    hipLaunchKernelGGL(gemm5_kernel<EPI>, dim3(grid), dim3(512), LDS5, st, p);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

}  // namespace

// bf16, N % 256 == 0, K % 32 == 0, operands below 2 GB.  Mode 1 picks the shapes where the kernel beats gemm_nt_kernel STAND-ALONE
// (profiles/r02_gemm5_notes.md: rotating operand sets): wide outputs (N >= 2048: the ViT's qkv / fc1 / fc2-backward GEMMs) with at least 3/4 of a
// round of tiles and few enough column tiles for an XCD's workgroups to share row panels in L2; mode 2 forces it (tests, micro-benchmarks), mode 0
// disables it.  CLHIP_GEMM5 / clhip_gemm5_config.
// DEFAULT since the end of round 4: mode 0.  Inside the ViT steps (operands hot from the previous launch, the step's epilogues) the register-staged
// kernel with its whole-round / tail split wins on every workload measured -- InfLoRA_OPT batch 128 16.5 -> 16.1 ms, batch 256 31.95 -> 31.05, L2P batch
// 256 50.5 -> 49.1, batch 16 unchanged (two alternating same-box runs each) -- one 8-wave workgroup per CU leaves nothing to cover its epilogue.
static int g_mode5 = -1;
bool clhip_gemm5_supported(int M, int N, int K, int lda, int ldb, int ldc, int ldr, int ldh, int dtype) {
    if (g_mode5 < 0) g_mode5 = clhip_cfg("GEMM5") ? atoi(clhip_cfg("GEMM5")) : 0;
    if (g_mode5 == 0 || dtype != CLHIP_BF16) return false;
    if (N % 256 != 0 || K % 32 != 0 || K < 128 || lda % 8 != 0 || ldb % 8 != 0 || ldc % 8 != 0 || ldr % 4 != 0 || ldh % 8 != 0) return false;
    if ((long long)M * lda * 2 >= (1ll << 31) || (long long)N * ldb * 2 >= (1ll << 31)) return false;
    if (g_mode5 == 2) return true;
    const long tiles = (long)((M + 255) / 256) * (N / 256);
    return N >= 2048 && N / 256 <= 16 && tiles >= 192;
}
extern "C" void clhip_gemm5_config(int mode) { g_mode5 = mode; }

int clhip_gemm5_launch(const void* A, const void* B, void* C, const float* bias, const void* R, void* H, int M, int N, int K,
                       int lda, int ldb, int ldc, int ldr, int ldh, int epilogue, hipStream_t st) {
    Gemm5Params p{static_cast<const bf16_t*>(A), static_cast<const bf16_t*>(B), static_cast<bf16_t*>(C), bias, static_cast<const bf16_t*>(R),
                  static_cast<bf16_t*>(H), M, N, K, lda, ldb, ldc, ldr, ldh, 0, 0, 0, g_trace5, g_debug5};
    p.nt = N / 256;
    p.items = ((M + 255) / 256) * p.nt;
    p.ipx = (p.items + 7) / 8;
    switch (epilogue) {
        case EPI_NONE: return launch5<EPI_NONE>(p, st);
        case EPI_BIAS: return launch5<EPI_BIAS>(p, st);
        case EPI_BIAS_RES: return launch5<EPI_BIAS_RES>(p, st);
        case EPI_BIAS_GELU: return launch5<EPI_BIAS_GELU>(p, st);
        case EPI_MUL: return launch5<EPI_MUL>(p, st);
    }
    clhip_set_error("gemm5: unknown epilogue %d", epilogue);
    return CLHIP_EINVAL;
}

// phase stamps of workgroup 0 (ablation build only; tools/ubench/gemm_bench trace)
void clhip_gemm5_set_trace(unsigned long long* dev_buf) { g_trace5 = dev_buf; }
void clhip_gemm5_set_debug(int bits) { g_debug5 = bits; }
