// gemm.hip -- C[M,N] = epilogue(A[M,K] . B[N,K]^T) for the ViT path (gfx950 MFMA; bf16 = performance mode,
// fp32 = parity mode through the exact fp32 MFMA, same kernel text).
//
// Every linear layer of the frozen ViT (reference core/model/backbone/transformer.py:172 qkv, :194 proj,
// :1267-1271 fc1/fc2, timm PatchEmbed) and every activation-gradient product of its backward (dX = dY . W) is an
// "NT" product once the frozen weight is kept in both orientations ([out,in] for the forward, [in,out] for dX),
// so ONE kernel serves forward and backward; the epilogue carries the elementwise work the reference runs as
// separate memory passes (bias add, residual add, exact GELU and -- saved for the backward -- its derivative).
//
// Tiling: (32 MT)x128 output tile per 256-thread block (2x2 waves, each (16 MT)x64 = MT x 4 MFMA 16x16 tiles; MT = 4 by default), K step 64,
// register-staged double-buffered LDS (one barrier per K step), XOR-swizzled 128-B rows so ds_read_b128 is
// bank-conflict free.  The MFMA "A" operand is the B (weight) tile, so a lane ends up with 4 consecutive n of
// one output row m -> 8-byte (bf16) / 16-byte (fp32) row-contiguous stores and bias / residual loads.
// Block ids are remapped so each XCD (own L2) walks a contiguous range of row panels over all column panels.
#include <stdlib.h>

#include <mutex>

#include "common.h"

namespace {

enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_RES = 2, EPI_BIAS_GELU = 3, EPI_MUL = 4 };

struct GemmParams {
    const void* A; const void* B; void* C;
    const float* bias; const void* R; void* H;
    int M, N, K, lda, ldb, ldc, ldr, ldh;
    int group_m;      // row panels per rasterisation group (gemm_nt_kernel), 1 = plain row-major
    // split-K (small M, long K: too few output tiles to fill the chip): blockIdx.y owns K range [y, y + 1) * K / ksplit and leaves its raw fp32 accumulators
    // in part[y][M][N]; splitk_epilogue_kernel sums the slices in index order and applies the epilogue -- bitwise reproducible
    int ksplit = 1;
    float* part = nullptr;
};

template <typename T> struct Chunk;
template <> struct Chunk<bf16_t> { uint4 a; };
template <> struct Chunk<float> { uint4 a, b; };

template <typename T> __device__ __forceinline__ Chunk<T> czero() {
    Chunk<T> c;
    c.a = make_uint4(0, 0, 0, 0);
    if constexpr (sizeof(T) == 4) c.b = make_uint4(0, 0, 0, 0);
    return c;
}
template <typename T> __device__ __forceinline__ Chunk<T> cload(const T* p) {
    Chunk<T> c;
    c.a = *reinterpret_cast<const uint4*>(p);
    if constexpr (sizeof(T) == 4) c.b = *reinterpret_cast<const uint4*>(p + 4);
    return c;
}
// LDS rows hold 8 chunks (64 elements).  bf16: 128-B rows, chunk index XORed with row&7.  fp32: 256-B rows, linear.
template <typename T> __device__ __forceinline__ int lds_off(int row, int chunk) {
    if constexpr (sizeof(T) == 2) return row * 128 + ((chunk ^ (row & 7)) << 4);
    else return row * 256 + (chunk << 5);
}
template <typename T> __device__ __forceinline__ void lds_st(char* base, int off, const Chunk<T>& c) {
    *reinterpret_cast<uint4*>(base + off) = c.a;
    if constexpr (sizeof(T) == 4) *reinterpret_cast<uint4*>(base + off + 16) = c.b;
}
template <typename T> __device__ __forceinline__ Chunk<T> lds_ld(const char* base, int off) {
    Chunk<T> c;
    c.a = *reinterpret_cast<const uint4*>(base + off);
    if constexpr (sizeof(T) == 4) c.b = *reinterpret_cast<const uint4*>(base + off + 16);
    return c;
}
template <typename T> __device__ __forceinline__ f32x4 mma(const Chunk<T>& a, const Chunk<T>& b, f32x4 acc) {
    if constexpr (sizeof(T) == 2) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a.a), __builtin_bit_cast(bf16x8_t, b.a), acc, 0, 0, 0);
    } else {
        const float* x = reinterpret_cast<const float*>(&a);
        const float* y = reinterpret_cast<const float*>(&b);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j], y[j], acc, 0, 0, 0);
        return acc;
    }
}

template <typename T> __device__ __forceinline__ void load4(const T* p, float (&v)[4]);
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float (&v)[4]) {
    uint2 u = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
}
template <> __device__ __forceinline__ void load4<float>(const float* p, float (&v)[4]) {
    float4 a = *reinterpret_cast<const float4*>(p);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float (&v)[4]);
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float (&v)[4]) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
}
template <> __device__ __forceinline__ void store4<float>(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}

// exact (erf) GELU of nn.GELU (transformer.py:1259): gelu = x Phi(x), gelu' = Phi(x) + x phi(x).  With z = |x|/sqrt(2) the
// Gaussian factor of Abramowitz-Stegun 7.1.26 (erf(z) = 1 - poly(t) exp(-z^2), |err| <= 1.5e-7, t = 1/(1 + p z)) is
// exp(-x^2/2) = sqrt(2 pi) phi(x): ONE exp serves both, and the forward epilogue emits gelu AND gelu' (the backward
// epilogue is then a plain multiply) for ~20 VALU per element instead of two libm erff/expf calls.
__device__ __forceinline__ void gelu_both(float x, float& y, float& dy) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __frcp_rn(fmaf(0.3275911f, z, 1.0f));
    const float e = __expf(-z * z);
    float pl = fmaf(t, 1.061405429f, -1.453152027f);
    pl = fmaf(t, pl, 1.421413741f);
    pl = fmaf(t, pl, -0.284496736f);
    pl = fmaf(t, pl, 0.254829592f);
    const float erf_abs = fmaf(-pl * t, e, 1.0f);
    const float phi = 0.5f * (1.0f + copysignf(erf_abs, x));
    y = x * phi;
    dy = fmaf(x * 0.3989422804014327f, e, phi);
}

// two elements at a time on <2 x float>: the FMA-type operations become v_pk_fma_f32 / v_pk_mul_f32 (half the issue slots);
// the two transcendental ops per element (v_rcp_f32, v_exp_f32) stay scalar
typedef __attribute__((ext_vector_type(2))) float f2;
__device__ __forceinline__ void gelu_both2(f2 x, f2& y, f2& dy) {
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

// epilogue of one accumulator: the lane holds C[m][n .. n+3]
template <typename T, int EPI>
__device__ __forceinline__ void epi_store(const GemmParams& p, const f32x4& a, int m, int n) {
    float v[4] = {a[0], a[1], a[2], a[3]};
    if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_RES || EPI == EPI_BIAS_GELU) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    if constexpr (EPI == EPI_BIAS_RES) {
        float r[4];
        load4<T>(static_cast<const T*>(p.R) + (size_t)m * p.ldr + n, r);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += r[e];
    }
    if constexpr (EPI == EPI_BIAS_GELU) {
        float d[4];
        f2 y0, d0, y1, d1;
        gelu_both2((f2){v[0], v[1]}, y0, d0);
        gelu_both2((f2){v[2], v[3]}, y1, d1);
        v[0] = y0.x; v[1] = y0.y; v[2] = y1.x; v[3] = y1.y;
        d[0] = d0.x; d[1] = d0.y; d[2] = d1.x; d[3] = d1.y;
        if (p.H) store4<T>(static_cast<T*>(p.H) + (size_t)m * p.ldh + n, d);
    }
    if constexpr (EPI == EPI_MUL) {
        float h[4];
        load4<T>(static_cast<const T*>(p.H) + (size_t)m * p.ldh + n, h);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= h[e];
    }
    store4<T>(static_cast<T*>(p.C) + (size_t)m * p.ldc + n, v);
}

constexpr int BK = 64;

// WM x WN waves per workgroup, each wave MT x NT MFMA tiles of 16 x 16: block tile (16 MT WM) x (16 NT WN).
//   2x2 waves, 4x4 tiles  = 128 x 128, two workgroups per CU: the default
//   2x2 waves, 5x4 / 2x2  = 160 x 128 / 64 x 64: better tile-count quantisation / problems too small to fill the chip
//   2x4 waves, 8x4 tiles  = 256 x 256, one 8-wave workgroup per CU: half the global->LDS and LDS->register traffic per MFMA of the
//                           128 x 128 tile.  Ablation of the 128 x 128 kernel (profiles/r01_gemm_ablation.txt): removing the MFMAs
//                           does not shorten it, removing the global loads / LDS stores / fragment reads does -- the operand
//                           delivery path, not the matrix cores, bounds the loop -- so the tile must grow where the shape allows
//                           (N >= 2304: enough tiles to keep 256 CUs busy).
template <typename T, int EPI, int MT, int NT, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, (sizeof(T) == 2 && WM * WN == 4) ? 2 : 1) void gemm_nt_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NTH = 64 * WM * WN, RPP = NTH / 8;                 // threads, tile rows covered per staging pass
    constexpr int BM = 16 * MT * WM, BN = 16 * NT * WN;
    constexpr int AL = BM / RPP, BL = BN / RPP;                       // staging chunks per thread
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must divide among the staging passes");
    constexpr int A_BYTES = BM * BK * (int)sizeof(T), B_BYTES = BN * BK * (int)sizeof(T), STAGE = A_BYTES + B_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware bijective remap of the block id (8 XCDs, round-robin dispatch): XCD x gets a contiguous range
    const int tiles_n = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int wg;
    {
        const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    // within an XCD's range the tiles are walked in groups of GM row panels, column-major inside a group: the ~32 workgroups an
    // XCD runs at a time then cover a GM x (32 / GM) block of tiles and share GM + 32 / GM operand panels in its L2 instead of
    // 1 + 32 (row-major order on 8192^3: L2 hit rate 50 %, 4.6 GB of misses per launch -- profiles/r01_gemm_ablation.txt)
    int tm, tn;
    {
        const int GM = p.group_m;
        const int tiles_m = (p.M + BM - 1) / BM;
        const int per = GM * tiles_n, grp = wg / per, in = wg - grp * per;
        const int first = grp * GM, gsz = min(tiles_m - first, GM);
        tn = in / gsz;
        tm = first + (in - tn * gsz);
    }
    const int m0 = tm * BM, n0 = tn * BN;

    const int KT = p.K / BK / p.ksplit;                              // K steps of this workgroup (split-K: its slice)
    const T* A = static_cast<const T*>(p.A) + (size_t)blockIdx.y * KT * BK;
    const T* B = static_cast<const T*>(p.B) + (size_t)blockIdx.y * KT * BK;

    // global -> register staging: thread owns chunk c of rows r0 + RPP i.  Rows beyond M / N are clamped to the last valid row:
    // their products land in accumulators that are never stored
    const int c = tid & 7, r0 = tid >> 3;
    const T* ag[AL];
    const T* bg[BL];
    constexpr int RMAX = AL > BL ? AL : BL;
    int st_off[RMAX];
#pragma unroll
    for (int i = 0; i < RMAX; ++i) st_off[i] = lds_off<T>(r0 + RPP * i, c);
#pragma unroll
    for (int i = 0; i < AL; ++i) ag[i] = A + (size_t)min(m0 + r0 + RPP * i, p.M - 1) * p.lda + c * 8;
#pragma unroll
    for (int i = 0; i < BL; ++i) bg[i] = B + (size_t)min(n0 + r0 + RPP * i, p.N - 1) * p.ldb + c * 8;
    // per-lane fragment read offsets (row & 7 == l15 & 7 because every tile row base is a multiple of 16)
    int a_off[2], b_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        a_off[kk] = lds_off<T>(wm * 16 * MT + l15, g + 4 * kk);
        b_off[kk] = lds_off<T>(wn * 16 * NT + l15, g + 4 * kk);
    }
    constexpr int ROW16 = 16 * BK * (int)sizeof(T);               // LDS bytes of 16 tile rows

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    Chunk<T> ra[AL], rb[BL];
#pragma unroll
    for (int i = 0; i < AL; ++i) ra[i] = cload<T>(ag[i]);
#pragma unroll
    for (int i = 0; i < BL; ++i) rb[i] = cload<T>(bg[i]);
#pragma unroll
    for (int i = 0; i < AL; ++i) lds_st<T>(smem, st_off[i], ra[i]);
#pragma unroll
    for (int i = 0; i < BL; ++i) lds_st<T>(smem + A_BYTES, st_off[i], rb[i]);
    __syncthreads();

    for (int kt = 0; kt < KT; ++kt) {
        const char* As = smem + (kt & 1) * STAGE;
        const char* Bs = As + A_BYTES;
        const bool more = kt + 1 < KT;
        if (more) {
            const int ko = (kt + 1) * BK;
#pragma unroll
            for (int i = 0; i < AL; ++i) ra[i] = cload<T>(ag[i] + ko);
#pragma unroll
            for (int i = 0; i < BL; ++i) rb[i] = cload<T>(bg[i] + ko);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            Chunk<T> fa[MT], fb[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[j] = lds_ld<T>(Bs, b_off[kk] + j * ROW16);
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[i] = lds_ld<T>(As, a_off[kk] + i * ROW16);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mma<T>(fb[j], fa[i], acc[i][j]);       // D[row = n][col = m]
        }
        if (more) {
            char* An = smem + ((kt + 1) & 1) * STAGE;
#pragma unroll
            for (int i = 0; i < AL; ++i) lds_st<T>(An, st_off[i], ra[i]);
#pragma unroll
            for (int i = 0; i < BL; ++i) lds_st<T>(An + A_BYTES, st_off[i], rb[i]);
        }
        __syncthreads();
    }

    if (p.ksplit > 1) {                                              // split-K slice: raw accumulators, the epilogue runs in splitk_epilogue_kernel
        float* out = p.part + (size_t)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m0 + wm * 16 * MT + i * 16 + l15;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = n0 + wn * 16 * NT + j * 16 + g * 4;
                if (n < p.N) *reinterpret_cast<f32x4*>(out + (size_t)m * p.N + n) = acc[i][j];
            }
        }
        return;
    }
    // 256 x 256 bf16 tile: the results leave through LDS.  A lane's accumulators are 4 columns of 16 different rows, so direct
    // stores (and the H loads of the x H epilogue) touch 16 cache lines per instruction with 32 bytes each -- the GELU epilogue
    // (two outputs) cost +78 us and the x H epilogue (one extra input) +55 us on top of a 179 us product.  Staged through the
    // 128 KB the K loop no longer needs (528-byte pitch: conflict-free 8-byte writes per 16-lane group), every global access
    // of the epilogue is a 16-byte-per-lane, row-contiguous one.
    if (sizeof(T) == 2 && (p.N & 7) == 0) {
        constexpr int CP = BN * 2 + 16;                           // LDS bytes per tile row: pitch / 4 = 4 (mod 64) for BN = 64, 128, 256
        constexpr int CPR = BN / 8, NCH = BM * CPR / NTH;         // 16-byte chunks per row / per thread
        static_assert((BM * CPR) % NTH == 0, "tile chunks must divide among the threads");
        const int cols_ok = min(BN, p.N - n0);
        const int rows_ok = min(BM, p.M - m0);
        auto stage_out = [&](void* dst_base, int ld) {           // LDS tile -> global, 16 chunks of 16 bytes per thread
            __syncthreads();
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
                const int cidx = tid + NTH * q, row = cidx / CPR, cc = cidx - row * CPR;
                if (row < rows_ok && cc * 8 < cols_ok)
                    *reinterpret_cast<uint4*>(static_cast<T*>(dst_base) + (size_t)(m0 + row) * ld + n0 + cc * 8) = *reinterpret_cast<const uint4*>(smem + row * CP + cc * 16);
            }
        };
        if constexpr (EPI == EPI_MUL || EPI == EPI_BIAS_RES) {    // the H (or residual) tile comes in the same way, coalesced
            const T* src = static_cast<const T*>(EPI == EPI_MUL ? p.H : p.R);
            const int lds_ = EPI == EPI_MUL ? p.ldh : p.ldr;
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
                const int cidx = tid + NTH * q, row = cidx / CPR, cc = cidx - row * CPR;
                if (row < rows_ok && cc * 8 < cols_ok)
                    *reinterpret_cast<uint4*>(smem + row * CP + cc * 16) = *reinterpret_cast<const uint4*>(src + (size_t)(m0 + row) * lds_ + n0 + cc * 8);
            }
            __syncthreads();
        }
        float4 bias4[NT];
        if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RES) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = n0 + wn * 16 * NT + j * 16 + g * 4;
                bias4[j] = n < p.N ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                char* slot = smem + (wm * 16 * MT + i * 16 + l15) * CP + (wn * 16 * NT + j * 16 + g * 4) * 2;
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RES) { v[0] += bias4[j].x; v[1] += bias4[j].y; v[2] += bias4[j].z; v[3] += bias4[j].w; }
                if constexpr (EPI == EPI_BIAS_RES) {
                    float r[4];
                    load4<T>(reinterpret_cast<const T*>(slot), r);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += r[e];
                }
                if constexpr (EPI == EPI_BIAS_GELU) {
                    f2 y0, d0, y1, d1;
                    gelu_both2((f2){v[0], v[1]}, y0, d0);
                    gelu_both2((f2){v[2], v[3]}, y1, d1);
                    v[0] = y0.x; v[1] = y0.y; v[2] = y1.x; v[3] = y1.y;
                    // GELU' leaves directly (a second staged pass would have to keep or recompute it).  Round 5 measured the alternatives on fc1 of ViT-B/16 at
                    // M = 25216 (this form: 239 us, the plain product 155-158): both outputs through LDS in two half-tile passes with coalesced 16-byte stores,
                    // no extra registers: 246 us; GELU' kept packed for a second staged pass: 303 spilled registers on the 256 x 256 tile, 517 us.  The 85 us
                    // are the erf arithmetic (one v_exp_f32 + one v_rcp_f32 + ~12 packed FMAs per element, 77 M elements) and the second tensor's bytes, not
                    // the shape of the stores; hiding them needs a second accumulator set under the next tile's K loop (not built).
                    if (p.H) {
                        const int m = m0 + wm * 16 * MT + i * 16 + l15;
                        if (m < p.M && n0 + wn * 16 * NT + j * 16 + g * 4 < p.N) {
                            const float d[4] = {d0.x, d0.y, d1.x, d1.y};
                            store4<T>(static_cast<T*>(p.H) + (size_t)m * p.ldh + n0 + wn * 16 * NT + j * 16 + g * 4, d);
                        }
                    }
                }
                if constexpr (EPI == EPI_MUL) {
                    float h[4];
                    load4<T>(reinterpret_cast<const T*>(slot), h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= h[e];
                }
                store4<T>(reinterpret_cast<T*>(slot), v);
            }
        stage_out(p.C, p.ldc);
        return;
    }
    // epilogue: lane holds C[m][n .. n+3], m = m0 + wm*16*MT + i*16 + l15, n = n0 + wn*16*NT + j*16 + g*4
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + wm * 16 * MT + i * 16 + l15;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + wn * 16 * NT + j * 16 + g * 4;
            if (n < p.N) epi_store<T, EPI>(p, acc[i][j], m, n);
        }
    }
}

// second pass of a split-K product: C[m][n .. n+3] = epilogue(sum over slices, in index order)
template <typename T, int EPI>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(GemmParams p) {
    const int n4 = p.N >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)p.M * n4) return;
    const int m = (int)(idx / n4), n = (int)(idx - (long long)m * n4) * 4;
    const float* src = p.part + (size_t)m * p.N + n;
    f32x4 a = *reinterpret_cast<const f32x4*>(src);
    for (int k = 1; k < p.ksplit; ++k) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)k * p.M * p.N);
        a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3];
    }
    epi_store<T, EPI>(p, a, m, n);
}

// The round-1 LDS-DMA experiments (128 x 128 two-stage, 256 x 128 three-stage ring, barrier-staggered ping-pong; none faster than the
// register-staged kernel above: profiles/r01_gemm_ablation.txt) are gone; their successor was gemm5.hip (rounds 2-5; deleted in round 6: gemm8.hip superseded it).

template <typename T, int EPI, int MT, int NT, int WM = 2, int WN = 2> int launch(const GemmParams& p, hipStream_t s) {
    constexpr int BM = 16 * MT * WM, BN = 16 * NT * WN;
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    size_t smem = 2 * (size_t)(BM + BN) * BK * sizeof(T);
    if (sizeof(T) == 2 && smem < (size_t)BM * (BN * 2 + 16)) smem = (size_t)BM * (BN * 2 + 16);      // LDS-staged epilogue
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<T, EPI, MT, NT, WM, WN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_done = true;
    }
    hipLaunchKernelGGL((gemm_nt_kernel<T, EPI, MT, NT, WM, WN>), dim3(tiles, p.ksplit), dim3(64 * WM * WN), smem, s, p);
    CLHIP_LAUNCH_CHECK();
    if (p.ksplit > 1) {
        const long long n = (long long)p.M * (p.N >> 2);
        hipLaunchKernelGGL((splitk_epilogue_kernel<T, EPI>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p);
        CLHIP_LAUNCH_CHECK();
    }
    return CLHIP_OK;
}

// fp32 scratch of the split-K products: one buffer per stream that ever needed one (two streams never share a buffer), grown on demand outside stream
// capture; nullptr = not available right now (the caller runs the unsplit kernel)
float* splitk_scratch(hipStream_t s, size_t bytes) {
    struct Slot { hipStream_t s; int dev; float* p; size_t n; };
    static Slot slots[16];
    static int nslots = 0;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    Slot* sl = nullptr;
    for (int i = 0; i < nslots; ++i) if (slots[i].s == s && slots[i].dev == dev) sl = &slots[i];
    bytes += 4096;                                            // (the first 4 KB stay unused: round 5's fused variant kept its tile tickets there)
    if (sl != nullptr && sl->n >= bytes) return sl->p;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
    if (sl == nullptr) { if (nslots == 16) return nullptr; sl = &slots[nslots++]; *sl = Slot{s, dev, nullptr, 0}; }
    // an outgrown buffer is NOT freed: a captured graph of this stream may hold its address (tens of MB at most; sizes are the ViT's few GEMM shapes)
    sl->p = nullptr; sl->n = 0;
    const size_t want = bytes + bytes / 4;
    if (hipMalloc(reinterpret_cast<void**>(&sl->p), want) != hipSuccess) { (void)hipGetLastError(); sl->p = nullptr; return nullptr; }
    if (hipMemset(sl->p, 0, 4096) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(sl->p); sl->p = nullptr; return nullptr; }
    sl->n = want;
    return sl->p;
}

// block tile: 64 x 64 when 128-wide tiles cannot fill half of the 512 resident workgroups (2 per CU); otherwise 128 columns and
// the row count (160 / 128) with the better work / (rounds x slots) quantisation.  fp32 (parity mode): 128 x 128.
// CLHIP_GEMM_MT forces 2 (64 x 64), 4, 5 or 8 (256 x 256, 8 waves).
int pick_tile(int M, int N) {
    static int forced = -1;
    if (forced < 0) { const char* e = clhip_cfg("GEMM_MT"); forced = e ? atoi(e) : 0; }
    if (forced == 2 || forced == 4 || forced == 5 || forced == 8) return forced;
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    if (t128 < 256) return 2;
    {   // 256 x 256 tiles (one 8-wave workgroup per CU) when they still fill the chip for >= 3.4 rounds at >= 85 % quantisation
        const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256);
        const long rounds = (t256 + 255) / 256;
        if (N % 256 == 0 && t256 >= 768 && (double)t256 / (double)(rounds * 256) >= 0.85) return 8;
    }
    int best = 4;
    double best_eff = -1.0;
    for (int mt : {5, 4}) {
        const long tiles = (long)((M + 32 * mt - 1) / (32 * mt)) * ((N + 127) / 128);
        const long rounds = (tiles + 511) / 512;
        const double eff = (double)M * ((N + 127) / 128) / ((double)rounds * 512 * 32 * mt);
        if (eff > best_eff + 1e-9) { best_eff = eff; best = mt; }
    }
    return best;
}

template <typename T, int EPI> int launch_mt(const GemmParams& p, hipStream_t s, bool allow_split = true) {
    if constexpr (sizeof(T) == 4) {
        return launch<T, EPI, 4, 4>(p, s);
    } else {
        // Narrow outputs with a long K (the ViT's N = 768 GEMMs, 36 % of an InfLoRA step): 256 x 256 tiles are 30 % faster per
        // tile (891 vs 647 TFLOP/s at K = 3072) but 297 of them on 256 CUs take two rounds.  Run exactly one round of 256 x 256
        // tiles on the leading rows and hand the remaining rows to the small-tile kernels (a second, short launch).
        static const bool no_split = clhip_cfg("GEMM_NO_SPLIT") != nullptr || clhip_cfg("GEMM_MT") != nullptr;
        // Few output tiles, long K (L2P at batch 16: [3552 x 3072] . [768 x 3072]^T is 168 tiles of 128 x 128 -- the 64 x 64 tiles that filled the chip instead ran
        // at 422 TFLOP/s against the vendor's 693): 128 x 128 tiles over 2-4 K slices, then one reduce + epilogue pass (profiles/r04_gemm_vs_blas.txt).
        // Measured (r04_gemm_vs_blas.txt, M = 3552, N = 768): K = 3072 39.7 -> 35.4 us (vendor 24.9), K = 2304 25.4 -> 29.4 us -- the slices' 128 x 128 tiles run at the same
        // ~600 TFLOP/s as the rest of this kernel and the second pass costs ~7 us, so only the longest K gains; CLHIP_GEMM_SPLITK = 0 disables, any other value = the minimum K.
        static const bool no_splitk = clhip_cfg("GEMM_SPLITK") != nullptr && atoi(clhip_cfg("GEMM_SPLITK")) == 0;
        static const int splitk_min_k = (clhip_cfg("GEMM_SPLITK") != nullptr && atoi(clhip_cfg("GEMM_SPLITK")) > 1) ? atoi(clhip_cfg("GEMM_SPLITK")) : 3072;
        if (allow_split && !no_split && !no_splitk && p.ksplit == 1 && p.K >= splitk_min_k && (p.N & 3) == 0) {
            const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
            if (t128 < 256) {
                const int ksteps = p.K / BK;
                int ks = (int)(512 / t128);
                if (ks > 4) ks = 4;
                while (ks > 1 && (ksteps % ks != 0 || ksteps / ks < 8)) --ks;
                if (ks > 1) {
                    float* ws = splitk_scratch(s, (size_t)ks * p.M * p.N * sizeof(float));
                    if (ws != nullptr) {
                        GemmParams q = p;
                        q.ksplit = ks; q.part = ws + 1024; q.group_m = 1;
                        // (round 5 also had the LAST slice of a tile reduce and store it -- one launch instead of two; measured slower, [3552 x 768 x 3072] 34.7 -> 54.7 us:
                        //  three 64-KB slices per tile read behind agent-scope fences cost the reducer more than the second launch does; removed in round 6)
                        return launch<T, EPI, 4, 4>(q, s);
                    }
                }
            }
        }
        if (allow_split && !no_split && p.N % 256 == 0 && p.K >= 2304) {
            const int tn = p.N / 256;
            const long t256 = (long)((p.M + 255) / 256) * tn;
            const int head_rows = (256 / tn) * 256;
            if (tn <= 4 && t256 > 256 && t256 < 448 && head_rows < p.M) {
                GemmParams h = p, t = p;
                h.M = head_rows;
                if (int rc = launch<T, EPI, 8, 4, 2, 4>(h, s)) return rc;
                const size_t es = sizeof(T);
                t.M = p.M - head_rows;
                t.A = static_cast<const char*>(p.A) + (size_t)head_rows * p.lda * es;
                t.C = static_cast<char*>(p.C) + (size_t)head_rows * p.ldc * es;
                if (p.R) t.R = static_cast<const char*>(p.R) + (size_t)head_rows * p.ldr * es;
                if (p.H) t.H = static_cast<char*>(p.H) + (size_t)head_rows * p.ldh * es;
                return launch_mt<T, EPI>(t, s, false);
            }
        }
        // short K, wide N (qkv / fc1 / the fc2 activation gradient, K = 768): the last of the 3.5 - 4.6 rounds of 256 x 256 tiles is
        // poorly filled.  Whole rounds of big tiles run on the leading rows, the remaining rows on the small tiles (qkv forward
        // 151.7 -> 139.0 us; CLHIP_GEMM_TAIL=0 disables).  Only when the last round is < 160 of 256 tiles: a fuller one costs more as small tiles.
        static const bool tail_split = !(clhip_cfg("GEMM_TAIL") != nullptr && atoi(clhip_cfg("GEMM_TAIL")) == 0);
        if (allow_split && tail_split && !no_split && p.N % 256 == 0 && pick_tile(p.M, p.N) == 8) {
            const int tn = p.N / 256;
            const long t256 = (long)((p.M + 255) / 256) * tn;
            const long full = t256 / 256;
            const long rest = t256 - full * 256;
            const int head_rows = (int)(full * 256 / tn) * 256;
            if (full >= 1 && rest > 0 && rest < 160 && head_rows < p.M) {
                GemmParams h = p, t = p;
                h.M = head_rows;
                if (int rc = launch<T, EPI, 8, 4, 2, 4>(h, s)) return rc;
                const size_t es = sizeof(T);
                t.M = p.M - head_rows;
                t.A = static_cast<const char*>(p.A) + (size_t)head_rows * p.lda * es;
                t.C = static_cast<char*>(p.C) + (size_t)head_rows * p.ldc * es;
                if (p.R) t.R = static_cast<const char*>(p.R) + (size_t)head_rows * p.ldr * es;
                if (p.H) t.H = static_cast<char*>(p.H) + (size_t)head_rows * p.ldh * es;
                return launch_mt<T, EPI>(t, s, false);
            }
        }
        switch (pick_tile(p.M, p.N)) {
            case 8: return launch<T, EPI, 8, 4, 2, 4>(p, s);
            case 5: return launch<T, EPI, 5, 4>(p, s);
            case 2: return launch<T, EPI, 2, 2>(p, s);
            default: return launch<T, EPI, 4, 4>(p, s);
        }
    }
}

template <typename T> int dispatch(int epi, const GemmParams& p, hipStream_t s) {
    switch (epi) {
        case EPI_NONE: return launch_mt<T, EPI_NONE>(p, s);
        case EPI_BIAS: return launch_mt<T, EPI_BIAS>(p, s);
        case EPI_BIAS_RES: return launch_mt<T, EPI_BIAS_RES>(p, s);
        case EPI_BIAS_GELU: return launch_mt<T, EPI_BIAS_GELU>(p, s);
        case EPI_MUL: return launch_mt<T, EPI_MUL>(p, s);
    }
    clhip_set_error("clhip_gemm_nt: unknown epilogue %d", epi);
    return CLHIP_EINVAL;
}

}  // namespace


int clhip_gemm8_rows(int M, int N, int K, int lda, int ldb, int ldc, int ldr, int ldh, int dtype);
int clhip_gemm8_launch(const void* A, const void* B, void* C, const float* bias, const void* R, void* H, int M, int N, int K,
                       int lda, int ldb, int ldc, int ldr, int ldh, int epilogue, hipStream_t st);

extern "C" int clhip_gemm_nt(const void* A, const void* B, void* C, const float* bias, const void* R, void* H, int M, int N, int K,
                             int lda, int ldb, int ldc, int ldr, int ldh, int epilogue, int dtype, void* stream) {
    CLHIP_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0);
    CLHIP_CHECK_ARG(K % 64 == 0 && N % 4 == 0);
    CLHIP_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0);
    CLHIP_CHECK_ARG(dtype == CLHIP_BF16 || dtype == CLHIP_F32);
    if (epilogue == EPI_BIAS || epilogue == EPI_BIAS_RES || epilogue == EPI_BIAS_GELU) CLHIP_CHECK_ARG(bias != nullptr);
    if (epilogue == EPI_BIAS_RES) CLHIP_CHECK_ARG(R != nullptr && ldr % 4 == 0);
    if (epilogue == EPI_MUL) CLHIP_CHECK_ARG(H != nullptr);
    if (H) CLHIP_CHECK_ARG(ldh % 4 == 0);
    static const int gm_env = clhip_cfg("GEMM_GROUP_M") ? atoi(clhip_cfg("GEMM_GROUP_M")) : 0;
    GemmParams p{A, B, C, bias, R, H, M, N, K, lda, ldb, ldc, ldr, ldh, gm_env > 0 ? gm_env : (N >= 4096 ? 4 : 1)};     // wide outputs: 8192^3 991 -> 1046 TFLOP/s; the ViT shapes (N <= 3072) are indifferent
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (const int m8 = clhip_gemm8_rows(M, N, K, lda, ldb, ldc, ldr, ldh, dtype)) {
        // gemm8.hip takes the leading m8 rows (whole rounds of its 256 persistent workgroups), the register-staged kernel the rest
        if (int rc = clhip_gemm8_launch(A, B, C, bias, R, H, m8, N, K, lda, ldb, ldc, ldr, ldh, epilogue, s)) return rc;
        if (m8 >= M) return CLHIP_OK;
        const size_t e = 2;                                    // bf16
        A = static_cast<const char*>(A) + (size_t)m8 * lda * e;
        C = static_cast<char*>(C) + (size_t)m8 * ldc * e;
        if (R) R = static_cast<const char*>(R) + (size_t)m8 * ldr * e;
        if (H) H = static_cast<char*>(H) + (size_t)m8 * ldh * e;
        M -= m8;
        p.A = A; p.C = C; p.R = R; p.H = H; p.M = M;
        return dispatch<bf16_t>(epilogue, p, s);
    }
    return dtype == CLHIP_BF16 ? dispatch<bf16_t>(epilogue, p, s) : dispatch<float>(epilogue, p, s);
}
