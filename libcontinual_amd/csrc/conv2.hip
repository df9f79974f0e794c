// conv2.hip -- second-generation implicit-GEMM convolution (forward + dgrad) for gfx950.
//
// Same math and orientation as conv.hip (D[row = out channel][col = pixel], 16x16x32 bf16 MFMA, fp32
// accumulate, BN statistics in the epilogue) but restructured around what the rocprofv3 trace of the
// first version showed (profiles/r01_*): 8 MFMAs per wave between two barriers left the matrix pipe idle
// ~85 % of the time.  Changes:
//   * K-step 64 (one 3x3 tap of a 64-channel layer per step), 2 MFMA sub-steps per step;
//   * per-wave tile up to 64 pixels x 64 channels (16 accumulator tiles, 8 ds_read_b128 per 16 MFMAs);
//   * LDS double buffering with register prefetch: global loads for step k+1 are issued before the MFMAs
//     of step k and written to the other LDS buffer afterwards -> ONE barrier per K-step;
//   * 128-byte LDS rows with the XOR swizzle chunk ^= (row & 7): every 16-lane service group of
//     ds_read_b128 hits 16 distinct 16-byte slots of the 256-byte bank row (conflict-free), and the
//     staging writes of 8 consecutive lanes fill one whole row.
//   * workgroup shapes chosen per layer so that >= 2 workgroups per CU exist even for the 4x4 / 8x8 layers.
#include <stdlib.h>

#include "common.h"

int clhip_wgrad_reduce_launch(const float* slab, float* dw, int64_t n4, int splits, hipStream_t st);      // conv3.hip

namespace {

constexpr int BK2 = 64;

struct ConvParams2 {
    const void* src;
    const void* wt;
    void* dst;
    float* stats;
    double* stat_acc;   // alternative to `stats`: per-channel [stat_rep][2][Cd] fp64 sums, accumulated with atomics
    int stat_rep;       // number of accumulator replicas (power of two); a workgroup adds into replica blockIdx.x & (stat_rep - 1)
    int N, Hs, Ws, Cs, log2Cs, Hd, Wd, Cd, ksize, stride, pad, accumulate;
    int M, K;
    int cls_tiles;      // > 0: stride-2 dgrad parity decomposition, tiles per (h&1, w&1) class
};

template <typename T> struct Chunk2;
template <> struct Chunk2<bf16_t> { uint4 a; };
template <> struct Chunk2<float> { uint4 a, b; };

template <typename T> __device__ __forceinline__ Chunk2<T> zero2() {
    Chunk2<T> c;
    if constexpr (sizeof(T) == 2) { c.a = make_uint4(0, 0, 0, 0); }
    else { c.a = make_uint4(0, 0, 0, 0); c.b = make_uint4(0, 0, 0, 0); }
    return c;
}
template <typename T> __device__ __forceinline__ Chunk2<T> gload2(const T* p) {
    Chunk2<T> c;
    if constexpr (sizeof(T) == 2) { c.a = *reinterpret_cast<const uint4*>(p); }
    else { c.a = *reinterpret_cast<const uint4*>(p); c.b = *reinterpret_cast<const uint4*>(p + 4); }
    return c;
}
// row pitch: 8 chunks.  bf16: 128 B rows, swizzled.  fp32 (parity mode): 256 B rows, linear.
template <typename T> __device__ __forceinline__ int lds_off2(int row, int chunk) {
    if constexpr (sizeof(T) == 2) return row * 128 + ((chunk ^ (row & 7)) << 4);
    else return row * 256 + (chunk << 5);
}
template <typename T> __device__ __forceinline__ void lds_st2(char* base, int row, int chunk, const Chunk2<T>& c) {
    char* p = base + lds_off2<T>(row, chunk);
    if constexpr (sizeof(T) == 2) { *reinterpret_cast<uint4*>(p) = c.a; }
    else { *reinterpret_cast<uint4*>(p) = c.a; *reinterpret_cast<uint4*>(p + 16) = c.b; }
}
template <typename T> __device__ __forceinline__ Chunk2<T> lds_ld2(const char* base, int row, int chunk) {
    const char* p = base + lds_off2<T>(row, chunk);
    Chunk2<T> c;
    if constexpr (sizeof(T) == 2) { c.a = *reinterpret_cast<const uint4*>(p); }
    else { c.a = *reinterpret_cast<const uint4*>(p); c.b = *reinterpret_cast<const uint4*>(p + 16); }
    return c;
}
template <typename T> __device__ __forceinline__ f32x4 mma2(const Chunk2<T>& wf, const Chunk2<T>& xf, f32x4 acc) {
    if constexpr (sizeof(T) == 2) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf.a), __builtin_bit_cast(bf16x8_t, xf.a), acc, 0, 0, 0);
    } else {
        const float* a = reinterpret_cast<const float*>(&wf);
        const float* b = reinterpret_cast<const float*>(&xf);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
        return acc;
    }
}

// Stride-2 dgrad: dx(h,w) only receives the taps with (h+pad-r) and (w+pad-s) even, i.e. 1, 2, 2 or 4 of the 9
// taps depending on the parity class (h&1, w&1).  Tiles are therefore formed from pixels of ONE class
// (blockIdx.x = class * cls_tiles + tile) and the K loop visits only that class's taps -- 2.25 taps per pixel
// on average instead of 9 with 75 % zero operands (the first version ran these three layers at 90 TFLOP/s).
__device__ __forceinline__ int map_pixel(const ConvParams2& p, int BM, int row) {
    if (p.cls_tiles == 0) { int pix = blockIdx.x * BM + row; return pix < p.M ? pix : -1; }
    const int cb = blockIdx.x / p.cls_tiles, cls = 3 - cb, tile = blockIdx.x - cb * p.cls_tiles;      // heaviest class first, see below
    const int h2 = p.Hd >> 1, w2 = p.Wd >> 1;
    const int q = tile * BM + row;
    if (q >= p.N * h2 * w2) return -1;
    const int wq = q % w2, t = q / w2, hq = t % h2, n = t / h2;
    return (n * p.Hd + 2 * hq + (cls >> 1)) * p.Wd + 2 * wq + (cls & 1);
}

// WM x WN waves (WM*WN == 4); each wave owns MT 16-pixel tiles x NT 16-channel tiles.
template <typename T, int WM, int WN, int MT, int NT, int MODE>
__global__ __launch_bounds__(256) void conv_igemm2_kernel(ConvParams2 p) {
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16;
    constexpr int ROWB = BK2 * sizeof(T);
    constexpr int AROWS = (BM + 31) / 32, BROWS = (BN + 31) / 32;
    constexpr int STAGE = (BM + BN) * ROWB;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 2 stages; the stats scratch aliases stage 0
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const T* __restrict__ src = static_cast<const T*>(p.src);
    const T* __restrict__ wt = static_cast<const T*>(p.wt);

    const int ca = tid & 7, rbase = tid >> 3;
    // Per staged pixel row: element offset of its (tap 0, channel 0) source element and a 9-bit mask of the taps
    // whose source pixel exists.  For the forward gather (any stride) and the stride-1 dgrad gather the source
    // offset is row_const + tap_const, so a K-step costs one bit test and one add per 16-byte load instead of
    // re-deriving coordinates (the first version spent ~12 VALU instructions per MFMA on that).
    constexpr bool LINEAR = true;
    // parity-class tiles of the stride-2 dgrad are linear too: with h + pad = 2a + e (e fixed by the class) the contributing taps are
    // r = e + 2m and read source row a - m, i.e. "origin a, tap index m" with the same formula as stride 1
    const bool cls_mode = (MODE == 1) && p.cls_tiles > 0;
    const int cls_id = cls_mode ? 3 - (int)(blockIdx.x / p.cls_tiles) : 0;
    const int e_h = cls_mode ? (((cls_id >> 1) + p.pad) & 1) : 0, e_w = cls_mode ? (((cls_id & 1) + p.pad) & 1) : 0;
    const bool lin = (MODE == 0) || (p.stride == 1) || cls_mode;
    long long a_off[AROWS];
    unsigned a_mask[AROWS];
    int a_h0[AROWS], a_w0[AROWS], a_base[AROWS];
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
        int row = rbase + 32 * i;
        int pix = row < BM ? map_pixel(p, BM, row) : -1;
        a_off[i] = 0; a_mask[i] = 0; a_h0[i] = 0; a_w0[i] = 0; a_base[i] = -1;
        if (pix >= 0) {
            int wd = pix % p.Wd;
            int t = pix / p.Wd;
            int hd = t % p.Hd;
            int n = t / p.Hd;
            a_base[i] = n * p.Hs * p.Ws;
            int h0, w0, dir;
            if (MODE == 0) { h0 = hd * p.stride - p.pad; w0 = wd * p.stride - p.pad; dir = 1; }
            else { h0 = hd + p.pad; w0 = wd + p.pad; dir = -1; }
            if (cls_mode) { h0 >>= 1; w0 >>= 1; }
            a_h0[i] = h0; a_w0[i] = w0;
            if (lin) {
                a_off[i] = ((long long)a_base[i] + (long long)h0 * p.Ws + w0) * p.Cs;
                unsigned m = 0;
                for (int r = 0; r < p.ksize; ++r)
                    for (int s2 = 0; s2 < p.ksize; ++s2) {
                        int hs = h0 + dir * r, ws = w0 + dir * s2;
                        if (cls_mode) {
                            if (((r - e_h) & 1) || ((s2 - e_w) & 1)) continue;        // this tap never reaches the class
                            hs = h0 - ((r - e_h) >> 1); ws = w0 - ((s2 - e_w) >> 1);
                        }
                        if ((unsigned)hs < (unsigned)p.Hs && (unsigned)ws < (unsigned)p.Ws) m |= 1u << (r * p.ksize + s2);
                    }
                a_mask[i] = m;
            }
        }
    }
    (void)LINEAR;
    long long b_off[BROWS];
    bool b_ok[BROWS];
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
        int row = rbase + 32 * i;
        int o = n0 + row;
        b_ok[i] = row < BN && o < p.Cd;
        b_off[i] = (long long)o * p.K;
    }
    Chunk2<T> ra[AROWS], rb[BROWS];
    auto gload = [&](int kstep) {
        const int kk = kstep * BK2 + ca * 8;
        const bool kvalid = kk < p.K;
        const int tap = kk >> p.log2Cs;
        const int cs = kk & (p.Cs - 1);
        const int r = (p.ksize == 3) ? tap / 3 : 0;
        const int s = (p.ksize == 3) ? tap - 3 * r : 0;
        if (lin) {
            const int dir = (MODE == 0) ? 1 : -1;
            const int rr = cls_mode ? ((r - e_h) >> 1) : r, ss = cls_mode ? ((s - e_w) >> 1) : s;
            const long long toff = (long long)dir * (rr * p.Ws + ss) * p.Cs + cs;   // thread-uniform for this K-step
            const unsigned bit = kvalid ? (1u << tap) : 0u;
#pragma unroll
            for (int i = 0; i < AROWS; ++i) ra[i] = (a_mask[i] & bit) ? gload2<T>(src + (a_off[i] + toff)) : zero2<T>();
        } else {
#pragma unroll
            for (int i = 0; i < AROWS; ++i) {
                bool ok = kvalid && a_base[i] >= 0;
                int th = a_h0[i] - r, tw = a_w0[i] - s;
                ok = ok && th >= 0 && tw >= 0 && (th % p.stride == 0) && (tw % p.stride == 0);
                int hs = th / p.stride, ws = tw / p.stride;
                ok = ok && (unsigned)hs < (unsigned)p.Hs && (unsigned)ws < (unsigned)p.Ws;
                ra[i] = ok ? gload2<T>(src + ((size_t)(a_base[i] + hs * p.Ws + ws) * p.Cs + cs)) : zero2<T>();
            }
        }
#pragma unroll
        for (int i = 0; i < BROWS; ++i) rb[i] = (b_ok[i] && kvalid) ? gload2<T>(wt + (b_off[i] + kk)) : zero2<T>();
    };
    auto sstore = [&](int stage) {
        char* As = smem + stage * STAGE;
        char* Bs = As + BM * ROWB;
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
            int row = rbase + 32 * i;
            if (row < BM) lds_st2<T>(As, row, ca, ra[i]);
        }
#pragma unroll
        for (int i = 0; i < BROWS; ++i) {
            int row = rbase + 32 * i;
            if (row < BN) lds_st2<T>(Bs, row, ca, rb[i]);
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK2 - 1) / BK2;
    const int fr = lane & 15, fg = lane >> 4;
    // K-steps to visit: all of them, or (parity-class mode; Cs >= 64 so a step lies inside one tap) those whose tap
    // can contribute to this class
    unsigned tapmask = 0x1ff;
    if (p.cls_tiles > 0) {
        // (the class with four taps, (1, 1), gets the lowest block indices: launched first, its tiles -- four times the work of the
        //  one-tap class's -- do not form the tail of the launch)
        const int cls = 3 - (int)(blockIdx.x / p.cls_tiles), ph = cls >> 1, pw = cls & 1;
        tapmask = 0;
        for (int t = 0; t < 9; ++t) {
            const int r = t / 3, s2 = t - 3 * r;
            if (((ph + p.pad - r) & 1) == 0 && ((pw + p.pad - s2) & 1) == 0) tapmask |= 1u << t;
        }
    }
    // a 64-element K-step lies inside one tap (Cs >= 64) or spans two (Cs == 32): visited when either can contribute -- the
    // per-element parity test of the gather zeroes the other one
    auto step_ok = [&](int k) {
        if (p.cls_tiles == 0) return true;
        const int t0 = (k * BK2) >> p.log2Cs, t1 = ((k + 1) * BK2 - 1) >> p.log2Cs;
        return (((tapmask >> t0) | (tapmask >> (t1 < 9 ? t1 : t0))) & 1u) != 0;
    };
    auto next_step = [&](int k) { ++k; while (k < nk && !step_ok(k)) ++k; return k; };
    int k = next_step(-1);
    if (k < nk) gload(k);
    sstore(0);
    __syncthreads();
    for (int it = 0; k < nk; ++it) {
        const int kn = next_step(k);
        const bool more = kn < nk;
        if (more) gload(kn);
        const char* As = smem + (it & 1) * STAGE;
        const char* Bs = As + BM * ROWB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            Chunk2<T> xf[MT], wf[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) xf[i] = lds_ld2<T>(As, (wm * MT + i) * 16 + fr, ks * 4 + fg);
#pragma unroll
            for (int j = 0; j < NT; ++j) wf[j] = lds_ld2<T>(Bs, (wn * NT + j) * 16 + fr, ks * 4 + fg);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mma2<T>(wf[j], xf[i], acc[i][j]);
        }
        if (more) sstore((it + 1) & 1);
        __syncthreads();
        k = kn;
    }

    T* __restrict__ dst = static_cast<T*>(p.dst);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int pix = map_pixel(p, BM, (wm * MT + i) * 16 + fr);
        if (pix >= 0) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int o = n0 + (wn * NT + j) * 16 + fg * 4;
                if (o < p.Cd) {
                    T* q = dst + (size_t)pix * p.Cd + o;
                    float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    if (p.accumulate) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += Elem<T>::ld(q + e);
                    }
                    if constexpr (sizeof(T) == 2) {
                        uint2 u;
                        u.x = pack_bf16x2(v[0], v[1]);
                        u.y = pack_bf16x2(v[2], v[3]);
                        *reinterpret_cast<uint2*>(q) = u;
                    } else {
                        *reinterpret_cast<float4*>(q) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                }
            }
        }
    }

    if (p.stats != nullptr || p.stat_acc != nullptr) {
        float* red = reinterpret_cast<float*>(smem);      // [WM][2][BN]; all LDS reads of the K loop are behind the last barrier
        float sv[8 * NT];                                 // [j][e] sums, then [j][e] sums of squares
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int i = 0; i < MT; ++i) { const float v = acc[i][j][e]; s1 += v; s2 = fmaf(v, v, s2); }
                sv[j * 4 + e] = s1;
                sv[4 * NT + j * 4 + e] = s2;
            }
        row16_sum_n(sv);
        if (fr == 0) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {                  // channels fg*4 .. fg*4+3 of tile j are contiguous: 16-byte stores
                const int c = (wn * NT + j) * 16 + fg * 4;
                *reinterpret_cast<float4*>(red + (wm * 2 + 0) * BN + c) = make_float4(sv[j * 4], sv[j * 4 + 1], sv[j * 4 + 2], sv[j * 4 + 3]);
                *reinterpret_cast<float4*>(red + (wm * 2 + 1) * BN + c) = make_float4(sv[4 * NT + j * 4], sv[4 * NT + j * 4 + 1], sv[4 * NT + j * 4 + 2], sv[4 * NT + j * 4 + 3]);
            }
        }
        __syncthreads();
        for (int idx = tid; idx < 2 * BN; idx += 256) {
            int which = idx / BN, c = idx - which * BN;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) t += red[(w * 2 + which) * BN + c];
            if (n0 + c < p.Cd) {
                if (p.stat_acc != nullptr) atomicAdd(p.stat_acc + ((size_t)(blockIdx.x & (p.stat_rep - 1)) * 2 + which) * p.Cd + n0 + c, (double)t);
                else p.stats[((size_t)blockIdx.x * 2 + which) * p.Cd + n0 + c] = t;
            }
        }
    }
}

struct TileCfg { int bm, bn; };

// candidate shapes, largest pixel tile first for each channel width
TileCfg pick_tile(int M, int Cd) {
    static const char* force = clhip_cfg("IGEMM_TILE");      // "bm,bn" (ablation runs)
    if (force) { int bm = 0, bn = 0; if (sscanf(force, "%d,%d", &bm, &bn) == 2 && bm > 0 && bn > 0 && bn <= Cd) return TileCfg{bm, bn}; }
    int bn = Cd >= 128 ? 128 : (Cd >= 64 ? 64 : (Cd >= 32 ? 32 : 16));
    int gy = (Cd + bn - 1) / bn;
    int cands128[2] = {128, 64};
    int cands[3] = {256, 128, 64};
    const int* c = bn == 128 ? cands128 : cands;
    int n = bn == 128 ? 2 : 3;
    for (int i = 0; i < n; ++i) {
        int64_t wgs = (int64_t)((M + c[i] - 1) / c[i]) * gy;
        if (wgs >= 512 || i == n - 1) return TileCfg{c[i], bn};
    }
    return TileCfg{64, bn};
}

template <typename T, int WM, int WN, int MT, int NT, int MODE>
int launch_cfg(const ConvParams2& p, hipStream_t st) {
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16;
    constexpr size_t lds = 2 * (size_t)(BM + BN) * BK2 * sizeof(T);
    static bool attr_set = false;
    auto kern = conv_igemm2_kernel<T, WM, WN, MT, NT, MODE>;
    if (!attr_set && lds > 64 * 1024) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    dim3 grid((p.M + BM - 1) / BM, (p.Cd + BN - 1) / BN);
    ConvParams2 q = p;
    if (q.cls_tiles > 0) {                           // parity-class tiling: 4 classes of M/4 pixels
        q.cls_tiles = (q.M / 4 + BM - 1) / BM;
        grid.x = 4 * q.cls_tiles;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, q);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

template <typename T, int MODE>
int launch2(const ConvParams2& p, hipStream_t st) {
    TileCfg t = pick_tile(p.cls_tiles > 0 ? p.M / 4 : p.M, p.Cd);
    // <WM, WN, MT, NT>
    if (t.bn == 128) {
        if (t.bm == 128) return launch_cfg<T, 2, 2, 4, 4, MODE>(p, st);
        return launch_cfg<T, 1, 4, 4, 2, MODE>(p, st);                   // 64 x 128
    }
    if (t.bn == 64) {
        if (t.bm == 256) return launch_cfg<T, 4, 1, 4, 4, MODE>(p, st);
        if (t.bm == 128) return launch_cfg<T, 4, 1, 2, 4, MODE>(p, st);
        return launch_cfg<T, 4, 1, 1, 4, MODE>(p, st);
    }
    if (t.bn == 32) {
        if (t.bm == 256) return launch_cfg<T, 4, 1, 4, 2, MODE>(p, st);
        if (t.bm == 128) return launch_cfg<T, 4, 1, 2, 2, MODE>(p, st);
        return launch_cfg<T, 4, 1, 1, 2, MODE>(p, st);
    }
    if (t.bm == 256) return launch_cfg<T, 4, 1, 4, 1, MODE>(p, st);
    if (t.bm == 128) return launch_cfg<T, 4, 1, 2, 1, MODE>(p, st);
    return launch_cfg<T, 4, 1, 1, 1, MODE>(p, st);
}

}  // namespace

// entry points used by conv.hip's C ABI functions
int clhip_conv2_tiles_m(int M, int Cd) { return (M + pick_tile(M, Cd).bm - 1) / pick_tile(M, Cd).bm; }

int clhip_conv2_launch(const void* src, const void* wt, void* dst, float* stats, double* stat_acc, int stat_rep, int N, int Hs, int Ws, int Cs, int Hd, int Wd,
                       int Cd, int ksize, int stride, int pad, int accumulate, int mode, int dtype, hipStream_t st) {
    ConvParams2 p;
    p.src = src; p.wt = wt; p.dst = dst; p.stats = stats; p.stat_acc = stat_acc; p.stat_rep = stat_rep > 0 ? stat_rep : 1;
    p.N = N; p.Hs = Hs; p.Ws = Ws; p.Cs = Cs; p.log2Cs = ilog2_exact(Cs); p.Hd = Hd; p.Wd = Wd; p.Cd = Cd;
    p.ksize = ksize; p.stride = stride; p.pad = pad; p.accumulate = accumulate;
    p.M = N * Hd * Wd; p.K = ksize * ksize * Cs;
    static const bool nocls = clhip_cfg("NO_PARITY_DGRAD") != nullptr;
    p.cls_tiles = (!nocls && mode == 1 && stride == 2 && ksize == 3 && Cs >= 32 && p.log2Cs >= 0 && (Hd % 2 == 0) && (Wd % 2 == 0)) ? 1 : 0;
    if (dtype == CLHIP_BF16) return mode == 0 ? launch2<bf16_t, 0>(p, st) : launch2<bf16_t, 1>(p, st);
    return mode == 0 ? launch2<float, 0>(p, st) : launch2<float, 1>(p, st);
}

// =============================================================================================== wgrad v2
// dw[o][tap][c] += sum_p dz[p][o] * x[p @ tap][c].   GEMM with the PIXEL axis as reduction: both operands
// are stored pixel-major ([pixel][channel], as they sit in HBM), and the MFMA wants 8 consecutive
// reduction elements per lane -> fragments are read with the gfx950 LDS transpose read
// (ds_read_b64_tr_b16).  Workgroup tile BO out-channels x 128 (tap,c) columns, 64 pixels per K-step,
// double-buffered LDS, split over the pixel axis (grid.z) with fp32 atomics into the gradient buffer.
namespace {

// ablation switches exist in ABL=1 builds only (a run-time test around every MFMA splits the K loop into basic blocks: conv6.hip's lesson)
#ifdef CLHIP_ABLATION
#define DBGW2(p) ((p).debug)
#else
#define DBGW2(p) 0
#endif

struct WgradParams2 {
    const void* x; const void* dz; float* dw;
    int N, H, W, C, log2C, Creal, Ho, Wo, lgHo, lgWo, K, ksize, stride, pad;
    int M, J, pix_per_split, debug;
    float* slab;              // deterministic form: [split][K][taps][Creal] fp32 partial blocks (plain stores, summed in a fixed order afterwards); nullptr: atomics
    int64_t slab_stride;
};

template <typename T>
__device__ __forceinline__ Chunk2<T> lds_ld_tr(const char* tile, int pitch, int pix0, int col0, int lane) {
    const int fr = lane & 15, fg = lane >> 4;
    Chunk2<T> c;
    if constexpr (sizeof(T) == 2) {
        const char* a0 = tile + (pix0 + fg * 8 + (fr >> 2)) * pitch + (col0 + (fr & 3) * 4) * 2;
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * pitch));
        uint2 l = __builtin_bit_cast(uint2, lo), h = __builtin_bit_cast(uint2, hi);
        c.a = make_uint4(l.x, l.y, h.x, h.y);
    } else {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float*>(tile + (pix0 + fg * 8 + j) * pitch + (col0 + fr) * 4);
        c.a = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
        c.b = make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7]));
    }
    return c;
}

// WO x WJ waves; each wave MT out-channel tiles x NT column tiles.  BJ = WJ*NT*16 must be 128.
template <typename T, int WO, int WJ, int MT, int NT>
__global__ __launch_bounds__(256) void conv_wgrad2_kernel(WgradParams2 p) {
    constexpr int BO = WO * MT * 16, BJ = WJ * NT * 16;
    static_assert(BJ == 128, "column tile is 128");
    constexpr int ZPITCH = BO * sizeof(T), XPITCH = BJ * sizeof(T);
    constexpr int STAGE = 64 * (ZPITCH + XPITCH);
    constexpr int ZCPR = BO / 8;                  // chunks per Z row
    constexpr int ZRPP = 256 / ZCPR;              // Z rows per pass
    constexpr int ZPASS = (64 + ZRPP - 1) / ZRPP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wo_ = wave / WJ, wj = wave % WJ;
    const int j0 = blockIdx.x * BJ, o0 = blockIdx.y * BO;
    const int pbeg = blockIdx.z * p.pix_per_split;
    const int pend = min(p.M, pbeg + p.pix_per_split);
    const T* __restrict__ x = static_cast<const T*>(p.x);
    const T* __restrict__ dz = static_cast<const T*>(p.dz);

    // X staging: chunk column fixed per thread -> (tap, c) fixed
    const int cj = tid & 15, xrow = tid >> 4;
    const int jx = j0 + cj * 8;
    const bool jvalid = jx < p.J;
    const int tap = jx >> p.log2C;
    const int cx = jx & (p.C - 1);
    const int r = (p.ksize == 3) ? tap / 3 : 0;
    const int s = (p.ksize == 3) ? tap - 3 * r : 0;
    const int cz = tid % ZCPR, zrow = tid / ZCPR;
    const bool ovalid = (o0 + cz * 8) < p.K;

    Chunk2<T> rx[4], rz[ZPASS];
    auto gload = [&](int pb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int pix = pb + xrow + 16 * i;
            bool ok = jvalid && pix < pend;
            size_t off = 0;
            if (ok) {
                int wo, ho, n;
                if (p.lgWo >= 0 && p.lgHo >= 0) { wo = pix & (p.Wo - 1); int t = pix >> p.lgWo; ho = t & (p.Ho - 1); n = t >> p.lgHo; }
                else { wo = pix % p.Wo; int t = pix / p.Wo; ho = t % p.Ho; n = t / p.Ho; }
                int hs = ho * p.stride - p.pad + r, ws = wo * p.stride - p.pad + s;
                ok = (unsigned)hs < (unsigned)p.H && (unsigned)ws < (unsigned)p.W;
                off = ((size_t)(n * p.H + hs) * p.W + ws) * p.C + cx;
            }
            rx[i] = ok ? gload2<T>(x + off) : zero2<T>();
        }
#pragma unroll
        for (int i = 0; i < ZPASS; ++i) {
            int row = zrow + ZRPP * i;
            int pix = pb + row;
            rz[i] = (row < 64 && pix < pend && ovalid) ? gload2<T>(dz + ((size_t)pix * p.K + o0 + cz * 8)) : zero2<T>();
        }
    };
    auto sstore = [&](int stage) {
        char* Zs = smem + stage * STAGE;
        char* Xs = Zs + 64 * ZPITCH;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            char* q = Xs + (xrow + 16 * i) * XPITCH + cj * 8 * sizeof(T);
            if constexpr (sizeof(T) == 2) { *reinterpret_cast<uint4*>(q) = rx[i].a; }
            else { *reinterpret_cast<uint4*>(q) = rx[i].a; *reinterpret_cast<uint4*>(q + 16) = rx[i].b; }
        }
#pragma unroll
        for (int i = 0; i < ZPASS; ++i) {
            int row = zrow + ZRPP * i;
            if (row < 64) {
                char* q = Zs + row * ZPITCH + cz * 8 * sizeof(T);
                if constexpr (sizeof(T) == 2) { *reinterpret_cast<uint4*>(q) = rz[i].a; }
                else { *reinterpret_cast<uint4*>(q) = rz[i].a; *reinterpret_cast<uint4*>(q + 16) = rz[i].b; }
            }
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (pbeg < pend) {
        gload(pbeg);
        sstore(0);
        __syncthreads();
        int st = 0;
        for (int pb = pbeg; pb < pend; pb += 64, st ^= 1) {
            const bool more = pb + 64 < pend;
            if (more) gload(pb + 64);
            const char* Zs = smem + st * STAGE;
            const char* Xs = Zs + 64 * ZPITCH;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                Chunk2<T> zf[MT], xf[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) zf[i] = lds_ld_tr<T>(Zs, ZPITCH, ks * 32, (wo_ * MT + i) * 16, lane);
#pragma unroll
                for (int j = 0; j < NT; ++j) xf[j] = lds_ld_tr<T>(Xs, XPITCH, ks * 32, (wj * NT + j) * 16, lane);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) if (!(DBGW2(p) & 2)) acc[i][j] = mma2<T>(zf[i], xf[j], acc[i][j]);
            }
            if (more) sstore(st ^ 1);
            __syncthreads();
        }
    }

    const int fr = lane & 15, fg = lane >> 4;
    const int taps = p.ksize * p.ksize;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        int jj = j0 + (wj * NT + j) * 16 + fr;
        if (jj >= p.J) continue;
        int tp = jj >> p.log2C, c = jj & (p.C - 1);
        if (c >= p.Creal) continue;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int o = o0 + (wo_ * MT + i) * 16 + fg * 4 + e;
                if (o < p.K && !(DBGW2(p) & 8)) {
                    const size_t at = ((size_t)o * taps + tp) * p.Creal + c;
                    // every (o, tap, c) belongs to exactly one (blockIdx.x, blockIdx.y) tile, and every split's tiles run (empty ones
                    // store zeros): the slab is fully written, no zeroing launch
                    if (p.slab != nullptr) p.slab[(size_t)blockIdx.z * p.slab_stride + at] = acc[i][j][e];
                    else atomicAdd(p.dw + at, acc[i][j][e]);
                }
            }
        }
    }
}

// tile and split geometry of a launch: BO output channels x 128 (tap, channel) columns per workgroup, the pixel range cut into splits
struct Wgrad2Geo { int bo, gx, gy, splits, pps; };
Wgrad2Geo wgrad2_geometry(int M, int J, int K, bool deterministic) {
    Wgrad2Geo g;
    g.bo = K >= 128 ? 128 : (K >= 64 ? 64 : (K >= 32 ? 32 : 16));
    g.gx = (J + 127) / 128; g.gy = (K + g.bo - 1) / g.bo;
    const int tiles = g.gx * g.gy;
    const int max_splits = (M + 255) / 256;            // >= 4 K-steps per workgroup
    static const int target = clhip_cfg("WGRAD_TARGET") ? atoi(clhip_cfg("WGRAD_TARGET")) : 256;     // measured: 256 beats 128 / 512 / 1536 on every stride-2 / 1x1 / stem shape (atomic contention vs parallelism)
    int splits = (target + tiles - 1) / tiles;
    // the deterministic form writes one partial block per split and reads them all back: fewer, longer splits (at most 64)
    if (deterministic && splits > 64) splits = 64;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int pps = (M + splits - 1) / splits;
    pps = (pps + 63) / 64 * 64;
    g.splits = (M + pps - 1) / pps;
    g.pps = pps;
    return g;
}

template <typename T, int WO, int WJ, int MT, int NT>
int launch_wgrad_cfg(WgradParams2& p, hipStream_t st) {
    constexpr int BO = WO * MT * 16, BJ = 128;
    constexpr size_t lds = 2 * 64 * (size_t)(BO + BJ) * sizeof(T);
    static bool attr_set = false;
    auto kern = conv_wgrad2_kernel<T, WO, WJ, MT, NT>;
    if (!attr_set && lds > 64 * 1024) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const Wgrad2Geo g = wgrad2_geometry(p.M, p.J, p.K, p.slab != nullptr);
    static const int dbg = clhip_cfg("WGRAD_DEBUG") ? atoi(clhip_cfg("WGRAD_DEBUG")) : 0;
    p.debug = dbg;
    p.pix_per_split = g.pps;
    hipLaunchKernelGGL(kern, dim3(g.gx, g.gy, g.splits), dim3(256), lds, st, p);
    CLHIP_LAUNCH_CHECK();
    if (p.slab != nullptr) return clhip_wgrad_reduce_launch(p.slab, p.dw, p.slab_stride / 4, g.splits, st);
    return CLHIP_OK;
}

template <typename T>
int launch_wgrad2(WgradParams2& p, hipStream_t st) {
    if (p.K >= 128) return launch_wgrad_cfg<T, 2, 2, 4, 4>(p, st);     // 128 x 128
    if (p.K >= 64) return launch_wgrad_cfg<T, 1, 4, 4, 2>(p, st);      //  64 x 128
    if (p.K >= 32) return launch_wgrad_cfg<T, 1, 4, 2, 2>(p, st);      //  32 x 128
    return launch_wgrad_cfg<T, 1, 4, 1, 2>(p, st);                     //  16 x 128
}

}  // namespace

// bytes of partial-block scratch the deterministic form needs (0: the element count is not a multiple of 4 -- the reduce works on float4 --
// or the slab would exceed 64 MB: such layers keep the atomic form)
size_t clhip_wgrad2_ws_bytes(int N, int H, int W, int C, int Creal, int K, int ksize, int stride, int pad) {
    static const bool off = clhip_cfg("WGRAD2_ATOMIC") != nullptr && atoi(clhip_cfg("WGRAD2_ATOMIC")) != 0;
    if (off) return 0;
    const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    const int64_t n = (int64_t)K * ksize * ksize * Creal;
    if (n % 4 != 0) return 0;
    const Wgrad2Geo g = wgrad2_geometry(N * Ho * Wo, ksize * ksize * C, K, true);
    const size_t bytes = (size_t)g.splits * n * sizeof(float);
    return bytes <= ((size_t)64 << 20) ? bytes : 0;
}

int clhip_wgrad2_launch(const void* x, const void* dz, float* dw, float* ws, int N, int H, int W, int C, int Creal, int K, int ksize, int stride,
                        int pad, int dtype, hipStream_t st) {
    WgradParams2 p;
    p.x = x; p.dz = dz; p.dw = dw;
    p.N = N; p.H = H; p.W = W; p.C = C; p.log2C = ilog2_exact(C); p.Creal = Creal;
    p.Ho = (H + 2 * pad - ksize) / stride + 1; p.Wo = (W + 2 * pad - ksize) / stride + 1;
    p.lgHo = ilog2_exact(p.Ho); p.lgWo = ilog2_exact(p.Wo);
    p.K = K; p.ksize = ksize; p.stride = stride; p.pad = pad;
    p.M = N * p.Ho * p.Wo; p.J = ksize * ksize * C;
    // with scratch from the caller: per-split partial blocks + the fixed-order reduce (bitwise reproducible); without: fp32 atomics
    const bool det = ws != nullptr && clhip_wgrad2_ws_bytes(N, H, W, C, Creal, K, ksize, stride, pad) > 0;
    p.slab = det ? ws : nullptr;
    p.slab_stride = (int64_t)K * ksize * ksize * Creal;
    if (dtype == CLHIP_BF16) return launch_wgrad2<bf16_t>(p, st);
    return launch_wgrad2<float>(p, st);
}
