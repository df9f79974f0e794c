// conv.hip -- convolution as implicit GEMM on the CDNA4 matrix cores (gfx950).
//
// Replaces nn.Conv2d forward / dgrad / wgrad of the reference ResNets
// (core/model/backbone/resnet.py:17-24, 295-298, 337, 367).  Activations NHWC, weights K,R,S,C.
//
//   forward:  z[p][o]   = sum_{tap,c} x[p @ tap][c] * w[o][tap][c]          p = output pixel
//   dgrad  :  dx[q][c]  = sum_{tap,o} dz[q @ tap^-1][o] * wd[c][tap][o]     q = input pixel
//   wgrad  :  dw[o][tap][c] += sum_p dz[p][o] * x[p @ tap][c]
//
// forward and dgrad share one kernel: a GEMM whose "pixel" operand is gathered on the fly (im2col is
// never materialised).  MFMA orientation: D[row = out channel][col = pixel], so a lane ends up with 4
// consecutive channels of one pixel (one 8-byte store in bf16).  16x16x32 bf16 MFMA (or 8x 16x16x4
// fp32 MFMA in the exact fp32 parity mode); fp32 accumulation.  The BatchNorm batch statistics are
// reduced from the fp32 accumulators in the epilogue (per-tile partials, no atomics -> deterministic).
#include <stdlib.h>

#include <algorithm>

#include "common.h"

namespace {

constexpr int BK = 32;   // K-step: 32 reduction elements = one bf16 MFMA

struct ConvParams {
    const void* src;   // gathered tensor [N, Hs, Ws, Cs]
    const void* wt;    // [Cd][taps][Cs]
    void* dst;         // [N, Hd, Wd, Cd]
    float* stats;      // [tiles_m][2][Cd] or nullptr
    int N, Hs, Ws, Cs, log2Cs, Hd, Wd, Cd, ksize, stride, pad, accumulate;
    int M;             // N*Hd*Wd
    int K;             // taps*Cs
};

template <typename T> struct Chunk;                       // 8 elements
template <> struct Chunk<bf16_t> { uint4 a; };
template <> struct Chunk<float> { uint4 a, b; };

template <typename T> __device__ __forceinline__ Chunk<T> zero_chunk() {
    Chunk<T> c;
    if constexpr (sizeof(T) == 2) { c.a = make_uint4(0, 0, 0, 0); }
    else { c.a = make_uint4(0, 0, 0, 0); c.b = make_uint4(0, 0, 0, 0); }
    return c;
}
template <typename T> __device__ __forceinline__ Chunk<T> load_chunk(const T* p) {
    Chunk<T> c;
    if constexpr (sizeof(T) == 2) { c.a = *reinterpret_cast<const uint4*>(p); }
    else { c.a = *reinterpret_cast<const uint4*>(p); c.b = *reinterpret_cast<const uint4*>(p + 4); }
    return c;
}

// LDS tile: rows of BK elements.  bf16 rows are 64 B; the 16-B chunk index is XOR-swizzled with
// ((row>>3)&1)<<1 so that every 16-lane service group of ds_read_b128 touches 16 distinct 16-B slots
// of the 256-B bank row (derivation in DESIGN.md); fp32 rows (parity mode) are left linear.
template <typename T> __device__ __forceinline__ int lds_off(int row, int chunk) {
    if constexpr (sizeof(T) == 2) return row * 64 + ((chunk ^ (((row >> 3) & 1) << 1)) << 4);
    else return row * 128 + (chunk << 5);
}
template <typename T> __device__ __forceinline__ void lds_store(char* base, int row, int chunk, const Chunk<T>& c) {
    char* p = base + lds_off<T>(row, chunk);
    if constexpr (sizeof(T) == 2) { *reinterpret_cast<uint4*>(p) = c.a; }
    else { *reinterpret_cast<uint4*>(p) = c.a; *reinterpret_cast<uint4*>(p + 16) = c.b; }
}
template <typename T> __device__ __forceinline__ Chunk<T> lds_load(const char* base, int row, int chunk) {
    const char* p = base + lds_off<T>(row, chunk);
    Chunk<T> c;
    if constexpr (sizeof(T) == 2) { c.a = *reinterpret_cast<const uint4*>(p); }
    else { c.a = *reinterpret_cast<const uint4*>(p); c.b = *reinterpret_cast<const uint4*>(p + 16); }
    return c;
}

// acc += W-frag (rows = out channels) x X-frag (cols = pixels) over 32 reduction elements
template <typename T> __device__ __forceinline__ f32x4 mma32(const Chunk<T>& wf, const Chunk<T>& xf, f32x4 acc) {
    if constexpr (sizeof(T) == 2) {
        bf16x8_t a = __builtin_bit_cast(bf16x8_t, wf.a);
        bf16x8_t b = __builtin_bit_cast(bf16x8_t, xf.a);
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    } else {
        // lane group g owns reduction elements 8g..8g+7; MFMA j consumes element j of every group:
        // any consistent assignment of reduction indices to (lane group, MFMA) is valid for a sum.
        const float* a = reinterpret_cast<const float*>(&wf);
        const float* b = reinterpret_cast<const float*>(&xf);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
        return acc;
    }
}

// MODE 0: forward gather  (hs = hd*stride + r - pad)
// MODE 1: dgrad gather    (hs = (hd + pad - r)/stride when divisible)
template <typename T, int BM, int BN, int MODE>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvParams p) {
    constexpr int MT = BM / 64;            // 16-pixel tiles per wave (4 waves split the BM pixels)
    constexpr int NT = BN / 16;            // 16-channel tiles
    constexpr int AROWS = BM / 64;         // A-tile rows staged per thread
    constexpr int BROWS = (BN + 63) / 64;  // B-tile rows staged per thread
    constexpr int ROWB = BK * sizeof(T);
    __shared__ __attribute__((aligned(16))) char smem[(BM + BN) * ROWB + 4 * BN * 2 * sizeof(float)];
    char* As = smem;
    char* Bs = smem + BM * ROWB;
    float* red = reinterpret_cast<float*>(smem + (BM + BN) * ROWB);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const T* __restrict__ src = static_cast<const T*>(p.src);
    const T* __restrict__ wt = static_cast<const T*>(p.wt);

    // ---- per-thread staging coordinates (fixed for the whole K loop)
    const int ca = tid & 3;                 // chunk column within the 32-wide K step
    int a_base[AROWS], a_h0[AROWS], a_w0[AROWS];
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
        int pix = m0 + (tid >> 2) + 64 * i;
        if (pix < p.M) {
            int wd = pix % p.Wd;
            int t = pix / p.Wd;
            int hd = t % p.Hd;
            int n = t / p.Hd;
            a_base[i] = n * p.Hs * p.Ws;
            if (MODE == 0) { a_h0[i] = hd * p.stride - p.pad; a_w0[i] = wd * p.stride - p.pad; }
            else { a_h0[i] = hd + p.pad; a_w0[i] = wd + p.pad; }
        } else {
            a_base[i] = -1; a_h0[i] = 0; a_w0[i] = 0;
        }
    }

    Chunk<T> ra[AROWS], rb[BROWS];
    auto gload = [&](int kstep) {
        const int kk = kstep * BK + ca * 8;
        const bool kvalid = kk < p.K;
        const int tap = kk >> p.log2Cs;
        const int cs = kk & (p.Cs - 1);
        const int r = (p.ksize == 3) ? tap / 3 : 0;
        const int s = (p.ksize == 3) ? tap - 3 * r : 0;
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
            bool ok = kvalid && a_base[i] >= 0;
            int hs, ws;
            if (MODE == 0) {
                hs = a_h0[i] + r; ws = a_w0[i] + s;
            } else {
                int th = a_h0[i] - r, tw = a_w0[i] - s;
                ok = ok && th >= 0 && tw >= 0;
                if (p.stride == 1) { hs = th; ws = tw; }
                else { ok = ok && (th % p.stride == 0) && (tw % p.stride == 0); hs = th / p.stride; ws = tw / p.stride; }
            }
            ok = ok && (unsigned)hs < (unsigned)p.Hs && (unsigned)ws < (unsigned)p.Ws;
            if (ok) ra[i] = load_chunk<T>(src + ((size_t)(a_base[i] + hs * p.Ws + ws) * p.Cs + cs));
            else ra[i] = zero_chunk<T>();
        }
#pragma unroll
        for (int i = 0; i < BROWS; ++i) {
            int row = (tid >> 2) + 64 * i;
            int o = n0 + row;
            if (row < BN && o < p.Cd && kvalid) rb[i] = load_chunk<T>(wt + ((size_t)o * p.K + kk));
            else rb[i] = zero_chunk<T>();
        }
    };
    auto sstore = [&]() {
#pragma unroll
        for (int i = 0; i < AROWS; ++i) lds_store<T>(As, (tid >> 2) + 64 * i, ca, ra[i]);
#pragma unroll
        for (int i = 0; i < BROWS; ++i) {
            int row = (tid >> 2) + 64 * i;
            if (row < BN) lds_store<T>(Bs, row, ca, rb[i]);
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    const int fr = lane & 15, fg = lane >> 4;
    gload(0);
    sstore();
    __syncthreads();
    for (int k = 0; k < nk; ++k) {
        if (k + 1 < nk) gload(k + 1);          // global loads in flight while the MFMAs run
        Chunk<T> xf[MT], wf[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) xf[i] = lds_load<T>(As, wave * (BM / 4) + i * 16 + fr, fg);
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[j] = lds_load<T>(Bs, j * 16 + fr, fg);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = mma32<T>(wf[j], xf[i], acc[i][j]);
        __syncthreads();
        if (k + 1 < nk) {
            sstore();
            __syncthreads();
        }
    }

    // ---- epilogue: D[row = channel fg*4+e][col = pixel fr]
    T* __restrict__ dst = static_cast<T*>(p.dst);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int pix = m0 + wave * (BM / 4) + i * 16 + fr;
        if (pix < p.M) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int o = n0 + j * 16 + fg * 4;
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

    if (p.stats != nullptr) {
        // per-channel sum / sum of squares over this tile's pixels (rows beyond M hold exact zeros)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int i = 0; i < MT; ++i) { float v = acc[i][j][e]; s1 += v; s2 += v * v; }
                s1 = row16_sum(s1);
                s2 = row16_sum(s2);
                if (fr == 0) {
                    int c = j * 16 + fg * 4 + e;
                    red[(wave * 2 + 0) * BN + c] = s1;
                    red[(wave * 2 + 1) * BN + c] = s2;
                }
            }
        }
        __syncthreads();
        if (tid < 2 * BN) {
            int which = tid / BN, c = tid - which * BN;
            float t = red[(0 * 2 + which) * BN + c] + red[(1 * 2 + which) * BN + c] + red[(2 * 2 + which) * BN + c] +
                      red[(3 * 2 + which) * BN + c];
            if (n0 + c < p.Cd) p.stats[((size_t)blockIdx.x * 2 + which) * p.Cd + n0 + c] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------ wgrad
struct WgradParams {
    const void* x;    // [N,H,W,C]
    const void* dz;   // [N,Ho,Wo,K]
    float* dw;        // [K][taps][Creal] fp32, atomically accumulated
    int N, H, W, C, log2C, Creal, Ho, Wo, K, ksize, stride, pad;
    int M;            // N*Ho*Wo
    int J;            // taps*C
    int pix_per_split;
};

// Fragment of the TRANSPOSED LDS tile: tile is [32 pixels][64 channels] (row = pixel); the MFMA
// operand wants, for channel column `col`, the 8 pixels 8g..8g+7 of lane group g.
template <typename T, bool TR>
__device__ __forceinline__ Chunk<T> lds_load_transposed(const char* tile, int col0, int lane) {
    const int fr = lane & 15, fg = lane >> 4;
    Chunk<T> c;
    if constexpr (sizeof(T) == 2) {
        constexpr int ROWB = 64 * 2;
        if constexpr (TR) {
            // ds_read_b64_tr_b16: the 16 lanes of a group fetch a [4 pixels][16 channels] block (lane i:
            // pixel i>>2, channels 4*(i&3)..+3) and receive it transposed (lane i: channel i, 4 pixels).
            const char* a0 = tile + (fg * 8 + (fr >> 2)) * ROWB + (col0 + (fr & 3) * 4) * 2;
            typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * ROWB));
            uint2 l = __builtin_bit_cast(uint2, lo), h = __builtin_bit_cast(uint2, hi);
            c.a = make_uint4(l.x, l.y, h.x, h.y);
        } else {
            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t e0 = *reinterpret_cast<const uint16_t*>(tile + (fg * 8 + 2 * j) * ROWB + (col0 + fr) * 2);
                uint32_t e1 = *reinterpret_cast<const uint16_t*>(tile + (fg * 8 + 2 * j + 1) * ROWB + (col0 + fr) * 2);
                w[j] = e0 | (e1 << 16);
            }
            c.a = make_uint4(w[0], w[1], w[2], w[3]);
        }
    } else {
        constexpr int ROWB = 64 * 4;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float*>(tile + (fg * 8 + j) * ROWB + (col0 + fr) * 4);
        c.a = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
        c.b = make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7]));
    }
    return c;
}

// grid: x = J tiles (64 (tap,c) columns), y = K tiles (64 out channels), z = pixel splits
template <typename T, bool TR>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradParams p) {
    constexpr int ROWB = 64 * sizeof(T);
    __shared__ __attribute__((aligned(16))) char smem[2 * 32 * ROWB];
    char* Zs = smem;                 // [32 pixels][64 out channels]
    char* Xs = smem + 32 * ROWB;     // [32 pixels][64 (tap,c) columns]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j0 = blockIdx.x * 64, o0 = blockIdx.y * 64;
    const int pbeg = blockIdx.z * p.pix_per_split;
    const int pend = min(p.M, pbeg + p.pix_per_split);
    const T* __restrict__ x = static_cast<const T*>(p.x);
    const T* __restrict__ dz = static_cast<const T*>(p.dz);

    const int row = tid >> 3, ch = tid & 7;          // staging: one 8-element chunk per thread per tile
    // (tap, c) of this thread's X chunk is fixed over the pixel loop
    const int jx = j0 + ch * 8;
    const bool jvalid = jx < p.J;
    const int tap = jx >> p.log2C;
    const int cx = jx & (p.C - 1);
    const int r = (p.ksize == 3) ? tap / 3 : 0;
    const int s = (p.ksize == 3) ? tap - 3 * r : 0;
    const bool ovalid = (o0 + ch * 8) < p.K;

    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    Chunk<T> rz, rx;
    auto gload = [&](int pb) {
        int pix = pb + row;
        bool pv = pix < pend;
        rz = (pv && ovalid) ? load_chunk<T>(dz + ((size_t)pix * p.K + o0 + ch * 8)) : zero_chunk<T>();
        bool ok = pv && jvalid;
        size_t off = 0;
        if (ok) {
            int wo = pix % p.Wo;
            int t = pix / p.Wo;
            int ho = t % p.Ho;
            int n = t / p.Ho;
            int hs = ho * p.stride - p.pad + r, ws = wo * p.stride - p.pad + s;
            ok = (unsigned)hs < (unsigned)p.H && (unsigned)ws < (unsigned)p.W;
            off = ((size_t)(n * p.H + hs) * p.W + ws) * p.C + cx;
        }
        rx = ok ? load_chunk<T>(x + off) : zero_chunk<T>();
    };
    auto sstore = [&]() {
        char* zq = Zs + row * ROWB + ch * 8 * sizeof(T);
        char* xq = Xs + row * ROWB + ch * 8 * sizeof(T);
        if constexpr (sizeof(T) == 2) {
            *reinterpret_cast<uint4*>(zq) = rz.a;
            *reinterpret_cast<uint4*>(xq) = rx.a;
        } else {
            *reinterpret_cast<uint4*>(zq) = rz.a; *reinterpret_cast<uint4*>(zq + 16) = rz.b;
            *reinterpret_cast<uint4*>(xq) = rx.a; *reinterpret_cast<uint4*>(xq + 16) = rx.b;
        }
    };

    if (pbeg < pend) {
        gload(pbeg);
        sstore();
        __syncthreads();
        for (int pb = pbeg; pb < pend; pb += 32) {
            const bool more = pb + 32 < pend;
            if (more) gload(pb + 32);
            Chunk<T> zf = lds_load_transposed<T, TR>(Zs, wave * 16, lane);   // rows = 16 out channels of this wave
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                Chunk<T> xf = lds_load_transposed<T, TR>(Xs, j * 16, lane);  // cols = 16 (tap,c) columns
                acc[j] = mma32<T>(zf, xf, acc[j]);
            }
            __syncthreads();
            if (more) {
                sstore();
                __syncthreads();
            }
        }
    }

    // D[row = out channel fg*4+e][col = (tap,c) column fr]
    const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int jj = j0 + j * 16 + fr;
        if (jj >= p.J) continue;
        int tp = jj >> p.log2C, c = jj & (p.C - 1);
        if (c >= p.Creal) continue;
        int taps = p.ksize * p.ksize;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int o = o0 + wave * 16 + fg * 4 + e;
            if (o < p.K) atomicAdd(p.dw + ((size_t)o * taps + tp) * p.Creal + c, acc[j][e]);
        }
    }
}

template <typename T, int MODE>
int launch_igemm(const ConvParams& p, hipStream_t st) {
    const int Cd = p.Cd;
    int BN = Cd >= 128 ? 128 : (Cd >= 64 ? 64 : (Cd >= 32 ? 32 : 16));
    int gy = (Cd + BN - 1) / BN;
    bool small = ((int64_t)((p.M + 127) / 128) * gy) < 512;
    int BM = small ? 64 : 128;
    dim3 grid((p.M + BM - 1) / BM, gy);
#define LAUNCH(bm, bn) hipLaunchKernelGGL((conv_igemm_kernel<T, bm, bn, MODE>), grid, dim3(256), 0, st, p)
    if (BM == 128) {
        if (BN == 128) LAUNCH(128, 128); else if (BN == 64) LAUNCH(128, 64); else if (BN == 32) LAUNCH(128, 32); else LAUNCH(128, 16);
    } else {
        if (BN == 128) LAUNCH(64, 128); else if (BN == 64) LAUNCH(64, 64); else if (BN == 32) LAUNCH(64, 32); else LAUNCH(64, 16);
    }
#undef LAUNCH
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

int fwd_tile_m(int M, int Cd) {
    int BN = Cd >= 128 ? 128 : (Cd >= 64 ? 64 : (Cd >= 32 ? 32 : 16));
    int gy = (Cd + BN - 1) / BN;
    bool small = ((int64_t)((M + 127) / 128) * gy) < 512;
    return small ? 64 : 128;
}

int check_conv(int N, int H, int W, int C, int K, int ksize, int stride, int pad) {
    CLHIP_CHECK_ARG(N > 0 && H > 0 && W > 0);
    CLHIP_CHECK_ARG(ksize == 1 || ksize == 3);
    CLHIP_CHECK_ARG(stride >= 1 && pad >= 0);
    CLHIP_CHECK_ARG(C >= 8 && ilog2_exact(C) >= 0);     // channel counts are powers of two (pad the stem to 8)
    CLHIP_CHECK_ARG(K >= 16 && K % 16 == 0);
    CLHIP_CHECK_ARG((int64_t)N * H * W * (C > K ? C : K) < (int64_t)1 << 31);
    return CLHIP_OK;
}

}  // namespace

// conv2.hip
int clhip_conv2_tiles_m(int M, int Cd);
int clhip_conv2_launch(const void* src, const void* wt, void* dst, float* stats, double* stat_acc, int stat_rep, int N, int Hs, int Ws, int Cs, int Hd, int Wd,
                       int Cd, int ksize, int stride, int pad, int accumulate, int mode, int dtype, hipStream_t st);
bool clhip_conv64_supported(int N, int H, int W, int Cs, int Cd, int ksize, int stride, int pad, int dtype);      // conv3.hip
int clhip_conv64_launch_ex(const void* src, const void* wt, void* dst, double* stat_acc, int stat_rep, int N, int H, int W, int accumulate, int mode,
                           const void* bn_z, const void* bn_y, const float* bn_mean, const float* bn_invstd, double* bn_acc, int bn_rep, const float* bn_coef,
                           const clhip_bn_input* in, hipStream_t st, const clhip_bn_res_input* rs = nullptr);

bool clhip_wgrad64_supported(int N, int H, int W, int C, int Creal, int K, int ksize, int stride, int pad, int dtype);      // conv3.hip
size_t clhip_wgrad64_ws_bytes(int N);
int clhip_wgrad64_launch(const void* x, const void* dz, float* dw, float* ws, int N, int H, const float* x_coef, hipStream_t st);
bool clhip_wgrad32_supported(int N, int H, int W, int C, int Creal, int K, int ksize, int stride, int pad, int dtype);      // conv3.hip
size_t clhip_wgrad32_ws_bytes(int N);
int clhip_wgrad32_launch(const void* x, const void* dz, float* dw, float* ws, int N, int H, const float* x_coef, hipStream_t st);
size_t clhip_wgrad2_ws_bytes(int N, int H, int W, int C, int Creal, int K, int ksize, int stride, int pad);
int clhip_wgrad2_launch(const void* x, const void* dz, float* dw, float* ws, int N, int H, int W, int C, int Creal, int K, int ksize, int stride,
                        int pad, int dtype, hipStream_t st);
bool clhip_conv3_supported(int H, int W, int Cs, int Cd, int ksize, int stride, int pad, int dtype);
int clhip_conv3_tiles_m(int M, int Cd);
int clhip_conv3_launch(const void* src, const void* wt, void* dst, float* stats, double* stat_acc, int stat_rep, int N, int H, int W, int Cs, int Cd, int accumulate,
                       int mode, hipStream_t st);
bool clhip_conv4_supported(int N, int H, int W, int Cs, int Cd, int ksize, int stride, int pad, int dtype);
int clhip_conv4_tiles_m(int M, int Cs, int Cd, int W);
int clhip_conv4_launch(const void* src, const void* wt, void* dst, float* stats, double* stat_acc, int stat_rep, int N, int H, int W, int Cs, int Cd, int accumulate,
                       int mode, hipStream_t st);
int clhip_conv4_launch_bn(const void* src, const void* wt, void* dst, float* stats, double* stat_acc, int stat_rep, int N, int H, int W, int Cs, int Cd,
                          int accumulate, int mode, const void* bn_z, const void* bn_y, const float* bn_mean, const float* bn_invstd, double* bn_acc, int bn_rep,
                          hipStream_t st);
bool clhip_conv16_supported(int H, int W, int Cs, int Cd, int ksize, int stride, int pad, int dtype);
int clhip_conv16_tiles_m(int M);
int clhip_conv16_launch(const void* src, const void* wt, void* dst, float* stats, double* stat_acc, int stat_rep, int N, int H, int W, int C, int accumulate, int mode,
                        hipStream_t st);
bool clhip_wgrad3_supported(int N, int H, int W, int C, int Creal, int K, int ksize, int stride, int pad, int dtype);
bool clhip_wgrad16_supported(int N, int H, int W, int C, int Creal, int K, int ksize, int stride, int pad, int dtype);
size_t clhip_wgrad16_ws_bytes(int N);
int clhip_wgrad16_launch(const void* x, const void* dz, float* dw, float* ws, int N, int H, const float* x_coef, hipStream_t st);
int clhip_wgrad3_launch(const void* x, const void* dz, float* dw, float* ws, int N, int H, int W, int C, int Creal, int K, hipStream_t st);
size_t clhip_wgrad3_ws_bytes(int N, int H, int W, int C, int K);
bool clhip_stem_supported(int N, int H, int W, int C, int K, int ksize, int stride, int pad, int dtype);
int clhip_stem_launch(const void* x, const void* w, void* z, double* acc, int rep, int N, int H, int W, int K, hipStream_t st);
bool clhip_shortcut_supported(int N, int H, int W, int C, int K, int ksize, int stride, int pad, int dtype);
int clhip_shortcut_dgrad(const void* dz, const void* w_dg, void* dx, int accumulate, int N, int H, int W, int C, int K, hipStream_t st);
bool clhip_stem_wgrad_supported(int N, int H, int W, int C, int Creal, int K, int ksize, int stride, int pad, int dtype);
size_t clhip_stem_wgrad_ws_bytes(int N, int H, int W, int Creal, int K);
int clhip_stem_wgrad_launch(const void* x, const void* dz, float* dw, float* ws, int N, int H, int W, int Creal, int K, hipStream_t st);
bool clhip_wgrad4_supported(int N, int H, int W, int C, int Creal, int K, int ksize, int stride, int pad, int dtype);
size_t clhip_wgrad4_ws_bytes(int N, int H, int W, int C, int K, int ksize, int stride);
int clhip_wgrad4_launch(const void* x, const void* dz, float* dw, float* ws, int N, int H, int W, int C, int K, int ksize, int stride, hipStream_t st);
bool clhip_conv5_supported(int N, int H, int W, int Cs, int Cd, int ksize, int stride, int pad, int dtype);      // conv5.hip
int clhip_conv5_launch(const void* src, const void* wt, void* dst, double* stat_acc, int stat_rep, int N, int H, int W, int accumulate, int mode, hipStream_t st);
bool clhip_conv8_supported(int N, int H, int W, int Cs, int Cd, int ksize, int stride, int pad, int dtype);      // conv8.hip
int clhip_conv8_launch(const void* src, const void* wt, void* dst, double* stat_acc, int stat_rep, int N, int H, int W, int accumulate, int mode, const LazyIn* in,
                       const void* bn_z, const void* bn_y, const void* bn_mask, const float* bn_gamma, const float* bn_beta, const float* bn_mean, const float* bn_invstd,
                       double* bn_acc, int bn_rep, hipStream_t st);
bool clhip_conv9_supported(int N, int H, int W, int Cs, int Cd, int ksize, int stride, int pad, int dtype);      // conv9.hip
int clhip_conv9_launch(const void* src, const void* wt, void* dst, double* stat_acc, int stat_rep, int N, int H, int W, int C, int accumulate, int mode, const LazyIn* in,
                       const void* bn_z, const void* bn_y, const float* bn_mean, const float* bn_invstd, double* bn_acc, int bn_rep, hipStream_t st);
static bool use_v3() {
    static const bool v = clhip_cfg("NO_CONV3") == nullptr;    // A/B switch: halo kernel for 3x3 stride-1 layers
    return v;
}
static bool use_v1() {
    static const bool v = clhip_cfg("CONV_V1") != nullptr;     // A/B switch: first-generation kernel
    return v;
}

extern "C" int clhip_conv_fwd_tiles(int N, int H, int W, int C, int K, int ksize, int stride, int pad) {
    int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    int M = N * Ho * Wo;
    // NB: the plan sizes the statistics scratch with this; dtype is not known here, so report the larger count
    if (!use_v1()) {
        int t2 = clhip_conv2_tiles_m(M, K);
        if (use_v3() && clhip_conv16_supported(H, W, C, K, ksize, stride, pad, CLHIP_BF16)) {
            int t16 = clhip_conv16_tiles_m(M);
            return t16 > t2 ? t16 : t2;
        }
        if (use_v3() && clhip_conv3_supported(H, W, C, K, ksize, stride, pad, CLHIP_BF16)) {
            int t3 = clhip_conv3_tiles_m(M, K);
            if (clhip_conv4_supported(N, H, W, C, K, ksize, stride, pad, CLHIP_BF16)) { int t4 = clhip_conv4_tiles_m(M, C, K, W); if (t4 > t3) t3 = t4; }
            return t3 > t2 ? t3 : t2;
        }
        return t2;
    }
    int bm = fwd_tile_m(M, K);
    return (M + bm - 1) / bm;
}

static bool conv64_fwd_on() { const char* c = clhip_cfg("CONV64_FWD"); return !(c != nullptr && atoi(c) == 0); }

static int conv_fwd_impl(const void* x, const void* w_fwd, void* z, float* stat_partials, double* stat_acc, int stat_rep, int N, int H, int W, int C,
                         int K, int ksize, int stride, int pad, int dtype, void* stream);

extern "C" int clhip_conv_fwd(const void* x, const void* w_fwd, void* z, float* stat_partials, int N, int H, int W, int C,
                              int K, int ksize, int stride, int pad, int dtype, void* stream) {
    return conv_fwd_impl(x, w_fwd, z, stat_partials, nullptr, 1, N, H, W, C, K, ksize, stride, pad, dtype, stream);
}

extern "C" int clhip_conv_fwd_acc(const void* x, const void* w_fwd, void* z, double* stat_acc, int replicas, int N, int H, int W, int C,
                                  int K, int ksize, int stride, int pad, int dtype, void* stream) {
    CLHIP_CHECK_ARG(stat_acc != nullptr && replicas >= 1 && replicas <= 64 && (replicas & (replicas - 1)) == 0);
    return conv_fwd_impl(x, w_fwd, z, nullptr, stat_acc, replicas, N, H, W, C, K, ksize, stride, pad, dtype, stream);
}

static int conv_fwd_impl(const void* x, const void* w_fwd, void* z, float* stat_partials, double* stat_acc, int stat_rep, int N, int H, int W, int C,
                         int K, int ksize, int stride, int pad, int dtype, void* stream) {
    if (int e = check_conv(N, H, W, C, K, ksize, stride, pad)) return e;
    CLHIP_CHECK_ARG(x && w_fwd && z);
    CLHIP_CHECK_ARG(!(stat_acc && use_v1()));
    ConvParams p;
    p.src = x; p.wt = w_fwd; p.dst = z; p.stats = stat_partials;
    p.N = N; p.Hs = H; p.Ws = W; p.Cs = C; p.log2Cs = ilog2_exact(C);
    p.Hd = (H + 2 * pad - ksize) / stride + 1; p.Wd = (W + 2 * pad - ksize) / stride + 1; p.Cd = K;
    p.ksize = ksize; p.stride = stride; p.pad = pad; p.accumulate = 0;
    p.M = N * p.Hd * p.Wd; p.K = ksize * ksize * C;
    hipStream_t st = static_cast<hipStream_t>(stream);
    CLHIP_CHECK_ARG(dtype == CLHIP_BF16 || dtype == CLHIP_F32);
    // the stems (<= 8 padded input channels): no LDS, weights in registers; serves the accumulator and the no-statistics forms
    if (!use_v1() && use_v3() && stat_partials == nullptr && clhip_stem_supported(N, H, W, C, K, ksize, stride, pad, dtype))
        return clhip_stem_launch(x, w_fwd, z, stat_acc, stat_rep, N, H, W, K, st);
    // 64 -> 64 channels on small maps: register-resident filters, out channels split over the waves (statistics through the accumulators only)
    if (!use_v1() && use_v3() && stat_partials == nullptr && conv64_fwd_on() && clhip_conv64_supported(N, H, W, C, K, ksize, stride, pad, dtype))
        return clhip_conv64_launch_ex(x, w_fwd, z, stat_acc, stat_rep, N, H, W, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 1, nullptr, nullptr, st);
    if (!use_v1() && use_v3() && clhip_conv16_supported(H, W, C, K, ksize, stride, pad, dtype)) {
        int tiles_alloc = clhip_conv_fwd_tiles(N, H, W, C, K, ksize, stride, pad);
        int tiles_used = clhip_conv16_tiles_m(p.M);
        if (stat_partials && tiles_alloc > tiles_used)
            hipMemsetAsync(stat_partials + (size_t)tiles_used * 2 * K, 0, (size_t)(tiles_alloc - tiles_used) * 2 * K * sizeof(float), st);
        return clhip_conv16_launch(x, w_fwd, z, stat_partials, stat_acc, stat_rep, N, H, W, C, 0, 0, st);
    }
    // 64 -> 64 channels on large activations: two four-wave workgroups per CU, the filters of 32 output channels resident in each wave (conv8.hip) ...
    if (!use_v1() && use_v3() && stat_partials == nullptr && clhip_conv8_supported(N, H, W, C, K, ksize, stride, pad, dtype))
        return clhip_conv8_launch(x, w_fwd, z, stat_acc, stat_rep, N, H, W, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, st);
    // ... or the weight-stationary kernel with one 512-register wave per SIMD (statistics through the accumulators only)
    if (!use_v1() && use_v3() && stat_partials == nullptr && clhip_conv5_supported(N, H, W, C, K, ksize, stride, pad, dtype))
        return clhip_conv5_launch(x, w_fwd, z, stat_acc, stat_rep, N, H, W, 0, 0, st);
    // 128 -> 128 / 256 -> 256 channels: resident patch, one barrier per 32-KB filter slab (conv9.hip; statistics through the accumulators only)
    if (!use_v1() && use_v3() && stat_partials == nullptr && clhip_conv9_supported(N, H, W, C, K, ksize, stride, pad, dtype))
        return clhip_conv9_launch(x, w_fwd, z, stat_acc, stat_rep, N, H, W, C, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, st);
    if (!use_v1() && use_v3() && clhip_conv4_supported(N, H, W, C, K, ksize, stride, pad, dtype)) {
        int tiles_alloc = clhip_conv_fwd_tiles(N, H, W, C, K, ksize, stride, pad);
        int tiles_used = clhip_conv4_tiles_m(p.M, C, K, W);
        if (stat_partials && tiles_alloc > tiles_used)
            hipMemsetAsync(stat_partials + (size_t)tiles_used * 2 * K, 0, (size_t)(tiles_alloc - tiles_used) * 2 * K * sizeof(float), st);
        return clhip_conv4_launch(x, w_fwd, z, stat_partials, stat_acc, stat_rep, N, H, W, C, K, 0, 0, st);
    }
    if (!use_v1() && use_v3() && clhip_conv3_supported(H, W, C, K, ksize, stride, pad, dtype)) {
        // the caller's partial buffer may hold more tiles than this kernel writes: zero the tail rows
        int tiles_alloc = clhip_conv_fwd_tiles(N, H, W, C, K, ksize, stride, pad);
        int tiles_used = clhip_conv3_tiles_m(p.M, K);
        if (stat_partials && tiles_alloc > tiles_used)
            hipMemsetAsync(stat_partials + (size_t)tiles_used * 2 * K, 0, (size_t)(tiles_alloc - tiles_used) * 2 * K * sizeof(float), st);
        return clhip_conv3_launch(x, w_fwd, z, stat_partials, stat_acc, stat_rep, N, H, W, C, K, 0, 0, st);
    }
    if (!use_v1()) {
        int tiles_alloc = clhip_conv_fwd_tiles(N, H, W, C, K, ksize, stride, pad);
        int tiles_used = clhip_conv2_tiles_m(p.M, K);
        if (stat_partials && tiles_alloc > tiles_used)
            hipMemsetAsync(stat_partials + (size_t)tiles_used * 2 * K, 0, (size_t)(tiles_alloc - tiles_used) * 2 * K * sizeof(float), st);
        return clhip_conv2_launch(x, w_fwd, z, stat_partials, stat_acc, stat_rep, N, H, W, C, p.Hd, p.Wd, K, ksize, stride, pad, 0, 0, dtype, st);
    }
    if (dtype == CLHIP_BF16) return launch_igemm<bf16_t, 0>(p, st);
    if (dtype == CLHIP_F32) return launch_igemm<float, 0>(p, st);
    CLHIP_CHECK_ARG(!"dtype");
    return CLHIP_EINVAL;
}

extern "C" int clhip_conv_dgrad(const void* dz, const void* w_dg, void* dx, int accumulate, int N, int H, int W, int C,
                                int K, int ksize, int stride, int pad, int dtype, void* stream) {
    if (int e = check_conv(N, H, W, C, K, ksize, stride, pad)) return e;
    CLHIP_CHECK_ARG(dz && w_dg && dx);
    CLHIP_CHECK_ARG(C % 16 == 0);
    ConvParams p;
    p.src = dz; p.wt = w_dg; p.dst = dx; p.stats = nullptr;
    p.N = N; p.Hs = (H + 2 * pad - ksize) / stride + 1; p.Ws = (W + 2 * pad - ksize) / stride + 1; p.Cs = K;
    p.log2Cs = ilog2_exact(K);
    CLHIP_CHECK_ARG(p.log2Cs >= 0);
    p.Hd = H; p.Wd = W; p.Cd = C;
    p.ksize = ksize; p.stride = stride; p.pad = pad; p.accumulate = accumulate;
    p.M = N * H * W; p.K = ksize * ksize * K;
    hipStream_t st = static_cast<hipStream_t>(stream);
    CLHIP_CHECK_ARG(dtype == CLHIP_BF16 || dtype == CLHIP_F32);
    if (!use_v1() && use_v3() && clhip_shortcut_supported(N, H, W, C, K, ksize, stride, pad, dtype))
        return clhip_shortcut_dgrad(dz, w_dg, dx, accumulate, N, H, W, C, K, st);
    if (!use_v1() && use_v3() && clhip_conv16_supported(H, W, K, C, ksize, stride, pad, dtype))
        return clhip_conv16_launch(dz, w_dg, dx, nullptr, nullptr, 1, N, H, W, C, accumulate, 1, st);
    if (!use_v1() && use_v3() && clhip_conv64_supported(N, H, W, K, C, ksize, stride, pad, dtype))
        return clhip_conv64_launch_ex(dz, w_dg, dx, nullptr, 1, N, H, W, accumulate, 1, nullptr, nullptr, nullptr, nullptr, nullptr, 1, nullptr, nullptr, st);
    if (!use_v1() && use_v3() && clhip_conv8_supported(N, H, W, K, C, ksize, stride, pad, dtype))
        return clhip_conv8_launch(dz, w_dg, dx, nullptr, 1, N, H, W, accumulate, 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, st);
    if (!use_v1() && use_v3() && clhip_conv5_supported(N, H, W, K, C, ksize, stride, pad, dtype))
        return clhip_conv5_launch(dz, w_dg, dx, nullptr, 1, N, H, W, accumulate, 1, st);
    if (!use_v1() && use_v3() && clhip_conv9_supported(N, H, W, K, C, ksize, stride, pad, dtype))
        return clhip_conv9_launch(dz, w_dg, dx, nullptr, 1, N, H, W, C, accumulate, 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, st);
    if (!use_v1() && use_v3() && clhip_conv4_supported(N, H, W, K, C, ksize, stride, pad, dtype))
        return clhip_conv4_launch(dz, w_dg, dx, nullptr, nullptr, 1, N, H, W, K, C, accumulate, 1, st);
    if (!use_v1() && use_v3() && clhip_conv3_supported(H, W, K, C, ksize, stride, pad, dtype))
        return clhip_conv3_launch(dz, w_dg, dx, nullptr, nullptr, 1, N, H, W, K, C, accumulate, 1, st);
    if (!use_v1()) return clhip_conv2_launch(dz, w_dg, dx, nullptr, nullptr, 1, N, p.Hs, p.Ws, K, H, W, C, ksize, stride, pad, accumulate, 1, dtype, st);
    if (dtype == CLHIP_BF16) return launch_igemm<bf16_t, 1>(p, st);
    if (dtype == CLHIP_F32) return launch_igemm<float, 1>(p, st);
    CLHIP_CHECK_ARG(!"dtype");
    return CLHIP_EINVAL;
}

int clhip_conv16_launch_bn(const void* src, const void* wt, void* dst, float* stats, double* stat_acc, int stat_rep, int N, int H, int W, int C, int accumulate, int mode,
                           const void* bn_z, const void* bn_y, const float* bn_mean, const float* bn_invstd, double* bn_acc, int bn_rep, hipStream_t st);      // conv3.hip

extern "C" int clhip_conv_dgrad_bn_reduce_supported(int N, int H, int W, int C, int K, int ksize, int stride, int pad, int dtype) {
    if (check_conv(N, H, W, C, K, ksize, stride, pad) != CLHIP_OK) return 0;
    if (use_v1() || !use_v3()) return 0;
    return (clhip_conv16_supported(H, W, K, C, ksize, stride, pad, dtype) || clhip_conv64_supported(N, H, W, K, C, ksize, stride, pad, dtype) ||
            clhip_conv8_supported(N, H, W, K, C, ksize, stride, pad, dtype) || clhip_conv4_supported(N, H, W, K, C, ksize, stride, pad, dtype)) ? 1 : 0;
}

// ... and where that epilogue is hidden (conv8.hip: the other workgroup of the CU multiplies meanwhile), so that the plan fuses it on large maps too
extern "C" int clhip_conv_dgrad_bn_reduce_overlapped(int N, int H, int W, int C, int K, int ksize, int stride, int pad, int dtype) {
    if (check_conv(N, H, W, C, K, ksize, stride, pad) != CLHIP_OK || use_v1() || !use_v3()) return 0;
    if (clhip_conv16_supported(H, W, K, C, ksize, stride, pad, dtype) || clhip_conv64_supported(N, H, W, K, C, ksize, stride, pad, dtype)) return 0;
    // OFF by default (CONV8_BNR=1 enables it).  Correct with every mask source (tests/test_kernels_gpu.py), z' staged through LDS by the wave that needs
    // it -- and the ResNet-18 step is 5 % SLOWER with the four layer-1 reductions fused (2.14 vs 2.02 ms, r05 A/B): the layer's backward is HBM-bound,
    // the fused launch takes 47-60 us instead of 22-32 (it reads z' on top of its 67 MB beside the weight-gradient stream), the apply pass that follows
    // loses the cache hits the reduce pass used to leave it (35-41 vs 19-24 us), and what disappears is a 26-us launch -- profiles/r05_conv_notes.md
    const char* cfg = clhip_cfg("CONV8_BNR");
    if (cfg == nullptr || atoi(cfg) == 0) return 0;
    return clhip_conv8_supported(N, H, W, K, C, ksize, stride, pad, dtype) ? 1 : 0;
}

extern "C" int clhip_conv_dgrad_bn_reduce(const void* dz, const void* w_dg, void* dx, int accumulate, const void* z_prod, const void* y_prod,
                                          const float* mean, const float* invstd, double* acc, int replicas, int N, int H, int W, int C, int K,
                                          int ksize, int stride, int pad, int dtype, void* stream) {
    return clhip_conv_dgrad_bn_reduce_ex(dz, w_dg, dx, accumulate, z_prod, y_prod, nullptr, nullptr, nullptr, mean, invstd, acc, replicas, N, H, W, C, K, ksize, stride, pad, dtype, stream);
}

extern "C" int clhip_conv_dgrad_bn_reduce_ex(const void* dz, const void* w_dg, void* dx, int accumulate, const void* z_prod, const void* y_prod, const void* mask_prod,
                                             const float* gamma_prod, const float* beta_prod, const float* mean, const float* invstd, double* acc, int replicas, int N, int H, int W, int C,
                                             int K, int ksize, int stride, int pad, int dtype, void* stream) {
    if (int e = check_conv(N, H, W, C, K, ksize, stride, pad)) return e;
    CLHIP_CHECK_ARG(dz && w_dg && dx && z_prod && mean && invstd && acc);
    CLHIP_CHECK_ARG(replicas >= 1 && replicas <= 64 && (replicas & (replicas - 1)) == 0);
    CLHIP_CHECK_ARG(clhip_conv_dgrad_bn_reduce_supported(N, H, W, C, K, ksize, stride, pad, dtype));
    if (clhip_conv64_supported(N, H, W, K, C, ksize, stride, pad, dtype))
        return clhip_conv64_launch_ex(dz, w_dg, dx, nullptr, 1, N, H, W, accumulate, 1, z_prod, y_prod, mean, invstd, acc, replicas, nullptr, nullptr,
                                      static_cast<hipStream_t>(stream));
    if (clhip_conv16_supported(H, W, K, C, ksize, stride, pad, dtype))      // 16 -> 16 / 32 -> 32 channels: the register-resident kernels' epilogue
        return clhip_conv16_launch_bn(dz, w_dg, dx, nullptr, nullptr, 1, N, H, W, C, accumulate, 1, z_prod, y_prod, mean, invstd, acc, replicas,
                                      static_cast<hipStream_t>(stream));
    if (clhip_conv8_supported(N, H, W, K, C, ksize, stride, pad, dtype))
        return clhip_conv8_launch(dz, w_dg, dx, nullptr, 1, N, H, W, accumulate, 1, nullptr, z_prod, y_prod, mask_prod, gamma_prod, beta_prod, mean, invstd, acc, replicas,
                                  static_cast<hipStream_t>(stream));
    if (clhip_conv9_supported(N, H, W, K, C, ksize, stride, pad, dtype))
        return clhip_conv9_launch(dz, w_dg, dx, nullptr, 1, N, H, W, C, accumulate, 1, nullptr, z_prod, y_prod, mean, invstd, acc, replicas, static_cast<hipStream_t>(stream));
    return clhip_conv4_launch_bn(dz, w_dg, dx, nullptr, nullptr, 1, N, H, W, K, C, accumulate, 1, z_prod, y_prod, mean, invstd, acc, replicas,
                                 static_cast<hipStream_t>(stream));
}

bool clhip_bwd_fused_supported(int N, int H, int W, int C, int Creal, int K, int ksize, int stride, int pad, int dtype);      // conv3.hip
int clhip_bwd_fused_launch(const void* x, const void* dz, const void* w_dg, void* dx, int accumulate, float* dw, float* ws, int N, int H, int W, int C,
                           const void* bn_z, const void* bn_y, const float* bn_mean, const float* bn_invstd, double* bn_acc, int bn_rep, const float* x_coef,
                           const clhip_bn_grad* lz, hipStream_t st);
int clhip_conv16_launch_ex(const void* src, const void* wt, void* dst, float* stats, double* stat_acc, int stat_rep, int N, int H, int W, int C, int accumulate, int mode,
                           const void* bn_z, const void* bn_y, const float* bn_mean, const float* bn_invstd, double* bn_acc, int bn_rep, const float* bn_coef,
                           const clhip_bn_input* in, hipStream_t st, const clhip_bn_res_input* rs = nullptr);

// ---- "lazy" BatchNorm input: the consumer applies relu(bn(z)) of its producer while it stages its operand (conv3.hip conv16 / conv32: the
//      kernels that stage through registers); the producer's apply launch and activation tensor do not exist
extern "C" int clhip_conv_bn_input_supported(int N, int H, int W, int C, int K, int ksize, int stride, int pad, int dtype) {
    if (check_conv(N, H, W, C, K, ksize, stride, pad) != CLHIP_OK) return 0;
    if (use_v1() || !use_v3()) return 0;
    const char* cfg = clhip_cfg("BN_INPUT");
    const bool off = cfg != nullptr && atoi(cfg) == 0;
    return (!off && (clhip_conv16_supported(H, W, C, K, ksize, stride, pad, dtype) || (conv64_fwd_on() && clhip_conv64_supported(N, H, W, C, K, ksize, stride, pad, dtype))) &&
            clhip_bwd_fused_supported(N, H, W, C, C, K, ksize, stride, pad, dtype)) ? 1 : 0;
}

static int conv_fwd_acc_bn_input(const void* z_in, const clhip_bn_input* bn, const clhip_bn_res_input* rs, const void* w_fwd, void* z, double* stat_acc,
                                 int replicas, int N, int H, int W, int C, int K, int ksize, int stride, int pad, int dtype, void* stream);
extern "C" int clhip_conv_fwd_acc_bn_input(const void* z_in, const clhip_bn_input* bn, const void* w_fwd, void* z, double* stat_acc, int replicas, int N,
                                           int H, int W, int C, int K, int ksize, int stride, int pad, int dtype, void* stream) {
    return conv_fwd_acc_bn_input(z_in, bn, nullptr, w_fwd, z, stat_acc, replicas, N, H, W, C, K, ksize, stride, pad, dtype, stream);
}
// ... whose producer is a conv -> BN -> +res -> ReLU layer: the launch also writes the producer's activation and packed ReLU mask
extern "C" int clhip_conv_fwd_acc_bn_res_input(const void* z_in, const clhip_bn_input* bn, const clhip_bn_res_input* rs, const void* w_fwd, void* z,
                                               double* stat_acc, int replicas, int N, int H, int W, int C, int K, int ksize, int stride, int pad, int dtype,
                                               void* stream) {
    CLHIP_CHECK_ARG(rs && rs->res && rs->y && (rs->relu_mask || (bn && bn->stat_acc == nullptr)));      // (eval-mode producers keep no mask)
    return conv_fwd_acc_bn_input(z_in, bn, rs, w_fwd, z, stat_acc, replicas, N, H, W, C, K, ksize, stride, pad, dtype, stream);
}
static int conv_fwd_acc_bn_input(const void* z_in, const clhip_bn_input* bn, const clhip_bn_res_input* rs, const void* w_fwd, void* z, double* stat_acc,
                                 int replicas, int N, int H, int W, int C, int K, int ksize, int stride, int pad, int dtype, void* stream) {
    if (int e = check_conv(N, H, W, C, K, ksize, stride, pad)) return e;
    CLHIP_CHECK_ARG(z_in && bn && w_fwd && z);
    if (bn->stat_acc == nullptr) {
        // eval-mode producer: scale / shift of its RUNNING statistics (read only), no by-products; the launch keeps no statistics of its own output either
        CLHIP_CHECK_ARG(bn->gamma && bn->beta && bn->running_mean && bn->running_var && stat_acc == nullptr);
        replicas = 1;
    } else {
        CLHIP_CHECK_ARG(stat_acc && replicas >= 1 && replicas <= 64 && (replicas & (replicas - 1)) == 0);
        CLHIP_CHECK_ARG(bn->gamma && bn->beta && bn->mean && bn->invstd && bn->coef && bn->replicas >= 1 && bn->replicas <= 64);
        CLHIP_CHECK_ARG((bn->running_mean == nullptr) == (bn->running_var == nullptr));
    }
    CLHIP_CHECK_ARG(clhip_conv_bn_input_supported(N, H, W, C, K, ksize, stride, pad, dtype));
    if (C == 64)
        return clhip_conv64_launch_ex(z_in, w_fwd, z, stat_acc, replicas, N, H, W, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 1, nullptr, bn,
                                      static_cast<hipStream_t>(stream), rs);
    return clhip_conv16_launch_ex(z_in, w_fwd, z, nullptr, stat_acc, replicas, N, H, W, C, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 1, nullptr, bn,
                                  static_cast<hipStream_t>(stream), rs);
}

// ---- the same for the LDS-DMA kernels of the wide layers (conv5.hip: 64 -> 64 channels at >= 512 tiles; conv4.hip: C, K multiples of 64): the
//      transform happens IN LDS on the landed patch and the launch always writes the activation (common.h LazyIn)
int clhip_conv5_launch_in(const void* src, const void* wt, void* dst, double* stat_acc, int stat_rep, int N, int H, int W, int accumulate, int mode, const LazyIn* in,
                          hipStream_t st);
bool clhip_conv4_in_supported(int N, int H, int W, int Cs, int Cd);
int clhip_conv4_launch_in(const void* src, const void* wt, void* dst, double* stat_acc, int stat_rep, int N, int H, int W, int Cs, int Cd, const LazyIn* in, hipStream_t st);

extern "C" int clhip_conv_bn_input_wt_supported(int N, int H, int W, int C, int K, int ksize, int stride, int pad, int dtype) {
    if (check_conv(N, H, W, C, K, ksize, stride, pad) != CLHIP_OK) return 0;
    if (use_v1() || !use_v3() || dtype != CLHIP_BF16) return 0;
    // OFF by default (BN_INPUT_WT=1 enables it): measured on ResNet-18 layer1 at batch 256 the fused launch is 41.5 us against 47.2 us for the two it
    // replaces stand-alone, but inside the step the apply launches run at 10-11 us out of the Infinity Cache and the step got SLOWER (2.095 ->
    // 2.14 ms with the four layer-1 units fused; the +res form loses stand-alone as well: 59 vs 55 us) -- profiles/r04_wt_notes.md
    // (round 5: conv8.hip hides the transform under the other workgroup's MFMAs, and the step is still neutral -- 2.00-2.06 vs 2.00-2.03 ms: the layer is
    //  HBM-bound, the fused launch re-reads the halo rows of z' AND r (1.55 x each at 128-pixel tiles) and saves one read of the activation)
    const char* cfg = clhip_cfg("BN_INPUT_WT");
    if (cfg == nullptr || atoi(cfg) == 0) return 0;
    if (conv64_fwd_on() && clhip_conv64_supported(N, H, W, C, K, ksize, stride, pad, dtype)) return 0;      // (that layer runs on the register-staged kernel)
    if (clhip_conv8_supported(N, H, W, C, K, ksize, stride, pad, dtype)) return 1;
    if (clhip_conv9_supported(N, H, W, C, K, ksize, stride, pad, dtype)) return 1;
    if (clhip_conv5_supported(N, H, W, C, K, ksize, stride, pad, dtype)) return 1;
    if (clhip_conv4_supported(N, H, W, C, K, ksize, stride, pad, dtype) && clhip_conv4_in_supported(N, H, W, C, K)) return 1;
    return 0;
}

extern "C" int clhip_conv_fwd_acc_bn_input_wt(const void* z_in, const clhip_bn_input* bn, const clhip_bn_res_input* rs, const void* w_fwd, void* z, double* stat_acc,
                                              int replicas, int N, int H, int W, int C, int K, int ksize, int stride, int pad, int dtype, void* stream) {
    if (int e = check_conv(N, H, W, C, K, ksize, stride, pad)) return e;
    CLHIP_CHECK_ARG(z_in && bn && rs && rs->y && w_fwd && z && stat_acc && replicas >= 1 && replicas <= 64 && (replicas & (replicas - 1)) == 0);
    CLHIP_CHECK_ARG(bn->stat_acc && bn->gamma && bn->beta && bn->mean && bn->invstd && bn->coef && bn->replicas >= 1 && bn->replicas <= 64);
    CLHIP_CHECK_ARG((bn->running_mean == nullptr) == (bn->running_var == nullptr));
    CLHIP_CHECK_ARG(rs->relu_mask == nullptr || rs->res != nullptr);
    CLHIP_CHECK_ARG(clhip_conv_bn_input_wt_supported(N, H, W, C, K, ksize, stride, pad, dtype));
    LazyIn in;
    in.acc = bn->stat_acc; in.rep = bn->replicas; in.gamma = bn->gamma; in.beta = bn->beta; in.rm = bn->running_mean; in.rv = bn->running_var;
    in.momentum = bn->momentum; in.eps = bn->eps; in.mean_o = bn->mean; in.invstd_o = bn->invstd; in.coef_o = bn->coef;
    const double M = (double)N * H * W;
    in.invM = 1.0 / M; in.unbias = M > 1.0 ? M / (M - 1.0) : 1.0;
    in.res = static_cast<const bf16_t*>(rs->res); in.y = static_cast<bf16_t*>(rs->y); in.mask = static_cast<unsigned char*>(rs->relu_mask);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (clhip_conv8_supported(N, H, W, C, K, ksize, stride, pad, dtype))
        return clhip_conv8_launch(z_in, w_fwd, z, stat_acc, replicas, N, H, W, 0, 0, &in, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, st);
    if (clhip_conv9_supported(N, H, W, C, K, ksize, stride, pad, dtype))
        return clhip_conv9_launch(z_in, w_fwd, z, stat_acc, replicas, N, H, W, C, 0, 0, &in, nullptr, nullptr, nullptr, nullptr, nullptr, 1, st);
    if (clhip_conv5_supported(N, H, W, C, K, ksize, stride, pad, dtype)) return clhip_conv5_launch_in(z_in, w_fwd, z, stat_acc, replicas, N, H, W, 0, 0, &in, st);
    return clhip_conv4_launch_in(z_in, w_fwd, z, stat_acc, replicas, N, H, W, C, K, &in, st);
}

extern "C" int clhip_conv_dgrad_wgrad_bn_input(const void* x_z, const float* x_coef, const void* dz, const void* w_dg, void* dx, int accumulate, float* dw,
                                               void* ws, const float* mean, const float* invstd, double* acc, int replicas, int N, int H, int W, int C,
                                               int Creal, int K, int ksize, int stride, int pad, int dtype, void* stream) {
    if (int e = check_conv(N, H, W, C, K, ksize, stride, pad)) return e;
    CLHIP_CHECK_ARG(x_z && x_coef && dz && w_dg && dx && dw && ws);
    CLHIP_CHECK_ARG(clhip_conv_bn_input_supported(N, H, W, C, K, ksize, stride, pad, dtype) && Creal == C);
    CLHIP_CHECK_ARG(acc == nullptr || (mean && invstd && replicas >= 1 && replicas <= 64 && (replicas & (replicas - 1)) == 0));
    return clhip_bwd_fused_launch(x_z, dz, w_dg, dx, accumulate, dw, static_cast<float*>(ws), N, H, W, C, acc ? x_z : nullptr, nullptr, mean, invstd, acc, replicas,
                                  x_coef, nullptr, static_cast<hipStream_t>(stream));
}

extern "C" int clhip_conv_dgrad_wgrad_bn_grad(const void* x, const float* x_coef, const clhip_bn_grad* bn, const void* w_dg, void* dx, int accumulate, float* dw, void* ws,
                                              const void* z_prod, const void* y_prod, const float* mean, const float* invstd, double* acc, int replicas,
                                              int N, int H, int W, int C, int Creal, int K, int ksize, int stride, int pad, int dtype, void* stream) {
    if (int e = check_conv(N, H, W, C, K, ksize, stride, pad)) return e;
    CLHIP_CHECK_ARG(x && bn && w_dg && dx && dw && ws);
    CLHIP_CHECK_ARG(bn->dy && bn->z && bn->sums && bn->mean && bn->invstd && bn->gamma && bn->beta && bn->dgamma && bn->dbeta && bn->replicas >= 1 && bn->replicas <= 64);
    CLHIP_CHECK_ARG(clhip_conv_bn_input_supported(N, H, W, C, K, ksize, stride, pad, dtype) && Creal == C);
    CLHIP_CHECK_ARG(z_prod == nullptr || (mean && invstd && acc && replicas >= 1 && replicas <= 64 && (replicas & (replicas - 1)) == 0));
    CLHIP_CHECK_ARG(bn->relu_mask != nullptr || bn->dres == nullptr);          // a residual gradient only exists behind a masked (+res) layer
    return clhip_bwd_fused_launch(x, nullptr, w_dg, dx, accumulate, dw, static_cast<float*>(ws), N, H, W, C, x_coef != nullptr ? (z_prod ? x : nullptr) : z_prod,
                                  x_coef != nullptr ? nullptr : y_prod, mean, invstd, acc, replicas, x_coef, bn, static_cast<hipStream_t>(stream));
}

extern "C" int clhip_conv_dgrad_wgrad_supported(int N, int H, int W, int C, int Creal, int K, int ksize, int stride, int pad, int dtype) {
    if (check_conv(N, H, W, C, K, ksize, stride, pad) != CLHIP_OK) return 0;
    if (use_v1() || !use_v3()) return 0;
    return clhip_bwd_fused_supported(N, H, W, C, Creal, K, ksize, stride, pad, dtype) ? 1 : 0;
}

extern "C" int clhip_conv_dgrad_wgrad(const void* x, const void* dz, const void* w_dg, void* dx, int accumulate, float* dw, void* ws,
                                      const void* z_prod, const void* y_prod, const float* mean, const float* invstd, double* acc, int replicas,
                                      int N, int H, int W, int C, int Creal, int K, int ksize, int stride, int pad, int dtype, void* stream) {
    if (int e = check_conv(N, H, W, C, K, ksize, stride, pad)) return e;
    CLHIP_CHECK_ARG(x && dz && w_dg && dx && dw && ws);
    CLHIP_CHECK_ARG(clhip_conv_dgrad_wgrad_supported(N, H, W, C, Creal, K, ksize, stride, pad, dtype));
    CLHIP_CHECK_ARG(z_prod == nullptr || (mean && invstd && acc && replicas >= 1 && replicas <= 64 && (replicas & (replicas - 1)) == 0));
    return clhip_bwd_fused_launch(x, dz, w_dg, dx, accumulate, dw, static_cast<float*>(ws), N, H, W, C, z_prod, y_prod, mean, invstd, acc, replicas, nullptr,
                                  nullptr, static_cast<hipStream_t>(stream));
}

bool clhip_dgrad6_supported(int N, int H, int W, int C, int K, int dtype);      // conv6.hip
size_t clhip_dgrad6_packed_bytes(int C, int K);
int clhip_dgrad6_pack(const void* w_dg, const void* w_sc_dg, void* packed, int C, int K, hipStream_t st);
int clhip_dgrad6_launch(const void* dz, const void* w_packed, const void* dz_sc, void* dx, int accumulate, int N, int H, int W, int C, int K, hipStream_t st);

// conv7.hip: the same for 16 -> 32 and 32 -> 64 channels (CifarResNet-32), packed [C][10][K]
bool clhip_dgrad7_supported(int N, int H, int W, int C, int K, int dtype);
size_t clhip_dgrad7_packed_bytes(int C, int K);
int clhip_dgrad7_pack(const void* w_dg, const void* w_sc_dg, void* packed, int C, int K, hipStream_t st);
int clhip_dgrad7_launch(const void* dz, const void* w_packed, const void* dz_sc, void* dx, int accumulate, int N, int H, int W, int C, int K, hipStream_t st,
                        const void* bn_z = nullptr, const void* bn_y = nullptr, const float* bn_mean = nullptr, const float* bn_invstd = nullptr, double* bn_acc = nullptr,
                        int bn_rep = 1);
static bool small_pair(int C, int K) { return (C == 16 || C == 32) && K == 2 * C; }

extern "C" int clhip_conv_dgrad_pair_supported(int N, int H, int W, int C, int K, int dtype) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0) return 0;
    if (use_v1() || !use_v3()) return 0;
    if (small_pair(C, K)) return clhip_dgrad7_supported(N, H, W, C, K, dtype) ? 1 : 0;
    return clhip_dgrad6_supported(N, H, W, C, K, dtype) ? 1 : 0;
}

extern "C" size_t clhip_conv_dgrad_pair_packed_bytes(int C, int K) {
    if (small_pair(C, K)) return clhip_dgrad7_packed_bytes(C, K);
    return (C > 0 && K > 0 && C % 64 == 0 && K % 16 == 0) ? clhip_dgrad6_packed_bytes(C, K) : 0;
}

extern "C" int clhip_conv_dgrad_pair_pack(const void* w_dg, const void* w_sc_dg, void* packed, int C, int K, int dtype, void* stream) {
    CLHIP_CHECK_ARG(w_dg && packed && dtype == CLHIP_BF16 && clhip_conv_dgrad_pair_packed_bytes(C, K) > 0);
    if (small_pair(C, K)) return clhip_dgrad7_pack(w_dg, w_sc_dg, packed, C, K, static_cast<hipStream_t>(stream));
    return clhip_dgrad6_pack(w_dg, w_sc_dg, packed, C, K, static_cast<hipStream_t>(stream));
}

extern "C" int clhip_conv_dgrad_pair(const void* dz, const void* w_packed, const void* dz_sc, void* dx, int accumulate, int N, int H, int W, int C, int K,
                                     int dtype, void* stream) {
    CLHIP_CHECK_ARG(dz && w_packed && dx);
    CLHIP_CHECK_ARG(clhip_conv_dgrad_pair_supported(N, H, W, C, K, dtype));
    if (small_pair(C, K)) return clhip_dgrad7_launch(dz, w_packed, dz_sc, dx, accumulate, N, H, W, C, K, static_cast<hipStream_t>(stream));
    return clhip_dgrad6_launch(dz, w_packed, dz_sc, dx, accumulate, N, H, W, C, K, static_cast<hipStream_t>(stream));
}

// conv7.hip: the weight gradients of a small-channel down-sampling entry (3x3/s2 + 1x1/s2 shortcut) in one launch
bool clhip_wgrad7_supported(int N, int H, int W, int C, int K, int dtype);
size_t clhip_wgrad7_ws_bytes(int N, int C, int K, int which);
int clhip_wgrad7_launch(const void* x, const void* dz, const void* dz_sc, float* dw, float* dw_sc, float* ws3, float* ws_sc, int N, int C, hipStream_t st);
static size_t wgrad_ws_bytes_single(int N, int H, int W, int C, int Creal, int K, int ksize, int stride, int pad, int dtype);

bool clhip_fwd7_supported(int N, int H, int W, int C, int K, int dtype);
int clhip_fwd7_launch(const void* x, const void* w3, const void* wsc, void* z3, void* zsc, double* acc3, int rep3, double* accsc, int repsc, int N, int H, int W, int C,
                      hipStream_t st);

// ... and their two FORWARD convolutions with the BatchNorm statistics of both (clhip_conv_fwd_acc twice) from one pass over x
extern "C" int clhip_conv_fwd_acc_pair_supported(int N, int H, int W, int C, int K, int dtype) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || use_v1() || !use_v3()) return 0;
    return clhip_fwd7_supported(N, H, W, C, K, dtype) ? 1 : 0;
}

extern "C" int clhip_conv_fwd_acc_pair(const void* x, const void* w_fwd, const void* w_sc_fwd, void* z, void* z_sc, double* stat_acc, int replicas, double* stat_acc_sc,
                                       int replicas_sc, int N, int H, int W, int C, int K, int dtype, void* stream) {
    CLHIP_CHECK_ARG(x && w_fwd && w_sc_fwd && z && z_sc && stat_acc && stat_acc_sc);
    CLHIP_CHECK_ARG(replicas >= 1 && replicas <= 64 && (replicas & (replicas - 1)) == 0 && replicas_sc >= 1 && replicas_sc <= 64 && (replicas_sc & (replicas_sc - 1)) == 0);
    CLHIP_CHECK_ARG(clhip_conv_fwd_acc_pair_supported(N, H, W, C, K, dtype));
    return clhip_fwd7_launch(x, w_fwd, w_sc_fwd, z, z_sc, stat_acc, replicas, stat_acc_sc, replicas_sc, N, H, W, C, static_cast<hipStream_t>(stream));
}

extern "C" int clhip_conv_wgrad_pair_supported(int N, int H, int W, int C, int K, int dtype) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || use_v1() || !use_v3()) return 0;
    return clhip_wgrad7_supported(N, H, W, C, K, dtype) ? 1 : 0;
}

extern "C" int clhip_conv_wgrad_pair(const void* x, const void* dz, const void* dz_sc, float* dw, float* dw_sc, void* ws, void* ws_sc, int N, int H, int W, int C, int K,
                                     int dtype, void* stream) {
    CLHIP_CHECK_ARG(x && dz && dz_sc && dw && dw_sc && ws && ws_sc);
    CLHIP_CHECK_ARG(clhip_conv_wgrad_pair_supported(N, H, W, C, K, dtype));
    return clhip_wgrad7_launch(x, dz, dz_sc, dw, dw_sc, static_cast<float*>(ws), static_cast<float*>(ws_sc), N, C, static_cast<hipStream_t>(stream));
}

// (the scratch of a layer that can be half of such a pair is large enough for either form)
// ... with the BatchNorm-backward sums of the layer that produced the block input in the epilogue (clhip_conv_dgrad_bn_reduce's contract; the
// small-channel kernel only)
extern "C" int clhip_conv_dgrad_pair_bn_reduce_supported(int N, int H, int W, int C, int K, int dtype) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || use_v1() || !use_v3()) return 0;
    return (small_pair(C, K) && clhip_dgrad7_supported(N, H, W, C, K, dtype)) ? 1 : 0;
}

extern "C" int clhip_conv_dgrad_pair_bn_reduce(const void* dz, const void* w_packed, const void* dz_sc, void* dx, int accumulate, const void* z_prod, const void* y_prod,
                                               const float* mean, const float* invstd, double* acc, int replicas, int N, int H, int W, int C, int K, int dtype,
                                               void* stream) {
    CLHIP_CHECK_ARG(dz && w_packed && dx && z_prod && mean && invstd && acc);
    CLHIP_CHECK_ARG(replicas >= 1 && replicas <= 64 && (replicas & (replicas - 1)) == 0);
    CLHIP_CHECK_ARG(clhip_conv_dgrad_pair_bn_reduce_supported(N, H, W, C, K, dtype));
    return clhip_dgrad7_launch(dz, w_packed, dz_sc, dx, accumulate, N, H, W, C, K, static_cast<hipStream_t>(stream), z_prod, y_prod, mean, invstd, acc, replicas);
}

extern "C" size_t clhip_conv_wgrad_ws_bytes(int N, int H, int W, int C, int Creal, int K, int ksize, int stride, int pad, int dtype) {
    size_t b = wgrad_ws_bytes_single(N, H, W, C, Creal, K, ksize, stride, pad, dtype);
    if (!use_v1() && use_v3() && Creal == C && stride == 2 && clhip_wgrad7_supported(N, H, W, C, K, dtype)) {
        if (ksize == 3 && pad == 1) b = std::max(b, clhip_wgrad7_ws_bytes(N, C, K, 0));
        if (ksize == 1 && pad == 0) b = std::max(b, clhip_wgrad7_ws_bytes(N, C, K, 1));
    }
    return b;
}

static size_t wgrad_ws_bytes_single(int N, int H, int W, int C, int Creal, int K, int ksize, int stride, int pad, int dtype) {
    if (!use_v1() && use_v3() && clhip_stem_wgrad_supported(N, H, W, C, Creal, K, ksize, stride, pad, dtype)) return clhip_stem_wgrad_ws_bytes(N, H, W, Creal, K);
    if (!use_v1() && use_v3() && clhip_wgrad64_supported(N, H, W, C, Creal, K, ksize, stride, pad, dtype)) return clhip_wgrad64_ws_bytes(N);
    if (!use_v1() && use_v3() && clhip_wgrad4_supported(N, H, W, C, Creal, K, ksize, stride, pad, dtype)) return clhip_wgrad4_ws_bytes(N, H, W, C, K, ksize, stride);
    if (!use_v1() && use_v3() && clhip_wgrad3_supported(N, H, W, C, Creal, K, ksize, stride, pad, dtype)) return clhip_wgrad3_ws_bytes(N, H, W, C, K);
    if (!use_v1() && use_v3() && clhip_wgrad16_supported(N, H, W, C, Creal, K, ksize, stride, pad, dtype)) return clhip_wgrad16_ws_bytes(N);
    if (!use_v1() && use_v3() && clhip_wgrad32_supported(N, H, W, C, Creal, K, ksize, stride, pad, dtype)) return clhip_wgrad32_ws_bytes(N);
    if (!use_v1()) return clhip_wgrad2_ws_bytes(N, H, W, C, Creal, K, ksize, stride, pad);      // the generic kernel's deterministic form
    return 0;
}

extern "C" int clhip_conv_wgrad(const void* x, const void* dz, float* dw, void* ws, int N, int H, int W, int C, int Creal, int K,
                                int ksize, int stride, int pad, int dtype, void* stream) {
    if (int e = check_conv(N, H, W, C, K, ksize, stride, pad)) return e;
    CLHIP_CHECK_ARG(x && dz && dw && Creal >= 1 && Creal <= C);
    WgradParams p;
    p.x = x; p.dz = dz; p.dw = dw;
    p.N = N; p.H = H; p.W = W; p.C = C; p.log2C = ilog2_exact(C); p.Creal = Creal;
    p.Ho = (H + 2 * pad - ksize) / stride + 1; p.Wo = (W + 2 * pad - ksize) / stride + 1;
    p.K = K; p.ksize = ksize; p.stride = stride; p.pad = pad;
    p.M = N * p.Ho * p.Wo; p.J = ksize * ksize * C;
    int gx = (p.J + 63) / 64, gy = (K + 63) / 64;
    int tiles = gx * gy;
    int max_splits = (p.M + 127) / 128;                 // at least 4 K-steps per block
    int splits = (2048 + tiles - 1) / tiles;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int pps = (p.M + splits - 1) / splits;
    pps = (pps + 31) / 32 * 32;
    splits = (p.M + pps - 1) / pps;
    p.pix_per_split = pps;
    dim3 grid(gx, gy, splits);
    hipStream_t st = static_cast<hipStream_t>(stream);
    CLHIP_CHECK_ARG(dtype == CLHIP_BF16 || dtype == CLHIP_F32);
    if (!use_v1() && use_v3() && ws != nullptr && clhip_stem_wgrad_supported(N, H, W, C, Creal, K, ksize, stride, pad, dtype))
        return clhip_stem_wgrad_launch(x, dz, dw, static_cast<float*>(ws), N, H, W, Creal, K, st);
    if (!use_v1() && use_v3() && ws != nullptr && clhip_wgrad64_supported(N, H, W, C, Creal, K, ksize, stride, pad, dtype))
        return clhip_wgrad64_launch(x, dz, dw, static_cast<float*>(ws), N, H, nullptr, st);
    if (!use_v1() && use_v3() && ws != nullptr && clhip_wgrad4_supported(N, H, W, C, Creal, K, ksize, stride, pad, dtype))
        return clhip_wgrad4_launch(x, dz, dw, static_cast<float*>(ws), N, H, W, C, K, ksize, stride, st);
    if (!use_v1() && use_v3() && clhip_wgrad3_supported(N, H, W, C, Creal, K, ksize, stride, pad, dtype))
        return clhip_wgrad3_launch(x, dz, dw, static_cast<float*>(ws), N, H, W, C, Creal, K, st);
    if (!use_v1() && use_v3() && ws != nullptr && clhip_wgrad16_supported(N, H, W, C, Creal, K, ksize, stride, pad, dtype))
        return clhip_wgrad16_launch(x, dz, dw, static_cast<float*>(ws), N, H, nullptr, st);
    if (!use_v1() && use_v3() && ws != nullptr && clhip_wgrad32_supported(N, H, W, C, Creal, K, ksize, stride, pad, dtype))
        return clhip_wgrad32_launch(x, dz, dw, static_cast<float*>(ws), N, H, nullptr, st);
    if (!use_v1()) return clhip_wgrad2_launch(x, dz, dw, static_cast<float*>(ws), N, H, W, C, Creal, K, ksize, stride, pad, dtype, st);
    static const bool no_tr = clhip_cfg("WGRAD_NO_TR") != nullptr;
    if (dtype == CLHIP_BF16) {
        if (no_tr) hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, false>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, true>), grid, dim3(256), 0, st, p);
    } else if (dtype == CLHIP_F32) {
        hipLaunchKernelGGL((conv_wgrad_kernel<float, false>), grid, dim3(256), 0, st, p);
    } else {
        CLHIP_CHECK_ARG(!"dtype");
    }
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}
