// augment.hip -- the input pipeline on the GPU: the whole uint8 image store of a dataset stays
// resident in HBM (CIFAR-100: 150 MB of 288 GB) and a batch is produced by ONE gather + augment + normalise kernel, instead of
// the reference's PIL-per-sample transforms in DataLoader workers (core/data/dataset.py:248-266, core/data/data.py:4-67), which
// cap a real epoch far below the GPU step rate.
//   crop_flip : RandomCrop(S, padding) + RandomHorizontalFlip + ColorJitter(brightness) + ToTensor + Normalize   (CIFAR ResNets)
//   rrc_flip  : RandomResizedCrop(S, box) with bilinear interpolation + RandomHorizontalFlip + ToTensor [+ Normalize]  (ViT)
//   rrc_aa    : the same on a RAGGED store (images of different sizes, ImageNet-R) with Pillow's anti-aliased filter, bit-exact:
//               a triangle filter of support max(scale, 1), 22-bit fixed-point weights, horizontal pass rounded to uint8, then
//               the vertical pass (what torchvision's RandomResizedCrop does on the PIL images of core/data/dataset.py:248-266)
// The random parameters (offsets, flips, brightness factors, crop boxes) are drawn on the host side of the boundary with the
// seeded torch generator and passed in; the kernels are pure functions of (store, index, params).
#include "common.h"

namespace {

// out[b, c, y, x] (fp32 NCHW) = normalise(clip(trunc(src * f))) where src is the zero-padded, shifted, optionally mirrored source
__global__ __launch_bounds__(256) void crop_flip_kernel(const uint8_t* __restrict__ store, const int64_t* __restrict__ index,
                                                        const int32_t* __restrict__ params /* [B,3]: dy, dx, flip */,
                                                        const float* __restrict__ bright /* [B] or null */, float* __restrict__ out, int B, int H,
                                                        int W, int S, int pad, float m0, float m1, float m2, float i0 /* std, not its reciprocal: ToTensor + Normalize divide (v / 255, (x - mean) / std), and so does this -- bit for bit */, float i1, float i2) {
    const int64_t total = (int64_t)B * S * S;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int x = (int)(t % S), y = (int)((t / S) % S), b = (int)(t / ((int64_t)S * S));
        const int dy = params[b * 3 + 0], dx = params[b * 3 + 1], flip = params[b * 3 + 2];
        const int xs = flip ? S - 1 - x : x;                  // flip acts on the cropped image
        const int sy = y + dy - pad, sx = xs + dx - pad;      // position in the unpadded source
        float r = 0.f, g = 0.f, bl = 0.f;
        if ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W) {
            const uint8_t* p = store + (((size_t)index[b] * H + sy) * W + sx) * 3;
            r = p[0]; g = p[1]; bl = p[2];
        }
        if (bright != nullptr) {
            const float f = bright[b];
            r = fminf(floorf(r * f), 255.f); g = fminf(floorf(g * f), 255.f); bl = fminf(floorf(bl * f), 255.f);
        }
        const size_t plane = (size_t)S * S;
        float* o = out + (size_t)b * 3 * plane + (size_t)y * S + x;
        o[0] = (r / 255.f - m0) / i0;
        o[plane] = (g / 255.f - m1) / i1;
        o[2 * plane] = (bl / 255.f - m2) / i2;
    }
}

// bilinear resize of the box (x0, y0, w, h) of a source image to S x S (half-pixel centres, edge clamp; the 32 -> 224 upscaling
// of the L2P / CIFAR pipeline never needs an anti-aliasing filter), then flip / scale to [0,1] / normalise
__global__ __launch_bounds__(256) void rrc_flip_kernel(const uint8_t* __restrict__ store, const int64_t* __restrict__ index,
                                                       const int32_t* __restrict__ params /* [B,5]: y0, x0, h, w, flip */, float* __restrict__ out,
                                                       int B, int H, int W, int S, float m0, float m1, float m2, float i0, float i1, float i2) {
    const int64_t total = (int64_t)B * S * S;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int x = (int)(t % S), y = (int)((t / S) % S), b = (int)(t / ((int64_t)S * S));
        const int32_t* q = params + b * 5;
        const int y0 = q[0], x0 = q[1], bh = q[2], bw = q[3], flip = q[4];
        const int xs = flip ? S - 1 - x : x;
        const float fy = ((float)y + 0.5f) * ((float)bh / (float)S) - 0.5f;
        const float fx = ((float)xs + 0.5f) * ((float)bw / (float)S) - 0.5f;
        const int iy = (int)floorf(fy), ix = (int)floorf(fx);
        const float wy = fy - (float)iy, wx = fx - (float)ix;
        const int ya = y0 + min(max(iy, 0), bh - 1), yb = y0 + min(max(iy + 1, 0), bh - 1);
        const int xa = x0 + min(max(ix, 0), bw - 1), xb = x0 + min(max(ix + 1, 0), bw - 1);
        const uint8_t* img = store + (size_t)index[b] * H * W * 3;
        const uint8_t* p00 = img + ((size_t)ya * W + xa) * 3;
        const uint8_t* p01 = img + ((size_t)ya * W + xb) * 3;
        const uint8_t* p10 = img + ((size_t)yb * W + xa) * 3;
        const uint8_t* p11 = img + ((size_t)yb * W + xb) * 3;
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float top = (float)p00[c] + wx * ((float)p01[c] - (float)p00[c]);
            const float bot = (float)p10[c] + wx * ((float)p11[c] - (float)p10[c]);
            v[c] = floorf(top + wy * (bot - top) + 0.5f);          // back to the uint8 grid like PIL, then ToTensor
        }
        const size_t plane = (size_t)S * S;
        float* o = out + (size_t)b * 3 * plane + (size_t)y * S + x;
        o[0] = (v[0] / 255.f - m0) / i0;
        o[plane] = (v[1] / 255.f - m1) / i1;
        o[2 * plane] = (v[2] / 255.f - m2) / i2;
    }
}

// ---- Pillow's resampling coefficients (Resample.c precompute_coeffs + normalize_coeffs_8bpc), one thread per (sample, axis,
// output position): tab[((b*2 + axis)*S + o)*(2+T)] = {first tap, tap count, T fixed-point weights}; axis 0 = rows (box height),
// axis 1 = columns (box width).  Double precision without contraction, the operation order of the C source, so that the
// integers are Pillow's.
constexpr int AA_BITS = 22;

__global__ __launch_bounds__(256) void aa_coef_kernel(const int32_t* __restrict__ params, int32_t* __restrict__ tab, int B, int S, int T) {
#pragma clang fp contract(off)
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= B * 2 * S) return;
    const int o = t % S, axis = (t / S) % 2, b = t / (2 * S);
    const int in = params[b * 5 + 2 + axis];
    int32_t* row = tab + (size_t)t * (2 + T);
    const double scale = (double)in / (double)S;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const double ss = 1.0 / filterscale;
    const double center = 0.0 + ((double)o + 0.5) * scale;
    int lo = (int)(center - support + 0.5);
    if (lo < 0) lo = 0;
    int hi = (int)(center + support + 0.5);
    if (hi > in) hi = in;
    int n = hi - lo;
    if (n > T) n = T;                                   // cannot happen when T = 2*ceil(support)+1 of the largest box; keeps the table in bounds
    double ww = 0.0;
    for (int x = 0; x < n; ++x) {
        double v = ((double)(x + lo) - center + 0.5) * ss;
        if (v < 0.0) v = -v;
        ww += v < 1.0 ? 1.0 - v : 0.0;
    }
    row[0] = lo;
    row[1] = n;
    for (int x = 0; x < T; ++x) {
        int k = 0;
        if (x < n) {
            double v = ((double)(x + lo) - center + 0.5) * ss;
            if (v < 0.0) v = -v;
            double w = v < 1.0 ? 1.0 - v : 0.0;
            if (ww != 0.0) w /= ww;
            k = w < 0.0 ? (int)(-0.5 + w * (double)(1 << AA_BITS)) : (int)(0.5 + w * (double)(1 << AA_BITS));
        }
        row[2 + x] = k;
    }
}

__device__ inline int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// out[b, c, y, x]: both passes of the resize for one output pixel -- for every row tap the horizontal filter, rounded to uint8
// as Pillow's intermediate image is, then the vertical filter over those
__global__ __launch_bounds__(256) void rrc_aa_kernel(const uint8_t* __restrict__ store, const int64_t* __restrict__ offsets, const int32_t* __restrict__ hw,
                                                     const int64_t* __restrict__ index, const int32_t* __restrict__ params /* [B,5]: y0, x0, h, w, flip */,
                                                     const int32_t* __restrict__ tab, float* __restrict__ out, int B, int H, int W, int S, int T, float m0,
                                                     float m1, float m2, float i0, float i1, float i2) {
    const int64_t total = (int64_t)B * S * S;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int x = (int)(t % S), y = (int)((t / S) % S), b = (int)(t / ((int64_t)S * S));
        const int32_t* q = params + b * 5;
        const int y0 = q[0], x0 = q[1], flip = q[4];
        const int xs = flip ? S - 1 - x : x;
        const int64_t im = index[b];
        const uint8_t* img = offsets != nullptr ? store + offsets[im] : store + (size_t)im * H * W * 3;
        const int Wi = hw != nullptr ? hw[im * 2 + 1] : W;
        const int32_t* ty = tab + ((size_t)(b * 2 + 0) * S + y) * (2 + T);
        const int32_t* tx = tab + ((size_t)(b * 2 + 1) * S + xs) * (2 + T);
        const int ylo = ty[0], yn = ty[1], xlo = tx[0], xn = tx[1];
        int v0 = 1 << (AA_BITS - 1), v1 = v0, v2 = v0;
        for (int j = 0; j < yn; ++j) {
            const uint8_t* p = img + ((size_t)(y0 + ylo + j) * Wi + x0 + xlo) * 3;
            int h0 = 1 << (AA_BITS - 1), h1 = h0, h2 = h0;
            for (int i = 0; i < xn; ++i) {
                const int k = tx[2 + i];
                h0 += (int)p[3 * i] * k; h1 += (int)p[3 * i + 1] * k; h2 += (int)p[3 * i + 2] * k;
            }
            const int k = ty[2 + j];
            v0 += clip8(h0 >> AA_BITS) * k; v1 += clip8(h1 >> AA_BITS) * k; v2 += clip8(h2 >> AA_BITS) * k;
        }
        const size_t plane = (size_t)S * S;
        float* o = out + (size_t)b * 3 * plane + (size_t)y * S + x;
        o[0] = ((float)clip8(v0 >> AA_BITS) / 255.f - m0) / i0;
        o[plane] = ((float)clip8(v1 >> AA_BITS) / 255.f - m1) / i1;
        o[2 * plane] = ((float)clip8(v2 >> AA_BITS) / 255.f - m2) / i2;
    }
}

inline int aa_taps(int max_box, int S) {               // Resample.c: ksize = ceil(support) * 2 + 1
    const double scale = (double)max_box / S;
    return (int)ceil(scale < 1.0 ? 1.0 : scale) * 2 + 1;
}

inline int blocks_for(int64_t n) { int64_t b = (n + 255) / 256; return (int)(b < 65536 ? b : 65536); }

}  // namespace

extern "C" int clhip_augment_crop_flip(const uint8_t* store, const int64_t* index, const int32_t* params, const float* brightness, float* out, int B,
                                       int H, int W, int S, int pad, const float* mean3, const float* std3, void* stream) {
    CLHIP_CHECK_ARG(store && index && params && out && mean3 && std3 && B > 0 && H > 0 && W > 0 && S > 0 && pad >= 0);
    CLHIP_CHECK_ARG(S <= H + 2 * pad && S <= W + 2 * pad);
    hipLaunchKernelGGL(crop_flip_kernel, dim3(blocks_for((int64_t)B * S * S)), dim3(256), 0, static_cast<hipStream_t>(stream), store, index, params,
                       brightness, out, B, H, W, S, pad, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_augment_rrc_flip(const uint8_t* store, const int64_t* index, const int32_t* params, float* out, int B, int H, int W, int S,
                                      const float* mean3, const float* std3, void* stream) {
    CLHIP_CHECK_ARG(store && index && params && out && mean3 && std3 && B > 0 && H > 0 && W > 0 && S > 0);
    hipLaunchKernelGGL(rrc_flip_kernel, dim3(blocks_for((int64_t)B * S * S)), dim3(256), 0, static_cast<hipStream_t>(stream), store, index, params, out,
                       B, H, W, S, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" size_t clhip_augment_rrc_aa_ws_bytes(int B, int S, int max_box) {
    if (B <= 0 || S <= 0 || max_box <= 0) return 0;
    return (size_t)B * 2 * S * (2 + aa_taps(max_box, S)) * sizeof(int32_t);
}

extern "C" int clhip_augment_rrc_aa(const uint8_t* store, const int64_t* offsets, const int32_t* hw, const int64_t* index, const int32_t* params,
                                    float* out, void* ws, int B, int H, int W, int S, int max_box, const float* mean3, const float* std3,
                                    void* stream) {
    CLHIP_CHECK_ARG(store && index && params && out && ws && mean3 && std3 && B > 0 && S > 0 && max_box > 0);
    CLHIP_CHECK_ARG((offsets != nullptr) == (hw != nullptr) && (offsets != nullptr || (H > 0 && W > 0)));
    const int T = aa_taps(max_box, S);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(aa_coef_kernel, dim3((B * 2 * S + 255) / 256), dim3(256), 0, st, params, static_cast<int32_t*>(ws), B, S, T);
    CLHIP_LAUNCH_CHECK();
    hipLaunchKernelGGL(rrc_aa_kernel, dim3(blocks_for((int64_t)B * S * S)), dim3(256), 0, st, store, offsets, hw, index, params,
                       static_cast<const int32_t*>(ws), out, B, H, W, S, T, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}
