// bn.hip -- BatchNorm2d (+ residual add + ReLU) forward / backward, global average pool, layout
// conversion and conv-weight shadow preparation.  All HBM-bound streaming kernels: 16-byte (bf16x8)
// or 32-byte (fp32x8) accesses per lane, coalesced along the NHWC channel axis, wavefront (64-lane)
// reductions, per-block partials instead of atomics (deterministic).
//
// Reference semantics: nn.BatchNorm2d(momentum=0.1, eps=1e-5) + F.relu + residual add,
// core/model/backbone/resnet.py:296-316; nn.AvgPool2d(8) / AdaptiveAvgPool2d(1), :160, :344.
#include <stdlib.h>

#include <hip/hip_ext.h>

#include "common.h"

namespace {

// ------------------------------------------------------------------------------- stats finalize
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int tiles, double invM, double unbias,
                                                          int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* rm, float* rv, float momentum, float eps, float* mean_o,
                                                          float* invstd_o, float* scale, float* shift) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + wave;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int t = lane; t < tiles; t += 64) {
        s1 += (double)part[((size_t)t * 2 + 0) * C + c];
        s2 += (double)part[((size_t)t * 2 + 1) * C + c];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if (lane == 0) {
        double mean = s1 * invM;
        double var = s2 * invM - mean * mean;
        if (var < 0.0) var = 0.0;
        float istd = (float)(1.0 / sqrt(var + (double)eps));
        float g = gamma[c], b = beta[c];
        float sc = g * istd;
        mean_o[c] = (float)mean;
        invstd_o[c] = istd;
        scale[c] = sc;
        shift[c] = b - (float)mean * sc;
        if (rm != nullptr) {
            rm[c] = (1.f - momentum) * rm[c] + momentum * (float)mean;
            rv[c] = (1.f - momentum) * rv[c] + momentum * (float)(var * unbias);
        }
    }
}

__global__ void bn_eval_affine_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int C,
                                      float* scale, float* shift) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float istd = 1.f / sqrtf(rv[c] + eps);
    float sc = gamma[c] * istd;
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
}

// ---------------------------------------------------------------------------------------- apply
template <typename T, bool RES, bool RELU>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ z, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const T* __restrict__ res,
                                                       T* __restrict__ y, int64_t nchunks, int C) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += stride) {
        const int c0 = (int)((i * 8) % C);
        float v[8], r[8];
        load8<T>(z + i * 8, v);
        if (RES) load8<T>(res + i * 8, r);
        const float4 s0 = *reinterpret_cast<const float4*>(scale + c0), s1 = *reinterpret_cast<const float4*>(scale + c0 + 4);
        const float4 h0 = *reinterpret_cast<const float4*>(shift + c0), h1 = *reinterpret_cast<const float4*>(shift + c0 + 4);
        const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float o = fmaf(v[e], sc[e], sh[e]);
            if (RES) o += r[e];
            if (RELU) o = fmaxf(o, 0.f);
            v[e] = o;
        }
        store8<T>(y + i * 8, v);
    }
}

// the eval-mode form in ONE launch (round 4): scale / shift of the running statistics are derived per workgroup into LDS -- bn_eval_affine_kernel's
// expressions, so the values equal the two-launch form bit for bit -- and applied; an eval forward of CifarResNet-32 was 33 affine launches of
// 4.7 us (25 % of its kernel time, profiles/r04_bench_kernel_stats_herding_b50.txt) in front of 33 apply launches
template <typename T, bool RES, bool RELU>
__global__ __launch_bounds__(256) void bn_apply_eval_kernel(const T* __restrict__ z, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ rm, const float* __restrict__ rv, float eps, const T* __restrict__ res,
                                                            T* __restrict__ y, int64_t nchunks, int C) {
    extern __shared__ __attribute__((aligned(16))) float ecoefs[];      // [2][C]
    for (int c = threadIdx.x; c < C; c += 256) {
        const float istd = 1.f / sqrtf(rv[c] + eps);
        const float sc = gamma[c] * istd;
        ecoefs[c] = sc;
        ecoefs[C + c] = beta[c] - rm[c] * sc;
    }
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += stride) {
        const int c0 = (int)((i * 8) % C);
        float v[8], r[8];
        load8<T>(z + i * 8, v);
        if (RES) load8<T>(res + i * 8, r);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float o = fmaf(v[e], ecoefs[c0 + e], ecoefs[C + c0 + e]);
            if (RES) o += r[e];
            if (RELU) o = fmaxf(o, 0.f);
            v[e] = o;
        }
        store8<T>(y + i * 8, v);
    }
}

// ------------------------------------------------------------------------------------- backward
// pass 1: per-channel partial sums of g = dy*mask and g*xhat over a slab of rows.
// thread layout: cpr = C/8 chunk columns per row, rpi = 256/cpr rows per iteration.
// RELU: 0 none, 1 mask = (y > 0), 2 mask recomputed from z -- units without a residual: y = max(fmaf(z, scale, shift), 0) with the
// forward's own fp32 scale / shift expressions, so the sign test is bit-identical and the y tensor is not read at all
template <typename T, int RELU>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                            const T* __restrict__ z, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, float* __restrict__ part,
                                                            double* __restrict__ acc, int rep, int64_t M, int C,
                                                            const float* __restrict__ gamma = nullptr, const float* __restrict__ beta = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float red[];     // [256][16]
    const int cpr = C >> 3;
    const int rpi = 256 / cpr;
    const int cc = threadIdx.x % cpr, ro = threadIdx.x / cpr;
    float a1[8], a2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
    // per-channel constants through LDS once per workgroup (behind the reduction scratch): [4][C] mean, invstd, scale, shift
    float* coef = red + 256 * 16;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float m = mean[c], is_c = invstd[c];
        coef[c] = m;
        coef[C + c] = is_c;
        if (RELU == 2) { const float sc_c = gamma[c] * is_c; coef[2 * C + c] = sc_c; coef[3 * C + c] = beta[c] - m * sc_c; }
    }
    __syncthreads();
    if (ro < rpi) {
        float mu[8], is[8], sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            mu[e] = coef[cc * 8 + e]; is[e] = coef[C + cc * 8 + e];
            if (RELU == 2) { sc[e] = coef[2 * C + cc * 8 + e]; sh[e] = coef[3 * C + cc * 8 + e]; }
        }
        constexpr int UN = 2;                                  // rows per trip, every load issued before the first use (r05 A/B inside the ResNet-18 step:
                                                               // four rows 2.013 ms, two 1.984 ms on one box; one row 2.039 vs 2.028 on another -- the pass shares HBM with the weight-gradient stream)
        const int64_t rstep = (int64_t)gridDim.x * rpi;
        for (int64_t r = (int64_t)blockIdx.x * rpi + ro; r < M; r += rstep * UN) {
            float g[UN][8], yy[UN][8], zz[UN][8];
            unsigned mk[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                mk[u] = 0;
                const int64_t ru = r + u * rstep;
                if (ru < M) {
                    const int64_t off = ru * C + cc * 8;
                    load8<T>(dy + off, g[u]);
                    load8<T>(z + off, zz[u]);
                    if (RELU == 1) load8<T>(y + off, yy[u]);
                    if (RELU == 3) mk[u] = reinterpret_cast<const unsigned char*>(y)[off >> 3];
                }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                if (r + u * rstep >= M) break;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float gg = g[u][e];
                    if (RELU == 1) gg = yy[u][e] > 0.f ? gg : 0.f;
                    if (RELU == 2) gg = fmaf(zz[u][e], sc[e], sh[e]) > 0.f ? gg : 0.f;
                    if (RELU == 3) gg = (mk[u] >> e) & 1u ? gg : 0.f;
                    a1[e] += gg;
                    a2[e] += gg * (zz[u][e] - mu[e]) * is[e];
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[threadIdx.x * 16 + e] = a1[e]; red[threadIdx.x * 16 + 8 + e] = a2[e]; }
    __syncthreads();
    // channel c, which w: sum over ro.  Narrow layers have few outputs and many rows per iteration (C = 16: 32 sums of 128
    // values): all 256 threads take a slice of the rows first (a serial loop of 128 LDS reads was a third of the kernel there)
    const int nout = 2 * C;
    if (nout <= 128) {
        const int nparts = 256 / nout;                       // power of two (C is)
        const int out = threadIdx.x % nout, prt = threadIdx.x / nout;
        const int which = out / C, c = out - which * C;
        const int col = c >> 3, e = c & 7;
        float s = 0.f;
        for (int q = prt; q < rpi; q += nparts) s += red[(q * cpr + col) * 16 + which * 8 + e];
        __syncthreads();                                      // everyone has read its slice: red is reused for the partial sums
        red[prt * nout + out] = s;
        __syncthreads();
        if (threadIdx.x < nout) {
            float t = 0.f;
            for (int q = 0; q < nparts; ++q) t += red[q * nout + threadIdx.x];
            if (acc != nullptr) atomicAdd(acc + ((size_t)(blockIdx.x & (rep - 1)) * 2 + which) * C + c, (double)t);
            else part[((size_t)blockIdx.x * 2 + which) * C + c] = t;
        }
        return;
    }
    for (int idx = threadIdx.x; idx < nout; idx += 256) {
        int which = idx / C, c = idx - which * C;
        int col = c >> 3, e = c & 7;
        float s = 0.f;
        for (int q = 0; q < rpi; ++q) s += red[(q * cpr + col) * 16 + which * 8 + e];
        if (acc != nullptr) atomicAdd(acc + ((size_t)(blockIdx.x & (rep - 1)) * 2 + which) * C + c, (double)s);
        else part[((size_t)blockIdx.x * 2 + which) * C + c] = s;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part, int G, double invM, int C,
                                                              float* dgamma, float* dbeta, float* coef) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + wave;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int t = lane; t < G; t += 64) {
        s1 += (double)part[((size_t)t * 2 + 0) * C + c];
        s2 += (double)part[((size_t)t * 2 + 1) * C + c];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if (lane == 0) {
        dbeta[c] += (float)s1;
        dgamma[c] += (float)s2;
        coef[c] = (float)(s1 * invM);
        coef[C + c] = (float)(s2 * invM);
    }
}

// pass 2: dz = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)); dres (+)= g
template <typename T, bool RELU, int DRES>   // DRES: 0 none, 1 write, 2 accumulate
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                           const T* __restrict__ z, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ coef, T* __restrict__ dz, T* __restrict__ dres,
                                                           int64_t nchunks, int C) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += stride) {
        const int c0 = (int)((i * 8) % C);
        float g[8], yy[8], zz[8], o[8], rr[8];
        load8<T>(dy + i * 8, g);
        load8<T>(z + i * 8, zz);
        if (RELU) load8<T>(y + i * 8, yy);
        if (DRES == 2) load8<T>(dres + i * 8, rr);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + e;
            float gg = g[e];
            if (RELU) gg = yy[e] > 0.f ? gg : 0.f;
            const float is = invstd[c];
            const float xh = (zz[e] - mean[c]) * is;
            o[e] = gamma[c] * is * (gg - coef[c] - xh * coef[C + c]);
            g[e] = gg;
            if (DRES == 2) rr[e] += gg;
        }
        store8<T>(dz + i * 8, o);
        if (DRES == 1) store8<T>(dres + i * 8, g);
        if (DRES == 2) store8<T>(dres + i * 8, rr);
    }
}

// is the value, as it is stored in T, positive?  (bf16: a positive fp32 below half the smallest bf16 subnormal rounds to +0)
template <typename T> __device__ __forceinline__ bool stored_positive(float v);
template <> __device__ __forceinline__ bool stored_positive<float>(float v) { return v > 0.f; }
template <> __device__ __forceinline__ bool stored_positive<bf16_t>(float v) { return v > 0.f && (pack_bf16x2(v, 0.f) & 0xffffu) != 0u; }

// ---------------------------------------------------------------------------- accumulator ("acc") variants
// The per-layer finalize launches (bn_finalize, bn_bwd_finalize: 40 x ~6 us of a 3.3 ms ResNet-18 step) disappear when the
// producer adds its per-channel sums into a [2][C] fp64 accumulator with hardware fp64 atomics (order-independent to ~1e-16)
// and the consumer derives scale / shift (forward) or the two mean terms (backward) itself: a handful of flops per thread,
// hoisted out of the grid-stride loop because a thread's 8 channels are fixed when the grid stride is a multiple of C/8.
template <typename T, bool RES, bool RELU>
__global__ __launch_bounds__(256) void bn_apply_train_kernel(const T* __restrict__ z, const double* __restrict__ acc, int rep, double invM, double unbias,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, float* rm, float* rv,
                                                             float momentum, float eps, float* __restrict__ mean_o, float* __restrict__ invstd_o,
                                                             const T* __restrict__ res, T* __restrict__ y, int64_t nchunks, int C,
                                                             unsigned char* __restrict__ relu_mask = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float coefs[];       // [2][C]: scale, shift
    // block-cooperative finalize: sum the accumulator replicas, derive scale / shift once per workgroup
    __shared__ double sred[256];
    const bool narrow = 2 * C <= 128;
    if (narrow) replica_parts(acc, rep, C, sred);
    for (int c = threadIdx.x; c < C; c += 256) {
        // (workgroup 0's read-modify-writes: the reads ride with the accumulator loads instead of adding a round trip behind them)
        const bool upd = blockIdx.x == 0 && rm != nullptr;
        const float rm_old = upd ? rm[c] : 0.f, rv_old = upd ? rv[c] : 0.f;
        double s1 = 0.0, s2 = 0.0;
        if (narrow) { for (int q = 0; q < 256 / (2 * C); ++q) { s1 += sred[q * 2 * C + c]; s2 += sred[q * 2 * C + C + c]; } }
        else sum_strided2(acc + c, acc + C + c, rep, 2 * (size_t)C, s1, s2);
        const double mean = s1 * invM;
        double var = s2 * invM - mean * mean;       // the cancellation happens in fp64 ...
        if (var < 0.0) var = 0.0;
        // ... the well-conditioned rest in fp32 in the bf16 mode; the fp32 parity mode keeps the finalize kernel's fp64 expression
        float istd;
        if constexpr (sizeof(T) == 4) istd = (float)(1.0 / sqrt(var + (double)eps));
        else istd = rsqrtf((float)var + eps);
        const float sc = gamma[c] * istd;
        coefs[c] = sc;
        coefs[C + c] = beta[c] - (float)mean * sc;
        if (blockIdx.x == 0) {                 // saved statistics for the backward + running-stat update, once
            mean_o[c] = (float)mean;
            invstd_o[c] = istd;
            if (rm != nullptr) {
                rm[c] = (1.f - momentum) * rm_old + momentum * (float)mean;
                rv[c] = (1.f - momentum) * rv_old + momentum * (float)(var * unbias);
            }
        }
    }
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c0 = ((int)i0 & ((C >> 3) - 1)) * 8;                     // fixed per thread: C is a power of two and the grid stride a multiple of C/8
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = coefs[c0 + e]; sh[e] = coefs[C + c0 + e]; }
    // UN chunks per trip, every load issued before the first use: the kernels are a few dependent round trips long on the small layers
    constexpr int UN = 4;
    for (int64_t i = i0; i < nchunks; i += stride * UN) {
        float v[UN][8], r[UN][8];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t iu = i + u * stride;
            if (iu < nchunks) { load8<T>(z + iu * 8, v[u]); if (RES) load8<T>(res + iu * 8, r[u]); }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t iu = i + u * stride;
            if (iu >= nchunks) break;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float o = fmaf(v[u][e], sc[e], sh[e]);
                if (RES) o += r[u][e];
                if (RELU) o = fmaxf(o, 0.f);
                v[u][e] = o;
            }
            store8<T>(y + iu * 8, v[u]);
            if (RELU && relu_mask != nullptr) {
                // bit e = (the STORED y > 0): the backward reads one byte per 8 elements instead of the 16 (32) bytes of y
                unsigned m = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) m |= (stored_positive<T>(v[u][e]) ? 1u : 0u) << e;
                relu_mask[iu] = (unsigned char)m;
            }
        }
    }
}

template <typename T, int RELU, int DRES>
__global__ __launch_bounds__(256) void bn_bwd_apply_acc_kernel(const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ z,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               const float* __restrict__ gamma, const double* __restrict__ acc, int rep, double invM,
                                                               float* dgamma, float* dbeta, T* __restrict__ dz, T* __restrict__ dres,
                                                               int64_t nchunks, int C, const float* __restrict__ beta = nullptr) {
    // [6][C]: mean(g), mean(g * xhat), mean, invstd, scale, shift -- every per-channel constant goes through LDS once per workgroup (32
    // four-byte gathers per thread from the four parameter arrays were most of this kernel's fixed cost on the small layers)
    extern __shared__ __attribute__((aligned(16))) float coefs[];
    __shared__ double sred[256];
    const bool narrow = 2 * C <= 128;
    if (narrow) replica_parts(acc, rep, C, sred);
    for (int c = threadIdx.x; c < C; c += 256) {
        const float db_old = blockIdx.x == 0 ? dbeta[c] : 0.f, dg_old = blockIdx.x == 0 ? dgamma[c] : 0.f;      // (read with the accumulator loads, see above)
        const float m = mean[c], is_c = invstd[c], gi_c = gamma[c] * is_c;                                     // gi = the forward's scale
        const float sh_c = RELU == 2 ? beta[c] - m * gi_c : 0.f;
        double s1 = 0.0, s2 = 0.0;
        if (narrow) { for (int q = 0; q < 256 / (2 * C); ++q) { s1 += sred[q * 2 * C + c]; s2 += sred[q * 2 * C + C + c]; } }
        else sum_strided2(acc + c, acc + C + c, rep, 2 * (size_t)C, s1, s2);
        coefs[c] = (float)(s1 * invM);
        coefs[C + c] = (float)(s2 * invM);
        coefs[2 * C + c] = m;
        coefs[3 * C + c] = is_c;
        coefs[4 * C + c] = gi_c;
        coefs[5 * C + c] = sh_c;
        if (blockIdx.x == 0) { dbeta[c] = db_old + (float)s1; dgamma[c] = dg_old + (float)s2; }
    }
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c0 = ((int)i0 & ((C >> 3) - 1)) * 8;
    float k0[8], k1[8], gi[8], mu[8], is[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        k0[e] = coefs[c0 + e];
        k1[e] = coefs[C + c0 + e];
        mu[e] = coefs[2 * C + c0 + e];
        is[e] = coefs[3 * C + c0 + e];
        gi[e] = coefs[4 * C + c0 + e];
        if (RELU == 2) sh[e] = coefs[5 * C + c0 + e];
    }
    constexpr int UN = sizeof(T) == 2 ? 4 : 2;
    for (int64_t i = i0; i < nchunks; i += stride * UN) {
        float g[UN][8], yy[UN][8], zz[UN][8], rr[UN][8];
        unsigned mk[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            mk[u] = 0;
            const int64_t iu = i + u * stride;
            if (iu < nchunks) {
                load8<T>(dy + iu * 8, g[u]);
                load8<T>(z + iu * 8, zz[u]);
                if (RELU == 1) load8<T>(y + iu * 8, yy[u]);
                if (RELU == 3) mk[u] = reinterpret_cast<const unsigned char*>(y)[iu];
                if (DRES == 2) load8<T>(dres + iu * 8, rr[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t iu = i + u * stride;
            if (iu >= nchunks) break;
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float gg = g[u][e];
                if (RELU == 1) gg = yy[u][e] > 0.f ? gg : 0.f;
                if (RELU == 2) gg = fmaf(zz[u][e], gi[e], sh[e]) > 0.f ? gg : 0.f;
                if (RELU == 3) gg = (mk[u] >> e) & 1u ? gg : 0.f;
                const float xh = (zz[u][e] - mu[e]) * is[e];
                o[e] = gi[e] * (gg - k0[e] - xh * k1[e]);
                g[u][e] = gg;
                if (DRES == 2) rr[u][e] += gg;
            }
            store8<T>(dz + iu * 8, o);
            if (DRES == 1) store8<T>(dres + iu * 8, g[u]);
            if (DRES == 2) store8<T>(dres + iu * 8, rr[u]);
        }
    }
}

int bn_bwd_blocks(int64_t M, int C) {
    int cpr = C >> 3;
    int rpi = 256 / cpr;
    int64_t g = (M + rpi - 1) / rpi;
    // up to 8 iterations per block, but small tensors (batch 32, the 8x8 / 4x4 layers) keep 128+ workgroups: with 32 of them the
    // kernel sat at its latency floor on a mostly idle chip; cap so the finalize reduction stays short
    static const int forced = clhip_cfg("BN_BWD_ITERS") ? atoi(clhip_cfg("BN_BWD_ITERS")) : 0;      // ablation runs
    int64_t iters = forced > 0 ? forced : g / 128;
    if (iters < 1) iters = 1;
    if (iters > 8) iters = 8;
    g = (g + iters - 1) / iters;
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    return (int)g;
}

int ew_blocks(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (int)b;
}

// ------------------------------------------------------------------------------------ pre-activation residual sums
// ResNet_BIC blocks (core/model/backbone/resnet.py:589-617) add the shortcut to the RAW output of conv2 and normalise the
// sum in the NEXT block (`out += residual` :615, then `bn1` :602 of the following block / the final `bn` :672): the statistics
// a BatchNorm needs are those of z + r, which no conv epilogue has seen.  One pass: z <- z + r (in place) and, in training,
// the per-channel sum / sum of squares of the result into the same fp64 accumulator the conv epilogues feed.
template <typename T>
__global__ __launch_bounds__(256) void add_stats_kernel(T* __restrict__ z, const T* __restrict__ r, double* __restrict__ acc, int rep,
                                                        int64_t M, int C) {
    extern __shared__ __attribute__((aligned(16))) float red[];     // [256][16]
    const int cpr = C >> 3;
    const int rpi = 256 / cpr;
    const int cc = threadIdx.x % cpr, ro = threadIdx.x / cpr;
    float a1[8], a2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
    if (ro < rpi) {
        for (int64_t row = (int64_t)blockIdx.x * rpi + ro; row < M; row += (int64_t)gridDim.x * rpi) {
            const int64_t off = row * C + cc * 8;
            float a[8], b[8];
            load8<T>(z + off, a);
            load8<T>(r + off, b);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += b[e];
            store8<T>(z + off, a);
            if (acc != nullptr) {
                if (sizeof(T) == 2) {                                    // the statistics of what BatchNorm will read (rounded)
#pragma unroll
                    for (int e = 0; e < 8; ++e) a[e] = bf16_to_f32(f32_to_bf16(a[e]));
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) { a1[e] += a[e]; a2[e] = fmaf(a[e], a[e], a2[e]); }
            }
        }
    }
    if (acc == nullptr) return;
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[threadIdx.x * 16 + e] = a1[e]; red[threadIdx.x * 16 + 8 + e] = a2[e]; }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 2 * C; idx += 256) {
        const int which = idx / C, c = idx - which * C;
        const int col = c >> 3, e = c & 7;
        float s = 0.f;
        for (int q = 0; q < rpi; ++q) s += red[(q * cpr + col) * 16 + which * 8 + e];
        atomicAdd(acc + ((size_t)(blockIdx.x & (rep - 1)) * 2 + which) * C + c, (double)s);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void add_inplace_kernel(T* __restrict__ a, const T* __restrict__ b, int64_t nchunks) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += stride) {
        float x[8], y[8];
        load8<T>(a + i * 8, x);
        load8<T>(b + i * 8, y);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] += y[e];
        store8<T>(a + i * 8, x);
    }
}

int add_stats_blocks(int64_t M, int C) {
    const int rpi = 256 / (C >> 3);
    int64_t g = ((M + rpi - 1) / rpi + 7) / 8;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

// ------------------------------------------------------------------------------------ windowed avg pool
// nn.AvgPool2d(win) + flatten of the NCHW result (ResNet_BIC.forward, resnet.py:675-676): feat[n][(c*Ph + ph)*Pw + pw]
template <typename T>
__global__ void avgpool_win_fwd_kernel(const T* __restrict__ a, float* __restrict__ feat, int N, int H, int W, int C, int win) {
    const int Ph = H / win, Pw = W / win;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;          // ((n*Ph + ph)*Pw + pw)*C + c : coalesced along c
    if (idx >= N * Ph * Pw * C) return;
    const int c = idx % C;
    int t = idx / C;
    const int pw = t % Pw; t /= Pw;
    const int ph = t % Ph;
    const int n = t / Ph;
    float s = 0.f;
    for (int i = 0; i < win; ++i)
        for (int j = 0; j < win; ++j) s += Elem<T>::ld(a + (((size_t)n * H + ph * win + i) * W + pw * win + j) * C + c);
    feat[(size_t)n * C * Ph * Pw + ((size_t)c * Ph + ph) * Pw + pw] = s / (float)(win * win);
}

template <typename T>
__global__ void avgpool_win_bwd_kernel(const float* __restrict__ dfeat, T* __restrict__ da, int N, int H, int W, int C, int win) {
    const int Ph = H / win, Pw = W / win;
    const int64_t nchunks = (int64_t)N * H * W * C / 8;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const float inv = 1.f / (float)(win * win);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += stride) {
        const int64_t e0 = i * 8;
        const int c0 = (int)(e0 % C);
        int64_t t = e0 / C;
        const int w = (int)(t % W); t /= W;
        const int h = (int)(t % H);
        const int n = (int)(t / H);
        const int ph = h / win, pw = w / win;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
            v[e] = (ph < Ph && pw < Pw) ? dfeat[(size_t)n * C * Ph * Pw + ((size_t)(c0 + e) * Ph + ph) * Pw + pw] * inv : 0.f;
        store8<T>(da + e0, v);
    }
}

// ------------------------------------------------------------------------------------ avg pool
// one thread per (image, channel); the pixel loop issues 8 independent loads per trip (the one-load-per-trip version was a chain of HW
// dependent L2 round trips: 20 us for the 256 x 8x8 x 64 map of CifarResNet-32, profiles/r03_bench_kernel_stats_ewc_resnet32_b50_task1.txt),
// summed in pixel order, so the result does not depend on the unrolling
template <typename T>
__global__ void avgpool_fwd_kernel(const T* __restrict__ a, float* __restrict__ feat, int N, int HW, int C) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * C) return;
    int n = idx / C, c = idx - n * C;
    const T* p = a + (size_t)n * HW * C + c;
    float s = 0.f;
    int i = 0;
    for (; i + 8 <= HW; i += 8) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = Elem<T>::ld(p + (size_t)(i + q) * C);
#pragma unroll
        for (int q = 0; q < 8; ++q) s += v[q];
    }
    for (; i < HW; ++i) s += Elem<T>::ld(p + (size_t)i * C);
    feat[idx] = s / (float)HW;
}

template <typename T>
__global__ void avgpool_bwd_kernel(const float* __restrict__ dfeat, T* __restrict__ da, int N, int HW, int C) {
    const int64_t nchunks = (int64_t)N * HW * C / 8;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const float inv = 1.f / (float)HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += stride) {
        int64_t e0 = i * 8;
        int c0 = (int)(e0 % C);
        int n = (int)(e0 / ((int64_t)HW * C));
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = dfeat[(size_t)n * C + c0 + e] * inv;
        store8<T>(da + e0, v);
    }
}

// avgpool backward that also reduces the BatchNorm backward of the LAST unit (whose activation the pooling is the only reader of): sum g and
// sum g * xhat per channel, g = da masked by that unit's ReLU -- the first pass of clhip_bn_bwd_acc over a tensor this kernel has in registers.
// A thread's 8 channels are fixed (the grid stride is a multiple of C / 8); threads of a workgroup with the same channels are summed through LDS,
// then one fp64 atomic per channel and sum (centred form as in the dgrad epilogues).
template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_bn_kernel(const float* __restrict__ dfeat, T* __restrict__ da, int N, int HW, int C, const T* __restrict__ z,
                                                             const T* __restrict__ y, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             double* __restrict__ acc, int rep) {
    __shared__ float red[256][17];
    const int64_t nchunks = (int64_t)N * HW * C / 8;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const float inv = 1.f / (float)HW;
    float s[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += stride) {
        const int64_t e0 = i * 8;
        const int c0 = (int)(e0 % C);
        const int n = (int)(e0 / ((int64_t)HW * C));
        float v[8], zz[8], yy[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = dfeat[(size_t)n * C + c0 + e] * inv;
        store8<T>(da + e0, v);
        load8<T>(z + e0, zz);
        if (y != nullptr) load8<T>(y + e0, yy);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float g = (y == nullptr || yy[e] > 0.f) ? v[e] : 0.f;
            s[e] += g; s[8 + e] = fmaf(g, zz[e], s[8 + e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) red[threadIdx.x][e] = s[e];
    __syncthreads();
    const int cpp = C / 8;                                   // chunks per pixel = distinct channel groups among the workgroup's threads
    for (int c = threadIdx.x; c < C; c += 256) {
        const int grp = c >> 3, e = c & 7;
        float a1 = 0.f, a2 = 0.f;
        for (int t = grp; t < 256; t += cpp) { a1 += red[t][e]; a2 += red[t][8 + e]; }
        double* a = acc + (size_t)(blockIdx.x & (rep - 1)) * 2 * C;
        atomicAdd(a + c, (double)a1);
        atomicAdd(a + C + c, (double)(invstd[c] * (a2 - mean[c] * a1)));
    }
}

// ------------------------------------------------------------------------------ layout converts
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int N, int C, int HW, int Cpad) {
    int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= (int64_t)N * HW) return;
    int n = (int)(pix / HW), hw = (int)(pix - (int64_t)n * HW);
    for (int c0 = 0; c0 < Cpad; c0 += 8) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (c0 + e < C) ? x[((size_t)n * C + c0 + e) * HW + hw] : 0.f;
        store8<T>(y + pix * Cpad + c0, v);
    }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ y, float* __restrict__ x, int N, int C, int HW) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * C * HW) return;
    int hw = (int)(idx % HW);
    int64_t t = idx / HW;
    int c = (int)(t % C);
    int n = (int)(t / C);
    x[idx] = Elem<T>::ld(y + ((size_t)n * HW + hw) * C + c);
}

template <typename T>
__global__ void weight_prep_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd, int K, int taps,
                                   int Creal, int Cpad) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t n = (int64_t)K * taps * Cpad;
    if (idx >= n) return;
    int c = (int)(idx % Cpad);
    int64_t t = idx / Cpad;
    int tap = (int)(t % taps);
    int k = (int)(t / taps);
    float v = c < Creal ? w[((size_t)k * taps + tap) * Creal + c] : 0.f;
    Elem<T>::st(wf + idx, v);
    if (wd != nullptr) Elem<T>::st(wd + ((size_t)c * taps + tap) * K + k, v);
}

}  // namespace

// =================================================================================== C ABI
extern "C" int clhip_bn_stats_finalize(const float* part, int tiles, int64_t M, int C, const float* gamma, const float* beta,
                                       float* rm, float* rv, float momentum, float eps, float* mean, float* invstd,
                                       float* scale, float* shift, void* stream) {
    CLHIP_CHECK_ARG(part && gamma && beta && mean && invstd && scale && shift && tiles > 0 && M > 0 && C > 0);
    CLHIP_CHECK_ARG((rm == nullptr) == (rv == nullptr));
    double unbias = M > 1 ? (double)M / (double)(M - 1) : 1.0;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, part, tiles, 1.0 / (double)M,
                       unbias, C, gamma, beta, rm, rv, momentum, eps, mean, invstd, scale, shift);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_bn_eval_affine(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int C,
                                    float* scale, float* shift, void* stream) {
    CLHIP_CHECK_ARG(gamma && beta && rm && rv && scale && shift && C > 0);
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, rm, rv, eps,
                       C, scale, shift);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

template <typename T>
static int bn_apply_t(const void* z, const float* scale, const float* shift, const void* res, void* y, int64_t M, int C, int relu,
                      hipStream_t st) {
    int64_t nch = M * C / 8;
    dim3 g(ew_blocks(nch)), b(256);
    const T* zz = (const T*)z; const T* rr = (const T*)res; T* yy = (T*)y;
    if (res && relu) hipLaunchKernelGGL((bn_apply_kernel<T, true, true>), g, b, 0, st, zz, scale, shift, rr, yy, nch, C);
    else if (res) hipLaunchKernelGGL((bn_apply_kernel<T, true, false>), g, b, 0, st, zz, scale, shift, rr, yy, nch, C);
    else if (relu) hipLaunchKernelGGL((bn_apply_kernel<T, false, true>), g, b, 0, st, zz, scale, shift, rr, yy, nch, C);
    else hipLaunchKernelGGL((bn_apply_kernel<T, false, false>), g, b, 0, st, zz, scale, shift, rr, yy, nch, C);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_bn_apply(const void* z, const float* scale, const float* shift, const void* res, void* y, int64_t M, int C,
                              int relu, int dtype, void* stream) {
    CLHIP_CHECK_ARG(z && scale && shift && y && M > 0 && C >= 8 && C % 8 == 0);
    if (dtype == CLHIP_BF16) return bn_apply_t<bf16_t>(z, scale, shift, res, y, M, C, relu, (hipStream_t)stream);
    if (dtype == CLHIP_F32) return bn_apply_t<float>(z, scale, shift, res, y, M, C, relu, (hipStream_t)stream);
    CLHIP_CHECK_ARG(!"dtype");
    return CLHIP_EINVAL;
}

template <typename T>
static int bn_apply_eval_t(const void* z, const float* gamma, const float* beta, const float* rm, const float* rv, float eps, const void* res, void* y, int64_t M, int C,
                           int relu, hipStream_t st) {
    const int64_t nch = M * C / 8;
    dim3 g(ew_blocks(nch)), b(256);
    const size_t lds = 2 * (size_t)C * sizeof(float);
    const T* zz = (const T*)z; const T* rr = (const T*)res; T* yy = (T*)y;
    if (res && relu) hipLaunchKernelGGL((bn_apply_eval_kernel<T, true, true>), g, b, lds, st, zz, gamma, beta, rm, rv, eps, rr, yy, nch, C);
    else if (res) hipLaunchKernelGGL((bn_apply_eval_kernel<T, true, false>), g, b, lds, st, zz, gamma, beta, rm, rv, eps, rr, yy, nch, C);
    else if (relu) hipLaunchKernelGGL((bn_apply_eval_kernel<T, false, true>), g, b, lds, st, zz, gamma, beta, rm, rv, eps, rr, yy, nch, C);
    else hipLaunchKernelGGL((bn_apply_eval_kernel<T, false, false>), g, b, lds, st, zz, gamma, beta, rm, rv, eps, rr, yy, nch, C);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_bn_apply_eval(const void* z, const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                                   const void* res, void* y, int64_t M, int C, int relu, int dtype, void* stream) {
    CLHIP_CHECK_ARG(z && gamma && beta && running_mean && running_var && y && M > 0 && C >= 8 && C % 8 == 0 && C <= 8192);
    if (dtype == CLHIP_BF16) return bn_apply_eval_t<bf16_t>(z, gamma, beta, running_mean, running_var, eps, res, y, M, C, relu, (hipStream_t)stream);
    if (dtype == CLHIP_F32) return bn_apply_eval_t<float>(z, gamma, beta, running_mean, running_var, eps, res, y, M, C, relu, (hipStream_t)stream);
    CLHIP_CHECK_ARG(!"dtype");
    return CLHIP_EINVAL;
}

extern "C" int clhip_bn_bwd_blocks(int64_t M, int C) { return bn_bwd_blocks(M, C); }

extern "C" size_t clhip_bn_bwd_ws_floats(int64_t M, int C) { return (size_t)bn_bwd_blocks(M, C) * 2 * C + 2 * (size_t)C; }

template <typename T>
static int bn_bwd_t(const void* dy, const void* y, const void* z, const float* mean, const float* invstd, const float* gamma,
                    float* dgamma, float* dbeta, void* dz, void* dres, int dres_acc, int64_t M, int C, int relu, float* ws,
                    hipStream_t st) {
    const int G = bn_bwd_blocks(M, C);
    float* part = ws;
    float* coef = ws + (size_t)G * 2 * C;
    const T* dyy = (const T*)dy; const T* yy = (const T*)y; const T* zz = (const T*)z;
    size_t lds = (256 * 16 + 4 * (size_t)C) * sizeof(float);
    if (relu) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 1>), dim3(G), dim3(256), lds, st, dyy, yy, zz, mean, invstd, part, (double*)nullptr, 1, M, C, (const float*)nullptr, (const float*)nullptr);
    else hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 0>), dim3(G), dim3(256), lds, st, dyy, yy, zz, mean, invstd, part, (double*)nullptr, 1, M, C, (const float*)nullptr, (const float*)nullptr);
    CLHIP_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, part, G, 1.0 / (double)M, C, dgamma, dbeta, coef);
    CLHIP_LAUNCH_CHECK();
    int64_t nch = M * C / 8;
    dim3 g(ew_blocks(nch)), b(256);
    T* dzz = (T*)dz; T* dr = (T*)dres;
    int mode = dres == nullptr ? 0 : (dres_acc ? 2 : 1);
#define BWD_APPLY(R, D) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, R, D>), g, b, 0, st, dyy, yy, zz, mean, invstd, gamma, coef, dzz, dr, nch, C)
    if (relu) { if (mode == 0) BWD_APPLY(true, 0); else if (mode == 1) BWD_APPLY(true, 1); else BWD_APPLY(true, 2); }
    else { if (mode == 0) BWD_APPLY(false, 0); else if (mode == 1) BWD_APPLY(false, 1); else BWD_APPLY(false, 2); }
#undef BWD_APPLY
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_bn_bwd(const void* dy, const void* y, const void* z, const float* mean, const float* invstd,
                            const float* gamma, float* dgamma, float* dbeta, void* dz, void* dres, int dres_accumulate, int64_t M,
                            int C, int relu, float* ws, int dtype, void* stream) {
    CLHIP_CHECK_ARG(dy && z && mean && invstd && gamma && dgamma && dbeta && dz && ws && M > 0);
    CLHIP_CHECK_ARG(C >= 8 && C % 8 == 0 && C <= 2048);
    CLHIP_CHECK_ARG(!relu || y);
    if (dtype == CLHIP_BF16)
        return bn_bwd_t<bf16_t>(dy, y, z, mean, invstd, gamma, dgamma, dbeta, dz, dres, dres_accumulate, M, C, relu, ws, (hipStream_t)stream);
    if (dtype == CLHIP_F32)
        return bn_bwd_t<float>(dy, y, z, mean, invstd, gamma, dgamma, dbeta, dz, dres, dres_accumulate, M, C, relu, ws, (hipStream_t)stream);
    CLHIP_CHECK_ARG(!"dtype");
    return CLHIP_EINVAL;
}

static bool acc_ok(int C) { return C >= 8 && C <= 2048 && (C & (C - 1)) == 0; }
// fewer, longer workgroups than the plain elementwise kernels: every workgroup pays the cooperative finalize prologue
static int acc_blocks(int64_t nchunks) {
    static const int cpt = clhip_cfg("BN_ACC_CPT") ? atoi(clhip_cfg("BN_ACC_CPT")) : 8;      // chunks per thread (swept 1..32: 8)
    int64_t b = (nchunks + 256 * cpt - 1) / (256 * cpt);
    if (b < 512) { b = (nchunks + 255) / 256; if (b > 512) b = 512; }      // small layers: fill the chip first
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

// one-shot: the next accumulator-path FORWARD apply launch completes this event (plan.hip forks the shortcut branch off it: the kernel's
// own completion signal instead of a marker packet in the caller's queue)
static thread_local hipEvent_t g_bn_fwd_stop_event = nullptr;
void clhip_bn_set_fwd_stop_event(hipEvent_t ev) { g_bn_fwd_stop_event = ev; }

template <typename T>
static int bn_apply_train_t(const void* z, const double* acc, int rep, int64_t M, const float* gamma, const float* beta, float* rm, float* rv, float momentum,
                            float eps, float* mean, float* invstd, const void* res, void* y, int C, int relu, hipStream_t st, unsigned char* mask = nullptr) {
    const int64_t nch = M * C / 8;
    dim3 g(acc_blocks(nch)), b(256);
    const size_t lds = 2 * (size_t)C * sizeof(float);
    const double invM = 1.0 / (double)M, unbias = M > 1 ? (double)M / (double)(M - 1) : 1.0;
    const T* zz = (const T*)z; const T* rr = (const T*)res; T* yy = (T*)y;
    hipEvent_t stop_ev = g_bn_fwd_stop_event;
    g_bn_fwd_stop_event = nullptr;
#define APPLY_TRAIN(R, L) hipExtLaunchKernelGGL((bn_apply_train_kernel<T, R, L>), g, b, (uint32_t)lds, st, (hipEvent_t) nullptr, stop_ev, 0u, zz, acc, rep, invM, unbias, gamma, beta, rm, rv, momentum, eps, mean, invstd, rr, yy, nch, C, mask)
    if (res && relu) APPLY_TRAIN(true, true);
    else if (res) APPLY_TRAIN(true, false);
    else if (relu) APPLY_TRAIN(false, true);
    else APPLY_TRAIN(false, false);
#undef APPLY_TRAIN
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_bn_apply_train(const void* z, const double* stat_acc, int replicas, int64_t M, int C, const float* gamma, const float* beta, float* rm, float* rv,
                                    float momentum, float eps, float* mean, float* invstd, const void* res, void* y, int relu, int dtype,
                                    void* stream) {
    CLHIP_CHECK_ARG(z && stat_acc && gamma && beta && mean && invstd && y && M > 0 && acc_ok(C));
    CLHIP_CHECK_ARG((rm == nullptr) == (rv == nullptr) && replicas >= 1 && replicas <= 64);
    if (dtype == CLHIP_BF16) return bn_apply_train_t<bf16_t>(z, stat_acc, replicas, M, gamma, beta, rm, rv, momentum, eps, mean, invstd, res, y, C, relu, (hipStream_t)stream);
    if (dtype == CLHIP_F32) return bn_apply_train_t<float>(z, stat_acc, replicas, M, gamma, beta, rm, rv, momentum, eps, mean, invstd, res, y, C, relu, (hipStream_t)stream);
    CLHIP_CHECK_ARG(!"dtype");
    return CLHIP_EINVAL;
}

extern "C" int clhip_bn_apply_train_mask(const void* z, const double* stat_acc, int replicas, int64_t M, int C, const float* gamma, const float* beta, float* rm,
                                         float* rv, float momentum, float eps, float* mean, float* invstd, const void* res, void* y, void* relu_mask,
                                         int dtype, void* stream) {
    CLHIP_CHECK_ARG(z && stat_acc && gamma && beta && mean && invstd && y && relu_mask && M > 0 && acc_ok(C));
    CLHIP_CHECK_ARG((rm == nullptr) == (rv == nullptr) && replicas >= 1 && replicas <= 64);
    unsigned char* mk = static_cast<unsigned char*>(relu_mask);
    if (dtype == CLHIP_BF16) return bn_apply_train_t<bf16_t>(z, stat_acc, replicas, M, gamma, beta, rm, rv, momentum, eps, mean, invstd, res, y, C, 1, (hipStream_t)stream, mk);
    if (dtype == CLHIP_F32) return bn_apply_train_t<float>(z, stat_acc, replicas, M, gamma, beta, rm, rv, momentum, eps, mean, invstd, res, y, C, 1, (hipStream_t)stream, mk);
    CLHIP_CHECK_ARG(!"dtype");
    return CLHIP_EINVAL;
}

// one-shot: the next accumulator-path backward apply launch completes this event (taken by the launch; see plan.hip)
static thread_local hipEvent_t g_bn_stop_event = nullptr;
void clhip_bn_set_stop_event(hipEvent_t ev) { g_bn_stop_event = ev; }
hipEvent_t clhip_bn_pending_stop_event() { return g_bn_stop_event; }

template <typename T>
static int bn_bwd_acc_t(const void* dy, const void* y, const void* z, const float* mean, const float* invstd, const float* gamma, float* dgamma,
                        float* dbeta, void* dz, void* dres, int dres_acc, int64_t M, int C, int relu, double* acc, int rep, hipStream_t st,
                        const float* beta = nullptr, bool sums_ready = false) {        // relu == 2: the mask comes from z, gamma and beta (see bn_bwd_reduce_kernel)
    const int G = bn_bwd_blocks(M, C);
    const T* dyy = (const T*)dy; const T* yy = (const T*)y; const T* zz = (const T*)z;
    size_t lds = (256 * 16 + 4 * (size_t)C) * sizeof(float);
    // the one-shot event is taken on entry: if the reduce launch below fails, it is dropped with the call instead of staying armed for an
    // unrelated later call
    hipEvent_t stop_ev = g_bn_stop_event;
    g_bn_stop_event = nullptr;
    if (sums_ready) { /* the two channel sums were accumulated by the producer of dy (clhip_conv_dgrad_bn_reduce): apply pass only */ }
    else if (relu == 2) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 2>), dim3(G), dim3(256), lds, st, dyy, yy, zz, mean, invstd, (float*)nullptr, acc, rep, M, C, gamma, beta);
    else if (relu == 3) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 3>), dim3(G), dim3(256), lds, st, dyy, yy, zz, mean, invstd, (float*)nullptr, acc, rep, M, C, (const float*)nullptr, (const float*)nullptr);
    else if (relu) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 1>), dim3(G), dim3(256), lds, st, dyy, yy, zz, mean, invstd, (float*)nullptr, acc, rep, M, C, (const float*)nullptr, (const float*)nullptr);
    else hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 0>), dim3(G), dim3(256), lds, st, dyy, yy, zz, mean, invstd, (float*)nullptr, acc, rep, M, C, (const float*)nullptr, (const float*)nullptr);
    CLHIP_LAUNCH_CHECK();
    const int64_t nch = M * C / 8;
    dim3 g(acc_blocks(nch)), b(256);
    const size_t lds2 = 6 * (size_t)C * sizeof(float);
    T* dzz = (T*)dz; T* dr = (T*)dres;
    const double invM = 1.0 / (double)M;
    int mode = dres == nullptr ? 0 : (dres_acc ? 2 : 1);
    // (plan.hip: the kernel's own completion signal is the event the weight-gradient stream waits for -- a separate hipEventRecord is a
    // marker packet in the caller's queue, ~5 us of bubble in front of the dgrad that follows)
#define BWD_ACC(R, D) hipExtLaunchKernelGGL((bn_bwd_apply_acc_kernel<T, R, D>), g, b, (uint32_t)lds2, st, (hipEvent_t) nullptr, stop_ev, 0u, dyy, yy, zz, mean, invstd, gamma, (const double*)acc, rep, invM, dgamma, dbeta, dzz, dr, nch, C, beta)
    if (relu == 2) { if (mode == 0) BWD_ACC(2, 0); else if (mode == 1) BWD_ACC(2, 1); else BWD_ACC(2, 2); }
    else if (relu == 3) { if (mode == 0) BWD_ACC(3, 0); else if (mode == 1) BWD_ACC(3, 1); else BWD_ACC(3, 2); }
    else if (relu) { if (mode == 0) BWD_ACC(1, 0); else if (mode == 1) BWD_ACC(1, 1); else BWD_ACC(1, 2); }
    else { if (mode == 0) BWD_ACC(0, 0); else if (mode == 1) BWD_ACC(0, 1); else BWD_ACC(0, 2); }
#undef BWD_ACC
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_bn_bwd_acc(const void* dy, const void* y, const void* z, const float* mean, const float* invstd, const float* gamma,
                                float* dgamma, float* dbeta, void* dz, void* dres, int dres_accumulate, int64_t M, int C, int relu,
                                double* acc, int replicas, int dtype, void* stream) {
    CLHIP_CHECK_ARG(dy && z && mean && invstd && gamma && dgamma && dbeta && dz && acc && M > 0 && acc_ok(C));
    CLHIP_CHECK_ARG(replicas >= 1 && replicas <= 64 && (replicas & (replicas - 1)) == 0);
    CLHIP_CHECK_ARG(!relu || y);
    if (dtype == CLHIP_BF16)
        return bn_bwd_acc_t<bf16_t>(dy, y, z, mean, invstd, gamma, dgamma, dbeta, dz, dres, dres_accumulate, M, C, relu, acc, replicas, (hipStream_t)stream);
    if (dtype == CLHIP_F32)
        return bn_bwd_acc_t<float>(dy, y, z, mean, invstd, gamma, dgamma, dbeta, dz, dres, dres_accumulate, M, C, relu, acc, replicas, (hipStream_t)stream);
    CLHIP_CHECK_ARG(!"dtype");
    return CLHIP_EINVAL;
}

extern "C" int clhip_bn_bwd_acc_zmask(const void* dy, const void* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                      float* dgamma, float* dbeta, void* dz, int64_t M, int C, double* acc, int replicas, int dtype, void* stream) {
    CLHIP_CHECK_ARG(dy && z && mean && invstd && gamma && beta && dgamma && dbeta && dz && acc && M > 0 && acc_ok(C));
    CLHIP_CHECK_ARG(replicas >= 1 && replicas <= 64 && (replicas & (replicas - 1)) == 0);
    if (dtype == CLHIP_BF16)
        return bn_bwd_acc_t<bf16_t>(dy, nullptr, z, mean, invstd, gamma, dgamma, dbeta, dz, nullptr, 0, M, C, 2, acc, replicas, (hipStream_t)stream, beta);
    if (dtype == CLHIP_F32)
        return bn_bwd_acc_t<float>(dy, nullptr, z, mean, invstd, gamma, dgamma, dbeta, dz, nullptr, 0, M, C, 2, acc, replicas, (hipStream_t)stream, beta);
    CLHIP_CHECK_ARG(!"dtype");
    return CLHIP_EINVAL;
}

extern "C" int clhip_bn_bwd_apply_acc(const void* dy, const void* y, const void* z, const float* mean, const float* invstd, const float* gamma,
                                      const float* beta, float* dgamma, float* dbeta, void* dz, void* dres, int dres_accumulate, int64_t M, int C,
                                      int relu, const double* acc, int replicas, int dtype, void* stream) {
    CLHIP_CHECK_ARG(dy && z && mean && invstd && gamma && dgamma && dbeta && dz && acc && M > 0 && acc_ok(C));
    CLHIP_CHECK_ARG(replicas >= 1 && replicas <= 64 && (replicas & (replicas - 1)) == 0);
    CLHIP_CHECK_ARG(relu >= 0 && relu <= 3 && ((relu != 1 && relu != 3) || y) && (relu != 2 || (beta && dres == nullptr)));
    double* a = const_cast<double*>(acc);
    if (dtype == CLHIP_BF16)
        return bn_bwd_acc_t<bf16_t>(dy, y, z, mean, invstd, gamma, dgamma, dbeta, dz, dres, dres_accumulate, M, C, relu, a, replicas, (hipStream_t)stream, beta, true);
    if (dtype == CLHIP_F32)
        return bn_bwd_acc_t<float>(dy, y, z, mean, invstd, gamma, dgamma, dbeta, dz, dres, dres_accumulate, M, C, relu, a, replicas, (hipStream_t)stream, beta, true);
    CLHIP_CHECK_ARG(!"dtype");
    return CLHIP_EINVAL;
}

extern "C" int clhip_avgpool_fwd(const void* a, float* feat, int N, int HW, int C, int dtype, void* stream) {
    CLHIP_CHECK_ARG(a && feat && N > 0 && HW > 0 && C > 0);
    dim3 g((N * C + 255) / 256), b(256);
    if (dtype == CLHIP_BF16) hipLaunchKernelGGL((avgpool_fwd_kernel<bf16_t>), g, b, 0, (hipStream_t)stream, (const bf16_t*)a, feat, N, HW, C);
    else if (dtype == CLHIP_F32) hipLaunchKernelGGL((avgpool_fwd_kernel<float>), g, b, 0, (hipStream_t)stream, (const float*)a, feat, N, HW, C);
    else { CLHIP_CHECK_ARG(!"dtype"); }
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

// C a power of two in [8, 2048] (a workgroup's threads then cover whole channel groups: thread t feeds channels (t mod C/8) * 8 ..), bf16 / f32
extern "C" int clhip_avgpool_bwd_bn_reduce_supported(int N, int HW, int C, int dtype) {
    return (N > 0 && HW > 0 && C >= 8 && C <= 2048 && (C & (C - 1)) == 0 && (dtype == CLHIP_BF16 || dtype == CLHIP_F32)) ? 1 : 0;
}
extern "C" int clhip_avgpool_bwd_bn_reduce(const float* dfeat, void* da, const void* z_prod, const void* y_prod, const float* mean, const float* invstd, double* acc,
                                           int replicas, int N, int HW, int C, int dtype, void* stream) {
    CLHIP_CHECK_ARG(dfeat && da && z_prod && mean && invstd && acc && replicas >= 1 && replicas <= 64 && (replicas & (replicas - 1)) == 0);
    CLHIP_CHECK_ARG(clhip_avgpool_bwd_bn_reduce_supported(N, HW, C, dtype));
    const int64_t nch = (int64_t)N * HW * C / 8;
    int blocks = (int)((nch + 256 * 4 - 1) / (256 * 4));     // ~4 chunks per thread: enough workgroups to fill the chip on the small last stages
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    dim3 g(blocks), b(256);
    if (dtype == CLHIP_BF16) hipLaunchKernelGGL((avgpool_bwd_bn_kernel<bf16_t>), g, b, 0, (hipStream_t)stream, dfeat, (bf16_t*)da, N, HW, C, (const bf16_t*)z_prod, (const bf16_t*)y_prod, mean, invstd, acc, replicas);
    else hipLaunchKernelGGL((avgpool_bwd_bn_kernel<float>), g, b, 0, (hipStream_t)stream, dfeat, (float*)da, N, HW, C, (const float*)z_prod, (const float*)y_prod, mean, invstd, acc, replicas);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_avgpool_bwd(const float* dfeat, void* da, int N, int HW, int C, int dtype, void* stream) {
    CLHIP_CHECK_ARG(dfeat && da && N > 0 && HW > 0 && C >= 8 && C % 8 == 0);
    dim3 g(ew_blocks((int64_t)N * HW * C / 8)), b(256);
    if (dtype == CLHIP_BF16) hipLaunchKernelGGL((avgpool_bwd_kernel<bf16_t>), g, b, 0, (hipStream_t)stream, dfeat, (bf16_t*)da, N, HW, C);
    else if (dtype == CLHIP_F32) hipLaunchKernelGGL((avgpool_bwd_kernel<float>), g, b, 0, (hipStream_t)stream, dfeat, (float*)da, N, HW, C);
    else { CLHIP_CHECK_ARG(!"dtype"); }
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_add_stats_blocks(int64_t M, int C) { return add_stats_blocks(M, C); }

extern "C" int clhip_add_stats(void* z, const void* r, double* stat_acc, int replicas, int64_t M, int C, int dtype, void* stream) {
    CLHIP_CHECK_ARG(z && r && M > 0 && C >= 8 && C % 8 == 0 && C <= 2048 && 256 % (C >> 3) == 0);
    CLHIP_CHECK_ARG(stat_acc == nullptr || (replicas >= 1 && (replicas & (replicas - 1)) == 0));
    dim3 g(add_stats_blocks(M, C)), b(256);
    const size_t lds = stat_acc ? 256 * 16 * sizeof(float) : 0;
    if (dtype == CLHIP_BF16) hipLaunchKernelGGL((add_stats_kernel<bf16_t>), g, b, lds, (hipStream_t)stream, (bf16_t*)z, (const bf16_t*)r, stat_acc, replicas, M, C);
    else if (dtype == CLHIP_F32) hipLaunchKernelGGL((add_stats_kernel<float>), g, b, lds, (hipStream_t)stream, (float*)z, (const float*)r, stat_acc, replicas, M, C);
    else { CLHIP_CHECK_ARG(!"dtype"); }
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_add_inplace(void* a, const void* b_, int64_t n, int dtype, void* stream) {
    CLHIP_CHECK_ARG(a && b_ && n > 0 && n % 8 == 0);
    dim3 g(ew_blocks(n / 8)), b(256);
    if (dtype == CLHIP_BF16) hipLaunchKernelGGL((add_inplace_kernel<bf16_t>), g, b, 0, (hipStream_t)stream, (bf16_t*)a, (const bf16_t*)b_, n / 8);
    else if (dtype == CLHIP_F32) hipLaunchKernelGGL((add_inplace_kernel<float>), g, b, 0, (hipStream_t)stream, (float*)a, (const float*)b_, n / 8);
    else { CLHIP_CHECK_ARG(!"dtype"); }
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_avgpool_win_fwd(const void* a, float* feat, int N, int H, int W, int C, int win, int dtype, void* stream) {
    CLHIP_CHECK_ARG(a && feat && N > 0 && C > 0 && win > 0 && H >= win && W >= win);
    const int total = N * (H / win) * (W / win) * C;
    dim3 g((total + 255) / 256), b(256);
    if (dtype == CLHIP_BF16) hipLaunchKernelGGL((avgpool_win_fwd_kernel<bf16_t>), g, b, 0, (hipStream_t)stream, (const bf16_t*)a, feat, N, H, W, C, win);
    else if (dtype == CLHIP_F32) hipLaunchKernelGGL((avgpool_win_fwd_kernel<float>), g, b, 0, (hipStream_t)stream, (const float*)a, feat, N, H, W, C, win);
    else { CLHIP_CHECK_ARG(!"dtype"); }
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_avgpool_win_bwd(const float* dfeat, void* da, int N, int H, int W, int C, int win, int dtype, void* stream) {
    CLHIP_CHECK_ARG(dfeat && da && N > 0 && C >= 8 && C % 8 == 0 && win > 0 && H >= win && W >= win);
    dim3 g(ew_blocks((int64_t)N * H * W * C / 8)), b(256);
    if (dtype == CLHIP_BF16) hipLaunchKernelGGL((avgpool_win_bwd_kernel<bf16_t>), g, b, 0, (hipStream_t)stream, dfeat, (bf16_t*)da, N, H, W, C, win);
    else if (dtype == CLHIP_F32) hipLaunchKernelGGL((avgpool_win_bwd_kernel<float>), g, b, 0, (hipStream_t)stream, dfeat, (float*)da, N, H, W, C, win);
    else { CLHIP_CHECK_ARG(!"dtype"); }
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_nchw_to_nhwc(const float* x, void* y, int N, int C, int H, int W, int Cpad, int dtype, void* stream) {
    CLHIP_CHECK_ARG(x && y && N > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C && Cpad % 8 == 0);
    int64_t npix = (int64_t)N * H * W;
    dim3 g((unsigned)((npix + 255) / 256)), b(256);
    if (dtype == CLHIP_BF16) hipLaunchKernelGGL((nchw_to_nhwc_kernel<bf16_t>), g, b, 0, (hipStream_t)stream, x, (bf16_t*)y, N, C, H * W, Cpad);
    else if (dtype == CLHIP_F32) hipLaunchKernelGGL((nchw_to_nhwc_kernel<float>), g, b, 0, (hipStream_t)stream, x, (float*)y, N, C, H * W, Cpad);
    else { CLHIP_CHECK_ARG(!"dtype"); }
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_nhwc_to_nchw(const void* y, float* x, int N, int C, int H, int W, int dtype, void* stream) {
    CLHIP_CHECK_ARG(x && y && N > 0 && C > 0 && H > 0 && W > 0);
    int64_t n = (int64_t)N * C * H * W;
    dim3 g((unsigned)((n + 255) / 256)), b(256);
    if (dtype == CLHIP_BF16) hipLaunchKernelGGL((nhwc_to_nchw_kernel<bf16_t>), g, b, 0, (hipStream_t)stream, (const bf16_t*)y, x, N, C, H * W);
    else if (dtype == CLHIP_F32) hipLaunchKernelGGL((nhwc_to_nchw_kernel<float>), g, b, 0, (hipStream_t)stream, (const float*)y, x, N, C, H * W);
    else { CLHIP_CHECK_ARG(!"dtype"); }
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_conv_weight_prep(const float* w, void* w_fwd, void* w_dg, int K, int taps, int Creal, int Cpad, int dtype,
                                      void* stream) {
    CLHIP_CHECK_ARG(w && w_fwd && K > 0 && taps > 0 && Creal > 0 && Cpad >= Creal);
    int64_t n = (int64_t)K * taps * Cpad;
    dim3 g((unsigned)((n + 255) / 256)), b(256);
    if (dtype == CLHIP_BF16) hipLaunchKernelGGL((weight_prep_kernel<bf16_t>), g, b, 0, (hipStream_t)stream, w, (bf16_t*)w_fwd, (bf16_t*)w_dg, K, taps, Creal, Cpad);
    else if (dtype == CLHIP_F32) hipLaunchKernelGGL((weight_prep_kernel<float>), g, b, 0, (hipStream_t)stream, w, (float*)w_fwd, (float*)w_dg, K, taps, Creal, Cpad);
    else { CLHIP_CHECK_ARG(!"dtype"); }
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}
