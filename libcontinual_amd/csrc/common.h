// common.h -- shared device helpers for libclhip (gfx950 / CDNA4 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/clhip.h"

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;   // MFMA 16x16x32 bf16 operand (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

void clhip_set_error(const char* fmt, ...);
// api.hip: the extra streams of the executors (role 0 = weight-gradient / LoRA side stream, 1 = shortcut-branch stream): process-wide, per device, never destroyed,
// BORROWED by the plans, and chosen so that a kernel on the returned stream really can start while one on `main_s` is running.  The HIP runtime maps streams to
// hardware queues round-robin in creation order (GPU_MAX_HW_QUEUES of them) and packets of one hardware queue execute in order, whichever stream they came from:
// a weight-gradient stream that lands on the caller's hardware queue turns the two-stream backward into one stream (batch-256 ResNet-18 step 2.55 instead of
// 2.0 ms -- every third / fourth model of a process, depending on the cap: round 5, tools/dp_step_micro.py, profiles/r05_notes.md).  No API tells a stream's
// queue, so the first request for a (main stream, role) pair MEASURES: a 300-us spin kernel on `main_s`, a stamp kernel on the candidate, and the candidate is taken if
// the stamp precedes the spin's end (two stream synchronisations, once per caller's stream; candidates are created until one passes, at most eight per device).
// Both roles get the SAME stream (they are busy in different halves of a step, and one shared hardware queue measured faster than two: api.hip).
hipStream_t clhip_shared_stream(int role, hipStream_t main_s, bool low_priority);
const char* clhip_cfg(const char* name);      // api.hip: value of a configuration switch (clhip_config(), else $CLHIP_<name>), nullptr if unset

#define CLHIP_CHECK_ARG(cond)                                                          \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            clhip_set_error("%s:%d: invalid argument: %s", __FILE__, __LINE__, #cond); \
            return CLHIP_EINVAL;                                                       \
        }                                                                              \
    } while (0)

#define CLHIP_LAUNCH_CHECK()                                                                          \
    do {                                                                                              \
        hipError_t e_ = hipGetLastError();                                                            \
        if (e_ != hipSuccess) {                                                                       \
            clhip_set_error("%s:%d: kernel launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return CLHIP_EHIP;                                                                        \
        }                                                                                             \
    } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {   // round-to-nearest-even
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// two fp32 -> packed bf16x2 (lo in bits 0..15) with the hardware round-to-nearest-even convert (v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// Element traits: T = bf16_t (storage uint16) or float.  A "chunk" is 8 consecutive elements.
template <typename T> struct Elem;
template <> struct Elem<bf16_t> {
    static constexpr int DTYPE = CLHIP_BF16;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};
template <> struct Elem<float> {
    static constexpr int DTYPE = CLHIP_F32;
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};

// 8-element vector load/store as fp32 registers
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&v)[8]) {
    uint4 u = *reinterpret_cast<const uint4*>(p);
    uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(w[i] << 16);
        v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
    float4 a = *reinterpret_cast<const float4*>(p);
    float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&v)[8]) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
}
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// eight bf16 values in a uint4 -> fp32
__device__ __forceinline__ void unpack8(uint4 u, float (&v)[8]) {
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(w[i] << 16);
        v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for 256-thread blocks; `red` = 4 floats of LDS per value
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

static inline int ilog2_exact(int v) {   // -1 if not a power of two
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// sum over the 16 lanes of a DPP row (lanes with equal lane >> 4), result in every lane: four v_add_f32 with a DPP source
// modifier (quad_perm xor 1, xor 2, then row_half_mirror, row_mirror) -- no LDS crossbar traffic, no address arithmetic
// (the __shfl_xor form compiles to ds_bpermute_b32 + address VALU + s_waitcnt per step)
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}

// N independent 16-lane sums, step-major: between two dependent DPP adds of one value sit the N - 1 others, so the
// VALU-write -> DPP-read hazard needs no s_nop padding
template <int N>
__device__ __forceinline__ void row16_sum_n(float (&v)[N]) {
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[q]), 0xB1, 0xF, 0xF, true));
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[q]), 0x4E, 0xF, 0xF, true));
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[q]), 0x141, 0xF, 0xF, true));
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[q]), 0x140, 0xF, 0xF, true));
}

// Sums of the fp64 accumulator replicas [rep][2][C] for the consumers' prologues: -> s1 (sum) and s2 (second sum) of channel c for the
// threads c < C (c + 256 k for wide layers).  Every load of a thread is independent and issued before the first add, and with 2 C < 256
// the replicas are split over 256 / (2 C) thread groups and combined through LDS: the serial loop this replaces made `rep` (16-32)
// dependent L2 round trips, ~10 us -- most of the 8 x 8 / 4 x 4 layers' BatchNorm launches (tools/bn_bench.py).
__device__ inline void sum_strided2(const double* __restrict__ p1, const double* __restrict__ p2, int n, size_t stride, double& s1, double& s2) {
    s1 = 0.0; s2 = 0.0;
    for (int r0 = 0; r0 < n; r0 += 8) {
        double v1[8], v2[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool in = r0 + i < n;
            v1[i] = in ? p1[(size_t)(r0 + i) * stride] : 0.0;
            v2[i] = in ? p2[(size_t)(r0 + i) * stride] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { s1 += v1[i]; s2 += v2[i]; }
    }
}

// narrow layers (2 C <= 128): part sums of thread group t / (2 C) into sred[256]; afterwards channel c reads sred[part * 2 C + which * C + c]
__device__ inline void replica_parts(const double* __restrict__ acc, int rep, int C, double* sred) {
    const int n2 = 2 * C, parts = 256 / n2;
    const int col = threadIdx.x % n2, part = threadIdx.x / n2;
    const int n = (rep - part + parts - 1) / parts;                     // replicas part, part + parts, ...
    double s = 0.0;
    for (int r0 = 0; r0 < n; r0 += 8) {
        double v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = r0 + i < n ? acc[(size_t)(part + (r0 + i) * parts) * n2 + col] : 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    sred[threadIdx.x] = s;
    __syncthreads();
}


// relu(scale * z + shift) of eight bf16 values, rounded back to bf16: exactly what bn_apply_train_kernel stores (fp32 fma, max, RNE), so
// a consumer that applies it while staging its operand sees the tensor the apply pass would have written
// relu(scale * z + shift + r), rounded to bf16, and the packed mask of the STORED values (bit e = element e > 0): what
// bn_apply_train_kernel<T, true, true> stores for a conv -> BN -> +res -> ReLU layer (fma, add, max, RNE -- in that order)
__device__ __forceinline__ uint4 bn_res_relu8_bf16(uint4 v, uint4 r, const float* sc, const float* sh, unsigned& mask) {
    const unsigned w[4] = {v.x, v.y, v.z, v.w}, rr[4] = {r.x, r.y, r.z, r.w};
    unsigned o[4];
    mask = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float lo = fmaxf(fmaf(__uint_as_float(w[k] << 16), sc[2 * k], sh[2 * k]) + __uint_as_float(rr[k] << 16), 0.f);
        const float hi = fmaxf(fmaf(__uint_as_float(w[k] & 0xffff0000u), sc[2 * k + 1], sh[2 * k + 1]) + __uint_as_float(rr[k] & 0xffff0000u), 0.f);
        o[k] = pack_bf16x2(lo, hi);
        mask |= ((o[k] & 0x7fffu) != 0u && (o[k] & 0x8000u) == 0u ? 1u : 0u) << (2 * k);
        mask |= ((o[k] & 0x7fff0000u) != 0u && (o[k] & 0x80000000u) == 0u ? 1u : 0u) << (2 * k + 1);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}
// ---- "lazy" BatchNorm input of the LDS-DMA convolution kernels (conv4.hip / conv5.hip, round 4): the consumer's patch arrives in LDS as the RAW
// pre-BatchNorm output z' of the producing layer (LDS-DMA has no arithmetic on the way), and the wave that issued a DMA piece rewrites the
// slots it landed IN PLACE with relu(scale * z' + shift [+ r]) before the workgroup barrier publishes the patch.  The workgroup that owns a
// pixel also writes that activation (and, for a conv -> BN -> +res -> ReLU producer, its packed ReLU mask) to global memory -- the weight
// gradient of the consumer and the backward of the producer read it -- so the producer's BatchNorm-apply launch disappears and the
// convolution's read of the activation becomes the read of z' [and r].  Values bit for bit what bn_apply_train_kernel stores.
struct LazyIn {
    const double* acc = nullptr;    // the producer's [rep][2][C] fp64 statistics accumulators; nullptr: the source is an ordinary activation
    int rep = 1;
    const float* gamma = nullptr; const float* beta = nullptr;
    float* rm = nullptr; float* rv = nullptr;           // running statistics (updated by workgroup 0), or both nullptr
    float momentum = 0.f, eps = 0.f;
    double invM = 0.0, unbias = 1.0;
    float* mean_o = nullptr; float* invstd_o = nullptr; float* coef_o = nullptr;      // [C], [C], [2][C]: what the backward reads (workgroup 0)
    const bf16_t* res = nullptr;    // r [N,H,W,C] or nullptr
    bf16_t* y = nullptr;            // out [N,H,W,C]
    unsigned char* mask = nullptr;  // out [N*H*W*C/8] or nullptr
};

// scale / shift of a lazy input into coefs[2][C] (LDS): the arithmetic of bn_apply_train_kernel's prologue (bn.hip), bf16 mode.  Called by
// every thread of a workgroup of >= 256 threads; sred = 256 doubles of LDS; ends with a workgroup barrier.
__device__ inline void lazy_in_coefs(const LazyIn& L, int C, float* coefs, double* sred, bool first) {
    const bool narrow = 2 * C <= 128;
    if (narrow) {
        if (threadIdx.x < 256) {
            const int n2 = 2 * C, parts = 256 / n2;
            const int col = threadIdx.x % n2, part = threadIdx.x / n2;
            const int n = (L.rep - part + parts - 1) / parts;
            double s = 0.0;
            for (int r0 = 0; r0 < n; r0 += 8) {
                double v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = r0 + i < n ? L.acc[(size_t)(part + (r0 + i) * parts) * n2 + col] : 0.0;
#pragma unroll
                for (int i = 0; i < 8; ++i) s += v[i];
            }
            sred[threadIdx.x] = s;
        }
        __syncthreads();
    }
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const bool upd = first && L.rm != nullptr;
        const float rm_old = upd ? L.rm[c] : 0.f, rv_old = upd ? L.rv[c] : 0.f;
        double s1 = 0.0, s2 = 0.0;
        if (narrow) { for (int q = 0; q < 256 / (2 * C); ++q) { s1 += sred[q * 2 * C + c]; s2 += sred[q * 2 * C + C + c]; } }
        else sum_strided2(L.acc + c, L.acc + C + c, L.rep, 2 * (size_t)C, s1, s2);
        const double mean = s1 * L.invM;
        double var = s2 * L.invM - mean * mean;
        if (var < 0.0) var = 0.0;
        const float istd = rsqrtf((float)var + L.eps);
        const float sc = L.gamma[c] * istd;
        const float sh = L.beta[c] - (float)mean * sc;
        coefs[c] = sc;
        coefs[C + c] = sh;
        if (first) {
            L.mean_o[c] = (float)mean;
            L.invstd_o[c] = istd;
            L.coef_o[c] = sc;
            L.coef_o[C + c] = sh;
            if (upd) {
                L.rm[c] = (1.f - L.momentum) * rm_old + L.momentum * (float)mean;
                L.rv[c] = (1.f - L.momentum) * rv_old + L.momentum * (float)(var * L.unbias);
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ uint4 bn_relu8_bf16(uint4 v, const float* sc, const float* sh) {
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    unsigned o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float lo = fmaxf(fmaf(__uint_as_float(w[k] << 16), sc[2 * k], sh[2 * k]), 0.f);
        const float hi = fmaxf(fmaf(__uint_as_float(w[k] & 0xffff0000u), sc[2 * k + 1], sh[2 * k + 1]), 0.f);
        o[k] = pack_bf16x2(lo, hi);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}
