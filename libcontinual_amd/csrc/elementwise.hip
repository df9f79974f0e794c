// elementwise.hip -- flat-buffer ("multi-tensor") fp32 kernels: EWC penalty + gradient, Fisher-diagonal
// accumulation / merge, fused SGD-momentum-weight-decay (+EWC term, +grad scale), Adam, squared norm.
// The reference launches ~3 tiny torch kernels per parameter tensor per step for each of these
// (ewc.py:221-225 loop over ~100 tensors; torch.optim foreach); here every family is ONE launch over
// the flat parameter buffer, float4 per lane, grid-stride, wave-reduced.
#include "common.h"

namespace {

int ew_blocks4(int64_t n4) {
    int64_t b = (n4 + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

__global__ __launch_bounds__(256) void ewc_penalty_kernel(const float* __restrict__ p, const float* __restrict__ ref,
                                                          const float* __restrict__ f, int64_t n, int64_t n4, float weight,
                                                          float* out) {
    __shared__ float red[4];
    float acc = 0.f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 a = reinterpret_cast<const float4*>(p)[i], b = reinterpret_cast<const float4*>(ref)[i],
               c = reinterpret_cast<const float4*>(f)[i];
        float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
        acc += c.x * d0 * d0 + c.y * d1 * d1 + c.z * d2 * d2 + c.w * d3 * d3;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float d = p[i] - ref[i];
        acc += f[i] * d * d;
    }
    float s = block_sum_256(acc, red);
    if (threadIdx.x == 0) atomicAdd(out, 0.5f * weight * s);
}

__global__ __launch_bounds__(256) void ewc_grad_kernel(const float* __restrict__ p, const float* __restrict__ ref,
                                                       const float* __restrict__ f, float* __restrict__ g, int64_t n, float weight,
                                                       const float* __restrict__ dev_scale) {
    if (dev_scale != nullptr) weight *= *dev_scale;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) g[i] += weight * f[i] * (p[i] - ref[i]);
}

// several (p, ref, F[, g]) segments in one launch: EWC's parameters are the backbone's flat buffer plus two head tensors, and three launches of a
// few microseconds each per pass were three places in a launch-bound step's queue
constexpr int kEwcSegs = 4;
struct EwcSeg { const float* p; const float* ref; const float* f; float* g; long long n, n4; unsigned first_block, blocks; };
struct EwcTable { int n; EwcSeg s[kEwcSegs]; };

__global__ __launch_bounds__(256) void ewc_penalty_multi_kernel(EwcTable t, float weight, float* out) {
    __shared__ float red[4];
    int u = 0;
    while (u + 1 < t.n && blockIdx.x >= t.s[u + 1].first_block) ++u;
    const EwcSeg& e = t.s[u];
    const long long stride = (long long)e.blocks * 256;
    const long long i0 = (long long)(blockIdx.x - e.first_block) * 256 + threadIdx.x;
    float acc = 0.f;
    for (long long i = i0; i < e.n4; i += stride) {
        float4 a = reinterpret_cast<const float4*>(e.p)[i], b = reinterpret_cast<const float4*>(e.ref)[i], c = reinterpret_cast<const float4*>(e.f)[i];
        float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
        acc += c.x * d0 * d0 + c.y * d1 * d1 + c.z * d2 * d2 + c.w * d3 * d3;
    }
    for (long long i = (e.n4 << 2) + i0; i < e.n; i += stride) {
        float d = e.p[i] - e.ref[i];
        acc += e.f[i] * d * d;
    }
    float s = block_sum_256(acc, red);
    if (threadIdx.x == 0) atomicAdd(out, 0.5f * weight * s);
}

__global__ __launch_bounds__(256) void ewc_grad_multi_kernel(EwcTable t, float weight, const float* __restrict__ dev_scale) {
    if (dev_scale != nullptr) weight *= *dev_scale;
    int u = 0;
    while (u + 1 < t.n && blockIdx.x >= t.s[u + 1].first_block) ++u;
    const EwcSeg& e = t.s[u];
    const long long stride = (long long)e.blocks * 256;
    for (long long i = (long long)(blockIdx.x - e.first_block) * 256 + threadIdx.x; i < e.n; i += stride) e.g[i] += weight * e.f[i] * (e.p[i] - e.ref[i]);
}

__global__ __launch_bounds__(256) void fisher_accum_kernel(float* __restrict__ fi, const float* __restrict__ g, int64_t n, float scale) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float v = g[i];
        fi[i] += v * v * scale;
    }
}

__global__ __launch_bounds__(256) void fisher_merge_kernel(float* __restrict__ nf, const float* __restrict__ of, int64_t n, float alpha) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) nf[i] = alpha * of[i] + (1.f - alpha) * nf[i];
}

// torch.optim.SGD: d = g + wd*p ; buf = momentum*buf + d ; p -= lr*buf   (first step: buf = d == momentum*0 + d)
template <bool MOM, bool EWC>
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, int64_t n,
                                                  float lr, float momentum, float wd, float gscale, const float* __restrict__ ref,
                                                  const float* __restrict__ fi, float ew) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float pv = p[i];
        float d = g[i] * gscale;
        if (EWC) d += ew * fi[i] * (pv - ref[i]);
        d = fmaf(wd, pv, d);
        if (MOM) {
            float b = fmaf(momentum, m[i], d);
            m[i] = b;
            d = b;
        }
        p[i] = fmaf(-lr, d, pv);
    }
}

// the same update over several tensors in one launch (a backbone's flat buffer + the head's weight and bias: three launches per step otherwise)
constexpr int kSgdSegs = 8;
struct SgdSeg { float* p; float* g; float* m; long long n; unsigned first_block, blocks; int zero; };
struct SgdTable { int n; SgdSeg s[kSgdSegs]; };
template <bool MOM>
__global__ __launch_bounds__(256) void sgd_multi_kernel(SgdTable t, float lr, float momentum, float wd, float gscale) {
    int u = 0;
    while (u + 1 < t.n && blockIdx.x >= t.s[u + 1].first_block) ++u;
    const SgdSeg& e = t.s[u];
    const long long stride = (long long)e.blocks * 256;
    for (long long i = (long long)(blockIdx.x - e.first_block) * 256 + threadIdx.x; i < e.n; i += stride) {
        float pv = e.p[i];
        float d = e.g[i] * gscale;
        d = fmaf(wd, pv, d);
        if (MOM) {
            float b = fmaf(momentum, e.m[i], d);
            e.m[i] = b;
            d = b;
        }
        e.p[i] = fmaf(-lr, d, pv);
        if (e.zero) e.g[i] = 0.f;          // the consumed gradient leaves zeroed: the next backward accumulates into it without a fill launch
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                                                   float gscale, float bc1, float bc2s) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float pv = p[i];
        float d = fmaf(wd, pv, g[i] * gscale);
        float mi = b1 * m[i] + (1.f - b1) * d;
        float vi = b2 * v[i] + (1.f - b2) * d * d;
        m[i] = mi; v[i] = vi;
        float denom = sqrtf(vi) / bc2s + eps;
        p[i] = pv - (lr / bc1) * (mi / denom);
    }
}

__global__ __launch_bounds__(256) void sq_norm_kernel(const float* __restrict__ g, int64_t n, float* out) {
    __shared__ float red[4];
    float acc = 0.f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += g[i] * g[i];
    float s = block_sum_256(acc, red);
    if (threadIdx.x == 0) atomicAdd(out, s);
}

__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ g, int64_t n, float s) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) g[i] *= s;
}

__global__ __launch_bounds__(256) void scale_dev_kernel(const float* __restrict__ g, float* __restrict__ out, int64_t n, float s,
                                                        const float* __restrict__ dev_scale) {
    if (dev_scale != nullptr) s *= *dev_scale;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = g[i] * s;
}

__global__ __launch_bounds__(256) void sigma_fwd_kernel(const float* __restrict__ sc, const float* __restrict__ sigma, float* __restrict__ out,
                                                        int64_t n) {
    const float s = *sigma;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = sc[i] * s;
}

__global__ __launch_bounds__(256) void sigma_bwd_kernel(const float* __restrict__ sc, const float* __restrict__ sigma,
                                                        const float* __restrict__ dl, float* __restrict__ dsc, float* dsigma, int64_t n) {
    __shared__ float red[4];
    const float s = *sigma;
    float acc = 0.f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float d = dl[i];
        acc = fmaf(d, sc[i], acc);
        if (dsc != nullptr) dsc[i] = d * s;
    }
    float t = block_sum_256(acc, red);
    if (threadIdx.x == 0 && dsigma != nullptr) atomicAdd(dsigma, t);
}

}  // namespace

extern "C" int clhip_ewc_penalty(const float* p, const float* ref, const float* fisher, int64_t n, float weight, float* loss_out,
                                 int loss_accumulate, void* stream) {
    CLHIP_CHECK_ARG(p && ref && fisher && loss_out && n >= 0);
    hipStream_t st = (hipStream_t)stream;
    if (!loss_accumulate) {
        if (hipMemsetAsync(loss_out, 0, sizeof(float), st) != hipSuccess) { clhip_set_error("memset failed"); return CLHIP_EHIP; }
    }
    if (n == 0) return CLHIP_OK;
    // float4 path only when all three pointers are 16-byte aligned (sliced heads may not be)
    bool aligned = (((uintptr_t)p | (uintptr_t)ref | (uintptr_t)fisher) & 15) == 0;
    int64_t n4 = aligned ? (n >> 2) : 0;
    hipLaunchKernelGGL(ewc_penalty_kernel, dim3(ew_blocks4(aligned ? (n >> 2) + 1 : n)), dim3(256), 0, st, p, ref, fisher, n, n4,
                       weight, loss_out);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

static int ewc_table(int count, const float* const* p, const float* const* ref, const float* const* fisher, float* const* g, const int64_t* n, bool vec4,
                     EwcTable& t, unsigned& blocks) {
    t.n = 0; blocks = 0;
    for (int k = 0; k < count; ++k) {
        if (n[k] == 0) continue;
        EwcSeg& e = t.s[t.n++];
        e.p = p[k]; e.ref = ref[k]; e.f = fisher[k]; e.g = g ? g[k] : nullptr; e.n = n[k];
        const bool aligned = vec4 && (((uintptr_t)p[k] | (uintptr_t)ref[k] | (uintptr_t)fisher[k]) & 15) == 0;
        e.n4 = aligned ? (n[k] >> 2) : 0;
        e.first_block = blocks;
        e.blocks = (unsigned)ew_blocks4(aligned ? (n[k] >> 2) + 1 : n[k]);
        blocks += e.blocks;
    }
    return t.n;
}

extern "C" int clhip_ewc_penalty_multi(int count, const float* const* p, const float* const* ref, const float* const* fisher, const int64_t* n, float weight,
                                       float* loss_out, int loss_accumulate, void* stream) {
    CLHIP_CHECK_ARG(count >= 1 && count <= kEwcSegs && p && ref && fisher && n && loss_out);
    for (int k = 0; k < count; ++k) CLHIP_CHECK_ARG(n[k] >= 0 && (n[k] == 0 || (p[k] && ref[k] && fisher[k])));
    hipStream_t st = (hipStream_t)stream;
    if (!loss_accumulate) {
        if (hipMemsetAsync(loss_out, 0, sizeof(float), st) != hipSuccess) { clhip_set_error("memset failed"); return CLHIP_EHIP; }
    }
    EwcTable t;
    unsigned blocks;
    if (ewc_table(count, p, ref, fisher, nullptr, n, true, t, blocks) == 0) return CLHIP_OK;
    hipLaunchKernelGGL(ewc_penalty_multi_kernel, dim3(blocks), dim3(256), 0, st, t, weight, loss_out);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_ewc_grad_multi(int count, const float* const* p, const float* const* ref, const float* const* fisher, float* const* g, const int64_t* n,
                                    float weight, const float* dev_scale, void* stream) {
    CLHIP_CHECK_ARG(count >= 1 && count <= kEwcSegs && p && ref && fisher && g && n);
    for (int k = 0; k < count; ++k) CLHIP_CHECK_ARG(n[k] >= 0 && (n[k] == 0 || (p[k] && ref[k] && fisher[k] && g[k])));
    EwcTable t;
    unsigned blocks;
    if (ewc_table(count, p, ref, fisher, g, n, false, t, blocks) == 0) return CLHIP_OK;
    hipLaunchKernelGGL(ewc_grad_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, weight, dev_scale);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_ewc_grad(const float* p, const float* ref, const float* fisher, float* g, int64_t n, float weight,
                              const float* dev_scale, void* stream) {
    CLHIP_CHECK_ARG(p && ref && fisher && g && n >= 0);
    if (n == 0) return CLHIP_OK;
    hipLaunchKernelGGL(ewc_grad_kernel, dim3(ew_blocks4(n)), dim3(256), 0, (hipStream_t)stream, p, ref, fisher, g, n, weight, dev_scale);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_fisher_accum(float* fisher, const float* g, int64_t n, float scale, void* stream) {
    CLHIP_CHECK_ARG(fisher && g && n >= 0);
    if (n == 0) return CLHIP_OK;
    hipLaunchKernelGGL(fisher_accum_kernel, dim3(ew_blocks4(n)), dim3(256), 0, (hipStream_t)stream, fisher, g, n, scale);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_fisher_merge(float* new_f, const float* old_f, int64_t n, float alpha, void* stream) {
    CLHIP_CHECK_ARG(new_f && old_f && n >= 0);
    if (n == 0) return CLHIP_OK;
    hipLaunchKernelGGL(fisher_merge_kernel, dim3(ew_blocks4(n)), dim3(256), 0, (hipStream_t)stream, new_f, old_f, n, alpha);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_sgd_step(float* p, const float* g, float* mom, int64_t n, float lr, float momentum, float weight_decay,
                              float grad_scale, const float* ewc_ref, const float* ewc_fisher, float ewc_weight, void* stream) {
    CLHIP_CHECK_ARG(p && g && n >= 0);
    CLHIP_CHECK_ARG(momentum == 0.f || mom != nullptr);
    CLHIP_CHECK_ARG((ewc_ref == nullptr) == (ewc_fisher == nullptr));
    if (n == 0) return CLHIP_OK;
    dim3 gr(ew_blocks4(n)), b(256);
    hipStream_t st = (hipStream_t)stream;
    bool mo = momentum != 0.f, ew = ewc_ref != nullptr;
#define SGD(M, E) hipLaunchKernelGGL((sgd_kernel<M, E>), gr, b, 0, st, p, g, mom, n, lr, momentum, weight_decay, grad_scale, ewc_ref, ewc_fisher, ewc_weight)
    if (mo && ew) SGD(true, true); else if (mo) SGD(true, false); else if (ew) SGD(false, true); else SGD(false, false);
#undef SGD
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_sgd_step_multi(int count, float* const* p, const float* const* g, float* const* mom, const int64_t* n, float lr, float momentum,
                                    float weight_decay, float grad_scale, void* stream) {
    return clhip_sgd_step_multi_zero(count, p, const_cast<float* const*>(reinterpret_cast<const float* const*>(g)), mom, n, lr, momentum, weight_decay, grad_scale, 0u, stream);
}

extern "C" int clhip_sgd_step_multi_zero(int count, float* const* p, float* const* g, float* const* mom, const int64_t* n, float lr, float momentum,
                                         float weight_decay, float grad_scale, unsigned zero_grad_mask, void* stream) {
    CLHIP_CHECK_ARG(count >= 1 && count <= kSgdSegs && p && g && n);
    CLHIP_CHECK_ARG(momentum == 0.f || mom != nullptr);
    SgdTable t;
    t.n = 0;
    unsigned blocks = 0;
    for (int k = 0; k < count; ++k) {
        CLHIP_CHECK_ARG(n[k] >= 0 && (n[k] == 0 || (p[k] && g[k] && (momentum == 0.f || mom[k]))));
        if (n[k] == 0) continue;
        SgdSeg& e = t.s[t.n++];
        e.p = p[k]; e.g = g[k]; e.m = momentum != 0.f ? mom[k] : nullptr; e.n = n[k]; e.zero = (zero_grad_mask >> k) & 1u;
        e.first_block = blocks; e.blocks = (unsigned)ew_blocks4(n[k]);
        blocks += e.blocks;
    }
    if (t.n == 0) return CLHIP_OK;
    if (momentum != 0.f) hipLaunchKernelGGL(sgd_multi_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, lr, momentum, weight_decay, grad_scale);
    else hipLaunchKernelGGL(sgd_multi_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, lr, momentum, weight_decay, grad_scale);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                               float eps, float weight_decay, float grad_scale, int step, void* stream) {
    CLHIP_CHECK_ARG(p && g && m && v && n >= 0 && step >= 1);
    if (n == 0) return CLHIP_OK;
    float bc1 = 1.f - powf(beta1, (float)step);
    float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks4(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                       weight_decay, grad_scale, bc1, bc2s);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_sq_norm(const float* g, int64_t n, float* out, int accumulate, void* stream) {
    CLHIP_CHECK_ARG(g && out && n >= 0);
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate) {
        if (hipMemsetAsync(out, 0, sizeof(float), st) != hipSuccess) { clhip_set_error("memset failed"); return CLHIP_EHIP; }
    }
    if (n == 0) return CLHIP_OK;
    hipLaunchKernelGGL(sq_norm_kernel, dim3(ew_blocks4(n)), dim3(256), 0, st, g, n, out);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_scale(float* g, int64_t n, float s, void* stream) {
    CLHIP_CHECK_ARG(g && n >= 0);
    if (n == 0) return CLHIP_OK;
    hipLaunchKernelGGL(scale_kernel, dim3(ew_blocks4(n)), dim3(256), 0, (hipStream_t)stream, g, n, s);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_scale_dev(const float* g, float* out, int64_t n, float s, const float* dev_scale, void* stream) {
    CLHIP_CHECK_ARG(g && out && n >= 0);
    if (n == 0) return CLHIP_OK;
    hipLaunchKernelGGL(scale_dev_kernel, dim3(ew_blocks4(n)), dim3(256), 0, (hipStream_t)stream, g, out, n, s, dev_scale);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_sigma_scale_fwd(const float* scores, const float* sigma_dev, float* logits, int64_t n, void* stream) {
    CLHIP_CHECK_ARG(scores && sigma_dev && logits && n > 0);
    hipLaunchKernelGGL(sigma_fwd_kernel, dim3(ew_blocks4(n)), dim3(256), 0, (hipStream_t)stream, scores, sigma_dev, logits, n);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_sigma_scale_bwd(const float* scores, const float* sigma_dev, const float* dlogits, float* dscores, float* dsigma,
                                     int dsigma_accumulate, int64_t n, void* stream) {
    CLHIP_CHECK_ARG(scores && sigma_dev && dlogits && n > 0);
    hipStream_t st = (hipStream_t)stream;
    if (dsigma && !dsigma_accumulate) {
        if (hipMemsetAsync(dsigma, 0, sizeof(float), st) != hipSuccess) { clhip_set_error("memset failed"); return CLHIP_EHIP; }
    }
    hipLaunchKernelGGL(sigma_bwd_kernel, dim3(ew_blocks4(n)), dim3(256), 0, st, scores, sigma_dev, dlogits, dscores, dsigma, n);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}
