// vit_ops.hip -- the memory-bound operators of the ViT path (everything between the GEMMs and attention).
//   LayerNorm fwd / input-gradient            transformer.py:1331-1336 (block LN, eps 1e-5), :2254 (final LN, eps 1e-6)
//   patchify (image -> patch rows), token assembly (cls + pos-embed + L2P prompt tokens), prompt-token gradient
//                                              timm PatchEmbed; transformer.py:2239-2243, :2010-2014
//   final LN + pooling (cls token / mean over the prompt tokens) fwd / bwd      transformer.py:2254-2261
//   weight preparation: fp32 master -> compute-dtype [out,in] and [in,out] copies; LoRA merge W + B A
//                                              transformer.py:249-255 (rebuilt every forward), :228-234 (merge_weight)
//   LoRA B gradient through the rank-r shortcut dB = dY^T (X A^T)  (the reference's autograd forms the dense dW)
//   input Gram X^T X for InfLoRA               transformer.py:241-244
//   L2P prompt selection, pull-constraint value and key gradient                prompt.py:375-404
// All LayerNorm / prompt parameters are frozen or tiny on this path, so the backward only produces input gradients.
#include <algorithm>

#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ LayerNorm
// one wave per row; a lane owns chunks lane, lane+64, ... of 8 elements (D % 8 == 0, D <= 512 NC: NC = 2 keeps ViT-B's 768 columns at half the registers)
template <typename T, int NC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      T* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int M, int D, float eps) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nch = D >> 3;
    float v[NC][8], gm[NC][8], bt[NC][8];
    float s = 0.f;
    // (every global read of the row's lifetime is requested up front: gamma / beta behind the two reductions were a second dependent round trip per wave)
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            load8<T>(x + (size_t)row * D + c * 8, v[i]);
            load8<float>(gamma + c * 8, gm[i]);
            load8<float>(beta + c * 8, bt[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < NC; ++i)
        if (lane + 64 * i < nch) {
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[i][j];
        }
    const float mu = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i)
        if (lane + 64 * i < nch) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mu; q += d * d; }
        }
    const float rs = rsqrtf(wave_sum(q) / D + eps);
    if (lane == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mu) * rs * gm[i][j] + bt[i][j];
            store8<T>(y + (size_t)row * D + c * 8, o);
        }
    }
}

// g[row] += LN'(x[row])^T dy[row]   (input gradient only; gamma frozen)
template <typename T, int NC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ g, int M, int D) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nch = D >> 3;
    const float mu = mean[row], rs = rstd[row];
    float dg[NC][8], xh[NC][8], go[NC][8];
    float s1 = 0.f, s2 = 0.f;
    // (the accumulated-into gradient row is requested with dy and x: behind the two reductions it was a second dependent round trip per wave)
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            float d[8], xv[8], gm[8];
            load8<T>(dy + (size_t)row * D + c * 8, d);
            load8<T>(x + (size_t)row * D + c * 8, xv);
            load8<float>(gamma + c * 8, gm);
            load8<T>(g + (size_t)row * D + c * 8, go[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                dg[i][j] = d[j] * gm[j];
                xh[i][j] = (xv[j] - mu) * rs;
                s1 += dg[i][j];
                s2 += dg[i][j] * xh[i][j];
            }
        }
    }
    s1 = wave_sum(s1) / D;
    s2 = wave_sum(s2) / D;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = go[i][j] + rs * (dg[i][j] - s1 - xh[i][j] * s2);
            store8<T>(g + (size_t)row * D + c * 8, o);
        }
    }
}

// final LN (eps 1e-6) + pooling over the first P tokens of every sample -> feat[b] fp32 (P = 1: the cls token)
template <typename T>
__global__ __launch_bounds__(256) void ln_pool_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ feat, int N, int D, int P, float eps) {
    __shared__ float red[8];
    const int b = blockIdx.x, tid = threadIdx.x;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;       // thread owns columns tid + 256*i  (D <= 2048)
    for (int t = 0; t < P; ++t) {
        const T* row = x + ((size_t)b * N + t) * D;
        float v[8], s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int c = tid + 256 * i; v[i] = c < D ? Elem<T>::ld(row + c) : 0.f; s += v[i]; }
        const float mu = block_sum_256(s, red) / D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int c = tid + 256 * i; if (c < D) { const float d = v[i] - mu; q += d * d; } }
        const float rs = rsqrtf(block_sum_256(q, red + 4) / D + eps);
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int c = tid + 256 * i; if (c < D) acc[i] += (v[i] - mu) * rs * gamma[c] + beta[c]; }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { const int c = tid + 256 * i; if (c < D) feat[(size_t)b * D + c] = acc[i] / P; }
}

// g[b, t<P] = LN'(x)^T (dfeat[b] / P); g[b, t>=P] = 0      (g is the residual-stream gradient entering the last block)
template <typename T>
__global__ __launch_bounds__(256) void ln_pool_bwd_kernel(const float* __restrict__ dfeat, const T* __restrict__ x, const float* __restrict__ gamma,
                                                           T* __restrict__ g, int N, int D, int P, float eps) {
    __shared__ float red[8];
    const int b = blockIdx.x / N, t = blockIdx.x - b * N, tid = threadIdx.x;
    T* grow = g + ((size_t)b * N + t) * D;
    if (t >= P) {
        for (int c = tid; c < D; c += 256) Elem<T>::st(grow + c, 0.f);
        return;
    }
    const T* row = x + ((size_t)b * N + t) * D;
    float v[8], s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const int c = tid + 256 * i; v[i] = c < D ? Elem<T>::ld(row + c) : 0.f; s += v[i]; }
    const float mu = block_sum_256(s, red) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const int c = tid + 256 * i; if (c < D) { const float d = v[i] - mu; q += d * d; } }
    const float rs = rsqrtf(block_sum_256(q, red + 4) / D + eps);
    float dg[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = tid + 256 * i;
        dg[i] = 0.f;
        if (c < D) {
            dg[i] = dfeat[(size_t)b * D + c] / P * gamma[c];
            v[i] = (v[i] - mu) * rs;
            s1 += dg[i];
            s2 += dg[i] * v[i];
        }
    }
    __syncthreads();
    s1 = block_sum_256(s1, red) / D;
    s2 = block_sum_256(s2, red + 4) / D;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const int c = tid + 256 * i; if (c < D) Elem<T>::st(grow + c, rs * (dg[i] - s1 - v[i] * s2)); }
}

// ------------------------------------------------------------------------------------------ tokens
// images fp32 NCHW [B,3,S,S] -> patch rows [B*np, 3*p*p] (column order c, i, j = Conv2d weight flattening)
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, T* __restrict__ out, int B, int S, int p) {
    const int g = S / p, np = g * g, K = 3 * p * p;
    const size_t total = (size_t)B * np * K;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int k = idx % K;
        const size_t r = idx / K;
        const int pt = r % np, b = r / np;
        const int c = k / (p * p), ij = k - c * p * p, i = ij / p, j = ij - i * p;
        const int py = pt / g, px = pt - py * g;
        Elem<T>::st(out + idx, img[(((size_t)b * 3 + c) * S + py * p + i) * S + px * p + j]);
    }
}

// x[b, 0..P) = prompt tokens (no pos-embed); x[b, P] = cls + pos[0]; x[b, P+1+t] = patch_emb[b, t] + pos[1+t]
template <typename T>
__global__ __launch_bounds__(256) void assemble_kernel(const T* __restrict__ pe, const float* __restrict__ cls, const float* __restrict__ pos,
                                                        const float* __restrict__ prompt, T* __restrict__ x, int B, int np, int P, int D) {
    const int N = P + 1 + np;
    const size_t total = (size_t)B * N * D;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int d = idx % D;
        const size_t r = idx / D;
        const int t = r % N, b = r / N;
        float v;
        if (t < P) v = prompt[(size_t)t * D + d];
        else if (t == P) v = cls[d] + pos[d];
        else v = Elem<T>::ld(pe + ((size_t)b * np + (t - P - 1)) * D + d) + pos[(size_t)(t - P) * D + d];
        Elem<T>::st(x + idx, v);
    }
}

// dprompt[t, d] = sum_b g[b, t, d]   (t < P)
template <typename T>
__global__ __launch_bounds__(256) void prompt_grad_kernel(const T* __restrict__ g, float* __restrict__ dprompt, int B, int N, int P, int D) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P * D) return;
    const int t = idx / D, d = idx - t * D;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += Elem<T>::ld(g + ((size_t)b * N + t) * D + d);
    dprompt[idx] = s;
}

// --------------------------------------------------------------------------------- weight preparation
// w fp32 [R, C] (+ optional low-rank update rows [r0, r0+Dl): + Bm[row-r0, :] . Am[:, col]) -> wt (T, [R,C]) and wtT (T, [C,R])
template <typename T>
__global__ __launch_bounds__(256) void weight_prep2_kernel(const float* __restrict__ w, T* __restrict__ wt, T* __restrict__ wtT, int R, int C,
                                                            const float* __restrict__ Ak, const float* __restrict__ Bk, const float* __restrict__ Av,
                                                            const float* __restrict__ Bv, int rank, int Dl) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        float v = 0.f;
        if (r < R && c < C) {
            v = w[(size_t)r * C + c];
            if (rank > 0 && r >= Dl) {                            // rows [Dl,2Dl): k part, [2Dl,3Dl): v part
                const bool isk = r < 2 * Dl;
                const float* Bm = (isk ? Bk : Bv) + (size_t)(r - (isk ? Dl : 2 * Dl)) * rank;
                const float* Am = isk ? Ak : Av;
                float a = 0.f;
                for (int q = 0; q < rank; ++q) a += Bm[q] * Am[(size_t)q * C + c];
                v += a;
            }
            if (wt) Elem<T>::st(wt + (size_t)r * C + c, v);
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    if (wtT) {
        for (int i = ty; i < 32; i += 8) {
            const int c = c0 + i, r = r0 + tx;
            if (r < R && c < C) Elem<T>::st(wtT + (size_t)c * R + r, tile[tx][i]);
        }
    }
}

// per-step refresh of the k / v rows of EVERY layer's effective qkv copies in one launch (lora_B moves every optimizer step;
// the q rows and lora_A do not): blockIdx.z = layer, descriptors by value
constexpr int kQkvMax = 32;
struct QkvEntry { const float *w, *Ak, *Bk, *Av, *Bv; void *wt, *wtT; };
struct QkvTable { QkvEntry e[kQkvMax]; };

template <typename T>
__global__ __launch_bounds__(256) void lora_qkv_refresh_kernel(QkvTable t, int D, int rank) {
    __shared__ float tile[32][33];
    const QkvEntry& d = t.e[blockIdx.z];
    const int R = 3 * D, C = D;
    const int c0 = blockIdx.x * 32, r0 = D + blockIdx.y * 32;        // rows [D, 3D): k then v
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    T* wt = static_cast<T*>(d.wt);
    T* wtT = static_cast<T*>(d.wtT);
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        float v = 0.f;
        if (r < R && c < C) {
            const bool isk = r < 2 * D;
            const float* Bm = (isk ? d.Bk : d.Bv) + (size_t)(r - (isk ? D : 2 * D)) * rank;
            const float* Am = isk ? d.Ak : d.Av;
            float a = 0.f;
            for (int q = 0; q < rank; ++q) a += Bm[q] * Am[(size_t)q * C + c];
            v = d.w[(size_t)r * C + c] + a;
            Elem<T>::st(wt + (size_t)r * C + c, v);
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (r < R && c < C) Elem<T>::st(wtT + (size_t)c * R + r, tile[tx][i]);
    }
}

// merged fp32 master: qkv_w[k rows] += B_k A_k, qkv_w[v rows] += B_v A_v   (merge_weight, transformer.py:228-234)
__global__ __launch_bounds__(256) void lora_merge_master_kernel(float* __restrict__ w, const float* __restrict__ Ak, const float* __restrict__ Bk,
                                                                 const float* __restrict__ Av, const float* __restrict__ Bv, int D, int rank) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)2 * D * D) return;
    const int r = idx / D, c = idx - (size_t)r * D;                // r in [0, 2D): k rows then v rows
    const bool isk = r < D;
    const float* Bm = (isk ? Bk : Bv) + (size_t)(isk ? r : r - D) * rank;
    const float* Am = isk ? Ak : Av;
    float a = 0.f;
    for (int q = 0; q < rank; ++q) a += Bm[q] * Am[(size_t)q * D + c];
    w[(size_t)(D + r) * D + c] += a;
}

// ------------------------------------------------------------------------------------ LoRA B gradient
// P[m, 0..r) = X[m,:] . A_k^T, P[m, r..2r) = X[m,:] . A_v^T        (one wave per row, rank <= 16)
template <typename T>
__global__ __launch_bounds__(256) void lora_proj_kernel(const T* __restrict__ x, const float* __restrict__ Ak, const float* __restrict__ Av,
                                                         float* __restrict__ P, int M, int D, int rank) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float acc[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) acc[q] = 0.f;
    for (int c = lane; c < D; c += 64) {
        const float xv = Elem<T>::ld(x + (size_t)row * D + c);
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (q < rank) { acc[q] += xv * Ak[(size_t)q * D + c]; acc[16 + q] += xv * Av[(size_t)q * D + c]; }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q)
        if (q < rank) {
            const float a = wave_sum(acc[q]), b = wave_sum(acc[16 + q]);
            if (lane == 0) { P[(size_t)row * 2 * rank + q] = a; P[(size_t)row * 2 * rank + rank + q] = b; }
        }
}

// slab[s, o, q] = sum_{m in slab s} dY[m, D + o] * P[m, (o >= D ? r : 0) + q]   (o in [0, 2D): dK columns then dV columns)
template <typename T>
__global__ __launch_bounds__(256) void lora_db_kernel(const T* __restrict__ dqkv, const float* __restrict__ P, float* __restrict__ slab, int M, int D,
                                                       int rank, int rows_per_slab) {
    __shared__ float ps[64][32];
    const int o = blockIdx.x * 256 + threadIdx.x;                 // column of [dK | dV]
    const int s = blockIdx.y;
    const int m0 = s * rows_per_slab, m1 = min(M, m0 + rows_per_slab);
    float acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    for (int mb = m0; mb < m1; mb += 64) {
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * 2 * rank; i += 256) {
            const int r = i / (2 * rank), q = i - r * 2 * rank;
            ps[r][q] = (mb + r) < m1 ? P[(size_t)(mb + r) * 2 * rank + q] : 0.f;
        }
        __syncthreads();
        if (o < 2 * D) {
            const int po = (o >= D) ? rank : 0;
            const int nr = min(64, m1 - mb);
            for (int r = 0; r < nr; ++r) {
                const float dy = Elem<T>::ld(dqkv + (size_t)(mb + r) * 3 * D + D + o);
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    if (q < rank) acc[q] += dy * ps[r][po + q];
            }
        }
    }
    if (o < 2 * D) {
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (q < rank) slab[((size_t)s * 2 * D + o) * rank + q] = acc[q];
    }
}

// ---- bf16 fast path.  P16 [M, 32] = X . Acat^T comes from the MFMA GEMM (Acat = [A_k; A_v; 0] as a [32, D] bf16 matrix);
// slab[s, o, q] = sum_m dY[m, D + o] P16[m, (o >= D ? r : 0) + q] is a "TN" product: both MFMA operands are read with the
// transposing LDS read from row-major tiles (rows = m), 32 rows per step, k-slot j of lane group g <-> row g*4 + (j&3) + 16*(j>>2)
// on BOTH operands (the slot order of an MFMA is free), which keeps the reads bank-conflict free at these pitches.
constexpr int YP = 160, PP = 96;      // LDS pitches (bytes) of the dY tile rows (64 bf16) and the P tile rows (32 bf16)

__device__ __forceinline__ uint4 tr8v(const char* base, int addr, int second) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + addr));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + addr + second));
    uint2 l = __builtin_bit_cast(uint2, lo), h = __builtin_bit_cast(uint2, hi);
    return make_uint4(l.x, l.y, h.x, h.y);
}

__global__ __launch_bounds__(256) void lora_db_mfma_kernel(const bf16_t* __restrict__ dqkv, const bf16_t* __restrict__ P16, float* __restrict__ slab,
                                                            int M, int D, int rank, int rows_per_slab) {
    __shared__ __attribute__((aligned(16))) char ys[2][32 * YP];
    __shared__ __attribute__((aligned(16))) char ps[2][32 * PP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
    const int o0 = blockIdx.x * 64;                                 // column block of [dK | dV]
    const int m0 = blockIdx.y * rows_per_slab, m1 = min(M, m0 + rows_per_slab);
    const int qoff = o0 >= D ? rank : 0;                            // D % 64 == 0: a block never straddles dK / dV
    const int yr = tid >> 3, yc = tid & 7;                          // dY tile: 32 rows x 8 chunks
    const int pr = tid >> 2, pc = tid & 3;                          // P tile: 32 rows x 4 chunks (threads < 128)
    const bf16_t* ysrc = dqkv + (size_t)D + o0 + yc * 8;
    uint4 ry, rp;
    auto gload = [&](int mb) {
        const int my = mb + yr, mp = mb + pr;
        ry = my < m1 ? *reinterpret_cast<const uint4*>(ysrc + (size_t)my * 3 * D) : make_uint4(0, 0, 0, 0);
        if (tid < 128) rp = mp < m1 ? *reinterpret_cast<const uint4*>(P16 + (size_t)mp * 32 + pc * 8) : make_uint4(0, 0, 0, 0);
    };
    auto sstore = [&](int st) {
        *reinterpret_cast<uint4*>(ys[st] + yr * YP + yc * 16) = ry;
        if (tid < 128) *reinterpret_cast<uint4*>(ps[st] + pr * PP + pc * 16) = rp;
    };
    const int ya = (g * 4 + (l15 >> 2)) * YP + (wave * 16 + (l15 & 3) * 4) * 2;
    const int pa = (g * 4 + (l15 >> 2)) * PP + (l15 & 3) * 8;
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    if (m0 < m1) {
        gload(m0);
        sstore(0);
        __syncthreads();
        int st = 0;
        for (int mb = m0; mb < m1; mb += 32, st ^= 1) {
            const bool more = mb + 32 < m1;
            if (more) gload(mb + 32);
            const uint4 a = tr8v(ys[st], ya, 16 * YP);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const uint4 b = tr8v(ps[st], pa + t * 32, 16 * PP);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc[t], 0, 0, 0);
            }
            if (more) sstore(st ^ 1);
            __syncthreads();
        }
    }
    // D[row = o (g*4+e)][col = q (l15 + 16 t)]; keep q in [qoff, qoff + rank)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int q = l15 + 16 * t - qoff;
        if (q >= 0 && q < rank) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int o = o0 + wave * 16 + g * 4 + e;
                slab[((size_t)blockIdx.y * 2 * D + o) * rank + q] = acc[t][e];
            }
        }
    }
}

// Acat [32, D] (compute dtype) = rows [A_k (rank) ; A_v (rank) ; zeros]
template <typename T>
__global__ __launch_bounds__(256) void lora_acat_kernel(const float* __restrict__ Ak, const float* __restrict__ Av, T* __restrict__ out, int D, int rank) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 32 * D) return;
    const int r = idx / D, c = idx - r * D;
    float v = 0.f;
    if (r < rank) v = Ak[(size_t)r * D + c];
    else if (r < 2 * rank) v = Av[(size_t)(r - rank) * D + c];
    Elem<T>::st(out + idx, v);
}

// dBk[o, q] += sum_s slab[s, o, q];  dBv[o, q] += sum_s slab[s, D + o, q]
__global__ __launch_bounds__(256) void lora_db_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dBk, float* __restrict__ dBv, int D, int rank,
                                                              int nslab) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 2 * D * rank) return;
    float a = 0.f;
    for (int s = 0; s < nslab; ++s) a += slab[(size_t)s * 2 * D * rank + idx];
    if (idx < D * rank) dBk[idx] += a;
    else dBv[idx - D * rank] += a;
}

// ------------------------------------------------------------------------------------------------ Gram
// G[i, j] += sum_m X[m, i] X[m, j]  (fp32 accumulate, atomics across row slabs; before/after-task only)
template <typename T>
__global__ __launch_bounds__(256) void gram_kernel(const T* __restrict__ x, float* __restrict__ G, int M, int D, int rows_per_slab) {
    __shared__ float xi[32][65], xj[32][65];
    const int i0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
    const int m0 = blockIdx.z * rows_per_slab, m1 = min(M, m0 + rows_per_slab);
    const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;         // 16 x 16 threads, each 4 x 4 outputs
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
    for (int mb = m0; mb < m1; mb += 32) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < 32 * 64; idx += 256) {
            const int r = idx >> 6, c = idx & 63;
            const bool ok = (mb + r) < m1;
            xi[r][c] = (ok && i0 + c < D) ? Elem<T>::ld(x + (size_t)(mb + r) * D + i0 + c) : 0.f;
            xj[r][c] = (ok && j0 + c < D) ? Elem<T>::ld(x + (size_t)(mb + r) * D + j0 + c) : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int r = 0; r < 32; ++r) {
            float a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] = xi[r][ti * 4 + u]; b[u] = xj[r][tj * 4 + u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[u][v] += a[u] * b[v];
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = i0 + ti * 4 + u, j = j0 + tj * 4 + v;
            if (i < D && j < D) atomicAdd(G + (size_t)i * D + j, acc[u][v]);
        }
}

// bf16: G_l [D, D] += X_l^T X_l on the matrix cores, all layers of a pass in ONE launch (the Gram of every
// attention input on the device, resident, instead of transformer.py:241-244's bmm + 2.4-MB transfer per layer and batch).  A "TN" product:
// the reduction runs over the ROWS of X, so both MFMA operands come out of row-major LDS tiles through the transposing read (the
// slot order of lora_db_mfma_kernel: k-slot j of lane group g <-> row 4 g + (j & 3) + 16 (j >> 2) on both operands).  Workgroup =
// one 128 x 128 tile of one layer over ALL rows (no row split: every element of G has one owner per launch, plain fp32 read-modify-
// write, bitwise reproducible); 2 x 2 waves of 64 x 64 = 16 accumulator tiles each; 32 rows per step, register-staged, double-buffered.
constexpr int GP = 544;               // LDS row pitch: 256 bf16 (the i panel, then the j panel) + 32 bytes -> rows 0..7 fall in disjoint 32-byte bank groups

__global__ __launch_bounds__(256) void gram_mfma_kernel(const bf16_t* __restrict__ x0, size_t layer_stride, float* __restrict__ G0, int M, int D, int tiles) {
    __shared__ __attribute__((aligned(16))) char xs[2][32 * GP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
    const int layer = blockIdx.x / (tiles * tiles), t = blockIdx.x - layer * tiles * tiles;
    const int i0 = (t / tiles) * 128, j0 = (t % tiles) * 128;
    const bf16_t* x = x0 + (size_t)layer * layer_stride;
    float* G = G0 + (size_t)layer * D * D;
    const int wi = wave >> 1, wj = wave & 1;
    // staging: chunk c = tid + 256 q (q = 0..3) is (row c >> 5, 16-byte column c & 31); columns 0..15 = the i panel, 16..31 = the j panel
    uint4 rg[4];
    int srow[4], scol[4], gcol[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = tid + 256 * q, cc = c & 31;
        srow[q] = c >> 5; scol[q] = cc * 16;
        const int col = cc < 16 ? i0 + cc * 8 : j0 + (cc - 16) * 8;
        gcol[q] = col < D ? col : -1;                  // D % 8 == 0: a chunk is inside or outside
    }
    auto gload = [&](int mb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = mb + srow[q];
            rg[q] = (m < M && gcol[q] >= 0) ? *reinterpret_cast<const uint4*>(x + (size_t)m * D + gcol[q]) : make_uint4(0, 0, 0, 0);
        }
    };
    auto sstore = [&](int st) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<uint4*>(xs[st] + srow[q] * GP + scol[q]) = rg[q];
    };
    const int ra = (g * 4 + (l15 >> 2)) * GP + (l15 & 3) * 8;
    const int aa = ra + (wi * 64) * 2, ba = ra + 256 + (wj * 64) * 2;
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    gload(0);
    sstore(0);
    __syncthreads();
    int st = 0;
    for (int mb = 0; mb < M; mb += 32, st ^= 1) {
        const bool more = mb + 32 < M;
        if (more) gload(mb + 32);
        uint4 af[4], bfr[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) af[a] = tr8v(xs[st], aa + a * 32, 16 * GP);
#pragma unroll
        for (int b = 0; b < 4; ++b) bfr[b] = tr8v(xs[st], ba + b * 32, 16 * GP);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[a]), __builtin_bit_cast(bf16x8_t, bfr[b]), acc[a][b], 0, 0, 0);
        if (more) sstore(st ^ 1);
        __syncthreads();
    }
    // D[row = i (4 g + e)][col = j (l15)] of tile (a, b)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int j = j0 + wj * 64 + b * 16 + l15;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = i0 + wi * 64 + a * 16 + g * 4 + e;
                if (i < D && j < D) G[(size_t)i * D + j] += acc[a][b][e];
            }
        }
}

// ------------------------------------------------------------------------------------------------- L2P
// One block.  q [B, D] (cls features), key [pool, D].  Per-sample cosine top-k, batch-majority top-k (ties -> lowest id),
// gathered prompt tokens, reduce_sim = sum_b sum_{k in ids} <kn_k, qn_b> / B and its gradient w.r.t. key.
__global__ __launch_bounds__(1024) void l2p_select_kernel(const float* __restrict__ q, const float* __restrict__ key, const float* __restrict__ prompt,
                                                          int B, int D, int pool, int top_k, int length, int* __restrict__ ids,
                                                          float* __restrict__ prompt_tokens, float* __restrict__ reduce_sim, float* __restrict__ dkey,
                                                          float* __restrict__ scratch /* [B + pool + D + B*pool] */) {
    __shared__ int counts[64];
    __shared__ int sel[64];
    __shared__ float red[16];
    // 16 waves (round 3): the B x pool cosine products are one wave each -- with 4 waves the 160 products of the bench batch were 40
    // dependent rounds of 12 loads per lane (120 us); every loop below strides by the block's own size
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = 16, NT = 1024;
    auto block_sum = [&](float v) {
        v = wave_sum(v);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < 16; ++w2) t += red[w2];
        return t;
    };
    float* qinv = scratch;                 // [B]    1 / max(|q_b|, eps)
    float* kinv = scratch + B;             // [pool]
    float* sbar = scratch + B + pool;      // [D]    sum_b qn_b / B
    for (int r = wave; r < B + pool; r += NW) {
        const float* v = r < B ? q + (size_t)r * D : key + (size_t)(r - B) * D;
        float s = 0.f;
        for (int c = lane; c < D; c += 64) s += v[c] * v[c];
        s = wave_sum(s);
        if (lane == 0) (r < B ? qinv[r] : kinv[r - B]) = 1.0f / fmaxf(sqrtf(s), 1e-12f);
    }
    if (tid < 64) counts[tid] = 0;
    __syncthreads();
    // cosine similarities sim[b][j] (one wave per pair, lanes over D: coalesced), then per-sample top-k: one wave per sample,
    // lane = prompt id (pool <= 64)
    float* sim = scratch + B + pool + D;   // [B * pool]
    for (int pr = wave; pr < B * pool; pr += NW) {
        const int b = pr / pool, j = pr - b * pool;
        float s = 0.f;
        for (int c = lane; c < D; c += 64) s += q[(size_t)b * D + c] * key[(size_t)j * D + c];
        s = wave_sum(s);
        if (lane == 0) sim[pr] = s * qinv[b] * kinv[j];
    }
    __syncthreads();
    for (int b = wave; b < B; b += NW) {
        float sv = lane < pool ? sim[b * pool + lane] : -INFINITY;
        for (int k = 0; k < top_k; ++k) {
            float best = wave_max(sv);
            unsigned long long m = __ballot(sv == best && lane < pool);
            const int win = __ffsll((long long)m) - 1;
            if (lane == win) { atomicAdd(&counts[lane], 1); sv = -INFINITY; }
        }
    }
    __syncthreads();
    if (tid == 0) {
        for (int k = 0; k < top_k; ++k) {
            int best = -1, bc = -1;
            for (int j = 0; j < pool; ++j)
                if (counts[j] > bc) { bc = counts[j]; best = j; }
            sel[k] = best;
            ids[k] = best;
            counts[best] = -2;
        }
    }
    __syncthreads();
    for (int i = tid; i < top_k * length * D; i += NT) {
        const int k = i / (length * D), rem = i - k * length * D;
        prompt_tokens[i] = prompt[(size_t)sel[k] * length * D + rem];
    }
    for (int c = tid; c < D; c += NT) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += q[(size_t)b * D + c] * qinv[b];
        sbar[c] = s / B;
    }
    for (int i = tid; i < pool * D; i += NT) dkey[i] = 0.f;
    __syncthreads();
    float total = 0.f;
    for (int k = 0; k < top_k; ++k) {
        const int j = sel[k];
        float dot = 0.f;
        for (int c = tid; c < D; c += NT) dot += key[(size_t)j * D + c] * kinv[j] * sbar[c];
        dot = block_sum(dot);
        total += dot;
        // d<kn, s>/dkey = (s - kn <kn, s>) / |key|
        for (int c = tid; c < D; c += NT) dkey[(size_t)j * D + c] = (sbar[c] - key[(size_t)j * D + c] * kinv[j] * dot) * kinv[j];
        __syncthreads();
    }
    if (tid == 0) *reduce_sim = total;
}

// dprompt_pool[ids[k], l, :] = dtokens[k*length + l, :]; other rows zero
__global__ __launch_bounds__(256) void l2p_scatter_kernel(const float* __restrict__ dtokens, const int* __restrict__ ids, float* __restrict__ dpool, int pool,
                                                           int top_k, int length, int D) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= pool * length * D) return;
    const int j = idx / (length * D), rem = idx - j * length * D;
    float v = 0.f;
    for (int k = 0; k < top_k; ++k)
        if (ids[k] == j) v = dtokens[(size_t)k * length * D + rem];
    dpool[idx] = v;
}

inline int grid1d(size_t total) { const size_t b = (total + 255) / 256; return (int)(b < 1048576 ? b : 1048576); }

}  // namespace

#define DT_DISPATCH(dtype, CALL_BF16, CALL_F32) \
    do { if ((dtype) == CLHIP_BF16) { CALL_BF16; } else { CALL_F32; } } while (0)

extern "C" int clhip_ln_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int M, int D, float eps, int dtype,
                            void* stream) {
    CLHIP_CHECK_ARG(x && gamma && beta && y && M > 0 && D % 8 == 0 && D <= 2048 && (mean == nullptr) == (rstd == nullptr));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (D <= 1024)
        DT_DISPATCH(dtype,
                    hipLaunchKernelGGL((ln_fwd_kernel<bf16_t, 2>), dim3((M + 3) / 4), dim3(256), 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, M, D, eps),
                    hipLaunchKernelGGL((ln_fwd_kernel<float, 2>), dim3((M + 3) / 4), dim3(256), 0, s, (const float*)x, gamma, beta, (float*)y, mean, rstd, M, D, eps));
    else
        DT_DISPATCH(dtype,
                    hipLaunchKernelGGL((ln_fwd_kernel<bf16_t, 4>), dim3((M + 3) / 4), dim3(256), 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, M, D, eps),
                    hipLaunchKernelGGL((ln_fwd_kernel<float, 4>), dim3((M + 3) / 4), dim3(256), 0, s, (const float*)x, gamma, beta, (float*)y, mean, rstd, M, D, eps));
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_ln_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* g, int M, int D, int dtype,
                            void* stream) {
    CLHIP_CHECK_ARG(dy && x && gamma && mean && rstd && g && M > 0 && D % 8 == 0 && D <= 2048);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (D <= 1024)
        DT_DISPATCH(dtype,
                    hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, 2>), dim3((M + 3) / 4), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, gamma, mean, rstd, (bf16_t*)g, M, D),
                    hipLaunchKernelGGL((ln_bwd_kernel<float, 2>), dim3((M + 3) / 4), dim3(256), 0, s, (const float*)dy, (const float*)x, gamma, mean, rstd, (float*)g, M, D));
    else
        DT_DISPATCH(dtype,
                    hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, 4>), dim3((M + 3) / 4), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, gamma, mean, rstd, (bf16_t*)g, M, D),
                    hipLaunchKernelGGL((ln_bwd_kernel<float, 4>), dim3((M + 3) / 4), dim3(256), 0, s, (const float*)dy, (const float*)x, gamma, mean, rstd, (float*)g, M, D));
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_ln_pool_fwd(const void* x, const float* gamma, const float* beta, float* feat, int B, int N, int D, int P, float eps, int dtype,
                                 void* stream) {
    CLHIP_CHECK_ARG(x && gamma && beta && feat && B > 0 && P >= 1 && P <= N && D <= 2048);
    hipStream_t s = static_cast<hipStream_t>(stream);
    DT_DISPATCH(dtype, hipLaunchKernelGGL(ln_pool_fwd_kernel<bf16_t>, dim3(B), dim3(256), 0, s, (const bf16_t*)x, gamma, beta, feat, N, D, P, eps),
                hipLaunchKernelGGL(ln_pool_fwd_kernel<float>, dim3(B), dim3(256), 0, s, (const float*)x, gamma, beta, feat, N, D, P, eps));
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_ln_pool_bwd(const float* dfeat, const void* x, const float* gamma, void* g, int B, int N, int D, int P, float eps, int dtype,
                                 void* stream) {
    CLHIP_CHECK_ARG(dfeat && x && gamma && g && B > 0 && P >= 1 && P <= N && D <= 2048);
    hipStream_t s = static_cast<hipStream_t>(stream);
    DT_DISPATCH(dtype, hipLaunchKernelGGL(ln_pool_bwd_kernel<bf16_t>, dim3(B * N), dim3(256), 0, s, dfeat, (const bf16_t*)x, gamma, (bf16_t*)g, N, D, P, eps),
                hipLaunchKernelGGL(ln_pool_bwd_kernel<float>, dim3(B * N), dim3(256), 0, s, dfeat, (const float*)x, gamma, (float*)g, N, D, P, eps));
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_patchify(const float* images, void* patches, int B, int img, int patch, int dtype, void* stream) {
    CLHIP_CHECK_ARG(images && patches && B > 0 && patch > 0 && img % patch == 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t total = (size_t)B * 3 * img * img;
    DT_DISPATCH(dtype, hipLaunchKernelGGL(patchify_kernel<bf16_t>, dim3(grid1d(total)), dim3(256), 0, s, images, (bf16_t*)patches, B, img, patch),
                hipLaunchKernelGGL(patchify_kernel<float>, dim3(grid1d(total)), dim3(256), 0, s, images, (float*)patches, B, img, patch));
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_vit_assemble(const void* patch_emb, const float* cls_token, const float* pos_embed, const float* prompt_tokens, void* x, int B,
                                  int n_patches, int n_prompt, int D, int dtype, void* stream) {
    CLHIP_CHECK_ARG(patch_emb && cls_token && pos_embed && x && B > 0 && n_patches > 0 && n_prompt >= 0 && (n_prompt == 0 || prompt_tokens));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t total = (size_t)B * (n_prompt + 1 + n_patches) * D;
    DT_DISPATCH(dtype,
                hipLaunchKernelGGL(assemble_kernel<bf16_t>, dim3(grid1d(total)), dim3(256), 0, s, (const bf16_t*)patch_emb, cls_token, pos_embed, prompt_tokens, (bf16_t*)x, B, n_patches, n_prompt, D),
                hipLaunchKernelGGL(assemble_kernel<float>, dim3(grid1d(total)), dim3(256), 0, s, (const float*)patch_emb, cls_token, pos_embed, prompt_tokens, (float*)x, B, n_patches, n_prompt, D));
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_vit_prompt_grad(const void* g, float* dprompt, int B, int N, int n_prompt, int D, int dtype, void* stream) {
    CLHIP_CHECK_ARG(g && dprompt && B > 0 && n_prompt > 0 && n_prompt <= N);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int blocks = (n_prompt * D + 255) / 256;
    DT_DISPATCH(dtype, hipLaunchKernelGGL(prompt_grad_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)g, dprompt, B, N, n_prompt, D),
                hipLaunchKernelGGL(prompt_grad_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)g, dprompt, B, N, n_prompt, D));
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_weight_prep2(const float* w, void* wt, void* wt_t, int rows, int cols, const float* lora_a_k, const float* lora_b_k,
                                  const float* lora_a_v, const float* lora_b_v, int rank, int dtype, void* stream) {
    CLHIP_CHECK_ARG(w && (wt || wt_t) && rows > 0 && cols > 0 && rank >= 0 && rank <= 16);
    if (rank > 0) CLHIP_CHECK_ARG(lora_a_k && lora_b_k && lora_a_v && lora_b_v && rows % 3 == 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid((cols + 31) / 32, (rows + 31) / 32);
    DT_DISPATCH(dtype,
                hipLaunchKernelGGL(weight_prep2_kernel<bf16_t>, grid, dim3(256), 0, s, w, (bf16_t*)wt, (bf16_t*)wt_t, rows, cols, lora_a_k, lora_b_k, lora_a_v, lora_b_v, rank, rows / 3),
                hipLaunchKernelGGL(weight_prep2_kernel<float>, grid, dim3(256), 0, s, w, (float*)wt, (float*)wt_t, rows, cols, lora_a_k, lora_b_k, lora_a_v, lora_b_v, rank, rows / 3));
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_lora_qkv_refresh(int layers, const float* const* qkv_w, const float* const* lora_a_k, const float* const* lora_b_k,
                                      const float* const* lora_a_v, const float* const* lora_b_v, void* const* wt, void* const* wt_t, int D, int rank,
                                      int dtype, void* stream) {
    CLHIP_CHECK_ARG(layers > 0 && qkv_w && lora_a_k && lora_b_k && lora_a_v && lora_b_v && wt && wt_t && D > 0 && D % 32 == 0 && rank > 0 && rank <= 16);
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int l0 = 0; l0 < layers; l0 += kQkvMax) {
        const int n = layers - l0 < kQkvMax ? layers - l0 : kQkvMax;
        QkvTable t;
        for (int i = 0; i < n; ++i) {
            CLHIP_CHECK_ARG(qkv_w[l0 + i] && lora_a_k[l0 + i] && lora_b_k[l0 + i] && lora_a_v[l0 + i] && lora_b_v[l0 + i] && wt[l0 + i] && wt_t[l0 + i]);
            t.e[i] = QkvEntry{qkv_w[l0 + i], lora_a_k[l0 + i], lora_b_k[l0 + i], lora_a_v[l0 + i], lora_b_v[l0 + i], wt[l0 + i], wt_t[l0 + i]};
        }
        dim3 grid(D / 32, 2 * D / 32, n);
        DT_DISPATCH(dtype, hipLaunchKernelGGL(lora_qkv_refresh_kernel<bf16_t>, grid, dim3(256), 0, s, t, D, rank),
                    hipLaunchKernelGGL(lora_qkv_refresh_kernel<float>, grid, dim3(256), 0, s, t, D, rank));
        CLHIP_LAUNCH_CHECK();
    }
    return CLHIP_OK;
}

extern "C" int clhip_lora_merge(float* qkv_w, const float* lora_a_k, const float* lora_b_k, const float* lora_a_v, const float* lora_b_v, int D, int rank,
                                void* stream) {
    CLHIP_CHECK_ARG(qkv_w && lora_a_k && lora_b_k && lora_a_v && lora_b_v && D > 0 && rank > 0);
    hipLaunchKernelGGL(lora_merge_master_kernel, dim3((int)(((size_t)2 * D * D + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), qkv_w, lora_a_k,
                       lora_b_k, lora_a_v, lora_b_v, D, rank);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" size_t clhip_lora_grad_ws_bytes(int M, int D, int rank) {
    const int nslab = (M + 511) / 512;
    const size_t p = std::max((size_t)M * 2 * rank * sizeof(float), (size_t)M * 32 * 2);
    return (p + 255) / 256 * 256 + (size_t)nslab * 2 * D * rank * sizeof(float);
}

extern "C" int clhip_lora_acat(const float* lora_a_k, const float* lora_a_v, void* a_cat, int D, int rank, int dtype, void* stream) {
    CLHIP_CHECK_ARG(lora_a_k && lora_a_v && a_cat && D > 0 && rank > 0 && rank <= 16);
    hipStream_t s = static_cast<hipStream_t>(stream);
    DT_DISPATCH(dtype, hipLaunchKernelGGL(lora_acat_kernel<bf16_t>, dim3((32 * D + 255) / 256), dim3(256), 0, s, lora_a_k, lora_a_v, (bf16_t*)a_cat, D, rank),
                hipLaunchKernelGGL(lora_acat_kernel<float>, dim3((32 * D + 255) / 256), dim3(256), 0, s, lora_a_k, lora_a_v, (float*)a_cat, D, rank));
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_lora_grad(const void* x, const void* dqkv, const float* lora_a_k, const float* lora_a_v, const void* a_cat, float* d_b_k, float* d_b_v,
                               void* ws, int M, int D, int rank, int dtype, void* stream) {
    CLHIP_CHECK_ARG(x && dqkv && lora_a_k && lora_a_v && d_b_k && d_b_v && ws && M > 0 && rank > 0 && rank <= 16);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == CLHIP_BF16 && a_cat != nullptr && D % 64 == 0) {
        // MFMA path: P16 = X . Acat^T through the GEMM kernel, then the transposing-read TN product per 1024-row slab
        const int rows = 1024, nslab = (M + rows - 1) / rows;
        CLHIP_CHECK_ARG(nslab <= (M + 511) / 512);
        bf16_t* P16 = static_cast<bf16_t*>(ws);
        float* slab = reinterpret_cast<float*>(static_cast<char*>(ws) + (std::max((size_t)M * 2 * rank * sizeof(float), (size_t)M * 32 * 2) + 255) / 256 * 256);
        if (int rc = clhip_gemm_nt(x, a_cat, P16, nullptr, nullptr, nullptr, M, 32, D, D, D, 32, 0, 0, 0, dtype, stream)) return rc;
        hipLaunchKernelGGL(lora_db_mfma_kernel, dim3(2 * D / 64, nslab), dim3(256), 0, s, (const bf16_t*)dqkv, P16, slab, M, D, rank, rows);
        hipLaunchKernelGGL(lora_db_reduce_kernel, dim3((2 * D * rank + 255) / 256), dim3(256), 0, s, slab, d_b_k, d_b_v, D, rank, nslab);
        CLHIP_LAUNCH_CHECK();
        return CLHIP_OK;
    }
    const int nslab = (M + 511) / 512;
    float* P = static_cast<float*>(ws);
    float* slab = reinterpret_cast<float*>(static_cast<char*>(ws) + (std::max((size_t)M * 2 * rank * sizeof(float), (size_t)M * 32 * 2) + 255) / 256 * 256);
    dim3 g2((2 * D + 255) / 256, nslab);
    DT_DISPATCH(dtype,
                { hipLaunchKernelGGL(lora_proj_kernel<bf16_t>, dim3((M + 3) / 4), dim3(256), 0, s, (const bf16_t*)x, lora_a_k, lora_a_v, P, M, D, rank);
                  hipLaunchKernelGGL(lora_db_kernel<bf16_t>, g2, dim3(256), 0, s, (const bf16_t*)dqkv, P, slab, M, D, rank, 512); },
                { hipLaunchKernelGGL(lora_proj_kernel<float>, dim3((M + 3) / 4), dim3(256), 0, s, (const float*)x, lora_a_k, lora_a_v, P, M, D, rank);
                  hipLaunchKernelGGL(lora_db_kernel<float>, g2, dim3(256), 0, s, (const float*)dqkv, P, slab, M, D, rank, 512); });
    hipLaunchKernelGGL(lora_db_reduce_kernel, dim3((2 * D * rank + 255) / 256), dim3(256), 0, s, slab, d_b_k, d_b_v, D, rank, nslab);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_gram_accum_batched(const void* x, size_t layer_stride_elems, int n_layers, float* G, int M, int D, int dtype, void* stream) {
    CLHIP_CHECK_ARG(x && G && M > 0 && D > 0 && n_layers > 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == CLHIP_BF16 && D % 8 == 0) {
        const int tiles = (D + 127) / 128;
        hipLaunchKernelGGL(gram_mfma_kernel, dim3(n_layers * tiles * tiles), dim3(256), 0, s, (const bf16_t*)x, layer_stride_elems, G, M, D, tiles);
        CLHIP_LAUNCH_CHECK();
        return CLHIP_OK;
    }
    const size_t esz = dtype == CLHIP_BF16 ? 2 : 4;
    for (int l = 0; l < n_layers; ++l)
        if (int rc = clhip_gram_accum(static_cast<const char*>(x) + (size_t)l * layer_stride_elems * esz, G + (size_t)l * D * D, M, D, dtype, stream)) return rc;
    return CLHIP_OK;
}

extern "C" int clhip_gram_accum(const void* x, float* G, int M, int D, int dtype, void* stream) {
    CLHIP_CHECK_ARG(x && G && M > 0 && D > 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == CLHIP_BF16 && D % 8 == 0) return clhip_gram_accum_batched(x, 0, 1, G, M, D, dtype, stream);
    const int rows = 1024;
    dim3 grid((D + 63) / 64, (D + 63) / 64, (M + rows - 1) / rows);
    DT_DISPATCH(dtype, hipLaunchKernelGGL(gram_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, G, M, D, rows),
                hipLaunchKernelGGL(gram_kernel<float>, grid, dim3(256), 0, s, (const float*)x, G, M, D, rows));
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_l2p_select(const float* cls_feat, const float* prompt_key, const float* prompt, int B, int D, int pool, int top_k, int length, int* ids,
                                float* prompt_tokens, float* reduce_sim, float* dkey, float* scratch, void* stream) {
    CLHIP_CHECK_ARG(cls_feat && prompt_key && prompt && ids && prompt_tokens && reduce_sim && dkey && scratch);
    CLHIP_CHECK_ARG(B > 0 && D > 0 && pool > 0 && pool <= 64 && top_k > 0 && top_k <= pool && length > 0);
    hipLaunchKernelGGL(l2p_select_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), cls_feat, prompt_key, prompt, B, D, pool, top_k, length, ids,
                       prompt_tokens, reduce_sim, dkey, scratch);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_l2p_scatter(const float* dtokens, const int* ids, float* dprompt_pool, int pool, int top_k, int length, int D, void* stream) {
    CLHIP_CHECK_ARG(dtokens && ids && dprompt_pool && pool > 0 && top_k > 0 && length > 0 && D > 0);
    hipLaunchKernelGGL(l2p_scatter_kernel, dim3((pool * length * D + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), dtokens, ids, dprompt_pool,
                       pool, top_k, length, D);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}
