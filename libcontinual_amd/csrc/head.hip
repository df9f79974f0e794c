// head.hip -- classifier heads and the continual-learning loss terms, fp32.
//
//   linear / cosine-linear heads   (ewc.py:50, lwf.py:29-40, icarl.py:31; backbone/resnet.py:418-463)
//   cross entropy on a column slice with label offset + argmax + correct count (ewc.py:87-108, lwf.py:61-62)
//   distillation KD (T=2)          (lwf.py:75-78, icarl.py:198-206)
//   LUCIR less-forget cosine embedding and top-K margin ranking (lucir.py:182-205)
//   iCaRL nearest-class-mean and herding selection (icarl.py:122-152, buffer/linearherdingbuffer.py:140-161)
//
// These are tiny (B x <=100 logits); each is one launch with one wavefront per row and fuses forward
// value, gradient, prediction and accuracy so the step needs no host synchronisation.
#include <atomic>
#include <mutex>

#include "common.h"

namespace {

// ------------------------------------------------------------------------------------- linear
// out[b][o] = x[b] . w[o] + bias[o]: one wave per (b, 4 outputs); the x row chunk is reused from registers
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ out, int B, int D, int O) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int og = (O + 3) / 4;
    if (gw >= B * og) return;
    const int row = gw / og, o0 = (gw - row * og) * 4;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int d = lane * 4; d < D; d += 256) {
        const int rem = D - d;
        float xv[4];
        if (rem >= 4) { float4 t = *reinterpret_cast<const float4*>(x + (size_t)row * D + d); xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w; }
        else { for (int e = 0; e < 4; ++e) xv[e] = e < rem ? x[(size_t)row * D + d + e] : 0.f; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (o0 + j < O) {
                const float* wr = w + (size_t)(o0 + j) * D + d;
                if (rem >= 4) { float4 t = *reinterpret_cast<const float4*>(wr); s[j] += xv[0] * t.x + xv[1] * t.y + xv[2] * t.z + xv[3] * t.w; }
                else { for (int e = 0; e < rem; ++e) s[j] = fmaf(xv[e], wr[e], s[j]); }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float t = wave_sum(s[j]);
        if (lane == 0 && o0 + j < O) out[(size_t)row * O + o0 + j] = t + (b ? b[o0 + j] : 0.f);
    }
}

// dx[b][d] = sum_o dout[b][o] w[o][d]
__global__ void linear_bwd_dx_kernel(const float* __restrict__ dout, const float* __restrict__ w, float* __restrict__ dx, int B, int D,
                                     int O, int acc) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * D) return;
    int b = idx / D, d = idx - b * D;
    float s = 0.f;
    for (int o = 0; o < O; ++o) s = fmaf(dout[(size_t)b * O + o], w[(size_t)o * D + d], s);
    dx[idx] = acc ? dx[idx] + s : s;
}
// dx and dw / db of a linear head in ONE launch (the two bodies behind one grid, input-gradient workgroups first: it is the one the backbone's
// backward waits for): workgroup < ndx -> 256 elements of dx with eight weight rows in flight per trip (the plain loop was one dependent
// load -> fma chain per output: 14-20 us inside the ResNet-18 step for 256 x 512 x 50), else -> one (output row, 64 columns) tile of dw as
// linear_bwd_dw_kernel below.  Same summation orders as the two kernels: bitwise reproducible.
__global__ __launch_bounds__(256) void linear_bwd_fused_kernel(const float* __restrict__ dout, const float* __restrict__ w, const float* __restrict__ x,
                                                               float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db, int B, int D, int O,
                                                               int accumulate, int ndx, int ndwx) {
    __shared__ float red[4][64], redb[4];
    if ((int)blockIdx.x < ndx) {
        const int idx = blockIdx.x * 256 + threadIdx.x;
        if (idx >= B * D) return;
        const int b = idx / D, d = idx - b * D;
        const float* dr = dout + (size_t)b * O;
        const float* wc = w + d;
        float s = 0.f;
        int o = 0;
        for (; o + 8 <= O; o += 8) {
            float g[8], wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { g[u] = dr[o + u]; wv[u] = wc[(size_t)(o + u) * D]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) s = fmaf(g[u], wv[u], s);
        }
        for (; o < O; ++o) s = fmaf(dr[o], wc[(size_t)o * D], s);
        dx[idx] = s;
        return;
    }
    const int t = (int)blockIdx.x - ndx;
    const int bx = t % ndwx, o = t / ndwx;
    const int dl = threadIdx.x & 63, bg = threadIdx.x >> 6;
    const int d = bx * 64 + dl;
    const bool in = d < D;
    float s = 0.f, sb = 0.f;
    for (int b0 = bg; b0 < B; b0 += 32) {
        float g[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int b = b0 + 4 * u;
            g[u] = b < B ? dout[(size_t)b * O + o] : 0.f;
            xv[u] = (b < B && in) ? x[(size_t)b * D + d] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { s = fmaf(g[u], xv[u], s); sb += g[u]; }
    }
    red[bg][dl] = s;
    if (dl == 0) redb[bg] = sb;
    __syncthreads();
    if (bg == 0) {
        const float tt = (red[0][dl] + red[1][dl]) + (red[2][dl] + red[3][dl]);
        if (in) { float* q = dw + (size_t)o * D + d; *q = accumulate ? *q + tt : tt; }
        if (db != nullptr && bx == 0 && dl == 0) { const float tb = (redb[0] + redb[1]) + (redb[2] + redb[3]); db[o] = accumulate ? db[o] + tb : tb; }
    }
}
// dw[o][d] (+)= sum_b dout[b][o] x[b][d] ; db[o] (+)= sum_b dout[b][o].  One workgroup = one output row o x 64 columns d; its four
// waves take every fourth batch row (eight independent loads in flight each) and are summed through LDS in a fixed order: no atomics,
// no zeroing launches, bitwise reproducible (the batch-split atomic version was the last fp32-atomic kernel of the ResNet step).
__global__ __launch_bounds__(256) void linear_bwd_dw_kernel(const float* __restrict__ dout, const float* __restrict__ x, float* __restrict__ dw,
                                                            float* __restrict__ db, int B, int D, int O, int accumulate) {
    __shared__ float red[4][64], redb[4];
    const int dl = threadIdx.x & 63, bg = threadIdx.x >> 6;
    const int d = blockIdx.x * 64 + dl, o = blockIdx.y;
    const bool in = d < D;
    float s = 0.f, sb = 0.f;
    for (int b0 = bg; b0 < B; b0 += 32) {
        float g[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int b = b0 + 4 * u;
            g[u] = b < B ? dout[(size_t)b * O + o] : 0.f;
            xv[u] = (b < B && in) ? x[(size_t)b * D + d] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { s = fmaf(g[u], xv[u], s); sb += g[u]; }
    }
    red[bg][dl] = s;
    if (dl == 0) redb[bg] = sb;
    __syncthreads();
    if (bg == 0) {
        const float t = (red[0][dl] + red[1][dl]) + (red[2][dl] + red[3][dl]);
        if (in) { float* q = dw + (size_t)o * D + d; *q = accumulate ? *q + t : t; }
        if (db != nullptr && blockIdx.x == 0 && dl == 0) { const float tb = (redb[0] + redb[1]) + (redb[2] + redb[3]); db[o] = accumulate ? db[o] + tb : tb; }
    }
}

// ---------------------------------------------------------------------------- CE on a column slice
// SINGLE: one workgroup of 16 waves walks all rows and writes (or adds to) the loss and the correct count itself, summed in a fixed
// order -- no zeroing launches in front, no atomics, a reproducible loss value (batches of at most 512 rows: the training steps).
template <bool SINGLE>
__global__ __launch_bounds__(SINGLE ? 1024 : 256) void ce_slice_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, int B,
                                                       int O, int lo, int hi, int pred_lo, int pred_hi, float weight, float* loss_out,
                                                       float* __restrict__ dlogits, int grad_acc, int64_t* pred, int32_t* correct, int loss_acc) {
    __shared__ float wl[16];
    __shared__ int wc[16];
    const int lane = threadIdx.x & 63;
    float my_loss = 0.f; int my_correct = 0;
    for (int row = SINGLE ? (int)(threadIdx.x >> 6) : (int)(blockIdx.x * 4 + (threadIdx.x >> 6)); row < B; row += SINGLE ? 16 : B) {
    const float* lr = logits + (size_t)row * O;
    const int y = (int)labels[row];
    // argmax over [pred_lo, pred_hi): first maximal index, as torch.argmax
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int c = pred_lo + lane; c < pred_hi; c += 64) { float v = lr[c]; if (v > bv) { bv = v; bi = c; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(bv, o, 64); int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    // softmax over the slice
    float mx = -INFINITY;
    for (int c = lo + lane; c < hi; c += 64) mx = fmaxf(mx, lr[c]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int c = lo + lane; c < hi; c += 64) se += expf(lr[c] - mx);
    se = wave_sum(se);
    const float lse = mx + logf(se);
    if (dlogits != nullptr) {
        float* dr = dlogits + (size_t)row * O;
        const float sc = weight / (float)B;
        for (int c = lane; c < O; c += 64) {
            float g = 0.f;
            if (c >= lo && c < hi) g = sc * (expf(lr[c] - lse) - (c == y ? 1.f : 0.f));
            if (grad_acc) { if (c >= lo && c < hi) dr[c] += g; }
            else dr[c] = g;
        }
    }
    if (lane == 0) {
        float li = (y >= lo && y < hi) ? (lse - lr[y]) : 0.f;
        if (pred) pred[row] = bi;
        if (SINGLE) { my_loss += weight * li / (float)B; my_correct += bi == y ? 1 : 0; }
        else {
            atomicAdd(loss_out, weight * li / (float)B);
            if (correct && bi == y) atomicAdd(correct, 1);
        }
    }
    }
    if (SINGLE) {
        if (lane == 0) { wl[threadIdx.x >> 6] = my_loss; wc[threadIdx.x >> 6] = my_correct; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f; int c = 0;
            for (int w = 0; w < 16; ++w) { t += wl[w]; c += wc[w]; }
            *loss_out = loss_acc ? *loss_out + t : t;
            if (correct) *correct = c;
        }
    }
}


// ---- the training steps' form (B <= 512 rows, O <= 16 * NC columns): ONE workgroup of 1024 threads = 64 sixteen-lane groups, a group
// owns rows g, g + 64, ...; a row's logits are loaded ONCE into registers (NC per lane) -- all of a group's rows up front, so the loads
// of every row are in flight together -- and argmax / max / sum-exp are 16-lane butterflies.  The one-wave-per-row form above walked
// 16 rows per wave with three dependent passes over memory per row: 40 us for 256 x 50 logits (profiles/r02_bench_kernel_stats_*),
// this one: profiles/r03_small_kernels.md.  Loss and correct count are still summed in a fixed order (per group over its rows, then a
// 64-value butterfly): reproducible, no atomics, no zeroing launches.
// 16-lane (one DPP row) butterflies without the LDS crossbar: quad_perm xor 1, xor 2, row_half_mirror, row_mirror (common.h: row16_sum)
__device__ __forceinline__ float dpp_f(float v, int ctrl) {
    return ctrl == 0 ? __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true))
         : ctrl == 1 ? __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true))
         : ctrl == 2 ? __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true))
                     : __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
}
__device__ __forceinline__ int dpp_i(int v, int ctrl) {
    return ctrl == 0 ? __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true) : ctrl == 1 ? __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true)
         : ctrl == 2 ? __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true) : __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);
}
__device__ __forceinline__ float row16_max(float v) {
#pragma unroll
    for (int c = 0; c < 4; ++c) v = fmaxf(v, dpp_f(v, c));
    return v;
}
__device__ __forceinline__ int row16_min_i(int v) {
#pragma unroll
    for (int c = 0; c < 4; ++c) v = min(v, dpp_i(v, c));
    return v;
}

// Several workgroups (gridDim.x = G <= 8, workgroup g owns rows g * 64 RPG ..): each leaves its fixed-order partial loss / count in a scratch slot, the
// LAST one to arrive (a ticket counter) adds the G partials in index order -- still no data-dependent summation order, no zeroing launch; the slot's
// counter is reset by that workgroup.  256 rows on four CUs: 13 -> 6.5 us.
struct CeSlot { float part[8]; int cpart[8]; unsigned count; unsigned pad[15]; };

template <int NC, int RPG>
__global__ __launch_bounds__(1024) void ce_rows_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, int B, int O, int lo, int hi,
                                                       int pred_lo, int pred_hi, float weight, float* loss_out, float* __restrict__ dlogits, int grad_acc,
                                                       int64_t* pred, int32_t* correct, int loss_acc, CeSlot* slot) {
    __shared__ float gl[64];
    __shared__ int gc[64];
    const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int row0 = blockIdx.x * 64 * RPG;
    float v[RPG][NC];
    int yy[RPG];
#pragma unroll
    for (int i = 0; i < RPG; ++i) {
        const int row = row0 + grp + 64 * i;
        const bool rv = row < B;
        yy[i] = rv ? (int)labels[row] : -1;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int c = l16 + 16 * j;
            v[i][j] = (rv && c < O) ? logits[(size_t)row * O + c] : 0.f;
        }
    }
    float my_loss = 0.f; int my_correct = 0;
    const float sc = weight / (float)B;
#pragma unroll
    for (int i = 0; i < RPG; ++i) {
        const int row = row0 + grp + 64 * i;
        if (row >= B) continue;                                   // uniform over the 16-lane group
        float bv = -INFINITY; int bi = 0x7fffffff;
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int c = l16 + 16 * j;
            if (c >= pred_lo && c < pred_hi && v[i][j] > bv) { bv = v[i][j]; bi = c; }      // ascending c: the first maximum stays
            if (c >= lo && c < hi) mx = fmaxf(mx, v[i][j]);
        }
        // first maximal index of the prediction window: the row maximum, then the smallest index that attains it
        const float bmax = row16_max(bv);
        bi = row16_min_i(bv == bmax ? bi : 0x7fffffff);
        mx = row16_max(mx);
        float e[NC], se = 0.f;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int c = l16 + 16 * j;
            e[j] = (c >= lo && c < hi) ? expf(v[i][j] - mx) : 0.f;
            se += e[j];
        }
        se = row16_sum(se);
        const float lse = mx + logf(se);
        const int y = yy[i];
        if (dlogits != nullptr) {
            float* dr = dlogits + (size_t)row * O;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const int c = l16 + 16 * j;
                if (c >= O) continue;
                const bool in = c >= lo && c < hi;
                const float g = in ? sc * (expf(v[i][j] - lse) - (c == y ? 1.f : 0.f)) : 0.f;
                if (grad_acc) { if (in) dr[c] += g; }
                else dr[c] = g;
            }
        }
        // the label's logit sits in lane y & 15, register y >> 4
        float ly = 0.f;
#pragma unroll
        for (int j = 0; j < NC; ++j) if (l16 + 16 * j == y) ly = v[i][j];
        ly = row16_sum(ly);
        if (l16 == 0) {
            const float li = (y >= lo && y < hi) ? (lse - ly) : 0.f;
            if (pred) pred[row] = bi;
            my_loss += weight * li / (float)B;
            my_correct += bi == y ? 1 : 0;
        }
    }
    if (l16 == 0) { gl[grp] = my_loss; gc[grp] = my_correct; }
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = gl[threadIdx.x]; int c = gc[threadIdx.x];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { t += __shfl_xor(t, o, 64); c += __shfl_xor(c, o, 64); }
        if (threadIdx.x == 0) {
            const int G = gridDim.x;
            if (G == 1) {
                *loss_out = loss_acc ? *loss_out + t : t;
                if (correct) *correct = c;
            } else {
                __hip_atomic_store(&slot->part[blockIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&slot->cpart[blockIdx.x], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned ticket = __hip_atomic_fetch_add(&slot->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                if (ticket == (unsigned)G - 1) {                 // every other workgroup's partial is visible (release / acquire on the counter)
                    float tt = 0.f; int cc = 0;
                    for (int g = 0; g < G; ++g) {
                        tt += __hip_atomic_load(&slot->part[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        cc += __hip_atomic_load(&slot->cpart[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    *loss_out = loss_acc ? *loss_out + tt : tt;
                    if (correct) *correct = cc;
                    __hip_atomic_store(&slot->count, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------- KD
__global__ __launch_bounds__(256) void kd_kernel(const float* __restrict__ pred, int ps, const float* __restrict__ soft, int ss, int B,
                                                 int k, float invT, float weight, float* loss_out, float* __restrict__ dpred, int grad_acc) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const float* pr = pred + (size_t)row * ps;
    const float* sr = soft + (size_t)row * ss;
    float mp = -INFINITY, mq = -INFINITY;
    for (int c = lane; c < k; c += 64) { mp = fmaxf(mp, pr[c] * invT); mq = fmaxf(mq, sr[c] * invT); }
    mp = wave_max(mp); mq = wave_max(mq);
    float sp = 0.f, sq = 0.f;
    for (int c = lane; c < k; c += 64) { sp += expf(pr[c] * invT - mp); sq += expf(sr[c] * invT - mq); }
    sp = wave_sum(sp); sq = wave_sum(sq);
    const float lsep = mp + logf(sp);
    float li = 0.f;
    const float sc = weight * invT / (float)B;
    for (int c = lane; c < k; c += 64) {
        float q = expf(sr[c] * invT - mq) / sq;
        float lp = pr[c] * invT - lsep;
        li -= q * lp;
        if (dpred != nullptr) {
            float g = sc * (expf(lp) - q);
            float* d = dpred + (size_t)row * ps + c;
            *d = grad_acc ? *d + g : g;
        }
    }
    li = wave_sum(li);
    if (lane == 0) atomicAdd(loss_out, weight * li / (float)B);
}

// -------------------------------------------------------------------------------- cosine linear
__global__ __launch_bounds__(256) void row_norm_kernel(const float* __restrict__ x, float* __restrict__ nrm, int R, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) { float v = x[(size_t)row * D + d]; s = fmaf(v, v, s); }
    s = wave_sum(s);
    if (lane == 0) nrm[row] = fmaxf(sqrtf(s), 1e-12f);     // F.normalize eps
}
__global__ __launch_bounds__(256) void cosine_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ xn, const float* __restrict__ wn,
                                                         float* __restrict__ out, int B, int D, int O) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    for (int o = 0; o < O; ++o) {
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s = fmaf(x[(size_t)row * D + d], w[(size_t)o * D + d], s);
        s = wave_sum(s);
        if (lane == 0) out[(size_t)row * O + o] = s / (xn[row] * wn[o]);
    }
}
// round 4: the whole forward in ONE launch for the head sizes of the in-scope nets (the weight matrix fits LDS): a block owns eight rows of x, keeps
// w [O, D] and its rows in LDS (pitch D + 1: conflict-free for consecutive outputs), derives both sets of norms itself (block 0 stores wnorm) and
// writes the cosines -- the three launches it replaces cost 5 + 5 + 30 us at 256 x 64 -> 55 (55 dependent wave reductions per row)
constexpr int kCosRows = 8;
__global__ __launch_bounds__(256) void cosine_fwd_fused_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ xn_o, float* __restrict__ wn_o,
                                                               float* __restrict__ out, int B, int D, int O) {
    extern __shared__ __attribute__((aligned(16))) float cs[];
    const int P = D + 1;
    float* ws = cs;                       // [O][P]
    float* xs = ws + (size_t)O * P;       // [kCosRows][P]
    float* wn = xs + kCosRows * P;        // [O]
    float* xn = wn + O;                   // [kCosRows]
    const int tid = threadIdx.x, r0 = blockIdx.x * kCosRows;
    for (int e = tid; e < O * D; e += 256) { const int o = e / D, d = e - o * D; ws[o * P + d] = w[e]; }
    for (int e = tid; e < kCosRows * D; e += 256) { const int r = e / D, d = e - r * D; xs[r * P + d] = (r0 + r) < B ? x[(size_t)(r0 + r) * D + d] : 0.f; }
    __syncthreads();
    for (int i = tid; i < O + kCosRows; i += 256) {
        const float* v = i < O ? ws + (size_t)i * P : xs + (i - O) * P;
        float s = 0.f;
        for (int d = 0; d < D; ++d) s = fmaf(v[d], v[d], s);
        const float n = fmaxf(sqrtf(s), 1e-12f);                 // F.normalize eps
        if (i < O) { wn[i] = n; if (blockIdx.x == 0) wn_o[i] = n; }
        else { xn[i - O] = n; if (r0 + i - O < B) xn_o[r0 + i - O] = n; }
    }
    __syncthreads();
    for (int i = tid; i < kCosRows * O; i += 256) {
        const int r = i / O, o = i - r * O;
        if (r0 + r >= B) continue;
        const float* a = xs + r * P;
        const float* b = ws + (size_t)o * P;
        float s = 0.f;
        for (int d = 0; d < D; ++d) s = fmaf(a[d], b[d], s);
        out[(size_t)(r0 + r) * O + o] = s / (xn[r] * wn[o]);
    }
}

// dw[o][d] (+)= (1 / wn[o]) sum_b dout[b,o] (x[b,d] / xn[b] - out[b,o] wh[o,d]): one block per output row o, the batch split over four thread groups
// whose partial sums are added in group order (fixed order: reproducible); the one-thread-per-element form walked the batch as 256 dependent loads
// per thread on 14 workgroups (76 us at 256 x 64 -> 55)
__global__ __launch_bounds__(256) void cosine_bwd_dw_rows_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ out,
                                                                 const float* __restrict__ xn, const float* __restrict__ wn, const float* __restrict__ dout,
                                                                 float* __restrict__ dw, int B, int D, int O, int acc) {
    __shared__ float part[4][64];
    const int o = blockIdx.x, dl = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const float wno = wn[o];
    const int per = (B + 3) / 4, b0 = grp * per, b1 = min(B, b0 + per);
    for (int d0 = 0; d0 < D; d0 += 64) {
        const int d = d0 + dl;
        const float wh = d < D ? w[(size_t)o * D + d] / wno : 0.f;
        float s = 0.f;
        if (d < D) {
            int b = b0;
            for (; b + 4 <= b1; b += 4) {
                float g[4], xv[4], ov[4], nv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { g[u] = dout[(size_t)(b + u) * O + o]; xv[u] = x[(size_t)(b + u) * D + d]; ov[u] = out[(size_t)(b + u) * O + o]; nv[u] = xn[b + u]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) s += g[u] * (xv[u] / nv[u] - ov[u] * wh);
            }
            for (; b < b1; ++b) s += dout[(size_t)b * O + o] * (x[(size_t)b * D + d] / xn[b] - out[(size_t)b * O + o] * wh);
        }
        part[grp][dl] = s;
        __syncthreads();
        if (grp == 0 && d < D) {
            const float t = (((part[0][dl] + part[1][dl]) + part[2][dl]) + part[3][dl]) / wno;
            const size_t idx = (size_t)o * D + d;
            dw[idx] = acc ? dw[idx] + t : t;
        }
        __syncthreads();
    }
}

// dx[b][d] = sum_o dout[b,o] * (wh[o][d] - s[b,o]*xh[b][d]) / xn[b]
__global__ void cosine_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ out,
                                     const float* __restrict__ xn, const float* __restrict__ wn, const float* __restrict__ dout,
                                     float* __restrict__ dx, int B, int D, int O, int acc) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * D) return;
    int b = idx / D, d = idx - b * D;
    float xh = x[idx] / xn[b];
    float s = 0.f;
    for (int o = 0; o < O; ++o) {
        float g = dout[(size_t)b * O + o];
        s += g * (w[(size_t)o * D + d] / wn[o] - out[(size_t)b * O + o] * xh);
    }
    s /= xn[b];
    dx[idx] = acc ? dx[idx] + s : s;
}

// nn.CosineEmbeddingLoss(target=1): mean_b (1 - cos(a_b, b_b)), cos = a.b / sqrt((|a|^2+eps)(|b|^2+eps)), eps=1e-8
__global__ __launch_bounds__(256) void cos_embed_kernel(const float* __restrict__ a, const float* __restrict__ b, int B, int D,
                                                        float weight, float* loss_out, float* __restrict__ da, int grad_acc) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const float* ar = a + (size_t)row * D;
    const float* br = b + (size_t)row * D;
    float ab = 0.f, aa = 0.f, bb = 0.f;
    for (int d = lane; d < D; d += 64) { float u = ar[d], v = br[d]; ab = fmaf(u, v, ab); aa = fmaf(u, u, aa); bb = fmaf(v, v, bb); }
    ab = wave_sum(ab); aa = wave_sum(aa) + 1e-8f; bb = wave_sum(bb) + 1e-8f;
    const float den = sqrtf(aa * bb);
    const float cs = ab / den;
    if (da != nullptr) {
        const float sc = -weight / (float)B;
        for (int d = lane; d < D; d += 64) {
            float g = sc * (br[d] / den - cs * ar[d] / aa);
            float* q = da + (size_t)row * D + d;
            *q = grad_acc ? *q + g : g;
        }
    }
    if (lane == 0) atomicAdd(loss_out, weight * (1.f - cs) / (float)B);
}

// lucir.py:187-205.  one wave per row; rows with label >= num_old contribute nothing.
// loss = weight * sum_{hard rows} sum_{k<K} max(0, margin - gt + novel_k) / (hard_num*K)
// two launches: count (host-free: count kernel writes hard_count) then loss.
__global__ void count_hard_kernel(const int64_t* __restrict__ labels, int B, int num_old, int32_t* hard_count) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B && labels[i] < num_old) atomicAdd(hard_count, 1);
}
__global__ __launch_bounds__(256) void margin_rank_kernel(const float* __restrict__ scores, const int64_t* __restrict__ labels, int B,
                                                          int O, int num_old, int K, float margin, float weight,
                                                          const int32_t* __restrict__ hard_count, float* loss_out,
                                                          float* __restrict__ ds, int grad_acc) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const int hn = *hard_count;
    const int y = (int)labels[row];
    float* dr = ds ? ds + (size_t)row * O : nullptr;
    const bool hard = (y < num_old) && hn > 0;
    int sel[8];                 // chosen novel columns whose hinge is active (wave-uniform), -1 otherwise
#pragma unroll
    for (int k = 0; k < 8; ++k) sel[k] = -1;
    float li = 0.f, dgt = 0.f, sc = 0.f;
    if (hard) {
        const float* sr = scores + (size_t)row * O;
        const float gt = sr[y];
        sc = weight / (float)(hn * K);
        unsigned long long taken = 0ull;   // one bit per column slot owned by this lane
        // top-K novel scores by repeated wave arg-max (K = 2 in the shipped config); ties -> lower index
        for (int k = 0; k < K; ++k) {
            float bv = -INFINITY; int bi = 0x7fffffff;
            int slot = 0;
            for (int c = num_old + lane; c < O; c += 64, ++slot) {
                if ((taken >> slot) & 1ull) continue;
                float v = sr[c];
                if (v > bv) { bv = v; bi = c; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                float ov = __shfl_xor(bv, o, 64); int oi = __shfl_xor(bi, o, 64);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if (bi == 0x7fffffff) break;
            if (((bi - num_old) & 63) == lane) taken |= 1ull << ((bi - num_old) >> 6);
            float h = margin - gt + bv;
            if (h > 0.f) { li += h; dgt -= sc; sel[k] = bi; }
        }
    }
    if (dr) {   // every column written exactly once by its owner lane
        for (int c = lane; c < O; c += 64) {
            float g = 0.f;
            if (hard) {
                if (c == y) g += dgt;
#pragma unroll
                for (int k = 0; k < 8; ++k) if (sel[k] == c) g += sc;
            }
            if (grad_acc) { if (g != 0.f) dr[c] += g; }
            else dr[c] = g;
        }
    }
    if (hard && lane == 0) atomicAdd(loss_out, sc * li);
}

// out[r] = x[r] / max(||x[r]||, eps)
__global__ __launch_bounds__(256) void l2_normalize_kernel(const float* __restrict__ x, float* __restrict__ out, int R, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) { float v = x[(size_t)row * D + d]; s = fmaf(v, v, s); }
    s = wave_sum(s);
    const float inv = 1.f / sqrtf(s);
    for (int d = lane; d < D; d += 64) out[(size_t)row * D + d] = x[(size_t)row * D + d] * inv;
}

// ------------------------------------------------------------------------------------------ NCM
__global__ __launch_bounds__(256) void ncm_kernel(const float* __restrict__ f, const float* __restrict__ means, int B, int M, int D,
                                                  int64_t* pred) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    float bv = INFINITY; int bi = 0;
    for (int m = 0; m < M; ++m) {
        float s = 0.f;
        for (int d = lane; d < D; d += 64) { float t = f[(size_t)row * D + d] - means[(size_t)m * D + d]; s = fmaf(t, t, s); }
        s = wave_sum(s);
        if (s < bv) { bv = s; bi = m; }     // first minimal index, as torch.argmin
    }
    if (lane == 0) pred[row] = bi;
}

// herding: one block per class; ws = [mu(D) | run(D) | taken(n)].  `fl` (nullable): the class's rows staged in LDS at a pitch of D + 1 floats
// (row i of thread i, coordinate d: bank (i + d) % 32 -- conflict-free), the arithmetic and its order are those of the global-memory form
__device__ __forceinline__ void herding_body(const float* __restrict__ f, int n, int D, int m, int32_t* chosen, float* ws, float* fl) {
    __shared__ float sv[256];
    __shared__ int si[256];
    float* mu = ws; float* run = ws + D; float* taken = ws + 2 * D;
    const int tid = threadIdx.x;
    const int P = fl != nullptr ? D + 1 : D;
    if (fl != nullptr) {
        for (int e = tid; e < n * D; e += 256) { const int i = e / D, d = e - i * D; fl[i * P + d] = f[e]; }
        __syncthreads();
        f = fl;
    }
    for (int d = tid; d < D; d += 256) {
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += f[(size_t)i * P + d];
        mu[d] = s / (float)n;
        run[d] = 0.f;
    }
    for (int i = tid; i < n; i += 256) taken[i] = 0.f;
    for (int k = n + tid; k < m; k += 256) chosen[k] = -1;      // fewer rows than picks: the tail stays marked
    __syncthreads();
    for (int k = 0; k < m && k < n; ++k) {
        float bv = INFINITY; int bi = 0x7fffffff;
        const float inv = 1.f / (float)(k + 1);
        for (int i = tid; i < n; i += 256) {
            float s = 0.f;
            // a selected row is "removed" by +1e6 on every coordinate (linearherdingbuffer.py:160)
            const float off = taken[i] * 1e6f;
            for (int d = 0; d < D; ++d) { float t = mu[d] - (f[(size_t)i * P + d] + off + run[d]) * inv; s = fmaf(t, t, s); }
            if (s < bv) { bv = s; bi = i; }
        }
        sv[tid] = bv; si[tid] = bi;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) {
                float ov = sv[tid + o]; int oi = si[tid + o];
                if (ov < sv[tid] || (ov == sv[tid] && oi < si[tid])) { sv[tid] = ov; si[tid] = oi; }
            }
            __syncthreads();
        }
        const int best = si[0];
        if (tid == 0) { chosen[k] = best; }
        for (int d = tid; d < D; d += 256) run[d] += f[(size_t)best * P + d] + taken[best] * 1e6f;
        __syncthreads();
        if (tid == 0) taken[best] += 1.f;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void herding_kernel(const float* __restrict__ f, int n, int D, int m, int32_t* chosen, float* ws, int use_lds) {
    extern __shared__ __attribute__((aligned(16))) float herd_lds[];
    herding_body(f, n, D, m, chosen, ws, use_lds ? herd_lds : nullptr);
}

// every class of a task in ONE launch (round 4): class c = rows [offsets[c], offsets[c + 1]) of f, its picks (class-local indices) in
// chosen[c * m ...], its scratch at ws + 2 D c + offsets[c].  The 50 classes of a B50 task: one launch of 50 blocks instead of 50 launches
// of one block (0.86 ms each: tools / bench.py --workload herding_b50)
__global__ __launch_bounds__(256) void herding_batched_kernel(const float* __restrict__ f, const int32_t* __restrict__ offsets, int D, int m, int32_t* chosen, float* ws,
                                                              int lds_rows) {
    extern __shared__ __attribute__((aligned(16))) float herd_lds[];
    const int c = blockIdx.x;
    const int o0 = offsets[c], n = offsets[c + 1] - o0;
    if (n <= 0) { for (int k = threadIdx.x; k < m; k += 256) chosen[(size_t)c * m + k] = -1; return; }
    herding_body(f + (size_t)o0 * D, n, D, m, chosen + (size_t)c * m, ws + (size_t)2 * D * c + o0, n <= lds_rows ? herd_lds : nullptr);
}

}  // namespace

#define ST ((hipStream_t)stream)
static int zero_scalar(void* p, size_t bytes, hipStream_t st) {
    if (hipMemsetAsync(p, 0, bytes, st) != hipSuccess) { clhip_set_error("hipMemsetAsync failed"); return CLHIP_EHIP; }
    return CLHIP_OK;
}

extern "C" int clhip_linear_fwd(const float* x, const float* w, const float* b, float* out, int B, int D, int O, void* stream) {
    CLHIP_CHECK_ARG(x && w && out && B > 0 && D > 0 && O > 0);
    CLHIP_CHECK_ARG(D % 4 == 0 || D < 4 || true);
    int waves = B * ((O + 3) / 4);
    hipLaunchKernelGGL(linear_fwd_kernel, dim3((waves + 3) / 4), dim3(256), 0, ST, x, w, b, out, B, D, O);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_linear_bwd(const float* x, const float* w, const float* dout, float* dx, float* dw, float* db, int B, int D,
                                int O, int accumulate, void* stream) {
    CLHIP_CHECK_ARG(x && w && dout && dw && B > 0 && D > 0 && O > 0);
    static const bool split = clhip_cfg("LINEAR_BWD_SPLIT") != nullptr && atoi(clhip_cfg("LINEAR_BWD_SPLIT")) != 0;
    if (dx && !split) {
        const int ndx = (B * D + 255) / 256, ndwx = (D + 63) / 64;
        hipLaunchKernelGGL(linear_bwd_fused_kernel, dim3(ndx + ndwx * O), dim3(256), 0, ST, dout, w, x, dx, dw, db, B, D, O, accumulate, ndx, ndwx);
        CLHIP_LAUNCH_CHECK();
        return CLHIP_OK;
    }
    if (dx) {
        hipLaunchKernelGGL(linear_bwd_dx_kernel, dim3((B * D + 255) / 256), dim3(256), 0, ST, dout, w, dx, B, D, O, 0);
        CLHIP_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(linear_bwd_dw_kernel, dim3((D + 63) / 64, O), dim3(256), 0, ST, dout, x, dw, db, B, D, O, accumulate);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_ce_window(const float* logits, const int64_t* labels, int B, int O, int lo, int hi, int pred_lo, int pred_hi, float weight,
                               float* loss_out, int loss_accumulate, float* dlogits, int grad_accumulate, int64_t* pred,
                               int32_t* correct, void* stream) {
    CLHIP_CHECK_ARG(logits && labels && loss_out && B > 0 && O > 0 && lo >= 0 && hi > lo && hi <= O);
    CLHIP_CHECK_ARG(pred_lo >= 0 && pred_hi > pred_lo && pred_hi <= O);
    static const bool rows_off = clhip_cfg("CE_ROWS") != nullptr && atoi(clhip_cfg("CE_ROWS")) == 0;      // A/B switch: the one-wave-per-row form
    if (B <= 512 && (O <= 128 || (O <= 256 && B <= 256)) && !rows_off) {      // (16 columns x 8 rows per lane would spill)
#define CE_ROWS(NC, RPG) hipLaunchKernelGGL((ce_rows_kernel<NC, RPG>), dim3(G), dim3(1024), 0, ST, logits, labels, B, O, lo, hi, pred_lo, pred_hi, weight, loss_out, \
                                            dlogits, grad_accumulate, pred, correct, loss_accumulate, slot)
        // 64 rows per workgroup from 128 rows up (at most 8 workgroups); a launch takes the next slot of a small ring, so launches in flight on
        // different streams do not share one
        static const bool one_wg = clhip_cfg("CE_ONE_WG") != nullptr && atoi(clhip_cfg("CE_ONE_WG")) != 0;
        // one ring PER DEVICE (a process that drives several GPUs must not hand device 0's memory to a kernel on device 1: ADVICE r3).  The first
        // multi-workgroup call on a device allocates it -- a synchronous hipMalloc + hipMemset, which a stream capture cannot contain: a capturing
        // stream that finds no ring yet takes the one-workgroup form instead (a different grouping of the same fixed-order sums; the trainer's
        // two eager warm steps allocate the ring before any capture, so a replayed step and an eager one group alike)
        constexpr int kMaxDev = 16;
        static std::atomic<CeSlot*> rings[kMaxDev];
        static std::atomic<unsigned> next{0};
        int G = (B >= 128 && !one_wg) ? (B + 63) / 64 : 1;
        if (G > 8) G = 8;
        int devid = 0;
        (void)hipGetDevice(&devid);
        if (devid < 0 || devid >= kMaxDev) G = 1;
        CeSlot* ring = G > 1 ? rings[devid].load(std::memory_order_acquire) : nullptr;
        if (G > 1 && ring == nullptr) {
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            (void)hipStreamIsCapturing(ST, &cap);
            if (cap != hipStreamCaptureStatusNone) G = 1;
            else {
                static std::mutex mu;
                std::lock_guard<std::mutex> lk(mu);
                ring = rings[devid].load(std::memory_order_acquire);
                if (ring == nullptr) {
                    CeSlot* r = nullptr;
                    if (hipMalloc(&r, 64 * sizeof(CeSlot)) != hipSuccess || hipMemset(r, 0, 64 * sizeof(CeSlot)) != hipSuccess) { clhip_set_error("clhip_ce_window: cannot allocate the partial-sum ring"); return CLHIP_EHIP; }
                    rings[devid].store(r, std::memory_order_release);
                    ring = r;
                }
            }
        }
        CeSlot* slot = G > 1 ? ring + (next.fetch_add(1) & 63u) : nullptr;
        const int rpg = (B + 64 * G - 1) / (64 * G);
        if (O <= 128) { if (rpg <= 1) CE_ROWS(8, 1); else if (rpg <= 2) CE_ROWS(8, 2); else if (rpg <= 4) CE_ROWS(8, 4); else CE_ROWS(8, 8); }
        else { if (rpg <= 1) CE_ROWS(16, 1); else if (rpg <= 2) CE_ROWS(16, 2); else CE_ROWS(16, 4); }
#undef CE_ROWS
        CLHIP_LAUNCH_CHECK();
        return CLHIP_OK;
    }
    if (B <= 512) {
        hipLaunchKernelGGL(ce_slice_kernel<true>, dim3(1), dim3(1024), 0, ST, logits, labels, B, O, lo, hi, pred_lo, pred_hi, weight, loss_out,
                           dlogits, grad_accumulate, pred, correct, loss_accumulate);
        CLHIP_LAUNCH_CHECK();
        return CLHIP_OK;
    }
    if (!loss_accumulate) { if (int e = zero_scalar(loss_out, 4, ST)) return e; }
    if (correct) { if (int e = zero_scalar(correct, 4, ST)) return e; }
    hipLaunchKernelGGL(ce_slice_kernel<false>, dim3((B + 3) / 4), dim3(256), 0, ST, logits, labels, B, O, lo, hi, pred_lo, pred_hi, weight, loss_out,
                       dlogits, grad_accumulate, pred, correct, loss_accumulate);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_ce_slice(const float* logits, const int64_t* labels, int B, int O, int lo, int hi, int pred_hi, float weight,
                              float* loss_out, int loss_accumulate, float* dlogits, int grad_accumulate, int64_t* pred,
                              int32_t* correct, void* stream) {
    return clhip_ce_window(logits, labels, B, O, lo, hi, 0, pred_hi, weight, loss_out, loss_accumulate, dlogits, grad_accumulate, pred, correct, stream);
}

extern "C" int clhip_kd_loss(const float* pred, int pred_stride, const float* soft, int soft_stride, int B, int k, float T,
                             float weight, float* loss_out, int loss_accumulate, float* dpred, int grad_accumulate, void* stream) {
    CLHIP_CHECK_ARG(pred && soft && loss_out && B > 0 && k > 0 && pred_stride >= k && soft_stride >= k && T > 0.f);
    if (!loss_accumulate) { if (int e = zero_scalar(loss_out, 4, ST)) return e; }
    hipLaunchKernelGGL(kd_kernel, dim3((B + 3) / 4), dim3(256), 0, ST, pred, pred_stride, soft, soft_stride, B, k, 1.f / T, weight,
                       loss_out, dpred, grad_accumulate);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_cosine_linear_fwd(const float* x, const float* w, float* out, float* xnorm, float* wnorm, int B, int D, int O,
                                       void* stream) {
    CLHIP_CHECK_ARG(x && w && out && xnorm && wnorm && B > 0 && D > 0 && O > 0);
    const size_t lds = ((size_t)(O + kCosRows) * (D + 1) + O + kCosRows) * sizeof(float);
    if (lds <= 60 * 1024) {           // every in-scope head (64 x 100, 512 x 100 needs 224 KB: the three-launch form below)
        hipLaunchKernelGGL(cosine_fwd_fused_kernel, dim3((B + kCosRows - 1) / kCosRows), dim3(256), lds, ST, x, w, xnorm, wnorm, out, B, D, O);
        CLHIP_LAUNCH_CHECK();
        return CLHIP_OK;
    }
    hipLaunchKernelGGL(row_norm_kernel, dim3((B + 3) / 4), dim3(256), 0, ST, x, xnorm, B, D);
    hipLaunchKernelGGL(row_norm_kernel, dim3((O + 3) / 4), dim3(256), 0, ST, w, wnorm, O, D);
    hipLaunchKernelGGL(cosine_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, ST, x, w, xnorm, wnorm, out, B, D, O);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_cosine_linear_bwd(const float* x, const float* w, const float* out, const float* xnorm, const float* wnorm,
                                       const float* dout, float* dx, float* dw, int B, int D, int O, int accumulate, void* stream) {
    CLHIP_CHECK_ARG(x && w && out && xnorm && wnorm && dout && B > 0 && D > 0 && O > 0);
    if (dx) hipLaunchKernelGGL(cosine_bwd_dx_kernel, dim3((B * D + 255) / 256), dim3(256), 0, ST, x, w, out, xnorm, wnorm, dout, dx, B, D, O, 0);
    if (dw) hipLaunchKernelGGL(cosine_bwd_dw_rows_kernel, dim3(O), dim3(256), 0, ST, x, w, out, xnorm, wnorm, dout, dw, B, D, O, accumulate);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_cos_embed_loss(const float* a, const float* b, int B, int D, float weight, float* loss_out, int loss_accumulate,
                                    float* da, int grad_accumulate, void* stream) {
    CLHIP_CHECK_ARG(a && b && loss_out && B > 0 && D > 0);
    if (!loss_accumulate) { if (int e = zero_scalar(loss_out, 4, ST)) return e; }
    hipLaunchKernelGGL(cos_embed_kernel, dim3((B + 3) / 4), dim3(256), 0, ST, a, b, B, D, weight, loss_out, da, grad_accumulate);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_margin_rank_loss(const float* scores, const int64_t* labels, int B, int O, int num_old, int K, float margin,
                                      float weight, float* loss_out, int loss_accumulate, float* dscores, int grad_accumulate,
                                      int32_t* hard_count, void* stream) {
    CLHIP_CHECK_ARG(scores && labels && loss_out && hard_count && B > 0 && O > num_old && num_old > 0 && K >= 1 && K <= O - num_old);
    CLHIP_CHECK_ARG(O - num_old <= 64 * 64 && K <= 8);
    if (!loss_accumulate) { if (int e = zero_scalar(loss_out, 4, ST)) return e; }
    if (int e = zero_scalar(hard_count, 4, ST)) return e;
    hipLaunchKernelGGL(count_hard_kernel, dim3((B + 255) / 256), dim3(256), 0, ST, labels, B, num_old, hard_count);
    hipLaunchKernelGGL(margin_rank_kernel, dim3((B + 3) / 4), dim3(256), 0, ST, scores, labels, B, O, num_old, K, margin, weight,
                       hard_count, loss_out, dscores, grad_accumulate);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_ncm_classify(const float* feats, const float* means, int B, int M, int D, int64_t* pred, void* stream) {
    CLHIP_CHECK_ARG(feats && means && pred && B > 0 && M > 0 && D > 0);
    hipLaunchKernelGGL(ncm_kernel, dim3((B + 3) / 4), dim3(256), 0, ST, feats, means, B, M, D, pred);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_herding_select(const float* feats, int n, int D, int m, int32_t* chosen, float* ws, void* stream) {
    CLHIP_CHECK_ARG(feats && chosen && ws && n > 0 && D > 0 && m > 0);
    // the class's rows in LDS when they fit (500 x 64 features: 130 KB): every pick re-reads all of them
    const size_t lds = (size_t)n * (D + 1) * sizeof(float);
    const bool in_lds = lds <= 150 * 1024;
    if (in_lds) {
        int dev = 0;                                     // function attributes are per device (a process may drive several)
        (void)hipGetDevice(&dev);
        static size_t attr[16] = {0};
        if (dev < 0 || dev >= 16 || lds > attr[dev]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(herding_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) { clhip_set_error("clhip_herding_select: cannot reserve 150 KB of LDS on device %d", dev); return CLHIP_EHIP; }
            if (dev >= 0 && dev < 16) attr[dev] = 150 * 1024;
        }
    }
    hipLaunchKernelGGL(herding_kernel, dim3(1), dim3(256), in_lds ? lds : 0, ST, feats, n, D, m, chosen, ws, in_lds ? 1 : 0);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_herding_select_batched(const float* feats, const int32_t* offsets, int n_classes, int max_rows, int D, int m, int32_t* chosen, float* ws,
                                            void* stream) {
    CLHIP_CHECK_ARG(feats && offsets && chosen && ws && n_classes > 0 && max_rows > 0 && D > 0 && m > 0);
    int lds_rows = (int)((150 * 1024) / ((size_t)(D + 1) * sizeof(float)));
    if (lds_rows > max_rows) lds_rows = max_rows;
    const size_t lds = (size_t)lds_rows * (D + 1) * sizeof(float);
    int dev = 0;
    (void)hipGetDevice(&dev);
    static size_t attr[16] = {0};
    if (dev < 0 || dev >= 16 || lds > attr[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(herding_batched_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) { clhip_set_error("clhip_herding_select_batched: cannot reserve 150 KB of LDS on device %d", dev); return CLHIP_EHIP; }
        if (dev >= 0 && dev < 16) attr[dev] = 150 * 1024;
    }
    hipLaunchKernelGGL(herding_batched_kernel, dim3(n_classes), dim3(256), lds, ST, feats, offsets, D, m, chosen, ws, lds_rows);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_l2_normalize_rows(const float* x, float* out, int R, int D, void* stream) {
    CLHIP_CHECK_ARG(x && out && R > 0 && D > 0);
    hipLaunchKernelGGL(l2_normalize_kernel, dim3((R + 3) / 4), dim3(256), 0, ST, x, out, R, D);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}
