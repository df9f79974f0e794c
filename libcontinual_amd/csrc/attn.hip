// attn.hip -- multi-head self-attention of the ViT path, forward and backward, directly on the packed qkv
// activation (reference core/model/backbone/transformer.py:169-197 and :239-274).
//
// The reference reshapes qkv to [3,B,H,N,d], materialises the [B,H,N,N] score matrix, softmaxes it in a separate
// pass, multiplies by V and transposes back.  Here one workgroup owns one (batch, head): K and V (and for the
// backward also Q and dO) of that head are staged ONCE in LDS straight from the [B*N, 3D] qkv buffer (row stride
// 3D, head offset h*64 -- no permute), the score tile never leaves registers, and the output is written in the
// [B*N, D] layout the projection GEMM consumes.  N <= 256 (197 / 222 tokens on this path), head dim 64.
//
// MFMA formulation (wave64, v_mfma_f32_16x16x32_bf16; D layout col = lane&15, rows = (lane>>4)*4+e):
//   fwd  S^T[key,q] = K . Q^T         (A = K rows, B = Q rows: both 16-byte LDS / global reads)
//        O^T[d,q]   = V^T . P^T       (A = V^T through ds_read_b64_tr_b16 on row-major V, B = P^T packed from the
//                                      S^T accumulators of two key tiles: no shuffle, the k-slot order of an MFMA
//                                      is free as long as A and B agree)
//   bwd  phase A (wave <- query tiles): dQ^T[d,q]  = K^T . dS^T
//        phase B (wave <- key tiles):   dV^T[d,k]  = dO^T . P,   dK^T[d,k] = Q^T . dS
//        with P = exp(S*scale - lse), dS = P * (dP - rowsum(dO*O)), dP = dO . V^T; lse is saved by the forward.
// fp32 (parity mode) and odd head sizes use the generic one-wave-per-row kernels at the end of the file.
#include "common.h"

namespace {

struct AttnParams {
    const void* qkv; void* out; float* lse;
    const void* dout; void* dqkv; float* dsum;     // backward only (dsum: [B,H,N] scratch, generic path)
    int B, N, H, D;
    float scale;
};

constexpr int KP = 160;      // LDS pitch (bytes) of a 64-element bf16 row: ds_read_b128 and tr reads are conflict-free

__device__ __forceinline__ uint4 tr8(const char* base, int addr, int second) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + addr));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + addr + second));
    uint2 l = __builtin_bit_cast(uint2, lo), h = __builtin_bit_cast(uint2, hi);
    return make_uint4(l.x, l.y, h.x, h.y);
}
__device__ __forceinline__ f32x4 mfma_bf16(uint4 a, uint4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ uint4 pack8(const f32x4& a, const f32x4& b) {
    return make_uint4(pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3]));
}
__device__ __forceinline__ uint4 ldsq(const char* base, int off) { return *reinterpret_cast<const uint4*>(base + off); }

// stage rows [0, NP2) of one head's 64-wide slice (global row stride `ld` elements) into LDS, zero beyond N
__device__ __forceinline__ void stage_rows(char* dst, const bf16_t* src, size_t ld, int N, int NP2) {
    for (int idx = threadIdx.x; idx < NP2 * 8; idx += blockDim.x) {
        const int row = idx >> 3, c = idx & 7;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < N) v = *reinterpret_cast<const uint4*>(src + (size_t)row * ld + c * 8);
        *reinterpret_cast<uint4*>(dst + row * KP + c * 16) = v;
    }
}

// NKT = number of 16-key tiles (compile time: 13 / 14 for N = 197 / 222; 0 = run-time count, up to 16).  Instruction diet
// (PMC of the first version: 18 VALU per MFMA, the kernel was issue-bound): the softmax scale is folded into the exponent
// (one v_fma + one v_exp per score, exp2 domain), only the last key tile is masked, P stays unnormalised in bf16 and 1/sum
// is applied to the 16 output accumulators instead of the 52 probabilities.
template <int NKT>
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
    const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
    const int N = p.N, D = p.D;
    const int nKT = NKT > 0 ? NKT : (N + 15) >> 4;
    constexpr int KTMAX = NKT > 0 ? NKT : 16;
    const int NP2 = ((nKT + 1) & ~1) * 16;
    const size_t ld = 3 * (size_t)D;
    const bf16_t* base = static_cast<const bf16_t*>(p.qkv) + (size_t)b * N * ld + h * 64;
    char* Ks = smem;
    char* Vs = smem + NP2 * KP;
    stage_rows(Ks, base + D, ld, N, NP2);
    stage_rows(Vs, base + 2 * D, ld, N, NP2);
    __syncthreads();
    const float c = p.scale * 1.4426950408889634f;          // scores -> exp2 domain
    const int last0 = (nKT - 1) * 16 + g * 4;               // first key this lane holds in the last tile

    for (int qt = wave; qt < nKT; qt += 4) {
        const int qrow = qt * 16 + l15;
        const bool qok = qrow < N;
        uint4 qf[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
            qf[kk] = qok ? *reinterpret_cast<const uint4*>(base + (size_t)qrow * ld + (g + 4 * kk) * 8) : make_uint4(0, 0, 0, 0);
        f32x4 s[KTMAX + 1];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < KTMAX; ++kt) {
            s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (NKT > 0 || kt < nKT) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) s[kt] = mfma_bf16(ldsq(Ks, (kt * 16 + l15) * KP + (g + 4 * kk) * 16), qf[kk], s[kt]);
                if (kt == nKT - 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (last0 + e >= N) s[kt][e] = -INFINITY;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) mx = fmaxf(mx, s[kt][e]);
            }
        }
        s[KTMAX] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mc = mx * c;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < KTMAX; ++kt) {
            if (NKT > 0 || kt < nKT) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float e_ = __builtin_amdgcn_exp2f(fmaf(s[kt][e], c, -mc)); s[kt][e] = e_; sum += e_; }
            }
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        if (g == 0 && qok && p.lse) p.lse[((size_t)b * p.H + h) * N + qrow] = mx * p.scale + __logf(sum);
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < (KTMAX + 1) / 2; ++ks) {
            if (NKT > 0 || 2 * ks < nKT) {
                const uint4 pb = pack8(s[2 * ks], s[2 * ks + 1]);           // tiles >= nKT are zero
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const uint4 vt = tr8(Vs, (2 * ks * 16 + g * 4 + (l15 >> 2)) * KP + (dt * 16 + (l15 & 3) * 4) * 2, 16 * KP);
                    o[dt] = mfma_bf16(vt, pb, o[dt]);
                }
            }
        }
        if (qok) {
            const float inv = 1.0f / sum;
            bf16_t* orow = static_cast<bf16_t*>(p.out) + ((size_t)b * N + qrow) * D + h * 64 + g * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<uint2*>(orow + dt * 16) = make_uint2(pack_bf16x2(o[dt][0] * inv, o[dt][1] * inv), pack_bf16x2(o[dt][2] * inv, o[dt][3] * inv));
        }
    }
}

constexpr int BWD_WAVES = 16;     // LDS (Q, K, V, dO of one head) limits the CU to one workgroup: give it 16 waves (4 per SIMD; 8 waves: 166 us, 16: 141 us)

__global__ __launch_bounds__(64 * BWD_WAVES) void attn_bwd_mfma_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
    const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
    const int N = p.N, D = p.D, nKT = (N + 15) >> 4, NP2 = ((nKT + 1) & ~1) * 16, nPair = NP2 >> 5;
    const size_t ld = 3 * (size_t)D;
    const bf16_t* base = static_cast<const bf16_t*>(p.qkv) + (size_t)b * N * ld + h * 64;
    const bf16_t* dob = static_cast<const bf16_t*>(p.dout) + (size_t)b * N * D + h * 64;
    const bf16_t* ob = static_cast<const bf16_t*>(p.out) + (size_t)b * N * D + h * 64;
    bf16_t* dqb = static_cast<bf16_t*>(p.dqkv) + (size_t)b * N * ld + h * 64;
    char* Qs = smem;
    char* Ks = Qs + NP2 * KP;
    char* Vs = Ks + NP2 * KP;
    char* Gs = Vs + NP2 * KP;                                  // dO
    float* lse_s = reinterpret_cast<float*>(Gs + NP2 * KP);    // [NP2]  lse * log2(e)  (+inf beyond N -> P = 0)
    float* dq_s = lse_s + NP2;                                 // [NP2]  rowsum(dO * O)
    stage_rows(Qs, base, ld, N, NP2);
    stage_rows(Ks, base + D, ld, N, NP2);
    stage_rows(Vs, base + 2 * D, ld, N, NP2);
    stage_rows(Gs, dob, D, N, NP2);
    for (int q = tid; q < NP2; q += 64 * BWD_WAVES) {
        float dsum = 0.f, l = INFINITY;
        if (q < N) {
            l = p.lse[((size_t)b * p.H + h) * N + q];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float x[8], y[8];
                load8<bf16_t>(dob + (size_t)q * D + c * 8, x);
                load8<bf16_t>(ob + (size_t)q * D + c * 8, y);
#pragma unroll
                for (int j = 0; j < 8; ++j) dsum += x[j] * y[j];
            }
        }
        lse_s[q] = l * 1.4426950408889634f;
        dq_s[q] = dsum;
    }
    __syncthreads();
    const float c = p.scale * 1.4426950408889634f;          // scores -> exp2 domain

    // ---- phase A: dQ.  wave <- query tile; per key-tile pair: S^T, dP^T (D layout: rows key g*4+e, col q l15)
    for (int qt = wave; qt < nKT; qt += BWD_WAVES) {
        const int qrow = qt * 16 + l15;
        uint4 qf[2], gf[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            qf[kk] = ldsq(Qs, qrow * KP + (g + 4 * kk) * 16);
            gf[kk] = ldsq(Gs, qrow * KP + (g + 4 * kk) * 16);
        }
        const float lq = lse_s[qrow], dq = dq_s[qrow];
        f32x4 acc[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < nPair; ++ks) {
            f32x4 ds[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int krow = (2 * ks + t) * 16 + l15;
                f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    s = mfma_bf16(ldsq(Ks, krow * KP + (g + 4 * kk) * 16), qf[kk], s);
                    dp = mfma_bf16(ldsq(Vs, krow * KP + (g + 4 * kk) * 16), gf[kk], dp);
                }
                const bool tail = (2 * ks + t) >= nKT - 1;               // only the last key tile holds keys >= N
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float pr = __builtin_amdgcn_exp2f(fmaf(s[e], c, -lq));
                    if (tail && (2 * ks + t) * 16 + g * 4 + e >= N) pr = 0.f;
                    ds[t][e] = pr * (dp[e] - dq);
                }
            }
            const uint4 db = pack8(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const uint4 kt_ = tr8(Ks, (2 * ks * 16 + g * 4 + (l15 >> 2)) * KP + (dt * 16 + (l15 & 3) * 4) * 2, 16 * KP);
                acc[dt] = mfma_bf16(kt_, db, acc[dt]);
            }
        }
        if (qrow < N) {
            bf16_t* r = dqb + (size_t)qrow * ld + g * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<uint2*>(r + dt * 16) = make_uint2(pack_bf16x2(acc[dt][0] * p.scale, acc[dt][1] * p.scale),
                                                                    pack_bf16x2(acc[dt][2] * p.scale, acc[dt][3] * p.scale));
        }
    }

    // ---- phase B: dK, dV.  wave <- key tile; per query-tile pair: S, dP (D layout: rows q g*4+e, col key l15)
    for (int kt = wave; kt < nKT; kt += BWD_WAVES) {
        const int krow = kt * 16 + l15;
        const bool kok = krow < N;
        uint4 kf[2], vf[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            kf[kk] = ldsq(Ks, krow * KP + (g + 4 * kk) * 16);
            vf[kk] = ldsq(Vs, krow * KP + (g + 4 * kk) * 16);
        }
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        for (int qs = 0; qs < nPair; ++qs) {
            f32x4 pr[2], ds[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int qr = (2 * qs + t) * 16 + l15;
                f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    s = mfma_bf16(ldsq(Qs, qr * KP + (g + 4 * kk) * 16), kf[kk], s);
                    dp = mfma_bf16(ldsq(Gs, qr * KP + (g + 4 * kk) * 16), vf[kk], dp);
                }
                const int q0 = (2 * qs + t) * 16 + g * 4;
                const float4 l4 = *reinterpret_cast<const float4*>(lse_s + q0);
                const float4 d4 = *reinterpret_cast<const float4*>(dq_s + q0);
                const float le[4] = {l4.x, l4.y, l4.z, l4.w}, de[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pe = kok ? __builtin_amdgcn_exp2f(fmaf(s[e], c, -le[e])) : 0.f;      // lse = +inf beyond N -> 0
                    pr[t][e] = pe;
                    ds[t][e] = pe * (dp[e] - de[e]);
                }
            }
            const uint4 pb = pack8(pr[0], pr[1]), db = pack8(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int a = (2 * qs * 16 + g * 4 + (l15 >> 2)) * KP + (dt * 16 + (l15 & 3) * 4) * 2;
                dv[dt] = mfma_bf16(tr8(Gs, a, 16 * KP), pb, dv[dt]);
                dk[dt] = mfma_bf16(tr8(Qs, a, 16 * KP), db, dk[dt]);
            }
        }
        if (kok) {
            bf16_t* r = dqb + (size_t)krow * ld + g * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                *reinterpret_cast<uint2*>(r + D + dt * 16) = make_uint2(pack_bf16x2(dk[dt][0] * p.scale, dk[dt][1] * p.scale),
                                                                        pack_bf16x2(dk[dt][2] * p.scale, dk[dt][3] * p.scale));
                *reinterpret_cast<uint2*>(r + 2 * D + dt * 16) = make_uint2(pack_bf16x2(dv[dt][0], dv[dt][1]), pack_bf16x2(dv[dt][2], dv[dt][3]));
            }
        }
    }
}


// Round 6, 197 / 222 tokens (NPAIR = 7).  Where the kernel above spends a 128 x 12-head launch (ablation: prologue alone 31 us -- it reads q, k, v, dO, O = 194 MB, HBM
// rate, but in TEN dependent round trips with one workgroup per CU and nothing beside it; + phase A 41 us; + phase B 63 us; PMC: matrix pipes 18 % busy, 59 % of the wave
// cycles parked in s_waitcnt, no bank conflicts; both phases sit at ~64 B/clk of LDS operand reads).  This form
//   * makes every global read of the prologue in ONE round trip (151 -> 137 us on the evidence box);
//   * drops the masks of keys >= N: a padded key's K and V rows are zero in LDS, so whatever finite dS it gets multiplies zeros in dQ (phase A), and in phase B it only
//     feeds its own output column, which is never stored (14 + 8 compare / select instructions per tile pair).  Padded QUERIES still vanish through lse = +inf;
//   * requests the next tile's fragments before this tile's arithmetic.
// The last two measure nothing by themselves (the phases are LDS-bandwidth bound: a two-tiles-per-wave variant that halves the LDS bytes per MFMA needs > 128 registers,
// i.e. eight waves instead of sixteen, and ran 167 us).  Same MFMAs on the same operands for every stored element: bit-identical to the kernel above (tests).
template <int NPAIR>
__global__ __launch_bounds__(64 * BWD_WAVES) void attn_bwd_mfma3_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
    const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
    const int N = p.N, D = p.D, nKT = (N + 15) >> 4;
    constexpr int NP2 = NPAIR * 32, nPair = NPAIR;
    const size_t ld = 3 * (size_t)D;
    const bf16_t* base = static_cast<const bf16_t*>(p.qkv) + (size_t)b * N * ld + h * 64;
    const bf16_t* dob = static_cast<const bf16_t*>(p.dout) + (size_t)b * N * D + h * 64;
    const bf16_t* ob = static_cast<const bf16_t*>(p.out) + (size_t)b * N * D + h * 64;
    bf16_t* dqb = static_cast<bf16_t*>(p.dqkv) + (size_t)b * N * ld + h * 64;
    char* Qs = smem;
    char* Ks = Qs + NP2 * KP;
    char* Vs = Ks + NP2 * KP;
    char* Gs = Vs + NP2 * KP;                                  // dO
    float* lse_s = reinterpret_cast<float*>(Gs + NP2 * KP);    // [NP2]  lse * log2(e)  (+inf beyond N -> P = 0)
    float* dq_s = lse_s + NP2;                                 // [NP2]  rowsum(dO * O)
    // every global read of the prologue in ONE round trip (the kernel above makes ten: four staging loops of two trips each, then the lse / dO / O rows; with one
    // workgroup per CU nothing overlaps them): the four tiles' chunks and, for the thread that owns a query row, its lse and O row
    constexpr int CH = (NP2 * 8 + 64 * BWD_WAVES - 1) / (64 * BWD_WAVES);
    uint4 sq[CH], sk[CH], sv[CH], sg[CH], orow[8];
    float lraw = INFINITY;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int idx = tid + i * 64 * BWD_WAVES, row = idx >> 3, cc = idx & 7;
        sq[i] = sk[i] = sv[i] = sg[i] = make_uint4(0, 0, 0, 0);
        if (idx < NP2 * 8 && row < N) {
            sq[i] = *reinterpret_cast<const uint4*>(base + (size_t)row * ld + cc * 8);
            sk[i] = *reinterpret_cast<const uint4*>(base + D + (size_t)row * ld + cc * 8);
            sv[i] = *reinterpret_cast<const uint4*>(base + 2 * D + (size_t)row * ld + cc * 8);
            sg[i] = *reinterpret_cast<const uint4*>(dob + (size_t)row * D + cc * 8);
        }
    }
    if (tid < N) {
        lraw = p.lse[((size_t)b * p.H + h) * N + tid];
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) orow[cc] = *reinterpret_cast<const uint4*>(ob + (size_t)tid * D + cc * 8);
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int idx = tid + i * 64 * BWD_WAVES, row = idx >> 3, cc = idx & 7;
        if (idx < NP2 * 8) {
            *reinterpret_cast<uint4*>(Qs + row * KP + cc * 16) = sq[i];
            *reinterpret_cast<uint4*>(Ks + row * KP + cc * 16) = sk[i];
            *reinterpret_cast<uint4*>(Vs + row * KP + cc * 16) = sv[i];
            *reinterpret_cast<uint4*>(Gs + row * KP + cc * 16) = sg[i];
        }
    }
    __syncthreads();
    if (tid < NP2) {
        float dsum = 0.f;
        if (tid < N) {
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) {                     // (the element order of the kernel above: the sums agree bit for bit)
                const uint4 xg = ldsq(Gs, tid * KP + cc * 16);
                const unsigned xw[4] = {xg.x, xg.y, xg.z, xg.w}, yw[4] = {orow[cc].x, orow[cc].y, orow[cc].z, orow[cc].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    dsum += __uint_as_float(xw[j] << 16) * __uint_as_float(yw[j] << 16);
                    dsum += __uint_as_float(xw[j] & 0xffff0000u) * __uint_as_float(yw[j] & 0xffff0000u);
                }
            }
        }
        lse_s[tid] = lraw * 1.4426950408889634f;
        dq_s[tid] = dsum;
    }
    __syncthreads();
    const float c = p.scale * 1.4426950408889634f;          // scores -> exp2 domain

    // ---- phase A: dQ.  wave <- query tile; per key-tile pair: S^T, dP^T (D layout: rows key g*4+e, col q l15).  The K / V fragments of the NEXT key tile and the
    //      transposed K fragments of this pair are requested before this tile's arithmetic (the rounds 2-5 loop waited for each small group of reads where it used it:
    //      nine LDS round trips per pair)
    auto frag = [&](const char* A, const char* B, int row, uint4 (&fa)[2], uint4 (&fb)[2]) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            fa[kk] = ldsq(A, row * KP + (g + 4 * kk) * 16);
            fb[kk] = ldsq(B, row * KP + (g + 4 * kk) * 16);
        }
    };
    for (int qt = wave; qt < nKT; qt += BWD_WAVES) {
        const int qrow = qt * 16 + l15;
        uint4 qf[2], gf[2];
        frag(Qs, Gs, qrow, qf, gf);
        const float lq = lse_s[qrow], dq = dq_s[qrow];
        f32x4 acc[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        uint4 k0[2], v0[2], k1[2], v1[2];
        frag(Ks, Vs, l15, k0, v0);
        auto tileA = [&](const uint4 (&kf)[2], const uint4 (&vf)[2], f32x4& ds) {
            f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                s = mfma_bf16(kf[kk], qf[kk], s);
                dp = mfma_bf16(vf[kk], gf[kk], dp);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) ds[e] = __builtin_amdgcn_exp2f(fmaf(s[e], c, -lq)) * (dp[e] - dq);
        };
#pragma unroll 1
        for (int ks = 0; ks < nPair; ++ks) {
            uint4 kt_[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) kt_[dt] = tr8(Ks, (2 * ks * 16 + g * 4 + (l15 >> 2)) * KP + (dt * 16 + (l15 & 3) * 4) * 2, 16 * KP);
            frag(Ks, Vs, (2 * ks + 1) * 16 + l15, k1, v1);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 ds[2];
            tileA(k0, v0, ds[0]);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < nPair) frag(Ks, Vs, (2 * ks + 2) * 16 + l15, k0, v0);
            __builtin_amdgcn_sched_barrier(0);
            tileA(k1, v1, ds[1]);
            const uint4 db = pack8(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) acc[dt] = mfma_bf16(kt_[dt], db, acc[dt]);
        }
        if (qrow < N) {
            bf16_t* r = dqb + (size_t)qrow * ld + g * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<uint2*>(r + dt * 16) = make_uint2(pack_bf16x2(acc[dt][0] * p.scale, acc[dt][1] * p.scale),
                                                                    pack_bf16x2(acc[dt][2] * p.scale, acc[dt][3] * p.scale));
        }
    }

    // ---- phase B: dK, dV.  wave <- key tile; per query-tile pair: S, dP (D layout: rows q g*4+e, col key l15); the same read-ahead
    for (int kt = wave; kt < nKT; kt += BWD_WAVES) {
        const int krow = kt * 16 + l15;
        const bool kok = krow < N;
        uint4 kf[2], vf[2];
        frag(Ks, Vs, krow, kf, vf);
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        uint4 q0[2], g0[2], q1[2], g1[2];
        frag(Qs, Gs, l15, q0, g0);
        auto tileB = [&](const uint4 (&qfr)[2], const uint4 (&gfr)[2], int qtile, f32x4& pr, f32x4& ds) {
            const float4 l4 = *reinterpret_cast<const float4*>(lse_s + qtile * 16 + g * 4);
            const float4 d4 = *reinterpret_cast<const float4*>(dq_s + qtile * 16 + g * 4);
            f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                s = mfma_bf16(qfr[kk], kf[kk], s);
                dp = mfma_bf16(gfr[kk], vf[kk], dp);
            }
            const float le[4] = {l4.x, l4.y, l4.z, l4.w}, de[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pe = __builtin_amdgcn_exp2f(fmaf(s[e], c, -le[e]));                  // lse = +inf beyond N -> 0
                pr[e] = pe;
                ds[e] = pe * (dp[e] - de[e]);
            }
        };
#pragma unroll 1
        for (int qs = 0; qs < nPair; ++qs) {
            const int a = (2 * qs * 16 + g * 4 + (l15 >> 2)) * KP + (l15 & 3) * 8;
            frag(Qs, Gs, (2 * qs + 1) * 16 + l15, q1, g1);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 pr[2], ds[2];
            tileB(q0, g0, 2 * qs, pr[0], ds[0]);
            __builtin_amdgcn_sched_barrier(0);
            if (qs + 1 < nPair) frag(Qs, Gs, (2 * qs + 2) * 16 + l15, q0, g0);
            uint4 gt[2], qt_[2];                                    // (the first half of the pair's transposed dO / Q fragments: requested behind the read-ahead, used last)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) { gt[dt] = tr8(Gs, a + dt * 32, 16 * KP); qt_[dt] = tr8(Qs, a + dt * 32, 16 * KP); }
            __builtin_amdgcn_sched_barrier(0);
            tileB(q1, g1, 2 * qs + 1, pr[1], ds[1]);
            const uint4 pb = pack8(pr[0], pr[1]), db = pack8(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dv[dt] = mfma_bf16(gt[dt], pb, dv[dt]);
                dk[dt] = mfma_bf16(qt_[dt], db, dk[dt]);
            }
#pragma unroll
            for (int dt = 2; dt < 4; ++dt) {
                dv[dt] = mfma_bf16(tr8(Gs, a + dt * 32, 16 * KP), pb, dv[dt]);
                dk[dt] = mfma_bf16(tr8(Qs, a + dt * 32, 16 * KP), db, dk[dt]);
            }
        }
        if (kok) {
            bf16_t* r = dqb + (size_t)krow * ld + g * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                *reinterpret_cast<uint2*>(r + D + dt * 16) = make_uint2(pack_bf16x2(dk[dt][0] * p.scale, dk[dt][1] * p.scale),
                                                                        pack_bf16x2(dk[dt][2] * p.scale, dk[dt][3] * p.scale));
                *reinterpret_cast<uint2*>(r + 2 * D + dt * 16) = make_uint2(pack_bf16x2(dv[dt][0], dv[dt][1]), pack_bf16x2(dv[dt][2], dv[dt][3]));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ generic path
// One wave per (batch, head, row); lane = key while scoring, lane = d while accumulating.  Used for fp32 (parity mode)
// and head sizes other than 64.  N <= 256, head dim <= 64.
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_generic_kernel(AttnParams p, int hd) {
    __shared__ float ps[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;                       // (b*H + h)*N + q
    if (row >= p.B * p.H * p.N) return;
    const int q = row % p.N, bh = row / p.N, h = bh % p.H, b = bh / p.H;
    const size_t ld = 3 * (size_t)p.D;
    const T* base = static_cast<const T*>(p.qkv) + (size_t)b * p.N * ld + h * hd;
    const T* qp = base + (size_t)q * ld;
    float s[4], mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int key = lane + 64 * i;
        s[i] = -INFINITY;
        if (key < p.N) {
            const T* kp = base + p.D + (size_t)key * ld;
            float a = 0.f;
            for (int d = 0; d < hd; ++d) a += Elem<T>::ld(qp + d) * Elem<T>::ld(kp + d);
            s[i] = a * p.scale;
        }
        mx = fmaxf(mx, s[i]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { s[i] = (lane + 64 * i) < p.N ? expf(s[i] - mx) : 0.f; sum += s[i]; }
    sum = wave_sum(sum);
#pragma unroll
    for (int i = 0; i < 4; ++i) ps[wave][lane + 64 * i] = s[i] / sum;
    if (lane == 0 && p.lse) p.lse[row] = mx + logf(sum);
    __builtin_amdgcn_wave_barrier();
    if (lane < hd) {
        float o = 0.f;
        for (int key = 0; key < p.N; ++key) o += ps[wave][key] * Elem<T>::ld(base + 2 * p.D + (size_t)key * ld + lane);
        Elem<T>::st(static_cast<T*>(p.out) + ((size_t)b * p.N + q) * p.D + h * hd + lane, o);
    }
}

// pass 1 (row = query): dsum[row] = sum_d dO*O; dQ.   pass 2 (row = key): dK, dV (deterministic, no atomics)
template <typename T, int PASS>
__global__ __launch_bounds__(256) void attn_bwd_generic_kernel(AttnParams p, int hd) {
    __shared__ float ps[4][256], ds_[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= p.B * p.H * p.N) return;
    const int r = row % p.N, bh = row / p.N, h = bh % p.H, b = bh / p.H;
    const size_t ld = 3 * (size_t)p.D;
    const T* base = static_cast<const T*>(p.qkv) + (size_t)b * p.N * ld + h * hd;
    const T* dob = static_cast<const T*>(p.dout) + (size_t)b * p.N * p.D + h * hd;
    const T* ob = static_cast<const T*>(p.out) + (size_t)b * p.N * p.D + h * hd;
    T* dqb = static_cast<T*>(p.dqkv) + (size_t)b * p.N * ld + h * hd;
    const float* lse = p.lse + (size_t)bh * p.N;
    float* dsum = p.dsum + (size_t)bh * p.N;
    if constexpr (PASS == 1) {
        float dd = lane < hd ? Elem<T>::ld(dob + (size_t)r * p.D + lane) * Elem<T>::ld(ob + (size_t)r * p.D + lane) : 0.f;
        dd = wave_sum(dd);
        if (lane == 0) dsum[r] = dd;
        const float l = lse[r];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = lane + 64 * i;
            float v = 0.f;
            if (key < p.N) {
                float a = 0.f, dp = 0.f;
                for (int d = 0; d < hd; ++d) {
                    a += Elem<T>::ld(base + (size_t)r * ld + d) * Elem<T>::ld(base + p.D + (size_t)key * ld + d);
                    dp += Elem<T>::ld(dob + (size_t)r * p.D + d) * Elem<T>::ld(base + 2 * p.D + (size_t)key * ld + d);
                }
                v = expf(a * p.scale - l) * (dp - dd);
            }
            ds_[wave][key] = v;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < hd) {
            float a = 0.f;
            for (int key = 0; key < p.N; ++key) a += ds_[wave][key] * Elem<T>::ld(base + p.D + (size_t)key * ld + lane);
            Elem<T>::st(dqb + (size_t)r * ld + lane, a * p.scale);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = lane + 64 * i;
            float pv = 0.f, dv = 0.f;
            if (q < p.N) {
                float a = 0.f, dp = 0.f;
                for (int d = 0; d < hd; ++d) {
                    a += Elem<T>::ld(base + (size_t)q * ld + d) * Elem<T>::ld(base + p.D + (size_t)r * ld + d);
                    dp += Elem<T>::ld(dob + (size_t)q * p.D + d) * Elem<T>::ld(base + 2 * p.D + (size_t)r * ld + d);
                }
                pv = expf(a * p.scale - lse[q]);
                dv = pv * (dp - dsum[q]);
            }
            ps[wave][q] = pv;
            ds_[wave][q] = dv;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < hd) {
            float ak = 0.f, av = 0.f;
            for (int q = 0; q < p.N; ++q) {
                ak += ds_[wave][q] * Elem<T>::ld(base + (size_t)q * ld + lane);
                av += ps[wave][q] * Elem<T>::ld(dob + (size_t)q * p.D + lane);
            }
            Elem<T>::st(dqb + p.D + (size_t)r * ld + lane, ak * p.scale);
            Elem<T>::st(dqb + 2 * p.D + (size_t)r * ld + lane, av);
        }
    }
}

bool force_generic() {
    static int v = -1;
    if (v < 0) { const char* e = clhip_cfg("ATTN_GENERIC"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}

int check(int B, int N, int H, int D, int dtype) {
    CLHIP_CHECK_ARG(B > 0 && H > 0 && N > 0 && N <= 256 && D % H == 0 && D / H <= 64 && D % 8 == 0);
    CLHIP_CHECK_ARG(dtype == CLHIP_BF16 || dtype == CLHIP_F32);
    return CLHIP_OK;
}

}  // namespace

extern "C" int clhip_attn_fwd(const void* qkv, void* out, float* lse, int B, int N, int H, int D, int dtype, void* stream) {
    CLHIP_CHECK_ARG(qkv && out);
    if (int rc = check(B, N, H, D, dtype)) return rc;
    const int hd = D / H;
    AttnParams p{qkv, out, lse, nullptr, nullptr, nullptr, B, N, H, D, 1.0f / sqrtf((float)hd)};
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == CLHIP_BF16 && hd == 64 && !force_generic()) {
        const int NP2 = ((((N + 15) >> 4) + 1) & ~1) * 16;
        const size_t smem = 2 * (size_t)NP2 * KP;
        static bool done = false;
        if (!done) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_mfma_kernel<13>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * KP);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_mfma_kernel<14>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * KP);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_mfma_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * KP);
            done = true;
        }
        const int nkt = (N + 15) >> 4;
        if (nkt == 13) hipLaunchKernelGGL(attn_fwd_mfma_kernel<13>, dim3(B * H), dim3(256), smem, s, p);
        else if (nkt == 14) hipLaunchKernelGGL(attn_fwd_mfma_kernel<14>, dim3(B * H), dim3(256), smem, s, p);
        else hipLaunchKernelGGL(attn_fwd_mfma_kernel<0>, dim3(B * H), dim3(256), smem, s, p);
    } else {
        const int rows = B * H * N;
        if (dtype == CLHIP_BF16) hipLaunchKernelGGL(attn_fwd_generic_kernel<bf16_t>, dim3((rows + 3) / 4), dim3(256), 0, s, p, hd);
        else hipLaunchKernelGGL(attn_fwd_generic_kernel<float>, dim3((rows + 3) / 4), dim3(256), 0, s, p, hd);
    }
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

extern "C" int clhip_attn_bwd(const void* qkv, const void* out, const float* lse, const void* dout, void* dqkv, float* dsum_ws, int B, int N, int H,
                              int D, int dtype, void* stream) {
    CLHIP_CHECK_ARG(qkv && out && lse && dout && dqkv);
    if (int rc = check(B, N, H, D, dtype)) return rc;
    const int hd = D / H;
    AttnParams p{qkv, const_cast<void*>(out), const_cast<float*>(lse), dout, dqkv, dsum_ws, B, N, H, D, 1.0f / sqrtf((float)hd)};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int NP2 = ((((N + 15) >> 4) + 1) & ~1) * 16;
    const size_t smem = 4 * (size_t)NP2 * KP + 2 * NP2 * sizeof(float);
    constexpr size_t kLdsMax = 160 * 1024;       // Q, K, V, dO of one head must fit the CU's LDS (N <= 240); else generic path
    if (dtype == CLHIP_BF16 && hd == 64 && smem <= kLdsMax && !force_generic()) {
        static bool done = false;
        if (!done) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_mfma3_kernel<7>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax);
            done = true;
        }
        // ATTN_BWD=1: the rounds 2-5 kernel at every token count (looked up per call: the tests compare the two bit for bit); default: the round-6 form at 197 / 222 tokens
        const char* v = clhip_cfg("ATTN_BWD");
        if ((v != nullptr && atoi(v) == 1) || NP2 != 224) hipLaunchKernelGGL(attn_bwd_mfma_kernel, dim3(B * H), dim3(64 * BWD_WAVES), smem, s, p);
        else hipLaunchKernelGGL(attn_bwd_mfma3_kernel<7>, dim3(B * H), dim3(64 * BWD_WAVES), smem, s, p);
    } else {
        CLHIP_CHECK_ARG(dsum_ws != nullptr);
        const int rows = B * H * N;
        if (dtype == CLHIP_BF16) {
            hipLaunchKernelGGL((attn_bwd_generic_kernel<bf16_t, 1>), dim3((rows + 3) / 4), dim3(256), 0, s, p, hd);
            hipLaunchKernelGGL((attn_bwd_generic_kernel<bf16_t, 2>), dim3((rows + 3) / 4), dim3(256), 0, s, p, hd);
        } else {
            hipLaunchKernelGGL((attn_bwd_generic_kernel<float, 1>), dim3((rows + 3) / 4), dim3(256), 0, s, p, hd);
            hipLaunchKernelGGL((attn_bwd_generic_kernel<float, 2>), dim3((rows + 3) / 4), dim3(256), 0, s, p, hd);
        }
    }
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}
