// conv3.hip -- "halo" convolution for 3x3 / stride 1 / pad 1 layers (forward and dgrad), bf16, gfx950.
//
// Why: the PMC passes on the generic implicit-GEMM kernel (profiles/r01_pmc_conv2_*.txt) show it is bound by
// the global->LDS path (~37 GB/s per CU, L2 hit rate 85 %): every one of the 9 filter taps re-gathers the same
// input pixels, so a 256x64 tile moves 40 KB per 2.1 MFLOP (51 FLOP/B).  MI355X has 160 KB of LDS per CU, enough
// to keep the whole input patch of a tile resident: here a workgroup loads the patch
//     pixels [m0 - W - 1, m0 + BM + W + 1) x 64 channels            (one contiguous, fully coalesced range)
// ONCE per 64-channel chunk and all 9 taps read their MFMA operand from it at a shifted LDS address
// (q = p + (W+1) + dh*W + dw); out-of-image taps are zeroed per lane with a precomputed 9-bit mask, so no
// padding is materialised and tiles may span image boundaries.  Only the weights of the current tap
// (BN x 64, 8-16 KB) are streamed per step.  Bytes moved per FLOP drop ~3x (165-205 FLOP/B).
//
// Orientation, MFMA shape, swizzle, epilogue (bf16 store + BatchNorm partial statistics) as in conv2.hip.
#include <stdlib.h>

#include "common.h"

namespace {

// the BatchNorm backward of a layer folded into the operand loads of its own dgrad / weight-gradient launch (see Conv3Params::lz)
struct LazyDz {
    const bf16_t* dy = nullptr;      // [M][C] gradient of the layer's activation, or nullptr: the gradient operand is an ordinary tensor
    const bf16_t* z = nullptr;       // [M][C] its pre-BatchNorm output
    const double* sums = nullptr;    // [rep][2][C]: sum g, sum g * xhat
    int rep = 1;
    const float* mean = nullptr; const float* invstd = nullptr; const float* gamma = nullptr; const float* beta = nullptr;
    double invM = 0.0;
    float* dgamma = nullptr; float* dbeta = nullptr;      // accumulated into by ONE workgroup of the launch
    const unsigned char* mask = nullptr;                  // packed ReLU mask of a conv -> BN -> +res -> ReLU layer (one byte per 8 elements), or nullptr: the mask from z
    bf16_t* dres = nullptr;                               // that layer's residual gradient g = dy * mask, written (dres_acc 0) or accumulated (1) by the dgrad body
    int dres_acc = 0;
};

// coef[6][C] (LDS): mean(g), mean(g xhat), mean, invstd, scale, shift -- bn_bwd_apply_acc_kernel's prologue (bn.hip); `first` = the one
// workgroup of the launch that also adds the sums to dgamma / dbeta
template <int C>
__device__ __forceinline__ void lazy_dz_coefs(const LazyDz& lz, bool first, float* coef) {
    __shared__ double sred[256];
    const int c = threadIdx.x;
    // the per-channel parameters are fetched BEFORE the replica sum: its barrier would otherwise put a second memory round trip on the launch's critical path
    float db_old = 0.f, dg_old = 0.f, m = 0.f, is_c = 0.f, ga_c = 0.f, be_c = 0.f;
    if (c < C) {
        if (first) { db_old = lz.dbeta[c]; dg_old = lz.dgamma[c]; }
        m = lz.mean[c]; is_c = lz.invstd[c]; ga_c = lz.gamma[c]; be_c = lz.beta[c];
    }
    replica_parts(lz.sums, lz.rep, C, sred);
    if (c < C) {
        const float gi_c = ga_c * is_c;
        const float sh_c = be_c - m * gi_c;
        double s1 = 0.0, s2 = 0.0;
        for (int q = 0; q < 256 / (2 * C); ++q) { s1 += sred[q * 2 * C + c]; s2 += sred[q * 2 * C + C + c]; }
        coef[c] = (float)(s1 * lz.invM);
        coef[C + c] = (float)(s2 * lz.invM);
        coef[2 * C + c] = m;
        coef[3 * C + c] = is_c;
        coef[4 * C + c] = gi_c;
        coef[5 * C + c] = sh_c;
        if (first) { lz.dbeta[c] = db_old + (float)s1; lz.dgamma[c] = dg_old + (float)s2; }
    }
    __syncthreads();
}

// this thread's eight channels c0 .. c0 + 7 of the table
struct LazyDz8 { float k0[8], k1[8], mu[8], is[8], gi[8], sh[8]; };
template <int C>
__device__ __forceinline__ void lazy_dz_load(const float* coef, int c0, LazyDz8& t) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        t.k0[e] = coef[c0 + e]; t.k1[e] = coef[C + c0 + e]; t.mu[e] = coef[2 * C + c0 + e];
        t.is[e] = coef[3 * C + c0 + e]; t.gi[e] = coef[4 * C + c0 + e]; t.sh[e] = coef[5 * C + c0 + e];
    }
}
// bits == true: the ReLU mask comes from the packed byte `mb` (bit e = element e), else from the sign of scale z + shift; gg = the masked gradient
__device__ __forceinline__ uint4 lazy_dz8(uint4 dy, uint4 z, const LazyDz8& t, bool bits, unsigned mb, float (&gg)[8]) {
    float g[8], zz[8], o[8];
    unpack8(dy, g);
    unpack8(z, zz);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const bool on = bits ? ((mb >> e) & 1u) != 0u : fmaf(zz[e], t.gi[e], t.sh[e]) > 0.f;
        gg[e] = on ? g[e] : 0.f;
        const float xh = (zz[e] - t.mu[e]) * t.is[e];
        o[e] = t.gi[e] * (gg[e] - t.k0[e] - xh * t.k1[e]);
    }
    return make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
}
__device__ __forceinline__ uint4 lazy_dz8(uint4 dy, uint4 z, const LazyDz8& t, bool bits, unsigned mb) {
    float gg[8];
    return lazy_dz8(dy, z, t, bits, mb, gg);
}
// the dgrad body owns the residual gradient of its tile's pixels: dres (+)= gg, as bn_bwd_apply_acc_kernel stores it
__device__ __forceinline__ void lazy_dres_store(bf16_t* q, const float (&gg)[8], int accumulate) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = gg[e];
    if (accumulate) {
        float old[8];
        unpack8(*reinterpret_cast<const uint4*>(q), old);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = old[e] + gg[e];
    }
    *reinterpret_cast<uint4*>(q) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
}

struct Conv3Params {
    const bf16_t* src;   // [N,H,W,Cs]   (forward: x ; dgrad: dz)
    const bf16_t* wt;    // [Cd][9][Cs]
    bf16_t* dst;         // [N,H,W,Cd]
    float* stats;
    double* stat_acc;   // alternative to `stats`: per-channel [stat_rep][2][Cd] fp64 sums, accumulated with atomics
    int stat_rep;       // number of accumulator replicas (power of two); a workgroup adds into replica blockIdx.x & (stat_rep - 1)
    int N, H, W, Cs, Cd, accumulate;
    int wshift, hshift;  // log2(W), log2(H) when they are powers of two, else -1
    int M;               // N*H*W
    int np;              // patch pixels = BM + 2W + 2
    int patch_bytes;     // np*128 rounded up to 256
    int nbuf;            // patch buffers (2 when Cs > 64)
    int debug;           // CLHIP_ABLATION builds only (CLHIP_CONV3_DEBUG): 1 = skip weight streaming, 2 = skip MFMA, 4 = skip patch loads, 8 = skip stores
    // conv16 / conv32 dgrad only: BatchNorm-backward sums of the layer that PRODUCED the tensor whose gradient this launch completes
    // (sum g and sum g * xhat per channel, g = dy masked by the producer's ReLU), from the fp32 results in the epilogue -- see conv4.hip
    const bf16_t* bn_z = nullptr;    // [N,H,W,Cd] pre-BatchNorm output of that layer, or nullptr: no reduction
    const bf16_t* bn_y = nullptr;    // its post-activation output (ReLU mask y > 0), or nullptr: no ReLU
    const float* bn_mean = nullptr;
    const float* bn_invstd = nullptr;
    double* bn_acc = nullptr;        // [bn_rep][2][Cd]
    int bn_rep = 1;
    const float* bn_coef = nullptr;  // [2][Cd] scale, shift of that layer: its ReLU mask from z when bn_y == nullptr (the activation was never written)
    // conv16 / conv32 FORWARD only ("lazy" input): src is the PRE-BatchNorm output z' of the producing layer and the operand is
    // relu(scale * z' + shift), applied while the patch is staged -- the producer's BatchNorm-apply launch and its activation tensor do
    // not exist.  Every workgroup derives scale / shift from the producer's fp64 statistics accumulators exactly as bn_apply_train_kernel
    // does; workgroup 0 also leaves mean / invstd / scale / shift for the backward and updates the running statistics.
    const double* in_acc = nullptr;  // [in_rep][2][Cs], or nullptr: src is an ordinary activation (or in_eval)
    int in_eval = 0;                 // 1: the producer's BatchNorm is in eval mode -- scale / shift from in_rm / in_rv (read only), nothing is written back
    int in_rep = 1;
    const float* in_gamma = nullptr; const float* in_beta = nullptr;
    float* in_rm = nullptr; float* in_rv = nullptr;
    float in_momentum = 0.f, in_eps = 0.f;
    double in_invM = 0.0, in_unbias = 1.0;
    float* in_mean_o = nullptr; float* in_invstd_o = nullptr; float* in_coef_o = nullptr;      // [Cs], [Cs], [2][Cs]
    // ... of a conv -> BN -> +res -> ReLU producer (LI == 2): the operand is relu(scale * z' + shift + r); the workgroup that owns a pixel also
    // WRITES that activation and its packed ReLU mask (the block output has later readers: the next residual add, the backward) -- everything
    // bn_apply_train_kernel<true, true> would have stored, from the consumer's staging loop
    const bf16_t* in_res = nullptr;  // r [N,H,W,Cs]
    bf16_t* in_y = nullptr;          // out [N,H,W,Cs]
    unsigned char* in_mask = nullptr;      // out [N*H*W*Cs/8], bit e = stored value > 0
    // conv16 / conv32 DGRAD only ("lazy" gradient): src is not read; the operand is the BatchNorm-backward result
    //   dz = scale * (g - mean(g) - xhat * mean(g xhat)),  g = dy * (scale z + shift > 0),  xhat = (z - mean) * invstd
    // of THIS layer's own BatchNorm, computed while the patch is staged from dy and z (bn_bwd_apply_acc_kernel's arithmetic, z-mask form);
    // the two sums come from the fp64 accumulators a consumer's dgrad epilogue filled.  Workgroup 0 adds them to dgamma / dbeta.
    LazyDz lz;
};

// scale / shift of a lazy input into coef[2][C] (LDS); the arithmetic of bn_apply_train_kernel's prologue (bn.hip), bf16 mode
template <int C>
__device__ __forceinline__ void lazy_input_coefs(const Conv3Params& p, int bx, float* coef) {
    __shared__ double sred[256];
    const int c = threadIdx.x;
    if (p.in_eval) {                                       // eval mode: bn_apply_eval_kernel's expressions (bn.hip), so the operand equals the tensor that launch would have written
        if (c < C) {
            const float istd = 1.f / sqrtf(p.in_rv[c] + p.in_eps);
            const float sc = p.in_gamma[c] * istd;
            coef[c] = sc;
            coef[C + c] = p.in_beta[c] - p.in_rm[c] * sc;
        }
        __syncthreads();
        return;
    }
    const bool first = bx == 0;
    const bool upd = first && p.in_rm != nullptr;
    float rm_old = 0.f, rv_old = 0.f, ga_c = 0.f, be_c = 0.f;       // fetched before the replica sum (see lazy_dz_coefs)
    if (c < C) {
        if (upd) { rm_old = p.in_rm[c]; rv_old = p.in_rv[c]; }
        ga_c = p.in_gamma[c]; be_c = p.in_beta[c];
    }
    replica_parts(p.in_acc, p.in_rep, C, sred);
    if (c < C) {
        double s1 = 0.0, s2 = 0.0;
        for (int q = 0; q < 256 / (2 * C); ++q) { s1 += sred[q * 2 * C + c]; s2 += sred[q * 2 * C + C + c]; }
        const double mean = s1 * p.in_invM;
        double var = s2 * p.in_invM - mean * mean;
        if (var < 0.0) var = 0.0;
        const float istd = rsqrtf((float)var + p.in_eps);
        const float sc = ga_c * istd;
        const float sh = be_c - (float)mean * sc;
        coef[c] = sc;
        coef[C + c] = sh;
        if (first) {
            p.in_mean_o[c] = (float)mean;
            p.in_invstd_o[c] = istd;
            p.in_coef_o[c] = sc;
            p.in_coef_o[C + c] = sh;
            if (upd) {
                p.in_rm[c] = (1.f - p.in_momentum) * rm_old + p.in_momentum * (float)mean;
                p.in_rv[c] = (1.f - p.in_momentum) * rv_old + p.in_momentum * (float)(var * p.in_unbias);
            }
        }
    }
    __syncthreads();
}

// BatchNorm-backward sums in the epilogue of the register-resident dgrad kernels (conv16 / conv32): `v[i][e]` = the final fp32 gradient
// of pixel m0 + wave*64 + i*16 + fr, channel c0 + e (c0 = this lane's first channel of the 4-channel group).  Per-channel sums over the
// workgroup's 256 pixels: 16-lane DPP sums, the four waves through LDS, then one fp64 atomic per channel and sum into the producer's
// accumulator replica; the centred form  sum g * xhat = invstd * (sum g z' - mean * sum g)  is taken once per channel.
// the epilogue's operands, fetched before the MFMA loop (fetched in the epilogue they were a memory round trip at the end of a launch a handful long)
template <int NJ> struct BwdPre { uint2 zz[4][NJ], yy[4][NJ]; float4 sc[NJ], sh[NJ]; float istd, mu; };
template <int C, int NJ>
__device__ __forceinline__ void bn_bwd_prefetch(const Conv3Params& p, int m0, int wave, int fr, int fg, int tid, BwdPre<NJ>& q) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pix = m0 + wave * 64 + i * 16 + fr;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            q.zz[i][j] = make_uint2(0u, 0u); q.yy[i][j] = make_uint2(0x3f803f80u, 0x3f803f80u);
            if (pix < p.M) {
                const size_t at = (size_t)pix * C + j * 16 + fg * 4;
                q.zz[i][j] = *reinterpret_cast<const uint2*>(p.bn_z + at);
                if (p.bn_y != nullptr) q.yy[i][j] = *reinterpret_cast<const uint2*>(p.bn_y + at);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        q.sc[j] = make_float4(0.f, 0.f, 0.f, 0.f); q.sh[j] = q.sc[j];
        if (p.bn_y == nullptr && p.bn_coef != nullptr) {
            q.sc[j] = *reinterpret_cast<const float4*>(p.bn_coef + j * 16 + fg * 4); q.sh[j] = *reinterpret_cast<const float4*>(p.bn_coef + C + j * 16 + fg * 4);
        }
    }
    q.istd = 0.f; q.mu = 0.f;
    if (tid < 2 * C) { const int cc = tid < C ? tid : tid - C; q.istd = p.bn_invstd[cc]; q.mu = p.bn_mean[cc]; }
}
template <int C, int NJ>
__device__ __forceinline__ void bn_bwd_sums_epilogue(const Conv3Params& p, const float (&v)[4][NJ][4], const BwdPre<NJ>& q, int m0, int wave, int fr, int fg, int tid, char* smem, int bx) {
    float sv[NJ * 8];
#pragma unroll
    for (int q = 0; q < NJ * 8; ++q) sv[q] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pix = m0 + wave * 64 + i * 16 + fr;
        if (pix < p.M) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const uint2 zz = q.zz[i][j];
                const uint2 yy = q.yy[i][j];
                const float z4[4] = {__uint_as_float(zz.x << 16), __uint_as_float(zz.x & 0xffff0000u), __uint_as_float(zz.y << 16), __uint_as_float(zz.y & 0xffff0000u)};
                float y4[4] = {__uint_as_float(yy.x << 16), __uint_as_float(yy.x & 0xffff0000u), __uint_as_float(yy.y << 16), __uint_as_float(yy.y & 0xffff0000u)};
                if (p.bn_y == nullptr && p.bn_coef != nullptr) {           // the producer's activation was never written: its sign from z (as bn.hip's z-mask kernels)
                    const float4 sc = q.sc[j], sh = q.sh[j];
                    y4[0] = fmaf(z4[0], sc.x, sh.x); y4[1] = fmaf(z4[1], sc.y, sh.y); y4[2] = fmaf(z4[2], sc.z, sh.z); y4[3] = fmaf(z4[3], sc.w, sh.w);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = y4[e] > 0.f ? v[i][j][e] : 0.f;
                    sv[j * 4 + e] += g;
                    sv[NJ * 4 + j * 4 + e] = fmaf(g, z4[e], sv[NJ * 4 + j * 4 + e]);
                }
            }
        }
    }
    row16_sum_n(sv);
    __syncthreads();                                         // the patch is dead
    float* red = reinterpret_cast<float*>(smem);             // [4 waves][2][C]
    if (fr == 0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            *reinterpret_cast<float4*>(red + (wave * 2 + 0) * C + j * 16 + fg * 4) = make_float4(sv[j * 4], sv[j * 4 + 1], sv[j * 4 + 2], sv[j * 4 + 3]);
            *reinterpret_cast<float4*>(red + (wave * 2 + 1) * C + j * 16 + fg * 4) = make_float4(sv[NJ * 4 + j * 4], sv[NJ * 4 + j * 4 + 1], sv[NJ * 4 + j * 4 + 2], sv[NJ * 4 + j * 4 + 3]);
        }
    }
    __syncthreads();
    if (tid < 2 * C) {
        const int which = tid / C, cc = tid - which * C;
        float t = 0.f, sg = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) { t += red[(w2 * 2 + which) * C + cc]; sg += red[(w2 * 2 + 0) * C + cc]; }
        if (which == 1) t = q.istd * (t - q.mu * sg);
        atomicAdd(p.bn_acc + ((size_t)(bx & (p.bn_rep - 1)) * 2 + which) * C + cc, (double)t);
    }
}

// the production build carries no ablation branches (ABL=1 csrc/build.sh builds libclhip_abl.so with them)
#ifdef CLHIP_ABLATION
#define DBG3(p) ((p).debug)
#else
#define DBG3(p) 0
#endif

// 9-bit mask of the taps whose source pixel lies inside the image, for output pixel g (tap t = 3 r + s reads the pixel at
// (h + dh, w + dw), dh = r - 1 / dw = s - 1 forward, mirrored for dgrad): border rows / columns knock out three taps each.
// Image sides that are powers of two (every CIFAR-style layer) avoid the integer divisions.
template <int MODE>
__device__ __forceinline__ unsigned tap_mask(int g, const Conv3Params& p) {
    int w, h;
    if (p.wshift >= 0 && p.hshift >= 0) { w = g & (p.W - 1); h = (g >> p.wshift) & (p.H - 1); }
    else { w = g % p.W; h = (g / p.W) % p.H; }
    constexpr unsigned UP = MODE == 0 ? 0x007u : 0x1c0u, DOWN = MODE == 0 ? 0x1c0u : 0x007u;      // taps with dh = -1 / dh = +1
    constexpr unsigned LEFT = MODE == 0 ? 0x049u : 0x124u, RIGHT = MODE == 0 ? 0x124u : 0x049u;  // taps with dw = -1 / dw = +1
    unsigned m = 0x1ffu;
    if (h == 0) m &= ~UP;
    if (h == p.H - 1) m &= ~DOWN;
    if (w == 0) m &= ~LEFT;
    if (w == p.W - 1) m &= ~RIGHT;
    return m;
}

__device__ __forceinline__ uint4 ldsq(const char* p) { return *reinterpret_cast<const uint4*>(p); }

// LDS rows are 64 bf16 (128 B) padded to a 160-byte pitch.  With 36+4 dwords per row the three lane
// sub-groups a ds_read_b128 is serviced in ({0-3,12-15} at chunk c, {20-27} at chunk c+1) land on 16 distinct
// 16-byte slots for ANY starting row, so tap-shifted reads need no per-read swizzle arithmetic: the address of
// a fragment is  base + shift*160  (one add per fragment per tap).  The kernel is instruction-issue bound
// (a wave issues at most one instruction every ~4 cycles; profiles/r01_pmc_conv3_*.txt), so every VALU
// instruction removed from the tap loop is worth ~1/8 of an MFMA.
constexpr int PITCH = 160;

// WM x WN waves, each wave 64 pixels x 64 channels.  MODE 0 forward (dh = r-1), MODE 1 dgrad (dh = 1-r).
template <int WM, int WN, int MODE>
__global__ __launch_bounds__(WM * WN * 64) void conv3_kernel(Conv3Params p) {
    constexpr int BM = WM * 64, BN = WN * 64, NTH = WM * WN * 64;
    constexpr int WROWS = (BN * 8 + NTH - 1) / NTH;           // weight chunks per thread per tap
    constexpr int PMAX = ((BM + 66) * 8 + NTH - 1) / NTH;     // patch chunks per thread (W <= 32)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* patch0 = smem;                                      // np rows + 1 zero row
    char* wst0 = smem + p.patch_bytes;                        // 2 weight stages of BN rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fg = lane >> 4;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int W = p.W, H = p.H, Cs = p.Cs, K = 9 * Cs;
    const int nchunk = Cs >> 6;
    const int halo = W + 1;

    // ---- per-lane tap masks and LDS base addresses of the 4 pixel fragments of this wave
    unsigned tmask[4];
    int xaddr[4];
    const int zaddr = p.np * PITCH + fg * 16;               // zero row: out-of-image taps read from here
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pl = wm * 64 + i * 16 + fr;               // tile-relative output pixel
        const int g = m0 + pl;
        tmask[i] = g < p.M ? tap_mask<MODE>(g, p) : 0u;
        xaddr[i] = (pl + halo) * PITCH + fg * 16;
    }
    int waddr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) waddr[j] = (wn * 64 + j * 16 + fr) * PITCH + fg * 16;

    // ---- staging (register-parked global loads -> LDS)
    uint4 pw[WROWS];
    const bf16_t* wsrc[WROWS];
    int wdst[WROWS];
#pragma unroll
    for (int i = 0; i < WROWS; ++i) {
        const int idx = tid + i * NTH;
        const int row = idx >> 3, ch = idx & 7;
        const int o = n0 + row;
        wsrc[i] = (row < BN && o < p.Cd) ? p.wt + ((size_t)o * K + ch * 8) : nullptr;
        wdst[i] = row < BN ? row * PITCH + ch * 16 : -1;
    }
    auto wload = [&](int koff) {     // koff = tap*Cs + c*64
#pragma unroll
        for (int i = 0; i < WROWS; ++i) pw[i] = wsrc[i] ? *reinterpret_cast<const uint4*>(wsrc[i] + koff) : make_uint4(0, 0, 0, 0);
    };
    auto wstore = [&](int stage) {
        char* ws = wst0 + stage * (BN * PITCH);
#pragma unroll
        for (int i = 0; i < WROWS; ++i)
            if (wdst[i] >= 0) *reinterpret_cast<uint4*>(ws + wdst[i]) = pw[i];
    };
    uint4 pp[PMAX];
    const int np8 = p.np * 8;
    auto pload = [&](int c) {
#pragma unroll
        for (int i = 0; i < PMAX; ++i) {
            const int idx = tid + i * NTH;
            const int q = idx >> 3, ch = idx & 7;
            const long long g = (long long)m0 - halo + q;
            pp[i] = (idx < np8 && g >= 0 && g < p.M && !(DBG3(p) & 4)) ? *reinterpret_cast<const uint4*>(p.src + ((size_t)g * Cs + c * 64 + ch * 8))
                                                                        : make_uint4(0, 0, 0, 0);
        }
    };
    auto pstore = [&]() {
#pragma unroll
        for (int i = 0; i < PMAX; ++i) {
            const int idx = tid + i * NTH;
            const int q = idx >> 3, ch = idx & 7;
            if (idx < np8) *reinterpret_cast<uint4*>(patch0 + q * PITCH + ch * 16) = pp[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: zero row, patch of chunk 0, weights of (chunk 0, tap 0)
    if (tid < 10) *reinterpret_cast<uint4*>(patch0 + p.np * PITCH + tid * 16) = make_uint4(0, 0, 0, 0);
    pload(0);
    wload(0);
    pstore();
    wstore(0);
    __syncthreads();

    int step = 0;
    for (int c = 0; c < nchunk; ++c) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap, ++step) {
            const bool last = (c == nchunk - 1) && (tap == 8);
            if (!last && !(DBG3(p) & 1)) wload(tap == 8 ? (c + 1) * 64 : (tap + 1) * Cs + c * 64);
            if (tap == 5 && c + 1 < nchunk) pload(c + 1);
            const char* ws = wst0 + (step & 1) * (BN * PITCH);
            constexpr int R = 0;
            (void)R;
            const int r = tap / 3, s = tap - 3 * r;
            const int shift = (MODE == 0 ? (r - 1) * W + (s - 1) : (1 - r) * W + (1 - s)) * PITCH;
            int xa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xa[i] = (tmask[i] & (1u << tap)) ? xaddr[i] + shift : zaddr;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 xf[4], wf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) xf[i] = ldsq(patch0 + xa[i] + ks * 64);
#pragma unroll
                for (int j = 0; j < 4; ++j) wf[j] = ldsq(ws + waddr[j] + ks * 64);
                if (!(DBG3(p) & 2))
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[j]),
                                                                           __builtin_bit_cast(bf16x8_t, xf[i]), acc[i][j], 0, 0, 0);
            }
            if (tap == 8 && c + 1 < nchunk) {
                __syncthreads();                    // every wave is done with this chunk's patch
                pstore();
            }
            if (!last && !(DBG3(p) & 1)) wstore((step + 1) & 1);
            __syncthreads();
        }
    }

    // ---- epilogue.  Accumulators hold D[row = channel fg*4+e][col = pixel fr]; they are staged through LDS as a
    //      [BM pixels][BN channels] bf16 image so that the global stores are full 16-byte-per-lane row segments.
    {
        char* ot = smem;
        constexpr int OPITCH = BN * 2 + 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pl = wm * 64 + i * 16 + fr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cl = wn * 64 + j * 16 + fg * 4;
                float v0 = acc[i][j][0], v1 = acc[i][j][1], v2 = acc[i][j][2], v3 = acc[i][j][3];
                if (MODE == 1 && p.accumulate) {                          // dx += result: add in fp32, round once
                    const int pix = m0 + pl, o = n0 + cl;
                    if (pix < p.M && o < p.Cd) {
                        const uint2 old = *reinterpret_cast<const uint2*>(p.dst + (size_t)pix * p.Cd + o);
                        v0 += __uint_as_float(old.x << 16); v1 += __uint_as_float(old.x & 0xffff0000u);
                        v2 += __uint_as_float(old.y << 16); v3 += __uint_as_float(old.y & 0xffff0000u);
                    }
                }
                uint2 u;
                u.x = pack_bf16x2(v0, v1);
                u.y = pack_bf16x2(v2, v3);
                *reinterpret_cast<uint2*>(ot + pl * OPITCH + cl * 2) = u;
            }
        }
        __syncthreads();
        constexpr int CPR = BN / 8;                        // 16-byte chunks per pixel row of the tile
        for (int idx = tid; idx < BM * CPR; idx += NTH) {
            const int pl = idx / CPR, ch = idx - pl * CPR;
            const int pix = m0 + pl, o = n0 + ch * 8;
            if (pix < p.M && o < p.Cd && !(DBG3(p) & 8))
                *reinterpret_cast<uint4*>(p.dst + (size_t)pix * p.Cd + o) = *reinterpret_cast<const uint4*>(ot + pl * OPITCH + ch * 16);
        }
        __syncthreads();
    }
    if (MODE == 0 && (p.stats != nullptr || p.stat_acc != nullptr)) {
        float* red = reinterpret_cast<float*>(smem);        // [WM][2][BN]
        float sv[32];                                     // [j][e] sums, then [j][e] sums of squares
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float v = acc[i][j][e]; s1 += v; s2 = fmaf(v, v, s2); }
                sv[j * 4 + e] = s1;
                sv[16 + j * 4 + e] = s2;
            }
        row16_sum_n(sv);
        if (fr == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {                   // channels fg*4 .. fg*4+3 of tile j are contiguous: 16-byte stores
                const int cc = wn * 64 + j * 16 + fg * 4;
                *reinterpret_cast<float4*>(red + (wm * 2 + 0) * BN + cc) = make_float4(sv[j * 4], sv[j * 4 + 1], sv[j * 4 + 2], sv[j * 4 + 3]);
                *reinterpret_cast<float4*>(red + (wm * 2 + 1) * BN + cc) = make_float4(sv[16 + j * 4], sv[17 + j * 4], sv[18 + j * 4], sv[19 + j * 4]);
            }
        }
        __syncthreads();
        for (int idx = tid; idx < 2 * BN; idx += NTH) {
            const int which = idx / BN, cc = idx - which * BN;
            float t = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < WM; ++w2) t += red[(w2 * 2 + which) * BN + cc];
            if (n0 + cc < p.Cd) {
                if (p.stat_acc != nullptr) atomicAdd(p.stat_acc + ((size_t)(blockIdx.x & (p.stat_rep - 1)) * 2 + which) * p.Cd + n0 + cc, (double)t);
                else p.stats[((size_t)blockIdx.x * 2 + which) * p.Cd + n0 + cc] = t;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// conv3g: same patch scheme, but the per-tap WEIGHT tiles are streamed by LDS-DMA (global_load_lds_dwordx4,
// no staging registers) into a 4-deep ring, 3 taps ahead of the MFMAs.  Motivation (profiles/r01_*): on the
// 8x8 / 4x4 layers there is only ~1 wave per SIMD, so nothing hides the ~1-2 us weight-load round trip when the
// prefetch distance is one tap (~0.25 us of MFMAs); register prefetch at distance 3 would cost 48+ VGPRs.
// The DMA writes lane-linearly (1 KB per wave-instruction = 8 rows of 128 B), so the weight ring uses 128-byte
// rows with the XOR swizzle applied to the per-lane SOURCE address and, pre-computed once, to the read address.
// Waits are counted (s_waitcnt vmcnt(N)) + raw s_barrier: __syncthreads() would drain the DMA queue.
template <int WM, int WN, int MODE>
__global__ __launch_bounds__(WM * WN * 64) void conv3g_kernel(Conv3Params p) {
    constexpr int BM = WM * 64, BN = WN * 64, NTH = WM * WN * 64, NW = NTH / 64;
    constexpr int NST = 4;
    constexpr int WINST = (BN / 8) / NW;                      // DMA instructions per wave per tap
    static_assert((BN / 8) % NW == 0, "row groups must divide among the waves");
    constexpr int PMAX = ((BM + 66) * 8 + NTH - 1) / NTH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* patch0 = smem;
    char* wst0 = smem + p.patch_bytes;                        // NST stages of BN x 128 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fg = lane >> 4;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int W = p.W, H = p.H, Cs = p.Cs, K = 9 * Cs;
    const int nchunk = Cs >> 6;
    const int halo = W + 1;
    const int nsteps = nchunk * 9;

    unsigned tmask[4];
    int xaddr[4];
    const int zaddr = p.np * PITCH + fg * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pl = wm * 64 + i * 16 + fr;
        const int g = m0 + pl;
        tmask[i] = g < p.M ? tap_mask<MODE>(g, p) : 0u;
        xaddr[i] = (pl + halo) * PITCH + fg * 16;
    }
    int waddr[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int row = wn * 64 + j * 16 + fr;
            waddr[j][ks] = row * 128 + (((ks * 4 + fg) ^ (row & 7)) << 4);
        }
    // DMA source pointers: instruction i of this wave fills rows [8g, 8g+8) of the stage, g = wave*WINST + i;
    // lane l lands at row 8g + (l>>3), 16-byte slot (l&7), which must hold logical chunk (l&7) ^ (row&7).
    const bf16_t* wsrc[WINST];
#pragma unroll
    for (int i = 0; i < WINST; ++i) {
        const int g = wave * WINST + i;
        const int row = g * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (row & 7);
        const int o = n0 + row < p.Cd ? n0 + row : p.Cd - 1;    // rows past Cd (never stored) read a valid address
        wsrc[i] = p.wt + ((size_t)o * K + chunk * 8);
    }
    typedef __attribute__((address_space(1))) const void gvoid;
    typedef __attribute__((address_space(3))) void lvoid;
    auto wdma = [&](int step) {          // stream the weights of K-step `step` into ring slot step % NST
        const int c = step / 9, tap = step - 9 * c;
        const int koff = tap * Cs + c * 64;
        char* ws = wst0 + (step & (NST - 1)) * (BN * 128) + wave * WINST * 1024;
#pragma unroll
        for (int i = 0; i < WINST; ++i)
            __builtin_amdgcn_global_load_lds((gvoid*)(wsrc[i] + koff), (lvoid*)(ws + i * 1024), 16, 0, 0);
    };

    uint4 pp[PMAX];
    const int np8 = p.np * 8;
    auto pload = [&](int c) {
#pragma unroll
        for (int i = 0; i < PMAX; ++i) {
            const int idx = tid + i * NTH;
            const int q = idx >> 3, ch = idx & 7;
            const long long g = (long long)m0 - halo + q;
            pp[i] = (idx < np8 && g >= 0 && g < p.M) ? *reinterpret_cast<const uint4*>(p.src + ((size_t)g * Cs + c * 64 + ch * 8))
                                                      : make_uint4(0, 0, 0, 0);
        }
    };
    auto pstore = [&]() {
#pragma unroll
        for (int i = 0; i < PMAX; ++i) {
            const int idx = tid + i * NTH;
            const int q = idx >> 3, ch = idx & 7;
            if (idx < np8) *reinterpret_cast<uint4*>(patch0 + q * PITCH + ch * 16) = pp[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: patch of chunk 0 (register path), weights of steps 0..2 (DMA)
    if (tid < 10) *reinterpret_cast<uint4*>(patch0 + p.np * PITCH + tid * 16) = make_uint4(0, 0, 0, 0);
    pload(0);
    pstore();                                    // compiler waits for the patch loads here (vmcnt(0)): nothing else in flight yet
    wdma(0);
    if (nsteps > 1) wdma(1);
    if (nsteps > 2) wdma(2);
    // wait until step 0's weights have landed: at most the 2 younger steps may be outstanding
    if (nsteps > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WINST) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int step = 0;
    for (int c = 0; c < nchunk; ++c) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap, ++step) {
            if (step + 3 < nsteps) wdma(step + 3);           // slot (step+3)%4 was last read at step-1: free since the last barrier
            if (tap == 5 && c + 1 < nchunk) pload(c + 1);
            const char* ws = wst0 + (step & (NST - 1)) * (BN * 128);
            const int r = tap / 3, s = tap - 3 * r;
            const int shift = (MODE == 0 ? (r - 1) * W + (s - 1) : (1 - r) * W + (1 - s)) * PITCH;
            int xa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xa[i] = (tmask[i] & (1u << tap)) ? xaddr[i] + shift : zaddr;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 xf[4], wf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) xf[i] = ldsq(patch0 + xa[i] + ks * 64);
#pragma unroll
                for (int j = 0; j < 4; ++j) wf[j] = ldsq(ws + waddr[j][ks]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[j]),
                                                                           __builtin_bit_cast(bf16x8_t, xf[i]), acc[i][j], 0, 0, 0);
            }
            if (tap == 8 && c + 1 < nchunk) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();            // every wave is done reading this chunk's patch
                pstore();                                // (waits for the parked patch loads and, in order, everything older)
            }
            // next step's weights must have landed; the two younger DMA batches may stay in flight.
            // (in-order completion: waiting for "<= 2 batches outstanding" retires step+1's batch and anything older)
            if (step + 3 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WINST) : "memory");
            else if (step + 2 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WINST) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }

    {
        char* ot = smem;
        constexpr int OPITCH = BN * 2 + 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pl = wm * 64 + i * 16 + fr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cl = wn * 64 + j * 16 + fg * 4;
                float v0 = acc[i][j][0], v1 = acc[i][j][1], v2 = acc[i][j][2], v3 = acc[i][j][3];
                if (MODE == 1 && p.accumulate) {
                    const int pix = m0 + pl, o = n0 + cl;
                    if (pix < p.M && o < p.Cd) {
                        const uint2 old = *reinterpret_cast<const uint2*>(p.dst + (size_t)pix * p.Cd + o);
                        v0 += __uint_as_float(old.x << 16); v1 += __uint_as_float(old.x & 0xffff0000u);
                        v2 += __uint_as_float(old.y << 16); v3 += __uint_as_float(old.y & 0xffff0000u);
                    }
                }
                uint2 u;
                u.x = pack_bf16x2(v0, v1);
                u.y = pack_bf16x2(v2, v3);
                *reinterpret_cast<uint2*>(ot + pl * OPITCH + cl * 2) = u;
            }
        }
        __syncthreads();
        constexpr int CPR = BN / 8;
        for (int idx = tid; idx < BM * CPR; idx += NTH) {
            const int pl = idx / CPR, ch = idx - pl * CPR;
            const int pix = m0 + pl, o = n0 + ch * 8;
            if (pix < p.M && o < p.Cd)
                *reinterpret_cast<uint4*>(p.dst + (size_t)pix * p.Cd + o) = *reinterpret_cast<const uint4*>(ot + pl * OPITCH + ch * 16);
        }
        __syncthreads();
    }
    if (MODE == 0 && (p.stats != nullptr || p.stat_acc != nullptr)) {
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) { float v = acc[i][j][e]; s1 += v; s2 += v * v; }
                s1 = row16_sum(s1);
                s2 = row16_sum(s2);
                if (fr == 0) {
                    const int cc = wn * 64 + j * 16 + fg * 4 + e;
                    red[(wm * 2 + 0) * BN + cc] = s1;
                    red[(wm * 2 + 1) * BN + cc] = s2;
                }
            }
        }
        __syncthreads();
        for (int idx = tid; idx < 2 * BN; idx += NTH) {
            const int which = idx / BN, cc = idx - which * BN;
            float t = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < WM; ++w2) t += red[(w2 * 2 + which) * BN + cc];
            if (n0 + cc < p.Cd) {
                if (p.stat_acc != nullptr) atomicAdd(p.stat_acc + ((size_t)(blockIdx.x & (p.stat_rep - 1)) * 2 + which) * p.Cd + n0 + cc, (double)t);
                else p.stats[((size_t)blockIdx.x * 2 + which) * p.Cd + n0 + cc] = t;
            }
        }
    }
}

// The patch of a register-staged kernel (conv16 / conv32 / conv64): pixels [m0 - halo, m0 + BM + halo) as 16-byte chunks into LDS at pitch PP,
// plus the zero row.  With a lazy operand (LI: BatchNorm input, LZ: BatchNorm-backward gradient) the RAW chunks of the first PF rounds are
// fetched BEFORE the coefficient prologue: that prologue is a memory round trip of its own (the fp64 replicas) behind two barriers, and with
// the loads after it a launch at its latency floor paid launch -> replicas -> patch -> MFMA in series (batch 32: 6.7-10.6 us per launch for
// 0.3 us of MFMAs).  Same arithmetic, same stores: bit-identical results.
struct RawChunk { uint4 a, b; unsigned m; };
template <int MODE, bool LZ, int LI, int C, int PP, int BM, int PF>
__device__ __forceinline__ void stage_patch(const Conv3Params& p, const int bx, const int m0, const int halo, char* smem) {
    constexpr int CPP = C / 8, LOG = CPP == 2 ? 1 : (CPP == 4 ? 2 : 3);
    constexpr bool lazy = MODE == 0 && LI != 0;
    constexpr bool lres = MODE == 0 && LI == 2;              // ... of a +res producer: two tensors in, the activation and its mask out
    constexpr bool lzd = MODE == 1 && LZ;                    // (template parameters: as run-time tests the tables' registers and the branches slowed EVERY launch)
    const int tid = threadIdx.x;
    const int nchunks = p.np * CPP;
    const int ch = tid & (CPP - 1);                          // chunk idx = tid + 256 * round: the same 8 channels in every round
    auto fetch = [&](int idx, RawChunk& rc) {
        rc.a = make_uint4(0, 0, 0, 0); rc.b = rc.a; rc.m = 0u;
        const int q = idx >> LOG;
        const long long g = (long long)m0 - halo + q;
        if (idx < nchunks && g >= 0 && g < p.M) {
            const size_t at = (size_t)g * C + ch * 8;
            if constexpr (lzd) {
                rc.a = *reinterpret_cast<const uint4*>(p.lz.dy + at);
                rc.b = *reinterpret_cast<const uint4*>(p.lz.z + at);
                if (p.lz.mask != nullptr) rc.m = p.lz.mask[(size_t)g * CPP + ch];
            } else {
                rc.a = *reinterpret_cast<const uint4*>(p.src + at);
                if constexpr (lres) rc.b = *reinterpret_cast<const uint4*>(p.in_res + at);
            }
        }
    };
    RawChunk raw[PF];
    if constexpr (lazy || lzd) {
#pragma unroll
        for (int it = 0; it < PF; ++it) fetch(tid + it * 256, raw[it]);
    }
    float isc[8], ish[8];
    LazyDz8 lt;
    float* coef = reinterpret_cast<float*>(smem + (p.np + 1) * PP);
    if constexpr (lazy) {
        lazy_input_coefs<C>(p, bx, coef);
#pragma unroll
        for (int e = 0; e < 8; ++e) { isc[e] = coef[ch * 8 + e]; ish[e] = coef[C + ch * 8 + e]; }
    }
    if constexpr (lzd) {                                     // the gradient operand = this layer's BatchNorm backward, from dy and z
        lazy_dz_coefs<C>(p.lz, bx == 0, coef);
        lazy_dz_load<C>(coef, ch * 8, lt);
    }
    auto stage = [&](int idx, const RawChunk& rc) {
        const int q = idx >> LOG;
        const long long g = (long long)m0 - halo + q;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (g >= 0 && g < p.M) {
            const size_t at = (size_t)g * C + ch * 8;
            if constexpr (lzd) {
                float gg[8];
                v = lazy_dz8(rc.a, rc.b, lt, p.lz.mask != nullptr, rc.m, gg);
                if (p.lz.dres != nullptr && q >= halo && q < halo + BM) lazy_dres_store(p.lz.dres + at, gg, p.lz.dres_acc);
            } else if constexpr (lres) {
                unsigned mb;
                v = bn_res_relu8_bf16(rc.a, rc.b, isc, ish, mb);
                if (q >= halo && q < halo + BM) { *reinterpret_cast<uint4*>(p.in_y + at) = v; if (p.in_mask != nullptr) p.in_mask[(size_t)g * CPP + ch] = (unsigned char)mb; }
            } else if constexpr (lazy) {
                v = bn_relu8_bf16(rc.a, isc, ish);
            } else {
                v = rc.a;
            }
        }
        *reinterpret_cast<uint4*>(smem + q * PP + ch * 16) = v;
    };
    int idx = tid;
    if constexpr (lazy || lzd) {
#pragma unroll
        for (int it = 0; it < PF; ++it)
            if (tid + it * 256 < nchunks) stage(tid + it * 256, raw[it]);
        idx = tid + PF * 256;
    }
    for (; idx < nchunks; idx += 256) {
        RawChunk rc;
        fetch(idx, rc);
        stage(idx, rc);
    }
    if (tid < CPP) *reinterpret_cast<uint4*>(smem + p.np * PP + tid * 16) = make_uint4(0, 0, 0, 0);
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------
// conv16: 3x3 / stride 1 / pad 1 with 16 input and 16 output channels (ResNet-32 stage 1: 11 of its 33 convolutions, on the
// largest maps).  The generic implicit-GEMM kernel spent 26 us (forward) / 40 us (dgrad) on [256,32,32,16] -- 2 % of the MFMA
// peak and 10-20x the 17 MB the layer moves -- because its tiling is built around wide channel counts.  Here the whole
// weight tensor (16 x 9 x 16 bf16 = 4.6 KB) lives in 20 registers per lane as five ready-made MFMA operands (two taps of 16
// channels fill one K = 32 step; the 10th half is zero), the pixel patch of a 256-pixel tile sits in 10 KB of LDS at a 32-byte
// pitch (conflict-free for ds_read_b128), and a 16-pixel x 16-channel output tile costs 5 LDS reads + 5 MFMAs with no barrier
// after the patch is staged.  Lane (fr, fg) ends with channels 4 fg .. 4 fg + 3 of pixel fr: one 8-byte store, 512 contiguous
// bytes per wave instruction, no output staging.  MODE as in conv3_kernel (the dgrad weight copy has the same [Cd][9][Cs] layout).
template <int MODE, bool LZ = false, int LI = 0>
__device__ __forceinline__ void conv16_body(const Conv3Params& p, const int bx, char* smem) {      // bx = tile index (a workgroup of a plain or a fused launch)
    constexpr int BM = 256, PP = 32;                         // pixels per workgroup, LDS bytes per pixel
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int m0 = bx * BM;
    const int W = p.W, halo = W + 1;
    const int half = fg & 1, tsel = fg >> 1;                // which 8 channels / which tap of the pair this lane feeds

    uint4 wreg[5];
    int sh[5];
    unsigned bit[5];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        const int tap = 2 * ks + tsel;
        const int r = tap / 3, s_ = tap - 3 * r;
        wreg[ks] = tap < 9 ? *reinterpret_cast<const uint4*>(p.wt + (size_t)fr * 144 + tap * 16 + half * 8) : make_uint4(0, 0, 0, 0);
        sh[ks] = (MODE == 0 ? (r - 1) * W + (s_ - 1) : (1 - r) * W + (1 - s_)) * PP;
        bit[ks] = tap < 9 ? 1u << tap : 0u;
    }
    // patch: pixels [m0 - halo, m0 + BM + halo) as 16-byte half rows, plus one zero row for the out-of-image taps
    stage_patch<MODE, LZ, LI, 16, PP, BM, 3>(p, bx, m0, halo, smem);

    const int zaddr = p.np * PP + half * 16;
    BwdPre<1> bpre;
    if (MODE == 1 && p.bn_z != nullptr) bn_bwd_prefetch<16, 1>(p, m0, wave, fr, fg, tid, bpre);
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pl = wave * 64 + i * 16 + fr;
        const int g = m0 + pl;
        const unsigned mask = g < p.M ? tap_mask<MODE>(g, p) : 0u;
        const int base = (pl + halo) * PP + half * 16;
        acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
            const uint4 x = ldsq(smem + ((mask & bit[ks]) ? base + sh[ks] : zaddr));
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wreg[ks]), __builtin_bit_cast(bf16x8_t, x), acc[i], 0, 0, 0);
        }
    }
    // D[row = channel fg*4 + e][col = pixel fr]
    float fin[4][1][4];                                      // the stored gradient before its bf16 rounding (dgrad + BatchNorm-backward sums)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pix = m0 + wave * 64 + i * 16 + fr;
        float v0 = acc[i][0], v1 = acc[i][1], v2 = acc[i][2], v3 = acc[i][3];
        if (pix < p.M) {
            bf16_t* o = p.dst + (size_t)pix * 16 + fg * 4;
            if (MODE == 1 && p.accumulate) {
                const uint2 old = *reinterpret_cast<const uint2*>(o);
                v0 += __uint_as_float(old.x << 16); v1 += __uint_as_float(old.x & 0xffff0000u);
                v2 += __uint_as_float(old.y << 16); v3 += __uint_as_float(old.y & 0xffff0000u);
            }
            *reinterpret_cast<uint2*>(o) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
        }
        fin[i][0][0] = v0; fin[i][0][1] = v1; fin[i][0][2] = v2; fin[i][0][3] = v3;
    }
    if (MODE == 1 && p.bn_z != nullptr) bn_bwd_sums_epilogue<16, 1>(p, fin, bpre, m0, wave, fr, fg, tid, smem, bx);
    if (MODE == 0 && (p.stats != nullptr || p.stat_acc != nullptr)) {
        float sv[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = m0 + wave * 64 + i * 16 + fr < p.M;
                const float v = ok ? acc[i][e] : 0.f;
                s1 += v; s2 = fmaf(v, v, s2);
            }
            sv[e] = s1; sv[4 + e] = s2;
        }
        row16_sum_n(sv);
        __syncthreads();                                     // the patch is dead
        float* red = reinterpret_cast<float*>(smem);         // [4 waves][2][16]
        if (fr == 0) {
            *reinterpret_cast<float4*>(red + (wave * 2 + 0) * 16 + fg * 4) = make_float4(sv[0], sv[1], sv[2], sv[3]);
            *reinterpret_cast<float4*>(red + (wave * 2 + 1) * 16 + fg * 4) = make_float4(sv[4], sv[5], sv[6], sv[7]);
        }
        __syncthreads();
        if (tid < 32) {
            const int which = tid >> 4, cc = tid & 15;
            const float t = red[(0 * 2 + which) * 16 + cc] + red[(1 * 2 + which) * 16 + cc] + red[(2 * 2 + which) * 16 + cc] + red[(3 * 2 + which) * 16 + cc];
            if (p.stat_acc != nullptr) atomicAdd(p.stat_acc + ((size_t)(bx & (p.stat_rep - 1)) * 2 + which) * 16 + cc, (double)t);
            else p.stats[((size_t)bx * 2 + which) * 16 + cc] = t;
        }
    }
}
template <int MODE, int LI = 0>
__global__ __launch_bounds__(256) void conv16_kernel(Conv3Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    conv16_body<MODE, false, LI>(p, blockIdx.x, smem);
}

// conv32: the same scheme for 32 -> 32 channels (ResNet-32 stage 2).  One tap fills a K = 32 step, the 32 output channels are two
// MFMA row tiles, the weights are 18 operands (72 registers) per lane; the patch pitch is 96 bytes (64 of data), which puts the 16
// lanes a ds_read_b128 services together on 16 distinct bank quartets.
template <int MODE, bool LZ = false, int LI = 0>
__device__ __forceinline__ void conv32_body(const Conv3Params& p, const int bx, char* smem) {
    constexpr int BM = 256, PP = 96;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int m0 = bx * BM;
    const int W = p.W, halo = W + 1;

    uint4 wreg[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j) wreg[t][j] = *reinterpret_cast<const uint4*>(p.wt + (size_t)(j * 16 + fr) * 288 + t * 32 + fg * 8);
    stage_patch<MODE, LZ, LI, 32, PP, BM, 5>(p, bx, m0, halo, smem);

    const int zaddr = p.np * PP + fg * 16;
    BwdPre<2> bpre;
    if (MODE == 1 && p.bn_z != nullptr) bn_bwd_prefetch<32, 2>(p, m0, wave, fr, fg, tid, bpre);
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pl = wave * 64 + i * 16 + fr;
        const int g = m0 + pl;
        const unsigned mask = g < p.M ? tap_mask<MODE>(g, p) : 0u;
        const int base = (pl + halo) * PP + fg * 16;
        acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int r = t / 3, s_ = t - 3 * r;
            const int shift = (MODE == 0 ? (r - 1) * W + (s_ - 1) : (1 - r) * W + (1 - s_)) * PP;
            const uint4 x = ldsq(smem + ((mask & (1u << t)) ? base + shift : zaddr));
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wreg[t][j]), __builtin_bit_cast(bf16x8_t, x), acc[i][j], 0, 0, 0);
        }
    }
    float fin[4][2][4];                                      // the stored gradient before its bf16 rounding (dgrad + BatchNorm-backward sums)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pix = m0 + wave * 64 + i * 16 + fr;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float v0 = acc[i][j][0], v1 = acc[i][j][1], v2 = acc[i][j][2], v3 = acc[i][j][3];
            if (pix < p.M) {
                bf16_t* o = p.dst + (size_t)pix * 32 + j * 16 + fg * 4;
                if (MODE == 1 && p.accumulate) {
                    const uint2 old = *reinterpret_cast<const uint2*>(o);
                    v0 += __uint_as_float(old.x << 16); v1 += __uint_as_float(old.x & 0xffff0000u);
                    v2 += __uint_as_float(old.y << 16); v3 += __uint_as_float(old.y & 0xffff0000u);
                }
                *reinterpret_cast<uint2*>(o) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
            }
            fin[i][j][0] = v0; fin[i][j][1] = v1; fin[i][j][2] = v2; fin[i][j][3] = v3;
        }
    }
    if (MODE == 1 && p.bn_z != nullptr) bn_bwd_sums_epilogue<32, 2>(p, fin, bpre, m0, wave, fr, fg, tid, smem, bx);
    if (MODE == 0 && (p.stats != nullptr || p.stat_acc != nullptr)) {
        float sv[16];                                        // [j][e] sums, then [j][e] sums of squares
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool ok = m0 + wave * 64 + i * 16 + fr < p.M;
                    const float v = ok ? acc[i][j][e] : 0.f;
                    s1 += v; s2 = fmaf(v, v, s2);
                }
                sv[j * 4 + e] = s1; sv[8 + j * 4 + e] = s2;
            }
        row16_sum_n(sv);
        __syncthreads();                                     // the patch is dead
        float* red = reinterpret_cast<float*>(smem);         // [4 waves][2][32]
        if (fr == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                *reinterpret_cast<float4*>(red + (wave * 2 + 0) * 32 + j * 16 + fg * 4) = make_float4(sv[j * 4], sv[j * 4 + 1], sv[j * 4 + 2], sv[j * 4 + 3]);
                *reinterpret_cast<float4*>(red + (wave * 2 + 1) * 32 + j * 16 + fg * 4) = make_float4(sv[8 + j * 4], sv[9 + j * 4], sv[10 + j * 4], sv[11 + j * 4]);
            }
        }
        __syncthreads();
        if (tid < 64) {
            const int which = tid >> 5, cc = tid & 31;
            float t = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) t += red[(w2 * 2 + which) * 32 + cc];
            if (p.stat_acc != nullptr) atomicAdd(p.stat_acc + ((size_t)(bx & (p.stat_rep - 1)) * 2 + which) * 32 + cc, (double)t);
            else p.stats[((size_t)bx * 2 + which) * 32 + cc] = t;
        }
    }
}
template <int MODE, int LI = 0>
__global__ __launch_bounds__(256) void conv32_kernel(Conv3Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    conv32_body<MODE, false, LI>(p, blockIdx.x, smem);
}

// conv64: 3x3 / s1 / p1 with 64 -> 64 channels on SMALL maps (CifarResNet-32 stage 3: 8x8 images, 16 384 pixels at batch 256 -- launches at their
// latency floor, where conv4.hip's ring of slabs has nothing to amortise).  Same scheme as conv16 / conv32 with the output channels split over the
// waves: wave w owns output channels 16 w .. 16 w + 15 and keeps THEIR filters in 72 registers (9 taps x two 32-wide K steps), a workgroup is 64
// pixels, the patch sits in LDS at a 144-byte pitch (conflict-free for ds_read_b128), every wave multiplies all four 16-pixel tiles.  No cross-wave
// reduction anywhere: a wave owns its channels' BatchNorm sums (forward statistics, backward sums) and adds them to the fp64 accumulators directly.
// The patch is staged through registers, so the lazy BatchNorm operands of conv16 / conv32 apply (LI: input, LZ: gradient).
template <int MODE, bool LZ = false, int LI = 0, int BM = 64>
__device__ __forceinline__ void conv64_body(const Conv3Params& p, const int bx, char* smem) {
    constexpr int PP = 144, C = 64, NT = BM / 16;                // BM pixels per workgroup (64: the fused backward launch; 128: the forward, half the statistics atomics)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int m0 = bx * BM;
    const int W = p.W, halo = W + 1;

    uint4 wreg[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wreg[t][ks] = *reinterpret_cast<const uint4*>(p.wt + (size_t)(wave * 16 + fr) * 576 + t * 64 + ks * 32 + fg * 8);
    stage_patch<MODE, LZ, LI, C, PP, BM, (BM == 64 ? 3 : 6)>(p, bx, m0, halo, smem);

    const int zaddr = p.np * PP + fg * 16;
    // dgrad + BatchNorm-backward sums of the producer: its z' (and y, or the z-mask coefficients) and the two statistics the final atomics need are
    // fetched HERE, under the MFMA loop -- fetched in the epilogue they were one more memory round trip at the end of a launch that is a handful long
    const int c4 = wave * 16 + fg * 4;                       // D[row = channel wave*16 + fg*4 + e][col = pixel fr]
    const bool bwd_sums = MODE == 1 && p.bn_z != nullptr;
    const bool zmask = bwd_sums && p.bn_y == nullptr && p.bn_coef != nullptr;
    uint2 pzz[NT], pyy[NT];
    float4 msc = make_float4(0.f, 0.f, 0.f, 0.f), msh = msc, pis = msc, pmu = msc;
    if (bwd_sums) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int pix = m0 + i * 16 + fr;
            pzz[i] = make_uint2(0u, 0u); pyy[i] = make_uint2(0u, 0u);
            if (pix < p.M) {
                const size_t at = (size_t)pix * C + c4;
                pzz[i] = *reinterpret_cast<const uint2*>(p.bn_z + at);
                if (p.bn_y != nullptr) pyy[i] = *reinterpret_cast<const uint2*>(p.bn_y + at);
            }
        }
        if (zmask) { msc = *reinterpret_cast<const float4*>(p.bn_coef + c4); msh = *reinterpret_cast<const float4*>(p.bn_coef + C + c4); }
        if (fr == 0) { pis = *reinterpret_cast<const float4*>(p.bn_invstd + c4); pmu = *reinterpret_cast<const float4*>(p.bn_mean + c4); }
    }
    f32x4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int pl = i * 16 + fr;
        const int g = m0 + pl;
        const unsigned mask = g < p.M ? tap_mask<MODE>(g, p) : 0u;
        const int base = (pl + halo) * PP + fg * 16;
        acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int r = t / 3, s_ = t - 3 * r;
            const int shift = (MODE == 0 ? (r - 1) * W + (s_ - 1) : (1 - r) * W + (1 - s_)) * PP;
            const int a = (mask & (1u << t)) ? base + shift : zaddr;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint4 x = ldsq(smem + a + ks * 64);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wreg[t][ks]), __builtin_bit_cast(bf16x8_t, x), acc[i], 0, 0, 0);
            }
        }
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};      // forward: sum z, sum z^2; dgrad + bn_z: sum g, sum g z'
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int pix = m0 + i * 16 + fr;
        float v[4] = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
        if (pix < p.M) {
            bf16_t* o = p.dst + (size_t)pix * C + c4;
            if (MODE == 1 && p.accumulate) {
                const uint2 old = *reinterpret_cast<const uint2*>(o);
                v[0] += __uint_as_float(old.x << 16); v[1] += __uint_as_float(old.x & 0xffff0000u);
                v[2] += __uint_as_float(old.y << 16); v[3] += __uint_as_float(old.y & 0xffff0000u);
            }
            *reinterpret_cast<uint2*>(o) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            if (MODE == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { s1[e] += acc[i][e]; s2[e] = fmaf(acc[i][e], acc[i][e], s2[e]); }
            } else if (bwd_sums) {
                const uint2 zz = pzz[i];
                const float z4[4] = {__uint_as_float(zz.x << 16), __uint_as_float(zz.x & 0xffff0000u), __uint_as_float(zz.y << 16), __uint_as_float(zz.y & 0xffff0000u)};
                float y4[4] = {1.f, 1.f, 1.f, 1.f};
                if (p.bn_y != nullptr) {
                    const uint2 yy = pyy[i];
                    y4[0] = __uint_as_float(yy.x << 16); y4[1] = __uint_as_float(yy.x & 0xffff0000u); y4[2] = __uint_as_float(yy.y << 16); y4[3] = __uint_as_float(yy.y & 0xffff0000u);
                } else if (zmask) {
                    y4[0] = fmaf(z4[0], msc.x, msh.x); y4[1] = fmaf(z4[1], msc.y, msh.y); y4[2] = fmaf(z4[2], msc.z, msh.z); y4[3] = fmaf(z4[3], msc.w, msh.w);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = y4[e] > 0.f ? v[e] : 0.f;
                    s1[e] += g; s2[e] = fmaf(g, z4[e], s2[e]);
                }
            }
        }
    }
    const bool fwd_stats = MODE == 0 && p.stat_acc != nullptr;
    if (fwd_stats || bwd_sums) {
        float sv[8] = {s1[0], s1[1], s1[2], s1[3], s2[0], s2[1], s2[2], s2[3]};
        row16_sum_n(sv);                                     // over the 16 pixels of a lane group: this wave saw all 64 pixels of its channels
        if (fr == 0) {
            const float is4[4] = {pis.x, pis.y, pis.z, pis.w}, mu4[4] = {pmu.x, pmu.y, pmu.z, pmu.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (fwd_stats) {
                    double* a = p.stat_acc + (size_t)(bx & (p.stat_rep - 1)) * 2 * C;
                    atomicAdd(a + c4 + e, (double)sv[e]);
                    atomicAdd(a + C + c4 + e, (double)sv[4 + e]);
                } else {
                    double* a = p.bn_acc + (size_t)(bx & (p.bn_rep - 1)) * 2 * C;
                    atomicAdd(a + c4 + e, (double)sv[e]);
                    atomicAdd(a + C + c4 + e, (double)(is4[e] * (sv[4 + e] - mu4[e] * sv[e])));
                }
            }
        }
    }
}
template <int MODE, int LI = 0, int BM = 64>
__global__ __launch_bounds__(256) void conv64_kernel(Conv3Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    conv64_body<MODE, false, LI, BM>(p, blockIdx.x, smem);
}

template <int WM, int WN, int MODE>
int launch3(Conv3Params& p, hipStream_t st) {
    constexpr int BM = WM * 64, BN = WN * 64;
    p.np = BM + 2 * p.W + 2;
    p.patch_bytes = ((p.np + 1) * PITCH + 255) / 256 * 256;       // + the zero row
    p.nbuf = 1;
    static const int gmode = clhip_cfg("CONV3G") ? atoi(clhip_cfg("CONV3G")) : 0;     // 0: register-staged weights everywhere
    const bool use_g = gmode == 2 || (gmode == 1 && p.M <= 32768);   // LDS-DMA weight ring: correct, not faster yet (r01 profiles) -> opt-in
    size_t lds = (size_t)p.patch_bytes + (use_g ? 4 * (size_t)BN * 128 : 2 * (size_t)BN * PITCH);
    size_t olds = (size_t)BM * (BN * 2 + 16);
    if (olds > lds) lds = olds;
    auto kern = use_g ? conv3g_kernel<WM, WN, MODE> : conv3_kernel<WM, WN, MODE>;
    static size_t attr_lds[2] = {0, 0};
    if (lds > attr_lds[use_g]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            clhip_set_error("conv3: cannot reserve %zu bytes of LDS", lds);
            return CLHIP_EHIP;
        }
        attr_lds[use_g] = lds;
    }
    dim3 grid((p.M + BM - 1) / BM, (p.Cd + BN - 1) / BN);
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, st, p);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

struct Cfg3 { int wm, wn; };
Cfg3 pick3(int M, int Cd) {
    static const char* ov = clhip_cfg("CONV3_CFG");       // tuning override "wm,wn"
    if (ov) { int a = 0, b = 0; if (sscanf(ov, "%d,%d", &a, &b) == 2 && (b == 1 || (b == 2 && Cd >= 128))) return Cfg3{a, b}; }
    // 4-wave workgroups of ~57 KB LDS: two of them share a CU, so one workgroup's patch load / output store
    // overlaps the other's MFMA phase (a single 8-wave workgroup per CU ran load -> compute -> store serially).
    if (Cd < 128) {
        if ((int64_t)(M + 255) / 256 >= 384) return Cfg3{4, 1};
        return Cfg3{2, 1};
    }
    int gy = (Cd + 127) / 128;
    if ((int64_t)((M + 127) / 128) * gy >= 384) return Cfg3{2, 2};
    // small images (8x8, 4x4): the per-tap weight tile dominates the staging work, so favour many pixels x 64
    // channels per workgroup (measured on 256x{8x8x256, 4x4x512}: (4,1) 33/54 us vs (1,2) 43/69 us, r01 profiles)
    if ((int64_t)((M + 255) / 256) * ((Cd + 63) / 64) >= 96) return Cfg3{4, 1};
    return Cfg3{2, 1};
}

}  // namespace

bool clhip_conv16_supported(int H, int W, int Cs, int Cd, int ksize, int stride, int pad, int dtype) {
    static const bool off = clhip_cfg("NO_CONV16") != nullptr;
    return !off && dtype == CLHIP_BF16 && ksize == 3 && stride == 1 && pad == 1 && Cs == Cd && (Cs == 16 || Cs == 32) && W <= 64 && W >= 2 && H >= 1;
}

int clhip_conv16_tiles_m(int M) { return (M + 255) / 256; }

int clhip_conv16_launch_bn(const void* src, const void* wt, void* dst, float* stats, double* stat_acc, int stat_rep, int N, int H, int W, int C, int accumulate, int mode,
                           const void* bn_z, const void* bn_y, const float* bn_mean, const float* bn_invstd, double* bn_acc, int bn_rep, hipStream_t st);
int clhip_conv16_launch_ex(const void* src, const void* wt, void* dst, float* stats, double* stat_acc, int stat_rep, int N, int H, int W, int C, int accumulate, int mode,
                           const void* bn_z, const void* bn_y, const float* bn_mean, const float* bn_invstd, double* bn_acc, int bn_rep, const float* bn_coef,
                           const clhip_bn_input* in, hipStream_t st, const clhip_bn_res_input* rs = nullptr);

int clhip_conv16_launch(const void* src, const void* wt, void* dst, float* stats, double* stat_acc, int stat_rep, int N, int H, int W, int C, int accumulate, int mode,
                        hipStream_t st) {
    return clhip_conv16_launch_bn(src, wt, dst, stats, stat_acc, stat_rep, N, H, W, C, accumulate, mode, nullptr, nullptr, nullptr, nullptr, nullptr, 1, st);
}

int clhip_conv16_launch_bn(const void* src, const void* wt, void* dst, float* stats, double* stat_acc, int stat_rep, int N, int H, int W, int C, int accumulate, int mode,
                           const void* bn_z, const void* bn_y, const float* bn_mean, const float* bn_invstd, double* bn_acc, int bn_rep, hipStream_t st) {
    return clhip_conv16_launch_ex(src, wt, dst, stats, stat_acc, stat_rep, N, H, W, C, accumulate, mode, bn_z, bn_y, bn_mean, bn_invstd, bn_acc, bn_rep, nullptr, nullptr, st);
}

// bn_coef: the dgrad epilogue's ReLU mask from z (bn_y == nullptr); in: forward with a lazy BatchNorm input (src = the producer's z)
int clhip_conv16_launch_ex(const void* src, const void* wt, void* dst, float* stats, double* stat_acc, int stat_rep, int N, int H, int W, int C, int accumulate, int mode,
                           const void* bn_z, const void* bn_y, const float* bn_mean, const float* bn_invstd, double* bn_acc, int bn_rep, const float* bn_coef,
                           const clhip_bn_input* in, hipStream_t st, const clhip_bn_res_input* rs) {
    Conv3Params p;
    p.bn_coef = bn_coef;
    if (rs != nullptr) { p.in_res = static_cast<const bf16_t*>(rs->res); p.in_y = static_cast<bf16_t*>(rs->y); p.in_mask = static_cast<unsigned char*>(rs->relu_mask); }
    if (in != nullptr) {
        p.in_acc = in->stat_acc; p.in_eval = in->stat_acc == nullptr ? 1 : 0; p.in_rep = in->replicas; p.in_gamma = in->gamma; p.in_beta = in->beta; p.in_rm = in->running_mean; p.in_rv = in->running_var;
        p.in_momentum = in->momentum; p.in_eps = in->eps; p.in_mean_o = in->mean; p.in_invstd_o = in->invstd; p.in_coef_o = in->coef;
        const double M = (double)N * H * W;
        p.in_invM = 1.0 / M; p.in_unbias = M > 1.0 ? M / (M - 1.0) : 1.0;
    }
    p.bn_z = static_cast<const bf16_t*>(bn_z); p.bn_y = static_cast<const bf16_t*>(bn_y); p.bn_mean = bn_mean; p.bn_invstd = bn_invstd;
    p.bn_acc = bn_acc; p.bn_rep = bn_rep > 0 ? bn_rep : 1;
    p.src = static_cast<const bf16_t*>(src); p.wt = static_cast<const bf16_t*>(wt); p.dst = static_cast<bf16_t*>(dst);
    p.stats = stats; p.stat_acc = stat_acc; p.stat_rep = stat_rep > 0 ? stat_rep : 1;
    p.N = N; p.H = H; p.W = W; p.wshift = ilog2_exact(W); p.hshift = ilog2_exact(H); p.Cs = C; p.Cd = C; p.accumulate = accumulate; p.M = N * H * W;
    p.np = 256 + 2 * W + 2; p.patch_bytes = (p.np + 1) * (C == 16 ? 32 : 96); p.nbuf = 1; p.debug = 0;
    const size_t lds = ((size_t)p.patch_bytes > 1024 ? (size_t)p.patch_bytes : 1024) + 1024;      // + the lazy operands' coefficient tables ([2][C] / [6][C])
    const dim3 grid(clhip_conv16_tiles_m(p.M));
    if (C == 16) {
        if (mode == 0 && in != nullptr && rs != nullptr) hipLaunchKernelGGL((conv16_kernel<0, 2>), grid, dim3(256), lds, st, p);
        else if (mode == 0 && in != nullptr) hipLaunchKernelGGL((conv16_kernel<0, 1>), grid, dim3(256), lds, st, p);
        else if (mode == 0) hipLaunchKernelGGL(conv16_kernel<0>, grid, dim3(256), lds, st, p);
        else hipLaunchKernelGGL(conv16_kernel<1>, grid, dim3(256), lds, st, p);
    } else {
        if (mode == 0 && in != nullptr && rs != nullptr) hipLaunchKernelGGL((conv32_kernel<0, 2>), grid, dim3(256), lds, st, p);
        else if (mode == 0 && in != nullptr) hipLaunchKernelGGL((conv32_kernel<0, 1>), grid, dim3(256), lds, st, p);
        else if (mode == 0) hipLaunchKernelGGL(conv32_kernel<0>, grid, dim3(256), lds, st, p);
        else hipLaunchKernelGGL(conv32_kernel<1>, grid, dim3(256), lds, st, p);
    }
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

bool clhip_conv64_supported(int N, int H, int W, int Cs, int Cd, int ksize, int stride, int pad, int dtype) {
    static const bool off = clhip_cfg("CONV64") != nullptr && atoi(clhip_cfg("CONV64")) == 0;
    // small maps only: from ~64 k pixels up conv4 / conv5 (LDS-DMA rings, persistent tiles) win; below, the launch is at its latency floor
    static const int max_w = clhip_cfg("CONV64_MAX_W") ? atoi(clhip_cfg("CONV64_MAX_W")) : 16;                 // (experiments: the high-occupancy kernel on the large maps)
    static const long long max_m = clhip_cfg("CONV64_MAX_M") ? atoll(clhip_cfg("CONV64_MAX_M")) : 32768;
    return !off && dtype == CLHIP_BF16 && ksize == 3 && stride == 1 && pad == 1 && Cs == 64 && Cd == 64 && W <= max_w && W >= 2 && H >= 1 && (int64_t)N * H * W <= max_m;
}

int clhip_conv64_launch_ex(const void* src, const void* wt, void* dst, double* stat_acc, int stat_rep, int N, int H, int W, int accumulate, int mode,
                           const void* bn_z, const void* bn_y, const float* bn_mean, const float* bn_invstd, double* bn_acc, int bn_rep, const float* bn_coef,
                           const clhip_bn_input* in, hipStream_t st, const clhip_bn_res_input* rs) {
    Conv3Params p;
    p.bn_coef = bn_coef;
    if (rs != nullptr) { p.in_res = static_cast<const bf16_t*>(rs->res); p.in_y = static_cast<bf16_t*>(rs->y); p.in_mask = static_cast<unsigned char*>(rs->relu_mask); }
    if (in != nullptr) {
        p.in_acc = in->stat_acc; p.in_eval = in->stat_acc == nullptr ? 1 : 0; p.in_rep = in->replicas; p.in_gamma = in->gamma; p.in_beta = in->beta; p.in_rm = in->running_mean; p.in_rv = in->running_var;
        p.in_momentum = in->momentum; p.in_eps = in->eps; p.in_mean_o = in->mean; p.in_invstd_o = in->invstd; p.in_coef_o = in->coef;
        const double M = (double)N * H * W;
        p.in_invM = 1.0 / M; p.in_unbias = M > 1.0 ? M / (M - 1.0) : 1.0;
    }
    p.bn_z = static_cast<const bf16_t*>(bn_z); p.bn_y = static_cast<const bf16_t*>(bn_y); p.bn_mean = bn_mean; p.bn_invstd = bn_invstd;
    p.bn_acc = bn_acc; p.bn_rep = bn_rep > 0 ? bn_rep : 1;
    p.src = static_cast<const bf16_t*>(src); p.wt = static_cast<const bf16_t*>(wt); p.dst = static_cast<bf16_t*>(dst);
    p.stats = nullptr; p.stat_acc = stat_acc; p.stat_rep = stat_rep > 0 ? stat_rep : 1;
    p.N = N; p.H = H; p.W = W; p.wshift = ilog2_exact(W); p.hshift = ilog2_exact(H); p.Cs = 64; p.Cd = 64; p.accumulate = accumulate; p.M = N * H * W;
    static const int bm_cfg = clhip_cfg("CONV64_BM") ? atoi(clhip_cfg("CONV64_BM")) : 64;       // (128-pixel tiles measured equal on ResNet-32 stage 3)
    const int bm = (bm_cfg == 128 && p.M >= 128 * 128) ? 128 : 64;       // 128-pixel tiles while they still give every second CU a workgroup
    p.np = bm + 2 * W + 2; p.patch_bytes = (p.np + 1) * 144; p.nbuf = 1; p.debug = 0;
    const size_t lds = (size_t)p.patch_bytes + 2048;
    const dim3 grid((p.M + bm - 1) / bm);
    if (bm == 128) {
        if (mode == 0 && in != nullptr && rs != nullptr) hipLaunchKernelGGL((conv64_kernel<0, 2, 128>), grid, dim3(256), lds, st, p);
        else if (mode == 0 && in != nullptr) hipLaunchKernelGGL((conv64_kernel<0, 1, 128>), grid, dim3(256), lds, st, p);
        else if (mode == 0) hipLaunchKernelGGL((conv64_kernel<0, 0, 128>), grid, dim3(256), lds, st, p);
        else hipLaunchKernelGGL((conv64_kernel<1, 0, 128>), grid, dim3(256), lds, st, p);
    } else {
        if (mode == 0 && in != nullptr && rs != nullptr) hipLaunchKernelGGL((conv64_kernel<0, 2>), grid, dim3(256), lds, st, p);
        else if (mode == 0 && in != nullptr) hipLaunchKernelGGL((conv64_kernel<0, 1>), grid, dim3(256), lds, st, p);
        else if (mode == 0) hipLaunchKernelGGL(conv64_kernel<0>, grid, dim3(256), lds, st, p);
        else hipLaunchKernelGGL(conv64_kernel<1>, grid, dim3(256), lds, st, p);
    }
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

bool clhip_conv3_supported(int H, int W, int Cs, int Cd, int ksize, int stride, int pad, int dtype) {
    return dtype == CLHIP_BF16 && ksize == 3 && stride == 1 && pad == 1 && (Cs % 64) == 0 && (Cd % 64) == 0 && W <= 32 && W >= 2 && H >= 1;
}

int clhip_conv3_tiles_m(int M, int Cd) { return (M + pick3(M, Cd).wm * 64 - 1) / (pick3(M, Cd).wm * 64); }

int clhip_conv3_launch(const void* src, const void* wt, void* dst, float* stats, double* stat_acc, int stat_rep, int N, int H, int W, int Cs, int Cd, int accumulate,
                       int mode, hipStream_t st) {
    Conv3Params p;
    p.src = static_cast<const bf16_t*>(src); p.wt = static_cast<const bf16_t*>(wt); p.dst = static_cast<bf16_t*>(dst);
    p.stats = stats; p.stat_acc = stat_acc; p.stat_rep = stat_rep > 0 ? stat_rep : 1; p.N = N; p.H = H; p.W = W; p.wshift = ilog2_exact(W); p.hshift = ilog2_exact(H); p.Cs = Cs; p.Cd = Cd; p.accumulate = accumulate; p.M = N * H * W;
    static const int dbg = clhip_cfg("CONV3_DEBUG") ? atoi(clhip_cfg("CONV3_DEBUG")) : 0;
    p.debug = dbg;
    Cfg3 c = pick3(p.M, Cd);
#define L3(a, b) (mode == 0 ? launch3<a, b, 0>(p, st) : launch3<a, b, 1>(p, st))
    if (c.wn == 1) {
        if (c.wm == 8) return L3(8, 1);
        if (c.wm == 4) return L3(4, 1);
        return L3(2, 1);
    }
    if (c.wm == 4) return L3(4, 2);
    if (c.wm == 2) return L3(2, 2);
    return L3(1, 2);
#undef L3
}

// =============================================================================================== wgrad3
// dw[o][tap][c] += sum_p dz[p][o] * x[p @ tap][c] for 3x3 / stride 1 / pad 1 layers.
//
// The first two generations spent their time outside the matrix pipe (profiles/r01_wgrad_ablation.txt: 61 of
// 112 us with the MFMAs removed, 46 us in fp32 atomics).  This version is built around three MI355X facts:
//  * LDS is big enough to hold a zero-PADDED copy of the input rows a 64-pixel step touches
//    ((R+2) x (W+2) pixels x 64 channels), so all 9 taps read the same resident patch at a constant address
//    offset (dh*(W+2)+dw) and image borders need no masking at all;
//  * ds_read_b64_tr_b16 delivers the pixel-major tiles transposed, i.e. directly as MFMA operands whose
//    reduction index is the pixel;
//  * one workgroup owns a 64(out) x 64(in) x 9(tap) block of dW in 72 accumulator tiles (8 waves x 18) and
//    walks a long pixel range, so the fp32 atomics at the end are amortised over >= 16 steps
//    (grid = ~1 workgroup per CU instead of ~6 short ones).
namespace {

struct Wgrad3Params {
    const bf16_t* x;    // [N,H,W,C]
    const bf16_t* dz;   // [N,H,W,K]
    float* dw;          // [K][9][Creal]
    float* slab;        // [splits][K][9][C] fp32 partials (deterministic path) or nullptr (atomics into dw)
    int N, H, W, C, Creal, K, M;
    int R, nimg;        // a 64-pixel step = nimg images x R rows x W cols
    int steps_per_split;
    int npatch;         // nimg*(R+2)*(W+2)
};

constexpr int P3 = 144;      // LDS pitch of a 64-channel (128 B) row: 4 consecutive rows fall in disjoint bank ranges

// two transposing 8-byte reads -> 8 reduction elements; `second` = byte distance of reduction elements k+4..k+7
__device__ __forceinline__ uint4 tr8(const char* base, int addr, int second) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + addr));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + addr + second));
    uint2 l = __builtin_bit_cast(uint2, lo), h = __builtin_bit_cast(uint2, hi);
    return make_uint4(l.x, l.y, h.x, h.y);
}

__global__ __launch_bounds__(512) void conv_wgrad3_kernel(Wgrad3Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int oh = wave >> 2, ct = wave & 3;            // wave -> (32-row half of the out-channel tile, 16-channel column)
    const int fr = lane & 15, fg = lane >> 4;
    const int c0 = blockIdx.x * 64, o0 = blockIdx.y * 64;
    const int W = p.W, H = p.H, PW = W + 2, R = p.R;
    const int zbytes = 64 * P3, xbytes = p.npatch * P3;
    const int stage_bytes = (zbytes + xbytes + 255) / 256 * 256;
    const int s_beg = blockIdx.z * p.steps_per_split;
    const int total_steps = (p.M + 63) / 64;
    const int s_end = min(total_steps, s_beg + p.steps_per_split);

    // ---- fragment addresses.  Reduction element k of a step is pixel (img, row, col) = (k/(R*W), (k%(R*W))/W, k%W).
    //      Lane (fr, fg) feeds k = ks*32 + fg*8 + {0..7}; a transposing read fetches 4 consecutive k (same image row).
    int zaddr[2], xaddr[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int k = ks * 32 + fg * 8 + (fr >> 2);
        zaddr[ks] = k * P3 + (oh * 32 + (fr & 3) * 4) * 2;
        const int img = k / (R * W), rem = k - img * (R * W), rr = rem / W, cc = rem - rr * W;
        // NB the "+4" second read of tr8 addresses pixel k+4: same row, 4 columns further -> +4 patch pixels
        xaddr[ks] = ((img * (R + 2) + rr + 1) * PW + cc + 1) * P3 + (ct * 16 + (fr & 3) * 4) * 2;
    }

    // pixels k+4..k+7 sit 4 columns further in the same patch row, except for 4-pixel-wide images (next row)
    const int xsecond = (W >= 8 ? 4 : PW) * P3;

    // ---- staging assignments (fixed per thread)
    // dz tile: 64 pixels x 8 chunks = 512 chunks -> one per thread
    const int zk = tid >> 3, zch = tid & 7;
    // patch: npatch pixels x 8 chunks, up to 3 per thread
    int pp_pix[3], pp_img[3], pp_row[3], pp_col[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int idx = tid + i * 512;
        const int pix = idx >> 3;
        pp_pix[i] = pix < p.npatch ? pix : -1;
        const int per = (R + 2) * PW;
        const int img = pix / per, rem = pix - img * per;
        pp_img[i] = img; pp_row[i] = rem / PW - 1; pp_col[i] = rem % PW - 1;
    }
    const int pch = tid & 7;
    uint4 rz, rx[3];
    auto gload = [&](int s) {
        const int p0 = s * 64;
        const int pz = p0 + zk;
        rz = pz < p.M ? *reinterpret_cast<const uint4*>(p.dz + ((size_t)pz * p.K + o0 + zch * 8)) : make_uint4(0, 0, 0, 0);
        const int n0 = p0 / (W * H), h0 = (p0 / W) % H;       // first image / first row of this step
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            bool ok = pp_pix[i] >= 0;
            const int n = n0 + pp_img[i], h = h0 + pp_row[i], w = pp_col[i];
            ok = ok && n < p.N && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
            rx[i] = ok ? *reinterpret_cast<const uint4*>(p.x + (((size_t)n * H + h) * W + w) * p.C + c0 + pch * 8) : make_uint4(0, 0, 0, 0);
        }
    };
    auto sstore = [&](int stage) {
        char* zs = smem + stage * stage_bytes;
        char* xs = zs + zbytes;
        *reinterpret_cast<uint4*>(zs + zk * P3 + zch * 16) = rz;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (pp_pix[i] >= 0) *reinterpret_cast<uint4*>(xs + pp_pix[i] * P3 + pch * 16) = rx[i];
    };

    f32x4 acc[2][9];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (s_beg < s_end) {
        gload(s_beg);
        sstore(0);
        __syncthreads();
        int st = 0;
        for (int s = s_beg; s < s_end; ++s, st ^= 1) {
            const bool more = s + 1 < s_end;
            if (more) gload(s + 1);
            const char* zs = smem + st * stage_bytes;
            const char* xs = zs + zbytes;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 zf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) zf[i] = tr8(zs, zaddr[ks] + i * 32, 4 * P3);      // out-channel tiles oh*32 + i*16
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int r = t / 3, sx = t - 3 * r;
                    const uint4 xf = tr8(xs, xaddr[ks] + ((r - 1) * PW + (sx - 1)) * P3, xsecond);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, zf[i]),
                                                                           __builtin_bit_cast(bf16x8_t, xf), acc[i][t], 0, 0, 0);
                }
            }
            if (more) sstore(st ^ 1);
            __syncthreads();
        }
    }

    // D[row = out channel fg*4+e][col = in channel fr]
    const int c = c0 + ct * 16 + fr;
    if (p.slab != nullptr) {
        // deterministic path: plain stores of this split's partial block; wgrad3_reduce_kernel sums the splits
        float* out = p.slab + (size_t)blockIdx.z * p.K * 9 * p.C;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int o = o0 + oh * 32 + i * 16 + fg * 4 + e;
                    out[((size_t)o * 9 + t) * p.C + c] = acc[i][t][e];
                }
    } else if (c < p.Creal) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int o = o0 + oh * 32 + i * 16 + fg * 4 + e;
                    if (o < p.K) atomicAdd(p.dw + ((size_t)o * 9 + t) * p.Creal + c, acc[i][t][e]);
                }
    }
}

// dw[i] += sum_s slab[s][i] in a fixed order (bitwise reproducible weight gradients).  blockIdx.y splits the
// split range into 4 groups whose partial sums are combined through LDS, 8 independent loads in flight per lane.
__global__ __launch_bounds__(256) void wgrad3_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int64_t n4, int splits) {
    __shared__ float4 part[4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
        const int per = (splits + 3) / 4;
        const int s0 = grp * per, s1 = min(splits, s0 + per);
        const float4* base = reinterpret_cast<const float4*>(slab) + i;
        int s = s0;
        for (; s + 8 <= s1; s += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = base[(size_t)(s + u) * n4];
#pragma unroll
            for (int u = 0; u < 8; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
        }
        for (; s < s1; ++s) { const float4 v = base[(size_t)s * n4]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    }
    part[grp][lane] = a;
    __syncthreads();
    if (grp == 0 && i < n4) {
        float4 d = reinterpret_cast<const float4*>(dw)[i];
#pragma unroll
        for (int g = 0; g < 4; ++g) { const float4 v = part[g][lane]; d.x += v.x; d.y += v.y; d.z += v.z; d.w += v.w; }
        reinterpret_cast<float4*>(dw)[i] = d;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// wgrad16: dw[o][tap][c] for 16 -> 16 channels on 32-pixel-wide images (ResNet-32 stage 1; the generic kernel took 44 us for a
// 1.2-GFLOP reduction whose result is 2304 numbers).  One workgroup = one image: the zero-padded 34 x 34 input image and the
// 32 x 32 output-gradient image sit in LDS (69 KB); an image row is exactly one K = 32 step, both operands are pixel-major
// columns fetched with transposing reads (a 16-lane group reads 4 pixels x 16 channels and lane i keeps channel i), the 9 taps
// are constant address offsets into the padded image.  Each of the 4 waves takes 8 rows into 9 accumulator tiles; the waves are
// summed through LDS and the image's 2304 partial sums go to a slab that wgrad3_reduce_kernel adds up in a fixed order.
namespace {
struct Wgrad16Params { const bf16_t* x; const bf16_t* dz; float* slab; int N, H; const float* x_coef = nullptr; LazyDz lz; int parts = 1; };      // parts: workgroups per image (row bands)      // x_coef [2][16]: x is a pre-BatchNorm tensor, the operand relu(scale x + shift)

template <bool LZ = false>
__device__ __forceinline__ void wgrad16_body(const Wgrad16Params& p, const int bx, char* smem) {      // bx = image * parts + row band
    constexpr int W = 32, PW = 34, PX = 32;                  // image width, padded width, bytes per pixel (16 bf16)
    const int H = p.H, RH = H / p.parts;                     // a workgroup owns RH rows of one image (two bands per 32-row image: twice the workgroups,
    const int im = bx / p.parts, r0 = (bx - im * p.parts) * RH;      // half the serial staging -> multiply -> reduce chain each; the halo rows come from the neighbour band)
    char* xs = smem;                                         // (RH + 2) x 34 pixels
    char* zs = smem + (RH + 2) * PW * PX;                    // RH x 32 pixels
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const size_t img = (size_t)im * H * W;
    // zero the padded band, then drop the real pixels in (rows r0 - 1 .. r0 + RH of the image where they exist); the gradient band is copied as is
    const int xchunks = (RH + 2) * PW * 2;
    for (int i = tid; i < xchunks; i += 256) *reinterpret_cast<uint4*>(xs + i * 16) = make_uint4(0, 0, 0, 0);
    __syncthreads();
    constexpr bool lzd = LZ;
    // lazy gradient: the band's raw dy / z chunks (RH = 16 rows: four rounds) are fetched first and the input band is staged next, so that the
    // coefficient prologue (a memory round trip of its own behind two barriers) overlaps them instead of preceding them
    constexpr int PFZ = 4;
    const int nzc = RH * W * 2;
    uint4 pdy[PFZ], pz[PFZ];
    unsigned pm[PFZ];
    if constexpr (lzd) {
#pragma unroll
        for (int it = 0; it < PFZ; ++it) {
            const int i = tid + it * 256;
            pdy[it] = make_uint4(0, 0, 0, 0); pz[it] = pdy[it]; pm[it] = 0u;
            if (i < nzc) {
                const size_t pix = img + (size_t)r0 * W + (i >> 1);
                pdy[it] = *reinterpret_cast<const uint4*>(p.lz.dy + pix * 16 + (i & 1) * 8);
                pz[it] = *reinterpret_cast<const uint4*>(p.lz.z + pix * 16 + (i & 1) * 8);
                if (p.lz.mask != nullptr) pm[it] = p.lz.mask[pix * 2 + (i & 1)];
            }
        }
    }
    float xsc[8], xsh[8];
    if (p.x_coef != nullptr) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { xsc[e] = p.x_coef[(tid & 1) * 8 + e]; xsh[e] = p.x_coef[16 + (tid & 1) * 8 + e]; }
    }
    for (int i = tid; i < (RH + 2) * W * 2; i += 256) {
        const int lp = i >> 1, half = i & 1, lr = lp >> 5, c = lp & 31;      // local padded row lr = image row r0 - 1 + lr
        const int row = r0 - 1 + lr;
        if (row < 0 || row >= H) continue;
        uint4 xv = *reinterpret_cast<const uint4*>(p.x + (img + (size_t)row * W + c) * 16 + half * 8);
        if (p.x_coef != nullptr) xv = bn_relu8_bf16(xv, xsc, xsh);
        *reinterpret_cast<uint4*>(xs + (lr * PW + c + 1) * PX + half * 16) = xv;
    }
    LazyDz8 lt;
    if constexpr (lzd) {                                     // (the gradient staging area is still free: the coefficient table sits at its start until the fill)
        float* coef = reinterpret_cast<float*>(zs);
        lazy_dz_coefs<16>(p.lz, false, coef);
        lazy_dz_load<16>(coef, (tid & 1) * 8, lt);
        __syncthreads();
#pragma unroll
        for (int it = 0; it < PFZ; ++it) {
            const int i = tid + it * 256;
            if (i < nzc) *reinterpret_cast<uint4*>(zs + (i >> 1) * PX + (i & 1) * 16) = lazy_dz8(pdy[it], pz[it], lt, p.lz.mask != nullptr, pm[it]);
        }
    }
    for (int i = lzd ? tid + PFZ * 256 : tid; i < nzc; i += 256) {
        const int lp = i >> 1, half = i & 1;
        const size_t pix = img + (size_t)r0 * W + lp;
        uint4 zv;
        if constexpr (lzd) zv = lazy_dz8(*reinterpret_cast<const uint4*>(p.lz.dy + pix * 16 + half * 8), *reinterpret_cast<const uint4*>(p.lz.z + pix * 16 + half * 8), lt,
                                         p.lz.mask != nullptr, p.lz.mask != nullptr ? p.lz.mask[pix * 2 + half] : 0u);
        else zv = *reinterpret_cast<const uint4*>(p.dz + pix * 16 + half * 8);
        *reinterpret_cast<uint4*>(zs + lp * PX + half * 16) = zv;
    }
    __syncthreads();

    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // lane (fr, fg) of a transposing read addresses pixel column 8 fg + (fr >> 2) (+4 for the second read), segment fr & 3
    const int col = fg * 8 + (fr >> 2), seg = (fr & 3) * 8;
    for (int h = wave; h < RH; h += 4) {
        const uint4 zf = tr8(zs, (h * W + col) * PX + seg, 4 * PX);
        const int xb = ((h + 1) * PW + col + 1) * PX + seg;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int r = t / 3, sx = t - 3 * r;
            const uint4 xf = tr8(xs, xb + ((r - 1) * PW + (sx - 1)) * PX, 4 * PX);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, zf), __builtin_bit_cast(bf16x8_t, xf), acc[t], 0, 0, 0);
        }
    }
    // D[row = out channel fg*4 + e][col = in channel fr]  ->  red[wave][(o*9 + t)*16 + c]
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave * 2304 + ((fg * 4 + e) * 9 + t) * 16 + fr] = acc[t][e];
    __syncthreads();
    float* out = p.slab + (size_t)bx * 2304;
    for (int i = tid; i < 2304; i += 256) out[i] = red[i] + red[2304 + i] + red[4608 + i] + red[6912 + i];
}
__global__ __launch_bounds__(256) void wgrad16_kernel(Wgrad16Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    wgrad16_body(p, blockIdx.x, smem);
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------
// wgrad32: dw[o][tap][c] for 32 -> 32 channels on 16-pixel-wide images (ResNet-32 stage 2: nine launches per step, until round 3 on the
// generic kernel's fp32 atomics -- the largest symbol of both ResNet-32 workloads, and not reproducible).  A per-image sibling of
// wgrad16 was measured at 30.7 us in round 2: an image's 9216 partial sums are more bytes than the image.  Here the OUTPUT channels are
// split over two workgroups (blockIdx.x = 16-channel out tile) and a workgroup walks a GROUP of images (N / 64 of them, so that ~128
// workgroups exist and the partial blocks total 128 x 18 KB): the zero-padded 18 x 18 x 32 input image and its 16 x 16 x 16 slice of the
// gradient sit in LDS, two image rows are one K = 32 step, both operands come from transposing reads, the taps are constant offsets.
// Wave w owns in-channel tile (w & 1) and taps {0..4} / {5..8} (w >> 1) over ALL K steps -- no cross-wave sum -- and the next image's
// global loads are in flight while the current one is multiplied.  Partial blocks + wgrad3_reduce_kernel: bitwise reproducible.
namespace {
struct Wgrad32Params { const bf16_t* x; const bf16_t* dz; float* slab; int N, H, img_per_group; const float* x_coef = nullptr; LazyDz lz; };      // x_coef [2][32] as in Wgrad16Params

template <bool LZ = false>
__device__ __forceinline__ void wgrad32_body(const Wgrad32Params& p, const int ot, const int grp, char* smem) {
    constexpr int W = 16, PW = 18, PX = 64, PZ = 32;          // image width, padded width, bytes per input pixel (32 bf16), per gradient pixel (this tile's 16)
    const int H = p.H, HW = H * W;
    char* xs = smem;                                          // (H + 2) x 18 pixels
    char* zs = smem + (H + 2) * PW * PX;                      // H x 16 pixels
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int it = wave & 1, th = wave >> 1;                  // in-channel tile, tap half
    const int n_beg = grp * p.img_per_group, n_end = min(p.N, n_beg + p.img_per_group);
    // zero the padded image once: only its interior is rewritten per image
    for (int i = tid; i < (H + 2) * PW * 4; i += 256) *reinterpret_cast<uint4*>(xs + i * 16) = make_uint4(0, 0, 0, 0);
    const int nx = HW * 4, nz = HW * 2;                       // 16-byte chunks per image: input (4 per pixel), gradient slice (2 per pixel)
    uint4 rx[4], rz[2], ry[2];
    unsigned rm[2] = {0u, 0u};
    constexpr bool lzd = LZ;
    LazyDz8 lt;
    float xsc[8], xsh[8];                                     // chunk q = tid + 256 i covers channels (q & 3) * 8 = (tid & 3) * 8 of its pixel
    if (p.x_coef != nullptr) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { xsc[e] = p.x_coef[(tid & 3) * 8 + e]; xsh[e] = p.x_coef[32 + (tid & 3) * 8 + e]; }
    }
    auto gload = [&](int n) {
        const size_t base = (size_t)n * HW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 256 * i;
            rx[i] = q < nx ? *reinterpret_cast<const uint4*>(p.x + (base + (q >> 2)) * 32 + (q & 3) * 8) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = tid + 256 * i;
            if constexpr (lzd) {
                rz[i] = q < nz ? *reinterpret_cast<const uint4*>(p.lz.dy + (base + (q >> 1)) * 32 + ot * 16 + (q & 1) * 8) : make_uint4(0, 0, 0, 0);
                ry[i] = q < nz ? *reinterpret_cast<const uint4*>(p.lz.z + (base + (q >> 1)) * 32 + ot * 16 + (q & 1) * 8) : make_uint4(0, 0, 0, 0);
                rm[i] = (q < nz && p.lz.mask != nullptr) ? p.lz.mask[(base + (q >> 1)) * 4 + ot * 2 + (q & 1)] : 0u;
            } else {
                rz[i] = q < nz ? *reinterpret_cast<const uint4*>(p.dz + (base + (q >> 1)) * 32 + ot * 16 + (q & 1) * 8) : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto lazy_or = [&](uint4 a, uint4 b, unsigned mb) { if constexpr (lzd) return lazy_dz8(a, b, lt, p.lz.mask != nullptr, mb); else { (void)b; (void)mb; return a; } };
    auto sstore = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 256 * i;
            if (q < nx) {
                const int px = q >> 2;
                *reinterpret_cast<uint4*>(xs + ((px / W + 1) * PW + (px % W) + 1) * PX + (q & 3) * 16) = p.x_coef != nullptr ? bn_relu8_bf16(rx[i], xsc, xsh) : rx[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = tid + 256 * i;
            if (q < nz) *reinterpret_cast<uint4*>(zs + (q >> 1) * PZ + (q & 1) * 16) = lazy_or(rz[i], ry[i], rm[i]);
        }
    };
    f32x4 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // lane (fr, fg) of a transposing read: reduction element 8 fg + (fr >> 2) (+4 for the second read) = pixel (row h0 + (fg >> 1),
    // column 8 (fg & 1) + (fr >> 2)), 8-byte segment fr & 3 of the 16-channel tile
    const int prow = fg >> 1, pcol = (fg & 1) * 8 + (fr >> 2), seg = (fr & 3) * 8;
    const int t0 = th * 5, nt = th == 0 ? 5 : 4;
    if (n_beg < n_end) gload(n_beg);                          // the first image's raw operands are in flight while the coefficient prologue (a round trip of its own) runs
    if constexpr (lzd) {                                      // gradient chunk q = tid + 256 i covers channels ot * 16 + (tid & 1) * 8
        float* coef = reinterpret_cast<float*>(zs);
        lazy_dz_coefs<32>(p.lz, false, coef);
        lazy_dz_load<32>(coef, ot * 16 + (tid & 1) * 8, lt);
    }
    for (int n = n_beg; n < n_end; ++n) {
        __syncthreads();                                      // the previous image's reads are done (first pass: the zero fill)
        sstore();
        __syncthreads();
        if (n + 1 < n_end) gload(n + 1);
        for (int h0 = 0; h0 < H; h0 += 2) {
            const uint4 zf = tr8(zs, ((h0 + prow) * W + pcol) * PZ + seg, 4 * PZ);
            const int xb = ((h0 + prow) * PW + pcol) * PX + it * 32 + seg;       // padded coordinates: tap (r, s) adds r rows, s columns
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                if (q < nt) {
                    const int t = t0 + q, r = t / 3, sx = t - 3 * r;
                    const uint4 xf = tr8(xs, xb + (r * PW + sx) * PX, 4 * PX);
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, zf), __builtin_bit_cast(bf16x8_t, xf), acc[q], 0, 0, 0);
                }
            }
        }
    }
    // D[row = out channel fg*4 + e][col = in channel fr]  ->  slab[grp][o][tap][c]
    float* out = p.slab + (size_t)grp * 9216;
#pragma unroll
    for (int q = 0; q < 5; ++q)
        if (q < nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) out[((ot * 16 + fg * 4 + e) * 9 + t0 + q) * 32 + it * 16 + fr] = acc[q][e];
}
__global__ __launch_bounds__(256) void wgrad32_kernel(Wgrad32Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    wgrad32_body(p, blockIdx.x, blockIdx.y, smem);
}

// wgrad64: dw[o][tap][c] for 64 -> 64 channels on 8-pixel-wide images (ResNet-32 stage 3).  Workgroup (out-channel tile ot of 16, image group):
// wave w owns in-channel tile w and all nine taps (9 accumulator tiles), an MFMA K step = 32 pixels = four image rows; the padded image
// (H + 2) x 10 pixels x 64 channels and the tile's 16 gradient channels sit in LDS, the next image's loads are in flight while this one is
// multiplied.  Partial blocks per group + the fixed-order reduce: bitwise reproducible.
struct Wgrad64Params { const bf16_t* x; const bf16_t* dz; float* slab; int N, H, img_per_group; const float* x_coef = nullptr; LazyDz lz; };

constexpr int WG64_IPI = 2;      // (default; WGRAD64_IPI2=1 keeps one) images staged and multiplied per barrier pair: an image is two K steps (18 MFMAs per wave) -- the per-image cost was the two barriers and the
                                 // LDS round trip, eight times in a row per workgroup at batch 256
template <bool LZ = false, int IPI = WG64_IPI>
__device__ __forceinline__ void wgrad64_body(const Wgrad64Params& p, const int ot, const int grp, char* smem) {
    constexpr int W = 8, PW = 10, PX = 144, PZ = 32, C = 64;  // image width, padded width, bytes per staged input pixel (128 + pad), per gradient pixel (this tile's 16 channels)
    const int H = p.H, HW = H * W;
    const int xsz = (H + 2) * PW * PX, zsz = HW * PZ;
    char* xs = smem;                                          // IPI x (H + 2) x 10 pixels
    char* zs = smem + IPI * xsz;                              // IPI x H x 8 pixels
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int it = wave;                                      // in-channel tile
    const int n_beg = grp * p.img_per_group, n_end = min(p.N, n_beg + p.img_per_group);
    for (int i = tid; i < IPI * xsz / 16; i += 256) *reinterpret_cast<uint4*>(xs + i * 16) = make_uint4(0, 0, 0, 0);
    const int nx = HW * 8, nz = HW * 2;                       // 16-byte chunks per image: input (8 per pixel), gradient slice (2 per pixel)
    constexpr int NXI = 4;                                    // input chunks per thread (H <= 16)
    uint4 rx[IPI][NXI], rz[IPI], ry[IPI];
    unsigned rm[IPI];
    constexpr bool lzd = LZ;
    LazyDz8 lt;
    float xsc[8], xsh[8];                                     // input chunk q = tid + 256 i covers channels (q & 7) * 8 = (tid & 7) * 8
    if (p.x_coef != nullptr) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { xsc[e] = p.x_coef[(tid & 7) * 8 + e]; xsh[e] = p.x_coef[C + (tid & 7) * 8 + e]; }
    }
    auto gload = [&](int n0) {
#pragma unroll
        for (int j = 0; j < IPI; ++j) {
            const bool iv = n0 + j < n_end;
            const size_t base = (size_t)(iv ? n0 + j : n0) * HW;
#pragma unroll
            for (int i = 0; i < NXI; ++i) {
                const int q = tid + 256 * i;
                rx[j][i] = (iv && q < nx) ? *reinterpret_cast<const uint4*>(p.x + (base + (q >> 3)) * C + (q & 7) * 8) : make_uint4(0, 0, 0, 0);
            }
            const int q = tid;
            rm[j] = 0u;
            if constexpr (lzd) {
                rz[j] = (iv && q < nz) ? *reinterpret_cast<const uint4*>(p.lz.dy + (base + (q >> 1)) * C + ot * 16 + (q & 1) * 8) : make_uint4(0, 0, 0, 0);
                ry[j] = (iv && q < nz) ? *reinterpret_cast<const uint4*>(p.lz.z + (base + (q >> 1)) * C + ot * 16 + (q & 1) * 8) : make_uint4(0, 0, 0, 0);
                rm[j] = (iv && q < nz && p.lz.mask != nullptr) ? p.lz.mask[(base + (q >> 1)) * 8 + ot * 2 + (q & 1)] : 0u;
            } else {
                rz[j] = (iv && q < nz) ? *reinterpret_cast<const uint4*>(p.dz + (base + (q >> 1)) * C + ot * 16 + (q & 1) * 8) : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto sstore = [&](int n0) {
#pragma unroll
        for (int j = 0; j < IPI; ++j) {
            if (n0 + j >= n_end) break;                       // (its K steps are skipped as well)
#pragma unroll
            for (int i = 0; i < NXI; ++i) {
                const int q = tid + 256 * i;
                if (q < nx) {
                    const int px = q >> 3;
                    *reinterpret_cast<uint4*>(xs + j * xsz + ((px / W + 1) * PW + (px % W) + 1) * PX + (q & 7) * 16) = p.x_coef != nullptr ? bn_relu8_bf16(rx[j][i], xsc, xsh) : rx[j][i];
                }
            }
            const int q = tid;
            if (q < nz) {
                uint4 v = rz[j];
                if constexpr (lzd) v = lazy_dz8(rz[j], ry[j], lt, p.lz.mask != nullptr, rm[j]);
                *reinterpret_cast<uint4*>(zs + j * zsz + (q >> 1) * PZ + (q & 1) * 16) = v;
            }
        }
    };
    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // lane (fr, fg) of a transposing read: reduction element 8 fg + (fr >> 2) (+4 for the second read) = pixel (row h0 + fg, column (fr >> 2) [+ 4]),
    // 8-byte segment fr & 3 of the 16-channel tile
    const int prow = fg, pcol = fr >> 2, seg = (fr & 3) * 8;
    if (n_beg < n_end) gload(n_beg);                          // in flight while the coefficient prologue runs (see wgrad32_body)
    if constexpr (lzd) {                                      // gradient chunk q = tid covers channels ot * 16 + (tid & 1) * 8
        float* coef = reinterpret_cast<float*>(zs);
        lazy_dz_coefs<C>(p.lz, false, coef);
        lazy_dz_load<C>(coef, ot * 16 + (tid & 1) * 8, lt);
    }
    for (int n = n_beg; n < n_end; n += IPI) {
        __syncthreads();
        sstore(n);
        __syncthreads();
        if (n + IPI < n_end) gload(n + IPI);
#pragma unroll
        for (int j = 0; j < IPI; ++j) {
            if (n + j >= n_end) break;
            for (int h0 = 0; h0 < H; h0 += 4) {
                const uint4 zf = tr8(zs + j * zsz, ((h0 + prow) * W + pcol) * PZ + seg, 4 * PZ);
                const int xb = ((h0 + prow) * PW + pcol) * PX + it * 32 + seg;       // padded coordinates: tap (r, s) adds r rows, s columns
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int r = t / 3, sx = t - 3 * r;
                    const uint4 xf = tr8(xs + j * xsz, xb + (r * PW + sx) * PX, 4 * PX);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, zf), __builtin_bit_cast(bf16x8_t, xf), acc[t], 0, 0, 0);
                }
            }
        }
    }
    // D[row = out channel fg*4 + e][col = in channel fr]  ->  slab[grp][o][tap][c]
    float* out = p.slab + (size_t)grp * (C * 9 * C);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) out[((ot * 16 + fg * 4 + e) * 9 + t) * C + it * 16 + fr] = acc[t][e];
}
template <int IPI>
__global__ __launch_bounds__(256) void wgrad64_kernel(Wgrad64Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    wgrad64_body<false, IPI>(p, blockIdx.x, blockIdx.y, smem);
}

int wgrad32_ipg(int N) {
    static const int forced = clhip_cfg("WGRAD32_IPG") ? atoi(clhip_cfg("WGRAD32_IPG")) : 0;
    if (forced > 0) return forced < N ? forced : N;
    return N >= 128 ? N / 64 : (N >= 32 ? 2 : 1);
}
int wgrad32_groups(int N) { const int ipg = wgrad32_ipg(N); return (N + ipg - 1) / ipg; }
}  // namespace

bool clhip_wgrad32_supported(int N, int H, int W, int C, int Creal, int K, int ksize, int stride, int pad, int dtype) {
    static const bool off = clhip_cfg("WGRAD32") != nullptr && atoi(clhip_cfg("WGRAD32")) == 0;
    return !off && dtype == CLHIP_BF16 && ksize == 3 && stride == 1 && pad == 1 && C == 32 && Creal == 32 && K == 32 && W == 16 && H >= 2 && H <= 32 && (H & 1) == 0 && N >= 1;
}

size_t clhip_wgrad32_ws_bytes(int N) { return (size_t)wgrad32_groups(N) * 9216 * sizeof(float); }

int clhip_wgrad_reduce_launch(const float* slab, float* dw, int64_t n4, int splits, hipStream_t st);

int clhip_wgrad32_launch(const void* x, const void* dz, float* dw, float* ws, int N, int H, const float* x_coef, hipStream_t st) {
    const int groups = wgrad32_groups(N);
    Wgrad32Params p{static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(dz), ws, N, H, (N + groups - 1) / groups, x_coef};
    // (groups was derived from the same images-per-group rule: recompute the per-group count exactly)
    p.img_per_group = wgrad32_ipg(N);
    const size_t lds = (size_t)(H + 2) * 18 * 64 + (size_t)H * 16 * 32;
    hipLaunchKernelGGL(wgrad32_kernel, dim3(2, groups), dim3(256), lds, st, p);
    CLHIP_LAUNCH_CHECK();
    return clhip_wgrad_reduce_launch(ws, dw, 2304, groups, st);
}

bool clhip_wgrad64_supported(int N, int H, int W, int C, int Creal, int K, int ksize, int stride, int pad, int dtype) {
    static const bool off = clhip_cfg("CONV64") != nullptr && atoi(clhip_cfg("CONV64")) == 0;
    return !off && dtype == CLHIP_BF16 && ksize == 3 && stride == 1 && pad == 1 && C == 64 && Creal == 64 && K == 64 && W == 8 && H >= 4 && H <= 16 && (H & 3) == 0 && N >= 1;
}

static bool wgrad64_pair_images() { static const bool one = clhip_cfg("WGRAD64_IPI2") != nullptr && atoi(clhip_cfg("WGRAD64_IPI2")) == 1; return !one; }
static int wgrad64_ipg(int N) {
    static const int forced = clhip_cfg("WGRAD64_IPG") ? atoi(clhip_cfg("WGRAD64_IPG")) : 0;
    if (forced > 0) return forced < N ? forced : N;
    // a group's partial block is the whole 64 x 9 x 64 fp32 gradient (147 KB): 64 groups = 9.4 MB per layer written and read back by the reduce, nine
    // layers per ResNet-32 step.  Measured on the EWC step at batch 256 (ms): 2 / 4 / 8 / 16 images per group: 1.52 / 1.445 / 1.426 / 1.506
    return N >= 128 ? N / 32 : (N >= 32 ? 2 : 1);
}
static int wgrad64_groups(int N) { const int ipg = wgrad64_ipg(N); return (N + ipg - 1) / ipg; }

size_t clhip_wgrad64_ws_bytes(int N) { return (size_t)wgrad64_groups(N) * 36864 * sizeof(float); }

int clhip_wgrad64_launch(const void* x, const void* dz, float* dw, float* ws, int N, int H, const float* x_coef, hipStream_t st) {
    const int groups = wgrad64_groups(N);
    Wgrad64Params p{static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(dz), ws, N, H, wgrad64_ipg(N), x_coef};
    const size_t lds = WG64_IPI * ((size_t)(H + 2) * 10 * 144 + (size_t)H * 8 * 32) + 2048;
    if (wgrad64_pair_images()) hipLaunchKernelGGL(wgrad64_kernel<WG64_IPI>, dim3(4, groups), dim3(256), lds, st, p);
    else hipLaunchKernelGGL(wgrad64_kernel<1>, dim3(4, groups), dim3(256), lds, st, p);
    CLHIP_LAUNCH_CHECK();
    return clhip_wgrad_reduce_launch(ws, dw, 9216, groups, st);
}

bool clhip_wgrad16_supported(int N, int H, int W, int C, int Creal, int K, int ksize, int stride, int pad, int dtype) {
    static const bool off = clhip_cfg("NO_CONV16") != nullptr;
    return !off && dtype == CLHIP_BF16 && ksize == 3 && stride == 1 && pad == 1 && C == 16 && Creal == 16 && K == 16 && W == 32 && H >= 1 && H <= 64 && N >= 1;
}

// row bands per image (env WGRAD16_PARTS overrides): enough workgroups at small batches, not more partial blocks than pay at large ones
static int wgrad16_parts(int N, int H) {
    static const int forced = clhip_cfg("WGRAD16_PARTS") ? atoi(clhip_cfg("WGRAD16_PARTS")) : 0;
    // measured on the ResNet-32 EWC step (ms): batch 256: 1 band 1.501, 2 bands 1.447, 4 bands 1.510; batch 32: 0.96-1.09 / 1.00-1.04 / 0.86-0.91
    const int want = forced > 0 ? forced : (H == 32 ? (N < 96 ? 4 : 2) : 1);
    if (want >= 4 && (H % 16) == 0) return 4;
    return (want >= 2 && (H % 8) == 0) ? 2 : 1;
}

size_t clhip_wgrad16_ws_bytes(int N) { return (size_t)N * 4 * 2304 * sizeof(float); }      // (room for four bands per image)

bool clhip_wgrad3_supported(int N, int H, int W, int C, int Creal, int K, int ksize, int stride, int pad, int dtype) {
    if (!(dtype == CLHIP_BF16 && ksize == 3 && stride == 1 && pad == 1 && C % 64 == 0 && K % 64 == 0 && Creal == C)) return false;
    if (W < 4 || W > 32 || (W & 3)) return false;
    int hw = H * W;
    if (hw >= 64) return (64 % W == 0) && (H % (64 / W) == 0);
    return 64 % hw == 0;
}

static void wgrad3_geometry(int N, int H, int W, int C, int K, Wgrad3Params& p, int& splits) {
    p.N = N; p.H = H; p.W = W; p.C = C; p.K = K; p.M = N * H * W;
    int hw = H * W;
    if (hw >= 64) { p.nimg = 1; p.R = 64 / W; } else { p.nimg = 64 / hw; p.R = H; }
    p.npatch = p.nimg * (p.R + 2) * (W + 2);
    int tiles = (C / 64) * (K / 64);
    int total_steps = (p.M + 63) / 64;
    static const int target = clhip_cfg("WGRAD_TARGET") ? atoi(clhip_cfg("WGRAD_TARGET")) : 256;
    splits = (target + tiles - 1) / tiles;
    int max_splits = (total_steps + 7) / 8;              // >= 8 steps per workgroup
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    p.steps_per_split = (total_steps + splits - 1) / splits;
    splits = (total_steps + p.steps_per_split - 1) / p.steps_per_split;
}

size_t clhip_wgrad3_ws_bytes(int N, int H, int W, int C, int K) {
    Wgrad3Params p; int splits;
    wgrad3_geometry(N, H, W, C, K, p, splits);
    return (size_t)splits * K * 9 * C * sizeof(float);
}

int clhip_wgrad3_launch(const void* x, const void* dz, float* dw, float* ws, int N, int H, int W, int C, int Creal, int K, hipStream_t st) {
    Wgrad3Params p; int splits;
    wgrad3_geometry(N, H, W, C, K, p, splits);
    p.x = static_cast<const bf16_t*>(x); p.dz = static_cast<const bf16_t*>(dz); p.dw = dw; p.slab = ws; p.Creal = Creal;
    if (p.npatch > 192) { clhip_set_error("wgrad3: patch too large"); return CLHIP_EINVAL; }
    size_t stage = ((size_t)64 * P3 + (size_t)p.npatch * P3 + 255) / 256 * 256;
    size_t lds = 2 * stage;
    static size_t attr = 0;
    if (lds > attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = lds;
    }
    hipLaunchKernelGGL(conv_wgrad3_kernel, dim3(C / 64, K / 64, splits), dim3(512), lds, st, p);
    CLHIP_LAUNCH_CHECK();
    if (ws != nullptr) return clhip_wgrad_reduce_launch(ws, dw, (int64_t)K * 9 * C / 4, splits, st);
    return CLHIP_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Horizontal fusion of a layer's two backward convolutions (round 3).  The input gradient (conv16 / conv32 in dgrad mode) and the weight
// gradient (wgrad16 / wgrad32) of a layer both read dz and depend on nothing else of each other; on CifarResNet-32 each is a launch at
// its latency floor (8-15 us and 10-12 us) on ONE stream -- a second stream does not pay there (events cost what the overlap wins,
// profiles/r03_step_notes.md).  One launch carries both: blocks [0, nw) run the weight-gradient body, blocks [nw, nw + nd) the
// dgrad body (the longer-running weight-gradient workgroups are dispatched first); same device functions as the stand-alone kernels,
// so the results are bit-identical to the two-launch path.
namespace {
template <bool LZ>
__global__ __launch_bounds__(256) void bwd16_fused_kernel(Conv3Params pd, Wgrad16Params pw, int nw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x < nw) wgrad16_body<LZ>(pw, blockIdx.x, smem);
    else conv16_body<1, LZ>(pd, (int)blockIdx.x - nw, smem);
}
template <bool LZ>
__global__ __launch_bounds__(256) void bwd32_fused_kernel(Conv3Params pd, Wgrad32Params pw, int nw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x < nw) wgrad32_body<LZ>(pw, blockIdx.x & 1, blockIdx.x >> 1, smem);
    else conv32_body<1, LZ>(pd, (int)blockIdx.x - nw, smem);
}
}  // namespace

template <bool LZ, int IPI = WG64_IPI>
__global__ __launch_bounds__(256) void bwd64_fused_kernel(Conv3Params pd, Wgrad64Params pw, int nw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x < nw) wgrad64_body<LZ, IPI>(pw, blockIdx.x & 3, blockIdx.x >> 2, smem);
    else conv64_body<1, LZ>(pd, (int)blockIdx.x - nw, smem);
}

bool clhip_bwd_fused_supported(int N, int H, int W, int C, int Creal, int K, int ksize, int stride, int pad, int dtype) {
    static const bool off = clhip_cfg("BWD_FUSED") != nullptr && atoi(clhip_cfg("BWD_FUSED")) == 0;
    if (off) return false;
    return clhip_wgrad16_supported(N, H, W, C, Creal, K, ksize, stride, pad, dtype) || clhip_wgrad32_supported(N, H, W, C, Creal, K, ksize, stride, pad, dtype) ||
           clhip_wgrad64_supported(N, H, W, C, Creal, K, ksize, stride, pad, dtype);
}

int clhip_bwd_fused_launch(const void* x, const void* dz, const void* w_dg, void* dx, int accumulate, float* dw, float* ws, int N, int H, int W, int C,
                           const void* bn_z, const void* bn_y, const float* bn_mean, const float* bn_invstd, double* bn_acc, int bn_rep, const float* x_coef,
                           const clhip_bn_grad* lz, hipStream_t st) {
    Conv3Params pd;
    LazyDz lzd;
    if (lz != nullptr) {                                     // dz = the layer's own BatchNorm backward, computed on both bodies' operand loads
        lzd.dy = static_cast<const bf16_t*>(lz->dy); lzd.z = static_cast<const bf16_t*>(lz->z); lzd.sums = lz->sums; lzd.rep = lz->replicas;
        lzd.mean = lz->mean; lzd.invstd = lz->invstd; lzd.gamma = lz->gamma; lzd.beta = lz->beta; lzd.dgamma = lz->dgamma; lzd.dbeta = lz->dbeta;
        lzd.mask = static_cast<const unsigned char*>(lz->relu_mask); lzd.dres = static_cast<bf16_t*>(lz->dres); lzd.dres_acc = lz->dres_accumulate;
        lzd.invM = 1.0 / ((double)N * H * W);
    }
    pd.lz = lzd;
    pd.bn_coef = x_coef;                                     // a lazy x IS the producer's z: its ReLU mask comes from z as well (bn_y is NULL then)
    pd.bn_z = static_cast<const bf16_t*>(bn_z); pd.bn_y = static_cast<const bf16_t*>(bn_y); pd.bn_mean = bn_mean; pd.bn_invstd = bn_invstd;
    pd.bn_acc = bn_acc; pd.bn_rep = bn_rep > 0 ? bn_rep : 1;
    pd.src = static_cast<const bf16_t*>(dz); pd.wt = static_cast<const bf16_t*>(w_dg); pd.dst = static_cast<bf16_t*>(dx);
    pd.stats = nullptr; pd.stat_acc = nullptr; pd.stat_rep = 1;
    pd.N = N; pd.H = H; pd.W = W; pd.wshift = ilog2_exact(W); pd.hshift = ilog2_exact(H); pd.Cs = C; pd.Cd = C; pd.accumulate = accumulate; pd.M = N * H * W;
    pd.np = 256 + 2 * W + 2; pd.patch_bytes = (pd.np + 1) * (C == 16 ? 32 : 96); pd.nbuf = 1; pd.debug = 0;
    if (C == 64) {
        pd.np = 64 + 2 * W + 2; pd.patch_bytes = (pd.np + 1) * 144;
        const int nd64 = (pd.M + 63) / 64, groups = wgrad64_groups(N);
        Wgrad64Params pw{static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(dz), ws, N, H, wgrad64_ipg(N), x_coef, lzd};
        size_t lds64 = (size_t)pd.patch_bytes + 2048;
        const size_t wl = WG64_IPI * ((size_t)(H + 2) * 10 * 144 + (size_t)H * 8 * 32) + 2048;
        if (wl > lds64) lds64 = wl;
        if (!wgrad64_pair_images()) {
            if (lz != nullptr) hipLaunchKernelGGL((bwd64_fused_kernel<true, 1>), dim3(4 * groups + nd64), dim3(256), lds64, st, pd, pw, 4 * groups);
            else hipLaunchKernelGGL((bwd64_fused_kernel<false, 1>), dim3(4 * groups + nd64), dim3(256), lds64, st, pd, pw, 4 * groups);
        } else if (lz != nullptr) hipLaunchKernelGGL((bwd64_fused_kernel<true>), dim3(4 * groups + nd64), dim3(256), lds64, st, pd, pw, 4 * groups);
        else hipLaunchKernelGGL((bwd64_fused_kernel<false>), dim3(4 * groups + nd64), dim3(256), lds64, st, pd, pw, 4 * groups);
        CLHIP_LAUNCH_CHECK();
        return clhip_wgrad_reduce_launch(ws, dw, 9216, groups, st);
    }
    const int nd = clhip_conv16_tiles_m(pd.M);
    size_t lds = ((size_t)pd.patch_bytes > 1024 ? (size_t)pd.patch_bytes : 1024) + 1024;
    if (C == 16) {
        const int parts = wgrad16_parts(N, H), nwg = N * parts;
        Wgrad16Params pw{static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(dz), ws, N, H, x_coef, lzd, parts};
        const int RH = H / parts;
        size_t wl = (size_t)((RH + 2) * 34 + RH * 32) * 32;
        if (wl < 4 * 2304 * sizeof(float)) wl = 4 * 2304 * sizeof(float);
        if (wl > lds) lds = wl;
        static size_t attr[2] = {0, 0};
        const int v = lz != nullptr;
        if (lds > attr[v]) {
            const void* kp = v ? reinterpret_cast<const void*>(bwd16_fused_kernel<true>) : reinterpret_cast<const void*>(bwd16_fused_kernel<false>);
            if (hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
                clhip_set_error("bwd16_fused: cannot reserve %zu bytes of LDS", lds);
                return CLHIP_EHIP;
            }
            attr[v] = lds;
        }
        if (v) hipLaunchKernelGGL(bwd16_fused_kernel<true>, dim3(nwg + nd), dim3(256), lds, st, pd, pw, nwg);
        else hipLaunchKernelGGL(bwd16_fused_kernel<false>, dim3(nwg + nd), dim3(256), lds, st, pd, pw, nwg);
        CLHIP_LAUNCH_CHECK();
        return clhip_wgrad_reduce_launch(ws, dw, (int64_t)576, nwg, st);
    }
    const int groups = wgrad32_groups(N);
    Wgrad32Params pw{static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(dz), ws, N, H, wgrad32_ipg(N), x_coef, lzd};
    const size_t wl = (size_t)(H + 2) * 18 * 64 + (size_t)H * 16 * 32;
    if (wl > lds) lds = wl;
    if (lz != nullptr) hipLaunchKernelGGL(bwd32_fused_kernel<true>, dim3(2 * groups + nd), dim3(256), lds, st, pd, pw, 2 * groups);
    else hipLaunchKernelGGL(bwd32_fused_kernel<false>, dim3(2 * groups + nd), dim3(256), lds, st, pd, pw, 2 * groups);
    CLHIP_LAUNCH_CHECK();
    return clhip_wgrad_reduce_launch(ws, dw, 2304, groups, st);
}

// ---- deferred reduces.  Every partial-block weight-gradient kernel ends with "dw += sum of my slab".  On CifarResNet-32 those are 33
// launches of ~5 us per step (the largest symbol of its profile once the atomics were gone: 8.9 % of kernel time,
// profiles/r03_bench_kernel_stats_ewc_resnet32_b50_task1.txt).  A caller that gives every layer its OWN scratch (plan.hip, networks
// without a weight-gradient stream) can collect the reduces and run them as ONE launch: clhip_wgrad_defer_begin() makes
// clhip_wgrad_reduce_launch() record its arguments instead of launching, clhip_wgrad_defer_flush() sums all recorded slabs -- same
// fixed order per element, so the results are bit-identical to the per-layer launches.
namespace {
constexpr int kDeferMax = 40;
struct ReduceEntry { const float* slab; float* dw; long long n4; int splits; unsigned first_block; };
struct ReduceTable { int n; ReduceEntry e[kDeferMax]; };
thread_local ReduceTable g_defer;
thread_local bool g_defer_on = false;

__global__ __launch_bounds__(256) void wgrad_multi_reduce_kernel(ReduceTable t) {
    __shared__ float4 part[4][64];
    // the entry whose block range holds blockIdx.x: one 64-lane load + ballot (a walk over e[1..].first_block is one dependent scalar load per
    // entry -- 33 of them on CifarResNet-32, most of this launch's 10.7 us at batch 32)
    static_assert(kDeferMax <= 64, "one lane per entry");
    const int l64 = threadIdx.x & 63;
    const bool le = l64 < t.n && t.e[l64 < kDeferMax ? l64 : 0].first_block <= blockIdx.x;
    const int u = __builtin_amdgcn_readfirstlane(__popcll(__ballot(le)) - 1);
    const ReduceEntry& d = t.e[u];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const long long i = (long long)(blockIdx.x - d.first_block) * 64 + lane;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < d.n4) {                                      // (the body of wgrad3_reduce_kernel: same grouping, same order)
        const int per = (d.splits + 3) / 4;
        const int s0 = grp * per, s1 = min(d.splits, s0 + per);
        const float4* base = reinterpret_cast<const float4*>(d.slab) + i;
        int s = s0;
        for (; s + 8 <= s1; s += 8) {
            float4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = base[(size_t)(s + q) * d.n4];
#pragma unroll
            for (int q = 0; q < 8; ++q) { a.x += v[q].x; a.y += v[q].y; a.z += v[q].z; a.w += v[q].w; }
        }
        for (; s < s1; ++s) { const float4 v = base[(size_t)s * d.n4]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    }
    part[grp][lane] = a;
    __syncthreads();
    if (grp == 0 && i < d.n4) {
        float4 o = reinterpret_cast<const float4*>(d.dw)[i];
#pragma unroll
        for (int g = 0; g < 4; ++g) { const float4 v = part[g][lane]; o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w; }
        reinterpret_cast<float4*>(d.dw)[i] = o;
    }
}
}  // namespace

void clhip_wgrad_defer_begin() { g_defer.n = 0; g_defer_on = true; }
void clhip_wgrad_defer_abort() { g_defer.n = 0; g_defer_on = false; }
void clhip_wgrad_defer_pause(bool paused) { g_defer_on = !paused; }      // a launch on another stream reduces on its own

int clhip_wgrad_defer_flush(hipStream_t st, bool end) {
    if (end) g_defer_on = false;
    if (g_defer.n == 0) return CLHIP_OK;
    unsigned blocks = 0;
    for (int k = 0; k < g_defer.n; ++k) { g_defer.e[k].first_block = blocks; blocks += (unsigned)((g_defer.e[k].n4 + 63) / 64); }
    hipLaunchKernelGGL(wgrad_multi_reduce_kernel, dim3(blocks), dim3(256), 0, st, g_defer);
    g_defer.n = 0;
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

// dw += the `splits` partial blocks of a workspace, fixed order (shared with wgrad4.hip, stem.hip, conv2.hip)
int clhip_wgrad_reduce_launch(const float* slab, float* dw, int64_t n4, int splits, hipStream_t st) {
    if (g_defer_on) {
        if (g_defer.n == kDeferMax) { if (int e = clhip_wgrad_defer_flush(st, false)) return e; }
        g_defer.e[g_defer.n++] = ReduceEntry{slab, dw, (long long)n4, splits, 0u};
        return CLHIP_OK;
    }
    hipLaunchKernelGGL(wgrad3_reduce_kernel, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, st, slab, dw, n4, splits);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

int clhip_wgrad16_launch(const void* x, const void* dz, float* dw, float* ws, int N, int H, const float* x_coef, hipStream_t st) {
    const int parts = wgrad16_parts(N, H);
    Wgrad16Params p{static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(dz), ws, N, H, x_coef, LazyDz{}, parts};
    const int RH = H / parts;
    size_t lds = (size_t)((RH + 2) * 34 + RH * 32) * 32;
    if (lds < 4 * 2304 * sizeof(float)) lds = 4 * 2304 * sizeof(float);
    static size_t attr = 0;
    if (lds > attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            clhip_set_error("wgrad16: cannot reserve %zu bytes of LDS", lds);
            return CLHIP_EHIP;
        }
        attr = lds;
    }
    hipLaunchKernelGGL(wgrad16_kernel, dim3(N * parts), dim3(256), lds, st, p);
    CLHIP_LAUNCH_CHECK();
    return clhip_wgrad_reduce_launch(ws, dw, (int64_t)576, N * parts, st);
}
