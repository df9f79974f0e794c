// gemm.hip -- C[M,N] = epilogue(A[M,K] . B[N,K]^T) for the ViT path (gfx950 MFMA; bf16 = performance mode,
// fp32 = parity mode through the exact fp32 MFMA, same kernel text).
//
// Every linear layer of the frozen ViT (reference core/model/backbone/transformer.py:172 qkv, :194 proj,
// :1267-1271 fc1/fc2, timm PatchEmbed) and every activation-gradient product of its backward (dX = dY . W) is an
// "NT" product once the frozen weight is kept in both orientations ([out,in] for the forward, [in,out] for dX),
// so ONE kernel serves forward and backward; the epilogue carries the elementwise work the reference runs as
// separate memory passes (bias add, residual add, exact GELU, GELU derivative).
//
// Tiling: 128x128 output tile per 256-thread block (2x2 waves, each 64x64 = 4x4 MFMA 16x16 tiles), K step 64,
// register-staged double-buffered LDS (one barrier per K step), XOR-swizzled 128-B rows so ds_read_b128 is
// bank-conflict free.  The MFMA "A" operand is the B (weight) tile, so a lane ends up with 4 consecutive n of
// one output row m -> 8-byte (bf16) / 16-byte (fp32) row-contiguous stores and bias / residual loads.
// Block ids are remapped so each XCD (own L2) walks a contiguous range of row panels over all column panels.
#include "common.h"

namespace {

enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_RES = 2, EPI_BIAS_GELU = 3, EPI_GELU_BWD = 4 };

struct GemmParams {
    const void* A; const void* B; void* C;
    const float* bias; const void* R; void* H;
    int M, N, K, lda, ldb, ldc, ldr, ldh;
};

template <typename T> struct Chunk;
template <> struct Chunk<bf16_t> { uint4 a; };
template <> struct Chunk<float> { uint4 a, b; };

template <typename T> __device__ __forceinline__ Chunk<T> czero() {
    Chunk<T> c;
    c.a = make_uint4(0, 0, 0, 0);
    if constexpr (sizeof(T) == 4) c.b = make_uint4(0, 0, 0, 0);
    return c;
}
template <typename T> __device__ __forceinline__ Chunk<T> cload(const T* p) {
    Chunk<T> c;
    c.a = *reinterpret_cast<const uint4*>(p);
    if constexpr (sizeof(T) == 4) c.b = *reinterpret_cast<const uint4*>(p + 4);
    return c;
}
// LDS rows hold 8 chunks (64 elements).  bf16: 128-B rows, chunk index XORed with row&7.  fp32: 256-B rows, linear.
template <typename T> __device__ __forceinline__ int lds_off(int row, int chunk) {
    if constexpr (sizeof(T) == 2) return row * 128 + ((chunk ^ (row & 7)) << 4);
    else return row * 256 + (chunk << 5);
}
template <typename T> __device__ __forceinline__ void lds_st(char* base, int off, const Chunk<T>& c) {
    *reinterpret_cast<uint4*>(base + off) = c.a;
    if constexpr (sizeof(T) == 4) *reinterpret_cast<uint4*>(base + off + 16) = c.b;
}
template <typename T> __device__ __forceinline__ Chunk<T> lds_ld(const char* base, int off) {
    Chunk<T> c;
    c.a = *reinterpret_cast<const uint4*>(base + off);
    if constexpr (sizeof(T) == 4) c.b = *reinterpret_cast<const uint4*>(base + off + 16);
    return c;
}
template <typename T> __device__ __forceinline__ f32x4 mma(const Chunk<T>& a, const Chunk<T>& b, f32x4 acc) {
    if constexpr (sizeof(T) == 2) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a.a), __builtin_bit_cast(bf16x8_t, b.a), acc, 0, 0, 0);
    } else {
        const float* x = reinterpret_cast<const float*>(&a);
        const float* y = reinterpret_cast<const float*>(&b);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j], y[j], acc, 0, 0, 0);
        return acc;
    }
}

template <typename T> __device__ __forceinline__ void load4(const T* p, float (&v)[4]);
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float (&v)[4]) {
    uint2 u = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
}
template <> __device__ __forceinline__ void load4<float>(const float* p, float (&v)[4]) {
    float4 a = *reinterpret_cast<const float4*>(p);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float (&v)[4]);
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float (&v)[4]) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
}
template <> __device__ __forceinline__ void store4<float>(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}

// exact GELU (nn.GELU default, transformer.py:1259): x * Phi(x), and its derivative Phi(x) + x * phi(x)
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

constexpr int BM = 128, BN = 128, BK = 64;

template <typename T, int EPI>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 2 : 1) void gemm_nt_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE_BYTES = BM * BK * (int)sizeof(T);          // one operand tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware bijective remap of the block id (8 XCDs, round-robin dispatch): XCD x gets a contiguous range
    const int tiles_n = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int wg;
    {
        const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const T* A = static_cast<const T*>(p.A);
    const T* B = static_cast<const T*>(p.B);

    // global -> register staging: thread owns chunk c of rows r0 + 32 i
    const int c = tid & 7, r0 = tid >> 3;
    const T* ag[4];
    const T* bg[4];
    bool av[4], bv[4];
    int st_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = r0 + 32 * i;
        av[i] = (m0 + row) < p.M;
        bv[i] = (n0 + row) < p.N;
        ag[i] = A + (size_t)(av[i] ? m0 + row : 0) * p.lda + c * 8;
        bg[i] = B + (size_t)(bv[i] ? n0 + row : 0) * p.ldb + c * 8;
        st_off[i] = lds_off<T>(row, c);
    }
    // per-lane fragment read offsets (row & 7 == l15 & 7 because every tile row base is a multiple of 16)
    int a_off[2], b_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        a_off[kk] = lds_off<T>(wm * 64 + l15, g + 4 * kk);
        b_off[kk] = lds_off<T>(wn * 64 + l15, g + 4 * kk);
    }
    constexpr int ROW16 = 16 * BK * (int)sizeof(T);               // LDS bytes of 16 tile rows

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    Chunk<T> ra[4], rb[4];
    const int KT = p.K / BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ra[i] = av[i] ? cload<T>(ag[i]) : czero<T>();
        rb[i] = bv[i] ? cload<T>(bg[i]) : czero<T>();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        lds_st<T>(smem, st_off[i], ra[i]);
        lds_st<T>(smem + TILE_BYTES, st_off[i], rb[i]);
    }
    __syncthreads();

    for (int kt = 0; kt < KT; ++kt) {
        const char* As = smem + (kt & 1) * 2 * TILE_BYTES;
        const char* Bs = As + TILE_BYTES;
        const bool more = kt + 1 < KT;
        if (more) {
            const int ko = (kt + 1) * BK;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = av[i] ? cload<T>(ag[i] + ko) : czero<T>();
                rb[i] = bv[i] ? cload<T>(bg[i] + ko) : czero<T>();
            }
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            Chunk<T> fa[4], fb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = lds_ld<T>(Bs, b_off[kk] + j * ROW16);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = lds_ld<T>(As, a_off[kk] + i * ROW16);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mma<T>(fb[j], fa[i], acc[i][j]);       // D[row = n][col = m]
        }
        if (more) {
            char* An = smem + ((kt + 1) & 1) * 2 * TILE_BYTES;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                lds_st<T>(An, st_off[i], ra[i]);
                lds_st<T>(An + TILE_BYTES, st_off[i], rb[i]);
            }
        }
        __syncthreads();
    }

    // epilogue: lane holds C[m][n .. n+3], m = m0 + wm*64 + i*16 + l15, n = n0 + wn*64 + j*16 + g*4
    T* C = static_cast<T*>(p.C);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + l15;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + g * 4;
            if (n >= p.N) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_RES || EPI == EPI_BIAS_GELU) {
                const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
            }
            if constexpr (EPI == EPI_BIAS_RES) {
                float r[4];
                load4<T>(static_cast<const T*>(p.R) + (size_t)m * p.ldr + n, r);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += r[e];
            }
            if constexpr (EPI == EPI_BIAS_GELU) {
                if (p.H) store4<T>(static_cast<T*>(p.H) + (size_t)m * p.ldh + n, v);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
            }
            if constexpr (EPI == EPI_GELU_BWD) {
                float h[4];
                load4<T>(static_cast<const T*>(p.H) + (size_t)m * p.ldh + n, h);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= gelu_grad_f(h[e]);
            }
            store4<T>(C + (size_t)m * p.ldc + n, v);
        }
    }
}

template <typename T, int EPI> int launch(const GemmParams& p, hipStream_t s) {
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    const size_t smem = 4 * (size_t)BM * BK * sizeof(T);
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<T, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_done = true;
    }
    hipLaunchKernelGGL((gemm_nt_kernel<T, EPI>), dim3(tiles), dim3(256), smem, s, p);
    CLHIP_LAUNCH_CHECK();
    return CLHIP_OK;
}

template <typename T> int dispatch(int epi, const GemmParams& p, hipStream_t s) {
    switch (epi) {
        case EPI_NONE: return launch<T, EPI_NONE>(p, s);
        case EPI_BIAS: return launch<T, EPI_BIAS>(p, s);
        case EPI_BIAS_RES: return launch<T, EPI_BIAS_RES>(p, s);
        case EPI_BIAS_GELU: return launch<T, EPI_BIAS_GELU>(p, s);
        case EPI_GELU_BWD: return launch<T, EPI_GELU_BWD>(p, s);
    }
    clhip_set_error("clhip_gemm_nt: unknown epilogue %d", epi);
    return CLHIP_EINVAL;
}

}  // namespace

extern "C" int clhip_gemm_nt(const void* A, const void* B, void* C, const float* bias, const void* R, void* H, int M, int N, int K,
                             int lda, int ldb, int ldc, int ldr, int ldh, int epilogue, int dtype, void* stream) {
    CLHIP_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0);
    CLHIP_CHECK_ARG(K % 64 == 0 && N % 4 == 0);
    CLHIP_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0);
    CLHIP_CHECK_ARG(dtype == CLHIP_BF16 || dtype == CLHIP_F32);
    if (epilogue == EPI_BIAS || epilogue == EPI_BIAS_RES || epilogue == EPI_BIAS_GELU) CLHIP_CHECK_ARG(bias != nullptr);
    if (epilogue == EPI_BIAS_RES) CLHIP_CHECK_ARG(R != nullptr && ldr % 4 == 0);
    if (epilogue == EPI_GELU_BWD) CLHIP_CHECK_ARG(H != nullptr);
    if (H) CLHIP_CHECK_ARG(ldh % 4 == 0);
    GemmParams p{A, B, C, bias, R, H, M, N, K, lda, ldb, ldc, ldr, ldh};
    hipStream_t s = static_cast<hipStream_t>(stream);
    return dtype == CLHIP_BF16 ? dispatch<bf16_t>(epilogue, p, s) : dispatch<float>(epilogue, p, s);
}
