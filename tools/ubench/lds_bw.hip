// lds_bw -- LDS read throughput by instruction width and number of reading waves per CU (conflict-free addresses, 12 independent reads
// in flight per wave):   hipcc --offload-arch=gfx950 -O2 tools/ubench/lds_bw.hip -o tools/ubench/lds_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

// fragment-shaped address patterns of a 32-row x 16-byte-per-lane read (row = lane & 31, k half = lane >> 5):
//   1: 64-byte rows, chunk ^ ((row >> 2) & 3)   (gemm5, conv4 CK = 32)      2: 128-byte rows, chunk ^ ((row >> 1) & 7)   (conv4 CK = 64)
//   3: 80-byte pitch (conv4 patch, CK = 32)      4: 144-byte pitch (conv4 patch CK = 64)     5: 64-byte rows, no swizzle (conflicting)
//   6: 64-byte rows, 16-lane groups rearranged: chunk ^ (row & 3) ...
__device__ __forceinline__ unsigned pat_addr(int pat, int lane, int base) {
    const int row = lane & 31, kh = lane >> 5;
    switch (pat) {
        case 1: return base + row * 64 + ((kh ^ ((row >> 2) & 3)) << 4);
        case 2: return base + row * 128 + ((kh ^ ((row >> 1) & 7)) << 4);
        case 3: return base + row * 80 + kh * 16;
        case 4: return base + row * 144 + kh * 16;
        case 5: return base + row * 64 + kh * 16;
        case 6: return base + row * 64 + ((kh ^ (row & 3)) << 4);
        default: return base + lane * 16;
    }
}

template <int NREAD>
__global__ __launch_bounds__(1024) void kp(unsigned long long* out, int iters, int nwaves, int pat) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 16; i += blockDim.x) reinterpret_cast<u32x4*>(smem)[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    if (wave >= nwaves) return;
    unsigned addr[NREAD];
    for (int q = 0; q < NREAD; ++q) addr[q] = pat_addr(pat, lane, ((wave * NREAD + q) * 256) % 57344);
    unsigned sink = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        u32x4 v[NREAD];
#pragma unroll
        for (int q = 0; q < NREAD; ++q) v[q] = *reinterpret_cast<const u32x4*>(smem + addr[q]);
#pragma unroll
        for (int q = 0; q < NREAD; ++q) sink ^= v[q].x + v[q].w;
        asm volatile("" : "+v"(sink));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[blockIdx.x * 32 + wave] = t1 - t0; out[blockIdx.x * 32 + 16 + wave] = sink; }
}

void runp(unsigned long long* d, int nwaves, int pat, const char* name) {
    const int iters = 20000;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kp<12>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipMemset(d, 0, 256 * 32 * 8));
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(kp<12>, dim3(256), dim3(1024), 65536, 0, d, iters, nwaves, pat); CK(hipDeviceSynchronize()); }
    static unsigned long long h[256 * 32];
    CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    double t = 0; int n = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < nwaves; ++w) { t += (double)h[b * 32 + w]; ++n; }
    const double per_batch = t / n / iters;
    printf("ds_read_b128 pattern %-46s %2d waves/CU: %5.1f ticks per instruction per wave, %6.1f B per tick per CU\n", name, nwaves, per_batch / 12, (double)nwaves * 12 * 1024 / per_batch);
}

template <int BYTES, int NREAD>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, int iters, int nwaves) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 16; i += blockDim.x) reinterpret_cast<u32x4*>(smem)[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    if (wave >= nwaves) return;
    unsigned addr[NREAD];
    for (int q = 0; q < NREAD; ++q) addr[q] = ((wave * NREAD + q) * 1024) % 61440 + lane * BYTES;     // lane-linear: conflict-free for every width
    unsigned sink = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (BYTES == 16) {
            u32x4 v[NREAD];
#pragma unroll
            for (int q = 0; q < NREAD; ++q) v[q] = *reinterpret_cast<const u32x4*>(smem + addr[q]);
#pragma unroll
            for (int q = 0; q < NREAD; ++q) sink ^= v[q].x + v[q].w;
        } else if constexpr (BYTES == 8) {
            u32x2 v[NREAD];
#pragma unroll
            for (int q = 0; q < NREAD; ++q) v[q] = *reinterpret_cast<const u32x2*>(smem + addr[q]);
#pragma unroll
            for (int q = 0; q < NREAD; ++q) sink ^= v[q].x + v[q].y;
        } else {
            unsigned v[NREAD];
#pragma unroll
            for (int q = 0; q < NREAD; ++q) v[q] = *reinterpret_cast<const unsigned*>(smem + addr[q]);
#pragma unroll
            for (int q = 0; q < NREAD; ++q) sink ^= v[q];
        }
        asm volatile("" : "+v"(sink));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[blockIdx.x * 32 + wave] = t1 - t0; out[blockIdx.x * 32 + 16 + wave] = sink; }
}

template <int BYTES, int NREAD> void run(unsigned long long* d, int nwaves) {
    const int iters = 20000;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<BYTES, NREAD>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipMemset(d, 0, 256 * 32 * 8));
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL((k<BYTES, NREAD>), dim3(256), dim3(1024), 65536, 0, d, iters, nwaves); CK(hipDeviceSynchronize()); }
    static unsigned long long h[256 * 32];
    CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    double t = 0; int n = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < nwaves; ++w) { t += (double)h[b * 32 + w]; ++n; }
    const double per_batch = t / n / iters;
    printf("ds_read_b%-3d x %2d in flight, %2d waves/CU: %7.1f ticks per batch = %5.1f ticks per instruction per wave, %6.1f B per tick per CU\n", BYTES * 8, NREAD, nwaves, per_batch,
           per_batch / NREAD, (double)nwaves * NREAD * 64 * BYTES / per_batch);
}

int main() {
    unsigned long long* d; CK(hipMalloc(&d, 256 * 32 * 8));
    const char* names[] = {"lane-linear", "64-B rows ^ (row>>2)&3", "128-B rows ^ (row>>1)&7", "80-B pitch", "144-B pitch", "64-B rows, no swizzle", "64-B rows ^ row&3"};
    for (int pat = 0; pat <= 6; ++pat) for (int nw : {4, 8}) runp(d, nw, pat, names[pat]);
    for (int nw : {1, 2, 4, 8, 16}) run<16, 12>(d, nw);
    for (int nw : {1, 4, 8, 16}) run<8, 12>(d, nw);
    for (int nw : {1, 4, 8, 16}) run<4, 12>(d, nw);
    for (int nw : {4, 8}) run<16, 4>(d, nw);
    for (int nw : {4, 8}) run<16, 2>(d, nw);
    return 0;
}
