// launch_floor -- what a kernel costs before it does anything: back-to-back launch cadence (hipEvents over 200 launches) of
//   null kernels (grid / block / dynamic LDS / register footprint varied) and of kernels that execute N straight-line or looped
//   VALU instructions once per wave (instruction-fetch cost of unrolled code on a cold instruction cache).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/launch_floor.hip -o tools/ubench/launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void null_k(float* out) { if (out == nullptr) return; }
__global__ void null_lds_k(float* out) { extern __shared__ char smem[]; if (out != nullptr && threadIdx.x == 12345) out[0] = smem[threadIdx.x]; }
__global__ __launch_bounds__(256, 2) void null_regs_k(float* out, int n) {     // forces a 256-register allocation
    float v[200];
#pragma unroll
    for (int i = 0; i < 200; ++i) v[i] = (float)(threadIdx.x + i);
    if (n > 0) {
#pragma unroll
        for (int i = 0; i < 200; ++i) v[i] = v[i] * v[(i + 1) % 200] + (float)n;
        float s = 0;
#pragma unroll
        for (int i = 0; i < 200; ++i) s += v[i];
        out[threadIdx.x] = s;
    }
}
template <int N>
__global__ void straight_k(float* out, float a, int go) {      // N dependent-free FMAs, straight line
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = a + i;
#pragma unroll
    for (int i = 0; i < N; ++i) x[i & 7] = fmaf(x[i & 7], a, (float)(i & 15));
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    if (go) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int N, int U>
__global__ void looped_k(float* out, float a, int go) {        // the same N FMAs as N/U iterations of U
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = a + i;
#pragma unroll 1
    for (int it = 0; it < N / U; ++it)
#pragma unroll
        for (int i = 0; i < U; ++i) x[i & 7] = fmaf(x[i & 7], a, (float)(i & 15));
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    if (go) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F> static double cadence(F launch, hipStream_t st, int n = 200) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 10; ++i) launch();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < n; ++i) launch();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / n;
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    float* out; CK(hipMalloc(&out, 64 << 20));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(null_lds_k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int grid : {1, 256, 512, 2048}) {
        printf("null            grid %5d x 256           : %6.2f us\n", grid, cadence([&] { hipLaunchKernelGGL(null_k, dim3(grid), dim3(256), 0, st, nullptr); }, st));
        printf("null            grid %5d x 512           : %6.2f us\n", grid, cadence([&] { hipLaunchKernelGGL(null_k, dim3(grid), dim3(512), 0, st, nullptr); }, st));
        printf("null + 72 KB    grid %5d x 256           : %6.2f us\n", grid, cadence([&] { hipLaunchKernelGGL(null_lds_k, dim3(grid), dim3(256), 72 * 1024, st, nullptr); }, st));
        printf("null + 150 KB   grid %5d x 512           : %6.2f us\n", grid, cadence([&] { hipLaunchKernelGGL(null_lds_k, dim3(grid), dim3(512), 150 * 1024, st, nullptr); }, st));
        printf("null 256 regs   grid %5d x 256           : %6.2f us\n", grid, cadence([&] { hipLaunchKernelGGL(null_regs_k, dim3(grid), dim3(256), 0, st, out, 0); }, st));
    }
    for (int grid : {8, 256, 1024}) {
        printf("straight  500   grid %5d x 256           : %6.2f us\n", grid, cadence([&] { hipLaunchKernelGGL(straight_k<500>, dim3(grid), dim3(256), 0, st, out, 1.0001f, 0); }, st));
        printf("straight 2000   grid %5d x 256           : %6.2f us\n", grid, cadence([&] { hipLaunchKernelGGL(straight_k<2000>, dim3(grid), dim3(256), 0, st, out, 1.0001f, 0); }, st));
        printf("straight 8000   grid %5d x 256           : %6.2f us\n", grid, cadence([&] { hipLaunchKernelGGL(straight_k<8000>, dim3(grid), dim3(256), 0, st, out, 1.0001f, 0); }, st));
        printf("looped   2000/50  grid %5d x 256         : %6.2f us\n", grid, cadence([&] { hipLaunchKernelGGL((looped_k<2000, 50>), dim3(grid), dim3(256), 0, st, out, 1.0001f, 0); }, st));
        printf("looped   8000/50  grid %5d x 256         : %6.2f us\n", grid, cadence([&] { hipLaunchKernelGGL((looped_k<8000, 50>), dim3(grid), dim3(256), 0, st, out, 1.0001f, 0); }, st));
        printf("looped   8000/500 grid %5d x 256         : %6.2f us\n", grid, cadence([&] { hipLaunchKernelGGL((looped_k<8000, 500>), dim3(grid), dim3(256), 0, st, out, 1.0001f, 0); }, st));
    }
    // a 33 MB streaming copy for scale
    return 0;
}
