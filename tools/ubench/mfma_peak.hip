// micro-benchmark: MFMA-only and MFMA+LDS-read loops, to calibrate what the box delivers
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[32768];
    for (int i = threadIdx.x; i < 8192; i += 256) ((float*)lds)[i] = i * 0.001f;
    __syncthreads();
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    uint4 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = make_uint4(threadIdx.x, i, 3, 4); b[i] = make_uint4(5, threadIdx.x, i, 8); }
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = *(const uint4*)(lds + ((lane * 16 + i * 1024 + it * 64) & 32767 & ~15));
                b[i] = *(const uint4*)(lds + ((lane * 16 + i * 1024 + 4096 + it * 64) & 32767 & ~15));
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[i]), __builtin_bit_cast(bf16x8_t, b[j]), acc[i * 4 + j], 0, 0, 0);
        if (MODE == 2) __syncthreads();
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 2000;
    for (int mode = 0; mode < 3; ++mode) for (int blocks : {256, 512, 1024, 2048}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
            else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double fl = (double)blocks * 4 * iters * 16 * 2.0 * 16 * 16 * 32;
            if (rep) printf("mode %d blocks %d: %.3f ms  %.0f TFLOP/s\n", mode, blocks, ms, fl / ms / 1e9);
        }
    }
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    printf("CUs %d clock %d kHz  name %s\n", pr.multiProcessorCount, pr.clockRate, pr.name);
    return 0;
}
