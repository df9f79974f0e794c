// barrier_cost -- cycles per s_barrier for a 512-thread workgroup, alone and with the staggered-halves pattern (waves 4-7 one barrier
// behind), with and without a little work between barriers:   hipcc --offload-arch=gfx950 -O2 tools/ubench/barrier_cost.hip -o tools/ubench/barrier_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE, int THREADS>     // 0: bare barriers; 1: staggered halves; 2: bare + 64 VALU ops between; 3: wait lgkmcnt(0) + barrier (nothing outstanding)
__global__ __launch_bounds__(THREADS) void k(unsigned long long* out, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float v = lane;
    if (MODE == 1 && wave >= 4) __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2) {
#pragma unroll
            for (int q = 0; q < 64; ++q) v = v * 1.0001f + 0.5f;
        }
        if (MODE == 3) __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (MODE == 1 && wave < 4) __builtin_amdgcn_s_barrier();
    if (lane == 0) { out[blockIdx.x * 16 + wave] = t1 - t0; if (v == 12345.f) out[0] = 1; }
}

template <int MODE, int THREADS> void run(const char* name, unsigned long long* d, int grid) {
    const int iters = 100000;
    CK(hipMemset(d, 0, 1024 * 16 * 8));
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL((k<MODE, THREADS>), dim3(grid), dim3(THREADS), 0, 0, d, iters); CK(hipDeviceSynchronize()); }
    static unsigned long long h[1024 * 16];
    CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    double t = 0; int n = 0;
    for (int b = 0; b < grid; ++b) for (int w = 0; w < THREADS / 64; ++w) { t += (double)h[b * 16 + w]; ++n; }
    printf("%-64s %7.1f ticks per barrier\n", name, t / n / iters);
}

int main() {
    unsigned long long* d; CK(hipMalloc(&d, 1024 * 16 * 8));
    run<0, 512>("8 waves, bare s_barrier, 1 workgroup per CU", d, 256);
    run<0, 256>("4 waves, bare s_barrier, 1 workgroup per CU", d, 256);
    run<0, 256>("4 waves, bare s_barrier, 2 workgroups per CU", d, 512);
    run<0, 64>("1 wave, bare s_barrier", d, 256);
    run<1, 512>("8 waves, halves one barrier apart", d, 256);
    run<2, 512>("8 waves, 64 dependent VALU ops + barrier", d, 256);
    run<3, 512>("8 waves, s_waitcnt lgkmcnt(0) + barrier", d, 256);
    return 0;
}
