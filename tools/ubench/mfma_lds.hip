// mfma_lds -- does a wave's LDS fragment reading slow down when the SIMD's other wave multiplies, and does it matter whether the
// accumulators live in VGPRs or AGPRs?   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_lds.hip -o tools/ubench/mfma_lds
// One 512-thread workgroup per CU: waves 4-7 stream batches of 12 conflict-free ds_read_b128 (the gemm5 / conv4 read phase), waves
// 0-3 either idle, or issue v_mfma_f32_32x32x16_bf16 back to back with the accumulator in VGPRs, or the same with it in AGPRs.
// Prints cycles per 12-read batch (s_memtime deltas) and MFMAs per 1000 cycles.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE>      // 0 readers only, 1 MFMA partner with VGPR accumulators, 2 with AGPR accumulators, 3 = 1 + readers idle, 4 = 2 + readers idle
__global__ __launch_bounds__(512) void k(unsigned long long* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 16; i += 512) reinterpret_cast<u32x4*>(smem)[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    if (wave >= 4) {
        if (MODE >= 3) return;
        // 32 rows x 64 B, swizzled: conflict-free
        const int row = lane & 31, kh = lane >> 5;
        unsigned addr[12];
        for (int q = 0; q < 12; ++q) addr[q] = ((wave - 4) * 12 + q) * 2048 % 61440 + row * 64 + ((kh ^ ((row >> 2) & 3)) << 4);
        u32x4 sink = {0, 0, 0, 0};
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
            u32x4 v[12];
#pragma unroll
            for (int q = 0; q < 12; ++q) v[q] = *reinterpret_cast<const u32x4*>(smem + addr[q]);
#pragma unroll
            for (int q = 0; q < 12; ++q) { sink.x ^= v[q].x; sink.y += v[q].y; }
            asm volatile("" : "+v"(sink));
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { out[blockIdx.x * 16 + wave] = t1 - t0; out[blockIdx.x * 16 + 8 + wave] = sink.x + sink.y; }
    } else {
        if (MODE == 0) return;
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * lane); b[i] = (__bf16)(0.002f * (lane + i)); }
        f32x16 acc[8];
        for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (MODE == 1 || MODE == 3) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a), "v"(b));
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        float s = 0.f;
        for (int j = 0; j < 8; ++j) s += acc[j][0];
        if (lane == 0) { out[blockIdx.x * 16 + wave] = t1 - t0; out[blockIdx.x * 16 + 8 + wave] = (unsigned long long)s; }
    }
}

template <int MODE> void run(const char* name, unsigned long long* d, int iters) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipMemset(d, 0, 256 * 16 * 8));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 65536, 0, d, iters);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 65536, 0, d, iters);
    CK(hipDeviceSynchronize());
    unsigned long long h[256 * 16];
    CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    double rd = 0, mf = 0; int nr = 0, nm = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) { if (!h[b * 16 + w]) continue; if (w >= 4) { rd += (double)h[b * 16 + w]; ++nr; } else { mf += (double)h[b * 16 + w]; ++nm; } }
    // s_memtime ticks at 100 MHz: convert with the ~2.1 GHz shader clock is left to the reader; ratios are what matters
    printf("%-44s reader ticks per 12-read batch %8.3f   mfma-wave ticks per 8 MFMAs %8.3f\n", name, nr ? rd / nr / iters : 0.0, nm ? mf / nm / iters : 0.0);
}

int main() {
    unsigned long long* d; CK(hipMalloc(&d, 256 * 16 * 8));
    const int iters = 20000;
    run<0>("readers alone", d, iters);
    run<3>("MFMA waves alone, VGPR accumulators", d, iters);
    run<4>("MFMA waves alone, AGPR accumulators", d, iters);
    run<1>("readers + MFMA partner (VGPR accumulators)", d, iters);
    run<2>("readers + MFMA partner (AGPR accumulators)", d, iters);
    return 0;
}
