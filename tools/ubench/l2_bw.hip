// L2 -> CU load-throughput probe: every wave streams 16-byte loads over an L2-resident buffer (no LDS, no MFMA).
// hipcc --offload-arch=gfx950 -O3 tools/ubench/l2_bw.hip -o /tmp/l2_bw && /tmp/l2_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void rd(const uint4* __restrict__ buf, size_t n16, int iters, uint4* out) {
    uint4 a = make_uint4(0, 0, 0, 0);
    size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            size_t i = (base + (size_t)(it * 8 + u) * 65536) % n16;
            uint4 v = buf[i];
            a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w;
        }
    }
    if (a.x == 0x12345678) out[0] = a;
}
int main() {
    for (size_t mb : {2, 8, 32, 128}) {
        size_t bytes = mb << 20, n16 = bytes / 16;
        uint4 *buf, *out;
        hipMalloc(&buf, bytes); hipMalloc(&out, 64); hipMemset(buf, 1, bytes);
        for (int blocks : {512, 1024, 2048}) {
            int iters = 2000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(rd, dim3(blocks), dim3(256), 0, 0, buf, n16, 10, out);
            hipEventRecord(e0);
            hipLaunchKernelGGL(rd, dim3(blocks), dim3(256), 0, 0, buf, n16, iters, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double tb = (double)blocks * 256 * iters * 8 * 16 / (ms * 1e-3) / 1e12;
            printf("buffer %4zu MB  blocks %5d : %.2f TB/s\n", mb, blocks, tb);
        }
        hipFree(buf); hipFree(out);
    }
    return 0;
}
