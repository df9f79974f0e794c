// xch_bench -- cost and correctness of the in-launch all-reduce of the stage-level training kernels (libcontinual_amd/csrc/xch.h) on a co-resident grid:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/xch_bench.hip -o tools/ubench/xch_bench && tools/ubench/xch_bench
// Every phase each workgroup publishes NV values that are exact in fp32 with an exact fp64 sum, so every total of every phase is checked bit for bit in
// every workgroup; `skew` adds a workgroup-dependent spin between phases (uneven arrival, L1-warm consumers -- the conditions that expose a missing acquire).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../libcontinual_amd/csrc/xch.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ float val_of(int wg, int v, int phase) { return (float)(wg + v + (phase & 1023)); }      // exact in fp32, the total in closed form

// MODE 0: two hops; MODE 1: one hop (every workgroup sweeps all G x NV hop-1 granules); `work` = cycles of independent work between the two halves
template <int MODE>
__global__ __launch_bounds__(256) void xch_kernel(XchBuf b, int G, int NV, int phases, int skew, int work, unsigned* err) {
    __shared__ float vals[128];
    __shared__ double tot[128];
    __shared__ double scratch[kXchScratchDoubles];
    const int t = threadIdx.x, wg = blockIdx.x;
    const unsigned base = xch_base(b);
    unsigned bad = 0;
    for (int p = 0; p < phases; ++p) {
        if (skew > 0) {
            const int spin = ((wg * 37 + p * 11) % 5) * skew;
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            while ((long long)(__builtin_amdgcn_s_memtime() - t0) < spin) {}
        }
        if (t < NV) vals[t] = val_of(wg, t, p);
        __syncthreads();
        const unsigned tag = base + p + 1;
        if (MODE == 2) xch_hier_begin(b, wg, G, NV, tag, vals, scratch + 256, scratch);
        else xch_publish(b, wg, NV, tag, vals);
        if (MODE == 0) xch_reduce(b, wg, G, NV, tag, scratch);
        if (work > 0) {
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            while ((long long)(__builtin_amdgcn_s_memtime() - t0) < work) {}
        }
        if (MODE == 0) xch_collect(b, NV, tag, tot, scratch);
        else if (MODE == 2) xch_hier_end(b, G, NV, tag, tot, scratch);
        else xch_sweep(b, G, NV, tag, tot, scratch);
        if (t < NV) {
            const double e = (double)G * (G - 1) * 0.5 + (double)G * (t + (p & 1023));
            if (tot[t] != e) ++bad;
        }
        __syncthreads();
    }
    if (bad) atomicAdd(err, bad);
    if (wg == 0 && t == 0) xch_advance(b, base, phases);
}

template <int MODE>
static void run(int G, int NV, int skew, int work, void* buf, unsigned* derr) {
    const int phases = 200;
    XchBuf b = xch_carve(buf, 256, 128);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(xch_kernel<MODE>, dim3(G), dim3(256), 0, 0, b, G, NV, phases, skew, work, derr);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    unsigned herr[2] = {0, 0}, hctl[4];
    CK(hipMemcpy(herr, derr, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hctl, b.ctl, 16, hipMemcpyDeviceToHost));
    printf("%s G=%3d NV=%3d skew=%4d work=%5d ticks : %7.2f us per phase   mismatches %u  timeout-word %u\n", MODE == 0 ? "two-hop" : (MODE == 1 ? "one-hop" : "3-level"), G, NV, skew, work,
           best * 1000.f / phases, herr[0], hctl[1]);
    CK(hipMemset(derr, 0, 4));
}

int main() {
    void* buf; unsigned* derr;
    const size_t bytes = xch_bytes(256, 128);
    CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
    CK(hipMalloc(&derr, 64)); CK(hipMemset(derr, 0, 64));
    const int Gs[] = {8, 32, 64, 72, 100, 128, 200, 256};
    const int NVs[] = {32, 64, 128};
    // s_memtime ticks at 100 MHz: work = 250 ticks = 2.5 us of independent work between publish and collect (the weight gradient of the backward)
    for (int work : {0, 250})
        for (int skew : {0, 40})
            for (int G : Gs)
                for (int NV : NVs) {
                    if (work && skew) continue;
                    run<0>(G, NV, skew, work, buf, derr);
                    if (G * NV <= kXchOneHop) run<1>(G, NV, skew, work, buf, derr);
                    if (G >= 32) run<2>(G, NV, skew, work, buf, derr);
                }
    return 0;
}
