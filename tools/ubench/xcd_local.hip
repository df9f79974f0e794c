// xcd_local -- (1) which XCD a workgroup of a 256-workgroup grid lands on (HW_REG_XCC_ID against blockIdx % 8), (2) the round trip of a tagged 8-byte granule between two
// workgroups of the SAME XCD through their shared L2 (plain store + sc0 load) against the agent-scope form of xch.h (sc1 store + sc1 load), (3) a one-hop sweep over the
// 32 workgroups of an XCD in both forms.     hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/xcd_local.hip -o /tmp/xcd_local && /tmp/xcd_local
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }

template <int AUX>
__device__ __forceinline__ unsigned long long ld(__amdgpu_buffer_rsrc_t r, int off) {
    asm volatile("" ::: "memory");                               // (a poll: never hoisted out of its loop)
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, AUX);
    asm volatile("" ::: "memory");
    return ((unsigned long long)v.y << 32) | v.x;
}
template <int AUX>
__device__ __forceinline__ void st(__amdgpu_buffer_rsrc_t r, int off, unsigned long long x) {
    __builtin_amdgcn_raw_buffer_store_b64((u32x2){(unsigned)x, (unsigned)(x >> 32)}, r, off, 0, AUX);
}

__global__ void where_kernel(unsigned* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

// ping-pong between workgroup w (even slot) and its partner w + 8 (same XCD if the mapping is blockIdx % 8): LAUX / SAUX = cache bits of the polls / stores
template <int LAUX, int SAUX>
__global__ __launch_bounds__(64) void pingpong_kernel(unsigned long long* buf, int rounds, unsigned long long* cycles, unsigned* bad) {
    const int w = blockIdx.x, slot = (w >> 3) & 1, pair = (w >> 4) * 8 + (w & 7);
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 1 << 20, 0x00020000);
    const int mine = (pair * 2 + slot) * 64, other = (pair * 2 + (slot ^ 1)) * 64;      // 64-byte apart: one cache line each... (512 B pitch per pair)
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned fails = 0;
    for (int i = 1; i <= rounds; ++i) {
        if (slot == 0) st<SAUX>(r, mine, (unsigned long long)i);
        unsigned spins = 0;
        while (ld<LAUX>(r, other) != (unsigned long long)i) { if (++spins > (1u << 14)) { ++fails; break; } }
        if (slot == 1) st<SAUX>(r, mine, (unsigned long long)i);
        if (fails) break;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    cycles[w] = t1 - t0;
    if (fails) atomicAdd(bad, 1u);
}

// one-hop sweep inside an XCD: the 32 workgroups with blockIdx % 8 == x publish NV tagged values each, every one of them polls all 32 x NV granules
template <int LAUX, int SAUX>
__global__ __launch_bounds__(256) void sweep_kernel(unsigned long long* buf, int NV, int phases, unsigned long long* cycles, unsigned* bad) {
    const int w = blockIdx.x, x = w & 7, m = w >> 3, t = threadIdx.x;         // member m of XCD x
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 1 << 22, 0x00020000);
    __shared__ double part[256];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned fails = 0;
    for (int p = 1; p <= phases; ++p) {
        const int par = p & 1;
        if (t < NV) st<SAUX>(r, (((par * 8 + x) * 32 + m) * 128 + t) * 8, ((unsigned long long)p << 32) | (unsigned)(m + t));
        // thread t: value t % NV, contributors t / NV, + 256 / NV, ...
        const int v = t % NV, g0 = t / NV, gs = 256 / NV;
        double s = 0.0;
        for (int c = g0; c < 32; c += gs) {
            unsigned spins = 0;
            unsigned long long q;
            while (((q = ld<LAUX>(r, (((par * 8 + x) * 32 + c) * 128 + v) * 8)) >> 32) != (unsigned long long)p) { if (++spins > (1u << 14)) { ++fails; break; } }
            s += (double)(unsigned)q;
        }
        part[t] = s;
        __syncthreads();
        if (t < NV) {
            double a = 0.0;
            for (int g = 0; g < gs; ++g) a += part[g * NV + t];
            if (!fails && a != 32.0 * t + 496.0) ++fails;
        }
        __syncthreads();
    }
    if (t == 0) cycles[w] = __builtin_amdgcn_s_memtime() - t0;
    if (fails) atomicAdd(bad, 1u);
}

int main() {
    unsigned *where, *bad;
    unsigned long long *buf, *cyc;
    CK(hipMalloc(&where, 256 * 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&buf, 1 << 22)); CK(hipMalloc(&cyc, 256 * 8));
    hipLaunchKernelGGL(where_kernel, dim3(256), dim3(64), 0, 0, where);
    unsigned h[256];
    CK(hipMemcpy(h, where, sizeof(h), hipMemcpyDeviceToHost));
    int match = 0;
    for (int i = 0; i < 256; ++i) match += h[i] == (unsigned)(i & 7);
    printf("workgroup -> XCD: %d of 256 on XCD blockIdx %% 8; first 16:", match);
    for (int i = 0; i < 16; ++i) printf(" %u", h[i]);
    printf("\n");
    for (int G : {32, 72, 100, 200}) {
        hipLaunchKernelGGL(where_kernel, dim3(G), dim3(64), 0, 0, where);
        CK(hipMemcpy(h, where, G * 4, hipMemcpyDeviceToHost));
        int mt = 0;
        for (int i = 0; i < G; ++i) mt += h[i] == (unsigned)(i & 7);
        printf("  grid %d: %d match\n", G, mt); fflush(stdout);
    }
    unsigned long long hc[256];
    auto report = [&](const char* name, int G, int rounds) {
        unsigned hb = 0;
        CK(hipMemcpy(hc, cyc, G * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        double mx = 0; for (int i = 0; i < G; ++i) mx = hc[i] > mx ? (double)hc[i] : mx;
        printf("%-64s %8.0f cycles per round (max over workgroups)%s\n", name, mx / rounds, hb ? "   ** FAILED / timed out **" : ""); fflush(stdout);
    };
#define PP(L, S, name) do { CK(hipMemset(buf, 0, 1 << 22)); CK(hipMemset(bad, 0, 4)); hipLaunchKernelGGL((pingpong_kernel<L, S>), dim3(256), dim3(64), 0, 0, buf, 300, cyc, bad); CK(hipDeviceSynchronize()); report(name, 256, 300); } while (0)
    PP(16, 16, "ping-pong, sc1 loads + sc1 stores (xch.h's form):");
    PP(16, 0, "ping-pong, sc1 loads + plain stores:");
    PP(17, 0, "ping-pong, sc0 sc1 loads + plain stores:");
#define SW(L, S, NV, name) do { CK(hipMemset(buf, 0, 1 << 22)); CK(hipMemset(bad, 0, 4)); hipLaunchKernelGGL((sweep_kernel<L, S>), dim3(256), dim3(256), 0, 0, buf, NV, 200, cyc, bad); CK(hipDeviceSynchronize()); report(name, 256, 200); } while (0)
    SW(16, 16, 32, "sweep of an XCD's 32 workgroups x 32 values, sc1 / sc1:");
    SW(17, 0, 32, "sweep of an XCD's 32 workgroups x 32 values, sc0 sc1 / plain:");
    SW(16, 0, 32, "sweep of an XCD's 32 workgroups x 32 values, sc1 / plain:");
    SW(16, 16, 128, "sweep of an XCD's 32 workgroups x 128 values, sc1 / sc1:");
    SW(17, 0, 128, "sweep of an XCD's 32 workgroups x 128 values, sc0 sc1 / plain:");
    SW(17, 0, 64, "sweep of an XCD's 32 workgroups x 64 values, sc0 sc1 / plain:");
    SW(16, 16, 64, "sweep of an XCD's 32 workgroups x 64 values, sc1 / sc1:");
    return 0;
}
