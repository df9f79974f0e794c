// wgrad_bench -- standalone timing + spot-check harness for the weight-gradient kernels of the 3x3 / stride-1 layers behind the C ABI
// (clhip_conv_wgrad with the deterministic workspace path).
//   build: hipcc --offload-arch=gfx950 -O2 tools/ubench/wgrad_bench.cpp -Iinclude -Llibcontinual_amd -lclhip -Wl,-rpath,'$ORIGIN/../../libcontinual_amd' -o tools/ubench/wgrad_bench
//   run  : tools/ubench/wgrad_bench [case-filter] [reps]
// Every case: random bf16 operands, dw pre-filled (the entry point accumulates), 1024 random entries of dW checked against a CPU fp64
// sum over all pixels, a second launch checked for bitwise reproducibility, then the timing over rotating buffer sets.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "clhip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

static uint16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float b2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static uint32_t rng_state = 777;
static uint32_t irand() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 4; }
static float urand() { return (irand() & 0xffff) / 32768.0f - 1.0f; }

// tuning / ablation hooks go through the library's one configuration entry point (include/clhip.h: clhip_config)
static void cfg_int(const char* key, long long v) { char b[32]; snprintf(b, sizeof(b), "%lld", v); clhip_config(key, b); }
static void cfg_ptr(const char* key, const void* ptr) { char b[32]; snprintf(b, sizeof(b), "%llu", (unsigned long long)(uintptr_t)ptr); clhip_config(key, b); }
static void clhip_wgrad4_set_trace(unsigned long long* d) { cfg_ptr("WGRAD4_TRACE", d); }

struct Case { const char* name; int N, H, W, C, K, ks = 3, st = 1; };      // H, W: INPUT size; ks x ks filter, stride st, pad ks / 2

int main(int argc, char** argv) {
    const char* filt = argc > 1 ? argv[1] : "";
    const int reps = argc > 2 ? atoi(argv[2]) : 40;
    const bool trace = argc > 3 && !strcmp(argv[3], "trace");      // phase stamps of workgroup 0 (link against libclhip_abl.so)
    std::vector<Case> cases = {
        {"L1w 256x32x32 64->64", 256, 32, 32, 64, 64},     {"L2w 256x16x16 128->128", 256, 16, 16, 128, 128},
        {"L3w 256x8x8 256->256", 256, 8, 8, 256, 256},     {"L4w 256x4x4 512->512", 256, 4, 4, 512, 512},
        {"S3w 256x8x8 64->64", 256, 8, 8, 64, 64},         {"S3w32 32x8x8 64->64", 32, 8, 8, 64, 64},
        {"L1w32 32x32x32 64->64", 32, 32, 32, 64, 64},     {"X2w 256x16x16 64->128", 256, 16, 16, 64, 128},
        {"odd 3x12x20 64->128", 3, 12, 20, 64, 128},       {"odd2 5x16x16 128->64", 5, 16, 16, 128, 64},
        {"T2w 256x32x32 64->128 s2", 256, 32, 32, 64, 128, 3, 2},   {"T3w 256x16x16 128->256 s2", 256, 16, 16, 128, 256, 3, 2},
        {"T4w 256x8x8 256->512 s2", 256, 8, 8, 256, 512, 3, 2},     {"T2w5 5x32x32 64->64 s2", 5, 32, 32, 64, 64, 3, 2},
        {"T4w7 7x8x8 64->128 s2", 7, 8, 8, 64, 128, 3, 2},
        {"P2w 256x32x32 64->128 1x1s2", 256, 32, 32, 64, 128, 1, 2}, {"P3w 256x16x16 128->256 1x1s2", 256, 16, 16, 128, 256, 1, 2},
        {"P4w 256x8x8 256->512 1x1s2", 256, 8, 8, 256, 512, 1, 2},   {"P4w7 7x8x8 64->64 1x1s2", 7, 8, 8, 64, 64, 1, 2},
    };
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Case& cs : cases) {
        if (!strstr(cs.name, filt)) continue;
        const int Ho = cs.H / cs.st, Wo = cs.W / cs.st, pad = cs.ks / 2, T = cs.ks * cs.ks;
        const size_t M = (size_t)cs.N * Ho * Wo, nx = (size_t)cs.N * cs.H * cs.W * cs.C, nz = M * cs.K, nw = (size_t)cs.K * T * cs.C;
        int nset = (int)((300u << 20) / ((nx + nz) * 2)) + 1; if (nset > 8) nset = 8; if (nset < 2) nset = 2;
        std::vector<uint16_t> hx(nx), hz(nz);
        for (auto& v : hx) v = f2b(urand());
        for (auto& v : hz) v = f2b(urand() * 0.05f);
        std::vector<float> h0(nw), h1(nw), h2(nw);
        for (auto& v : h0) v = urand();
        std::vector<void*> dx(nset), dz(nset);
        for (int i = 0; i < nset; ++i) {
            CK(hipMalloc(&dx[i], nx * 2)); CK(hipMalloc(&dz[i], nz * 2));
            CK(hipMemcpy(dx[i], hx.data(), nx * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dz[i], hz.data(), nz * 2, hipMemcpyHostToDevice));
        }
        float* dw; CK(hipMalloc(&dw, nw * 4));
        const size_t wsb = clhip_conv_wgrad_ws_bytes(cs.N, cs.H, cs.W, cs.C, cs.C, cs.K, cs.ks, cs.st, pad, CLHIP_BF16);
        void* ws = nullptr; if (wsb) CK(hipMalloc(&ws, wsb));
        auto run = [&](int set) { return clhip_conv_wgrad(dx[set], dz[set], dw, ws, cs.N, cs.H, cs.W, cs.C, cs.C, cs.K, cs.ks, cs.st, pad, CLHIP_BF16, st); };
        bool fail = false;
        for (int pass = 0; pass < 2 && !fail; ++pass) {
            CK(hipMemcpyAsync(dw, h0.data(), nw * 4, hipMemcpyHostToDevice, st));
            if (ws) CK(hipMemsetAsync(ws, 0xff, wsb, st));
            int rc = run(0);
            if (rc) { printf("%-26s -> error %d: %s\n", cs.name, rc, clhip_last_error()); fail = true; break; }
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy((pass ? h2 : h1).data(), dw, nw * 4, hipMemcpyDeviceToHost));
        }
        if (fail) continue;
        const bool repro = !memcmp(h1.data(), h2.data(), nw * 4);
        double worst = 0;
        for (int q = 0; q < 1024; ++q) {
            size_t e = q < 64 ? (size_t)q * (nw / 64) : irand() % nw;
            if (q == 1023) e = nw - 1;
            const int c = (int)(e % cs.C), t = (int)((e / cs.C) % T), o = (int)(e / cs.C / T), r = t / cs.ks, s = t % cs.ks;
            double a = h0[e], mag = fabs(a);
            for (int n = 0; n < cs.N; ++n) for (int h = 0; h < Ho; ++h) {
                const int hh = h * cs.st + r - pad; if (hh < 0 || hh >= cs.H) continue;
                for (int w = 0; w < Wo; ++w) {
                    const int ww = w * cs.st + s - pad; if (ww < 0 || ww >= cs.W) continue;
                    const double tt = (double)b2f(hz[(((size_t)n * Ho + h) * Wo + w) * cs.K + o]) * b2f(hx[(((size_t)n * cs.H + hh) * cs.W + ww) * cs.C + c]);
                    a += tt; mag += fabs(tt);
                }
            }
            const double err = fabs(h1[e] - a) / (mag * 2e-6 + 1e-6);      // 1 = fp32 accumulation slack over the whole reduction
            if (!(err <= worst)) worst = err;
        }
        if (trace) {
            unsigned long long* dt; CK(hipMalloc(&dt, 128 * 8)); CK(hipMemset(dt, 0, 128 * 8));
            clhip_wgrad4_set_trace(dt);
            run(0); CK(hipStreamSynchronize(st));
            clhip_wgrad4_set_trace(nullptr);
            std::vector<unsigned long long> ht(128); CK(hipMemcpy(ht.data(), dt, 128 * 8, hipMemcpyDeviceToHost));
            for (int h = 0; h < 2; ++h) {
                printf("wave %d stamps (deltas, ticks of the 100 MHz s_memtime counter = 10 ns):", h * 4);
                for (int i = 1; i < 64 && ht[h * 64 + i]; ++i) printf(" %llu", ht[h * 64 + i] - ht[h * 64 + i - 1]);
                printf("\n");
            }
            hipFree(dt);
        }
        for (int i = 0; i < 5; ++i) run(i % nset);
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) run(i % nset);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) run(0);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-26s %7.1f us %6.0f TF/s (hot %6.1f us)  ws %.1f MB  err %.2f %s%s\n", cs.name, us, 2.0 * M * T * cs.C * cs.K / us * 1e-6, ms * 1e3 / reps, wsb / 1048576.0, worst,
               repro ? "" : " NOT-REPRODUCIBLE", worst > 1.0 ? "  <-- MISMATCH" : "");
        fflush(stdout);
        for (int i = 0; i < nset; ++i) { hipFree(dx[i]); hipFree(dz[i]); }
        hipFree(dw); if (ws) hipFree(ws);
    }
    return 0;
}
