// gemm_bench -- standalone timing harness for clhip_gemm_nt (bf16, no epilogue) behind the C ABI: tools/ubench/gemm_bench M N K [reps] [trace]
//   build: hipcc --offload-arch=gfx950 -O2 tools/ubench/gemm_bench.cpp -Iinclude -Llibcontinual_amd -lclhip -Wl,-rpath,'$ORIGIN/../../libcontinual_amd' -o tools/ubench/gemm_bench
// `trace` prints the s_memtime phase stamps of workgroup 0 (link against libclhip_abl.so: ..._abl).  Correctness: 512 entries against fp64.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "clhip.h"
// tuning / ablation hooks go through the library's one configuration entry point (include/clhip.h: clhip_config)
static void cfg_int(const char* key, long long v) { char b[32]; snprintf(b, sizeof(b), "%lld", v); clhip_config(key, b); }
static void cfg_ptr(const char* key, const void* ptr) { char b[32]; snprintf(b, sizeof(b), "%llu", (unsigned long long)(uintptr_t)ptr); clhip_config(key, b); }
static void clhip_gemm5_set_trace(unsigned long long* d) { cfg_ptr("GEMM5_TRACE", d); }
static void clhip_gemm5_set_debug(int bits) { cfg_int("GEMM5_DEBUG", bits); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
static uint16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float b2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static uint32_t rng_state = 4242;
static uint32_t irand() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 4; }
static float urand() { return (irand() & 0xffff) / 32768.0f - 1.0f; }
int main(int argc, char** argv) {
    if (argc < 4) { printf("usage: gemm_bench M N K [reps] [trace]\n"); return 1; }
    const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]);
    const int reps = argc > 4 ? atoi(argv[4]) : 30;
    const bool trace = argc > 5 && !strcmp(argv[5], "trace");
    const size_t na = (size_t)M * K, nb = (size_t)N * K, nc = (size_t)M * N;
    int nset = (int)((300u << 20) / ((na + nb + nc) * 2)) + 1; if (nset > 8) nset = 8; if (nset < 2) nset = 2;
    std::vector<uint16_t> ha(na), hb(nb), hc(nc);
    for (auto& v : ha) v = f2b(urand());
    for (auto& v : hb) v = f2b(urand() * 0.05f);
    std::vector<void*> da(nset), db(nset), dc(nset);
    for (int i = 0; i < nset; ++i) {
        CK(hipMalloc(&da[i], na * 2)); CK(hipMalloc(&db[i], nb * 2)); CK(hipMalloc(&dc[i], nc * 2));
        CK(hipMemcpy(da[i], ha.data(), na * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(db[i], hb.data(), nb * 2, hipMemcpyHostToDevice));
    }
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](int s) { return clhip_gemm_nt(da[s], db[s], dc[s], nullptr, nullptr, nullptr, M, N, K, K, K, N, 0, 0, 0, CLHIP_BF16, st); };
    CK(hipMemsetAsync(dc[0], 0xff, nc * 2, st));
    if (int rc = run(0)) { printf("error %d: %s\n", rc, clhip_last_error()); return 1; }
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(hc.data(), dc[0], nc * 2, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int q = 0; q < 512; ++q) {
        const size_t m = q < 8 ? (size_t)(M - 1 - q) : irand() % M, n = q < 8 ? (size_t)(N - 1 - q * 3) : irand() % N;
        double a = 0, mag = 0;
        for (int k = 0; k < K; ++k) { const double t = (double)b2f(ha[m * K + k]) * b2f(hb[n * K + k]); a += t; mag += fabs(t); }
        const double err = fabs(b2f(hc[m * N + n]) - a) / (fabs(a) * 0.0079 + mag * 1e-6 + 1e-6);
        if (!(err <= worst)) worst = err;
    }
    if (trace) {
        unsigned long long* dt; CK(hipMalloc(&dt, 512 * 8)); CK(hipMemset(dt, 0, 512 * 8));
        clhip_gemm5_set_trace(dt);
        run(0); CK(hipStreamSynchronize(st));
        clhip_gemm5_set_trace(nullptr);
        std::vector<unsigned long long> ht(512); CK(hipMemcpy(ht.data(), dt, 512 * 8, hipMemcpyDeviceToHost));
        for (int h = 0; h < 2; ++h) {
            printf("wave %d stamps (deltas, shader-clock ticks):", h * 4);
            for (int i = 1; i < 256 && ht[h * 256 + i]; ++i) printf(" %llu", ht[h * 256 + i] - ht[h * 256 + i - 1]);
            printf("\n");
        }
    }
    if (argc > 5 && !strcmp(argv[5], "abl")) {            // ablation build: the loop with pieces removed (results are garbage)
        for (int mk : {0, 8, 1, 2, 4, 1 | 2 | 4, 1 | 2 | 4 | 8}) {
            clhip_gemm5_set_debug(mk);
            for (int i = 0; i < 3; ++i) run(i % nset);
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < reps; ++i) run(i % nset);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms2; CK(hipEventElapsedTime(&ms2, e0, e1));
            printf("debug %d (%s%s%s%s): %8.1f us\n", mk, mk & 1 ? "-mfma " : "", mk & 2 ? "-dma " : "", mk & 4 ? "-ldsread " : "", mk & 8 ? "-store " : "", ms2 * 1e3 / reps);
        }
        clhip_gemm5_set_debug(0);
    }
    for (int i = 0; i < 5; ++i) run(i % nset);
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) run(i % nset);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("gemm %d x %d x %d: %8.1f us %6.0f TF/s  err %.2f%s\n", M, N, K, ms * 1e3 / reps, 2.0 * M * N * K / (ms / reps) * 1e-9, worst, worst > 1.5 ? "  <-- MISMATCH" : "");
    return 0;
}
