// conv_bench -- standalone timing + spot-check harness for the 3x3 convolution kernels behind the C ABI (no torch, no Python:
// a gpurun call can sweep many kernel configurations in seconds).
//   build: hipcc --offload-arch=gfx950 -O2 tools/ubench/conv_bench.cpp -Iinclude -Llibcontinual_amd -lclhip -Wl,-rpath,'$ORIGIN/../../libcontinual_amd' -o tools/ubench/conv_bench
//   run  : tools/ubench/conv_bench [case-filter]
// Every case: random bf16 operands, the launch through clhip_conv_fwd_acc / clhip_conv_dgrad, a CPU fp64 check of every channel of
// the first and last image and of 256 random pixels, BatchNorm sums checked against the kernel's own output, then the timing
// (rotating over buffer sets larger than the Infinity Cache).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <string>
#include "clhip.h"

// tuning / ablation hooks go through the library's one configuration entry point (include/clhip.h: clhip_config)
static void cfg_int(const char* key, long long v) { char b[32]; snprintf(b, sizeof(b), "%lld", v); clhip_config(key, b); }
static void cfg_ptr(const char* key, const void* ptr) { char b[32]; snprintf(b, sizeof(b), "%llu", (unsigned long long)(uintptr_t)ptr); clhip_config(key, b); }
static void clhip_conv4_set_cfg(int wm, int wn, int kg, int ck) { char b[64]; snprintf(b, sizeof(b), "%d,%d,%d,%d", wm, wn, kg, ck); clhip_config("CONV4_FORCE_CFG", b); }
static void clhip_conv4_enable(int on) { cfg_int("CONV4_ENABLE", on); }
static void clhip_conv4_set_debug(int bits) { cfg_int("CONV4_DEBUG", bits); }
static void clhip_conv4_set_trace(unsigned long long* d) { cfg_ptr("CONV4_TRACE", d); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

static uint16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float b2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static uint32_t rng_state = 12345;
static float urand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xffff) / 32768.0f - 1.0f; }

struct Case { const char* name; int N, H, W, C, K, mode; };   // mode 0 fwd (C -> K), 1 dgrad (dz has K channels, dx has C)
struct Cfg { int on, wm, wn, kg, ck; };

int main(int argc, char** argv) {
    const char* filt = argc > 1 ? argv[1] : "";
    const int reps = argc > 2 ? atoi(argv[2]) : 40;
    // ablation mode: conv_bench <filter> <reps> abl wm,wn,kg,ck  -> one configuration under every debug mask of interest (no checks)
    const bool abl = argc > 4 && !strcmp(argv[3], "abl");
    const bool trace = argc > 5 && !strcmp(argv[5], "trace");   // ... one wm,wn,kg,ck trace : phase stamps of workgroup 0 (ablation build only)
    const bool one9 = argc > 4 && !strcmp(argv[3], "one9");    // ... conv9.hip only
    const bool one8 = argc > 4 && !strcmp(argv[3], "one8");    // conv_bench <filter> <reps> one8 - [trace] : conv8.hip only
    const bool one = argc > 4 && !strcmp(argv[3], "one");      // conv_bench <filter> <reps> one wm,wn,kg,ck : only that configuration (PMC passes)
    int acfg[4] = {0, 0, 0, 0};
    if (abl || one) sscanf(argv[4], "%d,%d,%d,%d", &acfg[0], &acfg[1], &acfg[2], &acfg[3]);
    std::vector<Case> cases = {
        {"L1f 256x32x32 64->64", 256, 32, 32, 64, 64, 0},   {"L1d 256x32x32 64->64", 256, 32, 32, 64, 64, 1},
        {"L2f 256x16x16 128->128", 256, 16, 16, 128, 128, 0}, {"L2d 256x16x16 128->128", 256, 16, 16, 128, 128, 1},
        {"L3f 256x8x8 256->256", 256, 8, 8, 256, 256, 0},   {"L3d 256x8x8 256->256", 256, 8, 8, 256, 256, 1},
        {"L4f 256x4x4 512->512", 256, 4, 4, 512, 512, 0},   {"L4d 256x4x4 512->512", 256, 4, 4, 512, 512, 1},
        {"S3f 256x8x8 64->64", 256, 8, 8, 64, 64, 0},       {"S3f32 32x8x8 64->64", 32, 8, 8, 64, 64, 0},
        {"X256f 256x16x16 256->128", 256, 16, 16, 256, 128, 0}, {"X512f 256x16x16 512->128", 256, 16, 16, 512, 128, 0},
        {"L1f32 32x32x32 64->64", 32, 32, 32, 64, 64, 0},   {"odd 3x12x20 64->128", 3, 12, 20, 64, 128, 0},
    };
    std::vector<Cfg> cfgs = {
        {0, 0, 0, 0, 0}, {1, 0, 0, 0, 0}, {2, 0, 0, 0, 0}, {3, 0, 0, 0, 0}, {4, 0, 0, 0, 0},      // 4: conv9.hip; on = 2: the weight-stationary kernel (conv5.hip) where it applies; 3: conv8.hip
        {1, 4, 1, 1, 32}, {1, 4, 1, 1, 64}, {1, 2, 1, 1, 32}, {1, 2, 1, 1, 64}, {1, 2, 2, 1, 32}, {1, 2, 2, 1, 64}, {1, 4, 2, 1, 32}, {1, 4, 2, 1, 64},
        {1, 2, 1, 2, 32}, {1, 2, 1, 2, 64}, {1, 1, 2, 2, 32}, {1, 2, 2, 2, 32}, {1, 1, 1, 4, 32}, {1, 1, 1, 4, 64}, {1, 2, 1, 4, 32}, {1, 2, 1, 4, 64}, {1, 1, 2, 4, 32},
    };
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Case& cs : cases) {
        if (!strstr(cs.name, filt)) continue;
        const int Cs = cs.mode == 0 ? cs.C : cs.K, Cd = cs.mode == 0 ? cs.K : cs.C;   // gathered / produced channel counts
        const size_t M = (size_t)cs.N * cs.H * cs.W;
        const size_t nsrc = M * Cs, ndst = M * Cd, nw = (size_t)Cd * 9 * Cs;
        size_t per_set = (nsrc + ndst) * 2;
        int nset = (int)((300u << 20) / per_set) + 1; if (nset > 8) nset = 8; if (nset < 2) nset = 2;
        std::vector<uint16_t> hs(nsrc), hw(nw), hd(ndst), hold(ndst);
        for (auto& v : hs) v = f2b(urand());
        for (auto& v : hw) v = f2b(urand() * 0.06f);
        for (auto& v : hold) v = f2b(urand());
        std::vector<void*> dsrc(nset), ddst(nset);
        void* dw; double* dacc; const int rep = 8;
        for (int i = 0; i < nset; ++i) { CK(hipMalloc(&dsrc[i], nsrc * 2)); CK(hipMalloc(&ddst[i], ndst * 2)); CK(hipMemcpy(dsrc[i], hs.data(), nsrc * 2, hipMemcpyHostToDevice)); }
        CK(hipMalloc(&dw, nw * 2)); CK(hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice));
        CK(hipMalloc(&dacc, rep * 2 * Cd * sizeof(double)));
        const double flop = 2.0 * M * 9.0 * Cs * Cd;
        if (abl) {
            clhip_conv4_enable(1);
            clhip_conv4_set_cfg(acfg[0], acfg[1], acfg[2], acfg[3]);
            const int masks[] = {0, 16, 8, 24, 1, 2, 4, 64, 32, 1 | 64, 2 | 32, 1 | 2 | 64, 1 | 2 | 4 | 64, 1 | 2 | 4 | 8 | 16 | 64, 1 | 2 | 4 | 8 | 16 | 32 | 64};
            for (int mk : masks) {
                clhip_conv4_set_debug(mk);
                auto run1 = [&](int set) { return cs.mode == 0 ? clhip_conv_fwd_acc(dsrc[set], dw, ddst[set], dacc, rep, cs.N, cs.H, cs.W, cs.C, cs.K, 3, 1, 1, CLHIP_BF16, st)
                                                             : clhip_conv_dgrad(dsrc[set], dw, ddst[set], 0, cs.N, cs.H, cs.W, cs.C, cs.K, 3, 1, 1, CLHIP_BF16, st); };
                for (int i = 0; i < 5; ++i) run1(i % nset);
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < reps; ++i) run1(i % nset);
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("%-26s cfg %s debug %3d (%s%s%s%s%s%s%s): %7.1f us\n", cs.name, argv[4], mk, mk & 1 ? "-mfma " : "", mk & 2 ? "-dma " : "", mk & 4 ? "-patch " : "",
                       mk & 8 ? "-store " : "", mk & 16 ? "-stats " : "", mk & 32 ? "-barrier " : "", mk & 64 ? "-ldsread " : "", ms * 1e3 / reps);
                fflush(stdout);
            }
            clhip_conv4_set_debug(0);
            continue;
        }
        if (one) { cfgs.clear(); cfgs.push_back(Cfg{acfg[0] > 0 ? 1 : 0, acfg[0], acfg[1], acfg[2], acfg[3]}); }
        if (one8) { cfgs.clear(); cfgs.push_back(Cfg{3, 0, 0, 0, 0}); }
        if (one9) { cfgs.clear(); cfgs.push_back(Cfg{4, 0, 0, 0, 0}); }
        for (const Cfg& cf : cfgs) {
            clhip_conv4_enable(cf.on ? 1 : 0);
            clhip_config("CONV5", cf.on == 2 ? "1" : "0");
            clhip_config("CONV5_MIN_TILES", "1");
            clhip_config("CONV8", cf.on == 3 ? "1" : "0");
            clhip_config("CONV8_MIN_TILES", "1");
            clhip_config("CONV9", cf.on == 4 ? "1" : "0");
            if (cf.on == 4 && !((Cs == 128 || Cs == 256) && Cs == Cd)) continue;
            if ((cf.on == 2 || cf.on == 3) && !(Cs == 64 && Cd == 64)) continue;
            if (cf.on == 3 && !(128 % cs.W == 0 && cs.W >= 8 && cs.H % (128 / cs.W) == 0)) continue;
            clhip_conv4_set_cfg(cf.wm, cf.wn, cf.kg, cf.ck);
            if (cf.wm > 0) {
                if (Cd % (cf.wn * 64) || Cs % cf.ck || (Cs / cf.ck) % cf.kg) continue;
            }
            auto run = [&](int set, int accumulate) -> int {
                if (cs.mode == 0) return clhip_conv_fwd_acc(dsrc[set], dw, ddst[set], dacc, rep, cs.N, cs.H, cs.W, cs.C, cs.K, 3, 1, 1, CLHIP_BF16, st);
                return clhip_conv_dgrad(dsrc[set], dw, ddst[set], accumulate, cs.N, cs.H, cs.W, cs.C, cs.K, 3, 1, 1, CLHIP_BF16, st);
            };
            // ---- correctness (accumulate variant for dgrad on the second pass)
            double worst = 0, worst_stat = 0;
            bool fail = false;
            for (int pass = 0; pass < (cs.mode == 1 ? 2 : 1); ++pass) {
                CK(hipMemsetAsync(dacc, 0, rep * 2 * Cd * sizeof(double), st));
                if (pass == 1) CK(hipMemcpyAsync(ddst[0], hold.data(), ndst * 2, hipMemcpyHostToDevice, st));
                else CK(hipMemsetAsync(ddst[0], 0xff, ndst * 2, st));
                int rc = run(0, pass);
                if (rc) { printf("%-26s cfg %d:%d,%d,%d,%d  -> error %d: %s\n", cs.name, cf.on, cf.wm, cf.wn, cf.kg, cf.ck, rc, clhip_last_error()); fail = true; break; }
                CK(hipStreamSynchronize(st));
                CK(hipMemcpy(hd.data(), ddst[0], ndst * 2, hipMemcpyDeviceToHost));
                std::vector<size_t> pix;
                const size_t hwp = (size_t)cs.H * cs.W;
                for (size_t q = 0; q < hwp; ++q) { pix.push_back(q); pix.push_back(M - hwp + q); }
                for (int q = 0; q < 256; ++q) { rng_state = rng_state * 1664525u + 1013904223u; pix.push_back((size_t)(rng_state >> 4) % M); }
                for (size_t g : pix) {
                    const int w = (int)(g % cs.W), h = (int)((g / cs.W) % cs.H); const size_t n = g / hwp;
                    for (int o = 0; o < Cd; ++o) {
                        double a = pass == 1 ? (double)b2f(hold[g * Cd + o]) : 0.0, mag = fabs(a);
                        for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) {
                            const int hh = cs.mode == 0 ? h + r - 1 : h + 1 - r, ww = cs.mode == 0 ? w + s - 1 : w + 1 - s;
                            if (hh < 0 || hh >= cs.H || ww < 0 || ww >= cs.W) continue;
                            const uint16_t* xs = &hs[((n * cs.H + hh) * cs.W + ww) * Cs];
                            const uint16_t* wr = &hw[((size_t)o * 9 + r * 3 + s) * Cs];
                            for (int c = 0; c < Cs; ++c) { const double t = (double)b2f(xs[c]) * b2f(wr[c]); a += t; mag += fabs(t); }
                        }
                        const double got = b2f(hd[g * Cd + o]);
                        const double err = fabs(got - a) / (fabs(a) * 0.0079 + mag * 1e-6 + 1e-6);   // 1 = one bf16 rounding + fp32 accumulation slack
                        if (!(err <= worst)) worst = err;
                    }
                }
                if (cs.mode == 0) {
                    std::vector<double> ha(rep * 2 * Cd), s1(Cd, 0.0), s2(Cd, 0.0);
                    CK(hipMemcpy(ha.data(), dacc, ha.size() * sizeof(double), hipMemcpyDeviceToHost));
                    for (size_t g = 0; g < M; ++g) for (int o = 0; o < Cd; ++o) { const double v = b2f(hd[g * Cd + o]); s1[o] += v; s2[o] += v * v; }
                    for (int o = 0; o < Cd; ++o) {
                        double g1 = 0, g2 = 0;
                        for (int r2 = 0; r2 < rep; ++r2) { g1 += ha[(r2 * 2 + 0) * Cd + o]; g2 += ha[(r2 * 2 + 1) * Cd + o]; }
                        const double e1_ = fabs(g1 - s1[o]) / (sqrt(s2[o] * M) * 2e-3 + 1e-3), e2_ = fabs(g2 - s2[o]) / (s2[o] * 4e-3 + 1e-3);
                        if (!(e1_ <= worst_stat)) worst_stat = e1_;
                        if (!(e2_ <= worst_stat)) worst_stat = e2_;
                    }
                }
            }
            if (fail) continue;
            if (trace) {
                unsigned long long* dt; CK(hipMalloc(&dt, 512 * 8)); CK(hipMemset(dt, 0, 512 * 8));
                clhip_conv4_set_trace(dt);
                cfg_ptr("CONV8_TRACE", dt);
                cfg_ptr("CONV9_TRACE", dt);
                run(0, 0); CK(hipStreamSynchronize(st));
                cfg_ptr("CONV9_TRACE", nullptr);
                if (cf.on == 4) {
                    std::vector<unsigned long long> h9(512); CK(hipMemcpy(h9.data(), dt, 512 * 8, hipMemcpyDeviceToHost));
                    for (int w9 = 0; w9 < 8; ++w9) {
                        printf("conv9 wave %d t0 %+lld:", w9, (long long)(h9[w9 * 64] - h9[0]));
                        for (int i = 1; i < 64 && h9[w9 * 64 + i]; ++i) printf(" %llu", h9[w9 * 64 + i] - h9[w9 * 64 + i - 1]);
                        printf("\n");
                    }
                }
                clhip_conv4_set_trace(nullptr);
                cfg_ptr("CONV8_TRACE", nullptr);
                if (cf.on == 3) {
                    std::vector<unsigned long long> h8(512); CK(hipMemcpy(h8.data(), dt, 512 * 8, hipMemcpyDeviceToHost));
                    for (int w8 = 0; w8 < 8; ++w8) {
                        printf("conv8 wg %d wave %d hwid %08llx xcc %llx t0 %+lld:", w8 >> 2, w8 & 3, h8[w8 * 64 + 63] & 0xffffffffull, h8[w8 * 64 + 63] >> 32, (long long)(h8[w8 * 64] - h8[0]));
                        for (int i = 1; i < 62 && h8[w8 * 64 + i]; ++i) printf(" %llu", h8[w8 * 64 + i] - h8[w8 * 64 + i - 1]);
                        printf("\n");
                    }
                }
                std::vector<unsigned long long> ht(512); CK(hipMemcpy(ht.data(), dt, 512 * 8, hipMemcpyDeviceToHost));
                for (int h = 0; h < 2; ++h) {
                    printf("wave %d stamps (deltas, cycles of the 100 MHz-free s_memtime counter):", h * 4);
                    for (int i = 1; i < 256 && ht[h * 256 + i]; ++i) printf(" %llu", ht[h * 256 + i] - ht[h * 256 + i - 1]);
                    printf("\n   first stamp offset vs wave 0: %lld\n", (long long)(ht[h * 256] - ht[0]));
                }
                hipFree(dt);
            }
            // ---- timing
            for (int i = 0; i < 5; ++i) run(i % nset, 0);
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < reps; ++i) run(i % nset, 0);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / reps;
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < reps; ++i) run(0, 0);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us_hot = ms * 1e3 / reps;
            printf("%-26s cfg %d:%d,%d,%d,%-2d  %7.1f us %6.0f TF/s (hot %6.1f us)  err %.2f stat %.2f %s\n", cs.name, cf.on, cf.wm, cf.wn, cf.kg, cf.ck, us, flop / us * 1e-6, us_hot,
                   worst, worst_stat, (worst > 1.5 || worst_stat > 1.0) ? "  <-- MISMATCH" : "");
            fflush(stdout);
        }
        for (int i = 0; i < nset; ++i) { hipFree(dsrc[i]); hipFree(ddst[i]); }
        hipFree(dw); hipFree(dacc);
    }
    return 0;
}
