// api.hip -- error reporting, version and the run-time configuration of the C ABI (include/clhip.h).
#include <map>
#include <utility>
#include <vector>
#include <mutex>
#include <string>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

static thread_local char g_err[512] = "";

void clhip_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* clhip_last_error(void) { return g_err; }
extern "C" int clhip_version(void) { return 103; }

// ---- configuration: ONE documented entry point (clhip_config) instead of ad-hoc exports and scattered getenv calls.  Every switch
//      has a name (the table below = the list in include/clhip.h); its value is what clhip_config() last set or, if never set, the
//      environment variable CLHIP_<NAME> (kept so that an unmodified caller can still be steered from the shell).  Kernels look a
//      switch up through clhip_cfg(); most look-ups are cached at their first use, so configure BEFORE the first launch / plan.
namespace {
const char* const kKeys[] = {
    // dispatch switches (A/B runs, tests that pin a code path)
    "ATTN_GENERIC", "ATTN_BWD", "BN_FUSE", "BN_FUSE_MAX_M", "BN_MASK_BITS", "BN_MASK_FROM_Y", "BN_PARTIALS", "CE_ROWS", "CE_ONE_WG", "LINEAR_BWD_SPLIT", "CONV3G", "CONV4", "CONV_V1",
    "GEMM_NO_SPLIT", "GEMM_TAIL", "NO_CONV16", "NO_CONV3", "NO_PARITY_DGRAD", "NO_SHORTCUT", "NO_STEM", "PREP_NARROW", "WGRAD4",
    "WGRAD_NO_TR", "WGRAD2_ATOMIC", "WGRAD_DEFER", "WGRAD_DEFER_SIDE", "BWD_FUSED", "WGRAD_STREAM", "BRANCH_STREAM", "WGRAD_ALWAYS_QUEUE", "SIDE_PRIO", "EVENT_FLAGS", "EVENT_RECORD", "CONV5", "CONV6", "WGRAD16_PARTS", "WGRAD32_IPG", "WGRAD64_IPG", "WGRAD64_IPI2", "CONV64", "CONV64_BM", "CONV64_FWD", "CONV6_PAIR", "CONV7", "CONV7_TPW", "WGRAD7", "FWD7", "PAIR_BN_FUSE", "POOL_BN_FUSE", "BN_INPUT", "BN_INPUT_WT", "BN_RES_INPUT", "BN_GRAD", "BN_GRAD_MINC", "BN_GRAD_RES", "CONV6_DEBUG", "BN_ONEPASS", "WGRAD5", "WGRAD32",
    // tuning values
    "BN_ACC_CPT", "BN_BWD_ITERS", "CONV3_CFG", "CONV4_CFG", "CONV4_GRID", "GEMM_GROUP_M", "GEMM_MT", "IGEMM_TILE",
    "SHORTCUT_MIN_PIXELS", "STEM_GRID", "STEM_WGRAD_GRID", "WGRAD4_MIN_STEPS", "WGRAD4_MIN_TOTAL", "WGRAD_NET_GFLOP", "WGRAD_TARGET",
    "CONV5_MIN_TILES", "CONV5_GRID", "CONV64_MAX_W", "CONV64_MAX_M", "CONV8", "CONV8_MIN_TILES", "DZ_BUFFERS", "CONV8_GRID", "CONV8_OPT", "CONV8_BNR", "CONV9", "PLAN_SKIP", "WT_DEBUG", "EVAL_LAZY", "GEMM_SPLITK", "STREAM_PROBE", "STAGE_EVAL", "STAGE_TRAIN", "STAGE_TRAIN_BWD", "STAGE_TRACE", "STAGE_ENTRY", "STAGE_ENTRY_BWD", "STAGE_POOL", "STAGE_XCH3",
    // micro-benchmark / ablation hooks (tools/ubench): applied immediately, not cached
    "CONV3_DEBUG", "WGRAD_DEBUG", "CONV4_FORCE_CFG", "CONV4_ENABLE", "CONV4_DEBUG", "CONV4_TRACE", "WGRAD4_TRACE", "CONV6_TRACE", "CONV8_TRACE", "CONV9_TRACE",
};
std::mutex g_stream_mu;
std::mutex g_cfg_mu;
// values are strdup'ed and never freed: look-up sites cache the pointer (a few bytes per clhip_config call, by design)
std::map<std::string, const char*>& cfg_map() { static std::map<std::string, const char*> m; return m; }
bool known_key(const char* k) {
    for (const char* q : kKeys) if (strcmp(q, k) == 0) return true;
    return false;
}
}  // namespace

namespace {
__global__ void stream_probe_spin_kernel(unsigned long long* out, long long ticks) {       // wall_clock64: the constant 100-MHz counter
    const unsigned long long t0 = wall_clock64();
    while ((long long)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(64);
    out[0] = wall_clock64();
}
__global__ void stream_probe_stamp_kernel(unsigned long long* out) { out[1] = wall_clock64(); }

struct SharedStreams {
    std::vector<hipStream_t> pool;                                   // candidates in creation order
    std::map<std::pair<hipStream_t, int>, hipStream_t> chosen;       // (main stream, role) -> stream
    hipStream_t low_prio = nullptr;                                   // SIDE_PRIO=1: a lowest-priority stream (a queue class of its own: no probing needed)
    unsigned long long* probe_buf = nullptr;
};
SharedStreams g_shared[16];

// can a kernel on `b` start while a kernel on `a` is running?  (false also when the probe itself fails: the caller then tries the next candidate)
bool runs_beside(SharedStreams& S, hipStream_t a, hipStream_t b) {
    if (S.probe_buf == nullptr && hipMalloc(reinterpret_cast<void**>(&S.probe_buf), 2 * sizeof(unsigned long long)) != hipSuccess) return false;
    hipLaunchKernelGGL(stream_probe_spin_kernel, dim3(1), dim3(1), 0, a, S.probe_buf, 30000ll);
    hipLaunchKernelGGL(stream_probe_stamp_kernel, dim3(1), dim3(1), 0, b, S.probe_buf);
    if (hipStreamSynchronize(b) != hipSuccess || hipStreamSynchronize(a) != hipSuccess) return false;
    unsigned long long h[2] = {0, 0};
    if (hipMemcpy(h, S.probe_buf, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return false;
    return h[1] < h[0];
}
}  // namespace

hipStream_t clhip_shared_stream(int role, hipStream_t main_s, bool low_priority) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16 || role < 0 || role > 1) return nullptr;
    std::lock_guard<std::mutex> lk(g_stream_mu);
    SharedStreams& S = g_shared[dev];
    if (low_priority) {
        if (S.low_prio == nullptr) {
            int prio_lo = 0, prio_hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
            if (hipStreamCreateWithPriority(&S.low_prio, hipStreamNonBlocking, prio_lo) != hipSuccess) S.low_prio = nullptr;
        }
        return S.low_prio;
    }
    // ONE extra stream serves both roles: the branch launches belong to the forward, the weight gradients to the backward, and the step measured 1.3-2 % FASTER
    // with the two on one hardware queue than with a queue each (1.957-1.962 vs 1.976-1.991 ms alternating on one box; 2.014 vs 2.058 on another;
    // profiles/r05_notes.md) -- what used to happen by accident at GPU_MAX_HW_QUEUES=3 and not at 4.
    (void)role;
    const auto key = std::make_pair(main_s, 0);
    auto it = S.chosen.find(key);
    if (it != S.chosen.end()) return it->second;
    {
        // the first request for this caller's stream MEASURES (spin + stamp kernels, two stream synchronisations, a hipMalloc): none of that may happen inside a
        // stream capture -- the probe's nodes would land in the caller's graph and the synchronisation would invalidate the capture.  A capturing caller gets an
        // already chosen stream of another main stream or the first candidate, unprobed and NOT remembered: the next eager call measures.
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(main_s, &cap);
        if (cap != hipStreamCaptureStatusNone) {
            for (const auto& kv : S.chosen) if (kv.second != main_s) return kv.second;
            for (hipStream_t c : S.pool) if (c != main_s) return c;
            return nullptr;                                          // (no stream yet and none may be created now: the caller keeps to its own stream)
        }
    }
    const int probe_cfg = clhip_cfg("STREAM_PROBE") != nullptr ? atoi(clhip_cfg("STREAM_PROBE")) : 1;      // 0: the first candidate, unmeasured
    hipStream_t pick = nullptr, fallback = nullptr;
    for (size_t k = 0; k < 8 && pick == nullptr; ++k) {
        if (k == S.pool.size()) {
            hipStream_t s = nullptr;
            if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) break;
            S.pool.push_back(s);
        }
        hipStream_t c = S.pool[k];
        if (c == main_s) continue;
        if (fallback == nullptr) fallback = c;
        if (probe_cfg == 0 || runs_beside(S, main_s, c)) pick = c;
    }
    if (pick == nullptr) pick = fallback;              // every candidate shares the caller's hardware queue (GPU_MAX_HW_QUEUES=1): correct, only slower
    if (pick != nullptr) S.chosen[key] = pick;
    return pick;
}

// hooks that take effect at once (defined next to the kernels they steer)
void clhip_conv4_set_cfg(int wm, int wn, int kg, int ck);
void clhip_conv4_enable(int on);
void clhip_conv4_set_debug(int bits);
void clhip_conv4_set_trace(unsigned long long* dev_buf);
void clhip_wgrad4_set_trace(unsigned long long* dev_buf);
void clhip_conv5_enable(int on);
void clhip_conv5_min_tiles(int n);
void clhip_conv6_enable(int on);
void clhip_conv8_enable(int on);
void clhip_conv9_enable(int on);
void clhip_conv9_set_trace(unsigned long long* dev_buf);
void clhip_conv8_min_tiles(int n);
void clhip_conv8_set_trace(unsigned long long* dev_buf);
void clhip_conv6_set_trace(unsigned long long* buf, int wg);

const char* clhip_cfg(const char* name) {
    {
        std::lock_guard<std::mutex> lk(g_cfg_mu);
        auto it = cfg_map().find(name);
        if (it != cfg_map().end()) return it->second;
    }
    char env[96];
    snprintf(env, sizeof(env), "CLHIP_%s", name);
    return getenv(env);
}

extern "C" const char* clhip_config_get(const char* key) {
    if (key == nullptr) return nullptr;
    if (strncmp(key, "CLHIP_", 6) == 0) key += 6;
    std::lock_guard<std::mutex> lk(g_cfg_mu);
    auto it = cfg_map().find(key);
    return it != cfg_map().end() ? it->second : nullptr;
}

extern "C" int clhip_config(const char* key, const char* value) {
    CLHIP_CHECK_ARG(key != nullptr);
    if (strncmp(key, "CLHIP_", 6) == 0) key += 6;
    if (!known_key(key)) { clhip_set_error("clhip_config: unknown switch '%s'", key); return CLHIP_EINVAL; }
    const char* v = value ? value : "";
    if (strcmp(key, "CONV4_FORCE_CFG") == 0) {
        int c[4] = {0, 0, 0, 0};
        if (value && sscanf(value, "%d,%d,%d,%d", &c[0], &c[1], &c[2], &c[3]) != 4) { clhip_set_error("clhip_config: CONV4_FORCE_CFG wants \"wm,wn,kg,ck\""); return CLHIP_EINVAL; }
        clhip_conv4_set_cfg(c[0], c[1], c[2], c[3]);
        return CLHIP_OK;
    }
    if (strcmp(key, "CONV4_ENABLE") == 0) { clhip_conv4_enable(value ? atoi(v) : -1); return CLHIP_OK; }
    if (strcmp(key, "CONV4_DEBUG") == 0) { clhip_conv4_set_debug(atoi(v)); return CLHIP_OK; }
    if (strcmp(key, "CONV4_TRACE") == 0) { clhip_conv4_set_trace(reinterpret_cast<unsigned long long*>(strtoull(v, nullptr, 0))); return CLHIP_OK; }
    if (strcmp(key, "CONV6_TRACE") == 0) {                         // "pointer[,workgroup]"
        char* end = nullptr;
        unsigned long long ptr = strtoull(v, &end, 0);
        clhip_conv6_set_trace(reinterpret_cast<unsigned long long*>(ptr), (end && *end == ',') ? atoi(end + 1) : 0);
        return CLHIP_OK;
    }
    if (strcmp(key, "CONV9_TRACE") == 0) { clhip_conv9_set_trace(reinterpret_cast<unsigned long long*>(strtoull(v, nullptr, 0))); return CLHIP_OK; }
    if (strcmp(key, "CONV8_TRACE") == 0) { clhip_conv8_set_trace(reinterpret_cast<unsigned long long*>(strtoull(v, nullptr, 0))); return CLHIP_OK; }
    if (strcmp(key, "WGRAD4_TRACE") == 0) { clhip_wgrad4_set_trace(reinterpret_cast<unsigned long long*>(strtoull(v, nullptr, 0))); return CLHIP_OK; }
    if (strcmp(key, "CONV5") == 0) clhip_conv5_enable(value ? atoi(v) : -1);                   // immediate AND recorded below
    if (strcmp(key, "CONV6") == 0) clhip_conv6_enable(value ? atoi(v) : -1);
    if (strcmp(key, "CONV8") == 0) clhip_conv8_enable(value ? atoi(v) : -1);
    if (strcmp(key, "CONV9") == 0) clhip_conv9_enable(value ? atoi(v) : -1);
    if (strcmp(key, "CONV8_MIN_TILES") == 0) clhip_conv8_min_tiles(value ? atoi(v) : -1);
    if (strcmp(key, "CONV5_MIN_TILES") == 0) clhip_conv5_min_tiles(value ? atoi(v) : -1);
    std::lock_guard<std::mutex> lk(g_cfg_mu);
    if (value == nullptr) {
        cfg_map().erase(key);                 // back to the environment's value
    } else {
        cfg_map()[key] = strdup(value);
    }
    return CLHIP_OK;
}
