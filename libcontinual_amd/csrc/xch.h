// xch.h -- the in-launch all-reduce of the stage-level TRAINING kernels (stage_train.hip): every workgroup of a co-resident grid contributes NV fp32 values
// (the per-channel BatchNorm sums of its image) and every workgroup leaves with the SAME fp64 totals, summed in ONE fixed order -- bit-reproducible from
// run to run and identical in every workgroup, whatever the dispatch order or the workgroup -> XCD placement (cdna_hip_programming.md section 6 Guideline 16).
//
// Two hops of 8-byte {tag, value} granules, each written by ONE relaxed agent-scope (write-through) store and polled with relaxed agent-scope loads:
// the data is its own flag, so there is no fence, no flag ordering and no L2 write-back on the path.
//   hop 1   workgroup w stores its NV values as granules g1[w][v]
//   reduce  value v belongs to workgroup v % G: it polls g1[0..G)[v], sums the G values in fp64 in a fixed tree (chunks of 16 contributors, then the chunk
//           sums in order) and stores the total as two granules (low / high word) g2[v][0..1]
//   hop 2   every workgroup polls the 2 NV granules of g2
// Small grids take ONE hop instead (xch_sweep: every workgroup sweeps all hop-1 granules itself).  g1 is double-buffered by phase parity: a workgroup is never more
// than one phase ahead of the slowest one (it can finish phase p + 1 only after every workgroup has published p + 1, i.e. has finished reading phase p), so
// what it overwrites when it publishes phase p + 2 has been read by everybody.
// A tag is  base + phase + 1,  `base` being a counter in device memory that the launch itself advances when it ends (a kernel ARGUMENT would be frozen under
// graph replay); the granule arrays are zeroed once, when the plan that owns them is created, and tags only grow, so no per-launch memset is needed and a
// stale granule of an earlier phase / launch can never match.  Reuse of g1 / g2 by the next phase is safe: a workgroup publishes phase p + 1 only after it
// has read ALL totals of phase p, i.e. after EVERY reducer has finished reading phase p's hop-1 granules; and a reducer overwrites g2 for phase p + 1 only
// after every workgroup has published hop 1 of p + 1, i.e. has read all of g2 for phase p.
// Every spin is bounded (kXchSpinLimit polls, then the launch sets ctl[1] and carries on with whatever it has): a grid that is not co-resident produces a
// reported error instead of a hung GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((address_space(1))) unsigned long long xch_gu64;
typedef __attribute__((address_space(1))) unsigned xch_gu32;

struct XchBuf {
    unsigned long long* g1;   // [2 parities][G][NVmax]
    unsigned long long* g2;   // [NVmax][2]
    unsigned long long* gx;   // [2 parities][8 XCDs][NVmax][2]  (three-level form: the XCDs' partial sums)
    unsigned* ctl;            // [0] tag base (advanced by the launch), [1] error word (sticky), [2..3] spare
    int NVmax;
    int G;                    // workgroups of the grid this buffer was carved for (the parity stride of g1)
};
__device__ __forceinline__ unsigned long long* xch_g1(const XchBuf& b, unsigned tag) { return b.g1 + (size_t)(tag & 1u) * b.G * b.NVmax; }

constexpr unsigned kXchSpinLimit = 1u << 18;      // polls: far beyond any legitimate wait (a poll is ~1 us), far below the driver's watchdog
constexpr int kXchScratchDoubles = 512;           // LDS scratch of xch_reduce / xch_collect

__device__ __forceinline__ void xch_store(unsigned long long* p, unsigned tag, unsigned value) {
    __hip_atomic_store((xch_gu64*)p, ((unsigned long long)tag << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long xch_load(const unsigned long long* p) {
    return __hip_atomic_load((xch_gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned xch_base(const XchBuf& b) { return __hip_atomic_load((xch_gu32*)b.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// the LAST thing a launch does (one lane of workgroup 0, after its last exchange: every workgroup has read the base long before)
__device__ __forceinline__ void xch_advance(const XchBuf& b, unsigned base, unsigned phases) {
    __hip_atomic_store((xch_gu32*)b.ctl, base + phases + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void xch_fail(const XchBuf& b, unsigned code) {
    __hip_atomic_fetch_or((xch_gu32*)(b.ctl + 1), code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// poll ONE granule per active lane until its tag matches; the wave leaves together
__device__ __forceinline__ unsigned xch_wait(const XchBuf& b, const unsigned long long* p, unsigned tag, bool active, unsigned code) {
    unsigned v = 0;
    bool ok = !active;
    for (unsigned spins = 0;; ++spins) {
        if (!ok) {
            const unsigned long long x = xch_load(p);
            if ((unsigned)(x >> 32) == tag) { v = (unsigned)x; ok = true; }
        }
        if (__all(ok)) break;
        if (spins >= kXchSpinLimit) { if (!ok) xch_fail(b, code); break; }
        // (once one wait of the launch has run out, the others give up at their next check instead of spinning their own limit out)
        if ((spins & 1023u) == 1023u && __hip_atomic_load((xch_gu32*)(b.ctl + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        __builtin_amdgcn_s_sleep(1);
    }
    return v;
}

// hop 1: thread v < NV publishes vals[v] (LDS or registers behind a barrier: the caller has made vals visible to these threads)
__device__ __forceinline__ void xch_publish(const XchBuf& b, int wg, int NV, unsigned tag, const float* vals) {
    const int t = threadIdx.x;
    if (t < NV) xch_store(xch_g1(b, tag) + (size_t)wg * b.NVmax + t, tag, __float_as_uint(vals[t]));
}

// the reducer role of workgroup `wg` of G for the values v = wg, wg + G, ... < NV: at most 256 (value, contributor) items (NV <= 128, G <= 256), one per
// thread of a 256-thread workgroup.  Contains workgroup barriers (uniform: every thread of the workgroup calls it).
__device__ __forceinline__ void xch_reduce(const XchBuf& b, int wg, int G, int NV, unsigned tag, double* scratch) {
    const int t = threadIdx.x;
    const int mine = wg < NV ? (NV - wg + G - 1) / G : 0;           // values this workgroup owns (uniform)
    if (mine == 0) return;
    const int items = mine * G;
    const int vl = t / G, src = t - vl * G;                           // item t = (local value vl, contributor src)
    const bool act = t < items;
    const int v = wg + vl * G;
    const unsigned bits = xch_wait(b, xch_g1(b, tag) + (size_t)(act ? src : 0) * b.NVmax + (act ? v : 0), tag, act, 1u);
    scratch[t] = act ? (double)__uint_as_float(bits) : 0.0;
    __syncthreads();
    // fixed tree: chunks of 16 contributors (sixteen independent LDS reads, then the adds in order), then the chunk sums in order
    const int chunks = (G + 15) / 16;                                 // mine * chunks <= 128
    if (t < mine * chunks) {
        const int q = t / chunks, c = t - q * chunks;
        const int lo = c * 16;
        double x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = lo + i < G ? scratch[q * G + lo + i] : 0.0;
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += x[i];
        scratch[256 + t] = s;
    }
    __syncthreads();
    if (t < mine) {
        double x[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) x[c] = c < chunks ? scratch[256 + t * chunks + c] : 0.0;
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < 16; ++c) s += x[c];
        const unsigned long long u = __double_as_longlong(s);
        const int vv = wg + t * G;
        xch_store(b.g2 + (size_t)vv * 2, tag, (unsigned)u);
        xch_store(b.g2 + (size_t)vv * 2 + 1, tag, (unsigned)(u >> 32));
    }
    __syncthreads();                                                  // scratch is free again
}

// hop 2: the NV totals into out[NV] (LDS doubles); ends with a workgroup barrier.  2 NV <= 256.
__device__ __forceinline__ void xch_collect(const XchBuf& b, int NV, unsigned tag, double* out, double* scratch) {
    const int t = threadIdx.x;
    unsigned* halves = reinterpret_cast<unsigned*>(scratch);
    const bool act = t < 2 * NV;
    const unsigned bits = xch_wait(b, b.g2 + (act ? t : 0), tag, act, 2u);
    if (act) halves[t] = bits;
    __syncthreads();
    if (t < NV) out[t] = __longlong_as_double(((unsigned long long)halves[2 * t + 1] << 32) | halves[2 * t]);
    __syncthreads();
}

// ONE hop for small grids (G NV <= kXchOneHop granules = 32 KB per sweep): every workgroup sweeps all hop-1 granules itself -- thread t takes value t % NV and the
// contributors t / NV, t / NV + 256 / NV, ... (at most 16: all polls of a pass in flight together), sums them in that order in fp64, and the 256 / NV partial
// sums of a value are added in order.  Same bits in every workgroup, one memory round trip instead of two.  NV a power of two, 32 <= NV <= 128.
constexpr int kXchOneHop = 4096;
__device__ __forceinline__ void xch_sweep(const XchBuf& b, int G, int NV, unsigned tag, double* out, double* scratch) {
    const int t = threadIdx.x;
    const int groups = 256 / NV, v = t & (NV - 1), grp = t / NV;
    unsigned long long x[16];
    const unsigned long long* g1 = xch_g1(b, tag);
    bool ok = false;
    for (unsigned spins = 0; !ok; ++spins) {
        ok = true;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int src = grp + k * groups;
            x[k] = src < G ? xch_load(g1 + (size_t)src * b.NVmax + v) : ((unsigned long long)tag << 32);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) ok &= (unsigned)(x[k] >> 32) == tag;
        ok = __all(ok);
        if (!ok) {
            if (spins >= kXchSpinLimit) { xch_fail(b, 4u); break; }
            if ((spins & 1023u) == 1023u && __hip_atomic_load((xch_gu32*)(b.ctl + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
            __builtin_amdgcn_s_sleep(1);
        }
    }
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += (grp + k * groups < G) ? (double)__uint_as_float((unsigned)x[k]) : 0.0;
    scratch[t] = s;
    __syncthreads();
    if (t < NV) {
        double a = 0.0;
        for (int g = 0; g < groups; ++g) a += scratch[g * NV + t];
        out[t] = a;
    }
    __syncthreads();
}
__device__ __forceinline__ bool xch_one_hop(int G, int NV) { return G * NV <= kXchOneHop; }

// THREE levels for big grids (round 6, late): a granule that only has to reach the OTHER workgroups of the same XCD need not be written through to memory -- they
// share the L2.  Measured (tools/ubench/xcd_local.hip): a workgroup of a grid of any size lands on XCD blockIdx % 8; a round trip between two workgroups of one XCD
// costs 1 270 shader cycles with plain stores + agent-scope polls against 2 500-2 600 with write-through stores; a sweep over an XCD's 32 workgroups 2 700 against 4 450.
//   level 1  workgroup w publishes its values with PLAIN stores; ONE of the <= 32 workgroups of its XCD (w % 8) -- member (tag mod members), another one every phase --
//            sweeps the members' granules and holds the XCD's partial sums (member order)
//   level 2  ... and publishes them -- fp64 as two granules -- with write-through stores into gx[parity][xcd]
//   level 3  every workgroup polls the <= 8 XCDs' partial sums and adds them in XCD order
// Same totals in every workgroup, one fixed order.  gx is double-buffered by phase parity like g1 (a leader publishes phase p + 2 only after every member of its XCD
// has finished phase p + 1, which needed every workgroup's phase-p + 1 contribution, i.e. everybody had finished reading phase p).  The form RELIES on the
// placement rule above: a workgroup whose XCD is not blockIdx % 8 would poll an L2 its partners never write -- the bounded spins turn that into the error word, and
// stage_train.hip checks the rule once per device before it lets a plan use this form (clhip_stage_train_xcd_rule).
constexpr int kXchXcds = 8;
__device__ __forceinline__ bool xch_hier(int G, int NV) { return G * NV > kXchOneHop && G >= 64; }
__device__ __forceinline__ void xch_store_local(unsigned long long* p, unsigned tag, unsigned value) {
    __hip_atomic_store((xch_gu64*)p, ((unsigned long long)tag << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // no cache bits: stays in the XCD's L2
}
// levels 1 + 2 (contains workgroup barriers; `part` = NV LDS doubles that receive the XCD's partial sums, scratch >= 256 doubles)
template <int W = 16>      // polls in flight per trip of the sweep: 16 where registers are free (forward), 4 in the backward kernels
__device__ __forceinline__ void xch_hier_begin(const XchBuf& b, int wg, int G, int NV, unsigned tag, const float* vals, double* part, double* scratch) {
    const int t = threadIdx.x;
    unsigned long long* g1 = xch_g1(b, tag);
    if (t < NV) xch_store_local(g1 + (size_t)wg * b.NVmax + t, tag, __float_as_uint(vals[t]));
    const int x = wg & (kXchXcds - 1), members = (G - x + kXchXcds - 1) / kXchXcds;      // <= 32
    // only ONE member of the XCD sums it, and a different one every phase: a workgroup that sweeps waits for its slowest partner before it gets to the work the caller
    // puts between begin and end (the weight gradient of the backward) -- with every member sweeping, the exchange was a barrier in front of that work (the backward
    // launches ran 8-15 % longer); the others publish and move on, like the workgroups of the two-hop form that own no value
    if ((wg >> 3) != (int)(tag % (unsigned)members)) return;      // (uniform per workgroup)
    const int groups = 256 / NV, v = t & (NV - 1), grp = t / NV;
    // (W polls in flight per trip, 16 / W trips: sixteen at once -- as xch_sweep does -- cost the backward kernels 32 more live registers at their tightest
    //  point: 70 -> 119 spilled in the 16-channel one; four trips of four cost the 64-channel forward, whose sweep has all sixteen contributors, 9 us per launch)
    double s = 0.0;
#pragma unroll 1
    for (int k0 = 0; k0 < 16; k0 += W) {
        if (k0 * groups >= members) break;                              // (uniform)
        unsigned long long q[W];
#pragma unroll
        for (int k = 0; k < W; ++k) q[k] = 0ull;
        bool ok = t >= 256;
        for (unsigned spins = 0; !ok; ++spins) {
            ok = true;
#pragma unroll
            for (int k = 0; k < W; ++k) {
                const int src = grp + (k0 + k) * groups;
                q[k] = src < members ? xch_load(g1 + (size_t)(x + kXchXcds * src) * b.NVmax + v) : ((unsigned long long)tag << 32);
            }
#pragma unroll
            for (int k = 0; k < W; ++k) ok &= (unsigned)(q[k] >> 32) == tag;
            ok = __all(ok);
            if (!ok) {
                if (spins >= kXchSpinLimit) { xch_fail(b, 8u); break; }
                if ((spins & 1023u) == 1023u && __hip_atomic_load((xch_gu32*)(b.ctl + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                __builtin_amdgcn_s_sleep(1);
            }
        }
#pragma unroll
        for (int k = 0; k < W; ++k) s += (grp + (k0 + k) * groups < members) ? (double)__uint_as_float((unsigned)q[k]) : 0.0;
    }
    if (t < 256) scratch[t] = s;
    __syncthreads();
    if (t < NV) {
        double a = 0.0;
        for (int g = 0; g < groups; ++g) a += scratch[g * NV + t];
        part[t] = a;
    }
    __syncthreads();
    if (t < 2 * NV) {
        const unsigned long long u = __double_as_longlong(part[t >> 1]);
        xch_store(b.gx + (((size_t)(tag & 1u) * kXchXcds + x) * b.NVmax + (t >> 1)) * 2 + (t & 1), tag, (t & 1) ? (unsigned)(u >> 32) : (unsigned)u);
    }
}
// level 3: the totals into out[NV] (LDS doubles; may be the `part` of xch_hier_begin); ends with a workgroup barrier
__device__ __forceinline__ void xch_hier_end(const XchBuf& b, int G, int NV, unsigned tag, double* out, double* scratch) {
    const int t = threadIdx.x;
    const int nx = G < kXchXcds ? G : kXchXcds;
    const int groups = 256 / NV, v = t & (NV - 1), grp = t / NV, per = kXchXcds / groups;      // NV = 32 / 64 / 128: 1 / 2 / 4 XCDs per thread
    const unsigned long long* gx = b.gx + (size_t)(tag & 1u) * kXchXcds * b.NVmax * 2;
    unsigned long long q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = 0ull;
    bool ok = t >= 256;
    for (unsigned spins = 0; !ok; ++spins) {
        ok = true;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int xi = grp * per + (k >> 1);
            q[k] = ((k >> 1) < per && xi < nx) ? xch_load(gx + ((size_t)xi * b.NVmax + v) * 2 + (k & 1)) : ((unsigned long long)tag << 32);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) ok &= (unsigned)(q[k] >> 32) == tag;
        ok = __all(ok);
        if (!ok) {
            if (spins >= kXchSpinLimit) { xch_fail(b, 16u); break; }
            if ((spins & 1023u) == 1023u && __hip_atomic_load((xch_gu32*)(b.ctl + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
            __builtin_amdgcn_s_sleep(1);
        }
    }
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < per && grp * per + k < nx) s += __longlong_as_double(((unsigned long long)(unsigned)q[2 * k + 1] << 32) | (unsigned)q[2 * k]);
    if (t < 256) scratch[t] = s;
    __syncthreads();
    if (t < NV) {
        double a = 0.0;
        for (int g = 0; g < groups; ++g) a += scratch[g * NV + t];
        out[t] = a;
    }
    __syncthreads();
}

// the whole exchange in two calls, so that work which does not need the totals can sit between them:
//   xch_begin  publishes this workgroup's values and (two-hop form) plays its reducer role
//   xch_end    leaves the totals in out[NV]
template <int W = 16>
__device__ __forceinline__ void xch_begin(const XchBuf& b, int wg, int G, int NV, unsigned tag, const float* vals, double* scratch) {
    if (b.gx != nullptr && xch_hier(G, NV)) { xch_hier_begin<W>(b, wg, G, NV, tag, vals, scratch + 256, scratch); return; }
    xch_publish(b, wg, NV, tag, vals);
    if (!xch_one_hop(G, NV)) xch_reduce(b, wg, G, NV, tag, scratch);
}
__device__ __forceinline__ void xch_end(const XchBuf& b, int G, int NV, unsigned tag, double* out, double* scratch) {
    if (b.gx != nullptr && xch_hier(G, NV)) xch_hier_end(b, G, NV, tag, out, scratch);
    else if (xch_one_hop(G, NV)) xch_sweep(b, G, NV, tag, out, scratch);
    else xch_collect(b, NV, tag, out, scratch);
}

// bytes of the three arrays for a grid of up to G workgroups and NVmax values, and their carving from one zeroed allocation
static inline size_t xch_bytes(int G, int NVmax) { return 256 + 2 * (size_t)G * NVmax * 8 + (size_t)NVmax * 16 + 2 * (size_t)kXchXcds * NVmax * 16; }
static inline XchBuf xch_carve(void* base, int G, int NVmax) {
    XchBuf b;
    char* p = static_cast<char*>(base);
    b.ctl = reinterpret_cast<unsigned*>(p);
    b.g1 = reinterpret_cast<unsigned long long*>(p + 256);
    b.g2 = reinterpret_cast<unsigned long long*>(p + 256 + 2 * (size_t)G * NVmax * 8);
    b.gx = reinterpret_cast<unsigned long long*>(p + 256 + 2 * (size_t)G * NVmax * 8 + (size_t)NVmax * 16);      // (nullptr: the three-level form is off)
    b.NVmax = NVmax; b.G = G;
    return b;
}
