// Workspace sizing and the MFMA peak probes used by bench.py to anchor the roofline.
#include "mfma32.h"
#include "mfma64.h"

namespace slk {

__global__ __launch_bounds__(256) void k_probe_f64(double *sink, int iters) {
    double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, c3, 0, 0, 0);
    }
    const double4_t s = c0 + c1 + c2 + c3;
    if (s[0] + s[1] + s[2] + s[3] == 12345.678) sink[0] = s[0];  // keep the chain alive
}

// Same probe with 8 or 16 independent accumulators per wave and distinct operands per MFMA.
template <int NACC>
__global__ __launch_bounds__(256) void k_probe_f64_n(double *sink, int iters) {
    double4_t c[NACC];
    double a[NACC], b[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        c[i] = (double4_t){0, 0, 0, 0};
        a[i] = 1.0 + (threadIdx.x + i) * 1e-3;
        b[i] = 1.0 - (threadIdx.x + 3 * i) * 1e-3;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[(i + 1) % NACC], c[i], 0, 0, 0);
    }
    double4_t s = c[0];
#pragma unroll
    for (int i = 1; i < NACC; ++i) s += c[i];
    if (s[0] + s[1] + s[2] + s[3] == 12345.678) sink[0] = s[0];
}

__global__ __launch_bounds__(256) void k_probe_f32(float *sink, int iters) {
    float16_t c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) c0[r] = c1[r] = c2[r] = c3[r] = 0.0f;
    const float a = 1.0f + threadIdx.x * 1e-3f, b = 1.0f - threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, c3, 0, 0, 0);
    }
    const float16_t s = c0 + c1 + c2 + c3;
    float tot = 0.0f;
    for (int r = 0; r < 16; ++r) tot += s[r];
    if (tot == 12345.678f) sink[0] = tot;
}

// Latency / issue-rate probe for the scalar-per-lane float64 pipeline (the leaf chain and the
// panel factorisation are dependency chains on it).  One wave; out[0] = shader cycles
// (s_memtime), out[1] = 100 MHz ticks (s_memrealtime), out[2] = checksum.
__global__ __launch_bounds__(64) void k_probe_chain(double *out, int iters, int mode) {
    double x0 = 1.0 + threadIdx.x * 1e-9, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const double a = 1.0000001, b = 1e-9;
    float f = 1.0f + threadIdx.x * 1e-6f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#define CHAIN(BODY)                                  \
    for (int i = 0; i < iters; i += 8) {             \
        BODY BODY BODY BODY BODY BODY BODY BODY      \
    }
    if (mode == 0) {
        CHAIN(x0 = __builtin_fma(x0, a, b);)
    } else if (mode == 1) {
        CHAIN(x0 = __builtin_fma(x0, a, b); x1 = __builtin_fma(x1, a, b); x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b);
              x4 = __builtin_fma(x4, a, b); x5 = __builtin_fma(x5, a, b); x6 = __builtin_fma(x6, a, b); x7 = __builtin_fma(x7, a, b);)
    } else if (mode == 2) {
        CHAIN(f = __builtin_fmaf(f, 1.0000001f, 1e-9f);)
    } else if (mode == 3) {
        CHAIN(x0 = __builtin_amdgcn_rsq(x0) + 1.0;)
    } else if (mode == 4) {
        CHAIN(x0 = 1.0 / x0 + 0.5;)
    } else if (mode == 5) {
        CHAIN(f = 1.0f / f + 0.5f;)
    } else if (mode == 6) {
        CHAIN(x0 = (double)(float)x0 * a;)
    } else if (mode == 7) {
        CHAIN(x0 = __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(__double_as_longlong(x0) >> 32), 5) << 32) | 7) * a;)
    } else {
        CHAIN(f = rintf(f * 1.0000001f) + 0.25f;)
    }
#undef CHAIN
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[0] = (double)(t1 - t0);
        out[1] = (double)(r1 - r0);
        out[2] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + f;
    }
}

}  // namespace slk

using namespace slk;

extern "C" {

int slk_probe_chain(double *out, int iters, int mode, slk_stream_t stream) {
    SLK_REQUIRE(out && iters > 0, "bad arguments");
    hipStream_t s = as_stream(stream);
    SLK_RUN("probe_chain", 0, 0, s, k_probe_chain<<<1, 64, 0, s>>>(out, iters, mode));
    return SLK_OK;
}

size_t slk_workspace_bytes(int R, int n) {
    if (R < 0 || n <= 0) return 0;
    const size_t ld = (size_t)slk_factor_ld(n);
    const size_t rn = (size_t)(R > 0 ? R : 1) * (size_t)n;
    size_t factor = 2 * ld * ld * sizeof(double);                      // X and S of slk_chol_inverse_upper
    size_t loop = 2 * rn * sizeof(float) + (size_t)n * sizeof(int);    // permuted Q and E, inverse order
    size_t search = rn * sizeof(float) + (size_t)(R + n) * sizeof(float) + (size_t)R * ((n + 127) / 128) * sizeof(float) + 4096;
    // layer error on the bfloat16 MFMA: three 2-byte planes of W - Q and of H, the per-tile partial sums
    size_t error = 6 * (rn + 128 * (size_t)n) + 6 * (size_t)n * n + 4 * (size_t)R * ((n + 127) / 128) * sizeof(float) + 8192;
    size_t prep = 64 * sizeof(float) + (size_t)n * (sizeof(double) + 1);
    search += error;  // the local search calls the layer error (for G = (W - Q) H) behind its own scratch
    size_t m = factor;
    if (loop > m) m = loop;
    if (search > m) m = search;
    if (error > m) m = error;
    if (prep > m) m = prep;
    return m + (1u << 16);
}

size_t slk_workspace_bytes_batch(int batch, int rows_per_layer, int n) {
    if (batch < 1 || batch > 64 || rows_per_layer < 0 || n <= 0 || (long long)batch * rows_per_layer > 0x7fffffffLL) return 0;
    // the stacked rows as one layer, plus the operand planes of the other Hessians and the other inverse orders
    return slk_workspace_bytes(batch * rows_per_layer, n) + (size_t)(batch - 1) * 6 * (size_t)n * n +
           (size_t)batch * n * (sizeof(int) + sizeof(float)) + 4096;  // (+ every layer's diagonal for a batched local search)
}

int slk_probe_mfma_f64(double *sink, int blocks, int iters, slk_stream_t stream) {
    SLK_REQUIRE(sink && blocks > 0 && iters > 0, "bad arguments");
    hipStream_t s = as_stream(stream);
    SLK_RUN("probe_mfma_f64", 4.0 * 2048 * 4 * iters * blocks, 0, s, k_probe_f64<<<blocks, 256, 0, s>>>(sink, iters));
    return SLK_OK;
}

int slk_probe_mfma_f64_acc(double *sink, int blocks, int iters, int nacc, slk_stream_t stream) {
    SLK_REQUIRE(sink && blocks > 0 && iters > 0 && (nacc == 1 || nacc == 2 || nacc == 4 || nacc == 8 || nacc == 16), "bad arguments");
    hipStream_t s = as_stream(stream);
    if (nacc == 1)
        SLK_RUN("probe_mfma_f64", 1.0 * 2048 * 4 * iters * blocks, 0, s, k_probe_f64_n<1><<<blocks, 256, 0, s>>>(sink, iters));
    else if (nacc == 2)
        SLK_RUN("probe_mfma_f64", 2.0 * 2048 * 4 * iters * blocks, 0, s, k_probe_f64_n<2><<<blocks, 256, 0, s>>>(sink, iters));
    else if (nacc == 4)
        SLK_RUN("probe_mfma_f64", 4.0 * 2048 * 4 * iters * blocks, 0, s, k_probe_f64_n<4><<<blocks, 256, 0, s>>>(sink, iters));
    else if (nacc == 8)
        SLK_RUN("probe_mfma_f64", 8.0 * 2048 * 4 * iters * blocks, 0, s, k_probe_f64_n<8><<<blocks, 256, 0, s>>>(sink, iters));
    else
        SLK_RUN("probe_mfma_f64", 16.0 * 2048 * 4 * iters * blocks, 0, s, k_probe_f64_n<16><<<blocks, 256, 0, s>>>(sink, iters));
    return SLK_OK;
}

int slk_probe_mfma_f32(float *sink, int blocks, int iters, slk_stream_t stream) {
    SLK_REQUIRE(sink && blocks > 0 && iters > 0, "bad arguments");
    hipStream_t s = as_stream(stream);
    SLK_RUN("probe_mfma_f32", 4.0 * 4096 * 4 * iters * blocks, 0, s, k_probe_f32<<<blocks, 256, 0, s>>>(sink, iters));
    return SLK_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ options, LDS opt-in
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include <strings.h>

namespace slk {
namespace {
const char *const kOptNames[OPT_COUNT] = {
    "NO_FAST_LEAF", "NO_DEFER", "NO_WINDOW2", "WIN_DBG", "NO_REGULAR_SEARCH", "NO_FAST_SEARCH_DIV",
    "NO_ERROR_SPLITK", "ERROR_CB", "NO_SYM_ERROR", "NO_BF16_ERROR", "NO_BF16_DMA", "NO_BF16_HESSIAN",
    "ERROR_F32_BELOW", "NO_BF16_ASYM", "NO_SYM_AVERAGE", "NO_WAVE_SEARCH", "LOOKAHEAD", "WINDOW_ROWS", "PANEL_SPLIT", "TALL_ERROR", "ROWS_BELOW_WIDE",
};
std::atomic<int> g_opts[OPT_COUNT];
std::once_flag g_opts_once;
void opts_from_env() {
    for (int i = 0; i < OPT_COUNT; ++i) {
        const std::string name = std::string("SLK_") + kOptNames[i];
        const char *v = getenv(name.c_str());
        int val = 0;
        if (v) {
            val = atoi(v);
            if (val == 0 && v[0] != '0') val = 1;  // set without a number: on
        }
        g_opts[i].store(val, std::memory_order_relaxed);
    }
}
int opt_index(const char *name) {
    if (!name) return -1;
    if (strncasecmp(name, "SLK_", 4) == 0) name += 4;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (strcasecmp(name, kOptNames[i]) == 0) return i;
    return -1;
}
std::mutex g_lds_mu;
std::map<std::pair<const void *, int>, size_t> g_lds_set;
}  // namespace

int opt(Opt o) {
    std::call_once(g_opts_once, opts_from_env);
    return g_opts[o].load(std::memory_order_relaxed);
}

namespace {
std::mutex g_helper_mu;
struct HelperRec {
    Helper h;
    int made;  // events created so far
};
std::map<std::pair<int, hipStream_t>, HelperRec> g_helpers;
}  // namespace

hipError_t helper_for(hipStream_t main, int need, Helper *out) {
    if (need > HELPER_EVENTS) return hipErrorInvalidValue;
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(g_helper_mu);
    auto it = g_helpers.find({device, main});
    if (it == g_helpers.end()) {
        HelperRec r{};
        e = hipStreamCreateWithFlags(&r.h.stream, hipStreamNonBlocking);
        if (e != hipSuccess) return e;
        e = hipStreamCreateWithFlags(&r.h.stream2, hipStreamNonBlocking);
        if (e != hipSuccess) {
            (void)hipStreamDestroy(r.h.stream);
            return e;
        }
        r.h.events = new hipEvent_t[HELPER_EVENTS]();
        r.made = 0;
        it = g_helpers.emplace(std::make_pair(device, main), r).first;
    }
    HelperRec &r = it->second;
    const int had = r.made;
    while (r.made < need) {
        e = hipEventCreateWithFlags(&r.h.events[r.made], hipEventDisableTiming);
        if (e != hipSuccess) {  // give back what this call made; what earlier calls made stays usable
            while (r.made > had) (void)hipEventDestroy(r.h.events[--r.made]);
            if (had == 0) {
                (void)hipStreamDestroy(r.h.stream);
                (void)hipStreamDestroy(r.h.stream2);
                delete[] r.h.events;
                g_helpers.erase(it);
            }
            return e;
        }
        ++r.made;
    }
    *out = r.h;
    return hipSuccess;
}

static hipError_t release_helpers() {
    std::lock_guard<std::mutex> lock(g_helper_mu);
    hipError_t first = hipSuccess;
    for (auto &kv : g_helpers) {
        HelperRec &r = kv.second;
        // The caller's stream (the map key) still waits on this record's join events when a look-ahead factorisation is in
        // flight: both streams are drained before anything is destroyed.  (Contract, include/sleekit_amd.h: no factorisation
        // may be ENQUEUED concurrently with this call -- helper_for hands out a copy of the record that this call invalidates.)
        int now = 0;
        (void)hipGetDevice(&now);
        if (now != kv.first.first) (void)hipSetDevice(kv.first.first);
        hipError_t e = hipStreamSynchronize(kv.first.second);
        if (e != hipSuccess && first == hipSuccess) first = e;
        e = hipStreamSynchronize(r.h.stream);
        if (e != hipSuccess && first == hipSuccess) first = e;
        e = hipStreamSynchronize(r.h.stream2);
        if (e != hipSuccess && first == hipSuccess) first = e;
        if (now != kv.first.first) (void)hipSetDevice(now);
        for (int i = 0; i < r.made; ++i) (void)hipEventDestroy(r.h.events[i]);
        (void)hipStreamDestroy(r.h.stream);
        (void)hipStreamDestroy(r.h.stream2);
        delete[] r.h.events;
    }
    g_helpers.clear();
    return first;
}

hipError_t lds_opt_in(const void *kernel, size_t bytes) {
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(g_lds_mu);
    size_t &have = g_lds_set[{kernel, device}];
    if (have >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) have = bytes;
    return e;
}
}  // namespace slk

extern "C" {
int slk_release_helpers(void) {
    SLK_HIP(slk::release_helpers());
    return SLK_OK;
}
int slk_set_option(const char *name, int value) {
    const int i = opt_index(name);
    SLK_REQUIRE(i >= 0, "unknown option %s", name ? name : "(null)");
    (void)opt((Opt)i);  // environment first, so that it cannot overwrite this later
    g_opts[i].store(value, std::memory_order_relaxed);
    return SLK_OK;
}
int slk_get_option(const char *name) {
    const int i = opt_index(name);
    return i < 0 ? 0 : opt((Opt)i);
}
}

// ------------------------------------------------------------------ per-launch profiler

namespace slk {
namespace {
struct ProfEntry {
    const char *name;
    double flops, bytes, width;  // width: share of the chip's 256 CUs the launch can occupy
    hipEvent_t start, stop;
};
struct ProfState {
    bool on = false;
    std::vector<ProfEntry> entries;
    std::vector<hipEvent_t> pool;
    std::mutex mu;
} g_prof;

hipEvent_t prof_event() {
    if (!g_prof.pool.empty()) {
        hipEvent_t e = g_prof.pool.back();
        g_prof.pool.pop_back();
        return e;
    }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}  // namespace

static thread_local long g_next_width = 0;
void prof_next_width(long workgroups) { g_next_width = workgroups; }

ProfScope::ProfScope(const char *name, double flops, double bytes, hipStream_t s) : slot(-1), stream(s) {
    const long wgs = g_next_width;
    g_next_width = 0;
    if (!g_prof.on) return;
    std::lock_guard<std::mutex> lock(g_prof.mu);
    ProfEntry e{name, flops, bytes, wgs > 0 && wgs < 256 ? wgs / 256.0 : 1.0, prof_event(), prof_event()};
    if (!e.start || !e.stop) return;
    (void)hipEventRecord(e.start, s);
    g_prof.entries.push_back(e);
    slot = (int)g_prof.entries.size() - 1;
}

ProfScope::~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lock(g_prof.mu);
    (void)hipEventRecord(g_prof.entries[slot].stop, stream);
}
}  // namespace slk

extern "C" {

int slk_profile_enable(int on) {
    std::lock_guard<std::mutex> lock(g_prof.mu);
    g_prof.on = on != 0;
    return SLK_OK;
}

int slk_profile_reset(void) {
    std::lock_guard<std::mutex> lock(g_prof.mu);
    for (auto &e : g_prof.entries) {
        g_prof.pool.push_back(e.start);
        g_prof.pool.push_back(e.stop);
    }
    g_prof.entries.clear();
    return SLK_OK;
}

// Waits for the recorded launches and writes a JSON array, one object per kernel name:
//   {"kernel": ..., "launches": n, "total_ms": t, "flops": sum, "bytes": sum}
// Returns the number of characters needed (like snprintf).
int slk_profile_report(char *buf, size_t cap) {
    std::lock_guard<std::mutex> lock(g_prof.mu);
    struct Agg {
        long n = 0;
        double ms = 0, flops = 0, bytes = 0, cu_ms = 0;
    };
    std::map<std::string, Agg> agg;
    std::vector<std::string> order;
    for (auto &e : g_prof.entries) {
        if (hipEventSynchronize(e.stop) != hipSuccess) continue;
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, e.start, e.stop) != hipSuccess) continue;
        if (!agg.count(e.name)) order.push_back(e.name);
        Agg &a = agg[e.name];
        a.n += 1;
        a.ms += ms;
        a.cu_ms += ms * e.width;
        a.flops += e.flops;
        a.bytes += e.bytes;
    }
    std::string out = "[";
    char line[512];
    for (size_t i = 0; i < order.size(); ++i) {
        const Agg &a = agg[order[i]];
        snprintf(line, sizeof(line), "%s{\"kernel\": \"%s\", \"launches\": %ld, \"total_ms\": %.6f, \"chip_ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e}",
                 i ? ", " : "", order[i].c_str(), a.n, a.ms, a.cu_ms, a.flops, a.bytes);
        out += line;
    }
    out += "]";
    if (buf && cap > 0) {
        snprintf(buf, cap, "%s", out.c_str());
    }
    return (int)out.size();
}

}  // extern "C"
