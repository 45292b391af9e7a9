// Shared host/device helpers for the gfx950 kernels.  Wave = 64 lanes everywhere.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/sleekit_amd.h"

namespace slk {

void set_error(const char *fmt, ...);

#define SLK_REQUIRE(cond, ...)          \
    do {                                \
        if (!(cond)) {                  \
            slk::set_error(__VA_ARGS__); \
            return SLK_E_ARG;           \
        }                               \
    } while (0)

#define SLK_HIP(call)                                                                  \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            slk::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return SLK_E_HIP;                                                          \
        }                                                                              \
    } while (0)

#define SLK_LAUNCH_CHECK() SLK_HIP(hipGetLastError())

// Optional per-launch timing (slk_profile_*): when enabled every kernel launch is bracketed
// by a pair of HIP events on its own stream, tagged with its name and ALGORITHMIC flops and
// bytes (the roofline numerators of DESIGN.md).  Disabled: one predictable branch.
struct ProfScope {
    int slot;
    hipStream_t stream;
    ProfScope(const char *name, double flops, double bytes, hipStream_t s);
    ~ProfScope();
};

// Workgroups of the NEXT profiled launch, for launches that cannot fill the chip (a latency chain on
// one workgroup is not "dominant" because it is long): the report weighs a launch's time by
// min(1, workgroups / 256).  0 = not stated, taken as chip-wide.
void prof_next_width(long workgroups);

// SLK_RUN("kernel name", flops, bytes, stream, kernel<<<grid, block, smem, stream>>>(args...));
#define SLK_RUN(name, flops, bytes, stream, ...)                 \
    do {                                                         \
        {                                                        \
            slk::ProfScope prof_scope_(name, flops, bytes, stream); \
            __VA_ARGS__;                                         \
        }                                                        \
        SLK_LAUNCH_CHECK();                                      \
    } while (0)

// SLK_RUN for a launch of `wgs` workgroups (see prof_next_width)
#define SLK_RUN_W(name, flops, bytes, wgs, stream, ...) \
    do {                                                \
        slk::prof_next_width((long)(wgs));              \
        SLK_RUN(name, flops, bytes, stream, __VA_ARGS__); \
    } while (0)

// Run-time switches (slk_set_option / slk_get_option; initial values read ONCE from the environment, SLK_<NAME>).
// They select between code paths that give the same results (tests hold them to that) or shape measurements.
enum Opt {
    OPT_NO_FAST_LEAF,          // leaves: true divides instead of the exact-division fma sequence
    OPT_NO_DEFER,              // window kernel: no deferred helper updates
    OPT_NO_WINDOW2,            // general window kernel for every window
    OPT_WIN_DBG,               // window kernels: measurement bits (8: cycle counters of workgroup 0, 16: U blocks from global)
    OPT_NO_REGULAR_SEARCH,     // scale search: staged row sums for every row length
    OPT_NO_FAST_SEARCH_DIV,    // scale search: true divides
    OPT_NO_ERROR_SPLITK,       // layer error: never cut K into chunks
    OPT_ERROR_CB,              // layer error: force this K-chunk count
    OPT_NO_SYM_ERROR,          // layer error: ignore the symmetry of H
    OPT_NO_BF16_ERROR,         // layer error: float32 MFMA kernel
    OPT_NO_BF16_DMA,           // bfloat16 x 3 GEMMs: stage operands through registers
    OPT_NO_BF16_HESSIAN,       // Hessian accumulation: float32 MFMA kernel
    OPT_ERROR_F32_BELOW,       // layer error of a batch: float32 kernel when a layer has fewer rows than this (0: never, the default)
    OPT_NO_BF16_ASYM,          // layer error: an H that is not symmetric goes to the float32 kernel
    OPT_NO_SYM_AVERAGE,        // layer error: an H that is not symmetric is NOT averaged with its transpose (planes of H^T, every k, instead)
    OPT_NO_WAVE_SEARCH,        // local search: the workgroup-per-row kernel for every row length
    OPT_LOOKAHEAD,             // factorisation: the bulk of an outer syrk on a helper stream, beside the next block's panels
    OPT_WINDOW_ROWS,           // window kernel: rows per workgroup forced to 16 or 32 (0: 32, or 16 where the caller asks for latency)
    OPT_PANEL_SPLIT,           // factorisation: 0 / 3: an outer block's panels as a chain of workgroups in one launch; 1 / 2: the panel kernel, two launches / one per panel
    OPT_TALL_ERROR,            // layer error of whole layers (from 2048 rows up): 1 = 256 x 128 tiles, 2 = 256 x 256 tiles over K16 (default: 128 x 128 everywhere)
    OPT_ROWS_BELOW_WIDE,       // factorisation: the rows below a diagonal block 64 rows per workgroup (1) or 16 (2) whatever the batch (0: 64 from 1024 strips per launch up)
    OPT_COUNT
};
int opt(Opt o);

// A helper stream beside the caller's (one per (device, caller stream), created at first use, with a pool of events):
// the factorisation forks the bulk of an outer syrk onto it and joins again before it returns, so the caller still sees
// one stream.  Fork and join are event record / wait pairs, which a hipGraph capture of the caller's stream follows.
struct Helper {
    hipStream_t stream;
    hipStream_t stream2;  // a second one: what the look-ahead runs beside the rest of an outer update (nodes of the inverse)
    hipEvent_t *events;   // room for HELPER_EVENTS; the first `need` of helper_for exist
};
constexpr int HELPER_EVENTS = 512;
// the helper of (current device, main) with at least `need` events (created as they are first asked for; on failure
// nothing made by the call is kept).  slk_release_helpers() destroys them all.
hipError_t helper_for(hipStream_t main, int need, Helper *out);

// Opt a kernel in to `bytes` of dynamic LDS (above the 64 KB default) on the CURRENT device; remembered per
// (kernel, device), safe to call from several threads.  Returns a hipError_t.
hipError_t lds_opt_in(const void *kernel, size_t bytes);
#define SLK_LDS_OPT_IN(kernel, bytes) SLK_HIP(slk::lds_opt_in(reinterpret_cast<const void *>(kernel), (bytes)))

// The float32 inputs of a batch of layers (caller-owned tensors: no common base), passed by value to the kernels
// that read them; blockIdx.z is the layer in every batched launch.
struct PtrTable {
    const float *p[64];
};

// the layer error of a stack of `batch` layers (rows [b rpl, (b + 1) rpl) against Hs[b]) WITH the products G = (Q - W) H
// the local search starts from (sgemm.hip; the public entries are slk_row_errors / slk_row_errors_batch)
int row_errors_products(const float *W, const float *Q, const float *const *Hs, int batch, int rpl, int n, const int *sym_known,
                        float *row_err, float *G, void *workspace, size_t ws_bytes, slk_stream_t stream);

static inline hipStream_t as_stream(slk_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Bump allocator over the caller's workspace.
struct Arena {
    char *base;
    size_t size, used;
    Arena(void *p, size_t bytes) : base(static_cast<char *>(p)), size(bytes), used(0) {}
    template <class T>
    T *take(size_t count) {
        size_t off = align_up(used, 256);
        size_t end = off + count * sizeof(T);
        if (end > size || base == nullptr) return nullptr;
        used = end;
        return reinterpret_cast<T *>(base + off);
    }
};

// Codebook in float32.  Uniform (table == nullptr): formed exactly like the reference
// (sleekit/codebook.py:35-41 under NEP 50: the Python-float step is cast to float32).  General
// (sleekit/codebook.py:98-190): `table` points to n sorted values followed by the n - 1 bin limits;
// a value's index is np.digitize(x, limits) = the number of limits <= x.
struct Grid {
    float zero, step, top;  // top = levels - 1
    const float *table;
    int n;
};

static inline Grid make_grid(int levels, double lo, double hi, const float *table = nullptr) {
    Grid g;
    g.zero = (float)lo;
    g.step = (float)((hi - lo) / (double)(levels - 1));
    g.top = (float)(levels - 1);
    g.table = table;
    g.n = levels;
    return g;
}

// t = clip(rint((x - zero) / step + shift), first, last): codebook.py:47-49, 71-74, 83-86.
// No contraction: the library is built with -ffp-contract=off.
__device__ __forceinline__ float grid_pos(float x, const Grid g, float shift, float first, float last) {
    float t = (x - g.zero) / g.step;
    t = t + shift;  // exact no-op for shift == 0
    t = rintf(t);
    return fminf(fmaxf(t, first), last);
}
__device__ __forceinline__ float grid_val(float t, const Grid g) { return t * g.step + g.zero; }
__device__ __forceinline__ float grid_value(float x, const Grid g) {
    return grid_val(grid_pos(x, g, 0.0f, 0.0f, g.top), g);
}

// np.digitize(x, limits) for increasing limits: how many of them are <= x (binary search).
__device__ __forceinline__ int table_index(float x, const Grid g) {
    const float *lim = g.table + g.n;
    int lo = 0, hi = g.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (lim[mid] <= x) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// The four maps of either kind of codebook (codebook.py:43-95 and 150-190).
__device__ __forceinline__ float cb_value(float x, const Grid g) { return g.table ? g.table[table_index(x, g)] : grid_value(x, g); }
__device__ __forceinline__ int cb_index(float x, const Grid g) {
    return g.table ? table_index(x, g) : (int)grid_pos(x, g, 0.0f, 0.0f, g.top);
}
__device__ __forceinline__ float cb_up(float x, const Grid g) {
    return g.table ? g.table[min(table_index(x, g) + 1, g.n - 1)] : grid_val(grid_pos(x, g, 1.0f, 1.0f, g.top), g);
}
__device__ __forceinline__ float cb_down(float x, const Grid g) {
    return g.table ? g.table[max(table_index(x, g) - 1, 0)] : grid_val(grid_pos(x, g, -1.0f, 0.0f, g.top - 1.0f), g);
}

// Broadcast lane `src` (wave-uniform index) of a double to the whole wave through SGPRs.
__device__ __forceinline__ double readlane_f64(double v, int src) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffLL), src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Zero-fill and copy as plain kernels rather than hipMemsetAsync / hipMemcpyAsync: inside a captured hipGraph the
// runtime's memset node did not take effect on replay (ROCm 7.2: a counting sort's rank buffer kept the previous
// replay's counts and the order it produced indexed out of bounds); kernel nodes replay faithfully.
__global__ __launch_bounds__(256) static void k_zero_words(unsigned *__restrict__ p, size_t words) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
__global__ __launch_bounds__(256) static void k_copy_words(unsigned *__restrict__ dst, const unsigned *__restrict__ src, size_t words) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// (sizes are multiples of 4 bytes everywhere they are used)
static inline void zero_async(void *p, size_t bytes, hipStream_t s) {
    const size_t words = bytes / 4;
    const int blocks = (int)((words + 1023) / 1024 < 2048 ? (words + 1023) / 1024 : 2048);
    if (words) k_zero_words<<<blocks ? blocks : 1, 256, 0, s>>>(static_cast<unsigned *>(p), words);
}
static inline void copy_async(void *dst, const void *src, size_t bytes, hipStream_t s) {
    const size_t words = bytes / 4;
    const int blocks = (int)((words + 1023) / 1024 < 2048 ? (words + 1023) / 1024 : 2048);
    if (words) k_copy_words<<<blocks ? blocks : 1, 256, 0, s>>>(static_cast<unsigned *>(dst), static_cast<const unsigned *>(src), words);
}

}  // namespace slk
