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

// Uniform codebook in float32, formed exactly like the reference
// (sleekit/codebook.py:35-41 under NEP 50: the Python-float step is cast to float32).
struct Grid {
    float zero, step, top;  // top = levels - 1
};

static inline Grid make_grid(int levels, double lo, double hi) {
    Grid g;
    g.zero = (float)lo;
    g.step = (float)((hi - lo) / (double)(levels - 1));
    g.top = (float)(levels - 1);
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

// Broadcast lane `src` (wave-uniform index) of a double to the whole wave through SGPRs.
__device__ __forceinline__ double readlane_f64(double v, int src) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffLL), src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

}  // namespace slk
