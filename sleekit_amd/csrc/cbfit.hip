// Fitting a general codebook to data (Lloyd-Max): the statistics every round of the reference's training code needs
// (sleekit/codebook.py:190-267, 322-367) in ONE pass over the data -- per-bin counts, per-bin sums (-> centroids) and
// the total squared miss (-> mse) -- plus the sort / distinct-values steps of its two initialisations.
//
// HBM-bound byte work: 4 B per element, read once.  Results do not depend on the order in which workgroups or lanes
// run, nor on the order of the data (counts and sums): counts are integers; the per-bin sums are accumulated in 64-bit
// FIXED POINT (integer adds commute), each workgroup at the scale of the largest |x| it has seen -- chosen so that
// `count` elements cannot overflow -- and brought to the common scale of the whole array when the workgroups'
// partials are gathered (one rounding per partial: resolution max|x| * 2^-(62 - ceil(log2 count)), i.e. 2^-38 of the
// largest element for 16M elements; the reference's float32 pairwise sums carry 2^-24); the squared miss is a
// float64 sum over a fixed tree (per thread, per workgroup, then over the workgroups in index order).
// No global atomics: thousands of device-scope atomics on a few addresses cost more than the pass over the data
// (measured: 103 us for a max|x| pass with one atomicMax per wave, against 22 us for the whole statistics kernel).
#include "common.h"

#include <hipcub/hipcub.hpp>

namespace slk {

constexpr int FIT_THREADS = 256;
constexpr int FIT_MAX_BLOCKS = 2048;
constexpr int FIT_MAX_LEVELS = 256;
constexpr int FIT_ROUND = FIT_THREADS * 4 * 8;  // elements a workgroup holds in registers at a time (8 float4 per thread)
constexpr int FIT_GATHER = 64;                  // workgroups of the gather stage
constexpr int FIT_NO_SHIFT = 1 << 30;           // "has seen only zeros so far"

// Workgroup b owns the CONTIGUOUS elements [b * chunk, (b + 1) * chunk), chunk a multiple of FIT_ROUND: sorted data
// (what lloyd_max passes) then gives every thread runs of one bin, which it sums in registers.
static inline int fit_blocks(size_t count) {
    const size_t b = (count + FIT_ROUND - 1) / FIT_ROUND;  // one round each up to 16M elements (8 workgroups per CU: measured faster
                                                           // than 4 per CU with two rounds each, 29 against 36 us); more rounds beyond
    return (int)(b < 1 ? 1 : b > (size_t)FIT_MAX_BLOCKS ? (size_t)FIT_MAX_BLOCKS : b);
}
static inline size_t fit_chunk(size_t count, int blocks) {
    const size_t per = (count + blocks - 1) / blocks;
    return (per + FIT_ROUND - 1) / FIT_ROUND * FIT_ROUND;
}

struct FitParts {              // workspace: what the workgroups of k_fit_stats leave, and the gather stage's sums
    unsigned *cnt;             // [blocks][levels]
    long long *fx;             // [blocks][levels], at 2^shift[block]
    int *shift;                // [blocks]
    double *miss;              // [blocks]
    unsigned long long *gcnt;  // [FIT_GATHER][levels]
    long long *gfx;            // [FIT_GATHER][levels], at the common scale
};

// 2^shift scales every |x| <= absmax below 2^(62 - log_count)
__device__ __forceinline__ int fit_shift(unsigned absmax_bits, int log_count) {
    if (absmax_bits == 0u) return FIT_NO_SHIFT;
    const int e = (int)((absmax_bits >> 23) & 0xffu) - 127;  // |x| < 2^(e + 1)
    return (62 - log_count) - (e + 1);
}
// float32 -> fixed point at 2^shift, rounded half away from zero, in integer arithmetic (the caller's shift keeps the
// result below 2^62; a float64 scalbn + convert costs four times the instructions)
__device__ __forceinline__ long long fit_fixed(float e, int shift) {
    const unsigned bits = __float_as_uint(e);
    const int ex = (int)((bits >> 23) & 0xffu);
    const long long man = (long long)((bits & 0x7fffffu) | (ex ? 0x800000u : 0u));  // |e| = man * 2^(max(ex, 1) - 150)
    const int up = (ex ? ex : 1) - 150 + shift;
    long long f;
    if (up >= 0) f = man << (up < 39 ? up : 39);
    else if (up > -26) f = (man + (1ll << (-up - 1))) >> -up;
    else f = 0;
    return (bits >> 31) ? -f : f;
}
// v * 2^-d rounded half up (d >= 0): deterministic, one rounding
__device__ __forceinline__ long long fit_rescale(long long v, int d) {
    if (d <= 0) return v;
    if (d >= 63) return 0;
    return (v + (1ll << (d - 1))) >> d;
}

// BY_POSITION: the "bin" of element i is the part of np.array_split(x, levels) it falls in (sizes q + 1 for the first
// r parts, q after them) instead of its codebook index: Codebook.equiprobable's part means.
template <bool BY_POSITION>
__global__ __launch_bounds__(FIT_THREADS) void k_fit_stats(const float *__restrict__ x, size_t count, size_t chunk, Grid g, FitParts out,
                                                           int log_count, size_t q, size_t r, bool vec_ok) {
    __shared__ unsigned cnt[FIT_MAX_LEVELS];
    __shared__ long long fx[FIT_MAX_LEVELS];
    __shared__ float table[2 * FIT_MAX_LEVELS];  // values, then limits: the binary search stays in LDS
    __shared__ double wave_miss[FIT_THREADS / 64];
    __shared__ unsigned wave_max[2][FIT_THREADS / 64];
    for (int k = threadIdx.x; k < FIT_MAX_LEVELS; k += FIT_THREADS) {
        cnt[k] = 0u;
        fx[k] = 0;
    }
    const bool tabled = g.table != nullptr;
    if (tabled) {
        for (int k = threadIdx.x; k < 2 * g.n - 1; k += FIT_THREADS) table[k] = g.table[k];
        g.table = table;
    }
    const float *lim = table + g.n;
    const size_t head = r * (q + 1);
    const size_t begin = (size_t)blockIdx.x * chunk, end = begin + chunk < count ? begin + chunk : count;
    double miss = 0.0;
    int shift = FIT_NO_SHIFT;  // of this workgroup's sums so far
    int round = 0;
    // a round's elements: 8 float4 per thread, fetched one round ahead (issued after the round's barriers, consumed at the
    // top of the next round: the loads fly while the current round is binned)
    float nxt[8][4];
    auto fetch = [&](size_t rbase) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const size_t base = rbase + (size_t)i * FIT_THREADS * 4 + (size_t)threadIdx.x * 4;
            if (base + 4 <= end && vec_ok) {
                const float4 t = *reinterpret_cast<const float4 *>(x + base);
                nxt[i][0] = t.x; nxt[i][1] = t.y; nxt[i][2] = t.z; nxt[i][3] = t.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) nxt[i][j] = base + j < end ? x[base + j] : 0.0f;
            }
        }
    };
    if (begin < end) fetch(begin);
    for (size_t rbase = begin; rbase < end; rbase += FIT_ROUND, round ^= 1) {
        // 1. this round's elements and their largest magnitude
        float v[8][4];
        unsigned mx = 0u;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[i][j] = nxt[i][j];
                mx = max(mx, __float_as_uint(v[i][j]) & 0x7fffffffu);
            }
        for (int off = 32; off; off >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, off));
        if ((threadIdx.x & 63) == 0) wave_max[round][threadIdx.x >> 6] = mx;
        __syncthreads();  // also: the table and the zeroed sums (first round), last round's atomics (later rounds)
        mx = max(max(wave_max[round][0], wave_max[round][1]), max(wave_max[round][2], wave_max[round][3]));
        // 2. a larger element than any before: the sums so far move to its scale
        const int need = fit_shift(mx, log_count);
        if (need < shift) {
            if (shift != FIT_NO_SHIFT)
                for (int k = threadIdx.x; k < g.n; k += FIT_THREADS) fx[k] = fit_rescale(fx[k], shift - need);
            shift = need;
            __syncthreads();
        }
        if (rbase + FIT_ROUND < end) fetch(rbase + FIT_ROUND);
        // 3. bins, sums, miss; a thread keeps a run of one bin in registers
        int run_bin = -1;
        unsigned run_cnt = 0u;
        long long run_fx = 0;
        float run_lo = 1.0f, run_hi = 0.0f;  // [run_lo, run_hi): the current bin of a table (empty interval: none yet)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const size_t base = rbase + (size_t)i * FIT_THREADS * 4 + (size_t)threadIdx.x * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (base + j >= end) break;
                const float e = v[i][j];
                int b;
                if (BY_POSITION) {
                    const size_t pos = base + j;
                    b = (int)(pos < head ? pos / (q + 1) : r + (pos - head) / q);
                } else {
                    float val;
                    if (tabled) {
                        if (e >= run_lo && e < run_hi) {
                            b = run_bin;
                        } else {
                            b = table_index(e, g);
                            run_lo = b > 0 ? lim[b - 1] : -INFINITY;
                            run_hi = b < g.n - 1 ? lim[b] : INFINITY;
                            if (!(e >= run_lo && e < run_hi)) run_lo = 1.0f, run_hi = 0.0f;  // +inf / NaN: searched every time
                        }
                        val = table[b];
                    } else {
                        const float t = grid_pos(e, g, 0.0f, 0.0f, g.top);
                        b = (int)t;
                        val = grid_val(t, g);
                    }
                    const float d = e - val;  // float32, like `data - quant` (codebook.py:210)
                    miss += (double)d * (double)d;
                }
                const long long f = shift == FIT_NO_SHIFT ? 0ll : fit_fixed(e, shift);
                if (b != run_bin) {
                    if (run_cnt) {
                        atomicAdd(&cnt[run_bin], run_cnt);
                        atomicAdd(reinterpret_cast<unsigned long long *>(&fx[run_bin]), (unsigned long long)run_fx);
                    }
                    run_bin = b;
                    run_cnt = 0u;
                    run_fx = 0;
                }
                run_cnt += 1u;
                run_fx += f;
            }
        }
        if (run_cnt) {
            atomicAdd(&cnt[run_bin], run_cnt);
            atomicAdd(reinterpret_cast<unsigned long long *>(&fx[run_bin]), (unsigned long long)run_fx);
        }
    }
    for (int off = 32; off; off >>= 1) miss += __shfl_xor(miss, off);
    if ((threadIdx.x & 63) == 0) wave_miss[threadIdx.x >> 6] = miss;
    __syncthreads();
    if (threadIdx.x == 0) {
        out.miss[blockIdx.x] = (wave_miss[0] + wave_miss[1]) + (wave_miss[2] + wave_miss[3]);
        out.shift[blockIdx.x] = shift;
    }
    for (int k = threadIdx.x; k < g.n; k += FIT_THREADS) {
        out.cnt[(size_t)blockIdx.x * g.n + k] = cnt[k];
        out.fx[(size_t)blockIdx.x * g.n + k] = fx[k];
    }
}

// The scale of the whole array: the smallest of the workgroups' shifts (every thread of the calling workgroup gets it).
__device__ int fit_common_shift(const int *__restrict__ shifts, int blocks, int *scratch /* FIT_THREADS / 64 ints of LDS */) {
    int s = FIT_NO_SHIFT;
    for (int b = threadIdx.x; b < blocks; b += FIT_THREADS) s = min(s, shifts[b]);
    for (int off = 32; off; off >>= 1) s = min(s, __shfl_xor(s, off));
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = s;
    __syncthreads();
    return min(min(scratch[0], scratch[1]), min(scratch[2], scratch[3]));
}

// Gather stage: workgroup j sums the partial rows j, j + FIT_GATHER, ... at the common scale; a thread takes level
// t % width and every (256 / width)-th of those rows, the slices meet in LDS (integers: any order gives the same bits).
__global__ __launch_bounds__(FIT_THREADS) void k_fit_gather(FitParts p, int levels, int width /* levels rounded up to a power of two */, int blocks) {
    __shared__ int scratch[FIT_THREADS / 64];
    __shared__ unsigned long long c_lds[FIT_THREADS];
    __shared__ long long f_lds[FIT_THREADS];
    const int common = fit_common_shift(p.shift, blocks, scratch);
    const int k = threadIdx.x % width, slice = threadIdx.x / width, slices = FIT_THREADS / width;
    unsigned long long c = 0ull;
    long long f = 0;
    if (k < levels) {
        const int step = FIT_GATHER * slices;
        int b = blockIdx.x + FIT_GATHER * slice;
        for (; b + 3 * step < blocks; b += 4 * step) {  // four rows' loads in flight
            int sh[4];
            unsigned cc[4];
            long long ff[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                sh[u] = p.shift[b + u * step];
                cc[u] = p.cnt[(size_t)(b + u * step) * levels + k];
                ff[u] = p.fx[(size_t)(b + u * step) * levels + k];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c += cc[u];
                if (sh[u] != FIT_NO_SHIFT) f += fit_rescale(ff[u], sh[u] - common);
            }
        }
        for (; b < blocks; b += step) {
            const int sh = p.shift[b];
            c += p.cnt[(size_t)b * levels + k];
            if (sh != FIT_NO_SHIFT) f += fit_rescale(p.fx[(size_t)b * levels + k], sh - common);
        }
    }
    c_lds[threadIdx.x] = c;
    f_lds[threadIdx.x] = f;
    __syncthreads();
    if (slice == 0 && k < levels) {
        for (int s2 = 1; s2 < slices; ++s2) {
            c += c_lds[s2 * width + k];
            f += f_lds[s2 * width + k];
        }
        p.gcnt[(size_t)blockIdx.x * levels + k] = c;
        p.gfx[(size_t)blockIdx.x * levels + k] = f;
    }
}

__global__ __launch_bounds__(FIT_THREADS) void k_fit_finish(FitParts p, int levels, int blocks, long long *counts, double *sums, double *sqerr) {
    __shared__ int scratch[FIT_THREADS / 64];
    __shared__ double part[FIT_THREADS];
    const int common = fit_common_shift(p.shift, blocks, scratch);
    for (int k = threadIdx.x; k < levels; k += FIT_THREADS) {
        unsigned long long c = 0ull;
        long long f = 0;
        if (blocks) {
#pragma unroll 8
            for (int j = 0; j < FIT_GATHER; ++j) {
                c += p.gcnt[(size_t)j * levels + k];
                f += p.gfx[(size_t)j * levels + k];
            }
        }
        counts[k] = (long long)c;
        sums[k] = common == FIT_NO_SHIFT ? 0.0 : scalbn((double)f, -common);
    }
    double s = 0.0;
    for (int b = threadIdx.x; b < blocks; b += FIT_THREADS) s += p.miss[b];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = FIT_THREADS / 2; w; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0 && sqerr) *sqerr = part[0];
}

static bool fit_parts(Arena &ws, FitParts *p) {
    p->cnt = ws.take<unsigned>((size_t)FIT_MAX_BLOCKS * FIT_MAX_LEVELS);
    p->fx = ws.take<long long>((size_t)FIT_MAX_BLOCKS * FIT_MAX_LEVELS);
    p->shift = ws.take<int>(FIT_MAX_BLOCKS);
    p->miss = ws.take<double>(FIT_MAX_BLOCKS);
    p->gcnt = ws.take<unsigned long long>((size_t)FIT_GATHER * FIT_MAX_LEVELS);
    p->gfx = ws.take<long long>((size_t)FIT_GATHER * FIT_MAX_LEVELS);
    return p->cnt && p->fx && p->shift && p->miss && p->gcnt && p->gfx;
}

}  // namespace slk

using namespace slk;

extern "C" {

size_t slk_codebook_stats_workspace_bytes(void) {
    Arena probe(reinterpret_cast<void *>(256), (size_t)1 << 40);  // addresses only: nothing is touched
    FitParts p;
    fit_parts(probe, &p);
    return probe.used + 256;
}

int slk_codebook_stats(const float *x, size_t count, int levels, double lo, double hi, const float *table, int by_position,
                       long long *counts, double *sums, double *sqerr, void *workspace, size_t ws_bytes, slk_stream_t stream) {
    SLK_REQUIRE(levels >= 1 && levels <= FIT_MAX_LEVELS, "codebook statistics take 1 to %d bins", FIT_MAX_LEVELS);
    SLK_REQUIRE(by_position || table || (levels >= 2 && lo < hi), "codebook needs levels >= 2 and lo < hi");
    SLK_REQUIRE(count < ((size_t)1 << 31), "at most 2^31 - 1 elements");
    SLK_REQUIRE(counts && sums && (x || count == 0), "null pointer");
    Arena ws(workspace, ws_bytes);
    FitParts parts;
    if (!fit_parts(ws, &parts)) {
        set_error("workspace too small for the codebook statistics");
        return SLK_E_WS;
    }
    hipStream_t s = as_stream(stream);
    int log_count = 0;
    while (((size_t)1 << log_count) < count) ++log_count;
    const int blocks = count ? fit_blocks(count) : 0;
    const Grid g = make_grid(levels, lo, hi, table);
    if (count) {
        const size_t chunk = fit_chunk(count, blocks);
        const bool vec_ok = (uintptr_t)x % 16 == 0;
        if (by_position) {
            SLK_RUN("codebook_stats", 0, 4.0 * count, s,
                    k_fit_stats<true><<<blocks, FIT_THREADS, 0, s>>>(x, count, chunk, g, parts, log_count, count / levels, count % levels, vec_ok));
        } else {
            SLK_RUN("codebook_stats", 0, 4.0 * count, s, k_fit_stats<false><<<blocks, FIT_THREADS, 0, s>>>(x, count, chunk, g, parts, log_count, 0, 0, vec_ok));
        }
        int width = 1;
        while (width < levels) width *= 2;
        SLK_RUN_W("codebook_stats_gather", 0, 12.0 * blocks * levels, FIT_GATHER, s,
                  k_fit_gather<<<FIT_GATHER, FIT_THREADS, 0, s>>>(parts, levels, width, blocks));
    }
    SLK_RUN_W("codebook_stats_finish", 0, 16.0 * levels, 1, s, k_fit_finish<<<1, FIT_THREADS, 0, s>>>(parts, levels, blocks, counts, sums, sqerr));
    return SLK_OK;
}

size_t slk_sort_workspace_bytes(size_t count) {
    size_t a = 0, b = 0;
    if (count >= ((size_t)1 << 31)) return 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, a, (const float *)nullptr, (float *)nullptr, (int)count, 0, 32, (hipStream_t)0);
    (void)hipcub::DeviceSelect::Unique(nullptr, b, (const float *)nullptr, (float *)nullptr, (int *)nullptr, (int)count, (hipStream_t)0);
    return align_up(a > b ? a : b, 256) + 256;
}

int slk_sort_f32(const float *x, size_t count, float *out, void *workspace, size_t ws_bytes, slk_stream_t stream) {
    SLK_REQUIRE(count < ((size_t)1 << 31), "at most 2^31 - 1 elements");
    if (count == 0) return SLK_OK;
    SLK_REQUIRE(x && out && x != out, "null or aliased pointer");
    size_t need = 0;
    SLK_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, need, x, out, (int)count, 0, 32, as_stream(stream)));
    if (need > ws_bytes || !workspace) {
        set_error("workspace too small for sorting %zu elements (%zu > %zu bytes)", count, need, ws_bytes);
        return SLK_E_WS;
    }
    hipStream_t s = as_stream(stream);
    {
        ProfScope prof("sort_f32", 0, 8.0 * 4.0 * (double)count, s);
        SLK_HIP(hipcub::DeviceRadixSort::SortKeys(workspace, need, x, out, (int)count, 0, 32, s));
    }
    return SLK_OK;
}

int slk_unique_f32(const float *sorted, size_t count, float *out, int *n_out, void *workspace, size_t ws_bytes, slk_stream_t stream) {
    SLK_REQUIRE(count < ((size_t)1 << 31), "at most 2^31 - 1 elements");
    SLK_REQUIRE(n_out && (count == 0 || (sorted && out && sorted != out)), "null or aliased pointer");
    size_t need = 0;
    SLK_HIP(hipcub::DeviceSelect::Unique(nullptr, need, sorted, out, n_out, (int)count, as_stream(stream)));
    if (need > ws_bytes || !workspace) {
        set_error("workspace too small for the distinct values of %zu elements", count);
        return SLK_E_WS;
    }
    hipStream_t s = as_stream(stream);
    {
        ProfScope prof("unique_f32", 0, 8.0 * (double)count, s);
        SLK_HIP(hipcub::DeviceSelect::Unique(workspace, need, sorted, out, n_out, (int)count, s));
    }
    return SLK_OK;
}

}  // extern "C"
