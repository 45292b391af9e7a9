// SURVEY.md 8(f) rows 1-2, the callers' pre-step: per-row scale selection.
//   compute_non_saturating_scaling   sleekit/scaling.py:44-55    k_row_minmax_scale
//   compute_norm_scaling             sleekit/scaling.py:35-41    k_row_norm_scale
//   compute_min_mse_scaling          sleekit/scaling.py:84-134   k_scale_search  (H = None or a diagonal)
// The grid search evaluates 100 round-to-nearest quantizations per row and keeps the FIRST
// scale with the smallest error, comparing float32 row sums; those sums follow NumPy's pairwise
// order exactly (same tree as prepare.hip: k_diag_mean) so that ties and near-ties fall the same way.
// One workgroup per row: the row stays in registers for the whole search, terms go through LDS,
// leaf pieces are summed one per thread, the tree is folded level by level.
#include <stdlib.h>

#include "npsum.h"

namespace slk {

// ------------------------------------------------------------------ closed-form scales
__global__ __launch_bounds__(256) void k_row_minmax_scale(const float *__restrict__ W, int R, int n, float lo_code,
                                                          float hi_code, float *__restrict__ scale) {
    __shared__ float smin[4], smax[4];
    const int r = blockIdx.x;
    const float *w = W + (size_t)r * n;
    float mn = w[0], mx = w[0];
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        mn = fminf(mn, w[j]);
        mx = fmaxf(mx, w[j]);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, m, 64));
        mx = fmaxf(mx, __shfl_xor(mx, m, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        smin[threadIdx.x >> 6] = mn;
        smax[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        mn = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
        mx = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
        const float s = fmaxf(mx / hi_code, mn / lo_code);  // scaling.py:53
        scale[r] = fmaxf(s, 1.0e-16f);                      // scaling.py:54
    }
}

__global__ __launch_bounds__(256) void k_row_norm_scale(const float *__restrict__ W, int R, int n,
                                                        float *__restrict__ scale) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    SumTree *trees = reinterpret_cast<SumTree *>(smem_raw);
    float *terms = reinterpret_cast<float *>(smem_raw + 2 * sizeof(SumTree));
    const float *w = W + (size_t)blockIdx.x * n;
    prepare_trees(trees, n);
    const float total = row_sum_numpy(trees, terms, n, [&](int j) { return w[j] * w[j]; });
    if (threadIdx.x == 0) scale[blockIdx.x] = sqrtf(fmaxf(total / (float)n, 1.0e-16f));  // scaling.py:40-41
}

// a / b, correctly rounded, from y = RN(1 / b) by Markstein's sequence (q0 = RN(a y); r = a - b q0, exact in one fma;
// q = RN(q0 + r y)): equal to the IEEE quotient when b's significand is not all ones and nothing over- or underflows
// (`ok`, decided once per divisor; the magnitude test sends the rare tiny or huge dividend to the true divide).
// Three dependent operations instead of the dozen of v_div_scale / v_div_fmas / v_div_fixup.
__device__ __forceinline__ float div_by(float a, float b, float y, bool ok) {
    const float mag = fabsf(a);
    if (ok && mag > 1.0e-30f && mag < 1.0e30f) {
        const float q0 = a * y;
        const float r = __builtin_fmaf(-b, q0, a);
        return __builtin_fmaf(r, y, q0);
    }
    return a / b;
}
__device__ __forceinline__ bool divisor_ok(float b) {
    const unsigned bits = __float_as_uint(b);
    const float mag = fabsf(b);
    return (bits & 0x7FFFFFu) != 0x7FFFFFu && mag > 1.0e-30f && mag < 1.0e30f;
}

// ------------------------------------------------------------------ grid search
// mode 0: error = sum E^2;  mode 1: error = sum hdiag_j * E_j^2   (scaling.py:84-95)
// E = quantize_with_scaling(w, s * base) - w with round-to-nearest quantization (scaling.py:73, 79-80)
template <int EPT>
__global__ __launch_bounds__(256) void k_scale_search(const float *__restrict__ W, const float *__restrict__ base,
                                                      const float *__restrict__ factors, int n_factors,
                                                      const float *__restrict__ hdiag, int R, int n, Grid g,
                                                      float *__restrict__ out, int fast_div) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    SumTree *trees = reinterpret_cast<SumTree *>(smem_raw);
    float *terms = reinterpret_cast<float *>(smem_raw + 2 * sizeof(SumTree));
    const int r = blockIdx.x, t = threadIdx.x;
    const float *w = W + (size_t)r * n;
    prepare_trees(trees, n);
    const float b = base[r];
    float best_err = __builtin_huge_valf(), best_f = __builtin_huge_valf();
    const float inv_step = 1.0f / g.step;
    const bool ok_step = fast_div && g.table == nullptr && divisor_ok(g.step) && divisor_ok(inv_step);
    for (int f = 0; f < n_factors; ++f) {
        const float fac = factors[f];
        const float sc = fac * b;         // scaling.py:128
        const float inv = 1.0f / sc;      // scaling.py:80
        const float back = 1.0f / inv;
        const bool ok_sc = fast_div && divisor_ok(sc) && divisor_ok(inv), ok_inv = ok_sc && divisor_ok(back);
        const float err = row_sum_numpy(trees, terms, n, [&](int j) {
            const float x = w[j];
            const float xs = div_by(x, sc, inv, ok_sc);
            float cv;
            if (ok_step) {
                float tq = div_by(xs - g.zero, g.step, inv_step, true);
                tq = fminf(fmaxf(rintf(tq), 0.0f), g.top);
                cv = tq * g.step + g.zero;
            } else {
                cv = cb_value(xs, g);
            }
            const float q = div_by(cv, inv, back, ok_inv);
            const float e = q - x;
            const float e2 = e * e;
            return hdiag ? hdiag[j] * e2 : e2;
        });
        if (err < best_err) {  // strict: the first minimum is kept (scaling.py:131-133)
            best_err = err;
            best_f = fac;
        }
    }
    if (t == 0) out[r] = b * best_f;  // scaling.py:134
}

// The same search for rows whose NumPy summation tree is REGULAR: n = L * m with L = 2^k leaves of m <= 128
// elements, m a multiple of 8 (4096 = 32 x 128, 3072 = 32 x 96, 1024 = 8 x 128, 768 = 8 x 96, ...).  NumPy's leaf is
// eight running sums r[a] += x[8 i + a] combined as ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)), and its tree a
// perfect binary one over the leaves in order -- i.e. 8 L independent chains of m / 8 sequential adds, then an
// xor-shuffle tree (offsets 1, 2, 4 inside a leaf, 8, 16, 32 over the leaves of a wave, LDS across waves): the same
// additions in the same order with every thread busy and no term ever written to LDS.  A thread keeps its m / 8
// elements of the row (and of the diagonal) in registers for the whole search.  When a row needs fewer than 256
// chains, 256 / (8 L) factors are evaluated side by side; the first smallest error wins, as in the sequential scan.
// (The general kernel above writes the terms to LDS and sums a leaf per thread: 4.5 ms for 100 factors at 4096 x 4096.)
template <int S>  // S = m / 8 adds per chain
__global__ __launch_bounds__(256) void k_scale_search_regular(const float *__restrict__ W, const float *__restrict__ base,
                                                              const float *__restrict__ factors, int n_factors,
                                                              const float *__restrict__ hdiag, int R, int n, Grid g,
                                                              float *__restrict__ out, int L, int fast_div) {
    __shared__ float wsum[2][4];
    __shared__ float g_err[4];
    __shared__ int g_idx[4];
    const int r = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int C = 8 * L, m = 8 * S;            // chains per factor (64, 128 or 256)
    const int groups = 256 / C, waves_per_group = C / 64;
    const int fg = t / C, chain = t % C;
    const int leaf = chain >> 3, a = chain & 7;
    const float *w = W + (size_t)r * n;
    float wv[S], hv[S];
#pragma unroll
    for (int i = 0; i < S; ++i) {
        const int j = leaf * m + a + 8 * i;
        wv[i] = w[j];
        hv[i] = hdiag ? hdiag[j] : 1.0f;
    }
    const float b = base[r];
    float best_err = __builtin_huge_valf();
    int best_idx = 0x7fffffff;
    const int rounds = (n_factors + groups - 1) / groups;
    const bool uniform = g.table == nullptr;
    const float inv_step = 1.0f / g.step;
    const bool ok_step = fast_div && uniform && divisor_ok(g.step) && divisor_ok(inv_step);
    for (int it = 0; it < rounds; ++it) {
        const int f = it * groups + fg;
        const float fac = factors[min(f, n_factors - 1)];
        const float sc = fac * b;     // scaling.py:128
        const float inv = 1.0f / sc;  // scaling.py:80
        const float back = 1.0f / inv;  // RN(1 / inv): the reciprocal the second division is built from
        const bool ok_sc = fast_div && divisor_ok(sc) && divisor_ok(inv), ok_inv = ok_sc && divisor_ok(back);
        float v = 0.0f;
#pragma unroll
        for (int i = 0; i < S; ++i) {
            const float x = wv[i];
            const float xs = div_by(x, sc, inv, ok_sc);  // x / sc
            float cv;
            if (ok_step) {  // codebook.py:47-54 with the division by the step replaced likewise
                float tq = div_by(xs - g.zero, g.step, inv_step, true);
                tq = fminf(fmaxf(rintf(tq), 0.0f), g.top);
                cv = tq * g.step + g.zero;
            } else {
                cv = cb_value(xs, g);
            }
            const float q = div_by(cv, inv, back, ok_inv);  // cv / inv
            const float e = q - x;
            const float e2 = e * e;
            const float term = hdiag ? hv[i] * e2 : e2;
            v = (i == 0) ? term : v + term;
        }
#pragma unroll
        for (int msk = 1; msk <= 32; msk <<= 1) v = v + __shfl_xor(v, msk, 64);  // leaf (1, 2, 4), then the wave's 8 leaves
        float total = v;
        if (waves_per_group > 1) {
            if (lane == 0) wsum[it & 1][wave] = v;
            __syncthreads();
            total = waves_per_group == 2 ? wsum[it & 1][wave & ~1] + wsum[it & 1][wave | 1]
                                         : (wsum[it & 1][0] + wsum[it & 1][1]) + (wsum[it & 1][2] + wsum[it & 1][3]);
        }
        if (f < n_factors && total < best_err) {  // strict: the first minimum is kept (scaling.py:131-133)
            best_err = total;
            best_idx = f;
        }
    }
    // the groups scanned interleaved factor lists: smallest error, ties to the smaller index = the sequential scan's choice
    if (chain == 0) {
        g_err[fg] = best_err;
        g_idx[fg] = best_idx;
    }
    __syncthreads();
    if (t == 0) {
        float be = g_err[0];
        int bi = g_idx[0];
        for (int q = 1; q < groups; ++q)
            if (g_err[q] < be || (g_err[q] == be && g_idx[q] < bi)) {
                be = g_err[q];
                bi = g_idx[q];
            }
        out[r] = bi < n_factors ? b * factors[bi] : b * __builtin_huge_valf();  // scaling.py:134
    }
}

// best/err bookkeeping for the full-Hessian and OBQ searches, whose errors come from slk_row_errors
__global__ __launch_bounds__(256) void k_search_update(const float *__restrict__ err, float factor, int R,
                                                       float *__restrict__ best_err, float *__restrict__ best_f) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    if (err[r] < best_err[r]) {
        best_err[r] = err[r];
        best_f[r] = factor;
    }
}

__global__ __launch_bounds__(256) void k_search_init(int R, float *__restrict__ best_err, float *__restrict__ best_f) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < R) best_err[r] = best_f[r] = __builtin_huge_valf();
}

__global__ __launch_bounds__(256) void k_scale_times(const float *__restrict__ a, const float *__restrict__ b, float c,
                                                     int R, float *__restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < R) out[r] = b ? a[r] * b[r] : a[r] * c;
}

}  // namespace slk

using namespace slk;

extern "C" {

int slk_scale_minmax(const float *W, int R, int n, double lo_code, double hi_code, float *scale, slk_stream_t stream) {
    SLK_REQUIRE(W && scale && R > 0 && n > 0, "bad arguments");
    hipStream_t s = as_stream(stream);
    SLK_RUN("scale_minmax", 0, 4.0 * R * n, s, k_row_minmax_scale<<<R, 256, 0, s>>>(W, R, n, (float)lo_code, (float)hi_code, scale));
    return SLK_OK;
}

static size_t search_smem(int n) { return 2 * sizeof(SumTree) + sizeof(float) * (size_t)(n < NP_CHUNK ? n : NP_CHUNK); }

int slk_scale_norm(const float *W, int R, int n, float *scale, slk_stream_t stream) {
    SLK_REQUIRE(W && scale && R > 0 && n > 0, "bad arguments");
    hipStream_t s = as_stream(stream);
    SLK_RUN("scale_norm", 0, 4.0 * R * n, s, k_row_norm_scale<<<R, 256, search_smem(n), s>>>(W, R, n, scale));
    return SLK_OK;
}

int slk_scale_search(const float *W, const float *base, const float *factors, int n_factors, const float *hdiag, int R,
                     int n, int levels, double lo, double hi, const float *table, float *out, slk_stream_t stream) {
    SLK_REQUIRE(W && base && factors && out && R > 0 && n > 0 && n_factors > 0, "bad arguments");
    SLK_REQUIRE(levels >= 2 && (table || lo < hi), "codebook needs levels >= 2 and lo < hi");
    hipStream_t s = as_stream(stream);
    const Grid g = make_grid(levels, lo, hi, table);
    // regular summation tree (see k_scale_search_regular): L = 2^k leaves of m = 8 S elements, 64 ... 256 chains
    int m = n, L = 1;
    bool regular = true;
    while (m > 128 && regular) {
        regular = m % 2 == 0 && (m / 2) % 8 == 0;
        m /= 2;
        L *= 2;
    }
    regular = regular && m % 8 == 0 && m >= 8 && (L == 8 || L == 16 || L == 32) && !opt(OPT_NO_REGULAR_SEARCH);
    const int fast_div = !opt(OPT_NO_FAST_SEARCH_DIV);
    if (regular) {
#define SLK_SEARCH_CASE(SV)                                                                                                     \
    case SV:                                                                                                                    \
        SLK_RUN("scale_search", 0, 4.0 * R * n, s,                                                                              \
                k_scale_search_regular<SV><<<R, 256, 0, s>>>(W, base, factors, n_factors, hdiag, R, n, g, out, L, fast_div));   \
        return SLK_OK;
        switch (m / 8) {
            SLK_SEARCH_CASE(16)
            SLK_SEARCH_CASE(15)
            SLK_SEARCH_CASE(14)
            SLK_SEARCH_CASE(13)
            SLK_SEARCH_CASE(12)
            SLK_SEARCH_CASE(11)
            SLK_SEARCH_CASE(10)
            SLK_SEARCH_CASE(9)
            default: break;  // leaves of 64 elements or fewer: the general kernel
        }
#undef SLK_SEARCH_CASE
    }
    SLK_RUN("scale_search", 0, 4.0 * R * n, s,
            k_scale_search<1><<<R, 256, search_smem(n), s>>>(W, base, factors, n_factors, hdiag, R, n, g, out, fast_div));
    return SLK_OK;
}

int slk_search_step(const float *err, float factor, int R, float *best_err, float *best_f, int init, slk_stream_t stream) {
    SLK_REQUIRE(best_err && best_f && R > 0, "bad arguments");
    hipStream_t s = as_stream(stream);
    if (init) SLK_RUN("search_init", 0, 8.0 * R, s, k_search_init<<<(R + 255) / 256, 256, 0, s>>>(R, best_err, best_f));
    if (err) SLK_RUN("search_update", 0, 12.0 * R, s, k_search_update<<<(R + 255) / 256, 256, 0, s>>>(err, factor, R, best_err, best_f));
    return SLK_OK;
}

int slk_scale_times(const float *a, const float *b, float c, int R, float *out, slk_stream_t stream) {
    SLK_REQUIRE(a && out && R > 0, "bad arguments");
    hipStream_t s = as_stream(stream);
    SLK_RUN("scale_times", 0, 12.0 * R, s, k_scale_times<<<(R + 255) / 256, 256, 0, s>>>(a, b, c, R, out));
    return SLK_OK;
}

}  // extern "C"
