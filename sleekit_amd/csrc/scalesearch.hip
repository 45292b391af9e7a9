// SURVEY.md 8(f) rows 1-2, the callers' pre-step: per-row scale selection.
//   compute_non_saturating_scaling   sleekit/scaling.py:44-55    k_row_minmax_scale
//   compute_norm_scaling             sleekit/scaling.py:35-41    k_row_norm_scale
//   compute_min_mse_scaling          sleekit/scaling.py:84-134   k_scale_search  (H = None or a diagonal)
// The grid search evaluates 100 round-to-nearest quantizations per row and keeps the FIRST
// scale with the smallest error, comparing float32 row sums; those sums follow NumPy's pairwise
// order exactly (same tree as prepare.hip: k_diag_mean) so that ties and near-ties fall the same way.
// One workgroup per row: the row stays in registers for the whole search, terms go through LDS,
// leaf pieces are summed one per thread, the tree is folded level by level.
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

// ------------------------------------------------------------------ grid search
// mode 0: error = sum E^2;  mode 1: error = sum hdiag_j * E_j^2   (scaling.py:84-95)
// E = quantize_with_scaling(w, s * base) - w with round-to-nearest quantization (scaling.py:73, 79-80)
template <int EPT>
__global__ __launch_bounds__(256) void k_scale_search(const float *__restrict__ W, const float *__restrict__ base,
                                                      const float *__restrict__ factors, int n_factors,
                                                      const float *__restrict__ hdiag, int R, int n, Grid g,
                                                      float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    SumTree *trees = reinterpret_cast<SumTree *>(smem_raw);
    float *terms = reinterpret_cast<float *>(smem_raw + 2 * sizeof(SumTree));
    const int r = blockIdx.x, t = threadIdx.x;
    const float *w = W + (size_t)r * n;
    prepare_trees(trees, n);
    const float b = base[r];
    float best_err = __builtin_huge_valf(), best_f = __builtin_huge_valf();
    for (int f = 0; f < n_factors; ++f) {
        const float fac = factors[f];
        const float sc = fac * b;         // scaling.py:128
        const float inv = 1.0f / sc;      // scaling.py:80
        const float err = row_sum_numpy(trees, terms, n, [&](int j) {
            const float x = w[j];
            const float q = cb_value(x / sc, g) / inv;
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
    SLK_RUN("scale_search", 0, 4.0 * R * n, s,
            k_scale_search<1><<<R, 256, search_smem(n), s>>>(W, base, factors, n_factors, hdiag, R, n, make_grid(levels, lo, hi, table), out));
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
