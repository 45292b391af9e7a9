// Hessian preparation ahead of the factorisation (sleekit/obq.py:198-204):
//   damping with NumPy's float32 diagonal mean, column order, and the gather of
//   the damped, permuted, index-reversed float64 matrix the factor kernels eat.
#include "npsum.h"

namespace slk {

typedef float float4v_t __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ diag mean
// mean(diag H) in float32 with NumPy's summation order (npsum.h): feeds the damping term.
__global__ __launch_bounds__(256) void k_diag_mean(const PtrTable hs, int n, int ld, float *__restrict__ out) {
    const float *__restrict__ H = hs.p[blockIdx.z];
    out += 64 * blockIdx.z;
    __shared__ SumTree trees[2];
    __shared__ float terms[NP_CHUNK];
    prepare_trees(trees, n);
    const float total = row_sum_numpy(trees, terms, n, [&](int j) { return H[(size_t)j * ld + j]; });
    if (threadIdx.x == 0) out[0] = total / (float)n;
}

// The head of slk_hessian_prepare in ONE launch (one workgroup per layer; n <= 16384): mean(diag H) in NumPy's order
// (the workgroup sum of the local search, npsum.h: HeapSum -- its tree needs no serial construction), the damping term,
// the sort keys and the zeroed rank counters -- three dependent launches of a few microseconds each before (the mean
// alone took 29 us at n = 4096, most of it one thread building the summation tree).
__global__ __launch_bounds__(256) void k_diag_prepare(const PtrTable hs, int n, float damp, const float *__restrict__ miss,
                                                      float *__restrict__ scal, double *__restrict__ keys, int *__restrict__ rank,
                                                      int want_rank) {
    __shared__ HeapSum plan;
    extern __shared__ float diag_terms[];  // heap_sum_floats(n)
    const float *__restrict__ H = hs.p[blockIdx.z];
    scal += 64 * blockIdx.z;
    if (keys) keys += (size_t)n * blockIdx.z;
    if (rank) rank += (size_t)n * blockIdx.z;
    heap_sum_plan(plan, n);
    for (int j = threadIdx.x; j < n; j += 256) diag_terms[heap_sum_pos(j)] = H[(size_t)j * n + j];
    const float mean = heap_sum(plan, diag_terms, n) / (float)n;
    const float add = damp * mean;  // float32 product (obq.py:198 under NEP 50)
    if (threadIdx.x == 0) {
        scal[0] = mean;
        if (keys) scal[1] = add;
    }
    if (keys == nullptr) return;  // (slk_diag_mean: the mean alone)
    for (int i = threadIdx.x; i < n; i += 256) {
        double k = -((double)diag_terms[heap_sum_pos(i)] + (double)add);
        if (miss) k = k * (double)miss[i];
        keys[i] = k;
        if (want_rank) rank[i] = 0;
    }
}

// ------------------------------------------------------------------ order keys
// key[i] = -(double(H_ii) + damp_add) [* double(miss_i)]     (obq.py:64, 69, 81)
// scal[0] = mean(diag); scal[1] <- damp_add = float32(damp) * mean   (float32 product)
__global__ __launch_bounds__(256) void k_order_keys(const PtrTable hs, int n, float damp,
                                                    const float *__restrict__ miss, float *scal,
                                                    double *__restrict__ keys) {
    const float *__restrict__ H = hs.p[blockIdx.z];
    scal += 64 * blockIdx.z;
    keys += (size_t)n * blockIdx.z;
    const float add = damp * scal[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) scal[1] = add;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double k = -((double)H[(size_t)i * n + i] + (double)add);
        if (miss) k = k * (double)miss[i];
        keys[i] = k;
    }
}

// Stable argsort by counting: order[rank(i)] = i with rank = #{j : key_j < key_i or (== and j < i)}.
// n threads, each scanning all keys through LDS tiles (n^2 compares: 17 M at n = 4096).
// Keys are compared through the usual monotone double -> int64 map, a TOTAL order, so the
// output is a permutation even for NaN keys (garbage in never becomes an out-of-range index).
__device__ __forceinline__ long long total_order_key(double k) {
    long long b = __double_as_longlong(k + 0.0);  // -0.0 -> +0.0: NumPy treats them as a tie
    return b ^ ((b >> 63) & 0x7FFFFFFFFFFFFFFFLL);
}

// grid (n / 256, SLICES): block (bx, by) counts, for its 256 keys, the smaller keys inside j-slice by
// and adds the partial rank atomically (integer adds: order-independent, deterministic).
__global__ __launch_bounds__(256) void k_rank_partial(const double *__restrict__ keys, int n, int slice,
                                                      int *__restrict__ rank) {
    __shared__ long long tile[1024];
    keys += (size_t)n * blockIdx.z;
    rank += (size_t)n * blockIdx.z;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const long long ki = i < n ? total_order_key(keys[i]) : 0;
    const int j_lo = blockIdx.y * slice, j_hi = min(n, j_lo + slice);
    int count = 0;
    for (int base = j_lo; base < j_hi; base += 1024) {
        const int m = min(1024, j_hi - base);
        __syncthreads();
        for (int t = threadIdx.x; t < m; t += blockDim.x) tile[t] = total_order_key(keys[base + t]);
        __syncthreads();
        for (int t = 0; t < m; ++t) {
            const long long kj = tile[t];
            // bitwise, not short-circuit: a taken branch costs ~45 cycles here
            count += (int)(kj < ki) | ((int)(kj == ki) & (int)(base + t < i));
        }
    }
    if (i < n && count) atomicAdd(&rank[i], count);
}

__global__ __launch_bounds__(256) void k_rank_scatter(const int *__restrict__ rank, int n, int identity,
                                                      long long *__restrict__ order) {
    rank += (size_t)n * blockIdx.z;
    order += (size_t)n * blockIdx.z;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) order[identity ? i : rank[i]] = i;
}

// ------------------------------------------------------------------ gather
// A[i][j] = double(H[p_i][p_j]) + (i == j ? damp_add : 0),  p_i = order[n-1-i],  for i >= j;
// rows/cols n .. ld-1 are the identity (padding to the factor tile size).
// The lower triangle of the index-reversed matrix is the triangle LAPACK's
// potrf('L') reads in the reference (obq.py:47-50), so an asymmetric H gives the same answer.
__global__ __launch_bounds__(256) void k_gather_reversed(const PtrTable hs, int n, int ld,
                                                         const long long *__restrict__ order,
                                                         const float *__restrict__ scal,
                                                         double *__restrict__ A) {
    const float *__restrict__ H = hs.p[blockIdx.z];
    order += (size_t)n * blockIdx.z;
    scal += 64 * blockIdx.z;
    A += (size_t)ld * ld * blockIdx.z;
    const double add = (double)scal[1];
    for (int i = blockIdx.x; i < ld; i += gridDim.x) {
        double *row = A + (size_t)i * ld;
        if (i >= n) {
            for (int j = threadIdx.x; j < ld; j += blockDim.x) row[j] = (i == j) ? 1.0 : 0.0;
            continue;
        }
        const float *src = H + (size_t)order[n - 1 - i] * n;
        for (int j = threadIdx.x; j < ld; j += blockDim.x) {
            double v = 0.0;
            if (j <= i) {
                v = (double)src[order[n - 1 - j]];
                if (j == i) v = v + add;
            }
            row[j] = v;
        }
    }
}

// The same through LDS: the source row is loaded once with coalesced 16-byte loads and gathered there (random
// 4-byte reads straight from global were one L1/L2 request each), and the tiles strictly above the diagonal,
// which no kernel of the factorisation reads, are not written at all (the 64 x 64 diagonal tiles are kept whole:
// the trailing update reads and rewrites them entirely).  n % 4 == 0, n <= 16384.
__global__ __launch_bounds__(256) void k_gather_reversed_lds(const PtrTable hs, int n, int ld,
                                                             const long long *__restrict__ order,
                                                             const float *__restrict__ scal, double *__restrict__ A) {
    extern __shared__ __attribute__((aligned(16))) float srow[];
    const float *__restrict__ H = hs.p[blockIdx.z];
    order += (size_t)n * blockIdx.z;
    scal += 64 * blockIdx.z;
    A += (size_t)ld * ld * blockIdx.z;
    const double add = (double)scal[1];
    const int t = threadIdx.x, n4 = n >> 2;
    for (int i = blockIdx.x; i < ld; i += gridDim.x) {
        double *row = A + (size_t)i * ld;
        const int stop = min(ld, (i / 64 + 1) * 64);  // end of the diagonal tile
        if (i >= n) {
            for (int j = t; j < stop; j += 256) row[j] = (i == j) ? 1.0 : 0.0;
            continue;
        }
        const float4v_t *src = reinterpret_cast<const float4v_t *>(H + (size_t)order[n - 1 - i] * n);
        __syncthreads();
        for (int c = t; c < n4; c += 256) reinterpret_cast<float4v_t *>(srow)[c] = src[c];
        __syncthreads();
        for (int j = t; j < stop; j += 256) {
            double v = 0.0;
            if (j <= i) {
                v = (double)srow[order[n - 1 - j]];
                if (j == i) v = v + add;
            }
            row[j] = v;
        }
    }
}

// A[i][j] = M[n-1-i][n-1-j] for i >= j (the triangle potrf('L') reads of flip(M)), padded.
__global__ __launch_bounds__(256) void k_load_reversed(const double *__restrict__ M, int n, int ld,
                                                       double *__restrict__ A) {
    for (int i = blockIdx.x; i < ld; i += gridDim.x) {
        double *row = A + (size_t)i * ld;
        for (int j = threadIdx.x; j < ld; j += blockDim.x) {
            double v = 0.0;
            if (i >= n) v = (i == j) ? 1.0 : 0.0;
            else if (j <= i) v = M[(size_t)(n - 1 - i) * n + (n - 1 - j)];
            row[j] = v;
        }
    }
}

// ------------------------------------------------------------------ factor payload (multi-GPU)
// One 8-byte-word buffer per layer travels over xGMI:  [0] status, [1 .. n] order,
// then the upper triangle of U row by row (row i: n - i doubles).  n (n + 1) / 2 + n + 1 words,
// half of what the square matrix would cost.
__device__ __forceinline__ size_t tri_offset(size_t i, size_t n) { return i * n - i * (i - 1) / 2; }

// Rows are taken in pairs (i, n - 1 - i): n + 1 elements per pair whatever i, so every workgroup moves the same amount
// (row by row, the short rows near the bottom left most of a workgroup idle: 65 us for n = 4096 against 38 paired).
template <class F>
__device__ __forceinline__ void for_each_upper_pair(int n, F f) {  // f(row, column) for every column >= row of the pairs
    const int half = (n + 1) / 2;
    for (int p = blockIdx.x; p < half; p += gridDim.x) {
        const int i0 = p, i1 = n - 1 - p;
        const int len0 = n - i0, total = len0 + (i1 != i0 ? n - i1 : 0);
        for (int e = threadIdx.x; e < total; e += 4 * blockDim.x) {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int x = e + h * blockDim.x;
                if (x < total) {
                    const bool first = x < len0;
                    f(first ? i0 : i1, first ? i0 + x : i1 + (x - len0));
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_pack_factor(const double *__restrict__ U, const long long *__restrict__ order,
                                                     const int *__restrict__ info, int n, long long *__restrict__ payload) {
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) payload[0] = info[0];
        for (int j = threadIdx.x; j < n; j += blockDim.x) payload[1 + j] = order[j];
    }
    double *tri = reinterpret_cast<double *>(payload + 1 + n);
    for_each_upper_pair(n, [&](int i, int j) { tri[tri_offset(i, n) - i + j] = U[(size_t)i * n + j]; });
}

// A round's payloads into the stacked factors in ONE launch: blockIdx.y is the layer (payload pointers by value; U (B, n, n),
// order (B, n), info (B) and -- when asked for -- verdict (B): the word AFTER the payload, the root's symmetry verdict of H).
__global__ __launch_bounds__(256) void k_unpack_factor_batch(PtrTable payloads, int n, size_t words, double *__restrict__ U,
                                                             long long *__restrict__ order, int *__restrict__ info,
                                                             int *__restrict__ verdict) {
    const long long *__restrict__ payload = reinterpret_cast<const long long *>(payloads.p[blockIdx.y]);
    U += (size_t)blockIdx.y * n * n;
    order += (size_t)blockIdx.y * n;
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) {
            info[blockIdx.y] = (int)payload[0];
            if (verdict) verdict[blockIdx.y] = (int)payload[words];  // the word after slk_factor_pack's
        }
        for (int j = threadIdx.x; j < n; j += blockDim.x) order[j] = payload[1 + j];
    }
    const double *tri = reinterpret_cast<const double *>(payload + 1 + n);
    for_each_upper_pair(n, [&](int i, int j) { U[(size_t)i * n + j] = tri[tri_offset(i, n) - i + j]; });
}

template <bool UPPER_ONLY>
__global__ __launch_bounds__(256) void k_unpack_factor(const long long *__restrict__ payload, int n, double *__restrict__ U,
                                                       long long *__restrict__ order, int *__restrict__ info) {
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) info[0] = (int)payload[0];
        for (int j = threadIdx.x; j < n; j += blockDim.x) order[j] = payload[1 + j];
    }
    const double *tri = reinterpret_cast<const double *>(payload + 1 + n);
    for_each_upper_pair(n, [&](int i, int j) { U[(size_t)i * n + j] = tri[tri_offset(i, n) - i + j]; });
    if (!UPPER_ONLY) {  // (UPPER_ONLY: the part below the diagonal is known to be zero already -- a buffer zeroed once and reused)
        for (int i = blockIdx.x; i < n; i += gridDim.x) {
            double *dst = U + (size_t)i * n;
            for (int j = threadIdx.x; j < i; j += blockDim.x) dst[j] = 0.0;
        }
    }
}

// ------------------------------------------------------------------ "pivot" order (obq.py:140-166)
// Greedy pivoted Cholesky of the damped Hessian: at step k the remaining column with the largest
// |conditional variance| (first one in the current POSITION order on ties: positions change with the
// reference's swaps) becomes column k.  Only the order is wanted, so the trailing matrix is never
// formed: row p_k of it is evaluated lazily,
//     M_k[p_k][v] = Hd[p_k][v] - sum_{m<k} (b_m[p_k] * b_m[v]) / d_m      (b_m = pivot row of step m)
// with the reference's three roundings per term (product, quotient, difference) in the order of its
// in-place update, and the diagonal is carried along the same way.  n steps of two small launches:
// n^3 / 6 divide-subtract terms on at most (n - k) lanes -- a cold path (no BASELINE config uses it).
__global__ __launch_bounds__(256) void k_pivot_init(const float *__restrict__ H, int n, const float *__restrict__ scal,
                                                    double *__restrict__ diag, int *__restrict__ pos) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    diag[v] = (double)H[(size_t)v * n + v] + (double)scal[1];
    pos[v] = v;
}

// one workgroup: first maximum of |diag| over positions k .. n-1, swapped into position k
__global__ __launch_bounds__(256) void k_pivot_select(const double *__restrict__ diag, int *__restrict__ pos, int n, int k,
                                                      int *__restrict__ piv, double *__restrict__ dpiv) {
    __shared__ double bv[256];
    __shared__ int bj[256];
    double best = -1.0;
    int at = n;
    for (int j = k + threadIdx.x; j < n; j += 256) {
        const double a = fabs(diag[pos[j]]);
        if (a > best || !(best >= 0.0)) {  // strict: keeps the first; a NaN never wins over a number
            if (a == a) best = a, at = j;
        }
    }
    bv[threadIdx.x] = best;
    bj[threadIdx.x] = at;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if (threadIdx.x < w) {
            const double o = bv[threadIdx.x + w];
            const int oj = bj[threadIdx.x + w];
            if (o > bv[threadIdx.x] || (o == bv[threadIdx.x] && oj < bj[threadIdx.x])) bv[threadIdx.x] = o, bj[threadIdx.x] = oj;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int j = bj[0] < n ? bj[0] : k;  // all NaN: keep the position
        const int p = pos[j];
        pos[j] = pos[k];
        pos[k] = p;
        piv[k] = p;
        dpiv[k] = diag[p];
    }
}

// row p_k of the trailing matrix for every remaining column, and the diagonal update of this step
__global__ __launch_bounds__(256) void k_pivot_row(const float *__restrict__ H, int n, const float *__restrict__ scal,
                                                   const int *__restrict__ pos, const int *__restrict__ piv,
                                                   const double *__restrict__ dpiv, int k, double *__restrict__ B,
                                                   double *__restrict__ diag) {
    const int j = k + 1 + blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int v = pos[j], p = piv[k];
    double x = (double)H[(size_t)p * n + v] + (p == v ? (double)scal[1] : 0.0);
    for (int m = 0; m < k; ++m) {
        const double prod = B[(size_t)m * n + p] * B[(size_t)m * n + v];
        x = x - prod / dpiv[m];
    }
    B[(size_t)k * n + v] = x;
    const double sq = x * x;
    diag[v] = diag[v] - sq / dpiv[k];
}

// keys[piv[k]] = k: sorting them ascending (slk_hessian_prepare, SLK_ORDER_KEYS) gives the pivot order
__global__ __launch_bounds__(256) void k_pivot_keys(const int *__restrict__ piv, int n, double *__restrict__ keys) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) keys[piv[k]] = (double)k;
}

}  // namespace slk

using namespace slk;

static inline PtrTable one_h(const float *H) {
    PtrTable t;
    for (int b = 0; b < 64; ++b) t.p[b] = b == 0 ? H : nullptr;
    return t;
}

extern "C" {

int slk_factor_ld(int n) { return (n + 63) / 64 * 64; }

int slk_factor_load(const double *M, int n, double *A, slk_stream_t stream) {
    SLK_REQUIRE(M && A && n > 0, "bad arguments");
    const int ld = slk_factor_ld(n);
    hipStream_t s = as_stream(stream);
    SLK_RUN("factor_load", 0, 4.0 * n * n + 8.0 * ld * ld, s, k_load_reversed<<<ld < 2048 ? ld : 2048, 256, 0, s>>>(M, n, ld, A));
    return SLK_OK;
}

// diag(H^-1)[j] = sum_i U[i][j]^2 for U^T U = H^-1.  One workgroup per 32 columns: 8 row groups x 32 columns, each
// thread takes the rows i = g, g + 8, ... down to the block's last diagonal (U is zero below its diagonal, so the
// rows past a column's own contribute nothing), partial sums combined in a fixed order through LDS.
// (One thread per column walking its rows alone: 1.07 ms at n = 4096.)
// op 0: keys = dinv (inv_diag, obq.py:73-75);  op 1: keys = -diag(Hd)[j] / dinv[j] (combined_diag, obq.py:70-72)
__global__ __launch_bounds__(256) void k_inverse_diag_keys(const double *__restrict__ U, const float *__restrict__ H,
                                                           const float *__restrict__ scal, int n, int op,
                                                           double *__restrict__ keys) {
    __shared__ double part[8][33];
    const int f = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + f;
    const int last = min(n - 1, blockIdx.x * 32 + 31);
    double a0 = 0.0, a1 = 0.0;
    if (j < n) {
        int i = g;
        for (; i + 8 <= last; i += 16) {
            const double u = U[(size_t)i * n + j], v = U[(size_t)(i + 8) * n + j];
            a0 = fma(u, u, a0);
            a1 = fma(v, v, a1);
        }
        for (; i <= last; i += 8) {
            const double u = U[(size_t)i * n + j];
            a0 = fma(u, u, a0);
        }
    }
    part[g][f] = a0 + a1;
    __syncthreads();
    if (g == 0 && j < n) {
        double acc = part[0][f];
#pragma unroll
        for (int q = 1; q < 8; ++q) acc = acc + part[q][f];
        keys[j] = op == 0 ? acc : -((double)H[(size_t)j * n + j] + (double)scal[1]) / acc;
    }
}

size_t slk_factor_payload_words(int n) { return n <= 0 ? 0 : (size_t)n * ((size_t)n + 1) / 2 + (size_t)n + 1; }

int slk_factor_pack(const double *U, const long long *order, const int *info, int n, void *payload, slk_stream_t stream) {
    SLK_REQUIRE(U && order && info && payload && n > 0, "bad arguments");
    hipStream_t s = as_stream(stream);
    SLK_RUN("factor_pack", 0, 8.0 * n * (n + 1.0), s,
            k_pack_factor<<<n < 2048 ? n : 2048, 256, 0, s>>>(U, order, info, n, static_cast<long long *>(payload)));
    return SLK_OK;
}

int slk_factor_unpack(const void *payload, int n, double *U, long long *order, int *info, slk_stream_t stream) {
    SLK_REQUIRE(U && order && info && payload && n > 0, "bad arguments");
    hipStream_t s = as_stream(stream);
    SLK_RUN("factor_unpack", 0, 4.0 * n * (n + 1.0) + 8.0 * n * n, s,
            k_unpack_factor<false><<<n < 2048 ? n : 2048, 256, 0, s>>>(static_cast<const long long *>(payload), n, U, order, info));
    return SLK_OK;
}

int slk_factor_unpack_upper(const void *payload, int n, double *U, long long *order, int *info, slk_stream_t stream) {
    SLK_REQUIRE(U && order && info && payload && n > 0, "bad arguments");
    hipStream_t s = as_stream(stream);
    SLK_RUN("factor_unpack", 0, 4.0 * n * (n + 1.0) + 4.0 * n * (n + 1.0), s,
            k_unpack_factor<true><<<n < 2048 ? n : 2048, 256, 0, s>>>(static_cast<const long long *>(payload), n, U, order, info));
    return SLK_OK;
}

int slk_factor_unpack_upper_batch(const void *const *payloads, int batch, int n, double *U, long long *order, int *info,
                                  int *verdict, slk_stream_t stream) {
    SLK_REQUIRE(U && order && info && payloads && n > 0 && batch >= 1 && batch <= 64, "bad arguments");
    PtrTable t;
    for (int b = 0; b < 64; ++b) {
        SLK_REQUIRE(b >= batch || payloads[b], "null payload in the batch");
        t.p[b] = b < batch ? static_cast<const float *>(payloads[b]) : nullptr;
    }
    hipStream_t s = as_stream(stream);
    SLK_RUN("factor_unpack", 0, batch * 8.0 * n * (n + 1.0), s,
            k_unpack_factor_batch<<<dim3(n < 2048 ? n : 2048, batch), 256, 0, s>>>(t, n, slk_factor_payload_words(n), U, order, info, verdict));
    return SLK_OK;
}

int slk_inverse_diag_keys(const double *U, const float *H, int n, float damp, int combined, double *keys,
                          void *workspace, size_t ws_bytes, slk_stream_t stream) {
    SLK_REQUIRE(U && H && keys && n > 0, "bad arguments");
    Arena ws(workspace, ws_bytes);
    float *scal = ws.take<float>(64);
    double *tmp = ws.take<double>((size_t)n);
    if (!scal || !tmp) {
        set_error("workspace too small");
        return SLK_E_WS;
    }
    hipStream_t s = as_stream(stream);
    SLK_RUN_W("diag_mean", 0, 4.0 * n, 1, s, k_diag_mean<<<1, 256, 0, s>>>(one_h(H), n, n, scal));
    SLK_RUN("order_keys", 0, 12.0 * n, s, k_order_keys<<<(n + 255) / 256, 256, 0, s>>>(one_h(H), n, damp, nullptr, scal, tmp));
    SLK_RUN("inverse_diag_keys", 0, 4.0 * n * n, s,
            k_inverse_diag_keys<<<(n + 31) / 32, 256, 0, s>>>(U, H, scal, n, combined, keys));
    return SLK_OK;
}

int slk_pivot_keys(const float *H, int n, float damp, double *keys, void *workspace, size_t ws_bytes,
                              slk_stream_t stream) {
    SLK_REQUIRE(H && keys && n > 0, "bad arguments");
    Arena ws(workspace, ws_bytes);
    float *scal = ws.take<float>(64);
    double *tmp = ws.take<double>((size_t)n);
    double *diag = ws.take<double>((size_t)n);
    double *dpiv = ws.take<double>((size_t)n);
    int *pos = ws.take<int>((size_t)n);
    int *piv = ws.take<int>((size_t)n);
    double *B = ws.take<double>((size_t)n * n);
    if (!scal || !tmp || !diag || !dpiv || !pos || !piv || !B) {
        set_error("workspace too small for the pivot order of a %d-column Hessian", n);
        return SLK_E_WS;
    }
    hipStream_t s = as_stream(stream);
    const int nb = (n + 255) / 256;
    SLK_RUN_W("diag_mean", 0, 4.0 * n, 1, s, k_diag_mean<<<1, 256, 0, s>>>(one_h(H), n, n, scal));
    SLK_RUN("order_keys", 0, 12.0 * n, s, k_order_keys<<<nb, 256, 0, s>>>(one_h(H), n, damp, nullptr, scal, tmp));  // scal[1] = damping term
    SLK_RUN("pivot_init", 0, 16.0 * n, s, k_pivot_init<<<nb, 256, 0, s>>>(H, n, scal, diag, pos));
    for (int k = 0; k < n; ++k) {
        SLK_RUN_W("pivot_select", 0, 12.0 * (n - k), 1, s, k_pivot_select<<<1, 256, 0, s>>>(diag, pos, n, k, piv, dpiv));
        if (k + 1 < n)
            SLK_RUN_W("pivot_row", 3.0 * k * (n - k - 1), 16.0 * k * (n - k - 1), (n - k - 1 + 255) / 256, s,
                      k_pivot_row<<<(n - k - 1 + 255) / 256, 256, 0, s>>>(H, n, scal, pos, piv, dpiv, k, B, diag));
    }
    SLK_RUN("pivot_keys", 0, 12.0 * n, s, k_pivot_keys<<<nb, 256, 0, s>>>(piv, n, keys));
    return SLK_OK;
}

int slk_diag_mean(const float *H, int n, float *out, void *, size_t, slk_stream_t stream) {
    SLK_REQUIRE(H && out && n > 0, "bad arguments");
    hipStream_t s = as_stream(stream);
    if (n <= 16384) {  // the kernel slk_hessian_prepare uses
        const size_t lds = (size_t)(n + 8 * (n / 128) + 8) * sizeof(float);
        SLK_LDS_OPT_IN(k_diag_prepare, lds);
        SLK_RUN_W("diag_mean", 0, 4.0 * n, 1, s, k_diag_prepare<<<1, 256, lds, s>>>(one_h(H), n, 0.0f, nullptr, out, nullptr, nullptr, 0));
        return SLK_OK;
    }
    SLK_RUN_W("diag_mean", 0, 4.0 * n, 1, s, k_diag_mean<<<1, 256, 0, s>>>(one_h(H), n, n, out));
    return SLK_OK;
}

static int hessian_prepare_impl(const PtrTable &hs, int batch, int n, float damp, int order_mode, const float *miss,
                                long long *order_out, double *A, void *workspace, size_t ws_bytes, slk_stream_t stream) {
    Arena ws(workspace, ws_bytes);
    float *scal = ws.take<float>(64 * (size_t)batch);
    double *keys = ws.take<double>((size_t)n * batch);
    int *rank = ws.take<int>((size_t)n * batch);
    if (!scal || !keys || !rank) {
        set_error("workspace too small");
        return SLK_E_WS;
    }
    hipStream_t s = as_stream(stream);
    const int ld = slk_factor_ld(n);
    const unsigned B = (unsigned)batch;
    bool aligned = true;
    for (int b = 0; b < batch; ++b) aligned = aligned && (uintptr_t)hs.p[b] % 16 == 0;
    const bool weighted = order_mode == SLK_ORDER_ERR || order_mode == SLK_ORDER_SQERR;
    const int identity = order_mode == SLK_ORDER_NONE;
    const bool fused = n <= 16384;
    if (fused) {
        const size_t lds = (size_t)(n + 8 * (n / 128) + 8) * sizeof(float);  // heap_sum_floats(n)
        SLK_LDS_OPT_IN(k_diag_prepare, lds);
        SLK_RUN_W("diag_prepare", 0, 16.0 * n * batch, batch, s,
                  k_diag_prepare<<<dim3(1, 1, B), 256, lds, s>>>(hs, n, damp, weighted ? miss : nullptr, scal, keys, rank, !identity));
    } else {
        SLK_RUN_W("diag_mean", 0, 4.0 * n * batch, batch, s, k_diag_mean<<<dim3(1, 1, B), 256, 0, s>>>(hs, n, n, scal));
        SLK_RUN("order_keys", 0, 12.0 * n * batch, s,
                k_order_keys<<<dim3((n + 255) / 256, 1, B), 256, 0, s>>>(hs, n, damp, weighted ? miss : nullptr, scal, keys));
    }
    if (order_mode == SLK_ORDER_KEYS)  // caller-supplied float64 sort keys (ascending)
        copy_async(keys, miss, sizeof(double) * (size_t)n, s);
    if (!identity) {
        if (!fused) zero_async(rank, sizeof(int) * (size_t)n * batch, s);
        const int slices = 16, slice = (n + slices - 1) / slices;
        SLK_RUN("rank_partial", 0, 16.0 * n * batch, s,
                k_rank_partial<<<dim3((n + 255) / 256, slices, B), 256, 0, s>>>(keys, n, slice, rank));
    }
    SLK_RUN("rank_scatter", 0, 12.0 * n * batch, s, k_rank_scatter<<<dim3((n + 255) / 256, 1, B), 256, 0, s>>>(rank, n, identity, order_out));
    if (n % 4 == 0 && n <= 16384 && aligned) {
        SLK_LDS_OPT_IN(k_gather_reversed_lds, 16384 * 4);
        SLK_RUN("gather_reversed", 0, (4.0 * n * n + 4.0 * ld * ld) * batch, s,
                k_gather_reversed_lds<<<dim3(ld < 2048 ? ld : 2048, 1, B), 256, (size_t)n * 4, s>>>(hs, n, ld, order_out, scal, A));
    } else {
        SLK_RUN("gather_reversed", 0, (2.0 * n * n + 8.0 * ld * ld) * batch, s,
                k_gather_reversed<<<dim3(ld < 2048 ? ld : 2048, 1, B), 256, 0, s>>>(hs, n, ld, order_out, scal, A));
    }
    return SLK_OK;
}

int slk_hessian_prepare(const float *H, int n, float damp, int order_mode, const float *miss,
                        long long *order_out, double *A, void *workspace, size_t ws_bytes,
                        slk_stream_t stream) {
    SLK_REQUIRE(H && order_out && A && n > 0, "bad arguments");
    SLK_REQUIRE(order_mode >= SLK_ORDER_NONE && order_mode <= SLK_ORDER_KEYS, "Invalid act_order value %d",
                order_mode);
    SLK_REQUIRE(order_mode < SLK_ORDER_ERR || miss, "err/sqerr/keys orders need their input vector");
    return hessian_prepare_impl(one_h(H), 1, n, damp, order_mode, miss, order_out, A, workspace, ws_bytes, stream);
}

int slk_hessian_prepare_batch(const float *const *H, int batch, int n, float damp, int order_mode,
                              long long *order_out, double *A, void *workspace, size_t ws_bytes, slk_stream_t stream) {
    SLK_REQUIRE(H && order_out && A && n > 0, "bad arguments");
    SLK_REQUIRE(batch >= 1 && batch <= 64, "batch must be 1..64");
    SLK_REQUIRE(order_mode == SLK_ORDER_NONE || order_mode == SLK_ORDER_DIAG, "the batch form takes the orders none and diag");
    PtrTable hs;
    for (int b = 0; b < 64; ++b) hs.p[b] = b < batch ? H[b] : nullptr;
    for (int b = 0; b < batch; ++b) SLK_REQUIRE(H[b], "null Hessian in the batch");
    return hessian_prepare_impl(hs, batch, n, damp, order_mode, nullptr, order_out, A, workspace, ws_bytes, stream);
}

}  // extern "C"
