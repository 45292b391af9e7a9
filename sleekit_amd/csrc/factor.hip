// a6  compute_hessian_chol (sleekit/obq.py:38-55) on the device, float64.
//
// The reference computes U = flip(inv(cholesky(flip(Hd)))).  slk_hessian_prepare
// already produced A = flip(Hd[order][:, order]) (lower triangle, padded with an
// identity block to a multiple of 64), so here:
//
//   1. L = chol(A), blocked right-looking with two levels (64-wide panels inside
//      256-wide outer blocks) so the trailing matrix is re-read once per 256
//      columns rather than once per 64;
//      every panel workgroup factors the 64 x 64 diagonal tile in LDS and inverts
//      it, so the panel solve is an MFMA product L21 = A21 * inv(L11)^T;
//   2. X = inv(L) by level-wise merging of inverted diagonal blocks:
//        X[B, A] = -X[B, B] * (L[B, A] * X[A, A])      (two MFMA products per level);
//   3. U[i][j] = X[n-1-i][n-1-j]  (index reversal back; upper triangular).
//
// All products run on v_mfma_f64_16x16x4_f64 through mfma64.h.
#include "mfma64.h"

namespace slk {

constexpr int PANEL = 64;
constexpr int OUTER = 256;
constexpr int DP = 65;  // pitch of the 64 x 64 LDS tiles (odd: column walks are conflict-free)

struct PanelSmem {
    double d[PANEL][DP];   // diagonal tile -> L11
    double x[PANEL][DP];   // inv(L11)
    Tile64Smem mm;
};

// Panel step at column k0: block 0 publishes inv(L11) into X; block b >= 1 overwrites the
// tile A[k0 + 64 b ..][k0 ..] with L21.
__global__ __launch_bounds__(256) void k_chol_panel(double *__restrict__ A, int ld, int k0,
                                                    double *__restrict__ X, int *__restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PanelSmem &sm = *reinterpret_cast<PanelSmem *>(smem_raw);
    const int t = threadIdx.x;

    // ---- load the diagonal tile (lower part; upper part cleared)
    for (int e = t; e < PANEL * PANEL; e += 256) {
        const int i = e >> 6, j = e & 63;
        sm.d[i][j] = (j <= i) ? A[(size_t)(k0 + i) * ld + k0 + j] : 0.0;
        sm.x[i][j] = 0.0;
    }
    // ---- unblocked right-looking Cholesky in LDS
    for (int j = 0; j < PANEL; ++j) {
        __syncthreads();
        double piv = sm.d[j][j];
        if (!(piv > 0.0)) {  // also catches NaN
            if (blockIdx.x == 0 && t == 0 && info[0] == 0) info[0] = k0 + j + 1;
            piv = 1.0;
        }
        const double s = sqrt(piv);
        const double inv = 1.0 / s;
        __syncthreads();
        if (t < PANEL) {
            if (t == j) sm.d[j][j] = s;
            else if (t > j) sm.d[t][j] = sm.d[t][j] * inv;
        }
        __syncthreads();
        // trailing update of the tile: columns c > j, rows i >= c
        const int i = t & 63;
        for (int c = j + 1 + (t >> 6); c < PANEL; c += 4)
            if (i >= c) sm.d[i][c] = sm.d[i][c] - sm.d[i][j] * sm.d[c][j];
    }
    __syncthreads();
    // ---- X11 = inv(L11): right-looking forward substitution on all columns at once.
    //      sm.x accumulates sum_t L[i][t] X[t][c]; row tt is finalised at step tt.
    for (int tt = 0; tt < PANEL; ++tt) {
        if (t <= tt) {
            const double acc = sm.x[tt][t];
            sm.x[tt][t] = ((t == tt ? 1.0 : 0.0) - acc) / sm.d[tt][tt];
        }
        __syncthreads();
        const int c = t & 63;
        if (c <= tt)
            for (int i = tt + 1 + (t >> 6); i < PANEL; i += 4) sm.x[i][c] = sm.x[i][c] + sm.d[i][tt] * sm.x[tt][c];
        __syncthreads();
    }

    if (blockIdx.x == 0) {
        for (int e = t; e < PANEL * PANEL; e += 256) {
            const int i = e >> 6, j = e & 63;
            X[(size_t)(k0 + i) * ld + k0 + j] = sm.x[i][j];  // upper part is exactly zero
        }
        return;
    }
    // ---- L21 = A21 * X11^T for this block's 64 rows
    const int r0 = k0 + PANEL * blockIdx.x;
    Acc64 acc;
    acc.zero();
    const double *arow = A + (size_t)r0 * ld + k0;
    tile64_mac<true, true>(
        acc, sm.mm, 0, PANEL, [&](int r, int k) { return arow[(size_t)r * ld + k]; },
        [&](int k, int c) { return sm.x[c][k]; });
    __syncthreads();
    tile64_foreach(acc, [&](int r, int c, double v) { A[(size_t)(r0 + r) * ld + k0 + c] = v; });
}

// C[bi][bj] -= L[bi][ka:kb] * L[bj][ka:kb]^T for tiles bi in [ti0, ti1), bj in [tj0, tj1), bj <= bi.
__global__ __launch_bounds__(256) void k_syrk_tiles(double *__restrict__ A, int ld, int ti0, int tj0, int ka,
                                                    int kb) {
    __shared__ Tile64Smem sm;
    const int bi = ti0 + blockIdx.y, bj = tj0 + blockIdx.x;
    if (bj > bi) return;
    Acc64 acc;
    acc.zero();
    const double *pa = A + (size_t)bi * TILE * ld;
    const double *pb = A + (size_t)bj * TILE * ld;
    tile64_mac<true, false>(
        acc, sm, ka, kb, [&](int r, int k) { return pa[(size_t)r * ld + k]; },
        [&](int k, int c) { return pb[(size_t)c * ld + k]; });
    double *pc = A + (size_t)bi * TILE * ld + (size_t)bj * TILE;
    tile64_foreach(acc, [&](int r, int c, double v) { pc[(size_t)r * ld + c] -= v; });
}

// Level s of the inverse: nodes [lo, lo + s) u [lo + s, min(lo + 2s, nt)), lo a multiple of 2s.
// STAGE 0:  S[bi][bj] =  sum_{kt = bj .. mid-1} L[bi][kt] X[kt][bj]
// STAGE 1:  X[bi][bj] = -sum_{kt = mid .. bi}   X[bi][kt] S[kt][bj]
template <int STAGE>
__global__ __launch_bounds__(256) void k_trtri_level(const double *__restrict__ L, double *__restrict__ X,
                                                     double *__restrict__ S, int ld, int nt, int s) {
    __shared__ Tile64Smem sm;
    const int bi = blockIdx.y, bj = blockIdx.x;
    const int lo = bi / (2 * s) * (2 * s), mid = lo + s;
    if (bi < mid || bj < lo || bj >= mid) return;
    Acc64 acc;
    acc.zero();
    if (STAGE == 0) {
        const double *pa = L + (size_t)bi * TILE * ld;
        const double *pb = X + (size_t)bj * TILE;
        tile64_mac<true, true>(
            acc, sm, bj * TILE, mid * TILE, [&](int r, int k) { return pa[(size_t)r * ld + k]; },
            [&](int k, int c) { return pb[(size_t)k * ld + c]; });
        double *pc = S + (size_t)bi * TILE * ld + (size_t)bj * TILE;
        tile64_foreach(acc, [&](int r, int c, double v) { pc[(size_t)r * ld + c] = v; });
    } else {
        const double *pa = X + (size_t)bi * TILE * ld;
        const double *pb = S + (size_t)bj * TILE;
        tile64_mac<true, true>(
            acc, sm, mid * TILE, (bi + 1) * TILE, [&](int r, int k) { return pa[(size_t)r * ld + k]; },
            [&](int k, int c) { return pb[(size_t)k * ld + c]; });
        double *pc = X + (size_t)bi * TILE * ld + (size_t)bj * TILE;
        tile64_foreach(acc, [&](int r, int c, double v) { pc[(size_t)r * ld + c] = -v; });
    }
}

// U[i][j] = X[n-1-i][n-1-j] for j >= i, 0 below the diagonal.
__global__ __launch_bounds__(256) void k_flip_out(const double *__restrict__ X, int ld, int n,
                                                  double *__restrict__ U) {
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const double *src = X + (size_t)(n - 1 - i) * ld;
        double *dst = U + (size_t)i * n;
        for (int j = threadIdx.x; j < n; j += blockDim.x) dst[j] = (j >= i) ? src[n - 1 - j] : 0.0;
    }
}

__global__ void k_clear_info(int *info) { info[0] = 0; }

}  // namespace slk

using namespace slk;

extern "C" int slk_chol_inverse_upper(double *A, int n, double *U, int *info, void *workspace,
                                      size_t ws_bytes, slk_stream_t stream) {
    SLK_REQUIRE(A && U && info && n > 0, "bad arguments");
    const int ld = slk_factor_ld(n);
    const int nt = ld / TILE;
    Arena ws(workspace, ws_bytes);
    double *X = ws.take<double>((size_t)ld * ld);
    double *S = ws.take<double>((size_t)ld * ld);
    if (!X || !S) {
        set_error("workspace too small for the %d x %d factorisation", n, n);
        return SLK_E_WS;
    }
    hipStream_t s = as_stream(stream);
    static bool attr_set = false;
    if (!attr_set) {
        SLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_chol_panel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PanelSmem)));
        attr_set = true;
    }
    SLK_RUN("clear_info", 0, 4, s, k_clear_info<<<1, 1, 0, s>>>(info));

    for (int K0 = 0; K0 < ld; K0 += OUTER) {
        const int K1 = K0 + OUTER < ld ? K0 + OUTER : ld;
        for (int k0 = K0; k0 < K1; k0 += PANEL) {
            const int below = (ld - k0) / PANEL;  // tiles from the diagonal tile down
            // potf2 + inverse of the 64-tile (2/3 * 64^3) and the triangular product below it
            SLK_RUN("chol_panel", 2.0 / 3.0 * 64 * 64 * 64 + (double)(below - 1) * 64 * 64 * 64, 16.0 * below * 64 * 64, s,
                    k_chol_panel<<<below, 256, sizeof(PanelSmem), s>>>(A, ld, k0, X, info));
            // inner update: columns of this outer block to the right of the panel
            const int tj0 = k0 / TILE + 1, tj1 = K1 / TILE;
            if (tj1 > tj0) {
                dim3 grid(tj1 - tj0, nt - tj0);
                double tiles = 0;
                for (int bj = tj0; bj < tj1; ++bj) tiles += nt - bj;
                SLK_RUN("chol_syrk", tiles * 2.0 * 64 * 64 * PANEL, 8.0 * (ld - k0) * PANEL + tiles * 16.0 * 64 * 64, s,
                        k_syrk_tiles<<<grid, 256, 0, s>>>(A, ld, tj0, tj0, k0, k0 + PANEL));
            }
        }
        const int t0 = K1 / TILE;
        if (nt > t0) {
            dim3 grid(nt - t0, nt - t0);
            const double tiles = 0.5 * (nt - t0) * (nt - t0 + 1);
            SLK_RUN("chol_syrk", tiles * 2.0 * 64 * 64 * (K1 - K0), 8.0 * (ld - K1) * (K1 - K0) + tiles * 16.0 * 64 * 64, s,
                    k_syrk_tiles<<<grid, 256, 0, s>>>(A, ld, t0, t0, K0, K1));
        }
    }
    for (int lvl = 1; lvl < nt; lvl *= 2) {
        dim3 grid(nt, nt);
        // work of this level: for every node, tiles (bi in B, bj in A) with their triangular K ranges
        double f0 = 0, f1 = 0, tiles = 0;
        for (int lo = 0; lo + lvl < nt; lo += 2 * lvl) {
            const int mid = lo + lvl, hi = lo + 2 * lvl < nt ? lo + 2 * lvl : nt;
            for (int bi = mid; bi < hi; ++bi)
                for (int bj = lo; bj < mid; ++bj) {
                    f0 += 2.0 * 64 * 64 * 64 * (mid - bj);
                    f1 += 2.0 * 64 * 64 * 64 * (bi + 1 - mid);
                    tiles += 1;
                }
        }
        SLK_RUN("trtri_stage0", f0, f0 / 64 * 8.0 / 2 + tiles * 8.0 * 64 * 64, s, k_trtri_level<0><<<grid, 256, 0, s>>>(A, X, S, ld, nt, lvl));
        SLK_RUN("trtri_stage1", f1, f1 / 64 * 8.0 / 2 + tiles * 8.0 * 64 * 64, s, k_trtri_level<1><<<grid, 256, 0, s>>>(A, X, S, ld, nt, lvl));
    }
    SLK_RUN("flip_out", 0, 12.0 * n * n, s, k_flip_out<<<n < 2048 ? n : 2048, 256, 0, s>>>(X, ld, n, U));
    return SLK_OK;
}
