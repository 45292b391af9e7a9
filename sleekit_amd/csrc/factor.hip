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
#include <stdlib.h>

#include "mfma64.h"

namespace slk {

constexpr int PANEL = 64;
constexpr int OUTER = 256;
constexpr int DP = 65;  // pitch of the 64 x 64 LDS tiles (odd: column walks are conflict-free)

struct PanelSmem {
    double d[PANEL][DP];      // staging of the diagonal tile, then L11
    double x[PANEL][DP];      // inv(L11)
    double col[2][4][PANEL];  // column j as seen by each wave (only the owner's slot is read)
    double row[2][2 * PANEL]; // row tt of the inverse being propagated, wave-major (+ a dump area)
    double rdiag[PANEL];      // 1 / L11[j][j]
    Tile64Smem mm;
};

// 1/sqrt(d) to double precision: hardware estimate + two Newton steps (3 dependent ops each).
__device__ __forceinline__ double rsqrt_newton(double d) {
    double y = __builtin_amdgcn_rsq(d);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double e = __builtin_fma(-(d * y), y, 1.0);
        y = __builtin_fma(0.5 * y, e, y);
    }
    return y;
}

// Panel step at column k0: block 0 publishes inv(L11) into X; block b >= 1 overwrites the
// tile A[k0 + 64 b ..][k0 ..] with L21 = A21 * inv(L11)^T.
//
// The 64 x 64 diagonal tile lives in REGISTERS: thread (i = t & 63, q = t >> 6) holds row i,
// columns q, q+4, ..., q+60, so wave q is one residue class of columns and its lanes are the
// rows.  Right-looking elimination, one barrier per column, a refined rsqrt instead of
// sqrt + divide.  A taken branch costs ~45 cycles on this chip and this loop is a pure
// latency chain, so both loops are FULLY UNROLLED and branch-free: column/step numbers are
// compile-time constants (static register indices, statically known triangular extents),
// every wave publishes its candidate for column j into its own LDS slot and readers index
// the owner's slot, failures are folded into a flag that is stored once at the end.
// The inverse is a forward substitution on all 64 right-hand sides with the same layout.
__global__ __launch_bounds__(256) void k_chol_panel(double *__restrict__ A, int ld, int k0,
                                                    double *__restrict__ X, int *__restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PanelSmem &sm = *reinterpret_cast<PanelSmem *>(smem_raw);
    const int t = threadIdx.x;
    const int i = t & 63, q = t >> 6;

    // ---- stage the tile coalesced, then pick this thread's row slice out of LDS
    for (int e = t; e < PANEL * PANEL; e += 256) {
        const int r = e >> 6, c = e & 63;
        sm.d[r][c] = (c <= r) ? A[(size_t)(k0 + r) * ld + k0 + c] : 0.0;
    }
    __syncthreads();
    double a[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) a[m] = sm.d[i][q + 4 * m];

    // ---- Cholesky of the tile
    int first_bad = PANEL;
#pragma unroll
    for (int j = 0; j < PANEL; ++j) {
        constexpr int dummy = 0;
        (void)dummy;
        const int jq = j & 3, jm = j >> 2, buf = j & 1;
        sm.col[buf][q][i] = a[jm];  // only wave jq's slot holds column j
        __syncthreads();
        double piv = sm.col[buf][jq][j];
        const double mine = sm.col[buf][jq][i];
        double lc[16];
#pragma unroll
        for (int m = jm; m < 16; ++m) lc[m] = sm.col[buf][jq][q + 4 * m];
        const bool bad = !(piv > 0.0);  // also catches NaN
        first_bad = (bad && first_bad == PANEL) ? j : first_bad;
        piv = bad ? 1.0 : piv;
        const double r = rsqrt_newton(piv);
        const double lij = (i < j) ? 0.0 : mine * r;  // i == j: piv * r = sqrt(piv)
        sm.rdiag[j] = r;                              // same value from every thread
        // element jm: the owner stores L, waves to its right (column q + 4 jm > j) update
        {
            const double upd = __builtin_fma(-lij, lc[jm] * r, a[jm]);
            a[jm] = (q == jq) ? lij : ((q > jq) ? upd : a[jm]);
        }
#pragma unroll
        for (int m = jm + 1; m < 16; ++m) a[m] = __builtin_fma(-lij, lc[m] * r, a[m]);
    }
    if (first_bad != PANEL && blockIdx.x == 0 && t == 0 && info[0] == 0) info[0] = k0 + first_bad + 1;
    // ---- L11 to LDS (lower part) for the substitution
#pragma unroll
    for (int m = 0; m < 16; ++m) sm.d[i][q + 4 * m] = (q + 4 * m <= i) ? a[m] : 0.0;
    __syncthreads();

    // ---- X11 = inv(L11): x[m] accumulates sum_t L[i][t] X[t][c] for c = q + 4m.
    //      Step tt touches only columns c <= tt, i.e. m <= tt >> 2 (m == tt >> 2 iff q <= tt & 3).
    double x[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) x[m] = 0.0;
#pragma unroll
    for (int tt = 0; tt < PANEL; ++tt) {
        const int tq = tt & 3, tm = tt >> 2, buf = tt & 1;
        const double rd = sm.rdiag[tt];
        const bool fin = i == tt;  // one lane per wave finalises its entries of row tt
#pragma unroll
        for (int m = 0; m <= tm; ++m) {
            const double unit = (m == tm) ? ((q == tq) ? 1.0 : 0.0) : 0.0;
            const double xv = (unit - x[m]) * rd;
            const bool live = (m < tm) || (q <= tq);
            x[m] = (fin && live) ? xv : x[m];
            sm.row[buf][fin ? q * 16 + m : PANEL + i] = xv;  // other lanes write to a dump slot: no branch
        }
        __syncthreads();
        const double l = (i > tt) ? sm.d[i][tt] : 0.0;
#pragma unroll
        for (int m = 0; m <= tm; ++m) {
            const double xr = sm.row[buf][q * 16 + m];
            const bool live = (m < tm) || (q <= tq);
            x[m] = live ? __builtin_fma(l, xr, x[m]) : x[m];  // l == 0 for rows already final
        }
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) sm.x[i][q + 4 * m] = x[m];  // exactly zero above the diagonal
    __syncthreads();

    if (blockIdx.x == 0) {
        for (int e = t; e < PANEL * PANEL; e += 256) {
            const int r = e >> 6, c = e & 63;
            X[(size_t)(k0 + r) * ld + k0 + c] = sm.x[r][c];
        }
        return;
    }
    // ---- L21 = A21 * X11^T for this block's 64 rows
    const int r0 = k0 + PANEL * blockIdx.x;
    Acc64 acc;
    acc.zero();
    const double *arow = A + (size_t)r0 * ld + k0;
    const double *pa = arow + (size_t)(t >> 2) * ld + (t & 3) * 8;
    tile64_mac<false>(
        acc, sm.mm, 0, PANEL, [&](int kb, double(&v)[8]) { load8d<true>(pa + kb, v); },
        [&](int kb, double(&v)[8]) {  // B[k][c] = X11[c][k], straight from LDS
            const double *px = &sm.x[t >> 2][kb + (t & 3) * 8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = px[e];
        });
    __syncthreads();
    tile64_foreach(acc, [&](int r, int c, double v) { A[(size_t)(r0 + r) * ld + k0 + c] = v; });
}

// C[bi][bj] -= L[bi][ka:kb] * L[bj][ka:kb]^T for tiles bi in [ti0, ti1), bj in [tj0, tj1), bj <= bi.
__global__ __launch_bounds__(256) void k_syrk_tiles(double *__restrict__ A, int ld, int ti0, int tj0, int ka,
                                                    int kb) {
    __shared__ __attribute__((aligned(16))) Tile64Smem sm;
    const int bi = ti0 + blockIdx.y, bj = tj0 + blockIdx.x;
    if (bj > bi) return;
    Acc64 acc;
    acc.zero();
    const int t = threadIdx.x;
    const double *pa = A + ((size_t)bi * TILE + (t >> 2)) * ld + (t & 3) * 8;
    const double *pb = A + ((size_t)bj * TILE + (t >> 2)) * ld + (t & 3) * 8;
    tile64_mac<false>(
        acc, sm, ka, kb, [&](int k0, double(&v)[8]) { load8d<true>(pa + k0, v); },
        [&](int k0, double(&v)[8]) { load8d<true>(pb + k0, v); });
    double *pc = A + (size_t)bi * TILE * ld + (size_t)bj * TILE;
    tile64_foreach(acc, [&](int r, int c, double v) { pc[(size_t)r * ld + c] -= v; });
}

// Level s of the inverse: nodes [lo, lo + s) u [lo + s, min(lo + 2s, nt)), lo a multiple of 2s.
// STAGE 0:  S[bi][bj] =  sum_{kt = bj .. mid-1} L[bi][kt] X[kt][bj]
// STAGE 1:  X[bi][bj] = -sum_{kt = mid .. bi}   X[bi][kt] S[kt][bj]
// Tiles of one level have K depths from 1 to s tiles; with all of them resident at once the
// level lasts as long as the CU that drew the deepest ones.  Blocks are therefore numbered
// deepest-first (stage 0: by column inside the node, stage 1: by row from the bottom), so the
// blocks id, id + 256, ... that land on one CU mix depths.
template <int STAGE>
__global__ __launch_bounds__(256) void k_trtri_level(const double *__restrict__ L, double *__restrict__ X,
                                                     double *__restrict__ S, int ld, int nt, int s) {
    __shared__ __attribute__((aligned(16))) Tile64Smem sm;
    // block id -> (node, slow, fast): `fast` runs over the s tiles of the balanced direction
    const int id = blockIdx.x;
    const int per_node = s * s, node = id / per_node, in_node = id % per_node;
    const int deep = in_node / s, other = in_node % s;  // deep = 0 is the deepest K range
    const int lo = node * 2 * s, mid = lo + s;
    const int bi = (STAGE == 0) ? mid + other : mid + (s - 1 - deep);
    const int bj = (STAGE == 0) ? lo + deep : lo + other;
    if (bi >= nt) return;
    Acc64 acc;
    acc.zero();
    const int t = threadIdx.x;
    if (STAGE == 0) {
        const double *pa = L + ((size_t)bi * TILE + (t >> 2)) * ld + (t & 3) * 8;
        const double *pb = X + (size_t)(t >> 3) * ld + (size_t)bj * TILE + (t & 7) * 8;
        tile64_mac<true>(
            acc, sm, bj * TILE, mid * TILE, [&](int k0, double(&v)[8]) { load8d<true>(pa + k0, v); },
            [&](int k0, double(&v)[8]) { load8d<true>(pb + (size_t)k0 * ld, v); });
        double *pc = S + (size_t)bi * TILE * ld + (size_t)bj * TILE;
        tile64_foreach(acc, [&](int r, int c, double v) { pc[(size_t)r * ld + c] = v; });
    } else {
        const double *pa = X + ((size_t)bi * TILE + (t >> 2)) * ld + (t & 3) * 8;
        const double *pb = S + (size_t)(t >> 3) * ld + (size_t)bj * TILE + (t & 7) * 8;
        tile64_mac<true>(
            acc, sm, mid * TILE, (bi + 1) * TILE, [&](int k0, double(&v)[8]) { load8d<true>(pa + k0, v); },
            [&](int k0, double(&v)[8]) { load8d<true>(pb + (size_t)k0 * ld, v); });
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
                SLK_RUN("chol_syrk_inner", tiles * 2.0 * 64 * 64 * PANEL, 8.0 * (ld - k0) * PANEL + tiles * 16.0 * 64 * 64, s,
                        k_syrk_tiles<<<grid, 256, 0, s>>>(A, ld, tj0, tj0, k0, k0 + PANEL));
            }
        }
        const int t0 = K1 / TILE;
        if (nt > t0) {
            dim3 grid(nt - t0, nt - t0);
            const double tiles = 0.5 * (nt - t0) * (nt - t0 + 1);
            SLK_RUN("chol_syrk_outer", tiles * 2.0 * 64 * 64 * (K1 - K0), 8.0 * (ld - K1) * (K1 - K0) + tiles * 16.0 * 64 * 64, s,
                    k_syrk_tiles<<<grid, 256, 0, s>>>(A, ld, t0, t0, K0, K1));
        }
    }
    for (int lvl = 1; lvl < nt; lvl *= 2) {
        const int nodes = (nt + 2 * lvl - 1) / (2 * lvl);
        dim3 grid(nodes * lvl * lvl);
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
