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
constexpr int OUTER_SMALL = 256, OUTER_LARGE = 512;  // columns per outer block (a host-side choice, see chol_inverse_impl)
constexpr int LOOKAHEAD_MIN_TILES = 24;  // trailing tile rows from which an outer syrk is split (below it lasts < 25 us whole)

// cycle counters of workgroup 1 of the panel kernel (wave 0; "lookahead"-free debug aid, read by slk_probe_panel_cycles)
__device__ long long g_panel_cycles[16];

constexpr int TP = 66;  // pitch of the 64 x 64 LDS tiles: MFMA operand reads walk banks 4 row + 2 k

struct PanelSmem {
    double t[PANEL][TP];    // diagonal tile -> L11 (lower)
    double x[PANEL][TP];    // inv(L11); before the factorisation starts: the previous panel's L of the diagonal tile's rows
    double a21[PANEL][TP];  // this workgroup's tile of A21
    double lb[PANEL][TP];   // the previous panel's L of this workgroup's rows
    double rdiag[PANEL];    // 1 / L11[j][j]
    int early_count;        // k_chol_chain: waves whose part of X's first three row blocks has left for memory
};

// 1/sqrt(d) to double precision: hardware estimate + two Newton steps (3 dependent ops each).
#ifndef NEWTON_STEPS
#define NEWTON_STEPS 2
#endif
__device__ __forceinline__ double rsqrt_newton(double d) {
    double y = __builtin_amdgcn_rsq(d);
#pragma unroll
    for (int it = 0; it < NEWTON_STEPS; ++it) {
        const double e = __builtin_fma(-(d * y), y, 1.0);
        y = __builtin_fma(0.5 * y, e, y);
    }
    return y;
}

// ---- 16 x 16 block helpers on v_mfma_f64_16x16x4_f64; operands straight from LDS ----------
// "A pattern": lane l reads M[row = l & 15][4 g + (l >> 4)] of a row-major block, which serves both
// as the A operand of M and as the B operand of M^T.
__device__ __forceinline__ double4_t blk_load_d(const double *blk, int lane) {  // D layout: reg r = [ (l>>4) + 4r ][ l & 15 ]
    double4_t c;
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = blk[((lane >> 4) + 4 * r) * TP + (lane & 15)];
    return c;
}
__device__ __forceinline__ void blk_store_d(double *blk, int lane, double4_t c) {
#pragma unroll
    for (int r = 0; r < 4; ++r) blk[((lane >> 4) + 4 * r) * TP + (lane & 15)] = c[r];
}
// acc += sign * A * Bt^T, both 16 x 16 row-major blocks in LDS
__device__ __forceinline__ double4_t blk_mma_abt(const double *a, const double *bt, double4_t acc, double sign, int lane) {
    const int o = (lane & 15) * TP + (lane >> 4);
#pragma unroll
    for (int g = 0; g < 4; ++g) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sign * a[o + 4 * g], bt[o + 4 * g], acc, 0, 0, 0);
    return acc;
}
// acc += A * B, B row-major [k][col] in LDS
__device__ __forceinline__ double4_t blk_mma_ab(const double *a, const double *b, double4_t acc, int lane) {
    const int oa = (lane & 15) * TP + (lane >> 4), ob = (lane >> 4) * TP + (lane & 15);
#pragma unroll
    for (int g = 0; g < 4; ++g) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[oa + 4 * g], b[ob + 4 * g * TP], acc, 0, 0, 0);
    return acc;
}
// acc += sign * A * S where S is a 16 x 16 block held in D layout: its register g IS the B operand of k-group g
__device__ __forceinline__ double4_t blk_mma_a_reg(const double *a, double4_t s, double4_t acc, double sign, int lane) {
    const int oa = (lane & 15) * TP + (lane >> 4);
#pragma unroll
    for (int g = 0; g < 4; ++g) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sign * a[oa + 4 * g], s[g], acc, 0, 0, 0);
    return acc;
}

// Panel step at column k0: block 0 publishes inv(L11) into X; block b >= 1 overwrites the
// tile A[k0 + 64 b ..][k0 ..] with L21 = A21 * inv(L11)^T.
//
// The chain of n pivots is the critical path of the whole factorisation, so the 64 x 64
// diagonal tile is processed in four 16-column strips and everything that is not inherently
// sequential runs on MFMA:
//   strip kb  (ONE wave, no barrier, no LDS in the loop): lane i holds row 16 kb + i of the
//             strip (all rows down to 63) in 16 registers.  Column j: pivot by v_readlane,
//             refined rsqrt, scale (this IS the triangular solve for the rows below), then
//             a[c] -= l_ij * l_cj with l_cj read-laned from lane c.  Fully unrolled, branch-free.
//   update    T[rb, cb] -= L[rb, kb] L[cb, kb]^T for the blocks right of the strip: 4 MFMAs each,
//             spread over the four waves; meanwhile one wave inverts the 16 x 16 diagonal block.
//   inverse   off-diagonal blocks X[rb, cb] = -X[rb, rb] * sum_k L[rb, k] X[k, cb] level by level;
//             the inner sum stays in registers: an accumulator tile is already the next B operand.
// Two barriers per strip, three for the inverse.
__global__ __launch_bounds__(256) void k_chol_panel(double *__restrict__ A, int ld, int k0, int kprev, int below, int rest_cols,
                                                    double *__restrict__ X, int *__restrict__ info, int dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // (blockIdx.z: the layer of a batched factorisation -- square matrices of one size, one after the other)
    A += (size_t)blockIdx.z * ld * ld;
    const int t = threadIdx.x;
    if ((int)blockIdx.x >= below) {
        // ---- part B: what is left of the PREVIOUS panel's update inside this outer block -- the tile columns right of this
        //      panel, C[bi][bj] -= L[bi][kprev ...] L[bj][kprev ...]^T (K = 64), which only the panels after this one read.
        //      (It used to be a launch of its own between two panels, k_syrk_tiles over all the block's columns: the part this
        //      panel needs is now its prologue below, the rest rides along here.)
        Tile64Smem &ts = *reinterpret_cast<Tile64Smem *>(smem_raw);
        const int nt = ld / TILE, tj0 = k0 / TILE + 1;
        int idx = (int)blockIdx.x - below, bj = tj0;
        for (int c = 0; c < rest_cols; ++c, ++bj) {
            const int count = nt - bj;
            if (idx < count) break;
            idx -= count;
        }
        const int bi = bj + idx;
        if (bj >= tj0 + rest_cols || bi >= nt) return;
        Acc64 acc;
        acc.zero();
        const double *pa = A + ((size_t)bi * TILE + (t >> 2)) * ld + (t & 3) * 8;
        const double *pb = A + ((size_t)bj * TILE + (t >> 2)) * ld + (t & 3) * 8;
        double *pc = A + (size_t)bi * TILE * ld + (size_t)bj * TILE;
        Acc64 old;  // fetched NOW: its trip to memory hides behind the products
        tile64_map(old, [&](int r, int c) { return pc[(size_t)r * ld + c]; });
        tile64_mac<false>(
            acc, ts, kprev, kprev + PANEL, [&](int k, double(&v)[8]) { load8d<true>(pa + k, v); },
            [&](int k, double(&v)[8]) { load8d<true>(pb + k, v); });
        tile64_foreach2(old, acc, [&](int r, int c, double o, double v) { pc[(size_t)r * ld + c] = o - v; });
        return;
    }
    PanelSmem &sm = *reinterpret_cast<PanelSmem *>(smem_raw);
    X += (size_t)blockIdx.z * ld * ld;
    info += blockIdx.z;
    const int lane = t & 63, wave = t >> 6;
    const bool timing = dbg && blockIdx.x == 1 && blockIdx.z == 0 && wave == 0;
    long long tmark = timing ? (long long)__builtin_readcyclecounter() : 0;
    auto lap = [&](int slot) {
        if (timing) {
            const long long now = (long long)__builtin_readcyclecounter();
            if (lane == 0) g_panel_cycles[slot] += now - tmark;
            tmark = now;
        }
    };
    const bool has_prev = kprev >= 0, has_rows = blockIdx.x > 0;

    // Everything this workgroup reads is fetched NOW: the diagonal tile and -- when a panel of this outer block came before --
    // that panel's L for the diagonal tile's rows FIRST (the pivots wait for them), then its own tile of A21 and that panel's
    // L for its own rows, which land while the diagonal tile is being updated (128 KB per workgroup arrive at ~26 GB/s: 4.4 us
    // when everything was waited for at once).
    const int sr = t >> 2, sc8 = (t & 3) * 16;  // a thread carries 16 consecutive doubles of one row of every tile
    double vt[16], vb[16];
    {
        const double *pd = A + (size_t)(k0 + sr) * ld + k0 + sc8;
        double vd[16], vk[16];
        load8d<true>(pd, *reinterpret_cast<double(*)[8]>(&vd[0]));
        load8d<true>(pd + 8, *reinterpret_cast<double(*)[8]>(&vd[8]));
        if (has_prev) {
            const double *pk = A + (size_t)(k0 + sr) * ld + kprev + sc8;
            load8d<true>(pk, *reinterpret_cast<double(*)[8]>(&vk[0]));
            load8d<true>(pk + 8, *reinterpret_cast<double(*)[8]>(&vk[8]));
        }
        if (has_rows) {
            const double *pt = A + (size_t)(k0 + PANEL * blockIdx.x + sr) * ld + k0 + sc8;
            load8d<true>(pt, *reinterpret_cast<double(*)[8]>(&vt[0]));
            load8d<true>(pt + 8, *reinterpret_cast<double(*)[8]>(&vt[8]));
            if (has_prev) {
                const double *pl = A + (size_t)(k0 + PANEL * blockIdx.x + sr) * ld + kprev + sc8;
                load8d<true>(pl, *reinterpret_cast<double(*)[8]>(&vb[0]));
                load8d<true>(pl + 8, *reinterpret_cast<double(*)[8]>(&vb[8]));
            }
        }
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            const int c = sc8 + e;
            // without an earlier panel the upper triangle is cleared (never read); with one the whole tile takes the update
            *reinterpret_cast<double2_t *>(&sm.t[sr][c]) = (double2_t){(has_prev || c <= sr) ? vd[e] : 0.0, (has_prev || c + 1 <= sr) ? vd[e + 1] : 0.0};
            *reinterpret_cast<double2_t *>(&sm.x[sr][c]) = has_prev ? (double2_t){vk[e], vk[e + 1]} : (double2_t){0.0, 0.0};
        }
    }
    __syncthreads();
    lap(0);  // staged

    // ---- prologue (has_prev): the previous panel's update of this panel's tile column, which used to be part of a launch
    //      of its own (k_syrk_tiles).  C -= L_rows L_diag^T, K = 64, accumulated from zero over ascending k and subtracted
    //      once -- the same chain of fused multiply-adds and the same single rounding of the difference as that kernel's
    //      (tile64_mac + `old - acc`), so the factor is unchanged bit for bit.
    //      Diagonal tile: the ten 16 x 16 blocks on and below its diagonal, all four waves, before the pivots can start.
    //      Own tile (16 blocks): waves 1 - 3, in the shadow of the first strip's pivot chain.
    auto block_syrk = [&](double (*c)[TP], const double (*a)[TP], int rb, int cb) {
        double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) acc = blk_mma_abt(&a[16 * rb][16 * kq], &sm.x[16 * cb][16 * kq], acc, 1.0, lane);
        double *cp = &c[16 * rb][16 * cb];
        double4_t o = blk_load_d(cp, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = o[r] - acc[r];
        blk_store_d(cp, lane, o);
    };
    if (has_prev) {
        int i = 0;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int cb = 0; cb <= rb; ++cb, ++i)
                if ((i & 3) == wave) block_syrk(sm.t, sm.x, rb, cb);
    }
    if (has_rows) {  // the own tile (and the previous panel's L of its rows) have landed meanwhile
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            *reinterpret_cast<double2_t *>(&sm.a21[sr][sc8 + e]) = (double2_t){vt[e], vt[e + 1]};
            if (has_prev) *reinterpret_cast<double2_t *>(&sm.lb[sr][sc8 + e]) = (double2_t){vb[e], vb[e + 1]};
        }
    }
    if (has_prev || has_rows) __syncthreads();
    lap(1);  // diagonal tile updated, everything staged

    // inverse of the 16 x 16 diagonal block of strip kb: lane c < 16 owns column c of X[kb, kb] (wave 3 runs it while
    // wave 0 is in the NEXT strip's chain: the block and its reciprocal diagonal are final by then, and nothing else
    // touches them)
    auto diag_inverse = [&](int kb) {
        const int c0 = 16 * kb, c = lane & 15;
        double xv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double s = 0.0;
#pragma unroll
            for (int u = 0; u < i; ++u) s = __builtin_fma(sm.t[c0 + i][c0 + u], xv[u], s);  // broadcast reads
            xv[i] = ((c == i ? 1.0 : 0.0) - s) * sm.rdiag[c0 + i];                      // 0 above the diagonal
        }
        if (lane < 16) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sm.x[c0 + i][c0 + c] = xv[i];
        }
    };

    // T[rb, cb] -= L[rb, kb] L[cb, kb]^T (16 x 16 blocks); column block 1's three blocks go to waves 1, 2, 3, the
    // others alternate between waves 1 and 2 (wave 3 also inverts the diagonal blocks)
    auto block_owner = [](int rb, int cb) { return cb == 1 ? rb : (rb == 3 && cb == 2 ? 2 : 1); };
    auto block_update = [&](int rb, int cb, int kb) {
        double *c = &sm.t[16 * rb][16 * cb];
        blk_store_d(c, lane, blk_mma_abt(&sm.t[16 * rb][16 * kb], &sm.t[16 * cb][16 * kb], blk_load_d(c, lane), -1.0, lane));
    };

    // X[rb][cb] = -X[rb][rb] * sum_{k = cb .. rb-1} L[rb][k] X[k][cb]  (cb < rb): the inner sum stays in registers
    auto x_block = [&](int rb, int cb) {
        double4_t sacc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k >= cb && k < rb) sacc = blk_mma_ab(&sm.t[16 * rb][16 * k], &sm.x[16 * k][16 * cb], sacc, lane);
        const double4_t z = {0.0, 0.0, 0.0, 0.0};
        blk_store_d(&sm.x[16 * rb][16 * cb], lane, blk_mma_a_reg(&sm.x[16 * rb][16 * rb], sacc, z, -1.0, lane));
    };
    // L21[rb][cb] = sum_{k <= cb} A21[rb][k] X[cb][k]^T (X11 is lower triangular), straight to memory
    const int r0 = k0 + PANEL * blockIdx.x;
    auto l21_block = [&](int rb, int cb) {
        double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k <= cb) acc = blk_mma_abt(&sm.a21[16 * rb][16 * k], &sm.x[16 * cb][16 * k], acc, 1.0, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) A[(size_t)(r0 + 16 * rb + (lane >> 4) + 4 * r) * ld + k0 + 16 * cb + (lane & 15)] = acc[r];
    };
    // The inverse's off-diagonal blocks and the columns of L21 do not wait for the last pivot: whatever their operands allow
    // runs on waves 1 - 3 in the shadow of the strips' pivot chains (X[kb][kb] is known one strip after strip kb):
    //   during strip 2: L21[:, 0]                         during strip 3: X[1][0], L21[:, 1]
    //   after the chain: X[2][0], X[2][1] beside the last diagonal block's inverse | X[3][0..2], L21[:, 2] | L21[:, 3]
    // -- 2.8 us after the last pivot instead of 6.3 (three levels of the inverse, then the whole of L21).

    int first_bad = PANEL;  // meaningful in wave 0
#ifdef SLK_PANEL_ROLLED
#pragma unroll 1
#else
#pragma unroll
#endif
    for (int kb = 0; kb < 4; ++kb) {
        const int c0 = 16 * kb;
        if (wave == 0) {
            // ---- strip: rows c0 .. 63, columns c0 .. c0+15; lane i = row c0 + i
            const int row = c0 + lane;
            const bool live = row < PANEL;
            double a[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = live ? sm.t[live ? row : 0][c0 + c] : 0.0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                double piv = readlane_f64(a[j], j);
                const bool bad = !(piv > 0.0);  // also catches NaN
                first_bad = (bad && first_bad == PANEL) ? c0 + j : first_bad;
                piv = bad ? 1.0 : piv;
                const double r = rsqrt_newton(piv);
                const double lj = a[j] * r;  // row j itself: piv * r = sqrt(piv)
                a[j] = lj;
                sm.rdiag[c0 + j] = r;  // uniform value, every lane stores it: no branch
#pragma unroll
                for (int c = j + 1; c < 16; ++c) a[c] = __builtin_fma(-lj, readlane_f64(lj, c), a[c]);
            }
            if (live) {
#pragma unroll
                for (int c = 0; c < 16; ++c) sm.t[row][c0 + c] = (c0 + c <= row) ? a[c] : 0.0;
            }
            lap(2);  // pivot chains
        } else if (kb == 0) {
            if (has_prev && has_rows) {  // the own tile's update (see the prologue), 16 blocks over waves 1 - 3
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (i % 3 == wave - 1) block_syrk(sm.a21, sm.lb, i >> 2, i & 3);
            }
        } else if (wave == 3) {
            diag_inverse(kb - 1);
        } else {
            // deferred updates of the strip before: the blocks that strip's successor did not need (cb > kb), done while
            // wave 0 is in this strip's chain.  Every block belongs to one wave (cb odd: wave 1, even: wave 2 -- and
            // (3, 3) to wave 1), which applies the strips' updates to it in order.
#pragma unroll
            for (int rb = kb + 1; rb < 4; ++rb)
#pragma unroll
                for (int cb = kb + 1; cb <= rb; ++cb)
                    if (block_owner(rb, cb) == wave) block_update(rb, cb, kb - 1);
            if (kb == 2 && has_rows) {  // X[0][0] is in (strip 1's side work)
                l21_block(wave - 1, 0);
                l21_block(wave + 1, 0);
            }
            if (kb == 3 && wave == 1) {  // X[1][1] is in (strip 2's side work)
                x_block(1, 0);
                // (this wave reads the block back at once: LDS operations of one wave complete in order)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (has_rows) {
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb) l21_block(rb, 1);
                }
            }
        }
        __syncthreads();
        if (kb == 0 && has_prev) {  // the previous panel's L has served: x is the inverse's from here on (zero above the diagonal)
            for (int e = t; e < PANEL * PANEL; e += 256) sm.x[e >> 6][e & 63] = 0.0;
        }
        // ---- the blocks the NEXT strip reads, (rb, kb + 1): one per wave 1 .. 3, then wave 0 goes on
        if (kb < 3) {
#pragma unroll
            for (int rb = kb + 1; rb < 4; ++rb)
                if (block_owner(rb, kb + 1) == wave) block_update(rb, kb + 1, kb);
            __syncthreads();
        }
        lap(3);  // barriers + next-strip blocks
    }
    // ---- after the last pivot
    if (wave == 3) diag_inverse(3);
    else if (wave == 0) x_block(2, 0);  // X[2][2] came with strip 3's side work, X[1][0] too
    else if (wave == 1) x_block(2, 1);
    __syncthreads();
    if (first_bad != PANEL && blockIdx.x == 0 && t == 0 && info[0] == 0) info[0] = k0 + first_bad + 1;
    if (wave < 3) x_block(3, wave);
    if (has_rows) {  // L21[:, 2] wants X[2][0 .. 2]: wave 3 two row blocks, waves 1 and 2 one each behind their X block
        if (wave == 3) {
            l21_block(0, 2);
            l21_block(1, 2);
        } else if (wave > 0) {
            l21_block(wave + 1, 2);
        }
    }
    __syncthreads();
    lap(4);  // tail: last diagonal inverse, X rows 2 and 3, L21[:, 2]

    if (blockIdx.x == 0) {
        for (int e = t; e < PANEL * PANEL; e += 256) {
            const int r = e >> 6, c = e & 63;
            X[(size_t)(k0 + r) * ld + k0 + c] = sm.x[r][c];  // exactly zero above the diagonal
        }
        return;
    }
    l21_block(wave, 3);  // the last block column wants the whole last block row of X
    lap(5);
    if (timing && lane == 0) g_panel_cycles[15] += 1;
}

// The panel step in TWO launches (the default when several layers are in flight): k_chol_panel with ONE workgroup per matrix
// factors and inverts the diagonal tile (X), then this kernel does what the tiles below it need -- the previous panel's
// update of the own tile, L21 = A21 inv(L11)^T -- and, in the same grid, the previous panel's update of the tile columns
// right of this panel (part B of k_chol_panel).  In the one-launch form every workgroup below repeats the diagonal tile's
// pivot chain to spare a launch: right for ONE factorisation alone on the GPU (the launch chain is its critical path), but
// `below` workgroups then hold a CU each (135 KB of LDS) for the 22 us of a chain they only wait for -- up to 63 CUs of a
// 4096-column factor, 171 of an 11008-column one, a quarter to two thirds of the chip that the wide kernels of the other
// layers could not use.  Here they hold it for the few us of their own work.  Same blocks, same products in the same
// order, X read back as it was written: the factor is the same bit for bit.
struct BelowSmem {
    double a21[PANEL][TP];  // this workgroup's current tile of A21
    double lb[PANEL][TP];   // the previous panel's L of that tile's rows
    double lk[PANEL][TP];   // the previous panel's L of the diagonal tile's rows
    double x[PANEL][TP];    // inv(L11)
};
__global__ __launch_bounds__(256) void k_panel_below(double *__restrict__ A, int ld, int k0, int kprev, int nb, int rest_cols,
                                                     const double *__restrict__ X, int tiles_below, int per_wg) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    A += (size_t)blockIdx.z * ld * ld;
    const int t = threadIdx.x;
    if ((int)blockIdx.x >= nb) {  // part B, as in k_chol_panel
        Tile64Smem &ts = *reinterpret_cast<Tile64Smem *>(smem_raw);
        const int nt = ld / TILE, tj0 = k0 / TILE + 1;
        int idx = (int)blockIdx.x - nb, bj = tj0;
        for (int c = 0; c < rest_cols; ++c, ++bj) {
            const int count = nt - bj;
            if (idx < count) break;
            idx -= count;
        }
        const int bi = bj + idx;
        if (bj >= tj0 + rest_cols || bi >= nt) return;
        Acc64 acc;
        acc.zero();
        const double *pa = A + ((size_t)bi * TILE + (t >> 2)) * ld + (t & 3) * 8;
        const double *pb = A + ((size_t)bj * TILE + (t >> 2)) * ld + (t & 3) * 8;
        double *pc = A + (size_t)bi * TILE * ld + (size_t)bj * TILE;
        Acc64 old;
        tile64_map(old, [&](int r, int c) { return pc[(size_t)r * ld + c]; });
        tile64_mac<false>(
            acc, ts, kprev, kprev + PANEL, [&](int k, double(&v)[8]) { load8d<true>(pa + k, v); },
            [&](int k, double(&v)[8]) { load8d<true>(pb + k, v); });
        tile64_foreach2(old, acc, [&](int r, int c, double o, double v) { pc[(size_t)r * ld + c] = o - v; });
        return;
    }
    BelowSmem &sm = *reinterpret_cast<BelowSmem *>(smem_raw);
    X += (size_t)blockIdx.z * ld * ld;
    const int lane = t & 63, wave = t >> 6;
    const bool has_prev = kprev >= 0;
    const int sr = t >> 2, sc8 = (t & 3) * 16;  // a thread carries 16 consecutive doubles of one row of every tile
    // This workgroup walks the row tiles first, first + 1, ... (`per_wg` of them): inv(L11) and the previous panel's L of the
    // diagonal tile's rows are the same for every row tile and are loaded ONCE; the next tile's two operands are in flight
    // while this one is worked on (a workgroup with one tile spends 4.4 of its 8 us waiting for its 128 KB).
    const int first = (int)blockIdx.x * per_wg, last = min(first + per_wg, tiles_below);
    double vt[16], vb[16];
    auto fetch_tile = [&](int rt) {
        const int r0 = k0 + PANEL * (rt + 1);
        const double *pt = A + (size_t)(r0 + sr) * ld + k0 + sc8;
        load8d<true>(pt, *reinterpret_cast<double(*)[8]>(&vt[0]));
        load8d<true>(pt + 8, *reinterpret_cast<double(*)[8]>(&vt[8]));
        if (has_prev) {
            const double *pl = A + (size_t)(r0 + sr) * ld + kprev + sc8;
            load8d<true>(pl, *reinterpret_cast<double(*)[8]>(&vb[0]));
            load8d<true>(pl + 8, *reinterpret_cast<double(*)[8]>(&vb[8]));
        }
    };
    {
        double vx[16], vk[16];
        const double *px = X + (size_t)(k0 + sr) * ld + k0 + sc8;
        load8d<true>(px, *reinterpret_cast<double(*)[8]>(&vx[0]));
        load8d<true>(px + 8, *reinterpret_cast<double(*)[8]>(&vx[8]));
        if (has_prev) {
            const double *pk = A + (size_t)(k0 + sr) * ld + kprev + sc8;
            load8d<true>(pk, *reinterpret_cast<double(*)[8]>(&vk[0]));
            load8d<true>(pk + 8, *reinterpret_cast<double(*)[8]>(&vk[8]));
        }
        fetch_tile(first);
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            *reinterpret_cast<double2_t *>(&sm.x[sr][sc8 + e]) = (double2_t){vx[e], vx[e + 1]};
            if (has_prev) *reinterpret_cast<double2_t *>(&sm.lk[sr][sc8 + e]) = (double2_t){vk[e], vk[e + 1]};
        }
    }
    for (int rt = first; rt < last; ++rt) {
        const int r0 = k0 + PANEL * (rt + 1);
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            *reinterpret_cast<double2_t *>(&sm.a21[sr][sc8 + e]) = (double2_t){vt[e], vt[e + 1]};
            if (has_prev) *reinterpret_cast<double2_t *>(&sm.lb[sr][sc8 + e]) = (double2_t){vb[e], vb[e + 1]};
        }
        __syncthreads();
        if (rt + 1 < last) fetch_tile(rt + 1);  // lands during the products below
        if (has_prev) {
            // the previous panel's update of the own tile: C -= L_rows L_diag^T, K = 64, accumulated from zero over ascending k
            // and subtracted once (k_chol_panel's block_syrk); wave w: row block w -- the blocks it multiplies next
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) acc = blk_mma_abt(&sm.lb[16 * wave][16 * kq], &sm.lk[16 * cb][16 * kq], acc, 1.0, lane);
                double *cp = &sm.a21[16 * wave][16 * cb];
                double4_t o = blk_load_d(cp, lane);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = o[r] - acc[r];
                blk_store_d(cp, lane, o);
            }
            // (a wave reads back only what it wrote itself: LDS operations of one wave complete in order)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        // L21[rb][cb] = sum_{k <= cb} A21[rb][k] X[cb][k]^T (X11 is lower triangular), straight to memory; wave w: row block w
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k <= cb) acc = blk_mma_abt(&sm.a21[16 * wave][16 * k], &sm.x[16 * cb][16 * k], acc, 1.0, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) A[(size_t)(r0 + 16 * wave + (lane >> 4) + 4 * r) * ld + k0 + 16 * cb + (lane & 15)] = acc[r];
        }
        if (rt + 1 < last) __syncthreads();  // every wave is through with this tile's images before the next one goes in
    }
}


// ===================================================================================================================
// Round 4: an outer block's panels in ONE launch, with no workgroup that waits while holding the chip.
//
// The panel step above is one (or two) launches per 64 columns: 64-72 dependent launches per 4096-column factor, each
// about 25 us of which 9.4 are the pivot chain itself -- the rest is the launch gap, the staging of operands nobody
// could fetch earlier, and the previous panel's update; and in the one-launch form every workgroup below the diagonal
// tile repeats the pivot chain and holds a CU for it.  Here the DIAGONAL block of an outer block (nb <= 8 tiles square)
// is factored by a CHAIN of nb workgroups, one per tile row, that hand the panels on through flags in memory:
//
//   workgroup r owns tile row r of the diagonal block:  L(r, q) for q < r, then the diagonal tile T_r.
//   It walks q = 0 .. r-1 left-looking, exactly the arithmetic of the panel kernels tile by tile:
//       C = A(r, q);  for s < q:  C -= L(r, s) L(q, s)^T      (K = 64 from zero, ONE subtraction per panel s: block_syrk)
//       L(r, q) = C inv(L_qq)^T                               (l21_block)            -- needs X_q, published by workgroup q
//       T_r -= L(r, q) L(r, q)^T                              (block_syrk on the diagonal tile, kept in registers)
//   and when L(r, r-1) is in, factors and inverts T_r (the strips of k_chol_panel, unchanged) and publishes X_r.
//   Everything that does not hang on X_q is done BEFORE it arrives: while workgroup q runs its pivot chain, workgroup r
//   applies the updates s <= q-1 to the NEXT tile C_{q+1} (registers), so that at its own turn only
//       [X_{r-1} crosses] -> L(r, r-1) -> T_r update -> pivots -> inverse -> [X_r crosses]
//   is on the critical path: no launch gap, no staging of anything but X, no update of older panels.
//   The rows BELOW the diagonal block wait for nobody: one wide launch afterwards (k_chol_rows_below) makes them all,
//   16 rows per workgroup, the row strip's own L tiles resident in LDS.  Then the outer update, as before.
//   Per 4096-column factor: 8 x (chain + rows below + outer update) = 22 launches instead of 72; the chain occupies
//   nb CUs instead of up to 64 (171 at 11008 columns).
//
// Hand-offs follow the write-through form of the CDNA4 guide (Guideline 16, R1; MI355X_MICROARCH.md, visibility, table row 1):
//   producer: every handed-off byte stored `sc1` (buffer stores, aux 16) -> every storing wave `s_waitcnt vmcnt(0)` ->
//             workgroup barrier -> ONE lane stores the flag (agent-scope relaxed atomic = `sc1` store);
//   consumer: every wave polls the flag itself (relaxed agent-scope load + s_sleep, bounded by the realtime clock), and
//             EVERY load of bytes another workgroup of this launch wrote -- and of its own earlier stores -- is an `sc1`
//             buffer load to registers (L1 is never consulted: no acquire fence needed, none of its 1.7 us).
//   One workgroup per CU (135 KB of LDS).  Workgroup r waits only for workgroups q < r of its own matrix, which the
//   dispatcher started before it; should that ever not hold, the spin gives up after 2 s and the status word says so
//   (SLK_INFO_HANDOFF_TIMEOUT) instead of hanging the GPU.
// Same blocks, same products in the same order, one rounding per panel update: U is the panel kernels' bit for bit.
constexpr int INFO_HANDOFF_TIMEOUT = 0x7fffffff;

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
constexpr int AUX_SC1 = 16;

struct GBuf {  // a matrix as a buffer resource: 16-byte accesses at 32-bit byte offsets, write-through / L1-bypassing on demand
    __amdgpu_buffer_rsrc_t rs;
    __device__ __forceinline__ GBuf(const double *base, size_t bytes)
        : rs(__builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(base), 0, (int)(bytes > 0xffffffffull ? 0xffffffffull : bytes), 0x00020000)) {}
    template <int AUX>
    __device__ __forceinline__ double2_t load2(unsigned elem) const {
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, elem * 8u, 0, AUX);
        return (double2_t){__hiloint2double((int)v[1], (int)v[0]), __hiloint2double((int)v[3], (int)v[2])};
    }
    template <int AUX>
    __device__ __forceinline__ void store2(unsigned elem, double2_t x) const {
        const u32x4_t v = {(unsigned)__double2loint(x[0]), (unsigned)__double2hiint(x[0]), (unsigned)__double2loint(x[1]), (unsigned)__double2hiint(x[1])};
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, elem * 8u, 0, AUX);
    }
};

__device__ __forceinline__ int flag_peek(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Every wave calls this for itself; true when *p >= want.  false: gave up (2 s on the 100 MHz realtime clock).
__device__ __forceinline__ bool flag_wait(const int *p, int want) {
    bool ok = flag_peek(p) >= want;
    if (!ok) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            __builtin_amdgcn_s_sleep(1);
            if (flag_peek(p) >= want) {
                ok = true;
                break;
            }
            if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) break;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // (no instruction: keeps the compiler from hoisting loads above the poll)
    return ok;
}
// after the stores of every wave: drain, meet, ONE lane signals
__device__ __forceinline__ void flag_publish(int *p, int value) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(p, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A 64 x 64 tile between memory and an LDS image of pitch TP, 16 bytes per lane and instruction: load j of thread t covers
// row 8 j + (t >> 5), doubles 2 (t & 31) .. +1 -- a wave reads two whole rows (2 x 512 contiguous bytes) per instruction.
struct TileRegs {
    double2_t v[8];
};
// (J0, J1: rows 8 J0 .. 8 J1 - 1 of the tile only)
template <int AUX, int J0 = 0, int J1 = 8>
__device__ __forceinline__ void tile_fetch(TileRegs &r, const GBuf &g, unsigned elem0, int ld) {
    const int t = threadIdx.x;
#pragma unroll
    for (int j = J0; j < J1; ++j) r.v[j] = g.load2<AUX>(elem0 + (unsigned)(8 * j + (t >> 5)) * (unsigned)ld + 2u * (t & 31));
}
template <int J0 = 0, int J1 = 8>
__device__ __forceinline__ void tile_stash(double (*img)[TP], const TileRegs &r) {
    const int t = threadIdx.x;
#pragma unroll
    for (int j = J0; j < J1; ++j) *reinterpret_cast<double2_t *>(&img[8 * j + (t >> 5)][2 * (t & 31)]) = r.v[j];
}
template <int AUX, int J0 = 0, int J1 = 8>
__device__ __forceinline__ void tile_store_from_lds(const GBuf &g, unsigned elem0, int ld, const double (*img)[TP]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int j = J0; j < J1; ++j)
        g.store2<AUX>(elem0 + (unsigned)(8 * j + (t >> 5)) * (unsigned)ld + 2u * (t & 31), *reinterpret_cast<const double2_t *>(&img[8 * j + (t >> 5)][2 * (t & 31)]));
}

// The diagonal tile in sm.t (blocks on and below the diagonal) -> L11 in sm.t, inv(L11) in sm.x (which must come in zeroed):
// the strips, deferred block updates and inverse blocks of k_chol_panel for a workgroup without rows of its own, instruction
// for instruction -- except for WHO runs the inverse's blocks and when: with no rows below to serve, wave 3 follows every diagonal
// block's inverse with the off-diagonal blocks that have become possible (X[1][0] during strip 2; X[2][1], X[2][0] during strip
// 3), and after the last pivot the inner sums of row 3 run beside the last diagonal block's inverse: one product and a barrier
// stand between the last pivot and the end instead of two levels of the inverse.  Same blocks, same operations.
// All 256 threads; returns (wave 0) the first column with a non-positive pivot, PANEL if none.
template <class Early>
__device__ __forceinline__ int diag_tile_factor(PanelSmem &sm, int lane, int wave, Early rows_0_and_1_are_in) {
    auto diag_inverse = [&](int kb) {
        const int c0 = 16 * kb, c = lane & 15;
        double xv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double s = 0.0;
#pragma unroll
            for (int u = 0; u < i; ++u) s = __builtin_fma(sm.t[c0 + i][c0 + u], xv[u], s);
            xv[i] = ((c == i ? 1.0 : 0.0) - s) * sm.rdiag[c0 + i];
        }
        if (lane < 16) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sm.x[c0 + i][c0 + c] = xv[i];
        }
    };
    auto block_owner = [](int rb, int cb) { return cb == 1 ? rb : (rb == 3 && cb == 2 ? 2 : 1); };
    auto block_update = [&](int rb, int cb, int kb) {
        double *c = &sm.t[16 * rb][16 * cb];
        blk_store_d(c, lane, blk_mma_abt(&sm.t[16 * rb][16 * kb], &sm.t[16 * cb][16 * kb], blk_load_d(c, lane), -1.0, lane));
    };
    auto x_inner = [&](int rb, int cb) {  // sum_{k = cb .. rb-1} L[rb][k] X[k][cb]
        double4_t sacc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k >= cb && k < rb) sacc = blk_mma_ab(&sm.t[16 * rb][16 * k], &sm.x[16 * k][16 * cb], sacc, lane);
        return sacc;
    };
    auto x_finish = [&](int rb, int cb, double4_t sacc) {  // X[rb][cb] = -X[rb][rb] * that
        const double4_t z = {0.0, 0.0, 0.0, 0.0};
        blk_store_d(&sm.x[16 * rb][16 * cb], lane, blk_mma_a_reg(&sm.x[16 * rb][16 * rb], sacc, z, -1.0, lane));
    };
    auto x_block = [&](int rb, int cb) { x_finish(rb, cb, x_inner(rb, cb)); };
    // (a wave that reads back LDS words it has just written: its LDS operations complete in order)
    auto own_writes = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
    int first_bad = PANEL;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const int c0 = 16 * kb;
        if (wave == 0) {
            const int row = c0 + lane;
            const bool live = row < PANEL;
            double a[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = live ? sm.t[live ? row : 0][c0 + c] : 0.0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                double piv = readlane_f64(a[j], j);
                const bool bad = !(piv > 0.0);
                first_bad = (bad && first_bad == PANEL) ? c0 + j : first_bad;
                piv = bad ? 1.0 : piv;
                const double r = rsqrt_newton(piv);
                const double lj = a[j] * r;
                a[j] = lj;
                sm.rdiag[c0 + j] = r;
#pragma unroll
                for (int c = j + 1; c < 16; ++c) a[c] = __builtin_fma(-lj, readlane_f64(lj, c), a[c]);
            }
            if (live) {
#pragma unroll
                for (int c = 0; c < 16; ++c) sm.t[row][c0 + c] = (c0 + c <= row) ? a[c] : 0.0;
            }
        } else if (kb == 0) {
        } else if (wave == 3) {
            diag_inverse(kb - 1);
            own_writes();
            if (kb == 2) x_block(1, 0);  // X[0][0] came with strip 1, X[1][1] just now
            if (kb == 3) {
                x_block(2, 1);
                x_block(2, 0);  // wants X[1][0] (strip 2's side work) and X[0][0]
            }
        } else {
#pragma unroll
            for (int rb = kb + 1; rb < 4; ++rb)
#pragma unroll
                for (int cb = kb + 1; cb <= rb; ++cb)
                    if (block_owner(rb, cb) == wave) block_update(rb, cb, kb - 1);
            // rows 0 and 1 of X are complete when strip 3 starts (X[1][0] came with strip 2's side work): they go out NOW, by
            // the two waves that have nothing else left to do, so that the next workgroup of the chain multiplies by them
            // while the last strip's pivots run here
            if (kb == 3) rows_0_and_1_are_in(wave - 1);
        }
        __syncthreads();
        if (kb < 3) {
#pragma unroll
            for (int rb = kb + 1; rb < 4; ++rb)
                if (block_owner(rb, kb + 1) == wave) block_update(rb, kb + 1, kb);
            __syncthreads();
        }
    }
    double4_t row3 = {0.0, 0.0, 0.0, 0.0};
    if (wave == 3) diag_inverse(3);
    else row3 = x_inner(3, wave);  // rows 0 .. 2 of X are complete
    __syncthreads();
    if (wave < 3) x_finish(3, wave, row3);
    __syncthreads();
    return first_bad;
}

// flags of one matrix: [0, nt) xready[p] = 1 when row blocks 0 and 1 of inv(L_pp) of panel p are in X, 2 when all of it is; [nt, 2 nt) rowdone[p] = how many L tiles of
// tile row p (inside its outer block) are in A.  Zeroed by k_clear_info before the factorisation starts.
// BELOW = true: the same walk for the tile rows BELOW the diagonal block, launched after the chain (every X_q and every L of
// the block are in memory: nothing to wait for, nothing to publish, no diagonal tile): workgroup x owns tile row nb + x and
// makes L(r, q) for all nb panels.  The wide form of k_chol_rows_below -- 64 rows per workgroup instead of 16: a quarter of
// the workgroups, each step 64 MFMAs a wave for two staged operands instead of 16 for one -- for factorisations that share
// the chip with other layers' kernels (what it costs the chip counts there, not how long a launch lasts).
template <bool BELOW>
__global__ __launch_bounds__(256) void k_chol_chain(double *__restrict__ Aall, int ld, int K0, int nb, double *__restrict__ Xall,
                                                    int *__restrict__ info, int *__restrict__ flags_all, int dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PanelSmem &sm = *reinterpret_cast<PanelSmem *>(smem_raw);
    // roles of the four images while this workgroup walks its row:  t: L(q', s) operand   x: X_q
    //                                                                a21: C_q (complete, waits for X_q)   lb: L(r, s) operand, L(r, q)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r = BELOW ? nb + (int)blockIdx.x : (int)blockIdx.x, nt = ld / PANEL, p0 = K0 / PANEL;
    const int rq = BELOW ? nb : r;  // panels this row walks
    const size_t mat = (size_t)ld * ld;
    double *A = Aall + (size_t)blockIdx.z * mat;
    const GBuf ga(A, mat * sizeof(double)), gx(Xall + (size_t)blockIdx.z * mat, mat * sizeof(double));
    info += blockIdx.z;
    int *xready = flags_all + (size_t)blockIdx.z * 2 * nt, *rowdone = xready + nt;
    const int row0 = K0 + PANEL * r;
    auto elem = [&](int tr, int tc) { return (unsigned)(K0 + PANEL * tr) * (unsigned)ld + (unsigned)(K0 + PANEL * tc); };  // tile (tr, tc) of the block
    bool alive = true;  // false once a wait gave up: the workgroup goes on without waiting (finite, results void, status says so)
    if (t == 0) sm.early_count = 0;  // (read after many barriers)
    // (measurement, SLK_WIN_DBG=8: where wave 0 of the workgroups r >= 1 spends the cycles between seeing X_{r-1} and
    // publishing X_r -- the chain's critical path; g_panel_cycles, read by slk_probe_panel_cycles / tools/micro_panel.py)
    const bool timing = !BELOW && dbg && r > 0 && blockIdx.z == 0 && wave == 0;
    long long tmark = 0;
    auto lap = [&](int slot) {
        if (timing) {
            const long long now = (long long)__builtin_readcyclecounter();
            if (lane == 0 && slot >= 0) atomicAdd(reinterpret_cast<unsigned long long *>(&g_panel_cycles[slot]), (unsigned long long)(now - tmark));
            tmark = now;
        }
    };

    // T_r, blocks on and below the diagonal, in registers (D layout): the i-th such block belongs to wave i & 3 (block_syrk's split)
    double4_t tp[3];
    if constexpr (!BELOW) {
        int i = 0;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int cb = 0; cb <= rb; ++cb, ++i)
                if ((i & 3) == wave) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int rr = 16 * rb + (lane >> 4) + 4 * j, cc = 16 * cb + (lane & 15);
                        // (workgroup 0 starts from the lower triangle as k_chol_panel's first panel of a block does)
                        tp[i >> 2][j] = (r > 0 || cc <= rr) ? A[(size_t)(row0 + rr) * ld + row0 + cc] : 0.0;
                    }
                }
    }
    // products of one panel's update: dst block (wave's row block, column block cb) -= lb[row block] t[cb]^T, K = 64 from zero
    auto update_cn = [&](double4_t(&cn)[4]) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) acc = blk_mma_abt(&sm.lb[16 * wave][16 * kq], &sm.t[16 * cb][16 * kq], acc, 1.0, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) cn[cb][j] = cn[cb][j] - acc[j];
        }
    };

    // L(r, 0 .. count-1) are in memory: this workgroup's own later loads may see them; the other rows of the chain are told
    auto row_done = [&](int count) {
        if constexpr (BELOW) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        } else {
            flag_publish(&rowdone[p0 + r], count);
        }
    };
    if (rq > 0) {
        TileRegs ra, rb_;
        double4_t cn[4];  // C_{q+1} in the making: this wave's row block
        // C_0 = A(r, 0) as it stands
        tile_fetch<0>(ra, ga, elem(r, 0), ld);
        tile_stash(sm.a21, ra);
        for (int q = 0; q < rq; ++q) {
            const bool more = q + 1 < rq;
            // (1) while workgroup q is in its pivot chain: the updates s < q of the NEXT tile, C_{q+1} = A(r, q+1) - ...
            if (more) {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        cn[cb][j] = A[(size_t)(row0 + 16 * wave + (lane >> 4) + 4 * j) * ld + K0 + PANEL * (q + 1) + 16 * cb + (lane & 15)];
                if (q > 0) {
                    if (!BELOW) alive = alive && flag_wait(&rowdone[p0 + q + 1], 1);
                    tile_fetch<AUX_SC1>(ra, ga, elem(r, 0), ld);
                    tile_fetch<AUX_SC1>(rb_, ga, elem(q + 1, 0), ld);
                }
                for (int s = 0; s < q; ++s) {
                    __syncthreads();  // the images' readers of the step before are through
                    tile_stash(sm.lb, ra);
                    tile_stash(sm.t, rb_);
                    __syncthreads();
                    if (s + 1 < q) {  // the next pair is in flight during the products
                        if (!BELOW && alive) alive = flag_wait(&rowdone[p0 + q + 1], s + 2);
                        tile_fetch<AUX_SC1>(ra, ga, elem(r, s + 1), ld);
                        tile_fetch<AUX_SC1>(rb_, ga, elem(q + 1, s + 1), ld);
                    }
                    update_cn(cn);
                }
            }
            // (2) X_q crosses in two parts -- its row blocks 0 and 1 while workgroup q is still in its last strip's pivots, then
            //     row blocks 2 and 3: L(r, q) = C_q inv(L_qq)^T column block by column block (block cb wants X's row block cb
            //     only), and of T_r's update by it, sum_kq L[rb][kq] L[cb][kq]^T, the terms kq = 0, 1 -- same chain of additions,
            //     kq ascending -- before the second part has arrived
            const bool last = !BELOW && q + 1 == r;
            auto l_block = [&](int cb) {
                double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k <= cb) acc = blk_mma_abt(&sm.a21[16 * wave][16 * k], &sm.x[16 * cb][16 * k], acc, 1.0, lane);
                blk_store_d(&sm.lb[16 * wave][16 * cb], lane, acc);
            };
            double4_t tacc[3];
            auto t_terms = [&](int kq0, int kq1) {
                int i = 0;
#pragma unroll
                for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                    for (int cb = 0; cb <= rb; ++cb, ++i)
                        if ((i & 3) == wave) {
                            double4_t acc = kq0 == 0 ? (double4_t){0.0, 0.0, 0.0, 0.0} : tacc[i >> 2];
#pragma unroll
                            for (int kq = 0; kq < 4; ++kq)
                                if (kq >= kq0 && kq < kq1) acc = blk_mma_abt(&sm.lb[16 * rb][16 * kq], &sm.lb[16 * cb][16 * kq], acc, 1.0, lane);
                            tacc[i >> 2] = acc;
                        }
            };
            if (!BELOW && alive) alive = flag_wait(&xready[p0 + q], 1);
            tile_fetch<AUX_SC1, 0, 4>(ra, gx, elem(q, q), ld);
            __syncthreads();
            tile_stash<0, 4>(sm.x, ra);
            __syncthreads();
            l_block(0);
            l_block(1);
            __syncthreads();
            if (!BELOW) t_terms(0, 2);
            if (!BELOW && alive) alive = flag_wait(&xready[p0 + q], 2);
            if (last) lap(-1);  // the clock of the critical path starts when the LAST part of X_{r-1} is seen
            tile_fetch<AUX_SC1, 4, 8>(ra, gx, elem(q, q), ld);
            tile_stash<4, 8>(sm.x, ra);  // (rows 32 .. 63 of the image: nobody has read them since the barriers above)
            __syncthreads();
            if (last) lap(0);  // row blocks 2 and 3 of X_{r-1} in LDS
            l_block(2);
            l_block(3);
            __syncthreads();
            if (last) lap(1);  // the last two column blocks of L(r, r-1)
            // L(r, q) to memory (write-through); the last terms of T_r's update meanwhile
            tile_store_from_lds<AUX_SC1>(ga, elem(r, q), ld, sm.lb);
            if constexpr (!BELOW) {
                t_terms(2, 4);
                int i = 0;
#pragma unroll
                for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                    for (int cb = 0; cb <= rb; ++cb, ++i)
                        if ((i & 3) == wave) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) tp[i >> 2][j] = tp[i >> 2][j] - tacc[i >> 2][j];
                            // the last panel's update completes T_r: straight into its image (t has served as an operand for
                            // the last time: nothing reads it in this iteration)
                            if (last) blk_store_d(&sm.t[16 * rb][16 * cb], lane, tp[i >> 2]);
                        }
            }
            if (more) {
                // (3) the last update of C_{q+1}: panel q's, with the L(r, q) just made (in lb) and L(q+1, q) from its owner
                if (!BELOW && alive) alive = flag_wait(&rowdone[p0 + q + 1], q + 1);
                tile_fetch<AUX_SC1>(rb_, ga, elem(q + 1, q), ld);
                tile_stash(sm.t, rb_);  // (t's readers finished before the barriers of (2))
                row_done(q + 1);  // (its barrier also closes the stash)
                update_cn(cn);
                // C_{q+1} is complete: into a21, each wave its own row block (the only one it reads back)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) blk_store_d(&sm.a21[16 * wave][16 * cb], lane, cn[cb]);
            } else {
                lap(2);  // the update of T_r
                row_done(q + 1);
                lap(3);  // L(r, r-1) drained, row flag
            }
        }
        // (flag_publish's barrier stands between the stores of T_r above and its readers below; x holds X_{r-1}, whose blocks
        // above the diagonal are exactly zero -- what the inverse below wants there -- and whose other blocks it overwrites)
    } else if (!BELOW) {
        // ---- workgroup 0: T_0 -> its image, x cleared
        int i = 0;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int cb = 0; cb <= rb; ++cb, ++i)
                if ((i & 3) == wave) blk_store_d(&sm.t[16 * rb][16 * cb], lane, tp[i >> 2]);
        for (int e = t; e < PANEL * PANEL; e += 256) sm.x[e >> 6][e & 63] = 0.0;
        __syncthreads();
    }
    if constexpr (BELOW) return;
    // ---- the turn: factor, invert, publish
    lap(4);  // T_r into its image
    // (the early publication: waves 1 and 2 store a row block of X each during the last strip; each drains its own stores, and the
    // one that is second to say so in LDS sets the flag to 1 -- Guideline 16's form with a counter in LDS instead of a workgroup
    // barrier, which the wave in the pivot chain must not be held up by)
    const int first_bad = diag_tile_factor(sm, lane, wave, [&](int rowblock) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = 16 * rowblock + 2 * j + (lane >> 5);
            gx.store2<AUX_SC1>(elem(r, r) + (unsigned)row * (unsigned)ld + 2u * (lane & 31), *reinterpret_cast<const double2_t *>(&sm.x[row][2 * (lane & 31)]));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int arrived = 0;
        if (lane == 0) arrived = __hip_atomic_fetch_add(&sm.early_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        arrived = __builtin_amdgcn_readfirstlane(arrived);
        if (arrived == 1) {
            if (lane == 0) __hip_atomic_store(&xready[p0 + r], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (this word is stored again, = 2, behind later barriers: in that order)
        }
    });
    lap(5);  // strips, inverse
    if (t == 0) {
        int expect = 0;
        if (!alive) {
            __hip_atomic_store(info, INFO_HANDOFF_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (first_bad != PANEL) {
            // (the workgroups factor their tiles one after the other: the first to report is the smallest column)
            __hip_atomic_compare_exchange_strong(info, &expect, row0 + first_bad + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    tile_store_from_lds<AUX_SC1, 4, 8>(gx, elem(r, r), ld, sm.x);  // row blocks 2 and 3 (the tile is exactly zero above the diagonal)
    flag_publish(&xready[p0 + r], 2);
    lap(6);  // X_r's last row block stored, drained, published
    if (timing && lane == 0) atomicAdd(reinterpret_cast<unsigned long long *>(&g_panel_cycles[15]), 1ull);
}

// The rows below an outer block's diagonal block, after k_chol_chain: for every 16-row strip, panel by panel (left-looking,
// the tile arithmetic of the panel kernels):  C = A(rows, q) - sum_{s < q} L(rows, s) L(q, s)^T  (one subtraction per s),
// L(rows, q) = C inv(L_qq)^T.  The strip's own L tiles stay in LDS; the 64 x 64 operands L(q, s), X_q stream through one
// image, the next one in flight during the products.  Wave w owns column block w of the strip's current tile.
// Measured and dropped (4096 columns, us per launch; this form: 45): two images and two register sets, one barrier per step
// and operands two steps ahead: 53; the operands straight from L2 into the MFMA's B layout, no image and no barrier between
// panels: 74 (a wave instruction then touches sixteen 128-byte lines for 512 bytes).  A step is 16 MFMAs per wave for 32 KB
// staged: with 16 rows per workgroup the kernel runs at the rate of its operand traffic; what it costs the chip (about half
// of it for 45 us, seven times per 4096-column factor) is what the panel kernels' waiting workgroups cost before.
struct RowsBelowSmem {
    double own[8][16][TP];  // L(rows, s), s < nb
    double bt[PANEL][TP];   // L(q, s) or X_q
    double c[16][TP];       // C before the triangular product
};
__global__ __launch_bounds__(256) void k_chol_rows_below(double *__restrict__ Aall, int ld, int K0, int nb, const double *__restrict__ Xall) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    RowsBelowSmem &sm = *reinterpret_cast<RowsBelowSmem *>(smem_raw);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const size_t mat = (size_t)ld * ld;
    double *A = Aall + (size_t)blockIdx.z * mat;
    const GBuf ga(A, mat * sizeof(double)), gx(Xall + (size_t)blockIdx.z * mat, mat * sizeof(double));
    const int rows0 = K0 + PANEL * nb + 16 * blockIdx.x;
    auto tile_elem = [&](int tr, int tc) { return (unsigned)(K0 + PANEL * tr) * (unsigned)ld + (unsigned)(K0 + PANEL * tc); };
    auto load_c = [&](int q) {  // this wave's column block of A(rows, q), D layout
        const double *p = A + (size_t)(rows0 + (lane >> 4)) * ld + K0 + PANEL * q + 16 * wave + (lane & 15);
        return (double4_t){p[0], p[(size_t)4 * ld], p[(size_t)8 * ld], p[(size_t)12 * ld]};
    };
    TileRegs nx;
    tile_fetch<0>(nx, gx, tile_elem(0, 0), ld);  // the first operand: X_0
    double4_t c = load_c(0), cnext = load_c(nb > 1 ? 1 : 0);  // (the tile after next is fetched a whole panel ahead)
    for (int q = 0; q < nb; ++q) {
        for (int s = 0; s <= q; ++s) {
            const bool tri = s == q;  // the triangular product with X_q closes the panel
            __syncthreads();          // the readers of bt (and of c) are through
            tile_stash(sm.bt, nx);
            if (tri) blk_store_d(&sm.c[0][16 * wave], lane, c);
            __syncthreads();
            // next operand in flight: L(q, s+1), or X_q, or the next panel's first (L(q+1, 0))
            if (!tri) {
                if (s + 1 < q) tile_fetch<0>(nx, ga, tile_elem(q, s + 1), ld);
                else tile_fetch<0>(nx, gx, tile_elem(q, q), ld);
            } else if (q + 1 < nb) {
                tile_fetch<0>(nx, ga, tile_elem(q + 1, 0), ld);
            }
            if (!tri) {
                double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) acc = blk_mma_abt(&sm.own[s][0][16 * kq], &sm.bt[16 * wave][16 * kq], acc, 1.0, lane);
                c = c - acc;
            } else {
                double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k <= wave) acc = blk_mma_abt(&sm.c[0][16 * k], &sm.bt[16 * wave][16 * k], acc, 1.0, lane);
#pragma unroll
                for (int j = 0; j < 4; ++j) A[(size_t)(rows0 + (lane >> 4) + 4 * j) * ld + K0 + PANEL * q + 16 * wave + (lane & 15)] = acc[j];
                blk_store_d(&sm.own[q][0][16 * wave], lane, acc);
                c = cnext;
                if (q + 2 < nb) cnext = load_c(q + 2);
            }
        }
    }
}

// C[bi][bj] -= L[bi][ka:kb] * L[bj][ka:kb]^T for tiles bi in [ti0, ti1), bj in [tj0, tj1), bj <= bi.
__global__ __launch_bounds__(256) void k_syrk_tiles(double *__restrict__ A, int ld, int ti0, int tj0, int ka,
                                                    int kb) {
    __shared__ __attribute__((aligned(16))) Tile64Smem sm;
    A += (size_t)blockIdx.z * ld * ld;
    const int bi = ti0 + blockIdx.y, bj = tj0 + blockIdx.x;
    if (bj > bi) return;
    Acc64 acc;
    acc.zero();
    const int t = threadIdx.x;
    const double *pa = A + ((size_t)bi * TILE + (t >> 2)) * ld + (t & 3) * 8;
    const double *pb = A + ((size_t)bj * TILE + (t >> 2)) * ld + (t & 3) * 8;
    // the tile of C is fetched NOW, not when the products are done: its trip to memory then hides behind them instead of
    // sitting at the tail of this K = 64 update (at most ~240 workgroups: registers are free)
    double *pc = A + (size_t)bi * TILE * ld + (size_t)bj * TILE;
    Acc64 old;
    tile64_map(old, [&](int r, int c) { return pc[(size_t)r * ld + c]; });
    tile64_mac<false>(
        acc, sm, ka, kb, [&](int k0, double(&v)[8]) { load8d<true>(pa + k0, v); },
        [&](int k0, double(&v)[8]) { load8d<true>(pb + k0, v); });
    tile64_foreach2(old, acc, [&](int r, int c, double o, double v) { pc[(size_t)r * ld + c] = o - v; });
}

// The same update over the whole lower triangle of tiles [t0, nt) x [t0, nt), one-dimensional grid in
// an XCD-aware order: workgroup i runs on XCD i % 8 (own L2 each), so XCD x takes the x-th eighth of
// the row-major list of tiles -- a band of a few tile rows whose A panels stay in its L2 while the
// column panels stream through once per band, instead of every XCD streaming every panel.  At
// K = 256 the update moves 6.5 flop per operand byte: it runs at the speed of the L2 misses.
__global__ __launch_bounds__(256) void k_syrk_triangle(double *__restrict__ A, int ld, int t0, int m, int ka, int kb) {
    __shared__ __attribute__((aligned(16))) Tile64Smem sm;
    A += (size_t)blockIdx.z * ld * ld;
    const int total = m * (m + 1) / 2, per_xcd = (total + 7) / 8;
    const int lin = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (lin >= total) return;
    int row = (int)((sqrtf(8.0f * (float)lin + 1.0f) - 1.0f) * 0.5f);
    while (row * (row + 1) / 2 > lin) --row;
    while ((row + 1) * (row + 2) / 2 <= lin) ++row;
    const int bi = t0 + row, bj = t0 + lin - row * (row + 1) / 2;
    Acc64 acc;
    acc.zero();
    const int t = threadIdx.x;
    const double *pa = A + ((size_t)bi * TILE + (t >> 2)) * ld + (t & 3) * 8;
    const double *pb = A + ((size_t)bj * TILE + (t >> 2)) * ld + (t & 3) * 8;
    tile64_mac<false>(
        acc, sm, ka, kb, [&](int k0, double(&v)[8]) { load8d<true>(pa + k0, v); },
        [&](int k0, double(&v)[8]) { load8d<true>(pb + k0, v); });
    // (fetching the tile of C before the products, as the inner update does, costs 48 registers: 3 waves per SIMD
    // instead of 4, which this chip-wide kernel needs more than the shorter tail)
    // The tile of C is fetched AFTER the products, but ALL of it before the first store: written as `pc[...] -= v` element by
    // element the compiler cannot tell the sixteen addresses of a thread apart (ld is a run-time value) and emitted sixteen
    // load -> s_waitcnt vmcnt(0) -> store round trips one behind the other -- as long as the products of a K = 512 tile.
    double *pc = A + (size_t)bi * TILE * ld + (size_t)bj * TILE;
    Acc64 old;
    tile64_map(old, [&](int r, int c) { return pc[(size_t)r * ld + c]; });
    tile64_foreach2(old, acc, [&](int r, int c, double o, double v) { pc[(size_t)r * ld + c] = o - v; });
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
                                                     double *__restrict__ S, int ld, int nt, int s, int node0) {
    __shared__ __attribute__((aligned(16))) Tile64Smem sm;
    L += (size_t)blockIdx.z * ld * ld;
    X += (size_t)blockIdx.z * ld * ld;
    S += (size_t)blockIdx.z * ld * ld;
    // block id -> (node, slow, fast): `fast` runs over the s tiles of the balanced direction
    const int id = blockIdx.x;
    const int per_node = s * s, node = node0 + id / per_node, in_node = id % per_node;  // (node0: a launch may cover a range of nodes)
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
        const double *pb = X + (size_t)(t >> 3) * ld + (size_t)bj * TILE + (t & 7) * 2;
        tile64_mac<true>(
            acc, sm, bj * TILE, mid * TILE, [&](int k0, double(&v)[8]) { load8d<true>(pa + k0, v); },
            [&](int k0, double(&v)[8]) { load8d_cols(pb + (size_t)k0 * ld, v); });
        double *pc = S + (size_t)bi * TILE * ld + (size_t)bj * TILE;
        tile64_foreach(acc, [&](int r, int c, double v) { pc[(size_t)r * ld + c] = v; });
    } else {
        const double *pa = X + ((size_t)bi * TILE + (t >> 2)) * ld + (t & 3) * 8;
        const double *pb = S + (size_t)(t >> 3) * ld + (size_t)bj * TILE + (t & 7) * 2;
        tile64_mac<true>(
            acc, sm, mid * TILE, (bi + 1) * TILE, [&](int k0, double(&v)[8]) { load8d<true>(pa + k0, v); },
            [&](int k0, double(&v)[8]) { load8d_cols(pb + (size_t)k0 * ld, v); });
        double *pc = X + (size_t)bi * TILE * ld + (size_t)bj * TILE;
        tile64_foreach(acc, [&](int r, int c, double v) { pc[(size_t)r * ld + c] = -v; });
    }
}

// U[i][j] = X[n-1-i][n-1-j] for j >= i, 0 below the diagonal.
__global__ __launch_bounds__(256) void k_flip_out(const double *__restrict__ X, int ld, int n,
                                                  double *__restrict__ U) {
    X += (size_t)blockIdx.z * ld * ld;
    U += (size_t)blockIdx.z * n * n;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const double *src = X + (size_t)(n - 1 - i) * ld;
        double *dst = U + (size_t)i * n;
        for (int j = threadIdx.x; j < n; j += blockDim.x) dst[j] = (j >= i) ? src[n - 1 - j] : 0.0;
    }
}

// status words and hand-off flags of a (batched) factorisation, zeroed before it starts
__global__ void k_clear_info(int *info, int *flags, int per_matrix) {
    if (threadIdx.x == 0) info[blockIdx.x] = 0;
    if (flags)
        for (int i = threadIdx.x; i < per_matrix; i += blockDim.x) flags[(size_t)blockIdx.x * per_matrix + i] = 0;
}

}  // namespace slk

using namespace slk;

static int chol_inverse_impl(double *A, int batch, int n, double *U, int *info, void *workspace, size_t ws_bytes,
                             slk_stream_t stream, bool want_lookahead) {
    const int ld = slk_factor_ld(n);
    const int nt = ld / TILE;
    const unsigned B = (unsigned)batch;
    const double Bd = (double)batch;
    Arena ws(workspace, ws_bytes);
    double *X = ws.take<double>((size_t)ld * ld * batch);
    double *S = ws.take<double>((size_t)ld * ld * batch);
    int *flags = ws.take<int>((size_t)batch * 2 * nt);  // k_chol_chain's hand-off words
    if (!X || !S || !flags) {
        set_error("workspace too small for %d factorisation(s) of %d x %d", batch, n, n);
        return SLK_E_WS;
    }
    hipStream_t s = as_stream(stream);
    // Outer blocks of 256 columns, of 512 from 4096 columns up: the outer update then runs at K = 512 (0.49 against 0.63 ms
    // per 4096-column factor alone) while the inner updates it displaces ride in the panel launches' grids.  Measured on whole
    // streams (alternating runs): the headline batch no difference (25.8 ms per step either way), OPT-350M 60.9 -> 59.3 and
    // BLOOM-560M 69.65 -> 69.0 (their 4096-column factorisations serve 1024 rows each), 11008 columns +2.5 %; 3072 columns
    // and fewer: none or a loss (OPT-125M 14.11 -> 14.14 / 14.19 with 512 from 3072 / 2048 columns).
    const int OUTER = n >= 4096 ? OUTER_LARGE : OUTER_SMALL;
    SLK_LDS_OPT_IN(k_chol_panel, sizeof(PanelSmem));
    SLK_RUN_W("clear_info", 0, 4, 1, s, k_clear_info<<<B, 64, 0, s>>>(info, flags, 2 * nt));

    // Look-ahead over the outer blocks (below): a helper stream and events.  For ONE factorisation at a time it takes
    // 0.6 ms off a 4096-column layer (6.2 -> 5.6 ms end to end); with several factorisations in flight on streams of
    // their own (sleekit_amd.dist) the chip is full already and the split launches and event joins only cost
    // (4850 -> 3560 Mweights/s on one rank whatever GPU_MAX_HW_QUEUES is, 4.7 -> 7.3 ms per step of a rank of 8).  So it
    // is a switch, off by default, which the single-layer API turns on (sleekit_amd/engine.py: quantize_layer).
    Helper helper{};
    const bool lookahead = (want_lookahead || opt(OPT_LOOKAHEAD)) && nt >= LOOKAHEAD_MIN_TILES + OUTER / TILE && nt / (OUTER / TILE) < HELPER_EVENTS / 3;
    if (lookahead) SLK_HIP(helper_for(s, 3 * (nt / (OUTER / TILE) + 1), &helper));  // up to three events per outer block
    // The panel step in one launch (this factorisation is alone on the GPU: its launch chain is the critical path) or in two
    // (others are in flight: the tiles below the diagonal tile must not hold their CUs while its pivot chain runs)
    // Measured on whole streams (ms per step, one launch against two): batches of small matrices gain (OPT-125M 14.50 -> 14.06,
    // OPT-350M 61.3 -> 61.0, BLOOM-560M 70.1 -> 69.7) and so do 11008-column factors (512.6 -> 503.8, up to 171 workgroups
    // below); single 4096-column factors do not (headline 25.45 -> 26.7 with two factor streams, 25.55 with three: the longer
    // launch chain starves the loop stream; OPT-350M 61.6 -> 64.8 when its 4096-column layers split too).  Hence the rule;
    // slk_set_option("panel_split", 1 | 2) forces two launches | one.
    // Since round 4 the default is neither: an outer block's panels are ONE launch of a chain of workgroups that hand the
    // panels on through flags (k_chol_chain), plus one wide launch for the rows below (k_chol_rows_below) -- same U bit for
    // bit; panel_split = 1 | 2 still force the panel kernels (3 = the chain, explicitly).
    const int sp = opt(OPT_PANEL_SPLIT);
    const bool chain = sp == 0 || sp == 3;
    const bool split = !chain && !want_lookahead && sp != 2 && (sp == 1 || batch > 1 || n >= 8192);
    if (split) SLK_LDS_OPT_IN(k_panel_below, sizeof(BelowSmem));
    if (chain) {
        SLK_LDS_OPT_IN(k_chol_chain<false>, sizeof(PanelSmem));
        SLK_LDS_OPT_IN(k_chol_chain<true>, sizeof(PanelSmem));
        SLK_LDS_OPT_IN(k_chol_rows_below, sizeof(RowsBelowSmem));
    }
    // The inverse, level by level: level `lvl` merges the inverted halves of nodes of 2 lvl tiles (k_trtri_level).  A node can
    // be merged as soon as its tiles are factored, so a factorisation that looks ahead hands the nodes that an outer block
    // completes to the helper stream at once (below); what is left -- without look-ahead: everything -- follows the last block.
    // inv_done[i]: nodes of level 2^i already launched.  Same kernels, same nodes, same K ranges: same bits.
    // A node's FIRST stage (S = L[B, A] X[A, A]) only wants its left half inverted and the columns of that half factored:
    // it is ready when the factorisation has passed the node's MIDDLE, half a node before the second stage (X[B, A] = -X[B, B]
    // S), which wants the right half's inverse.  inv_done[stage][i]: nodes of level 2^i whose stage has been launched.
    int inv_done[2][16] = {{0}, {0}};
    auto inverse_nodes = [&](int tiles_done, hipStream_t st) -> int {
        int li = 0;
        for (int lvl = 1; lvl < nt; lvl *= 2, ++li) {
            const int all = (nt + 2 * lvl - 1) / (2 * lvl);
            for (int stage = 0; stage < 2; ++stage) {
                int ready;
                if (tiles_done >= nt) ready = all;
                else if (stage == 0) ready = tiles_done >= lvl ? (tiles_done - lvl) / (2 * lvl) + 1 : 0;  // mid <= tiles_done
                else ready = tiles_done / (2 * lvl);                                                       // hi <= tiles_done
                if (ready > all) ready = all;
                const int first = inv_done[stage][li], count = ready - first;
                if (count <= 0) continue;
                inv_done[stage][li] = ready;
                // work of these nodes: tiles (bi in B, bj in A) with their triangular K ranges
                double fl = 0, tiles = 0;
                for (int nd = first; nd < ready; ++nd) {
                    const int lo = nd * 2 * lvl, mid = lo + lvl, hi = lo + 2 * lvl < nt ? lo + 2 * lvl : nt;
                    for (int bi = mid; bi < hi; ++bi)
                        for (int bj = lo; bj < mid; ++bj) {
                            fl += 2.0 * 64 * 64 * 64 * (stage == 0 ? mid - bj : bi + 1 - mid);
                            tiles += 1;
                        }
                }
                if (tiles == 0) continue;  // (a last node without a second half)
                dim3 grid(count * lvl * lvl, 1, B);
                if (stage == 0)
                    SLK_RUN_W("trtri_stage0", Bd * fl, Bd * (fl / 64 * 8.0 / 2 + tiles * 8.0 * 64 * 64), tiles * batch, st,
                              k_trtri_level<0><<<grid, 256, 0, st>>>(A, X, S, ld, nt, lvl, first));
                else
                    SLK_RUN_W("trtri_stage1", Bd * fl, Bd * (fl / 64 * 8.0 / 2 + tiles * 8.0 * 64 * 64), tiles * batch, st,
                              k_trtri_level<1><<<grid, 256, 0, st>>>(A, X, S, ld, nt, lvl, first));
            }
        }
        return SLK_OK;
    };
    const int EV = chain ? 3 : 2;  // helper events per outer block (chain: + the inverse's nodes behind the rest of the update)
    int block = 0, forked = -1;  // forked: the last block whose rest went to the helper and has not been joined
    int inverse_on_helper = -1;  // the last block behind whose rest the helper also runs nodes of the inverse
    for (int K0 = 0; K0 < ld; K0 += OUTER) {
        const int K1 = K0 + OUTER < ld ? K0 + OUTER : ld;
        if (chain) {
            const int nb = (K1 - K0) / PANEL, below_tiles = (ld - K1) / PANEL;
            const double e = 64.0 * nb;
            SLK_RUN_W("chol_chain", Bd * e * e * e / 3.0, Bd * 12.0 * e * e, nb * batch, s,
                      k_chol_chain<false><<<dim3(nb, 1, B), 256, sizeof(PanelSmem), s>>>(A, ld, K0, nb, X, info, flags, opt(OPT_WIN_DBG) & 8));
            if (below_tiles > 0 && forked >= 0) {
                // look-ahead: the rows below read tiles that the REST of the previous block's outer update (helper stream)
                // writes; the chain above did not -- it ran beside it
                SLK_HIP(hipStreamWaitEvent(s, helper.events[EV * forked + 1], 0));
                forked = -1;
            }
            // the rows below: 16 rows per workgroup (k_chol_rows_below: 45 us a launch at 4096 columns, most of the chip) or,
            // "rows_below_wide" = 1, 64 rows per workgroup (k_chol_chain<true>: 103 us, a quarter of the workgroups: 0.6 of the
            // chip time).  Measured on whole streams, wide against narrow: headline 5257 / 5272 Mweights/s, one rank of 8
            // 3.66 / 3.61 ms per step, BLOOM-560M 5139 / 5084 -- nothing in it, and a single factorisation is 0.4 ms longer:
            // narrow stays.
            // ... EXCEPT for chains of several wide factorisations (OPT-350M / BLOOM-560M's 1024 x 4096 layers go six to eight
            // to a chain): 4 x 56 x 8 strips are seven rounds of the chip at 45 us, the 64-row form under two at 103.
            const int rw = opt(OPT_ROWS_BELOW_WIDE);
            const bool wide = rw == 1 || (rw == 0 && 4 * below_tiles * batch >= 1024);
            if (below_tiles > 0 && wide)
                SLK_RUN_W("chol_rows_below", Bd * 64.0 * below_tiles * e * e, Bd * (16.0 * 64 * below_tiles * e + 4.0 * e * e), below_tiles * batch, s,
                          k_chol_chain<true><<<dim3(below_tiles, 1, B), 256, sizeof(PanelSmem), s>>>(A, ld, K0, nb, X, info, flags, 0));
            else if (below_tiles > 0)
                SLK_RUN_W("chol_rows_below", Bd * 64.0 * below_tiles * e * e, Bd * (16.0 * 64 * below_tiles * e + 4.0 * e * e), 4 * below_tiles * batch, s,
                          k_chol_rows_below<<<dim3(4 * below_tiles, 1, B), 256, sizeof(RowsBelowSmem), s>>>(A, ld, K0, nb, X));
        }
        for (int k0 = K0; k0 < K1 && !chain; k0 += PANEL) {
            const int below = (ld - k0) / PANEL;  // tiles from the diagonal tile down
            // ONE launch per panel: [the previous panel's update of this tile column (prologue of every workgroup); potf2 +
            // inverse of the diagonal tile; the triangular product below it] in `below` workgroups, and in the same grid what
            // is left of the previous panel's update inside this outer block: the tile columns right of this panel (`rest`).
            // (Until round 3 the whole inner update was a launch of its own between two panels: 112 dependent launches per
            // 4096-column factor, now 64 + 16 -- 64 + 8 with outer blocks of 512 columns.)
            const int kprev = k0 > K0 ? k0 - PANEL : -1;
            const int rest_cols = kprev >= 0 ? (K1 - k0) / PANEL - 1 : 0;
            int rest = 0;
            for (int c = 0; c < rest_cols; ++c) rest += nt - (k0 / TILE + 1 + c);
            const double prologue = kprev >= 0 ? (double)below * 2.0 * 64 * 64 * 64 + (double)below * 2.0 * 64 * 64 * 64 : 0.0;  // own + diagonal tile
            if (split) {
                // two launches: the diagonal tile by one workgroup per matrix, then the tiles below it (and part B) -- see k_panel_below
                const double pro1 = kprev >= 0 ? 2.0 * 64 * 64 * 64 : 0.0;
                SLK_RUN_W("chol_panel", Bd * (2.0 / 3.0 * 64 * 64 * 64 + pro1), Bd * 16.0 * 64 * 64, batch, s,
                          k_chol_panel<<<dim3(1, 1, B), 256, sizeof(PanelSmem), s>>>(A, ld, k0, kprev, 1, 0, X, info, opt(OPT_WIN_DBG) & 8));
                if (below - 1 + rest > 0) {
                    // two row tiles per workgroup (inv(L11) and the diagonal rows' L loaded once for both, the second tile's
                    // operands in flight during the first's products): OPT-350M 51.55 -> 51.25 ms per step, 52.0 with four
                    const int per_wg = 2;
                    const int nbw = (below - 1 + per_wg - 1) / per_wg;  // workgroups that walk `per_wg` row tiles each
                    SLK_RUN_W("chol_panel_below",
                              Bd * ((double)(below - 1) * 64 * 64 * 64 + (kprev >= 0 ? (double)(below - 1) * 2.0 * 64 * 64 * 64 : 0.0) +
                                    (double)rest * 2.0 * 64 * 64 * PANEL),
                              Bd * (24.0 * (below - 1) * 64 * 64 + (double)rest * 16.0 * 64 * 64), (nbw + rest) * batch, s,
                              k_panel_below<<<dim3(nbw + rest, 1, B), 256, sizeof(BelowSmem), s>>>(A, ld, k0, kprev, nbw, rest_cols, X, below - 1,
                                                                                                  per_wg));
                }
            } else {
                SLK_RUN_W("chol_panel",
                          Bd * (2.0 / 3.0 * 64 * 64 * 64 + (double)(below - 1) * 64 * 64 * 64 + prologue + (double)rest * 2.0 * 64 * 64 * PANEL),
                          Bd * (16.0 * below * 64 * 64 + (double)rest * 16.0 * 64 * 64), (below + rest) * batch, s,
                          k_chol_panel<<<dim3(below + rest, 1, B), 256, sizeof(PanelSmem), s>>>(A, ld, k0, kprev, below, rest_cols, X, info,
                                                                                                opt(OPT_WIN_DBG) & 8));
            }
        }
        const int t0 = K1 / TILE;
        if (nt > t0) {
            const int m = nt - t0, total = m * (m + 1) / 2;
            const double tiles = total;
            const int ahead = OUTER / TILE;  // tile columns of the next outer block
            if (lookahead && chain && m >= LOOKAHEAD_MIN_TILES) {
                // LOOK-AHEAD with the chain (round 4).  The next block's chain reads its DIAGONAL block and nothing else; the
                // rows below it are not read before that chain is through (k_chol_rows_below).  So only the diagonal block's
                // tiles (<= 36) are updated here, on the caller's stream, and the next chain starts at once; all the rest of
                // the outer update -- the rectangle under that block and the triangle beyond -- goes to the helper stream and
                // runs beside the chain (8 CUs for ~130 us), joined before the next block's rows below are made (above).
                // Same tiles, same K ranges, one update per tile and block in the order of the blocks (events): same bits.
                SLK_HIP(hipEventRecord(helper.events[EV * block], s));  // this block's chain and rows below are done
                const int nbn = ahead < m ? ahead : m, td = nbn * (nbn + 1) / 2;
                SLK_RUN_W("chol_syrk_ahead", Bd * td * 2.0 * 64 * 64 * (K1 - K0), Bd * (16.0 * nbn * 64 * (K1 - K0) + td * 16.0 * 64 * 64), td * batch, s,
                          k_syrk_tiles<<<dim3(nbn, nbn, B), 256, 0, s>>>(A, ld, t0, t0, K0, K1));
                SLK_HIP(hipStreamWaitEvent(helper.stream, helper.events[EV * block], 0));
                const int m2 = m - nbn, total2 = m2 * (m2 + 1) / 2;
                if (m2 > 0) {
                    SLK_RUN_W("chol_syrk_outer", Bd * (double)nbn * m2 * 2.0 * 64 * 64 * (K1 - K0), Bd * (8.0 * (ld - K1) * (K1 - K0) + (double)nbn * m2 * 16.0 * 64 * 64),
                              (double)nbn * m2 * batch, helper.stream,
                              k_syrk_tiles<<<dim3(nbn, m2, B), 256, 0, helper.stream>>>(A, ld, t0 + nbn, t0, K0, K1));
                    SLK_RUN_W("chol_syrk_outer", Bd * total2 * 2.0 * 64 * 64 * (K1 - K0), Bd * (8.0 * (ld - K1) * (K1 - K0) + total2 * 16.0 * 64 * 64),
                              (double)total2 * batch, helper.stream,
                              k_syrk_triangle<<<dim3(8 * ((total2 + 7) / 8), 1, B), 256, 0, helper.stream>>>(A, ld, t0 + nbn, m2, K0, K1));
                }
                SLK_HIP(hipEventRecord(helper.events[EV * block + 1], helper.stream));
                forked = block;
                // ... and beside it, on a helper stream of their own (behind the rest of the update they would hold up the next
                // block's rows below), the nodes of the inverse that this block completes (tiles [0, K1) are factored): they
                // read L and X tiles left of K1, which nothing writes any more
                SLK_HIP(hipStreamWaitEvent(helper.stream2, helper.events[EV * block], 0));
                {
                    const int rc = inverse_nodes(t0, helper.stream2);
                    if (rc != SLK_OK) return rc;
                }
                SLK_HIP(hipEventRecord(helper.events[EV * block + 2], helper.stream2));
                inverse_on_helper = block;
            } else if (lookahead && m >= LOOKAHEAD_MIN_TILES) {
                // LOOK-AHEAD.  Only the next outer block's columns are needed before its panels can start: those
                // (4 m - 6 tiles) are updated here, on the caller's stream; the rest of the triangle goes to the helper
                // stream and runs beside the next block's panels (<= 64 workgroups each), which it used to hold up.
                // Same tiles, same K ranges, a fixed order per tile (events below), so the results do not depend on timing.
                if (block > 0) SLK_HIP(hipStreamWaitEvent(s, helper.events[2 * (block - 1) + 1], 0));  // rest of block - 1 is in
                SLK_HIP(hipEventRecord(helper.events[2 * block], s));                                // panels of this block are done
                const double ta = (double)ahead * m - 6;
                SLK_RUN_W("chol_syrk_ahead", Bd * ta * 2.0 * 64 * 64 * (K1 - K0), Bd * (8.0 * (ld - K1) * (K1 - K0) + ta * 16.0 * 64 * 64),
                          ta * batch, s, k_syrk_tiles<<<dim3(ahead, m, B), 256, 0, s>>>(A, ld, t0, t0, K0, K1));
                SLK_HIP(hipStreamWaitEvent(helper.stream, helper.events[2 * block], 0));
                const int m2 = m - ahead, total2 = m2 * (m2 + 1) / 2;
                SLK_RUN_W("chol_syrk_outer", Bd * total2 * 2.0 * 64 * 64 * (K1 - K0), Bd * (8.0 * (ld - K1) * (K1 - K0) + total2 * 16.0 * 64 * 64),
                          (double)total2 * batch, helper.stream,
                          k_syrk_triangle<<<dim3(8 * ((total2 + 7) / 8), 1, B), 256, 0, helper.stream>>>(A, ld, t0 + ahead, m2, K0, K1));
                SLK_HIP(hipEventRecord(helper.events[2 * block + 1], helper.stream));
                forked = block;
            } else {
                if (forked >= 0) {  // the helper's last piece touches these tiles too: join first
                    SLK_HIP(hipStreamWaitEvent(s, helper.events[EV * forked + 1], 0));
                    forked = -1;
                }
                if (lookahead && chain) {
                    // (a short trailing triangle: the update stays on the caller's stream, but the nodes of the inverse that this
                    // block completes still go to the second helper stream, beside the next chain)
                    SLK_HIP(hipEventRecord(helper.events[EV * block], s));
                    SLK_HIP(hipStreamWaitEvent(helper.stream2, helper.events[EV * block], 0));
                    const int rc = inverse_nodes(t0, helper.stream2);
                    if (rc != SLK_OK) return rc;
                    SLK_HIP(hipEventRecord(helper.events[EV * block + 2], helper.stream2));
                    inverse_on_helper = block;
                }
                SLK_RUN_W("chol_syrk_outer", Bd * tiles * 2.0 * 64 * 64 * (K1 - K0), Bd * (8.0 * (ld - K1) * (K1 - K0) + tiles * 16.0 * 64 * 64),
                          tiles * batch, s, k_syrk_triangle<<<dim3(8 * ((total + 7) / 8), 1, B), 256, 0, s>>>(A, ld, t0, m, K0, K1));
            }
        }
        ++block;
    }
    if (forked >= 0) SLK_HIP(hipStreamWaitEvent(s, helper.events[EV * forked + 1], 0));
    if (inverse_on_helper >= 0) SLK_HIP(hipStreamWaitEvent(s, helper.events[EV * inverse_on_helper + 2], 0));
    {
        const int rc = inverse_nodes(nt, s);  // every node not merged yet (without look-ahead: all of them)
        if (rc != SLK_OK) return rc;
    }
    SLK_RUN("flip_out", 0, Bd * 12.0 * n * n, s, k_flip_out<<<dim3(n < 2048 ? n : 2048, 1, B), 256, 0, s>>>(X, ld, n, U));
    return SLK_OK;
}

extern "C" int slk_chol_inverse_upper(double *A, int n, double *U, int *info, void *workspace,
                                      size_t ws_bytes, slk_stream_t stream) {
    SLK_REQUIRE(A && U && info && n > 0, "bad arguments");
    return chol_inverse_impl(A, 1, n, U, info, workspace, ws_bytes, stream, false);
}

extern "C" int slk_chol_inverse_upper_lookahead(double *A, int n, double *U, int *info, void *workspace,
                                                size_t ws_bytes, slk_stream_t stream) {
    SLK_REQUIRE(A && U && info && n > 0, "bad arguments");
    return chol_inverse_impl(A, 1, n, U, info, workspace, ws_bytes, stream, true);
}

extern "C" int slk_chol_inverse_upper_batch(double *A, int batch, int n, double *U, int *info, void *workspace,
                                            size_t ws_bytes, slk_stream_t stream) {
    SLK_REQUIRE(A && U && info && n > 0, "bad arguments");
    SLK_REQUIRE(batch >= 1 && batch <= 64, "batch must be 1..64");
    return chol_inverse_impl(A, batch, n, U, info, workspace, ws_bytes, stream, false);
}

extern "C" int slk_probe_panel_cycles(long long *host_out, int reset) {
    SLK_REQUIRE(host_out, "null pointer");
    SLK_HIP(hipDeviceSynchronize());
    SLK_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_panel_cycles), sizeof(long long) * 16));
    if (reset) {
        long long zero[16] = {0};
        SLK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_panel_cycles), zero, sizeof(zero)));
    }
    return SLK_OK;
}

extern "C" size_t slk_factor_workspace_bytes_batch(int batch, int n) {
    if (batch < 1 || batch > 64 || n <= 0) return 0;
    const size_t ld = (size_t)slk_factor_ld(n);
    return 2 * ld * ld * sizeof(double) * batch + (size_t)batch * (64 * sizeof(float) + (size_t)n * (sizeof(double) + sizeof(int))) +
           (size_t)batch * 2 * (ld / 64) * sizeof(int) + (1u << 16);
}
