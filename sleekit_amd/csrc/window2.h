// The window kernel for the reference's standard schedules: row-independent chain waves.
//
// A window (<= 512 columns of the permuted layer, 16 rows per workgroup) is a sequence of PERIODS:
//   LEAF(s, w1) [ UPDATE(s, w1 -> w2) LEAF(s + w1, w2) ]   then   UPDATE(s, K = w1 + w2 -> rest of the window)
// with w1, w2 <= 32 (obq.py:121-137 with min_block_size <= 32: n = 768 ... 11008 all give this).
// Rows never interact, so a CHAIN wave (waves 0-3, four rows each, lane = (row, column mod 16))
// runs its rows through a whole period alone: the leaf chain in registers, then the update of the
// columns it needs next (the other leaf of the period, the next period) with
// v_mfma_f64_4x4x4_4b_f64, whose result layout -- lane 16 i + c holds row i, column c -- is the
// leaf's own.  That instruction, v_mfma_f64_16x16x4_f64 and a chain of v_fma_f64 over k all round
// identically (tools/scratch/mfma4b.hip: bit-equal for K = 4 ... 512), so who computes which
// columns is free.  The HELPER waves (4-7) feed the chain waves through LDS (leaf tables and the
// U blocks of the local updates, one period ahead) and fold the rest of each period's update,
// the columns beyond the next period, in behind their back on the 16-row MFMA.  One barrier per period:
//
//   chain:    [leaves of period p] B_p [update of period p+1's columns][leaves of period p+1] B_p+1 ...
//   helpers:  ... B_p [rest of period p's update; tables, blocks for period p+2 / update p+1] B_p+1 ...
#pragma once

namespace slk {

constexpr int ERING = 128;  // columns of E kept in LDS (two periods)
constexpr int MAXP = 32;    // periods per window

struct Period {
    int s, w1, w2, nw;  // start column, leaf widths (w2 = 0: one leaf), width of the next period (0: last)
};
struct PeriodTable {
    int count;
    Period p[MAXP];
};

// SETS = 1: 16 rows per workgroup, every staging buffer double (158 KB).  SETS = 2: 32 rows per workgroup -- a chain
// wave runs TWO sets of four rows interleaved (the chain is latency-bound: tools/micro_leaf.py measures 126 cycles a
// column for one, two, three or four chains per SIMD alike), a helper wave folds both 16-row tiles in behind one read
// of U -- so that a launch of R rows occupies R / 32 CUs for about the same time instead of R / 16: the chip time of
// the window kernel per row drops by a third and the CUs it leaves go to the other streams' kernels.  The two U
// blocks are single buffers then (158 KB again), guarded by two LDS counters instead of the buffer parity.
struct LeafTables8 {
    double u[ULEAF][ULEAF];  // leaf block of U, strictly upper part, columns permuted: [i][(j & 7) * 4 + (j >> 3)]
    double udr[ULEAF][2];    // its diagonal (1 beyond the width) and 1 / diagonal by true division
};

template <int SETS>
struct Window2SmemT {
    // [period parity][leaf]; SETS == 2: eight rows per chain wave (leaf_chain8), a lane's four columns side by side
    typename std::conditional<SETS == 1, LeafTables, LeafTables8>::type lt[2][2];
    double sblk[SETS == 1 ? 2 : 1][16 * 64];   // U block of the update between the two leaves, in chunks of 4 k x 16 columns
    double nblk[SETS == 1 ? 2 : 1][64 * 64];   // U block of the update into the next period, same chunking
    float q[RB * SETS][WPITCH];
    float e[RB * SETS][ERING + 4];
    int odd[2][2][4];        // [parity][leaf][helper wave]: a diagonal defeats the exact-division shortcut
    int n_done;              // SETS == 2: chain waves through with nblk, 4 per period
    int s_count;             // SETS == 2: helper waves that have stored their part of sblk, 4 per period from period 1 on
    float cbt[512];          // a general codebook's values and limits (<= 256 entries)
};
using Window2Smem = Window2SmemT<1>;

// spin on an LDS counter another wave of the workgroup bumps (acquire: what that wave wrote before is visible after)
__device__ __forceinline__ void lds_wait_ge(int *counter, int target) {
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ void lds_signal(int *counter, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // this wave's LDS reads / writes are done
    if (lane == 0) __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The leaf chain on registers (see leaf_registers).  FAST: Markstein divisions (exact unless a
// significand is all ones, which the caller has excluded); otherwise true divides.
template <int NSTEP, bool FAST, int SETS>
__device__ __forceinline__ void leaf_chain(const LeafTables &lt, int c16, float (&x0)[SETS], float (&x1)[SETS], float (&q0)[SETS],
                                           float (&q1)[SETS], float (&e0)[SETS], float (&e1)[SETS], const Grid g, float inv_step) {
    double u0n = lt.u[0][c16], u1n = lt.u[0][c16 + 16], uiin = lt.udr[0][0], riin = lt.udr[0][1];
    static_for<0, NSTEP>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const double u0 = u0n, u1 = u1n, uii = uiin, rii = riin;
        if constexpr (i + 1 < NSTEP) {
            u0n = lt.u[i + 1][c16];
            u1n = lt.u[i + 1][c16 + 16];
            uiin = lt.udr[i + 1][0];
            riin = lt.udr[i + 1][1];
        }
        // the reads above are issued before step i starts
        if constexpr (SETS == 1) asm volatile("" : "+v"(x0[0]), "+v"(x1[0])::"memory");
        else asm volatile("" : "+v"(x0[0]), "+v"(x1[0]), "+v"(x0[1]), "+v"(x1[1])::"memory");
        constexpr int src = i & 15;
#pragma unroll
        for (int st = 0; st < SETS; ++st) {  // independent chains: the compiler interleaves them
            // column i of each of the wave's rows, broadcast inside its 16-lane DPP row; every lane recomputes the
            // column's error for its own row.  The chain is bound by the ISSUE of these instructions: nothing is kept
            // per step (the column's own q and e come after the loop, below)
            const float xi = row_bcast<src>(i < 16 ? x0[st] : x1[st]);
            double err;
            if (FAST) {
                const float q = grid_value_fast_med3(xi, g, inv_step);
                const double d = (double)(xi - q);
                const double qq = d * rii;
                const double rem = __builtin_fma(-uii, qq, d);
                err = __builtin_fma(rem, rii, qq);
            } else {
                err = (double)(xi - cb_value(xi, g)) / uii;
            }
            if (i < 15) x0[st] = (float)((double)x0[st] - err * u0);
            if (NSTEP > 16) x1[st] = (float)((double)x1[st] - err * u1);
        }
    });
    // A lane's own columns are final once their step has passed: the block of U is zero on and below the diagonal, so
    // the later steps subtract err * 0 (at most the sign of a zero changes, which no result can see).  Their q and e
    // are the same expressions on the same value as in the step that broadcast it: computed once here instead of being
    // selected into place in every step (two v_cndmask and a conversion per step less on the chain).
    const double d0 = lt.udr[c16][0], r0 = lt.udr[c16][1];
    const double d1 = lt.udr[NSTEP > 16 ? c16 + 16 : c16][0], r1 = lt.udr[NSTEP > 16 ? c16 + 16 : c16][1];
#pragma unroll
    for (int st = 0; st < SETS; ++st) {
#pragma unroll
        for (int half = 0; half < (NSTEP > 16 ? 2 : 1); ++half) {
            const float xv = half ? x1[st] : x0[st];
            const double uii = half ? d1 : d0, rii = half ? r1 : r0;
            float q;
            double err;
            if (FAST) {
                q = grid_value_fast_med3(xv, g, inv_step);
                const double d = (double)(xv - q);
                const double qq = d * rii;
                const double rem = __builtin_fma(-uii, qq, d);
                err = __builtin_fma(rem, rii, qq);
            } else {
                q = cb_value(xv, g);
                err = (double)(xv - q) / uii;
            }
            if (half) q1[st] = q, e1[st] = (float)err;
            else q0[st] = q, e0[st] = (float)err;
        }
    }
}

// lane S of every 8-lane group, to all lanes of that group: row_newbcast of lane S over the 16-lane DPP row, then lanes
// 8-15 of the row (bank mask 0xC) take lane 8 + S instead
template <int S>
__device__ __forceinline__ float bcast8(float x) {
    int t = __builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + S, 0xf, 0xf, true);
    t = __builtin_amdgcn_update_dpp(t, __float_as_int(x), 0x150 + 8 + S, 0xf, 0xc, false);
    return __int_as_float(t);
}

// The leaf chain on registers: lane (rg = lane >> 3, sub = lane & 7) holds columns sub + 8 k of row rg in x[k].
// FAST: Markstein divisions (exact unless a significand is all ones, which the caller has excluded); otherwise true
// divides / table look-ups.  Step i (obq.py:106-118): q = quantizer(x_i); err = float64(x_i - q) / U[i][i];
// x_c <- float32(float64(x_c) - err * U[i][c]) for the columns c > i, product and difference rounded separately.
template <int NSTEP, bool FAST>
__device__ __forceinline__ void leaf_chain8(const LeafTables8 &lt, int sub, float (&x)[4], float (&q)[4], float (&e)[4], const Grid g,
                                            float inv_step) {
    constexpr int NK = NSTEP / 8;
    const double2_t *urow = reinterpret_cast<const double2_t *>(&lt.u[0][4 * sub]);  // + 16 doubles per row of U
    double2_t ulo = urow[0], uhi = NK > 2 ? urow[1] : (double2_t){0.0, 0.0};
    double2_t dr = *reinterpret_cast<const double2_t *>(&lt.udr[0][0]);
    static_for<0, NSTEP>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int kk = i >> 3, s = i & 7;
        const double u[4] = {ulo[0], ulo[1], uhi[0], uhi[1]};
        const double uii = dr[0], rii = dr[1];
        if constexpr (i + 1 < NSTEP) {
            // the next step's operands are read before this step starts; only the pairs it still updates
            constexpr int first = ((i + 1) >> 3) + (((i + 1) & 7) == 7 ? 1 : 0);  // its first live register
            if constexpr (first < 2) ulo = urow[16 * (i + 1)];
            if constexpr (first < NK && NK > 2) uhi = urow[16 * (i + 1) + 1];
            dr = *reinterpret_cast<const double2_t *>(&lt.udr[i + 1][0]);
        }
        if constexpr (NK == 4) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3])::"memory");
        else asm volatile("" : "+v"(x[0]), "+v"(x[1])::"memory");
        // column i of each of the wave's eight rows, broadcast inside its 8-lane group; every lane recomputes the
        // column's error for its own row.  Nothing is kept per step (a lane's own q and e come after the loop).
        const float xi = bcast8<s>(x[kk]);
        double err;
        if (FAST) {
            const float qv = grid_value_fast_med3(xi, g, inv_step);
            const double d = (double)(xi - qv);
            const double qq = d * rii;
            const double rem = __builtin_fma(-uii, qq, d);
            err = __builtin_fma(rem, rii, qq);
        } else {
            err = (double)(xi - cb_value(xi, g)) / uii;
        }
#pragma unroll
        for (int k = kk; k < NK; ++k) {
            if (k == kk && s == 7) continue;  // the last column of register kk: nothing right of it in that register
            x[k] = (float)((double)x[k] - err * u[k]);
        }
    });
    // A lane's own columns are final once their step has passed: the block of U is zero on and below the diagonal, so the
    // later steps of their register subtract err * 0.  Their q and e are the same expressions on the same value as in the
    // step that broadcast it: computed once here.
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const double2_t d = *reinterpret_cast<const double2_t *>(&lt.udr[sub + 8 * k][0]);
        const float xv = x[k];
        float qv;
        double err;
        if (FAST) {
            qv = grid_value_fast_med3(xv, g, inv_step);
            const double dd = (double)(xv - qv);
            const double qq = dd * d[1];
            const double rem = __builtin_fma(-d[0], qq, dd);
            err = __builtin_fma(rem, d[1], qq);
        } else {
            qv = cb_value(xv, g);
            err = (double)(xv - qv) / d[0];
        }
        q[k] = qv;
        e[k] = (float)err;
    }
}

template <int SETS>
__global__ __launch_bounds__(512) void k_gptq_window2(float *__restrict__ Qp, float *__restrict__ Eg,
                                                      const double *__restrict__ U, int R, int n, int w0, int w1,
                                                      Grid g, float inv_step, int fast_ok, int prof, PeriodTable tab, int rpl) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Window2SmemT<SETS> &sm = *reinterpret_cast<Window2SmemT<SETS> *>(smem_raw);
    constexpr int RBX = RB * SETS;  // rows of this workgroup
    // readfirstlane: tells the compiler the wave index is wave-uniform (scalar branches, SGPR addressing)
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // cycle accounting (debug, SLK_WIN_DBG bit 3): wave-uniform, workgroup 0, waves 0 and 4
    const bool timing = (prof & 8) && blockIdx.x == 0;
    long long tmark = timing ? (long long)__builtin_readcyclecounter() : 0;
    const long long tstart = tmark;
    long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tper = tmark;  // start of the current period / interval
    auto lap = [&](int slot) {
        if (timing) {
            const long long now = (long long)__builtin_readcyclecounter();
            tacc[slot] += now - tmark;
            tmark = now;
        }
    };
    // Waves go round the four SIMDs (wave i on SIMD i % 4).  A leaf chain is bound by the ISSUE of its own instructions,
    // not by their latency: two chains on one SIMD take turns, oldest first -- tools/micro_leaf.py: 127 cycles a column
    // for one chain wave per SIMD, 189 for the slower of two, 348 of four (round 1 timed wave 0 only, the oldest, and
    // concluded the opposite) -- and a wave issuing float64 MFMAs back to back doubles the chain on its SIMD (255).  The
    // helpers issue MFMAs a fraction of their time, so one chain wave AND one helper wave per SIMD (chain waves 0-3,
    // helpers 4-7) beats two chains on SIMDs 0-1 and the helpers on 2-3: 80.5 -> 73.3 us per launch, 4880 -> 4980
    // Mweights/s.
    const bool helper = wave >= 4;
    const int role_wave = wave & 3;  // 0-3 within the role
    constexpr int first_helper = 4;
    // The oldest wave of a SIMD issues first, so a helper (waves 4-7) would only get the slots its chain wave leaves; in
    // the first periods the helpers are the longer side, and with raised priority the two meet in the middle (73.0 ->
    // 70.0 us per launch; the same priority on the chain waves instead changes nothing, they are the oldest already).
    // (prof bits 64 / 128, measurement only: no helper priority / the chain waves raised instead)
    if (helper && !(prof & (64 | 128))) __builtin_amdgcn_s_setprio(3);
    if (!helper && (prof & 128)) __builtin_amdgcn_s_setprio(3);
    const int ht = role_wave * 64 + lane;                   // thread index within the role, 0-255
    const int r0 = blockIdx.x * RBX;
    U += (size_t)(r0 / rpl) * n * n;  // a batch of layers stacked by rows: rows [b rpl, (b + 1) rpl) use factor b
    const int np = tab.count;
    const int width = w1 - w0;
    if (g.table) {  // the leaves search the codebook once per column: keep it next to them
        for (int i = t; i < 2 * g.n - 1; i += 512) sm.cbt[i] = g.table[i];
        g.table = sm.cbt;  // visible after the first barrier (B_start)
    }
    auto ring = [&](int c) { return (c - w0) & (ERING - 1); };

    // columns [c_lo, c_hi) of the Q tile, global -> LDS, by `nth` threads of which this is number `tid`:
    // 16-byte loads when the layout allows, eight (four) loads in flight per thread either way
    // (a load-wait-store loop pays the full latency per element)
    const bool vec4 = n % 4 == 0 && (w0 & 3) == 0 && ((uintptr_t)Qp & 15) == 0;
    auto load_cols = [&](int c_lo, int c_hi, int tid, int nth) {
        const int cw = c_hi - c_lo;
        if (vec4 && (c_lo & 3) == 0 && (cw & 3) == 0) {
            const int cw4 = cw >> 2, total = RBX * cw4;
            for (int e0 = tid; e0 < total; e0 += 4 * nth) {
                float4v_t v[4];
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const int e = min(e0 + h * nth, total - 1);
                    const int r = e / cw4, c = c_lo + 4 * (e % cw4);
                    v[h] = *reinterpret_cast<const float4v_t *>(Qp + (size_t)min(r0 + r, R - 1) * n + c);
                }
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const int e = e0 + h * nth;
                    const int r = e / cw4, c = c_lo + 4 * (e % cw4);
                    if (e < total) *reinterpret_cast<float4v_t *>(&sm.q[r][c - w0]) = (r0 + r < R) ? v[h] : (float4v_t){0.0f, 0.0f, 0.0f, 0.0f};
                }
            }
            return;
        }
        const int total = RBX * cw;
        for (int e0 = tid; e0 < total; e0 += 8 * nth) {
            float v[8];
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                const int e = min(e0 + h * nth, total - 1);
                const int r = e / cw, c = c_lo + e % cw;
                v[h] = Qp[(size_t)min(r0 + r, R - 1) * n + c];
            }
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                const int e = e0 + h * nth;
                const int r = e / cw, c = c_lo + e % cw;
                if (e < total) sm.q[r][c - w0] = (r0 + r < R) ? v[h] : 0.0f;
            }
        }
    };
    const Period P0 = tab.p[0];
    const int end0 = P0.s + P0.w1 + P0.w2;
    if (SETS == 2 && t == 0) sm.n_done = 0, sm.s_count = 0;  // visible after B_start

    if (!helper) {
        // ======================================================== chain waves
        const int c16 = lane & 15, rg = lane >> 4;
        int row[SETS];
        bool row_live[SETS];
#pragma unroll
        for (int st = 0; st < SETS; ++st) row[st] = RB * st + 4 * role_wave + rg, row_live[st] = r0 + row[st] < R;
        // Q[rows, dst0 : dst0 + N] -= E[rows, src0 : src0 + K] @ (U block in `chunks`), this wave's four rows.
        // A operand: lane 16 k + 4 q + i carries E[i][k] (the same for the four column quads q);
        // B operand: lane 16 k + c carries U[k][c]: chunk (cg, ks) is read as 512 contiguous bytes.
        auto local_update = [&](const double *chunks, int src0, int K, int dst0, int N) {
            const int ncg = (N + 15) >> 4, nks = (K + 3) >> 2;
            const int ai = lane & 3, ak = lane >> 4;
            const float *erow[SETS];
            double acc[SETS][4];
#pragma unroll
            for (int st = 0; st < SETS; ++st) {
                erow[st] = &sm.e[RB * st + 4 * role_wave + ai][0];
#pragma unroll
                for (int cg = 0; cg < 4; ++cg) acc[st][cg] = 0.0;
            }
            auto body = [&](auto nks_c, auto ncg_c) {
                constexpr int NKS = decltype(nks_c)::value, NCG = decltype(ncg_c)::value;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    double a[SETS];
#pragma unroll
                    for (int st = 0; st < SETS; ++st) a[st] = (double)erow[st][ring(src0 + 4 * ks + ak)];
#pragma unroll
                    for (int cg = 0; cg < NCG; ++cg) {
                        const double bchunk = chunks[(size_t)(cg * NKS + ks) * 64 + lane];
#pragma unroll
                        for (int st = 0; st < SETS; ++st) acc[st][cg] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[st], bchunk, acc[st][cg], 0, 0, 0);
                    }
                }
            };
            using std::integral_constant;
            if (K == 64 && ncg == 4) body(integral_constant<int, 16>{}, integral_constant<int, 4>{});
            else if (K == 32 && ncg == 2) body(integral_constant<int, 8>{}, integral_constant<int, 2>{});
            else {
                for (int ks = 0; ks < nks; ++ks) {
                    const int kk = src0 + 4 * ks + ak;
                    double a[SETS];
#pragma unroll
                    for (int st = 0; st < SETS; ++st) {
                        const float ev = erow[st][ring(min(kk, src0 + K - 1))];
                        a[st] = kk < src0 + K ? (double)ev : 0.0;
                    }
#pragma unroll
                    for (int cg = 0; cg < 4; ++cg)
                        if (cg < ncg) {
                            const double bchunk = chunks[(size_t)(cg * nks + ks) * 64 + lane];
#pragma unroll
                            for (int st = 0; st < SETS; ++st) acc[st][cg] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[st], bchunk, acc[st][cg], 0, 0, 0);
                        }
                }
            }
#pragma unroll
            for (int st = 0; st < SETS; ++st)
#pragma unroll
                for (int cg = 0; cg < 4; ++cg) {
                    const int col = dst0 + 16 * cg + c16;
                    if (cg < ncg && col < dst0 + N) sm.q[row[st]][col - w0] = (float)((double)sm.q[row[st]][col - w0] - acc[st][cg]);
                }
        };

        load_cols(w0, end0, t, 512);  // with the helpers' half
        __syncthreads();  // B_start
        lap(0);
        for (int p = 0; p < np; ++p) {
            const Period P = tab.p[p];
            const int par = p & 1;
#pragma unroll 1
            for (int lf = 0; lf < 2; ++lf) {
                const int w = lf ? P.w2 : P.w1;
                if (w == 0) break;
                const int a = lf ? P.s + P.w1 : P.s;
                if (lf) {
                    if (SETS == 2) lds_wait_ge(&sm.s_count, 4 * p);  // the helpers store period p's block at the top of interval p
                    local_update(sm.sblk[SETS == 1 ? par : 0], P.s, P.w1, a, w);
                }
                lap(2);
                const bool fast = fast_ok && (sm.odd[par][lf][0] | sm.odd[par][lf][1] | sm.odd[par][lf][2] | sm.odd[par][lf][3]) == 0;
                if constexpr (SETS == 2) {
                    // EIGHT rows per chain wave, 8 lanes x 4 columns per lane (leaf_chain8): one quantizer instruction stream
                    // for both sets of four rows instead of two interleaved ones -- the chain is bound by instruction issue
                    const auto &lt = sm.lt[par][lf];
                    const int sub = lane & 7, g8 = lane >> 3;
                    const int lrow = RB * (g8 >> 2) + 4 * role_wave + (g8 & 3);  // this lane's row: the two sets of local_update
                    const bool live = r0 + lrow < R;
                    float x[4], qv[4], ev[4];
                    bool m[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        m[k] = sub + 8 * k < w;
                        x[k] = m[k] ? sm.q[lrow][a - w0 + sub + 8 * k] : 0.0f;
                        qv[k] = ev[k] = 0.0f;
                    }
                    if (!fast) leaf_chain8<32, false>(lt, sub, x, qv, ev, g, inv_step);
                    else if (w <= 16) leaf_chain8<16, true>(lt, sub, x, qv, ev, g, inv_step);
                    else leaf_chain8<32, true>(lt, sub, x, qv, ev, g, inv_step);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (m[k]) {
                            const int col = a + sub + 8 * k;
                            sm.q[lrow][col - w0] = qv[k];
                            sm.e[lrow][ring(col)] = ev[k];
                            if (live) Eg[(size_t)(r0 + lrow) * n + col] = ev[k];
                        }
                    }
                } else {
                const auto &lt = sm.lt[par][lf];
                const bool m0 = c16 < w, m1 = c16 + 16 < w;
                float x0[SETS], x1[SETS], q0[SETS], q1[SETS], e0[SETS], e1[SETS];
#pragma unroll
                for (int st = 0; st < SETS; ++st) {
                    x0[st] = m0 ? sm.q[row[st]][a - w0 + c16] : 0.0f;
                    x1[st] = m1 ? sm.q[row[st]][a - w0 + 16 + c16] : 0.0f;
                    q0[st] = q1[st] = e0[st] = e1[st] = 0.0f;
                }
                if (!fast) leaf_chain<32, false, SETS>(lt, c16, x0, x1, q0, q1, e0, e1, g, inv_step);
                else if (w <= 16) leaf_chain<16, true, SETS>(lt, c16, x0, x1, q0, q1, e0, e1, g, inv_step);
                else leaf_chain<32, true, SETS>(lt, c16, x0, x1, q0, q1, e0, e1, g, inv_step);
#pragma unroll
                for (int st = 0; st < SETS; ++st) {
                    if (m0) {
                        sm.q[row[st]][a - w0 + c16] = q0[st];
                        sm.e[row[st]][ring(a + c16)] = e0[st];
                        if (row_live[st]) Eg[(size_t)(r0 + row[st]) * n + a + c16] = e0[st];
                    }
                    if (m1) {
                        sm.q[row[st]][a - w0 + 16 + c16] = q1[st];
                        sm.e[row[st]][ring(a + 16 + c16)] = e1[st];
                        if (row_live[st]) Eg[(size_t)(r0 + row[st]) * n + a + 16 + c16] = e1[st];
                    }
                }
                }
                lap(1);
            }
            if (timing && lane == 0 && wave == 0) g_win_trace[p] += (long long)__builtin_readcyclecounter() - tper;
            __syncthreads();  // B_p
            tper = timing ? (long long)__builtin_readcyclecounter() : 0;
            lap(3);
            if (P.nw) local_update(sm.nblk[SETS == 1 ? par : 0], P.s, P.w1 + P.w2, P.s + P.w1 + P.w2, P.nw);
            if (SETS == 2) lds_signal(&sm.n_done, lane);  // nblk may be overwritten (every period, to keep the count simple)
            lap(2);
        }
    } else {
        // ======================================================== helper waves
        // leaf tables: registers <- U during one interval, LDS <- registers during the next
        double pu[2][4];
        auto fetch_tables = [&](int p) {  // both leaves of period p
            if (p >= np) return;
            const Period P = tab.p[p];
#pragma unroll
            for (int lf = 0; lf < 2; ++lf) {
                const int a = lf ? P.s + P.w1 : P.s, w = max(1, lf ? P.w2 : P.w1);
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const int e = ht + 256 * h, i = min(e >> 5, w - 1), j = min(e & 31, w - 1);
                    pu[lf][h] = U[(size_t)(a + i) * n + a + j];  // clamped, selected when written
                }
            }
        };
        auto write_tables = [&](int p) {
            if (p >= np) return;
            const Period P = tab.p[p];
#pragma unroll
            for (int lf = 0; lf < 2; ++lf) {
                auto &lt = sm.lt[p & 1][lf];
                const int w = lf ? P.w2 : P.w1;
                // a thread meets at most one diagonal slot (e = 33 i) per leaf: one division, not four
                double dgv = 1.0;
                int di = -1;
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const int e = ht + 256 * h, i = e >> 5, j = e & 31;
                    const bool in = i < w && j < w;
                    lt.u[i][SETS == 1 ? j : (j & 7) * 4 + (j >> 3)] = (in && j > i) ? pu[lf][h] : 0.0;
                    if (i == j) {
                        dgv = in ? pu[lf][h] : 1.0;
                        di = i;
                    }
                }
                bool odd = false;
                if (di >= 0) {
                    lt.udr[di][0] = dgv;
                    lt.udr[di][1] = 1.0 / dgv;
                    odd = (__double_as_longlong(dgv) & 0xFFFFFFFFFFFFFLL) == 0xFFFFFFFFFFFFFLL;
                }
                const bool any = __builtin_amdgcn_ballot_w64(odd) != 0;
                if (lane == 0) sm.odd[p & 1][lf][role_wave] = any ? 1 : 0;
            }
        };
        // U[a : a + K, lo : lo + N] -> chunks (cg * nks + ks) of (4 k) x (16 columns) doubles in LDS, through
        // registers: the loads are issued at the start of an interval, the LDS writes come at its end.
        // (global_load_lds would need no registers, but the compiler then makes every LDS read wait
        // for all loads in flight -- it cannot tell the two apart -- which serialises the MFMA rounds.)
        // Thread (role_wave, lane) of step `it` carries 16 bytes of chunk 2 (role_wave + 4 it) + (lane >> 5).
        // Out-of-range rows and columns are clamped: their products meet a zero E operand or land in
        // columns that are never stored.
        typedef double double2_t __attribute__((ext_vector_type(2)));
        auto stage_fetch = [&](auto steps_c, double2_t *regs, int a, int K, int lo, int N) {
            constexpr int STEPS = decltype(steps_c)::value;
            const int ncg = (N + 15) >> 4, nks = (K + 3) >> 2, total = ncg * nks;
            const int half = lane >> 5, k = (lane >> 3) & 3, piece = lane & 7;
            int ch = 2 * role_wave + half;
            int cg = ch / nks, ks = ch - cg * nks;
            const double *colp = U + (size_t)a * n + lo;
#pragma unroll
            for (int it = 0; it < STEPS; ++it) {
                const bool over = ch >= total;  // beyond the block: repeat the last chunk (never stored)
                const int cgc = over ? ncg - 1 : cg, ksc = over ? nks - 1 : ks;
                const int row = min(4 * ksc + k, K - 1);
                const int col = min(16 * cgc + 2 * piece, N - 2);
                regs[it] = *reinterpret_cast<const double2_t *>(colp + (size_t)row * n + col);
                ch += 8;
                ks += 8;
                while (ks >= nks) ks -= nks, ++cg;
            }
        };
        auto stage_store = [&](auto steps_c, const double2_t *regs, double *chunks, int K, int N) {
            constexpr int STEPS = decltype(steps_c)::value;
            const int total = ((N + 15) >> 4) * ((K + 3) >> 2);
#pragma unroll
            for (int it = 0; it < STEPS; ++it) {
                const int pr = role_wave + 4 * it;
                if (2 * pr + (lane >> 5) < total) *reinterpret_cast<double2_t *>(chunks + (size_t)pr * 128 + 2 * lane) = regs[it];
            }
        };
        using std::integral_constant;
        const integral_constant<int, 8> steps_n{};  // 64 x 64 block: 64 chunks = 32 pairs = 8 steps of 4 waves
        const integral_constant<int, 2> steps_s{};  // 32 x 32 block: 16 chunks
        double2_t nreg[8], sreg[2];

        // ---- the 16-row MFMA update of v1 on four waves: Q[:, lo:hi] -= E[:, a:b] @ U[a:b, lo:hi]
        const int lr = lane & 15, lk = lane >> 4;
        double cur[16];
        // (Scalar per-row-group bases with one vector offset were tried to take the address arithmetic
        // off the vector ALU, which the float64 MFMA shares: the SALU chain it needs is slower, 36 vs 31 us.)
        // The helpers share their SIMDs' issue slots with the chain waves, so the address arithmetic counts: a full chunk
        // of 64 rows (the common case) is read at a uniform base + a 32-bit byte offset that advances by a constant
        // (one v_add_u32 per load; the clamped form below costs an add, a min, a 64-bit multiply-add and a 64-bit
        // shift-add each).  n <= 16384 (checked on the host) keeps the offset below 2^32.
        const char *Ubytes = reinterpret_cast<const char *>(U);
        const unsigned row4 = 32u * (unsigned)n;  // bytes from row k to row k + 4
        auto load_round = [&](int a, int b, int lo, int hi, int blk, int kc, double(&bv)[16]) {
            const int cc = min(lo + blk * 16 + lr, hi - 1);
            const int kbase = a + 64 * kc;
            if (kbase + 64 <= b) {
                unsigned off = ((unsigned)(kbase + lk) * (unsigned)n + (unsigned)cc) * 8u;
#pragma unroll
                for (int s4 = 0; s4 < 16; ++s4) {
                    bv[s4] = *reinterpret_cast<const double *>(Ubytes + off);
                    off += row4;
                }
                return;
            }
#pragma unroll
            for (int s4 = 0; s4 < 16; ++s4) {
                const int k = kbase + 4 * s4 + lk;
                bv[s4] = U[(size_t)min(k, b - 1) * n + cc];
            }
        };
        auto mac_round = [&](int a, int b, int kc, const double(&bv)[16], double4_t(&acc)[SETS]) {
            const int kbase = a + 64 * kc, kcount = min(64, b - kbase);
            if ((kcount == 64 || kcount == 32) && ((kbase - w0) & 31) == 0) {
#pragma unroll
                for (int st = 0; st < SETS; ++st) {  // the round of U in registers serves every 16-row tile
                    const float *ep = &sm.e[RB * st + lr][ring(kbase) + lk];  // 32-aligned: the chunk does not wrap
                    float av[16];
#pragma unroll
                    for (int s4 = 0; s4 < 8; ++s4) av[s4] = ep[4 * s4];
                    if (kcount == 64) {
#pragma unroll
                        for (int s4 = 8; s4 < 16; ++s4) av[s4] = ep[4 * s4];
                    }
#pragma unroll
                    for (int s4 = 0; s4 < 8; ++s4) acc[st] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av[s4], bv[s4], acc[st], 0, 0, 0);
                    if (kcount == 64) {
#pragma unroll
                        for (int s4 = 8; s4 < 16; ++s4) acc[st] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av[s4], bv[s4], acc[st], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int s4 = 0; s4 < 16; ++s4) {
                    const int k = kbase + 4 * s4 + lk;
#pragma unroll
                    for (int st = 0; st < SETS; ++st) {
                        const float ev = sm.e[RB * st + lr][ring(min(k, b - 1))];
                        const double av = k < b ? (double)ev : 0.0;
                        if (4 * s4 < kcount) acc[st] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[s4], acc[st], 0, 0, 0);
                    }
                }
            }
        };
        auto store_block = [&](int lo, int hi, int blk, double4_t(&acc)[SETS]) {  // and clear the accumulators
            const int col = lo + blk * 16 + lr;
#pragma unroll
            for (int st = 0; st < SETS; ++st) {
                if (col < hi) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rr = RB * st + lk + 4 * r;
                        sm.q[rr][col - w0] = (float)((double)sm.q[rr][col - w0] - acc[st][r]);
                    }
                }
                acc[st] = (double4_t){0.0, 0.0, 0.0, 0.0};
            }
        };
        // Rounds in pairs on two register buffers.  The loads of the round after next are issued
        // UNCONDITIONALLY (clamped to the last round when there is none): with a fixed number of loads
        // between a buffer's fill and its use the compiler can wait with vmcnt(16); a conditional
        // load in the loop makes it fall back to vmcnt(0), which serialises load and MFMA.
        auto run_update = [&](int a, int b, int lo, int hi, int wid, int nw) {
            const int nblk = (hi - lo + 15) / 16, nchunk = (b - a + 63) / 64;
            if (wid >= nblk) return;
            const int nr = (nblk - wid + nw - 1) / nw * nchunk;
            const int last_blk = wid + ((nblk - wid - 1) / nw) * nw;
            double other[16];
            load_round(a, b, lo, hi, wid, 0, cur);
            double4_t acc[SETS];
#pragma unroll
            for (int st = 0; st < SETS; ++st) acc[st] = (double4_t){0.0, 0.0, 0.0, 0.0};
            int blk = wid, kc = 0;
            for (int r = 0; r < nr; r += 2) {
                int blk1 = blk, kc1 = kc + 1;
                if (kc1 == nchunk) kc1 = 0, blk1 += nw;
                const bool has1 = r + 1 < nr;
                load_round(a, b, lo, hi, has1 ? blk1 : last_blk, has1 ? kc1 : nchunk - 1, other);
                mac_round(a, b, kc, cur, acc);
                if (kc + 1 == nchunk) store_block(lo, hi, blk, acc);
                int blk2 = blk1, kc2 = kc1 + 1;
                if (kc2 == nchunk) kc2 = 0, blk2 += nw;
                const bool has2 = r + 2 < nr;
                load_round(a, b, lo, hi, has2 ? blk2 : last_blk, has2 ? kc2 : nchunk - 1, cur);
                if (has1) {
                    mac_round(a, b, kc1, other, acc);
                    if (kc1 + 1 == nchunk) store_block(lo, hi, blk1, acc);
                }
                blk = blk2, kc = kc2;
            }
        };

        // ---- prologue: tables and the inner block of period 0
        fetch_tables(0);
        load_cols(w0, end0, t, 512);  // with the chain waves' half
        if (P0.w2) {
            stage_fetch(steps_s, sreg, P0.s, P0.w1, P0.s + P0.w1, P0.w2);
            stage_store(steps_s, sreg, sm.sblk[0], P0.w1, P0.w2);
        }
        write_tables(0);
        fetch_tables(1);
        __syncthreads();  // B_start
        lap(0);
        for (int p = 0; p < np; ++p) {
            // interval I_p, while the chain waves run period p
            const Period P = tab.p[p];
            const int K = P.w1 + P.w2;
            if (SETS == 2 && p >= 1) {
                // single sblk: period p's block (in registers since the last interval) goes in now, after the barrier
                // that ended the chain waves' reads of period p - 1's; they wait for all four helper waves' parts
                if (P.w2) stage_store(steps_s, sreg, sm.sblk[0], P.w1, P.w2);
                lds_signal(&sm.s_count, lane);
            }
            // registers first: a wait for the table values (fetched an interval ago) must not have the
            // block loads below in front of it -- vmcnt completes in order
            write_tables(p + 1);
            lap(6);
            if (P.nw) stage_fetch(steps_n, nreg, P.s, K, P.s + K, P.nw);
            const Period N1 = tab.p[min(p + 1, np - 1)];
            const bool has_s = p + 1 < np && N1.w2;
            if (has_s) stage_fetch(steps_s, sreg, N1.s, N1.w1, N1.s + N1.w1, N1.w2);
            lap(4);
            fetch_tables(p + 2);
            lap(10);
            if (p == 0) load_cols(end0, w1, ht, 256);  // the rest of the tile
            lap(11);
            if (p >= 1) {
                // the rest of period p-1's update: beyond period p, whose columns the chain waves have done
                const Period M = tab.p[p - 1];
                const int lo = P.s + K;
                if (lo < w1) run_update(M.s, M.s + M.w1 + M.w2, lo, w1, role_wave, 4);
            }
            lap(5);
            if (SETS == 2) lds_wait_ge(&sm.n_done, 4 * p);  // single nblk: the chain waves are through with period p - 1's block
            if (P.nw) stage_store(steps_n, nreg, sm.nblk[SETS == 1 ? (p & 1) : 0], K, P.nw);
            if (SETS == 1 && has_s) stage_store(steps_s, sreg, sm.sblk[(p + 1) & 1], N1.w1, N1.w2);
            if (timing && lane == 0 && wave == first_helper) g_win_trace[32 + p] += (long long)__builtin_readcyclecounter() - tper;
            __syncthreads();  // B_p
            tper = timing ? (long long)__builtin_readcyclecounter() : 0;
            lap(7);
        }
    }

    // every column is final: the tile goes back (E went out leaf by leaf)
    if (vec4 && (width & 3) == 0) {
        const int cw4 = width >> 2;
        for (int e = t; e < RBX * cw4; e += 512) {
            const int r = e / cw4, c = 4 * (e % cw4);
            if (r0 + r < R) *reinterpret_cast<float4v_t *>(Qp + (size_t)(r0 + r) * n + w0 + c) = *reinterpret_cast<const float4v_t *>(&sm.q[r][c]);
        }
    } else {
        for (int e = t; e < RBX * width; e += 512) {
            const int r = e / width, c = e % width;
            if (r0 + r < R) Qp[(size_t)(r0 + r) * n + w0 + c] = sm.q[r][c];
        }
    }
    if (timing && lane == 0 && (wave == 0 || wave == first_helper)) {
#pragma unroll
        for (int k = 0; k < 12; ++k)
            if (tacc[k] && k != 9) g_win_cycles[k] += tacc[k];
        if (wave == 0) g_win_cycles[9] += (long long)__builtin_readcyclecounter() - tstart;
    }
}

}  // namespace slk
