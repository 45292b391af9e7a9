// a5 + a8 + a9 + a10: the column-sequential quantize / error-propagation loop
// (sleekit/obq.py:106-137, 202-213) with the reference's recursion and rounding points.
//
// Host side flattens the recursion of _quantize_opt_block into
//   LEAF(a, b)        quantize columns a..b-1 one at a time           (obq.py:106-118)
//   UPDATE(a, b, c)   Q[:, b:c] -= E[:, a:b] @ U[a:b, b:c]            (obq.py:137)
// and cuts it into WINDOWS: maximal sub-trees at most 512 columns wide.  A window runs in
// one kernel, a 16-row tile per workgroup with its Q and E columns resident in LDS; the
// updates that reach beyond a window run as chip-wide float64 MFMA GEMMs.
//
// Numerics follow the reference exactly: float32 quantizer (true divide, rint, separate
// multiply/add), err = float64(w - q) / U[i][i], rank-1 and block updates in float64
// (product rounded, then subtraction rounded), one rounding to float32 per update.
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <vector>

#include "mfma64.h"

namespace slk {

constexpr int OP_LEAF = 0, OP_UPDATE = 1;
constexpr int WMAX = 512;     // widest window held in LDS
constexpr int WPITCH = WMAX + 4;
constexpr int RB = 16;        // rows per window workgroup
constexpr int MAX_OPS = 60;   // ops per window kernel (kernel-argument table)
constexpr int ULEAF = 32;     // leaves up to this width stage their U block in LDS

struct Op {
    int kind, a, b, c;
    int m;   // UPDATE: columns [b, m) are needed at once, [m, c) may trail behind the next leaves
    int nl;  // UPDATE: number of leaves before [m, c) is touched again
};
struct OpTable {
    int count;
    Op op[MAX_OPS];
};

// ------------------------------------------------------------------ permute in / out
// Both are row gathers by a fixed column permutation.  A row goes through LDS: coalesced 16-byte loads in,
// the gather reads LDS (random 4-byte reads there cost a few cycles; straight from global every one of them
// was its own L1/L2 request: 2 TB/s), coalesced 16-byte stores out.  Rows longer than PERM_MAX floats take
// the plain kernels.
constexpr int PERM_MAX = 16384;

// Qp[r][c] = W[r][order[c]] (/ scale[r]);  E is cleared by the window kernels as they go.
__global__ __launch_bounds__(256) void k_permute_in(const float *__restrict__ W, const float *__restrict__ scale,
                                                    const long long *__restrict__ order, int R, int n,
                                                    float *__restrict__ Qp, int *__restrict__ inv_order, int rpl) {
    // (a batch of layers stacked by rows: rows [b rpl, (b + 1) rpl) follow order[b], inv_order[b])
    for (int r = blockIdx.x; r < R; r += gridDim.x) {
        const float *src = W + (size_t)r * n;
        float *dst = Qp + (size_t)r * n;
        const long long *ord = order ? order + (size_t)(r / rpl) * n : nullptr;
        if (scale) {
            const float s = scale[r];
            for (int c = threadIdx.x; c < n; c += blockDim.x) dst[c] = src[ord ? ord[c] : c] / s;
        } else {
            for (int c = threadIdx.x; c < n; c += blockDim.x) dst[c] = src[ord ? ord[c] : c];
        }
    }
    for (int b = blockIdx.x; b < (R + rpl - 1) / rpl; b += gridDim.x)
        for (int c = threadIdx.x; c < n; c += blockDim.x) inv_order[(size_t)b * n + (order ? order[(size_t)b * n + c] : c)] = c;
}
__global__ __launch_bounds__(256) void k_permute_in_lds(const float *__restrict__ W, const float *__restrict__ scale,
                                                        const long long *__restrict__ order, int R, int n,
                                                        float *__restrict__ Qp, int *__restrict__ inv_order, int rpl) {
    extern __shared__ __attribute__((aligned(16))) float row[];
    const int t = threadIdx.x, n4 = n >> 2;
    const long long *order_all = order;
    for (int r = blockIdx.x; r < R; r += gridDim.x) {
        order = order_all + (size_t)(r / rpl) * n;
        const float4v_t *src = reinterpret_cast<const float4v_t *>(W + (size_t)r * n);
        __syncthreads();  // the previous row's gathers are done
        for (int c = t; c < n4; c += 256) reinterpret_cast<float4v_t *>(row)[c] = src[c];
        __syncthreads();
        const float s = scale ? scale[r] : 1.0f;
        float4v_t *dst = reinterpret_cast<float4v_t *>(Qp + (size_t)r * n);
        for (int c = t; c < n4; c += 256) {
            float4v_t v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = row[order[4 * c + e]];
                v[e] = scale ? x / s : x;
            }
            dst[c] = v;
        }
    }
    for (int b = blockIdx.x; b < (R + rpl - 1) / rpl; b += gridDim.x)
        for (int c = t; c < n; c += 256) inv_order[(size_t)b * n + order_all[(size_t)b * n + c]] = c;
}

// Q[r][j] = Qp[r][inv[j]];  idx[r][j] = grid index of that value (codebook.py:43-54).
__global__ __launch_bounds__(256) void k_permute_out(const float *__restrict__ Qp, const int *__restrict__ inv_order,
                                                     int R, int n, Grid g, const float *__restrict__ unscale,
                                                     float *__restrict__ Q, uint8_t *__restrict__ idx, int rpl) {
    for (int r = blockIdx.x; r < R; r += gridDim.x) {
        const float *src = Qp + (size_t)r * n;
        const int *inv_o = inv_order + (size_t)(r / rpl) * n;
        const float inv = unscale ? 1.0f / unscale[r] : 1.0f;  // scaling.py:80: a division by the reciprocal
        for (int j = threadIdx.x; j < n; j += blockDim.x) {
            const float v = src[inv_o[j]];
            Q[(size_t)r * n + j] = unscale ? v / inv : v;
            if (idx) idx[(size_t)r * n + j] = (uint8_t)cb_index(v, g);
        }
    }
}
__global__ __launch_bounds__(256) void k_permute_out_lds(const float *__restrict__ Qp, const int *__restrict__ inv_order,
                                                         int R, int n, Grid g, const float *__restrict__ unscale,
                                                         float *__restrict__ Q, uint8_t *__restrict__ idx, int rpl) {
    extern __shared__ __attribute__((aligned(16))) float row[];
    const int t = threadIdx.x, n4 = n >> 2;
    const int *inv_all = inv_order;
    for (int r = blockIdx.x; r < R; r += gridDim.x) {
        inv_order = inv_all + (size_t)(r / rpl) * n;
        const float4v_t *src = reinterpret_cast<const float4v_t *>(Qp + (size_t)r * n);
        __syncthreads();
        for (int c = t; c < n4; c += 256) reinterpret_cast<float4v_t *>(row)[c] = src[c];
        __syncthreads();
        float4v_t *dst = reinterpret_cast<float4v_t *>(Q + (size_t)r * n);
        unsigned *di = idx ? reinterpret_cast<unsigned *>(idx + (size_t)r * n) : nullptr;
        const float inv = unscale ? 1.0f / unscale[r] : 1.0f;  // scaling.py:80: a division by the reciprocal
        for (int c = t; c < n4; c += 256) {
            float4v_t v;
            unsigned packed = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = row[inv_order[4 * c + e]];
                if (idx) packed |= (unsigned)(cb_index(v[e], g) & 255) << (8 * e);
                if (unscale) v[e] = v[e] / inv;
            }
            dst[c] = v;
            if (idx) di[c] = packed;
        }
    }
}

// ------------------------------------------------------------------ window kernel
struct LeafTables {
    double u[ULEAF][ULEAF + 1];  // leaf block of U, STRICTLY upper part (zero on/below the diagonal and beyond the width)
    double udr[ULEAF][2];        // its diagonal (1 beyond the width) and 1 / diagonal by true division
};
struct WindowSmem {
    // leaf tables first: their offsets fit the 16-bit immediate of ds_read, so the unrolled leaf
    // needs one address register instead of one per step.  Two copies: the helper waves fill the
    // tables of the next leaf while the chain waves read those of the current one.
    LeafTables lt[2];
    int odd[2][4];  // per helper wave: a diagonal entry of that block defeats the exact-division shortcut
    float q[RB][WPITCH];
    float e[RB][WPITCH];
    float cbt[512];  // a general codebook's values and limits (<= 256 entries), copied here for the leaves
};

// codebook.py:56-65 with the divide replaced by Markstein's sequence: with y = RN(1/step),
//   q0 = RN(t0 y);  r = t0 - step q0 (exact in one fma);  t = RN(q0 + r y) == RN(t0 / step)
// for every t0 whose quotient neither overflows nor underflows (rint of an underflowing
// quotient is 0 either way).  Saves ~40 dependent cycles per column on the leaf's critical
// path.  tests/test_gpu_parity.py::test_fast_quantizer_matches_true_divide sweeps it against
// the true-divide kernel.
__device__ __forceinline__ float grid_value_fast(float x, const Grid g, float inv_step) {
    const float t0 = x - g.zero;
    const float q0 = t0 * inv_step;
    const float r = __builtin_fmaf(-g.step, q0, t0);
    float t = __builtin_fmaf(r, inv_step, q0);
    t = rintf(t);
    t = fminf(fmaxf(t, 0.0f), g.top);
    return t * g.step + g.zero;
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// lane L of every 16-lane row, to all lanes of that row (DPP row_newbcast, gfx90a and later)
template <int L>
__device__ __forceinline__ float row_bcast(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + L, 0xf, 0xf, true));
}
// grid_value_fast with the clamp as one v_med3 (the leaf chain counts instructions)
__device__ __forceinline__ float grid_value_fast_med3(float x, const Grid g, float inv_step) {
    const float t0 = x - g.zero;
    const float q0 = t0 * inv_step;
    const float r = __builtin_fmaf(-g.step, q0, t0);
    float t = __builtin_fmaf(r, inv_step, q0);
    t = rintf(t);
    t = __builtin_amdgcn_fmed3f(t, 0.0f, g.top);
    return t * g.step + g.zero;
}

// LEAF, register path (width <= 32, window in LDS), run by waves 0-3 of the workgroup: a wave
// owns FOUR rows, 16 lanes per row; lane c keeps columns a + c and a + 16 + c of its row in two
// registers.  Step i: column i's value is read-laned out of the four rows and selected per
// row group, every lane recomputes q_i and err_i for its own row (no second broadcast), then
// updates its two columns:  x_c <- float32(float64(x_c) - err_i * U[i][c])   (obq.py:114-118)
// err_i = float64(x_i - q_i) / U[i][i] uses the exact-division fma sequence with reciprocals
// taken once per leaf by a true division.  The loop is issue-bound (~35 instructions a step),
// which is why four rows share a wave and the other four waves of the workgroup stay parked
// at the barrier: two waves per SIMD would just take turns (measured 430 -> ~230 cycles/step).
template <int NSTEP>
__device__ __forceinline__ void leaf_registers(WindowSmem &sm, const LeafTables &lt, int wave, int lane, int a_rel, int w,
                                               const Grid g, float inv_step) {
    const int c16 = lane & 15, rg = lane >> 4;
    const int row = 4 * wave + rg;
    const bool m0 = c16 < w, m1 = c16 + 16 < w;
    float x0 = m0 ? sm.q[row][a_rel + c16] : 0.0f, x1 = m1 ? sm.q[row][a_rel + 16 + c16] : 0.0f;
    float q0 = 0.0f, q1 = 0.0f, e0 = 0.0f, e1 = 0.0f;
    // Steps beyond the width run on the padding (x = 0, U row = 0, diagonal = 1) and change
    // nothing: no per-step branch, so the whole leaf is one basic block and the LDS reads of
    // step i + 1 (U row, diagonal, reciprocal) are issued before the arithmetic of step i.
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
        // the chain values pass through this point: the reads above are issued before step i starts
        asm volatile("" : "+v"(x0), "+v"(x1)::"memory");
        constexpr int src = i & 15;
        // column i of each of the wave's four rows, broadcast inside its 16-lane DPP row
        const float xi = row_bcast<src>(i < 16 ? x0 : x1);
        const float q = grid_value_fast(xi, g, inv_step);
        const double d = (double)(xi - q);
        const double qq = d * rii;
        const double rem = __builtin_fma(-uii, qq, d);
        const double err = __builtin_fma(rem, rii, qq);
        const float ef = (float)err;
        const bool here = c16 == src;
        if (i < 16) {
            q0 = here ? q : q0;
            e0 = here ? ef : e0;
        } else {
            q1 = here ? q : q1;
            e1 = here ? ef : e1;
        }
        // the staged block is zero on and below the diagonal: only later columns move
        if (i < 15) x0 = (float)((double)x0 - err * u0);
        if (NSTEP > 16) x1 = (float)((double)x1 - err * u1);
    });
    if (m0) {
        sm.q[row][a_rel + c16] = q0;
        sm.e[row][a_rel + c16] = e0;
    }
    if (m1) {
        sm.q[row][a_rel + 16 + c16] = q1;
        sm.e[row][a_rel + 16 + c16] = e1;
    }
}

// cycle counters of workgroup 0 (SLK_WIN_DBG bit 3), read back by slk_probe_window_cycles
__device__ long long g_win_cycles[16];
__device__ long long g_win_trace[64];  // window2: busy cycles per period, chain wave 0 / helper wave 2 (+32)

// One workgroup = 512 threads = 8 waves = RB rows, Q and E of the window resident in LDS.
//
// LEAF (fast path): waves 0-3, the CHAIN waves, run the column chain, four rows each.  Waves 4-7,
// the HELPERS, meanwhile (a) write the U tables of the next leaf into the other LDS buffer and
// fetch those of the one after, (b) run a share of the DEFERRED part of the last update on the MFMA
// pipe: UPDATE(a, b, c) is split at m, the end of the sub-tree that follows it -- columns [b, m) are
// needed at once (urgent, all eight waves, the chain waits for them), columns [m, c) are not touched
// again before the next update with the same c and are folded in behind the chain's back, spread
// over the `nl` leaves in between.  Every column still sees the same updates in the same order,
// each rounded to float32 once, so the result is the reference's bit for bit.
// UPDATE: the target columns are cut in 16-wide MFMA blocks (M = the 16 rows of the tile), dealt
// round-robin to the participating waves, K in chunks of 64.
template <bool IN_LDS>
__global__ __launch_bounds__(512) void k_gptq_window(float *__restrict__ Qp, float *__restrict__ Eg,
                                                     const double *__restrict__ U, int R, int n, int w0, int w1,
                                                     Grid g, float inv_step, int fast_ok, int dbg, OpTable tab, int rpl) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    WindowSmem &sm = *reinterpret_cast<WindowSmem *>(smem_raw);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool helper = wave >= 4;
    const int ht = t - 256;  // helper thread index
    const int r0 = blockIdx.x * RB;
    U += (size_t)(r0 / rpl) * n * n;  // a batch of layers stacked by rows: rows [b rpl, (b + 1) rpl) use factor b
    if (IN_LDS && g.table) {  // the leaves search the codebook once per column: keep it next to them
        for (int i = t; i < 2 * g.n - 1; i += 512) sm.cbt[i] = g.table[i];
        g.table = sm.cbt;  // visible after the first barrier below
    }
    const int width = w1 - w0;
    // cycle accounting (debug): wave-uniform accumulators, written out once at the end
    const bool timing = (dbg & 8) && blockIdx.x == 0;
    long long tmark = timing ? (long long)__builtin_readcyclecounter() : 0;
    const long long tstart = tmark;
    long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // adds the cycles since the last mark to counter `slot` (wave `w` only)
    auto lap = [&](int slot, int w) {
        if (timing) {
            const long long now = (long long)__builtin_readcyclecounter();
            if (wave == w) tacc[slot] += now - tmark;
            tmark = now;
        }
    };

    // tile accessors: LDS copy of the window, or the global arrays themselves (very wide leaves).
    // Plain accesses: every producer/consumer pair below is separated by a __syncthreads().
    auto qld = [&](int r, int c) -> float { return IN_LDS ? sm.q[r][c - w0] : Qp[(size_t)(r0 + r) * n + c]; };
    auto eld = [&](int r, int c) -> float { return IN_LDS ? sm.e[r][c - w0] : Eg[(size_t)(r0 + r) * n + c]; };
    auto qst = [&](int r, int c, float v) {
        if (IN_LDS) sm.q[r][c - w0] = v; else Qp[(size_t)(r0 + r) * n + c] = v;
    };
    auto est = [&](int r, int c, float v) {
        if (IN_LDS) sm.e[r][c - w0] = v; else Eg[(size_t)(r0 + r) * n + c] = v;
    };
    // columns [c_lo, c_hi) of the Q tile, global -> LDS, by `nth` threads of which this is number `tid`
    auto load_cols = [&](int c_lo, int c_hi, int tid, int nth) {
        const int cw = c_hi - c_lo;
        for (int e = tid; e < RB * cw; e += nth) {
            const int r = e / cw, c = c_lo + e % cw;
            sm.q[r][c - w0] = (r0 + r < R) ? Qp[(size_t)(r0 + r) * n + c] : 0.0f;
        }
    };

    // ---- staging pipeline of the leaf tables (helpers only): registers <- U two staged leaves
    // ahead, LDS buffer <- registers one leaf ahead.
    double pu[4] = {0.0, 0.0, 0.0, 0.0};
    int fetch_w = 0;     // width of the block held in pu, 0 = none
    int next_fetch = 0;  // op from which to look for the next staged leaf
    auto fetch_block = [&]() {
        fetch_w = 0;
        for (; next_fetch < tab.count; ++next_fetch) {
            const Op nx = tab.op[next_fetch];
            const int w = nx.b - nx.a;
            if (nx.kind != OP_LEAF || w > ULEAF) continue;
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int e = ht + 256 * h, i = min(e >> 5, w - 1), j = min(e & 31, w - 1);
                pu[h] = U[(size_t)(nx.a + i) * n + nx.a + j];  // clamped, selected when written
            }
            fetch_w = w;
            ++next_fetch;
            return;
        }
    };
    auto write_block = [&](int buf) {
        LeafTables &lt = sm.lt[buf];
        const int w = fetch_w;
        bool odd = false;  // exact-division exception: a diagonal entry whose significand is all ones
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int e = ht + 256 * h, i = e >> 5, j = e & 31;  // all 32 x 32 slots
            const bool in = i < w && j < w;
            lt.u[i][j] = (in && j > i) ? pu[h] : 0.0;
            if (i == j) {
                const double dg = in ? pu[h] : 1.0;
                lt.udr[i][0] = dg;
                lt.udr[i][1] = 1.0 / dg;
                odd = odd || (__double_as_longlong(dg) & 0xFFFFFFFFFFFFFLL) == 0xFFFFFFFFFFFFFLL;
            }
        }
        const bool any = __builtin_amdgcn_ballot_w64(odd) != 0;
        if (lane == 0) sm.odd[buf][wave - 4] = any ? 1 : 0;
    };

    // Warm this XCD's L2 with the window's block of U (workgroup i runs on XCD i % 8; the
    // workgroups of an XCD share the lines, one 4-byte touch per 128-byte line).
    float warm = 0.0f;
    if (IN_LDS && !(dbg & 16)) {
        const int per_xcd = max(1, min(32, (int)gridDim.x >> 3));
        const int slice = (blockIdx.x >> 3) % per_xcd;
        const int lpr = (width + 15) >> 4, total = width * lpr;
        const int per_slice = (total + per_xcd - 1) / per_xcd;
        for (int id = slice * per_slice + t; id < min(total, (slice + 1) * per_slice); id += 512) {
            const int row = id / lpr, c0 = (id % lpr) << 4;
            if (c0 + 15 >= row) warm += reinterpret_cast<const float *>(U + (size_t)(w0 + row) * n + w0 + min(c0, width - 1))[0];
        }
    }

    // Prologue: only the first leaf's columns are needed before the first barrier; the rest of the
    // tile is loaded by the helpers during that leaf.
    int rest_from = w1;  // columns [rest_from, w1) of the tile still to be loaded
    if (IN_LDS) {
        if (helper) fetch_block();
        const int first_end = (tab.op[0].kind == OP_LEAF && !(dbg & 32)) ? tab.op[0].b : w1;
        load_cols(w0, first_end, t, 512);
        rest_from = first_end;
        if (helper) {
            if (fetch_w) write_block(0);
            fetch_block();
        }
    }
    __syncthreads();
    lap(7, 0);

    // ---- UPDATE machinery: Q[:, lo:hi] -= E[:, a:b] @ U[a:b, lo:hi] on the 16 rows of this tile.
    // A ROUND is (16-column block, 64-deep K chunk).  Operands of U come straight from global memory
    // (every workgroup streams the same panel out of L2), so the loads of round r+1 are issued before
    // the MFMAs of round r, and the chain waves issue round 0 of the urgent part BEFORE their leaf.
    const int lr = lane & 15, lk = lane >> 4;
    const bool row_ok = IN_LDS || r0 + lr < R;
    double cur[16];
    bool primed = false;
    auto load_round = [&](int a, int b, int lo, int hi, int blk, int kc, double(&bv)[16]) {
        // Clamped addresses, no masking: rows beyond b meet a zero E operand, columns beyond hi are
        // never stored -- and any arithmetic on the loaded value here would make the wave wait for
        // the load at once instead of after the MFMAs of the round before.
        const int cc = min(lo + blk * 16 + lr, hi - 1);
#pragma unroll
        for (int s4 = 0; s4 < 16; ++s4) {
            const int k = a + 64 * kc + 4 * s4 + lk;
            bv[s4] = U[(size_t)min(k, b - 1) * n + cc];
        }
    };
    // one round of MFMAs: acc += E[:, chunk kc of a:b] @ bv.  Full and half chunks (K = 64, 32: all
    // the reference's default schedules) take straight-line code: 16 (8) LDS reads, then the MFMAs.
    auto mac_round = [&](int a, int b, int kc, const double(&bv)[16], double4_t &acc) {
        const int kbase = a + 64 * kc, kcount = min(64, b - kbase);
        if (IN_LDS && (kcount == 64 || kcount == 32)) {
            const float *ep = &sm.e[lr][kbase - w0 + lk];
            float av[16];
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4) av[s4] = ep[4 * s4];
            if (kcount == 64) {
#pragma unroll
                for (int s4 = 8; s4 < 16; ++s4) av[s4] = ep[4 * s4];
            }
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av[s4], bv[s4], acc, 0, 0, 0);
            if (kcount == 64) {
#pragma unroll
                for (int s4 = 8; s4 < 16; ++s4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av[s4], bv[s4], acc, 0, 0, 0);
            }
        } else {
            const int er = IN_LDS ? lr : min(lr, R - 1 - r0);  // clamped: the load is unconditional, the value is selected
#pragma unroll
            for (int s4 = 0; s4 < 16; ++s4) {
                const int k = kbase + 4 * s4 + lk;
                const float ev = eld(er, min(k, b - 1));
                const double av = (k < b && row_ok) ? (double)ev : 0.0;
                if (4 * s4 < kcount) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[s4], acc, 0, 0, 0);
            }
        }
    };
    auto store_block = [&](int lo, int hi, int blk, const double4_t &acc) {
        const int col = lo + blk * 16 + lr;
        if (col < hi) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = lk + 4 * r;
                if (IN_LDS || r0 + row < R) qst(row, col, (float)((double)qld(row, col) - acc[r]));
            }
        }
    };
    // blocks wid, wid + nw, ... of [lo, hi); `ready`: cur already holds (block wid, chunk 0).
    // Two register buffers in ping-pong, so that the loads of a round are in flight during the
    // MFMAs of the round before and no copy ties the two together.
    auto run_update = [&](int a, int b, int lo, int hi, int wid, int nw, bool ready) {
        const int nblk = (hi - lo + 15) / 16, nchunk = (b - a + 63) / 64;
        if (wid >= nblk) return;
        const int nr = (nblk - wid + nw - 1) / nw * nchunk;
        double other[16];
        if (!ready) load_round(a, b, lo, hi, wid, 0, cur);
        double4_t acc = {0.0, 0.0, 0.0, 0.0};
        const double4_t zero4 = {0.0, 0.0, 0.0, 0.0};
        int blk = wid, kc = 0;
        for (int r = 0; r < nr; r += 2) {
            int blk1 = blk, kc1 = kc + 1;
            if (kc1 == nchunk) kc1 = 0, blk1 += nw;
            const bool has1 = r + 1 < nr;
            if (has1) load_round(a, b, lo, hi, blk1, kc1, other);
            mac_round(a, b, kc, cur, acc);
            if (kc + 1 == nchunk) {
                store_block(lo, hi, blk, acc);
                acc = zero4;
            }
            if (!has1) break;
            int blk2 = blk1, kc2 = kc1 + 1;
            if (kc2 == nchunk) kc2 = 0, blk2 += nw;
            if (r + 2 < nr) load_round(a, b, lo, hi, blk2, kc2, cur);
            mac_round(a, b, kc1, other, acc);
            if (kc1 + 1 == nchunk) {
                store_block(lo, hi, blk1, acc);
                acc = zero4;
            }
            blk = blk2, kc = kc2;
        }
    };
    // generic leaf (any width, true divides): the whole workgroup in lockstep, 32 lanes per row,
    // one barrier per column.  Rare (width > 32, or a diagonal hitting the exact-division
    // exception), so simplicity wins.
    auto generic_leaf = [&](int a, int b, bool staged, int buf) {
        const LeafTables &lt = sm.lt[buf];
        const int myrow = 2 * wave + (lane >> 5), l = lane & 31;
        const bool live = IN_LDS || (r0 + myrow < R);
        for (int i = a; i < b; ++i) {
            float x = 0.0f, q = 0.0f;
            double err = 0.0;
            if (live) {
                x = qld(myrow, i);
                q = cb_value(x, g);
                const double uii = staged ? lt.udr[i - a][0] : U[(size_t)i * n + i];
                err = (double)(x - q) / uii;
            }
            __syncthreads();  // everyone has read column i before lane 0 overwrites it
            if (live) {
                for (int j = i + 1 + l; j < b; j += 32) {
                    const double uij = staged ? lt.u[i - a][j - a] : U[(size_t)i * n + j];
                    const double p = err * uij;
                    qst(myrow, j, (float)((double)qld(myrow, j) - p));
                }
                if (l == 0) {
                    est(myrow, i, (float)err);
                    qst(myrow, i, q);
                }
            }
            __syncthreads();
        }
    };

    // deferred part of the last update: Q[:, pm:pc] -= E[:, pa:pb] @ U[pa:pb, pm:pc], of which the
    // first pdone 16-column blocks are done and the rest is due within pleaves more leaves
    int pa = 0, pb = 0, pm = 0, pc = 0, pdone = 0, pleaves = 1;
#ifdef SLK_WINDOW_EXPERIMENTS  // measurement builds only: these bits switch parts of the work OFF (wrong values)
    const bool no_updates = dbg & 2, no_leaves = dbg & 1, no_leaf_regs = dbg & 4;
#else
    constexpr bool no_updates = false, no_leaves = false, no_leaf_regs = false;
#endif
    int sbuf = 0;  // LDS buffer holding the tables of the next staged leaf

    // One pass per op plus a final pass that folds in whatever is still pending.  Each pass: at most
    // one update job per wave (ONE call site of run_update: it is big), then the leaf, then a barrier.
    for (int oi = 0; oi <= tab.count; ++oi) {
        const bool fin = oi == tab.count;
        const Op op = tab.op[fin ? 0 : oi];
        const bool is_leaf = !fin && op.kind == OP_LEAF;
        if (!fin && (is_leaf ? no_leaves : no_updates)) continue;
        const int prem = (pc - pm + 15) / 16 - pdone;  // pending blocks
        const int plo = pm + 16 * pdone;
        // the update job of this wave in this pass: Q[:, ulo:uhi] -= E[:, ua:ub] @ U[ua:ub, ulo:uhi]
        int ua = 0, ub = 0, ulo = 0, uhi = 0, uwid = wave, unw = 8;
        bool uready = false, redo = false, use_fast = false, staged = false;
        if (is_leaf) {
            const int w = op.b - op.a;
            staged = IN_LDS && w <= ULEAF;
            use_fast = staged && fast_ok;
            if (use_fast) use_fast = (sm.odd[sbuf][0] | sm.odd[sbuf][1] | sm.odd[sbuf][2] | sm.odd[sbuf][3]) == 0;
            lap(1, 0);
            if (use_fast) {
                // this leaf's share of the pending blocks, in whole turns of the four helpers
                const int take = prem > 0 ? min(prem, ((prem + pleaves - 1) / pleaves + 3) & ~3) : 0;
                if (helper) {
                    if (rest_from < w1) load_cols(rest_from, w1, ht, 256);
                    if (fetch_w) write_block(sbuf ^ 1);
                    fetch_block();
                    ua = pa, ub = pb, ulo = plo, uhi = min(pc, plo + 16 * take), uwid = wave - 4, unw = 4;
                } else if (oi + 1 < tab.count && tab.op[oi + 1].kind == OP_UPDATE && !no_updates) {
                    const Op nx = tab.op[oi + 1];
                    load_round(nx.a, nx.b, nx.b, nx.m, wave, 0, cur);
                    primed = true;
                }
                pdone += take;
                pleaves = max(1, pleaves - 1);
            } else {
                if (rest_from < w1) load_cols(rest_from, w1, t, 512);
                if (helper && staged) {
                    if (fetch_w) write_block(sbuf ^ 1);
                    fetch_block();
                }
                ua = pa, ub = pb, ulo = plo, uhi = pc;  // everything pending, all eight waves
                pdone += prem;
            }
            rest_from = w1;
        } else if (fin) {
            const bool rest = rest_from < w1;  // no leaf ran (debug modes)
            if (rest) load_cols(rest_from, w1, t, 512);
            rest_from = w1;
            if (prem <= 0 && !rest) break;
            ua = pa, ub = pb, ulo = plo, uhi = pc;
            pdone += prem;
        } else if (prem > 0 && ((op.c > plo && op.b < pc) || op.m < op.c)) {
            // an update that meets columns still pending, or has a deferred part of its own while the
            // older one is unfinished (never under the host's leaf counts, unless leaves were
            // skipped): fold the pending columns in first, then come back
            ua = pa, ub = pb, ulo = plo, uhi = pc;
            pdone += prem;
            redo = true;
        } else {
            ua = op.a, ub = op.b, ulo = op.b, uhi = op.m, uready = primed;
            primed = false;
            // an update without a deferred part (a small one between two leaves) leaves the older
            // deferral alone: that one keeps trailing behind the following leaves
            if (op.m < op.c) pa = op.a, pb = op.b, pm = op.m, pc = op.c, pdone = 0, pleaves = max(1, op.nl);
        }
        if (!is_leaf) lap(10, 0);
        if (ulo < uhi) run_update(ua, ub, ulo, uhi, uwid, unw, uready);
        if (is_leaf) {
            if (use_fast) {
                if (!helper && !no_leaf_regs) {
                    const int w = op.b - op.a;
                    if (w <= 16) leaf_registers<16>(sm, sm.lt[sbuf], wave, lane, op.a - w0, w, g, inv_step);
                    else leaf_registers<32>(sm, sm.lt[sbuf], wave, lane, op.a - w0, w, g, inv_step);
                }
                lap(0, 0);
                lap(5, 4);
            } else {
                __syncthreads();
                generic_leaf(op.a, op.b, staged, sbuf);
            }
            if (staged) sbuf ^= 1;
        } else {
            lap(3, 0);
        }
        __syncthreads();
        if (is_leaf) {
            lap(2, 0);
            lap(6, 4);
        } else {
            lap(4, 0);
        }
        if (redo) --oi;
    }

    if (IN_LDS) {
        for (int e = t; e < RB * width; e += 512) {
            const int r = e / width, c = e % width;
            if (r0 + r < R) {
                Qp[(size_t)(r0 + r) * n + w0 + c] = sm.q[r][c];
                Eg[(size_t)(r0 + r) * n + w0 + c] = sm.e[r][c];
            }
        }
    }
    if (warm == 1.2345e-30f) g_win_cycles[15] = 1;  // keeps the touches alive
    lap(8, 0);
    if (timing && lane == 0) {
        if (wave == 0) tacc[9] += (long long)__builtin_readcyclecounter() - tstart;
        if (wave == 0 || wave == 4) {
#pragma unroll
            for (int k = 0; k < 12; ++k)
                if (tacc[k]) g_win_cycles[k] += tacc[k];
        }
    }
}

}  // namespace slk
#include "window2.h"
namespace slk {

// ------------------------------------------------------------------ trailing update
// Qp[:, ja:jb] = float32(float64(Qp[:, ja:jb]) - E[:, ka:kb] @ U[ka:kb, ja:jb]), 64 x 64 tiles.
__global__ __launch_bounds__(256) void k_gptq_trailing(float *__restrict__ Qp, const float *__restrict__ Eg,
                                                       const double *__restrict__ U, int R, int n, int ka, int kb,
                                                       int ja, int jb, int vec_ok, int rpl) {
#ifdef SLK_TRAILING_F64_IMAGE
    __shared__ __attribute__((aligned(16))) Tile64Smem sm;
#else
    __shared__ __attribute__((aligned(16))) Tile64SmemAf sm;  // E stays float32 in LDS (mfma64.h)
#endif
    const int r0 = blockIdx.y * TILE, j0 = ja + blockIdx.x * TILE;
    U += (size_t)(r0 / rpl) * n * n;  // batch of layers stacked by rows (rpl a multiple of the tile)
    const int t = threadIdx.x;
    Acc64 acc;
    acc.zero();
    const int kend = ka + (kb - ka + KSTEP - 1) / KSTEP * KSTEP;
    const int a_row = r0 + (t >> 2), a_k = (t & 3) * 8;  // A = E (float32), K contiguous
    const int b_k = t >> 3, b_c2 = (t & 7) * 2;          // B = U (float64), columns contiguous: pieces at 16 h + b_c2
    // interior tiles take the unguarded loaders over the whole K-steps; a ragged last step (K = 172 at n = 11008)
    // and the edge tiles take the guarded ones
    int k_fast = ka;
    if (vec_ok && r0 + TILE <= R && j0 + TILE <= jb) {
        k_fast = ka + (kb - ka) / KSTEP * KSTEP;
        const float *pe = Eg + (size_t)a_row * n + a_k;
        const double *pu = U + (size_t)b_k * n + j0 + b_c2;
        tile64_mac<true, float>(
            acc, sm, ka, k_fast, [&](int k0, float(&v)[8]) { load8f<true>(pe + k0, v); },
            [&](int k0, double(&v)[8]) { load8d_cols(pu + (size_t)k0 * n, v); });
    }
    if (k_fast < kend) {
        const bool row_ok = a_row < R;
        const float *pe = Eg + (size_t)min(a_row, R - 1) * n;
        tile64_mac<true, float>(
            acc, sm, k_fast, kend,
            [&](int k0, float(&v)[8]) {
                const int k = k0 + a_k;
                load8f_guarded(pe + min(k, kb - 1), kb - 1 - k, row_ok, v);
            },
            [&](int k0, double(&v)[8]) {
                const int k = k0 + b_k;
                load8d_cols_guarded(U + (size_t)min(k, kb - 1) * n + j0, b_c2, jb - 1 - j0, k < kb, v);
            });
    }
    if (r0 + TILE <= R && j0 + TILE <= jb) {
        // interior tile: ALL sixteen old values of a thread first, then the sixteen stores.  Element by element (below) every
        // guarded `*p = *p - v` is a load -> s_waitcnt vmcnt(0) -> store round trip of its own, sixteen in a row at the end of
        // every tile: the compiler cannot tell the addresses apart.
        float *q0 = Qp + (size_t)r0 * n + j0;
        float oldq[16];
        int e = 0;
        tile64_foreach(acc, [&](int r, int c, double) { oldq[e++] = q0[(size_t)r * n + c]; });
        e = 0;
        tile64_foreach(acc, [&](int r, int c, double v) { q0[(size_t)r * n + c] = (float)((double)oldq[e++] - v); });
        return;
    }
    tile64_foreach(acc, [&](int r, int c, double v) {
        if (r0 + r < R && j0 + c < jb) {
            float *p = Qp + (size_t)(r0 + r) * n + j0 + c;
            *p = (float)((double)*p - v);
        }
    });
}

// ------------------------------------------------------------------ host planning
static void flatten(int a, int b, int mb, int nb, std::vector<Op> &ops) {
    const int size = b - a;
    if (size <= mb) {
        ops.push_back({OP_LEAF, a, b, 0, 0, 0});
        return;
    }
    int step = (size + nb - 1) / nb;
    if (step < mb) step = mb;
    for (int s = a; s < b; s += step) {
        const int e = s + step < b ? s + step : b;
        flatten(s, e, mb, nb, ops);
        if (e < b) ops.push_back({OP_UPDATE, s, e, b, b, 1});
    }
}

struct Plan {
    // kind 0: window [a, b) with ops; kind 1: chip-wide update (a, b, c)
    struct Step {
        int kind, a, b, c;
        std::vector<Op> ops;
    };
    std::vector<Step> steps;
};

static void plan(int a, int b, int mb, int nb, Plan &p) {
    const int size = b - a;
    std::vector<Op> ops;
    flatten(a, b, mb, nb, ops);
    if (size <= mb || (size <= WMAX && (int)ops.size() <= MAX_OPS)) {
        p.steps.push_back({0, a, b, 0, ops});
        return;
    }
    int step = (size + nb - 1) / nb;
    if (step < mb) step = mb;
    for (int s = a; s < b; s += step) {
        const int e = s + step < b ? s + step : b;
        plan(s, e, mb, nb, p);
        if (e < b) p.steps.push_back({1, s, e, b, {}});
    }
}

// A window's ops as periods (window2.h), or false when the window is not of the standard shape.
static bool as_periods(const std::vector<Op> &ops, int wa, int wb, PeriodTable &pt) {
    pt.count = 0;
    size_t i = 0;
    while (i < ops.size()) {
        if (pt.count == MAXP || ops[i].kind != OP_LEAF) return false;
        Period P = {ops[i].a, ops[i].b - ops[i].a, 0, 0};
        ++i;
        if (i + 1 < ops.size() && ops[i].kind == OP_UPDATE && ops[i].a == P.s && ops[i].b == P.s + P.w1 &&
            ops[i + 1].kind == OP_LEAF && ops[i + 1].a == P.s + P.w1 && ops[i].c == ops[i + 1].b) {
            P.w2 = ops[i + 1].b - ops[i + 1].a;
            i += 2;
        }
        const int K = P.w1 + P.w2;
        if (i < ops.size()) {
            if (ops[i].kind != OP_UPDATE || ops[i].a != P.s || ops[i].b != P.s + K || ops[i].c != wb) return false;
            ++i;
        } else if (P.s + K != wb) {
            return false;
        }
        if (P.w1 > ULEAF || P.w2 > ULEAF || (P.s & 1) || (P.w1 & 1) || (P.w2 & 1) || P.w1 < 2) return false;
        pt.p[pt.count++] = P;
    }
    if (pt.count == 0 || pt.p[0].s != wa) return false;
    for (int k = 0; k + 1 < pt.count; ++k) {
        if (pt.p[k + 1].s != pt.p[k].s + pt.p[k].w1 + pt.p[k].w2) return false;
        pt.p[k].nw = pt.p[k + 1].w1 + pt.p[k + 1].w2;
    }
    return true;
}

}  // namespace slk

using namespace slk;

namespace slk {
// the leaf chain alone: `iters` leaves of 32 columns on the tables of a made-up block; out[0] = cycles
// of wave 0, out[1] = checksum.  blockDim = 256 (one wave per SIMD) or 512 (two).
__global__ void k_probe_leaf(double *out, int iters, int mode, Grid g, float inv_step) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    WindowSmem &sm = *reinterpret_cast<WindowSmem *>(smem_raw);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int e = t; e < 32 * 32; e += blockDim.x) {
        const int i = e >> 5, j = e & 31;
        sm.lt[0].u[i][j] = j > i ? 0.01 * ((i * 7 + j * 3) % 11 - 5) : 0.0;
        if (i == j) sm.lt[0].udr[i][0] = 1.0 + 0.01 * i, sm.lt[0].udr[i][1] = 1.0 / (1.0 + 0.01 * i);
    }
    for (int e = t; e < RB * 64; e += blockDim.x) sm.q[e >> 6][e & 63] = 0.37f * ((e * 13) % 17 - 8);
    __syncthreads();
    const long long t0 = (long long)__builtin_readcyclecounter();
    double4_t macc = {0, 0, 0, 0};
    double vacc = 1.0;
    // modes 1-3: leaf waves 0-3, companions 4-7 (same SIMDs if waves go round the SIMDs);
    // mode 4: leaf waves 0, 1, 4, 5 and MFMA companions 2, 3, 6, 7 (other SIMDs under that mapping)
    const bool leafer = mode == 4 ? (wave & 2) == 0 : (wave < 4 || mode == 0);
    if (mode == 4) mode = leafer ? 0 : 1;
    if (t == 0) out[2] = (double)__builtin_amdgcn_s_getreg((4 << 11) | (0 << 6) | 4);  // HW_ID, 32 bits... low 16 here
    if (leafer || mode == 0) {
        // (with more than two leaf waves per SIMD several waves run the same four rows: garbage values, honest timing)
        for (int it = 0; it < iters; ++it) leaf_registers<32>(sm, sm.lt[0], ((wave & 1) + ((wave >> 2) << 1)) & 3, lane, (it & 1) * 32, 32, g, inv_step);
    } else if (mode == 5) {
        // companion wave on the same SIMD: back-to-back bfloat16 MFMAs (what the layer-error kernel issues)
        typedef __bf16 probe_bf16x8_t __attribute__((ext_vector_type(8)));
        typedef float probe_f32x16_t __attribute__((ext_vector_type(16)));
        probe_bf16x8_t av, bv;
        for (int k = 0; k < 8; ++k) av[k] = (__bf16)(1.0f + lane), bv[k] = (__bf16)2.0f;
        probe_f32x16_t c16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int it = 0; it < iters * 64; ++it) c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c16, 0, 0, 0);
        vacc = c16[0];
    } else if (mode == 1) {
        // companion wave on the same SIMD: back-to-back 16x16x4 MFMAs for about as long
        for (int it = 0; it < iters * 64; ++it) macc = __builtin_amdgcn_mfma_f64_16x16x4f64(1.0 + lane, 2.0, macc, 0, 0, 0);
    } else if (mode == 2) {
        // companion wave: a dependent float64 fma chain
        for (int it = 0; it < iters * 400; ++it) vacc = __builtin_fma(vacc, 1.0000001, 1e-9);
    } else {
        // companion wave: LDS reads
        for (int it = 0; it < iters * 100; ++it) vacc += sm.lt[1].u[(it + lane) & 31][lane & 31];
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    __shared__ long long leaf_cycles[16];
    if (lane == 0) leaf_cycles[wave] = leafer || mode == 0 ? t1 - t0 : 0;
    __syncthreads();
    if (t == 0) {
        out[0] = (double)(t1 - t0);
        out[1] = sm.q[3][5] + sm.e[2][7];
        long long slowest = 0;  // the oldest wave of a SIMD issues first: wave 0 alone says nothing about the others
        for (int w = 0; w < (int)blockDim.x / 64; ++w) slowest = leaf_cycles[w] > slowest ? leaf_cycles[w] : slowest;
        out[3] = (double)slowest;
    }
    if (macc[0] + vacc == 1.2345e-30) out[1] = macc[0];
}
}  // namespace slk

extern "C" int slk_probe_leaf_chain(double *out, int iters, int waves_per_simd, slk_stream_t stream) {
    // waves_per_simd: 1, 2, 3, 4 (all leaf waves), or 2 + 10 * mode for companions (1 float64 MFMA, 2 fma chain, 3 LDS reads,
    // 5 bfloat16 MFMA)
    const int mode = waves_per_simd / 10;
    waves_per_simd %= 10;
    SLK_REQUIRE(out && iters > 0 && waves_per_simd >= 1 && waves_per_simd <= 4 && mode >= 0 && mode <= 5, "bad arguments");
    hipStream_t s = as_stream(stream);
    const Grid g = make_grid(8, -1.0, 1.0);
    SLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_probe_leaf), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)sizeof(WindowSmem)));
    SLK_RUN("probe_leaf", 0, 0, s, k_probe_leaf<<<1, 256 * waves_per_simd, sizeof(WindowSmem), s>>>(out, iters, mode, g, 1.0f / g.step));
    return SLK_OK;
}

extern "C" int slk_probe_window_cycles(long long *host_out, int reset) {
    SLK_REQUIRE(host_out, "null pointer");
    SLK_HIP(hipDeviceSynchronize());
    SLK_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_win_cycles), sizeof(long long) * 16));
    SLK_HIP(hipMemcpyFromSymbol(host_out + 16, HIP_SYMBOL(g_win_trace), sizeof(long long) * 64));
    if (reset) {
        long long zero[64] = {0};
        SLK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_win_cycles), zero, sizeof(long long) * 16));
        SLK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_win_trace), zero, sizeof(zero)));
    }
    return SLK_OK;
}

extern "C" int slk_gptq_quantize(const float *W, const float *scale, const long long *order, const double *U,
                                 int R, int n, int levels, double lo, double hi, const float *table, int min_block, int num_blocks,
                                 int flags, float *Q, uint8_t *idx, float *E_out, void *workspace, size_t ws_bytes,
                                 slk_stream_t stream) {
    return slk_gptq_quantize_batch(W, scale, order, U, 1, R, n, levels, lo, hi, table, min_block, num_blocks, flags, Q, idx,
                                   E_out, workspace, ws_bytes, stream);
}

// `batch` layers of one shape stacked by rows: every launch of the loop covers all of them, each row tile
// reading its own layer's factor.  What a row shard of a multi-GPU run needs: R / G rows alone leave most of
// the chip idle (the window kernel runs one workgroup per 16 rows), G layers' shards together fill it.
extern "C" int slk_gptq_quantize_batch(const float *W, const float *scale, const long long *order, const double *U,
                                       int batch, int rows_per_layer, int n, int levels, double lo, double hi,
                                       const float *table, int min_block, int num_blocks, int flags, float *Q, uint8_t *idx,
                                       float *E_out, void *workspace, size_t ws_bytes, slk_stream_t stream) {
    SLK_REQUIRE(W && U && Q, "null pointer");
    SLK_REQUIRE((flags & ~(SLK_LOOP_UNSCALE | SLK_LOOP_LATENCY)) == 0, "unknown flags");
    const int unscale = flags & SLK_LOOP_UNSCALE;
    SLK_REQUIRE(!unscale || scale, "unscale needs the row scales");
    SLK_REQUIRE(rows_per_layer > 0 && n > 0, "empty layer");
    SLK_REQUIRE(batch >= 1 && batch <= 64, "batch must be 1..64");
    SLK_REQUIRE(batch == 1 || rows_per_layer % TILE == 0, "a batch needs rows_per_layer to be a multiple of 64");
    SLK_REQUIRE(batch == 1 || order, "a batch needs the column orders");
    SLK_REQUIRE((long long)batch * rows_per_layer <= 0x7fffffffLL, "too many rows");
    const int R = batch * rows_per_layer, rpl = rows_per_layer;
    SLK_REQUIRE(levels >= 2 && (table || lo < hi), "codebook needs levels >= 2 and lo < hi");
    SLK_REQUIRE(table == nullptr || levels <= 256, "general codebooks hold at most 256 entries");
    SLK_REQUIRE(idx == nullptr || levels <= 256, "uint8 indices need levels <= 256");
    SLK_REQUIRE(min_block >= 1 && num_blocks >= 1, "min_block_size and num_blocks must be >= 1");
    Arena ws(workspace, ws_bytes);
    float *Qp = ws.take<float>((size_t)R * n);
    float *Eg = ws.take<float>((size_t)R * n);
    int *inv_order = ws.take<int>((size_t)batch * n);
    if (!Qp || !Eg || !inv_order) {
        set_error("workspace too small for a %d x %d layer", R, n);
        return SLK_E_WS;
    }
    hipStream_t s = as_stream(stream);
    const Grid g = make_grid(levels, lo, hi, table);
    // exact-division shortcut of the leaf: needs a sane step whose significand is not all ones
    const float inv_step = 1.0f / g.step;
    unsigned step_bits;
    memcpy(&step_bits, &g.step, 4);
    const int fast_ok = table == nullptr && (step_bits & 0x7FFFFFu) != 0x7FFFFFu && g.step > 9.0e-13f && g.step < 1.0e12f &&
                        !opt(OPT_NO_FAST_LEAF);
#ifdef SLK_WINDOW_EXPERIMENTS
    const int dbg = opt(OPT_WIN_DBG);
#else
    const int dbg = opt(OPT_WIN_DBG) & (8 | 16 | 32 | 64 | 128);  // cycle counters, no L2 warm-up, whole tile loaded up front: same results
#endif
    const bool no_defer = opt(OPT_NO_DEFER) != 0;
    SLK_LDS_OPT_IN(k_gptq_window<true>, sizeof(WindowSmem));
    SLK_LDS_OPT_IN(k_gptq_window2<1>, sizeof(Window2SmemT<1>));
    SLK_LDS_OPT_IN(k_gptq_window2<2>, sizeof(Window2SmemT<2>));
    // 32 rows per workgroup (eight rows per chain wave, ONE quantizer instruction stream for them: leaf_chain8) halve the CUs
    // a window launch occupies for 1.45 times the duration (96 against 66 us at 4096 rows): less chip time per row, which is
    // what counts when other streams' kernels fill the CUs it leaves (round 3, every BASELINE stream on one MI355X: headline
    // 5040 -> 5180 Mweights/s, OPT-125M 5350 -> 5700, OPT-350M 4430 -> 4650, BLOOM-560M 3620 -> 3860) -- the default.  16
    // rows (four per chain wave) are faster for ONE layer alone: callers ask with SLK_LOOP_LATENCY (the single-layer API
    // does).  slk_set_option("window_rows", 16 | 32) forces either.
    // (fewer than 2048 rows -- the row shards of small layers on several ranks -- leave most CUs idle either way: what
    // counts then is the length of the launch chain, 16 rows again: one rank of 8 on OPT-125M 14.0 -> 12.6 ms per step)
    const int window_rows = opt(OPT_WINDOW_ROWS) == 16 || opt(OPT_WINDOW_ROWS) == 32 ? opt(OPT_WINDOW_ROWS)
                                                                                     : (((flags & SLK_LOOP_LATENCY) || R < 2048) ? 16 : 32);
    const bool periods_ok = n % 2 == 0 && n <= 16384 && (uintptr_t)U % 16 == 0 && !opt(OPT_NO_WINDOW2) && (dbg & ~(24 | 64 | 128)) == 0;

    // rows staged through LDS when they fit and 16-byte accesses line up
    const bool perm_lds = order && n % 4 == 0 && n <= PERM_MAX && ((uintptr_t)W | (uintptr_t)Q | (uintptr_t)workspace) % 16 == 0 &&
                          (idx == nullptr || (uintptr_t)idx % 4 == 0);
    if (perm_lds) {
        SLK_LDS_OPT_IN(k_permute_in_lds, PERM_MAX * 4);
        SLK_LDS_OPT_IN(k_permute_out_lds, PERM_MAX * 4);
    }
    if (perm_lds)
        SLK_RUN("permute_in", 0, 8.0 * R * n, s, k_permute_in_lds<<<R < 2048 ? R : 2048, 256, (size_t)n * 4, s>>>(W, scale, order, R, n, Qp, inv_order, rpl));
    else
        SLK_RUN("permute_in", 0, 8.0 * R * n, s, k_permute_in<<<R < 2048 ? R : 2048, 256, 0, s>>>(W, scale, order, R, n, Qp, inv_order, rpl));

    Plan p;
    plan(0, n, min_block, num_blocks, p);
    const int row_tiles = (R + RB - 1) / RB;
    for (const Plan::Step &st : p.steps) {
        if (st.kind == 0) {
            const bool in_lds = (st.b - st.a) <= WMAX;
            // a window's op list may exceed the table only when it is a single wide leaf
            for (size_t o = 0; o < st.ops.size(); o += MAX_OPS) {
                OpTable tab;
                tab.count = (int)(st.ops.size() - o < (size_t)MAX_OPS ? st.ops.size() - o : MAX_OPS);
                double fl = 0, ub = 0;  // float64 flops per row; bytes of U the window touches
                for (int i = 0; i < tab.count; ++i) {
                    const Op &q = tab.op[i] = st.ops[o + i];
                    if (q.kind == OP_LEAF) {
                        const double w = q.b - q.a;
                        fl += w * (w - 1);
                        ub += 4.0 * w * (w + 1);
                    } else {
                        fl += 2.0 * (q.b - q.a) * (q.c - q.b);
                        ub += 8.0 * (q.b - q.a) * (q.c - q.b);
                    }
                }
                // split every update at the end of the sibling sub-tree that follows it: the columns
                // beyond are not touched again before that sibling's own update (same c, a == this b)
                for (int i = 0; i < tab.count; ++i) {
                    Op &q = tab.op[i];
                    if (q.kind != OP_UPDATE) continue;
                    q.m = q.c;
                    q.nl = 1;
                    if (no_defer) continue;
                    int leaves = 0;
                    for (int j = i + 1; j < tab.count; ++j) {
                        const Op &x = tab.op[j];
                        if (x.kind == OP_LEAF) ++leaves;
                        if (x.kind == OP_UPDATE && x.c == q.c && x.a == q.b) {
                            q.m = x.b;
                            q.nl = leaves > 0 ? leaves : 1;
                            break;
                        }
                    }
                }
                const double wbytes = 12.0 * R * (st.b - st.a) + ub;  // Q in/out + E out, U once
                PeriodTable pt;
                if (in_lds && periods_ok && st.ops.size() <= (size_t)MAX_OPS && as_periods(st.ops, st.a, st.b, pt)) {
                    if (window_rows == 32)
                        SLK_RUN_W("gptq_window", fl * R, wbytes, (R + 2 * RB - 1) / (2 * RB), s,
                                  k_gptq_window2<2><<<(R + 2 * RB - 1) / (2 * RB), 512, sizeof(Window2SmemT<2>), s>>>(
                                      Qp, Eg, U, R, n, st.a, st.b, g, inv_step, fast_ok, dbg & (24 | 64 | 128), pt, rpl));
                    else
                        SLK_RUN_W("gptq_window", fl * R, wbytes, row_tiles, s,
                                  k_gptq_window2<1><<<row_tiles, 512, sizeof(Window2SmemT<1>), s>>>(Qp, Eg, U, R, n, st.a, st.b, g, inv_step,
                                                                                                 fast_ok, dbg & 24, pt, rpl));
                }
                else if (in_lds)
                    SLK_RUN("gptq_window", fl * R, wbytes, s,
                            k_gptq_window<true><<<row_tiles, 512, sizeof(WindowSmem), s>>>(Qp, Eg, U, R, n, st.a, st.b, g,
                                                                                        inv_step, fast_ok, dbg, tab, rpl));
                else
                    SLK_RUN("gptq_window_wide", fl * R, wbytes, s,
                            k_gptq_window<false><<<row_tiles, 512, 0, s>>>(Qp, Eg, U, R, n, st.a, st.b, g, inv_step, fast_ok,
                                                                           dbg, tab, rpl));
            }
        } else {
            const double K = st.b - st.a, N = st.c - st.b;
            const int vec_ok = n % 4 == 0 && st.a % 4 == 0 && st.b % 2 == 0 && (uintptr_t)U % 16 == 0;
            {
                dim3 grid((st.c - st.b + TILE - 1) / TILE, (R + TILE - 1) / TILE);
                SLK_RUN("gptq_trailing", 2.0 * R * K * N, 4.0 * R * K + 8.0 * K * N + 8.0 * R * N, s,
                        k_gptq_trailing<<<grid, 256, 0, s>>>(Qp, Eg, U, R, n, st.a, st.b, st.b, st.c, vec_ok, rpl));
            }
        }
    }
    if (perm_lds)
        SLK_RUN("permute_out", 0, (idx ? 9.0 : 8.0) * R * n, s, k_permute_out_lds<<<R < 2048 ? R : 2048, 256, (size_t)n * 4, s>>>(Qp, inv_order, R, n, g, unscale ? scale : nullptr, Q, idx, rpl));
    else
        SLK_RUN("permute_out", 0, (idx ? 9.0 : 8.0) * R * n, s, k_permute_out<<<R < 2048 ? R : 2048, 256, 0, s>>>(Qp, inv_order, R, n, g, unscale ? scale : nullptr, Q, idx, rpl));
    if (E_out) copy_async(E_out, Eg, sizeof(float) * (size_t)R * n, s);
    return SLK_OK;
}
