// a12 + a13: best-first local search (sleekit/obq.py:220-358).
//
// Rows are independent.  One 256-thread workgroup owns one row and keeps the row's state
// -- W, Q and both gain vectors -- in registers (column j = t + 256 e lives in slot e of
// thread t), so a move costs one streamed row of H (4n bytes, usually from L2 / Infinity
// Cache) plus three workgroup reductions:
//   argmax(gain_up), argmax(gain_down)   first maximum wins, like np.argmax
//   S = sum_k (Q_old[k] - W[k]) H[c][k]  the interaction term of obq.py:328
// The candidates Q_up / Q_down of the reference are pure functions of Q (obq.py:257-258,
// 273-278) and are recomputed instead of stored.  Gains are updated in the reference's
// order: diagonal term, interaction part 1 (both on the moved column), then part 2 on
// the whole row (obq.py:322-334), each step rounded to float32.
//
// Initial gains (obq.py:231) reuse G = (W - Q) @ H from the error GEMM: delta @ H = -G exactly.
#include "common.h"
#include "npsum.h"

namespace slk {

__device__ __forceinline__ float cand_up(float q, const Grid g) { return cb_up(q, g); }
__device__ __forceinline__ float cand_down(float q, const Grid g) { return cb_down(q, g); }

struct Best {
    float v;
    int j;
    float q;  // the row's current value at column j (saves a broadcast once the move is chosen)
};
__device__ __forceinline__ Best better(Best a, Best b) {
    // larger value wins; on a tie the smaller column (first occurrence).  Bit-wise on purpose: with || and && hipcc builds
    // a branch region (two exec-mask saves and a jump) around every one of the 32 comparisons of a move
    const bool take = (b.v > a.v) | ((b.v == a.v) & (b.j < a.j));
    Best r;
    r.v = take ? b.v : a.v;
    r.j = take ? b.j : a.j;
    r.q = take ? b.q : a.q;
    return r;
}
// a thread's own slots come in increasing column order: a strictly larger value is the only way to replace the first maximum
__device__ __forceinline__ void pick_first_max(Best &b, float v, int j, float q) {
    const bool take = v > b.v;
    b.v = take ? v : b.v;
    b.j = take ? j : b.j;
    b.q = take ? q : b.q;
}
// Cross-lane steps on the vector ALU (DPP) instead of the LDS crossbar (__shfl_xor compiles to ds_bpermute_b32: ~150 cycles
// each, six dependent levels per reduction, three reductions per move -- a third of a move's time).  A butterfly level only
// needs SOME lane of the partner group once the earlier levels have made the groups uniform:
//   xor 1, xor 2: quad_perm;  xor 4: row_half_mirror (lane i <-> 7 - i of each 8);  xor 8: row_mirror (i <-> 15 - i of each 16);
//   across the four 16-lane rows: v_readlane of lanes 0, 16, 32, 48.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int x) {
    return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;
__device__ __forceinline__ float lane_f(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }

// sum over the wave in the order of six xor-butterfly levels (1, 2, 4, 8, 16, 32): every lane ends with
// ((((x + x^1) + ..^2) + ..^4) + ..^8) combined as ((r0 + r1) + (r2 + r3)) over the four rows -- the operands of every addition
// are those of acc + __shfl_xor(acc, level) (float addition commutes), so the result is that loop's bit for bit
__device__ __forceinline__ float wave_sum_tree(float acc) {
    acc = acc + dpp_f<DPP_XOR1>(acc);
    acc = acc + dpp_f<DPP_XOR2>(acc);
    acc = acc + dpp_f<DPP_HALF_MIRROR>(acc);
    acc = acc + dpp_f<DPP_MIRROR>(acc);
    const float r0 = lane_f(acc, 0), r1 = lane_f(acc, 16), r2 = lane_f(acc, 32), r3 = lane_f(acc, 48);
    return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ float max_sel(float a, float b) { return b > a ? b : a; }  // (no NaN among gains; -inf pads)
// The wave's winner, the same on every lane: the largest value (a maximum is exact in any order), then -- the rule is a
// total order on (value, column) -- the smallest column among the lanes that hold it.  Almost always ONE lane holds it: its
// column and q are read off that lane; only a tie between lanes pays a second reduction.
__device__ __forceinline__ Best wave_best(Best x) {
    float m = x.v;
    m = max_sel(m, dpp_f<DPP_XOR1>(m));
    m = max_sel(m, dpp_f<DPP_XOR2>(m));
    m = max_sel(m, dpp_f<DPP_HALF_MIRROR>(m));
    m = max_sel(m, dpp_f<DPP_MIRROR>(m));
    m = max_sel(max_sel(lane_f(m, 0), lane_f(m, 16)), max_sel(lane_f(m, 32), lane_f(m, 48)));
    const bool holds = x.v == m;
    const unsigned long long who = __builtin_amdgcn_ballot_w64(holds);
    Best r;
    r.v = m;
    int lane = (int)__builtin_ctzll(who | (1ull << 63));  // (who is never empty: some lane holds the maximum)
    if (__builtin_popcountll(who) > 1) {  // wave-uniform
        int j = holds ? x.j : 0x7fffffff;
        j = min(j, dpp_i<DPP_XOR1>(j));
        j = min(j, dpp_i<DPP_XOR2>(j));
        j = min(j, dpp_i<DPP_HALF_MIRROR>(j));
        j = min(j, dpp_i<DPP_MIRROR>(j));
        j = min(min(__builtin_amdgcn_readlane(j, 0), __builtin_amdgcn_readlane(j, 16)),
                min(__builtin_amdgcn_readlane(j, 32), __builtin_amdgcn_readlane(j, 48)));
        lane = (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(holds && x.j == j) | (1ull << 63));
    }
    r.j = __builtin_amdgcn_readlane(x.j, lane);
    r.q = lane_f(x.q, lane);
    return r;
}

__global__ void k_extract_diag(PtrTable hs, int n, float *__restrict__ d) {  // blockIdx.y: the layer of a stack
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[(size_t)blockIdx.y * n + i] = hs.p[blockIdx.y][(size_t)i * n + i];
}

// The steps to a column's candidates, cand_up(q) - q and cand_down(q) - q (obq.py:257-258), are pure functions of q: they
// change with the moved column only.  Recomputed per element and move (a true division each) they were half of a move's
// vector instructions.  CAND = 1 keeps them as floats (short rows, where registers are free); CAND = 2, uniform grids of
// <= 256 levels, keeps the candidates' LEVELS, four a register -- computed once by the reference's own formula
// (codebook.py:71-86), so exact for any Q -- and forms a step as (level * step + zero) - q, the grid value as the reference
// forms it (v_cvt_f32_ubyte, multiply, add, subtract); CAND = 0 recomputes.
__device__ __forceinline__ float byte_as_float(unsigned pk, int k) { return (float)((pk >> (8 * k)) & 0xffu); }
__device__ __forceinline__ unsigned byte_set(unsigned pk, int k, unsigned v) { return (pk & ~(0xffu << (8 * k))) | (v << (8 * k)); }
template <int EPT, int CAND>
struct Steps {
    float du[CAND == 1 ? EPT : 1], dd[CAND == 1 ? EPT : 1];
    unsigned pu[CAND == 2 ? (EPT + 3) / 4 : 1], pd[CAND == 2 ? (EPT + 3) / 4 : 1];
    __device__ __forceinline__ void clear() {
        if (CAND == 2) {
#pragma unroll
            for (int k = 0; k < (EPT + 3) / 4; ++k) pu[k] = pd[k] = 0u;
        }
    }
    // slot E of a row whose value there is qv: the steps (up, down), kept as CAND says
    __device__ __forceinline__ void set(int E, float qv, const Grid g, float &su, float &sd) {  // E: constant once unrolled
        if (CAND == 2) {
            const float tu = grid_pos(qv, g, 1.0f, 1.0f, g.top), td = grid_pos(qv, g, -1.0f, 0.0f, g.top - 1.0f);
            pu[E / 4] = byte_set(pu[E / 4], E % 4, (unsigned)tu);
            pd[E / 4] = byte_set(pd[E / 4], E % 4, (unsigned)td);
            su = grid_val(tu, g) - qv;
            sd = grid_val(td, g) - qv;
        } else {
            su = cand_up(qv, g) - qv;
            sd = cand_down(qv, g) - qv;
            if (CAND == 1) du[E] = su, dd[E] = sd;
        }
    }
    // slot E takes what `one` keeps for its slot 0
    __device__ __forceinline__ void put(int E, const Steps<1, CAND> &one) {
        if (CAND == 2) {
            pu[E / 4] = byte_set(pu[E / 4], E % 4, one.pu[0] & 0xffu);
            pd[E / 4] = byte_set(pd[E / 4], E % 4, one.pd[0] & 0xffu);
        } else if (CAND == 1) {
            du[E] = one.du[0];
            dd[E] = one.dd[0];
        }
    }
    __device__ __forceinline__ void get(int E, float qv, const Grid g, float &su, float &sd) const {
        if (CAND == 2) {
            su = grid_val(byte_as_float(pu[E / 4], E % 4), g) - qv;
            sd = grid_val(byte_as_float(pd[E / 4], E % 4), g) - qv;
        } else if (CAND == 1) {
            su = du[E];
            sd = dd[E];
        } else {
            su = cand_up(qv, g) - qv;
            sd = cand_down(qv, g) - qv;
        }
    }
};

// trace (may be NULL): moves ints per row, 2 * column + (1 = up, 0 = down) of every move taken, -1 from the
// first move on at which the row had nothing left to gain (or only a "move" onto the value it already has).
// TABLE: a general codebook (binary searches); false: the uniform grid alone -- as one kernel every one of a move's candidate
// computations sat between two jumps on g.table.
template <int EPT, bool TABLE, int CAND>
__global__ __launch_bounds__(256) void k_local_search(const float *__restrict__ W, float *__restrict__ Q,
                                                      PtrTable hs, int rpl, const float *__restrict__ G,
                                                      const float *__restrict__ hdiag, int R, int n, Grid g,
                                                      int moves, uint8_t *__restrict__ idx, int *__restrict__ trace,
                                                      float *__restrict__ gains, int gains_mode, float *__restrict__ row_err) {
    if (!TABLE) g.table = nullptr;
    // (a stack of layers by rows: rows [b rpl, (b + 1) rpl) search against Hessian b)
    const float *__restrict__ H = hs.p[blockIdx.x / rpl];
    hdiag += (size_t)(blockIdx.x / rpl) * n;
    __shared__ Best red_up[4], red_dn[4];
    __shared__ HeapSum plan;
    extern __shared__ float terms[];  // heap_sum_floats(n): the products of the interaction sum, staged for NumPy's order
    // ... followed by W of the row, slot-major (a thread reads only its own: consecutive lanes, consecutive banks): read once
    // per move for the products, it need not hold EPT registers
    float *wl = terms + ((heap_sum_floats(n) + 3) & ~3) + threadIdx.x;  // slot e at wl[256 e]
    const int row = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const size_t base = (size_t)row * n;
    const float NEG = -__builtin_huge_valf();

    heap_sum_plan(plan, n);

    float q[EPT], gu[EPT], gd[EPT];
    Steps<EPT, CAND> st;
    st.clear();
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int j = t + 256 * e;
        if (j < n) {
            wl[256 * e] = W[base + j];
            q[e] = Q[base + j];
            float su, sd;
            st.set(e, q[e], g, su, sd);
            if (gains_mode == 2) {
                // carried over from an earlier call (the stateful LocalSearchQuantizer, obq.py:234-346): the gains as
                // the reference holds them between two do_move() calls, incrementally updated, not rebuilt
                gu[e] = gains[2 * base + j];
                gd[e] = gains[2 * base + n + j];
            } else {
                const float gj = G[base + j], hjj = hdiag[j];
                // -D^2 * H_jj - 2 * (delta @ H)_j * D  with (delta @ H)_j = -G_j     (obq.py:229-231)
                gu[e] = (-(su * su)) * hjj + (2.0f * gj) * su;
                gd[e] = (-(sd * sd)) * hjj + (2.0f * gj) * sd;
            }
        } else {
            wl[256 * e] = q[e] = 0.0f;
            gu[e] = gd[e] = NEG;
            float su, sd;
            st.set(e, 0.0f, g, su, sd);
        }
    }

    // the row's error, carried like the reference's own (obq.py:290, `self.err[r] -= gain` per move): in: the error of the
    // rows as they come (from the product that made G), out: the error after the moves -- no second product for it
    float e_run = row_err ? row_err[row] : 0.0f;
    int mv = 0;
    for (; mv < moves; ++mv) {
        // ---- best up / best down of the row
        // (the thread index through an opaque copy per move: everything formed from it below -- EPT load offsets, EPT
        // positions of the staged products -- is loop-invariant, and hipcc keeps each in a register of its own across the
        // moves otherwise: 2 EPT registers for values that cost one addition to form)
        int tz = t;
        asm volatile("" : "+v"(tz));
        Best bu = {NEG, 0x7fffffff, 0.0f}, bd = {NEG, 0x7fffffff, 0.0f};
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            // (a thread's columns increase with the slot; the SLOT is tracked, a constant per comparison)
            pick_first_max(bu, gu[e], e, q[e]);
            pick_first_max(bd, gd[e], e, q[e]);
        }
        bu.j = bu.j == 0x7fffffff ? bu.j : tz + 256 * bu.j;
        bd.j = bd.j == 0x7fffffff ? bd.j : tz + 256 * bd.j;
        bu = wave_best(bu);
        bd = wave_best(bd);
        if (lane == 0) {  // (the previous move's readers of these slots passed the barriers of its sum)
            red_up[wave] = bu;
            red_dn[wave] = bd;
        }
        __syncthreads();
        bu = better(better(red_up[0], red_up[1]), better(red_up[2], red_up[3]));
        bd = better(better(red_dn[0], red_dn[1]), better(red_dn[2], red_dn[3]));

        // ---- decision (obq.py:338-346)
        const bool go_up = (bu.v > bd.v) && (bu.v > 0.0f);
        const bool go_down = !go_up && (bd.v > 0.0f);
        if (!go_up && !go_down) break;  // uniform: nothing can change in later moves either
        const int c = __builtin_amdgcn_readfirstlane(go_up ? bu.j : bd.j);
        const int ce = c >> 8, ct = c & 255;  // owner slot / thread of column c
        const float q_old = go_up ? bu.q : bd.q;
        // ---- row c of H is asked for as soon as its number is known (with H[c][c], from the diagonal: after the sum it
        // would be a second trip to memory per move); what only needs the decision is computed while the loads are out.
        // All loads of (a chunk of) the row are issued before the first is waited for, and the slots then run the same
        // instructions, selects instead of jumps: with `if (j == c) ... else ...` per slot hipcc put every load in a block
        // of its own, right in front of its use -- sixteen trips to memory one after the other per move.  A slot beyond the
        // row reads the row's last element; its gains stay -inf under any finite update.
        const float *hrow = H + (size_t)c * n;
        constexpr int CHUNK = EPT < 16 ? EPT : 16;
        float h[CHUNK];
#pragma unroll
        for (int k = 0; k < CHUNK; ++k) h[k] = hrow[min(tz + 256 * k, n - 1)];
        const float hd = hdiag[c];
        asm volatile("" ::: "memory");
        const float q_new = go_up ? cand_up(q_old, g) : cand_down(q_old, g);
        // A "move" onto the value the weight already has (the up-candidate of the top level is the top level; a
        // rounding residue can leave such a candidate a positive gain): the reference carries it out, and every
        // term of its gain update (obq.py:322-334) is then a product with a zero difference -- Q and the gains stay as
        // they are, so it repeats the same non-move until the moves run out.  Same final state: stop here.  The trace
        // says -1 ("stay") from here on, the convention of the move records the parity tests hold it to
        // (oracle/obq_ref.py: move_record counts the reference's non-moves as "stay").
        if (q_new == q_old) break;
        if (trace && t == 0) trace[(size_t)row * moves + mv] = 2 * c + (go_up ? 1 : 0);
        e_run = e_run - (go_up ? bu.v : bd.v);

        // ---- stream row c of H: the products of the interaction sum with the OLD Q (obq.py:328), and part 2
        //      off the moved column
        const float two_dq = 2.0f * (q_old - q_new);
        // the moved column's candidates before and after the move (obq.py:322-334)
        const float d1u = cand_up(q_old, g) - q_old, d1d = cand_down(q_old, g) - q_old;
        Steps<1, CAND> moved;
        moved.clear();
        float d2u, d2d;
        moved.set(0, q_new, g, d2u, d2d);
        const float sq_u = d1u * d1u - d2u * d2u, sq_d = d1d * d1d - d2d * d2d;
        const float df_u = d1u - d2u, df_d = d1d - d2d;

        // ---- the products of the interaction sum with the OLD Q (obq.py:328), and part 2 off the moved column (whose own
        //      slot keeps its gains here: rebuilt below)
#pragma unroll
        for (int e0 = 0; e0 < EPT; e0 += CHUNK) {
            if (e0 > 0) {
#pragma unroll
                for (int k = 0; k < CHUNK; ++k) h[k] = hrow[min(tz + 256 * (e0 + k), n - 1)];
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int k = 0; k < CHUNK; ++k) {
                const int e = e0 + k, j = tz + 256 * e;
                if (e < EPT) {
                    if (j < n) terms[heap_sum_pos(j)] = (q[e] - wl[256 * e]) * h[k];  // Q still holds the old value at column c
                    float su, sd;
                    st.get(e, q[e], g, su, sd);
                    const float f = two_dq * h[k];
                    const float nu = gu[e] + f * su, nd = gd[e] + f * sd;
                    gu[e] = j == c ? gu[e] : nu;
                    gd[e] = j == c ? gd[e] : nd;
                }
                // (slots in turn: left alone the scheduler runs all of them side by side, at three times the registers)
                if (k % 2 == 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
        // np.sum(axis=-1) of the products, in NumPy's pairwise order: the sum feeds a gain that later moves compare
        const float s = heap_sum(plan, terms, n);

        // ---- the moved column: diagonal term, interaction part 1, then part 2 with H[c][c] (the diagonal holds the row's
        //      own element), each rounded in turn (obq.py:322-334)
        if (t == ct) {
            const float f_cc = two_dq * hd, s2x = 2.0f * s;
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                if (e == ce) {
                    float x = gu[e];
                    x = x + hd * sq_u;
                    x = x + s2x * df_u;
                    x = x + f_cc * d2u;
                    gu[e] = x;
                    float y = gd[e];
                    y = y + hd * sq_d;
                    y = y + s2x * df_d;
                    y = y + f_cc * d2d;
                    gd[e] = y;
                    q[e] = q_new;
                    st.put(e, moved);
                }
            }
        }
    }
    if (trace && t == 0)
        for (; mv < moves; ++mv) trace[(size_t)row * moves + mv] = -1;
    if (row_err && t == 0) row_err[row] = e_run;

#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int j = t + 256 * e;
        if (j < n) {
            Q[base + j] = q[e];
            if (idx) idx[base + j] = (uint8_t)cb_index(q[e], g);
            if (gains_mode) {
                gains[2 * base + j] = gu[e];
                gains[2 * base + n + j] = gd[e];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// The same search with ONE WAVE per row, for row lengths whose NumPy summation tree is regular and small:
// n = L leaves of m elements (m <= 128, a multiple of 8; L = 8 or 16), i.e. 8 L = 64 or 128 of NumPy's accumulator
// chains -- one or two per lane.  Lane l, register set s holds chain c = l + 64 s: leaf c >> 3, accumulator c & 7, the
// columns leaf * m + (c & 7) + 8 i.  A move then needs no LDS and no barrier: the chain is added up in the lane, the 8
// accumulators of a leaf are 8 neighbouring lanes (xor-shuffles 1, 2, 4 give NumPy's ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))),
// the 8 leaves of a register set the shuffles 8, 16, 32 (NumPy's halving), two sets one more add.  Four rows per
// workgroup and no barrier in the move: 135 -> 95 us for 3072 x 1024 with 10 moves (taken from 2048 rows up, see the
// dispatch below).  Same arithmetic in the same order: bit-equal to k_local_search.
// WAVES = 4: the same with ONE ROW per 256-thread workgroup, for trees of 32 or 64 leaves (n = 3072, 4096, 6144, 8192):
// thread t, register set s holds chain t + 256 s; a wave's 8 leaves are a sub-tree of NumPy's recursion, the four waves'
// sums meet in LDS as (T0 + T1) + (T2 + T3).  Two barriers per move (best move, sum) against the general kernel's five,
// and no staging of the products: 684 -> 532 us for 4096 x 4096 with 10 moves, 242 -> 184 for 1024 x 4096.
// CAND (see Steps): floats for the one-wave-per-row form, which is bound by the issue of its instructions and has registers to
// spare (4096 x 1024, 10 moves: 150 -> 113 us); the workgroup-per-row form is bound by the LATENCY of a move (one trip to
// memory for the row of H, two barriers, its own instructions one after the other), which resident rows hide and registers
// cost (1024 x 4096 with floats: 121 -> 132 us, three rows a CU instead of four): packed levels there.
#ifndef SLK_LS_WAVES_PER_SIMD
#define SLK_LS_WAVES_PER_SIMD 4  // rows of 3072 / 4096 columns a CU keeps in flight (register cap 128)
#endif
template <int M8, int S, int WAVES, bool TABLE, int CAND>
__global__ __launch_bounds__(256, (WAVES == 4 && S == 1 && CAND == 2 ? SLK_LS_WAVES_PER_SIMD : 1)) void k_local_search_wave(const float *__restrict__ W, float *__restrict__ Q,
                                                           PtrTable hs, int rpl, const float *__restrict__ G,
                                                           const float *__restrict__ hdiag, int R, int n, Grid g, int moves,
                                                           uint8_t *__restrict__ idx, int *__restrict__ trace,
                                                           float *__restrict__ gains, int gains_mode, float *__restrict__ row_err) {
    if (!TABLE) g.table = nullptr;
    constexpr int EPT = M8 * S;
    __shared__ Best red_up[4], red_dn[4];
    __shared__ float red_s[2][4];
    // W of the row(s), slot-major (a thread reads only its own: consecutive lanes, consecutive banks): read once per move
    // for the products, it need not hold EPT registers -- they decide how many rows a CU keeps in flight
    __shared__ float wl[EPT][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int row = WAVES == 1 ? blockIdx.x * 4 + wv : blockIdx.x;
    if (row >= R) return;  // (WAVES == 1: whole waves, and no barrier anywhere below; WAVES == 4: the whole workgroup)
    const float *__restrict__ H = hs.p[row / rpl];  // a stack of layers by rows (a wave's row belongs to one layer)
    hdiag += (size_t)(row / rpl) * n;
    const int m = 8 * M8;
    const int tr = WAVES == 1 ? lane : threadIdx.x;  // thread within the row
    const size_t base = (size_t)row * n;
    const float NEG = -__builtin_huge_valf();
    // slot e = s2 * M8 + i of this thread: column col0 + 8 WAVES m s2 + 8 i, increasing with e (one register and constant
    // offsets: as an array of EPT columns they cost EPT registers, and the row's loads a register each for their address)
    const int col0 = (tr >> 3) * m + (tr & 7);
    auto col = [&](int e) { return col0 + (e / M8) * (8 * WAVES * m) + 8 * (e % M8); };
    float q[EPT], gu[EPT], gd[EPT];
    Steps<EPT, CAND> st;
    st.clear();
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int j = col(e);
        wl[e][threadIdx.x] = W[base + j];
        q[e] = Q[base + j];
        float su, sd;
        st.set(e, q[e], g, su, sd);
        if (gains_mode == 2) {
            gu[e] = gains[2 * base + j];
            gd[e] = gains[2 * base + n + j];
        } else {
            const float gj = G[base + j], hjj = hdiag[j];
            gu[e] = (-(su * su)) * hjj + (2.0f * gj) * su;
            gd[e] = (-(sd * sd)) * hjj + (2.0f * gj) * sd;
        }
    }

    // the row's error, carried like the reference's own (obq.py:290, `self.err[r] -= gain` per move): in: the error of the
    // rows as they come (from the product that made G), out: the error after the moves -- no second product for it
    float e_run = row_err ? row_err[row] : 0.0f;
    int mv = 0;
    for (; mv < moves; ++mv) {
        Best bu = {NEG, 0x7fffffff, 0.0f}, bd = {NEG, 0x7fffffff, 0.0f};
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            // (a thread's columns increase with e: within a leaf by 8, and the second register set's leaves lie 8 WAVES
            // further.  The SLOT is what is tracked -- a constant per comparison; the columns would be EPT registers)
            pick_first_max(bu, gu[e], e, q[e]);
            pick_first_max(bd, gd[e], e, q[e]);
        }
        bu.j = bu.j == 0x7fffffff ? bu.j : col0 + (S > 1 && bu.j >= M8 ? 8 * WAVES * m - 8 * M8 : 0) + 8 * bu.j;
        bd.j = bd.j == 0x7fffffff ? bd.j : col0 + (S > 1 && bd.j >= M8 ? 8 * WAVES * m - 8 * M8 : 0) + 8 * bd.j;
        bu = wave_best(bu);
        bd = wave_best(bd);
        if (WAVES > 1) {
            if (lane == 0) {  // (the previous move's readers of these slots have passed the barrier of its sum)
                red_up[wv] = bu;
                red_dn[wv] = bd;
            }
            __syncthreads();
            bu = better(better(red_up[0], red_up[1]), better(red_up[2], red_up[3]));
            bd = better(better(red_dn[0], red_dn[1]), better(red_dn[2], red_dn[3]));
        }
        const bool go_up = (bu.v > bd.v) && (bu.v > 0.0f);
        const bool go_down = !go_up && (bd.v > 0.0f);
        if (!go_up && !go_down) break;
        const int c = go_up ? bu.j : bd.j;
        const float q_old = go_up ? bu.q : bd.q;
        // the row of H is asked for as soon as its number is known (with H[c][c], from the diagonal: after the sum it would
        // be a second trip to memory per move); what only needs the decision is computed while the loads are out
        const float *hrow = H + (size_t)c * n;
        float h[M8];
#pragma unroll
        for (int i = 0; i < M8; ++i) h[i] = hrow[col(i)];
        const float hd = hdiag[c];
        asm volatile("" ::: "memory");
        const float q_new = go_up ? cand_up(q_old, g) : cand_down(q_old, g);
        if (q_new == q_old) break;  // a "move" onto the same value: see k_local_search
        if (trace && tr == 0) trace[(size_t)row * moves + mv] = 2 * c + (go_up ? 1 : 0);
        e_run = e_run - (go_up ? bu.v : bd.v);
        const float two_dq = 2.0f * (q_old - q_new);
        // the moved column's candidates before and after the move (obq.py:322-334)
        const float d1u = cand_up(q_old, g) - q_old, d1d = cand_down(q_old, g) - q_old;
        Steps<1, CAND> moved;
        moved.clear();
        float d2u, d2d;
        moved.set(0, q_new, g, d2u, d2d);
        const float sq_u = d1u * d1u - d2u * d2u, sq_d = d1d * d1d - d2d * d2d;
        const float df_u = d1u - d2u, df_d = d1d - d2d;

        float total = 0.0f;
#pragma unroll
        for (int s2 = 0; s2 < S; ++s2) {
            // (every load of a set in flight before the first is waited for, then selects, not jumps: see k_local_search)
            if (s2 > 0) {
#pragma unroll
                for (int i = 0; i < M8; ++i) h[i] = hrow[col(s2 * M8 + i)];
                asm volatile("" ::: "memory");
            }
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < M8; ++i) {
                const int e = s2 * M8 + i;
                const float p = (q[e] - wl[e][threadIdx.x]) * h[i];  // Q still holds the old value at column c
                acc = i == 0 ? p : acc + p;
                float su, sd;
                st.get(e, q[e], g, su, sd);
                const float f = two_dq * h[i];
                const float nu = gu[e] + f * su, nd = gd[e] + f * sd;
                const bool own = col(e) == c;  // the moved column keeps its gains here: rebuilt below
                gu[e] = own ? gu[e] : nu;
                gd[e] = own ? gd[e] : nd;
                // (slots in turn: left alone the scheduler runs all of them side by side, at three times the registers)
                if (i % 2 == 1) __builtin_amdgcn_sched_barrier(0);
            }
            // the leaf (8 accumulators = 8 lanes), then the 8 leaves of this register set
            acc = wave_sum_tree(acc);
            if (WAVES > 1 && lane == 0) red_s[s2][wv] = acc;  // this wave's 8 leaves of the set
            total = s2 == 0 ? acc : total + acc;
        }
        if (WAVES > 1) {
            __syncthreads();
            total = (red_s[0][0] + red_s[0][1]) + (red_s[0][2] + red_s[0][3]);
            if (S > 1) total = total + ((red_s[1][0] + red_s[1][1]) + (red_s[1][2] + red_s[1][3]));
        }
        const float ssum = 0.0f + total;  // (NumPy's reduction starts from 0 and adds the chunk's pairwise sum)
        // the moved column: diagonal term, interaction part 1, then part 2 with H[c][c] (the diagonal holds the row's own
        // element), each rounded in turn (obq.py:322-334)
        const float f_cc = two_dq * hd, s2x = 2.0f * ssum;
        // its owner: leaf c / m, accumulator chain (c % m) & 7 -> thread; slot (leaf / (8 WAVES)) M8 + (c % m) / 8.  c is
        // wave-uniform: one lane goes in, and the slot is picked by scalar compares
        const int leaf = c / m, within = c - leaf * m;
        const int ce = (leaf / (8 * WAVES)) * M8 + (within >> 3);
        const int ct = ((leaf % (8 * WAVES)) << 3) | (within & 7);
        if (tr == ct) {
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                if (e == ce) {
                    float x = gu[e];
                    x = x + hd * sq_u;
                    x = x + s2x * df_u;
                    x = x + f_cc * d2u;
                    gu[e] = x;
                    float y = gd[e];
                    y = y + hd * sq_d;
                    y = y + s2x * df_d;
                    y = y + f_cc * d2d;
                    gd[e] = y;
                    q[e] = q_new;
                    st.put(e, moved);
                }
            }
        }
    }
    if (trace && tr == 0)
        for (; mv < moves; ++mv) trace[(size_t)row * moves + mv] = -1;
    if (row_err && tr == 0) row_err[row] = e_run;

#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int j = col(e);
        Q[base + j] = q[e];
        if (idx) idx[base + j] = (uint8_t)cb_index(q[e], g);
        if (gains_mode) {
            gains[2 * base + j] = gu[e];
            gains[2 * base + n + j] = gd[e];
        }
    }
}

}  // namespace slk

using namespace slk;

// One search over `batch` layers stacked by rows (rows [b rpl, (b + 1) rpl) against H[b]): the initial interaction product
// is ONE batched product (row_errors_impl), the moves ONE launch -- the row shards of a round on several ranks are a few
// hundred rows per layer, and a search per layer is ten launches of a few microseconds each (the host's launch rate
// then bounds the rank: BLOOM-560M on 8 ranks spent 18 of its 34 ms per step enqueueing).
static int local_search_impl(const float *W, float *Q, const float *const *Hs, int batch, int rpl, int n, int levels, double lo,
                             double hi, const float *table, int moves, uint8_t *idx, int *trace, float *gains,
                             int gains_mode, const int *sym_known, float *row_err_out, void *workspace, size_t ws_bytes,
                             slk_stream_t stream) {
    const int R = batch * rpl;
    SLK_REQUIRE(row_err_out == nullptr || gains_mode != 2, "the carried error needs the initial product (gains_mode 0 or 1)");
    Arena ws(workspace, ws_bytes);
    float *G = ws.take<float>((size_t)R * n);
    float *row_err = row_err_out ? row_err_out : ws.take<float>((size_t)R);
    float *hdiag = ws.take<float>((size_t)n * batch);
    if (!G || !row_err || !hdiag) {
        set_error("workspace too small");
        return SLK_E_WS;
    }
    const size_t used = align_up(ws.used, 256);
    if (gains_mode != 2) {  // (carried gains: no initial product)
        int rc = row_errors_products(W, Q, Hs, batch, rpl, n, sym_known, row_err, G, static_cast<char *>(workspace) + used,
                                              ws_bytes - used, stream);
        if (rc != SLK_OK) return rc;
    }
    PtrTable hs;
    for (int b = 0; b < 64; ++b) hs.p[b] = b < batch ? Hs[b] : nullptr;
    hipStream_t s = as_stream(stream);
    SLK_RUN("extract_diag", 0, 8.0 * n * batch, s, k_extract_diag<<<dim3((n + 255) / 256, batch), 256, 0, s>>>(hs, n, hdiag));
    const Grid g = make_grid(levels, lo, hi, table);
    const int ept = (n + 255) / 256;
    // heap_sum_floats(n) for the staged products, then W of the row (256 floats per slot)
    const size_t lds = (size_t)(((n + 8 * (n / 128) + 8 + 3) & ~3) + 256 * std::max(4, (ept <= 8 ? (ept <= 4 ? 4 : 8) : ept <= 16 ? 16 : ept <= 32 ? 32 : ept <= 48 ? 48 : 64))) * sizeof(float);
    // how the candidates are kept (Steps): floats on short rows, packed levels for a uniform grid of <= 256 levels
    const bool small_grid = !g.table && levels <= 256;
#define SLK_LS_T(E, T, C)                                                                                           \
    do {                                                                                                            \
        SLK_LDS_OPT_IN((k_local_search<E, T, C>), lds);                                                             \
        SLK_RUN("local_search", 10.0 * n * R * moves, 4.0 * n * R * moves + 13.0 * R * n, s,                       \
                (k_local_search<E, T, C><<<R, 256, lds, s>>>(W, Q, hs, rpl, G, hdiag, R, n, g, moves, idx, trace, gains, gains_mode, row_err_out))); \
    } while (0)
#define SLK_LS(E)                                                                                                   \
    do {                                                                                                            \
        if constexpr (E <= 8) { /* short rows: the steps as floats */                                               \
            if (g.table) SLK_LS_T(E, true, 1);                                                                      \
            else SLK_LS_T(E, false, 1);                                                                             \
        } else {                                                                                                    \
            if (g.table) SLK_LS_T(E, true, 0);                                                                      \
            else if (small_grid) SLK_LS_T(E, false, 2);                                                             \
            else SLK_LS_T(E, false, 0);                                                                             \
        }                                                                                                           \
    } while (0)
    // regular row lengths (8 or 16 leaves of m <= 128 elements, m % 8 == 0): one wave per row, no LDS, no barrier
    {
        int m = n, leaves = 1;
        bool regular = true;
        while (m > 128 && regular) {
            regular = m % 2 == 0 && (m / 2) % 8 == 0;  // NumPy splits at (m / 2) rounded down to 8: only even halves keep the tree regular
            m /= 2;
            leaves *= 2;
        }
        // (round 3 kept few rows on the general kernel -- 1024 x 1024: 68 against 80 us, the wave kernel's shuffles were LDS
        // permutes then; since round 4 the wave kernel wins at every height: 1024 x 1024 x 10 moves 48 -> 29 us, 128 x 1024
        // 39 -> 27)
        const int wave_opt = opt(OPT_NO_WAVE_SEARCH);  // 1: never (tests: the general kernel on regular rows)
        regular = regular && m % 8 == 0 && m >= 8 && wave_opt <= 0;
        const int m8 = m / 8;
#define SLK_LSW_T(M8, S, WAVES, T, C)                                                                                   \
    SLK_RUN("local_search", 10.0 * n * R * moves, 4.0 * n * R * moves + 13.0 * R * n, s,                               \
            (k_local_search_wave<M8, S, WAVES, T, C><<<WAVES == 1 ? (R + 3) / 4 : R, 256, 0, s>>>(                      \
                W, Q, hs, rpl, G, hdiag, R, n, g, moves, idx, trace, gains, gains_mode, row_err_out)))
#define SLK_LSW(M8, S, WAVES, LEAVES)                                                                                   \
    if (regular && m8 == M8 && leaves == LEAVES) {                                                                      \
        constexpr bool keep = WAVES == 1 && S == 1;                                                                     \
        if (g.table) SLK_LSW_T(M8, S, WAVES, true, (keep ? 1 : 0));                                                     \
        else SLK_LSW_T(M8, S, WAVES, false, (keep ? 1 : 2));                                                            \
        return SLK_OK;                                                                                                  \
    }
        regular = regular && (g.table || small_grid);  // (a uniform grid of more than 256 levels: the general kernel)
        // a wave per row: 8 or 16 leaves
        SLK_LSW(16, 1, 1, 8)   // n = 1024
        SLK_LSW(12, 1, 1, 8)   // n = 768
        SLK_LSW(16, 2, 1, 16)  // n = 2048
        SLK_LSW(12, 2, 1, 16)  // n = 1536
        // a workgroup per row, a chain per thread: 32 or 64 leaves
        SLK_LSW(16, 1, 4, 32)  // n = 4096
        SLK_LSW(12, 1, 4, 32)  // n = 3072
        SLK_LSW(16, 2, 4, 64)  // n = 8192
        SLK_LSW(12, 2, 4, 64)  // n = 6144
#undef SLK_LSW
#undef SLK_LSW_T
    }
    if (ept <= 4) SLK_LS(4);
    else if (ept <= 8) SLK_LS(8);
    else if (ept <= 16) SLK_LS(16);
    else if (ept <= 32) SLK_LS(32);
    else if (ept <= 48) SLK_LS(48);
    else SLK_LS(64);
#undef SLK_LS
#undef SLK_LS_T
    return SLK_OK;
}

extern "C" int slk_local_search(const float *W, float *Q, const float *H, int R, int n, int levels, double lo,
                                double hi, const float *table, int moves, uint8_t *idx, int *trace, float *gains,
                                int gains_mode, float *row_err, void *workspace, size_t ws_bytes, slk_stream_t stream) {
    SLK_REQUIRE(W && Q && H && R > 0 && n > 0 && moves >= 0, "bad arguments");
    SLK_REQUIRE(levels >= 2 && (table || lo < hi), "codebook needs levels >= 2 and lo < hi");
    SLK_REQUIRE(idx == nullptr || levels <= 256, "uint8 indices need levels <= 256");
    SLK_REQUIRE(n <= 256 * 64, "local search supports n <= 16384");
    SLK_REQUIRE(gains_mode >= 0 && gains_mode <= 2 && (gains_mode == 0 || gains), "gains_mode 1 / 2 needs the gains buffer");
    return local_search_impl(W, Q, &H, 1, R, n, levels, lo, hi, table, moves, idx, trace, gains, gains_mode, nullptr, row_err, workspace,
                             ws_bytes, stream);
}

extern "C" int slk_local_search_batch(const float *W, float *Q, const float *const *H, int batch, int rows_per_layer, int n,
                                      int levels, double lo, double hi, const float *table, int moves, uint8_t *idx,
                                      const int *symmetric, float *row_err, void *workspace, size_t ws_bytes, slk_stream_t stream) {
    SLK_REQUIRE(W && Q && H && rows_per_layer > 0 && n > 0 && moves >= 0, "bad arguments");
    SLK_REQUIRE(batch >= 1 && batch <= 64, "batch must be 1..64");
    SLK_REQUIRE(batch == 1 || rows_per_layer % 128 == 0, "a batch needs rows_per_layer to be a multiple of 128");
    SLK_REQUIRE(levels >= 2 && (table || lo < hi), "codebook needs levels >= 2 and lo < hi");
    SLK_REQUIRE(idx == nullptr || levels <= 256, "uint8 indices need levels <= 256");
    SLK_REQUIRE(n <= 256 * 64, "local search supports n <= 16384");
    SLK_REQUIRE((long long)batch * rows_per_layer <= 0x7fffffffLL, "too many rows");
    for (int b = 0; b < batch; ++b) SLK_REQUIRE(H[b], "null Hessian in the batch");
    return local_search_impl(W, Q, H, batch, rows_per_layer, n, levels, lo, hi, table, moves, idx, nullptr, nullptr, 0, symmetric, row_err,
                             workspace, ws_bytes, stream);
}

