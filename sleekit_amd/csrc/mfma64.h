// float64 MFMA tile core for gfx950: one 256-thread workgroup (4 waves, one per
// SIMD) accumulates a 64 x 64 output tile with v_mfma_f64_16x16x4_f64.
//
//   wave w owns the 32 x 32 sub-tile (w >> 1, w & 1): 2 x 2 MFMA blocks of 16 x 16.
//   K is consumed in steps of 16 staged through LDS: As[k][row], Bs[k][col]
//   with a row pitch of 80 doubles (== 16 mod 32): the two 16-lane groups that a
//   32-lane half of ds_read_b64 serves then fall on disjoint bank ranges.
//   Global loads of step s+1 are issued into registers before the MFMAs of
//   step s and written to LDS after them (register double buffering).
//
// Operand maps of v_mfma_f64_16x16x4_f64 (guide section 3, checked by
// tests/test_gpu_kernels.py::test_dgemm_tile_asymmetric):
//   A: lane l holds A[row = l & 15][k = l >> 4]
//   B: lane l holds B[k = l >> 4][col = l & 15]
//   D: register r of lane l is D[row = (l >> 4) + 4 r][col = l & 15]
#pragma once

#include "common.h"

namespace slk {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int TILE = 64;       // output tile edge
constexpr int KSTEP = 16;      // K depth staged per LDS round
constexpr int LDS_PITCH = 80;  // doubles per staged k-row

struct Tile64Smem {
    double a[KSTEP][LDS_PITCH];
    double b[KSTEP][LDS_PITCH];
};

struct Acc64 {
    double4_t c[2][2];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) c[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
    }
};

// Accumulate acc += A(64 x K) * B(K x 64) for k in [k_begin, k_end), k_end - k_begin a
// multiple of KSTEP.  `fa(r, k)` / `fb(k, c)` return operand elements as double for
// tile-local r, c in [0, 64) and absolute k.  A_K_FAST / B_C_FAST say which index is
// contiguous in the operand's memory so the staging loads coalesce.
template <bool A_K_FAST, bool B_C_FAST, class FA, class FB>
__device__ __forceinline__ void tile64_mac(Acc64 &acc, Tile64Smem &sm, int k_begin, int k_end, FA fa, FB fb) {
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    double ra[4], rb[4];

    // staging coordinates of this thread's 4 elements
    const int a_r = A_K_FAST ? (t >> 2) : ((t & 15) * 4);
    const int a_k = A_K_FAST ? ((t & 3) * 4) : (t >> 4);
    const int b_c = B_C_FAST ? ((t & 15) * 4) : (t >> 2);
    const int b_k = B_C_FAST ? (t >> 4) : ((t & 3) * 4);

    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ra[e] = A_K_FAST ? fa(a_r, k0 + a_k + e) : fa(a_r + e, k0 + a_k);
            rb[e] = B_C_FAST ? fb(k0 + b_k, b_c + e) : fb(k0 + b_k + e, b_c);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (A_K_FAST) sm.a[a_k + e][a_r] = ra[e]; else sm.a[a_k][a_r + e] = ra[e];
            if (B_C_FAST) sm.b[b_k][b_c + e] = rb[e]; else sm.b[b_k + e][b_c] = rb[e];
        }
    };

    if (k_begin >= k_end) return;
    fetch(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += KSTEP) {
        __syncthreads();  // previous step's LDS reads are done
        stash();
        __syncthreads();
        if (k0 + KSTEP < k_end) fetch(k0 + KSTEP);
#pragma unroll
        for (int kk = 0; kk < KSTEP; kk += 4) {
            const int kr = kk + (lane >> 4);
            const double a0 = sm.a[kr][wr * 32 + (lane & 15)];
            const double a1 = sm.a[kr][wr * 32 + 16 + (lane & 15)];
            const double b0 = sm.b[kr][wc * 32 + (lane & 15)];
            const double b1 = sm.b[kr][wc * 32 + 16 + (lane & 15)];
            acc.c[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc.c[0][0], 0, 0, 0);
            acc.c[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc.c[0][1], 0, 0, 0);
            acc.c[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc.c[1][0], 0, 0, 0);
            acc.c[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc.c[1][1], 0, 0, 0);
        }
    }
}

// Visit every accumulator element of this thread: f(row, col, value), tile-local coordinates.
template <class F>
__device__ __forceinline__ void tile64_foreach(const Acc64 &acc, F f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                f(wr * 32 + i * 16 + (lane >> 4) + 4 * r, wc * 32 + j * 16 + (lane & 15), acc.c[i][j][r]);
}

}  // namespace slk
