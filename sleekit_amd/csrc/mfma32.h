// float32 MFMA tile core for gfx950: one 256-thread workgroup (4 waves, one per SIMD)
// accumulates a 128 x 128 output tile with v_mfma_f32_32x32x2_f32 (exact float32, a
// k-ordered fma chain per element).
//
//   wave w owns the 64 x 64 sub-tile (w >> 1, w & 1): 2 x 2 MFMA blocks of 32 x 32.
//   K is consumed in steps of 16 staged through LDS as As[k][row], Bs[k][col] with a
//   pitch of 132 floats, register-prefetching the next step during the MFMAs.
//
// Operands come from FRAGMENT LOADERS: la(k0, ra) / lb(k0, rb) fill the 8 floats this
// thread stages per K-step.  Loaders are written per kernel so that interior tiles use
// unconditional 16-byte loads; a guard expressed as `cond ? load : 0` per element makes
// hipcc branch around every load and wait for each one (guide section 5, trap (c)), which is
// what held the first version of these kernels at a quarter of the MFMA rate.
//
// Thread -> staged elements (t = threadIdx.x):
//   A, K contiguous in memory : row  t >> 1,        k = k0 + (t & 1) * 8 + e
//   A, rows contiguous        : rows (t & 15) * 8 + e,  k = k0 + (t >> 4)
//   B, columns contiguous     : k = k0 + (t >> 4),  cols (t & 15) * 8 + e
//   B, K contiguous           : col  t >> 1,        k = k0 + (t & 1) * 8 + e
//
// MFMA operand maps (guide section 3):
//   A: lane l holds A[row = l & 31][k = l >> 5]      B: lane l holds B[k = l >> 5][col = l & 31]
//   D: register r of lane l is D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31]
#pragma once

#include "common.h"

namespace slk {

typedef float float16_t __attribute__((ext_vector_type(16)));
typedef float float4_t __attribute__((ext_vector_type(4)));

constexpr int T32 = 128;        // output tile edge
constexpr int K32 = 16;         // K depth per LDS round
constexpr int PITCH32 = 132;    // floats per staged k-row

struct Tile128Smem {
    float a[K32][PITCH32];
    float b[K32][PITCH32];
};

struct Acc128 {
    float16_t c[2][2];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) c[i][j][r] = 0.0f;
    }
};

// 8 consecutive floats from p (16-byte aligned when VEC), as two 16-byte loads.
template <bool VEC>
__device__ __forceinline__ void load8(const float *p, float (&v)[8]) {
    if (VEC) {
        const float4_t lo = *reinterpret_cast<const float4_t *>(p), hi = *reinterpret_cast<const float4_t *>(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = lo[e];
            v[4 + e] = hi[e];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = p[e];
    }
}

// Edge-safe version: element e is taken from p[min(e, last)] and zeroed by a MULTIPLY when
// e > last or !row_ok (a multiply cannot be turned back into a branch around the load).
__device__ __forceinline__ void load8_guarded(const float *p, int last, bool row_ok, float (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ee = min(e, max(last, 0));
        v[e] = p[ee] * ((row_ok && e <= last) ? 1.0f : 0.0f);
    }
}

// acc += A(128 x K) * B(K x 128) for k in [k_begin, k_end), a multiple of K32 deep.
template <bool A_K_FAST, bool B_C_FAST, class LA, class LB>
__device__ __forceinline__ void tile128_mac(Acc128 &acc, Tile128Smem &sm, int k_begin, int k_end, LA la, LB lb) {
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    float ra[8], rb[8];
    const int a_r = A_K_FAST ? (t >> 1) : ((t & 15) * 8);
    const int a_k = A_K_FAST ? ((t & 1) * 8) : (t >> 4);
    const int b_c = B_C_FAST ? ((t & 15) * 8) : (t >> 1);
    const int b_k = B_C_FAST ? (t >> 4) : ((t & 1) * 8);

    auto stash = [&]() {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (A_K_FAST) sm.a[a_k + e][a_r] = ra[e]; else sm.a[a_k][a_r + e] = ra[e];
            if (B_C_FAST) sm.b[b_k][b_c + e] = rb[e]; else sm.b[b_k + e][b_c] = rb[e];
        }
    };

    if (k_begin >= k_end) return;
    la(k_begin, ra);
    lb(k_begin, rb);
    for (int k0 = k_begin; k0 < k_end; k0 += K32) {
        __syncthreads();
        stash();
        __syncthreads();
        if (k0 + K32 < k_end) {
            la(k0 + K32, ra);
            lb(k0 + K32, rb);
        }
#pragma unroll
        for (int kk = 0; kk < K32; kk += 2) {
            const int kr = kk + (lane >> 5);
            const float a0 = sm.a[kr][wr * 64 + (lane & 31)];
            const float a1 = sm.a[kr][wr * 64 + 32 + (lane & 31)];
            const float b0 = sm.b[kr][wc * 64 + (lane & 31)];
            const float b1 = sm.b[kr][wc * 64 + 32 + (lane & 31)];
            acc.c[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc.c[0][0], 0, 0, 0);
            acc.c[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc.c[0][1], 0, 0, 0);
            acc.c[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc.c[1][0], 0, 0, 0);
            acc.c[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc.c[1][1], 0, 0, 0);
        }
    }
}

// f(row, col, value) for every accumulator element of this thread (tile-local coordinates).
template <class F>
__device__ __forceinline__ void tile128_foreach(const Acc128 &acc, F f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                f(wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), wc * 64 + j * 32 + (lane & 31),
                  acc.c[i][j][r]);
}

}  // namespace slk
