// float32 MFMA tile core for gfx950: one 256-thread workgroup (4 waves, one per SIMD)
// accumulates a 128 x 128 output tile with v_mfma_f32_32x32x2_f32 (exact float32, a
// k-ordered fma chain per element).
//
//   wave w owns the 64 x 64 sub-tile (w >> 1, w & 1): 2 x 2 MFMA blocks of 32 x 32.
//   K is consumed in steps of 16 staged through LDS as As[k][row], Bs[k][col] with a
//   pitch of 132 floats, register-prefetching the next step during the MFMAs.
//
// Operand maps (guide section 3):
//   A: lane l holds A[row = l & 31][k = l >> 5]      B: lane l holds B[k = l >> 5][col = l & 31]
//   D: register r of lane l is D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31]
#pragma once

#include "common.h"

namespace slk {

typedef float float16_t __attribute__((ext_vector_type(16)));

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

// acc += A(128 x K) * B(K x 128) for k in [k_begin, k_end), a multiple of K32 deep.
// fa(r, k), fb(k, c): operand elements for tile-local r, c and absolute k.
// Each thread stages 8 elements per operand per step:
//   A_K_FAST : rows t >> 1, k-octet (t & 1) * 8          else: k = t >> 4, rows (t & 15) * 8 ..+7
//   B_C_FAST : k = t >> 4, cols (t & 15) * 8 ..+7         else: cols t >> 1, k-octet (t & 1) * 8
template <bool A_K_FAST, bool B_C_FAST, class FA, class FB>
__device__ __forceinline__ void tile128_mac(Acc128 &acc, Tile128Smem &sm, int k_begin, int k_end, FA fa, FB fb) {
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    float ra[8], rb[8];
    const int a_r = A_K_FAST ? (t >> 1) : ((t & 15) * 8);
    const int a_k = A_K_FAST ? ((t & 1) * 8) : (t >> 4);
    const int b_c = B_C_FAST ? ((t & 15) * 8) : (t >> 1);
    const int b_k = B_C_FAST ? (t >> 4) : ((t & 1) * 8);

    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ra[e] = A_K_FAST ? fa(a_r, k0 + a_k + e) : fa(a_r + e, k0 + a_k);
            rb[e] = B_C_FAST ? fb(k0 + b_k, b_c + e) : fb(k0 + b_k + e, b_c);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (A_K_FAST) sm.a[a_k + e][a_r] = ra[e]; else sm.a[a_k][a_r + e] = ra[e];
            if (B_C_FAST) sm.b[b_k][b_c + e] = rb[e]; else sm.b[b_k + e][b_c] = rb[e];
        }
    };

    if (k_begin >= k_end) return;
    fetch(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += K32) {
        __syncthreads();
        stash();
        __syncthreads();
        if (k0 + K32 < k_end) fetch(k0 + K32);
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
