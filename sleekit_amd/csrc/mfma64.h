// float64 MFMA tile core for gfx950: one 256-thread workgroup (4 waves, one per SIMD)
// accumulates a 64 x 64 output tile with v_mfma_f64_16x16x4_f64.
//
//   wave w owns the 32 x 32 sub-tile (w >> 1, w & 1): 2 x 2 MFMA blocks of 16 x 16.
//   K is consumed in steps of 32 staged through LDS.  A is staged row-major
//   As[row][k] with a pitch of 34 doubles: a thread's 8 consecutive k go down as four
//   ds_write_b128, and the MFMA operand read (lane l: row l & 15, k l >> 4) walks banks
//   4 row + 2 k, conflict-free for a 32-lane half.  B is staged either the same way
//   (operand given as [col][k]) or as Bs[k][col] with a pitch of 80 (operand given as [k][col]).
//   Global loads of step s+1 are issued before the MFMAs of step s (register double buffer).
//
// Operands come from FRAGMENT LOADERS la(k0, v) / lb(k0, v) that fill this thread's 8
// doubles per K-step; kernels write them so interior tiles use unconditional 16-byte loads
// (see mfma32.h for why per-element guards are poison).
//
// Thread -> staged elements (t = threadIdx.x):
//   A and K-contiguous B : row/col t >> 2,     k = k0 + (t & 3) * 8 + e
//   column-contiguous B  : k = k0 + (t >> 3),  four PIECES of two columns: 16 h + 2 (t & 7) + {0, 1}, h = 0..3
//     (load8d_cols).  One ds_write_b128 per piece: the 16 lanes of two k rows then cover all 64 banks
//     (row pitch 80 doubles = 32 banks mod 64).  With 8 consecutive columns per thread the same 16 lanes hit
//     16 banks four deep -- half of the kernel's LDS cycles were bank conflicts (SQ_LDS_BANK_CONFLICT) --
//     and each global load instruction touched 64-byte-strided 16-byte bits instead of 128-byte runs.
//
// Operand maps of v_mfma_f64_16x16x4_f64 (guide section 3; exercised with asymmetric data by
// tests/test_gpu_parity.py::test_factor_of_plain_matrix_and_not_pd):
//   A: lane l holds A[row = l & 15][k = l >> 4]     B: lane l holds B[k = l >> 4][col = l & 15]
//   D: register r of lane l is D[row = (l >> 4) + 4 r][col = l & 15]
#pragma once

#include <type_traits>

#include "common.h"

namespace slk {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));
typedef float float4v_t __attribute__((ext_vector_type(4)));

constexpr int TILE = 64;       // output tile edge
constexpr int KSTEP = 32;      // K depth staged per LDS round
constexpr int PITCH_RK = 34;   // [row][k] images
constexpr int PITCH_KC = 80;   // [k][col] image

struct Tile64Smem {
    double a[TILE * PITCH_RK];
    double b[KSTEP * PITCH_KC];  // >= TILE * PITCH_RK
};

// The same with a float32 A operand kept as float32 in LDS (k_gptq_trailing: A = E): half the A image (9 KB instead of 17: five
// workgroups to a CU instead of four, a quarter less LDS write traffic per K-step), widened AFTER the LDS read.  Row pitch 36
// floats: a thread's 8 consecutive k go down as two ds_write_b128, and the operand read (lane l: row l & 15, k l >> 4) walks
// banks 36 row + k -- 36 r mod 64 are the sixteen multiples of 4, so the 64 lanes cover the 64 banks.
constexpr int PITCH_AF = 36;
struct Tile64SmemAf {
    float a[TILE * PITCH_AF];
    double b[KSTEP * PITCH_KC];
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

// 8 consecutive doubles (16-byte aligned when VEC) as four 16-byte loads.
template <bool VEC>
__device__ __forceinline__ void load8d(const double *p, double (&v)[8]) {
    if (VEC) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const double2_t x = *reinterpret_cast<const double2_t *>(p + 2 * h);
            v[2 * h] = x[0];
            v[2 * h + 1] = x[1];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = p[e];
    }
}

// The column-contiguous B fragment: pieces h = 0..3 at p[16 h], p[16 h + 1] (p already offset by 2 (t & 7)).
__device__ __forceinline__ void load8d_cols(const double *p, double (&v)[8]) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const double2_t x = *reinterpret_cast<const double2_t *>(p + 16 * h);
        v[2 * h] = x[0];
        v[2 * h + 1] = x[1];
    }
}
// ... edge-safe: `row` points at column col0 of the tile, columns beyond `last` (relative) or !ok give zero.
__device__ __forceinline__ void load8d_cols_guarded(const double *row, int c2, int last, bool ok, double (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int col = 16 * (e >> 1) + c2 + (e & 1);
        v[e] = row[min(col, max(last, 0))] * ((ok && col <= last) ? 1.0 : 0.0);
    }
}

// Edge-safe: element e from p[min(e, last)], zeroed by a multiply when e > last or !ok.
__device__ __forceinline__ void load8d_guarded(const double *p, int last, bool ok, double (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = p[min(e, max(last, 0))] * ((ok && e <= last) ? 1.0 : 0.0);
}

// 8 consecutive floats widened to double.
template <bool VEC>
__device__ __forceinline__ void load8f_as_d(const float *p, double (&v)[8]) {
    if (VEC) {
        const float4v_t lo = *reinterpret_cast<const float4v_t *>(p), hi = *reinterpret_cast<const float4v_t *>(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = (double)lo[e];
            v[4 + e] = (double)hi[e];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (double)p[e];
    }
}
// 8 consecutive floats, kept as floats.
template <bool VEC>
__device__ __forceinline__ void load8f(const float *p, float (&v)[8]) {
    if (VEC) {
        const float4v_t lo = *reinterpret_cast<const float4v_t *>(p), hi = *reinterpret_cast<const float4v_t *>(p + 4);
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
__device__ __forceinline__ void load8f_guarded(const float *p, int last, bool ok, float (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = p[min(e, max(last, 0))] * ((ok && e <= last) ? 1.0f : 0.0f);
}
__device__ __forceinline__ void load8f_as_d_guarded(const float *p, int last, bool ok, double (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (double)p[min(e, max(last, 0))] * ((ok && e <= last) ? 1.0 : 0.0);
}

// acc += A(64 x K) * B(K x 64) for k in [k_begin, k_end), k_end - k_begin a multiple of KSTEP.
// TA: type of the A fragment held in registers between its global load and the LDS write (double, or
// float for a float32 operand: the widening then happens at the LDS write, a whole MFMA phase after the
// load was issued -- converted at once it would make the wave wait for the load before its MFMAs).
template <bool B_C_FAST, class TA = double, class SM, class LA, class LB>
__device__ __forceinline__ void tile64_mac(Acc64 &acc, SM &sm, int k_begin, int k_end, LA la, LB lb) {
    constexpr bool AF = std::is_same<SM, Tile64SmemAf>::value;  // A stays float32 in LDS
    static_assert(!AF || (std::is_same<TA, float>::value && B_C_FAST), "the float32 A image is for float32 A and column-contiguous B");
    using TL = typename std::conditional<AF, float, double>::type;  // element type of the A image
    constexpr int PITCH_A = AF ? PITCH_AF : PITCH_RK;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    TA ra[8];
    double rb[8];
    TL *a_dst = sm.a + (t >> 2) * PITCH_A + (t & 3) * 8;
    double *b_dst = B_C_FAST ? sm.b + (t >> 3) * PITCH_KC + (t & 7) * 2 : sm.b + (t >> 2) * PITCH_RK + (t & 3) * 8;
    constexpr int B_PIECE = B_C_FAST ? 16 : 2;  // distance between a thread's pieces in the image

    auto stash = [&]() {
        if constexpr (AF) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
                *reinterpret_cast<float4v_t *>(a_dst + 4 * h) = (float4v_t){(float)ra[4 * h], (float)ra[4 * h + 1], (float)ra[4 * h + 2], (float)ra[4 * h + 3]};
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            if constexpr (!AF) *reinterpret_cast<double2_t *>(a_dst + 2 * h) = (double2_t){(double)ra[2 * h], (double)ra[2 * h + 1]};
            *reinterpret_cast<double2_t *>(b_dst + B_PIECE * h) = (double2_t){rb[2 * h], rb[2 * h + 1]};
        }
    };

    if (k_begin >= k_end) return;
    la(k_begin, ra);
    lb(k_begin, rb);
    const TL *a_src = sm.a + (wr * 32 + (lane & 15)) * PITCH_A + (lane >> 4);
    const double *b_src = B_C_FAST ? sm.b + (lane >> 4) * PITCH_KC + wc * 32 + (lane & 15)
                                   : sm.b + (wc * 32 + (lane & 15)) * PITCH_RK + (lane >> 4);
    // The second operand of each pair gets a base of its own that the compiler cannot relate to the
    // first: left to itself it fuses the pair into ds_read2_b64, which runs at HALF the rate of two
    // ds_read_b64 and banks modulo 32 over 16-lane groups -- 2-way conflicts on these images, laid
    // out for ds_read_b64 (32-lane halves, 64 banks).  Measured: 46 % of the LDS cycles were conflicts.
    // (the distance goes through an opaque register; laundering the pointer itself would lose the
    // LDS address space and turn the reads into flat loads)
    int a_off1 = 16 * PITCH_A, b_off1 = B_C_FAST ? 16 : 16 * PITCH_RK;
    asm volatile("" : "+v"(a_off1), "+v"(b_off1));
    const TL *a_src1 = a_src + a_off1;
    const double *b_src1 = b_src + b_off1;
    for (int k0 = k_begin; k0 < k_end; k0 += KSTEP) {
#ifdef SLK_T64_NO_STAGE  // (timing only: one image, no loads, no barriers after the first step)
        if (k0 == k_begin) {
            __syncthreads();
            stash();
            __syncthreads();
        }
#else
        __syncthreads();  // previous step's LDS reads are done
        stash();
        __syncthreads();
        if (k0 + KSTEP < k_end) {
            la(k0 + KSTEP, ra);
            lb(k0 + KSTEP, rb);
        }
#endif
        // operands of group kk + 4 are read from LDS before the MFMAs of group kk are issued
#ifdef SLK_T64_NO_READS  // (timing only: the MFMAs alone, on whatever the registers hold)
        TL a0n = 1.0, a1n = 2.0;
        double b0n = 3.0, b1n = 4.0;
        asm volatile("" : "+v"(a0n), "+v"(a1n), "+v"(b0n), "+v"(b1n));
#else
        TL a0n = a_src[0], a1n = a_src1[0];
        double b0n = b_src[0], b1n = b_src1[0];
#endif
        // (fence: otherwise these four are fused with the reads of group 4 just below into ds_read2_b64 /
        // ds_read2st64_b64 -- a quarter of all operand reads, and the remaining bank conflicts)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int kk = 0; kk < KSTEP; kk += 4) {
            const double a0 = (double)a0n, a1 = (double)a1n, b0 = b0n, b1 = b1n;  // (AF: widened here, after the LDS read)
#ifndef SLK_T64_NO_READS
            if (kk + 4 < KSTEP) {
                a0n = a_src[kk + 4];
                a1n = a_src1[kk + 4];
                b0n = B_C_FAST ? b_src[(kk + 4) * PITCH_KC] : b_src[kk + 4];
                b1n = B_C_FAST ? b_src1[(kk + 4) * PITCH_KC] : b_src1[kk + 4];
            }
#endif
            asm volatile("" : "+v"(acc.c[0][0]), "+v"(acc.c[0][1])::"memory");  // keep the reads above the MFMAs
            acc.c[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc.c[0][0], 0, 0, 0);
            acc.c[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc.c[0][1], 0, 0, 0);
            acc.c[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc.c[1][0], 0, 0, 0);
            acc.c[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc.c[1][1], 0, 0, 0);
        }
    }
}

// A 128 x 128 variant (64 accumulator doubles per lane, one wave per SIMD) was measured
// SLOWER (33 vs 49 TFLOP/s on the trailing update): the float64 MFMA pipe of this chip only
// saturates with >= 4 waves per SIMD (tools/micro_mfma.py: 32 TFLOP/s at one wave per SIMD,
// 49 at eight), so small tiles at high occupancy win.

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

// Fill an accumulator-shaped register tile: value(row, col), tile-local coordinates (same element -> thread map as above).
template <class F>
__device__ __forceinline__ void tile64_map(Acc64 &acc, F value) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc.c[i][j][r] = value(wr * 32 + i * 16 + (lane >> 4) + 4 * r, wc * 32 + j * 16 + (lane & 15));
}
// Visit two tiles element by element: f(row, col, a, b).
template <class F>
__device__ __forceinline__ void tile64_foreach2(const Acc64 &x, const Acc64 &y, F f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                f(wr * 32 + i * 16 + (lane >> 4) + 4 * r, wc * 32 + j * 16 + (lane & 15), x.c[i][j][r], y.c[i][j][r]);
}

}  // namespace slk
