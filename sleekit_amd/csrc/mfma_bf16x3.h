// float32-grade products on the bfloat16 MFMA: each float32 operand is split into three bfloat16 pieces
// (a = a1 + a2 + a3 exactly, to 24 bits) and the product takes the six piece products above 2^-24,
//     a b ~ a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1),
// every one exact in float32, accumulated in float32 by v_mfma_f32_32x32x16_bf16 (smallest terms first).
// Measured (tools/scratch/mfma_bf16.hip): error / sum|a b| = 9e-8 ... 1.7e-7 for K = 512 ... 4096, the same as a
// sequential float32 fma chain (1.3e-7).  The bfloat16 MFMA does 16x the float32 MFMA's flops per cycle, so six
// terms cost 3/8 of the float32 instruction stream.
//
// One 256-thread workgroup accumulates a 128 x 128 tile; wave w owns the 64 x 64 sub-tile (w >> 1, w & 1) as
// 2 x 2 blocks of 32 x 32.  BOTH operands are given K-contiguous: A[row][k] and Bt[col][k] (the callers' B is
// symmetric, so B[k][col] = B[col][k] and its rows serve), already split into three bfloat16 planes in the
// slab order of k_split3 (splitting in the loop was measured: 656 us against 550, the vector ALU work of
// 7.5 instructions per element and tile).  K goes in steps of 32 through LDS: three planes per operand,
// [row][32 k] bfloat16 with a pitch of 80 bytes, which makes the 16-byte operand reads (lane l: row l & 31,
// k group l >> 5) conflict-free over ds_read_b128's 16-lane groups.
//
// Operand maps of v_mfma_f32_32x32x16_bf16 (checked by the scratch program above):
//   A: lane l holds A[row = l & 31][k = 8 (l >> 5) .. + 7]    B: lane l holds B[k = 8 (l >> 5) .. + 7][col = l & 31]
//   D: as v_mfma_f32_32x32x2_f32 (mfma32.h)
#pragma once

#include "mfma32.h"

namespace slk {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned uint4v_t __attribute__((ext_vector_type(4)));

constexpr int KB16 = 32;          // K depth per LDS round
constexpr int PITCH_B16 = 80;     // bytes per staged row of one plane (64 of data)

struct TileBf16Smem {
    unsigned char a[3][T32 * PITCH_B16];
    unsigned char b[3][T32 * PITCH_B16];
};

// Two floats -> three words holding their bfloat16 pieces pairwise (round to nearest even at each level:
// v_cvt_pk_bf16_f32, gfx950's hardware conversion; the residuals x - piece are exact in float32).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3_pair(float x, float y, unsigned &w1, unsigned &w2, unsigned &w3) {
    const __bf16 x1 = (__bf16)x, y1 = (__bf16)y;
    const float rx = x - (float)x1, ry = y - (float)y1;
    const __bf16 x2 = (__bf16)rx, y2 = (__bf16)ry;
    const float sx = rx - (float)x2, sy = ry - (float)y2;
    const __bf16 x3 = (__bf16)sx, y3 = (__bf16)sy;
    w1 = __builtin_bit_cast(unsigned, (bf16x2_t){x1, y1});
    w2 = __builtin_bit_cast(unsigned, (bf16x2_t){x2, y2});
    w3 = __builtin_bit_cast(unsigned, (bf16x2_t){x3, y3});
}

// acc += A(128 x K) * Bt(128 x K)^T from operands split beforehand (k_split3 in sgemm.hip): plane p of A is a row-major
// bfloat16 image A_p[row][k], likewise Bt_p[col][k].  The loop then has no vector arithmetic at all: six
// 16-byte loads per operand and round, twelve LDS writes, the reads and the MFMAs.
// la(k0, v) / lb(k0, v): v[p][h] = 16 bytes h of plane p of this thread's row (threadIdx.x >> 1), k = k0 + 16 (threadIdx.x & 1) + 8 h ...
template <class LA, class LB>
__device__ __forceinline__ void tile128_mac_planes(Acc128 &acc, TileBf16Smem &sm, int k_begin, int k_end, LA la, LB lb) {
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // A round's MFMAs last ~0.7 us, less than a trip to memory: the loads run TWO rounds ahead, in two register
    // sets used in turn.  Rounds go in pairs and every load is issued unconditionally (clamped to the last
    // round), so that the compiler can count what is in flight (s_waitcnt vmcnt(12), not 0).
    uint4v_t ra0[3][2], rb0[3][2], ra1[3][2], rb1[3][2];
    const int s_off = (t >> 1) * PITCH_B16 + (t & 1) * 32;
    if (k_begin >= k_end) return;
    const int k_last = k_end - KB16;
    la(k_begin, ra0);
    lb(k_begin, rb0);
    la(min(k_begin + KB16, k_last), ra1);
    lb(min(k_begin + KB16, k_last), rb1);
    const int r_off = (lane & 31) * PITCH_B16 + (lane >> 5) * 16;
    auto stash = [&](const uint4v_t(&va)[3][2], const uint4v_t(&vb)[3][2]) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                *reinterpret_cast<uint4v_t *>(sm.a[p] + s_off + 16 * h) = va[p][h];
                *reinterpret_cast<uint4v_t *>(sm.b[p] + s_off + 16 * h) = vb[p][h];
            }
    };
    auto mac = [&]() {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8_t a[2][3], b[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    a[i][p] = *reinterpret_cast<const bf16x8_t *>(sm.a[p] + (wr * 64 + i * 32) * PITCH_B16 + r_off + s * 32);
                    b[i][p] = *reinterpret_cast<const bf16x8_t *>(sm.b[p] + (wc * 64 + i * 32) * PITCH_B16 + r_off + s * 32);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float16_t c = acc.c[i][j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
                    acc.c[i][j] = c;
                }
        }
    };
    // pairs of rounds; an odd count runs its last round once more with a zero weight?  No: every caller's depth
    // is a multiple of 64 or ends on a pair boundary -- handled by the single-round tail below.
    int k0 = k_begin;
    for (; k0 + 2 * KB16 <= k_end; k0 += 2 * KB16) {
        __syncthreads();
        stash(ra0, rb0);
        __syncthreads();
        la(min(k0 + 2 * KB16, k_last), ra0);
        lb(min(k0 + 2 * KB16, k_last), rb0);
        mac();
        __syncthreads();
        stash(ra1, rb1);
        __syncthreads();
        la(min(k0 + 3 * KB16, k_last), ra1);
        lb(min(k0 + 3 * KB16, k_last), rb1);
        mac();
    }
    if (k0 < k_end) {  // one round left: its operands are in set 0
        __syncthreads();
        stash(ra0, rb0);
        __syncthreads();
        mac();
    }
}

// ---- the same product with the operands brought to LDS by the memory pipeline itself (global_load_lds_dwordx4).
// LDS writes are slow (80 bytes a cycle against 256 for reads, profiles/r01_lds_access_microbench.txt): staged through
// registers a round costs 48 ds_write_b128 a workgroup -- more LDS time than all its operand reads -- and 96 VGPRs
// of staging (230 -> 122 VGPRs here).  Measured: error GEMM 520 -> 493 us, Hessian accumulation 0.786 -> 0.771 ms; no
// more, because at 128 x 128 tiles the kernel moves 48 KB from L2 per 6.3 Mflop round -- it runs at the rate the
// CUs can pull operands out of L2 (32 bytes a cycle and CU), not at the LDS's or the MFMA's.  A slab of k_split3 (128 rows x 32 k of one plane, 8 KB contiguous) is instead copied
// verbatim, eight 1 KB wave instructions; no padding is possible in such an image, so the planes are stored
// SWIZZLED: the 16-byte chunk c (of 4) of row r sits at chunk c ^ ((r >> 2) & 3), which makes the operand reads
// (lane l: row l & 31, chunk (l >> 5) + 2 s) conflict-free -- the eight quads of a half-wave split into two sets of
// four with disjoint banks.
struct TileBf16DmaSmem {
    unsigned char a[3][8192];
    unsigned char b[3][8192];
};
__device__ __forceinline__ int swizzled_chunk(int row, int c) { return c ^ ((row >> 2) & 3); }

// SLK_BF16_COPY_AFTER_HALF = 1 (round 4): every operand read of a round is issued up front (a scheduling barrier keeps hipcc from
// sinking the second k-half's reads to their MFMAs), the MFMAs of the first k-half run while the second half's reads land,
// and only then comes the barrier behind which the next round's copy is issued.  Same MFMAs in the same order.  Layer error
// of a 4096 x 4096 layer, alternating on one box: 449 / 474 / 457 us -> 443 / 409 / 423.  (0: wait for all reads, barrier, copy,
// then the round's MFMAs.)
#ifndef SLK_BF16_COPY_AFTER_HALF
#define SLK_BF16_COPY_AFTER_HALF 1
#endif
// a_slabs / b_slabs: this tile's slab of K-step 0 in plane 0 (consecutive steps 4096 elements apart, planes *_plane apart).
__device__ __forceinline__ void tile128_mac_dma(Acc128 &acc, TileBf16DmaSmem &sm, int k_begin, int k_end,
                                                const unsigned short *__restrict__ a_slabs, size_t a_plane,
                                                const unsigned short *__restrict__ b_slabs, size_t b_plane) {
    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    if (k_begin >= k_end) return;
    // this wave's quarter of a slab: 2 KB = two instructions of 64 lanes x 16 bytes
    const unsigned short *ga = a_slabs + (size_t)(k_begin >> 5) * 4096 + wave * 1024 + lane * 8;
    const unsigned short *gb = b_slabs + (size_t)(k_begin >> 5) * 4096 + wave * 1024 + lane * 8;
    const int sw = ((lane & 31) >> 2) & 3;
    const int r_off0 = (lane & 31) * 64 + (((lane >> 5) ^ sw) * 16), r_off1 = (lane & 31) * 64 + ((((lane >> 5) + 2) ^ sw) * 16);
    // One LDS buffer, but the copy of round r + 1 is in flight while the MFMAs of round r run: the operands of a round are
    // read into registers first (barrier: every wave has them), THEN the next round's copy is issued, then the MFMAs.
    // (Measured against issuing the copy at the top of the round and waiting for it there: 0.441 against 0.445 ms for the
    // layer error of a 4096 x 4096 layer -- the three workgroups of a CU already overlap one another's copies; what bounds
    // the kernel is the rate at which a CU pulls operands out of L2, 48 KB per 6.3 Mflop round.)
    auto copy_round = [&](const unsigned short *pa, const unsigned short *pb) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                __builtin_amdgcn_global_load_lds(pa + p * a_plane + i * 512, sm.a[p] + wave * 2048 + i * 1024, 16, 0, 0);
                __builtin_amdgcn_global_load_lds(pb + p * b_plane + i * 512, sm.b[p] + wave * 2048 + i * 1024, 16, 0, 0);
            }
    };
    __syncthreads();  // whatever used this LDS before is done
    copy_round(ga, gb);
    for (int k0 = k_begin; k0 < k_end; k0 += KB16) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's copies have landed before it passes the barrier
        __syncthreads();
        bf16x8_t a[2][2][3], b[2][2][3];  // [s][i][plane]
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    a[s][i][p] = *reinterpret_cast<const bf16x8_t *>(sm.a[p] + (wr * 64 + i * 32) * 64 + (s ? r_off1 : r_off0));
                    b[s][i][p] = *reinterpret_cast<const bf16x8_t *>(sm.b[p] + (wc * 64 + i * 32) * 64 + (s ? r_off1 : r_off0));
                }
        ga += 4096, gb += 4096;
        if (SLK_BF16_COPY_AFTER_HALF) __builtin_amdgcn_sched_barrier(0);  // every read of the round is issued before its first MFMA
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s == SLK_BF16_COPY_AFTER_HALF && k0 + KB16 < k_end) {
                // (the reads above must have completed in EVERY wave before the copy overwrites the buffer)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __syncthreads();
                copy_round(ga, gb);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float16_t c = acc.c[i][j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][i][2], b[s][j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][i][1], b[s][j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][i][0], b[s][j][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][i][1], b[s][j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][i][0], b[s][j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][i][0], b[s][j][0], c, 0, 0, 0);
                    acc.c[i][j] = c;
                }
        }
    }
}

// ---- 256 x 128 tiles (round 4).  At 128 x 128 a round moves 48 KB out of L2 per 6.3 Mflop and the MFMA is busy 0.42 of the
// time (profiles/r03_pmc_mfma.json).  A workgroup of 512 threads (eight waves, each its 64 x 64 sub-tile as before: wave w =
// row quarter w >> 1, column half w & 1) takes TWO 128-row slabs of A against one slab of B: 72 KB per 12.6 Mflop round, 0.75
// of the bytes per flop, one workgroup per CU.  What a round waits for is memory LATENCY -- a fifth of the bytes miss the
// XCD's L2 (an H slab is shared by the two row tiles an XCD walks, no more) and come from the Infinity Cache in 2 us or more,
// while the round's MFMAs last 1.3 -- so the operands travel in three stages: memory -> registers TWO rounds ahead (two
// register sets of nine 16-byte loads a lane, every load unconditional and clamped so that the compiler can count what is in
// flight), registers -> the LDS image the round before last was read from, LDS -> MFMA; two images, ONE barrier per round, and
// the waves drift apart inside a round so that one wave's operand reads fall under another's MFMAs.  (With the copy of round
// r + 1 issued by global_load_lds at the start of round r -- one round ahead, all the two images allow -- a launch took 492 us
// against 518 for the square tiles: a round lasted 3.7 us.)  Same slabs (k_split3's layout is untouched), same six products
// per element in the same order, rounds in the same order: every accumulator is the 128 x 128 kernel's bit for bit.
struct TileBf16TallSmem {
    unsigned char a[2][3][2][8192];
    unsigned char b[2][3][8192];
};
// a_slabs: the slab (row block 2 * tile_y, K-step 0) of plane 0; the second row block lies a_rb elements further.
__device__ __forceinline__ void tile256_mac(Acc128 &acc, TileBf16TallSmem &sm, int k_begin, int k_end,
                                            const unsigned short *__restrict__ a_slabs, size_t a_rb, size_t a_plane,
                                            const unsigned short *__restrict__ b_slabs, size_t b_plane) {
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    if (k_begin >= k_end) return;
    // this thread's share of a round: 16 bytes of each half of a quarter (2 KB) of one of the two A slabs and of an eighth
    // (1 KB) of the B slab, per plane
    const unsigned short *ga = a_slabs + (size_t)(wave >> 2) * a_rb + (size_t)(k_begin >> 5) * 4096 + (wave & 3) * 1024 + lane * 8;
    const unsigned short *gb = b_slabs + (size_t)(k_begin >> 5) * 4096 + wave * 512 + lane * 8;
    const int rounds = (k_end - k_begin) / KB16;
    const int sw = ((lane & 31) >> 2) & 3;
    const int r_off0 = (lane & 31) * 64 + (((lane >> 5) ^ sw) * 16), r_off1 = (lane & 31) * 64 + ((((lane >> 5) + 2) ^ sw) * 16);
    auto fetch = [&](uint4v_t(&x)[9], int round) {
        const size_t o = (size_t)min(round, rounds - 1) * 4096;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            x[3 * p + 0] = *reinterpret_cast<const uint4v_t *>(ga + o + p * a_plane);
            x[3 * p + 1] = *reinterpret_cast<const uint4v_t *>(ga + o + p * a_plane + 512);
            x[3 * p + 2] = *reinterpret_cast<const uint4v_t *>(gb + o + p * b_plane);
        }
    };
    auto stash = [&](int buf, const uint4v_t(&x)[9]) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            *reinterpret_cast<uint4v_t *>(sm.a[buf][p][wave >> 2] + (wave & 3) * 2048 + lane * 16) = x[3 * p + 0];
            *reinterpret_cast<uint4v_t *>(sm.a[buf][p][wave >> 2] + (wave & 3) * 2048 + 1024 + lane * 16) = x[3 * p + 1];
            *reinterpret_cast<uint4v_t *>(sm.b[buf][p] + wave * 1024 + lane * 16) = x[3 * p + 2];
        }
    };
    auto mac = [&](int buf) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8_t a[2][3], b[2][3];  // [i][plane]
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    a[i][p] = *reinterpret_cast<const bf16x8_t *>(sm.a[buf][p][wr >> 1] + ((wr & 1) * 64 + i * 32) * 64 + (s ? r_off1 : r_off0));
                    b[i][p] = *reinterpret_cast<const bf16x8_t *>(sm.b[buf][p] + (wc * 64 + i * 32) * 64 + (s ? r_off1 : r_off0));
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float16_t c = acc.c[i][j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
                    acc.c[i][j] = c;
                }
        }
    };
    uint4v_t xa[9], xb[9];
    fetch(xb, 0);
    fetch(xa, 1);
    __syncthreads();  // whatever used this LDS before is done
    stash(0, xb);
    fetch(xb, 2);
    // round r is multiplied out of image r & 1; meanwhile image (r + 1) & 1 takes round r + 1 from the register set that was
    // loaded two rounds ago, and that set goes out again for round r + 3
    for (int r = 0; r < rounds; r += 2) {
        __syncthreads();  // image 0 holds round r; everybody is through with image 1 (round r - 1)
        if (r + 1 < rounds) stash(1, xa);
        fetch(xa, r + 3);
        mac(0);
        if (r + 1 < rounds) {
            __syncthreads();
            if (r + 2 < rounds) stash(0, xb);
            fetch(xb, r + 4);
            mac(1);
        }
    }
}

// ---- 256 x 256 tiles over K in steps of 16 (round 4, the verdict's form).  At 128 x 128 the kernel runs at the rate a CU
// pulls operands out of L2 (48 KB per 6.3 Mflop); a 256 x 256 tile moves the same 48 KB per 12.6 Mflop.  512 threads, wave w =
// rows 64 (w >> 1), columns 128 (w & 1): 2 x 4 blocks of 32 x 32, 128 accumulator registers.  One workgroup per CU, so nothing
// else hides a round's waits: the operands come by global_load_lds into a RING OF THREE images (144 KB), the copy of round
// r + 2 issued at the top of round r, ONE barrier per round.  hipcc cannot tell an LDS read from the landing of such a copy
// and would wait for every copy in flight before each read; the reads are therefore inline assembly with their own
// s_waitcnt, the barrier is the bare instruction.  Planes in the K16 layout of k_split3 (lay16): slab (256-row block, 16-k
// step) = [k half][256 rows][8 k], 8 KB contiguous per plane -- copied verbatim, read as lane l: row l & 31, half l >> 5:
// 512 contiguous bytes per half-wave, conflict-free.  Same six products per element and 16-k step, steps in order: every
// accumulator is the 128 x 128 kernel's bit for bit.
struct Acc256 {
    float16_t c[2][4];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) c[i][j][r] = 0.0f;
    }
};
constexpr int BIG = 256;        // tile edge
constexpr int BIG_RING = 3;     // images in flight
constexpr int BIG_IMAGE = 2 * 3 * 8192;  // bytes per round: A and B, three planes of [2][256][8] bfloat16 each
struct TileBf16BigSmem {
    unsigned char ring[BIG_RING][BIG_IMAGE];  // [image][operand 0 = A, 1 = B][plane][8192]
};
__device__ __forceinline__ bf16x8_t lds_read16_asm(unsigned addr) {
    bf16x8_t v;
#ifndef SLK_BIG_NO_READS
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
#else
    asm volatile("; no read %0 %1" : "=v"(v) : "v"(addr));  // (timing only)
#endif
    return v;
}
// a_slabs / b_slabs: plane 0 of this tile's slab of K-step 0 (consecutive 16-k steps 4096 elements apart, planes *_plane apart);
// rounds = (k_end - k_begin) / 16.  Ends with every copy landed and every read done (a barrier): the LDS is free afterwards.
__device__ __forceinline__ void tile256sq_mac(Acc256 &acc, TileBf16BigSmem &sm, int k_begin, int k_end,
                                              const unsigned short *__restrict__ a_slabs, size_t a_plane,
                                              const unsigned short *__restrict__ b_slabs, size_t b_plane) {
    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int rounds = (k_end - k_begin) >> 4;
    if (rounds <= 0) return;
    // this wave's eighth of a slab plane: 1 KB = one instruction of 64 lanes x 16 bytes
    const unsigned short *ga = a_slabs + (size_t)(k_begin >> 4) * 4096 + wave * 512 + lane * 8;
    const unsigned short *gb = b_slabs + (size_t)(k_begin >> 4) * 4096 + wave * 512 + lane * 8;
    auto copy_round = [&](int r) {  // round r of this call into image r % 3
        unsigned char *img = sm.ring[r % BIG_RING] + wave * 1024;
        const unsigned short *pa = ga + (size_t)r * 4096, *pb = gb + (size_t)r * 4096;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            __builtin_amdgcn_global_load_lds(pa + p * a_plane, img + p * 8192, 16, 0, 0);
            __builtin_amdgcn_global_load_lds(pb + p * b_plane, img + 3 * 8192 + p * 8192, 16, 0, 0);
        }
    };
    // LDS byte addresses of this lane's operands in image 0: A block i at + 512 i, B block j at + 512 j, plane p at + 8192 p
    const unsigned lds0 = (unsigned)(uintptr_t)&sm.ring[0][0];
    const unsigned a_rd = lds0 + (lane >> 5) * 4096 + (wr * 64 + (lane & 31)) * 16;
    const unsigned b_rd = lds0 + 3 * 8192 + (lane >> 5) * 4096 + (wc * 128 + (lane & 31)) * 16;
    __builtin_amdgcn_s_barrier();  // whatever used this LDS before is done (the caller's earlier reads were waited for)
    copy_round(0);
    if (rounds > 1) copy_round(1);
    for (int r = 0; r < rounds; ++r) {
        // this wave's copies of round r have landed (those of round r + 1, issued a round ago, may still be out)
#if defined(SLK_BIG_NO_WAIT)
        if (r + 1 == rounds) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (timing only: rounds read whatever has landed)
#elif !defined(SLK_BIG_NO_COPY)
        if (r + 1 < rounds) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#ifndef SLK_BIG_NO_BARRIER
        __builtin_amdgcn_s_barrier();  // every wave's have; and every wave is through with the reads of round r - 1
#endif
#ifndef SLK_BIG_NO_COPY
        if (r + 2 < rounds) copy_round(r + 2);  // into the image round r - 1 was read from
#endif
        const unsigned img = (unsigned)(r % BIG_RING) * BIG_IMAGE;
        bf16x8_t a[2][3], b[4][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) a[0][p] = lds_read16_asm(a_rd + img + p * 8192);
#pragma unroll
        for (int p = 0; p < 3; ++p) a[1][p] = lds_read16_asm(a_rd + img + p * 8192 + 512);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) b[j][p] = lds_read16_asm(b_rd + img + p * 8192 + 512 * j);
        // four accumulators' chains side by side (columns j0, j0 + 1 of both row blocks), the six products of each in the
        // 128 x 128 kernel's order
        auto mac4 = [&](int j0) {
            constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc.c[i][j0 + jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[t6]], b[j0 + jj][PB[t6]], acc.c[i][j0 + jj], 0, 0, 0);
        };
        // 18 reads are out, in the order a0 a1 b0 b1 b2 b3 (three planes each); LDS returns in order.
        // (scheduling barriers: nothing ties a wait to the MFMAs before it, and hipcc otherwise hoists all the waits in front
        // of the first MFMA -- the reads would no longer run under the MFMAs)
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]),
                     "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2])::"memory");
        mac4(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[2][0]), "+v"(b[2][1]), "+v"(b[2][2]), "+v"(b[3][0]), "+v"(b[3][1]), "+v"(b[3][2])::"memory");
        mac4(2);
    }
    __builtin_amdgcn_s_barrier();  // every wave is through with the last image
}

}  // namespace slk
