// float32 MFMA products on the path:
//   a11  channelwise_error (sleekit/obq.py:89-95): row_err = rowsum(((W-Q) @ H) * (W-Q))
//   a1   Sleekit.add_batch (sleekit/statistics.py:76-87): H = H f + X^T X / c'
// Both are dense contractions on v_mfma_f32_32x32x2_f32 through mfma32.h.
#include <stdlib.h>

#include <algorithm>

#include "mfma32.h"
#include "mfma_bf16x3.h"

namespace slk {

// ------------------------------------------------------------------ row errors
// Tile (rows r0.., cols j0..) of G = D @ H with D = W - Q formed on load.  The epilogue
// multiplies by D again and reduces each row over the tile's 128 columns in a fixed
// order (lane tree, then the two column waves); partial[r][tile] goes to scratch and a
// second kernel adds the tiles left to right, so results are run-to-run identical.
using HPtrs = PtrTable;  // the Hessians of a batch of layers stacked by rows

__global__ __launch_bounds__(256) void k_error_tiles(const float *__restrict__ W, const float *__restrict__ Q,
                                                     const HPtrs hs, int R, int n,
                                                     float *__restrict__ G, float *__restrict__ partial,
                                                     int n_tiles, int vec_ok, const int *__restrict__ sym_flag,
                                                     int bf16_takes_sym, int p_stride, int rpl) {
    __shared__ Tile128Smem sm;
    __shared__ float rowpart[2][T32];
    // Column tiles are rotated by the row-tile index: with the symmetric shortcut below a tile's
    // depth grows with its column, and blocks id, id + 256, ... (same CU) must not all be deep.
    // XCD-aware tile order.  Workgroup i runs on XCD i % 8; each XCD has its own L2.  XCD x takes a
    // contiguous run of the row-major tile list (for a square 4096 layer: four row tiles by all
    // 32 column tiles = its 128 resident workgroups), so its L2 holds 4 row panels of W - Q and one
    // sweep of H instead of every XCD streaming every panel (2.8 GB past the L2 per launch before).
    // Within a row the column is rotated with the row and mirrored on odd rows: with the symmetric
    // shortcut below a tile's depth grows with its column, and the workgroups that share a CU
    // (i, i + 256, ...) must not all be deep.
    const int n_rt = (R + T32 - 1) / T32, n_all = n_tiles * n_rt;
    const int per_xcd = (n_all + 7) / 8;
    const int lin = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (lin >= n_all) return;
    const int tile_y = lin / n_tiles, in_row = lin - tile_y * n_tiles;
    const int rot = (in_row + tile_y * max(1, n_tiles / 4)) % n_tiles;
    const int tile_x = (tile_y & 1) ? n_tiles - 1 - rot : rot;
    const int r0 = tile_y * T32, j0 = tile_x * T32;
    const int layer = r0 / rpl;  // rows [b rpl, (b + 1) rpl) belong to layer b (rpl a multiple of the tile when there are several)
    const float *__restrict__ H = hs.p[layer];
    if (sym_flag) sym_flag += layer;
    const int t = threadIdx.x;
    Acc128 acc;
    acc.zero();
    const int kfull = (n + K32 - 1) / K32 * K32;
    // H symmetric (checked bit-wise on the device) and G not wanted:
    //   sum_k D_k H_kj over all k  ==  2 * sum_{k < j0} + the 128-wide diagonal band, after the
    //   final multiplication by D_j and the sum over j.  Halves the flops of the layer error.
    const bool sym = sym_flag != nullptr && sym_flag[0] > 0;  // (a negative verdict = unknown: the general route)
    if (sym && bf16_takes_sym) return;  // (that layer's rows are the bfloat16 kernel's)
    const int kend = sym ? min(j0 + T32, kfull) : kfull;
    const int ksplit = sym ? j0 : 0;  // [0, ksplit) counted twice
    const int a_row = r0 + (t >> 1), a_k = (t & 1) * 8;  // A: D = W - Q, K contiguous
    const int b_k = t >> 4, b_col = j0 + (t & 15) * 8;   // B: H, columns contiguous
    auto twice = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc.c[i][j][r] = acc.c[i][j][r] * 2.0f;
    };
    if (vec_ok && r0 + T32 <= R && j0 + T32 <= n && kfull == n) {
        // interior tile: unconditional 16-byte loads
        const float *pw = W + (size_t)a_row * n + a_k, *pq = Q + (size_t)a_row * n + a_k;
        const float *ph = H + (size_t)b_k * n + b_col;
        auto la = [&](int k0, float(&v)[8]) {
            float w[8], q[8];
            load8<true>(pw + k0, w);
            load8<true>(pq + k0, q);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = w[e] - q[e];
        };
        auto lb = [&](int k0, float(&v)[8]) { load8<true>(ph + (size_t)k0 * n, v); };
        if (ksplit > 0) {
            tile128_mac<true, true>(acc, sm, 0, ksplit, la, lb);
            twice();
            __syncthreads();
        }
        tile128_mac<true, true>(acc, sm, ksplit, kend, la, lb);
    } else if (ksplit > 0) {
        const bool row_ok = a_row < R;
        const size_t arow = (size_t)min(a_row, R - 1) * n;
        auto la = [&](int k0, float(&v)[8]) {
            const int k = min(k0 + a_k, n - 1), last = n - 1 - (k0 + a_k);
            float w[8], q[8];
            load8_guarded(W + arow + k, last, row_ok, w);
            load8_guarded(Q + arow + k, last, row_ok, q);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = w[e] - q[e];
        };
        auto lb = [&](int k0, float(&v)[8]) {
            const int k = k0 + b_k;
            load8_guarded(H + (size_t)min(k, n - 1) * n + min(b_col, n - 1), n - 1 - b_col, k < n, v);
        };
        tile128_mac<true, true>(acc, sm, 0, ksplit, la, lb);
        twice();
        __syncthreads();
        tile128_mac<true, true>(acc, sm, ksplit, kend, la, lb);
    } else {
        const bool row_ok = a_row < R;
        const size_t arow = (size_t)min(a_row, R - 1) * n;
        tile128_mac<true, true>(
            acc, sm, 0, kend,
            [&](int k0, float(&v)[8]) {
                const int k = min(k0 + a_k, n - 1), last = n - 1 - (k0 + a_k);
                float w[8], q[8];
                load8_guarded(W + arow + k, last, row_ok, w);
                load8_guarded(Q + arow + k, last, row_ok, q);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = w[e] - q[e];
            },
            [&](int k0, float(&v)[8]) {
                const int k = k0 + b_k;
                load8_guarded(H + (size_t)min(k, n - 1) * n + min(b_col, n - 1), n - 1 - b_col, k < n, v);
            });
    }

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = j0 + wc * 64 + j * 32 + (lane & 31);
                const float v = acc.c[i][j][r];
                if (r0 + row < R && col < n) {
                    const size_t o = (size_t)(r0 + row) * n + col;
                    if (G) G[o] = v;
                    s = s + v * (W[o] - Q[o]);
                }
            }
            // reduce over the 32 lanes that share this row
#pragma unroll
            for (int m = 16; m >= 1; m >>= 1) s = s + __shfl_xor(s, m, 64);
            if ((lane & 31) == 0) rowpart[wc][row] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x < T32 && r0 + threadIdx.x < R)
        partial[(size_t)(r0 + threadIdx.x) * p_stride + tile_x] = rowpart[0][threadIdx.x] + rowpart[1][threadIdx.x];
}

// x (or x - y when y is given), rows x n row-major -> its three bfloat16 pieces in the GEMM's own order:
// plane p at out + p * plane, and inside a plane the slab (128-row block rb, 32-deep k step ks) is one
// contiguous 8 KB run, [row in block][32 k] -- what a workgroup of k_error_tiles_bf16 loads per round, so
// its 16-byte loads walk memory linearly (row-major planes gave 64 contiguous bytes per row and round).
// Rows beyond `rows` in the last block repeat the last row (never stored by the GEMM).  n % 32 == 0.
// One pass at HBM speed; it takes all the splitting arithmetic out of the GEMM, where every element
// would otherwise be split once per tile that uses it (32 times at n = 4096).
// blockIdx.y: the layer of a stack (its own x, flag and planes: out + blockIdx.y * 3 * plane); one launch for a round's
// Hessians -- eight 4 MB matrices split one by one are eight launches of 17 us each, bound by nothing but their number.
// swz == 2: the K16 layout of the 256 x 256 tiles (mfma_bf16x3.h, tile256sq_mac): slab (256-row block rb, 16-k step ks) =
// [k half][256 rows][8 k], no swizzle.
__global__ __launch_bounds__(256) void k_split3(PtrTable xs, const float *__restrict__ y, int rows, int n,
                                                unsigned short *__restrict__ out, size_t plane,
                                                const int *__restrict__ sym_flag, int swz, int band_only, int average) {
    const float *__restrict__ x = xs.p[blockIdx.y];
    out += (size_t)blockIdx.y * 3 * plane;
    // `average`: a square matrix that is NOT symmetric is split as (x + x^T) / 2 -- the quadratic form d x d^T, which is all
    // the layer error wants of it, is the same for both, and the symmetric half-product route then serves every Hessian
    // (the second read is a column gather: twice the time of this pass, a tenth of the products it saves)
    const bool avg = average && sym_flag && sym_flag[blockIdx.y] <= 0;
    if (sym_flag && sym_flag[blockIdx.y] <= 0 && !avg) return;
    const int ksteps = n / 32;
    const size_t quads = plane / 4;  // groups of four consecutive k
    for (size_t qd = (size_t)blockIdx.x * blockDim.x + threadIdx.x; qd < quads; qd += (size_t)gridDim.x * blockDim.x) {
        const size_t e = qd * 4;                      // position in the plane
        const size_t slab = e >> 12;                  // rb * ksteps + ks
        int k4, r, row, kcol;
        if (swz == 2) {
            const int w = (int)(e & 4095), ksteps16 = n / 16;
            const int ks = (int)(slab % ksteps16), rb = (int)(slab / ksteps16);
            r = (w >> 3) & 255;
            k4 = w & 7;
            if (band_only && ks >= 16 * (rb + 1)) continue;  // (tile column rb reads k < 256 (rb + 1))
            row = min(rb * 256 + r, rows - 1);
            kcol = ks * 16 + (w >> 11) * 8 + k4;
        } else {
            k4 = (int)(e & 31), r = (int)((e >> 5) & 127);
            const int ks = (int)(slab % ksteps), rb = (int)(slab / ksteps);
            // (a symmetric H whose error alone is wanted: tile column rb of the GEMM reads the k steps up to its diagonal
            // band only, k < 128 (rb + 1) -- the other half of the plane is never looked at)
            if (band_only && ks >= 4 * (rb + 1)) continue;
            row = min(rb * 128 + r, rows - 1);
            kcol = ks * 32 + k4;
        }
        const size_t src = (size_t)row * n + kcol;
        float4_t v = *reinterpret_cast<const float4_t *>(x + src);
        if (avg) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = 0.5f * (v[c] + x[(size_t)(kcol + c) * n + row]);
        }
        if (y) {
            const float4_t u = *reinterpret_cast<const float4_t *>(y + src);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = v[c] - u[c];
        }
        unsigned w[3][2];
        split3_pair(v[0], v[1], w[0][0], w[1][0], w[2][0]);
        split3_pair(v[2], v[3], w[0][1], w[1][1], w[2][1]);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            // (swz: the 16-byte chunk moves inside its row, see tile128_mac_dma)
            const size_t eo = swz == 1 ? (e & ~(size_t)31) + (size_t)(swizzled_chunk(r, k4 >> 3) * 8 + (k4 & 7)) : e;
            unsigned *o = reinterpret_cast<unsigned *>(out + (size_t)p * plane + eo);
            o[0] = w[p][0];
            o[1] = w[p][1];
        }
    }
}

// The symmetric case on the bfloat16 MFMA (mfma_bf16x3.h): same tile order, same epilogue, the product
// emulated to float32 grade with six bfloat16 terms from operands split by k_split3.  Runs only when the
// device-side flag says H is symmetric (its rows then serve as the K-contiguous B operand) -- the float32
// kernel above takes the other case; each returns at once when the flag is not its own.  n % 128 == 0.
template <bool DMA>
__global__ __launch_bounds__(256) void k_error_tiles_bf16(const float *__restrict__ W, const float *__restrict__ Q,
                                                          const unsigned short *__restrict__ Dp,
                                                          const unsigned short *__restrict__ Hp, int R, int n,
                                                          float *__restrict__ partial, int n_tiles,
                                                          const int *__restrict__ sym_flag, int p_stride, int cb, int rpl,
                                                          float *__restrict__ G, int asym_mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    TileBf16Smem &sm = *reinterpret_cast<TileBf16Smem *>(smem_raw);
    TileBf16DmaSmem &smd = *reinterpret_cast<TileBf16DmaSmem *>(smem_raw);  // DMA: operands copied to LDS by the loads themselves
    __shared__ float rowpart[2][T32];
    // XCD-aware tile order, column-major inside the XCD.  Workgroup i runs on XCD i % 8 (own L2).  XCD x takes
    // the row tiles [x rpx, (x + 1) rpx) (rpx = 4 for a 4096-row layer) and walks the column tiles in the order
    // `rank`; its slot s = rank * rpx + row, so the workgroups resident together (two per CU, 64 per XCD) are
    // ALL the band's rows times 16 column tiles: an H slab is fetched once and serves every row of the band, the
    // band's own D slabs stay in L2.  (Row-major slots left only two of the four rows resident at a time and
    // cost 1.9 GB past the L2 per launch.)  The column ranks go up and down the depths in groups of 8 so that
    // the tiles a CU gets over time (slots s, s + 32, ...) add up to the same depth.
    //
    // Few rows (a row shard of a multi-GPU run): the tiles alone do not fill the chip and the deepest one is the
    // whole critical path, so K is cut into chunks of `cb` 128-blocks (cb > 0) and a tile of depth d = tile_x + 1
    // blocks becomes ceil(d / cb) workgroups, each with its own slot of partial sums.  Workgroup order: the row
    // tiles of one (column tile, chunk) are neighbours, so the H slabs they share are fetched once.
    const int n_rt = (R + T32 - 1) / T32;
    int tile_x, tile_y, p_slot, blk_lo, blk_hi;  // blocks [blk_lo, blk_hi) of K; block tile_x is the diagonal band
    if (cb > 0) {
        tile_y = blockIdx.x % n_rt;
        int sl = blockIdx.x / n_rt;
        p_slot = sl;
        tile_x = 0;
        while (tile_x < n_tiles && sl >= (tile_x + cb) / cb) {
            sl -= (tile_x + cb) / cb;
            ++tile_x;
        }
        if (tile_x >= n_tiles) return;
        blk_lo = sl * cb;
        blk_hi = min(blk_lo + cb, tile_x + 1);
    } else {
        const int rpx = (n_rt + 7) / 8, per_xcd = rpx * n_tiles;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        if (slot >= per_xcd) return;
        const int rank = slot / rpx;
        tile_y = xcd * rpx + (slot - rank * rpx);
        if (tile_y >= n_rt) return;
        const int grp = rank >> 3, in_grp = rank & 7;
        const int g_lo = grp * 8, g_hi = min(n_tiles, g_lo + 8) - 1;
        tile_x = (grp & 1) ? g_hi - in_grp : g_lo + in_grp;
        if (tile_x < g_lo || tile_x > g_hi) return;  // a last group shorter than 8
        p_slot = tile_x;
        blk_lo = 0;
        blk_hi = tile_x + 1;
    }
    const int r0 = tile_y * T32, j0 = tile_x * T32;
    // a batch of layers stacked by rows (rpl rows each, a multiple of the tile): layer b has its own flag and planes of H
    const int layer = r0 / rpl;
    // H not symmetric (or not known to be): its planes then hold H^T (k_split3_transposed), every k is multiplied and
    // nothing counted twice -- when the caller asked for that (asym_mode); otherwise those rows are the float32 kernel's
    const bool full_k = G != nullptr || (sym_flag[layer] <= 0 && asym_mode != 2);  // (2: the planes hold (H + H^T) / 2)
    if (sym_flag[layer] <= 0 && !asym_mode) return;
    Hp += (size_t)layer * 3 * n * n;
    const int t = threadIdx.x;
    Acc128 acc;
    acc.zero();
    // slab (row block, k step) of a plane: 128 x 32 bfloat16, this thread's 32 bytes at (t >> 1, 16 (t & 1))
    const int ksteps = n / 32;
    const size_t d_plane = (size_t)n_rt * T32 * n, h_plane = (size_t)n * n;
    const unsigned short *pd = Dp + (size_t)tile_y * ksteps * 4096 + (size_t)t * 16;
    const unsigned short *ph = Hp + (size_t)tile_x * ksteps * 4096 + (size_t)t * 16;
    auto la = [&](int k0, uint4v_t(&v)[3][2]) {
        const unsigned short *q = pd + (size_t)(k0 >> 5) * 4096;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int h = 0; h < 2; ++h) v[p][h] = *reinterpret_cast<const uint4v_t *>(q + p * d_plane + 8 * h);
    };
    auto lb = [&](int k0, uint4v_t(&v)[3][2]) {
        const unsigned short *q = ph + (size_t)(k0 >> 5) * 4096;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int h = 0; h < 2; ++h) v[p][h] = *reinterpret_cast<const uint4v_t *>(q + p * h_plane + 8 * h);
    };
    // sum_k D_k H_kj over all k == 2 * sum_{k < j0} + the 128-wide diagonal band (see k_error_tiles)
    const int k_lo = blk_lo * T32, k_below = min(blk_hi * T32, j0);  // [k_lo, k_below) lies under the band: twice
    const unsigned short *a_slabs = Dp + (size_t)tile_y * ksteps * 4096, *b_slabs = Hp + (size_t)tile_x * ksteps * 4096;
    if (full_k) {
        // the product itself is wanted (local search: G = (W - Q) H, obq.py:231), or H is not symmetric: every k, nothing
        // counted twice
        if (DMA)
            tile128_mac_dma(acc, smd, 0, n, a_slabs, d_plane, b_slabs, h_plane);
        else
            tile128_mac_planes(acc, sm, 0, n, la, lb);
    } else if (k_below > k_lo) {
        if (DMA)
            tile128_mac_dma(acc, smd, k_lo, k_below, a_slabs, d_plane, b_slabs, h_plane);
        else
            tile128_mac_planes(acc, sm, k_lo, k_below, la, lb);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc.c[i][j][r] = acc.c[i][j][r] * 2.0f;
        __syncthreads();
    }
    if (!full_k && blk_hi == tile_x + 1) {
        if (DMA)
            tile128_mac_dma(acc, smd, j0, j0 + T32, a_slabs, d_plane, b_slabs, h_plane);
        else
            tile128_mac_planes(acc, sm, j0, j0 + T32, la, lb);
    }

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // (epilogue as in k_error_tiles_bf16_tall: the differences of half a sub-tile fetched together -- rows beyond R read the
    // last row and count for nothing --, the sixteen row sums down the lane tree side by side)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float dd[16][2], sv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const bool live = r0 + row < R;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const size_t o = (size_t)min(r0 + row, R - 1) * n + j0 + wc * 64 + j * 32 + (lane & 31);
                dd[r][j] = W[o] - Q[o];
                if (G && live) G[o] = acc.c[i][j][r];
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            float sacc = 0.0f;
            if (r0 + row < R) {
#pragma unroll
                for (int j = 0; j < 2; ++j) sacc = sacc + acc.c[i][j][r] * dd[r][j];
            }
            sv[r] = sacc;
        }
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1)
#pragma unroll
            for (int r = 0; r < 16; ++r) sv[r] = sv[r] + __shfl_xor(sv[r], m, 64);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if ((lane & 31) == 0) rowpart[wc][row] = sv[r];
        }
    }
    __syncthreads();
    if (threadIdx.x < T32 && r0 + threadIdx.x < R)
        partial[(size_t)(r0 + threadIdx.x) * p_stride + p_slot] = rowpart[0][threadIdx.x] + rowpart[1][threadIdx.x];
}

// The same product on 256 x 128 tiles (mfma_bf16x3.h: tile256_mac): 512 threads, two 128-row slabs of D against one slab
// of H per round.  Same column tiles, same rounds, same epilogue per 64 x 64 sub-tile: every partial sum is the 128 x 128
// kernel's bit for bit (tests/test_gpu_parity.py::test_layer_error_on_tall_tiles_is_the_square_tile_kernel).  An option for whole
// layers (at least 2048 rows, a multiple of 256; no K chunks): measured 1-5 % ahead of the square tiles before the epilogues of
// both were batched, behind them since (one workgroup per CU: nothing runs under its epilogue).
constexpr int TALL = 256;
__global__ __launch_bounds__(512) void k_error_tiles_bf16_tall(const float *__restrict__ W, const float *__restrict__ Q,
                                                                  const unsigned short *__restrict__ Dp, const unsigned short *__restrict__ Hp,
                                                                  int R, int n, float *__restrict__ partial, int n_tiles,
                                                                  const int *__restrict__ sym_flag, int p_stride, int rpl, float *__restrict__ G,
                                                                  int asym_mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    TileBf16TallSmem &sm = *reinterpret_cast<TileBf16TallSmem *>(smem_raw);
    __shared__ float rowpart[2][TALL];
    // the tile order of k_error_tiles_bf16 with row tiles of 256: XCD x takes the row tiles [x rpx, (x + 1) rpx) and walks the
    // column tiles up and down the depths in groups of 8
    const int n_rt = R / TALL;
    const int rpx = (n_rt + 7) / 8, per_xcd = rpx * n_tiles;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    if (slot >= per_xcd) return;
    const int rank = slot / rpx;
    const int tile_y = xcd * rpx + (slot - rank * rpx);
    if (tile_y >= n_rt) return;
    const int grp = rank >> 3, in_grp = rank & 7;
    const int g_lo = grp * 8, g_hi = min(n_tiles, g_lo + 8) - 1;
    const int tile_x = (grp & 1) ? g_hi - in_grp : g_lo + in_grp;
    if (tile_x < g_lo || tile_x > g_hi) return;  // a last group shorter than 8
    const int r0 = tile_y * TALL, j0 = tile_x * T32;
    const int layer = r0 / rpl;
    const bool full_k = G != nullptr || (sym_flag[layer] <= 0 && asym_mode != 2);
    if (sym_flag[layer] <= 0 && !asym_mode) return;
    Hp += (size_t)layer * 3 * n * n;
    Acc128 acc;
    acc.zero();
    const int ksteps = n / 32;
    const size_t d_plane = (size_t)(R / T32) * T32 * n, h_plane = (size_t)n * n, a_rb = (size_t)ksteps * 4096;
    const unsigned short *a_slabs = Dp + (size_t)(2 * tile_y) * a_rb, *b_slabs = Hp + (size_t)tile_x * a_rb;
    if (full_k) {
        tile256_mac(acc, sm, 0, n, a_slabs, a_rb, d_plane, b_slabs, h_plane);
    } else {
        if (j0 > 0) {  // [0, j0) lies under the band: twice
            tile256_mac(acc, sm, 0, j0, a_slabs, a_rb, d_plane, b_slabs, h_plane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc.c[i][j][r] = acc.c[i][j][r] * 2.0f;
            __syncthreads();
        }
        tile256_mac(acc, sm, j0, j0 + T32, a_slabs, a_rb, d_plane, b_slabs, h_plane);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // (epilogue: the 32 differences W - Q of a half of the sub-tile are fetched TOGETHER, then the sixteen row sums go down the
    // lane tree side by side -- row by row, four loads -> wait -> five dependent shuffles each, the tail of a tile was 32
    // trips to memory one behind the other.  Same products, same additions in the same order.)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float dd[16][2], sv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const size_t o = (size_t)(r0 + row) * n + j0 + wc * 64 + j * 32 + (lane & 31);
                dd[r][j] = W[o] - Q[o];
                if (G) G[o] = acc.c[i][j][r];
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float sacc = 0.0f;
#pragma unroll
            for (int j = 0; j < 2; ++j) sacc = sacc + acc.c[i][j][r] * dd[r][j];
            sv[r] = sacc;
        }
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1)
#pragma unroll
            for (int r = 0; r < 16; ++r) sv[r] = sv[r] + __shfl_xor(sv[r], m, 64);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if ((lane & 31) == 0) rowpart[wc][row] = sv[r];
        }
    }
    __syncthreads();
    if (threadIdx.x < TALL) partial[(size_t)(r0 + threadIdx.x) * p_stride + tile_x] = rowpart[0][threadIdx.x] + rowpart[1][threadIdx.x];
}

// The same product on 256 x 256 tiles (mfma_bf16x3.h: tile256sq_mac; "tall_error" = 2): half the bytes out of L2 per flop.
// One workgroup (512 threads, 144 KB of LDS) per CU at a time, so the tiles themselves must balance:
//   * the product wanted (G, the local search's): every tile runs all of K -- equal work; XCD x (workgroups x, x + 8, ...)
//     takes patches of 4 x 8 tiles, 12 slab sets per round for its 32 CUs;
//   * the error of a symmetric H alone: tile column x needs the k blocks 0 ... x (the part under its diagonal band twice,
//     the band once), a triangle of depths 1 ... n / 256.  It is cut into ITEMS of two 256-k blocks (one for the band of an
//     even column), each with its own slot of partial row sums; XCD x takes the row tiles x, x + 8, ... and walks the items,
//     the two-block ones first.
// Every accumulator is the 128 x 128 kernel's bit for bit (same products, same order over k); the partial sums of a row are
// added in another order (256 columns and two k blocks per slot), so the row errors agree to rounding, not bit for bit.
__global__ __launch_bounds__(512) void k_error_tiles_bf16_big(const float *__restrict__ W, const float *__restrict__ Q,
                                                                 const unsigned short *__restrict__ Dp, const unsigned short *__restrict__ Hp,
                                                                 int R, int n, float *__restrict__ partial, const int *__restrict__ sym_flag,
                                                                 int p_stride, int rpl, float *__restrict__ G, int asym_mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    TileBf16BigSmem &sm = *reinterpret_cast<TileBf16BigSmem *>(smem_raw);
    __shared__ float rowpart[2][BIG];
    const int n_rt = R / BIG, n_ct = n / BIG;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int tile_x, tile_y, blk_lo, blk_hi, p_slot;
    if (G != nullptr) {
        const int pr_n = (n_rt + 3) / 4, pc_n = (n_ct + 7) / 8;
        const int patch = xcd + 8 * (slot >> 5), in_patch = slot & 31;
        if (patch >= pr_n * pc_n) return;
        tile_y = (patch / pc_n) * 4 + (in_patch & 3);
        tile_x = (patch % pc_n) * 8 + (in_patch >> 2);
        if (tile_y >= n_rt || tile_x >= n_ct) return;
        blk_lo = 0, blk_hi = n_ct, p_slot = tile_x;
    } else {
        const int rows_here = (n_rt - xcd + 7) / 8;  // row tiles xcd, xcd + 8, ...
        if (rows_here <= 0) return;
        const int q = slot / rows_here;
        tile_y = xcd + 8 * (slot - q * rows_here);
        // item q: the two-block chunks of every column first (column x has (x + 1) / 2 of them), then the single blocks of
        // the even columns.  Slots: column x starts at sum_{x' < x} ceil((x' + 1) / 2).
        int count2 = 0;
        for (int x = 0; x < n_ct; ++x) count2 += (x + 1) / 2;
        if (q >= count2 + (n_ct + 1) / 2) return;
        int chunk;
        if (q < count2) {
            int x = 0, left = q;
            while (left >= (x + 1) / 2) left -= (x + 1) / 2, ++x;
            tile_x = x, chunk = left;
            blk_lo = 2 * chunk, blk_hi = 2 * chunk + 2;
        } else {
            tile_x = 2 * (q - count2), chunk = tile_x / 2;
            blk_lo = tile_x, blk_hi = tile_x + 1;
        }
        p_slot = chunk;
        for (int x = 0; x < tile_x; ++x) p_slot += (x + 2) / 2;
    }
    const int r0 = tile_y * BIG, j0 = tile_x * BIG;
    const int layer = r0 / rpl;
    const bool full_k = G != nullptr;
    if (sym_flag[layer] <= 0 && !asym_mode) return;  // (not symmetric and no averaged planes: the float32 kernel's rows)
    Hp += (size_t)layer * 3 * n * n;
    Acc256 acc;
    acc.zero();
    const int ksteps = n / 16;
    const size_t d_plane = (size_t)R * n, h_plane = (size_t)n * n;
    const unsigned short *a_slabs = Dp + (size_t)tile_y * ksteps * 4096, *b_slabs = Hp + (size_t)tile_x * ksteps * 4096;
    if (full_k) {
        tile256sq_mac(acc, sm, 0, n, a_slabs, d_plane, b_slabs, h_plane);
    } else {
        const int k_lo = blk_lo * BIG, k_below = min(blk_hi * BIG, j0);  // [k_lo, k_below) lies under the band: twice
        if (k_below > k_lo) {
            tile256sq_mac(acc, sm, k_lo, k_below, a_slabs, d_plane, b_slabs, h_plane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc.c[i][j][r] = acc.c[i][j][r] * 2.0f;
        }
        if (blk_hi == tile_x + 1) tile256sq_mac(acc, sm, j0, j0 + BIG, a_slabs, d_plane, b_slabs, h_plane);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // epilogue as in k_error_tiles_bf16: a half sub-tile's differences fetched together (rows beyond R do not exist here: whole
    // tiles), the sixteen row sums down the lane tree side by side
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float dd[16][4], sv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const size_t o = (size_t)(r0 + row) * n + j0 + wc * 128 + j * 32 + (lane & 31);
                dd[r][j] = W[o] - Q[o];
                if (G) G[o] = acc.c[i][j][r];
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float sacc = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) sacc = sacc + acc.c[i][j][r] * dd[r][j];
            sv[r] = sacc;
        }
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1)
#pragma unroll
            for (int r = 0; r < 16; ++r) sv[r] = sv[r] + __shfl_xor(sv[r], m, 64);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if ((lane & 31) == 0) rowpart[wc][row] = sv[r];
        }
    }
    __syncthreads();
    if (threadIdx.x < BIG) partial[(size_t)(r0 + threadIdx.x) * p_stride + p_slot] = rowpart[0][threadIdx.x] + rowpart[1][threadIdx.x];
}

// flag[0] = 1 iff H is bit-wise symmetric (flag must be preset to 1).  One workgroup per pair of mirrored
// 64 x 64 tiles (a triangular list of pairs): tile (bi, bj) goes to LDS through coalesced 16-byte loads, tile
// (bj, bi) is read the same way and compared with the transpose out of LDS -- every element is read once.
__global__ __launch_bounds__(256) void k_symmetry_flag(PtrTable hs, int n, int *__restrict__ flag) {
    __shared__ float tile[64][65];
    const float *__restrict__ H = hs.p[blockIdx.y];  // blockIdx.y: the layer of a stack, with a flag of its own
    flag += blockIdx.y;
    // pair index -> (bi >= bj)
    int bi = (int)((sqrtf(8.0f * (float)blockIdx.x + 1.0f) - 1.0f) * 0.5f);
    while ((bi + 1) * (bi + 2) / 2 <= (int)blockIdx.x) ++bi;
    while (bi * (bi + 1) / 2 > (int)blockIdx.x) --bi;
    const int bj = (int)blockIdx.x - bi * (bi + 1) / 2;
    const int t = threadIdx.x, c4 = (t & 15) * 4, r_first = t >> 4;  // 16 threads x 16 bytes per row, 16 rows per pass
    const bool vec = n % 4 == 0 && (uintptr_t)H % 16 == 0;
    bool ok = true;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = r_first + 16 * p, i = bi * 64 + r, j = bj * 64 + c4;
        float4_t v = (float4_t){0.0f, 0.0f, 0.0f, 0.0f};
        if (i < n) {
            if (vec && j + 3 < n) {
                v = *reinterpret_cast<const float4_t *>(H + (size_t)i * n + j);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (j + e < n) v[e] = H[(size_t)i * n + j + e];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[r][c4 + e] = v[e];
    }
    __syncthreads();
    {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = r_first + 16 * p, i = bj * 64 + r, j = bi * 64 + c4;  // the mirrored tile, read row-wise
            if (i >= n) continue;
            float4_t v = (float4_t){0.0f, 0.0f, 0.0f, 0.0f};
            if (vec && j + 3 < n) {
                v = *reinterpret_cast<const float4_t *>(H + (size_t)i * n + j);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (j + e < n) v[e] = H[(size_t)i * n + j + e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (j + e < n) ok = ok && (__float_as_int(v[e]) == __float_as_int(tile[c4 + e][r]));
        }
    }
    if (!ok) flag[0] = 0;
}

__global__ void k_set_flag(int *flag, int v, int count) {
    if ((int)threadIdx.x < count) flag[threadIdx.x] = v;
}

__global__ __launch_bounds__(256) void k_error_reduce(const float *__restrict__ partial, int R, int n_tiles,
                                                      float *__restrict__ row_err) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float s = 0.0f;
    for (int t = 0; t < n_tiles; ++t) s = s + partial[(size_t)r * n_tiles + t];
    row_err[r] = s;
}

// ------------------------------------------------------------------ Hessian accumulation
// Lower tiles of X^T X, mirrored on store so H stays exactly symmetric.
__global__ __launch_bounds__(256) void k_hessian_tiles(float *__restrict__ H, const float *__restrict__ X, int n, int T,
                                                       float factor, float count, int vec_ok) {
    __shared__ Tile128Smem sm;
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj > bi) return;
    const int i0 = bi * T32, j0 = bj * T32;
    const int t = threadIdx.x;
    Acc128 acc;
    acc.zero();
    const int kend = (T + K32 - 1) / K32 * K32;
    // both operands are rows of X (tokens x features): A[r][k] = X[k][i0 + r], B[k][c] = X[k][j0 + c]
    const int x_k = t >> 4, a_col = i0 + (t & 15) * 8, b_col = j0 + (t & 15) * 8;
    if (vec_ok && i0 + T32 <= n && kend == T) {  // j0 <= i0, so the B tile is interior too
        const float *pa = X + (size_t)x_k * n + a_col, *pb = X + (size_t)x_k * n + b_col;
        tile128_mac<false, true>(
            acc, sm, 0, kend, [&](int k0, float(&v)[8]) { load8<true>(pa + (size_t)k0 * n, v); },
            [&](int k0, float(&v)[8]) { load8<true>(pb + (size_t)k0 * n, v); });
    } else {
        tile128_mac<false, true>(
            acc, sm, 0, kend,
            [&](int k0, float(&v)[8]) {
                const int k = k0 + x_k;
                load8_guarded(X + (size_t)min(k, T - 1) * n + min(a_col, n - 1), n - 1 - a_col, k < T, v);
            },
            [&](int k0, float(&v)[8]) {
                const int k = k0 + x_k;
                load8_guarded(X + (size_t)min(k, T - 1) * n + min(b_col, n - 1), n - 1 - b_col, k < T, v);
            });
    }
    tile128_foreach(acc, [&](int r, int c, float v) {
        const int i = i0 + r, j = j0 + c;
        if (i < n && j < n && (bi != bj || j <= i)) {
            const float h = H[(size_t)i * n + j] * factor + v / count;
            H[(size_t)i * n + j] = h;
            if (i != j) H[(size_t)j * n + i] = h;
        }
    });
}

// X (T x n, tokens x features) -> the three bfloat16 pieces of its TRANSPOSE in the GEMM's slab order: slab
// (128-feature block ib, 32-token step ks) is [feature in block][32 tokens], contiguous; tokens beyond T are
// zero.  One workgroup per slab: 32 coalesced rows of 128 floats in, through LDS, 8 KB per plane out.
__global__ __launch_bounds__(256) void k_split3_transposed(PtrTable xs, int n, int T, int t_first, int t_count,
                                                           unsigned short *__restrict__ out, size_t plane, int swz,
                                                           const int *__restrict__ skip_if_symmetric) {
    __shared__ float tile[32][T32 + 1];
    // blockIdx.z: the layer of a stack (its own X, flag and planes)
    const float *__restrict__ X = xs.p[blockIdx.z];
    out += (size_t)blockIdx.z * 3 * plane;
    if (skip_if_symmetric && skip_if_symmetric[blockIdx.z] > 0) return;  // (the layer error: a symmetric H is split as it stands)
    const int ib = blockIdx.x, ks = blockIdx.y;
    const int t = threadIdx.x;
    for (int e = t; e < 32 * T32; e += 256) {
        const int tt = e >> 7, ii = e & 127;
        const int tok = t_first + ks * 32 + tt;
        tile[tt][ii] = (ks * 32 + tt < t_count && tok < T) ? X[(size_t)tok * n + ib * T32 + ii] : 0.0f;
    }
    __syncthreads();
    const int ii = t >> 1, half = (t & 1) * 16;
    unsigned w[3][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) split3_pair(tile[half + 2 * e][ii], tile[half + 2 * e + 1][ii], w[0][e], w[1][e], w[2][e]);
    const size_t slab = ((size_t)ib * gridDim.y + ks) * 4096 + (size_t)ii * 32;
    const int c0 = (t & 1) * 2;  // this thread's two 16-byte chunks of the row
    const int p0 = swz ? swizzled_chunk(ii, c0) : c0, p1 = swz ? swizzled_chunk(ii, c0 + 1) : c0 + 1;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        unsigned short *o = out + p * plane + slab;
        *reinterpret_cast<uint4v_t *>(o + p0 * 8) = (uint4v_t){w[p][0], w[p][1], w[p][2], w[p][3]};
        *reinterpret_cast<uint4v_t *>(o + p1 * 8) = (uint4v_t){w[p][4], w[p][5], w[p][6], w[p][7]};
    }
}

// Lower tiles of X^T X on the bfloat16 MFMA (mfma_bf16x3.h), from the planes of k_split3_transposed; same
// epilogue as k_hessian_tiles.  Tiles of the lower triangle in an XCD-aware order (see k_syrk_triangle).
template <bool DMA>
__global__ __launch_bounds__(256) void k_hessian_tiles_bf16(float *__restrict__ H, const unsigned short *__restrict__ Xp, int n,
                                                            int ksteps, size_t plane, float factor, float count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    TileBf16Smem &sm = *reinterpret_cast<TileBf16Smem *>(smem_raw);
    const int m = n / T32, total = m * (m + 1) / 2, per_xcd = (total + 7) / 8;
    const int lin = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (lin >= total) return;
    int bi = (int)((sqrtf(8.0f * (float)lin + 1.0f) - 1.0f) * 0.5f);
    while (bi * (bi + 1) / 2 > lin) --bi;
    while ((bi + 1) * (bi + 2) / 2 <= lin) ++bi;
    const int bj = lin - bi * (bi + 1) / 2;
    const int t = threadIdx.x;
    Acc128 acc;
    acc.zero();
    const unsigned short *pa = Xp + (size_t)bi * ksteps * 4096 + (size_t)t * 16;
    const unsigned short *pb = Xp + (size_t)bj * ksteps * 4096 + (size_t)t * 16;
    auto la = [&](int k0, uint4v_t(&v)[3][2]) {
        const unsigned short *q = pa + (size_t)(k0 >> 5) * 4096;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int h = 0; h < 2; ++h) v[p][h] = *reinterpret_cast<const uint4v_t *>(q + p * plane + 8 * h);
    };
    auto lb = [&](int k0, uint4v_t(&v)[3][2]) {
        const unsigned short *q = pb + (size_t)(k0 >> 5) * 4096;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int h = 0; h < 2; ++h) v[p][h] = *reinterpret_cast<const uint4v_t *>(q + p * plane + 8 * h);
    };
    if (DMA)
        tile128_mac_dma(acc, *reinterpret_cast<TileBf16DmaSmem *>(smem_raw), 0, ksteps * 32, Xp + (size_t)bi * ksteps * 4096, plane,
                        Xp + (size_t)bj * ksteps * 4096, plane);
    else
        tile128_mac_planes(acc, sm, 0, ksteps * 32, la, lb);
    const int i0 = bi * T32, j0 = bj * T32;
    // (the 64 read-modify-writes of a thread come out as 64 load -> wait -> store round trips one behind the other; fetching
    // the old values first was measured: 310 against 300 us per 2048 tokens -- 256 VGPRs, and two workgroups per CU hide the
    // tail as it is)
    tile128_foreach(acc, [&](int r, int c, float v) {
        const int i = i0 + r, j = j0 + c;
        if (bi != bj || j <= i) {
            const float h = H[(size_t)i * n + j] * factor + v / count;
            H[(size_t)i * n + j] = h;
            if (i != j) H[(size_t)j * n + i] = h;
        }
    });
}

// mean[j] = mean[j] * factor + (sum over the T tokens of X[t][j]) / count  (statistics.py:76-87).
// One workgroup per 32 features: 8 token groups x 32 features, each thread walks its tokens t = g, g + 8, ... with four
// running sums (128-byte row segments, loads independent of one another), then a fixed-order reduction through LDS.
// (One thread per feature walking all T tokens, the first version, took 470 us for 2048 x 4096: longer than the
// Hessian update itself.)
__global__ __launch_bounds__(256) void k_mean_update(float *__restrict__ mean, const float *__restrict__ X, int n, int T,
                                                     float factor, float count) {
    __shared__ float part[8][33];
    const int f = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + f;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (j < n) {
        int t = g;
        for (; t + 24 < T; t += 32) {
            s0 = s0 + X[(size_t)t * n + j];
            s1 = s1 + X[(size_t)(t + 8) * n + j];
            s2 = s2 + X[(size_t)(t + 16) * n + j];
            s3 = s3 + X[(size_t)(t + 24) * n + j];
        }
        for (; t < T; t += 8) s0 = s0 + X[(size_t)t * n + j];
    }
    part[g][f] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && j < n) {
        float s = part[0][f];
#pragma unroll
        for (int q = 1; q < 8; ++q) s = s + part[q][f];
        mean[j] = mean[j] * factor + s / count;
    }
}

}  // namespace slk

using namespace slk;

extern "C" {

static PtrTable one_ptr(const float *p) {
    PtrTable t;
    for (int b = 0; b < 64; ++b) t.p[b] = b == 0 ? p : nullptr;
    return t;
}

static int row_errors_impl(const float *W, const float *Q, const float *const *Hs, int batch, int rpl, int n, float *row_err,
                           float *G, void *workspace, size_t ws_bytes, slk_stream_t stream, const int *sym_known = nullptr) {
    const int R = batch * rpl;
    const int n_tiles = (n + T32 - 1) / T32;
    Arena ws(workspace, ws_bytes);
    // K split of the bfloat16 kernel for few rows (see the kernel): the largest chunk that still gives >= 512 workgroups
    const int n_rt = (R + T32 - 1) / T32;
    int cb = 0, n_slots = n_tiles;
    if (G == nullptr && n_rt * n_tiles < 512 && !opt(OPT_NO_ERROR_SPLITK)) {
        const int force = opt(OPT_ERROR_CB);
        for (int c = force ? force : 16; c >= 2; c >>= 1) {
            int slots = 0;
            for (int x = 0; x < n_tiles; ++x) slots += (x + c) / c;
            if (slots > 4 * n_tiles) break;
            cb = c;
            n_slots = slots;
            if (n_rt * slots >= 512 || force) break;
        }
    }
    // whole layers on 256 x 256 tiles ("tall_error" = 2, k_error_tiles_bf16_big): slots of its own (one per tile column with
    // G, one per item of the triangle without); the float32 kernel's n_tiles slots stay addressable
    const bool big = cb == 0 && R % BIG == 0 && rpl % BIG == 0 && n % BIG == 0 && R / BIG >= 8 && opt(OPT_TALL_ERROR) == 2 &&
                     !opt(OPT_NO_BF16_DMA) && !opt(OPT_NO_SYM_ERROR) && !opt(OPT_NO_BF16_ERROR) && !(G == nullptr && opt(OPT_NO_SYM_AVERAGE));
    int big_items = 0;
    if (big) {
        for (int x = 0; x < n / BIG; ++x) big_items += (x + 2) / 2;
        n_slots = std::max(n_tiles, G ? n / BIG : big_items);
    }
    float *partial = ws.take<float>((size_t)R * n_slots);
    int *sym = ws.take<int>(64);
    if (!partial || !sym) {
        set_error("workspace too small");
        return SLK_E_WS;
    }
    hipStream_t s = as_stream(stream);
    if (cb > 0 || big) zero_async(partial, (size_t)R * n_slots * sizeof(float), s);  // the float32 kernel fills n_tiles slots only
    dim3 grid(8 * ((n_tiles * n_rt + 7) / 8));  // a multiple of the 8 XCDs: see the tile order in the kernel
    bool aligned = ((uintptr_t)W | (uintptr_t)Q) % 16 == 0;
    HPtrs hp;
    for (int b = 0; b < 64; ++b) hp.p[b] = b < batch ? Hs[b] : nullptr;
    for (int b = 0; b < batch; ++b) aligned = aligned && (uintptr_t)Hs[b] % 16 == 0;
    const int vec_ok = n % 4 == 0 && aligned;
    const bool try_sym = !opt(OPT_NO_SYM_ERROR);  // (with G too: the bfloat16 kernel needs H symmetric)
    if (try_sym && sym_known) {
        sym = const_cast<int *>(sym_known);  // verdicts computed elsewhere (slk_symmetry_flag), read only from here on
    } else if (try_sym) {
        SLK_RUN("set_flag", 0, 4, s, k_set_flag<<<1, 64, 0, s>>>(sym, 1, batch));
        const int t64 = (n + 63) / 64;
        SLK_RUN("symmetry_check", 0, 4.0 * n * n * batch, s, k_symmetry_flag<<<dim3(t64 * (t64 + 1) / 2, batch), 256, 0, s>>>(hp, n, sym));
    }
    // the bfloat16 x 3 kernel when the shape allows (16-byte loads, whole tiles of columns).  Its operands are split into
    // planes first, 10 bytes of traffic per element of H and layer; until the end of round 2 a batch with fewer than 1024
    // rows per layer (the row shards of a round on 8 ranks: 512 rows each) took the float32 kernel instead (H as it
    // stands).  Measured again after the non-symmetric route and the banded splits: the bfloat16 kernel wins at every
    // shard size by now (a rank of 8: 4.31 -> 4.14 ms per round; OPT-125M ... BLOOM-560M shards 0.5-1.5 %), so the
    // threshold is a switch that is off by default ("error_f32_below" = rows).
    const int f32_below = opt(OPT_ERROR_F32_BELOW) > 0 ? opt(OPT_ERROR_F32_BELOW) : 0;
    const bool few_rows = G == nullptr && batch > 1 && rpl < f32_below;
    const size_t d_plane = (size_t)n_rt * T32 * n;  // rows padded to whole tiles
    unsigned short *Dp = ws.take<unsigned short>(3 * d_plane), *Hp = ws.take<unsigned short>(3 * (size_t)n * n * batch);
    const int bf16_ok = try_sym && vec_ok && n % T32 == 0 && Dp && Hp && !opt(OPT_NO_BF16_ERROR) && !few_rows;
    // a Hessian that is NOT symmetric stays on the bfloat16 kernel too (planes of H^T, every k) unless K is cut into chunks
    // ... or, when only the error is wanted (no G), is AVERAGED with its transpose on the way into the planes (asym_mode 2):
    // d H d^T = d ((H + H^T) / 2) d^T, so every Hessian takes the symmetric half-product route
    // (256 x 256 tiles: no transposed planes in their layout -- with G a Hessian that is not symmetric goes to the float32 kernel)
    const int asym_mode = !bf16_ok ? 0 : (G == nullptr && !opt(OPT_NO_SYM_AVERAGE)) ? 2 : (cb == 0 && !opt(OPT_NO_BF16_ASYM) && !big) ? 1 : 0;
    if (batch > 1 && !few_rows && !(Dp && Hp)) {
        set_error("workspace too small for the operand planes of %d layers (slk_workspace_bytes_batch)", batch);
        return SLK_E_WS;
    }
    // every layer's rows through the float32 kernel in one launch (those whose flag says "symmetric" are skipped when
    // the bfloat16 kernel has taken them)
    auto f32_all = [&](const int *flags, int bf16_takes_sym) {
        k_error_tiles<<<grid, 256, 0, s>>>(W, Q, hp, R, n, G, partial, n_tiles, vec_ok, flags, bf16_takes_sym, n_slots, rpl);
    };
    // algorithmic flops: the definition (2 R n^2) whichever way they are obtained
    if (bf16_ok) {
        SLK_LDS_OPT_IN(k_error_tiles_bf16<false>, sizeof(TileBf16Smem));
        SLK_LDS_OPT_IN(k_error_tiles_bf16<true>, sizeof(TileBf16DmaSmem));
        const int dma = big ? 2 : !opt(OPT_NO_BF16_DMA);  // operands to LDS by global_load_lds (1: swizzled planes, 2: the K16 layout)
        SLK_RUN("error_split", 0, 14.0 * R * n, s,
                k_split3<<<2048, 256, 0, s>>>(one_ptr(W), Q, R, n, Dp, d_plane, batch == 1 && !asym_mode ? sym : nullptr, dma, 0, 0));
        {
            // every layer's H in one launch each (blockIdx.y / z = layer)
            const int split_blocks = (int)std::min<size_t>(2048, ((size_t)n * n / 4 + 255) / 256);
            SLK_RUN("error_split", 0, 10.0 * n * n * batch, s,
                    k_split3<<<dim3(split_blocks, batch), 256, 0, s>>>(hp, nullptr, n, n, Hp, (size_t)n * n, sym, dma, G == nullptr,
                                                                       asym_mode == 2));
            if (asym_mode == 1)  // (a layer's blocks return at once when its flag says symmetric)
                SLK_RUN("error_split_t", 0, 0, s,
                        k_split3_transposed<<<dim3(n / T32, n / 32, batch), 256, 0, s>>>(hp, n, n, 0, n, Hp, (size_t)n * n, dma, sym));
        }
        // flops as executed: six bfloat16 products per float32 product, over k <= j only (the definition of the
        // layer error, SURVEY.md 8d, counts 2 R n^2 float32 flops: a third of this, twice over)
        const int wgs = cb > 0 ? n_rt * n_slots : 8 * ((n_rt + 7) / 8) * ((n_tiles + 7) / 8 * 8);  // see the tile order in the kernel
        // whole layers on 256 x 128 tiles: an option ("tall_error" = 1, see k_error_tiles_bf16_tall).  It led by 1-5 % until the
        // epilogues were batched (round 4); since then the square tiles, two workgroups to a CU so that one's epilogue runs
        // under the other's MFMAs, are ahead (451 against 489 us on one box).
        const bool tall = dma && cb == 0 && R % TALL == 0 && rpl % TALL == 0 && R / TALL >= 8 && opt(OPT_TALL_ERROR) == 1;
        if (big) {
            SLK_LDS_OPT_IN(k_error_tiles_bf16_big, sizeof(TileBf16BigSmem));
            const int n_rt_b = R / BIG, n_ct_b = n / BIG;
            const int wgs_big = G ? 8 * 32 * ((((n_rt_b + 3) / 4) * ((n_ct_b + 7) / 8) + 7) / 8)
                                  : 8 * ((n_rt_b + 7) / 8) * big_items;
            SLK_RUN_W("error_gemm_bf16", 6.0 * R * n * (n + (double)T32), 6.0 * R * n + 3.0 * n * n * batch, 2 * wgs_big, s,
                      k_error_tiles_bf16_big<<<wgs_big, 512, sizeof(TileBf16BigSmem), s>>>(W, Q, Dp, Hp, R, n, partial, sym, n_slots, rpl, G,
                                                                                             asym_mode));
        } else if (tall) {
            SLK_LDS_OPT_IN(k_error_tiles_bf16_tall, sizeof(TileBf16TallSmem));
            const int wgs_tall = 8 * ((R / TALL + 7) / 8) * ((n_tiles + 7) / 8 * 8);
            SLK_RUN_W("error_gemm_bf16", 6.0 * R * n * (n + (double)T32), 6.0 * R * n + 3.0 * n * n * batch, 2 * (R / TALL) * n_tiles, s,
                      k_error_tiles_bf16_tall<<<wgs_tall, 512, sizeof(TileBf16TallSmem), s>>>(W, Q, Dp, Hp, R, n, partial, n_tiles, sym, n_slots,
                                                                                              rpl, G, asym_mode));
        } else if (dma)
            SLK_RUN_W("error_gemm_bf16", 6.0 * R * n * (n + (double)T32), 6.0 * R * n + 3.0 * n * n * batch, cb > 0 ? wgs : n_rt * n_tiles, s,
                      k_error_tiles_bf16<true><<<wgs, 256, sizeof(TileBf16DmaSmem), s>>>(W, Q, Dp, Hp, R, n, partial, n_tiles, sym, n_slots,
                                                                                         cb, rpl, G, asym_mode));
        else
            SLK_RUN_W("error_gemm_bf16", 6.0 * R * n * (n + (double)T32), 6.0 * R * n + 3.0 * n * n * batch, cb > 0 ? wgs : n_rt * n_tiles, s,
                      k_error_tiles_bf16<false><<<wgs, 256, sizeof(TileBf16Smem), s>>>(W, Q, Dp, Hp, R, n, partial, n_tiles, sym, n_slots,
                                                                                        cb, rpl, G, asym_mode));
        if (!asym_mode) SLK_RUN("error_gemm_f32", 0, 0, s, f32_all(sym, 1));
    } else {
        SLK_RUN("error_gemm", 2.0 * R * n * n, 8.0 * R * n + 4.0 * n * n * batch + (G ? 4.0 * R * n : 0.0), s,
                f32_all(try_sym && G == nullptr ? sym : nullptr, 0));  // (its symmetric shortcut does not produce G)
    }
    SLK_RUN("error_reduce", 0, 4.0 * R * n_slots, s, k_error_reduce<<<(R + 255) / 256, 256, 0, s>>>(partial, R, n_slots, row_err));
    return SLK_OK;
}

int slk_row_errors(const float *W, const float *Q, const float *H, int R, int n, float *row_err, float *G,
                   void *workspace, size_t ws_bytes, slk_stream_t stream) {
    SLK_REQUIRE(W && Q && H && row_err && R > 0 && n > 0, "bad arguments");
    return row_errors_impl(W, Q, &H, 1, R, n, row_err, G, workspace, ws_bytes, stream);
}

int slk_row_errors_batch(const float *W, const float *Q, const float *const *H, int batch, int rows_per_layer, int n,
                         const int *symmetric, float *row_err, void *workspace, size_t ws_bytes, slk_stream_t stream) {
    SLK_REQUIRE(W && Q && H && row_err && rows_per_layer > 0 && n > 0, "bad arguments");
    SLK_REQUIRE(batch >= 1 && batch <= 64, "batch must be 1..64");
    SLK_REQUIRE(batch == 1 || rows_per_layer % T32 == 0, "a batch needs rows_per_layer to be a multiple of 128");
    for (int b = 0; b < batch; ++b) SLK_REQUIRE(H[b], "null Hessian in the batch");
    return row_errors_impl(W, Q, H, batch, rows_per_layer, n, row_err, nullptr, workspace, ws_bytes, stream, symmetric);
}

int slk_symmetry_flag(const float *H, int n, int *flag, slk_stream_t stream) {
    SLK_REQUIRE(H && flag && n > 0, "bad arguments");
    hipStream_t s = as_stream(stream);
    const int t64 = (n + 63) / 64;
    SLK_RUN("set_flag", 0, 4, s, k_set_flag<<<1, 64, 0, s>>>(flag, 1, 1));
    SLK_RUN("symmetry_check", 0, 4.0 * n * n, s, k_symmetry_flag<<<t64 * (t64 + 1) / 2, 256, 0, s>>>(one_ptr(H), n, flag));
    return SLK_OK;
}

int slk_hessian_accumulate(float *H, float *mean, const float *X, int n, int T, long long count_before,
                           void *workspace, size_t ws_bytes, slk_stream_t stream) {
    SLK_REQUIRE(H && mean && X && n > 0 && T > 0 && count_before >= 0, "bad arguments");
    const long long after = count_before + T;
    const float factor = (float)((double)count_before / (double)after);
    const float count = (float)after;
    hipStream_t s = as_stream(stream);
    const int nt = (n + T32 - 1) / T32;
    SLK_RUN("mean_update", 0, 4.0 * T * n, s, k_mean_update<<<(n + 31) / 32, 256, 0, s>>>(mean, X, n, T, factor, count));
    // bfloat16 x 3 path: whole tiles of features, and room for the planes of at least 32 tokens
    Arena ws(workspace, ws_bytes);
    const size_t room = workspace && ws_bytes > 4096 ? (ws_bytes - 4096) / ((size_t)6 * n) / 32 * 32 : 0;  // tokens per chunk
    if (n % T32 == 0 && room >= 32 && !opt(OPT_NO_BF16_HESSIAN)) {
        SLK_LDS_OPT_IN(k_hessian_tiles_bf16<false>, sizeof(TileBf16Smem));
        SLK_LDS_OPT_IN(k_hessian_tiles_bf16<true>, sizeof(TileBf16DmaSmem));
        const int dma = !opt(OPT_NO_BF16_DMA);
        const int chunk = (int)(room < (size_t)((T + 31) / 32 * 32) ? room : (size_t)((T + 31) / 32 * 32));
        unsigned short *Xp = ws.take<unsigned short>((size_t)3 * n * chunk);
        const int m = n / T32, total = m * (m + 1) / 2;
        for (int t0 = 0; t0 < T; t0 += chunk) {
            const int cnt = T - t0 < chunk ? T - t0 : chunk, ksteps = (cnt + 31) / 32;
            const size_t plane = (size_t)n * ksteps * 32;
            SLK_RUN("hessian_split", 0, 10.0 * cnt * n, s,
                    k_split3_transposed<<<dim3(m, ksteps), 256, 0, s>>>(one_ptr(X), n, T, t0, cnt, Xp, plane, dma, nullptr));
            // the running-mean factor applies once per batch: later chunks add to what the first one scaled
            if (dma)
                SLK_RUN("hessian_syrk_bf16", 6.0 * cnt * n * (n + (double)T32), 6.0 * cnt * n + 8.0 * n * n, s,
                        k_hessian_tiles_bf16<true><<<8 * ((total + 7) / 8), 256, sizeof(TileBf16DmaSmem), s>>>(
                            H, Xp, n, ksteps, plane, t0 == 0 ? factor : 1.0f, count));
            else
                SLK_RUN("hessian_syrk_bf16", 6.0 * cnt * n * (n + (double)T32), 6.0 * cnt * n + 8.0 * n * n, s,
                        k_hessian_tiles_bf16<false><<<8 * ((total + 7) / 8), 256, sizeof(TileBf16Smem), s>>>(
                            H, Xp, n, ksteps, plane, t0 == 0 ? factor : 1.0f, count));
        }
        return SLK_OK;
    }
    dim3 grid(nt, nt);
    SLK_RUN("hessian_syrk", (double)T * n * (n + 1), 4.0 * T * n + 8.0 * n * n, s,
            k_hessian_tiles<<<grid, 256, 0, s>>>(H, X, n, T, factor, count, n % 4 == 0 && (uintptr_t)X % 16 == 0));
    return SLK_OK;
}

}  // extern "C"

namespace slk {
int row_errors_products(const float *W, const float *Q, const float *const *Hs, int batch, int rpl, int n, const int *sym_known,
                        float *row_err, float *G, void *workspace, size_t ws_bytes, slk_stream_t stream) {
    return row_errors_impl(W, Q, Hs, batch, rpl, n, row_err, G, workspace, ws_bytes, stream, sym_known);
}
}  // namespace slk

