// NumPy's float32 pairwise summation order on the device (oracle/npsum.py states the order):
//   chunks of 8192 elements added left to right; inside a chunk, split at (m / 2 rounded down to 8)
//   until the piece has <= 128 elements; a piece is summed with 8 interleaved accumulators.
// Used wherever a float32 sum of the reference decides something downstream: the damping term
// (mean of the Hessian diagonal) and the row errors compared by the scale searches.
// One workgroup cooperates: leaf pieces one per thread, then the tree level by level.
#pragma once

#include "common.h"

namespace slk {

#define NP_CHUNK 8192
#define NP_BLOCK 128
#define NP_MAX_NODES 256

__device__ __forceinline__ float np_piece_sum_lds(const float *a, int m) {
    if (m < 8) {
        float r = -0.0f;
        for (int i = 0; i < m; ++i) r = r + a[i];
        return r;
    }
    float r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i = 8;
    for (; i < m - (m % 8); i += 8) {
        r0 = r0 + a[i + 0];
        r1 = r1 + a[i + 1];
        r2 = r2 + a[i + 2];
        r3 = r3 + a[i + 3];
        r4 = r4 + a[i + 4];
        r5 = r5 + a[i + 5];
        r6 = r6 + a[i + 6];
        r7 = r7 + a[i + 7];
    }
    float res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < m; ++i) res = res + a[i];
    return res;
}

// The summation tree of one chunk of `m` elements, built once per workgroup by thread 0.
struct SumTree {
    int n_nodes, max_depth;
    short lo[NP_MAX_NODES], len[NP_MAX_NODES], left[NP_MAX_NODES], right[NP_MAX_NODES], depth[NP_MAX_NODES];
    float val[NP_MAX_NODES];
};

__device__ void build_tree(SumTree &tr, int m) {
    // breadth-first numbering: children always have larger ids than their parent
    int count = 1, maxd = 0;
    tr.lo[0] = 0;
    tr.len[0] = (short)m;
    tr.depth[0] = 0;
    for (int v = 0; v < count; ++v) {
        const int len = tr.len[v];
        if (len <= NP_BLOCK) {
            tr.left[v] = tr.right[v] = -1;
        } else {
            int h = len / 2;
            h -= h % 8;
            tr.left[v] = (short)count;
            tr.lo[count] = tr.lo[v];
            tr.len[count] = (short)h;
            tr.depth[count] = tr.depth[v] + 1;
            ++count;
            tr.right[v] = (short)count;
            tr.lo[count] = tr.lo[v] + h;
            tr.len[count] = (short)(len - h);
            tr.depth[count] = tr.depth[v] + 1;
            ++count;
        }
        maxd = max(maxd, (int)tr.depth[v]);
    }
    tr.n_nodes = count;
    tr.max_depth = maxd;
}

// NumPy-ordered sum of terms[0 .. m) (all threads participate; result valid everywhere after return).
__device__ float tree_sum(SumTree &tr, const float *terms) {
    for (int v = threadIdx.x; v < tr.n_nodes; v += blockDim.x)
        if (tr.left[v] < 0) tr.val[v] = np_piece_sum_lds(terms + tr.lo[v], tr.len[v]);
    __syncthreads();
    for (int d = tr.max_depth - 1; d >= 0; --d) {
        for (int v = threadIdx.x; v < tr.n_nodes; v += blockDim.x)
            if (tr.depth[v] == d && tr.left[v] >= 0) tr.val[v] = tr.val[tr.left[v]] + tr.val[tr.right[v]];
        __syncthreads();
    }
    return tr.val[0];
}

// Sum over a whole row of n terms staged chunk by chunk: total = ((0 + c0) + c1) + ...
template <class TermFn>
__device__ float row_sum_numpy(SumTree *trees, float *terms, int n, TermFn term) {
    float total = 0.0f;
    for (int base = 0, ci = 0; base < n; base += NP_CHUNK, ++ci) {
        const int m = min(NP_CHUNK, n - base);
        for (int j = threadIdx.x; j < m; j += blockDim.x) terms[j] = term(base + j);
        __syncthreads();
        SumTree &tr = trees[(m == NP_CHUNK || n <= NP_CHUNK) ? 0 : 1];
        total = total + tree_sum(tr, terms);
        __syncthreads();
    }
    return total;
}

__device__ __forceinline__ void prepare_trees(SumTree *trees, int n) {
    if (threadIdx.x == 0) {
        build_tree(trees[0], min(n, NP_CHUNK));
        if (n > NP_CHUNK && n % NP_CHUNK) build_tree(trees[1], n % NP_CHUNK);
    }
    __syncthreads();
}

}  // namespace slk
