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


// ---------------------------------------------------------------------------------------------------
// The same order for a 256-thread workgroup that sums one row per call, many times (the local search:
// one interaction sum per move).  The tree of a chunk is held in HEAP numbering -- node h has children
// 2h and 2h + 1 -- so every thread derives its node's range by walking the bits of h from the root: no
// serial construction.  A chunk of <= 8192 elements is at most 7 levels deep (a right child is at most
// len / 2 + 7.5 long: <= 8192 / 128 + 15 = 79 at depth 7), i.e. 256 heap slots per chunk; rows of up to
// two chunks (n <= 16384).  Leaves are summed by 8 threads each -- NumPy's 8 interleaved accumulators --
// and combined by xor-shuffles in NumPy's order ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)); one wave then walks
// the levels bottom-up.
struct HeapSum {
    short lo[512], len[512];  // slot = 256 * chunk + heap index (1..255); len 0: no such node
    short leaf[512];          // the slots that are leaves, in no particular order
    int n_leaves;
    float val[512];
};

__device__ __forceinline__ void heap_sum_plan(HeapSum &p, int n) {  // all 256 threads; ends with a barrier
    const int t = threadIdx.x;
    if (t == 0) p.n_leaves = 0;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        int lo = c * NP_CHUNK, len = min(NP_CHUNK, n - c * NP_CHUNK);
        bool ok = t >= 1 && len > 0;
        if (ok) {
            const int depth = 31 - __clz(t);
            for (int b = depth - 1; b >= 0; --b) {
                if (len <= NP_BLOCK) {  // an ancestor is a leaf already
                    ok = false;
                    break;
                }
                const int h = (len / 2) & ~7;
                if ((t >> b) & 1) {
                    lo += h;
                    len -= h;
                } else {
                    len = h;
                }
            }
        }
        p.lo[256 * c + t] = (short)lo;
        p.len[256 * c + t] = ok ? (short)len : (short)0;
        if (ok && len <= NP_BLOCK) p.leaf[atomicAdd(&p.n_leaves, 1)] = (short)(256 * c + t);
    }
    __syncthreads();
}

// position of element j in the staged row: a gap of 8 floats after every 128 keeps the 8 x 8 lanes of a wave
// (8 leaves, 8 accumulators each) on distinct LDS banks when the leaves are 128 apart
__device__ __forceinline__ int heap_sum_pos(int j) { return j + ((j >> 7) << 3); }
__device__ __forceinline__ int heap_sum_floats(int n) { return heap_sum_pos(n) + 8; }

// Sum of the n terms staged at terms[heap_sum_pos(j)] (written before the call; the call starts with a barrier).
// Every thread returns the total.
__device__ __forceinline__ float heap_sum(HeapSum &p, const float *terms, int n) {
    const int t = threadIdx.x;
    __syncthreads();
    for (int u = t; u < 8 * p.n_leaves; u += 256) {
        const int slot = p.leaf[u >> 3], r = u & 7;
        const int lo = p.lo[slot], len = p.len[slot];
        float acc;
        if (len < 8) {
            acc = -0.0f;
            if (r == 0)
                for (int i = 0; i < len; ++i) acc = acc + terms[heap_sum_pos(lo + i)];
        } else {
            const int body = len - (len & 7);
            acc = terms[heap_sum_pos(lo + r)];
            for (int i = 8; i < body; i += 8) acc = acc + terms[heap_sum_pos(lo + i + r)];
            acc = acc + __shfl_xor(acc, 1, 64);
            acc = acc + __shfl_xor(acc, 2, 64);
            acc = acc + __shfl_xor(acc, 4, 64);
            if (r == 0)
                for (int i = body; i < len; ++i) acc = acc + terms[heap_sum_pos(lo + i)];
        }
        if (r == 0) p.val[slot] = acc;
    }
    __syncthreads();
    if (t < 64) {
        for (int d = 6; d >= 0; --d) {
            for (int k = t; k < (2 << d); k += 64) {
                const int base = 256 * (k >> d), h = (1 << d) + (k & ((1 << d) - 1));
                if (p.len[base + h] > NP_BLOCK) p.val[base + h] = p.val[base + 2 * h] + p.val[base + 2 * h + 1];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // one wave, LDS in order: only the compiler must not reorder
            __builtin_amdgcn_wave_barrier();
        }
        if (t == 0) {
            float total = 0.0f + p.val[1];
            if (n > NP_CHUNK) total = total + p.val[257];
            p.val[0] = total;
        }
    }
    __syncthreads();
    return p.val[0];
}

}  // namespace slk
