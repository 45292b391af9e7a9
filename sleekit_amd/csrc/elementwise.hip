// Element-wise stages of the path: codebook maps, row scaling, Hessian fix-ups,
// column statistics.  All HBM-bound streaming kernels: 16-byte loads where the
// layout allows, grid-stride over at most 2048 blocks (guide: Guideline 11/13).
#include <stdarg.h>

#include "common.h"

namespace slk {

static thread_local char g_error[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

static inline int stream_blocks(size_t work_items, int per_block) {
    size_t b = (work_items + per_block - 1) / per_block;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

// ---------------------------------------------------------------- codebook maps
template <int WHAT>
__global__ __launch_bounds__(256) void k_codebook(const float *__restrict__ x, size_t count, Grid g,
                                                  void *__restrict__ out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const float v = x[i];
        if (WHAT == SLK_CB_VALUE) {
            static_cast<float *>(out)[i] = cb_value(v, g);
        } else if (WHAT == SLK_CB_INDEX) {
            static_cast<uint8_t *>(out)[i] = (uint8_t)cb_index(v, g);
        } else if (WHAT == SLK_CB_INDEX16) {
            static_cast<uint16_t *>(out)[i] = (uint16_t)cb_index(v, g);
        } else if (WHAT == SLK_CB_INDEX32) {
            static_cast<uint32_t *>(out)[i] = (uint32_t)cb_index(v, g);
        } else if (WHAT == SLK_CB_UP) {
            static_cast<float *>(out)[i] = cb_up(v, g);
        } else {
            static_cast<float *>(out)[i] = cb_down(v, g);
        }
    }
}

// ---------------------------------------------------------------- row scaling
__global__ __launch_bounds__(256) void k_rows_divide(const float *__restrict__ x,
                                                     const float *__restrict__ scale, int R, int n,
                                                     int invert, float *__restrict__ out) {
    // one block walks rows; threads walk columns (coalesced)
    for (int r = blockIdx.x; r < R; r += gridDim.x) {
        float s = scale[r];
        if (invert) s = 1.0f / s;
        const float *xr = x + (size_t)r * n;
        float *orow = out + (size_t)r * n;
        for (int j = threadIdx.x; j < n; j += blockDim.x) orow[j] = xr[j] / s;
    }
}

// ---------------------------------------------------------------- stacking row shards
// dst[b][r][:] = src_b[r][:] for r < rows, `fill` for the padding rows up to rows_padded; blockIdx.z = layer
__global__ __launch_bounds__(256) void k_stack_rows(PtrTable srcs, int rows, int rows_padded, int cols, float fill,
                                                    float *__restrict__ dst) {
    const float *__restrict__ src = srcs.p[blockIdx.z];
    float *__restrict__ out = dst + (size_t)blockIdx.z * rows_padded * cols;
    const size_t live = (size_t)rows * cols, all = (size_t)rows_padded * cols;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < all; i += (size_t)gridDim.x * 256) out[i] = i < live ? src[i] : fill;
}

// ---------------------------------------------------------------- H - m m^T
__global__ __launch_bounds__(256) void k_strip_mean(const float *__restrict__ H,
                                                    const float *__restrict__ mean, int n,
                                                    float *__restrict__ out) {
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const float mi = mean[i];
        for (int j = threadIdx.x; j < n; j += blockDim.x) {
            const float p = mi * mean[j];  // np.outer in float32, then one subtraction
            out[(size_t)i * n + j] = H[(size_t)i * n + j] - p;
        }
    }
}

// ---------------------------------------------------------------- dead columns
__global__ __launch_bounds__(256) void k_patch_dead(float *__restrict__ H, float *__restrict__ W,
                                                    int R, int n, const float *__restrict__ fill,
                                                    uint8_t *__restrict__ dead) {
    // pass 1 (block 0 .. ): flag dead columns and patch the diagonal
    const float f = fill[0];
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const bool d = H[(size_t)j * n + j] == 0.0f;
        dead[j] = d;
        if (d) H[(size_t)j * n + j] = f;
    }
}

__global__ __launch_bounds__(256) void k_zero_dead(float *__restrict__ W, int R, int n,
                                                   const uint8_t *__restrict__ dead) {
    for (int r = blockIdx.x; r < R; r += gridDim.x)
        for (int j = threadIdx.x; j < n; j += blockDim.x)
            if (dead[j]) W[(size_t)r * n + j] = 0.0f;
}

// ---------------------------------------------------------------- column miss
// miss[j] = sum_r f(q(W[r][j]) - W[r][j]), accumulated in row order in float32
// (NumPy's axis-0 reduction adds row after row, and the sort that follows sees every bit of it).
// One workgroup per 32 columns.  The order of the ADDS is fixed, everything else is not: all 256 threads evaluate
// the terms of a 256-row chunk (coalesced 128-byte row segments, the next chunk's loads already in flight) into
// LDS, then one thread per column adds its 256 terms top to bottom.  (One thread per column doing everything
// took 1.4 ms for a 4096 x 4096 layer: 4096 dependent trips to memory.)
typedef float float4v_t __attribute__((ext_vector_type(4)));
constexpr int CM_COLS = 32, CM_ROWS = 256;
__global__ __launch_bounds__(256) void k_column_miss(const float *__restrict__ W, int R, int n, Grid g,
                                                     int squared, float *__restrict__ miss, int vec_ok) {
    __shared__ float term[CM_ROWS][CM_COLS + 1];
    const int t = threadIdx.x;
    const int j0 = blockIdx.x * CM_COLS;
    const int lr = t >> 3, c4 = (t & 7) * 4;  // this thread's row inside a pass of 32 rows, its four columns
    float4v_t cur[8], nxt[8];
    auto fetch = [&](int base, float4v_t(&v)[8]) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int r = min(base + p * 32 + lr, R - 1);  // clamped: rows beyond R are never added
            const float *src = W + (size_t)r * n + j0 + c4;
            if (vec_ok && j0 + c4 + 3 < n) {
                v[p] = *reinterpret_cast<const float4v_t *>(src);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[p][e] = (j0 + c4 + e < n) ? src[e] : 0.0f;
            }
        }
    };
    float acc = 0.0f;
    if (R > 0) fetch(0, cur);
    for (int base = 0; base < R; base += CM_ROWS) {
        if (base + CM_ROWS < R) fetch(base + CM_ROWS, nxt);
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float w = cur[p][e];
                const float d = cb_value(w, g) - w;
                term[p * 32 + lr][c4 + e] = squared ? d * d : fabsf(d);
            }
        __syncthreads();
        if (t < CM_COLS) {
            const int rows = min(CM_ROWS, R - base);
            for (int r = 0; r < rows; ++r) acc = acc + term[r][t];
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 8; ++p) cur[p] = nxt[p];
    }
    if (t < CM_COLS && j0 + t < n) miss[j0 + t] = acc;
}

}  // namespace slk

using namespace slk;

extern "C" {

int slk_abi_version(void) { return 8; }

const char *slk_last_error(void) { return g_error; }

int slk_codebook_apply(const float *x, size_t count, int levels, double lo, double hi, const float *table, int what,
                       void *out, slk_stream_t stream) {
    SLK_REQUIRE(levels >= 2 && (table || lo < hi), "codebook needs levels >= 2 and lo < hi");
    SLK_REQUIRE(what >= 0 && what <= 5, "unknown codebook map %d", what);
    SLK_REQUIRE(what != SLK_CB_INDEX || levels <= 256, "uint8 indices need levels <= 256");
    SLK_REQUIRE(what != SLK_CB_INDEX16 || levels <= 65536, "uint16 indices need levels <= 65536");
    SLK_REQUIRE(table == nullptr || levels <= 256, "general codebooks hold at most 256 entries");
    if (count == 0) return SLK_OK;
    SLK_REQUIRE(x && out, "null pointer");
    const Grid g = make_grid(levels, lo, hi, table);
    const int blocks = stream_blocks(count, 256 * 4);
    hipStream_t s = as_stream(stream);
    const double bytes = (double)count * (what == SLK_CB_INDEX ? 5.0 : 8.0);
    switch (what) {
        case SLK_CB_VALUE: SLK_RUN("codebook_apply", 0, bytes, s, k_codebook<SLK_CB_VALUE><<<blocks, 256, 0, s>>>(x, count, g, out)); break;
        case SLK_CB_INDEX: SLK_RUN("codebook_apply", 0, bytes, s, k_codebook<SLK_CB_INDEX><<<blocks, 256, 0, s>>>(x, count, g, out)); break;
        case SLK_CB_UP: SLK_RUN("codebook_apply", 0, bytes, s, k_codebook<SLK_CB_UP><<<blocks, 256, 0, s>>>(x, count, g, out)); break;
        case SLK_CB_INDEX16: SLK_RUN("codebook_apply", 0, bytes, s, k_codebook<SLK_CB_INDEX16><<<blocks, 256, 0, s>>>(x, count, g, out)); break;
        case SLK_CB_INDEX32: SLK_RUN("codebook_apply", 0, bytes, s, k_codebook<SLK_CB_INDEX32><<<blocks, 256, 0, s>>>(x, count, g, out)); break;
        default: SLK_RUN("codebook_apply", 0, bytes, s, k_codebook<SLK_CB_DOWN><<<blocks, 256, 0, s>>>(x, count, g, out)); break;
    }
    return SLK_OK;
}

int slk_rows_divide(const float *x, const float *scale, int R, int n, int invert, float *out,
                    slk_stream_t stream) {
    SLK_REQUIRE(R >= 0 && n >= 0, "negative shape");
    if (R == 0 || n == 0) return SLK_OK;
    SLK_REQUIRE(x && scale && out, "null pointer");
    hipStream_t s = as_stream(stream);
    SLK_RUN("rows_divide", 0, 8.0 * R * n, s, k_rows_divide<<<stream_blocks(R, 1), 256, 0, s>>>(x, scale, R, n, invert, out));
    return SLK_OK;
}

int slk_stack_rows(const float *const *src, int batch, int rows, int rows_padded, int cols, float fill, float *dst,
                   slk_stream_t stream) {
    SLK_REQUIRE(batch >= 0 && rows >= 0 && rows_padded >= rows && cols >= 0, "bad shape");
    if (batch == 0 || rows_padded == 0 || cols == 0) return SLK_OK;
    SLK_REQUIRE(src && dst, "null pointer");
    hipStream_t s = as_stream(stream);
    const size_t per = (size_t)rows_padded * cols;
    for (int b0 = 0; b0 < batch; b0 += 64) {  // (a table of 64 pointers travels by value with each launch)
        const int nb = batch - b0 < 64 ? batch - b0 : 64;
        PtrTable t;
        for (int b = 0; b < 64; ++b) t.p[b] = b < nb ? src[b0 + b] : nullptr;
        for (int b = 0; b < nb; ++b) SLK_REQUIRE(t.p[b] || rows == 0, "null source in the batch");
        const int bx = (int)std::min<size_t>((per + 1023) / 1024, 1024);
        SLK_RUN("stack_rows", 0, 8.0 * per * nb, s, k_stack_rows<<<dim3(bx, 1, nb), 256, 0, s>>>(t, rows, rows_padded, cols, fill, dst + (size_t)b0 * per));
    }
    return SLK_OK;
}

int slk_hessian_strip_mean(const float *H, const float *mean, int n, float *out, slk_stream_t stream) {
    SLK_REQUIRE(n >= 0, "negative shape");
    if (n == 0) return SLK_OK;
    SLK_REQUIRE(H && mean && out, "null pointer");
    hipStream_t s = as_stream(stream);
    SLK_RUN("hessian_strip_mean", 0, 8.0 * n * n, s, k_strip_mean<<<stream_blocks(n, 1), 256, 0, s>>>(H, mean, n, out));
    return SLK_OK;
}

int slk_hessian_patch_dead(float *H, float *W, int R, int n, void *workspace, size_t ws_bytes,
                           slk_stream_t stream) {
    SLK_REQUIRE(n > 0 && R >= 0 && H, "bad arguments");
    Arena ws(workspace, ws_bytes);
    float *fill = ws.take<float>(64);
    uint8_t *dead = ws.take<uint8_t>((size_t)n);
    if (!fill || !dead) {
        set_error("workspace too small");
        return SLK_E_WS;
    }
    size_t used = ws.used;
    int rc = slk_diag_mean(H, n, fill, static_cast<char *>(workspace) + align_up(used, 256),
                           ws_bytes - align_up(used, 256), stream);
    if (rc != SLK_OK) return rc;
    hipStream_t s = as_stream(stream);
    SLK_RUN("patch_dead", 0, 8.0 * n, s, k_patch_dead<<<stream_blocks(n, 256), 256, 0, s>>>(H, W, R, n, fill, dead));
    if (W && R > 0) SLK_RUN("zero_dead", 0, 1.0 * n, s, k_zero_dead<<<stream_blocks(R, 1), 256, 0, s>>>(W, R, n, dead));
    return SLK_OK;
}

int slk_column_miss(const float *W, int R, int n, int levels, double lo, double hi, const float *table, int squared,
                    float *miss, slk_stream_t stream) {
    SLK_REQUIRE(levels >= 2 && (table || lo < hi), "codebook needs levels >= 2 and lo < hi");
    SLK_REQUIRE(R >= 0 && n > 0 && W && miss, "bad arguments");
    hipStream_t s = as_stream(stream);
    SLK_RUN("column_miss", 0, 4.0 * R * n, s,
            k_column_miss<<<(n + CM_COLS - 1) / CM_COLS, 256, 0, s>>>(W, R, n, make_grid(levels, lo, hi, table), squared, miss,
                                                                      n % 4 == 0 && (uintptr_t)W % 16 == 0));
    return SLK_OK;
}

}  // extern "C"
