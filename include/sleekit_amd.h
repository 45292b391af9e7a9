/*
 * sleekit_amd.h -- C ABI of the MI355X (gfx950) GPTQ/OBQ layer-quantization engine.
 *
 * Drop-in boundary for the hot path of Coloquinte/sleekit.  The reference is pure
 * Python/NumPy and has no FFI of its own; each entry point below replaces one
 * NumPy function (cited as file:line relative to the reference tree) and is what
 * a ctypes binding in the reference would bind.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (hipMalloc / torch);
 *     matrices are dense row-major; `R` = output rows of W, `n` = input columns;
 *   - `stream` is a hipStream_t passed as void*; calls only enqueue work and never
 *     synchronise, allocate or copy from host memory, so they can be captured
 *     into a hipGraph (zero-fills and copies inside the library are kernels, not
 *     hipMemsetAsync / hipMemcpyAsync, whose graph nodes did not replay faithfully on ROCm 7.2);
 *   - scratch comes from a caller-provided workspace of at least
 *     slk_workspace_bytes(R, n) bytes, 256-byte aligned; one workspace may be
 *     shared by consecutive calls on the same stream;
 *   - return value: SLK_OK or a negative SLK_E_* code; slk_last_error() gives the
 *     message of the last failure on the calling thread;
 *   - a uniform codebook is (levels, lo, hi): `levels` evenly spaced values on
 *     [lo, hi]; step and zero are formed in float32 exactly as the reference does.
 */
#ifndef SLEEKIT_AMD_H
#define SLEEKIT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLK_OK 0
#define SLK_E_ARG (-1)    /* invalid argument (shape, mode, null pointer)            */
#define SLK_E_NOT_PD (-2) /* reported through `info`, see slk_chol_inverse_upper     */
#define SLK_E_HIP (-3)    /* a HIP runtime call or kernel launch failed              */
#define SLK_E_WS (-4)     /* workspace too small                                     */
/* value of a factorisation's status word (`info`) when its chain of workgroups lost one another: a hand-off between two
 * workgroups of one launch was not seen within 2 s (never observed; the wait is bounded so that nothing can hang the GPU) */
#define SLK_INFO_HANDOFF_TIMEOUT 0x7fffffff

/* Codebooks.  Every entry point that quantizes takes (levels, lo, hi, table):
 *   table == NULL  UniformCodebook(levels, lo, hi)            (sleekit/codebook.py:4-95)
 *   table != NULL  general Codebook of `levels` <= 256 entries (sleekit/codebook.py:98-190): a DEVICE array of
 *                  `levels` increasing float32 values followed by the `levels - 1` bin limits; the index of x
 *                  is np.digitize(x, limits); lo / hi are ignored.
 * codebook maps, slk_codebook_apply `what` */
#define SLK_CB_VALUE 0 /* float32 out */
#define SLK_CB_INDEX 1 /* uint8 out (levels <= 256) */
#define SLK_CB_UP 2    /* float32 out */
#define SLK_CB_DOWN 3  /* float32 out */
#define SLK_CB_INDEX16 4 /* uint16 out (levels <= 65536): UniformCodebook above 256 entries, codebook.py:50-54 */
#define SLK_CB_INDEX32 5 /* uint32 out */

/* column orders, slk_hessian_prepare `order_mode` (obq.py:58-86) */
#define SLK_ORDER_NONE 0
#define SLK_ORDER_DIAG 1
#define SLK_ORDER_ERR 2   /* needs `miss` = column sums of |q(W) - W|   */
#define SLK_ORDER_SQERR 3 /* needs `miss` = column sums of (q(W) - W)^2 */
#define SLK_ORDER_KEYS 4  /* `miss` is reinterpreted as n float64 sort keys (ascending), e.g. from
                             slk_inverse_diag_keys for the inv_diag / combined_diag orders       */

typedef void *slk_stream_t;

/* Library / device ---------------------------------------------------------
 * slk_abi_version: 8.  History: 2 = every quantizing entry takes (levels, lo, hi, table); 3 adds the batch forms
 * (slk_gptq_quantize_batch, slk_row_errors_batch, slk_workspace_bytes_batch) and slk_symmetry_flag; 4 adds the
 * `trace` and `gains` arguments of slk_local_search, slk_set_option / slk_get_option and the batched factorisation
 * (slk_hessian_prepare_batch, slk_chol_inverse_upper_batch, slk_factor_workspace_bytes_batch); 5 adds codebook training
 * (slk_codebook_stats, slk_sort_f32, slk_unique_f32), slk_local_search_batch and slk_factor_unpack_upper_batch; 6 adds
 * slk_chol_inverse_upper_lookahead and slk_release_helpers (the look-ahead is an argument of the call, not a process-wide switch)
 * and turns the loop's `unscale` argument into `flags` (SLK_LOOP_UNSCALE = 1 as before, SLK_LOOP_LATENCY = 2); slk_local_search(_batch)
 * gain `row_err` (the rows' errors after the moves), slk_probe_panel_cycles is new; 7 adds slk_stack_rows and the option "panel_split";
 * 8: no new entry point -- the factorisation's default form is the CHAIN (an outer block's panels in one launch of workgroups that
 * hand the panels on through flags in memory: option "panel_split" 0 / 3; 1 / 2 = round 3's panel kernels), whose status word can
 * read SLK_INFO_HANDOFF_TIMEOUT; the factorisation's workspace holds its flags (slk_factor_workspace_bytes_batch grew by
 * 8 * (ld / 64) bytes per matrix); new options "tall_error", "rows_below_wide".   */
int slk_abi_version(void);
const char *slk_last_error(void);
/* Run-time switches between code paths that give the same results (the tests hold them to that) or that shape a
 * measurement: "no_window2", "no_fast_leaf", "no_defer", "win_dbg", "no_regular_search", "no_fast_search_div",
 * "no_error_splitk", "error_cb", "no_sym_error", "no_bf16_error", "no_bf16_dma", "no_bf16_hessian", "no_bf16_asym", "no_sym_average",
 * "error_f32_below", "no_wave_search", "lookahead" (EVERY factorisation forks the bulk of its outer updates onto a helper stream: a measurement
 * switch; one call at a time asks for it through slk_chol_inverse_upper_lookahead instead), "window_rows" (16 or 32 rows per
 * window workgroup, forced; 0 = 32, or 16 under SLK_LOOP_LATENCY), "panel_split" (the factorisation's panel step: 1 = two launches, diagonal tile then
 * the rest, 2 = one launch in which every workgroup below the diagonal tile repeats its pivot chain; 0 = two for batches and from 8192
 * columns up, one otherwise and always in the look-ahead form) (case-insensitive,
 * an "SLK_" prefix is accepted).  Initial values are read ONCE from the environment (SLK_NO_WINDOW2=1 ...);
 * afterwards only these calls change them.  Process-wide, thread-safe; no reference counterpart.              */
int slk_set_option(const char *name, int value);
int slk_get_option(const char *name);
/* Scratch bytes that any call below may use for an (R, n) layer. */
size_t slk_workspace_bytes(int R, int n);

/* a7  UniformCodebook / Codebook .quantize_value/index/up/down  (sleekit/codebook.py:43-95, 150-190)
 *     out[i] = map(x[i]); float32 arithmetic, IEEE divide, round-half-even.    */
int slk_codebook_apply(const float *x, size_t count, int levels, double lo, double hi, const float *table,
                       int what, void *out, slk_stream_t stream);

/* Codebook TRAINING (Lloyd-Max).  OUTSIDE the hot path of SURVEY.md section 8 (section 2 row 3 marks it out of scope): the
 * rest of the `sleekit.codebook` surface, built after the path's own work.  What one round of Codebook.improve / centroids /
 * probabilities / mse reads off the data  (sleekit/codebook.py:190-267), in one pass:
 *     counts[k] = #{i : index(x[i]) == k}                      (np.bincount of quantize_index)
 *     sums[k]   = sum of those x[i], float64                   (-> centroid = sums / counts)
 *     *sqerr    = sum_i (x[i] - value(x[i]))^2, the difference in float32 and the sum in float64  (-> mse = sqerr / count)
 * for a codebook of 1..256 entries (a general one may hold a single value: every x falls in bin 0).
 * by_position != 0: the "bin" of x[i] is the part of np.array_split(x, levels) that POSITION i falls in, the codebook
 * is ignored and *sqerr = 0: the part means of Codebook.equiprobable (codebook.py:327-331) on sorted data.
 * Results do not depend on scheduling: integer counts, sums in 64-bit fixed point scaled by max|x| (exact to
 * max|x| * 2^-(62 - ceil(log2 count))), sqerr over a fixed tree.  count < 2^31.  Device outputs; workspace of
 * slk_codebook_stats_workspace_bytes().                                                                          */
size_t slk_codebook_stats_workspace_bytes(void);
int slk_codebook_stats(const float *x, size_t count, int levels, double lo, double hi, const float *table, int by_position,
                       long long *counts, double *sums, double *sqerr, void *workspace, size_t ws_bytes, slk_stream_t stream);
/* np.sort / np.unique of float32 data on the device, the first step of lloyd_max and of its two initialisations
 * (codebook.py:283, 327, 356).  `out` must not alias the input; *n_out (device) receives the number of distinct
 * values.  Workspace of slk_sort_workspace_bytes(count) serves either call.                                       */
size_t slk_sort_workspace_bytes(size_t count);
int slk_sort_f32(const float *x, size_t count, float *out, void *workspace, size_t ws_bytes, slk_stream_t stream);
int slk_unique_f32(const float *sorted, size_t count, float *out, int *n_out, void *workspace, size_t ws_bytes,
                   slk_stream_t stream);

/* a14 apply_scaling on axis 0  (sleekit/scaling.py:21-25, 73, 80)
 *     invert == 0: out[r][j] = x[r][j] / scale[r]
 *     invert == 1: out[r][j] = x[r][j] / (1.0f / scale[r])   (two IEEE divides)  */
int slk_rows_divide(const float *x, const float *scale, int R, int n, int invert, float *out,
                    slk_stream_t stream);

/* The row shards of `batch` layers (host array of device pointers, each to rows x cols contiguous float32) stacked into
 * dst[batch][rows_padded][cols], the padding rows set to `fill`: what a rank of several does with its rows of a round's
 * layers before the batch entry points (whole 128-row tiles per layer).  No reference counterpart (the reference has
 * one layer, all rows: sleekit/obq.py:310-336); one launch per 64 layers instead of a copy per layer.               */
int slk_stack_rows(const float *const *src, int batch, int rows, int rows_padded, int cols, float fill, float *dst,
                   slk_stream_t stream);

/* a2  remove_input_bias  (sleekit/obq.py:14-25): out = H - mean mean^T (float32). */
int slk_hessian_strip_mean(const float *H, const float *mean, int n, float *out,
                           slk_stream_t stream);

/* a2  remove_dead_values  (sleekit/obq.py:28-35), in place:
 *     H[d][d] = mean(diag H) and W[:, d] = 0 for every d with H[d][d] == 0.
 *     The mean follows NumPy's float32 pairwise order. `W` may be NULL (R = 0). */
int slk_hessian_patch_dead(float *H, float *W, int R, int n, void *workspace, size_t ws_bytes,
                           slk_stream_t stream);

/* a1  Sleekit.add_batch, Linear branch  (sleekit/statistics.py:41-43, 76-87)
 *     X: T tokens x n features, row-major.  With c = count_before, c' = c + T:
 *     mean = mean * (c/c') + colsum(X) / c';   H = H * (c/c') + X^T X / c'.
 *     Matches the reference to float32 GEMM tolerance: float32 MFMA, or -- n a multiple of 128 and a
 *     workspace given (6 n bytes per token, chunked to what fits) -- float32-grade products of three
 *     bfloat16 pieces per operand on the bfloat16 MFMA.  workspace may be NULL.                    */
int slk_hessian_accumulate(float *H, float *mean, const float *X, int n, int T,
                           long long count_before, void *workspace, size_t ws_bytes, slk_stream_t stream);

/* a4  column statistics for the err / sqerr orders  (sleekit/obq.py:60-69)
 *     miss[j] = sum over rows, in row order, of |q(W) - W| (squared == 0) or its square. */
int slk_column_miss(const float *W, int R, int n, int levels, double lo, double hi, const float *table,
                    int squared, float *miss, slk_stream_t stream);

/* a3+a4+a5  damping, ordering, permutation  (sleekit/obq.py:198-204)
 *     Hd = float64(H) + float32(damp * mean(diag H)) * I
 *     order = argsort(-diag(Hd) [* miss])            (stable on ties)
 *     order_out[n] (int64) and the permuted, index-reversed damped Hessian that
 *     slk_chol_inverse_upper consumes are written to the workspace-independent
 *     outputs `order_out` and `A` (float64, ld = slk_factor_ld(n)).            */
int slk_hessian_prepare(const float *H, int n, float damp, int order_mode, const float *miss,
                        long long *order_out, double *A, void *workspace, size_t ws_bytes,
                        slk_stream_t stream);
/* Sort keys of the orders that need diag(Hd^-1) (sleekit/obq.py:70-75): given the factor U of the
 * damped Hessian in its ORIGINAL order (order_mode NONE), keys[j] = diag(Hd^-1)[j] = sum_i U[i][j]^2
 * (combined == 0, "inv_diag") or -diag(Hd)[j] / diag(Hd^-1)[j] (combined != 0, "combined_diag").  */
int slk_inverse_diag_keys(const double *U, const float *H, int n, float damp, int combined, double *keys,
                          void *workspace, size_t ws_bytes, slk_stream_t stream);
/* Sort keys of the "pivot" order (greedy pivoted Cholesky, sleekit/obq.py:78, 140-166): keys[j] = the
 * step at which column j of Hd = float64(H) + float32(damp * mean(diag H)) * I is picked.  A cold path:
 * 2 n small launches, n^3 / 6 divide-subtract terms; needs 8 n^2 + O(n) bytes of workspace.          */
int slk_pivot_keys(const float *H, int n, float damp, double *keys, void *workspace, size_t ws_bytes,
                   slk_stream_t stream);
/* Leading dimension (and row count) of the padded float64 matrices A / scratch. */
int slk_factor_ld(int n);
/* Same layout from a float64 matrix as is (no damping, no order): A = reversed lower
 * triangle of M, padded.  Entry for compute_hessian_chol on a caller's own matrix.  */
int slk_factor_load(const double *M, int n, double *A, slk_stream_t stream);

/* a6  compute_hessian_chol  (sleekit/obq.py:38-55)
 *     A: output of slk_hessian_prepare (destroyed).  U: n x n float64 row-major,
 *     upper triangular with U^T U = Hd[order][:, order]^-1, zeros below the diagonal.
 *     info[0] = 0 on success, else 1 + the (permuted) column where a non-positive
 *     pivot appeared -- the reference raises numpy.linalg.LinAlgError there.    */
int slk_chol_inverse_upper(double *A, int n, double *U, int *info, void *workspace,
                           size_t ws_bytes, slk_stream_t stream);
/*     The same with LOOK-AHEAD: after a block's panels only the next block's tile columns are updated on `stream`, the
 *     rest of the trailing triangle runs on a helper stream beside the next block's panels (forked and joined by events
 *     inside the call: the caller still sees one stream).  Same updates per tile in the same order: U bit for bit.  For
 *     ONE layer at a time (latency); with several factorisations in flight on streams of their own the plain form is
 *     faster.  The helper stream and its events are made at first use per (device, stream) and live until
 *     slk_release_helpers(), which drains BOTH the helper streams and the caller streams they belong to, then destroys
 *     the helpers (call it before destroying such a stream).  The one call of this library that is not safe beside others:
 *     no look-ahead factorisation may be enqueued from another thread while it runs.                                        */
int slk_chol_inverse_upper_lookahead(double *A, int n, double *U, int *info, void *workspace,
                                     size_t ws_bytes, slk_stream_t stream);
int slk_release_helpers(void);

/* (a3-a6 for `batch` layers of ONE width n at once: every launch covers all the layers -- blockIdx.z is the layer --
 * so a round of small layers costs the launches of one.  Small layers are bound by the host's launch rate, not by the
 * GPU (a 768-column layer is ~50 launches of a few microseconds each): OPT-125M's 72 layers go from 48 ms to the time
 * of ~10 rounds.  H: HOST array of `batch` device pointers; order_out batch x n; A batch x ld x ld (ld = slk_factor_ld(n));
 * U batch x n x n; info `batch` ints.  Orders: SLK_ORDER_NONE / SLK_ORDER_DIAG.  Same results as the single-layer calls.
 * Workspace: slk_factor_workspace_bytes_batch(batch, n).                                                          */
int slk_hessian_prepare_batch(const float *const *H, int batch, int n, float damp, int order_mode,
                              long long *order_out, double *A, void *workspace, size_t ws_bytes, slk_stream_t stream);
int slk_chol_inverse_upper_batch(double *A, int batch, int n, double *U, int *info, void *workspace,
                                 size_t ws_bytes, slk_stream_t stream);
size_t slk_factor_workspace_bytes_batch(int batch, int n);

/* Multi-GPU payload of one layer's factor: one buffer of 8-byte words,
 *   [0] status word, [1..n] order, then the upper triangle of U row by row,
 * slk_factor_payload_words(n) = n (n + 1) / 2 + n + 1 words -- what the single RCCL broadcast per
 * layer carries.  Unpack restores the square U (zeros below the diagonal).   */
size_t slk_factor_payload_words(int n);
int slk_factor_pack(const double *U, const long long *order, const int *info, int n, void *payload,
                    slk_stream_t stream);
int slk_factor_unpack(const void *payload, int n, double *U, long long *order, int *info,
                      slk_stream_t stream);
/*     ... writing the diagonal and above only: for a U buffer whose lower triangle is zero already (zeroed once,
 *     reused round after round: a third less traffic than rewriting the zeros).                          */
int slk_factor_unpack_upper(const void *payload, int n, double *U, long long *order, int *info,
                            slk_stream_t stream);
/*     ... and a round's payloads (a HOST array of `batch` device pointers) into stacked factors U (batch, n, n),
 *     order (batch, n), info (batch) in one launch; verdict (may be NULL): (batch) int32, the word that FOLLOWS each
 *     payload's slk_factor_payload_words(n) (sleekit_amd.dist appends the root's symmetry verdict of the layer's
 *     Hessian there: the buffers must then be one word longer).                                            */
int slk_factor_unpack_upper_batch(const void *const *payloads, int batch, int n, double *U, long long *order, int *info,
                                  int *verdict, slk_stream_t stream);

/* a5+a8+a9+a10  quantize_opt without local search  (sleekit/obq.py:106-137, 202-213)
 *     W: R x n float32.  scale: per-row divisor applied on load (NULL: W is used as is).
 *     Runs the blocked column-sequential loop in the order `order` with factor U and
 *     the reference's recursion (min_block, num_blocks), float64 updates rounded to
 *     float32 at the reference's rounding points.
 *     order (may be NULL): identity, i.e. _quantize_opt_block on Q as given (obq.py:121-137).
 *     Q (R x n float32, original column order): codebook VALUES in the scaled domain, or -- flags & SLK_LOOP_UNSCALE,
 *       scale given -- de-scaled like quantize_with_scaling's result (scaling.py:80: q / (1 / scale[r])).
 *     flags: SLK_LOOP_UNSCALE | SLK_LOOP_LATENCY.  SLK_LOOP_LATENCY: this layer is alone on the GPU -- the window kernel
 *       takes 16 rows per workgroup (shortest launch) instead of 32 (least chip time, for streams of layers whose kernels
 *       overlap); the results are the same bit for bit.
 *     idx (may be NULL): codebook indices, uint8, original column order.
 *     E_out (may be NULL): the scaled errors E of obq.py:115, R x n, in PROCESSING order. */
#define SLK_LOOP_UNSCALE 1
#define SLK_LOOP_LATENCY 2
int slk_gptq_quantize(const float *W, const float *scale, const long long *order, const double *U,
                      int R, int n, int levels, double lo, double hi, const float *table, int min_block,
                      int num_blocks, int flags, float *Q, uint8_t *idx, float *E_out, void *workspace,
                      size_t ws_bytes, slk_stream_t stream);

/* (e) The same loop over `batch` layers of one shape at once, stacked by rows: W, Q, idx, E_out are
 *     (batch * rows_per_layer) x n, scale has batch * rows_per_layer entries, order is batch x n and U is
 *     batch x n x n (layer b's rows use order[b], U[b]).  Results are those of `batch` separate calls, bit for
 *     bit -- rows never interact (obq.py:106-137) -- but every launch covers all the layers: the row shards of
 *     a multi-GPU run (R / G rows each) fill the chip together where each alone leaves most of it
 *     idle.  batch in 1..64; batch > 1 needs rows_per_layer % 64 == 0 and the orders.
 *     Workspace: slk_workspace_bytes_batch(batch, rows_per_layer, n).                                       */
int slk_gptq_quantize_batch(const float *W, const float *scale, const long long *order, const double *U,
                            int batch, int rows_per_layer, int n, int levels, double lo, double hi,
                            const float *table, int min_block, int num_blocks, int flags, float *Q,
                            uint8_t *idx, float *E_out, void *workspace, size_t ws_bytes, slk_stream_t stream);
size_t slk_workspace_bytes_batch(int batch, int rows_per_layer, int n);

/* a11 channelwise_error  (sleekit/obq.py:89-95): row_err[r] = (W-Q)[r] H (W-Q)[r]^T.
 *     G (may be NULL): the R x n product (W - Q) @ H, reused by the local search. */
int slk_row_errors(const float *W, const float *Q, const float *H, int R, int n, float *row_err,
                   float *G, void *workspace, size_t ws_bytes, slk_stream_t stream);
/*     ... of `batch` layers stacked by rows (see slk_gptq_quantize_batch): H is a HOST array of `batch` device
 *     pointers, one n x n Hessian each; batch > 1 needs rows_per_layer % 128 == 0.                          */
int slk_row_errors_batch(const float *W, const float *Q, const float *const *H, int batch, int rows_per_layer,
                         int n, const int *symmetric, float *row_err, void *workspace, size_t ws_bytes,
                         slk_stream_t stream);
/*     The error of a bit-wise symmetric H takes half the products; the verdict is reached on the device.
 *     `symmetric` (device, `batch` ints, may be NULL = checked inside): verdicts from slk_symmetry_flag, so that
 *     ranks sharing a Hessian check it once (the factor's root) instead of once each.  flag[0] = 1 iff
 *     H[i][j] == H[j][i] bit for bit.                                                                   */
int slk_symmetry_flag(const float *H, int n, int *flag, slk_stream_t stream);

/* a12+a13 quantize_local_search  (sleekit/obq.py:220-358)
 *     W, Q: R x n float32 in the scaled domain; Q is updated in place, idx
 *     (may be NULL) receives the indices of the result.  `moves` best-first
 *     single-weight moves per row.  The interaction sum of every move (obq.py:328) is taken in NumPy's
 *     pairwise order, so that from equal initial gains the moves are the reference's bit for bit.
 *     trace (may be NULL): R x moves int32, the moves taken -- 2 * column + (1: up, 0: down), or -1 from
 *     the first move on at which the row had nothing left to gain (parity tests compare it with the
 *     reference's sequence of moves to find where, if anywhere, a near-tie fell the other way).  A "move" onto the
 *     value a weight already holds changes nothing and the reference repeats it until its moves run out: the search
 *     of that row ends there and the trace says -1 ("stay") from that move on, where a record of the reference's own
 *     do_move() calls would show the repeated non-move.  (The carried gains of gains_mode 1 / 2 then lack the reference's
 *     additions of +0.0: at most the sign of a zero differs.)
 *     gains / gains_mode: the state of the reference's stateful LocalSearchQuantizer (obq.py:234-346) between calls --
 *     R x 2 x n float32, per row the n up-gains then the n down-gains.  gains_mode 0: none (gains may be NULL);
 *     1: the initial gains are built from (W - Q) H as usual, and the gains after the moves are stored (moves == 0
 *     gives the constructor's state, obq.py:259-262); 2: the gains are LOADED, the moves made, the gains stored --
 *     k calls with moves = 1 are then one call with moves = k, bit for bit, like k calls of do_move().
 *     row_err (may be NULL; gains_mode 0 or 1): R float32, the rows' errors (W - Q) H (W - Q)^T AFTER the moves, in the domain of
 *     W and Q: the error before them comes out of the product that makes the initial gains, and every move takes its gain off
 *     it -- how the reference's LocalSearchQuantizer carries `err` (obq.py:254, 290) -- so the layer error of a searched layer
 *     needs no product of its own (a row shard of BLOOM-560M spent as long on that product as on the search).                */
int slk_local_search(const float *W, float *Q, const float *H, int R, int n, int levels, double lo,
                     double hi, const float *table, int moves, uint8_t *idx, int *trace, float *gains,
                     int gains_mode, float *row_err, void *workspace, size_t ws_bytes, slk_stream_t stream);
/* The same search over `batch` layers of one shape stacked by rows (rows [b R, (b + 1) R) of W, Q, idx against H[b], a
 * HOST array of `batch` device pointers; R = rows_per_layer, a multiple of 128 when batch > 1): the results of `batch`
 * separate calls bit for bit, in ONE product and ONE launch of moves -- the row shards of a round on several ranks (a few
 * hundred rows per layer) are bound by the host's launch rate otherwise.  symmetric: as in slk_row_errors_batch (may be NULL).
 * row_err (may be NULL): batch * rows_per_layer float32, the rows' errors after the moves (see slk_local_search).
 * Workspace: slk_workspace_bytes_batch(batch, rows_per_layer, n).                                                     */
int slk_local_search_batch(const float *W, float *Q, const float *const *H, int batch, int rows_per_layer, int n,
                           int levels, double lo, double hi, const float *table, int moves, uint8_t *idx,
                           const int *symmetric, float *row_err, void *workspace, size_t ws_bytes, slk_stream_t stream);

/* Scale selection: the callers' pre-step -------------------------- */
/* compute_non_saturating_scaling (sleekit/scaling.py:44-55): scale[r] = max(max_r / hi_code,
 * min_r / lo_code, 1e-16) with the codebook's extreme values lo_code < 0 < hi_code.          */
int slk_scale_minmax(const float *W, int R, int n, double lo_code, double hi_code, float *scale,
                     slk_stream_t stream);
/* compute_norm_scaling (sleekit/scaling.py:35-41): sqrt(max(mean(row^2), 1e-16)), float32,
 * row sums in NumPy's pairwise order.                                                        */
int slk_scale_norm(const float *W, int R, int n, float *scale, slk_stream_t stream);
/* compute_min_mse_scaling with H = None (hdiag NULL) or a diagonal Hessian (sleekit/scaling.py:
 * 84-134): for each factor f (float32, in order) quantize the row round-to-nearest with scale
 * f * base[r], take the error sum_j [hdiag_j] E_j^2 in NumPy's summation order, keep the first
 * minimum; out[r] = base[r] * best factor.                                                   */
int slk_scale_search(const float *W, const float *base, const float *factors, int n_factors,
                     const float *hdiag, int R, int n, int levels, double lo, double hi, const float *table,
                     float *out, slk_stream_t stream);
/* Book-keeping of the searches whose row errors come from slk_row_errors (full Hessian, OBQ-aware;
 * sleekit/scaling.py:131-133, 187-189): init != 0 resets best_err / best_f to +inf; err != NULL
 * applies `better = err < best_err`.                                                          */
int slk_search_step(const float *err, float factor, int R, float *best_err, float *best_f, int init,
                    slk_stream_t stream);
/* out[r] = a[r] * b[r]  (b != NULL)  or  a[r] * c. */
int slk_scale_times(const float *a, const float *b, float c, int R, float *out, slk_stream_t stream);

/* Diagnostics used by tests ------------------------------------------------ */
/* NumPy-ordered float32 mean of diag(H) -> out[0]. */
int slk_diag_mean(const float *H, int n, float *out, void *workspace, size_t ws_bytes,
                  slk_stream_t stream);
/* Peak probes: every wave of `blocks` 256-thread workgroups issues `iters` x 4 independent
 * v_mfma_f64_16x16x4_f64 (2048 flop each) / v_mfma_f32_32x32x2_f32 (4096 flop each).      */
int slk_probe_mfma_f64(double *sink, int blocks, int iters, slk_stream_t stream);
int slk_probe_mfma_f32(float *sink, int blocks, int iters, slk_stream_t stream);
/* float64 probe with `nacc` (8 or 16) independent accumulators per wave: iters x nacc MFMAs per wave. */
int slk_probe_mfma_f64_acc(double *sink, int blocks, int iters, int nacc, slk_stream_t stream);
/* One wave runs `iters` steps of a dependent chain (mode 0: fma f64, 1: 8 independent fma f64,
 * 2: fma f32, 3: rsq f64, 4: divide f64, 5: divide f32, 6: f64->f32->f64 + mul);
 * out[0] = shader cycles, out[1] = 100 MHz ticks, out[2] = checksum.                      */
int slk_probe_chain(double *out, int iters, int mode, slk_stream_t stream);
/* Debug: cycle counters of workgroup 0 of the window kernel, filled when SLK_WIN_DBG has bit 3 set.
 * host_out: 80 int64 on the HOST (16 counters + 64 per-period entries of the standard-schedule kernel).  Synchronises the device.  No reference counterpart. */
int slk_probe_window_cycles(long long *host_out, int reset);
/* Debug: cycle counters of workgroup 1 (wave 0) of the panel kernel, filled when SLK_WIN_DBG has bit 3 set: 16 int64 on the
 * HOST -- [0] staging, [1] diagonal-tile update, [2] pivot chains, [3] barriers and next-strip blocks, [4] tail up to the last
 * barrier, [5] last column of L21 + stores, [15] launches counted.  Synchronises the device.  No reference counterpart. */
int slk_probe_panel_cycles(long long *host_out, int reset);
/* Debug: the leaf chain alone (32-column leaves, 8-level grid) on one workgroup with one or two waves per
 * SIMD.  out (DEVICE, 2 doubles): cycles wave 0 spent on `iters` leaves, checksum. */
int slk_probe_leaf_chain(double *out, int iters, int waves_per_simd, slk_stream_t stream);

/* Per-launch timing (off by default).  While enabled, every kernel launch is bracketed by
 * HIP events on its own stream.  slk_profile_report synchronises on them and writes a JSON
 * array with, per kernel name, the launch count, total milliseconds and the ALGORITHMIC
 * flops / bytes of those launches (the roofline numerators of DESIGN.md); it returns the
 * length needed, like snprintf.  Do not enable while capturing a hipGraph.          */
int slk_profile_enable(int on);
int slk_profile_reset(void);
int slk_profile_report(char *buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* SLEEKIT_AMD_H */
