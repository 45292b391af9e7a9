"""Oracle: GPTQ/OBQ column-sequential quantization with error propagation.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates the hot path of the reference `sleekit/obq.py`:

    strip_input_mean      obq.py:14-25    (remove_input_bias)
    patch_dead_columns    obq.py:28-35    (remove_dead_values)
    inverse_factor_upper  obq.py:38-55    (compute_hessian_chol)
    column_order          obq.py:58-86    (compute_hessian_order)
    pivot_order           obq.py:140-166  (_cholesky_ordering, without forming the trailing matrix)
    row_errors/mean_error obq.py:89-103   (channelwise_error/quantization_error)
    block_schedule        obq.py:121-137  (the recursion of _quantize_opt_block, flattened)
    run_schedule          obq.py:106-137  (_quantize_opt_core + block updates)
    quantize_layer        obq.py:169-217  (quantize_opt)
    flip_gains            obq.py:220-231  (compute_gain)
    local_search          obq.py:234-358  (LocalSearchQuantizer + quantize_local_search)
    move_record           (no counterpart) how close each local-search decision was: parity evidence

The reference recurses over views; here the recursion is flattened once into
an explicit list of operations on absolute column ranges.  Every operation
touches exactly the same memory with the same NumPy call as the reference, so
the results are bit-identical (checked in tests/test_oracle_golden.py).

Mixed precision that is part of the semantics (NumPy >= 2, NEP 50):
  * `np.eye` makes the damped Hessian, hence the factor `U`, float64;
  * `err = (w - q) / U[i, i]` is therefore float64 and so are both updates,
    while `Q` and `E` are float32 stores (one rounding per update).
"""

import numpy as np

LEAF = 0
UPDATE = 1


# --------------------------------------------------------------------------
# Hessian preparation
# --------------------------------------------------------------------------
def strip_input_mean(H, mean):
    """H - m m^T: the part of the Hessian a bias correction absorbs (obq.py:14-25)."""
    assert H.ndim == 2 and mean.ndim == 1
    assert H.shape[0] == H.shape[1] == mean.shape[0]
    return H - np.outer(mean, mean)


def patch_dead_columns(H, W):
    """In place (obq.py:28-35): inputs that never fired get the mean diagonal and zero weights."""
    diag = H.diagonal()
    fill = diag.mean()
    dead = diag == 0
    H[dead, dead] = fill
    W[:, dead] = 0


def inverse_factor_upper(H):
    """Upper-triangular U with U^T U = H^-1 (obq.py:38-55).

    Cholesky of the index-reversed matrix, inverted, reversed back.
    Raises numpy.linalg.LinAlgError when H is not positive definite.
    """
    R = np.flip(H)
    R = np.linalg.cholesky(R)
    R = np.linalg.inv(R)
    return np.ascontiguousarray(np.flip(R))


def column_order(W, H, grid, mode, ties="numpy"):
    """Column processing order (obq.py:58-86), restricted to the hot-path modes.

    `ties`: the reference calls `argsort()` with NumPy's default UNSTABLE sort, so the
    order of columns whose keys are exactly equal is whatever that sort leaves (it depends
    on the NumPy build's SIMD dispatch).  "numpy" reproduces the reference call as is;
    "stable" breaks exact ties by column index, which is what the device kernel does.
    The two differ only when two keys are bit-identical (it happens: float32 diagonals of
    a 4096-column Hessian collide now and then).
    """
    kind = {"numpy": None, "stable": "stable"}[ties]
    if mode == "diag":
        return (-H.diagonal()).argsort(kind=kind)
    if mode == "none":
        return np.arange(W.shape[1])
    if mode == "err":
        miss = np.abs(grid(W) - W).sum(axis=0)
        return (-H.diagonal() * miss).argsort(kind=kind)
    if mode == "sqerr":
        miss = np.square(grid(W) - W).sum(axis=0)
        return (-H.diagonal() * miss).argsort(kind=kind)
    if mode == "combined_diag":  # obq.py:70-72
        return (-H.diagonal() / np.linalg.inv(H).diagonal()).argsort(kind=kind)
    if mode == "inv_diag":  # obq.py:73-75
        return np.linalg.inv(H).diagonal().argsort(kind=kind)
    if mode == "pivot":  # obq.py:76-78
        return pivot_order(H)
    raise RuntimeError(f"Invalid act_order value {mode}")


def pivot_order(H):
    """Greedy pivoted-Cholesky order (obq.py:140-166), restated without the trailing matrix.

    The reference swaps rows/columns of a full copy and applies `L[k+1:, k+1:] -= outer(b, b) / L[k, k]`
    at every step.  Only the order is wanted, so this restatement keeps the pivot rows b_m and the
    diagonal, and evaluates row p_k of the trailing matrix when column p_k is picked:
        M_k[p_k][v] = H[p_k][v] - sum_{m<k} (b_m[p_k] * b_m[v]) / d_m
    term by term in step order, each with the reference's three roundings (product, quotient,
    difference).  Ties in |diagonal| go to the first POSITION, positions following the reference's swaps.
    """
    H = np.asarray(H, dtype=np.float64)
    n = H.shape[0]
    pos = np.arange(n)          # column held by each position (the reference's `order`)
    diag = H.diagonal().copy()
    B = np.zeros((n, n))        # B[m, v] = b_m[v] for the columns v still free at step m
    d = np.zeros(n)
    for k in range(n):
        j = int(np.argmax(np.abs(diag[pos[k:]]))) + k
        pos[[k, j]] = pos[[j, k]]
        p = pos[k]
        d[k] = diag[p]
        rest = pos[k + 1 :]
        x = H[p, rest].copy()
        for m in range(k):
            x = x - (B[m, p] * B[m, rest]) / d[m]
        B[k, rest] = x
        diag[rest] = diag[rest] - (x * x) / d[k]
    return pos


# --------------------------------------------------------------------------
# Error metric
# --------------------------------------------------------------------------
def row_errors(W, Q, H):
    """(W-Q) H (W-Q)^T per output row (obq.py:89-95)."""
    D = W - Q
    return ((D @ H) * D).sum(axis=-1)


def mean_error(W, Q, H):
    """Layer error: mean over rows (obq.py:98-103)."""
    return row_errors(W, Q, H).mean()


# --------------------------------------------------------------------------
# The column-sequential loop
# --------------------------------------------------------------------------
def block_schedule(width, min_block=32, num_blocks=8):
    """Flatten the recursion of obq.py:121-137 into a list of operations.

    Returns tuples
        (LEAF,   c0, c1, 0)   quantize columns [c0, c1) one by one (obq.py:106-118)
        (UPDATE, k0, k1, j1)  Q[:, k1:j1] -= E[:, k0:k1] @ U[k0:k1, k1:j1]   (obq.py:137)
    in execution order, on absolute column indices.  Updates whose target
    range is empty (the last block of every level) are dropped: the reference
    executes them as no-ops on zero-width views.
    """
    assert min_block >= 1
    ops = []
    # Explicit stack of (start, stop) ranges still to be expanded; a range
    # [a, b) lives inside its parent [pa, pb) whose end bounds its update.
    def expand(a, b):
        size = b - a
        if size <= min_block:
            ops.append((LEAF, a, b, 0))
            return
        step = max((size + num_blocks - 1) // num_blocks, min_block)
        for s in range(a, b, step):
            e = min(s + step, b)
            expand(s, e)
            if e < b:
                ops.append((UPDATE, s, e, b))

    expand(0, width)
    return ops


def run_schedule(Q, E, U, grid, ops):
    """Execute a schedule in place on Q (float32 weights -> grid values) and E (errors)."""
    for kind, a, b, c in ops:
        if kind == LEAF:
            for i in range(a, b):
                w = Q[:, i]
                q = grid(w)
                err = (w - q) / U[i, i]
                E[:, i] = err
                Q[:, i] = q
                Q[:, i + 1 : b] -= np.outer(err, U[i, i + 1 : b])
        else:
            Q[:, b:c] -= E[:, a:b] @ U[a:b, b:c]


def quantize_layer(
    W, H, grid, order_mode="diag", damp=0.01, ls_moves=0, min_block=32, num_blocks=8, ties="numpy", ls_records=None
):
    """GPTQ-style quantization of one layer (obq.py:169-217). Returns grid values, float32.
    ls_records: see local_search."""
    assert W.ndim == 2 and H.ndim == 2
    assert H.shape[0] == H.shape[1] == W.shape[1]
    assert min_block >= 1
    W = W.astype(np.float32)
    H = H.astype(np.float32)
    n = H.shape[0]

    H_damped = H + damp * H.diagonal().mean() * np.eye(n)
    order = column_order(W, H_damped, grid, order_mode, ties)

    Wp = W[:, order]
    Q = Wp.copy()
    H_damped = H_damped[order][:, order]
    U = inverse_factor_upper(H_damped)

    E = np.zeros_like(Wp)
    run_schedule(Q, E, U, grid, block_schedule(n, min_block, num_blocks))

    back = np.argsort(order)
    W0 = Wp[:, back]
    Q = Q[:, back]
    return local_search(W0, Q, H, grid, ls_moves, ls_records)


def quantize_layer_debug(W, H, grid, order_mode="diag", damp=0.01, min_block=32, num_blocks=8):
    """Same as quantize_layer without local search, also returning order, U and E (permuted)."""
    W = W.astype(np.float32)
    H = H.astype(np.float32)
    n = H.shape[0]
    H_damped = H + damp * H.diagonal().mean() * np.eye(n)
    order = column_order(W, H_damped, grid, order_mode)
    Wp = W[:, order]
    Q = Wp.copy()
    H_damped = H_damped[order][:, order]
    U = inverse_factor_upper(H_damped)
    E = np.zeros_like(Wp)
    run_schedule(Q, E, U, grid, block_schedule(n, min_block, num_blocks))
    return Q[:, np.argsort(order)], order, U, E


# --------------------------------------------------------------------------
# Local search
# --------------------------------------------------------------------------
def flip_gains(W, Q, H, candidates):
    """Error decrease from moving each weight alone to its candidate (obq.py:220-231)."""
    delta = Q - W
    D = candidates - Q
    return -np.square(D) * H.diagonal() - 2 * (delta @ H) * D


class _SearchState:
    """Per-row best-first search state (obq.py:234-346)."""

    def __init__(self, W, Q, H, grid):
        assert W.ndim == 2 and H.ndim == 2
        assert H.shape[0] == H.shape[1] == W.shape[1]
        assert Q.shape == W.shape
        self.W, self.H, self.grid = W, H, grid
        self.Q = Q.copy()
        self.err = row_errors(W, self.Q, H)
        self.cand = {+1: grid.quantize_up(self.Q), -1: grid.quantize_down(self.Q)}
        self.gain = {s: flip_gains(W, self.Q, H, self.cand[s]) for s in (+1, -1)}

    def _refresh(self, sign, rows, cols, q_old, c_old):
        """Incremental gain update after Q[rows, cols] changed (obq.py:299-336)."""
        gains, cand = self.gain[sign], self.cand[sign]
        k = np.arange(len(rows))
        H = self.H
        Wr = self.W[rows].copy()
        Q_new = self.Q[rows].copy()
        Q_old = Q_new.copy()
        Q_old[k, cols] = q_old
        C_new = cand[rows].copy()
        D_new = C_new - Q_new
        Hr = H[cols].copy()

        c_new, q_new = C_new[k, cols], Q_new[k, cols]
        d_old, d_new = c_old - q_old, c_new - q_new
        hd = H.diagonal()[cols]

        gains[rows, cols] += hd * (np.square(d_old) - np.square(d_new))
        gains[rows, cols] += 2 * ((Q_old - Wr) * Hr).sum(axis=-1) * (d_old - d_new)
        gains[rows] += 2 * np.expand_dims(q_old - q_new, 1) * Hr * D_new

    def _apply(self, sign, mask):
        """Move the best weight of every selected row one step (obq.py:264-297)."""
        gains, cand = self.gain[sign], self.cand[sign]
        rows = np.arange(self.W.shape[0])[mask]
        delta = gains.max(axis=1)[mask]
        cols = gains.argmax(axis=1)[mask]
        new = cand[rows, cols]
        q_old = self.Q[rows, cols].copy()
        self.Q[rows, cols] = new
        up_old = self.cand[+1][rows, cols].copy()
        self.cand[+1][rows, cols] = self.grid.quantize_up(new)
        down_old = self.cand[-1][rows, cols].copy()
        self.cand[-1][rows, cols] = self.grid.quantize_down(new)
        self.err[rows] -= delta
        self._refresh(+1, rows, cols, q_old, up_old)
        self._refresh(-1, rows, cols, q_old, down_old)

    def move(self):
        """One move per row: the better of best-up / best-down if it helps (obq.py:338-346)."""
        best_up = self.gain[+1].max(axis=1)
        best_down = self.gain[-1].max(axis=1)
        go_up = (best_up > best_down) & (best_up > 0)
        go_down = ~go_up & (best_down > 0)
        self._apply(+1, go_up)
        self._apply(-1, go_down)


def local_search(W, Q, H, grid, moves, records=None):
    """obq.py:349-358: returns the input object itself when moves == 0.

    records (a list, optional): receives one `move_record` per move -- how close every row's decision was
    (see near_tie_summary); parity tests use it to tell a near-tie that fell the other way from a wrong move."""
    if moves == 0:
        return Q
    state = _SearchState(W, Q, H, grid)
    noise = gain_noise_scale(W, Q, H) if records is not None else None
    for _ in range(moves):
        if records is not None:
            records.append(move_record(state.gain[+1], state.gain[-1], state.cand[+1] - state.Q, state.cand[-1] - state.Q, noise))
        state.move()
    return state.Q


# --------------------------------------------------------------------------
# How close a move was: evidence for local-search parity (not part of the reference)
# --------------------------------------------------------------------------
def gain_noise_scale(W, Q0, H):
    """A[r, j] = sum_k |W - Q0|[r, k] |H|[k, j]: the magnitude that bounds the rounding error of the initial
    `delta @ H` (obq.py:231) -- two float32 dot products of length n over the same terms, summed in different
    orders, differ by at most about 2 n 2^-24 A.  Every later gain update is computed from identical inputs by
    identical arithmetic on both sides, so this initial difference is what separates two implementations."""
    return np.abs(W - Q0).astype(np.float32) @ np.abs(H).astype(np.float32)


def move_record(gain_up, gain_down, D_up, D_down, noise):
    """Per row, for the move about to be taken (obq.py:338-346): the choice, the runner-up and their distance.

    choice / runner : 2 * column + (1 = up, 0 = down), or -1 for "stay"
    ratio           : (value of the choice - value of the runner-up) / (2^-24 * (noise of both)), where the
                      noise of a candidate is 2 |step| A[r, column] (0 for a plain "stay"): the margin in units
                      of one float32 rounding of the terms that make up the two gains.  A ratio below ~n means
                      the decision is within the rounding error of the initial GEMM -- a near-tie.
    "Stay" covers the reference's "no gain is positive" AND its moves onto the value a weight already has
    (the top level's up-candidate is the top level; a rounding residue can make such a candidate's gain positive):
    such a move changes neither Q nor any gain (every update term of obq.py:322-334 is multiplied by a zero
    difference), so the row is in the same state at the next move and takes the same non-move again.
    The alternatives are every real up move, every real down move and "stay" (value: the largest residue of
    the non-moves, at least 0); the runner-up is the best of the others.
    """
    R, n = gain_up.shape
    rows = np.arange(R)
    both = np.concatenate([gain_down, gain_up], axis=1).astype(np.float64)  # code of entry e: 2 * (e % n) + (e // n)
    real = np.concatenate([D_down, D_up], axis=1) != 0
    cu, cd = gain_up.max(axis=1), gain_down.max(axis=1)
    ju, jd = gain_up.argmax(axis=1), gain_down.argmax(axis=1)
    go_up = (cu > cd) & (cu > 0)
    go_down = ~go_up & (cd > 0)
    entry = np.where(go_up, n + ju, jd)                       # the reference's pick among the 2 n entries
    moves = (go_up | go_down) & real[rows, entry]
    # "stay": its value is the best residue among the non-moves (or 0), its noise that candidate's
    idle = np.where(real, -np.inf, both)
    idle_entry = idle.argmax(axis=1)
    idle_value = np.maximum(idle[rows, idle_entry], 0.0)
    idle_code = np.where(idle[rows, idle_entry] > 0, 2 * (idle_entry % n) + idle_entry // n, -1)
    # best real move other than the chosen one
    others = np.where(real, both, -np.inf)
    others[rows[moves], entry[moves]] = -np.inf
    other_entry = others.argmax(axis=1)
    other_value = others[rows, other_entry]
    other_code = 2 * (other_entry % n) + other_entry // n

    choice = np.where(moves, 2 * (entry % n) + entry // n, -1)
    value = np.where(moves, both[rows, entry], idle_value)
    stay_is_runner = moves & (idle_value >= other_value)
    runner = np.where(stay_is_runner, -1, other_code)
    runner_value = np.where(stay_is_runner, idle_value, other_value)

    def cand_noise(code):
        # the larger of the column's two steps: a gain carried across a move INTO the end of the codebook keeps the
        # rounding residue of the step it was built with although its own step is then 0
        col = np.maximum(code, 0) // 2
        d = np.maximum(np.abs(D_up[rows, col]), np.abs(D_down[rows, col]))
        return np.where(code >= 0, 2.0 * d.astype(np.float64) * noise[rows, col], 0.0)

    stay_noise = cand_noise(idle_code)
    scale = np.where(moves, cand_noise(choice), stay_noise) + np.where(stay_is_runner, stay_noise, cand_noise(other_code))
    scale = scale * 2.0 ** -24
    margin = value - runner_value
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = np.where(scale > 0, margin / scale, np.where(margin > 0, np.inf, 0.0))
    ratio = np.where(np.isneginf(runner_value), np.inf, ratio)  # nothing else to do at all
    return dict(choice=choice.astype(np.int32), runner=runner.astype(np.int32), ratio=ratio.astype(np.float32))


def near_tie_summary(records, limit):
    """Rows whose decision came within `limit` (a ratio, see move_record) at some move, with their full records:
    dict(rows int32[K], choice int32[K, moves], runner int32[K, moves], ratio float32[K, moves])."""
    choice = np.stack([r["choice"] for r in records], axis=1)
    runner = np.stack([r["runner"] for r in records], axis=1)
    ratio = np.stack([r["ratio"] for r in records], axis=1)
    rows = np.flatnonzero((ratio < limit).any(axis=1)).astype(np.int32)
    return dict(rows=rows, choice=choice[rows], runner=runner[rows], ratio=ratio[rows])
