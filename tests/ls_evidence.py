"""Local-search parity, proven row by row (used by the GPU parity and fuzz tests).

The reference's initial gains come out of a BLAS product `delta @ H` (obq.py:231) whose summation order is the
BLAS's own; every later step is reproduced bit for bit on the device (moves in the reference's order, the
interaction sum in NumPy's pairwise order).  A row may therefore end differently only if, at some move, the
reference's best candidate led its runner-up by less than the rounding of that initial product.  This module
checks exactly that, per row, from the reference's recorded moves (tests/golden/ls_traces.npz, or the oracle's
records when the oracle is the comparison): the device's sequence of moves must follow the reference's up to a
move whose margin was below LS_NEAR_TIE roundings (oracle/obq_ref.py: move_record), and there it must have taken
the reference's runner-up.  Anything else is a failure, however few rows it concerns.
"""

import numpy as np

# Margin, in units of one float32 rounding of the magnitude of the terms behind the two gains, below which the order
# of two candidates is not determined by the inputs (two GEMMs that differ only in summation order can disagree by
# ~2 n such roundings in the worst case and by a fraction of ONE typically).  Observed on the BASELINE-sized cases:
# every departure of the device from the reference's moves happened at a ratio <= 0.52.
LS_NEAR_TIE = 2.0


def explain_rows(bad_rows, device_trace, near):
    """near: dict(rows, choice, runner, ratio) for the rows that came near a tie; device_trace: (R, moves) int32.
    Asserts that every row of `bad_rows` is a proven near-tie; returns the largest ratio at a departure."""
    where = {int(r): i for i, r in enumerate(near["rows"])}
    worst = 0.0
    for r in bad_rows:
        r = int(r)
        assert r in where, f"row {r} differs from the reference, yet none of its decisions was close"
        i = where[r]
        ref, runner, ratio = near["choice"][i], near["runner"][i], near["ratio"][i]
        departs = np.flatnonzero(ref != device_trace[r])
        assert len(departs), f"row {r}: same moves as the reference but different indices"
        m = int(departs[0])
        assert device_trace[r][m] == runner[m], f"row {r}, move {m}: device took {device_trace[r][m]}, reference {ref[m]}, runner-up {runner[m]}"
        assert ratio[m] <= LS_NEAR_TIE, f"row {r}, move {m}: the reference led by {ratio[m]:.3g} roundings -- not a near-tie"
        worst = max(worst, float(ratio[m]))
    return worst
