"""CPU oracle for the GPTQ/OBQ layer-quantization hot path.

TEST INFRASTRUCTURE ONLY.  This package is a NumPy restatement of the
algorithm in the reference (Coloquinte/sleekit, `sleekit/obq.py`,
`sleekit/codebook.py`, `sleekit/scaling.py`, `sleekit/statistics.py`).
It is imported only by `tests/`, by `__graft_entry__.smoke()` and by the
`cpu_baseline` leg of `bench.py` -- never by the product package
`sleekit_amd`, which has no CPU path and fails loudly without its HIP library.

Parity status: PINNED.  `tests/golden/make_golden.py` imported the real
reference in the build container (NumPy 2.2.6 / OpenBLAS 0.3.29, NEP-50
promotion rules) and wrote the fixtures under `tests/golden/`;
`tests/test_oracle_golden.py` checks every function here against them
bit-for-bit (indices, orders) or to 1e-12 (float64 factors).
"""

from . import grid, obq_ref, scaling_ref, stats_ref, npsum  # noqa: F401
