"""`sleekit` by name: the reference's package name over the MI355X engine.

Put this directory's parent in front of the path and the reference's experiments run as they are:

    PYTHONPATH=/path/to/repo/dropin:/path/to/repo  python experiments/compare.py data/ --codebook-size 8

`from sleekit.codebook import *` / `sleekit.obq` / `sleekit.scaling` (experiments/compare.py:1-3 and every other
experiment script) and `from sleekit import Sleekit` (sleekit/__init__.py:1-4) resolve to sleekit_amd's modules, which
mirror the reference's names, arguments, defaults and exceptions for the hot path.  The experiments use
`np` without importing it: it leaks out of the star imports here as it does in the reference.
"""

from sleekit_amd.statistics import Sleekit  # noqa: F401
