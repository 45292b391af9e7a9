"""sleekit.obq (reference: sleekit/obq.py) -> sleekit_amd.obq; see sleekit/__init__.py here."""

import numpy as np  # noqa: F401  (the experiments rely on the star import leaking it)

import sleekit_amd.obq as _impl
from sleekit_amd.obq import *  # noqa: F401,F403

# private names the reference's modules import from one another (sleekit/scaling.py:3-8) travel too
globals().update({k: v for k, v in vars(_impl).items() if k.startswith("_quantize_opt") or k.startswith("_compute_")})
