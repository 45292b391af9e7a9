"""sleekit_amd: MI355X-native GPTQ/OBQ layer quantization behind the sleekit function surface.

    from sleekit_amd import obq, scaling, codebook, Sleekit

mirrors `sleekit.obq`, `sleekit.scaling`, `sleekit.codebook` and `sleekit.Sleekit` of
Coloquinte/sleekit for the hot path.  Submodules are imported lazily
so that `sleekit_amd.synth` (pure NumPy test-data generation) stays importable before the
HIP library is built; everything else needs libsleekit_amd.so and fails loudly without it.
"""

import importlib
import os as _os

# Streams only overlap when they sit in different hardware queues, and HIP gives a process 4 of them by default
# ("Hardware queues"): the pipelines of sleekit_amd.dist and bench.py want a queue per stream.  Read by the
# HIP runtime when it initialises, so this has no effect once the process has touched the GPU.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_SUBMODULES = ("codebook", "obq", "scaling", "statistics", "engine", "dist", "synth", "_lib", "_device")


def __getattr__(name):
    if name in _SUBMODULES:
        return importlib.import_module(f"{__name__}.{name}")
    if name == "Sleekit":
        return importlib.import_module(f"{__name__}.statistics").Sleekit
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
