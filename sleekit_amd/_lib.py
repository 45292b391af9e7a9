"""ctypes binding of libsleekit_amd.so (the C ABI declared in include/sleekit_amd.h).

There is no CPU path: if the library has not been built, importing this module
raises; if no MI355X is visible, the first call that needs the device raises.
"""

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_longlong, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SLK_LIB_PATH") or os.path.join(_HERE, "libsleekit_amd.so")  # (override: A/B builds in tools/)

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C sleekit_amd/csrc` "
        "(or `python -c 'import __graft_entry__ as g; g.build()'`). sleekit_amd has no CPU fallback."
    )

lib = ctypes.CDLL(LIB_PATH)

OK, E_ARG, E_NOT_PD, E_HIP, E_WS = 0, -1, -2, -3, -4
CB_VALUE, CB_INDEX, CB_UP, CB_DOWN, CB_INDEX16, CB_INDEX32 = 0, 1, 2, 3, 4, 5
ORDER_NONE, ORDER_DIAG, ORDER_ERR, ORDER_SQERR, ORDER_KEYS = 0, 1, 2, 3, 4
ORDER_MODES = {"none": ORDER_NONE, "diag": ORDER_DIAG, "err": ORDER_ERR, "sqerr": ORDER_SQERR}

P = c_void_p  # device pointers travel as plain addresses

# name -> (restype, argtypes); every symbol of include/sleekit_amd.h is listed here and
# tests/test_abi.py checks the two stay in step.
PROTOTYPES = {
    "slk_abi_version": (c_int, []),
    "slk_last_error": (c_char_p, []),
    "slk_set_option": (c_int, [c_char_p, c_int]),
    "slk_get_option": (c_int, [c_char_p]),
    "slk_workspace_bytes": (c_size_t, [c_int, c_int]),
    "slk_codebook_apply": (c_int, [P, c_size_t, c_int, c_double, c_double, P, c_int, P, P]),
    "slk_rows_divide": (c_int, [P, P, c_int, c_int, c_int, P, P]),
    "slk_stack_rows": (c_int, [P, c_int, c_int, c_int, c_int, c_float, P, P]),
    "slk_hessian_strip_mean": (c_int, [P, P, c_int, P, P]),
    "slk_hessian_patch_dead": (c_int, [P, P, c_int, c_int, P, c_size_t, P]),
    "slk_hessian_accumulate": (c_int, [P, P, P, c_int, c_int, c_longlong, P, c_size_t, P]),
    "slk_codebook_stats_workspace_bytes": (c_size_t, []),
    "slk_codebook_stats": (c_int, [P, c_size_t, c_int, c_double, c_double, P, c_int, P, P, P, P, c_size_t, P]),
    "slk_sort_workspace_bytes": (c_size_t, [c_size_t]),
    "slk_sort_f32": (c_int, [P, c_size_t, P, P, c_size_t, P]),
    "slk_unique_f32": (c_int, [P, c_size_t, P, P, P, c_size_t, P]),
    "slk_column_miss": (c_int, [P, c_int, c_int, c_int, c_double, c_double, P, c_int, P, P]),
    "slk_hessian_prepare": (c_int, [P, c_int, c_float, c_int, P, P, P, P, c_size_t, P]),
    "slk_inverse_diag_keys": (c_int, [P, P, c_int, c_float, c_int, P, P, c_size_t, P]),
    "slk_pivot_keys": (c_int, [P, c_int, c_float, P, P, c_size_t, P]),
    "slk_factor_ld": (c_int, [c_int]),
    "slk_factor_load": (c_int, [P, c_int, P, P]),
    "slk_chol_inverse_upper": (c_int, [P, c_int, P, P, P, c_size_t, P]),
    "slk_chol_inverse_upper_lookahead": (c_int, [P, c_int, P, P, P, c_size_t, P]),
    "slk_release_helpers": (c_int, []),
    "slk_hessian_prepare_batch": (c_int, [P, c_int, c_int, c_float, c_int, P, P, P, c_size_t, P]),
    "slk_chol_inverse_upper_batch": (c_int, [P, c_int, c_int, P, P, P, c_size_t, P]),
    "slk_factor_workspace_bytes_batch": (c_size_t, [c_int, c_int]),
    "slk_factor_payload_words": (c_size_t, [c_int]),
    "slk_factor_pack": (c_int, [P, P, P, c_int, P, P]),
    "slk_factor_unpack": (c_int, [P, c_int, P, P, P, P]),
    "slk_factor_unpack_upper": (c_int, [P, c_int, P, P, P, P]),
    "slk_factor_unpack_upper_batch": (c_int, [P, c_int, c_int, P, P, P, P, P]),
    "slk_gptq_quantize": (
        c_int,
        [P, P, P, P, c_int, c_int, c_int, c_double, c_double, P, c_int, c_int, c_int, P, P, P, P, c_size_t, P],
    ),
    "slk_gptq_quantize_batch": (
        c_int,
        [P, P, P, P, c_int, c_int, c_int, c_int, c_double, c_double, P, c_int, c_int, c_int, P, P, P, P, c_size_t, P],
    ),
    "slk_workspace_bytes_batch": (c_size_t, [c_int, c_int, c_int]),
    "slk_row_errors": (c_int, [P, P, P, c_int, c_int, P, P, P, c_size_t, P]),
    "slk_row_errors_batch": (c_int, [P, P, P, c_int, c_int, c_int, P, P, P, c_size_t, P]),
    "slk_symmetry_flag": (c_int, [P, c_int, P, P]),
    "slk_local_search": (c_int, [P, P, P, c_int, c_int, c_int, c_double, c_double, P, c_int, P, P, P, c_int, P, P, c_size_t, P]),
    "slk_local_search_batch": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_double, c_double, P, c_int, P, P, P, P, c_size_t, P]),
    "slk_scale_minmax": (c_int, [P, c_int, c_int, c_double, c_double, P, P]),
    "slk_scale_norm": (c_int, [P, c_int, c_int, P, P]),
    "slk_scale_search": (c_int, [P, P, P, c_int, P, c_int, c_int, c_int, c_double, c_double, P, P, P]),
    "slk_search_step": (c_int, [P, c_float, c_int, P, P, c_int, P]),
    "slk_scale_times": (c_int, [P, P, c_float, c_int, P, P]),
    "slk_diag_mean": (c_int, [P, c_int, P, P, c_size_t, P]),
    "slk_probe_mfma_f64": (c_int, [P, c_int, c_int, P]),
    "slk_probe_mfma_f32": (c_int, [P, c_int, c_int, P]),
    "slk_probe_mfma_f64_acc": (c_int, [P, c_int, c_int, c_int, P]),
    "slk_probe_chain": (c_int, [P, c_int, c_int, P]),
    "slk_probe_window_cycles": (c_int, [P, c_int]),
    "slk_probe_panel_cycles": (c_int, [P, c_int]),
    "slk_probe_leaf_chain": (c_int, [P, c_int, c_int, P]),
    "slk_profile_enable": (c_int, [c_int]),
    "slk_profile_reset": (c_int, []),
    "slk_profile_report": (c_int, [c_char_p, c_size_t]),
}


def profile_report():
    """Parsed slk_profile_report(): list of dicts per kernel name."""
    import json

    need = lib.slk_profile_report(None, 0)
    buf = ctypes.create_string_buffer(need + 1)
    lib.slk_profile_report(buf, need + 1)
    return json.loads(buf.value.decode())

for _name, (_res, _args) in PROTOTYPES.items():
    _fn = getattr(lib, _name)  # AttributeError here = header and library out of step
    _fn.restype = _res
    _fn.argtypes = _args


def set_option(name, value):
    """slk_set_option: run-time switch between equivalent code paths (see include/sleekit_amd.h)."""
    check(lib.slk_set_option(name.encode(), int(value)))


class option:
    """`with option("no_window2", 1): ...` -- set for the block, restored afterwards."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = lib.slk_get_option(self.name.encode())
        set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.old)
        return False


class SleekitAmdError(RuntimeError):
    pass


def check(rc):
    """Turn a negative return code into an exception carrying slk_last_error()."""
    if rc == OK:
        return
    msg = (lib.slk_last_error() or b"").decode("utf-8", "replace")
    if rc == E_ARG:
        raise RuntimeError(msg or "invalid argument")
    raise SleekitAmdError(f"libsleekit_amd error {rc}: {msg}")
