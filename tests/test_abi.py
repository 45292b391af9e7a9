"""CPU checks of the drop-in boundary: the library loads and exports what the header declares."""

import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "sleekit_amd.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(slk_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from sleekit_amd import _lib

    names = declared_symbols()
    assert len(names) >= 18
    for name in names:
        assert hasattr(_lib.lib, name), f"{name} declared in the header but not exported"
    assert sorted(_lib.PROTOTYPES) == names, "ctypes prototypes and header out of step"
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r" T (slk_[a-z0-9_]+)", out)))
    assert exported == names


def test_abi_version_and_sizes():
    from sleekit_amd import _lib

    assert _lib.lib.slk_abi_version() == 8
    assert _lib.lib.slk_factor_ld(1) == 64 and _lib.lib.slk_factor_ld(64) == 64 and _lib.lib.slk_factor_ld(11008) == 11008
    assert _lib.lib.slk_factor_ld(1100) == 1152
    # 4096 x 4096: two float64 n x n scratch matrices dominate
    assert _lib.lib.slk_workspace_bytes(4096, 4096) >= 2 * 4096 * 4096 * 8
    assert _lib.lib.slk_workspace_bytes(512, 11008) >= 2 * 11008 * 11008 * 8


def test_options_are_read_once_and_set_through_the_abi():
    """SLK_* environment switches are read at first use only; afterwards slk_set_option is the way."""
    from sleekit_amd import _lib

    assert _lib.lib.slk_get_option(b"no_window2") in (0, 1)
    old = _lib.lib.slk_get_option(b"NO_WINDOW2")
    os.environ["SLK_NO_WINDOW2"] = "1" if not old else "0"   # too late: not looked at again
    try:
        assert _lib.lib.slk_get_option(b"SLK_NO_WINDOW2") == old
        with _lib.option("no_window2", 5):
            assert _lib.lib.slk_get_option(b"no_window2") == 5
        assert _lib.lib.slk_get_option(b"no_window2") == old
    finally:
        del os.environ["SLK_NO_WINDOW2"]
    assert _lib.lib.slk_set_option(b"no_such_switch", 1) == _lib.E_ARG
    assert b"unknown option" in _lib.lib.slk_last_error()


def test_argument_errors_do_not_touch_the_gpu():
    """Bad arguments are rejected on the host before any launch (safe without a GPU)."""
    from sleekit_amd import _lib

    assert _lib.lib.slk_codebook_apply(None, 4, 1, -1.0, 1.0, None, 0, None, None) == _lib.E_ARG
    assert b"levels" in _lib.lib.slk_last_error()
    assert _lib.lib.slk_gptq_quantize(None, None, None, None, 4, 4, 8, -1.0, 1.0, None, 32, 8, 0, None, None, None, None, 0, None) == _lib.E_ARG
    assert _lib.lib.slk_hessian_prepare(None, 0, 0.01, 1, None, None, None, None, 0, None) == _lib.E_ARG
    # entry points added with ABI version 4
    import ctypes

    one = (ctypes.c_void_p * 1)(8)
    assert _lib.lib.slk_hessian_prepare_batch(one, 1, 64, 0.01, _lib.ORDER_ERR, 8, 8, None, 0, None) == _lib.E_ARG
    assert b"none and diag" in _lib.lib.slk_last_error()
    assert _lib.lib.slk_hessian_prepare_batch(one, 65, 64, 0.01, _lib.ORDER_DIAG, 8, 8, None, 0, None) == _lib.E_ARG
    assert _lib.lib.slk_chol_inverse_upper_batch(8, 0, 64, 8, 8, None, 0, None) == _lib.E_ARG
    assert _lib.lib.slk_factor_workspace_bytes_batch(8, 768) >= 8 * 2 * 768 * 768 * 8
    assert _lib.lib.slk_local_search(8, 8, 8, 4, 4, 8, -1.0, 1.0, None, 1, None, None, None, 2, None, None, 0, None) == _lib.E_ARG
    assert b"gains" in _lib.lib.slk_last_error()
    # ... with ABI version 7
    assert _lib.lib.slk_stack_rows(one, 1, 8, 4, 16, 0.0, 8, None) == _lib.E_ARG  # fewer padded rows than rows
    assert _lib.lib.slk_stack_rows(None, 0, 8, 8, 16, 0.0, None, None) == _lib.OK  # an empty batch is nothing to do
    assert _lib.lib.slk_set_option(b"panel_split", 0) == _lib.OK and _lib.lib.slk_get_option(b"panel_split") == 0
    # ... with ABI version 8 (no new entry point: the chain is the factorisation's default form; its flags live in the workspace)
    for name in (b"tall_error", b"rows_below_wide"):
        assert _lib.lib.slk_set_option(name, 0) == _lib.OK and _lib.lib.slk_get_option(name) == 0
    assert _lib.lib.slk_set_option(b"no_such_option", 1) == _lib.E_ARG
    per_matrix = _lib.lib.slk_factor_workspace_bytes_batch(2, 4096) - _lib.lib.slk_factor_workspace_bytes_batch(1, 4096)
    assert per_matrix >= 2 * 4096 * 4096 * 8 + 2 * 64 * 4  # X and S, and the chain's 2 x 64 flags
    from sleekit_amd import _device as sdev

    with pytest.raises(RuntimeError, match="hand-off timed out"):
        sdev.raise_not_pd(sdev.HANDOFF_TIMEOUT, "compute_hessian_chol")
    with pytest.raises(np.linalg.LinAlgError, match="pivot 41"):
        sdev.raise_not_pd(42, "compute_hessian_chol")
    with pytest.raises(RuntimeError):
        _lib.check(_lib.E_ARG)


def test_product_has_no_cpu_path():
    """Without a GPU the product raises instead of computing on the host; it never imports the oracle."""
    import sys

    import numpy as np
    import torch

    from sleekit_amd import codebook, obq, scaling

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cb = codebook.UniformCodebook(8, -1, 1)
    W = np.zeros((4, 8), np.float32)
    H = np.eye(8, dtype=np.float32)
    for call in (
        lambda: scaling.compute_min_mse_scaling(W, cb),
        lambda: cb(W),
        lambda: obq.quantize_opt(W, H, cb),
        lambda: scaling.quantize_with_scaling(W, np.ones(4, np.float32), cb, H),
        lambda: obq.quantization_error(W, W, H),
    ):
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            call()
    for mod in ("sleekit_amd.obq", "sleekit_amd.engine", "sleekit_amd.scaling", "sleekit_amd.codebook"):
        src = open(sys.modules[mod].__file__).read()
        assert "import oracle" not in src and "from oracle" not in src


def test_interface_mirrors_reference_names():
    from sleekit_amd import codebook, obq, scaling
    import sleekit_amd

    for name in ("remove_input_bias", "remove_dead_values", "compute_hessian_chol", "compute_hessian_order",
                 "channelwise_error", "quantization_error", "_quantize_opt_core", "_quantize_opt_block",
                 "quantize_opt", "compute_gain", "LocalSearchQuantizer", "quantize_local_search", "random_psd_matrix", "np"):
        assert hasattr(obq, name), name
    for name in ("apply_scaling", "apply_scaling_in_place", "compute_norm_scaling", "compute_non_saturating_scaling",
                 "quantize_with_scaling", "compute_min_mse_scaling", "compute_obq_scaling", "compute_scaling", "np"):
        assert hasattr(scaling, name), name
    for name in ("UniformCodebook", "Codebook", "lloyd_max", "np"):
        assert hasattr(codebook, name), name
    assert sleekit_amd.Sleekit is not None
    cb = codebook.UniformCodebook(8, -1, 1)
    assert len(cb) == 8 and cb.min() == -1 and cb.max() == 1 and cb.zero == -1 and abs(cb.scale - 2 / 7) < 1e-15
    import inspect

    sig = inspect.signature(obq.quantize_opt)
    assert [p for p in sig.parameters] == ["W", "H", "quantizer", "act_order", "damp", "nb_ls_moves", "min_block_size", "num_blocks"]
    assert sig.parameters["damp"].default == 0.01 and sig.parameters["min_block_size"].default == 32
    sig = inspect.signature(scaling.quantize_with_scaling)
    assert [p for p in sig.parameters] == ["data", "scale", "quantizer", "H", "act_order", "damp", "nb_ls_moves"]


def test_bench_workloads_are_the_baseline_configs():
    """bench.py --config cfgN builds the layer streams SURVEY.md 8 names (counts from results/compare_3b.csv: 72 / 144 / 96
    layers) with each config's codebook size, moves and Hessian correction."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    W = bench.WORKLOADS
    assert [len(W[c]["block"]) * W[c]["blocks"] for c in ("cfg2", "cfg3", "cfg4", "cfg5")] == [72, 144, 96, 32]
    assert (W["cfg2"]["levels"], W["cfg3"]["levels"], W["cfg4"]["levels"], W["cfg5"]["levels"]) == (8, 3, 8, 4)
    assert W["cfg3"]["strip"] and W["cfg4"]["moves"] == 10 and not W["cfg2"]["moves"]
    weights = {c: sum(r * n for r, n in W[c]["block"]) * W[c]["blocks"] for c in W}
    assert weights["cfg2"] == 84934656 and weights["cfg3"] == weights["cfg4"] == 301989888 and weights["cfg5"] == 32 * 4096 * 11008
