"""pytest configuration: the `gpu` marker and shared fixture loaders."""

import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def small_cases():
    return np.load(os.path.join(GOLDEN, "small_cases.npz"))


@pytest.fixture(scope="session")
def pieces():
    return np.load(os.path.join(GOLDEN, "pieces.npz"))


@pytest.fixture(scope="session")
def large_cases():
    with open(os.path.join(GOLDEN, "large_cases.json")) as f:
        return json.load(f)["cases"]


def parse_case(name):
    """'r64_n96_s2001_N8_diag_ls10[_strip_d0.03]' -> dict."""
    p = name.split("_")
    d = dict(R=int(p[0][1:]), n=int(p[1][1:]), seed=int(p[2][1:]), levels=int(p[3][1:]), order=p[4],
             moves=int(p[5][2:]), strip=False, damp=0.01)
    if len(p) > 6:
        d["strip"] = True
        d["damp"] = float(p[7][1:])
    return d
