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


@pytest.fixture(scope="session")
def ls_traces():
    return np.load(os.path.join(GOLDEN, "ls_traces.npz"))


@pytest.fixture(scope="session")
def orders():
    return np.load(os.path.join(GOLDEN, "orders.npz"))


@pytest.fixture(scope="session")
def experiments():
    with open(os.path.join(GOLDEN, "experiments.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def codebook_fit():
    return np.load(os.path.join(GOLDEN, "codebook_fit.npz"))


def parse_case(name):
    """'r64_n96_s2001_N8_diag_ls10[_strip_d0.03]' -> dict."""
    p = name.split("_")
    d = dict(R=int(p[0][1:]), n=int(p[1][1:]), seed=int(p[2][1:]), levels=int(p[3][1:]), order=p[4],
             moves=int(p[5][2:]), strip=False, damp=0.01)
    if len(p) > 6:
        d["strip"] = True
        d["damp"] = float(p[7][1:])
    return d


class _Options:
    """setenv / delenv look-alikes over slk_set_option: the library reads SLK_* environment switches once, at its
    first use, so tests flip them through the C ABI; everything touched is restored afterwards."""

    def __init__(self):
        from sleekit_amd import _lib

        self._lib, self._old = _lib, {}

    def setenv(self, name, value):
        if name not in self._old:
            self._old[name] = self._lib.lib.slk_get_option(name.encode())
        self._lib.set_option(name, int(value))

    def delenv(self, name, raising=True):
        self.setenv(name, 0)

    def restore(self):
        for name, value in self._old.items():
            self._lib.set_option(name, value)


@pytest.fixture
def slkopt():
    o = _Options()
    yield o
    o.restore()
