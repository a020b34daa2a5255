import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "one_arith: the test does not depend on the arithmetic set; run it once")


ARITH_SETS = {"opencv": 0, "legacy": 1}   # uwt_params.arith / uwo_params.arith


def pytest_generate_tests(metafunc):
    """Every test runs under both arithmetic sets (the default: OpenCV's generic paths; legacy: rounds 1-3), unless its module
    sets ARITH_INDEPENDENT = True or it carries @pytest.mark.one_arith."""
    if "arith" in metafunc.fixturenames:
        once = getattr(metafunc.module, "ARITH_INDEPENDENT", False) or metafunc.definition.get_closest_marker("one_arith")
        metafunc.parametrize("arith", ["opencv"] if once else ["opencv", "legacy"], indirect=True)


@pytest.fixture(autouse=True)
def arith(request, monkeypatch):
    """The arithmetic set of this test: the default both Python bindings put into fresh parameter blocks (oracle.oracle and
    uw-slam_amd.capi default_params) and the oracle's per-stage switch.  Neither library reads the environment: child processes
    of tests are told the set explicitly (--arith / an argument), and the Python snippets tests run as children pick it up from
    UWT_TEST_ARITH through child_arith_header() below."""
    name = request.param
    from oracle import oracle
    capi = importlib.import_module("uw-slam_amd.capi")
    monkeypatch.setattr(oracle, "DEFAULT_ARITH", ARITH_SETS[name])
    monkeypatch.setattr(capi, "DEFAULT_ARITH", ARITH_SETS[name])
    monkeypatch.setenv("UWT_TEST_ARITH", name)
    if os.path.exists(oracle._LIB_PATH):
        prev = oracle.set_arith(ARITH_SETS[name])
        yield name
        oracle.set_arith(prev)
    else:
        yield name


# first lines of a Python snippet a test runs as a child process: both bindings' default set = the parent test's
CHILD_ARITH_HEADER = r'''
import importlib as _il, os as _os
_ar = {"opencv": 0, "legacy": 1}[_os.environ.get("UWT_TEST_ARITH", "opencv")]
_il.import_module("uw-slam_amd.capi").DEFAULT_ARITH = _ar
from oracle import oracle as _O
_O.DEFAULT_ARITH = _ar
_O.set_arith(_ar)
'''


class GoldenView:
    """A tests/golden/*.npz under one arithmetic set: g["pose"] reads "legacy_pose" when the legacy set is active and the file
    has it, the plain key otherwise (arithmetic-independent vectors exist once)."""

    def __init__(self, path, arith):
        self._g = np.load(path)
        self._prefix = "legacy_" if arith == "legacy" else ""
        self.files = [k for k in self._g.files if not k.startswith("legacy_")]

    def __getitem__(self, k):
        if self._prefix and (self._prefix + k) in self._g.files:
            return self._g[self._prefix + k]
        return self._g[k]

    def __contains__(self, k):
        return k in self._g.files


@pytest.fixture
def golden(arith):
    """golden("pair_160x96_fixed") -> GoldenView of tests/golden/pair_160x96_fixed.npz under the test's arithmetic set"""
    def load(name):
        return GoldenView(os.path.join(ROOT, "tests", "golden", name if name.endswith(".npz") else name + ".npz"), arith)
    return load


def pytest_collection_modifyitems(config, items):
    """A stalled test fails after 10 minutes instead of holding the run (pytest-timeout, when it is installed)."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(600))


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("uw-slam_amd.synth")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
