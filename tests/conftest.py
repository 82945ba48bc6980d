import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
PKG = os.path.join(ROOT, "dba-fusion_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionstart(session):
    """the parity report (tests/util.py::check_state) starts empty in the outermost pytest session; sessions spawned by
    tests (other solver kernels) append to it"""
    if not os.environ.get("DBA_PARITY_SESSION"):
        os.environ["DBA_PARITY_SESSION"] = "1"
        rep = os.path.join(ROOT, "gpurun_out", "parity_report.jsonl")
        if os.path.exists(rep):
            os.remove(rep)


@pytest.fixture(params=["auto", "resident", "rowtile"])
def lookup_kernel(request):
    """runs a test under each form of the sheared lookup kernel (csrc/corr_sheared.hip); results must not differ"""
    from dbaf_amd import _lib
    lib = _lib.load()
    assert lib.dba_corr_lookup_select({"auto": 0, "resident": 2, "rowtile": 5}[request.param]) == 0
    yield request.param
    lib.dba_corr_lookup_select(0)
