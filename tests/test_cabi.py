"""CPU: the C-ABI library loads and exports every symbol include/dba_hip.h declares (no compute calls)."""
import ctypes
import os
import re

from dbaf_amd import _lib

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "dba_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dba_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), "missing export %s" % name
        assert name in _lib.SYMBOLS, "python binding table lacks %s" % name
    assert b"gfx950" in lib.dba_version()


def test_workspace_and_layout_are_host_side_only():
    lib = _lib.load()
    nbytes = lib.dba_ba_workspace_bytes(96, 26, 64, 64, 1, 25)
    assert 8e6 < nbytes < 64e6
    lay = _lib.BaLayout()
    assert lib.dba_ba_get_layout(96, 26, 64, 64, 1, 25, ctypes.byref(lay)) == 0
    assert lay.P == 24 and lay.Mmax == 26 and lay.nchunks in (4, 8, 16)
    assert lay.H % 256 == 0 and lay.b > lay.H
    assert lib.dba_ba_workspace_bytes(96, 26, 64, 64, 5, 2) == 0  # t1 < t0 rejected


def test_product_path_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "dba-fusion_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f
