"""Every `droid_backends.<name>` and every `lietorch` name the reference's Python uses resolves in the adapters
(SURVEY.md section 8(b)).  Reads the reference tree, so it only runs where /root/reference is mounted."""
import glob
import os
import re

import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted")


def _sources():
    files = glob.glob(os.path.join(REF, "dbaf", "**", "*.py"), recursive=True) + glob.glob(os.path.join(REF, "*.py"))
    return [(f, open(f, errors="ignore").read()) for f in files]


def test_every_droid_backends_attribute_the_reference_uses_exists():
    import droid_backends
    used = set()
    for _, text in _sources():
        used |= set(re.findall(r"\bdroid_backends\.([A-Za-z_]\w*)", text))
    assert {"ba", "corr_index_forward", "frame_distance", "depth_filter", "iproj"} <= used  # the scan sees the call sites
    missing = sorted(n for n in used if not hasattr(droid_backends, n))
    assert not missing, missing


def test_every_lietorch_name_the_reference_imports_exists():
    import lietorch
    used = set()
    for _, text in _sources():
        for names in re.findall(r"from\s+lietorch\s+import\s+([^\n]+)", text):
            used |= {n.strip() for n in names.replace("(", "").replace(")", "").split(",") if n.strip()}
    assert "SE3" in used
    missing = sorted(n for n in used if not hasattr(lietorch, n))
    assert not missing, missing
    # the SE3 methods and attributes called on pose objects in the hot path and its callers
    se3_used = set()
    for _, text in _sources():
        se3_used |= set(re.findall(r"\bSE3\.([A-Za-z_]\w*)", text))
        se3_used |= set(re.findall(r"\bSE3\([^)]*\)\.([A-Za-z_]\w*)", text))
        se3_used |= set(re.findall(r"\b(?:Gs?|Gij|Gi|Gj|poses|Ps?|dP|d)\s*(?:\[[^\]]*\])?\.(inv|retr|adjT|adj|matrix|translation|data|log|exp|scale|vec|cpu|act)\b", text))
    missing = sorted(n for n in se3_used if not hasattr(lietorch.SE3, n) and n != "data")
    assert not missing, missing


def _ref_corr_module():
    """the reference's dbaf/modules/corr.py imported with an empty stand-in for its native module"""
    import importlib.util
    import sys
    import types
    sys.dont_write_bytecode = True
    saved = sys.modules.get("droid_backends")
    sys.modules["droid_backends"] = types.ModuleType("droid_backends")
    try:
        spec = importlib.util.spec_from_file_location("_ref_modules_corr", os.path.join(REF, "dbaf", "modules", "corr.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        if saved is not None:
            sys.modules["droid_backends"] = saved
        else:
            del sys.modules["droid_backends"]
    return mod


def test_corr_mirrors_have_the_reference_signatures():
    """dbaf_amd.corr is what INTEGRATION.md swaps in for `from modules.corr import CorrBlock, AltCorrBlock`
    (covisible_graph.py:7, motion_filter.py:8): every public name and every parameter list must match"""
    import inspect
    from dbaf_amd import corr as mine
    ref = _ref_corr_module()
    for cls in ("CorrBlock", "AltCorrBlock", "CorrSampler", "CorrLayer"):
        assert hasattr(mine, cls), cls
        rc, mc = getattr(ref, cls), getattr(mine, cls)
        for name, member in inspect.getmembers(rc, predicate=inspect.isfunction):
            if name.startswith("_") and name not in ("__init__", "__call__", "__getitem__"):
                continue
            assert hasattr(mc, name), (cls, name)
            rp = list(inspect.signature(member).parameters)
            mp = list(inspect.signature(getattr(mc, name)).parameters)
            assert mp[:len(rp)] == rp, (cls, name, rp, mp)   # the mirror may add trailing keyword options only
            extra = [inspect.signature(getattr(mc, name)).parameters[k] for k in mp[len(rp):]]
            assert all(e.default is not inspect.Parameter.empty for e in extra), (cls, name, mp)


def _nparams(fn):
    """number of parameters of a Python function or of a pybind11 builtin (whose signature is the first docstring line)"""
    import inspect
    try:
        return len(inspect.signature(fn).parameters)
    except ValueError:
        head = fn.__doc__.splitlines()[0]
        inner = head[head.index("(") + 1:head.rindex(")")]
        return len([a for a in inner.split(",") if a.strip()]) if inner.strip() else 0


def test_droid_backends_signatures_match_the_bindings():
    """src/droid.cpp: every m.def / class method bound there exists here with the same number of parameters -- in the
    package AND in the compiled adapter (csrc_ext/droid_backends_ext.cpp), which must be a drop-in by itself"""
    import inspect
    import droid_backends
    text = open(os.path.join(REF, "src", "droid.cpp")).read()
    bound = re.findall(r'm\.def\("(\w+)",\s*&(\w+)', text)
    assert len(bound) >= 10
    for pyname, cname in bound:
        m = re.search(r"\b%s\s*\(([^)]*)\)\s*\{" % cname, text)
        assert m, cname
        nargs = len([a for a in m.group(1).split(",") if a.strip()])
        fn = getattr(droid_backends, pyname)
        assert _nparams(fn) == nargs, (pyname, nargs)
        if droid_backends.compiled is not None and pyname != "ba_extend":   # (ba_extend: debug variant, Python only)
            assert _nparams(getattr(droid_backends.compiled, pyname)) == nargs, ("compiled", pyname, nargs)
    methods = re.findall(r'\.def\("(\w+)",\s*&BACore::(\w+)\)', text)
    assert {m for m, _ in methods} >= {"init", "hessian", "optimize", "retract"}
    hdr = open(os.path.join(REF, "src", "bacore.h")).read()
    for pyname, cname in methods:
        m = re.search(r"\b%s\s*\(([^)]*)\)\s*;" % cname, hdr)
        assert m, cname
        nargs = len([a for a in m.group(1).split(",") if a.strip()])
        fn = getattr(droid_backends.BACore, pyname)
        assert len(inspect.signature(fn).parameters) == nargs + 1, (pyname, nargs)  # + self
        if droid_backends.compiled is not None:
            assert _nparams(getattr(droid_backends.compiled.BACore, pyname)) == nargs + 1, ("compiled", pyname, nargs)
