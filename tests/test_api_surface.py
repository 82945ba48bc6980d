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
