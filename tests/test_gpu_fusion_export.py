"""The device side of the BACore -> factor-graph hand-off (round 6): the export kernel that ends BACore.hessian writes the
reduced system into pinned host memory itself -- plain (src/droid_kernels.cu:1889-1897) or already in GTSAM's tangent
coordinates, the augmented [Hg | vg] matrix of the fork's gtsam.BA2GTSAM (/root/reference/dbaf/depth_video.py:20-29, :397-401).
Held against the vectors the reference's own numpy BA2GTSAM produced (tests/golden/ba2gtsam.npz, make_golden.gen_ba2gtsam) and
against the host-side restatement (dbaf_amd/fusion.py) on a real window."""
import ctypes
import os

import numpy as np
import pytest
import torch

from dbaf_amd import _lib, fusion
from dbaf_amd import synthetic as syn
from util import to_dev

pytestmark = pytest.mark.gpu


def _export(H, v, layout, A=None, stab=0.0, poison=True):
    """H (symmetric, numpy), v -> the workspace's slots (upper triangle poisoned: only the lower one may be read) -> export"""
    lib = _lib.load()
    n = H.shape[0]
    P = n // 6
    dims = (1, P + 2, 8, 8, 1, 1 + P)
    nbytes = lib.dba_ba_workspace_bytes(*dims)
    ws = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    lay = _lib.BaLayout()
    _lib.check(lib.dba_ba_get_layout(*dims, ctypes.byref(lay)), "dba_ba_get_layout")
    Hl = np.tril(H) + (np.triu(np.full_like(H, np.nan), 1) if poison else np.triu(H, 1))
    ws[lay.H:lay.H + 8 * n * n].view(torch.float64).copy_(torch.from_numpy(Hl.reshape(-1)).cuda())
    ws[lay.b:lay.b + 8 * n].view(torch.float64).copy_(torch.from_numpy(v).cuda())
    out = ctypes.c_void_p()
    a = None if A is None else ctypes.cast((ctypes.c_double * 36)(*A.reshape(-1)), ctypes.c_void_p)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.dba_bacore_export_host(*dims, ctypes.c_void_p(ws.data_ptr()), nbytes, stream, layout, a, float(stab),
                                          ctypes.byref(out)), "dba_bacore_export_host")
    count = n * (n + 1)
    return np.ctypeslib.as_array((ctypes.c_double * count).from_address(out.value)).copy()


def test_device_ba2gtsam_matches_the_reference_vectors():
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ba2gtsam.npz"))
    H, v, n = g["H"], g["v"], g["H"].shape[0]
    aug = _export(H, v, 1, fusion.tangent_block(g["Tbc"])).reshape(n, n + 1)
    np.testing.assert_allclose(aug[:, :n], g["Hg"], rtol=1e-12, atol=1e-12 * np.abs(g["Hg"]).max())
    np.testing.assert_allclose(aug[:, n], g["vg"], rtol=1e-12, atol=1e-12 * np.abs(g["vg"]).max())
    # the caller's stabiliser (depth_video.py:397) goes onto the first pose's diagonal BEFORE the change of coordinates
    Hs = H.copy()
    Hs[np.arange(6), np.arange(6)] += 0.00025
    ref = fusion.BA2GTSAM_augmented(Hs, v, g["Tbc"])
    np.testing.assert_allclose(_export(H, v, 1, fusion.tangent_block(g["Tbc"]), 0.00025).reshape(n, n + 1), ref, rtol=1e-12,
                               atol=1e-12 * np.abs(ref).max())
    # the plain layout: the full symmetric matrix mirrored from the lower triangle, v behind it
    flat = _export(H, v, 0)
    assert np.array_equal(flat[:n * n].reshape(n, n), np.tril(H) + np.tril(H, -1).T) and np.array_equal(flat[n * n:n * n + n], v)


@pytest.mark.parametrize("P", [1, 24, 63, 64])
def test_export_at_window_sizes(P):
    rng = np.random.default_rng(P)
    n = 6 * P
    M = rng.standard_normal((n, n))
    H, v = M + M.T, rng.standard_normal(n)
    Tbc = np.array([0.05, -0.02, 0.11, 0.1, -0.2, 0.3, 0.9])
    ref = fusion.BA2GTSAM_augmented(H, v, Tbc)
    for rep in range(3):   # (the completion word carries a sequence number: repeated calls on fresh workspaces and on the same block)
        aug = _export(H, v, 1, fusion.tangent_block(Tbc)).reshape(n, n + 1)
        np.testing.assert_allclose(aug, ref, rtol=1e-11, atol=1e-12 * np.abs(ref).max())


def test_bacore_hessian_gtsam_is_hessian_plus_the_callers_two_statements():
    """BACore.hessian_gtsam(Tbc) == hessian(H, v); H[i,i] += 0.00025 (i < 6); gtsam.BA2GTSAM(H, v, Tbc) -- depth_video.py:394-401"""
    import droid_backends
    W = syn.make_window(*syn.graph_banded(10, 3), 10, 48, 64, seed=4, intr=(30.0, 30.0, 31.5, 23.7), sensor_frac=0.2)
    d = to_dev(W)
    n = 6 * (W.t1 - W.t0)
    Tbc = np.array([0.03, 0.01, -0.08, 0.02, -0.01, 0.7, 0.71])
    core = droid_backends.BACore()
    core.init(d["poses"], d["disps"], d["intrinsics"], d["disps_sens"], d["target"], d["weight"], d["eta"], d["ii"], d["jj"],
              W.t0, W.t1, 2, W.lm, W.ep, False)
    H, v = torch.zeros(n, n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
    core.hessian(H, v)
    Hn = H.numpy().copy()
    assert np.array_equal(Hn, Hn.T) and np.abs(Hn).max() > 0
    Hs, vs = core.hessian_staging()
    assert torch.equal(Hs, H) and torch.equal(vs, v)
    Hn[np.arange(6), np.arange(6)] += 0.00025
    ref = fusion.BA2GTSAM_augmented(Hn, v.numpy(), Tbc)
    aug = core.hessian_gtsam(Tbc).copy()
    assert aug.shape == (n, n + 1)
    # (the two hessian calls accumulate H with float64 atomics in different orders: the inputs agree to ~1e-13 relative)
    np.testing.assert_allclose(aug, ref, rtol=0, atol=1e-10 * np.abs(ref).max())
    # a partial H (the reference fills H_accessor.size(0) x size(1) entries, droid_kernels.cu:1892-1897)
    Hp, vp = torch.zeros(n - 6, n - 6, dtype=torch.float64), torch.zeros(n - 6, dtype=torch.float64)
    core.hessian(Hp, vp)
    np.testing.assert_allclose(Hp.numpy(), H.numpy()[:n - 6, :n - 6], rtol=0, atol=1e-10 * np.abs(Hn).max())
    # retract still works after the device H was left lower-triangular by the export
    dx = np.linalg.solve(H.numpy() + np.diag(W.ep + W.lm * np.diag(H.numpy())), v.numpy())
    core.retract(torch.from_numpy(dx))
    torch.cuda.synchronize()
    assert torch.isfinite(d["poses"]).all() and torch.isfinite(d["disps"]).all()
