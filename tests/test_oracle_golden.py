"""CPU: pins the oracle (oracle/) against vectors captured from the reference's own Python
(tests/golden/make_golden.py) -- SURVEY.md section 8(c).  The reference has no tests of its own."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from dbaf_amd import synthetic as syn


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_corr_pyramid_matches_reference_corrblock(golden_dir):
    g = _load(golden_dir, "corr_pyramid.npz")
    for tag in ("a", "b"):
        pyr = orc.corr_pyramid(g[f"{tag}_fmap1"], g[f"{tag}_fmap2"], 4)
        for lvl in range(4):
            ref = g[f"{tag}_lvl{lvl}"]
            assert pyr[lvl].shape == ref.shape
            np.testing.assert_allclose(pyr[lvl], ref, rtol=2e-5, atol=2e-5)


def _dense_schur(H, E, C, v, w, ep, lm):
    """numpy restatement of the algebra the goldens pin: S = H + damping - E Q E^T (chol.py:46-73)."""
    B, P, M, D, HW = E.shape
    Hd = H.transpose(0, 1, 3, 2, 4).reshape(P * D, P * D)
    Ed = E.transpose(0, 1, 3, 2, 4).reshape(P * D, M * HW)
    Q = (1.0 / C).reshape(M * HW)
    Hd = Hd + (ep + lm * Hd) * np.eye(P * D)
    S = Hd - (Ed * Q) @ Ed.T
    b = v.reshape(-1) - Ed @ (Q * w.reshape(-1))
    dx = np.linalg.solve(S, b)
    dz = Q * (w.reshape(-1) - Ed.T @ dx)
    return dx.reshape(P, D), dz.reshape(M, HW)


def test_schur_algebra_matches_reference_schur_solve(golden_dir):
    g = _load(golden_dir, "schur_solve.npz")
    dx, dz = _dense_schur(g["H"], g["E"], g["C"], g["v"], g["w"], float(g["ep"]), float(g["lm"]))
    np.testing.assert_allclose(dx, g["dx"][0], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(dz, g["dz"][0], rtol=1e-9, atol=1e-12)


def test_pinhole_conventions(golden_dir):
    g = _load(golden_dir, "pinhole.npz")
    disps, intr = g["disps"][0], g["intr"][0, 0]
    # x = column, y = row (projective_ops.py:11-16)
    assert g["grid"][2, 5, 0] == 5 and g["grid"][2, 5, 1] == 2
    ident = np.tile(np.array([0, 0, 0, 0, 0, 0, 1], np.float32), (2, 1))
    pts = orc.iproj(ident, disps, intr)  # world points = [X, Y, 1] / d with identity poses
    ref = g["pts"][0][..., :3] / g["pts"][0][..., 3:4]
    np.testing.assert_allclose(pts, ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_reproject_matches_reference_projective_transform(golden_dir, tag):
    g = _load(golden_dir, "projective.npz")
    coords, valid = orc.reproject(g[f"{tag}_poses"], g[f"{tag}_disps"], g[f"{tag}_intr"], g[f"{tag}_ii"],
                                  g[f"{tag}_jj"], np.float32)
    np.testing.assert_allclose(coords, g[f"{tag}_coords"], rtol=1e-4, atol=2e-4)
    assert (valid == g[f"{tag}_valid"]).mean() > 0.999


def test_linearisation_matches_reference_jacobians(golden_dir):
    """Hii,Hij,Hjj,vi,vj,Ei,Ej,Ck,wk of the CUDA restatement equal the products formed from the
    reference's torch Jacobians (geom/ba.py:44-67) wherever Z > 0.25."""
    g = _load(golden_dir, "projective.npz")
    W = syn.window_tiny_a(3)
    assert np.array_equal(W.poses, g["a_poses"])
    Ji, Jj, Jz = g["a_Ji"].astype(np.float64), g["a_Jj"].astype(np.float64), g["a_Jz"].astype(np.float64)
    N, h, w = len(W.ii), W.h, W.w
    lin = orc.linearize(W.poses, W.disps, W.intrinsics, W.target, W.weight, W.ii, W.jj, np.float64)
    _, Z = syn.reproject_np(W.poses, W.disps, W.intrinsics, W.ii, W.jj)
    assert Z.min() > 0.25  # no pixel hits the depth threshold in this fixture
    coords = g["a_coords"].astype(np.float64)
    r = (W.target.transpose(0, 2, 3, 1).astype(np.float64) - coords).reshape(N, h * w, 2)
    wgt = 0.001 * W.weight.transpose(0, 2, 3, 1).astype(np.float64).reshape(N, h * w, 2)
    Ji = Ji.reshape(N, h * w, 2, 6)
    Jj = Jj.reshape(N, h * w, 2, 6)
    Jz = Jz.reshape(N, h * w, 2)
    Hii = np.einsum("npc,npca,npcb->nab", wgt, Ji, Ji)
    Hij = np.einsum("npc,npca,npcb->nab", wgt, Ji, Jj)
    Hjj = np.einsum("npc,npca,npcb->nab", wgt, Jj, Jj)
    vi = np.einsum("npc,npca,npc->na", wgt, Ji, r)
    vj = np.einsum("npc,npca,npc->na", wgt, Jj, r)
    Ei = np.einsum("npc,npca,npc->nap", wgt, Ji, Jz)
    Ej = np.einsum("npc,npca,npc->nap", wgt, Jj, Jz)
    Ck = np.einsum("npc,npc,npc->np", wgt, Jz, Jz)
    wk = np.einsum("npc,npc,npc->np", wgt, r, Jz)
    tol = dict(rtol=2e-3, atol=1e-7)  # golden Jacobians are fp32
    np.testing.assert_allclose(lin["Hs"][0], Hii, **tol)
    np.testing.assert_allclose(lin["Hs"][1], Hij, **tol)
    np.testing.assert_allclose(lin["Hs"][2], Hij.transpose(0, 2, 1), **tol)
    np.testing.assert_allclose(lin["Hs"][3], Hjj, **tol)
    np.testing.assert_allclose(lin["vs"][0], vi, rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(lin["vs"][1], vj, rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(lin["Eii"], Ei, rtol=2e-3, atol=1e-8)
    np.testing.assert_allclose(lin["Eij"], Ej, rtol=2e-3, atol=1e-8)
    np.testing.assert_allclose(lin["Cii"], Ck, rtol=2e-3, atol=1e-10)
    np.testing.assert_allclose(lin["bz"], wk, rtol=2e-3, atol=1e-9)
