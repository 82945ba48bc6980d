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


def _schur_case_to_rows(E, t0):
    """dense coupling blocks E [P, M, 6, HW] (chol.py:49-50) -> the CUDA path's row form (droid_kernels.cu:1480-1481):
    rows 0..P-1 belong to the poses themselves (source frame == pose frame), one further row per (source frame m ->
    pose p) pair."""
    P, M, D, HW = E.shape
    rows = [E[p, t0 + p] for p in range(P)]
    ii, jj = [], []
    for p in range(P):
        for m in range(M):
            if m != t0 + p:
                rows.append(E[p, m])
                ii.append(m)
                jj.append(t0 + p)
    return np.stack(rows), np.asarray(ii, np.int64), np.asarray(jj, np.int64)


@pytest.mark.parametrize("tag", ["", "z_"])
def test_oracle_schur_and_backsub_match_reference_schur_solve(golden_dir, tag):
    """The C oracle's OWN step_schur / sys_solve / step_backsub (not a formula written here) against what the
    reference's schur_solve computed (chol.py:46-73): the reduced system S, v that reached its Cholesky, dx and dz.
    The torch path damps before the Schur complement, so the damped pose block goes in and the solve runs undamped."""
    g = _load(golden_dir, "schur_solve.npz")
    H, E, C, v, w = (g[tag + k][0] for k in ("H", "E", "C", "v", "w"))
    ep, lm = float(g["ep"]), float(g["lm"])
    P, M, D, HW = E.shape
    t0 = 1
    A = H.transpose(0, 2, 1, 3).reshape(P * D, P * D).copy()
    A[np.diag_indices(P * D)] += ep + lm * np.diag(A)  # chol.py:55-56
    rows, ii, jj = _schur_case_to_rows(E, t0)
    S, b, _ = orc.schur_rows(rows, C, w, ii, jj, M, 1, HW, t0, t0 + P, A, v.reshape(-1))
    np.testing.assert_allclose(S, g[tag + "S"][0], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(b, g[tag + "vred"][0, :, 0], rtol=1e-11, atol=1e-12)
    x, ok = orc.sys_solve(S, b, 0.0, 0.0)
    assert ok
    np.testing.assert_allclose(x.reshape(P, D), g[tag + "dx"][0], rtol=1e-9, atol=1e-12)
    _, _, dz = orc.schur_rows(rows, C, w, ii, jj, M, 1, HW, t0, t0 + P, A, v.reshape(-1), dx=g[tag + "dx"][0])
    if tag == "z_":  # first pose has no depth coupling: the two back-substitutions are the same function
        np.testing.assert_allclose(dz, g[tag + "dz"][0], rtol=1e-9, atol=1e-12)
    else:
        # EvT6x1_kernel skips rows whose pose index is <= 0 (droid_kernels.cu:1150): the CUDA path's dz lacks the
        # first optimised pose's term, Q * E[0]^T dx[0], and nothing else
        missing = np.einsum("mdk,d->mk", E[0], g["dx"][0, 0]) / C
        assert np.abs(missing).max() > 1e-3
        np.testing.assert_allclose(dz - missing, g["dz"][0], rtol=1e-9, atol=1e-12)


def _dense_from_rows(Erows, ii_exp, jj_exp, kx, t0, P):
    M = len(kx)
    dense = np.zeros((P, M) + Erows.shape[1:], Erows.dtype)
    for r in range(len(ii_exp)):
        p = int(jj_exp[r]) - t0
        if 0 <= p < P:
            dense[p, int(np.searchsorted(kx, ii_exp[r]))] += Erows[r]
    return dense


def test_oracle_gauss_newton_step_matches_reference_torch_BA(golden_dir):
    """One Gauss-Newton step of the reference's torch BA (geom/ba.py:29-104, captured by make_golden.gen_ba_step):
    the oracle's assembled system (pose block, coupling rows, depth block), its Schur complement, solve,
    back-substitution and retraction reproduce it where the two variants coincide (every
    Z > 0.25, no depth measurements; eta shifted by the torch path's +1e-7, damping moved before the Schur
    complement, first-pose skip added back)."""
    g = _load(golden_dir, "ba_step.npz")
    W = syn.window_tiny_a(int(g["seed"]))
    K, t0 = W.num_kf, int(g["t0"])
    P, HW = K - t0, W.h * W.w
    kx = g["kx"]
    eta = W.eta[:len(kx)].astype(np.float64) + 1e-7  # ba.py:89
    core = orc.BACore(W.poses[:K], W.disps[:K], W.intrinsics, W.disps_sens[:K], W.target, W.weight, eta, W.ii, W.jj,
                      t0, K, dtype=np.float64)
    assert core.M == len(kx)
    A, v, C = core.presystem(0.05)
    Erows, Q, w = core.get_EQw()
    f32 = dict(rtol=2e-5)  # the golden is fp32 end to end; measured agreement 1e-7 .. 9e-7 of the largest entry
    Hg = g["H"].transpose(0, 2, 1, 3).reshape(6 * P, 6 * P).astype(np.float64)
    np.testing.assert_allclose(A, Hg, atol=5e-6 * np.abs(Hg).max(), **f32)
    np.testing.assert_allclose(v, g["v"].reshape(-1), atol=5e-6 * np.abs(g["v"]).max(), **f32)
    np.testing.assert_allclose(C, g["C"], atol=1e-12, **f32)
    np.testing.assert_allclose(w, g["w"], atol=5e-6 * np.abs(g["w"]).max(), **f32)
    ii_exp = np.concatenate([np.arange(t0, K), W.ii])
    jj_exp = np.concatenate([np.arange(t0, K), W.jj])
    Ed = _dense_from_rows(Erows, ii_exp, jj_exp, kx, t0, P)
    np.testing.assert_allclose(Ed, g["E"], atol=5e-6 * np.abs(g["E"]).max(), **f32)
    # reduced system, with the torch path's damping order (chol.py:55-56)
    Ad = A.copy()
    Ad[np.diag_indices(6 * P)] += 0.1 + 1e-4 * np.diag(A)
    S, b, _ = orc.schur_rows(Erows, C, w, W.ii, W.jj, K, W.h, W.w, t0, K, Ad, v)
    np.testing.assert_allclose(S, g["S"], atol=5e-6 * np.abs(g["S"]).max(), **f32)
    np.testing.assert_allclose(b, g["vred"][:, 0], atol=5e-6 * np.abs(g["vred"]).max(), **f32)
    x, ok = orc.sys_solve(S, b, 0.0, 0.0)
    assert ok
    np.testing.assert_allclose(x.reshape(P, 6), g["dx"], atol=2e-4 * np.abs(g["dx"]).max(), rtol=2e-4)
    _, _, dz = orc.schur_rows(Erows, C, w, W.ii, W.jj, K, W.h, W.w, t0, K, Ad, v, dx=g["dx"])
    missing = np.einsum("mdk,d->mk", Ed[0], g["dx"][0].astype(np.float64)) / C   # droid_kernels.cu:1150
    np.testing.assert_allclose(dz - missing, g["dz"], atol=2e-4 * np.abs(g["dz"]).max(), rtol=2e-4)
    # retraction: T <- Exp(dx) T (ba.py:25-27 / droid_kernels.cu:922-976); disps + dz, > 10 -> 0, clamp at 0 (ba.py:98-102)
    poses1 = orc.pose_retr(W.poses[:K], g["dx"], t0, K, np.float64)
    np.testing.assert_allclose(poses1, g["poses1"], rtol=0, atol=2e-6)
    d1 = W.disps[:K].astype(np.float64) + g["dz"].reshape(len(kx), W.h, W.w)
    d1 = np.clip(np.where(d1 > 10, 0.0, d1), 0.0, None)
    np.testing.assert_allclose(d1, g["disps1"], rtol=0, atol=2e-6)


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
