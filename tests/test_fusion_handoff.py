"""CPU: the numpy side of the BACore <-> GTSAM hand-off (dbaf_amd/fusion.py) against the reference's own numpy
BA2GTSAM (dbaf/depth_video.py:20-29; committed vectors: tests/golden/ba2gtsam.npz, make_golden.gen_ba2gtsam) and the
algebra that ties GTSAM2BA to it."""
import os

import numpy as np

from dbaf_amd import fusion


def test_ba2gtsam_matches_reference_vectors(golden_dir):
    g = np.load(os.path.join(golden_dir, "ba2gtsam.npz"))
    Hg, vg = fusion.BA2GTSAM(g["H"], g["v"], g["Tbc"])
    np.testing.assert_allclose(Hg, g["Hg"], rtol=1e-12, atol=1e-12 * np.abs(g["Hg"]).max())
    np.testing.assert_allclose(vg, g["vg"], rtol=1e-12, atol=1e-12 * np.abs(g["vg"]).max())
    aug = fusion.BA2GTSAM_augmented(g["H"], g["v"], g["Tbc"])
    n = g["H"].shape[0]
    assert aug.shape == (n, n + 1) and np.array_equal(aug[:, :n], Hg) and np.array_equal(aug[:, n], vg)
    np.testing.assert_allclose(fusion.adjoint_map(g["Tbc"]), g["Ad"], rtol=1e-13, atol=1e-15)


def test_gtsam2ba_is_the_map_under_which_ba2gtsam_is_a_congruence():
    rng = np.random.default_rng(0)
    P = 5
    M = rng.standard_normal((6 * P, 6 * P + 3))
    H, v = M @ M.T, rng.standard_normal(6 * P)
    Tbc = np.array([0.05, -0.02, 0.11, 0.1, -0.2, 0.3, 0.9])
    Hg, vg = fusion.BA2GTSAM(H, v, Tbc)
    dxg = np.linalg.solve(Hg, vg)
    np.testing.assert_allclose(fusion.GTSAM2BA(dxg, Tbc), np.linalg.solve(H, v), rtol=1e-9, atol=1e-12)
    # the quadratic model is the same function of the increment in either coordinates
    d = rng.standard_normal(6 * P)
    db = fusion.GTSAM2BA(d, Tbc)
    np.testing.assert_allclose(0.5 * d @ Hg @ d - vg @ d, 0.5 * db @ H @ db - v @ db, rtol=1e-10)


def test_marginal_prior_is_the_schur_complement_of_the_joint_solve():
    rng = np.random.default_rng(1)
    P, keep = 6, 2
    M = rng.standard_normal((6 * P, 6 * P + 3))
    H, v = M @ M.T, rng.standard_normal(6 * P)
    Hk, vk = fusion.marginal_prior(H, v, keep)
    np.testing.assert_allclose(np.linalg.solve(Hk, vk), np.linalg.solve(H, v)[6 * keep:], rtol=1e-8, atol=1e-11)
