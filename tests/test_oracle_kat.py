"""Known-answer tests that stand in for the reference tests that do not exist:
finite-difference Jacobians, Gauss-Newton contraction on a noise-free scene, the zero-residual fixed point, the
stereo-edge special case, and the SE3 surface (shim and oracle) against the matrix exponential.  CPU only."""
import os
import sys

import numpy as np
import pytest
import scipy.linalg

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import oracle as orc  # noqa: E402  (tests may use the oracle)
from dbaf_amd import synthetic as syn  # noqa: E402


def _twist_matrix(xi):
    """se(3) element for xi = (tau, phi): translation first, as lietorch and droid_kernels.cu:141-178 order it"""
    tau, phi = xi[:3], xi[3:]
    M = np.zeros((4, 4))
    M[:3, :3] = np.array([[0, -phi[2], phi[1]], [phi[2], 0, -phi[0]], [-phi[1], phi[0], 0]])
    M[:3, 3] = tau
    return M


def _pose_matrix(p):
    t, q = p[:3], p[3:]
    x, y, z, w = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    return T


def test_oracle_retraction_is_left_multiplication_by_the_matrix_exponential():
    """pose_retr_kernel (droid_kernels.cu:922-951): T <- Exp(xi) * T, xi = (tau, phi)"""
    rng = np.random.default_rng(0)
    W = syn.window_tiny_a(1)
    poses = W.poses.astype(np.float64)
    poses[:, 3:] /= np.linalg.norm(poses[:, 3:], axis=1, keepdims=True)  # the fixture's quaternions are unit in f32 only
    P = W.t1 - W.t0
    dx = 0.2 * rng.standard_normal((P, 6))
    out = orc.pose_retr(poses, dx, W.t0, W.t1, np.float64)
    for p in range(P):
        want = scipy.linalg.expm(_twist_matrix(dx[p])) @ _pose_matrix(poses[W.t0 + p])
        got = _pose_matrix(out[W.t0 + p])
        np.testing.assert_allclose(got, want, atol=1e-10)
    untouched = [k for k in range(len(poses)) if not (W.t0 <= k < W.t1)]
    assert np.array_equal(out[untouched], poses[untouched])


def test_se3_shim_against_matrix_exponential_and_group_identities():
    """the lietorch surface DBA-Fusion uses: exp/log, retr, mul, inv, act, adj/adjT, matrix"""
    import torch
    from lietorch import SE3
    rng = np.random.default_rng(1)
    xi = torch.from_numpy(0.5 * rng.standard_normal((5, 6)))
    X = SE3.exp(xi)
    M = X.matrix().numpy()
    for k in range(5):
        np.testing.assert_allclose(M[k], scipy.linalg.expm(_twist_matrix(xi[k].numpy())), atol=1e-10)
    np.testing.assert_allclose(X.log().numpy(), xi.numpy(), atol=1e-9)
    Y = SE3.exp(torch.from_numpy(0.3 * rng.standard_normal((5, 6))))
    np.testing.assert_allclose((X * Y).matrix().numpy(), M @ Y.matrix().numpy(), atol=1e-10)
    np.testing.assert_allclose((X * X.inv()).matrix().numpy(), np.broadcast_to(np.eye(4), (5, 4, 4)), atol=1e-10)
    d = torch.from_numpy(0.1 * rng.standard_normal((5, 6)))
    np.testing.assert_allclose(Y.retr(d).matrix().numpy(), (SE3.exp(d) * Y).matrix().numpy(), atol=1e-10)
    pts = torch.from_numpy(rng.standard_normal((5, 4)))
    np.testing.assert_allclose((X * pts).numpy(), np.einsum("kab,kb->ka", M, pts.numpy()), atol=1e-10)
    # Adjoint: X Exp(a) X^-1 = Exp(Adj_X a); adjT is its transpose (used on Jacobian rows, projective_ops.py:118-121)
    a = torch.from_numpy(1e-3 * rng.standard_normal((5, 6)))
    lhs = (X * SE3.exp(a) * X.inv()).matrix().numpy()
    rhs = SE3.exp(X.adj(a)).matrix().numpy()
    np.testing.assert_allclose(lhs, rhs, atol=1e-10)
    b = torch.from_numpy(rng.standard_normal((5, 6)))
    np.testing.assert_allclose((X.adjT(b) * a).sum(-1).numpy(), (b * X.adj(a)).sum(-1).numpy(), atol=1e-12)


def _cost(W, poses, disps):
    """0.5 * sum 0.001 * w * (target - pi)^2 over pixels with Z >= 0.25 on both sides (the weights the
    linearisation uses, droid_kernels.cu:296-311)"""
    coords, valid = orc.reproject(poses, disps, W.intrinsics, W.ii, W.jj, np.float64)
    r = W.target.transpose(0, 2, 3, 1).astype(np.float64) - coords
    w = 0.001 * W.weight.transpose(0, 2, 3, 1).astype(np.float64) * valid.reshape(valid.shape[0], valid.shape[1], valid.shape[2], 1)
    return 0.5 * float((w * r * r).sum())


def test_linearisation_gradient_matches_finite_differences():
    """vi, vj (pose gradients under the LEFT retraction) and bz (per-pixel depth gradient) of
    projective_transform_kernel equal -d cost / d (xi_i, xi_j, d): pins the Jacobian sign and frame conventions
    independently of the reference's torch Jacobians"""
    W = syn.window_tiny_a(3)
    poses, disps = W.poses.astype(np.float64), W.disps.astype(np.float64)
    _, Z = syn.reproject_np(W.poses, W.disps, W.intrinsics, W.ii, W.jj)
    assert Z.min() > 0.3  # away from the validity threshold, so the cost is smooth
    lin = orc.linearize(poses, disps, W.intrinsics, W.target, W.weight, W.ii, W.jj, np.float64)
    B = len(poses)
    g_pose = np.zeros((B, 6))
    for n, (i, j) in enumerate(zip(W.ii, W.jj)):
        g_pose[i] += lin["vs"][0][n]
        g_pose[j] += lin["vs"][1][n]
    eps = 1e-6
    for k in range(1, B):
        for c in range(6):
            dx = np.zeros((B, 6))
            dx[k, c] = eps
            cp = _cost(W, orc.pose_retr(poses, dx, 0, B, np.float64), disps)
            dx[k, c] = -eps
            cm = _cost(W, orc.pose_retr(poses, dx, 0, B, np.float64), disps)
            fd = -(cp - cm) / (2 * eps)
            assert abs(fd - g_pose[k, c]) <= 1e-5 * max(1.0, abs(fd)), (k, c, fd, g_pose[k, c])
    # depth gradient at a few pixels: bz[n] summed over the edges leaving the pixel's frame
    h, w = W.h, W.w
    rng = np.random.default_rng(0)
    for _ in range(6):
        f = int(rng.integers(0, B))
        y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
        g = sum(lin["bz"][n][y * w + x] for n in range(len(W.ii)) if W.ii[n] == f)
        dp, dm = disps.copy(), disps.copy()
        dp[f, y, x] += eps
        dm[f, y, x] -= eps
        fd = -(_cost(W, poses, dp) - _cost(W, poses, dm)) / (2 * eps)
        assert abs(fd - g) <= 1e-5 * max(1e-3, abs(fd)), (f, y, x, fd, g)


def _noise_free(W):
    coords, _ = orc.reproject(W.poses_gt, W.disps_gt, W.intrinsics, W.ii, W.jj, np.float64)
    target = np.ascontiguousarray(coords.transpose(0, 3, 1, 2))  # float64: exact zero residual
    weight = np.ones_like(W.weight)
    return target, weight


def test_zero_residual_is_a_fixed_point_and_gauss_newton_contracts():
    """target = reproject(ground truth): at the ground truth the update is zero; from a perturbed state a few
    Gauss-Newton iterations shrink the reprojection residual by orders of magnitude"""
    W = syn.window_tiny_a(2)
    target, weight = _noise_free(W)
    eta = 1e-6 * np.ones_like(W.eta)
    p0, d0 = W.poses_gt.astype(np.float64), W.disps_gt.astype(np.float64)
    out = orc.ba(p0, d0, W.intrinsics, W.disps_sens, target, weight, eta, W.ii, W.jj, W.t0, W.t1, 1, 1e-4, 0.1, False,
                 0.05, np.float64)
    assert np.abs(out["dx"]).max() < 1e-9 and np.abs(out["dz"]).max() < 1e-9
    assert np.allclose(out["poses"], p0, atol=1e-9) and np.allclose(out["disps"], d0, atol=1e-9)

    def resid(p, d):
        c, v = orc.reproject(p, d, W.intrinsics, W.ii, W.jj, np.float64)
        r = (target.transpose(0, 2, 3, 1) - c) * v.reshape(v.shape[0], v.shape[1], v.shape[2], 1)
        return float(np.sqrt((r * r).mean()))

    r0 = resid(W.poses.astype(np.float64), W.disps.astype(np.float64))
    p, d = W.poses.astype(np.float64), W.disps.astype(np.float64)
    out = orc.ba(p, d, W.intrinsics, W.disps_sens, target, weight, eta, W.ii, W.jj, W.t0, W.t1, 8, 1e-4, 1e-6, False,
                 0.05, np.float64)
    r1 = resid(out["poses"], out["disps"])
    assert r0 > 1e-2, r0
    assert r1 < 1e-2 * r0, (r0, r1)


def test_stereo_edge_uses_the_fixed_baseline_and_has_no_pose_gradient():
    """ii == jj marks a stereo edge: relative pose = (-0.1, 0, 0, identity) and both pose Jacobians are zero
    (droid_kernels.cu:262-270, :331-335); only the depth block receives information"""
    W = syn.window_tiny_a(4)
    ii = np.array([1, 2], np.int64)
    jj = np.array([1, 2], np.int64)
    poses, disps = W.poses.astype(np.float64), W.disps.astype(np.float64)
    coords, valid = orc.reproject(poses, disps, W.intrinsics, ii, jj, np.float64)
    fx, cx = float(W.intrinsics[0]), float(W.intrinsics[2])
    y, x = np.meshgrid(np.arange(W.h, dtype=np.float64), np.arange(W.w, dtype=np.float64), indexing="ij")
    for n, f in enumerate(ii):
        # X' = X - 0.1 d  =>  u' = fx (X - 0.1 d) + cx = x - 0.1 fx d ;  v' = y
        np.testing.assert_allclose(coords[n, ..., 0], x - 0.1 * fx * disps[f], atol=1e-9)
        np.testing.assert_allclose(coords[n, ..., 1], y, atol=1e-9)
    tgt = np.ascontiguousarray(coords.transpose(0, 3, 1, 2)) + 0.25
    lin = orc.linearize(poses, disps, W.intrinsics, tgt, np.ones_like(tgt), ii, jj, np.float64)
    assert np.abs(lin["Hs"]).max() == 0.0 and np.abs(lin["vs"]).max() == 0.0
    assert np.abs(lin["Eii"]).max() == 0.0 and np.abs(lin["Eij"]).max() == 0.0
    assert (lin["Cii"] > 0).all() and (np.abs(lin["bz"]) > 0).all()
    del cx


def test_cholesky_failure_returns_a_zero_pose_update():
    """a non-SPD reduced system is not an error: dx = 0 (droid_kernels.cu:193-196, :1263-1266)"""
    W = syn.window_tiny_a(5)
    weight = -np.abs(W.weight) - 1.0  # negative weights make the normal matrix negative definite
    out = orc.ba(W.poses, W.disps, W.intrinsics, W.disps_sens, W.target, weight, W.eta, W.ii, W.jj, W.t0, W.t1, 1,
                 1e-4, 1e-6, True, 0.05, np.float64)
    assert np.abs(out["dx"]).max() == 0.0
    assert np.allclose(out["poses"], W.poses)
