"""Host side of the BACore <-> GTSAM hand-off.

DBA-Fusion's multi-sensor path (dbaf/depth_video.py:350-462, :469-559) moves the Schur-reduced camera system from
`BACore.hessian` into a GTSAM factor and the solved increment back into `BACore.retract`.  Two changes of tangent
coordinates sit on that boundary:

  BA2GTSAM(H, v, Tbc)   dbaf/depth_video.py:20-29 (pure numpy in the reference; its runtime calls the C++ twin of a
                        GTSAM fork, `gtsam.BA2GTSAM`, which returns the augmented [Hg | vg] matrix)
  GTSAM2BA(dx, Tbc)     only exists in the GTSAM fork (yuxuanzhou97/gtsam, un-vendored, absent here: parity at that
                        boundary is unpinned); it must be the linear map under which BA2GTSAM is a congruence:
                        dx_ba = J dx_gtsam.

BA tangents are (translation, rotation) per camera pose, left-multiplied on the world->camera pose; GTSAM's Pose3
tangents are (rotation, translation) on the body pose.  With A = -Ad(Tbc^-1) and its row halves swapped,
J = blockdiag(A), H_g = J^T H J, v_g = J^T v.

These run on the host in numpy: the system is 144x144 (24 poses) .. 384x384 (64 poses).  A `gtsam.Pose3` also works
as `Tbc` (anything with .matrix(), or .inverse().AdjointMap()).
"""
import numpy as np


def pose_matrix(T):
    """4x4 homogeneous matrix from: a 4x4 array, a (t, q_xyzw) 7-vector, or an object with .matrix()"""
    if hasattr(T, "matrix") and callable(T.matrix):
        return np.asarray(T.matrix(), np.float64).reshape(4, 4)
    a = np.asarray(T, np.float64)
    if a.shape == (4, 4):
        return a
    if a.shape == (7,):
        x, y, z, w = a[3:] / np.linalg.norm(a[3:])
        M = np.eye(4)
        M[:3, :3] = [[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]]
        M[:3, 3] = a[:3]
        return M
    raise ValueError("Tbc must be a 4x4 matrix, a (t, q) 7-vector or a Pose3-like object")


def adjoint_map(T):
    """gtsam::Pose3::AdjointMap of T = (R, t) in GTSAM's (rotation, translation) tangent order:
    [[R, 0], [[t]x R, R]]"""
    M = pose_matrix(T)
    R, t = M[:3, :3], M[:3, 3]
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    A = np.zeros((6, 6))
    A[:3, :3] = R
    A[3:, :3] = tx @ R
    A[3:, 3:] = R
    return A


def tangent_block(Tbc):
    """the 6x6 block A of depth_video.py:21-23: -Ad(Tbc^-1) with its row halves swapped (rotation rows last)"""
    if hasattr(Tbc, "inverse") and hasattr(Tbc.inverse(), "AdjointMap"):
        A = -np.asarray(Tbc.inverse().AdjointMap(), np.float64)
    else:
        A = -adjoint_map(np.linalg.inv(pose_matrix(Tbc)))
    return np.concatenate([A[3:6, :], A[0:3, :]], axis=0)


def BA2GTSAM(H, v, Tbc):
    """(J^T H J, J^T v), J = blockdiag(A) -- depth_video.py:20-29, block-wise (no 6P x 6P J is formed)"""
    H = np.asarray(H, np.float64)
    v = np.asarray(v, np.float64).reshape(-1)
    n = H.shape[0]
    assert H.shape == (n, n) and n % 6 == 0 and v.shape[0] == n
    A = tangent_block(Tbc)
    P = n // 6
    Hb = H.reshape(P, 6, P, 6)
    Hg = np.einsum("ka,ikjl,lb->iajb", A, Hb, A).reshape(n, n)
    vg = (v.reshape(P, 6) @ A).reshape(n)
    return Hg, vg


def BA2GTSAM_augmented(H, v, Tbc):
    """what the fork's `gtsam.BA2GTSAM` returns and depth_video.py:398-401, :527-529 slice: [n, n+1] = [Hg | vg]"""
    Hg, vg = BA2GTSAM(H, v, Tbc)
    return np.concatenate([Hg, vg[:, None]], axis=1)


def GTSAM2BA(dx, Tbc):
    """increment in GTSAM tangents [6P] -> BA tangents [6P]: dx_ba = J dx_gtsam (depth_video.py:557)"""
    dx = np.asarray(dx, np.float64).reshape(-1, 6)
    return (dx @ tangent_block(Tbc).T).reshape(-1)


def marginal_prior(Hg, vg, keep_from):
    """Schur complement of a Gaussian information pair onto the poses [keep_from, P): the dense stand-in for
    gtsam.marginalizeOut on a graph that holds only the visual factor (depth_video.py:443), used by the tests and by
    callers without a GTSAM build.  Returns (H_kept, v_kept)."""
    k = 6 * int(keep_from)
    Hmm, Hmk, Hkk = Hg[:k, :k], Hg[:k, k:], Hg[k:, k:]
    sol = np.linalg.solve(Hmm, np.concatenate([Hmk, vg[:k, None]], axis=1))
    return Hkk - Hmk.T @ sol[:, :-1], vg[k:] - Hmk.T @ sol[:, -1]
