"""Absolute trajectory error the way the reference's evaluation scripts compute it
(/root/reference/evaluation_scripts/evaluate_tumvi.py:132-135, :150-175: evo's APE with pose_relation =
translation part, align = True, correct_scale only for visual-only runs, RMSE statistic).

evo is not a dependency here; the alignment is Umeyama's closed-form least-squares similarity / rigid registration
(S. Umeyama, "Least-squares estimation of transformation parameters between two point patterns", TPAMI 13(4), 1991),
which is what evo's `trajectory.align` applies.  Host-side numpy: it scores trajectories, it is not on the hot path.
"""
import numpy as np


def camera_centres(poses):
    """world -> camera poses [n, 7] (tx, ty, tz, qx, qy, qz, qw), as DepthVideo.poses stores them -> centres [n, 3]"""
    p = np.asarray(poses, np.float64)
    t, q = p[:, :3], p[:, 3:] / np.linalg.norm(p[:, 3:], axis=1, keepdims=True)
    qv, w = q[:, :3], q[:, 3:4]
    # R^T t with R from q: rotate t by the conjugate quaternion
    uv = 2.0 * np.cross(-qv, t)
    rt = t + w * uv + np.cross(-qv, uv)
    return -rt


def umeyama(src, dst, with_scale=False):
    """least-squares (R, t, s) with dst ~ s R src + t"""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    mu_s, mu_d = src.mean(0), dst.mean(0)
    xs, xd = src - mu_s, dst - mu_d
    cov = xd.T @ xs / len(src)
    U, D, Vt = np.linalg.svd(cov)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1.0
    R = U @ S @ Vt
    var_s = (xs ** 2).sum() / len(src)
    s = float(np.trace(np.diag(D) @ S) / var_s) if with_scale and var_s > 0 else 1.0
    t = mu_d - s * R @ mu_s
    return R, t, s


def ape_translation_rmse(ref_xyz, est_xyz, align=True, correct_scale=False):
    """evo APE, translation part, RMSE (metres): est aligned onto ref first (SE3, or Sim3 with correct_scale)"""
    ref, est = np.asarray(ref_xyz, np.float64), np.asarray(est_xyz, np.float64)
    assert ref.shape == est.shape and ref.shape[1] == 3
    if align and len(ref) >= 3:
        R, t, s = umeyama(est, ref, with_scale=correct_scale)
        est = s * est @ R.T + t
    return float(np.sqrt(((ref - est) ** 2).sum(1).mean()))


def ate(ref_poses, est_poses, **kw):
    """ATE RMSE between two world -> camera pose arrays [n, 7]"""
    return ape_translation_rmse(camera_centres(ref_poses), camera_centres(est_poses), **kw)
