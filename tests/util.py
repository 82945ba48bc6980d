import numpy as np
import torch


def to_dev(W, device="cuda"):
    """Window (numpy, reference layouts) -> dict of torch device tensors."""
    t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dt)  # noqa: E731
    return dict(poses=t(W.poses), disps=t(W.disps), intrinsics=t(W.intrinsics), disps_sens=t(W.disps_sens),
                target=t(W.target), weight=t(W.weight), eta=t(W.eta), ii=t(W.ii), jj=t(W.jj))


def quat_angle(qa, qb):
    """rotation angle (rad) between quaternion arrays [...,4] (xyzw), not assuming exact unit norm"""
    qa = qa / np.linalg.norm(qa, axis=-1, keepdims=True)
    qb = qb / np.linalg.norm(qb, axis=-1, keepdims=True)
    d = np.abs((qa * qb).sum(-1)).clip(0, 1)
    # 2*acos(d) loses precision near d=1: use the sine form
    cr = np.linalg.norm(qa - qb * np.sign((qa * qb).sum(-1, keepdims=True)), axis=-1)
    return 2.0 * np.arcsin(np.clip(cr / 2.0, 0, 1))


def check_state(poses, disps, ref_poses, ref_disps, old_disps, ref32_disps=None, t_tol=1e-5, r_tol=1e-6,
                d_rtol=1e-4, frac=0.995):
    """north_star tolerances: poses 1e-5 m / 1e-6 rad; inverse depths 1e-4 relative -- measured against the
    float64 arbiter instantiation of the oracle.

    Depth criterion, per pixel:  |d - d_ref| <= max(d_rtol * max(|d_ref|, |d_old|), 2 * |d_ref32 - d_ref|)
      * the update is d_old + dz and may cancel, so the natural scale of fp32 rounding is the larger of the two;
      * at the few ill-conditioned pixels where the reference's OWN fp32 arithmetic (the faithful fp32
        restatement, ref32) is further than that from exact arithmetic, the HIP path must be no worse than
        twice the reference's own deviation (measured: oracle32 vs oracle64 reaches 2e-4 on such pixels).
    In addition at least `frac` of the non-cancelling pixels (|d_ref| >= 0.1 |d_old|) must meet the pure relative
    bound d_rtol * |d_ref|."""
    poses, ref_poses = np.asarray(poses, np.float64), np.asarray(ref_poses, np.float64)
    dt = np.abs(poses[:, :3] - ref_poses[:, :3]).max()
    dr = quat_angle(poses[:, 3:], ref_poses[:, 3:]).max()
    d, r, o = np.asarray(disps, np.float64), np.asarray(ref_disps, np.float64), np.asarray(old_disps, np.float64)
    err = np.abs(d - r)
    scale = np.maximum(np.abs(r), np.abs(o))
    allowed = d_rtol * scale
    if ref32_disps is not None:
        allowed = np.maximum(allowed, 2.0 * np.abs(np.asarray(ref32_disps, np.float64) - r))
    worst = (err / np.maximum(allowed, 1e-300)).max()
    solid = np.abs(r) >= 0.1 * np.abs(o)
    pure = (err[solid] <= d_rtol * np.abs(r[solid])).mean() if solid.any() else 1.0
    msg = "dt=%.3e m dr=%.3e rad depth max(err/allowed)=%.3f max(err/scale)=%.3e pure-rel frac=%.6f" % (
        dt, dr, worst, (err / np.maximum(scale, 1e-12)).max(), pure)
    assert dt <= t_tol, msg
    assert dr <= r_tol, msg
    assert worst <= 1.0, msg
    assert pure >= frac, msg
    return msg
