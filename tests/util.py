import json
import os

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


REPORT = os.environ.get("DBA_PARITY_REPORT") or os.path.join(
    os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "parity_report.jsonl")


def _record(rec):
    """one line per parity comparison (test id, measured worst cases): the committed copy under profiles/ makes the
    slack between the measured error and the asserted tolerance visible"""
    rec = dict(test=os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0],
               solver=os.environ.get("DBA_SOLVE_KERNEL", "default"), **rec)
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        with open(REPORT, "a") as fh:
            fh.write(json.dumps(rec) + "\n")
    except OSError:
        pass


def check_depths_every_pixel(disps, ref_disps, old_disps, ref32_disps, d_rtol=1e-4, floor_factor=1.25):
    """THE statement of the depth tolerance on the windows of BASELINE's configs, every pixel, no fraction:

    comparator: the float64 arbiter instantiation of the oracle (the reference's algorithm in exact-enough arithmetic);
      (1) |d - d_ref| <= 1e-4 * max(|d_ref|, |d_old|)                          -- the inverse depth's own scale, before / after
      (2) |d - d_ref| <= max(1e-4 * |d_ref|, 1.25 * |d_ref32 - d_ref|)        -- relative to the NEW depth alone, except where
          the reference's own fp32 arithmetic (the fp32-faithful oracle, ref32) is at the same noise floor: a pixel the update
          shrinks strongly turns the ~3e-7 m of fp32 noise in the pose increment into ~1e-4 of its depth (dz = Q (w - E^T dx),
          Q E ~ 300 / m), for the reference exactly as for this path; there the device may be at most 25 % further from the
          arbiter than the reference's own arithmetic is.  (2) is asked of the pixels whose depth the update does not cancel
          (|d_ref| >= 0.1 |d_old|: all but a handful; where it does cancel, |d_new| is no scale for an fp32 sum and (1) is the bound).
    Returns (worst of (1) as a multiple of its bound, worst of (2), pixels that needed the noise-floor clause)."""
    d, r, o = np.asarray(disps, np.float64), np.asarray(ref_disps, np.float64), np.asarray(old_disps, np.float64)
    q = np.asarray(ref32_disps, np.float64)
    err = np.abs(d - r)
    b1 = d_rtol * np.maximum(np.abs(r), np.abs(o))
    b2 = np.maximum(d_rtol * np.abs(r), floor_factor * np.abs(q - r))
    solid = np.abs(r) >= 0.1 * np.abs(o)
    ratio2 = np.where(solid, err / np.maximum(b2, 1e-300), 0.0)
    w1, w2 = float((err / np.maximum(b1, 1e-300)).max()), float(ratio2.max())
    needed = int(((err > d_rtol * np.abs(r)) & solid).sum())
    i2 = int(np.argmax(ratio2))
    _record(dict(kind="depths_every_pixel", worst_over_scale_bound=w1, worst_over_dref_or_floor_bound=w2,
                 pixels_beyond_1e4_of_dref=needed, pixels=int(err.size), cancelling_pixels=int((~solid).sum()),
                 worst_pixel=dict(index=[int(v) for v in np.unravel_index(i2, err.shape)], d_ref=float(r.flat[i2]), d_old=float(o.flat[i2]),
                                  err_over_dref=float(err.flat[i2] / max(abs(r.flat[i2]), 1e-300)),
                                  ref32_err_over_dref=float(abs(q.flat[i2] - r.flat[i2]) / max(abs(r.flat[i2]), 1e-300)))))
    assert w1 <= 1.0, "a pixel is %.3f x the bound 1e-4 max(|d_ref|, |d_old|)" % w1
    assert w2 <= 1.0, "a pixel is %.3f x the bound max(1e-4 |d_ref|, %.2f |d_ref32 - d_ref|)" % (w2, floor_factor)
    return w1, w2, needed


def check_state(poses, disps, ref_poses, ref_disps, old_disps, ref32_disps=None, t_tol=1e-5, r_tol=1e-6,
                d_rtol=1e-4, frac=0.995, ref32_factor=2.0, ref32_poses=None, log32_disps=None, log32_poses=None):
    """north_star tolerances: poses 1e-5 m / 1e-6 rad; inverse depths 1e-4 relative -- measured against the
    float64 arbiter instantiation of the oracle.

    Depth criterion, per pixel:  |d - d_ref| <= max(d_rtol * max(|d_ref|, |d_old|), ref32_factor * |d_ref32 - d_ref|)
      * the update is d_old + dz and may cancel, so the natural scale of fp32 rounding is the larger of the two;
      * at the few ill-conditioned pixels where the reference's OWN fp32 arithmetic (the faithful fp32
        restatement, ref32) is further than that from exact arithmetic, the HIP path must be no worse than
        ref32_factor times the reference's own deviation (pass ref32_disps=None to switch the allowance off).
    In addition at least `frac` of the non-cancelling pixels (|d_ref| >= 0.1 |d_old|) must meet the pure relative
    bound d_rtol * |d_ref|.
    Pose criterion: t_tol / r_tol; with ref32_poses (the fp32-faithful oracle's poses) the bound of each is widened to
    ref32_factor x the fp32 oracle's OWN distance from the arbiter where that is larger -- and that distance is recorded,
    so the report shows whether "the reference's fp32 arithmetic is no better" is true of the poses too.
    log32_disps / log32_poses: the fp32-faithful oracle's state FOR THE REPORT ONLY -- north_star's comparator is the
    reference's fp32 CUDA path, so |device - ref32| is logged next to the two distances from the arbiter; nothing is widened.
    Every call appends its measured worst cases to gpurun_out/parity_report.jsonl."""
    poses, ref_poses = np.asarray(poses, np.float64), np.asarray(ref_poses, np.float64)
    dt = np.abs(poses[:, :3] - ref_poses[:, :3]).max()
    dr = quat_angle(poses[:, 3:], ref_poses[:, 3:]).max()
    dt32 = dr32 = None
    if ref32_poses is not None:
        p32 = np.asarray(ref32_poses, np.float64)
        dt32 = float(np.abs(p32[:, :3] - ref_poses[:, :3]).max())
        dr32 = float(quat_angle(p32[:, 3:], ref_poses[:, 3:]).max())
        t_tol, r_tol = max(t_tol, ref32_factor * dt32), max(r_tol, ref32_factor * dr32)
    d, r, o = np.asarray(disps, np.float64), np.asarray(ref_disps, np.float64), np.asarray(old_disps, np.float64)
    err = np.abs(d - r)
    scale = np.maximum(np.abs(r), np.abs(o))
    base = d_rtol * scale
    allowed = base
    n_allow = 0
    ref32_worst = None
    if ref32_disps is not None:
        dev32 = np.abs(np.asarray(ref32_disps, np.float64) - r)
        allowed = np.maximum(base, ref32_factor * dev32)
        n_allow = int((err > base).sum())             # pixels that needed the allowance
        ref32_worst = float((dev32 / np.maximum(scale, 1e-12)).max())
    worst = (err / np.maximum(allowed, 1e-300)).max()
    solid = np.abs(r) >= 0.1 * np.abs(o)
    pure_err = err[solid] / np.abs(r[solid]) if solid.any() else np.zeros(1)
    pure = (pure_err <= d_rtol).mean()
    rec = dict(dt_m=float(dt), dr_rad=float(dr), depth_max_err_over_scale=float((err / np.maximum(scale, 1e-12)).max()),
               depth_max_err_over_dref_noncancelling=float(pure_err.max()), depth_p999_err_over_dref=float(
                   np.quantile(pure_err, 0.999)), frac_pure_rel_ok=float(pure), pixels=int(err.size),
               pixels_needing_ref32_allowance=n_allow, ref32_own_max_dev_over_scale=ref32_worst,
               ref32_own_dt_m=dt32, ref32_own_dr_rad=dr32,
               tol=dict(t=t_tol, r=r_tol, d_rtol=d_rtol, frac=frac,
                        ref32_factor=ref32_factor if ref32_disps is not None else None))
    # device against the fp32-faithful oracle directly (what north_star names: the reference's fp32 path)
    l32d = log32_disps if log32_disps is not None else ref32_disps
    l32p = log32_poses if log32_poses is not None else ref32_poses
    if l32d is not None:
        q = np.asarray(l32d, np.float64)
        e32 = np.abs(d - q)
        sc32 = np.maximum(np.abs(q), np.abs(o))
        sol32 = np.abs(q) >= 0.1 * np.abs(o)
        rec["device_vs_ref32_depth_max_err_over_scale"] = float((e32 / np.maximum(sc32, 1e-12)).max())
        rec["device_vs_ref32_depth_max_err_over_dref32_noncancelling"] = float(
            (e32[sol32] / np.abs(q[sol32])).max()) if sol32.any() else 0.0
        rec["ref32_own_depth_max_dev_over_scale"] = float((np.abs(q - r) / np.maximum(scale, 1e-12)).max())
        rec["ref32_own_depth_max_dev_over_dref_noncancelling"] = float(
            (np.abs(q - r)[solid] / np.abs(r[solid])).max()) if solid.any() else 0.0
    if l32p is not None:
        p32 = np.asarray(l32p, np.float64)
        rec["device_vs_ref32_dt_m"] = float(np.abs(poses[:, :3] - p32[:, :3]).max())
        rec["device_vs_ref32_dr_rad"] = float(quat_angle(poses[:, 3:], p32[:, 3:]).max())
        if rec.get("ref32_own_dt_m") is None:
            rec["ref32_own_dt_m"] = float(np.abs(p32[:, :3] - ref_poses[:, :3]).max())
            rec["ref32_own_dr_rad"] = float(quat_angle(p32[:, 3:], ref_poses[:, 3:]).max())
    # where the worst pixel is and what kind of pixel it is (a depth the update shrinks, a far point ...)
    wi = int(np.argmax(err / np.maximum(scale, 1e-12)))
    rec["worst_pixel"] = dict(index=[int(v) for v in np.unravel_index(wi, err.shape)], d_ref=float(r.flat[wi]),
                              d_old=float(o.flat[wi]), err=float(err.flat[wi]),
                              ref32_err=(float(np.abs(np.asarray(l32d, np.float64) - r).flat[wi]) if l32d is not None else None))
    wp = int(np.argmax(np.where(solid, err / np.maximum(np.abs(r), 1e-300), 0.0)))      # ... and on the |d_ref| scale
    rec["worst_pixel_over_dref"] = dict(index=[int(v) for v in np.unravel_index(wp, err.shape)], d_ref=float(r.flat[wp]),
                                        d_old=float(o.flat[wp]), err_over_dref=float(err.flat[wp] / max(abs(r.flat[wp]), 1e-300)))
    _record(rec)
    msg = ("dt=%.3e m dr=%.3e rad depth max(err/allowed)=%.3f max(err/scale)=%.3e max(err/|d_ref|)=%.3e "
           "pure-rel frac=%.6f ref32-allowance pixels=%d/%d" % (
               dt, dr, worst, rec["depth_max_err_over_scale"], rec["depth_max_err_over_dref_noncancelling"], pure,
               n_allow, err.size))
    assert dt <= t_tol, msg
    assert dr <= r_tol, msg
    assert worst <= 1.0, msg
    assert pure >= frac, msg
    return msg
