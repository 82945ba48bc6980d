"""GPU parity: droid_backends.ba / BACore (HIP, through the C ABI) vs the CPU oracle on identical inputs.
Tolerances are the ones BASELINE.json's north_star states: inverse depths 1e-4 relative, poses 1e-5 m /
1e-6 rad."""
import os

import numpy as np
import pytest
import torch

from dbaf_amd import synthetic as syn
from util import to_dev, check_state

pytestmark = pytest.mark.gpu


def _oracle():
    from oracle import oracle as orc
    return orc


def _run_gpu_ba(W, itrs=2, motion_only=False, lm=None, ep=None):
    import droid_backends
    d = to_dev(W)
    dx, dz = droid_backends.ba(d["poses"], d["disps"], d["intrinsics"], d["disps_sens"], d["target"], d["weight"],
                               d["eta"], d["ii"], d["jj"], W.t0, W.t1, itrs, W.lm if lm is None else lm,
                               W.ep if ep is None else ep, motion_only)
    torch.cuda.synchronize()
    return d["poses"].cpu().numpy(), d["disps"].cpu().numpy(), dx.cpu().numpy(), None if dz is None else dz.cpu().numpy()


WINDOWS = {
    "tiny_a": lambda: syn.window_tiny_a(11),
    "tiny_b_stereo_fixed_sensor": lambda: syn.window_tiny_b(12),
    "kitti_shape_8kf": lambda: syn.make_window(*syn.graph_banded(8, 2), 8, 28, 107, seed=13,
                                               intr=(69.0, 69.5, 53.2, 14.1)),
    "25kf_96edges_64x64": lambda: syn.window_25_96(0),
    "kitti360_32kf_122edges_28x107": lambda: syn.window_32_122(0),
    "64kf_512edges_64x64": lambda: syn.window_64_512(0),  # n = 378: solver runs from the global scratch
    # BASELINE configs[0]: TUM-VI corridor at the demo's resolution (55x55 maps), max_factors = 36
    "tumvi_corridor_9kf_36edges_55x55": lambda: syn.make_window(*syn.graph_banded(9, 2, extra=[(0, 3), (1, 4), (2, 5)]),
                                                                 9, 55, 55, seed=14, intr=(20.5, 20.5, 27.4, 27.6)),
    # BASELINE configs[4] map shape (WHU, 48x64) with depth measurements on a quarter of the pixels
    "whu_10kf_48x64_sensor_depth": lambda: syn.make_window(*syn.graph_banded(10, 3), 10, 48, 64, seed=15,
                                                           intr=(30.0, 30.0, 31.5, 23.7), sensor_frac=0.25),
}


STRICT = ("25kf_96edges_64x64", "kitti360_32kf_122edges_28x107", "64kf_512edges_64x64", "tumvi_corridor_9kf_36edges_55x55",
          "whu_10kf_48x64_sensor_depth")


@pytest.mark.parametrize("name", list(WINDOWS))
def test_ba_matches_oracle(name):
    orc = _oracle()
    W = WINDOWS[name]()
    args = (W.poses, W.disps, W.intrinsics, W.disps_sens, W.target, W.weight, W.eta, W.ii, W.jj, W.t0, W.t1, 2,
            W.lm, W.ep, False, 0.05)
    r32 = orc.ba(*args, np.float32)
    r64 = orc.ba(*args, np.float64)
    poses, disps, dx, dz = _run_gpu_ba(W)
    assert dz.shape == r32["dz"].shape
    # a14: DepthVideo.ba clamps all disps after the call (depth_video.py:560)
    clamp = lambda a: np.maximum(a, 0.001)  # noqa: E731
    if name in STRICT:
        # the windows of BASELINE.json's configs: the tolerance as util.check_depths_every_pixel states it -- comparator = the
        # float64 arbiter; EVERY pixel within 1e-4 of max(|d_new|, |d_old|), and within 1e-4 of |d_new| alone except where the
        # reference's own fp32 arithmetic sits at the same noise floor (then: no more than 25 % further from the arbiter than
        # it) -- no fraction of pixels is exempt (round 5 kept frac = 0.9999 for one pixel of the 25-KF window at 1.05e-4 of
        # |d_new|, where the fp32-faithful oracle is at 0.91e-4).  Poses: 1e-5 m / 1e-6 rad, no allowance.
        from util import check_depths_every_pixel
        w1, w2, needed = check_depths_every_pixel(clamp(disps), clamp(r64["disps"]), W.disps, clamp(r32["disps"]))
        msg = check_state(poses, clamp(disps), r64["poses"], clamp(r64["disps"]), W.disps, ref32_disps=None, frac=0.0,
                          log32_disps=clamp(r32["disps"]), log32_poses=r32["poses"])
        msg += " | every pixel: %.3f of the max-scale bound, %.3f of the |d_new| / noise-floor bound (%d pixels beyond 1e-4 |d_new|)" % (
            w1, w2, needed)
    else:
        # small fixtures (16x16, 24x32, 28x107 maps, 4-8 keyframes): the fp32-faithful oracle is itself 0.9e-4 / 2.0e-4
        # (tiny_a / KITTI shape) from the arbiter at its worst pixel; bound 1.5e-4 or 2 x the oracle's own deviation
        msg = check_state(poses, clamp(disps), r64["poses"], clamp(r64["disps"]), W.disps,
                          ref32_disps=clamp(r32["disps"]), d_rtol=1.5e-4, frac=0.99)
    # (the fp32-faithful oracle's own pose deviation goes into the report next to the device's; it widens nothing here)
    from util import quat_angle, _record
    _record(dict(kind="oracle32_pose_deviation", window=name,
                 ref32_own_dt_m=float(np.abs(r32["poses"][:, :3].astype(np.float64) - r64["poses"][:, :3]).max()),
                 ref32_own_dr_rad=float(quat_angle(r32["poses"][:, 3:].astype(np.float64), r64["poses"][:, 3:]).max()),
                 device_dt_m=float(np.abs(poses[:, :3].astype(np.float64) - r64["poses"][:, :3]).max()),
                 device_dr_rad=float(quat_angle(poses[:, 3:].astype(np.float64), r64["poses"][:, 3:]).max())))
    print(name, "vs fp64 arbiter:", msg)
    np.testing.assert_allclose(dx, r64["dx"], rtol=1e-3, atol=2e-6)


@pytest.mark.parametrize("name", ["tiny_a", "25kf_96edges_64x64", "whu_10kf_48x64_sensor_depth"])
@pytest.mark.parametrize("itrs,motion_only", [(2, False), (1, False), (3, False), (2, True)])
def test_ba_clamped_is_ba_followed_by_the_callers_clamp(name, itrs, motion_only):
    """droid_backends.ba_clamped (dba_ba_run with disp_floor > 0): DepthVideo.ba's two statements -- droid_backends.ba(...)
    and self.disps.clamp_(min=0.001) (dbaf/depth_video.py:559-560) -- in one call, the clamp riding in the last launch.
    Same poses, inverse depths, dx and dz bit for bit as the two statements; the floor is chosen so that a good part of the
    pixels IS clamped (the default 0.001 rarely bites on a synthetic window)."""
    import droid_backends
    W = WINDOWS[name]()
    a, b = to_dev(W), to_dev(W)
    args = lambda d: (d["poses"], d["disps"], d["intrinsics"], d["disps_sens"], d["target"], d["weight"], d["eta"],  # noqa: E731
                      d["ii"], d["jj"], W.t0, W.t1, itrs, W.lm, W.ep, motion_only)
    # the tracker's invariant: every frame is at or above the floor before the call (each was clamped after the call that
    # updated it).  With the floor at the median, half of the pixels start ON it and the update pushes part of them below.
    floor = float(np.median(W.disps))
    for d in (a, b):
        d["disps"].clamp_(min=floor)
    dxa, dza = droid_backends.ba(*args(a))
    below = float((a["disps"] < floor).float().mean())
    a["disps"].clamp_(min=floor)
    dxb, dzb = droid_backends.ba_clamped(*args(b), disp_floor=floor)
    torch.cuda.synchronize()
    assert torch.equal(a["poses"], b["poses"]) and torch.equal(dxa, dxb)
    assert torch.equal(a["disps"], b["disps"])
    if motion_only:
        assert dza is None and dzb is None and below == 0.0
    else:
        assert torch.equal(dza, dzb)
        assert below > 0.02, below      # the clamp did something
    with pytest.raises(RuntimeError):
        droid_backends.ba_clamped(*args(b), disp_floor=0.0)


@pytest.mark.parametrize("motion_only", [False, True])
def test_ba_clamped_floors_the_whole_buffer_like_the_callers_clamp(motion_only):
    """`self.disps.clamp_(min=0.001)` is over the WHOLE buffer, and the reference's caller rescales inverse depths between BA
    calls (dbaf_frontend.py:570,814 `disps[i] /= s`), so frames outside kx -- and every frame of a motion_only call -- can sit
    below the floor when ba is entered: ba_clamped floors them in its last launch too (NaNs stay NaNs, as with clamp_)"""
    import droid_backends
    W = syn.make_window(*syn.graph_banded(6, 2), 6, 12, 16, seed=5, buffer=11)     # frames 6..10 of the buffer are in no edge
    assert W.B > W.M
    a, b = to_dev(W), to_dev(W)
    floor = float(np.median(W.disps))
    for d in (a, b):
        d["disps"][W.B - 2] *= 0.01            # a frame outside kx far below the floor
        d["disps"][W.B - 1, 0, 0] = float("nan")
        d["disps"][1] *= 0.5                   # ... and one inside
    args = lambda d: (d["poses"], d["disps"], d["intrinsics"], d["disps_sens"], d["target"], d["weight"], d["eta"],  # noqa: E731
                      d["ii"], d["jj"], W.t0, W.t1, 2, W.lm, W.ep, motion_only)
    droid_backends.ba(*args(a))
    a["disps"].clamp_(min=floor)
    droid_backends.ba_clamped(*args(b), disp_floor=floor)
    torch.cuda.synchronize()
    assert torch.equal(a["poses"], b["poses"])
    assert torch.equal(a["disps"].view(torch.int32), b["disps"].view(torch.int32))     # (bit patterns: the NaN included)
    assert torch.isnan(b["disps"][W.B - 1, 0, 0]) and float(b["disps"][W.B - 2].min()) == floor


def _bacore_system(W, form):
    """the Schur-reduced camera system of W as BACore.hessian hands it to the host, with the given Schur kernel form"""
    import droid_backends
    from dbaf_amd import _lib
    lib = _lib.load()
    assert lib.dba_ba_schur_select(form) == 0
    try:
        d = to_dev(W)
        core = droid_backends.BACore()
        core.init(d["poses"], d["disps"], d["intrinsics"], d["disps_sens"], d["target"], d["weight"], d["eta"], d["ii"],
                  d["jj"], W.t0, W.t1, 2, W.lm, W.ep, False)
        n = 6 * (W.t1 - W.t0)
        H, v = torch.zeros(n, n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
        core.hessian(H, v)
        return H.numpy().copy(), v.numpy().copy()
    finally:
        lib.dba_ba_schur_select(0)


def _dense_graph(num_kf, radius):
    return syn.graph_banded(num_kf, radius)


SCHUR_WINDOWS = dict(WINDOWS)
SCHUR_WINDOWS.update({
    # frames with up to 17 rows: more than the Gram tiles hold (13), the row-pair path inside the per-frame kernel
    "dense_16kf_radius8_24x32": lambda: syn.make_window(*_dense_graph(16, 8), 16, 24, 32, seed=61,
                                                         intr=(12.0, 11.5, 15.6, 12.2)),
    # every tile count 1..5 in one window (out-degrees 0 .. 12), duplicate edges, an odd pixel count (15 x 17)
    "ragged_degrees_15x17": lambda: syn.make_window(
        np.array([1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 5, 5, 0], np.int64),
        np.array([2, 2, 1, 3, 4, 5, 1, 2, 4, 5, 6, 7, 0, 1, 2, 3, 5, 6, 7, 8, 0, 1, 2, 3, 5, 4, 6, 1], np.int64),
        9, 15, 17, seed=62, intr=(8.0, 8.5, 8.2, 7.1)),
})


@pytest.mark.parametrize("name", list(SCHUR_WINDOWS))
def test_schur_kernel_forms_agree_on_the_reduced_system(name):
    """The per-source-frame kernel (float64 Gram tiles on the matrix cores, csrc/ba_kernels.hip) against the (row, partner)
    grid: the same H, v up to the fp32 rounding of the latter's products (the former's are exact), symmetric, and the
    frame form reproducible bit for bit (its sums are ordered; float64 atomics of fp32-exact addends)."""
    W = SCHUR_WINDOWS[name]()
    Hr, vr = _bacore_system(W, 1)
    Hf, vf = _bacore_system(W, 2)
    Hf2, vf2 = _bacore_system(W, 2)
    scale = np.abs(Hr).max()
    assert scale > 0 and np.isfinite(Hf).all() and np.isfinite(vf).all()
    # blockwise scale: an entry is a difference A - E Q E^T of two sums; compare against the larger block norm
    assert np.abs(Hf - Hr).max() <= 3e-6 * scale, np.abs(Hf - Hr).max() / scale
    assert np.abs(vf - vr).max() <= 3e-6 * max(np.abs(vr).max(), scale), np.abs(vf - vr).max()
    assert np.abs(Hf - Hf.T).max() == 0.0 and np.abs(Hr - Hr.T).max() == 0.0   # mirrored from the lower triangle
    assert np.abs(Hf - Hf2).max() <= 1e-9 * scale and np.abs(vf - vf2).max() <= 1e-9 * scale


@pytest.mark.parametrize("name", STRICT)
def test_ba_matches_oracle_with_the_per_frame_schur_kernel(name):
    """the windows of BASELINE.json's configs through the per-source-frame Schur kernel whatever the automatic choice is
    (the 64-KF window takes it anyway), at the north-star tolerance with no allowance"""
    from dbaf_amd import _lib
    orc = _oracle()
    W = WINDOWS[name]()
    r64 = orc.ba(W.poses, W.disps, W.intrinsics, W.disps_sens, W.target, W.weight, W.eta, W.ii, W.jj, W.t0, W.t1, 2,
                 W.lm, W.ep, False, 0.05, np.float64)
    lib = _lib.load()
    lib.dba_ba_schur_select(2)
    try:
        poses, disps, dx, dz = _run_gpu_ba(W)
    finally:
        lib.dba_ba_schur_select(0)
    clamp = lambda a: np.maximum(a, 0.001)  # noqa: E731
    r32 = orc.ba(W.poses, W.disps, W.intrinsics, W.disps_sens, W.target, W.weight, W.eta, W.ii, W.jj, W.t0, W.t1, 2,
                 W.lm, W.ep, False, 0.05, np.float32)
    from util import check_depths_every_pixel
    check_depths_every_pixel(clamp(disps), clamp(r64["disps"]), W.disps, clamp(r32["disps"]))
    check_state(poses, clamp(disps), r64["poses"], clamp(r64["disps"]), W.disps, ref32_disps=None, frac=0.0)


def test_ba_full_size_properties_64kf():
    """BASELINE.json configs[3] size (64 KF / 512 edges / 64x64), size-independent properties:
    (i) a noise-free window is a fixed point; (ii) Gauss-Newton contracts towards the ground truth;
    (iii) the run is reproducible (f64 atomics only reorder sums at the 1e-16 level)."""
    ii, jj = syn.graph_64_512()
    W0 = syn.make_window(ii, jj, 64, 64, 64, seed=3, pose_noise=0.0, disp_noise=0.0, target_noise=0.0)
    poses, disps, dx, dz = _run_gpu_ba(W0)
    assert np.abs(dx).max() < 5e-5 and np.abs(poses - W0.poses).max() < 5e-5
    W = syn.make_window(ii, jj, 64, 64, 64, seed=3, pose_noise=0.01, disp_noise=0.05, target_noise=0.0)
    e0p = np.abs(W.poses - W.poses_gt)[:64].max()
    e0d = np.abs(W.disps - W.disps_gt)[:64].mean()
    p1, d1, _, _ = _run_gpu_ba(W, itrs=4)
    p2, d2, _, _ = _run_gpu_ba(W, itrs=4)
    assert np.abs(p1 - W.poses_gt)[:64].max() < 0.2 * e0p
    assert np.abs(d1 - W.disps_gt)[:64].mean() < 0.2 * e0d
    assert np.abs(p1 - p2).max() < 1e-6 and np.abs(d1 - d2).max() < 1e-4


def test_ba_single_iteration_and_dz():
    orc = _oracle()
    W = syn.window_tiny_b(21)
    r64 = orc.ba(W.poses, W.disps, W.intrinsics, W.disps_sens, W.target, W.weight, W.eta, W.ii, W.jj, W.t0, W.t1, 1,
                 W.lm, W.ep, False, 0.05, np.float64)
    poses, disps, dx, dz = _run_gpu_ba(W, itrs=1)
    check_state(poses, disps, r64["poses"], r64["disps"], W.disps)
    scale = np.maximum(np.abs(r64["dz"]), np.abs(W.disps[W.kx].reshape(W.M, -1)))
    assert (np.abs(dz - r64["dz"]) <= 1e-4 * scale + 1e-7).all()


def test_ba_motion_only():
    orc = _oracle()
    W = syn.window_tiny_b(31)
    r64 = orc.ba(W.poses, W.disps, W.intrinsics, W.disps_sens, W.target, W.weight, W.eta, W.ii, W.jj, W.t0, W.t1, 2,
                 W.lm, W.ep, True, 0.05, np.float64)
    poses, disps, dx, dz = _run_gpu_ba(W, motion_only=True)
    assert dz is None
    assert np.array_equal(disps, W.disps)  # depth untouched
    check_state(poses, disps, r64["poses"], r64["disps"], W.disps)


def test_ba_zero_residual_is_fixed_point():
    """target = reprojection of the state, no noise => dx = dz = 0 up to rounding (SURVEY 8(c) KAT)."""
    W = syn.make_window(*syn.graph_banded(5, 2), 5, 24, 32, seed=41, intr=(12.0, 11.5, 15.6, 12.2),
                        pose_noise=0.0, disp_noise=0.0, target_noise=0.0)
    poses, disps, dx, dz = _run_gpu_ba(W)
    assert np.abs(dx).max() < 2e-5
    assert np.abs(poses - W.poses).max() < 2e-5
    assert np.abs(disps - W.disps)[:5].max() < 2e-3


def test_ba_cholesky_failure_gives_zero_pose_update():
    """non-SPD system: the reference returns dx = 0 (droid_kernels.cu:1263-1266) and still back-substitutes"""
    orc = _oracle()
    W = syn.window_tiny_a(51)
    r64 = orc.ba(W.poses, W.disps, W.intrinsics, W.disps_sens, W.target, W.weight, W.eta, W.ii, W.jj, W.t0, W.t1, 1,
                 W.lm, -1.0e6, False, 0.05, np.float64)
    assert not r64["ok"]
    poses, disps, dx, dz = _run_gpu_ba(W, itrs=1, ep=-1.0e6)
    assert np.array_equal(dx, np.zeros_like(dx))
    np.testing.assert_allclose(poses, W.poses, rtol=0, atol=1e-7)
    check_state(poses, disps, r64["poses"], r64["disps"], W.disps)


def test_bacore_hessian_and_retract():
    import droid_backends
    orc = _oracle()
    W = syn.window_tiny_b(61)
    oc = orc.BACore(W.poses, W.disps, W.intrinsics, W.disps_sens, W.target, W.weight, W.eta, W.ii, W.jj, W.t0, W.t1,
                    W.lm, W.ep, np.float64)
    Ho, vo = oc.hessian()
    d = to_dev(W)
    core = droid_backends.BACore()
    core.init(d["poses"], d["disps"], d["intrinsics"], d["disps_sens"], d["target"], d["weight"], d["eta"], d["ii"],
              d["jj"], W.t0, W.t1, 2, W.lm, W.ep, False)
    n = 6 * (W.t1 - W.t0)
    H = torch.zeros(n, n, dtype=torch.float64)
    v = torch.zeros(n, dtype=torch.float64)
    core.hessian(H, v)
    np.testing.assert_allclose(H.numpy(), Ho, rtol=0, atol=2e-5 * np.abs(Ho).max())
    np.testing.assert_allclose(v.numpy(), vo, rtol=0, atol=2e-5 * np.abs(vo).max())
    # external solve (stands in for GTSAM, depth_video.py:528-557), then retract on both sides
    L = Ho + np.diag(W.ep + W.lm * np.diag(Ho))
    dx = np.linalg.solve(L, vo)
    dxo, dzo = oc.retract(dx)
    dxg, dzg = core.retract(torch.from_numpy(dx))
    torch.cuda.synchronize()
    check_state(d["poses"].cpu().numpy(), d["disps"].cpu().numpy(), oc.poses, oc.disps, W.disps)
    assert dzg.shape == dzo.shape
    # optimize(): damped dense solve of a caller-supplied system
    core.optimize(torch.from_numpy(Ho), torch.from_numpy(vo))
    np.testing.assert_allclose(core.dx.cpu().numpy().reshape(-1), dx, rtol=1e-4, atol=1e-7)
    del core


def _compare_with_oracle(W, itrs=2, **kw):
    orc = _oracle()
    args = (W.poses, W.disps, W.intrinsics, W.disps_sens, W.target, W.weight, W.eta, W.ii, W.jj, W.t0, W.t1, itrs,
            W.lm, W.ep, False, 0.05)
    r32, r64 = orc.ba(*args, np.float32), orc.ba(*args, np.float64)
    poses, disps, dx, dz = _run_gpu_ba(W, itrs=itrs)
    assert dz.shape == r64["dz"].shape
    return check_state(poses, disps, r64["poses"], r64["disps"], W.disps, ref32_disps=r32["disps"], **kw)


def test_ba_duplicate_edges_and_edgeless_window_frame():
    """collisions: the same (i, j) twice (the reference sums both, SparseBlock::update_lhs :1186-1200);
    a frame inside [t0, t1) with no edge at all keeps its pose (zero block + damping) and its depths."""
    ii = np.array([0, 1, 1, 2, 2, 1, 0, 2, 4, 2], np.int64)
    jj = np.array([1, 0, 2, 1, 1, 2, 2, 0, 2, 4], np.int64)  # (2,1) and (1,2) twice; frame 3 has no edge
    W = syn.make_window(ii, jj, 5, 16, 24, seed=81, intr=(9.0, 9.3, 11.6, 7.9))
    assert 3 in W.kx and 3 not in ii and 3 not in jj
    print(_compare_with_oracle(W))
    poses, disps, dx, dz = _run_gpu_ba(W)
    np.testing.assert_allclose(poses[3], W.poses[3], atol=1e-6)  # untouched by any residual
    assert np.array_equal(disps[3], W.disps[3])


def test_ba_source_frames_outside_the_window_and_later_t0():
    """t0 = 3: frames 0..2 are fixed (their pose blocks are dropped, :1191) but their depths still move"""
    W = syn.make_window(*syn.graph_banded(7, 2), 7, 16, 16, seed=82, intr=(6.0, 6.0, 7.7, 8.1), t0=3)
    assert W.t0 == 3 and W.M == 7
    print(_compare_with_oracle(W))
    poses, disps, _, _ = _run_gpu_ba(W)
    assert np.array_equal(poses[:3], W.poses[:3]) and not np.array_equal(disps[1], W.disps[1])


def test_ba_without_edges_is_a_noop():
    W = syn.make_window(np.zeros(0, np.int64), np.zeros(0, np.int64), 4, 8, 16, seed=83, intr=(4.0, 4.0, 7.5, 3.5))
    poses, disps, dx, dz = _run_gpu_ba(W)
    assert np.array_equal(dx, np.zeros_like(dx))
    np.testing.assert_allclose(poses, W.poses, atol=1e-7)
    assert np.array_equal(disps, W.disps)


def test_ba_eta_broadcast_row():
    """eta given as a single [1, h, w] row (view(-1, HW) broadcast in the reference, :1476)"""
    orc = _oracle()
    W = syn.window_tiny_a(84)
    W.eta = W.eta[:1].copy()
    r64 = orc.ba(W.poses, W.disps, W.intrinsics, W.disps_sens, W.target, W.weight, W.eta, W.ii, W.jj, W.t0, W.t1, 2,
                 W.lm, W.ep, False, 0.05, np.float64)
    poses, disps, dx, dz = _run_gpu_ba(W)
    assert dz.shape == (W.M, W.h * W.w)
    check_state(poses, disps, r64["poses"], r64["disps"], W.disps)


def test_ba_rejects_an_eta_with_the_wrong_number_of_rows():
    """1 < rows != |kx|: the reference's eta.view(-1, HW) fails to broadcast against C (:1476) before anything is touched.  |kx|
    only exists on the device: stage 0 compares, turns the rest of the call into a no-op on the device (state untouched, zero
    dx / dz: round 6) and records the mismatch in the workspace's pinned words; the module raises at the next call on that
    workspace (or at check_async_errors() after a synchronisation) -- the call itself never stops the host.  The graph is the
    same in every call here, so the check also runs on the early-out path."""
    import droid_backends
    W = syn.window_tiny_b(85)
    d = to_dev(W)
    assert W.M > 2
    torch.cuda.synchronize()
    droid_backends.check_async_errors()
    args = lambda eta: (d["poses"], d["disps"], d["intrinsics"], d["disps_sens"], d["target"], d["weight"], eta, d["ii"],  # noqa: E731
                        d["jj"], W.t0, W.t1, 2, W.lm, W.ep, False)
    for rows in (2, W.M - 1, W.M + 1):
        eta = torch.full((rows, W.h, W.w), 3e-7, device="cuda")
        if rows > (W.t1 - W.t0) + W.N:          # more rows than kx can have entries: known without the device
            with pytest.raises(RuntimeError):
                droid_backends.ba(*args(eta))
            continue
        p_before, d_before = d["poses"].clone(), d["disps"].clone()
        dx_bad, dz_bad = droid_backends.ba(*args(eta))   # (asynchronous: returns -- and changes nothing, like the reference's raise)
        torch.cuda.synchronize()
        assert torch.equal(d["poses"], p_before) and torch.equal(d["disps"], d_before)
        assert not dx_bad.any() and not dz_bad.any()
        with pytest.raises(RuntimeError, match="eta with %d rows.*= %d rows" % (rows, W.M)):
            droid_backends.check_async_errors()
        droid_backends.check_async_errors()      # reported once
        droid_backends.ba(*args(eta))
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError, match="earlier"):     # ... or by whatever call comes next
            droid_backends.ba(*args(d["eta"]))
    p0 = d["poses"].clone()
    droid_backends.ba(*args(d["eta"]))
    torch.cuda.synchronize()
    droid_backends.check_async_errors()
    assert not torch.equal(d["poses"], p0)


def test_ba_never_synchronises_the_host():
    """droid_backends.ba on NEW edge tensors in every call (what CovisibleGraph.update hands over with use_inactive=True:
    torch.cat creates them, covisible_graph.py:242-247) must not stop the host: torch's sync debug mode raises on any
    synchronising torch call inside, and the library's own calls are checked by timing the host against a long kernel
    queued in front"""
    import time
    import droid_backends
    W = syn.window_25_96(3)
    d = to_dev(W)
    n_in = W.N // 3
    ii_in, jj_in, ii_ac, jj_ac = d["ii"][:n_in].clone(), d["jj"][:n_in].clone(), d["ii"][n_in:].clone(), d["jj"][n_in:].clone()

    def call():
        ii = torch.cat([ii_in, ii_ac], 0)       # new objects, same contents
        jj = torch.cat([jj_in, jj_ac], 0)
        return droid_backends.ba(d["poses"], d["disps"], d["intrinsics"], d["disps_sens"], d["target"], d["weight"], d["eta"],
                                 ii, jj, W.t0, W.t1, 2, W.lm, W.ep, False)

    call()
    torch.cuda.synchronize()
    big = torch.empty(1 << 28, device="cuda")    # 1 GiB fills: ~10 ms of queued work in front of the calls
    torch.cuda.set_sync_debug_mode("error")
    try:
        for _ in range(40):
            big.fill_(1.0)
        t = time.perf_counter()
        for _ in range(5):
            call()
        host = time.perf_counter() - t
    finally:
        torch.cuda.set_sync_debug_mode("default")
    t = time.perf_counter()
    torch.cuda.synchronize()
    drained = time.perf_counter() - t
    assert drained > 2 * host, "the host waited for the device inside droid_backends.ba (%.1f ms host, %.1f ms left)" % (
        1e3 * host, 1e3 * drained)


def test_ba_stage0_recognises_the_graph_by_its_contents():
    """the key stage 0 leaves in the workspace: new tensor objects with the same edges leave stage 0 at once, a changed edge,
    another window, another Schur form or a re-ordered list rebuild -- every call gives what a cold call gives, and the
    skip is visible in the workspace (meta[7], which only a rebuild resets)"""
    import ctypes
    import droid_backends
    from droid_backends import _BA_WS
    from dbaf_amd import _lib
    W = syn.window_tiny_b(83)

    def cold(Wx):
        saved = _BA_WS.enabled
        _BA_WS.enabled = False
        try:
            return _run_gpu_ba(Wx)
        finally:
            _BA_WS.enabled = saved

    def warm(Wx):
        dd = to_dev(Wx)                            # new tensors for everything, ii / jj included
        droid_backends.ba(dd["poses"], dd["disps"], dd["intrinsics"], dd["disps_sens"], dd["target"], dd["weight"], dd["eta"],
                          dd["ii"], dd["jj"], Wx.t0, Wx.t1, 2, Wx.lm, Wx.ep, False)
        return dd["poses"].cpu().numpy(), dd["disps"].cpu().numpy()

    def meta7():
        dims = (W.N, W.B, W.h, W.w, W.t0, W.t1)
        key = [k for k in _BA_WS.ws if k[-1] == dims and k[0] != "bacore"][0]
        lay = _lib.BaLayout()
        _lib.load().dba_ba_get_layout(*dims, ctypes.byref(lay))
        return _BA_WS.ws[key][0][lay.meta + 28:lay.meta + 32].view(torch.int32)

    ref = cold(W)
    for rep in range(3):
        p, z = warm(W)
        assert np.array_equal(p, ref[0]) and np.array_equal(z, ref[1]), rep
        if rep == 0:
            meta7().fill_(77)                      # a mark a rebuild would erase
        else:
            assert int(meta7().item()) == 77, "stage 0 rebuilt the tables of an unchanged graph"
    W2 = syn.window_tiny_b(83)
    W2.jj = W2.jj.copy()
    W2.jj[3] = 5 if W2.jj[3] != 5 else 4           # one edge re-targeted: same shapes, same workspace
    ref2 = cold(W2)
    p, z = warm(W2)
    assert np.array_equal(p, ref2[0]) and np.array_equal(z, ref2[1])
    assert int(meta7().item()) != 77, "stage 0 kept the tables of another graph"
    p, z = warm(W)                                 # ... and back
    assert np.array_equal(p, ref[0]) and np.array_equal(z, ref[1])
    W3 = syn.window_tiny_b(83)                     # the same edges in another order: other tables (edge ids), a rebuild
    perm = np.arange(W.N)[::-1].copy()
    for name in ("ii", "jj", "target", "weight"):
        setattr(W3, name, np.ascontiguousarray(getattr(W, name)[perm]))
    ref3 = cold(W3)
    meta7().fill_(77)
    p, z = warm(W3)
    assert int(meta7().item()) != 77
    np.testing.assert_allclose(p, ref3[0], rtol=0, atol=0)
    assert np.array_equal(z, ref3[1])
    lib = _lib.load()                               # another Schur form needs other tables
    try:
        p, z = warm(W)
        meta7().fill_(77)
        lib.dba_ba_schur_select(2)
        refs = cold(W)
        p, z = warm(W)
        assert int(meta7().item()) != 77
        assert np.array_equal(p, refs[0]) and np.array_equal(z, refs[1])
    finally:
        lib.dba_ba_schur_select(0)


def test_ba_rejects_cpu_and_noncontiguous():
    import droid_backends
    W = syn.window_tiny_a(71)
    d = to_dev(W)
    with pytest.raises(RuntimeError):
        droid_backends.ba(d["poses"].cpu(), d["disps"], d["intrinsics"], d["disps_sens"], d["target"], d["weight"],
                          d["eta"], d["ii"], d["jj"], W.t0, W.t1, 2, W.lm, W.ep, False)
    with pytest.raises(RuntimeError):
        droid_backends.ba(d["poses"], d["disps"], d["intrinsics"], d["disps_sens"], d["target"].transpose(2, 3),
                          d["weight"], d["eta"], d["ii"], d["jj"], W.t0, W.t1, 2, W.lm, W.ep, False)


def test_ba_three_and_four_iterations_match_the_oracle():
    """iterations > 2: the retracted window ping-pongs between the two workspace copies while back-substitution +
    retraction ride in the next linearisation (csrc/ba_host.hip: ba_run); motion_only folds the pose part alone"""
    orc = _oracle()
    for itrs, motion_only in ((3, False), (4, False), (3, True)):
        W = syn.window_tiny_b(70 + itrs)
        r64 = orc.ba(W.poses, W.disps, W.intrinsics, W.disps_sens, W.target, W.weight, W.eta, W.ii, W.jj, W.t0, W.t1, itrs,
                     W.lm, W.ep, motion_only, 0.05, np.float64)
        poses, disps, dx, dz = _run_gpu_ba(W, itrs=itrs, motion_only=motion_only)
        if motion_only:
            assert np.array_equal(disps, W.disps)
        check_state(poses, disps, r64["poses"], r64["disps"], W.disps)
        np.testing.assert_allclose(dx, r64["dx"], rtol=2e-3, atol=5e-6)


def test_ba_with_update_and_linearisation_in_separate_launches():
    """DBA_BA_FUSE_UPDATE=0: one update launch per iteration, as the sharded driver runs them -- the same parity cases"""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, DBA_BA_FUSE_UPDATE="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.join(here, "test_gpu_ba.py"),
                        "-k", "test_ba_matches_oracle or three_and_four or motion_only or prepared"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_ba_prepared_workspace_is_reused_only_for_the_same_graph():
    """droid_backends.ba keeps one workspace per window shape and does not launch stage 0 when it finds it prepared for the
    very same edge tensor OBJECTS (CovisibleGraph.update with use_inactive=False calls ba on self.ii / self.jj): repeated
    calls, an in-place edit of jj, another graph of the same shape in between, and fresh tensors with the same contents all
    give what a cold call gives"""
    import droid_backends
    from droid_backends import _BA_WS
    assert _BA_WS.enabled
    W = syn.window_tiny_b(81)

    def cold(Wx):
        saved = _BA_WS.enabled
        _BA_WS.enabled = False
        try:
            return _run_gpu_ba(Wx)
        finally:
            _BA_WS.enabled = saved

    ref = cold(W)
    d = to_dev(W)
    args = lambda dd: (dd["poses"], dd["disps"], dd["intrinsics"], dd["disps_sens"], dd["target"], dd["weight"], dd["eta"],
                       dd["ii"], dd["jj"], W.t0, W.t1, 2, W.lm, W.ep, False)  # noqa: E731
    ii, jj = d["ii"], d["jj"]
    state = lambda dd: (dd["poses"].cpu().numpy(), dd["disps"].cpu().numpy())  # noqa: E731
    for rep in range(3):     # rep 0 prepares, rep 1, 2 find the tables in place
        dd = to_dev(W)
        dd["ii"], dd["jj"] = ii, jj
        droid_backends.ba(*args(dd))
        p, z = state(dd)
        assert np.array_equal(p, ref[0]) and np.array_equal(z, ref[1]), rep
    key = [k for k in _BA_WS.graph if k[-1] == (W.N, W.B, W.h, W.w, W.t0, W.t1)]
    assert key and _BA_WS.prepared_for(key[0], ii, jj, W.M)
    # a different graph of the same shape (same N, so the same workspace): edge 3 re-targeted, in place (version bump)
    W2 = syn.window_tiny_b(81)
    W2.jj = W2.jj.copy()
    W2.jj[3] = 5 if W2.jj[3] != 5 else 4
    ref2 = cold(W2)
    jj[3] = int(W2.jj[3])
    assert not _BA_WS.prepared_for(key[0], ii, jj, W.M)
    dd = to_dev(W2)
    dd["ii"], dd["jj"] = ii, jj
    droid_backends.ba(*args(dd))
    p, z = state(dd)
    assert np.array_equal(p, ref2[0]) and np.array_equal(z, ref2[1])
    # fresh tensors holding the first graph again: not the noted objects; stage 0 decides by the key (here: a rebuild)
    dd = to_dev(W)
    droid_backends.ba(*args(dd))
    p, z = state(dd)
    assert np.array_equal(p, ref[0]) and np.array_equal(z, ref[1])
    assert not _BA_WS.prepared_for(key[0], ii, jj, W.M) and _BA_WS.prepared_for(key[0], dd["ii"], dd["jj"], W.M)


def test_ba_on_the_64_pose_window_is_taken_by_the_window_solver_without_a_host_sync():
    """BASELINE configs[3] (64 KF / 512 edges: the reduced system is 8-10 poses wide): round 6's 80-row window with its ring of
    panels takes it (stage 0's verdict: 1), so the adapter no longer reads the skyline kernel's plan back from the workspace
    (one stream synchronisation per graph in round 5); repeated calls on the same edge tensors give the same bits"""
    import ctypes
    import droid_backends
    from dbaf_amd import _lib
    from droid_backends import _BA_WS
    W = syn.window_64_512(4)
    saved = _BA_WS.enabled
    _BA_WS.enabled = False
    try:
        ref = _run_gpu_ba(W)
    finally:
        _BA_WS.enabled = saved
    d0 = to_dev(W)
    ii, jj = d0["ii"], d0["jj"]
    for rep in range(4):
        d = to_dev(W)
        droid_backends.ba(d["poses"], d["disps"], d["intrinsics"], d["disps_sens"], d["target"], d["weight"], d["eta"], ii, jj,
                          W.t0, W.t1, 2, W.lm, W.ep, False)
        torch.cuda.synchronize()
        assert np.array_equal(d["poses"].cpu().numpy(), ref[0]) and np.array_equal(d["disps"].cpu().numpy(), ref[1]), rep
    dims = (W.N, W.B, W.h, W.w, W.t0, W.t1)
    key = [k for k in _BA_WS.graph if k[-1] == dims]
    assert key and _BA_WS.plan.get(key[0]) is None      # never asked
    ws, nbytes = _BA_WS.ws[key[0]]
    assert _lib.load().dba_ba_solver_verdict(*dims, ctypes.c_void_p(ws.data_ptr()), nbytes) == 1


def test_ba_general_size_solver_path_matches_too():
    """Every solver kernel behind the same parity cases: DBA_SOLVE_KERNEL = general (blocked Cholesky, csrc/ba_solve.hip) |
    band (skyline kernel, csrc/ba_solve_band.hip) | tile (register-tile LDL^T, csrc/ba_solve_tile.hip)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    # (banded windows go to the five-wave window kernel by default, csrc/ba_solve_wave.hip; the register-tile, skyline and
    # general kernels serve the other structures and must pass the same parity cases)
    # ... and the register-tile kernel with one elimination front only (two fronts are the default on banded systems).
    # (the four processes share the device: side by side they take as long as one)
    base = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.join(here, "test_gpu_ba.py"),
            os.path.join(here, "test_gpu_solve.py"), "-k"]
    sel = "matches_oracle or cholesky_failure or host_cholesky or non_spd"
    runs = [(k, dict(os.environ, DBA_SOLVE_KERNEL=k), sel) for k in ("general", "band", "tile")]
    runs.append(("one front", dict(os.environ, DBA_SOLVE_TWIST="0"), sel + " or two_front or failing_pivot"))
    procs = [(name, subprocess.Popen(base + [k], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
             for name, env, k in runs]
    for name, pr in procs:
        out, _ = pr.communicate(timeout=900)
        assert pr.returncode == 0, name + out[-3000:]


def _random_graph(rng, num_kf, n_edges, t0, long_range):
    """random covisibility graph: mostly near-neighbour edges, a few long-range ones (loop closures inside the
    window), sources on both sides of t0, duplicates allowed"""
    ii, jj = [], []
    for _ in range(n_edges):
        i = int(rng.integers(0, num_kf))
        if rng.uniform() < long_range:
            j = int(rng.integers(0, num_kf))
        else:
            j = int(np.clip(i + rng.integers(-3, 4), 0, num_kf - 1))
        if i == j:
            j = (i + 1) % num_kf
        ii.append(i)
        jj.append(j)
    return np.array(ii, np.int64), np.array(jj, np.int64)


RANDOM_GRAPHS = [(0, 6, 14, 1, 0.3), (1, 12, 40, 2, 0.0), (2, 20, 70, 1, 0.1), (3, 31, 110, 1, 0.0), (4, 34, 120, 3, 0.05),
                 (5, 40, 150, 1, 0.0), (6, 46, 170, 1, 0.02), (7, 27, 90, 5, 0.5), (8, 64, 200, 1, 0.0), (9, 33, 100, 1, 1.0)]


def test_ba_random_graphs_match_oracle():
    """irregular graphs exercise every solver path (register tiles up to 29 poses, the skyline variants above, the
    general kernel for wide skylines) and the pose-level skyline table of the prepare kernel: a skyline that missed
    a coupling would show up as a wrong pose update here.

    The maps are 8 x 12 pixels, so the systems are poorly conditioned and fp32 arithmetic -- the reference's included --
    sits well above the north-star pose bound on some of them.  Per graph: the pose update against the fp32-faithful
    oracle's (same fp32-built system, fp64 solve on both sides), the depths against the arbiter with the allowance of the
    reference arithmetic's own deviation, the poses within 1e-5 m / 1e-6 rad of the arbiter or within 6 x the fp32 oracle's
    own distance from it.  Over the ten graphs: the device is no further from the arbiter than the reference arithmetic is
    (median ratio <= 1.5; measured 0.2 .. 2.0 per graph either way: profiles/ parity report)."""
    orc = _oracle()
    ratios_t, ratios_r = [], []
    for seed, num_kf, n_edges, t0, long_range in RANDOM_GRAPHS:
        rng = np.random.default_rng(1000 + seed)
        ii, jj = _random_graph(rng, num_kf, n_edges, t0, long_range)
        W = syn.make_window(ii, jj, num_kf, h=8, w=12, seed=seed, t0=t0, target_noise=0.2)
        args = (W.poses, W.disps, W.intrinsics, W.disps_sens, W.target, W.weight, W.eta, W.ii, W.jj, W.t0, W.t1, 2,
                W.lm, W.ep, False, 0.05)
        r32 = orc.ba(*args, np.float32)
        r64 = orc.ba(*args, np.float64)
        poses, disps, dx, dz = _run_gpu_ba(W)
        clamp = lambda a: np.maximum(a, 0.001)  # noqa: E731
        if np.abs(r64["dx"]).max() == 0.0:  # the damped system was not positive definite: zero update on both sides
            assert np.abs(dx).max() == 0.0
            continue
        np.testing.assert_allclose(dx, r32["dx"], rtol=2e-2, atol=2e-4)
        os.environ["PYTEST_CURRENT_TEST"] = "tests/test_gpu_ba.py::test_ba_random_graphs_match_oracle[%d-%d-%d-%d-%s] (call)" % (
            seed, num_kf, n_edges, t0, long_range)
        print(check_state(poses, clamp(disps), r64["poses"], clamp(r64["disps"]), W.disps,
                          ref32_disps=clamp(r32["disps"]), ref32_poses=r32["poses"], d_rtol=2e-3, frac=0.95,
                          ref32_factor=6.0))
        from util import quat_angle
        p64 = r64["poses"]
        dt = np.abs(poses[:, :3].astype(np.float64) - p64[:, :3]).max()
        dr = quat_angle(poses[:, 3:].astype(np.float64), p64[:, 3:]).max()
        dt32 = np.abs(r32["poses"][:, :3].astype(np.float64) - p64[:, :3]).max()
        dr32 = quat_angle(r32["poses"][:, 3:].astype(np.float64), p64[:, 3:]).max()
        ratios_t.append(dt / max(dt32, 1e-9))
        ratios_r.append(dr / max(dr32, 1e-10))
    print("device / fp32-oracle distance from the arbiter, per graph: translation", np.round(ratios_t, 2), "rotation",
          np.round(ratios_r, 2))
    assert np.median(ratios_t) <= 1.5 and np.median(ratios_r) <= 1.5, (ratios_t, ratios_r)


def test_ba_extend_exports_the_system_and_matches_ba_with_a_zero_prior():
    """droid.cpp:140-178 (debug binding, unused by the runtime): with skip_solve the reduced system of the first
    iteration is exported to the CPU float64 H, v and the state is untouched; otherwise the solve uses Ad + Adprior"""
    import droid_backends
    W = syn.window_tiny_a(21)
    P = W.t1 - W.t0
    d = to_dev(W)
    H = torch.zeros(6 * P, 6 * P, dtype=torch.float64)
    v = torch.zeros(6 * P, dtype=torch.float64)
    A0 = torch.zeros(6 * P, 6 * P, dtype=torch.float64)
    p0, z0 = d["poses"].clone(), d["disps"].clone()
    droid_backends.ba_extend(d["poses"], d["disps"], d["intrinsics"], d["disps_sens"], d["target"], d["weight"],
                             d["eta"], d["ii"], d["jj"], H, v, A0, W.t0, W.t1, 1, W.lm, W.ep, False, True)
    assert torch.equal(d["poses"], p0) and torch.equal(d["disps"], z0)
    Hn, vn = H.numpy(), v.numpy()
    # (the stage mirrors the lower triangle, which is what the solvers read)
    assert np.abs(Hn).max() > 0 and np.array_equal(Hn, Hn.T)
    Hn = np.tril(Hn) + np.tril(Hn, -1).T
    # the exported system is the one ba() solves: dx = (H + damping)^-1 v
    Hd = Hn.copy()
    Hd[np.diag_indices_from(Hd)] += W.ep + W.lm * np.diag(Hn)
    dx_ref = np.linalg.solve(Hd, vn).reshape(P, 6)
    d2 = to_dev(W)
    dx, _ = droid_backends.ba(d2["poses"], d2["disps"], d2["intrinsics"], d2["disps_sens"], d2["target"], d2["weight"],
                              d2["eta"], d2["ii"], d2["jj"], W.t0, W.t1, 1, W.lm, W.ep, False)
    np.testing.assert_allclose(dx.cpu().numpy(), dx_ref, rtol=1e-4, atol=1e-7)
    # the prior REPLACES the corresponding part of a copy of Ad (droid_kernels.cu:1624-1632): an all-zero prior of full
    # size leaves Ad alone, a prior equal to Ad itself doubles the system
    for prior, factor in ((A0, 1.0), (torch.from_numpy(Hn.copy()), 2.0)):
        d3 = to_dev(W)
        dx3, dz3 = droid_backends.ba_extend(d3["poses"], d3["disps"], d3["intrinsics"], d3["disps_sens"], d3["target"],
                                            d3["weight"], d3["eta"], d3["ii"], d3["jj"], H, v, prior, W.t0, W.t1, 1,
                                            W.lm, W.ep, False, False)
        Hf = factor * Hn
        Hf[np.diag_indices_from(Hf)] += W.ep + W.lm * np.diag(Hf)
        np.testing.assert_allclose(dx3.cpu().numpy(), np.linalg.solve(Hf, vn).reshape(P, 6), rtol=1e-4, atol=1e-7)
        assert dz3 is not None and torch.isfinite(dz3).all()


def test_stage0_tells_the_host_which_solver_the_new_graph_gets():
    """when stage 0 rebuilds a graph's tables it runs the window solver's admission test on the new skyline and writes the
    verdict where launch_ba_solve reads it (pinned host memory, no synchronisation): a workspace whose graph changes from banded
    to far-reaching and back is sent to the right solver from the NEXT call on, not after the window kernel has found out by
    solving (or a probe every 1024 solves); the states are the oracle's either way"""
    import ctypes
    import droid_backends
    from dbaf_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    num_kf, n_edges, h, w = 20, 64, 16, 24      # (19 free poses: 114 unknowns -- a coupling of the window's ends does not fit the 80-row window)

    def graph(long_range):
        ii, jj = [], []
        for k in range(n_edges):
            i = int(rng.integers(0, num_kf))
            j = int(rng.integers(0, num_kf)) if (long_range and k % 3 == 0) else int(np.clip(i + rng.integers(-2, 3), 0, num_kf - 1))
            if i == j:
                j = (i + 1) % num_kf
            ii.append(i), jj.append(j)
        if long_range:
            ii += [1, num_kf - 1]
            jj += [num_kf - 1, 1]          # couples the ends of the window: no band
            del ii[:2], jj[:2]
        return np.array(ii, np.int64), np.array(jj, np.int64)

    verdicts = []
    for long_range in (False, True, False):
        ii, jj = graph(long_range)
        W = syn.make_window(ii, jj, num_kf, h, w, seed=3, target_noise=0.2)
        d = to_dev(W)
        for rep in range(2):
            droid_backends.ba(d["poses"].clone(), d["disps"].clone(), d["intrinsics"], d["disps_sens"], d["target"], d["weight"],
                              d["eta"], d["ii"], d["jj"], W.t0, W.t1, 2, W.lm, W.ep, False)
        torch.cuda.synchronize()
        dims = (W.N, W.B, W.h, W.w, W.t0, W.t1)
        key = [k for k in droid_backends._BA_WS.ws if k[-1] == dims]
        assert key, "the adapter keeps one workspace per window shape"
        ws, nbytes = droid_backends._BA_WS.ws[key[0]]
        verdicts.append(lib.dba_ba_solver_verdict(*dims, ctypes.c_void_p(ws.data_ptr()), nbytes))
        # ... and the result is the oracle's whichever kernel solved
        # (a small random window: bound like the other small fixtures -- 2 x the fp32-faithful oracle's own deviation)
        orc = _oracle()
        args = (W.poses, W.disps, W.intrinsics, W.disps_sens, W.target, W.weight, W.eta, W.ii, W.jj, W.t0, W.t1, 2,
                W.lm, W.ep, False, 0.05)
        r32, r64 = orc.ba(*args, np.float32), orc.ba(*args, np.float64)
        p, dd = d["poses"].clone(), d["disps"].clone()
        droid_backends.ba(p, dd, d["intrinsics"], d["disps_sens"], d["target"], d["weight"], d["eta"], d["ii"], d["jj"],
                          W.t0, W.t1, 2, W.lm, W.ep, False)
        check_state(p.cpu().numpy(), dd.cpu().numpy(), r64["poses"], r64["disps"], W.disps, ref32_disps=r32["disps"],
                    ref32_poses=r32["poses"], d_rtol=2e-3, frac=0.95, ref32_factor=6.0)   # (the bounds of test_ba_random_graphs_match_oracle)
    assert verdicts == [1, 2, 1], verdicts
