"""CPU: the trajectory-error metric (dbaf_amd/ate.py) -- evo-style APE as the reference's evaluation scripts use it."""
import numpy as np

from dbaf_amd import ate
from dbaf_amd import synthetic as syn


def test_alignment_removes_a_rigid_motion_and_scale_only_when_asked():
    rng = np.random.default_rng(0)
    ref = np.cumsum(rng.standard_normal((40, 3)), 0)
    R = syn._qrot  # noqa: F841  (synthetic's quaternion helpers are exercised below)
    q = np.array([0.2, -0.1, 0.3, 0.9])
    q /= np.linalg.norm(q)
    rot = np.stack([syn._qrot(q, e) for e in np.eye(3)], 1)
    est = 1.7 * ref @ rot.T + np.array([3.0, -2.0, 5.0])
    assert ate.ape_translation_rmse(ref, est, align=True, correct_scale=True) < 1e-9
    assert ate.ape_translation_rmse(ref, est, align=True, correct_scale=False) > 0.1
    est1 = ref @ rot.T + np.array([3.0, -2.0, 5.0])
    assert ate.ape_translation_rmse(ref, est1, align=True) < 1e-9
    assert ate.ape_translation_rmse(ref, est1, align=False) > 1.0
    noisy = est1 + 0.01 * rng.standard_normal(est1.shape)
    assert 0.005 < ate.ape_translation_rmse(ref, noisy) < 0.03


def test_camera_centres_invert_world_to_camera_poses():
    W = syn.window_tiny_b(1)
    c = ate.camera_centres(W.poses_gt)
    for k in range(W.num_kf):   # T maps the centre to the camera origin: R c + t = 0
        assert np.abs(syn._qrot(W.poses_gt[k, 3:].astype(np.float64), c[k]) + W.poses_gt[k, :3]).max() < 1e-6
    assert ate.ate(W.poses_gt, W.poses_gt) < 1e-12
    assert ate.ate(W.poses_gt[:W.num_kf], W.poses[:W.num_kf]) > 1e-4   # the perturbed state is off the ground truth
