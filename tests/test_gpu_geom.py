"""GPU parity for the geometry kernels around the hot path."""
import numpy as np
import pytest
import torch

from dbaf_amd import synthetic as syn
from util import to_dev

pytestmark = pytest.mark.gpu


def _oracle():
    from oracle import oracle as orc
    return orc


@pytest.mark.parametrize("mk", [lambda: syn.window_tiny_b(3), lambda: syn.window_25_96(1)])
def test_reproject_matches_oracle_and_reference_golden(mk):
    from dbaf_amd import projective_ops as pops
    orc = _oracle()
    W = mk()
    d = to_dev(W)
    K = d["intrinsics"][None, None].expand(1, W.B, 4).contiguous()
    coords, valid = pops.projective_transform(d["poses"][None], d["disps"][None], K, d["ii"], d["jj"])
    rc, rv = orc.reproject(W.poses, W.disps, W.intrinsics, W.ii, W.jj, np.float64)
    np.testing.assert_allclose(coords[0].cpu().numpy(), rc, rtol=1e-4, atol=2e-4)
    assert (valid[0].cpu().numpy() == rv).mean() > 0.9999


def test_reproject_matches_committed_reference_vectors(golden_dir):
    import os
    from dbaf_amd import projective_ops as pops
    g = np.load(os.path.join(golden_dir, "projective.npz"))
    for tag in ("a", "b"):
        poses = torch.from_numpy(g[f"{tag}_poses"]).cuda()
        disps = torch.from_numpy(g[f"{tag}_disps"]).cuda()
        K = torch.from_numpy(np.tile(g[f"{tag}_intr"], (disps.shape[0], 1))).cuda()
        coords, valid = pops.projective_transform(poses[None], disps[None], K[None],
                                                  torch.from_numpy(g[f"{tag}_ii"]).cuda(),
                                                  torch.from_numpy(g[f"{tag}_jj"]).cuda())
        np.testing.assert_allclose(coords[0].cpu().numpy(), g[f"{tag}_coords"], rtol=1e-4, atol=2e-4)
        assert (valid[0].cpu().numpy() == g[f"{tag}_valid"]).mean() > 0.999


def test_frame_distance_projmap_iproj_depth_filter():
    import droid_backends
    orc = _oracle()
    W = syn.window_25_96(2)
    d = to_dev(W)
    ii = torch.arange(0, 10, device="cuda").repeat_interleave(3)
    jj = (ii + torch.tensor([1, 2, 3], device="cuda").repeat(10)).clamp(max=24)
    dist = droid_backends.frame_distance(d["poses"], d["disps"], d["intrinsics"], ii, jj, 0.3)
    ref = orc.frame_distance(W.poses, W.disps, W.intrinsics, ii.cpu().numpy(), jj.cpu().numpy(), 0.3, np.float64)
    np.testing.assert_allclose(dist.cpu().numpy(), ref, rtol=2e-4, atol=1e-4)

    coords, valid = droid_backends.projmap(d["poses"], d["disps"], d["intrinsics"], ii, jj)
    rc, rv = orc.projmap(W.poses, W.disps, W.intrinsics, ii.cpu().numpy(), jj.cpu().numpy(), np.float64)
    np.testing.assert_allclose(coords.cpu().numpy(), rc, rtol=1e-4, atol=5e-4)
    assert (valid.cpu().numpy() == rv).mean() > 0.9999

    pts = droid_backends.iproj(d["poses"][:25].contiguous(), d["disps"][:25].contiguous(), d["intrinsics"])
    rp = orc.iproj(W.poses[:25], W.disps[:25], W.intrinsics, np.float64)
    np.testing.assert_allclose(pts.cpu().numpy(), rp, rtol=1e-4, atol=1e-4)

    inds = torch.tensor([0, 3, 12, 24], device="cuda")
    thresh = torch.tensor([0.05, 0.1, 0.2, 0.4], device="cuda")
    cnt = droid_backends.depth_filter(d["poses"], d["disps"], d["intrinsics"], inds, thresh)
    rcnt = orc.depth_filter(W.poses, W.disps, W.intrinsics, inds.cpu().numpy(), thresh.cpu().numpy(), np.float32)
    assert (cnt.cpu().numpy() == rcnt).mean() > 0.999
