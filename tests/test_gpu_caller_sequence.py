"""GPU: the hot path driven in the ORDER AND WITH THE BOOK-KEEPING of its callers, stage by stage against the oracle.

1. `CovisibleGraph.update()` (/root/reference/dbaf/covisible_graph.py:214-342) after `add_factors` (:103-140) and
   `rm_factors`: volumes of two batches of edges concatenated (`corr.cat`), some edges retired into the inactive set
   (`corr[mask]`, target_inac / weight_inac), then per update: reproject -> motion features -> 4-level lookup ->
   [update operator: a deterministic stand-in, the ConvGRU is out of scope] -> target / weight -> inactive edges
   concatenated in front (:242-247) -> newest-frame down-weighting (:323-326) -> eta = .2 * damping[unique(ii)] + EP
   (:330) -> permute to [N,2,h,w] -> DepthVideo.ba -> clamp (depth_video.py:560).  Both correlation layouts: the fused
   flow-aligned CorrBlock of INTEGRATION.md section 2 and the zero-edit route (reference-layout volume from stock torch
   ops + droid_backends.corr_index_forward per level, i.e. what the reference's modules/corr.py executes).
2. `DepthVideo.ba` with IMU enabled (depth_video.py:350-462 marginalisation, :469-559 fusion): BACore on the
   marginalised edge set with depth priors switched off, H + 0.00025 on the first pose, BA2GTSAM, a dense stand-in for
   gtsam.marginalizeOut / LevenbergMarquardt, then two rounds of hessian -> solve -> GTSAM2BA -> retract on the active
   edge set, on a WHU-shaped (48x64) window with depth measurements (disps_sens > 0).
Every stage gets IDENTICAL inputs on both sides (the oracle continues from the GPU's state), so each comparison is at
the stage's own tolerance: reprojection 1e-4 px, lookups bit-exact, BA at the north-star tolerances."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from dbaf_amd import fusion
from dbaf_amd import synthetic as syn
from util import check_state

pytestmark = pytest.mark.gpu


def _oracle():
    from oracle import oracle as orc
    return orc


class ZeroEditCorrBlock:
    """what the reference's own CorrBlock (modules/corr.py:23-71) executes when `droid_backends` is this repo's module:
    stock torch ops for the volume and pyramid, droid_backends.corr_index_forward per level for the lookup"""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=3):
        self.num_levels, self.radius = num_levels, radius
        self.corr_pyramid = []
        batch, num, dim, ht, wd = fmap1.shape
        f1 = fmap1.reshape(batch * num, dim, ht * wd) / 4.0
        f2 = fmap2.reshape(batch * num, dim, ht * wd) / 4.0
        corr = torch.matmul(f1.transpose(1, 2), f2).view(batch, num, ht, wd, ht, wd)
        corr = corr.reshape(batch * num * ht * wd, 1, ht, wd)
        for i in range(num_levels):
            self.corr_pyramid.append(corr.view(batch * num, ht, wd, ht // 2 ** i, wd // 2 ** i))
            corr = F.avg_pool2d(corr, 2, stride=2)

    def __call__(self, coords):
        import droid_backends
        batch, num, ht, wd, _ = coords.shape
        coords = coords.permute(0, 1, 4, 2, 3).contiguous().view(batch * num, 2, ht, wd)
        out = []
        for i in range(self.num_levels):
            corr, = droid_backends.corr_index_forward(self.corr_pyramid[i], coords / 2 ** i, self.radius)
            out.append(corr.view(batch, num, -1, ht, wd))
        return torch.cat(out, dim=2)

    def cat(self, other):
        for i in range(self.num_levels):
            self.corr_pyramid[i] = torch.cat([self.corr_pyramid[i], other.corr_pyramid[i]], 0)
        return self

    def __getitem__(self, index):
        for i in range(self.num_levels):
            self.corr_pyramid[i] = self.corr_pyramid[i][index]
        return self


def _update_operator_stand_in(corr, motn):
    """deterministic stand-in for the ConvGRU update operator (out of scope): flow revision, confidence, damping from
    the correlation features and the motion features, with the reference's output layouts (droid_net.py)"""
    c = corr.float()
    delta = torch.stack([0.25 * torch.tanh(c[:, :, 0:98].mean(2)), 0.25 * torch.tanh(c[:, :, 98:].mean(2))], -1)
    delta = delta + 0.1 * motn[:, :, 2:4].permute(0, 1, 3, 4, 2)
    weight = torch.sigmoid(torch.stack([c[:, :, 24], c[:, :, 73]], -1))
    return delta, weight


@pytest.mark.parametrize("layout", ["sheared", "zero_edit_reference_layout"])
def test_covisible_graph_update_sequence(layout):
    import droid_backends
    from dbaf_amd import projective_ops as pops
    from dbaf_amd.corr import CorrBlock
    orc = _oracle()
    h, w, kf = 32, 64, 7
    ii_all, jj_all = syn.graph_banded(kf, 2)
    W = syn.make_window(ii_all, jj_all, kf, h, w, seed=9, intr=(23.9, 23.9, 31.9, 16.1))
    N = W.N
    dev = "cuda"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    fm = syn.make_fmaps(W.B, 128, h, w, 77)
    fmaps = t(fm)
    poses, disps = t(W.poses), t(W.disps)
    intr, dsens = t(W.intrinsics), t(W.disps_sens)
    K = intr[None, None].expand(1, W.B, 4).contiguous()
    ii, jj = t(W.ii), t(W.jj)
    Cls = CorrBlock if layout == "sheared" else ZeroEditCorrBlock
    kw = dict(layout="sheared") if layout == "sheared" else {}

    # add_factors twice (covisible_graph.py:127-132), oracle pyramids alongside
    first = slice(0, 10)
    second = slice(10, N)
    corr = Cls(fmaps[ii[first]][None], fmaps[jj[first]][None], **kw)
    corr = corr.cat(Cls(fmaps[ii[second]][None], fmaps[jj[second]][None], **kw))
    if layout == "sheared":
        # the volume the oracle's lookup reads: the same edges in the reference layout from the MFMA build (its parity
        # with the oracle's volume and the equality of the fused flow-aligned build are test_gpu_corr*.py's subject)
        parts = [CorrBlock.build_pyramid(fmaps[ii[s]][None], fmaps[jj[s]][None], 4) for s in (first, second)]
        pyr = [np.concatenate([parts[0][l].cpu().numpy(), parts[1][l].cpu().numpy()], 0) for l in range(4)]
    else:   # the stock-torch volume is the reference's own arithmetic: it is the oracle's volume too
        pyr = [p.cpu().numpy() for p in corr.corr_pyramid]
    # rm_factors(mask, store=True) (:142-170): the three oldest edges become inactive, their volumes are dropped
    retired = np.zeros(N, bool)
    retired[[0, 1, 4]] = True
    keep = torch.from_numpy(~retired).to(dev)
    rng = np.random.default_rng(3)
    target_inac = t(W.target[retired].transpose(0, 2, 3, 1))[None]      # [1, n_inac, h, w, 2]
    weight_inac = t(W.weight[retired].transpose(0, 2, 3, 1))[None]
    ii_inac, jj_inac = ii[~keep], jj[~keep]
    corr = corr[keep]
    pyr = [p[~retired] for p in pyr]
    ii, jj = ii[keep], jj[keep]
    iin, jjn = W.ii[~retired], W.jj[~retired]
    target = t(W.target[~retired].transpose(0, 2, 3, 1))[None].contiguous()   # [1, n, h, w, 2]
    coords0 = torch.stack(torch.meshgrid(torch.arange(h, device=dev).float(), torch.arange(w, device=dev).float(),
                                         indexing="ij")[::-1], -1)
    damping = 1e-6 * torch.ones(W.B, h, w, device=dev)
    damping[2] = 3e-6
    EP, t0, t1 = 1e-7, 1, kf

    for it in range(2):
        # ---- reproject + motion features (:219-222) ----
        coords1, _ = pops.projective_transform(poses[None], disps[None], K, ii, jj)
        oc, _ = orc.reproject(poses.cpu().numpy(), disps.cpu().numpy(), W.intrinsics, iin, jjn, np.float32)
        np.testing.assert_allclose(coords1[0].cpu().numpy(), oc, rtol=0, atol=2e-4)
        motn = torch.cat([coords1 - coords0, target - coords1], dim=-1).permute(0, 1, 4, 2, 3).clamp(-64.0, 64.0)
        # ---- lookup (:224): bit-exact on the coordinates the device produced ----
        c = corr(coords1)
        ref = orc.corr_lookup_pyramid(pyr, coords1[0].cpu().numpy(), 3)
        got = c[0].cpu().numpy()
        assert np.array_equal(got.view(np.uint16), ref.view(np.uint16)), (got != ref).mean()
        # ---- update operator stand-in, target / weight (:226-238) ----
        delta, weight = _update_operator_stand_in(c, motn)
        target = coords1 + delta.float()
        weight = weight.float()
        # ---- inactive edges in front (:242-247), down-weighting (:323-326), eta (:330), layout (:332-333) ----
        m = (ii_inac >= t0 - 3) & (jj_inac >= t0 - 3)
        ii_b = torch.cat([ii_inac[m], ii], 0)
        jj_b = torch.cat([jj_inac[m], jj], 0)
        target_b = torch.cat([target_inac[:, m], target], 1)
        weight_b = torch.cat([weight_inac[:, m], weight], 1).clone()
        weight_b[:, ii_b == ii_b.max()] /= 10.0
        weight_b[:, jj_b == jj_b.max()] /= 4.0
        eta = .2 * damping[torch.unique(ii_b)].contiguous() + EP
        tgt = target_b.view(-1, h, w, 2).permute(0, 3, 1, 2).contiguous()
        wgt = weight_b.view(-1, h, w, 2).permute(0, 3, 1, 2).contiguous()
        # ---- DepthVideo.ba (depth_video.py:331-333, :560) on both sides from the same state ----
        p0, d0 = poses.cpu().numpy(), disps.cpu().numpy()
        args = (p0, d0, W.intrinsics, W.disps_sens, tgt.cpu().numpy(), wgt.cpu().numpy(), eta.cpu().numpy(),
                ii_b.cpu().numpy(), jj_b.cpu().numpy(), t0, t1, 2, 1e-4, 0.1, False, 0.05)
        r64 = orc.ba(*args, np.float64)
        droid_backends.ba(poses, disps, intr, dsens, tgt, wgt, eta, ii_b, jj_b, t0, t1, 2, 1e-4, 0.1, False)
        disps.clamp_(min=0.001)
        clamp = lambda a: np.maximum(a, 0.001)  # noqa: E731
        print(layout, "update", it, check_state(poses.cpu().numpy(), disps.cpu().numpy(), r64["poses"],
                                                 clamp(r64["disps"]), d0, ref32_disps=None, frac=1.0))
    assert rng is not None


def test_marginalisation_and_fusion_sequence_whu_shape():
    import droid_backends
    orc = _oracle()
    h, w, kf = 48, 64, 8
    ii_all, jj_all = syn.graph_banded(kf, 2)
    W = syn.make_window(ii_all, jj_all, kf, h, w, seed=21, intr=(30.0, 30.0, 31.5, 23.7), sensor_frac=0.25)
    assert (W.disps_sens > 0).mean() > 0.1
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    poses, disps, intr, dsens = t(W.poses), t(W.disps), t(W.intrinsics), t(W.disps_sens)
    cur_ii, cur_jj = t(W.ii), t(W.jj)
    cur_target, cur_weight, cur_eta = t(W.target), t(W.weight), t(W.eta)
    Tbc = np.array([0.04, -0.01, 0.07, 0.02, -0.03, 0.7, 0.713])
    last_t0, last_t1, t0, t1 = 1, kf, 3, kf
    lm, ep = 1e-4, 0.1

    # ---- marginalisation (depth_video.py:362-407) ----
    marg_idx = (cur_ii >= last_t0) & (cur_ii < t0) & (cur_ii < last_t1 - 2) & (cur_jj < last_t1 - 2)
    marg_ii, marg_jj = cur_ii[marg_idx], cur_jj[marg_idx]
    assert len(marg_ii) > 0
    marg_t0, marg_t1 = last_t0, int(marg_jj.max().item()) + 1
    nm = 6 * (marg_t1 - marg_t0)
    core = droid_backends.BACore()
    core.init(poses, disps, intr, torch.zeros_like(dsens), cur_target[marg_idx], cur_weight[marg_idx],
              cur_eta[0:marg_t1 - marg_t0], marg_ii, marg_jj, marg_t0, marg_t1, 2, lm, ep, False)
    H = torch.zeros(nm, nm, dtype=torch.float64)
    v = torch.zeros(nm, dtype=torch.float64)
    p_before, d_before = poses.clone(), disps.clone()
    core.hessian(H, v)
    assert torch.equal(poses, p_before) and torch.equal(disps, d_before)     # hessian() is read-only on the state
    Hs, vs = core.hessian_staging()                                           # pinned staging: same numbers, zero copy
    assert Hs.is_pinned() and torch.equal(Hs, H) and torch.equal(vs, v)
    del core
    mi = marg_idx.cpu().numpy()
    oc = orc.BACore(W.poses, W.disps, W.intrinsics, np.zeros_like(W.disps_sens), W.target[mi], W.weight[mi],
                    W.eta[0:marg_t1 - marg_t0], W.ii[mi], W.jj[mi], marg_t0, marg_t1, lm, ep, np.float64)
    Ho, vo = oc.hessian()
    np.testing.assert_allclose(H.numpy(), Ho, rtol=0, atol=2e-5 * np.abs(Ho).max())
    np.testing.assert_allclose(v.numpy(), vo, rtol=0, atol=2e-5 * np.abs(vo).max())

    def prior_from(Hn, vn):
        Hn = Hn.copy()
        Hn[np.arange(6), np.arange(6)] += 0.00025                             # "for stability" (:395)
        aug = fusion.BA2GTSAM_augmented(Hn, vn, Tbc)                          # (:398-400)
        Hg, vg = aug[:nm, :nm], aug[:nm, nm]
        return fusion.marginal_prior(Hg, vg, t0 - marg_t0)                    # stand-in for marginalizeOut (:443)

    (Hp, vp), (Hpo, vpo) = prior_from(H.numpy(), v.numpy()), prior_from(Ho, vo)

    # ---- optimisation (:464-559) ----
    active = (cur_ii >= t0) & (cur_jj >= t0)
    a_ii, a_jj = cur_ii[active], cur_jj[active]
    a_target, a_weight = cur_target[active], cur_weight[active]
    a_eta = cur_eta[(t0 - int(cur_ii.min().item())):]
    n = 6 * (t1 - t0)
    core = droid_backends.BACore()
    core.init(poses, disps, intr, dsens, a_target, a_weight, a_eta, a_ii, a_jj, t0, t1, 2, lm, ep, False)
    ai = active.cpu().numpy()
    oc = orc.BACore(W.poses, W.disps, W.intrinsics, W.disps_sens, W.target[ai], W.weight[ai],
                    W.eta[(t0 - int(W.ii.min())):], W.ii[ai], W.jj[ai], t0, t1, lm, ep, np.float64)

    def fuse_and_solve(Hn, vn, Hprior, vprior):
        """dense stand-in for the LevenbergMarquardt solve of visual factor + marginalisation prior (:523-547)"""
        Hg, vg = fusion.BA2GTSAM(Hn, vn, Tbc)
        k = Hprior.shape[0]
        Hg[:k, :k] += Hprior
        vg[:k] += vprior
        Hg[np.diag_indices(n)] += 1e-3
        return fusion.GTSAM2BA(np.linalg.solve(Hg, vg), Tbc)                  # (:557)

    for it in range(2):
        H = torch.zeros(n, n, dtype=torch.float64)
        v = torch.zeros(n, dtype=torch.float64)
        core.hessian(H, v)
        Ho, vo = oc.hessian()
        np.testing.assert_allclose(H.numpy(), Ho, rtol=0, atol=3e-5 * np.abs(Ho).max())
        # each side solves ITS OWN system (as a deployment would), then retracts
        dx_dz = core.retract(torch.from_numpy(fuse_and_solve(H.numpy(), v.numpy(), Hp, vp)))
        oc.retract(fuse_and_solve(Ho, vo, Hpo, vpo))
        assert dx_dz[0].shape == (t1 - t0, 6) and dx_dz[1].shape[1] == h * w
    del core
    disps.clamp_(min=0.001)                                                   # (:560)
    torch.cuda.synchronize()
    print(check_state(poses.cpu().numpy(), disps.cpu().numpy(), oc.poses, np.maximum(oc.disps, 0.001), W.disps))
    assert not torch.equal(poses, p_before)
