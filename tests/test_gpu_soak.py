"""GPU soak (BASELINE configs[4], "long sequence"): 300 consecutive updates of a sliding keyframe window with the graph
changing every 6 updates -- the cadence of the frontend (dbaf_frontend: one add_factors / rm_factors round per keyframe,
then its updates; /root/reference/dbaf/covisible_graph.py:103-170,214-342) -- through every cache on the path:

  * the slot-addressed CorrBlock (edges dropped and added in place: no store is ever re-allocated),
  * droid_backends.ba's workspace LRU (the window shape changes with the edge count), its prepared-graph cache (same edge
    tensors between two graph changes, new ones after), the solver plan,
  * the zero-edit route's flow-aligned shadows (reference-layout level tensors re-created at every graph change, looked up
    through droid_backends.corr_index_forward; budget + LRU).
Asserted: the fused lookup equals the zero-edit lookup bit for bit at every update; at checkpoints the oracle, continuing
from the device's state, agrees with the device's BA at the north-star tolerance and its lookup bit for bit; device memory
does not grow once the first cycles have been seen."""
import numpy as np
import pytest
import torch

from dbaf_amd import synthetic as syn
from test_gpu_caller_sequence import _update_operator_stand_in
from util import check_state

pytestmark = pytest.mark.gpu


def test_300_updates_with_graph_churn():
    import droid_backends
    from droid_backends import _BA_WS, _SHADOWS
    from dbaf_amd.corr import CorrBlock
    from oracle import oracle as orc
    h, w, kf, C = 24, 32, 9, 32
    dev = "cuda"
    g_i, g_j = syn.graph_banded(kf, 3)                      # 42 candidate edges; 24-30 of them are active at a time
    W = syn.make_window(g_i, g_j, kf, h, w, seed=21, intr=(12.0, 12.0, 15.7, 11.9), buffer=kf + 3)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    fmaps = t(syn.make_fmaps(W.B, C, h, w, 5))
    poses, disps = t(W.poses), t(W.disps)
    poses0, disps0 = poses.clone(), disps.clone()
    intr, dsens = t(W.intrinsics), t(W.disps_sens)
    K = intr[None].expand(W.B, 4).contiguous()
    coords0 = torch.stack(torch.meshgrid(torch.arange(h, device=dev).float(), torch.arange(w, device=dev).float(),
                                         indexing="ij")[::-1], -1)
    all_ii, all_jj = t(W.ii), t(W.jj)
    rng = np.random.default_rng(0)
    active = list(range(26))                                # indices into the candidate list
    spare = list(range(26, W.N))
    CorrBlock.default_capacity, cap_before = 32, CorrBlock.default_capacity
    # (caches of earlier tests hold memory that this run would release: start from empty ones)
    _SHADOWS.clear()
    for dct in (_BA_WS.ws, _BA_WS.graph, _BA_WS.plan):
        dct.clear()
    torch.cuda.empty_cache()
    try:
        corr = CorrBlock(fmaps[all_ii[active]][None], fmaps[all_jj[active]][None])
        corr.build()
        store_ptrs = [s.data_ptr() for s in corr._stores]
        ii, jj = all_ii[active].contiguous(), all_jj[active].contiguous()
        ref_pyr = CorrBlock.build_pyramid(fmaps[ii][None], fmaps[jj][None], 4)       # the zero-edit caller's tensors
        target = None
        damping = 1e-6 * torch.ones(W.B, h, w, device=dev)
        t0, t1 = 1, kf
        mem_mark, checkpoints, n_changes = None, 0, 0
        for u in range(300):
            if u % 6 == 0 and u > 0:
                # ---- a keyframe's graph change: drop 1-3 edges, add as many or one more / fewer (the edge COUNT moves, and
                # with it the BA workspace's shape) ----
                n_drop = int(rng.integers(1, 4))
                n_add = min(len(spare), max(1, n_drop + int(rng.integers(-1, 2))))
                n_add = max(1, min(n_add, 30 - (len(active) - n_drop)))      # 22..30 active edges: inside the 32 slots
                if len(active) - n_drop + n_add < 22:
                    n_add += 1
                drop_pos = sorted(rng.choice(len(active), size=n_drop, replace=False).tolist())
                keep = torch.ones(len(active), dtype=torch.bool, device=dev)
                keep[drop_pos] = False
                dropped = [active[p] for p in drop_pos]
                add = [spare.pop(int(rng.integers(len(spare)))) for _ in range(n_add)]
                active = [a for p, a in enumerate(active) if p not in drop_pos] + add
                spare += dropped
                new = torch.tensor(add, device=dev)
                corr = corr[keep].cat(CorrBlock(fmaps[all_ii[new]][None], fmaps[all_jj[new]][None]))   # rm + add_factors
                add_pyr = CorrBlock.build_pyramid(fmaps[all_ii[new]][None], fmaps[all_jj[new]][None], 4)
                ref_pyr = [torch.cat([p[keep], q], 0) for p, q in zip(ref_pyr, add_pyr)]               # the reference's way
                ii, jj = all_ii[active].contiguous(), all_jj[active].contiguous()
                if target is not None:
                    c_new, _, _ = _reproject(poses, disps, K, all_ii[new], all_jj[new])
                    target = torch.cat([target[:, keep], c_new], 1)
                n_changes += 1
                assert [s.data_ptr() for s in corr._stores] == store_ptrs and corr.stats["grown"] == 0
            n = len(active)
            # ---- update(): lookup with the reprojection in its launch; the zero-edit route beside it ----
            c, coords1, _ = corr.lookup_reprojected(poses, disps, K, ii, jj)
            cp = coords1[0].permute(0, 3, 1, 2).contiguous()
            ze = torch.cat([droid_backends.corr_index_forward(ref_pyr[l], cp / 2 ** l, 3)[0].view(1, n, -1, h, w)
                            for l in range(4)], 2)
            assert torch.equal(c, ze), "update %d: fused and zero-edit lookups differ" % u
            if target is None:
                target = coords1.clone()
            motn = torch.cat([coords1 - coords0, target - coords1], dim=-1).permute(0, 1, 4, 2, 3).clamp(-64.0, 64.0)
            delta, weight = _update_operator_stand_in(c, motn)
            target = coords1 + 0.2 * delta.float()
            wgt = weight.float().clone()
            wgt[:, ii == ii.max()] /= 10.0
            wgt[:, jj == jj.max()] /= 4.0
            kx = torch.unique(torch.cat([torch.arange(t0, t1, device=dev), ii]))     # the rows C has (droid_kernels.cu:1416-1424)
            eta = .2 * damping[kx].contiguous() + 1e-7
            tg = target.view(-1, h, w, 2).permute(0, 3, 1, 2).contiguous()
            wg = wgt.view(-1, h, w, 2).permute(0, 3, 1, 2).contiguous()
            check = (u % 50 == 49) or u in (0, 7)
            if check:
                p0, d0 = poses.cpu().numpy(), disps.cpu().numpy()
                args = (p0, d0, W.intrinsics, W.disps_sens, tg.cpu().numpy(), wg.cpu().numpy(), eta.cpu().numpy(),
                        ii.cpu().numpy(), jj.cpu().numpy(), t0, t1, 2, 1e-4, 0.1, False, 0.05)
                r64 = orc.ba(*args, np.float64)
                r32 = orc.ba(*args, np.float32)
                ref = orc.corr_lookup_pyramid([p.cpu().numpy() for p in ref_pyr], coords1[0].cpu().numpy(), 3)
                assert np.array_equal(c[0].cpu().numpy().view(np.uint16), ref.view(np.uint16))
            droid_backends.ba(poses, disps, intr, dsens, tg, wg, eta, ii, jj, t0, t1, 2, 1e-4, 0.1, False)
            disps.clamp_(min=0.001)
            if check:
                clamp = lambda a: np.maximum(a, 0.001)  # noqa: E731
                check_state(poses.cpu().numpy(), disps.cpu().numpy(), r64["poses"], clamp(r64["disps"]), d0,
                            ref32_disps=clamp(r32["disps"]), ref32_poses=r32["poses"], d_rtol=1.5e-4)
                checkpoints += 1
            if u % 30 == 29:   # the scene must not run away under the stand-in operator: pull the state back
                poses.copy_(poses0)
                disps.copy_(disps0)
            if u == 60:
                torch.cuda.synchronize()
                mem_mark, n_mark = torch.cuda.memory_allocated(), n
        torch.cuda.synchronize()
        assert checkpoints == 8 and n_changes == 49
        assert torch.isfinite(poses).all() and torch.isfinite(disps).all()
        # memory: what is held after 300 updates is what was held after 60 (+ slack for the allocator's rounding of the
        # tensors whose size follows the edge count)
        per_edge = sum(p.numel() * 2 for p in ref_pyr) / n          # the caller's level tensors follow n (and their shadows,
        shadowed = 2 if _SHADOWS.bytes_held() else 1                # when the policy builds any at this cadence)
        grown = torch.cuda.memory_allocated() - mem_mark - shadowed * per_edge * (n - n_mark)
        assert grown < 16 * 2 ** 20, "device memory grew by %.1f MiB over 240 updates" % (grown / 2 ** 20)
        assert len(_BA_WS.ws) <= _BA_WS.max_entries
        if _SHADOWS.enabled:
            assert _SHADOWS.bytes_held() <= _SHADOWS.budget
            assert len(_SHADOWS.seen) <= 64, "shadow entries of dead tensors are not released: %d" % len(_SHADOWS.seen)
    finally:
        CorrBlock.default_capacity = cap_before


def _reproject(poses, disps, K, ii, jj):
    from dbaf_amd import projective_ops as pops
    c, v = pops.projective_transform(poses[None], disps[None], K[None], ii, jj)
    return c, v, None
