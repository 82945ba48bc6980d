"""GPU: one whole update of the hot path -- reprojection + 4-level lookup in one launch, the caller's edge-list statements,
ba(itrs=2) with the clamp -- recorded into a hipGraph (dbaf_amd.graphed.GraphedUpdate) and replayed: the replay returns the
eager call's results bit for bit (deterministic accumulation switched on for the comparison), from new input VALUES too."""
import numpy as np
import pytest
import torch

from dbaf_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(6, 2, 24, 32), (9, 2, 55, 55), (8, 3, 32, 64)])
def test_graph_replay_of_one_update_equals_the_eager_update(shape):
    import droid_backends
    from dbaf_amd import _lib
    from dbaf_amd.corr import CorrBlock
    from dbaf_amd.graphed import GraphedUpdate
    nkf, rad, h, w = shape
    dev = torch.device("cuda", 0)
    W = syn.make_window(*syn.graph_banded(nkf, rad), nkf, h, w, seed=5)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    poses0, disps0 = t(W.poses), t(W.disps)
    intr, dsens, eta = t(W.intrinsics), t(W.disps_sens), t(W.eta)
    ii, jj, target0, weight0 = t(W.ii), t(W.jj), t(W.target), t(W.weight)
    fmaps = t(syn.make_fmaps(W.B, 128, h, w, 77))
    corr = CorrBlock(fmaps[ii][None], fmaps[jj][None], num_levels=4, radius=3).build()
    n_in = W.N // 3
    # the caller's book-keeping (covisible_graph.py:30-60, :242-247, :332-333): inactive + active lists, [1, n, h, w, 2] layout
    tgt5, wgt5 = target0.permute(0, 2, 3, 1)[None].contiguous(), weight0.permute(0, 2, 3, 1)[None].contiguous()
    st = dict(poses=poses0.clone(), disps=disps0.clone(), tgt_inac=tgt5[:, :n_in].clone(), tgt_act=tgt5[:, n_in:].clone(),
              wgt_inac=wgt5[:, :n_in].clone(), wgt_act=wgt5[:, n_in:].clone())
    ii_inac, jj_inac, ii_act, jj_act = ii[:n_in].clone(), jj[:n_in].clone(), ii[n_in:].clone(), jj[n_in:].clone()
    m = torch.arange(n_in, device=dev)

    def update():
        c, coords, _ = corr.lookup_reprojected(st["poses"], st["disps"], intr, ii, jj)
        ii_n = torch.cat([ii_inac[m], ii_act], 0)
        jj_n = torch.cat([jj_inac[m], jj_act], 0)
        tg = torch.cat([st["tgt_inac"][:, m], st["tgt_act"]], 1).view(-1, h, w, 2).permute(0, 3, 1, 2).contiguous()
        wt = torch.cat([st["wgt_inac"][:, m], st["wgt_act"]], 1).view(-1, h, w, 2).permute(0, 3, 1, 2).contiguous()
        droid_backends.ba_clamped(st["poses"], st["disps"], intr, dsens, tg, wt, eta, ii_n, jj_n, W.t0, W.t1, 2, W.lm, W.ep,
                                  False, 0.001)
        return c, coords

    def set_inputs(scale):
        st["poses"].copy_(poses0)
        st["disps"].copy_(disps0 * scale)
        st["tgt_act"].copy_(tgt5[:, n_in:] + (scale - 1.0))
        st["wgt_act"].copy_(wgt5[:, n_in:] * scale)

    lib = _lib.load()
    assert lib.dba_ba_set_deterministic(1) == 0
    try:
        eager = []
        for scale in (1.0, 0.97):
            set_inputs(scale)
            c, coords = update()
            eager.append([x.clone() for x in (c, coords, st["poses"], st["disps"])])
        set_inputs(1.0)
        g = GraphedUpdate(update)
        for k, scale in enumerate((1.0, 0.97, 1.0)):
            set_inputs(scale)
            c, coords = g.replay()
            torch.cuda.synchronize()
            ref = eager[k % 2]
            assert torch.equal(c.view(torch.int16), ref[0].view(torch.int16)), (k, "lookup")
            assert torch.equal(coords, ref[1]), (k, "coords")
            assert torch.equal(st["poses"], ref[2]), (k, "poses")
            assert torch.equal(st["disps"], ref[3]), (k, "disps")
        assert not torch.equal(eager[0][3], eager[1][3])   # (the two input sets do give different results)
    finally:
        lib.dba_ba_set_deterministic(0)
        droid_backends.check_async_errors()
