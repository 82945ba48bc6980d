"""GPU: the compiled `droid_backends` (csrc_ext/droid_backends_ext.cpp -> _droid_backends_C, the pybind11 module that takes
the place of the reference's src/droid.cpp:297-316) against the ctypes adapter: same C ABI underneath, so the same results --
`ba` in the deterministic accumulation mode to the bit -- and the same error behaviour (RuntimeError on non-contiguous or
host tensors, CHECK_CONTIGUOUS at droid.cpp:105-106)."""
import numpy as np
import pytest
import torch

from dbaf_amd import synthetic as syn
from util import to_dev

pytestmark = pytest.mark.gpu


def _both():
    import droid_backends
    assert droid_backends.compiled is not None, "the compiled adapter is not built (make ext)"
    return droid_backends, droid_backends.compiled


def test_package_serves_stateless_operators_from_the_compiled_module():
    db, C = _both()
    for name in ("frame_distance", "projmap", "depth_filter", "iproj", "corr_index_backward", "altcorr_forward", "altcorr_backward"):
        assert getattr(db, name) is getattr(C, name), name
    assert "gfx950" in C.version() and db.ADAPTER.startswith("compiled")


def test_compiled_ba_equals_the_ctypes_adapter_bit_for_bit_in_deterministic_mode():
    db, C = _both()
    from dbaf_amd import _lib
    lib = _lib.load()
    W = syn.window_tiny_b(4)
    lib.dba_ba_set_deterministic(1)
    try:
        outs = []
        for fn in (db.ba, C.ba):
            d = to_dev(W)
            r = fn(d["poses"], d["disps"], d["intrinsics"], d["disps_sens"], d["target"], d["weight"], d["eta"], d["ii"], d["jj"],
                   W.t0, W.t1, 2, W.lm, W.ep, False)
            assert tuple(r[1].shape) == (W.M, W.h * W.w)      # [|kx|, ht*wd], the reference's shape, from both
            outs.append((d["poses"].clone(), d["disps"].clone(), r[0].clone(), r[1].clone()))
        for a, b in zip(*outs):
            assert torch.equal(a, b)
        # ... and ba_clamped: the clamp over the whole buffer in the last launch
        outs = []
        floor = float(np.median(W.disps))
        for fn in (db.ba_clamped, C.ba_clamped):
            d = to_dev(W)
            fn(d["poses"], d["disps"], d["intrinsics"], d["disps_sens"], d["target"], d["weight"], d["eta"], d["ii"], d["jj"],
               W.t0, W.t1, 2, W.lm, W.ep, False, floor)
            assert float(d["disps"].min()) >= floor
            outs.append((d["poses"].clone(), d["disps"].clone()))
        for a, b in zip(*outs):
            assert torch.equal(a, b)
        # motion-only: no depth update; iterations <= 0: nothing touched
        d = to_dev(W)
        r = C.ba(d["poses"], d["disps"], d["intrinsics"], d["disps_sens"], d["target"], d["weight"], d["eta"], d["ii"], d["jj"],
                 W.t0, W.t1, 2, W.lm, W.ep, True)
        assert r[1] is None and torch.equal(d["disps"], to_dev(W)["disps"])
        r = C.ba(d["poses"], d["disps"], d["intrinsics"], d["disps_sens"], d["target"], d["weight"], d["eta"], d["ii"], d["jj"],
                 W.t0, W.t1, 0, W.lm, W.ep, False)
        assert r == [None, None]
    finally:
        lib.dba_ba_set_deterministic(0)


def test_compiled_bacore_and_operators_equal_the_ctypes_adapter():
    db, C = _both()
    W = syn.window_tiny_b(6)
    d = to_dev(W)
    P = W.t1 - W.t0
    res = []
    for cls in (db.BACore, C.BACore):
        s = to_dev(W)
        core = cls()
        core.init(s["poses"], s["disps"], s["intrinsics"], s["disps_sens"], s["target"], s["weight"], s["eta"], s["ii"], s["jj"],
                  W.t0, W.t1, 2, W.lm, W.ep, False)
        H = torch.zeros(6 * P, 6 * P, dtype=torch.float64)
        v = torch.zeros(6 * P, dtype=torch.float64)
        core.hessian(H, v)
        # ... and in the factor-graph side's coordinates (round 6): both adapters' hessian_gtsam against the host-side BA2GTSAM
        from dbaf_amd import fusion
        Tbc = np.array([0.03, 0.01, -0.08, 0.02, -0.01, 0.7, 0.71])
        Hn = H.numpy().copy()
        Hn[np.arange(6), np.arange(6)] += 0.00025
        ref = fusion.BA2GTSAM_augmented(Hn, v.numpy(), Tbc)
        aug = core.hessian_gtsam(Tbc) if cls is db.BACore else core.hessian_gtsam(torch.from_numpy(fusion.tangent_block(Tbc)), 0.00025).numpy()
        np.testing.assert_allclose(np.array(aug), ref, rtol=0, atol=1e-10 * np.abs(ref).max())
        dx = torch.linalg.solve(H + 1e-3 * torch.eye(6 * P, dtype=torch.float64), v)
        core.retract(dx)
        res.append((H, v, s["poses"].cpu(), s["disps"].cpu()))
    np.testing.assert_allclose(res[0][0].numpy(), res[1][0].numpy(), rtol=1e-12, atol=1e-12 * float(res[0][0].abs().max()))
    np.testing.assert_allclose(res[0][1].numpy(), res[1][1].numpy(), rtol=1e-12, atol=1e-12 * float(res[0][1].abs().max()))
    np.testing.assert_allclose(res[0][2].numpy(), res[1][2].numpy(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(res[0][3].numpy(), res[1][3].numpy(), rtol=1e-5, atol=1e-6)
    # stateless operators: compiled module vs the ctypes implementations kept in the package
    ci = db._ctypes_impl
    ii, jj = d["ii"], d["jj"]
    assert torch.equal(C.frame_distance(d["poses"], d["disps"], d["intrinsics"], ii, jj, 0.3),
                       ci["frame_distance"](d["poses"], d["disps"], d["intrinsics"], ii, jj, 0.3))
    for a, b in zip(C.projmap(d["poses"], d["disps"], d["intrinsics"], ii, jj), ci["projmap"](d["poses"], d["disps"], d["intrinsics"], ii, jj)):
        assert torch.equal(a, b)
    assert torch.equal(C.iproj(d["poses"], d["disps"], d["intrinsics"]), ci["iproj"](d["poses"], d["disps"], d["intrinsics"]))
    ix = torch.tensor([0, 2, 4], device="cuda")
    th = torch.tensor([0.05, 0.1, 0.2], device="cuda")
    assert torch.equal(C.depth_filter(d["poses"], d["disps"], d["intrinsics"], ix, th),
                       ci["depth_filter"](d["poses"], d["disps"], d["intrinsics"], ix, th))
    g = torch.Generator(device="cuda").manual_seed(0)
    vol = torch.randn(3, 12, 16, 12, 16, device="cuda", generator=g).half()
    c = (torch.rand(3, 2, 12, 16, device="cuda", generator=g) * 14).float()
    assert torch.equal(C.corr_index_forward(vol, c, 3)[0], db.corr_index_forward(vol, c, 3)[0])
    f1 = torch.randn(2, 12, 16, 32, device="cuda", generator=g)
    f2 = torch.randn(2, 12, 16, 32, device="cuda", generator=g)
    cc = (torch.rand(2, 1, 12, 16, 2, device="cuda", generator=g) * 12).float()
    assert torch.equal(C.altcorr_forward(f1, f2, cc, 3)[0], ci["altcorr_forward"](f1, f2, cc, 3)[0])


def test_compiled_error_behaviour():
    _, C = _both()
    W = syn.window_tiny_a(1)
    d = to_dev(W)
    with pytest.raises(RuntimeError):   # CHECK_CONTIGUOUS
        C.frame_distance(d["poses"].t().contiguous().t(), d["disps"], d["intrinsics"], d["ii"], d["jj"], 0.3)
    with pytest.raises(RuntimeError):   # host tensor: no CPU path
        C.iproj(d["poses"].cpu(), d["disps"], d["intrinsics"])
    # eta with a row count that is neither 1 nor |kx|: found by stage 0 on the device, raised by the module's next call
    torch.cuda.synchronize()
    C.check_async_errors()
    assert W.M > 2
    eta = torch.full((W.M - 1, W.h, W.w), 3e-7, device="cuda")
    C.ba(d["poses"], d["disps"], d["intrinsics"], d["disps_sens"], d["target"], d["weight"], eta, d["ii"], d["jj"], W.t0, W.t1, 1,
         W.lm, W.ep, False)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="eta with %d rows" % (W.M - 1)):
        C.check_async_errors()
    with pytest.raises(RuntimeError):   # more rows than kx can have: known on the host
        C.ba(d["poses"], d["disps"], d["intrinsics"], d["disps_sens"], d["target"], d["weight"],
             torch.zeros(W.t1 - W.t0 + W.N + 1, W.h, W.w, device="cuda"), d["ii"], d["jj"], W.t0, W.t1, 1, W.lm, W.ep, False)
    with pytest.raises(RuntimeError):   # BACore.hessian wants CPU float64
        core = C.BACore()
        core.init(d["poses"], d["disps"], d["intrinsics"], d["disps_sens"], d["target"], d["weight"], d["eta"], d["ii"], d["jj"],
                  W.t0, W.t1, 2, W.lm, W.ep, False)
        core.hessian(torch.zeros(6, 6, device="cuda", dtype=torch.float64), torch.zeros(6, dtype=torch.float64))


@pytest.mark.parametrize("sel_kind", ["index", "none", "mask", "negative"])
def test_gather_edges_equals_the_callers_statements(sel_kind):
    """droid_backends.gather_edges (both adapters) == covisible_graph.py:242-247 + :332-333 statement by statement, bit for bit"""
    db, C = _both()
    g = torch.Generator(device="cpu").manual_seed(5)
    h, w, n_in, n_act = 12, 20, 7, 5
    dev = "cuda"
    tgt_inac = torch.randn(1, n_in, h, w, 2, generator=g).to(dev)
    wgt_inac = torch.rand(1, n_in, h, w, 2, generator=g).to(dev)
    ii_inac = torch.randint(0, 9, (n_in,), generator=g).to(dev)
    jj_inac = torch.randint(0, 9, (n_in,), generator=g).to(dev)
    tgt = torch.randn(1, n_act, h, w, 2, generator=g).to(dev)
    wgt = torch.rand(1, n_act, h, w, 2, generator=g).to(dev)
    ii = torch.randint(0, 9, (n_act,), generator=g).to(dev)
    jj = torch.randint(0, 9, (n_act,), generator=g).to(dev)
    if sel_kind == "index":
        m = torch.tensor([5, 0, 3], device=dev)
    elif sel_kind == "negative":
        m = torch.tensor([-1, 2, -7], device=dev)
    elif sel_kind == "mask":
        m = (ii_inac >= 3) & (jj_inac >= 2)            # covisible_graph.py:240
    else:
        m = None
    mm = slice(None) if m is None else m
    ii_r = torch.cat([ii_inac[mm], ii], 0)
    jj_r = torch.cat([jj_inac[mm], jj], 0)
    tg_r = torch.cat([tgt_inac[:, mm], tgt], 1).view(-1, h, w, 2).permute(0, 3, 1, 2).contiguous()
    wt_r = torch.cat([wgt_inac[:, mm], wgt], 1).view(-1, h, w, 2).permute(0, 3, 1, 2).contiguous()
    for mod in (db, C):
        ii_n, jj_n, tg, wt = mod.gather_edges(tgt_inac, wgt_inac, ii_inac, jj_inac, m, tgt, wgt, ii, jj)
        assert torch.equal(ii_n, ii_r) and torch.equal(jj_n, jj_r)
        assert tg.shape == tg_r.shape and torch.equal(tg.view(torch.int32), tg_r.view(torch.int32))
        assert torch.equal(wt.view(torch.int32), wt_r.view(torch.int32))
    # no inactive edges selected / no active edges
    e = torch.empty(0, dtype=torch.int64, device=dev)
    ii_n, jj_n, tg, wt = db.gather_edges(tgt_inac, wgt_inac, ii_inac, jj_inac, e, tgt, wgt, ii, jj)
    assert torch.equal(ii_n, ii) and torch.equal(tg, tgt.view(-1, h, w, 2).permute(0, 3, 1, 2).contiguous())
    ii_n, jj_n, tg, wt = db.gather_edges(tgt_inac, wgt_inac, ii_inac, jj_inac, None, tgt[:, :0], wgt[:, :0], ii[:0], jj[:0])
    assert torch.equal(jj_n, jj_inac) and torch.equal(wt, wgt_inac.view(-1, h, w, 2).permute(0, 3, 1, 2).contiguous())
    with pytest.raises(RuntimeError):
        db.gather_edges(tgt_inac, wgt_inac, ii_inac, jj_inac, None, tgt, wgt[:, :2], ii, jj)
    with pytest.raises(RuntimeError):
        db.gather_edges(tgt_inac.cpu(), wgt_inac, ii_inac, jj_inac, None, tgt, wgt, ii, jj)
