"""`droid_backends` for MI355X -- drop-in for the pybind11 module of the same name that the reference
builds from src/droid.cpp (bindings at /root/reference/src/droid.cpp:297-316).

Same functions, argument order, in-place semantics and error behaviour (a non-contiguous tensor raises
RuntimeError like TORCH_CHECK at droid.cpp:105-106); every call is a thin adapter over the C ABI of
include/dba_hip.h (libdba_hip.so, hand-written HIP for gfx950).  Tensors must live on the HIP device
("cuda" in PyTorch-ROCm): there is no CPU path, calls with CPU tensors raise.

Call sites in the reference that run unmodified against this module:
  dbaf/depth_video.py:255-265,331-346,392-397,469-478,527,558   (frame_distance, ba, BACore)
  dbaf/modules/corr.py:9-13,17-20,79-88                          (corr_index_*, altcorr_*)
  dbaf/dbaf.py:76-79,119-121                                     (iproj, depth_filter)
"""
import ctypes

import numpy as np
import torch

from dbaf_amd import _lib
from dbaf_amd._lib import DBA_F16, DBA_F32

__all__ = ["ba", "ba_extend", "frame_distance", "projmap", "depth_filter", "iproj", "altcorr_forward",
           "altcorr_backward", "corr_index_forward", "corr_index_backward", "BACore", "ba_clamped", "gather_edges",
           "check_async_errors"]

# The compiled adapter (csrc_ext/droid_backends_ext.cpp -> _droid_backends_C.so, `make ext`): the pybind11 module a
# maintainer would build in place of the reference's src/droid.cpp -- every binding of droid.cpp:297-316 over the same C ABI.
# `import droid_backends._droid_backends_C as droid_backends` is a complete drop-in by itself.  This package serves the
# stateless operators straight from it and keeps, in Python, the policies the reference's call pattern rewards: the BA
# workspace / prepared-graph cache (`ba`), pinned staging (`BACore`), the flow-aligned shadows (`corr_index_forward`).
# DBA_ADAPTER=ctypes keeps every call on the ctypes path (the two are the same C ABI calls; tests run both).
import os as _os

compiled = None
if _os.environ.get("DBA_ADAPTER", "") != "ctypes":
    try:
        from . import _droid_backends_C as compiled   # noqa: F401
    except ImportError as _e:
        if _os.environ.get("DBA_ADAPTER", "") == "compiled":
            raise
        import warnings as _w   # once per process: the ctypes path serves everything, but nobody should find out by accident
        _w.warn("droid_backends: the compiled adapter (_droid_backends_C, `make ext`) failed to import (%s); every call goes "
                "through ctypes" % _e)
ADAPTER = "compiled (pybind11) + ctypes policies" if compiled is not None else "ctypes"


def _check(x, name, dtype=None):
    if not isinstance(x, torch.Tensor):
        raise RuntimeError("%s must be a torch.Tensor" % name)
    if not x.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)  # CHECK_CONTIGUOUS, droid.cpp:105
    if not x.is_cuda:
        raise RuntimeError("droid_backends (MI355X): %s must be a HIP device tensor; there is no CPU path" % name)
    if dtype is not None and x.dtype != dtype:
        raise RuntimeError("%s must be %s, got %s" % (name, dtype, x.dtype))
    return x


def _ptr(x):
    return ctypes.c_void_p(x.data_ptr()) if x is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ws(N, B, ht, wd, t0, t1, device):
    nbytes = _lib.load().dba_ba_workspace_bytes(N, B, ht, wd, t0, t1)
    if nbytes == 0:
        raise RuntimeError("dba_ba_workspace_bytes: invalid sizes N=%d B=%d ht=%d wd=%d t0=%d t1=%d"
                           % (N, B, ht, wd, t0, t1))
    return torch.empty(nbytes, dtype=torch.uint8, device=device), nbytes


class _BaWorkspaces:
    """One BA workspace per (device, stream, window shape), kept across calls.

    The index tables of stage 0 depend on nothing but (ii, jj, t0, t1, sizes, Schur form), and CovisibleGraph.update() runs
    droid_backends.ba on the same EDGE LIST many times before the graph changes -- but, with use_inactive=True (every update
    of the frontend, dbaf_frontend.py:251,357,474-483), through NEW tensor objects each time (`torch.cat([self.ii_inac[m],
    self.ii])`, covisible_graph.py:242-247).  So "the same graph" is decided on the device, on the contents: stage 0 keeps a
    key of the edge list it built its tables for in the workspace, and a call whose edges compare equal leaves that launch
    after ~2 us (dba_ba_run with prepared = 2; no host synchronisation on either outcome).  On top of that, a call that hands
    over the very same tensor OBJECTS at the same in-place version (what update() does with use_inactive=False) does not
    launch stage 0 at all (prepared = 1): identity through weak references -- an id() can only be reused after its object
    died, which clears the note -- and `_version`, which every in-place write bumps.
    All calls are stream-ordered on the caller's current stream, like the reference's (one workspace is safe to reuse
    there); DBA_WS_CACHE=0 falls back to a fresh workspace and a full dba_ba per call."""

    def __init__(self):
        import os
        self.enabled = os.environ.get("DBA_WS_CACHE", "1") != "0"
        self.ws = {}        # (device, stream, dims) -> (tensor, nbytes)
        self.graph = {}     # key -> (ref(ii), version, ref(jj), version, schur form generation, eta rows it was checked with)
        self.plan = {}      # key -> which skyline-solver variant solved this graph last time (meta[7])
        self.max_entries = 4

    def workspace(self, key, dims, device):
        ent = self.ws.get(key)
        if ent is None:
            if len(self.ws) >= self.max_entries:      # window shapes change with the graph: keep the latest few
                old = next(iter(self.ws))
                self.ws.pop(old)
                self.graph.pop(old, None)
                self.plan.pop(old, None)
            ent = _ws(*dims, device)
            # "no graph prepared yet": the key area stage 0 compares against must not be whatever the allocator left there
            _lib.check(_lib.load().dba_ba_workspace_init(*dims, _ptr(ent[0]), ent[1], _stream()), "dba_ba_workspace_init")
            self.ws[key] = ent
        return ent

    def prepared_for(self, key, ii, jj, eta_rows):
        g = self.graph.get(key)
        return (g is not None and g[0]() is ii and g[2]() is jj and g[1] == ii._version and g[3] == jj._version
                and g[4] == _lib.schur_generation() and g[5] == eta_rows)

    def note(self, key, ii, jj, eta_rows):
        import weakref
        self.graph[key] = (weakref.ref(ii), ii._version, weakref.ref(jj), jj._version, _lib.schur_generation(), eta_rows)
        self.plan.pop(key, None)

    def solver_hint(self, key, dims):
        """windows of 30-64 poses: which skyline-solver variant took this graph's structure in the previous call (meta[7] of
        the workspace; ONE stream synchronisation per graph, at its second update on the same tensor objects) -> what
        dba_ba_prepared need not queue"""
        N, B, ht, wd, t0, t1 = dims
        if not (174 < 6 * (t1 - t0) <= 384):
            return 0
        # (round 6: the window kernel takes bands of up to ten poses at these sizes too -- nothing is queued behind it, and the
        # verdict stage 0 left in pinned host memory says so without a synchronisation; only a graph it refuses gets here)
        ws, nbytes = self.ws[key]
        if _lib.load().dba_ba_solver_verdict(N, B, ht, wd, t0, t1, _ptr(ws), nbytes) != 2:
            return 0
        h = self.plan.get(key)
        if h is None:
            lay = _lib.BaLayout()
            _lib.load().dba_ba_get_layout(N, B, ht, wd, t0, t1, ctypes.byref(lay))
            ws = self.ws[key][0]
            h = int(ws[lay.meta + 28:lay.meta + 32].view(torch.int32).item())
            self.plan[key] = h
        return 1 if h == 1 else 0


_BA_WS = _BaWorkspaces()
_BACORE_OWNER = {}   # workspace address -> id of the BACore whose hessian() linearised into it last


def _num_kx(eta, ii, t0, t1, ht, wd):
    """|kx| = |unique(arange(t0,t1) U ii)| without a device sync when eta has one row per kx entry
    (as DBA-Fusion always passes it, covisible_graph.py:330; the row count is verified on the device, see _ba)."""
    rows = eta.numel() // (ht * wd)
    if rows > 1:
        return rows
    return _num_kx_exact(ii, t0, t1)


def _num_kx_exact(ii, t0, t1):
    ts = torch.arange(t0, t1, device=ii.device, dtype=ii.dtype)
    return int(torch.unique(torch.cat([ts, ii])).numel())


_ETA_CHECK_SYNC = _os.environ.get("DBA_ETA_CHECK", "") == "sync"


def _raise_pending_eta_error(ws=None, dims=None):
    """eta.view(-1, HW) must broadcast against the |kx| rows of C (droid_kernels.cu:1476): one row, or one per kx entry.  The
    reference raises a broadcast error in the call itself, before it touches anything (its torch::_unique synchronises the host
    anyway); here |kx| only exists on the device, stage 0 compares, and a mismatch (a) makes the offending call a no-op ON THE
    DEVICE -- poses and inverse depths stay as they were, dx and dz come back zero -- and (b) surfaces at the NEXT call on that
    workspace (the report lives in the workspace's own pinned words: other devices, streams or threads do not see it), or at an
    explicit droid_backends.check_async_errors() after a synchronisation -- the price of a call that never stops the host.
    DBA_ETA_CHECK=sync restores the synchronous check."""
    r, k = ctypes.c_int(0), ctypes.c_int(0)
    lib = _lib.load()
    if ws is not None:
        hit = lib.dba_ba_poll_eta_error_ws(*dims, _ptr(ws[0]), ws[1], ctypes.byref(r), ctypes.byref(k)) == 1
    else:
        hit = lib.dba_ba_poll_eta_error(ctypes.byref(r), ctypes.byref(k)) == 1
    if hit:
        _BA_WS.graph.clear()   # (whichever note remembered that row count as checked must not skip stage 0 with it again)
        raise RuntimeError("an earlier droid_backends.ba / BACore.hessian call was given eta with %d rows; it must have 1 or "
                           "|unique(arange(t0,t1) U ii)| = %d rows (droid_kernels.cu:1476: eta.view(-1, ht*wd) is added to C "
                           "row by row); that call changed nothing (poses and inverse depths untouched, dx = dz = 0)"
                           % (r.value, k.value))


def check_async_errors():
    """raises what an earlier asynchronous call of this module found wrong on the device (today: the eta row count); only
    complete after the stream was synchronised"""
    _raise_pending_eta_error()


def _check_eta_rows(eta_rows, ii, t0, t1):
    """the bounds on the row count that are known without the device (the exact count is checked by stage 0)"""
    if eta_rows == 1:
        return
    P, N = max(t1 - t0, 0), int(ii.shape[0])
    if eta_rows > P + N or eta_rows < min(P, 1):
        raise RuntimeError("eta has %d rows; it must have 1 or |unique(arange(t0,t1) U ii)| rows (at most %d here)"
                           % (eta_rows, P + N))


def _ba_args(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj):
    _check(targets, "targets", torch.float32)
    _check(weights, "weights", torch.float32)
    _check(poses, "poses", torch.float32)
    _check(disps, "disps", torch.float32)
    _check(intrinsics, "intrinsics", torch.float32)
    _check(disps_sens, "disps_sens", torch.float32)
    _check(ii, "ii", torch.int64)
    _check(jj, "jj", torch.int64)
    if not eta.is_contiguous():
        eta = eta.contiguous()  # the reference takes eta.view(-1, ht*wd) of whatever it is given
    eta = _check(eta, "eta", torch.float32)
    B, ht, wd = disps.shape
    if eta.numel() % (ht * wd) != 0:
        raise RuntimeError("eta must view as [-1, ht*wd] (droid_kernels.cu:1476), got %s" % (tuple(eta.shape),))
    return eta, int(ii.shape[0]), int(B), int(ht), int(wd), int(eta.numel() // (ht * wd))


def ba(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, t0, t1, iterations, lm, ep,
       motion_only):
    """Dense bundle adjustment, in place on poses[t0:t1] and disps[kx] (droid.cpp:109-138)."""
    return _ba(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, t0, t1, iterations, lm, ep,
               motion_only, 0.0)


def ba_clamped(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, t0, t1, iterations, lm, ep,
               motion_only, disp_floor=0.001):
    """`ba(...)` followed by the caller's `disps.clamp_(min=disp_floor)` (DepthVideo.ba, dbaf/depth_video.py:559-560) in ONE
    call: the clamp rides in the call's last launch instead of being an elementwise launch of its own -- over the whole
    buffer, like the caller's (frames outside kx and motion_only calls included: the caller rescales depths between BA
    calls, dbaf_frontend.py:570,814), so the state is the same bit for bit (dba_ba_run, include/dba_hip.h).  Not a
    reference binding: an integration that wants it replaces the two statements of depth_video.py by this one call."""
    if not (float(disp_floor) > 0.0):
        raise RuntimeError("ba_clamped: disp_floor must be positive")
    return _ba(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, t0, t1, iterations, lm, ep,
               motion_only, float(disp_floor))


def gather_edges(target_inac, weight_inac, ii_inac, jj_inac, sel, target, weight, ii, jj):
    """The edge tensors of one BA call as CovisibleGraph.update(use_inactive=True) assembles them
    (dbaf/covisible_graph.py:242-247: `torch.cat([self.ii_inac[m], self.ii], 0)`, the same for jj, target, weight; :332-333:
    `target.view(-1, ht, wd, 2).permute(0, 3, 1, 2).contiguous()`, the same for weight) in ONE launch instead of ten.
    target_inac, weight_inac [1, n_inac, ht, wd, 2] (or without the leading 1), ii_inac, jj_inac [n_inac]; sel: int64 indices of
    the inactive edges to take (None: all; a boolean mask is converted with torch's own nonzero, i.e. with its host
    synchronisation); target, weight [1, n, ht, wd, 2], ii, jj [n]: the active edges.
    Returns (ii, jj, target, weight) with target, weight [n_sel + n, 2, ht, wd]: the arguments of `ba`.  Not a reference
    binding: an integration replaces the six statements above by this call."""
    lib = _lib.load()
    ht, wd = int(target.shape[-3]), int(target.shape[-2])
    dev = target.device
    if not target.is_cuda:
        raise RuntimeError("gather_edges (MI355X): tensors must be HIP device tensors; no CPU path")

    def five(x, name):
        if x.shape[-1] != 2 or tuple(x.shape[-3:-1]) != (ht, wd):
            raise RuntimeError("gather_edges: %s must be [..., %d, %d, 2], got %s" % (name, ht, wd, tuple(x.shape)))
        return _check(x.reshape(-1, ht, wd, 2), name, torch.float32)

    ta, wa = five(target, "target"), five(weight, "weight")
    ti, wi = five(target_inac, "target_inac"), five(weight_inac, "weight_inac")
    iia, jja = _check(ii.reshape(-1), "ii", torch.int64), _check(jj.reshape(-1), "jj", torch.int64)
    iii, jji = _check(ii_inac.reshape(-1), "ii_inac", torch.int64), _check(jj_inac.reshape(-1), "jj_inac", torch.int64)
    n_act, n_inac = int(ta.shape[0]), int(ti.shape[0])
    if wa.shape[0] != n_act or iia.shape[0] != n_act or jja.shape[0] != n_act:
        raise RuntimeError("gather_edges: target, weight, ii, jj disagree on the number of active edges")
    if wi.shape[0] != n_inac or iii.shape[0] != n_inac or jji.shape[0] != n_inac:
        raise RuntimeError("gather_edges: target_inac, weight_inac, ii_inac, jj_inac disagree on the number of inactive edges")
    if sel is None:
        n_sel = n_inac
    else:
        if sel.dtype == torch.bool:
            sel = sel.reshape(-1).nonzero().reshape(-1)      # (synchronises the host, like `x[mask]` itself)
        sel = _check(sel.reshape(-1), "sel", torch.int64)
        n_sel = int(sel.shape[0])
    n = n_sel + n_act
    tg = torch.empty(n, 2, ht, wd, dtype=torch.float32, device=dev)
    wt = torch.empty(n, 2, ht, wd, dtype=torch.float32, device=dev)
    ii_n = torch.empty(n, dtype=torch.int64, device=dev)
    jj_n = torch.empty(n, dtype=torch.int64, device=dev)
    _lib.check(lib.dba_ba_gather_edges(_ptr(ti), _ptr(wi), _ptr(iii), _ptr(jji), n_inac, _ptr(sel), n_sel, _ptr(ta), _ptr(wa),
                                       _ptr(iia), _ptr(jja), n_act, ht, wd, _ptr(tg), _ptr(wt), _ptr(ii_n), _ptr(jj_n),
                                       _stream()), "dba_ba_gather_edges")
    return ii_n, jj_n, tg, wt


def _ba(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, t0, t1, iterations, lm, ep,
        motion_only, disp_floor):
    eta, N, B, ht, wd, eta_rows = _ba_args(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj)
    t0, t1 = int(t0), int(t1)
    P = t1 - t0
    dims = (N, B, ht, wd, t0, t1)
    if _BA_WS.enabled:   # what an earlier call on THIS workspace (device, stream, window shape) was found to be wrong with
        key = (poses.device, torch.cuda.current_stream().cuda_stream, dims)
        _raise_pending_eta_error(_BA_WS.workspace(key, dims, poses.device), dims)
    else:
        _raise_pending_eta_error()
    _check_eta_rows(eta_rows, ii, t0, t1)
    if int(iterations) <= 0:   # the reference returns two undefined tensors and touches nothing (:1437, :1511)
        return [None, None]
    lib = _lib.load()
    if _BA_WS.enabled:
        ws, nbytes = _BA_WS.workspace(key, dims, poses.device)
        # the same tensor objects as last time: stage 0 is not even launched; anything else: stage 0 compares the edge list
        # with the key in the workspace and leaves at once when it is the graph the tables were built for
        prepared = 1 if _BA_WS.prepared_for(key, ii, jj, eta_rows) else 2
        hint = _BA_WS.solver_hint(key, dims) if prepared == 1 else 0
    else:
        key, prepared, hint = None, 0, 0
        ws, nbytes = _ws(*dims, poses.device)
    if _ETA_CHECK_SYNC and eta_rows > 1 and prepared != 1:
        nk = _num_kx_exact(ii, t0, t1)
        if eta_rows != nk:
            raise RuntimeError("eta has %d rows; it must have 1 or |unique(arange(t0,t1) U ii)| = %d rows "
                               "(droid_kernels.cu:1476: eta.view(-1, ht*wd) is added to C row by row)" % (eta_rows, nk))
    dx = torch.empty(P, 6, dtype=torch.float32, device=poses.device)       # fully written by the last iteration
    Mmax = min(B, P + N)
    dz_full = torch.empty(Mmax, ht * wd, dtype=torch.float32, device=poses.device)  # rows [0,|kx|) written
    rc = lib.dba_ba_run(_ptr(poses), _ptr(disps), _ptr(intrinsics), _ptr(disps_sens), _ptr(targets),
                        _ptr(weights), _ptr(eta), eta_rows, _ptr(ii), _ptr(jj), N, B, ht, wd, t0, t1,
                        int(iterations), float(lm), float(ep), int(bool(motion_only)), _ptr(dx),
                        _ptr(dz_full), _ptr(ws), nbytes, _stream(), prepared, int(hint), float(disp_floor))
    _lib.check(rc, "dba_ba")
    if key is not None and prepared != 1:
        _BA_WS.note(key, ii, jj, eta_rows)
    if motion_only:
        return [dx, None]
    # (eta with one row per kx entry: that IS the count; a single-row eta needs the count from the device, which
    # synchronises -- as the reference's own torch::_unique does in every call)
    return [dx, dz_full[:_num_kx(eta, ii, t0, t1, ht, wd)]]


def ba_extend(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, H, v, A_prior, t0, t1,
              iterations, lm, ep, motion_only, skip_solve):
    """Debug variant bound at droid.cpp:140-178 (never called by the DBA-Fusion runtime): per iteration
    the reduced system is exported to the CPU float64 tensors H, v; unless skip_solve, it is solved with
    the prior added (solveDense, droid_kernels.cu:180-198) and the state retracted."""
    eta, N, B, ht, wd, eta_rows = _ba_args(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj)
    t0, t1 = int(t0), int(t1)
    P = t1 - t0
    lib = _lib.load()
    ws, nbytes = _ws(N, B, ht, wd, t0, t1, poses.device)
    dx = dz = None
    if motion_only:
        return ba(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, t0, t1, iterations, lm,
                  ep, True)
    _lib.check(lib.dba_ba_prepare(_ptr(ii), _ptr(jj), N, B, ht, wd, t0, t1, _ptr(ws), nbytes, _stream()),
               "dba_ba_prepare")
    Hn = np.zeros((6 * P, 6 * P), np.float64)
    vn = np.zeros((6 * P,), np.float64)
    for _ in range(int(iterations)):
        _lib.check(lib.dba_ba_linearize(_ptr(poses), _ptr(disps), _ptr(intrinsics), _ptr(disps_sens),
                                        _ptr(targets), _ptr(weights), _ptr(eta), eta_rows, _ptr(ii), _ptr(jj),
                                        None, N, B, ht, wd, t0, t1, 0.05, _ptr(ws), nbytes, _stream()),
                   "dba_ba_linearize")
        _lib.check(lib.dba_ba_reduce(_ptr(ii), _ptr(jj), None, N, B, ht, wd, t0, t1, 0, _ptr(ws), nbytes,
                                     _stream()), "dba_ba_reduce")
        lay = _lib.BaLayout()
        lib.dba_ba_get_layout(N, B, ht, wd, t0, t1, ctypes.byref(lay))
        n = 6 * P
        Hd = ws[lay.H:lay.H + 8 * n * n].view(torch.float64).view(n, n)
        bd = ws[lay.b:lay.b + 8 * n].view(torch.float64)
        Hn[:] = Hd.cpu().numpy()
        vn[:] = bd.cpu().numpy()
        H.copy_(torch.from_numpy(Hn)[:H.shape[0], :H.shape[1]])
        v.copy_(torch.from_numpy(vn)[:v.shape[0]])
        if skip_solve:
            return [dx, dz]
        prior = Hn.copy()  # Adprior = Ad, then overwritten where A_prior is defined (:1624-1630)
        Ap = A_prior.detach().cpu().numpy().astype(np.float64)
        prior[:Ap.shape[0], :Ap.shape[1]] = Ap
        Ht = np.ascontiguousarray(Hn + prior)
        dx = torch.zeros(P, 6, dtype=torch.float32, device=poses.device)
        _lib.check(lib.dba_bacore_optimize(Ht.ctypes.data_as(ctypes.c_void_p),
                                           vn.ctypes.data_as(ctypes.c_void_p), N, B, ht, wd, t0, t1,
                                           float(lm), float(ep), _ptr(dx), _ptr(ws), nbytes, _stream()),
                   "dba_bacore_optimize")
        Mmax = min(B, P + N)
        dz_full = torch.zeros(Mmax, ht * wd, dtype=torch.float32, device=poses.device)
        _lib.check(lib.dba_ba_update(_ptr(poses), _ptr(disps), _ptr(ii), _ptr(jj), None, N, B, ht, wd, t0, t1,
                                     1, 1, _ptr(dz_full), _ptr(ws), nbytes, _stream()), "dba_ba_update")
        dz = dz_full[:_num_kx(eta, ii, t0, t1, ht, wd)]
    return [dx, dz]


class BACore:
    """Split-phase BA for the GTSAM fusion path (src/bacore.h:4-70, droid_kernels.cu:1786-1956):
    init caches the problem, hessian(H, v) fills CPU float64 H [6P,6P], v [6P] with the Schur-reduced
    camera system (alpha = 0.001), retract(dx) applies an externally solved float64 update."""

    def __init__(self):
        self._ready = False

    def init(self, poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, t0, t1, iterations, lm,
             ep, motion_only):
        eta, N, B, ht, wd, eta_rows = _ba_args(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj)
        self.poses, self.disps, self.intrinsics, self.disps_sens = poses, disps, intrinsics, disps_sens
        self.targets, self.weights, self.eta, self.ii, self.jj = targets, weights, eta, ii, jj
        self.t0, self.t1, self.lm, self.ep = int(t0), int(t1), float(lm), float(ep)
        self.N, self.B, self.ht, self.wd, self.eta_rows = N, B, ht, wd, eta_rows
        self.P = self.t1 - self.t0
        _raise_pending_eta_error()
        _check_eta_rows(eta_rows, ii, self.t0, self.t1)
        # one workspace per window shape, kept across the BACore objects DepthVideo.ba creates per call (its own pool:
        # nothing droid_backends.ba does to its workspaces can fall between hessian and retract); stage 0 recognises an
        # unchanged edge list by its key, so the second hessian() of an update finds its tables in place
        if _BA_WS.enabled:
            key = ("bacore", poses.device, torch.cuda.current_stream().cuda_stream, self._dims())
            self.ws, self.nbytes = _BA_WS.workspace(key, self._dims(), poses.device)
            self._prepared = 2
        else:
            self.ws, self.nbytes = _ws(N, B, ht, wd, self.t0, self.t1, poses.device)
            self._prepared = 0
        self.dx = None
        # the hand-off to the factor-graph side: the reduced system leaves the device in pinned host memory
        # the LIBRARY keeps per workspace -- written there by the last kernel of hessian() itself, which the call waits for on a
        # completion word (round 6: no DMA launches, no stream synchronisation, nothing allocated here)
        self._stage_ptr, self._stage_layout = None, None
        self._args, self._linearised = None, None
        self._ready = True

    def _dims(self):
        return (self.N, self.B, self.ht, self.wd, self.t0, self.t1)

    def _front_args(self):
        """the 16 leading arguments of the hessian entries (pointers and sizes: fixed once init() has run)"""
        if self._args is None:
            self._args = (_ptr(self.poses), _ptr(self.disps), _ptr(self.intrinsics), _ptr(self.disps_sens), _ptr(self.targets),
                          _ptr(self.weights), _ptr(self.eta), self.eta_rows, _ptr(self.ii), _ptr(self.jj)) + self._dims()
        return self._args

    def _stage0_mode(self):
        """1: this object's previous hessian() built (or confirmed) the tables for exactly these edge tensors -- the second
        hessian of an update (depth_video.py:527) does not even launch stage 0; else the workspace's mode (2: compare by key)"""
        if self._prepared == 2 and self._linearised == (self.ii._version, self.jj._version) and \
                _BACORE_OWNER.get(self.ws.data_ptr()) == id(self):
            return 1
        return self._prepared

    def _after_hessian(self, stage_ptr, layout):
        # the linearisation retract() back-substitutes with lives in the workspace, which BACore objects of one window shape
        # share: remember whose it is (the reference's objects are independent; DepthVideo.ba has one alive at a time)
        _BACORE_OWNER[self.ws.data_ptr()] = id(self)
        self._linearised = (self.ii._version, self.jj._version)
        self._stage_ptr, self._stage_layout = stage_ptr, layout
        _raise_pending_eta_error()    # (the call has waited for its last kernel: the verdict of its own stage 0 is in)

    def _hessian_host(self, layout, A36=None, stabilizer=0.0):
        out = ctypes.c_void_p()
        a = None if A36 is None else (ctypes.c_double * 36)(*[float(x) for x in A36])
        rc = _lib.load().dba_bacore_hessian_host(
            *self._front_args(), _ptr(self.ws), self.nbytes, _stream(), self._stage0_mode(), layout,
            None if a is None else ctypes.cast(a, ctypes.c_void_p), float(stabilizer), ctypes.byref(out))
        _lib.check(rc, "dba_bacore_hessian")
        self._after_hessian(out.value, layout)

    def _stage_view(self, count):
        import numpy as _np
        return _np.ctypeslib.as_array((ctypes.c_double * count).from_address(self._stage_ptr))

    def hessian(self, H, v):
        assert self._ready, "BACore.init must be called first"
        if H.is_cuda or v.is_cuda or H.dtype != torch.float64 or v.dtype != torch.float64:
            raise RuntimeError("BACore.hessian: H, v must be CPU float64 tensors (droid_kernels.cu:1889-1890)")
        n = 6 * self.P
        if tuple(H.shape) == (n, n) and tuple(v.shape) == (n,) and H.is_contiguous() and v.is_contiguous():
            # the library copies its pinned block into the caller's tensors itself (two memcpys, no tensor views on the way)
            rc = _lib.load().dba_bacore_hessian_run(*self._front_args(), ctypes.c_void_p(H.data_ptr()),
                                                    ctypes.c_void_p(v.data_ptr()), _ptr(self.ws), self.nbytes, _stream(),
                                                    self._stage0_mode())
            _lib.check(rc, "dba_bacore_hessian")
            self._after_hessian(None, 0)
            return
        self._hessian_host(0)
        if n == 0:
            return
        Hh, vh = self.hessian_staging()   # the reference fills H_accessor.size(0) x size(1) entries (:1892-1897)
        H.copy_(Hh[:H.shape[0], :H.shape[1]])
        v.copy_(vh[:v.shape[0]])

    def hessian_staging(self):
        """zero-copy access to what the last hessian() left in the library's pinned block of this workspace: (H [6P,6P], v [6P])
        float64 CPU tensors, valid until the next hessian call on a BACore of this window shape"""
        assert self._stage_layout == 0, "hessian() has not run"
        n = 6 * self.P
        if self._stage_ptr is None:   # (the last hessian() went straight into the caller's tensors: ask where the block is)
            out = ctypes.c_void_p()
            _lib.check(_lib.load().dba_bacore_staging(*self._dims(), _ptr(self.ws), self.nbytes, ctypes.byref(out)), "dba_bacore_staging")
            self._stage_ptr = out.value
        flat = torch.from_numpy(self._stage_view(n * n + n))
        return flat[:n * n].view(n, n), flat[n * n:]

    def hessian_gtsam(self, Tbc, stabilizer=0.00025):
        """hessian() + `for i in range(6): H[i,i] += 0.00025` + gtsam.BA2GTSAM(H, v, Tbc) of dbaf/depth_video.py:394-401 (:524-529)
        in one call: the export kernel applies the stabiliser and the change of tangent coordinates (Hg = J^T H J, vg = J^T v,
        J = blockdiag(-Ad(Tbc^-1) with swapped row halves), :20-29) on the device and writes the augmented matrix
        [Hg | vg] of shape [6P, 6P + 1] -- what the fork's gtsam.BA2GTSAM returns -- into pinned host memory.  Returns a numpy
        view of it (valid until the next hessian call on this window shape).  Tbc: a gtsam.Pose3, a 4x4 matrix, a (t, q) 7-vector,
        or the 6x6 block itself."""
        assert self._ready, "BACore.init must be called first"
        import numpy as _np
        from dbaf_amd import fusion
        A = _np.asarray(Tbc, _np.float64) if (not hasattr(Tbc, "matrix") and _np.shape(Tbc) == (6, 6)) else fusion.tangent_block(Tbc)
        n = 6 * self.P
        self._hessian_host(1, A.reshape(-1), stabilizer)
        return self._stage_view(n * (n + 1)).reshape(n, n + 1)

    def optimize(self, H, v):
        assert self._ready, "BACore.init must be called first"
        n = 6 * self.P
        Hh = H.detach().to("cpu", torch.float64).contiguous()
        vh = v.detach().to("cpu", torch.float64).contiguous()
        self.dx = torch.zeros(self.P, 6, dtype=torch.float32, device=self.poses.device)
        rc = _lib.load().dba_bacore_optimize(ctypes.c_void_p(Hh.data_ptr()), ctypes.c_void_p(vh.data_ptr()),
                                             *self._dims(), self.lm, self.ep, _ptr(self.dx), _ptr(self.ws),
                                             self.nbytes, _stream())
        _lib.check(rc, "dba_bacore_optimize")
        assert Hh.shape[0] >= n

    def retract(self, _dx):
        assert self._ready, "BACore.init must be called first"
        if _BACORE_OWNER.get(self.ws.data_ptr(), id(self)) != id(self):
            raise RuntimeError("BACore.retract: another BACore of the same window shape has linearised into the shared "
                               "workspace since this object's hessian(); call hessian() again before retract()")
        dxh = _dx.detach().to("cpu", torch.float64).contiguous().view(-1)
        if dxh.numel() < 6 * self.P:
            raise RuntimeError("BACore.retract: dx must have %d entries" % (6 * self.P))
        dev = self.poses.device
        dx = torch.empty(self.P, 6, dtype=torch.float32, device=dev)        # (both fully written: dx by the copy of the
        Mmax = min(self.B, self.P + self.N)                                 #  workspace's slot, dz rows [0, |kx|) by the update)
        dz_full = torch.empty(Mmax, self.ht * self.wd, dtype=torch.float32, device=dev)
        rc = _lib.load().dba_bacore_retract(_ptr(self.poses), _ptr(self.disps), _ptr(self.ii), _ptr(self.jj),
                                            *self._dims(), ctypes.c_void_p(dxh.data_ptr()), _ptr(dx),
                                            _ptr(dz_full), _ptr(self.ws), self.nbytes, _stream())
        _lib.check(rc, "dba_bacore_retract")
        self.dx = dx
        return [dx, dz_full[:_num_kx(self.eta, self.ii, self.t0, self.t1, self.ht, self.wd)]]


def frame_distance(poses, disps, intrinsics, ii, jj, beta):
    """droid.cpp:181-197."""
    for x, nm, dt in ((poses, "poses", torch.float32), (disps, "disps", torch.float32),
                      (intrinsics, "intrinsics", torch.float32), (ii, "ii", torch.int64), (jj, "jj", torch.int64)):
        _check(x, nm, dt)
    N = int(ii.shape[0])
    _, ht, wd = disps.shape
    dist = torch.empty(N, dtype=torch.float32, device=poses.device)     # (every entry is written by its workgroup)
    _lib.check(_lib.load().dba_frame_distance(_ptr(poses), _ptr(disps), _ptr(intrinsics), _ptr(ii), _ptr(jj), N,
                                              int(ht), int(wd), float(beta), _ptr(dist), _stream()),
               "dba_frame_distance")
    return dist


def projmap(poses, disps, intrinsics, ii, jj):
    """droid.cpp:200-215."""
    for x, nm, dt in ((poses, "poses", torch.float32), (disps, "disps", torch.float32),
                      (intrinsics, "intrinsics", torch.float32), (ii, "ii", torch.int64), (jj, "jj", torch.int64)):
        _check(x, nm, dt)
    N = int(ii.shape[0])
    _, ht, wd = disps.shape
    coords = torch.zeros(N, ht, wd, 3, dtype=torch.float32, device=poses.device)
    valid = torch.zeros(N, ht, wd, 1, dtype=torch.float32, device=poses.device)
    _lib.check(_lib.load().dba_projmap(_ptr(poses), _ptr(disps), _ptr(intrinsics), _ptr(ii), _ptr(jj), N, int(ht),
                                       int(wd), _ptr(coords), _ptr(valid), _stream()), "dba_projmap")
    return [coords, valid]


def depth_filter(poses, disps, intrinsics, ix, thresh):
    """droid.cpp:281-295."""
    for x, nm, dt in ((poses, "poses", torch.float32), (disps, "disps", torch.float32),
                      (intrinsics, "intrinsics", torch.float32), (ix, "ix", torch.int64),
                      (thresh, "thresh", torch.float32)):
        _check(x, nm, dt)
    num = int(ix.shape[0])
    nbuf, ht, wd = disps.shape
    counter = torch.zeros(num, ht, wd, dtype=disps.dtype, device=disps.device)
    _lib.check(_lib.load().dba_depth_filter(_ptr(poses), _ptr(disps), _ptr(intrinsics), _ptr(ix), _ptr(thresh), num,
                                            int(nbuf), int(ht), int(wd), _ptr(counter), _stream()),
               "dba_depth_filter")
    return counter


def iproj(poses, disps, intrinsics):
    """droid.cpp:218-227."""
    for x, nm in ((poses, "poses"), (disps, "disps"), (intrinsics, "intrinsics")):
        _check(x, nm, torch.float32)
    nm_, ht, wd = disps.shape
    points = torch.zeros(nm_, ht, wd, 3, dtype=disps.dtype, device=disps.device)
    _lib.check(_lib.load().dba_iproj(_ptr(poses), _ptr(disps), _ptr(intrinsics), int(nm_), int(ht), int(wd),
                                     _ptr(points), _stream()), "dba_iproj")
    return points


def _vol_dtype(t):
    if t.dtype == torch.float16:
        return DBA_F16
    if t.dtype == torch.float32:
        return DBA_F32
    raise RuntimeError("corr_index: volume dtype %s not supported on the MI355X path (half / float)" % t.dtype)


class _ShadowStore:
    """flow-aligned shadow volumes of ONE pyramid level shape on one device: a slot-addressed store (like CorrBlock's), the
    sampled signature of the edge each slot holds, and who refers to it"""

    def __init__(self, lvl, h1, w1, h2l, w2l, dtype, device, nsig):
        self.lvl, self.h1, self.w1, self.h2l, self.w2l = lvl, h1, w1, h2l, w2l
        self.dtype, self.device = dtype, device
        self.hw1p = _lib.load().dba_corr_sheared_plane_elems(int(h1), int(w1))
        self.store = None                    # [cap, h2l, w2l, hw1p]
        self.sig = None                      # [cap, nsig] int16: sampled elements of the edge in the slot
        self.ref = []                        # per slot: live tensors that use it
        self.filled = []                     # per slot: holds an edge (its signature is valid)
        self.last = []                       # per slot: tick of the last use (eviction order among unreferenced slots)
        self.nsig = nsig
        self.event = None                    # (stream, event) of the last re-layout pass

    def slot_bytes(self):
        return self.h2l * self.w2l * self.hw1p * 2

    def bytes_held(self):
        return 0 if self.store is None else self.store.numel() * self.store.element_size()

    def capacity(self):
        return 0 if self.store is None else int(self.store.shape[0])

    def grow(self, cap):
        new = torch.empty(cap, self.h2l, self.w2l, self.hw1p, dtype=self.dtype, device=self.device)
        sig = torch.zeros(cap, self.nsig, dtype=torch.int16, device=self.device)
        old = self.capacity()
        if old:
            new[:old].copy_(self.store)
            sig[:old].copy_(self.sig)
        self.store, self.sig = new, sig
        self.ref += [0] * (cap - old)
        self.filled += [False] * (cap - old)
        self.last += [0] * (cap - old)


class _VolumeShadows:
    """Flow-aligned shadows of reference-layout pyramid levels, for callers that run the reference's OWN CorrBlock
    (dbaf/modules/corr.py:24-50, no import swapped) against this module.

    In the reference layout [n, y1, x1, y2, x2] a lookup touches 512 different cache lines per wave and level (0.08 of the
    HBM roofline); in the flow-aligned layout of csrc/corr_sheared.hip it streams (0.47).  CorrBlock.__call__ looks the
    SAME level tensors up once per update() for as long as the graph stands, so a level that keeps being asked about gets
    a shadow (one pass of corr_shear_kernel over the edges that need it) and later lookups are served from it -- bit for bit
    the same result.  The shadows of a level shape live in ONE slot-addressed store; a tensor's entry is its slot table.

    What a keyframe does to this: torch.cat in add_factors and the boolean index in rm_factors hand over NEW tensors, whose
    edges are mostly the old ones.  Two policies:
      * default: a new tensor starts over; its shadow is built when it has been looked up `min_uses` times (13: the
        re-layout of a whole 96-edge window costs what 13 lookups save, so a window that changes every 6 updates never
        pays for one -- the ski-rental rule; DBA_ZERO_EDIT_SHADOW_USES);
      * DBA_ZERO_EDIT_SHADOW_MATCH=1 (`match`): the edges of a new tensor are matched to the shadows already held by a
        SIGNATURE -- 128 elements sampled per edge and level, compared exactly -- and only the unmatched (new) edges are
        re-laid out, at the tensor's second use.  A matched edge is trusted to be the same volume: true for anything
        cat / index produce, and for two different correlation volumes to agree in 128 sampled half values is not a
        practical event, but it is a sampled comparison, not a proof -- hence opt-in.  In-place writes are never matched
        (the tensor's entry is dropped and its edges are re-laid out).
    Memory: the stores double the pyramid; a byte budget (DBA_ZERO_EDIT_SHADOW_BYTES, default 16 GiB) bounds them, slots no
    live tensor refers to are reused least-recently-used first.  DBA_ZERO_EDIT_SHADOW=0 switches everything off."""

    NSIG = 128

    def __init__(self):
        import os
        self.enabled = os.environ.get("DBA_ZERO_EDIT_SHADOW", "1") != "0"
        self.match = os.environ.get("DBA_ZERO_EDIT_SHADOW_MATCH", "0") == "1"
        uses = os.environ.get("DBA_ZERO_EDIT_SHADOW_USES")
        self._min_uses = max(2, int(uses)) if uses else None
        self.budget = int(float(os.environ.get("DBA_ZERO_EDIT_SHADOW_BYTES", 16 * 2 ** 30)))
        self.seen = {}      # id(volume) -> entry (see _entry)
        self.stores = {}    # (device, dtype, lvl, h1, w1) -> _ShadowStore
        self.builds = 0     # re-layout passes
        self.built_edges = 0
        self.matched_edges = 0
        self.hits = 0
        self.tick = 0
        self._sig_idx = {}

    @property
    def min_uses(self):
        return self._min_uses if self._min_uses is not None else (2 if self.match else 13)

    @min_uses.setter
    def min_uses(self, v):
        self._min_uses = None if v is None else max(2, int(v))

    @staticmethod
    def level_of(volume):
        """pyramid level of a reference-layout tensor [n, h1, w1, h2 >> lvl, w2 >> lvl] -- inferred from the shapes, which
        is only possible when source and target maps have the same size (h2 == h1, w2 == w1: CorrBlock's only use in the
        reference, covisible_graph.py:127-131); anything else gets no shadow"""
        n, h1, w1, h2l, w2l = volume.shape
        for lvl in range(8):
            if (h1 >> lvl) == h2l and (w1 >> lvl) == w2l and h2l >= 1 and w2l >= 1:
                return lvl
        return None

    def bytes_held(self):
        return sum(st.bytes_held() for st in self.stores.values())

    def clear(self):
        """drop every shadow and every entry (tests; a caller that wants the memory back)"""
        self.seen.clear()
        self.stores.clear()

    def _release(self, key):
        ent = self.seen.pop(key, None)
        if ent is not None and ent["slots_host"] is not None:
            st = ent["store"]
            for sl in ent["slots_host"]:
                st.ref[sl] -= 1

    def _signature(self, volume):
        """[n, NSIG] int16: the elements of every edge at NSIG fixed positions spread over its level volume"""
        n = volume.shape[0]
        per = volume[0].numel()
        key = (per, volume.device)
        idx = self._sig_idx.get(key)
        if idx is None:
            k = torch.arange(self.NSIG, dtype=torch.int64)
            idx = ((k * 2654435761 + 97) % per).to(volume.device)
            self._sig_idx[key] = idx
        return volume.reshape(n, per).index_select(1, idx).view(torch.int16)

    def _slots_for(self, st, need):
        """`need` slots no live tensor refers to: empty ones first, then the least recently used; grows the store inside
        the byte budget (old and new store coexist while one is copied into the other); None if there is no room"""
        free = [s for s in range(st.capacity()) if st.ref[s] == 0]
        if len(free) < need:
            short = need - len(free)
            want = st.capacity() + max(short, st.capacity() // 4)
            others = self.bytes_held() - st.bytes_held()
            cap = min(want, int((self.budget - others) // st.slot_bytes()))
            if cap < st.capacity() + short:
                return None
            st.grow(cap)
            free = [s for s in range(st.capacity()) if st.ref[s] == 0]
        free.sort(key=lambda s: (st.filled[s], st.last[s]))
        return free[:need]

    def lookup(self, volume, radius):
        """-> (shadow store, slots int32 [n] on the device, lvl) when this call should be served from shadows, else None"""
        if not self.enabled or int(radius) != 3 or volume.dtype != torch.float16 or volume.dim() != 5:
            return None
        key = id(volume)
        ent = self.seen.get(key)
        if ent is not None and (ent["ref"]() is not volume or ent["version"] != volume._version):
            self._release(key)        # another object behind a recycled id, or written in place: nothing of it is trusted
            ent, fresh_object = None, False
        else:
            fresh_object = ent is None
        if ent is None:
            lvl = self.level_of(volume)
            if lvl is None:
                return None
            import weakref
            ent = dict(ref=weakref.ref(volume, lambda _r, k=key: self._release(k)), version=volume._version, uses=0,
                       slots=None, slots_host=None, store=None, lvl=lvl, matchable=fresh_object)
            self.seen[key] = ent
        ent["uses"] += 1
        self.tick += 1
        n, h1, w1, h2l, w2l = (int(x) for x in volume.shape)
        if ent["slots"] is None:
            skey = (volume.device, volume.dtype, ent["lvl"], h1, w1)
            st = self.stores.get(skey)
            can_match = self.match and ent["matchable"] and st is not None and st.capacity() > 0
            # a level looked up once (motion_filter's one-edge block) is not worth a re-layout -- unless most of it is
            # already here (match mode: the tensors a graph change creates are served from shadows at their first use)
            if ent["uses"] < self.min_uses and not (can_match and ent["uses"] == 1):
                return None
            if st is None:
                st = self.stores[skey] = _ShadowStore(ent["lvl"], h1, w1, h2l, w2l, volume.dtype, volume.device, self.NSIG)
            sig = self._signature(volume)
            mapping = [-1] * n
            if can_match:
                filled = torch.tensor(st.filled, device=volume.device)
                eq = (sig[:, None, :] == st.sig[None, :, :]).all(-1) & filled[None, :]          # [n, cap]
                hit = eq.any(1)
                first = eq.float().argmax(1)
                mapping = [int(s) if h else -1 for s, h in zip(first.tolist(), hit.tolist())]   # (one small D2H per new tensor)
            todo = [e for e in range(n) if mapping[e] < 0]
            if ent["uses"] < self.min_uses and 4 * len(todo) > n:
                return None       # more than a quarter unknown: this tensor waits for its min_uses-th lookup like any other
            if todo:
                for s_ in set(m for m in mapping if m >= 0):
                    st.ref[s_] += 1                     # (held while slots are chosen: a matched slot must not be handed out)
                slots_new = self._slots_for(st, len(todo))
                for s_ in set(m for m in mapping if m >= 0):
                    st.ref[s_] -= 1
                if slots_new is None:
                    ent["uses"] = -(1 << 40)             # over budget: the direct kernel serves this tensor from now on (no
                    return None                          # signature / match work, no D2H, at every later lookup)
                src = torch.tensor(todo, dtype=torch.int32, device=volume.device)
                dst = torch.tensor(slots_new, dtype=torch.int32, device=volume.device)
                _lib.check(_lib.load().dba_corr_shear_level_slots(_ptr(volume), _ptr(st.store), _ptr(src), _ptr(dst),
                                                                   len(todo), h1, w1, h2l, w2l, ent["lvl"], _stream()),
                           "dba_corr_shear_level_slots")
                st.sig.index_copy_(0, dst.long(), sig.index_select(0, src.long()))
                st.event = (torch.cuda.current_stream(), torch.cuda.Event())
                st.event[1].record(st.event[0])
                for e, sl in zip(todo, slots_new):
                    mapping[e] = sl
                    st.filled[sl] = True
                self.builds += 1
                self.built_edges += len(todo)
            self.matched_edges += n - len(todo)
            for sl in mapping:
                st.ref[sl] += 1
            ent["slots_host"], ent["store"] = mapping, st
            ent["slots"] = torch.tensor(mapping, dtype=torch.int32, device=volume.device)
        else:
            self.hits += 1
        st = ent["store"]
        for sl in ent["slots_host"]:
            st.last[sl] = self.tick
        if st.event is not None and st.event[0] != torch.cuda.current_stream():
            torch.cuda.current_stream().wait_event(st.event[1])   # re-laid out on another stream: order this lookup behind it
        return st.store, ent["slots"], ent["lvl"]


_SHADOWS = _VolumeShadows()


def corr_index_forward(volume, coords, radius):
    """droid.cpp:231-239: volume [n,h1,w1,h2,w2] (half/float), coords [n,2,h1,w1] f32 -> [corr]."""
    _check(volume, "volume")
    _check(coords, "coords", torch.float32)
    n, h1, w1, h2, w2 = volume.shape
    r = int(radius)
    corr = torch.empty(n, 2 * r + 1, 2 * r + 1, h1, w1, dtype=volume.dtype, device=volume.device)
    sh = _SHADOWS.lookup(volume, r) if n > 0 else None
    if sh is not None:
        store, slots, lvl = sh
        _lib.check(_lib.load().dba_corr_lookup_level_sheared_slots(_ptr(store), _ptr(slots), _ptr(coords), _ptr(corr), int(n),
                                                                   int(h1), int(w1), int(h1), int(w1), lvl, r, _stream()),
                   "dba_corr_lookup_level_sheared_slots")
        return [corr]
    _lib.check(_lib.load().dba_corr_index_forward(_ptr(volume), _ptr(coords), _ptr(corr), int(n), int(h1), int(w1),
                                                  int(h2), int(w2), r, _vol_dtype(volume), _stream()),
               "dba_corr_index_forward")
    return [corr]


def corr_index_backward(volume, coords, corr_grad, radius):
    """droid.cpp:241-252 (training only)."""
    _check(volume, "volume")
    _check(coords, "coords", torch.float32)
    _check(corr_grad, "corr_grad")
    n, h1, w1, h2, w2 = volume.shape
    vg = torch.zeros(volume.shape, dtype=torch.float32, device=volume.device)
    cg = corr_grad.float().contiguous()
    _lib.check(_lib.load().dba_corr_index_backward(_ptr(coords), _ptr(cg), _ptr(vg), int(n), int(h1), int(w1),
                                                   int(h2), int(w2), int(radius), _stream()),
               "dba_corr_index_backward")
    return [vg.to(volume.dtype)]


def altcorr_forward(fmap1, fmap2, coords, radius):
    """droid.cpp:254-264: fmap [B,H,W,C] channels-last (float or half, like the reference's
    AT_DISPATCH_FLOATING_TYPES_AND_HALF), coords [B,S,H1,W1,2] f32 -> [corr [B,S,rd*rd,H1,W1]] of the maps' dtype."""
    _check(fmap1, "fmap1")
    _check(fmap2, "fmap2")
    _check(coords, "coords", torch.float32)
    if fmap1.dtype != fmap2.dtype or fmap1.dtype not in (torch.float32, torch.float16):
        raise RuntimeError("altcorr_forward: fmap1 / fmap2 must both be float32 or both float16")
    B, S, H1, W1, _ = coords.shape
    _, H2, W2, C = fmap2.shape
    r = int(radius)
    corr = torch.empty(B, S, (2 * r + 1) ** 2, H1, W1, dtype=fmap1.dtype, device=coords.device)
    _lib.check(_lib.load().dba_altcorr_forward_t(_ptr(fmap1), _ptr(fmap2), _ptr(coords), _ptr(corr), int(B), int(S),
                                                 int(H1), int(W1), int(H2), int(W2), int(C), r, _vol_dtype(fmap1),
                                                 _stream()), "dba_altcorr_forward")
    return [corr]


def altcorr_backward(fmap1, fmap2, coords, corr_grad, radius):
    """droid.cpp:266-278 (training only; dead in the DBA-Fusion runtime).  Returns
    [fmap1_grad, fmap2_grad, coords_grad]; coords_grad is all zeros like the reference's (never written)."""
    _check(fmap1, "fmap1")
    _check(fmap2, "fmap2")
    _check(coords, "coords", torch.float32)
    _check(corr_grad, "corr_grad")
    dt = fmap1.dtype   # half inputs (the reference dispatches them too): the adjoint runs in float, results are cast back
    if dt != torch.float32:
        fmap1, fmap2, corr_grad = fmap1.float(), fmap2.float(), corr_grad.float()
    B, S, H1, W1, _ = coords.shape
    _, H2, W2, C = fmap2.shape
    g1 = torch.zeros(B, H1, W1, C, dtype=torch.float32, device=fmap1.device)
    g2 = torch.zeros(B, H2, W2, C, dtype=torch.float32, device=fmap1.device)
    gc = torch.zeros(B, S, H1, W1, 2, dtype=torch.float32, device=fmap1.device)
    _lib.check(_lib.load().dba_altcorr_backward(_ptr(fmap1), _ptr(fmap2), _ptr(coords), _ptr(corr_grad), _ptr(g1),
                                                _ptr(g2), int(B), int(S), int(H1), int(W1), int(H2), int(W2),
                                                int(C), int(radius), _stream()), "dba_altcorr_backward")
    return [g1.to(dt), g2.to(dt), gc]


# ---- the stateless operators come from the compiled adapter when it is built (same C ABI calls as the functions above) ----
_ctypes_impl = dict(frame_distance=frame_distance, projmap=projmap, depth_filter=depth_filter, iproj=iproj,
                    corr_index_backward=corr_index_backward, altcorr_forward=altcorr_forward, altcorr_backward=altcorr_backward)
if compiled is not None:
    frame_distance = compiled.frame_distance
    projmap = compiled.projmap
    depth_filter = compiled.depth_filter
    iproj = compiled.iproj
    corr_index_backward = compiled.corr_index_backward
    altcorr_forward = compiled.altcorr_forward
    altcorr_backward = compiled.altcorr_backward
