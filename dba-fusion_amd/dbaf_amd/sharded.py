"""Edge-sharded dense bundle adjustment over the GPUs of one node.

The reference is single-GPU (no NCCL anywhere); this is new design for the MI355X node:
  * edges are sharded BY SOURCE FRAME (all edges leaving frame k live on one rank), so the per-frame depth
    block C, w, Q, the rows E and the Schur products E Q E^T of a frame are rank-local;
  * the only coupling is the reduced camera system: ONE all-reduce (sum) of H [6P,6P] + b [6P] in float64 per
    Gauss-Newton iteration (83 KB at P = 24) over RCCL/xGMI -- latency-bound, not bandwidth-bound;
  * every rank then solves the identical system redundantly (identical bits in -> identical dx out; RCCL
    delivers the same reduced buffer to all ranks), retracts all poses, back-substitutes the depths of the
    frames it owns; the per-frame depth updates are exchanged once per call with one more small all-reduce
    (a sum in which exactly one rank contributes a non-zero per element, hence exact).
The stage executor is pluggable so the partition / exchange logic is testable on CPU with the gloo backend
(tests/test_sharded_cpu.py drives it with a CPU stand-in); the product executor is `HipStages` (C ABI).
"""
import ctypes

import numpy as np
import torch

from . import _lib


def partition_source_frames(ii, jj, t0, t1, world):
    """Deterministic balanced assignment of source frames (and edge-less window frames) to ranks.
    Greedy longest-processing-time on out-degree; ties broken by frame id."""
    ii = np.asarray(ii, np.int64)
    kx = np.unique(np.concatenate([np.arange(t0, t1, dtype=np.int64), ii]))
    deg = {int(k): 0 for k in kx}
    for f in ii:
        deg[int(f)] += 1
    order = sorted(deg, key=lambda k: (-deg[k], k))
    load = [0] * world
    count = [0] * world
    owner = {}
    for k in order:
        r = min(range(world), key=lambda q: (load[q], count[q], q))
        owner[k] = r
        load[r] += deg[k]
        count[r] += 1
    return kx, owner


class ShardedWindow:
    def __init__(self, ii, jj, t0, t1, B, world, rank):
        self.ii_all = np.asarray(ii, np.int64)
        self.jj_all = np.asarray(jj, np.int64)
        self.t0, self.t1, self.B, self.world, self.rank = int(t0), int(t1), int(B), int(world), int(rank)
        self.kx_global, self.owner = partition_source_frames(self.ii_all, self.jj_all, t0, t1, world)
        self.local_edges = np.nonzero(np.array([self.owner[int(f)] == rank for f in self.ii_all]))[0]
        owned = np.zeros(B, np.uint8)
        for k, r in self.owner.items():
            if r == rank:
                owned[k] = 1
        self.owned = owned
        # rows of the GLOBAL eta ([|kx_global|, h, w]) that correspond to this rank's kx
        kx_local = np.unique(np.concatenate([np.arange(t0, t1, dtype=np.int64), self.ii_all[self.local_edges]]))
        self.kx_local = kx_local
        self.eta_rows = np.searchsorted(self.kx_global, kx_local)
        self._dev = {}

    def _on(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = dict(owned=torch.from_numpy(self.owned).to(device),
                                  eta_rows=torch.from_numpy(self.eta_rows).to(device),
                                  kx=torch.from_numpy(self.kx_global).to(device))
        return self._dev[key]

    def ba(self, poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, iterations, lm, ep, dist,
           stages=None, alpha=0.05, motion_only=False):
        """In-place sharded BA.  targets/weights/ii/jj are THIS rank's edges; eta is the global
        [|kx|,h,w] (or [1,h,w]) damping; poses/disps are replicated and stay coherent on return."""
        d = self._on(poses.device)
        if stages is None:
            stages = HipStages()
        eta2 = eta.reshape(-1, eta.shape[-2], eta.shape[-1])
        eta_loc = eta2 if eta2.shape[0] == 1 else eta2.index_select(0, d["eta_rows"]).contiguous()
        n6 = 6 * (self.t1 - self.t0)
        ctx = stages.begin(poses, disps, intrinsics, disps_sens, targets, weights, eta_loc, ii, jj, d["owned"],
                           self.t0, self.t1, alpha)
        kmin, kmax = int(self.kx_global[0]), int(self.kx_global[-1]) + 1
        own_rows = d["owned"][kmin:kmax].to(torch.bool)
        before = disps[kmin:kmax].clone()
        inplace = getattr(stages, "system_view", None)
        for _ in range(int(iterations)):
            stages.linearize_reduce(ctx, motion_only)
            hb = inplace(ctx) if inplace is not None else None
            if hb is not None:                              # float64 view of [H | pad | b] inside the workspace
                if dist is not None and self.world > 1:
                    dist.all_reduce(hb)                     # RCCL sum over xGMI, in place: no staging copies
            else:
                hb = stages.get_system(ctx)                 # float64 [n6*n6 + n6], this rank's partial sums
                if dist is not None and self.world > 1:
                    dist.all_reduce(hb)                     # (gloo in the CPU tests)
                stages.set_system(ctx, hb)
            stages.solve(ctx, lm, ep)
            stages.update(ctx, update_disps=not motion_only)
        # Between iterations a rank only reads the depths of the frames it owns (the source frames of its own
        # edges), so the replicas are made coherent ONCE per call: owner-masked deltas, summed over ranks
        # (exactly one non-zero contributor per element, hence exact).
        if not motion_only:
            delta = disps[kmin:kmax] - before
            delta[~own_rows] = 0
            if dist is not None and self.world > 1:
                dist.all_reduce(delta)
            disps[kmin:kmax] = before + delta
        assert hb.numel() >= n6 * n6 + n6
        return stages.finish(ctx)


class HipStages:
    """Stage executor over the C ABI (include/dba_hip.h): dba_ba_prepare / linearize / reduce / solve / update."""

    def begin(self, poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, owned, t0, t1, alpha):
        lib = _lib.load()
        B, ht, wd = disps.shape
        N = int(ii.shape[0])
        dims = (N, int(B), int(ht), int(wd), int(t0), int(t1))
        nbytes = lib.dba_ba_workspace_bytes(*dims)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=poses.device)
        lay = _lib.BaLayout()
        _lib.check(lib.dba_ba_get_layout(*dims, ctypes.byref(lay)), "dba_ba_get_layout")
        ctx = dict(lib=lib, dims=dims, ws=ws, nbytes=nbytes, lay=lay, poses=poses, disps=disps, intr=intrinsics,
                   dsens=disps_sens, targets=targets, weights=weights, eta=eta.contiguous(), ii=ii, jj=jj,
                   owned=owned.contiguous(), alpha=float(alpha),
                   eta_rows=int(eta.reshape(-1, ht * wd).shape[0]))
        _lib.check(lib.dba_ba_prepare(self._p(ii), self._p(jj), *dims, self._p(ws), nbytes, self._s()),
                   "dba_ba_prepare")
        n = 6 * (int(t1) - int(t0))
        ctx["H"] = ws[lay.H:lay.H + 8 * n * n].view(torch.float64)
        ctx["b"] = ws[lay.b:lay.b + 8 * n].view(torch.float64)
        # H and b are neighbours in the workspace: one in-place all-reduce covers both (the alignment gap is zeroed
        # once so that it sums to zero)
        if lay.b >= lay.H + 8 * n * n and (lay.b - lay.H) % 8 == 0 and lay.b - (lay.H + 8 * n * n) <= 4096:
            ws[lay.H + 8 * n * n:lay.b].zero_()
            ctx["hb"] = ws[lay.H:lay.b + 8 * n].view(torch.float64)
        return ctx

    @staticmethod
    def _p(x):
        return ctypes.c_void_p(x.data_ptr()) if x is not None and x.numel() > 0 else None

    @staticmethod
    def _s():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def linearize_reduce(self, c, motion_only):
        lib, p = c["lib"], self._p
        _lib.check(lib.dba_ba_linearize(p(c["poses"]), p(c["disps"]), p(c["intr"]), p(c["dsens"]), p(c["targets"]),
                                        p(c["weights"]), p(c["eta"]), c["eta_rows"], p(c["ii"]), p(c["jj"]),
                                        p(c["owned"]), *c["dims"], c["alpha"], p(c["ws"]), c["nbytes"], self._s()),
                   "dba_ba_linearize")
        _lib.check(lib.dba_ba_reduce(p(c["ii"]), p(c["jj"]), p(c["owned"]), *c["dims"], int(bool(motion_only)),
                                     p(c["ws"]), c["nbytes"], self._s()), "dba_ba_reduce")

    def system_view(self, c):
        return c["hb"] if "hb" in c else None

    def get_system(self, c):
        return torch.cat([c["H"], c["b"]])

    def set_system(self, c, hb):
        n2 = c["H"].numel()
        c["H"].copy_(hb[:n2])
        c["b"].copy_(hb[n2:])

    def solve(self, c, lm, ep):
        _lib.check(c["lib"].dba_ba_solve(*c["dims"], float(lm), float(ep), self._p(c["ws"]), c["nbytes"], self._s()),
                   "dba_ba_solve")

    def update(self, c, update_disps=True):
        p = self._p
        _lib.check(c["lib"].dba_ba_update(p(c["poses"]), p(c["disps"]), p(c["ii"]), p(c["jj"]), p(c["owned"]),
                                          *c["dims"], 1, int(bool(update_disps)), None, p(c["ws"]), c["nbytes"],
                                          self._s()), "dba_ba_update")

    def finish(self, c):
        n = 6 * (c["dims"][5] - c["dims"][4])
        lay = c["lay"]
        return c["ws"][lay.dx:lay.dx + 4 * n].view(torch.float32).view(-1, 6).clone()
