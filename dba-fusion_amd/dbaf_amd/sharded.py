"""Edge-sharded dense bundle adjustment over the GPUs of one node.

The reference is single-GPU (no NCCL anywhere); this is new design for the MI355X node:
  * edges are sharded BY SOURCE FRAME (all edges leaving frame k live on one rank), so the per-frame depth
    block C, w, Q, the rows E and the Schur products E Q E^T of a frame are rank-local;
  * the only coupling is the reduced camera system: ONE all-reduce (sum) of H [6P,6P] + b [6P] in float64 per
    Gauss-Newton iteration (83 KB at P = 24) over RCCL/xGMI -- latency-bound, not bandwidth-bound;
  * every rank then solves the identical system redundantly (identical bits in -> identical dx out; RCCL
    delivers the same reduced buffer to all ranks), retracts all poses, back-substitutes the depths of the
    frames it owns; the updated depth maps are exchanged once per call with ONE all-gather of the rows each rank
    owns (64 KF / 8 ranks: 8 rows x 16 KB per rank), after which the replicas are bit-identical;
  * the IMU / GNSS fusion path (`ShardedBACore`, mirroring droid_backends.BACore as DepthVideo.ba drives it,
    dbaf/depth_video.py:469-559): the reduced system is summed onto rank 0 (one RCCL reduce), handed to the host
    there (GTSAM runs on rank 0 only), the externally solved dx is broadcast, every rank retracts.
The stage executor is pluggable so the partition / exchange logic is testable on CPU with the gloo backend
(tests/test_sharded_cpu.py drives it with a CPU stand-in); the product executor is `HipStages` (C ABI).
"""
import ctypes
import os

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


def window_fpose(ii, jj, t0, t1):
    """Pose-level skyline of the reduced camera system of the COMPLETE graph (what ba_prepare_kernel builds on the device
    for one rank's edges, csrc/ba_kernels.hip): the window poses in S_i = {targets of the edges leaving frame i} U {i}
    are mutually coupled (pose blocks, and the Schur products of frame i), so fpose[a] = min over the sets containing a
    of min S_i.  int32 [t1 - t0], pose indices relative to t0."""
    ii = np.asarray(ii, np.int64)
    jj = np.asarray(jj, np.int64)
    P = int(t1) - int(t0)
    big = np.iinfo(np.int32).max
    mins = {}
    for f in np.unique(ii):
        p = int(f) - t0
        mins[int(f)] = p if 0 <= p < P else big
    for f, g in zip(ii, jj):
        tg = int(g) - t0
        if 0 <= tg < P:
            mins[int(f)] = min(mins[int(f)], tg)
    fp = np.arange(P, dtype=np.int64)
    for f, g in zip(ii, jj):
        tg = int(g) - t0
        if 0 <= tg < P:
            fp[tg] = min(fp[tg], mins[int(f)])
    for f, m in mins.items():
        p = f - t0
        if 0 <= p < P:
            fp[p] = min(fp[p], m)
    return fp.astype(np.int32)


def hand_round_id(dist, make_id):
    """the communicator's 128-byte unique id: made by rank 0 (make_id() -> bytes | None), the same bytes on every rank of `dist`
    afterwards (None everywhere if rank 0 could not make one).  torch.distributed is only this set-up channel."""
    box = [make_id() if dist.get_rank() == 0 else None]
    dist.broadcast_object_list(box, src=0)
    idb = box[0]
    if idb is not None and len(idb) != 128:
        raise RuntimeError("a communicator id is 128 bytes, got %d" % len(idb))
    return idb


class _Comms:
    """RCCL communicators of libdba_hip.so (dba_comm_*), one per process group: the collectives of a sharded ba are then
    issued by the library itself on the caller's stream, inside dba_ba_sharded_run, instead of by Python between stage
    calls.  The unique id is made by rank 0 and handed round through the process group ONCE (torch.distributed is only the
    set-up channel).  DBA_SHARDED_IN_STREAM=0 keeps the staged path."""
    _by_group = {}
    enabled = os.environ.get("DBA_SHARDED_IN_STREAM", "1") != "0"

    @classmethod
    def for_dist(cls, dist, device):
        """a dba_comm* for `dist` if it is a real device process group (backend nccl = RCCL), else None"""
        if not cls.enabled or dist is None or not hasattr(dist, "get_backend"):
            return None
        try:
            if str(dist.get_backend()) != "nccl":
                return None
        except Exception:
            return None
        key = (id(dist), str(device))
        ent = cls._by_group.get(key)
        if ent is None:
            lib = _lib.load()
            world, rank = dist.get_world_size(), dist.get_rank()

            def make_id():
                buf = (ctypes.c_ubyte * 128)()
                return bytes(buf) if lib.dba_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)) == 0 else None
            idbytes = hand_round_id(dist, make_id)
            if idbytes is None:
                cls._by_group[key] = ent = (None,)
                return None
            idb = (ctypes.c_ubyte * 128).from_buffer_copy(idbytes)
            comm = ctypes.c_void_p()
            with torch.cuda.device(device):
                rc = lib.dba_comm_create(ctypes.cast(idb, ctypes.c_void_p), world, rank, ctypes.byref(comm))
            # every rank must take the same route: if the communicator could not be made on ANY of them, all keep the staged
            # path (the collectives of torch.distributed between stage calls)
            ok = torch.tensor([1 if (rc == 0 and comm.value) else 0], dtype=torch.int32, device=device)
            dist.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if rc == 0 and comm.value:
                    lib.dba_comm_destroy(comm)
                import warnings
                warnings.warn("dbaf_amd.sharded: the library's own RCCL communicator could not be created on every rank (%s); "
                              "the sharded BA keeps the staged path" % (_lib.load().dba_last_error().decode() if rc else "another rank failed"))
                cls._by_group[key] = ent = (None,)
                return None
            cls._by_group[key] = ent = (comm,)
        return ent[0]

    @classmethod
    def any_existing(cls):
        """a communicator this process has already made (None if the sharded BA has not run in-stream): never collective"""
        for ent in cls._by_group.values():
            if ent[0] is not None:
                return ent[0]
        return None

    @staticmethod
    def info(comm):
        """(ranks the communicator spans, this process's rank in it) as RCCL reports them, or None without a communicator"""
        if comm is None:
            return None
        w, r = ctypes.c_int(0), ctypes.c_int(0)
        if _lib.load().dba_comm_info(comm, ctypes.byref(w), ctypes.byref(r)) != 0:
            return None
        return w.value, r.value

    @classmethod
    def close(cls):
        lib = _lib.load()
        for ent in cls._by_group.values():
            if ent[0] is not None:
                lib.dba_comm_destroy(ent[0])
        cls._by_group.clear()


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
        # depth exchange: rank r contributes the rows (frames) it owns, padded to the largest share
        self.rows_of = [np.array(sorted(k for k, r in self.owner.items() if r == q), np.int64) for q in range(world)]
        self.kmax = max(1, max(len(r) for r in self.rows_of))
        self.fpose = window_fpose(self.ii_all, self.jj_all, self.t0, self.t1)
        self._dev = {}
        self.schur_form = None   # Schur kernel form of the COMPLETE graph (HipStages asks the library once)
        self._solver_plan = None  # which skyline-solver variant solved the summed system in the first call (HipStages)

    def _on(self, device):
        key = str(device)
        if key not in self._dev:
            flat = np.concatenate([np.pad(r, (0, self.kmax - len(r)), constant_values=-1) for r in self.rows_of])
            self._dev[key] = dict(owned=torch.from_numpy(self.owned).to(device),
                                  eta_rows=torch.from_numpy(self.eta_rows).to(device),
                                  kx=torch.from_numpy(self.kx_global).to(device),
                                  fpose=torch.from_numpy(self.fpose).to(device),
                                  my_rows=torch.from_numpy(self.rows_of[self.rank]).to(device),
                                  all_rows=torch.from_numpy(flat[flat >= 0]).to(device),        # frames, rank-major
                                  all_slots=torch.from_numpy(np.nonzero(flat >= 0)[0]).to(device))  # their gather slots
        return self._dev[key]

    def band_index(self, device, hb_len):
        """Flat indices, into the float64 view [H (6P x 6P, row-major) | pad | b (6P)] of length hb_len, of what the
        exchange step has to move: the lower triangle of H inside the pose-level skyline of the COMPLETE graph (the solvers
        read nothing else of H: lower triangle, columns from 6 fpose[p] on) and b.  64 KF / 512 edges: 19 k of 144 k
        doubles (153 KB instead of 1.15 MB per all-reduce)."""
        key = ("band", str(device), int(hb_len))
        d = self._on(device)
        if key not in d:
            P = self.t1 - self.t0
            n6 = 6 * P
            rows, cols = [], []
            for p in range(P):
                c0 = 6 * int(self.fpose[p])
                for a in range(6):
                    r = 6 * p + a
                    cc = np.arange(c0, r + 1)
                    rows.append(np.full(cc.shape, r))
                    cols.append(cc)
            flat = np.concatenate(rows) * n6 + np.concatenate(cols) if rows else np.zeros(0, np.int64)
            flat = np.concatenate([flat, hb_len - n6 + np.arange(n6)]).astype(np.int64)
            d[key] = (torch.from_numpy(flat).to(device), None)
        return d[key][0]

    def exchange_system(self, hb, dist):
        """the one exchange step of a Gauss-Newton iteration: sum of the partial [H | b] over the ranks, in place.
        Systems of 32 poses and more move only the skyline band (gather -> all-reduce -> scatter: two small launches
        for 7-9 x fewer bytes on the links); smaller ones go as they are (83 KB at 24 poses: latency-bound either way).
        DBA_BAND_EXCHANGE=0 / 1 forces the choice."""
        if dist is None:
            return
        import os
        mode = os.environ.get("DBA_BAND_EXCHANGE")
        n6 = 6 * (self.t1 - self.t0)
        band = (n6 >= 192) if mode is None else (mode == "1")
        if not band:
            dist.all_reduce(hb)
            return
        idx = self.band_index(hb.device, hb.numel())
        packed = hb.take(idx)
        dist.all_reduce(packed)
        hb.put_(idx, packed)

    def merge_disps(self, disps, dist):
        """all-gather of the depth maps each rank owns (and has just updated) -> coherent replicas.
        [kmax, h, w] per rank; slots beyond a rank's share are padding."""
        if dist is None:
            return
        d = self._on(disps.device)
        key = ("merge", tuple(disps.shape[1:]), disps.dtype)
        if key not in d:   # staging buffers live as long as the window object (one allocation, not one per call)
            d[key] = (disps.new_zeros((self.kmax,) + tuple(disps.shape[1:])),
                      disps.new_empty((self.world * self.kmax,) + tuple(disps.shape[1:])))
        send, recv = d[key]
        n_mine = int(d["my_rows"].numel())
        if n_mine:
            torch.index_select(disps, 0, d["my_rows"], out=send[:n_mine])
        dist.all_gather_into_tensor(recv, send)
        disps.index_copy_(0, d["all_rows"], recv.index_select(0, d["all_slots"]))

    def ba(self, poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, iterations, lm, ep, dist,
           stages=None, alpha=0.05, motion_only=False):
        """In-place sharded BA.  targets/weights/ii/jj are THIS rank's edges; eta is the global
        [|kx|,h,w] (or [1,h,w]) damping; poses/disps are replicated and stay coherent on return."""
        if stages is None:
            if getattr(self, "_hip_stages", None) is None:
                self._hip_stages = HipStages()
            stages = self._hip_stages
        with _FormPin(stages, self):   # the form the complete graph would get, on every rank, for this call only
            return self._ba(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, iterations, lm, ep, dist,
                            stages, alpha, motion_only)

    def _ba(self, poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, iterations, lm, ep, dist, stages,
            alpha, motion_only):
        d = self._on(poses.device)
        eta2 = eta.reshape(-1, eta.shape[-2], eta.shape[-1])
        # (one rank: every row of the global damping is this rank's, in order -- nothing to select)
        whole = self.world == 1 and eta2.shape[0] == len(self.eta_rows)
        eta_loc = eta2 if (eta2.shape[0] == 1 or whole) else eta2.index_select(0, d["eta_rows"]).contiguous()
        n6 = 6 * (self.t1 - self.t0)
        # the whole call as ONE enqueued sequence of the library (stage 0, front / exchange / back per iteration, depth
        # all-gather): whenever the exchange is something the library can issue itself -- nothing (one rank), its own RCCL
        # communicator, or the peer-read kernel
        route = self._in_stream_route(stages, dist, poses.device)
        ctx = stages.begin(poses, disps, intrinsics, disps_sens, targets, weights, eta_loc, ii, jj, d["owned"],
                           self.t0, self.t1, alpha, **({"defer_prepare": True} if route is not None else {}))
        if isinstance(ctx, dict):
            ctx["fpose"] = d["fpose"]      # the complete graph's skyline for the solver (the summed system has it)
            ctx["solver_hint"] = 1 if self._solver_plan == 1 else 0
        if route is not None:
            return self._ba_in_stream(route, ctx, stages, d, disps, dist, iterations, lm, ep, motion_only, n6)
        inplace = getattr(stages, "system_view", None)
        for _ in range(int(iterations)):
            stages.linearize_reduce(ctx, motion_only)
            hb = inplace(ctx) if inplace is not None else None
            if hb is not None:                              # float64 view of [H | pad | b] inside the workspace
                self.exchange_system(hb, dist)              # RCCL sum over xGMI, in place (band only on large windows)
            else:
                hb = stages.get_system(ctx)                 # float64 [n6*n6 + n6], this rank's partial sums
                if dist is not None:
                    dist.all_reduce(hb)                     # (gloo in the CPU tests)
                stages.set_system(ctx, hb)
            if hasattr(stages, "solve_update"):
                stages.solve_update(ctx, lm, ep, update_disps=not motion_only)
            else:
                stages.solve(ctx, lm, ep)
                stages.update(ctx, update_disps=not motion_only)
        # Between iterations a rank only reads the depths of the frames it owns (the source frames of its own
        # edges), so the replicas are made coherent ONCE per call: every rank sends the rows it owns.
        if not motion_only:
            self.merge_disps(disps, dist)
        if hasattr(dist, "check"):   # the peer-read exchange reports a missing peer through a status word: surface it
            dist.check()
        # 30-64 poses: the summed system goes to the skyline solver, whose fall-back variants are queued behind it until it
        # is known that the first one takes this window's structure (one stream synchronisation, after the first call)
        if self._solver_plan is None and 174 < n6 <= 384 and hasattr(stages, "solver_plan") and int(iterations) > 0:
            self._solver_plan = stages.solver_plan(ctx)
        assert hb.numel() >= n6 * n6 + n6
        return stages.finish(ctx)

    def _in_stream_route(self, stages, dist, device):
        """("none" | "comm" | "peer", handle) when dba_ba_sharded_run can carry this call, else None (CPU stages of the tests,
        host-staged / in-process stand-ins for torch.distributed, DBA_SHARDED_IN_STREAM=0)"""
        if not _Comms.enabled or not isinstance(stages, HipStages) or not hasattr(_lib.load(), "dba_ba_sharded_run"):
            return None
        if dist is None:
            return ("none", None)
        from .peer import PeerDist
        if isinstance(dist, PeerDist):
            return ("peer", dist.peer)
        comm = _Comms.for_dist(dist, device)
        return ("comm", comm) if comm is not None else None

    def _ba_in_stream(self, route, ctx, stages, d, disps, dist, iterations, lm, ep, motion_only, n6):
        kind, handle = route
        x = _lib.ShardExchange()
        x.world, x.rank = self.world, self.rank
        keep = []
        if kind == "comm":
            x.comm = handle
            key = ("merge", tuple(disps.shape[1:]), disps.dtype)
            if key not in d:
                d[key] = (disps.new_zeros((self.kmax,) + tuple(disps.shape[1:])),
                          disps.new_empty((self.world * self.kmax,) + tuple(disps.shape[1:])))
            send, recv = d[key]
            x.my_rows, x.n_mine, x.kmax = d["my_rows"].data_ptr(), int(d["my_rows"].numel()), self.kmax
            x.all_rows, x.all_slots, x.n_all = d["all_rows"].data_ptr(), d["all_slots"].data_ptr(), int(d["all_rows"].numel())
            x.send, x.recv = send.data_ptr(), recv.data_ptr()
        elif kind == "peer":
            x.peer_regions = ctypes.cast(handle._regions, ctypes.c_void_p)
            if not hasattr(handle, "_epoch_c"):
                handle._epoch_c = ctypes.c_uint(handle.epoch)
            handle._epoch_c.value = handle.epoch
            x.peer_epoch = ctypes.pointer(handle._epoch_c)
            x.peer_max_doubles, x.peer_status = handle.max_doubles, handle._status.data_ptr()
        if kind != "none":
            mode = os.environ.get("DBA_BAND_EXCHANGE")
            if ((n6 >= 192) if mode is None else (mode == "1")) and "hb" in ctx:
                idx = self.band_index(ctx["hb"].device, ctx["hb"].numel())
                bkey = ("band_buf", int(idx.numel()))
                if bkey not in d:
                    d[bkey] = torch.empty(idx.numel(), dtype=torch.float64, device=idx.device)
                x.band_idx, x.band_len, x.band_buf = idx.data_ptr(), int(idx.numel()), d[bkey].data_ptr()
        stages.run_in_stream(ctx, x, iterations, lm, ep, motion_only)
        if kind == "peer":
            handle.epoch = handle._epoch_c.value
            if not motion_only:
                self.merge_disps(disps, dist)          # (the peer-read exchange carries the reduced system only)
            dist.check()
        if self._solver_plan is None and 174 < n6 <= 384 and int(iterations) > 0:
            self._solver_plan = stages.solver_plan(ctx)
        return stages.finish(ctx)

    def bacore(self, dist, stages=None):
        return ShardedBACore(self, dist, stages)


class ShardedBACore:
    """droid_backends.BACore over the ranks of a node, for the IMU / GNSS fusion path
    (/root/reference/dbaf/depth_video.py:469-559; src/bacore.h:4-70):

        init(...)        this rank's edges (targets / weights / ii / jj), replicated poses / disps, GLOBAL eta
        hessian(H, v)    every rank linearises and Schur-reduces its share (alpha = 0.001, droid_kernels.cu:1872); the
                         partial systems are summed onto rank 0 with one RCCL reduce; rank 0 copies the sum into the
                         caller's CPU float64 H [6P,6P], v [6P] (through pinned staging) -- other ranks' H, v stay as
                         they are (GTSAM runs on rank 0 only)
        retract(dx)      rank 0 passes the externally solved CPU float64 dx [6P]; it is broadcast, every rank retracts
                         all poses and back-substitutes the depths of the frames it owns with the E, Q, w its last
                         hessian() cached; owned depth rows are all-gathered.  Returns [dx, None].
    """

    def __init__(self, window, dist, stages=None):
        self.win, self.dist = window, dist
        self.stages = stages if stages is not None else HipStages()
        self.ctx = None

    def init(self, poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, t0, t1, iterations, lm, ep,
             motion_only):
        win = self.win
        assert (int(t0), int(t1)) == (win.t0, win.t1), "ShardedBACore.init: window differs from the partition's"
        d = win._on(poses.device)
        eta2 = eta.reshape(-1, eta.shape[-2], eta.shape[-1])
        eta_loc = eta2 if eta2.shape[0] == 1 else eta2.index_select(0, d["eta_rows"]).contiguous()
        self.poses, self.disps = poses, disps
        with _FormPin(self.stages, win):
            self.ctx = self.stages.begin(poses, disps, intrinsics, disps_sens, targets, weights, eta_loc, ii, jj, d["owned"],
                                         win.t0, win.t1, 0.001)
        n = 6 * (win.t1 - win.t0)
        self.n = n
        pin = poses.is_cuda
        self._Hpin = torch.zeros(n * n + n, dtype=torch.float64, pin_memory=pin)
        self._dxdev = torch.zeros(n, dtype=torch.float64, device=poses.device)

    def hessian(self, H, v):
        st, c, n = self.stages, self.ctx, self.n
        with _FormPin(st, self.win):
            st.linearize_reduce(c, False)
        hb = st.system_view(c) if hasattr(st, "system_view") else None
        if hb is None:
            hb = st.get_system(c)
        if self.dist is not None:
            self.dist.reduce(hb, dst=0)          # RCCL sum onto rank 0 (in place on the workspace view there)
        if self.win.rank != 0:
            return
        if hasattr(st, "symmetrize"):            # the front stage keeps only the lower triangle up; GTSAM gets the full H
            st.symmetrize(c)
        flat = torch.cat([hb[:n * n], hb[-n:]]) if hb.numel() != n * n + n else hb
        self._Hpin.copy_(flat, non_blocking=True)  # one D2H into pinned staging
        if flat.is_cuda:
            torch.cuda.current_stream().synchronize()
        H.copy_(self._Hpin[:n * n].view(n, n)[:H.shape[0], :H.shape[1]])
        v.copy_(self._Hpin[n * n:][:v.shape[0]])

    def retract(self, dx):
        st, c, n = self.stages, self.ctx, self.n
        if self.win.rank == 0:
            if dx is None:
                raise RuntimeError("ShardedBACore.retract: rank 0 must pass the solved dx")
            self._dxdev.copy_(dx.detach().reshape(-1)[:n].to(torch.float64), non_blocking=False)
        if self.dist is not None:
            self.dist.broadcast(self._dxdev, src=0)
        st.set_dx(c, self._dxdev)
        st.update(c, update_disps=True)
        self.win.merge_disps(self.disps, self.dist)
        return [st.finish(c), None]


class HostStagedDist:
    """`dist` stand-in that runs the collectives of the sharded drivers through a process group that cannot take device
    tensors (gloo), by staging through the host: the WHOLE multi-rank path -- partition, front stage, exchange, redundant
    solves, depth all-gather -- can then be executed by several processes on a box with one GPU (bench.py --backend gloo;
    tests/test_gpu_entrypoints.py).  Not a performance path: RCCL is."""

    def __init__(self, dist):
        self._dist = dist
        self.ReduceOp = dist.ReduceOp

    @staticmethod
    def _host(t):
        return t.detach().to("cpu").contiguous()

    def all_reduce(self, t, op=None, **kw):
        h = self._host(t)
        self._dist.all_reduce(h, **({} if op is None else {"op": op}))
        t.copy_(h)

    def reduce(self, t, dst=0, **kw):
        h = self._host(t)
        self._dist.reduce(h, dst=dst)
        if self._dist.get_rank() == dst:
            t.copy_(h)

    def broadcast(self, t, src=0, **kw):
        h = self._host(t)
        self._dist.broadcast(h, src=src)
        t.copy_(h)

    def all_gather_into_tensor(self, out, inp, **kw):
        hi = self._host(inp)
        parts = [torch.empty_like(hi) for _ in range(self._dist.get_world_size())]
        self._dist.all_gather(parts, hi)
        out.copy_(torch.cat([p.reshape((-1,) + tuple(hi.shape[1:])) for p in parts], 0).reshape(out.shape))

    def __getattr__(self, name):   # barrier, get_rank, get_world_size, all_gather_object, destroy_process_group, ...
        return getattr(self._dist, name)


class _FormPin:
    """`with _FormPin(stages, window):` -- the Schur kernel form of the window's COMPLETE graph is pinned on this thread for
    the duration of a sharded call (stage executors without kernels of their own -- the CPU stages of the tests -- have
    nothing to pin)"""

    def __init__(self, stages, window):
        self.stages, self.window = stages, window

    def __enter__(self):
        self.prev = 0
        if hasattr(self.stages, "select_schur_form"):
            w = self.window
            self.prev = self.stages.select_schur_form(w, len(w.ii_all), w.t1 - w.t0) or 0

    def __exit__(self, *exc):
        if hasattr(self.stages, "release_schur_form"):
            self.stages.release_schur_form(self.prev)
        return False


class HipStages:
    """Stage executor over the C ABI (include/dba_hip.h): dba_ba_prepare / linearize / reduce / solve / update."""

    def __init__(self):
        self._cache = {}   # (dims, device) -> workspace, layout and views: one allocation per window shape, not per call

    def select_schur_form(self, window, n_edges, n_poses):
        """A rank's share of the graph has the complete graph's rows per frame (all out-edges of a frame live on its owner)
        but few edges over all the window's frames, so the library's automatic choice would differ from a single GPU's:
        ask it with the complete graph's numbers and pin that form (every rank of a job asks the same question)."""
        lib = _lib.load()
        if window.schur_form is None:
            window.schur_form = int(lib.dba_ba_schur_auto_form(int(n_edges), int(n_poses)))
        prev = int(lib.dba_ba_schur_thread_form())           # whatever this thread had pinned before (usually nothing: 0)
        lib.dba_ba_schur_select_thread(window.schur_form)   # (a per-thread pin: cheap, and other callers keep their choice)
        self._form_pin = window.schur_form
        return prev

    def release_schur_form(self, prev=0):
        """the pin lasts for one sharded call: afterwards the thread has what it had before (the automatic choice, or a pin
        the caller set with dba_ba_schur_select_thread)"""
        _lib.load().dba_ba_schur_select_thread(int(prev))

    def begin(self, poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, owned, t0, t1, alpha,
              defer_prepare=False):
        lib = _lib.load()
        B, ht, wd = disps.shape
        N = int(ii.shape[0])
        dims = (N, int(B), int(ht), int(wd), int(t0), int(t1))
        key = (dims, str(poses.device))
        base = self._cache.get(key)
        if base is None:
            nbytes = lib.dba_ba_workspace_bytes(*dims)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=poses.device)
            _lib.check(lib.dba_ba_workspace_init(*dims, self._p(ws), nbytes, self._s()), "dba_ba_workspace_init")
            lay = _lib.BaLayout()
            _lib.check(lib.dba_ba_get_layout(*dims, ctypes.byref(lay)), "dba_ba_get_layout")
            n = 6 * (int(t1) - int(t0))
            base = dict(ws=ws, nbytes=nbytes, lay=lay, H=ws[lay.H:lay.H + 8 * n * n].view(torch.float64),
                        b=ws[lay.b:lay.b + 8 * n].view(torch.float64))
            # H and b are neighbours in the workspace: one in-place all-reduce covers both (the alignment gap is zeroed
            # once so that it sums to zero; no kernel writes there)
            if lay.b >= lay.H + 8 * n * n and (lay.b - lay.H) % 8 == 0 and lay.b - (lay.H + 8 * n * n) <= 4096:
                ws[lay.H + 8 * n * n:lay.b].zero_()
                base["hb"] = ws[lay.H:lay.b + 8 * n].view(torch.float64)
            self._cache = {key: base}   # (one shape at a time: a new window shape replaces the old workspace)
        ctx = dict(base, lib=lib, dims=dims, poses=poses, disps=disps, intr=intrinsics,
                   dsens=disps_sens, targets=targets, weights=weights, eta=eta.contiguous(), ii=ii, jj=jj,
                   owned=owned.contiguous(), alpha=float(alpha),
                   eta_rows=int(eta.reshape(-1, ht * wd).shape[0]))
        # stage 0 depends on (ii, jj, t0, t1, sizes, Schur form) alone: a window that calls ba() again with the very same edge
        # tensors (same objects, same in-place version -- the workspace is this object's and holds that graph's tables)
        # skips the launch, as droid_backends.ba does
        import weakref
        form = getattr(self, "_form_pin", None)
        pk = base.get("prepared")
        same = (pk is not None and pk[0]() is ii and pk[2]() is jj and pk[1] == ii._version and pk[3] == jj._version
                and pk[4] == (lib.dba_ba_schur_generation(), form))
        # (new tensor objects: stage 0 compares the edge list with the key it left in the workspace and rebuilds only when the
        # graph changed -- dba_ba_prepare_keyed; on the in-stream path that launch belongs to dba_ba_sharded_run)
        ctx["prepared"] = 1 if same else 2
        if not same:
            if not defer_prepare:
                _lib.check(lib.dba_ba_prepare_keyed(self._p(ii), self._p(jj), *dims, ctx["eta_rows"], 1, self._p(ctx["ws"]),
                                                    ctx["nbytes"], self._s()), "dba_ba_prepare_keyed")
            base["prepared"] = (weakref.ref(ii), ii._version, weakref.ref(jj), jj._version,
                                (lib.dba_ba_schur_generation(), form))
        return ctx

    def run_in_stream(self, c, x, iterations, lm, ep, motion_only):
        """dba_ba_sharded_run: stage 0, `iterations` x (front, exchange, back), depth all-gather -- one call, one stream"""
        p = self._p
        _lib.check(c["lib"].dba_ba_sharded_run(p(c["poses"]), p(c["disps"]), p(c["intr"]), p(c["dsens"]), p(c["targets"]),
                                               p(c["weights"]), p(c["eta"]), c["eta_rows"], p(c["ii"]), p(c["jj"]),
                                               p(c["owned"]), *c["dims"], int(iterations), float(lm), float(ep), c["alpha"],
                                               int(bool(motion_only)), p(c.get("fpose")), int(c.get("solver_hint", 0)),
                                               int(c["prepared"]), ctypes.byref(x), p(c["ws"]), c["nbytes"], self._s()),
                   "dba_ba_sharded_run")

    @staticmethod
    def _p(x):
        return ctypes.c_void_p(x.data_ptr()) if x is not None and x.numel() > 0 else None

    @staticmethod
    def _s():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def linearize_reduce(self, c, motion_only):
        lib, p = c["lib"], self._p
        _lib.check(lib.dba_ba_shard_front(p(c["poses"]), p(c["disps"]), p(c["intr"]), p(c["dsens"]), p(c["targets"]),
                                          p(c["weights"]), p(c["eta"]), c["eta_rows"], p(c["ii"]), p(c["jj"]),
                                          p(c["owned"]), *c["dims"], c["alpha"], int(bool(motion_only)), p(c["ws"]),
                                          c["nbytes"], self._s()), "dba_ba_shard_front")

    def solve_update(self, c, lm, ep, update_disps=True):
        p = self._p
        _lib.check(c["lib"].dba_ba_shard_back(p(c["poses"]), p(c["disps"]), p(c["ii"]), p(c["jj"]), p(c["owned"]),
                                              *c["dims"], float(lm), float(ep), int(bool(update_disps)),
                                              p(c.get("fpose")), int(c.get("solver_hint", 0)), p(c["ws"]), c["nbytes"],
                                              self._s()),
                   "dba_ba_shard_back")

    def solver_plan(self, c):
        """which skyline-solver variant took the summed system (meta[7] of the workspace; synchronises the stream: the
        window asks once, after its first call) -> 1: the one-tile-per-thread variant, the fall-back behind it need not be
        queued"""
        lay = _lib.BaLayout()
        c["lib"].dba_ba_get_layout(*c["dims"], ctypes.byref(lay))
        return int(c["ws"][lay.meta + 28:lay.meta + 32].view(torch.int32).item())

    def system_view(self, c):
        return c["hb"] if "hb" in c else None

    def get_system(self, c):
        return torch.cat([c["H"], c["b"]])

    def set_system(self, c, hb):
        n2 = c["H"].numel()
        c["H"].copy_(hb[:n2])
        c["b"].copy_(hb[n2:])

    def symmetrize(self, c):
        _lib.check(c["lib"].dba_ba_symmetrize(*c["dims"], self._p(c["ws"]), c["nbytes"], self._s()), "dba_ba_symmetrize")

    def solve(self, c, lm, ep):
        _lib.check(c["lib"].dba_ba_solve(*c["dims"], float(lm), float(ep), self._p(c["ws"]), c["nbytes"], self._s()),
                   "dba_ba_solve")

    def update(self, c, update_disps=True):
        p = self._p
        _lib.check(c["lib"].dba_ba_update(p(c["poses"]), p(c["disps"]), p(c["ii"]), p(c["jj"]), p(c["owned"]),
                                          *c["dims"], 1, int(bool(update_disps)), None, p(c["ws"]), c["nbytes"],
                                          self._s()), "dba_ba_update")

    def set_dx(self, c, dx64):
        """externally solved update (float64, device) -> the workspace's dx slot (float32, droid_kernels.cu:1929-1930)"""
        n = 6 * (c["dims"][5] - c["dims"][4])
        lay = c["lay"]
        c["ws"][lay.dx:lay.dx + 4 * n].view(torch.float32).copy_(dx64.reshape(-1)[:n].to(torch.float32))

    def finish(self, c):
        n = 6 * (c["dims"][5] - c["dims"][4])
        lay = c["lay"]
        return c["ws"][lay.dx:lay.dx + 4 * n].view(torch.float32).view(-1, 6).clone()
