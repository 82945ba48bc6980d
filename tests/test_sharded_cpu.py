"""CPU, world_size 2, gloo: the edge-sharding / exchange logic of dbaf_amd.sharded gives the same
state as the unsharded solve.  The stage executor here is a CPU stand-in built on the oracle (test
infrastructure); on the GPU the same driver runs with HipStages over the C ABI."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist

from dbaf_amd import synthetic as syn
from dbaf_amd.sharded import ShardedWindow, partition_source_frames


class OracleStages:
    """float64 CPU stand-in for HipStages: BACore.hessian / damped solve / BACore.retract of the oracle."""

    def begin(self, poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, owned, t0, t1, alpha):
        return dict(poses=poses, disps=disps, intr=intrinsics, dsens=disps_sens, targets=targets, weights=weights,
                    eta=eta, ii=ii, jj=jj, t0=t0, t1=t1)

    def linearize_reduce(self, c, motion_only):
        from oracle import oracle as orc
        c["core"] = orc.BACore(c["poses"].numpy(), c["disps"].numpy(), c["intr"].numpy(), c["dsens"].numpy(),
                               c["targets"].numpy(), c["weights"].numpy(), c["eta"].numpy(), c["ii"].numpy(),
                               c["jj"].numpy(), c["t0"], c["t1"], 1e-4, 0.1, np.float64)
        c["H"], c["v"] = c["core"].hessian()

    def get_system(self, c):
        return torch.from_numpy(np.concatenate([c["H"].reshape(-1), c["v"]]))

    def set_system(self, c, hb):
        n = c["v"].shape[0]
        a = hb.numpy()
        c["H"], c["v"] = a[:n * n].reshape(n, n).copy(), a[n * n:].copy()

    def solve(self, c, lm, ep):
        L = c["H"] + np.diag(ep + lm * np.diag(c["H"]))
        c["dx"] = np.linalg.solve(L, c["v"])

    def set_dx(self, c, dx64):
        c["dx"] = dx64.numpy().copy()

    def update(self, c, update_disps=True):
        c["core"].retract(c["dx"])
        c["poses"].copy_(torch.from_numpy(c["core"].poses))
        c["disps"].copy_(torch.from_numpy(c["core"].disps))

    def finish(self, c):
        return torch.from_numpy(c["dx"].reshape(-1, 6))


def _window():
    return syn.make_window(*syn.graph_banded(7, 2, extra=[(0, 4)]), 7, 12, 16, seed=5, intr=(8.0, 8.2, 7.7, 5.9),
                           sensor_frac=0.15)


def _t(a, dt=torch.float64):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dt)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W = _window()
    sh = ShardedWindow(W.ii, W.jj, W.t0, W.t1, W.B, world, rank)
    sel = sh.local_edges
    poses, disps = _t(W.poses), _t(W.disps)
    dx = sh.ba(poses, disps, _t(W.intrinsics), _t(W.disps_sens), _t(W.target[sel]), _t(W.weight[sel]), _t(W.eta),
               _t(W.ii[sel], torch.int64), _t(W.jj[sel], torch.int64), 2, W.lm, W.ep, dist, stages=OracleStages())
    # replicas must be coherent
    ref = [torch.zeros_like(disps) for _ in range(world)]
    dist.all_gather(ref, disps)
    assert all(torch.equal(r, ref[0]) for r in ref)
    if rank == 0:
        np.savez(out, poses=poses.numpy(), disps=disps.numpy(), dx=dx.numpy(), nloc=len(sel))
    dist.destroy_process_group()


def _worker_bacore(rank, world, port, out):
    """the fusion path: hessian -> (rank 0: external solve) -> retract, twice (depth_video.py:524-558)"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W = _window()
    sh = ShardedWindow(W.ii, W.jj, W.t0, W.t1, W.B, world, rank)
    sel = sh.local_edges
    poses, disps = _t(W.poses), _t(W.disps)
    core = sh.bacore(dist, stages=OracleStages())
    core.init(poses, disps, _t(W.intrinsics), _t(W.disps_sens), _t(W.target[sel]), _t(W.weight[sel]), _t(W.eta),
              _t(W.ii[sel], torch.int64), _t(W.jj[sel], torch.int64), W.t0, W.t1, 2, W.lm, W.ep, False)
    n = 6 * (W.t1 - W.t0)
    Hs = []
    for _ in range(2):
        H, v = torch.zeros(n, n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
        core.hessian(H, v)
        dx = None
        if rank == 0:
            Hn, vn = H.numpy(), v.numpy()
            Hs.append(Hn.copy())
            dx = torch.from_numpy(np.linalg.solve(Hn + np.diag(W.ep + W.lm * np.diag(Hn)), vn))
        else:
            assert float(H.abs().max()) == 0.0   # the system goes to rank 0 only
        core.retract(dx)
    ref = [torch.zeros_like(disps) for _ in range(world)]
    dist.all_gather(ref, disps)
    assert all(torch.equal(r, ref[0]) for r in ref)
    refp = [torch.zeros_like(poses) for _ in range(world)]
    dist.all_gather(refp, poses)
    assert all(torch.equal(r, refp[0]) for r in refp)
    if rank == 0:
        np.savez(out, poses=poses.numpy(), disps=disps.numpy(), H0=Hs[0], nloc=len(sel))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_partition_is_balanced_and_complete():
    ii, jj = syn.graph_64_512()
    kx, owner = partition_source_frames(ii, jj, 1, 64, 8)
    assert set(owner) == set(int(k) for k in kx)
    loads = [sum(1 for f in ii if owner[int(f)] == r) for r in range(8)]
    assert sum(loads) == 512 and max(loads) - min(loads) <= 8
    # every edge of a source frame lives on one rank
    for f in np.unique(ii):
        assert len({owner[int(f)]}) == 1


def test_sharded_ba_matches_unsharded_world2(tmp_path):
    from oracle import oracle as orc
    out = str(tmp_path / "rank0.npz")
    # one OS process per rank, like the driver launches bench.py (torch.multiprocessing.spawn re-imports the
    # whole pytest session in every child and is ~10x slower here)
    port = _free_port()
    env = dict(os.environ, OMP_NUM_THREADS="2", OMP_WAIT_POLICY="passive", PYTHONPATH=os.pathsep.join(sys.path))
    code = "import sys; import test_sharded_cpu as T; T._worker(int(sys.argv[1]), 2, int(sys.argv[2]), sys.argv[3])"
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(port), out], env=env,
                              cwd=os.path.dirname(os.path.abspath(__file__))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = np.load(out)
    assert 0 < got["nloc"] < 20  # the edge set really was split
    # unsharded reference: same flow (hessian -> damped solve -> retract), two iterations
    W = _window()
    poses, disps = W.poses.astype(np.float64), W.disps.astype(np.float64)
    for _ in range(2):
        core = orc.BACore(poses, disps, W.intrinsics, W.disps_sens, W.target, W.weight, W.eta, W.ii, W.jj, W.t0,
                          W.t1, W.lm, W.ep, np.float64)
        H, v = core.hessian()
        dx = np.linalg.solve(H + np.diag(W.ep + W.lm * np.diag(H)), v)
        core.retract(dx)
        poses, disps = core.poses, core.disps
    np.testing.assert_allclose(got["poses"], poses, rtol=0, atol=1e-9)
    np.testing.assert_allclose(got["disps"], disps, rtol=1e-8, atol=1e-9)


def test_sharded_bacore_matches_unsharded_world2(tmp_path):
    """the IMU-path counterpart (BASELINE config 5): reduce onto rank 0, external solve there, dx broadcast"""
    from oracle import oracle as orc
    out = str(tmp_path / "rank0_core.npz")
    port = _free_port()
    env = dict(os.environ, OMP_NUM_THREADS="2", OMP_WAIT_POLICY="passive", PYTHONPATH=os.pathsep.join(sys.path))
    code = "import sys; import test_sharded_cpu as T; T._worker_bacore(int(sys.argv[1]), 2, int(sys.argv[2]), sys.argv[3])"
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(port), out], env=env,
                              cwd=os.path.dirname(os.path.abspath(__file__))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = np.load(out)
    W = _window()
    core = orc.BACore(W.poses.astype(np.float64), W.disps.astype(np.float64), W.intrinsics, W.disps_sens, W.target,
                      W.weight, W.eta, W.ii, W.jj, W.t0, W.t1, W.lm, W.ep, np.float64)
    for it in range(2):
        H, v = core.hessian()
        if it == 0:
            np.testing.assert_allclose(got["H0"], H, rtol=1e-10, atol=1e-12 * np.abs(H).max())
        core.retract(np.linalg.solve(H + np.diag(W.ep + W.lm * np.diag(H)), v))
    np.testing.assert_allclose(got["poses"], core.poses, rtol=0, atol=1e-9)
    np.testing.assert_allclose(got["disps"], core.disps, rtol=1e-8, atol=1e-9)


def test_window_fpose_is_the_first_coupled_pose_of_the_complete_graph():
    """the skyline table the sharded driver hands to the solver: brute force over the coupling sets
    S_i = {targets of the edges leaving i} U {i} (window poses), on random graphs incl. edges that leave the window"""
    from dbaf_amd.sharded import window_fpose
    rng = np.random.default_rng(4)
    for trial in range(30):
        B = int(rng.integers(6, 40))
        t0 = int(rng.integers(0, 3))
        t1 = int(rng.integers(t0 + 2, B + 1))
        N = int(rng.integers(1, 120))
        ii = rng.integers(0, B, N)
        jj = np.clip(ii + rng.integers(-6, 7, N), 0, B - 1)
        P = t1 - t0
        coupled = np.eye(P, dtype=bool)
        for f in np.unique(ii):
            S = set(int(g) - t0 for g in jj[ii == f] if t0 <= g < t1)
            if t0 <= f < t1:
                S.add(int(f) - t0)
            for a in S:
                for b in S:
                    coupled[a, b] = True
        want = np.array([int(np.nonzero(coupled[a])[0].min()) for a in range(P)], np.int32)
        got = window_fpose(ii, jj, t0, t1)
        assert np.array_equal(got, want), (trial, got, want)


def test_band_index_covers_the_reduced_system_of_the_complete_graph_and_the_exchange_only_moves_it():
    """The band-only exchange (ShardedWindow.exchange_system on 32 poses and more) sums the entries inside the pose-level
    skyline of the lower triangle plus b, and nothing else: (i) every entry the reduced system of the COMPLETE graph can
    have (pose blocks of an edge, Schur fill between the targets of a common source frame) lies inside the index set,
    (ii) an exchange over a fake two-rank `dist` leaves the entries outside untouched and sums the ones inside."""
    import torch
    from dbaf_amd.sharded import ShardedWindow
    rng = np.random.default_rng(9)
    for trial in range(6):
        KF = int(rng.integers(34, 48))
        ii, jj = syn.graph_banded(KF, int(rng.integers(1, 4)), extra=[(0, 5), (7, 2)])
        t0, t1, B = 1, KF, KF + 1
        P, n6 = t1 - t0, 6 * (t1 - t0)
        sh = ShardedWindow(ii, jj, t0, t1, B, 2, 0)
        pad = 3
        hb_len = n6 * n6 + pad + n6
        idx = sh.band_index(torch.device("cpu"), hb_len).numpy()
        inside = np.zeros(hb_len, bool)
        inside[idx] = True
        assert inside[hb_len - n6:].all() and not inside[n6 * n6:hb_len - n6].any()      # all of b, none of the padding
        # (i) structure of the complete graph's reduced system, lower triangle
        need = np.zeros((P, P), bool)
        for f in np.unique(ii):
            S = set(int(g) - t0 for g in jj[ii == f] if t0 <= g < t1)
            if t0 <= f < t1:
                S.add(int(f) - t0)
            for a in S:
                for b in S:
                    need[max(a, b), min(a, b)] = True
        Hin = inside[:n6 * n6].reshape(n6, n6)
        for a in range(P):
            for b in range(a + 1):
                if need[a, b]:
                    blk = Hin[6 * a:6 * a + 6, 6 * b:6 * b + 6]
                    assert (blk[np.tril_indices(6)] if a == b else blk).all(), (trial, a, b)
        assert not np.triu(Hin, 1).any()                                                  # nothing above the diagonal
        assert idx.size < 0.5 * n6 * n6                                                   # it IS a band

        # (ii) the exchange
        class TwoRanks:
            def __init__(self, other):
                self.other = other

            def all_reduce(self, t):
                t += self.other

        mine = torch.from_numpy(rng.standard_normal(hb_len))
        theirs = torch.from_numpy(rng.standard_normal(hb_len))
        before = mine.clone()
        os.environ["DBA_BAND_EXCHANGE"] = "1"
        try:
            sh.exchange_system(mine, TwoRanks(theirs.take(torch.from_numpy(idx))))
        finally:
            del os.environ["DBA_BAND_EXCHANGE"]
        want = before.clone()
        want[torch.from_numpy(idx)] += theirs[torch.from_numpy(idx)]
        assert torch.equal(mine, want)


def _worker_world8(rank, world, port, out):
    """what bench.py --gpus 8 sets up, on CPU: the communicator id made by rank 0 and handed round through the process group
    (dbaf_amd.sharded.hand_round_id: the set-up channel of the library's own RCCL communicator), then one sharded ba over all
    eight ranks"""
    import hashlib
    from dbaf_amd.sharded import hand_round_id
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    made = []

    def make_id():          # (only rank 0 may be asked)
        made.append(rank)
        return hashlib.sha512(b"dba-fusion").digest() * 2
    idb = hand_round_id(dist, make_id)
    assert made == ([0] if rank == 0 else []) and len(idb) == 128
    ids = [None] * world
    dist.all_gather_object(ids, idb)
    assert all(x == ids[0] for x in ids)
    assert hand_round_id(dist, lambda: None) is None          # rank 0 cannot make one: every rank learns it
    W = syn.make_window(*syn.graph_banded(16, 3), 16, 8, 8, seed=9, intr=(6.0, 6.1, 3.9, 4.1))
    sh = ShardedWindow(W.ii, W.jj, W.t0, W.t1, W.B, world, rank)
    sel = sh.local_edges
    counts = [None] * world
    dist.all_gather_object(counts, len(sel))
    assert sum(counts) == W.N and min(counts) > 0             # every rank works, every edge has one owner
    poses, disps = _t(W.poses), _t(W.disps)
    sh.ba(poses, disps, _t(W.intrinsics), _t(W.disps_sens), _t(W.target[sel]), _t(W.weight[sel]), _t(W.eta),
          _t(W.ii[sel], torch.int64), _t(W.jj[sel], torch.int64), 1, W.lm, W.ep, dist, stages=OracleStages())
    ref = [torch.zeros_like(disps) for _ in range(world)]
    dist.all_gather(ref, disps)
    assert all(torch.equal(r, ref[0]) for r in ref)
    if rank == 0:
        np.savez(out, poses=poses.numpy(), disps=disps.numpy())
    dist.destroy_process_group()


def test_eight_ranks_id_hand_round_and_sharded_ba(tmp_path):
    """world_size 8 (the node the scaling bench runs on), gloo: the id hand-round reaches every rank, the partition gives every
    rank edges, the sharded ba equals the unsharded flow"""
    from oracle import oracle as orc
    out = str(tmp_path / "rank0.npz")
    port = _free_port()
    env = dict(os.environ, OMP_NUM_THREADS="1", OMP_WAIT_POLICY="passive", PYTHONPATH=os.pathsep.join(sys.path))
    code = "import sys; import test_sharded_cpu as T; T._worker_world8(int(sys.argv[1]), 8, int(sys.argv[2]), sys.argv[3])"
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(port), out], env=env,
                              cwd=os.path.dirname(os.path.abspath(__file__))) for r in range(8)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    got = np.load(out)
    W = syn.make_window(*syn.graph_banded(16, 3), 16, 8, 8, seed=9, intr=(6.0, 6.1, 3.9, 4.1))
    core = orc.BACore(W.poses.astype(np.float64), W.disps.astype(np.float64), W.intrinsics, W.disps_sens, W.target, W.weight, W.eta,
                      W.ii, W.jj, W.t0, W.t1, W.lm, W.ep, np.float64)
    H, v = core.hessian()
    core.retract(np.linalg.solve(H + np.diag(W.ep + W.lm * np.diag(H)), v))
    np.testing.assert_allclose(got["poses"], core.poses, rtol=0, atol=1e-9)
    np.testing.assert_allclose(got["disps"], core.disps, rtol=1e-8, atol=1e-9)
