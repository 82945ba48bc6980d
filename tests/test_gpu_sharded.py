"""GPU: the edge-sharded BA driver with the real stage executor (HipStages over the C ABI, incl. the
frame_owned masks in the kernels).  The box has one GPU, so the two ranks run as two Python threads of one
process and exchange through an in-process stand-in for torch.distributed (RCCL itself is exercised by the
driver's multi-GPU bench; the exchange logic is also covered on CPU with gloo in test_sharded_cpu.py)."""
import threading

import numpy as np
import pytest
import torch

from dbaf_amd import synthetic as syn
from dbaf_amd.sharded import ShardedWindow
from util import to_dev, check_state

pytestmark = pytest.mark.gpu


class LoopbackDist:
    """all_reduce(sum) between `world` threads of this process"""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.local = threading.local()

    def bind(self, rank):
        self.local.rank = rank

    def all_reduce(self, t):
        torch.cuda.synchronize()
        self.slots[self.local.rank] = t.clone()
        self.barrier.wait()
        total = self.slots[0].clone()
        for r in range(1, self.world):
            total += self.slots[r]
        torch.cuda.synchronize()
        self.barrier.wait()
        t.copy_(total)
        torch.cuda.synchronize()
        self.barrier.wait()


@pytest.mark.parametrize("mk,world", [(lambda: syn.window_tiny_b(3), 2), (lambda: syn.window_25_96(1), 2),
                                     (lambda: syn.window_25_96(2), 4)])
def test_sharded_hip_ba_matches_single_gpu_ba(mk, world):
    import droid_backends
    W = mk()
    d = to_dev(W)
    droid_backends.ba(d["poses"], d["disps"], d["intrinsics"], d["disps_sens"], d["target"], d["weight"], d["eta"],
                      d["ii"], d["jj"], W.t0, W.t1, 2, W.lm, W.ep, False)
    torch.cuda.synchronize()
    ref_poses, ref_disps = d["poses"].cpu().numpy(), d["disps"].cpu().numpy()

    dist = LoopbackDist(world)
    results, errors = [None] * world, []

    def run(rank):
        try:
            dist.bind(rank)
            torch.cuda.set_device(0)
            sh = ShardedWindow(W.ii, W.jj, W.t0, W.t1, W.B, world, rank)
            sel = sh.local_edges
            dd = to_dev(W)
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
            sh.ba(dd["poses"], dd["disps"], dd["intrinsics"], dd["disps_sens"], t(W.target[sel]), t(W.weight[sel]),
                  dd["eta"], t(W.ii[sel]), t(W.jj[sel]), 2, W.lm, W.ep, dist)
            torch.cuda.synchronize()
            results[rank] = (dd["poses"].cpu().numpy(), dd["disps"].cpu().numpy(), len(sel))
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            dist.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    assert sum(r[2] for r in results) == W.N and all(r[2] < W.N for r in results)
    for r in range(1, world):  # replicas coherent
        assert np.array_equal(results[r][0], results[0][0]) and np.array_equal(results[r][1], results[0][1])
    # same state as the single-GPU call (summation order of the f64 atomics differs, nothing else)
    check_state(results[0][0], results[0][1], ref_poses, ref_disps, W.disps, t_tol=2e-6, r_tol=2e-7, d_rtol=2e-5)
