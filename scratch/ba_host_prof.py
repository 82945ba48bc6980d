"""host-side cost of droid_backends.ba_clamped (Python policy) against the compiled adapter's, and where the Python time goes"""
import cProfile, os, pstats, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dba-fusion_amd"))
from dbaf_amd import synthetic as syn  # noqa: E402
import droid_backends  # noqa: E402

dev = torch.device("cuda", 0)
W = syn.make_window(*syn.graph_banded(9, 2, extra=[(0, 3), (1, 4), (2, 5)]), 9, 55, 55, seed=1, intr=(20.5, 20.5, 27.4, 27.6))
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
poses, disps, intr, dsens, eta = t(W.poses), t(W.disps), t(W.intrinsics), t(W.disps_sens), t(W.eta)
ii, jj, tg, wt = t(W.ii), t(W.jj), t(W.target), t(W.weight)


def run(fn, fresh, reps=2000):
    for _ in range(20):
        fn(poses, disps, intr, dsens, tg, wt, eta, ii, jj, W.t0, W.t1, 2, W.lm, W.ep, False, 0.001)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        a, b = (ii.clone(), jj.clone()) if fresh else (ii, jj)
        fn(poses, disps, intr, dsens, tg, wt, eta, a, b, W.t0, W.t1, 2, W.lm, W.ep, False, 0.001)
    host = (time.perf_counter() - t0) / reps * 1e6
    torch.cuda.synchronize()
    return host, (time.perf_counter() - t0) / reps * 1e6


for name, fn in (("python policy", droid_backends.ba_clamped), ("compiled", droid_backends.compiled.ba_clamped)):
    for fresh in (False, True):
        h, w_ = run(fn, fresh)
        print("%-14s %-22s host %.1f us per call (enqueue only), %.1f us incl. the device's tail" % (name, "new ii / jj objects" if fresh else "same tensor objects", h, w_))
pr = cProfile.Profile()
pr.enable()
run(droid_backends.ba_clamped, False, 1000)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
