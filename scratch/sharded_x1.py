"""sharded driver on one rank against plain ba, wall clock per call (python scratch/sharded_x1.py)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dba-fusion_amd"))
import numpy as np, torch
from dbaf_amd import synthetic as syn
from dbaf_amd.sharded import ShardedWindow
import droid_backends
W = syn.window_25_96(0)
dev = "cuda"
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
poses0, disps0 = t(W.poses), t(W.disps)
intr, dsens, eta = t(W.intrinsics), t(W.disps_sens), t(W.eta)
ii, jj, target, weight = t(W.ii), t(W.jj), t(W.target), t(W.weight)
poses, disps = poses0.clone(), disps0.clone()
sh1 = ShardedWindow(W.ii, W.jj, W.t0, W.t1, W.B, 1, 0)
def loop(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t_ = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t_) / reps * 1e6
def plain():
    poses.copy_(poses0); disps.copy_(disps0)
    droid_backends.ba(poses, disps, intr, dsens, target, weight, eta, ii, jj, W.t0, W.t1, 2, W.lm, W.ep, False)
def sharded1():
    poses.copy_(poses0); disps.copy_(disps0)
    sh1.ba(poses, disps, intr, dsens, target, weight, eta, ii, jj, 2, W.lm, W.ep, None)
for k in range(4):
    print("plain %.1f us  sharded x1 %.1f us" % (loop(plain, 20), loop(sharded1, 20)))
# host time of the calls themselves (enqueue only) and the slowest single call
import statistics
for name, fn in (("plain", plain), ("sharded x1", sharded1)):
    ts = []
    for _ in range(30):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append(((t1 - t0) * 1e6, (t2 - t0) * 1e6))
    print("%-10s enqueue us: median %.0f max %.0f | to completion: median %.0f max %.0f" % (name, statistics.median(a for a, _ in ts), max(a for a, _ in ts), statistics.median(b for _, b in ts), max(b for _, b in ts)))
