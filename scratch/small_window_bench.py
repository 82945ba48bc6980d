import sys, time, numpy as np, torch
sys.path.insert(0,'dba-fusion_amd'); sys.path.insert(0,'tests')
from dbaf_amd import synthetic as syn
from util import to_dev
import droid_backends
for kf, band in [(8,2),(12,2),(16,3),(25,2)]:
    ii,jj = syn.graph_banded(kf, band)
    W = syn.make_window(ii, jj, kf, 64, 64, seed=1)
    d = to_dev(W); p0, z0 = d["poses"].clone(), d["disps"].clone()
    def run():
        d["poses"].copy_(p0); d["disps"].copy_(z0)
        droid_backends.ba(d["poses"], d["disps"], d["intrinsics"], d["disps_sens"], d["target"], d["weight"], d["eta"], d["ii"], d["jj"], W.t0, W.t1, 2, W.lm, W.ep, False)
    for _ in range(5): run()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(50): run()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/50
    print("%2d KF %3d edges: ba(itrs=2) %.1f us" % (kf, len(ii), dt*1e6))
