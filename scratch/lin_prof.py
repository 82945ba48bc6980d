import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dba-fusion_amd"))
from dbaf_amd import synthetic as syn, _lib
import droid_backends
lib = _lib.load()
W = getattr(syn, "window_" + (sys.argv[1] if len(sys.argv) > 1 else "25_96"))(0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
poses, disps = t(W.poses), t(W.disps)
args = (t(W.intrinsics), t(W.disps_sens), t(W.target), t(W.weight), t(W.eta), t(W.ii), t(W.jj))
dump = ctypes.CDLL(_lib.LIB_PATH).dba_lin_prof_dump
for it in range(6):
    p, d = poses.clone(), disps.clone()
    droid_backends.ba(p, d, *args, W.t0, W.t1, 1, W.lm, W.ep, False)
    dump()   # one linearisation per dump
