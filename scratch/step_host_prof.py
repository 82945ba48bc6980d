"""where the HOST's time of one eager update goes on a small window (9 KF / 36 edges at 55x55: the step is host-bound there)"""
import cProfile, os, pstats, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dba-fusion_amd"))
from dbaf_amd import synthetic as syn  # noqa: E402
from dbaf_amd.corr import CorrBlock  # noqa: E402
import droid_backends  # noqa: E402

dev = torch.device("cuda", 0)
h = w = 55
W = syn.make_window(*syn.graph_banded(9, 2, extra=[(0, 3), (1, 4), (2, 5)]), 9, h, w, seed=1, intr=(20.5, 20.5, 27.4, 27.6))
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
poses0, disps0, intr, dsens, eta = t(W.poses), t(W.disps), t(W.intrinsics), t(W.disps_sens), t(W.eta)
ii, jj, tg, wt = t(W.ii), t(W.jj), t(W.target), t(W.weight)
K = intr[None].expand(W.B, 4).contiguous()
fm = t(syn.make_fmaps(W.B, 128, h, w, 3))
corr = CorrBlock(fm[ii][None], fm[jj][None]).build()
n_in = W.N // 3
tgt5, wgt5 = tg.permute(0, 2, 3, 1)[None].contiguous(), wt.permute(0, 2, 3, 1)[None].contiguous()
tgt_inac, tgt_act, wgt_inac, wgt_act = tgt5[:, :n_in].clone(), tgt5[:, n_in:].clone(), wgt5[:, :n_in].clone(), wgt5[:, n_in:].clone()
ii_inac, jj_inac, ii_act, jj_act = ii[:n_in].clone(), jj[:n_in].clone(), ii[n_in:].clone(), jj[n_in:].clone()
m = torch.arange(n_in, device=dev)
poses, disps = poses0.clone(), disps0.clone()
parts = {}


def lookup():
    return corr.lookup_reprojected(poses, disps, K, ii, jj)


def caller():
    ii_n = torch.cat([ii_inac[m], ii_act], 0)
    jj_n = torch.cat([jj_inac[m], jj_act], 0)
    a = torch.cat([tgt_inac[:, m], tgt_act], 1).view(-1, h, w, 2).permute(0, 3, 1, 2).contiguous()
    b = torch.cat([wgt_inac[:, m], wgt_act], 1).view(-1, h, w, 2).permute(0, 3, 1, 2).contiguous()
    return ii_n, jj_n, a, b


def ba(g):
    droid_backends.ba_clamped(poses, disps, intr, dsens, g[2], g[3], eta, g[0], g[1], W.t0, W.t1, 2, W.lm, W.ep, False, 0.001)


def step():
    poses.copy_(poses0)
    disps.copy_(disps0)
    lookup()
    ba(caller())


for _ in range(30):
    step()
torch.cuda.synchronize()
N = 400
t0 = time.perf_counter()
for _ in range(N):
    step()
th = time.perf_counter() - t0
torch.cuda.synchronize()
print("eager step: host %.1f us, wall %.1f us" % (th / N * 1e6, (time.perf_counter() - t0) / N * 1e6))
# each piece alone with an idle queue in front of it (sleep between calls): its true host cost
for name, fn in (("reset (2 copies)", lambda: (poses.copy_(poses0), disps.copy_(disps0))), ("lookup_reprojected", lookup), ("caller's ten statements", caller)):
    tot = 0.0
    for _ in range(200):
        torch.cuda.synchronize()
        a = time.perf_counter()
        fn()
        tot += time.perf_counter() - a
    print("%-26s %.1f us host per call (idle queue)" % (name, tot / 200 * 1e6))
g = caller()
tot = 0.0
for _ in range(200):
    torch.cuda.synchronize()
    a = time.perf_counter()
    ba(g)
    tot += time.perf_counter() - a
print("%-26s %.1f us host per call (idle queue)" % ("ba_clamped (python policy)", tot / 200 * 1e6))
tot = 0.0
for _ in range(200):
    torch.cuda.synchronize()
    a = time.perf_counter()
    droid_backends.compiled.ba_clamped(poses, disps, intr, dsens, g[2], g[3], eta, g[0], g[1], W.t0, W.t1, 2, W.lm, W.ep, False, 0.001)
    tot += time.perf_counter() - a
print("%-26s %.1f us host per call (idle queue)" % ("ba_clamped (compiled)", tot / 200 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    torch.cuda.synchronize()
    lookup()
    torch.cuda.synchronize()
    ba(g)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
