"""BACore update unit (bench.py's bacore_unit) with many repetitions, for A/B runs of two libraries on one box (DBA_HIP_LIB)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dba-fusion_amd"))
from dbaf_amd import synthetic as syn
import droid_backends
W = getattr(syn, "window_" + (sys.argv[1] if len(sys.argv) > 1 else "25_96"))(0)
nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 200
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
st0 = torch.cat([t(W.poses).reshape(-1), t(W.disps).reshape(-1)]); stt = st0.clone()
poses, disps = stt[:W.poses.size].view(W.B, 7), stt[W.poses.size:].view(W.B, W.h, W.w)
intr, dsens, target, weight, eta, ii, jj = t(W.intrinsics), t(W.disps_sens), t(W.target), t(W.weight), t(W.eta), t(W.ii), t(W.jj)
P6 = 6 * (W.t1 - W.t0)
H = torch.zeros([P6, P6], dtype=torch.float64); v = torch.zeros([P6], dtype=torch.float64)
th = ts = tr = 0.0
def unit():
    global th, ts, tr
    stt.copy_(st0)
    core = droid_backends.BACore()
    core.init(poses, disps, intr, dsens, target.clone(), weight.clone(), eta, ii.clone(), jj.clone(), W.t0, W.t1, 2, W.lm, W.ep, False)
    for _ in range(2):
        a = time.perf_counter(); core.hessian(H, v); b = time.perf_counter()
        Hn = H.numpy().copy(); Hn[np.diag_indices(P6)] += W.ep + W.lm * np.diag(Hn)
        dxn = np.linalg.solve(Hn, v.numpy()); c = time.perf_counter()
        core.retract(torch.from_numpy(dxn)); d = time.perf_counter()
        th += b - a; ts += c - b; tr += d - c
    disps.clamp_(min=0.001)
for _ in range(10): unit()
torch.cuda.synchronize(); th = ts = tr = 0.0
t0 = time.perf_counter()
for _ in range(nrep): unit()
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / nrep * 1e6
print("%s %s: update %.1f us, hessian x2 %.1f, host solve x2 %.1f, retract enqueue x2 %.1f, device side %.1f" % (
    os.path.basename(os.environ.get("DBA_HIP_LIB", "libdba_hip.so")), sys.argv[1] if len(sys.argv) > 1 else "25_96", tot, th / nrep * 1e6, ts / nrep * 1e6, tr / nrep * 1e6, tot - ts / nrep * 1e6))
