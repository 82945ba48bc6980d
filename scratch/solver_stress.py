"""Random stress of the reduced-system solvers through dba_ba_solve: pose counts 16..64, random bands, extra random couplings
(block level), sparse right-hand sides, occasional failing pivots; compares with numpy and reports which paths ran.
   python scratch/solver_stress.py [count] [seed]"""
import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dba-fusion_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from test_gpu_solve import _solve_on_device

count = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
stats = dict(split=0, whole=0, failed_ok=0, bad=0)
worst = 0.0
for it in range(count):
    P = int(rng.integers(16, 65))
    n = 6 * P
    bandp = int(rng.integers(1, 8))               # band in poses
    A = np.zeros((n, n))
    for p in range(P):
        for q in range(max(0, p - bandp), p + 1):
            blk = rng.uniform(-1, 1, (6, 6)) / (1 + 3 * (p - q))
            A[6 * p:6 * p + 6, 6 * q:6 * q + 6] = blk
    for _ in range(int(rng.integers(0, 3))):     # extra couplings anywhere
        p, q = sorted(rng.integers(0, P, 2))
        A[6 * q:6 * q + 6, 6 * p:6 * p + 6] += rng.uniform(-.3, .3, (6, 6))
    H = np.tril(A) + np.tril(A, -1).T
    H[np.diag_indices(n)] = 8.0 + rng.uniform(0, 2, n) + np.abs(H).sum(1) * 0.5
    b = np.sin(1.3 * np.arange(n))
    if rng.random() < 0.2:
        b[:] = 0
        lo = int(rng.integers(0, n - 4)); b[lo:lo + 4] = 1 + np.arange(4)
    bad_pivot = rng.random() < 0.1
    if bad_pivot:
        w = int(rng.integers(0, n)); H[w, w] = -abs(H[w, w])
    lm, ep = 1e-4, 0.1
    Hd = H.copy(); Hd[np.diag_indices(n)] += ep + lm * np.diag(H)
    dx, failed = _solve_on_device(H, b, P, lm, ep)
    taken = _solve_on_device.split[0] if n > 174 else 0
    spd = np.all(np.linalg.eigvalsh(Hd) > 0)
    if not spd:
        ok = failed == 1 and np.all(dx == 0)
        stats["failed_ok" if ok else "bad"] += 1
        if not ok: print("MISMATCH (non-SPD)", it, P, bandp, failed)
        continue
    ref = np.linalg.solve(Hd, b)
    err = np.abs(dx - ref).max() / max(1.0, np.abs(ref).max())
    worst = max(worst, err)
    if failed or err > 3e-7:
        stats["bad"] += 1
        print("MISMATCH", it, P, bandp, failed, err)
    stats["split" if taken else "whole"] += 1
print(stats, "worst relative error %.2e" % worst)
