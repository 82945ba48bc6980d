"""stress for read-before-write races in the window solver: systems of the same size but different values are solved
alternately, so that LDS left behind by the previous launch never holds the right values (python scratch/solve_stress.py [reps])"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dba-fusion_amd"))
import numpy as np, torch
import test_gpu_solve as T
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
for P, w in [(40, 4), (24, 4), (63, 4), (29, 4), (33, 4), (16, 3), (24, 6), (45, 5), (8, 4), (3, 2)]:
    systems = []
    for k in range(3):
        rng = np.random.default_rng(1000 * k + 7 * P + w)
        H, b, fpose = T._pose_system(rng, P, w)
        H = H * (1.0 + 0.5 * k)          # other magnitudes too
        b = b * (1.0 - 0.3 * k)
        systems.append((H, b, fpose, T._ref(H, b)))
    S = T._SkylineSolver(P)
    nb = 0
    for r in range(reps):
        for k, (H, b, fpose, ref) in enumerate(systems):
            dx, failed = S.solve(H, b, fpose)
            tol = 3e-7 * max(1.0, np.abs(ref).max())
            e = np.abs(dx - ref).max()
            if failed or e > tol:
                nb += 1
                if nb <= 5:
                    idx = np.nonzero(np.abs(dx - ref) > tol)[0]
                    print("  P=%d w=%d rep %d system %d: failed=%d max err %.3e (tol %.1e) bad unknowns %d..%d (%d)" % (P, w, r, k, failed, e, tol, idx.min() if len(idx) else -1, idx.max() if len(idx) else -1, len(idx)))
    print("P=%d w=%d: %d bad of %d" % (P, w, nb, 3 * reps))
    bad += nb
print("TOTAL bad", bad)
