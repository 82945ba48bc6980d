"""one cold solve per process (python scratch/solve_cold.py P w): prints the error profile per pose when it is wrong"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dba-fusion_amd"))
import numpy as np, torch
import test_gpu_solve as T
P, w = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(7 * P + w)
H, b, fpose = T._pose_system(rng, P, w)
ref = T._ref(H, b)
S = T._SkylineSolver(P)
dx, failed = S.solve(H, b, fpose)
e = np.abs(dx - ref)
tol = 3e-7 * max(1.0, np.abs(ref).max())
if failed or e.max() > tol:
    print("BAD failed=%d max err %.3e; per pose max err:" % (failed, e.max()), " ".join("%.1e" % e[6 * p:6 * p + 6].max() for p in range(P)))
    dx2, _ = S.solve(H, b, fpose)
    print("   second solve max err %.3e" % np.abs(dx2 - ref).max())
else:
    print("ok")
