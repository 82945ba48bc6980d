"""Cold-start stress of the solvers whose waves / workgroups meet through flags instead of barriers (round 6).

`tests/native/solve_cold` (built by `make testbin`) is ONE process = ONE first solve: device initialisation, one system, the
solve as the first kernel of the process, a float64 host Cholesky as the judge.  A protocol race shows when the waves start out
of step (cold instruction cache, clocks down, waves dealt out one after the other) and hides in warm back-to-back solves: round
5's hand-over race on flagE[NT - 1] gave one wrong solve in ~40 cold starts and none in 300 warm ones.  Here: 240 fresh
processes over the window kernel's three window heights (48 / 64 / 80 rows: three, four, five factor waves), with and without the
ring of panels (n = 378), the literal skyline of BASELINE's 64-KF / 512-edge graph, and the skyline kernel's two-workgroup
handshake through global memory (DBA_SOLVE_KERNEL=band); eight processes share the GPU at a time, which perturbs the start-up
further.  What the unfixed kernel does under this test: profiles/r06_cold_start_stress.txt (scratch/build_unfixed_lib.sh).
Replaces the host Eigen solve of /root/reference/src/droid_kernels.cu:1248-1269."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "native", "solve_cold")
LIB = os.path.join(ROOT, "dba-fusion_amd", "lib", "libdba_hip.so")

# (poses, band in poses | -2: the 64/512 skyline, forced kernel)
CONFIGS = [(24, 4, None), (29, 4, None), (40, 4, None), (63, 4, None),       # n = 144 / 174 / 240 / 378, 48-row window
           (24, 6, None), (63, 6, None),                                     # 64-row window, whole store / ring
           (30, 8, None), (63, -2, None),                                    # 80-row window, whole store / ring: BASELINE configs[3]
           (40, 4, "band"), (63, 4, "band")]                                 # two workgroups, handshake through global memory
SEEDS = 24


def run_cold(lib, configs, seeds, parallel=8, extra=0):
    """every (config, seed) in a process of its own, `parallel` at a time; returns (lines of the bad ones, total)"""
    jobs = [(P, w, forced, seed) for seed in range(seeds) for (P, w, forced) in configs]
    bad, running, done = [], [], 0

    def reap(block):
        nonlocal done
        for item in list(running):
            proc, job = item
            if block:
                proc.wait()
            if proc.poll() is None:
                continue
            out = proc.stdout.read().decode(errors="replace").strip()
            if proc.returncode != 0:
                bad.append("%r rc=%d: %s" % (job, proc.returncode, out))
            running.remove(item)
            done += 1
            if block:
                return

    for job in jobs:
        while len(running) >= parallel:
            reap(True)
            reap(False)
        P, w, forced, seed = job
        env = dict(os.environ)
        env.pop("DBA_SOLVE_KERNEL", None)
        if forced:
            env["DBA_SOLVE_KERNEL"] = forced
        running.append((subprocess.Popen([BIN, lib, str(P), str(w), str(seed), str(extra)], stdout=subprocess.PIPE,
                                         stderr=subprocess.STDOUT, env=env), job))
    while running:
        reap(True)
    return bad, done


def test_first_solve_of_a_process_is_right_every_time():
    if not os.path.exists(BIN):
        pytest.fail("tests/native/solve_cold is not built (make testbin)")
    bad, total = run_cold(LIB, CONFIGS, SEEDS)
    assert total == len(CONFIGS) * SEEDS >= 200
    assert not bad, "%d of %d cold starts wrong:\n%s" % (len(bad), total, "\n".join(bad[:20]))


def test_a_cold_start_followed_by_other_systems():
    """the same processes go on to solve three other systems of the same structure: nothing a solve leaves behind (LDS, the
    scratch the ring's early panels are parked in, the handshake words) may reach the next one"""
    bad, total = run_cold(LIB, CONFIGS, 3, extra=3)
    assert not bad, "%d of %d wrong:\n%s" % (len(bad), total, "\n".join(bad[:20]))


if __name__ == "__main__":   # python tests/test_gpu_solve_cold.py <library> [seeds]: the stress as an experiment on any build
    import sys
    import time
    lib = sys.argv[1] if len(sys.argv) > 1 else LIB
    seeds = int(sys.argv[2]) if len(sys.argv) > 2 else SEEDS
    t = time.time()
    bad, total = run_cold(lib, CONFIGS, seeds)
    print("%s: %d of %d cold starts wrong (%.1f s)" % (lib, len(bad), total, time.time() - t))
    for line in bad[:40]:
        print("  ", line)
