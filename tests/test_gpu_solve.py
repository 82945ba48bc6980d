"""The damped reduced-system solve (csrc/ba_solve_tile.hip, csrc/ba_solve.hip) through the C ABI stage function
dba_ba_solve, against a float64 Cholesky on the host: dense, block-banded, block-diagonal and non-SPD systems at
window sizes on both sides of the register-tile kernel's limit (replaces droid_kernels.cu:200-218, :1248-1269)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from dbaf_amd import _lib

pytestmark = pytest.mark.gpu


def _system(rng, P, band, spd=True):
    n = 6 * P
    A = np.zeros((n, n))
    for i in range(n):
        for j in range(max(0, i - band + 1), i + 1):
            A[i, j] = A[j, i] = rng.uniform(-1, 1) / (1 + i - j)
    A[np.diag_indices(n)] = 6.0 + rng.uniform(0, 2, n)
    if not spd:
        A[n // 2, n // 2] = -1.0
    return A, np.sin(1.3 * np.arange(n))


def _solve_on_device(H, b, P, lm, ep):
    lib = _lib.load()
    N, B, ht, wd, t0, t1 = 1, P + 2, 8, 8, 1, 1 + P
    dims = (N, B, ht, wd, t0, t1)
    nbytes = lib.dba_ba_workspace_bytes(*dims)
    ws = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    lay = _lib.BaLayout()
    _lib.check(lib.dba_ba_get_layout(*dims, ctypes.byref(lay)), "dba_ba_get_layout")
    n = 6 * P
    ws[lay.H:lay.H + 8 * n * n].view(torch.float64).copy_(torch.from_numpy(H.reshape(-1)).cuda())
    ws[lay.b:lay.b + 8 * n].view(torch.float64).copy_(torch.from_numpy(b).cuda())
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.dba_ba_solve(*dims, lm, ep, ctypes.c_void_p(ws.data_ptr()), nbytes, stream), "dba_ba_solve")
    torch.cuda.synchronize()
    dx = ws[lay.dx:lay.dx + 4 * n].view(torch.float32).cpu().numpy().copy()
    meta = ws[lay.meta:lay.meta + 32].view(torch.int32).cpu().numpy()
    failed = int(meta[1])
    _solve_on_device.split = (int(meta[4]), 4 * int(meta[5]), 4 * int(meta[6]))   # (taken?, top unknowns, bottom unknowns)
    return dx, failed


@pytest.mark.parametrize("P,band", [(1, 6), (2, 12), (3, 7), (23, 30), (24, 24), (24, 144), (24, 1), (24, 3), (28, 40),
                                     (29, 174), (30, 36), (33, 50), (40, 60), (63, 36), (63, 18), (64, 40), (63, 378), (50, 300), (45, 1), (64, 5)])
def test_solve_matches_host_cholesky(P, band):
    rng = np.random.default_rng(100 * P + band)
    H, b = _system(rng, P, band)
    lm, ep = 1e-4, 0.1
    Hd = H.copy()
    Hd[np.diag_indices_from(Hd)] += ep + lm * np.diag(H)  # droid_kernels.cu:1252-1253
    ref = np.linalg.solve(Hd, b)
    dx, failed = _solve_on_device(H, b, P, lm, ep)
    assert failed == 0
    np.testing.assert_allclose(dx, ref, rtol=0, atol=3e-7 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("P", [5, 24, 29, 33, 63])
def test_non_spd_system_gives_a_zero_update(P):
    rng = np.random.default_rng(P)
    H, b = _system(rng, P, 18, spd=False)
    dx, failed = _solve_on_device(H, b, P, 1e-4, 0.1)
    assert failed == 1 and np.all(dx == 0.0)


def _check(H, b, P, lm=1e-4, ep=0.1):
    Hd = H.copy()
    Hd[np.diag_indices_from(Hd)] += ep + lm * np.diag(H)
    ref = np.linalg.solve(Hd, b)
    dx, failed = _solve_on_device(H, b, P, lm, ep)
    assert failed == 0
    np.testing.assert_allclose(dx, ref, rtol=0, atol=3e-7 * max(1.0, np.abs(ref).max()))


# The register-tile kernel eliminates banded systems with n % 4 == 0 from both ends (two fronts, csrc/ba_solve_tile.hip).
# Sizes with an even and an odd number of tile columns, bands around the tile width, and skylines that are not
# monotone (a long coupling inside the top front, the bottom front, the middle, and an arrow that leaves one front).
@pytest.mark.parametrize("P,band", [(4, 2), (4, 6), (6, 4), (8, 9), (10, 8), (12, 7), (16, 16), (20, 12), (22, 30), (24, 30), (24, 12), (24, 60), (28, 30)])
def test_two_front_elimination_on_banded_systems(P, band):
    rng = np.random.default_rng(7 * P + band)
    H, b = _system(rng, P, band)
    _check(H, b, P)


@pytest.mark.parametrize("P,band,i,j", [(24, 24, 141, 2), (24, 24, 90, 40), (24, 12, 139, 100), (24, 12, 40, 4), (16, 6, 30, 1),
                                         (24, 18, 143, 120), (24, 18, 72, 60), (20, 10, 119, 0)])
def test_two_front_elimination_with_a_long_coupling(P, band, i, j):
    rng = np.random.default_rng(P + band + i)
    H, b = _system(rng, P, band)
    H[i, j] = H[j, i] = 0.37
    _check(H, b, P)


def test_two_front_elimination_with_sparse_right_hand_sides():
    # the bottom front fills the right-hand-side row to the left of its first non-zero
    rng = np.random.default_rng(5)
    H, _ = _system(rng, 24, 24)
    for lo, hi in ((140, 144), (0, 4), (70, 74), (100, 144)):
        b = np.zeros(144)
        b[lo:hi] = 1.0 + np.arange(hi - lo)
        _check(H, b, 24)


@pytest.mark.parametrize("P,where", [(24, 3), (24, 140), (24, 70), (20, 118), (16, 1)])
def test_a_failing_pivot_in_either_front_gives_a_zero_update(P, where):
    rng = np.random.default_rng(P + where)
    H, b = _system(rng, P, 18)
    H[where, where] = -1.0
    dx, failed = _solve_on_device(H, b, P, 1e-4, 0.1)
    assert failed == 1 and np.all(dx == 0.0)


# The skyline kernel (30 - 64 poses) cuts banded systems at a separator and runs the two halves on two workgroups
# (csrc/ba_solve_band.hip): odd and even pose counts (n % 4 == 2 and 0), bands from one pose to the separator limit
# (64 unknowns; wider bands and arrows keep one workgroup), sparse right-hand sides, long couplings on either side of
# and across the separator, failing pivots in the top block, the bottom block and the separator.
@pytest.mark.parametrize("P,band", [(30, 6), (30, 36), (31, 36), (31, 13), (32, 48), (33, 61), (35, 70), (40, 36), (41, 30),
                                     (47, 24), (50, 36), (63, 36), (63, 60), (64, 36), (64, 12), (64, 66)])
def test_two_workgroup_skyline_solve(P, band):
    rng = np.random.default_rng(11 * P + band)
    H, b = _system(rng, P, band)
    _check(H, b, P)
    taken, top, bottom = _solve_on_device.split
    n = 6 * P
    if os.environ.get("DBA_SOLVE_SPLIT") == "0":   # (the switch that keeps the kernel to one workgroup)
        assert taken == 0
    elif band <= 60:   # a separator of at most 64 unknowns exists: both workgroups must have worked, on balanced halves
        assert taken == 1 and min(top, bottom) >= 24 and abs(top - bottom) <= 8 and band - 4 <= n - top - bottom <= 64
    elif band >= 70:
        assert taken == 0


@pytest.mark.parametrize("P,band,i,j", [(40, 24, 100, 20), (40, 24, 230, 150), (40, 24, 140, 100), (40, 24, 239, 0),
                                         (63, 36, 377, 300), (63, 36, 60, 2), (63, 36, 200, 170), (31, 18, 185, 150)])
def test_two_workgroup_skyline_solve_with_a_long_coupling(P, band, i, j):
    rng = np.random.default_rng(P + band + i)
    H, b = _system(rng, P, band)
    H[i, j] = H[j, i] = 0.37
    _check(H, b, P)


def test_two_workgroup_skyline_solve_with_sparse_right_hand_sides():
    rng = np.random.default_rng(9)
    H, _ = _system(rng, 40, 30)
    for lo, hi in ((236, 240), (0, 4), (118, 124), (60, 64), (180, 240)):
        b = np.zeros(240)
        b[lo:hi] = 1.0 + np.arange(hi - lo)
        _check(H, b, 40)


@pytest.mark.parametrize("P,where", [(40, 3), (40, 236), (40, 120), (40, 70), (40, 170), (31, 184), (63, 190), (63, 377)])
def test_two_workgroup_skyline_failing_pivot_gives_a_zero_update(P, where):
    rng = np.random.default_rng(P + where)
    H, b = _system(rng, P, 18)
    H[where, where] = -1.0
    dx, failed = _solve_on_device(H, b, P, 1e-4, 0.1)
    assert failed == 1 and np.all(dx == 0.0)


def test_two_workgroup_skyline_solve_repeated_calls_share_the_exchange_buffer():
    # the handshake flags carry a per-launch generation: back-to-back solves on one workspace must not see each other's
    rng = np.random.default_rng(3)
    for P, band in ((40, 24), (40, 30), (63, 36), (40, 24)):
        H, b = _system(rng, P, band)
        for _ in range(3):
            _check(H, b, P)


def test_random_window_structures_against_numpy():
    """80 random reduced systems at pose level (16-64 poses, bands of 1-7 poses, up to two extra couplings anywhere,
    sparse right-hand sides, a negative pivot now and then): whatever solver path takes them, the result is numpy's or,
    for a system that is not positive definite, a zero update"""
    rng = np.random.default_rng(2026)
    paths = {"two workgroups": 0, "one workgroup": 0, "not SPD": 0}
    for _ in range(80):
        P = int(rng.integers(16, 65))
        n = 6 * P
        bandp = int(rng.integers(1, 8))
        A = np.zeros((n, n))
        for p in range(P):
            for q in range(max(0, p - bandp), p + 1):
                A[6 * p:6 * p + 6, 6 * q:6 * q + 6] = rng.uniform(-1, 1, (6, 6)) / (1 + 3 * (p - q))
        for _k in range(int(rng.integers(0, 3))):
            p, q = sorted(rng.integers(0, P, 2))
            A[6 * q:6 * q + 6, 6 * p:6 * p + 6] += rng.uniform(-.3, .3, (6, 6))
        H = np.tril(A) + np.tril(A, -1).T
        H[np.diag_indices(n)] = 8.0 + rng.uniform(0, 2, n) + np.abs(H).sum(1) * 0.5
        b = np.sin(1.3 * np.arange(n))
        if rng.random() < 0.2:
            b[:] = 0
            lo = int(rng.integers(0, n - 4))
            b[lo:lo + 4] = 1 + np.arange(4)
        if rng.random() < 0.1:
            w = int(rng.integers(0, n))
            H[w, w] = -abs(H[w, w])
        lm, ep = 1e-4, 0.1
        Hd = H.copy()
        Hd[np.diag_indices(n)] += ep + lm * np.diag(H)
        dx, failed = _solve_on_device(H, b, P, lm, ep)
        if not np.all(np.linalg.eigvalsh(Hd) > 0):
            assert failed == 1 and np.all(dx == 0.0)
            paths["not SPD"] += 1
            continue
        ref = np.linalg.solve(Hd, b)
        assert failed == 0
        np.testing.assert_allclose(dx, ref, rtol=0, atol=3e-7 * max(1.0, np.abs(ref).max()))
        paths["two workgroups" if (n > 174 and _solve_on_device.split[0]) else "one workgroup"] += 1
    print(paths)
    if os.environ.get("DBA_SOLVE_SPLIT") != "0":
        assert paths["two workgroups"] >= 20 and paths["one workgroup"] >= 10


# ---- the five-wave window kernel (csrc/ba_solve_wave.hip), reached through the stage function that takes a skyline table ----

def _pose_system(rng, P, w, spd=True, extra=()):
    """P poses of 6 unknowns, pose p coupled with p-w .. p (+ extra pairs); returns H, b and the pose-level skyline"""
    n = 6 * P
    H = np.zeros((n, n))
    fpose = list(range(P))
    for p, q in [(p, q) for p in range(P) for q in range(max(0, p - w), p + 1)] + list(extra):
        Bk = rng.standard_normal((6, 6)) * 0.3
        if p == q:
            Bk = Bk + Bk.T
        H[6 * p:6 * p + 6, 6 * q:6 * q + 6] += Bk
        if p != q:
            H[6 * q:6 * q + 6, 6 * p:6 * p + 6] += Bk.T
        fpose[p] = min(fpose[p], q)
    H += np.eye(n) * (np.abs(H).sum(1).max() + 1.0)
    if not spd:
        H[n // 2, n // 2] = -1.0
    return H, np.sin(1.3 * np.arange(n)), np.array(fpose, np.int32)


class _SkylineSolver:
    """one workspace, several solves: the solver's choice of kernel for a workspace depends on what its previous system was"""

    def __init__(self, P, init=False):
        self.lib = _lib.load()
        self.P = P
        self.dims = (1, P + 2, 8, 8, 1, 1 + P)
        self.nbytes = self.lib.dba_ba_workspace_bytes(*self.dims)
        self.ws = torch.zeros(self.nbytes, dtype=torch.uint8, device="cuda")
        self.lay = _lib.BaLayout()
        _lib.check(self.lib.dba_ba_get_layout(*self.dims, ctypes.byref(self.lay)), "dba_ba_get_layout")
        if init:   # forget what the library was told about an earlier workspace at this address
            _lib.check(self.lib.dba_ba_workspace_init(*self.dims, ctypes.c_void_p(self.ws.data_ptr()), self.nbytes,
                                                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "dba_ba_workspace_init")

    def solve(self, H, b, fpose, lm=1e-4, ep=0.1):
        n, lay, ws = 6 * self.P, self.lay, self.ws
        Hl = np.tril(H) + np.triu(np.full_like(H, 1e300), 1)       # the solvers read the lower triangle only
        ws[lay.H:lay.H + 8 * n * n].view(torch.float64).copy_(torch.from_numpy(Hl.reshape(-1)).cuda())
        ws[lay.b:lay.b + 8 * n].view(torch.float64).copy_(torch.from_numpy(b).cuda())
        ws[lay.dx:lay.dx + 4 * n].view(torch.float32).fill_(7.0)
        fp = torch.from_numpy(fpose).cuda()
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(self.lib.dba_ba_solve_skyline(*self.dims, lm, ep, ctypes.c_void_p(fp.data_ptr()),
                                                 ctypes.c_void_p(ws.data_ptr()), self.nbytes, stream), "dba_ba_solve_skyline")
        torch.cuda.synchronize()
        dx = ws[lay.dx:lay.dx + 4 * n].view(torch.float32).cpu().numpy().copy()
        meta = ws[lay.meta:lay.meta + 32].view(torch.int32).cpu().numpy()
        self.fronts = (int(meta[4]), 4 * int(meta[5]), 4 * int(meta[6]))   # two fronts taken?, top / bottom unknowns (window kernel, round 6)
        return dx, int(meta[1])


def _ref(H, b, lm=1e-4, ep=0.1):
    return np.linalg.solve(H + np.diag(ep + lm * np.diag(H)), b)


@pytest.mark.parametrize("P,w", [(1, 0), (2, 1), (3, 2), (8, 4), (16, 3), (23, 4), (24, 4), (24, 3), (24, 1), (24, 0), (25, 4),
                                  (29, 4), (33, 4), (40, 4), (63, 4), (64, 3)])
def test_window_kernel_solves_banded_systems(P, w):
    rng = np.random.default_rng(7 * P + w)
    H, b, fpose = _pose_system(rng, P, w)
    dx, failed = _SkylineSolver(P).solve(H, b, fpose)
    ref = _ref(H, b)
    assert failed == 0
    np.testing.assert_allclose(dx, ref, rtol=0, atol=3e-7 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("P,w", [(24, 5), (24, 6), (24, 7), (12, 7), (40, 6), (45, 5)])
def test_window_kernel_with_four_factor_waves_takes_wider_bands(P, w):
    """bands of 5-7 poses: the 64-row window (four factor waves); up to 48 poses its whole panel store fits LDS"""
    rng = np.random.default_rng(13 * P + w)
    H, b, fpose = _pose_system(rng, P, w)
    S = _SkylineSolver(P)
    dx, failed = S.solve(H, b, fpose)
    ref = _ref(H, b)
    assert failed == 0
    np.testing.assert_allclose(dx, ref, rtol=0, atol=3e-7 * max(1.0, np.abs(ref).max()))
    first = dx
    for rep in range(50):
        dx, failed = S.solve(H, b, fpose)
        assert failed == 0 and np.array_equal(dx, first), rep


def _skyline_pairs_64_512():
    """pose-level couplings of the REDUCED system of BASELINE configs[3] (synthetic.graph_64_512: |i - j| <= 4 plus (i, i + 5) for
    i < 10; frame 0 fixed): two poses are coupled when they see the same source frame's depths -- 8 poses wide, 9-10 among the
    first 18 (what stage 0's skyline table records)"""
    from dbaf_amd import synthetic
    ii, jj = synthetic.graph_64_512()
    pairs = set()
    for m in set(ii):
        grp = sorted({m} | {j for i, j in zip(ii, jj) if i == m})
        pairs |= {(a - 1, b - 1) for a in grp for b in grp if a > b >= 1}
    return sorted(pairs)


@pytest.mark.parametrize("P,w", [(63, 5), (63, 6), (63, 7), (64, 5), (64, 6), (64, 7), (49, 5), (50, 7), (56, 6), (61, 6), (48, 7),
                                  (24, 8), (24, 10), (30, 9), (37, 10), (38, 8), (45, 9), (52, 10), (59, 8), (60, 10), (63, 8),
                                  (63, 9), (63, 10), (64, 8), (64, 10)])
def test_window_kernel_with_the_ring_of_panels_takes_wide_bands_up_to_64_poses(P, w):
    """round 6: bands of 5-7 poses take the 64-row window (four factor waves), 8-10 poses -- what a covisibility graph of radius
    4-5 gives -- the 80-row window (five).  Beyond 48 (37) poses their panel stores (2 / 2.5 KB per step) do not fit LDS: a ring
    of 56-72 steps, the early panels parked in scratch by the substitution wave and brought back by the loader wave on the
    way back (csrc/ba_solve_wave.hip); the window kernel must TAKE these systems (verdict 1), not hand them to the skyline kernel"""
    rng = np.random.default_rng(13 * P + w)
    H, b, fpose = _pose_system(rng, P, w)
    S = _SkylineSolver(P)
    dx, failed = S.solve(H, b, fpose)
    ref = _ref(H, b)
    assert failed == 0
    np.testing.assert_allclose(dx, ref, rtol=0, atol=3e-7 * max(1.0, np.abs(ref).max()))
    assert S.lib.dba_ba_solver_verdict(*S.dims, ctypes.c_void_p(S.ws.data_ptr()), S.nbytes) == 1
    # from 46 poses on the system is cut in two: top part | separator | bottom part, a workgroup per part, each chain half as long
    if 6 * P >= 276 and os.environ.get("DBA_SOLVE_FRONTS") != "0":
        taken, top, bottom = S.fronts
        n4 = 6 * P + ((6 * P) & 2)
        assert taken == 1 and min(top, bottom) >= 96 and abs(top - bottom) <= 8 and 6 * (w - 1) <= n4 - top - bottom <= 64, S.fronts
    first = dx
    for rep in range(40):
        dx, failed = S.solve(H, b, fpose)
        assert failed == 0 and np.array_equal(dx, first), rep


def test_window_kernel_takes_the_literal_64_512_skyline():
    """BASELINE configs[3]: 63 free poses, the reduced system 8 poses wide (9-10 among the first 18: the (i, i + 5) edges) --
    round 5 solved this graph with the skyline kernel (101 us); three systems of that structure alternate so that no solve finds
    its own values left behind in LDS or in the scratch"""
    pairs = _skyline_pairs_64_512()
    P = 63
    systems = []
    for k in range(3):
        rng = np.random.default_rng(600 + k)
        H, b, fpose = _pose_system(rng, P, 0, extra=pairs)
        systems.append((H * (1.0 + 0.5 * k), b * (1.0 - 0.3 * k), fpose))
    S = _SkylineSolver(P)
    firsts = [None] * 3
    for rep in range(20):
        for k, (H, b, fpose) in enumerate(systems):
            dx, failed = S.solve(H, b, fpose)
            assert failed == 0
            if firsts[k] is None:
                ref = _ref(H, b)
                np.testing.assert_allclose(dx, ref, rtol=0, atol=3e-7 * max(1.0, np.abs(ref).max()))
                firsts[k] = dx
            assert np.array_equal(dx, firsts[k]), (rep, k)
    assert S.lib.dba_ba_solver_verdict(*S.dims, ctypes.c_void_p(S.ws.data_ptr()), S.nbytes) == 1


@pytest.mark.parametrize("P,w,where", [(63, 5, 3), (63, 5, 190), (63, 5, 377), (64, 7, 100), (49, 6, 293), (63, 8, 5), (63, 10, 200),
                                        (64, 9, 383), (30, 8, 90)])
def test_window_kernel_with_the_ring_gives_a_zero_update_for_an_indefinite_system(P, w, where):
    rng = np.random.default_rng(P + where)
    H, b, fpose = _pose_system(rng, P, w)
    H[where, where] = -1.0
    dx, failed = _SkylineSolver(P).solve(H, b, fpose)
    assert failed == 1 and np.all(dx == 0.0)


@pytest.mark.parametrize("P,w,extra", [(24, 12, ()), (24, 23, ()), (24, 2, ((20, 3),)), (50, 12, ()), (63, 13, ()),
                                        (63, 2, ((60, 1),))])
def test_systems_that_are_not_banded_are_solved_in_the_same_launch_and_then_by_the_other_kernels(P, w, extra):
    """the window kernel's admission test refuses these: the general kernel's code solves them inside its launch (first solve
    on a workspace), the pinned verdict sends the workspace's next solves to the register-tile / skyline kernels; a banded
    system on that workspace is then still solved (by whichever kernel gets it), and every 1024th solve of the workspace probes the window kernel again"""
    rng = np.random.default_rng(11 * P + w)
    H, b, fpose = _pose_system(rng, P, w, extra=extra)
    S = _SkylineSolver(P)
    ref = _ref(H, b)
    for rep in range(3):
        dx, failed = S.solve(H, b, fpose)
        assert failed == 0, rep
        np.testing.assert_allclose(dx, ref, rtol=0, atol=3e-7 * max(1.0, np.abs(ref).max()), err_msg="solve %d" % rep)
    Hb, bb, fb = _pose_system(rng, P, 3)
    refb = _ref(Hb, bb)
    for rep in range(34):            # (past the probe)
        dx, failed = S.solve(Hb, bb, fb)
        assert failed == 0
        np.testing.assert_allclose(dx, refb, rtol=0, atol=3e-7 * max(1.0, np.abs(refb).max()), err_msg="banded solve %d" % rep)


@pytest.mark.parametrize("P,w", [(5, 2), (24, 4), (29, 4), (63, 4), (24, 7)])
def test_window_kernel_and_its_fall_back_give_a_zero_update_for_an_indefinite_system(P, w):
    rng = np.random.default_rng(P)
    H, b, fpose = _pose_system(rng, P, w, spd=False)
    dx, failed = _SkylineSolver(P).solve(H, b, fpose)
    assert failed == 1 and np.all(dx == 0.0)


def test_window_kernel_many_back_to_back_solves_are_identical():
    """the waves of the kernel meet through flags in LDS, not barriers: 300 solves of one system, each bit-identical to the
    first (a lost or early flag would show as a different or failed solve)"""
    rng = np.random.default_rng(3)
    H, b, fpose = _pose_system(rng, 24, 4)
    S = _SkylineSolver(24)
    first, failed = S.solve(H, b, fpose)
    assert failed == 0
    for rep in range(300):
        dx, failed = S.solve(H, b, fpose)
        assert failed == 0 and np.array_equal(dx, first), rep


def test_opt_in_residual_check_turns_a_wrong_solve_into_a_zero_update():
    """dba_ba_set_solve_check / DBA_SOLVE_CHECK=1: a kernel behind the solver checks the residual of the damped system at
    the solution; a right solve passes untouched, a solution that is wrong beyond rounding becomes dx = 0 with the failure flag set"""
    lib = _lib.load()
    rng = np.random.default_rng(77)
    P = 24
    H, b, fpose = _pose_system(rng, P, 4)
    S = _SkylineSolver(P)
    lib.dba_ba_set_solve_check(1)
    try:
        dx, failed = S.solve(H, b, fpose)
    finally:
        lib.dba_ba_set_solve_check(0)
    ref = _ref(H, b)
    assert failed == 0
    np.testing.assert_allclose(dx, ref, rtol=0, atol=3e-7 * max(1.0, np.abs(ref).max()))
    # the same workspace, one unknown of the solution spoilt by 1e-3 of its size: the check alone
    lay, ws, n = S.lay, S.ws, 6 * P
    bad = dx.copy()
    bad[70] *= 1.001
    ws[lay.dx:lay.dx + 4 * n].view(torch.float32).copy_(torch.from_numpy(bad).cuda())
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.dba_ba_solve_check(*S.dims, 1e-4, 0.1, ctypes.c_void_p(ws.data_ptr()), S.nbytes, stream), "dba_ba_solve_check")
    torch.cuda.synchronize()
    assert int(ws[lay.meta:lay.meta + 32].view(torch.int32).cpu().numpy()[1]) == 1
    assert not ws[lay.dx:lay.dx + 4 * n].view(torch.float32).any()


@pytest.mark.parametrize("P,w", [(46, 4), (50, 3), (63, 4), (64, 4), (63, 8), (55, 6)])
def test_two_fronts_on_two_workgroups_equal_one_front(P, w):
    """round 6: systems of 46+ poses are cut into top | separator | bottom; workgroup 0 eliminates the top part, workgroup 1 the
    bottom part in reverse order (the same code on the mirrored matrix), both add up what they leave of the separator block, solve
    it redundantly and back-substitute their part.  Against the host Cholesky, through the workspace's own skyline path too (the
    plan is then kept in the workspace between the solves of one graph), alternating systems, and a failing pivot in either part
    or in the separator gives a zero update of ALL unknowns"""
    rng = np.random.default_rng(5 * P + w)
    S = _SkylineSolver(P)
    systems = [_pose_system(rng, P, w) for _ in range(2)]
    for rep in range(6):
        H, b, fpose = systems[rep & 1]
        dx, failed = S.solve(H, b, fpose)
        ref = _ref(H, b)
        assert failed == 0 and S.fronts[0] == 1, (rep, S.fronts)
        np.testing.assert_allclose(dx, ref, rtol=0, atol=3e-7 * max(1.0, np.abs(ref).max()))
    n = 6 * P
    top, bottom = S.fronts[1], S.fronts[2]
    for where in (5, top - 3, top + 2, n - bottom + 1, n - 4, n // 2):
        H, b, fpose = systems[0]
        Hb = H.copy()
        Hb[where, where] = -1.0
        dx, failed = S.solve(Hb, b, fpose)
        assert failed == 1 and np.all(dx == 0.0), where
    dx, failed = S.solve(*systems[0])
    assert failed == 0


@pytest.mark.parametrize("kernel", ["window", "skyline"])
def test_two_workgroup_solves_replay_from_a_captured_graph(kernel):
    """the hand-shake words of the two-workgroup kernels carry a generation number counted ON THE DEVICE (meta[24..27] for the
    window kernel's two fronts, meta[28..31] for the skyline kernel), not a kernel argument: a launch captured into a hipGraph gets
    a new number on every replay.  The same launch is replayed over alternating systems (and a failing one in between); every
    replay must give that system's solution -- a replay that took the previous replay's flags for its own would read the partner's
    separator block before it is written.  Which kernel a fresh workspace gets follows the verdict on the last workspace that had
    one (launch_ba_solve): a banded system in another workspace first -> the window kernel; one with a long coupling -> skyline"""
    P, w = (63, 8) if kernel == "window" else (63, 4)
    rng = np.random.default_rng(99)
    prime = _SkylineSolver(40, init=True)
    Hp, bp, fpp = _pose_system(rng, 40, 4, extra=() if kernel == "window" else [(39, 0)])
    prime.solve(Hp, bp, fpp)
    fpd = torch.from_numpy(fpp).cuda()
    want = 1 if kernel == "window" else 2
    for _ in range(1100):     # (a workspace sent to the skyline kernel is offered to the window kernel again every 1024th solve)
        if prime.lib.dba_ba_solver_verdict(*prime.dims, ctypes.c_void_p(prime.ws.data_ptr()), prime.nbytes) == want:
            break
        _lib.check(prime.lib.dba_ba_solve_skyline(*prime.dims, 1e-4, 0.1, ctypes.c_void_p(fpd.data_ptr()), ctypes.c_void_p(prime.ws.data_ptr()),
                                                  prime.nbytes, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "prime")
        torch.cuda.synchronize()
    assert prime.lib.dba_ba_solver_verdict(*prime.dims, ctypes.c_void_p(prime.ws.data_ptr()), prime.nbytes) == want
    prime.solve(Hp, bp, fpp)  # (launch_ba_solve notes the verdict it finds)
    S = _SkylineSolver(P, init=True)
    systems = [_pose_system(rng, P, w) for _ in range(3)]
    dx, failed = S.solve(*systems[0])          # eager once: the plan for this structure is in the workspace, attributes are set
    assert failed == 0 and S.fronts[0] == 1
    n, lay, ws = 6 * P, S.lay, S.ws
    fp = torch.from_numpy(systems[0][2]).cuda()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            _lib.check(S.lib.dba_ba_solve_skyline(*S.dims, 1e-4, 0.1, ctypes.c_void_p(fp.data_ptr()), ctypes.c_void_p(ws.data_ptr()),
                                                  S.nbytes, stream), "dba_ba_solve_skyline (capture)")
    torch.cuda.synchronize()
    for rep in range(24):
        H, b, _ = systems[rep % 3]
        bad = rep % 7 == 5
        Hl = np.tril(H) + np.triu(np.full_like(H, 1e300), 1)
        if bad:
            Hl[200, 200] = -1.0
        ws[lay.H:lay.H + 8 * n * n].view(torch.float64).copy_(torch.from_numpy(Hl.reshape(-1)).cuda())
        ws[lay.b:lay.b + 8 * n].view(torch.float64).copy_(torch.from_numpy(b).cuda())
        ws[lay.dx:lay.dx + 4 * n].view(torch.float32).fill_(7.0)
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        dx = ws[lay.dx:lay.dx + 4 * n].view(torch.float32).cpu().numpy()
        meta = ws[lay.meta:lay.meta + 32].view(torch.int32).cpu().numpy()
        if bad:
            assert int(meta[1]) == 1 and np.all(dx == 0.0), rep
            continue
        ref = _ref(H, b)
        assert int(meta[1]) == 0 and int(meta[4]) == 1, (rep, meta[:8])
        np.testing.assert_allclose(dx, ref, rtol=0, atol=3e-7 * max(1.0, np.abs(ref).max()))
    meta = ws[lay.meta:lay.meta + 128].view(torch.int32).cpu().numpy()
    c = 24 if kernel == "window" else 28
    assert int(meta[c]) == int(meta[c + 1]) >= 25 and int(meta[52 - c]) == 0, meta[24:32]   # both workgroups counted every launch
