"""CPU: the lane-level numpy model of the five-wave window solver (tests/wave_solver_model.py <-> csrc/ba_solve_wave.hip) against
dense solves of the damped system (droid_kernels.cu:1252-1253 damping, :1263-1266 zero update on failure), and the kernel's
admission test: which skylines stay inside the 48-row window."""
import numpy as np
import pytest

import wave_solver_model as wsm
from wave_solver_model import WaveSolver, band_ok


def _system(P, w, seed, extra=None):
    """P poses of 6 unknowns, pose p coupled with p-w .. p (+ extra pairs); diagonally dominant; fpose as stage 0 builds it"""
    rng = np.random.default_rng(seed)
    n = 6 * P
    H = np.zeros((n, n))
    fpose = list(range(P))
    pairs = [(p, q) for p in range(P) for q in range(max(0, p - w), p + 1)] + list(extra or [])
    for p, q in pairs:
        Bk = rng.standard_normal((6, 6)) * 0.3
        if p == q:
            Bk = Bk + Bk.T
        H[6 * p:6 * p + 6, 6 * q:6 * q + 6] += Bk
        if p != q:
            H[6 * q:6 * q + 6, 6 * p:6 * p + 6] += Bk.T
        fpose[p] = min(fpose[p], q)
    H += np.eye(n) * (np.abs(H).sum(1).max() + 1.0)
    return H, rng.standard_normal(n), fpose


@pytest.mark.parametrize("P,w", [(24, 4), (24, 3), (25, 4), (8, 4), (3, 2), (29, 4), (63, 4), (24, 1), (16, 0), (2, 1), (1, 0)])
def test_model_solves_banded_systems(P, w):
    H, b, fpose = _system(P, w, 100 * P + w)
    assert band_ok(fpose, 6 * P)
    lm, ep = 1e-4, 0.1
    ref = np.linalg.solve(H + np.diag(ep + lm * np.diag(H)), b)
    x, failed = WaveSolver(np.tril(H), b, lm, ep).solve()      # (only the lower triangle is read)
    assert not failed
    np.testing.assert_allclose(x, ref, rtol=0, atol=1e-13 * np.abs(ref).max())


def test_admission():
    """4 poses wide always fits; 5 and more, an arrow (a coupling back to an early pose) and a dense system do not; an extra
    coupling inside the window does"""
    for P in (8, 24, 25, 29, 63, 64):
        assert band_ok(_system(P, 4, 1)[2], 6 * P)
    assert not band_ok(_system(24, 5, 2)[2], 144)
    assert not band_ok(_system(24, 6, 3)[2], 144)
    assert not band_ok(_system(24, 2, 4, extra=[(20, 3)])[2], 144)
    assert not band_ok(_system(24, 23, 5)[2], 144)
    assert band_ok(_system(24, 2, 6, extra=[(9, 5)])[2], 144)
    assert band_ok(_system(24, 3, 7, extra=[(23, 19)])[2], 144)


def test_model_reports_an_indefinite_system():
    H, b, _ = _system(24, 4, 9)
    H[70, 70] = -5.0
    x, failed = WaveSolver(np.tril(H), b, 1e-4, 0.1).solve()
    assert failed and np.all(x == 0.0)


@pytest.mark.parametrize("P,w", [(24, 5), (24, 6), (24, 7), (12, 7), (40, 6)])
def test_model_with_the_64_row_window(P, w):
    """four factor waves / tile rows: bands the 48-row window refuses"""
    H, b, fpose = _system(P, w, 31 * P + w)
    assert not band_ok(fpose, 6 * P)
    wsm.set_window(4)
    try:
        assert band_ok(fpose, 6 * P)
        ref = np.linalg.solve(H + np.diag(0.1 + 1e-4 * np.diag(H)), b)
        x, failed = WaveSolver(np.tril(H), b, 1e-4, 0.1).solve()
        assert not failed
        np.testing.assert_allclose(x, ref, rtol=0, atol=1e-13 * np.abs(ref).max())
        assert not band_ok(_system(24, 9, 1)[2], 144)
    finally:
        wsm.set_window(3)


@pytest.mark.parametrize("P,w,nt", [(24, 4, 3), (24, 3, 3), (25, 4, 3), (29, 4, 3), (23, 2, 3), (16, 4, 3), (40, 4, 3), (63, 4, 3),
                                     (24, 5, 4), (24, 6, 4), (40, 6, 4), (24, 1, 3), (12, 4, 3)])
def test_model_two_fronts_around_a_separator(P, w, nt):
    """the top and the bottom part eliminated independently (the bottom one in reverse order), the separator's system assembled
    from what both leave of it, the three solutions stitched together"""
    H, b, fpose = _system(P, w, 57 * P + w)
    wsm.set_window(nt)
    try:
        plan = wsm.split_plan(fpose, 6 * P)
        if P >= 16:
            assert plan is not None
        if plan is None:
            pytest.skip("no separator for this system: one front")
        a_t, sep, a_b = plan
        assert a_t % 4 == 0 and a_b % 4 == 0 and sep % 4 == 0 and a_t + sep + a_b == 6 * P + ((6 * P) & 2)
        ref = np.linalg.solve(H + np.diag(0.1 + 1e-4 * np.diag(H)), b)
        x, failed = wsm.TwoFrontSolver(np.tril(H), b, 1e-4, 0.1, fpose).solve()
        assert not failed
        np.testing.assert_allclose(x, ref, rtol=0, atol=1e-13 * np.abs(ref).max())
    finally:
        wsm.set_window(3)


def test_split_plan_refuses_what_two_fronts_cannot_take():
    assert wsm.split_plan(_system(4, 2, 1)[2], 24) is None                 # too small to be worth it
    assert wsm.split_plan(_system(24, 2, 2, extra=[(20, 3)])[2], 144) is None   # an arrow: every separator is crossed
    H, b, fpose = _system(24, 4, 3)
    H[70, 70] = -5.0
    x, failed = wsm.TwoFrontSolver(np.tril(H), b, 1e-4, 0.1, fpose).solve()
    assert failed and np.all(x == 0.0)


@pytest.mark.parametrize("PR", [48, 64])
def test_backward_address_recurrence(PR):
    """the closed form the kernel's backward substitution steps its panel-row address with (csrc/ba_solve_wave.hip: u = sp - 1 -
    slot) against the direct index of the lane's pending step, for every lane and step"""
    PD = PR * 4
    slot, kk = np.arange(64) >> 2, np.arange(64) & 3
    C = slot * PD + 16 * (slot & 3) + kk + 16
    dmax = PR // 4 - 2 - (slot & 3)
    for S in (4, 8, 12, 36, 96):
        u = S - 2 - slot
        idx_inc = C + 16 * u + 16 * (PD - 16) * (u >> 4)
        for sp in range(S - 1, -1, -1):
            dd = (sp - 1 - slot) & 15
            sq = sp - 1 - dd
            lrow = 4 * sp - 16 * (sq >> 2)
            valid = (sq >= 0) & (lrow + 3 <= PR - 1)
            assert np.array_equal(valid, (u >= 0) & ((u & 15) <= dmax))
            assert np.array_equal((sq * PD + lrow * 4 + kk)[valid], idx_inc[valid])
            u = u - 1                                                     # the kernel's step: rp -= 16 or 16 (PD - 15)
            idx_inc = idx_inc - np.where(slot == ((sp - 1) & 15), 16 * (PD - 15), 16)


# ---- round 6: the panel store as a ring (64-row window on systems of more than 45 poses: BASELINE's 64-KF / 512-edge graph)

def _kernel_ring(n, nt=4):
    """E, K as csrc/ba_solve_wave.hip::wv_ring_early / wv_lds_doubles_k choose them"""
    npad = (n + 15) // 16 * 16
    S = npad // 4
    fixed = (S * 4 + npad + 88 + nt * 4 * 64) * 8
    E = wsm.ring_early(S, 16 * nt * 4 * 8, fixed)
    return E, S - E


def test_ring_sizes_the_kernel_chooses():
    assert _kernel_ring(6 * 48) == (0, 72)             # 48 poses: the last size whose whole 64-row store fits
    assert _kernel_ring(6 * 49) == (4, 72)
    assert _kernel_ring(6 * 63) == (24, 72)            # BASELINE's 64-KF window (63 free poses)
    assert _kernel_ring(6 * 63, 5) == (40, 56)         # ... whose reduced system is 8-10 poses wide: the 80-row window
    assert _kernel_ring(6 * 60, 5) == (36, 56)
    assert _kernel_ring(6 * 37, 5) == (0, 56) and _kernel_ring(6 * 38, 5) == (4, 56)
    for P in range(1, 65):
        assert _kernel_ring(6 * P, 3)[0] == 0          # the 48-row window never needs it
        for nt in (4, 5):
            E, K = _kernel_ring(6 * P, nt)
            assert E >= 0 and E % 4 == 0 and E <= K and (E == 0 or K >= 32)


def _skyline_64_512():
    """the pose-level couplings of the REDUCED system of BASELINE configs[3] (synthetic.graph_64_512: |i - j| <= 4 plus (i, i + 5)
    for i < 10; first pose fixed: 63 free poses): two poses are coupled when they see the same source frame's depths -- a
    frame's own pose and all its targets, pairwise (what stage 0's skyline table records: ba_kernels.hip::ba_prepare_kernel)"""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dba-fusion_amd"))
    from dbaf_amd import synthetic
    ii, jj = synthetic.graph_64_512()
    pairs = set()
    for m in set(ii):
        grp = sorted({m} | {j for i, j in zip(ii, jj) if i == m})
        pairs |= {(a - 1, b - 1) for a in grp for b in grp if a > b >= 1}
    return sorted(pairs)


@pytest.mark.parametrize("reload", ["late", "early"])
@pytest.mark.parametrize("P,w", [(63, 5), (64, 6), (63, 7), (49, 5), (61, 6), (50, 7)])
def test_model_with_the_ring_of_panels(P, w, reload):
    H, b, fpose = _system(P, w, 17 * P + w)
    E, K = _kernel_ring(6 * P)
    assert E > 0
    wsm.set_window(4)
    wsm.set_ring(K, reload)
    try:
        assert band_ok(fpose, 6 * P)
        ref = np.linalg.solve(H + np.diag(0.1 + 1e-4 * np.diag(H)), b)
        x, failed = WaveSolver(np.tril(H), b, 1e-4, 0.1).solve()
        assert not failed
        np.testing.assert_allclose(x, ref, rtol=0, atol=1e-13 * np.abs(ref).max())
    finally:
        wsm.set_window(3)
        wsm.set_ring(None)


def test_model_on_the_literal_64_512_skyline():
    pairs = _skyline_64_512()
    P = 63
    H, b, fpose = _system(P, 0, 5, extra=pairs)
    assert max(p - f for p, f in enumerate(fpose)) == 10
    wsm.set_window(4)
    assert not band_ok(fpose, 6 * P)                   # 8-10 poses wide: round 5 sent this graph to the skyline kernel
    E, K = _kernel_ring(6 * P, 5)
    wsm.set_window(5)
    wsm.set_ring(K)
    try:
        assert band_ok(fpose, 6 * P)
        ref = np.linalg.solve(H + np.diag(0.1 + 1e-4 * np.diag(H)), b)
        x, failed = WaveSolver(np.tril(H), b, 1e-4, 0.1).solve()
        assert not failed
        np.testing.assert_allclose(x, ref, rtol=0, atol=1e-13 * np.abs(ref).max())
    finally:
        wsm.set_window(3)
        wsm.set_ring(None)


@pytest.mark.parametrize("S,E", [(96, 32), (92, 32), (72, 16), (80, 16), (96, 16), (96, 24), (76, 4), (92, 28), (88, 44)])
def test_backward_address_recurrence_across_the_ring_seam(S, E):
    """the same recurrence with the ring's base: -E PD, + K PD for the steps below the seam; a lane's pending step slot + 16 m
    crosses the seam when its u becomes 16 ceil((E - slot) / 16) - 1"""
    PR = 64
    PD, K = PR * 4, S - E
    slot, kk = np.arange(64) >> 2, np.arange(64) & 3
    C = slot * PD + 16 * (slot & 3) + kk + 16
    dmax = PR // 4 - 2 - (slot & 3)
    u = S - 2 - slot
    idx = C + 16 * u + 16 * (PD - 16) * (u >> 4)
    pending = slot + 16 * (u >> 4)
    idx = idx + np.where((u >= 0) & (pending < E), K - E, -E) * PD
    for sp in range(S - 1, -1, -1):
        dd = (sp - 1 - slot) & 15
        sq = sp - 1 - dd
        lrow = 4 * sp - 16 * (sq >> 2)
        valid = (sq >= 0) & (lrow + 3 <= PR - 1)
        assert np.array_equal(valid, (u >= 0) & ((u & 15) <= dmax))
        ring_slot = sq - E + np.where(sq < E, K, 0)
        assert (ring_slot[valid] >= 0).all() and (ring_slot[valid] < K).all()
        assert np.array_equal((ring_slot * PD + lrow * 4 + kk)[valid], idx[valid]), sp
        u = u - 1
        idx = idx - np.where(slot == ((sp - 1) & 15), 16 * (PD - 15), 16)
        idx = idx + np.where(u == 16 * ((E - slot + 15) >> 4) - 1, K * PD, 0)


@pytest.mark.parametrize("P,w,ring", [(24, 8, False), (24, 9, False), (37, 10, False), (38, 10, True), (30, 8, False), (63, 8, True), (64, 10, True),
                                       (63, 9, True), (45, 8, True), (52, 10, True), (63, 4, True)])
def test_model_with_the_80_row_window(P, w, ring):
    """five factor waves / tile rows: bands of 8-10 poses (covisibility radius 4-5); a panel spans up to 19 steps, so a lane of the
    backward pass collects for two pending steps at a time; beyond 38 poses the store is a ring"""
    H, b, fpose = _system(P, w, 41 * P + w)
    E, K = _kernel_ring(6 * P, 5)
    assert (E > 0) == ring
    wsm.set_window(5)
    for reload in (("late", "early") if ring else ("late",)):
        wsm.set_ring(K if ring else None, reload)
        try:
            assert band_ok(fpose, 6 * P)
            ref = np.linalg.solve(H + np.diag(0.1 + 1e-4 * np.diag(H)), b)
            x, failed = WaveSolver(np.tril(H), b, 1e-4, 0.1).solve()
            assert not failed
            np.testing.assert_allclose(x, ref, rtol=0, atol=1e-13 * np.abs(ref).max())
        finally:
            wsm.set_ring(None)
    wsm.set_window(3)


@pytest.mark.parametrize("S,E", [(96, 48), (60, 0), (64, 16), (80, 32), (36, 0), (96, 40), (92, 36), (60, 4)])
def test_backward_address_recurrence_with_the_far_accumulator(S, E):
    """80-row window: the second pending step of a lane (16 steps further down) reads the same rows of x, 64 rows further into its
    panel: rp - 16 PD + 256, + K PD when the seam lies between the two panels (u >> 4 == E / 16)"""
    PR = 80
    PD, K = PR * 4, S - E
    slot, kk = np.arange(64) >> 2, np.arange(64) & 3
    C = slot * PD + 16 * (slot & 3) + kk + 16
    dmax = PR // 4 - 2 - (slot & 3)
    u = S - 2 - slot
    idx = C + 16 * u + 16 * (PD - 16) * (u >> 4)
    pending = slot + 16 * (u >> 4)
    if E:
        idx = idx + np.where((u >= 0) & (pending < E), K - E, -E) * PD
    phys = lambda sq: (sq - E + np.where(sq < E, K, 0)) if E else sq   # noqa: E731
    for sp in range(S - 1, -1, -1):
        dd = (sp - 1 - slot) & 15
        sq = sp - 1 - dd
        lrow = 4 * sp - 16 * (sq >> 2)
        valid = (sq >= 0) & (lrow + 3 <= PR - 1)
        assert np.array_equal(valid, (u >= 0) & ((u & 15) <= dmax)) and np.array_equal(valid, u >= 0)
        assert np.array_equal((phys(sq) * PD + lrow * 4 + kk)[valid], idx[valid]), sp
        sq2 = sq - 16
        lrow2 = 4 * sp - 16 * (sq2 >> 2)
        valid2 = (sq2 >= 0) & (lrow2 + 3 <= PR - 1)
        assert np.array_equal(valid2, (u >= 16) & ((u & 15) <= dmax - 16))
        idx2 = idx - 16 * PD + 256 + (np.where((u >> 4) == ((E - slot + 15) >> 4), K * PD, 0) if E else 0)
        assert np.array_equal((phys(sq2) * PD + lrow2 * 4 + kk)[valid2], idx2[valid2]), sp
        u = u - 1
        idx = idx - np.where(slot == ((sp - 1) & 15), 16 * (PD - 15), 16)
        if E:
            idx = idx + np.where(u == 16 * ((E - slot + 15) >> 4) - 1, K * PD, 0)


def test_two_front_plan_of_the_literal_64_512_skyline():
    """what the two-workgroup kernel (csrc/ba_solve_wave.hip::wv_front_plan) must find for BASELINE configs[3]: 41 steps per front
    around a separator of 52 unknowns in the 80-row window; the two-front arithmetic solves it"""
    pairs = _skyline_64_512()
    H, b, fpose = _system(63, 0, 5, extra=pairs)
    wsm.set_window(5)
    try:
        assert wsm.split_plan(fpose, 378) == (164, 52, 164)
        ref = np.linalg.solve(H + np.diag(0.1 + 1e-4 * np.diag(H)), b)
        x, failed = wsm.TwoFrontSolver(np.tril(H), b, 1e-4, 0.1, fpose).solve()
        assert not failed
        np.testing.assert_allclose(x, ref, rtol=0, atol=1e-13 * np.abs(ref).max())
    finally:
        wsm.set_window(3)
