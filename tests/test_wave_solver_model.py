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
