"""A numpy model of the two-workgroup cut of csrc/ba_solve_band.hip, checked against numpy.linalg.solve: the decision
rule (tile-level skyline -> suffix minimum -> top block / separator / bottom block), the two local systems (top block
then S; bottom block MIRRORED then S, whose S x S block and S right-hand side start from zero), the sum of the two
Schur-complement contributions on S, the common separator solve and the back-substitution of each block.  The kernel
itself is tested against a host Cholesky in test_gpu_solve.py; here the rules are pinned on the CPU."""
import numpy as np
import pytest

XS_MAX = 64


def tile_skyline(H):
    n = H.shape[0]
    Tl = (n + 4) // 4 - 1          # row tiles that hold matrix rows only (the last tile row is the right-hand side's)
    KT = (n + 3) // 4
    first = []
    for I in range(Tl):
        rows = H[4 * I:min(4 * I + 4, n), :min(4 * I + 4, n)]
        nz = np.nonzero(np.any(rows != 0, axis=0))[0]
        first.append(min(int(nz.min()) // 4 if len(nz) else KT - 1, min(I, KT - 1)))
    flast = KT - 1
    if n % 4 == 2:                 # the two rows that share the right-hand side's tile row
        nz = np.nonzero(np.any(H[n - 2:, :n - 2] != 0, axis=0))[0]
        flast = (int(nz.min()) if len(nz) else n - 2) // 4
    return first, flast


def choose_split(H):
    """(ua, ub) = first unknown of S, first unknown of the bottom block; None if the system stays whole"""
    n = H.shape[0]
    first, flast = tile_skyline(H)
    Tl, KT = len(first), (n + 3) // 4
    sm = [0] * Tl
    m = min(KT, flast)
    for I in range(Tl - 1, -1, -1):
        m = min(m, first[I])
        sm[I] = m
    best = (0, 0, 0)
    kb = 1
    while 4 * kb + 8 <= n:
        ub = n - 4 * kb
        Ta = min(sm[min(ub // 4, Tl - 1)], (ub - 4) // 4)
        ws = ub - 4 * Ta
        score = min(Ta, kb) if (Ta > 0 and 4 <= ws <= XS_MAX) else 0
        if score > best[0] or (score == best[0] and score > 0 and Ta < best[1]):
            best = (score, Ta, kb)
        kb += 1
    if n < 96 or best[0] < 6:
        return None
    return 4 * best[1], n - 4 * best[2]


def eliminate_block(A, rhs, nown):
    """eliminate the first nown unknowns of the local system; returns the Schur complement and reduced right-hand side
    on the rest, and what the back-substitution of the block needs"""
    A11, A12, A22 = A[:nown, :nown], A[:nown, nown:], A[nown:, nown:]
    X = np.linalg.solve(A11, np.column_stack([A12, rhs[:nown]]))
    S = A22 - A12.T @ X[:, :-1]
    r = rhs[nown:] - A12.T @ X[:, -1]
    return S, r, (A11, A12)


def solve_split(H, b, lm, ep):
    n = H.shape[0]
    Hd = H.copy()
    Hd[np.diag_indices(n)] += ep + lm * np.diag(H)
    cut = choose_split(H)
    if cut is None:
        return np.linalg.solve(Hd, b), None
    ua, ub = cut
    # no row from ub on may reach a column before ua: the blocks only meet through S
    assert not np.any(H[ub:, :ua] != 0)
    s_idx = np.arange(ua, ub)
    # workgroup 0: top block, then S (with the original S x S block and right-hand side)
    i0 = np.concatenate([np.arange(0, ua), s_idx])
    S0, r0, keep0 = eliminate_block(Hd[np.ix_(i0, i0)], b[i0], ua)
    # workgroup 1: bottom block mirrored, then S; its S x S block and S right-hand side start from zero
    i1 = np.concatenate([np.arange(n - 1, ub - 1, -1), s_idx])
    A1 = Hd[np.ix_(i1, i1)].copy()
    b1 = b[i1].copy()
    A1[n - ub:, n - ub:] = 0.0
    b1[n - ub:] = 0.0
    S1, r1, keep1 = eliminate_block(A1, b1, n - ub)
    xs = np.linalg.solve(S0 + S1, r0 + r1)          # both workgroups: the same sum, the same solve
    x = np.zeros(n)
    x[s_idx] = xs
    x[:ua] = np.linalg.solve(keep0[0], b[:ua] - keep0[1] @ xs)
    x[np.arange(n - 1, ub - 1, -1)] = np.linalg.solve(keep1[0], b1[:n - ub] - keep1[1] @ xs)
    return x, (ua, ub)


def _system(rng, n, band):
    A = np.zeros((n, n))
    for i in range(n):
        for j in range(max(0, i - band + 1), i + 1):
            A[i, j] = A[j, i] = rng.uniform(-1, 1) / (1 + i - j)
    A[np.diag_indices(n)] = 6.0 + rng.uniform(0, 2, n)
    return A, np.sin(1.3 * np.arange(n))


@pytest.mark.parametrize("n,band", [(180, 36), (186, 36), (186, 13), (198, 61), (240, 30), (378, 36), (378, 60), (384, 12)])
def test_split_solution_equals_the_direct_solve(n, band):
    rng = np.random.default_rng(n + band)
    H, b = _system(rng, n, band)
    x, cut = solve_split(H, b, 1e-4, 0.1)
    assert cut is not None
    ua, ub = cut
    assert ua % 4 == 0 and (n - ub) % 4 == 0 and band - 4 <= ub - ua <= XS_MAX and abs(ua - (n - ub)) <= 8
    Hd = H.copy()
    Hd[np.diag_indices(n)] += 0.1 + 1e-4 * np.diag(H)
    np.testing.assert_allclose(x, np.linalg.solve(Hd, b), rtol=0, atol=1e-12)


@pytest.mark.parametrize("n,band,i,j", [(240, 24, 100, 20), (240, 24, 230, 150), (240, 24, 140, 100), (378, 36, 200, 170)])
def test_a_long_coupling_moves_or_widens_the_separator(n, band, i, j):
    rng = np.random.default_rng(n + i)
    H, b = _system(rng, n, band)
    H[i, j] = H[j, i] = 0.37
    x, cut = solve_split(H, b, 1e-4, 0.1)
    Hd = H.copy()
    Hd[np.diag_indices(n)] += 0.1 + 1e-4 * np.diag(H)
    np.testing.assert_allclose(x, np.linalg.solve(Hd, b), rtol=0, atol=1e-12)
    if cut is not None:
        assert not np.any(H[cut[1]:, :cut[0]] != 0)


@pytest.mark.parametrize("n,band", [(210, 70), (240, 240), (60, 12)])
def test_systems_without_a_separator_stay_whole(n, band):
    rng = np.random.default_rng(n)
    H, b = _system(rng, n, band)
    assert choose_split(H) is None


def test_an_arrow_keeps_the_system_whole():
    rng = np.random.default_rng(1)
    H, b = _system(rng, 240, 24)
    H[239, 0] = H[0, 239] = 0.2   # the last pose sees the first one (loop closure): no row range is cut off from column 0
    assert choose_split(H) is None
