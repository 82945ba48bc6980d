"""A numpy model of the two-front (twisted) block LDL^T that csrc/ba_solve_tile.hip runs on banded systems, checked
against numpy.linalg.solve.  It restates the kernel's rule set at tile level - the monotone skyline, the choice of
c1 (tile columns each front eliminates while their fill regions are disjoint), the activity rules of the two fronts,
what a column panel (top front) and a row panel (bottom front) contain, and the substitution order - so that the
rules themselves are pinned on the CPU; the kernel is tested against a host Cholesky in test_gpu_solve.py."""
import numpy as np
import pytest


def _skyline(A, n):
    KT, T = n // 4, n // 4 + 1
    first = [min(I, KT - 1) for I in range(T)]
    for I in range(T):
        for K in range(min(I, KT - 1) + 1):
            if np.any(A[4 * I:4 * I + 4, 4 * K:4 * K + 4] != 0):
                first[I] = K
                break
    return first


def solve_two_fronts(H, b):
    n = H.shape[0]
    assert n % 4 == 0
    KT, T, npairs = n // 4, n // 4 + 1, n // 2
    S = np.zeros((n + 4, n))
    S[:n] = H
    S[n] = b  # the right-hand side rides along as row n
    first = _skyline(S, n)
    fp = first[:KT]
    for I in range(KT - 2, -1, -1):  # monotone skyline: bottom-up elimination fills a row as far left as any row below
        fp[I] = min(fp[I], fp[I + 1])
    cm = [max(I for I in range(KT) if fp[I] <= K) for K in range(KT)]
    c1 = 0
    while c1 < (KT - 1) // 2 and cm[c1] < fp[KT - 1 - c1]:
        c1 += 1
    if c1 < 2:
        c1 = 0
        first = _skyline(S, n)
    else:
        first[:KT] = fp
        first[T - 1] = min(first[T - 1], fp[KT - c1])
    frozen = set(range(KT - c1, KT)) if c1 else set()
    nsteps = npairs - 2 * c1
    Ptop, Pbot, pinv = {}, {}, {}

    def el(i, k):  # symmetric access to the working matrix (row n = right-hand side)
        return S[i, k] if (i >= k or i >= n) else S[k, i]

    def inv2(j0):
        pa, pb, pc = el(j0, j0), el(j0 + 1, j0), el(j0 + 1, j0 + 1)
        det = pa * pc - pb * pb
        assert pa > 0 and det > 0
        return np.array([[pc, -pb], [-pb, pa]]) / det

    def publish_top(sp):  # column panel: rows of the tile column, except the rows the bottom front owns
        j0, Kp = 2 * sp, sp >> 1
        P = np.zeros((n + 4, 2))
        for I in range(Kp, T):
            if I in frozen:
                continue
            for i in range(4 * I, min(4 * I + 4, n + 1)):
                P[i] = (el(i, j0), el(i, j0 + 1))
        Ptop[sp], pinv[sp] = P, inv2(j0)

    def publish_bottom(slot, sp):  # row panel: the pivot rows over the columns left of the pivot, rhs at index n
        j0 = 2 * sp
        P = np.zeros((n + 4, 2))
        for k in range(j0):
            P[k] = (el(j0, k), el(j0 + 1, k))
        P[n] = (S[n, j0], S[n, j0 + 1])
        Pbot[slot], pinv[sp] = P, inv2(j0)

    publish_top(0)
    if c1:
        publish_bottom(0, npairs - 1)
    for s in range(nsteps):
        Ks, h = s >> 1, s & 1
        sb = npairs - 1 - s
        Kb, hb = sb >> 1, sb & 1
        new = S.copy()
        for I in range(T):
            for K in range(min(I, KT - 1) + 1):
                rhs = I == T - 1
                sstart = max(first[I], first[K])
                top = Ks >= sstart and (K > Ks or (K == Ks and h == 0)) and I not in frozen
                bot = (s < 2 * c1 and Kb <= cm[K] and (K < Kb or (K == Kb and hb == 1))
                       and (I < Kb or (I == Kb and hb == 1) or rhs))
                assert not (top and bot), "the fronts must never meet in a tile"
                if not (top or bot):
                    continue
                P, Pi = (Pbot[s], pinv[sb]) if bot else (Ptop[s], pinv[s])
                for i in range(4 * I, min(4 * I + 4, n + 1)):
                    for k in range(4 * K, 4 * K + 4):
                        if I == K and k > i:
                            continue
                        new[i, k] -= P[i] @ Pi @ P[k]  # operands by global row / column index, for both fronts
        S = new
        if s + 1 < nsteps:
            publish_top(s + 1)
        if s + 1 < 2 * c1:
            publish_bottom(s + 1, sb - 1)

    def coef(i, j):
        sj = j >> 1
        return Pbot[npairs - 1 - sj][i, j & 1] if sj >= npairs - 2 * c1 else Ptop[sj][i, j & 1]

    t = np.array([coef(n, j) for j in range(n)])
    x = np.zeros(n)
    order = (list(range(npairs - 1 - 2 * c1, 2 * c1 - 1, -1)) + list(range(2 * c1 - 1, -1, -1))
             + list(range(npairs - 2 * c1, npairs)))  # reverse elimination: middle, top front, bottom front
    done = set()
    for s in order:
        xs = pinv[s] @ t[2 * s:2 * s + 2]
        x[2 * s:2 * s + 2] = xs
        done.add(s)
        for j in range(n):
            if (j >> 1) not in done:
                t[j] -= coef(2 * s, j) * xs[0] + coef(2 * s + 1, j) * xs[1]
    return x, c1


def _banded(rng, n, band, extra=()):
    H = np.zeros((n, n))
    for i in range(n):
        for j in range(max(0, i - band + 1), i + 1):
            H[i, j] = H[j, i] = rng.uniform(-1, 1) / (1 + i - j)
        H[i, i] = 6 + rng.uniform(0, 2)
    for i, j in extra:
        H[i, j] = H[j, i] = 0.37
    return H, np.sin(1.3 * np.arange(n))


@pytest.mark.parametrize("n,band,extra,two", [
    (24, 2, (), True), (48, 9, (), True), (64, 6, (), True), (96, 24, (), True), (144, 30, (), True),
    (24, 6, (), False),            # too wide for its size: one front
    (48, 48, (), False),           # dense
    (96, 12, ((93, 2),), False),   # arrow
    (96, 12, ((60, 30),), True),   # a long coupling in the middle
    (96, 8, ((90, 70),), True),    # non-monotone skyline inside the bottom front
    (96, 8, ((30, 4),), True),     # ... inside the top front
])
def test_two_front_elimination_model(n, band, extra, two):
    rng = np.random.default_rng(n + band)
    H, b = _banded(rng, n, band, extra)
    x, c1 = solve_two_fronts(H, b)
    assert (c1 > 0) == two
    np.testing.assert_allclose(x, np.linalg.solve(H, b), rtol=0, atol=1e-12)


def test_two_front_model_with_a_sparse_right_hand_side():
    rng = np.random.default_rng(3)
    H, _ = _banded(rng, 96, 16)
    for lo, hi in ((92, 96), (0, 4), (44, 48)):
        b = np.zeros(96)
        b[lo:hi] = 1.0
        x, c1 = solve_two_fronts(H, b)
        assert c1 > 0
        np.testing.assert_allclose(x, np.linalg.solve(H, b), rtol=0, atol=1e-12)
