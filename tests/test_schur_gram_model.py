"""numpy model of the per-source-frame Schur kernel's index logic (csrc/ba_kernels.hip: ba_schur_gram_kernel).

The kernel stacks a frame's values per pixel as x = [w | E_0 | E_1 | ...], accumulates the lower triangle of
G = sum_k q_k x_k x_k^T in 16 x 16 tiles of v_mfma_f64_16x16x4_f64 and scatters tile (ti, tj), register r, lane l to
H / b (both triangles, or -- what dba_ba asks for -- the lower triangle only).  This model walks the same lanes, registers, pixel groups and scatter rule in float64 and compares with the
definition the row-pair kernel implements (schur_block + EEt6x6 + Ev6x1, droid_kernels.cu:1046-1138, :1297-1391):
for every ordered pair of rows (a, b) of one source frame, H[tgt_a, tgt_b] -= E_a diag(Q) E_b^T, b[tgt_a] -= E_a (Q o w).
What it pins is the bookkeeping (operand lanes, tile order, triangle handling, duplicates, ragged chunks), not the rounding.
"""
import numpy as np
import pytest


F32_LAYOUT = False   # result rows of the float instruction: 4 (l >> 4) + r; of the float64 one: (l >> 4) + 4 r


def _row(l, r):
    return 4 * (l >> 4) + r if F32_LAYOUT else (l >> 4) + 4 * r


def mfma_16x16x4(a_lanes, b_lanes, acc):
    """D = A B + C with the gfx950 operand layout: lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15];
    register r of lane l holds D[(l >> 4) + 4 r][l & 15] for the FLOAT64 instruction (the layout ba_solve_kernel's
    trailing update relies on) and D[4 (l >> 4) + r][l & 15] for the float one (ba_linearize_kernel's): the kernel has
    both forms (gram_mac / gram_mac_f32) and scatters accordingly."""
    A = np.zeros((16, 4)); B = np.zeros((4, 16))
    for l in range(64):
        A[l & 15, l >> 4] = a_lanes[l]
        B[l >> 4, l & 15] = b_lanes[l]
    D = A @ B
    for l in range(64):
        for r in range(4):
            acc[l, r] += D[_row(l, r), l & 15]


def gram_kernel_model(E, Q, w, rows, tgts, P, HW, nch, lower=False, nw=8):
    """E [nrows_total, 6, HW], rows/tgts = the frame's list; returns the H, b contribution of the frame"""
    n6 = 6 * P
    H = np.zeros((n6, n6)); b = np.zeros(n6)
    nrows = len(rows)
    R = 1 + 6 * nrows
    T = (R + 15) // 16
    NT = T * (T + 1) // 2
    cpx = ((HW + nch - 1) // nch + 15) // 16 * 16
    for ch in range(nch):
        c0, c1 = ch * cpx, min(HW, ch * cpx + cpx)
        if c0 >= c1:
            continue
        red = np.zeros((NT, 4, 64))
        ngroups = (c1 - c0 + 15) // 16
        per = (ngroups + nw - 1) // nw
        for wv in range(nw):
            acc = np.zeros((NT, 64, 4))
            for g in range(wv * per, min(ngroups, wv * per + per)):
                # operand of lane (li, lk) for tile t, k-step s: value 16 t + li of pixel c0 + 16 g + 4 lk + s
                def val(c, pix):
                    if pix >= c1:
                        return 0.0
                    if c == 0:
                        return w[pix]
                    a = (c - 1) // 6
                    if a >= nrows:
                        return w[pix]  # padding lanes read w: never looked at
                    return E[rows[a], (c - 1) % 6, pix]
                for s in range(4):
                    ops = np.zeros((T, 64)); qs = np.zeros(64)
                    for l in range(64):
                        pix = c0 + 16 * g + 4 * (l >> 4) + s
                        qs[l] = Q[pix] if pix < c1 else 0.0
                        for t in range(T):
                            ops[t, l] = val(16 * t + (l & 15), pix)
                    for ti in range(T):
                        for tj in range(ti + 1):
                            mfma_16x16x4(ops[ti] * qs, ops[tj], acc[ti * (ti + 1) // 2 + tj])
            for i in range(NT):
                for r in range(4):
                    red[i, r] += acc[i, :, r]
        ti = tj = 0
        for idx in range(NT):
            for r in range(4):
                for l in range(64):
                    i, j = 16 * ti + _row(l, r), 16 * tj + (l & 15)
                    if i < R and j <= i and i >= 1:
                        s = -red[idx, r, l]
                        a, ca = (i - 1) // 6, (i - 1) % 6
                        hr = 6 * tgts[a] + ca
                        if j == 0:
                            b[hr] += s
                        else:
                            bq, cb = (j - 1) // 6, (j - 1) % 6
                            hc = 6 * tgts[bq] + cb
                            if i == j:
                                H[hr, hc] += s
                            elif not lower:
                                H[hr, hc] += s
                                H[hc, hr] += s
                            elif hr == hc:
                                H[hr, hc] += 2 * s
                            else:
                                H[max(hr, hc), min(hr, hc)] += s
            tj += 1
            if tj > ti:
                ti, tj = ti + 1, 0
    return H, b


def definition(E, Q, w, rows, tgts, P):
    n6 = 6 * P
    H = np.zeros((n6, n6)); b = np.zeros(n6)
    for x, (ra, ta) in enumerate(zip(rows, tgts)):
        b[6 * ta:6 * ta + 6] -= E[ra] @ (Q * w)
        for rb, tb in zip(rows, tgts):
            H[6 * ta:6 * ta + 6, 6 * tb:6 * tb + 6] -= (E[ra] * Q) @ E[rb].T
    return H, b


@pytest.mark.parametrize("nrows,HW,nch,seed", [(1, 64, 1, 0), (2, 100, 2, 1), (5, 144, 2, 2), (6, 121, 3, 3), (11, 80, 1, 4),
                                               (15, 48, 2, 5), (3, 37, 4, 6)])
@pytest.mark.parametrize("f32_layout", [False, True])
def test_gram_scatter_matches_the_pairwise_definition(nrows, HW, nch, seed, f32_layout):
    global F32_LAYOUT
    F32_LAYOUT = f32_layout
    rng = np.random.default_rng(seed)
    P = 7
    total = nrows + 3
    E = rng.standard_normal((total, 6, HW))
    Q = rng.uniform(0.5, 2.0, HW)
    w = rng.standard_normal(HW)
    rows = list(rng.permutation(total)[:nrows])
    tgts = list(rng.integers(0, P, nrows))  # duplicates on purpose (two edges to one target, own row + edge)
    H, b = gram_kernel_model(E, Q, w, rows, tgts, P, HW, nch)
    Hd, bd = definition(E, Q, w, rows, tgts, P)
    assert np.allclose(H, Hd, rtol=0, atol=1e-9 * np.abs(Hd).max())
    assert np.allclose(b, bd, rtol=0, atol=1e-9 * np.abs(bd).max())
    Hl, bl = gram_kernel_model(E, Q, w, rows, tgts, P, HW, nch, lower=True, nw=4)
    assert np.allclose(Hl, np.tril(Hd), rtol=0, atol=1e-9 * np.abs(Hd).max()) and np.allclose(bl, bd)
