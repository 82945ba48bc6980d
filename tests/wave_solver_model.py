"""Lane-level model of csrc/ba_solve_wave.hip (the window solver): every array below has one entry per lane of a wave,
every helper mirrors one hardware operation (the f64 16x16x4 matrix instruction with its operand layouts, v_readlane, LDS
reads / writes by per-lane address), and the data flow is the kernel's, statement by statement -- what the kernel spreads
over three factor waves (one tile row of the window each, roles rotating with the window), a loader and a substitution wave
is executed here in program order, which changes no index and no value.
Test infrastructure: tests/test_wave_solver_model.py holds it against dense solves, which pins the index arithmetic of the
kernel (panel store, pivot block read with indices XOR k and its cofactor inverse, window rotation, the substitution ring)
and its admission test without a GPU."""
import numpy as np

LANES = np.arange(64)
LI = LANES & 15          # column of the C / D layout, row of the A layout, column of the B layout
LK = LANES >> 4          # row group of the C / D layout, k of the A and B layouts
NT = 3                   # tile rows of the window (16 rows each): 3 or 4 in the kernel (set_window)
PAN_ROWS = 16 * NT


def set_window(nt):
    """the kernel's template parameter: 3 (48-row window) or 4 (64 rows, four factor waves)"""
    global NT, PAN_ROWS
    NT, PAN_ROWS = nt, 16 * nt


def mfma_16x16x4(a, b, c):
    """v_mfma_f64_16x16x4_f64: a[lane] = A[lane & 15][lane >> 4], b[lane] = B[lane >> 4][lane & 15],
    c[r][lane] = C[(lane >> 4) + 4 r][lane & 15]; returns D = A B + C in the layout of c"""
    A = np.zeros((16, 4))
    B = np.zeros((4, 16))
    A[LI, LK] = a
    B[LK, LI] = b
    P = A @ B
    d = np.empty_like(c)
    for r in range(4):
        d[r] = c[r] + P[LK + 4 * r, LI]
    return d


def invert_row0(P):
    """row 0 of the inverse of the symmetric positive definite 4x4 block whose LOWER triangle is P[i][j] (per lane), by
    cofactors (the formulas of the kernel's wv_invert_row0_cof); returns (w0..w3, ok)"""
    a, b, c = P[0][0], P[1][0], P[1][1]
    d, e, f, g = P[2][0], P[2][1], P[3][0], P[3][1]
    h, i, j = P[2][2], P[3][2], P[3][3]
    m01, m02, m03 = d * g - e * f, d * i - h * f, d * j - i * f      # 2x2 minors of rows 2, 3
    m12, m13, m23 = e * i - h * g, e * j - i * g, h * j - i * i
    C0 = c * m23 - e * m13 + g * m12                                 # cofactors of row 0 (rows 1..3 expanded along row 1 = b c e g)
    C1 = -(b * m23 - e * m03 + g * m02)
    C2 = b * m13 - c * m03 + g * m01
    C3 = -(b * m12 - c * m02 + e * m01)
    det = a * C0 + b * C1 + d * C2 + f * C3
    det2 = a * c - b * b
    det3 = d * (b * e - c * d) - e * (a * e - b * d) + h * det2
    ok = (a > 0) & (det2 > 0) & (det3 > 0) & (det > 0)
    idet = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0)
    return [C0 * idet, C1 * idet, C2 * idet, C3 * idet], ok


def band_ok(fpose, n):
    """the kernel's admission test: with the skyline made monotone, every column of step s must end inside the window of
    its tile column (rows below 16 (s >> 2) + 48)"""
    P = len(fpose)
    g = np.minimum.accumulate(np.asarray(fpose)[::-1])[::-1]          # suffix minimum
    last = np.array([max(p for p in range(P) if g[p] <= q) for q in range(P)])
    npad = (n + 15) // 16 * 16
    for s in range(npad // 4):
        c = 4 * s
        if c >= n:
            continue
        q3 = min(c + 3, n - 1) // 6
        if 6 * last[q3] + 5 > 16 * (s >> 2) + PAN_ROWS - 1:
            return False
    return True


class WaveSolver:
    def __init__(self, H, b, lm, ep):
        self.n = n = H.shape[0]
        self.H, self.b, self.lm, self.ep = H, b, lm, ep
        self.np_ = (n + 15) // 16 * 16
        self.S = self.np_ // 4
        self.PAN = np.full((self.S, PAN_ROWS, 4), np.nan)      # LDS: panel storage, NaN = never written
        self.ZST = np.full((self.S, 4), np.nan)
        self.BV = np.zeros(self.np_ + 64)
        self.BV[:n] = b
        self.bad = np.zeros(64, bool)

    # ---- a tile of the damped, padded system in the C layout: reg r, lane -> (16 TI + (lane >> 4) + 4 r, 16 TJ + (lane & 15))
    def load_tile(self, TI, TJ):
        n, np_ = self.n, self.np_
        t = np.zeros((4, 64))
        for r in range(4):
            row, col = 16 * TI + LK + 4 * r, 16 * TJ + LI
            for l in range(64):
                i, j = row[l], col[l]
                if i >= np_ or j >= np_:
                    v = 0.0
                elif i >= n or j >= n:
                    v = 1.0 if i == j else 0.0
                else:
                    v = self.H[max(i, j), min(i, j)]
                    if i == j:
                        v += self.ep + self.lm * v
                t[r, l] = v
        return t

    def extract(self, s_next, acc):
        """the columns of step s_next out of the tile column 0 of the window -> PAN[s_next]"""
        qn = s_next & 3
        m = (LI >> 2) == qn
        for t in range(NT):
            for r in range(4):
                rows = 16 * t + LK + 4 * r
                self.PAN[s_next, rows[m], (LI & 3)[m]] = acc[(t, 0)][r][m]

    def factor(self):
        S = self.S
        acc = {(ti, tj): self.load_tile(ti, tj) for ti in range(NT) for tj in range(ti + 1)}
        nxt = [self.load_tile(NT, 1 + j) for j in range(NT)]
        self.extract(0, acc)
        for s in range(S):
            tb, q = s >> 2, s & 3
            cl, c = 4 * q, 4 * s
            pan = self.PAN[s]
            # 1. the pivot block as lane group k sees it (indices XOR k), lower triangle only; the panel rows of the window
            Pp = [[None] * 4 for _ in range(4)]
            for i in range(4):
                for j in range(i + 1):
                    ii, jj = i ^ LK, j ^ LK
                    Pp[i][j] = pan[cl + np.maximum(ii, jj), np.minimum(ii, jj)]
            raw = [[np.where(16 * t + LI > cl + 3, pan[16 * t + LI, j ^ LK], 0.0) for j in range(4)] for t in range(NT)]
            assert not any(np.isnan(x).any() for row in Pp for x in row if x is not None)
            assert not any(np.isnan(x).any() for t in raw for x in t)
            # 2. this lane group's row of the inverse
            w, ok = invert_row0(Pp)
            self.bad |= ~ok
            # 3. operands
            a = [-raw[t][0] for t in range(NT)]
            u = [sum(w[j] * raw[t][j] for j in range(4)) for t in range(NT)]
            # 4. rank-4 update of the window
            for ti in range(NT):
                for tj in range(ti + 1):
                    acc[(ti, tj)] = mfma_16x16x4(a[ti], u[tj], acc[(ti, tj)])
            # 5. W over the pivot block's rows (lanes with li == 0)
            for l in range(0, 64, 16):
                k = l >> 4
                for j in range(4):
                    pan[cl + k, k ^ j] = w[j][l]
            # 6. right-hand side: z = W b1, b2 -= R z
            b1 = self.BV[c:c + 4].copy()
            Wf = pan[cl:cl + 4, :].copy()
            z = Wf @ b1
            for l in range(PAN_ROWS):
                if l > cl + 3:
                    self.BV[16 * tb + l] -= pan[l, :] @ z
            self.ZST[s] = z
            # 7. the window moves on by one tile column
            if q == 3:
                for ti in range(NT - 1):
                    for tj in range(ti + 1):
                        acc[(ti, tj)] = acc[(ti + 1, tj + 1)]
                for j in range(NT):
                    acc[(NT - 1, j)] = nxt[j]
                nxt = [self.load_tile(tb + 1 + NT, tb + 2 + j) for j in range(NT)]
            # 8. next step's panel
            if s + 1 < S:
                self.extract(s + 1, acc)

    def substitute(self):
        S = self.S
        slot, k = LANES >> 2, LANES & 3
        v = np.zeros(64)
        x = np.zeros(self.np_ + 64)
        for sp in range(S - 1, -1, -1):
            tbp, clp, cp = sp >> 2, 4 * (sp & 3), 4 * sp
            vk = np.array([v[4 * (sp & 15) + m] for m in range(4)])          # v_readlane
            Wf = self.PAN[sp, clp:clp + 4, :]
            x1 = self.ZST[sp] - Wf @ vk
            x[cp:cp + 4] = x1
            d = (sp - 1 - slot) & 15
            s = sp - 1 - d
            lrow = cp - 16 * (s >> 2)
            valid = (s >= 0) & (lrow + 3 <= PAN_ROWS - 1)
            for l in range(64):
                if valid[l]:
                    v[l] += self.PAN[s[l], lrow[l]:lrow[l] + 4, k[l]] @ x1
            v[slot == (sp & 15)] = 0.0
        return x[:self.n]

    def solve(self):
        self.factor()
        x = self.substitute()
        failed = self.bad.any() or not np.isfinite(x).all()
        return (np.zeros(self.n) if failed else x), failed
