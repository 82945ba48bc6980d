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
RING_K = None            # panels the LDS store holds (None: all of them); set_ring
RELOAD = "late"          # when the early panels come back during the backward pass: as late / as early as the flags allow


def set_window(nt):
    """the kernel's template parameter: 3 (48-row window) or 4 (64 rows, four factor waves)"""
    global NT, PAN_ROWS
    NT, PAN_ROWS = nt, 16 * nt


def set_ring(k, reload="late"):
    """round 6: the panel store as a ring of k steps (the kernel's wv_ring_early decides E = S - k from the LDS budget); the
    first S - k panels are copied to scratch by the forward substitution and brought back during the backward pass"""
    global RING_K, RELOAD
    RING_K, RELOAD = k, reload


def ring_early(S, pd_bytes, fixed_bytes, budget=160 * 1024):
    """the kernel's wv_ring_early: the smallest number of early panels (whole tile columns) to evict so that the store fits"""
    E = 0
    while E == 0 or (2 * E <= S and S - E >= 32):
        if (S - E) * pd_bytes + fixed_bytes <= budget:
            return E
        E += 4
    return -1


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


class _RingView:
    """PAN[s, ...] of the kernel's layout: step s lives in slot s - E (s >= E) or s - E + K (s < E); a read of a step the slot
    does not hold at that moment is an error of the protocol"""

    def __init__(self, inst):
        self.i = inst

    def slot(self, s):
        i = self.i
        return s - i.E + (i.K if s < i.E else 0)

    def _split(self, key):
        if not isinstance(key, tuple):
            key = (key,)
        return key[0], key[1:]

    def __getitem__(self, key):
        s, rest = self._split(key)
        i = self.i
        if np.ndim(s) == 0:
            s = int(s)
            if s >= i.S:                      # (the model's harmless extra panel behind the last step)
                return np.full((PAN_ROWS, 4), np.nan)[rest] if rest else np.full((PAN_ROWS, 4), np.nan)
            assert i.holds[self.slot(s)] == s, "panel %d is not in its slot (which holds %d)" % (s, i.holds[self.slot(s)])
            v = i.PAN_[self.slot(s)]
            return v[rest] if rest else v      # (a view: writes through `pan = self.PAN[s]` land in the store)
        raise TypeError("vector step index: use Instance.pan_read")

    def __setitem__(self, key, val):
        s, rest = self._split(key)
        i = self.i
        s = int(s)
        if s >= i.S:
            return
        sl = self.slot(s)
        if i.holds[sl] != s:
            # a new occupant: the previous one must be gone for good (forward: copied out; backward: not read any more)
            prev = i.holds[sl]
            assert prev < 0 or prev in i.released, "panel %d would overwrite panel %d, which is still needed" % (s, prev)
            i.PAN_[sl] = np.nan
            i.holds[sl] = s
        i.PAN_[sl][rest] = val


class Instance:
    """one (sub-)system the waves of the kernel work on: `elem(i, j)` = the damped local matrix (i >= j < nloc; identity padding
    beyond nloc is added here), `rhs(i)`; `sel` = steps to eliminate (all of them for a complete solve)"""

    def __init__(self, elem, rhs, nloc, sel=None):
        self.elem, self.nloc = elem, nloc
        self.np_ = (nloc + 15) // 16 * 16
        self.S = self.np_ // 4
        self.sel = self.S if sel is None else sel
        self.K = self.S if RING_K is None else min(RING_K, self.S)
        self.E = self.S - self.K
        assert self.E % 4 == 0 and self.E <= self.K
        self.PAN_ = np.full((self.K, PAN_ROWS, 4), np.nan)          # LDS: panel storage (a ring of K steps), NaN = never written
        self.holds = [-1] * self.K                                  # which step's panel a slot holds
        self.SPILL = np.full((self.E, PAN_ROWS, 4), np.nan)         # global scratch: the early panels
        self.PAN = _RingView(self)
        self.ZST = np.full((self.S, 4), np.nan)
        self.BV = np.zeros(self.np_ + 64)
        self.BV[:nloc] = [rhs(i) for i in range(nloc)]
        self.bad = np.zeros(64, bool)
        self.acc = None
        self.released = set()                                       # panels whose slot may be taken by another step

    # ---- a tile in the C layout: reg r, lane -> (16 TI + (lane >> 4) + 4 r, 16 TJ + (lane & 15))
    def load_tile(self, TI, TJ):
        nloc, np_ = self.nloc, self.np_
        t = np.zeros((4, 64))
        for r in range(4):
            row, col = 16 * TI + LK + 4 * r, 16 * TJ + LI
            for l in range(64):
                i, j = row[l], col[l]
                if i >= np_ or j >= np_:
                    v = 0.0
                elif i >= nloc or j >= nloc:
                    v = 1.0 if i == j else 0.0
                else:
                    v = self.elem(max(i, j), min(i, j))
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
        """the first `sel` steps of the elimination, the right-hand side riding along"""
        acc = {(ti, tj): self.load_tile(ti, tj) for ti in range(NT) for tj in range(ti + 1)}
        nxt = [self.load_tile(NT, 1 + j) for j in range(NT)]
        self.extract(0, acc)
        for s in range(self.sel):
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
            if s < self.E:                      # the substitution wave has read the whole panel: the early ones leave for scratch
                self.SPILL[s] = pan
                self.released.add(s)
            # 7. the window moves on by one tile column
            if q == 3:
                for ti in range(NT - 1):
                    for tj in range(ti + 1):
                        acc[(ti, tj)] = acc[(ti + 1, tj + 1)]
                for j in range(NT):
                    acc[(NT - 1, j)] = nxt[j]
                nxt = [self.load_tile(tb + 1 + NT, tb + 2 + j) for j in range(NT)]
            # 8. next step's panel (the kernel stops at the last eliminated step; one more here is harmless)
            if s + 1 < self.S:
                self.extract(s + 1, acc)
        self.acc = acc

    def dump(self, a, sep):
        """after a partial elimination (a = 4 sel): the lower triangle of the block [a, a + sep)^2 out of the window's tiles
        (tile (ti, tj) of the window sits at tile row / column (a >> 4) + ti / tj of the local system)"""
        tb = a >> 4
        D = np.full((sep, sep), np.nan)
        for (ti, tj), t in self.acc.items():
            for r in range(4):
                row, col = 16 * (tb + ti) + LK + 4 * r, 16 * (tb + tj) + LI
                for l in range(64):
                    i, j = row[l] - a, col[l] - a
                    if 0 <= j <= i < sep:
                        D[i, j] = t[r, l]
        assert not np.isnan(D[np.tril_indices(sep)]).any()
        return D

    def substitute(self, given=None):
        """backward substitution over all steps; `given`: the unknowns behind the eliminated part (local 4 sel ...), solved
        elsewhere -- their steps only hand their values on"""
        Sx = self.S if given is None else self.sel + (len(given) + 3) // 4
        slot, k = LANES >> 2, LANES & 3
        v = np.zeros(64)
        vfar = np.zeros(64)      # five tile rows: a panel spans up to 19 steps, the lane's step 16 further down already collects
        x = np.zeros(self.np_ + 64)
        if given is not None:
            x[4 * self.sel:4 * self.sel + len(given)] = given
        E4, K4, TB = self.E // 4, self.K // 4, self.S // 4

        def reload(c):
            for s4 in range(4 * c, 4 * c + 4):
                self.PAN[s4] = self.SPILL[s4]
        reloaded = E4                            # tile columns >= reloaded are in LDS
        self.released = set()                    # (forward releases are history: on the way back a slot is free when its column is done)
        for sp in range(Sx - 1, -1, -1):
            if self.E and (sp & 3) == 3 and given is None:
                c = sp >> 2                      # iteration c of the kernel's backward loop begins: it reads tile columns c - NT .. c
                if RELOAD == "early":            # everything the flags allow: column j may return once column j + K / 4 is finished
                    while reloaded > 0 and (reloaded - 1) + K4 > c:
                        reloaded -= 1
                        reload(reloaded)
                else:                            # only what this iteration waits for
                    while reloaded > max(c - NT, 0):
                        reloaded -= 1
                        assert (reloaded + K4) > c, "the slots of column %d are still in use" % (reloaded + K4)
                        reload(reloaded)
            clp, cp = 4 * (sp & 3), 4 * sp
            if sp < self.sel:
                vk = np.array([v[4 * (sp & 15) + m] for m in range(4)])          # v_readlane
                Wf = self.PAN[sp, clp:clp + 4, :]
                x1 = self.ZST[sp] - Wf @ vk
                x[cp:cp + 4] = x1
            else:
                x1 = x[cp:cp + 4].copy()
            d = (sp - 1 - slot) & 15
            s = sp - 1 - d
            lrow = cp - 16 * (s >> 2)
            valid = (s >= 0) & (s < self.sel) & (lrow + 3 <= PAN_ROWS - 1)
            s2 = s - 16                          # the lane's previous pending step: distances 16 .. (only with PAN_ROWS > 64)
            lrow2 = cp - 16 * (s2 >> 2)
            valid2 = (s2 >= 0) & (s2 < self.sel) & (lrow2 + 3 <= PAN_ROWS - 1)
            own = slot == (sp & 15)              # this lane's step has just been solved: the next one's collection takes over
            base = np.where(own, vfar, v)
            upd, upd2 = np.zeros(64), np.zeros(64)
            for l in range(64):
                if valid[l]:
                    upd[l] = self.PAN[s[l], lrow[l]:lrow[l] + 4, k[l]] @ x1
                if valid2[l]:
                    assert not own[l]
                    upd2[l] = self.PAN[s2[l], lrow2[l]:lrow2[l] + 4, k[l]] @ x1
            v = base + upd
            vfar = np.where(own, 0.0, vfar + upd2)
            if (sp & 3) == 0:                    # flagB: tile column sp >> 2 and everything above it is finished
                self.released.update(range(sp, self.S))
        return x


class WaveSolver:
    """the whole system on one front (what the kernel does when it finds no separator)"""

    def __init__(self, H, b, lm, ep):
        self.n = n = H.shape[0]
        self.inst = Instance(lambda i, j: H[i, j] + ((ep + lm * H[i, j]) if i == j else 0.0), lambda i: b[i], n)

    def solve(self):
        self.inst.factor()
        x = self.inst.substitute()[:self.n]
        failed = self.inst.bad.any() or not np.isfinite(x).all()
        return (np.zeros(self.n) if failed else x), failed


# ---- two fronts around a separator: the model of scratch/ba_solve_wave_two_fronts.hip (built, correct, slower than one front:
# profiles/SOLVER_NOTES.md); kept because the instance form above is also how the product kernel's arithmetic is pinned
def split_plan(fpose, n):
    """the kernel's decision (ba_solve_wave_split): the system padded to n4 = n + (n & 2) unknowns (identity behind the last one),
    top part [0, a_t) | separator [a_t, a_t + sep) | bottom part [a_t + sep, n4).  The smallest separator (a multiple of 4) such
    that no column of the top part reaches the bottom part, both blocks [a, a + sep) lie inside the window that is left when
    a front stops, and the bottom part -- eliminated in REVERSE order -- passes the window test too.  None: one front."""
    P = len(fpose)
    g = np.minimum.accumulate(np.asarray(fpose)[::-1])[::-1]
    last = np.array([max(p for p in range(P) if g[p] <= q) for q in range(P)])
    n4 = n + (n & 2)
    for sep in range(4, PAN_ROWS - 12 + 1, 4):
        a_t = ((n4 - sep) // 2) & ~3
        a_b = n4 - sep - a_t
        if a_t < 16 or a_b < 16:
            return None
        if 6 * last[(a_t - 1) // 6] + 5 >= a_t + sep:                    # a top column reaches past the separator
            continue
        if 4 * ((a_t // 4) % 4) + sep > PAN_ROWS or 4 * ((a_b // 4) % 4) + sep > PAN_ROWS:
            continue
        ok = True
        for s in range(a_b // 4):                                         # the bottom part's own window test, reversed order
            omin = n4 - 1 - (4 * s + 3)
            first = 6 * g[omin // 6] if omin < n else omin
            if (n4 - 1 - first) > 16 * (s >> 2) + PAN_ROWS - 1:
                ok = False
        if ok:
            return a_t, sep, a_b
    return None


class TwoFrontSolver:
    def __init__(self, H, b, lm, ep, fpose):
        self.n, self.H, self.b, self.lm, self.ep = H.shape[0], H, b, lm, ep
        self.plan = split_plan(fpose, self.n)

    def _elem(self, o_r, o_c):          # damped, padded system in ORIGINAL indices, o_r >= o_c
        n = self.n
        if o_r >= n:
            return 1.0 if o_r == o_c else 0.0
        v = self.H[o_r, o_c]
        return v + (self.ep + self.lm * v) if o_r == o_c else v

    def solve(self):
        n, n4 = self.n, self.n + (self.n & 2)
        a_t, sep, a_b = self.plan
        rhs = lambda o: self.b[o] if o < n else 0.0                      # noqa: E731
        top = Instance(lambda i, j: self._elem(i, j), rhs, a_t + sep, sel=a_t // 4)
        # bottom: local i' <-> original n4 - 1 - i' (lower triangle of the reversed matrix = upper of the original = its mirror)
        bot = Instance(lambda i, j: self._elem(n4 - 1 - j, n4 - 1 - i), lambda i: rhs(n4 - 1 - i), a_b + sep, sel=a_b // 4)
        top.factor()
        bot.factor()
        Dt, Db = top.dump(a_t, sep), bot.dump(a_b, sep)
        # separator: both dumps hold the damped original block plus their own front's update
        def selem(i, j):
            return Dt[i, j] + Db[sep - 1 - j, sep - 1 - i] - self._elem(a_t + i, a_t + j)
        def srhs(i):
            return top.BV[a_t + i] + bot.BV[a_b + sep - 1 - i] - rhs(a_t + i)
        mid = Instance(selem, srhs, sep)
        mid.factor()
        xs = mid.substitute()[:sep]
        xt = top.substitute(given=xs)[:a_t]
        xb = bot.substitute(given=xs[::-1])[:a_b]
        x = np.concatenate([xt, xs, xb[::-1]])[:n]
        failed = top.bad.any() or bot.bad.any() or mid.bad.any() or not np.isfinite(x).all()
        return (np.zeros(n) if failed else x), failed
