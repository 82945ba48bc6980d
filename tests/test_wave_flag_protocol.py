"""CPU: the flag protocol of the window solver (dba-fusion_amd/csrc/ba_solve_wave.hip) under random interleavings of its waves.

The numpy model (wave_solver_model.py) checks the arithmetic of one wave after the other; what it cannot see is the ORDER in
which the waves' LDS accesses land.  This is a discrete-event restatement of who writes which panel row, W block and ring slot,
who announces it through which monotone counter, and who reads it behind which wait -- the kernel's run_column / loader /
substitution loops with everything but the accesses and the flags removed -- run under a random scheduler.  It must never read
something that has not been written, and no counter may step back.

Round 5 found such a race on the device (one wrong solve in ~40 cold starts): flagE[NT - 1] changes hands at a rotation from the
wave furthest behind to the old chain wave, which waited for nobody.  `handover_wait=False` reproduces it here; the shipped
protocol (True) passes every schedule."""
import random

import pytest

def _waves(TB, handover_wait, log, NT=3, K=None, gate=True):
    """K: the panel store as a ring of K steps (round 6; None: every panel has its own place); gate=False leaves out the loader's
    wait for the substitution wave (the bug the wait prevents)"""
    S = 4 * TB
    E = 0 if K is None else S - K
    E4, K4 = E // 4, (K or S) // 4

    def factor(w):
        role = w
        yield ("write", ("panel", 0, w))
        yield ("publish", ("E", w), 1)
        for tb in range(TB):
            for q in range(4):
                s = 4 * tb + q
                rot = q == 3
                more = s + 1 < S
                if role >= 1 and rot and not more:
                    break                                         # (the very last step only concerns the pivot tile)
                if role >= 1:
                    yield ("await", ("W",), s + 1)
                    yield ("read", ("W", s))
                    for t in range(role):
                        yield ("await", ("E", t), s + 1)
                        yield ("read", ("panel", s, t))
                yield ("read", ("panel", s, role))              # its own rows (role 0: the pivot block, requested a step ahead)
                if role == 0:
                    yield ("write", ("W", s))
                    yield ("publish", ("W",), s + 1)
                if not rot:
                    yield ("write", ("panel", s + 1, role))
                    yield ("publish", ("E", role), s + 2)
                elif role >= 1:
                    if more:
                        yield ("write", ("panel", s + 1, role - 1))
                        yield ("publish", ("E", role - 1), s + 2)
                elif more:                                        # the pivot tile is finished: take the entering tile row
                    yield ("await", ("L",), tb + 1)
                    yield ("read", ("ring", tb))
                    yield ("write", ("panel", s + 1, NT - 1))
                    if handover_wait:
                        yield ("await", ("E", NT - 1), s + 1)
                    yield ("publish", ("E", NT - 1), s + 2)
                    yield ("publish", ("C",), tb + 1)
            role = NT - 1 if role == 0 else role - 1

    def loader():
        for k in range(TB - 1):
            yield ("await", ("C",), k)
            if E and gate and 4 * (k + NT) + 5 - K > 0:
                yield ("await", ("F",), 4 * (k + NT) + 5 - K)
            yield ("write", ("ring", k))
            yield ("publish", ("L",), k + 1)
        if E:                                                     # the way back: the early tile columns return
            yield ("await", ("B",), 1)
            for c in range(E4 - 1, -1, -1):
                yield ("await", ("B",), TB - (c + K4))
                yield ("reload", c)
                yield ("publish", ("R",), E4 - c)

    def subst():
        for s in range(S):
            yield ("await", ("W",), s + 1)
            for t in range(NT):
                yield ("await", ("E", t), s + 1)
            yield ("read", ("W", s))
            for t in range(NT):
                yield ("read", ("panel", s, t))
            if s < E:
                yield ("spill", s)
            if E and (s & 3) == 3:
                yield ("publish", ("F",), s + 1)
        if E:
            for c in range(TB - 1, -1, -1):
                if c - NT < E4:
                    yield ("await", ("R",), min(E4, E4 - (c - NT)))
                yield ("backread", c)                             # panels of the tile columns c - NT .. c
                yield ("publish", ("B",), TB - c)

    return [factor(w) for w in range(NT)] + [subst(), loader()]


def _run(TB, handover_wait, seed, bias, NT=3, K=None, gate=True, slow_wave=None):
    """random scheduler; bias > 1 lets one factor wave (or `slow_wave`: NT = the substitution wave, NT + 1 = the loader) run that
    much less often (the wave that falls behind on the device)"""
    rng = random.Random(seed)
    log = []
    waves = _waves(TB, handover_wait, log, NT, K, gate)
    S = 4 * TB
    E = 0 if K is None else S - K
    E4, K4 = E // 4, (K or S) // 4
    flags, written, taken, violations = {}, set(), set(), []
    reads, spilled, back_done, in_lds = {}, set(), set(), set(range(E4, TB))
    # who reads row t of panel s on the way forward: the roles t .. NT - 1 (not at the very last step) and the substitution wave
    expected = lambda s, t: (NT - t if s < S - 1 else (1 if t == 0 else 0)) + 1   # noqa: E731
    pending = [next(g, None) for g in waves]
    slow = rng.randrange(NT) if slow_wave is None else slow_wave
    weights = [1.0 / bias if w == slow else 1.0 for w in range(NT + 2)]
    steps = 0
    while any(a is not None for a in pending):
        runnable = [i for i, a in enumerate(pending) if a is not None and not (a[0] == "await" and flags.get(a[1], 0) < a[2])]
        assert runnable, "deadlock: %r" % (pending,)
        i = rng.choices(runnable, [weights[k] for k in runnable])[0]
        kind = pending[i][0]
        if kind == "write":
            item = pending[i][1]
            # the loader's ring is ONE slot: tile row k may only land there when row k - 1 has been taken out
            if item[0] == "ring" and item[1] > 0 and ("ring", item[1] - 1) not in taken:
                violations.append(("slot overwritten before it was read", i, item))
            if E and item[0] == "panel" and item[1] >= K:
                old = ("panel", item[1] - K, item[2])              # the occupant of that place in the ring
                if reads.get(old, 0) < expected(old[1], old[2]):
                    violations.append(("panel overwritten before everybody had read it", i, old, reads.get(old, 0)))
                if old[1] not in spilled:
                    violations.append(("panel overwritten before it was copied out", i, old))
                written.discard(old)
            written.add(item)
        elif kind == "read":
            if pending[i][1] not in written:
                violations.append(("read before write", i, pending[i][1]))
            taken.add(pending[i][1])
            reads[pending[i][1]] = reads.get(pending[i][1], 0) + 1
        elif kind == "spill":
            spilled.add(pending[i][1])
        elif kind == "reload":
            c = pending[i][1]
            # it takes the places of tile column c + K / 4: every backward iteration that reads that column (c + K/4 .. c + K/4 + 4)
            # must be over, and so must the way forward
            if not all(j in back_done for j in range(c + K4, min(c + K4 + NT + 1, TB))):
                violations.append(("reload over a tile column the way back still reads", i, c))
            if any(reads.get(("panel", s, t), 0) < expected(s, t) for s in range(4 * (c + K4), 4 * (c + K4) + 4) for t in range(NT)):
                violations.append(("reload over a panel the way forward still reads", i, c))
            if not all(s in spilled for s in range(4 * c, 4 * c + 4)):
                violations.append(("reload of a panel that was never copied out", i, c))
            in_lds.discard(c + K4)
            in_lds.add(c)
        elif kind == "backread":
            c = pending[i][1]
            for j in range(max(c - NT, 0), c + 1):
                if j not in in_lds:
                    violations.append(("the way back reads a tile column that is not in LDS", i, c, j))
            back_done.add(c)
        elif kind == "publish":
            if pending[i][2] <= flags.get(pending[i][1], 0):
                violations.append(("counter steps back", i, pending[i][1], flags.get(pending[i][1], 0), pending[i][2]))
            flags[pending[i][1]] = pending[i][2]
        pending[i] = next(waves[i], None)
        steps += 1
    return violations


@pytest.mark.parametrize("TB,NT", [(2, 3), (3, 3), (9, 3), (9, 4), (3, 4), (9, 5), (6, 5)])
def test_shipped_protocol_never_reads_what_is_not_written(TB, NT):
    """48-row window (three factor waves) and 64-row window (four)"""
    for seed in range(150):
        for bias in (1.0, 4.0, 20.0):
            assert _run(TB, True, seed, bias, NT) == [], (TB, NT, seed, bias)


def test_the_round5_handover_race_is_what_the_wait_closes():
    """without the ring taker's wait for flagE[NT - 1] the last tile row's counter is announced over the laggard's head: rows are
    read before they are written and the counter steps back -- found by schedules that let one factor wave fall behind"""
    for NT in (3, 4):
        bad = [v for seed in range(300) for v in _run(9, False, seed, 20.0, NT)]
        assert any(v[0] == "read before write" and v[2][0] == "panel" and v[2][2] == NT - 1 for v in bad)
        assert any(v[0] == "counter steps back" and v[2] == ("E", NT - 1) for v in bad)


# ---- round 6: the panel store as a ring of K steps (the 64-row window on BASELINE's 64-KF / 512-edge graph: S = 96, K = 64)

@pytest.mark.parametrize("TB,K,NT", [(24, 72, 4), (23, 60, 4), (20, 64, 4), (12, 32, 4), (24, 56, 5), (23, 56, 5), (15, 56, 5), (16, 32, 5)])
def test_ring_of_panels_is_never_overwritten_early_nor_read_late(TB, K, NT):
    """no panel's place is rewritten before every factor wave and the substitution wave have read it and it has been copied out;
    on the way back no tile column is read before it has returned, none returns over a column still being read -- with one of
    the factor waves, the substitution wave or the loader falling behind"""
    for seed in range(12):
        for slow in (None, NT, NT + 1):
            for bias in (1.0, 20.0):
                assert _run(TB, True, seed, bias, NT, K, True, slow) == [], (TB, K, NT, seed, slow, bias)


def test_the_loaders_wait_for_the_substitution_wave_is_what_keeps_the_ring_safe():
    """without it a substitution wave that falls behind finds its panel replaced (on the device it never is that far behind:
    the margin is K - 4 NT - 5 steps; the wait is there so that it cannot be)"""
    bad = [v for seed in range(40) for v in _run(24, True, seed, 200.0, 4, 64, False, 4)]
    assert any(v[0].startswith("panel overwritten") for v in bad)
