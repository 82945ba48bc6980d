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

def _waves(TB, handover_wait, log, NT=3):
    S = 4 * TB

    def factor(w):
        role = w
        yield ("write", ("panel", 0, w))
        yield ("publish", ("E", w), 1)
        for tb in range(TB):
            for q in range(4):
                s = 4 * tb + q
                rot = q == 3
                more = s + 1 < S
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
            yield ("write", ("ring", k))
            yield ("publish", ("L",), k + 1)

    def subst():
        for s in range(S):
            yield ("await", ("W",), s + 1)
            for t in range(NT):
                yield ("await", ("E", t), s + 1)
            yield ("read", ("W", s))
            for t in range(NT):
                yield ("read", ("panel", s, t))

    return [factor(w) for w in range(NT)] + [subst(), loader()]


def _run(TB, handover_wait, seed, bias, NT=3):
    """random scheduler; bias > 1 lets one factor wave run that much more often (the wave that falls behind on the device)"""
    rng = random.Random(seed)
    log = []
    waves = _waves(TB, handover_wait, log, NT)
    flags, written, taken, violations = {}, set(), set(), []
    pending = [next(g, None) for g in waves]
    slow = rng.randrange(NT)
    weights = [1.0 / bias if w == slow else 1.0 for w in range(NT)] + [1.0, 1.0]
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
            written.add(item)
        elif kind == "read":
            if pending[i][1] not in written:
                violations.append(("read before write", i, pending[i][1]))
            taken.add(pending[i][1])
        elif kind == "publish":
            if pending[i][2] <= flags.get(pending[i][1], 0):
                violations.append(("counter steps back", i, pending[i][1], flags.get(pending[i][1], 0), pending[i][2]))
            flags[pending[i][1]] = pending[i][2]
        pending[i] = next(waves[i], None)
        steps += 1
    return violations


@pytest.mark.parametrize("TB,NT", [(2, 3), (3, 3), (9, 3), (9, 4), (3, 4)])
def test_shipped_protocol_never_reads_what_is_not_written(TB, NT):
    """48-row window (three factor waves) and 64-row window (four)"""
    for seed in range(300):
        for bias in (1.0, 4.0, 20.0):
            assert _run(TB, True, seed, bias, NT) == [], (TB, NT, seed, bias)


def test_the_round5_handover_race_is_what_the_wait_closes():
    """without the ring taker's wait for flagE[NT - 1] the last tile row's counter is announced over the laggard's head: rows are
    read before they are written and the counter steps back -- found by schedules that let one factor wave fall behind"""
    for NT in (3, 4):
        bad = [v for seed in range(300) for v in _run(9, False, seed, 20.0, NT)]
        assert any(v[0] == "read before write" and v[2][0] == "panel" and v[2][2] == NT - 1 for v in bad)
        assert any(v[0] == "counter steps back" and v[2] == ("E", NT - 1) for v in bad)
