"""The persistent panel chain hands out ONE task per workgroup by a ticket counter, and a workgroup polls device flags
for its inputs: that is deadlock-free without any co-residency guarantee only if every task waits for tasks with EARLIER
tickets (a workgroup that holds ticket t is running or done once ticket t + 1 is taken).  This test enumerates every
ticket of a launch through the library's own map (csrc/chain_tasks.h via tgp_chain_task: the text the kernel decodes
its ticket with and launch_chain sizes the grid by) and checks, on the CPU,
  * that the tasks of the launch are exactly the ones the factorisation needs, each once;
  * that every wait of chain_kernel (csrc/chol.hip) is for an earlier ticket.
Round 6: the kernel reads its task from a TABLE the host builds from the same header (chain_build: the ticket order with
the K-batched updates of a policy folded in, tgp_chain_tasks); the same two checks run on that list for several policies --
an update (i, c, k) is covered exactly once, by its own task or inside ONE batch, and a batch waits for earlier tickets only."""
import ctypes as C
import functools

import pytest

from tinygp_amd import _ffi



@functools.lru_cache(maxsize=None)
def tasks_of(R, nblk, cb, ce):
    lib = _ffi.lib()
    n = C.c_int64()
    _ffi.check(lib.tgp_chain_task(R, nblk, cb, ce, -1, None, C.byref(n)), "tgp_chain_task")
    out = (C.c_int32 * 5)()
    tasks = []
    for t in range(n.value):
        _ffi.check(lib.tgp_chain_task(R, nblk, cb, ce, t, out, None), "tgp_chain_task")
        tasks.append(tuple(out))
    return tuple(tasks)


def table_of(R, nblk, cb, ce, batch, lag, rowlag, minrows=0, fwd=0):
    """the list launch_chain uploads: (kind, i, c, last k, part, first k) per ticket"""
    lib = _ffi.lib()
    n = C.c_int64()
    _ffi.check(lib.tgp_chain_tasks(R, nblk, cb, ce, batch, lag, rowlag, minrows, fwd, None, 0, C.byref(n)), "tgp_chain_tasks")
    out = (C.c_int32 * (6 * max(n.value, 1)))()
    _ffi.check(lib.tgp_chain_tasks(R, nblk, cb, ce, batch, lag, rowlag, minrows, fwd, out, n.value, C.byref(n)), "tgp_chain_tasks")
    return [tuple(out[6 * t: 6 * t + 6]) for t in range(n.value)]


import functools


@functools.lru_cache(maxsize=None)
def crit_parts():
    """CHAIN_CRIT_PARTS: workgroups sharing the one critical update.  Asked of the library on first USE, never at import:
    pytest imports every module at collection, and a dlopen there once put the library's HIP runtime into the process
    before torch's (GPUTEST_r04)."""
    parts = sum(1 for t in tasks_of(4, 2, 0, 2) if t[0] == 4)
    assert parts in (4, 8)
    return parts


def expected(R, nblk, cb, ce):
    """What a launch over block columns [cb, ce) has to do (kind, i, c, k, part)."""
    want = []
    if cb > 0:
        want.append((1, 0, cb, 0, 0))  # diag(cb): tile (cb, cb-1) is final since the launch before
    for k in range(cb, ce):
        factored_here = k + 1 < ce
        if factored_here:
            want += [(5, 0, k + 1, 0, 0), (1, 0, k + 1, 0, 0)]
        for i in range(k + 1, R):  # column k: tile (k+1, k) belongs to xsolve(k+1) when that exists
            if not (factored_here and i == k + 1):
                want.append((0, i, k, 0, 0))
        for c in range(k + 1, nblk):  # right-looking: column k -> every tile right of it
            for i in range(c, R):
                if (i, c) == (k + 1, k + 1):
                    continue  # the fold inside diag(k+1)
                if i == c:
                    want.append((3, i, c, k, 0))
                elif (i, c) == (k + 2, k + 1):
                    want += [(4, i, c, k, p) for p in range(crit_parts())]
                else:
                    want.append((2, i, c, k, 0))
    return want


# (nblk == R: the launch covers the whole rest of the matrix -- the order with the next step's diagonal lane in front of the bulk;
# otherwise a panel's launch -- only its two diagonal tasks move: csrc/chain_tasks.h)
SHAPES = [(1, 1, 0, 1), (2, 1, 0, 1), (2, 2, 0, 2), (3, 2, 0, 2), (8, 8, 0, 8), (12, 8, 0, 8), (32, 32, 0, 32), (40, 8, 0, 5),
          (40, 8, 5, 8), (9, 8, 3, 4), (64, 64, 0, 64), (70, 64, 60, 64), (128, 8, 0, 8), (16, 8, 7, 8), (3, 3, 0, 3), (4, 4, 0, 4),
          (5, 5, 0, 5), (16, 16, 0, 16), (16, 16, 4, 16), (32, 32, 8, 20), (48, 8, 0, 8), (64, 64, 30, 64)]


@pytest.mark.parametrize("shape", SHAPES, ids=[f"R{r}-nblk{n}-cols{a}to{b}" for r, n, a, b in SHAPES])
def test_every_task_once_and_waits_only_for_earlier_tickets(shape):
    R, nblk, cb, ce = shape
    tasks = tasks_of(R, nblk, cb, ce)
    assert sorted(tasks) == sorted(expected(R, nblk, cb, ce))
    assert len(set(tasks)) == len(tasks)
    at = {t: n for n, t in enumerate(tasks)}

    def final_of_tile(i, c):
        """ticket of the task that makes tile (i, c), i > c, final (None: final before the launch)"""
        if c < cb:
            return None
        if i == c + 1 and (5, 0, i, 0, 0) in at:
            return at[(5, 0, i, 0, 0)]
        return at[(0, i, c, 0, 0)]

    def factor_of(c):
        """ticket of diag(c) (None: factored in front of the launch / by an earlier one)"""
        return at.get((1, 0, c, 0, 0))

    def updates_of(i, c, upto):
        """tickets of the updates of tile (i, c) from columns cb .. upto-1 of this launch"""
        out = []
        for k in range(cb, upto):
            if i == c:
                out.append(at[(3, i, c, k, 0)])
            elif (4, i, c, k, 0) in at:
                out += [at[(4, i, c, k, p)] for p in range(crit_parts())]
            else:
                out.append(at[(2, i, c, k, 0)])
        return out

    for task, me in at.items():
        kind, i, c, k, part = task
        deps = []
        if kind == 0:  # solve(i, c): the tile with every update from this launch, L_cc
            deps = updates_of(i, c, c) + [factor_of(c)]
        elif kind == 5:  # xsolve(c): tile (c, c-1) with its updates, the progress of potf2(c-1)
            deps = updates_of(c, c - 1, c - 1) + [factor_of(c - 1)]
        elif kind == 1:  # diag(c): the solve of tile (c, c-1) (its fold), tile (c, c) with its updates
            deps = ([at[(5, 0, c, 0, 0)]] if (5, 0, c, 0, 0) in at else []) + updates_of(c, c, c - 1)
        else:  # update(i, c, k): both operands final, the tile carries every earlier update
            deps = [final_of_tile(i, k), final_of_tile(c, k)] + updates_of(i, c, k)
        late = [d for d in deps if d is not None and d >= me]
        assert not late, (task, me, late)


POLICIES = [(1, 1, 2, 0), (4, 1, 4, 0), (4, 1, 2, 0), (2, 1, 2, 0), (8, 2, 3, 0), (3, 1, 5, 0), (16, 1, 4, 0), (4, 1, 4, 32), (8, 1, 4, 20)]


@pytest.mark.parametrize("policy", POLICIES, ids=[f"batch{b}-lag{g}-rowlag{r}-minrows{m}" for b, g, r, m in POLICIES])
@pytest.mark.parametrize("shape", SHAPES, ids=[f"R{r}-nblk{n}-cols{a}to{b}" for r, n, a, b in SHAPES])
def test_table_covers_every_update_once_and_batches_wait_for_earlier_tickets(shape, policy):
    R, nblk, cb, ce = shape
    batch, lag, rowlag, minrows = policy
    table = table_of(R, nblk, cb, ce, batch, lag, rowlag, minrows)
    plain = tasks_of(R, nblk, cb, ce)
    if batch <= 1:
        assert [t[:5] for t in table] == list(plain) and all(t[5] == t[3] for t in table)
    # every task of the unbatched launch is there once -- as itself, or (kind 2 only) inside exactly one batch
    covered = {}
    for n, t in enumerate(table):
        kind, i, c, k, part, k0 = t
        if kind == 6:
            assert k - k0 + 1 >= 2 and k - k0 + 1 <= batch and k0 >= cb and k < ce
            assert i >= c + rowlag and k + 1 + lag <= c  # off the diagonal lane, done lag + 1 steps ahead of column c
            assert R - (k + 1) >= minrows                # only while the launch is throughput-bound
            for kk in range(k0, k + 1):
                assert (2, i, c, kk, 0) not in covered
                covered[(2, i, c, kk, 0)] = n
        else:
            assert k0 == k and t[:5] not in covered
            covered[t[:5]] = n
    assert sorted(covered) == sorted(expected(R, nblk, cb, ce))
    # a batch keeps the ticket position of its LAST update relative to everything that is not absorbed
    kept = [t[:5] if t[0] != 6 else (2, t[1], t[2], t[3], 0) for t in table]
    ks = set(kept)
    assert kept == [t for t in plain if t in ks]

    def final_of_tile(i, c):
        if c < cb:
            return None
        if i == c + 1 and (5, 0, i, 0, 0) in covered:
            return covered[(5, 0, i, 0, 0)]
        return covered[(0, i, c, 0, 0)]

    def updates_of(i, c, upto):
        out = []
        for k in range(cb, upto):
            if i == c:
                out.append(covered[(3, i, c, k, 0)])
            elif (4, i, c, k, 0) in covered:
                out += [covered[(4, i, c, k, p)] for p in range(crit_parts())]
            else:
                out.append(covered[(2, i, c, k, 0)])
        return out

    for me, t in enumerate(table):
        kind, i, c, k, part, k0 = t
        if kind == 0:
            deps = updates_of(i, c, c) + [covered.get((1, 0, c, 0, 0))]
        elif kind == 5:
            deps = updates_of(c, c - 1, c - 1) + [covered.get((1, 0, c - 1, 0, 0))]
        elif kind == 1:
            deps = ([covered[(5, 0, c, 0, 0)]] if (5, 0, c, 0, 0) in covered else []) + updates_of(c, c, c - 1)
        elif kind == 6:  # operands of EVERY column of the batch final (the kernel waits for the last: it implies the others)
            deps = [final_of_tile(i, kk) for kk in range(k0, k + 1)] + [final_of_tile(c, kk) for kk in range(k0, k + 1)]
            deps += updates_of(i, c, k0)
            # "final in column k implies final in the columns before": the solve of tile (i, k) waited for all of them
            for kk in range(k0, k):
                a, b = final_of_tile(i, kk), final_of_tile(i, k)
                assert a is None or a < b
        else:
            deps = [final_of_tile(i, k), final_of_tile(c, k)] + updates_of(i, c, k)
        late = [d for d in deps if d is not None and d >= me]
        assert not late, (t, me, late)


def test_batches_are_staggered_over_the_steps():
    """the boundaries of a tile's batches depend on (i + c): every step of a long launch ends about the same share of them"""
    table = table_of(64, 64, 0, 64, 4, 1, 4)
    ends = {}
    for kind, i, c, k, part, k0 in table:
        if kind == 6:
            ends[k] = ends.get(k, 0) + 1
    for k in range(8, 28):  # (the tiles right of a step get fewer as the launch advances: compare neighbouring steps)
        win = [ends.get(kk, 0) for kk in range(k, k + 4)]
        assert min(win) > 0.7 * max(win), (k, win)
    n_updates = sum(1 for t in tasks_of(64, 64, 0, 64) if t[0] == 2)
    n_left = sum(1 for t in table if t[0] == 2)
    n_batched = sum(t[3] - t[5] + 1 for t in table if t[0] == 6)
    assert n_left + n_batched == n_updates and n_batched > 0.7 * n_updates  # most of the one-launch tail's updates are batched


FWD_GROUP = 16  # CHAIN_FWD_GROUP of csrc/chain_tasks.h: row tiles per fupdate task


@pytest.mark.parametrize("policy", [(1, 1, 2, 0), (4, 1, 4, 0)], ids=["plain", "batch4"])
@pytest.mark.parametrize("shape", SHAPES, ids=[f"R{r}-nblk{n}-cols{a}to{b}" for r, n, a, b in SHAPES])
def test_forward_substitution_tasks_ride_along_and_wait_for_earlier_tickets(shape, policy):
    """Round 6: fsolve(c) / fupdate(c, g) as tasks of the launch (kinds 7, 8).  The rest of the list is unchanged, every
    block column of the launch has its fsolve, every row tile below c + 1 is in exactly one fupdate(c, .), and what
    chain_kernel waits for in those tasks has an earlier ticket."""
    R, nblk, cb, ce = shape
    table = table_of(R, nblk, cb, ce, *policy, fwd=1)
    assert [t for t in table if t[0] not in (7, 8)] == table_of(R, nblk, cb, ce, *policy, fwd=0)
    at = {}
    for n, t in enumerate(table):
        kind, i, c, k, part, k0 = t
        if kind in (7, 8):
            key = (kind, c, i if kind == 8 else 0)
            assert key not in at
            at[key] = n
    final = {}  # ticket that makes tile (i, c) of the launch final / factors block c
    for n, t in enumerate(table):
        kind, i, c, k, part, k0 = t
        if kind == 0:
            final[(i, c)] = n
        elif kind == 5:
            final[(c, c - 1)] = n
        elif kind == 1:
            final[(c, c)] = n
    for c in range(cb, ce):
        me = at[(7, c, 0)]
        deps = [final.get((c, c))]                                  # L_cc (None: factored in front of the launch)
        if c > cb:
            deps.append(at[(7, c - 1, 0)])                          # z_{c-1}
            deps.append(final.get((c, c - 1)))                      # the tile fsolve(c) applies itself
        if c - 1 > cb:
            deps.append(at[(8, c - 2, c // FWD_GROUP)])             # row c carries the columns cb .. c-2
        assert all(d is None or d < me for d in deps), (c, me, deps)
        rows = set()
        for g in range(0, (R - 1) // FWD_GROUP + 1):
            if (8, c, g) not in at:
                continue
            mine = [i for i in range(g * FWD_GROUP, min((g + 1) * FWD_GROUP, R)) if i > c + 1]
            assert mine, "a task without rows"
            rows |= set(mine)
            me = at[(8, c, g)]
            deps = [at[(7, c, 0)]] + [final[(i, c)] for i in range(c + 1, R)] + [final.get((c, c))]
            if c > cb:
                deps.append(at[(8, c - 1, g)])                      # column order per group
            assert all(d is None or d < me for d in deps), (c, g, me, deps)
        assert rows == set(range(c + 2, R))
    assert len(at) == sum(1 for t in table if t[0] in (7, 8))


def test_chain_task_rejects_bad_shapes():
    lib = _ffi.lib()
    n = C.c_int64()
    assert lib.tgp_chain_task(4, 8, 0, 8, -1, None, C.byref(n)) != 0   # more block columns than row tiles
    assert lib.tgp_chain_task(8, 8, 3, 3, -1, None, C.byref(n)) != 0   # empty column range
    assert lib.tgp_chain_task(8, 8, 0, 8, 10**6, (C.c_int32 * 5)(), None) != 0
