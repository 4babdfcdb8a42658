"""Host logic of the factorisation schedule, checked without a GPU.

``tgp_trace_factor`` replays what ``tgp_solver_factor`` / ``tgp_solver_factor_logprob`` would
enqueue on the library's streams (same code path, launches replaced by records).  The
checker below models every record's reads and writes at 128 x 128 tile granularity and
verifies with vector clocks that each pair of conflicting accesses (write-write, write-read,
read-write) is ordered by stream order or by an event record / wait pair -- i.e. that the
multi-stream schedule has no data race by construction.  (The GPU suite's determinism test is
the empirical counterpart.)
"""
import ctypes as C

import numpy as np
import pytest

from tinygp_amd import _ffi

NSTREAMS = 5
EV_G1, EV_G2 = 7, 8  # split gate: column block 1 / column blocks 2.. of the next panel
KIND = {1: "potf2", 2: "trsm", 3: "gemm", 4: "trsv_step", 5: "record", 6: "wait", 7: "assembly",
        8: "residual_copy", 9: "reductions", 10: "panel_step", 11: "chain", 12: "chain_poll", 13: "prefix_poll"}


def trace(n_pad, nb=1024, lookahead=1, first_split=5, first_small=1100, fused=1, wide_rows=0, **more):
    """fused: bit 0 = forward substitution fused into the factorisation, bit 1 = the unfused panel
    chain (potf2 | trsm | in-panel update per block -- the library's default) instead of one panel-step
    launch per block, bit 2 = the block-column update between two chains in one piece (no split gate).
    `more`: further context options by name (sub_panel=512, nb_first=256, ...)."""
    lib = _ffi.load_library()
    cap = 48 * (n_pad // 128) + 256
    out = np.zeros(cap * 10, dtype=np.int64)
    n = C.c_int64()
    opts = dict(nb_outer=nb, lookahead=lookahead, first_split=first_split, first_small_tiles=first_small,
                nb_wide_rows=wide_rows, fused_step=0 if fused & 2 else 1, gate_split=0 if fused & 4 else 1)
    opts["chain_kernel"] = 0  # (the library's default is the persistent chain: its configurations say so)
    opts.update(more)
    text = ",".join(f"{k}={v}" for k, v in opts.items())
    st = lib.tgp_trace_factor(n_pad, text.encode(), fused & 1, out.ctypes.data_as(C.POINTER(C.c_int64)), cap,
                              C.byref(n))
    assert st == 0, lib.tgp_last_error()
    return out[: n.value * 10].reshape(-1, 10).tolist()


def accesses(rec, T, part=None):
    """(reads, writes): sets of resources.  ('A', tr, tc) matrix tile, ('D', j) 16x16 inverses
    of block j, ('Y', j) 128 entries of the solved vector."""
    kind, _, *v = rec
    R, W = set(), set()

    def tile(off, ld):
        assert off >= 0 and off % 128 == 0 and (off % ld) % 128 == 0, (rec, off)
        return (off % ld) // 128, (off // ld) // 128

    if kind == 7:  # assembly of lower tiles of column tiles [tc0, tc0 + ntc)
        tc0, ntc = v[0], v[1]
        W |= {("A", tr, tc) for tc in range(tc0, tc0 + ntc) for tr in range(tc, T)}
    elif kind == 8:
        W |= {("Y", j) for j in range(T)}
    elif kind == 9:
        R |= {("A", j, j) for j in range(T)} | {("Y", j) for j in range(T)}
    elif kind == 1:
        tr, tc = tile(v[0], v[2])
        assert tr == tc
        W.add(("A", tr, tc)); R.add(("A", tr, tc)); W.add(("D", tr))
        if v[1] >= 0:
            R.add(("A",) + tile(v[1], v[2]))
    elif kind == 2:
        ld = v[3]
        lr, lc = tile(v[0], ld)
        assert lr == lc
        R |= {("A", lr, lc), ("D", lr)}
        br, bc = tile(v[1], ld)
        for i in range(v[2] // 128):
            R.add(("A", br + i, bc)); W.add(("A", br + i, bc))
    elif kind == 3:
        ld = v[7]
        (ar, ac), (brr, bcc), (cr, cc) = tile(v[0], ld), tile(v[1], ld), tile(v[2], ld)
        m, n, k = v[3] // 128, v[4] // 128, v[5] // 128
        lower, role, pfx = v[6] & 0xFF, (v[6] >> 8) & 0xFF, v[6] >> 16
        # a merged trailing update (round 6) is stepped through in two parts: its PREFIX (the first pfx tile columns, what a
        # prefix poll waits for) and the rest; `part` = None for an ordinary launch
        cols = range(n) if part is None else (range(pfx) if part == 0 else range(pfx, n))
        for tj in cols:
            for ti in range(m):
                if lower and ti < tj:
                    continue
                if role in (3, 5) and ti == 0 and tj == 0:
                    continue  # folded into the next potf2 (3) / updated and factored on the side stream (5)
                R.add(("A", cr + ti, cc + tj)); W.add(("A", cr + ti, cc + tj))
        R |= {("A", ar + ti, ac + kk) for ti in range(m) for kk in range(k)}
        R |= {("A", brr + tj, bcc + kk) for tj in cols for kk in range(k)}
    elif kind == 10:  # fused panel step: potf2 (has_p) + per row tile below: pending update, trsm
        ld, m, has_p = v[3], v[2] // 128, v[4]
        tr, tc = tile(v[0], ld)
        assert tr == tc
        R.add(("A", tr, tc)); R.add(("D", tr))
        if has_p:
            W.add(("A", tr, tc)); W.add(("D", tr))
        if v[1] >= 0:
            xr, xc = tile(v[1], ld)
            assert (xr, xc) == (tr, tc - 1)
            R.add(("A", xr, xc))
        for i in range(1, m + 1):
            R.add(("A", tr + i, tc)); W.add(("A", tr + i, tc))
            if v[1] >= 0:
                R.add(("A", tr + i, tc - 1))
    elif kind == 11:  # persistent chain: the whole launch (find_races steps through it column by column)
        for c in range(v[3], v[4]):
            r, w = chain_column_accesses(rec, c)
            R |= r; W |= w
    elif kind == 4:
        ld = v[2]
        lr, lc = tile(v[0], ld)
        assert lr == lc
        R |= {("A", lr, lc), ("D", lr)}
        R.add(("Y", lr)); W.add(("Y", lr))
        for i in range(1, v[1] // 128 + 1):
            R.add(("A", lr + i, lc)); R.add(("Y", lr + i)); W.add(("Y", lr + i))
    return R, W


def chain_column_accesses(rec, c):
    """What the persistent chain launch `rec` (kind 11: block columns [cb, ce) of the panel at (t0, t0), `rows` row
    tiles, nblk block columns) reads and writes while it makes block column c final: the column's own tiles (potf2 --
    the panel's very first block is factored by the potf2 launch in front --, the solves of the rows below) and,
    right-looking, every LATER block column of the panel (the update tasks behind column c)."""
    _, _, *v = rec
    ld, rows, cb, ce, nblk = v[1], v[2], v[3], v[4], v[5]
    off = v[0]
    assert off >= 0 and off % 128 == 0 and (off % ld) % 128 == 0
    t0, t0c = (off % ld) // 128, (off // ld) // 128
    assert t0 == t0c and 0 <= cb <= c < ce <= nblk <= rows
    R, W = set(), set()
    R.add(("A", t0 + c, t0 + c)); R.add(("D", t0 + c))
    if c > 0:
        W.add(("A", t0 + c, t0 + c)); W.add(("D", t0 + c))
    for i in range(c + 1, rows):
        R.add(("A", t0 + i, t0 + c)); W.add(("A", t0 + i, t0 + c))
    if c == cb and cb > 0:  # diag(cb) folds tile (cb, cb-1), solved by the launch before
        R.add(("A", t0 + cb, t0 + cb - 1))
    for cc in range(c + 1, nblk):
        for i in range(cc, rows):
            R.add(("A", t0 + i, t0 + cc)); W.add(("A", t0 + i, t0 + cc))
    if len(v) > 6 and v[6]:  # round 6: the forward substitution of block column c rides along as tasks of the launch
        R.add(("Y", t0 + c)); W.add(("Y", t0 + c))
        if t0 + c > 0:  # fsolve(c) applies tile (c, c-1) -- for a panel's first block the previous panel's last column
            R.add(("A", t0 + c, t0 + c - 1)); R.add(("Y", t0 + c - 1))
        for i in range(c + 1, rows):
            R.add(("Y", t0 + i)); W.add(("Y", t0 + i))
    return R, W


def find_races(recs, T, limit=5):
    clock = [[0] * NSTREAMS for _ in range(NSTREAMS)]
    snap = {}
    last_w = {}   # resource -> (stream, counter, index)
    readers = {}  # resource -> list of (stream, counter, index) since the last write
    races = []

    def ordered(prev, now_clock):
        s, c, _ = prev
        return now_clock[s] >= c

    def check_and_register(R, W, now, me, idx):
        for r in R | W:
            lw = last_w.get(r)
            if lw is not None and lw[2] != idx and not ordered(lw, now):
                races.append((r, lw[2], idx, "after write"))
        for r in W:
            for rd in readers.get(r, ()):
                if rd[2] != idx and not ordered(rd, now):
                    races.append((r, rd[2], idx, "write after read"))
        for r in W:
            last_w[r] = me
            readers[r] = []
        for r in R - W:
            readers.setdefault(r, []).append(me)

    for idx, rec in enumerate(recs):
        kind, s = rec[0], rec[1]
        assert 0 <= s < NSTREAMS, rec
        if kind == 5:
            snap[rec[2]] = list(clock[s])
            continue
        if kind == 6:
            if rec[2] in snap:
                clock[s] = [max(a, b) for a, b in zip(clock[s], snap[rec[2]])]
            continue
        if kind == 12:  # one-wave poll: this stream continues once block column c of the chain launch at `off` is final
            key = ("chain", rec[2], rec[4])
            assert key in snap, ("poll without a chain launch in front of it (host order)", rec)
            clock[s] = [max(a, b) for a, b in zip(clock[s], snap[key])]
            continue
        if kind == 13:  # one-wave poll: this stream continues once the PREFIX of the merged update at `off` is complete
            key = ("prefix", rec[2])
            assert key in snap, ("prefix poll without its update in front of it (host order)", rec)
            clock[s] = [max(a, b) for a, b in zip(clock[s], snap[key])]
            continue
        if kind == 3 and (rec[8] >> 16) > 0:  # merged trailing update: the prefix, then the rest
            for part in (0, 1):
                clock[s][s] += 1
                now = list(clock[s])
                me = (s, now[s], idx)
                R, W = accesses(rec, T, part)
                check_and_register(R, W, now, me, idx)
                if part == 0:
                    snap[("prefix", rec[4])] = list(clock[s])
            if len(races) >= limit:
                break
            continue
        if kind == 11:
            # the launch makes its block columns final one after the other; a poller waits for ONE of them: step
            # through the columns, each with its own tick and its own snapshot for the pollers
            for c in range(rec[5], rec[6]):
                clock[s][s] += 1
                now = list(clock[s])
                me = (s, now[s], idx)
                R, W = chain_column_accesses(rec, c)
                check_and_register(R, W, now, me, idx)
                snap[("chain", rec[2], c)] = list(clock[s])
            if len(races) >= limit:
                break
            continue
        clock[s][s] += 1
        now = list(clock[s])
        me = (s, now[s], idx)
        R, W = accesses(rec, T)
        check_and_register(R, W, now, me, idx)
        if len(races) >= limit:
            break
    return [(r, f"#{i} {KIND[recs[i][0]]} s{recs[i][1]}", f"#{j} {KIND[recs[j][0]]} s{recs[j][1]}", why)
            for r, i, j, why in races[:limit]]


CONFIGS = [
    # n_pad, nb, lookahead, first_split, first_small, fused
    (128, 1024, 1, 5, 1100, 1),
    (1024, 1024, 1, 5, 1100, 1),
    (1152, 1024, 1, 5, 1100, 1),
    (2560, 1024, 1, 5, 1100, 1),
    (2560, 1024, 1, 5, 1100, 0),
    (3456, 1024, 1, 5, 1100, 1),
    (5120, 1024, 1, 5, 1100, 1),
    (5120, 1024, 0, 5, 1100, 1),
    (5120, 1024, 1, 0, 1100, 1),
    (5120, 1024, 1, 7, 0, 1),
    (5120, 512, 1, 3, 1100, 1),
    (4096, 2048, 1, 5, 1100, 1),
    (16384, 1024, 1, 5, 1100, 1),
    # panels of 2 nb while at least `wide_rows` rows are left (never the first)
    (8192, 1024, 1, 5, 1100, 1, 3000),
    (5120, 512, 1, 3, 1100, 0, 1024),
    (6144, 1024, 1, 0, 0, 1, 128),
    (4224, 1024, 0, 5, 1100, 1, 2048),
    # the gate in one piece (fused bit 2)
    (5120, 1024, 1, 5, 1100, 5),
    (5120, 512, 1, 3, 1100, 4),
    (8192, 1024, 1, 0, 0, 5, 3000),
    # the unfused chain (fused bit 1)
    (2560, 1024, 1, 5, 1100, 3),
    (5120, 1024, 1, 5, 1100, 3),
    (5120, 512, 1, 3, 1100, 2),
    (5120, 1024, 0, 5, 1100, 3),
    (8192, 1024, 1, 5, 1100, 3, 3000),
    # two-level panels (sub-panels of 512 / 256 inside the outer block), with and without a narrow first panel
    (5120, 1024, 1, 5, 1100, 3, 0, dict(sub_panel=512)),
    (5120, 1024, 1, 4, 1100, 3, 0, dict(sub_panel=512)),   # the early share branches off at a sub-panel boundary
    (5120, 1024, 1, 5, 0, 2, 0, dict(sub_panel=256)),
    (3456, 1024, 0, 5, 1100, 3, 0, dict(sub_panel=512)),
    (8192, 1024, 1, 5, 1100, 3, 3000, dict(sub_panel=512, sub_panel_min_rows=4096)),
    (5120, 1024, 1, 5, 1100, 3, 0, dict(nb_first=256)),
    (5248, 1024, 1, 5, 1100, 3, 0, dict(nb_first=512, sub_panel=512)),
    (16384, 1024, 1, 5, 1100, 3, 0, dict(sub_panel=512, nb_first=768)),
    # the persistent chain: one launch per panel (two with an early share), no in-panel update launches
    (128, 1024, 1, 5, 1100, 3, 0, dict(chain_kernel=1)),
    (1152, 1024, 1, 5, 1100, 3, 0, dict(chain_kernel=1)),
    (2560, 1024, 1, 5, 1100, 3, 0, dict(chain_kernel=1)),
    (2560, 1024, 1, 5, 1100, 2, 0, dict(chain_kernel=1)),
    (5120, 1024, 1, 5, 1100, 3, 0, dict(chain_kernel=1)),
    (5120, 1024, 0, 5, 1100, 3, 0, dict(chain_kernel=1)),
    (5120, 1024, 1, 0, 1100, 3, 0, dict(chain_kernel=1)),
    (5120, 512, 1, 3, 1100, 3, 0, dict(chain_kernel=1)),
    (3456, 1024, 1, 7, 0, 3, 0, dict(chain_kernel=1)),
    (8192, 1024, 1, 5, 1100, 3, 3000, dict(chain_kernel=1)),
    (16384, 1024, 1, 5, 1100, 3, 0, dict(chain_kernel=1)),
    (5248, 1024, 1, 5, 1100, 3, 0, dict(chain_kernel=1, nb_first=512)),
    # ... panel by panel only (chain_full_rows = 0), and the whole rest in one launch from 2048 / 8192 rows
    (5120, 1024, 1, 5, 1100, 3, 0, dict(chain_kernel=1, chain_full_rows=0)),
    (5120, 1024, 0, 5, 1100, 3, 0, dict(chain_kernel=1, chain_full_rows=0)),
    (5120, 512, 1, 3, 1100, 3, 0, dict(chain_kernel=1, chain_full_rows=2048)),
    (16384, 1024, 1, 5, 1100, 3, 0, dict(chain_kernel=1, chain_full_rows=8192)),
    (16384, 1024, 1, 5, 1100, 2, 0, dict(chain_kernel=1, chain_full_rows=0)),
    # ... the next panel's first diagonal block updated and factored on the update stream beside the gate: chain_gate_split = 1
    (5120, 1024, 1, 5, 1100, 3, 0, dict(chain_kernel=1, chain_gate_split=1)),
    (16384, 1024, 1, 5, 1100, 3, 0, dict(chain_kernel=1, chain_gate_split=1)),
    (16384, 1024, 1, 5, 1100, 2, 0, dict(chain_kernel=1, chain_gate_split=1)),
    (16384, 512, 1, 5, 1100, 3, 0, dict(chain_kernel=1, chain_full_rows=0)),
    # ... round 5's followers (forward steps as launches behind pollers) are still there: chain_fwd_tasks = 0
    (5120, 1024, 1, 5, 1100, 3, 0, dict(chain_kernel=1, chain_fwd_tasks=0)),
    (16384, 1024, 1, 5, 1100, 3, 0, dict(chain_kernel=1, chain_fwd_tasks=0)),
    (5120, 1024, 0, 5, 1100, 3, 0, dict(chain_kernel=1, chain_fwd_tasks=0, chain_full_rows=0)),
    # ... and without pollers (forward steps and early shares behind the whole launch)
    (5120, 1024, 1, 5, 1100, 3, 0, dict(chain_kernel=1, chain_polls=0, chain_fwd_tasks=0)),
    (16384, 1024, 1, 5, 1100, 3, 0, dict(chain_kernel=1, chain_polls=0, chain_fwd_tasks=0)),
    (16384, 1024, 0, 5, 1100, 3, 0, dict(chain_kernel=1, chain_polls=0)),
    # ... the chain of a panel sub-panel by sub-panel, each finished sub-panel applied to the panel's remaining columns by one
    # product on the priority stream (merged schedule only: chain_sub_panel)
    (16384, 1024, 1, 5, 1100, 3, 0, dict(chain_kernel=1, chain_merged=1, chain_sub_panel=512)),
    (16384, 1024, 1, 5, 1100, 3, 0, dict(chain_kernel=1, chain_merged=1, chain_sub_panel=256, chain_sub_role=0)),
    (9216, 1024, 1, 5, 1100, 3, 0, dict(chain_kernel=1, chain_merged=1, chain_sub_panel=512, chain_sub_min_rows=6144)),
]


# round 6: the default schedule with the persistent chain is the MERGED trailing update (one launch per panel, the next
# panel's block column first, its chain behind a prefix poll); every chain configuration above is also run on round 5's
# depth-2 schedule (gate | pre | rest)
CONFIGS += [c[:7] + (dict(c[7], chain_merged=0),) for c in CONFIGS
            if len(c) > 7 and c[7].get("chain_kernel") and "chain_merged" not in c[7]]


def _cfg_id(c):
    s = f"n{c[0]}-nb{c[1]}-la{c[2]}-fs{c[3]}-st{c[4]}-f{c[5]}" + (f"-wide{c[6]}" if len(c) > 6 and c[6] else "")
    return s + ("".join(f"-{k}{v}" for k, v in c[7].items()) if len(c) > 7 else "")


@pytest.mark.parametrize("cfg", CONFIGS, ids=[_cfg_id(c) for c in CONFIGS])
def test_schedule_has_no_data_race(cfg):
    n_pad = cfg[0]
    recs = trace(*cfg[:7], **(cfg[7] if len(cfg) > 7 else {}))
    T = n_pad // 128
    # every block column is factored exactly once, in order
    def factored(r):  # block columns a record factors
        if r[0] == 1:
            return [(r[2] % r[4]) // 128]
        if r[0] == 10 and r[6] == 1:
            return [(r[2] % r[5]) // 128]
        if r[0] == 11:  # chain over [cb, ce) of a panel: every diagonal block but the panel's first
            t0 = (r[2] % r[3]) // 128
            return [t0 + c for c in range(max(r[5], 1), r[6])]
        return []

    order = [j for r in recs for j in factored(r)]
    assert order == list(range(T))
    if len(cfg) > 7 and cfg[7].get("chain_kernel"):
        assert not any(r[0] in (2, 10) for r in recs)            # no trsm / panel-step launches ...
        # ... and nothing on the update stream's GEMM queue but the next panel's first diagonal block (chain_gate_split)
        assert all(r[5] == 128 and r[6] == 128 and cfg[7].get("chain_gate_split") for r in recs if r[0] == 3 and r[1] == 3)
    elif cfg[5] & 2:
        assert not any(r[0] == 10 for r in recs)
    else:
        assert not any(r[0] == 2 for r in recs)  # every trsm of the chain rides in a panel step
    in_chain = len(cfg) > 7 and cfg[7].get("chain_kernel") and cfg[7].get("chain_fwd_tasks", 1)
    if cfg[5] & 1 and in_chain:
        # round 6: the forward substitution rides in the chain launches as tasks -- no step launch, no poller for it
        assert not any(r[0] == 4 for r in recs)
        assert all(r[8] == 1 for r in recs if r[0] == 11)
        if cfg[7].get("chain_depth2", 1) and cfg[2]:
            assert not any(r[0] == 12 for r in recs)  # (the default schedule has no early shares either)
    elif cfg[5] & 1:
        assert sum(1 for r in recs if r[0] == 4) == T  # one forward-substitution step per block
    assert find_races(recs, T) == []


def test_checker_sees_a_missing_dependency():
    """The checker is not vacuous: dropping the waits that order the far in-panel updates (two
    alternating markers) before the panel step two blocks later, or the join of the side-stream
    assembly, must be reported."""
    recs = trace(2560)
    T = 2560 // 128
    no_update_wait = [r for r in recs if not (r[0] == 6 and r[1] == 1 and r[2] == 4)]  # panel waits ev_e
    assert len(no_update_wait) < len(recs)
    assert find_races(no_update_wait, T)
    no_update_wait2 = [r for r in recs if not (r[0] == 6 and r[1] == 1 and r[2] == 6)]  # panel waits ev_f
    assert len(no_update_wait2) < len(recs)
    assert find_races(no_update_wait2, T)
    for ev, who in ((EV_G1, 1), (EV_G2, 3)):  # the panel's second step / the first far update wait for the gate pieces
        dropped = [r for r in recs if not (r[0] == 6 and r[1] == who and r[2] == ev)]
        assert len(dropped) < len(recs)
        assert find_races(dropped, T)
    no_join = [r for r in recs if not (r[0] == 6 and r[1] == 0 and r[2] == 5)]  # main waits ev_asm
    assert len(no_join) < len(recs)
    assert find_races(no_join, T)
    no_chain_wait = [r for r in recs if not (r[0] == 6 and r[1] == 0 and r[2] == 1)]  # main waits ev_b
    assert len(no_chain_wait) < len(recs)
    assert find_races(no_chain_wait, T)


def test_checker_sees_a_missing_join_of_the_split_gate():
    """chain_gate_split: the next panel's first diagonal block is updated and factored on the update stream beside the gate.
    Without the event that makes the side stream wait for the gate's inputs, or the one that joins it in front of the chain
    launch, the checker must report the race."""
    recs = trace(16384, 1024, 1, 5, 1100, 3, 0, chain_kernel=1, chain_gate_split=1, chain_merged=0)
    T = 16384 // 128
    assert any(r[0] == 3 and r[1] == 3 for r in recs) and any(r[0] == 1 and r[1] == 3 for r in recs)  # product + potf2 on stream 3
    assert find_races(recs, T) == []
    for ev, who in ((10, 3), (11, 1)):  # ev_i: side stream behind the gate's inputs; ev_j: chain launch behind the potf2
        dropped = [r for r in recs if not (r[0] == 6 and r[1] == who and r[2] == ev)]
        assert len(dropped) < len(recs)
        assert find_races(dropped, T), ev


def test_checker_sees_a_missing_prefix_poll_of_the_merged_update():
    """The merged trailing update (round 6): the chain of the next panel starts behind a one-wave poll of the update's
    PREFIX (that panel's block column) while the rest of the launch still runs.  Without the poll, or without the wait that
    keeps the next update behind that chain, the checker must report the race; the poll must follow its update in host
    order."""
    recs = trace(16384, 1024, 1, 5, 1100, 3, 0, chain_kernel=1)
    T = 16384 // 128
    merged = [r for r in recs if r[0] == 3 and (r[8] >> 16) > 0]
    polls = [r for r in recs if r[0] == 13]
    assert len(merged) == len(polls) == 11 and all((r[8] >> 8) & 0xFF == 0 and r[1] == 0 for r in merged)
    assert not any(r[0] == 3 and (r[8] >> 8) & 0xFF in (4, 5) for r in recs)  # no 64 x 64-tile launch is left
    assert find_races(recs, T) == []
    assert find_races([r for r in recs if r[0] != 13], T)                      # chain of panel p+1 beside its own columns' update
    no_chain_wait = [r for r in recs if not (r[0] == 6 and r[1] == 0 and r[2] in (1, 7))]  # main waits ev_chain (ev_b / ev_g1)
    assert len(no_chain_wait) < len(recs) and find_races(no_chain_wait, T)
    # the tail (the last panel covers every remaining column) waits for the WHOLE update: no poll for it
    last = [r for r in recs if r[0] == 3][-1]
    assert (last[8] >> 16) == 0


def test_checker_sees_a_missing_poll_of_the_persistent_chain():
    """Persistent chain: the forward-substitution step of block j and the early share of the next block-column
    update start behind a one-wave poll of block column j's count of final tiles while the launch still runs.
    Without the polls (or without the event that keeps a poller behind the zeroing of the counters, or the one that
    keeps the next launch's zeroing behind the last poller) the checker must report the race / refuse the order."""
    recs = trace(5120, 1024, 1, 5, 1100, 3, 0, chain_kernel=1, chain_full_rows=0, chain_fwd_tasks=0, chain_merged=0)
    T = 5120 // 128
    assert find_races(recs, T) == []
    polls = [r for r in recs if r[0] == 12]
    assert len(polls) >= T - 1  # one per forward-substitution step (+ the early shares on panels without a solve)
    assert find_races([r for r in recs if r[0] != 12], T)
    # every poll sits behind a wait for ev_d (recorded between the launch's memset and the kernel) on ITS stream,
    # and every chain launch but the first behind a wait for ev_f (the previous launch's last poller)
    chain_idx = [i for i, r in enumerate(recs) if r[0] == 11]
    for q, i in enumerate(chain_idx):
        before = recs[:i]
        assert any(r[0] == 5 and r[1] == recs[i][1] and r[2] == 3 for r in before[-3:]), "ev_d in front of the launch"
        if q > 0:
            assert any(r[0] == 6 and r[1] == recs[i][1] and r[2] == 6 for r in before[chain_idx[q - 1]:]), "wait for ev_f"


def test_bench_accounting_comes_from_the_launch_records():
    """bench.py's algorithmic bytes / flops per launch (roofline.algorithmic_bytes_per_launch) are summed over the
    library's own launch records for the profiled kernel (128 x 128-tile GEMM, role 0, main stream): without
    look-ahead that is one lower-triangular update per panel, whose closed form is checked here; with the default
    schedule the launch count is the one the timeline shows (14 at N = 16 384)."""
    import importlib.util
    from pathlib import Path

    spec = importlib.util.spec_from_file_location("bench", Path(__file__).resolve().parents[1] / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n_pad, nb = 8192, 1024
    total, launches, flops = bench.traced_update_bytes(dict(nb_outer=nb, lookahead=0, chain_kernel=0), n_pad, 8)
    want_b, want_f = 0, 0.0
    for k0 in range(0, n_pad - nb, nb):
        m = n_pad - k0 - nb
        entries = m * m - m * (m - 1) // 2
        want_b += 8 * (2 * entries + m * nb)
        want_f += 2.0 * entries * nb
    assert (total, launches, flops) == (want_b, n_pad // nb - 1, want_f)
    _, launches, flops = bench.traced_update_bytes(dict(chain_kernel=0), 16384, 8)  # the per-block chain's schedule
    assert launches == 14
    assert 1.0e12 < flops < 16384**3 / 3  # the rest of the N^3 / 3 runs on the 64 x 64-tile kernel (gates, in-panel)
    # round 5's depth-2 schedule (persistent chain, the last 4 096 rows ONE chain launch): the ten `rest` updates of panels
    # 0..9; gates and pre-updates have at most 1 100 tiles and run on the 64 x 64-tile kernel
    _, launches, flops = bench.traced_update_bytes(dict(chain_merged=0), 16384, 8)
    assert launches == 10
    assert 0.8e12 < flops < 16384**3 / 3  # (less than the per-block schedule: the last 4 096 rows are chain tasks)
    # library defaults (round 6): ONE merged update per panel 0..11 on the 128 x 128-tile kernel -- every flop of the
    # factorisation outside the chain launches: sum over panels of m^2 nb with m = the rows right of the panel
    _, launches, flops = bench.traced_update_bytes({}, 16384, 8)
    assert launches == 12
    want = sum(2.0 * (m * m - m * (m - 1) // 2) * 1024 for m in range(16384 - 1024, 4096 - 1, -1024))
    assert flops == want and 1.2e12 < flops < 16384**3 / 3
