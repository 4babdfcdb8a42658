"""Where a block of the persistent panel chain spends its time: the device-side time stamps of every tile task
(ctx option chain_stamps = 1, tgp_chain_stamps) of one factorisation, printed as the diagonal chain's timeline
(microseconds from the first task's start) and the bulk tiles' phase averages.

  python scripts/chain_timeline.py [N ...] [option=value ...] [brief]

`brief`: the diagonal chain's period every 4th block and the per-kind totals only (long launches).

xsolve task stamps: 0 start | 1 tile (c,c-1) in registers | 3 X_{c,c-1} solved and stored (column block by column block behind
                    the step flags of potf2(c-1)) | 4 X published
diag task stamps:   0 start | 3 fold of X_{c,c-1} accumulated (16 columns at a time behind xsolve(c)'s counter) |
                    5 tile (c,c) ready, potf2 starts | 6 factored | 7 L_cc published
solve task stamps:  0 start | 1 tile and L_cc ready | 2 staged | 3 solved | 4 published
update task stamps: 0 start | 1 operands and tile ready | 2 computed, stores issued | 3 published"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tinygp_amd import _ffi, kernels, noise, synthetic  # noqa: E402
from tinygp_amd.solvers import DirectSolver  # noqa: E402

ctx = _ffi.default_ctx()
ctx.set_option("chain_kernel", 1)
brief = "brief" in sys.argv[1:]
for a in sys.argv[1:]:
    if "=" in a:
        for kv in a.split(","):
            key, val = kv.split("=")
            ctx.set_option(key, int(val))
        print("option", a)
for n in [int(a) for a in sys.argv[1:] if a.isdigit()] or [1024, 4096]:
    X, y = synthetic.make_inputs(n, 1, "float64")
    k = 1.5**2 * kernels.ExpSquared(2.5)
    solver = DirectSolver(k, X, noise.Diagonal(np.full(n, 0.01)))
    solver.set_residual(y)
    for _ in range(3):
        solver.factor_log_probability(None, k)
    ctx.set_option("chain_stamps", 1)
    solver.factor_log_probability(None, k)
    ctx.set_option("chain_stamps", 0)
    cap = 32768
    out = np.zeros(cap * 16, dtype=np.int64)
    cnt = C.c_int64()
    _ffi.check(_ffi.lib().tgp_chain_stamps(ctx.handle, out.ctypes.data_as(C.POINTER(C.c_int64)), cap, C.byref(cnt)),
               "tgp_chain_stamps")
    recs = out[: cnt.value * 16].reshape(-1, 16)
    t0 = recs[:, 4].min()
    us = lambda v: (v - t0) / 100.0  # noqa: E731  (100 MHz counter)
    print(f"== N = {n}: {cnt.value} tasks in {recs[:, 3].max() + 1} launches")
    diag = recs[recs[:, 0] == 1]
    diag = diag[np.lexsort((diag[:, 2], diag[:, 3] & 255))]
    prev_pub = None
    xs = {int(r[2]): r for r in recs[recs[:, 0] == 5]}
    print("launch col | xsolve: start   tile streamed   Xpub | diag: start  folded potf2in factored published | "
          "prev.published->potf2  potf2  period")
    for nd, r in enumerate(diag):
        s_ = [us(v) if v else float("nan") for v in r[4:14]]
        x = xs.get(int(r[2]))
        x_ = [us(v) if v else float("nan") for v in x[4:14]] if x is not None else [float("nan")] * 10
        per = s_[7] - prev_pub if prev_pub is not None else float("nan")
        gap = s_[5] - prev_pub if prev_pub is not None else float("nan")
        if not brief or nd % 4 == 0:
          print(f"{r[3] & 255:4d} {r[2]:4d}   | {x_[0]:13.1f} {x_[1]:6.1f} {x_[3]:8.1f} {x_[4]:6.1f} | {s_[0]:11.1f} {s_[3]:7.1f} "
              f"{s_[5]:7.1f} {s_[6]:8.1f} {s_[7]:9.1f} | {gap:12.1f} {s_[6] - s_[5]:14.1f}  {per:7.1f}")
        prev_pub = s_[7]
    for kind, name, ph in ((0, "solve", ("wait", "stage", "solve", "publish")), (2, "update", ("wait", "compute", "publish")),
                           (4, "update of tile (k+2, k+1), one of CHAIN_CRIT_PARTS parts per task", ("wait", "compute", "publish")),
                           (3, "update (diagonal tile)", ("wait", "compute", "publish")),
                           (6, "BATCHED update (block columns first .. last in one product)", ("wait", "compute", "publish"))):
        sel = recs[recs[:, 0] == kind]
        if not len(sel):
            continue
        dur = [(sel[:, 5 + q] - sel[:, 4 + q]) / 100.0 for q in range(len(ph))]
        print(f"{name}: {len(sel)} tasks, CU time in ms: " + "  ".join(f"{p_} {x.sum() / 1e3:8.2f}" for p_, x in zip(ph, dur)) +
              "   mean us: " + "  ".join(f"{x.mean():6.1f}" for x in dur))
        if brief:
            continue
        print(f"{name} tasks: phase durations in us (mean over tasks), by block column" +
              (" updated FROM" if kind >= 2 else ""))
        col = (sel[:, 3] >> 8) & 255 if kind >= 2 else sel[:, 2]
        for c in sorted(set(col.tolist())):
            m = sel[col == c]
            d = [(m[:, 5 + q] - m[:, 4 + q]) / 100.0 for q in range(len(ph))]
            print(f"  column {c}: {len(m):5d} tasks  " + "  ".join(f"{p} {x.mean():6.1f}" for p, x in zip(ph, d)) +
                  f"   first start {us(m[:, 4]).min():7.1f}  last published at {us(m[:, 4 + len(ph)]).max():7.1f}")
    last = np.where(recs[:, 4:16] > 0, recs[:, 4:16], 0).max(axis=1)
    span = (last.max() - t0) / 100.0
    busy = ((last - recs[:, 4]) / 100.0).sum()
    print(f"launches span {span:.0f} us; task time (start -> last stamp) summed {busy / 1e3:.2f} ms = {busy / span / 256:.2f} of 256 CUs")
    solver.close()
