"""Hunts a rare wrong evaluation: repeats the fused evaluation at one size and, on the first result that differs
from the reference, reports info, where the factor differs (128-tile coordinates) and whether it holds NaNs.

usage: stress_nan.py <n> <reps> [key=value ...]   (context options)"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tinygp_amd import _ffi, kernels, noise, synthetic  # noqa: E402
from tinygp_amd.solvers import DirectSolver  # noqa: E402

n, reps = int(sys.argv[1]), int(sys.argv[2])
ctx = _ffi.default_ctx()
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
X, y = synthetic.make_inputs(n, 1, "float64")
ks = [1.5**2 * kernels.ExpSquared(2.5), 1.4**2 * kernels.ExpSquared(2.2)]
solver = DirectSolver(ks[0], X, noise.Diagonal(np.full(n, 0.01)))
solver.set_residual(y)
ref, Lref = [], []
for k in ks:
    ref.append(solver.factor_log_probability(None, k))
    Lref.append(np.array(solver.scale_tril))
bad = 0
for r in range(reps):
    k = r % 2
    ll = solver.factor_log_probability(None, ks[k])
    if ll != ref[k]:
        bad += 1
        L = np.array(solver.scale_tril)
        d = (L != Lref[k]) & ~(np.isnan(L) & np.isnan(Lref[k]))
        rows, cols = np.nonzero(d)
        msg = f"rep {r}: ll={ll!r} ref={ref[k]!r} info={solver._info} factor: {d.sum()} differing entries, {np.isnan(L).sum()} NaNs"
        if d.any():
            msg += (f"; tiles rows {rows.min() // 128}..{rows.max() // 128} cols {cols.min() // 128}..{cols.max() // 128};"
                    f" first col {cols.min()} (rows {rows[cols == cols.min()].min()}..{rows[cols == cols.min()].max()})")
            c0 = cols.min()
            rr = rows[cols == c0]
            msg += f"; max |diff| in first col {np.nanmax(np.abs(L[rr, c0] - Lref[k][rr, c0])):.3e}"
        print(msg, flush=True)
        if bad >= 12:
            break
print(f"n={n} reps={reps} options={sys.argv[3:]}: {bad} wrong evaluations", flush=True)
