"""One process, many tunings: median wall-clock of the fused evaluation (assembly + Cholesky + forward solve + reductions,
what bench.py times) at one size for each option set.   usage: sweep.py <n>[,<n>...] [reps] "k=v,k=v" "k=v" ... ("" = defaults)"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tinygp_amd import _ffi, kernels, noise, synthetic  # noqa: E402
from tinygp_amd.solvers import DirectSolver  # noqa: E402

sizes = [int(s) for s in sys.argv[1].split(",")]
args = sys.argv[2:]
reps = 9
if args and args[0].isdigit():
    reps, args = int(args[0]), args[1:]
ctx = _ffi.default_ctx()
for n in sizes:
    X, y = synthetic.make_inputs(n, 1, "float64")
    ks = [1.5**2 * kernels.ExpSquared(2.5), 1.4**2 * kernels.ExpSquared(2.2)]
    solver = DirectSolver(ks[0], X, noise.Diagonal(np.full(n, 0.01)), ctx=ctx)
    solver.set_residual(y)
    ref = None
    for spec in args or [""]:
        old = {}
        for kv in filter(None, spec.split(",")):
            k, v = kv.split("=")
            old[k] = ctx.set_option(k, int(v))
        try:
            for r in range(3):
                ll = solver.factor_log_probability(None, ks[r % 2])
            ts = []
            for r in range(reps):
                t0 = time.perf_counter()
                ll = solver.factor_log_probability(None, ks[0])
                ts.append(time.perf_counter() - t0)
            ts.sort()
            if ref is None:
                ref = ll
            tag = "" if ll == ref else f"  (ll differs from the first set by {abs(ll - ref) / abs(ref):.1e} rel)"
            print(f"n={n:6d} {spec or 'defaults':48s} median {ts[len(ts) // 2] * 1e3:8.3f} ms  min {ts[0] * 1e3:8.3f}{tag}", flush=True)
        finally:
            for k, v in old.items():
                ctx.set_option(k, v)
    del solver
