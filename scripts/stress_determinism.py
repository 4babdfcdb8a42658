"""Race detector for the multi-stream factorisation schedule: the fused evaluation is
deterministic (fixed reduction orders, no atomics in the arithmetic), so repeated evaluations
of the same (kernel, X, y) must return bit-identical log-likelihoods and factors.  A missing
stream dependency shows up as an occasional different bit pattern."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tinygp_amd import kernels, noise, synthetic  # noqa: E402
from tinygp_amd.solvers import DirectSolver  # noqa: E402

bad = 0
for n, reps in ((16384, 150), (5000, 300), (4096, 600), (1100, 600)):
    X, y = synthetic.make_inputs(n, 1, "float64")
    ks = [1.5**2 * kernels.ExpSquared(2.5), 1.4**2 * kernels.ExpSquared(2.2)]
    solver = DirectSolver(ks[0], X, noise.Diagonal(np.full(n, 0.01)))
    solver.set_residual(y)
    ref = [solver.factor_log_probability(None, k) for k in ks]
    t0 = time.perf_counter()
    diffs = 0
    for r in range(reps):
        k = r % 2
        ll = solver.factor_log_probability(None, ks[k])
        if ll != ref[k]:
            diffs += 1
            if diffs <= 5:
                print(f"  N={n} rep {r}: {ll!r} != {ref[k]!r} (rel {abs(ll-ref[k])/abs(ref[k]):.2e})")
    dt = time.perf_counter() - t0
    # the factor itself, twice
    solver.refactor(ks[0]); L1 = np.array(solver.scale_tril)
    solver.refactor(ks[0]); L2 = np.array(solver.scale_tril)
    same = bool(np.array_equal(L1, L2))
    print(f"N={n}: {reps} evaluations in {dt:.2f} s, {diffs} differing log-likelihoods, factor bit-identical: {same}", flush=True)
    bad += diffs + (0 if same else 1)
print("STRESS", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
