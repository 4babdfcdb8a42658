"""Resident-factor log_probability (ONE streaming forward substitution + reductions) by workgroups per block row.

  python scripts/trsv_bench.py [N ...]            (ctx option trsv_groups = 2, 3, 4, 6, 8)

bytes read = s N (N + 1) / 2; HBM roofline 8 TB/s (bench.py's `roofline_secondary`)."""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tinygp_amd import _ffi, kernels, noise, synthetic  # noqa: E402
from tinygp_amd.solvers import DirectSolver  # noqa: E402

ctx = _ffi.default_ctx()
for n in [int(a) for a in sys.argv[1:]] or [16384]:
    X, y = synthetic.make_inputs(n, 1, "float64")
    k = 1.5**2 * kernels.ExpSquared(2.5)
    solver = DirectSolver(k, X, noise.Diagonal(np.full(n, 0.01)))
    solver.set_residual(y)
    solver.refactor()
    out = C.c_double()
    ref = None
    for g in (2, 3, 4, 6, 8, 4, 2):
        ctx.set_option("trsv_groups", g)
        for _ in range(3):
            _ffi.check(_ffi.lib().tgp_solver_logprob(solver._handle, None, C.byref(out)), "logprob")
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            _ffi.check(_ffi.lib().tgp_solver_logprob(solver._handle, None, C.byref(out)), "logprob")
        dt = (time.perf_counter() - t0) / reps
        ref = out.value if ref is None else ref
        nbytes = 8 * n * (n + 1) / 2
        print(f"N = {n:6d}  groups {g}:  {dt * 1e3:7.3f} ms  {nbytes / dt / 1e12:5.2f} TB/s = {nbytes / dt / 8e12:.3f} of 8 TB/s   "
              f"ll {out.value:.10f}  rel. diff to groups 2: {abs(out.value - ref) / abs(ref):.1e}", flush=True)
    solver.close()
