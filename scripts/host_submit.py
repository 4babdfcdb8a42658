"""How far ahead of the device the submitting thread runs: per size, the wall-clock of one log_probability with fresh
hyper-parameters (assembly + factorisation + solve) next to the HOST time its enqueues took (tgp_solver_timings ms[5])."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tinygp_amd import GaussianProcess, kernels, synthetic  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096, 8192, 16384]:
    X, y = synthetic.make_inputs(n, 1, "float64")
    kern = 1.5**2 * kernels.ExpSquared(2.5)
    gp = GaussianProcess(kern, X, diag=0.01)
    gp.log_probability(y)
    s = gp.solver
    tot, sub = [], []
    for rep in range(8):
        t = time.perf_counter()
        s.factor_log_probability(np.asarray(y) - gp.loc)
        tot.append(1e3 * (time.perf_counter() - t))
        sub.append(s.timings()["host_submit_ms"])
    print(f"N = {n:6d}: wall {np.median(tot):8.3f} ms   host submission {np.median(sub):8.3f} ms", flush=True)
