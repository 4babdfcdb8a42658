"""Wall-clock of the paths around log_probability (resident solver, warm): gradient,
conditional mean, conditional covariance.  One line per path -> stdout."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tinygp_amd import GaussianProcess, kernels, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
m = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
X, y = synthetic.make_inputs(n, 1, "float64")
Xt = np.linspace(0, n / 100, m)
kern = 1.5**2 * kernels.ExpSquared(2.5)


def timed(label, fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    dt = (time.perf_counter() - t0) / reps
    print(f"{label:58s} {dt*1e3:9.2f} ms", flush=True)
    return out


gp = GaussianProcess(kern, X, diag=0.01)
timed(f"GaussianProcess(...) build + factor, N={n}", lambda: GaussianProcess(kern, X, diag=0.01))
timed("log_probability (factor resident: solve + reduce only)", lambda: gp.log_probability(y))
timed("log_probability_and_grad (3 kernel params, noise, mean)", lambda: gp.log_probability_and_grad(y))
timed(f"predict mean at M={m}", lambda: gp.predict(y, Xt))
timed(f"predict mean + variance at M={m}", lambda: gp.predict(y, Xt, return_var=True))
timed(f"condition(...).gp.covariance (M x M) at M={m}", lambda: gp.condition(y, Xt).gp.covariance)
