"""Experiment: K independent log_probability evaluations in flight on ONE GPU (one context + solver per host thread,
separate streams and matrices) -- does one evaluation's chain-bound start and tail hide behind another's updates?
usage (GPU box): python scripts/two_in_flight.py [n] [threads] [steps]"""
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from tinygp_amd import _ffi, kernels, noise, synthetic  # noqa: E402
from tinygp_amd.solvers import DirectSolver  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
T = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
stagger = float(sys.argv[4]) * 1e-3 if len(sys.argv) > 4 else 0.0  # ms between the threads' first evaluations
spec = bench.workload_spec("c2" if n == 16384 else f"n{n}")
X, y = bench.make_inputs(spec)


def kernel_at(step, who):
    u = ((step * 7 + who * 3) % 11 - 5) / 5.0
    return synthetic.config_kernel(kernels, spec["kernel"], amp=1.5 * (1 + 0.02 * u), scale=2.5 * (1 + 0.03 * u))


solvers = []
for who in range(T):
    ctx = _ffi.Ctx(device=0)
    s = DirectSolver(kernel_at(-1, who), X, noise.Diagonal(np.full(n, spec["diag"])), ctx=ctx)
    s.set_residual(y)
    solvers.append(s)
out = [[None] * steps for _ in range(T)]
go = threading.Barrier(T + 1)


def work(who):
    s = solvers[who]
    for k in range(3):
        s.factor_log_probability(None, kernel_at(k, who))
    go.wait()
    time.sleep(who * stagger)  # out of phase: one evaluation's chain-bound start / tail beside another's updates
    for k in range(steps):
        out[who][k] = s.factor_log_probability(None, kernel_at(k, who))
    go.wait()


th = [threading.Thread(target=work, args=(w,)) for w in range(T)]
for t in th:
    t.start()
go.wait()
t0 = time.perf_counter()
go.wait()
el = time.perf_counter() - t0
for t in th:
    t.join()
assert all(np.isfinite(v) for o in out for v in o) and all(s.info == 0 for s in solvers)
# same hyper-parameter point -> same value whatever ran beside it
ref = DirectSolver(kernel_at(-1, 0), X, noise.Diagonal(np.full(n, spec["diag"])), ctx=_ffi.Ctx(device=0))
ref.set_residual(y)
same = all(ref.factor_log_probability(None, kernel_at(k, w)) == out[w][k] for w in range(T) for k in range(min(steps, 4)))
print(f"n={n} in_flight={T} steps={steps} stagger={stagger * 1e3:.1f} ms: {T * steps / el:.2f} evals/s aggregate ({el / steps * 1e3:.2f} ms per round), "
      f"bit-identical to a lone evaluation: {same}")
