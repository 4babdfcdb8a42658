"""Value-and-gradient on a resident factor at N (default 16 384): wall-clock per call; run under
rocprofv3 --kernel-trace for the per-kernel split (scripts/timeline.py on the database)."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tinygp_amd import GaussianProcess, kernels, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
X, y = synthetic.make_inputs(n, 1, "float64")
kern = 1.5**2 * kernels.ExpSquared(2.5)
gp = GaussianProcess(kern, X, diag=0.01)
for i in range(reps):
    t = time.perf_counter()
    ll, g = gp.log_probability_and_grad(y)
    print(f"call {i}: {1e3 * (time.perf_counter() - t):.2f} ms  ll = {ll:.6f}", flush=True)
