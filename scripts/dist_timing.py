"""Wall-clock of the block-column path's resident-factor operations at world size 1 (RCCL self-collectives issued by the
library; no torch in the process): value-and-gradient, (N, R) transposed solve in ONE blocked pass, conditional variance at
M test points with the right- and the left-looking forward solve.  python scripts/dist_timing.py [N] [nb]"""
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29671")
from tinygp_amd import GaussianProcess, kernels, synthetic  # noqa: E402
from tinygp_amd.distributed import BlockCyclicCholesky  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
X, y = synthetic.make_inputs(n, 1)
k = synthetic.config_kernel(kernels, "expsq")
s = BlockCyclicCholesky(k, X, np.full(n, 0.01), nb=nb)


def timed(f, reps=3):
    f()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = f()
        t.append(time.perf_counter() - t0)
    return min(t) * 1e3, out


ms, ll = timed(lambda: s.log_probability(y))
print(f"N = {n}, nb = {nb}, world 1 ({type(s.comm).__name__}): log_probability {ms:.1f} ms")
ms, (ll2, g) = timed(lambda: s.log_probability_and_grad(y), reps=2)
print(f"  value-and-gradient (K^-1 in chunks of {s.GRAD_CHUNK} columns: fan-in forward + right-looking backward): {ms:.1f} ms")
if n <= 20000:
    ms1, (ll1, g1) = timed(lambda: GaussianProcess(k, X, diag=0.01).log_probability_and_grad(y), reps=2)
    flat = [g["kernel"][2 * i] for i, op in enumerate(k.program()) if op[0] < 16]
    print(f"  single-GPU value-and-gradient (explicit K^-1): {ms1:.1f} ms; gradients agree to "
          f"{np.max(np.abs(np.array(flat) - np.array(g1['kernel'])) / np.abs(np.array(g1['kernel']))):.1e} relative")
Y = np.random.default_rng(1).normal(size=(n, 64))
ms, xb = timed(lambda: s.solve_triangular(Y, transpose=True))
print(f"  solve_triangular(Y (N, 64), transpose=True), ONE blocked pass: {ms:.1f} ms")
ms1, x1 = timed(lambda: np.stack([s.solve_triangular(np.ascontiguousarray(Y[:, r]), transpose=True) for r in range(8)], axis=1), reps=1)
print(f"  the same column by column (round 4's form), 8 of the 64 columns: {ms1:.1f} ms -> {ms1 * 8:.0f} ms for 64; max |diff| "
      f"{np.max(np.abs(x1 - xb[:, :8])):.1e}")
xt = np.linspace(X[0], X[-1], 4096)
for mode in ("right", "left"):
    s.FORWARD = mode
    ms, v = timed(lambda: s.condition_colsumsq(xt), reps=2)
    print(f"  conditional variance at 4 096 test points, {mode}-looking forward solve: {ms:.1f} ms")
s.ops.close()
