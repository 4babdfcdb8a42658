"""Synthetic inputs of BASELINE.json's configs: host NumPy only.

"Constant density" (about 100 points per unit length / volume) keeps the conditioning of K
independent of N, so the same noise level works from N = 1 024 to N = 131 072.
"""

from __future__ import annotations

import numpy as np

__all__ = ["make_inputs", "make_reference_inputs", "CONFIGS", "config_kernel", "kernel_text"]

SEED = 49382  # the reference benchmark's seed (docs/benchmarks.ipynb:135)

# name -> (N, D, dtype, diag, kernel spec)
CONFIGS = {
    "c1": dict(n=1024, d=1, dtype="float64", diag=0.01, kernel="expsq"),
    "c2": dict(n=16384, d=1, dtype="float64", diag=0.01, kernel="expsq"),
    "c3": dict(n=65536, d=3, dtype="float64", diag=0.01, kernel="matern52"),
    "c4": dict(n=131072, d=1, dtype="float64", diag=0.01, kernel="expsq"),
    "c5": dict(n=262144, d=1, dtype="float32", diag=0.1, kernel="sum", m_test=4096),
}


def make_inputs(n: int, d: int = 1, dtype="float64", seed: int = SEED):
    """X then the noise from ONE fresh generator: X = sort(U(0, n/100, n)) for d = 1 or
    U(0, (n/100)^(1/d), (n, d)) unsorted; y = sin(x_0) + 0.1 N(0, 1)."""
    rng = np.random.default_rng(seed)
    if d == 1:
        X = np.sort(rng.uniform(0.0, n / 100.0, size=n))
        x0 = X
    else:
        X = rng.uniform(0.0, (n / 100.0) ** (1.0 / d), size=(n, d))
        x0 = X[:, 0]
    y = np.sin(x0) + 0.1 * rng.normal(0.0, 1.0, n)
    return X.astype(dtype), y.astype(dtype)


def make_reference_inputs(n: int, seed: int = SEED):
    """The reference's own benchmark recipe (docs/benchmarks.ipynb:131-159): 100 000 sorted points
    in [0, 10], y = sin x + 0.1 N(0, 1), the first n of them; used with 1.5^2 Matern32(2.5),
    diag = 0.01 (published rows: N <= 20 000)."""
    rng = np.random.default_rng(seed)
    x = np.sort(rng.uniform(0, 10, 100_000))
    y = np.sin(x) + 0.1 * rng.normal(size=len(x))
    return x[:n].copy(), y[:n].copy()


def kernel_text(spec: str) -> str:
    """Human-readable definition of a config kernel (goes into bench.py's `config.workload`)."""
    return {
        "expsq": "1.5^2 ExpSquared(2.5)",
        "matern52": "1.5^2 Matern52(2.5, distance=L2Distance) [the reference's default L1 metric is "
                    "indefinite in 3-D: DESIGN.md 5]",
        "matern52_l1": "1.5^2 Matern52(2.5) [default L1 metric]",
        "matern32": "1.5^2 Matern32(2.5)",
        "sum": "1.5^2 ExpSquared(2.5) + 0.5^2 Matern32(1.0)",
    }[spec]


def config_kernel(kernels_module, spec: str, amp: float = 1.5, scale: float = 2.5):
    """Build the config's kernel from ANY module exposing the tinygp kernel classes
    (``tinygp_amd.kernels`` or the oracle), so product and checker share one definition."""
    k = kernels_module
    if spec == "expsq":
        return amp**2 * k.ExpSquared(scale)
    if spec == "matern52":
        # Euclidean metric: with the reference's DEFAULT (L1) metric a Matern-5/2 of a 3-D
        # distance is not positive definite (LAPACK and the HIP path both stop at pivot 17
        # on config 3's inputs; tests/test_gpu_1_gp.py pins that), so config 3 names L2.
        return amp**2 * k.Matern52(scale, distance=k.L2Distance())
    if spec == "matern52_l1":
        return amp**2 * k.Matern52(scale)
    if spec == "matern32":
        return amp**2 * k.Matern32(scale)
    if spec == "sum":
        return amp**2 * k.ExpSquared(scale) + 0.5**2 * k.Matern32(1.0)
    raise ValueError(spec)
