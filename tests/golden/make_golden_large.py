"""Full-size oracle values that are too slow for a test run (hours of one CPU core), computed
ONCE on the host and committed as tests/golden/large.npz:

  * BASELINE config 3 (Matern52/L2 + diagonal noise, 3-D X, N = 65 536, fp64): log-likelihood and
    normalisation by LAPACK dpotrf / dtrtrs (SURVEY.md 8d "and 65 536 once");
  * BASELINE config 5's kernel (Sum(ExpSquared, Matern32)) at N = 32 768: posterior mean at 4 096
    test points in fp64 -- the value the fp32 HIP path is held to at 5e-4.

The formulas are the oracle's (oracle/tinygp_np.py, itself pinned to the reference's execution by
tests/golden/ref_*.npz); only the assembly is blocked so that the N^2 temporaries stay small.

    python tests/golden/make_golden_large.py [c3] [c5]
"""
import sys
import time
from pathlib import Path

import numpy as np
import scipy.linalg as sla
from scipy.linalg import lapack

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from oracle import tinygp_np as o  # noqa: E402
from tinygp_amd import synthetic  # noqa: E402

OUT = Path(__file__).resolve().parent / "large.npz"


def assemble(kern, X, diag, bs=1024):
    n = X.shape[0]
    K = np.empty((n, n), order="F")
    for j0 in range(0, n, bs):  # column blocks of the symmetric matrix (Fortran order: contiguous)
        K[:, j0:j0 + bs] = kern(X, X[j0:j0 + bs])
    K[np.diag_indices(n)] += diag
    return K


def factor_inplace(K, bs=4096):
    """Right-looking blocked Cholesky in place (lower), LAPACK dpotrf on the diagonal blocks,
    dtrsm / dgemm for the rest.  (One dpotrf call on the whole matrix segfaults inside the
    bundled OpenBLAS at N = 32 768 in this container.)"""
    n = K.shape[0]
    for k0 in range(0, n, bs):
        k1 = min(k0 + bs, n)
        c, info = lapack.dpotrf(np.array(K[k0:k1, k0:k1], order="F"), lower=1, clean=1)
        assert info == 0, (k0, info)
        K[k0:k1, k0:k1] = c
        if k1 < n:
            K[k1:, k0:k1] = sla.solve_triangular(c, K[k1:, k0:k1].T, lower=True, check_finite=False).T
            for j0 in range(k1, n, bs):
                j1 = min(j0 + bs, n)
                K[j0:, j0:j1] -= K[j0:, k0:k1] @ K[j0:j1, k0:k1].T
    return K


def config3():
    c = synthetic.CONFIGS["c3"]
    X, y = synthetic.make_inputs(c["n"], c["d"], c["dtype"])
    kern = synthetic.config_kernel(o, c["kernel"])
    t0 = time.time()
    L = factor_inplace(assemble(kern, X, c["diag"]))
    alpha = sla.solve_triangular(L, y, lower=True, check_finite=False)
    norm = float(np.sum(np.log(np.diag(L))) + 0.5 * c["n"] * np.log(2 * np.pi))
    logp = float(-0.5 * alpha @ alpha - norm)
    print(f"c3: logp {logp!r} norm {norm!r} ({time.time() - t0:.0f} s)", flush=True)
    return {"c3_n65536__logp": np.float64(logp), "c3_n65536__norm": np.float64(norm),
            "c3_n65536__alpha_head": alpha[:16], "c3_n65536__alpha_tail": alpha[-16:]}


def config5(n=32768, m=4096):
    c = synthetic.CONFIGS["c5"]
    X, y = synthetic.make_inputs(n, 1, "float32")  # the fp32 inputs the device sees, in fp64 arithmetic
    X, y = X.astype(np.float64), y.astype(np.float64)
    kern = synthetic.config_kernel(o, c["kernel"])
    t0 = time.time()
    L = factor_inplace(assemble(kern, X, c["diag"]))
    z = sla.solve_triangular(L, y, lower=True, check_finite=False)
    norm = float(np.sum(np.log(np.diag(L))) + 0.5 * n * np.log(2 * np.pi))
    logp = float(-0.5 * z @ z - norm)
    alpha = sla.solve_triangular(L, z, lower=True, trans=1, check_finite=False)
    xt = np.linspace(0.0, n / 100.0, m).astype(np.float32).astype(np.float64)
    mean = np.zeros(m)
    for j0 in range(0, n, 4096):
        mean += kern(xt, X[j0:j0 + 4096]) @ alpha[j0:j0 + 4096]
    print(f"c5 (N={n}): logp {logp!r} ({time.time() - t0:.0f} s)", flush=True)
    return {f"c5_n{n}__logp": np.float64(logp), f"c5_n{n}__test_loc": mean}


if __name__ == "__main__":
    which = sys.argv[1:] or ["c5", "c3"]
    res = dict(np.load(OUT)) if OUT.exists() else {}
    for w in which:
        res.update({"c3": config3, "c5": config5}[w]())
        np.savez_compressed(OUT, **res)
    print(sorted(res))
