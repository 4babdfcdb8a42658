"""Regenerates tests/golden/*.npz from the NumPy/SciPy oracle.

    python tests/golden/make_golden.py

The reference (dfm/tinygp) cannot be imported in the build container (needs jax/equinox),
so these vectors are produced by ``oracle/tinygp_np.py`` -- the line-by-line restatement of
the reference path -- on the seeds and shapes of the reference's own tests
(tests/_cases.py cites them).  They pin the oracle against regressions and travel to the
GPU box, where the HIP path is checked against them without importing anything else.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import _cases  # noqa: E402
from oracle import tinygp_np as o  # noqa: E402

OUT = Path(__file__).resolve().parent


def kernels_golden():
    x1, x2 = _cases.data_kernels()
    xs, _, ts = _cases.data_solver()
    res = {}
    for name, k in _cases.kernel_zoo(o).items():
        res[f"{name}__5d"] = k(x1, x2)
        res[f"{name}__1d"] = k(xs, ts)
        res[f"{name}__diag"] = k(x1)
    np.savez_compressed(OUT / "kernels.npz", **res)


def gp_golden():
    res = {}
    for name, (gp, y, t) in _cases.gp_cases(o, o.GaussianProcess).items():
        res[f"{name}__logp"] = np.float64(gp.log_probability(y))
        res[f"{name}__norm"] = np.float64(gp.solver.normalization())
        res[f"{name}__var"] = gp.variance
        c0 = gp.condition(y)
        res[f"{name}__self_loc"] = c0.gp.loc
        res[f"{name}__self_var"] = c0.gp.variance
        c1 = gp.condition(y, t)
        res[f"{name}__test_logp"] = np.float64(c1.log_probability)
        res[f"{name}__test_loc"] = c1.gp.loc
        res[f"{name}__test_var"] = c1.gp.variance
        res[f"{name}__test_cov"] = c1.gp.covariance
    np.savez_compressed(OUT / "gp.npz", **res)


def config_golden():
    """BASELINE.json config 1 (N=1024) in full, N=4096 / reference-recipe N=2000 as scalars."""
    res = {}
    for n in (1024, 4096):
        X, y = _cases.synthetic.make_inputs(n, 1)
        gp = o.GaussianProcess(_cases.synthetic.config_kernel(o, "expsq"), X, diag=0.01)
        alpha = gp.solver.solve_triangular(y)
        res[f"expsq_n{n}__logp"] = np.float64(gp.log_probability(y))
        res[f"expsq_n{n}__norm"] = np.float64(gp.solver.normalization())
        res[f"expsq_n{n}__alpha_head"] = alpha[:16]
        res[f"expsq_n{n}__alpha_tail"] = alpha[-16:]
        res[f"expsq_n{n}__Ldiag_head"] = np.diag(gp.solver.scale_tril)[:16]
        res[f"expsq_n{n}__Ldiag_tail"] = np.diag(gp.solver.scale_tril)[-16:]
        if n == 1024:
            xt = np.linspace(0, n / 100, 64)
            c = gp.condition(y, xt)
            res["expsq_n1024__test_loc"] = c.gp.loc
            res["expsq_n1024__test_var"] = c.gp.variance
    X3, y3 = _cases.synthetic.make_inputs(2048, 3)
    gp = o.GaussianProcess(_cases.synthetic.config_kernel(o, "matern52"), X3, diag=0.01)
    res["m52_3d_n2048__logp"] = np.float64(gp.log_probability(y3))
    xb, yb = _cases.data_benchmark(2000)
    gp = o.GaussianProcess(_cases.kernel_zoo(o)["bench_m32"], xb, diag=0.01)
    res["bench_m32_n2000__logp"] = np.float64(gp.log_probability(yb))
    np.savez_compressed(OUT / "configs.npz", **res)


if __name__ == "__main__":
    kernels_golden()
    gp_golden()
    config_golden()
    for f in sorted(OUT.glob("*.npz")):
        print(f.name, f.stat().st_size, "bytes")
