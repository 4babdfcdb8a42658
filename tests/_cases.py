"""Shared definitions of the parity cases: seeds, shapes and kernels lifted from the
reference's own tests (file:line cited per case) plus BASELINE.json's config 1.

Every builder takes a module exposing the tinygp kernel classes (the oracle
``oracle.tinygp_np`` or the product ``tinygp_amd.kernels``) so both sides see identical
definitions.
"""
import numpy as np

from tinygp_amd import synthetic


def kernel_zoo(k):
    """name -> kernel.  Stationary family at scale 1.5 (test_george_compat.py:57-84), the
    x0.3 product (:174-179), the Sum/Product pairs of test_kernels.py:62-69 and the
    test_solver.py:27-57 set."""
    return {
        "exp": k.Exp(1.5),
        "expsq": k.ExpSquared(1.5),
        "matern32": k.Matern32(1.5),
        "matern52": k.Matern52(1.5),
        "cosine": k.Cosine(2.3),
        "expsine2": k.ExpSineSquared(scale=2.3, gamma=1.3),
        "ratquad": k.RationalQuadratic(alpha=1.5),
        "expsq_x0.3": k.ExpSquared(1.5) * 0.3,
        "const": k.Constant(1.5) + 0.0 * k.Exp(1.0),
        "sum_ops": 1.5 * k.Matern32(2.5) + 0.9 * k.ExpSineSquared(scale=1.5, gamma=0.3),
        "prod_ops": (1.5 * k.Matern32(2.5)) * (0.9 * k.ExpSineSquared(scale=1.5, gamma=0.3)),
        "solver_m32": 1.8**2 * k.Matern32(1.5),
        "solver_m52": 1.8**2 * k.Matern52(1.5),
        "solver_exp": 1.8**2 * k.Exp(1.5),
        "solver_cos": 1.8**2 * k.Cosine(1.5),
        "solver_sum": 1.8**2 * k.Matern32(1.5) + 0.9**2 * k.Matern52(0.7),
        "l2_m32": k.Matern32(1.5, distance=k.L2Distance()),
        "l1_expsq": k.ExpSquared(1.5, distance=k.L1Distance()),
        "bench_m32": 1.5**2 * k.Matern32(2.5),
    }


def data_kernels():
    """test_kernels.py:13-20 / test_gp.py:14-21: 5-D uniform clouds, seed 1058390."""
    rng = np.random.default_rng(1058390)
    x1 = rng.uniform(-3, 3, (50, 5))
    x2 = rng.uniform(-5, 5, (50, 5))
    return x1, x2


def data_solver():
    """test_solver.py:16-24: x = sort(U(-3,3,50)), y = sin x, t = sort(U(-3,3,10))."""
    rng = np.random.default_rng(84930)
    x = np.sort(rng.uniform(-3, 3, 50))
    y = np.sin(x)
    t = np.sort(rng.uniform(-3, 3, 10))
    return x, y, t


def data_george(ndim=1):
    """test_george_compat.py:105-108: per-point noise in U(0.1, 0.2)."""
    rng = np.random.default_rng(1058390)
    x = np.sort(rng.uniform(0, 10, (50, ndim)), axis=0)
    t = np.sort(rng.uniform(0, 10, (12, ndim)), axis=0)
    y = np.sin(x[:, 0])
    diag = rng.uniform(0.1, 0.2, 50)
    return x, y, t, diag


def data_benchmark(n):
    """docs/benchmarks.ipynb:131-159: x in [0,10], Matern32, diag 0.01."""
    rng = np.random.default_rng(49382)
    x = np.sort(rng.uniform(0, 10, 100_000))
    y = np.sin(x) + 0.1 * rng.normal(size=len(x))
    return x[:n].copy(), y[:n].copy()


def data_config(name):
    c = synthetic.CONFIGS[name]
    X, y = synthetic.make_inputs(c["n"], c["d"], c["dtype"])
    return X, y, c


def gp_cases(mod, gp_cls):
    """name -> (gp, y, t) for the GP-level parity cases.  `mod` holds the kernel classes,
    `gp_cls` the GaussianProcess class of the same implementation."""
    out = {}
    x, y, t = data_solver()
    zoo = kernel_zoo(mod)
    for name in ["solver_m32", "solver_m52", "solver_exp", "solver_cos", "solver_sum"]:
        out[name] = (gp_cls(zoo[name], x, diag=0.1), y, t)
    for nd in (1, 3):
        xg, yg, tg, dg = data_george(nd)
        for name in ["exp", "expsq", "matern32", "matern52", "ratquad"]:
            out[f"george{nd}d_{name}"] = (gp_cls(zoo[name], xg, diag=dg), yg, tg)
    x5, _ = data_kernels()
    y5 = np.sin(x5[:, 0]) + 0.3 * np.cos(x5[:, 1])
    out["cloud5d_m32"] = (gp_cls(zoo["matern32"], x5, diag=0.01, mean=0.25), y5, x5[:7] + 0.1)
    return out


def data_transforms():
    """3-D cloud for the transform / dense-noise / matmul cases (shapes like test_kernels.py:13-20, its own seed)."""
    rng = np.random.default_rng(1058397)
    X = rng.uniform(-3, 3, (40, 3))
    T = rng.uniform(-3, 3, (9, 3))
    y = np.sin(X[:, 0]) + 0.2 * X[:, 1] - 0.1 * X[:, 2] ** 2
    B = rng.normal(size=(40, 40))
    dense = 0.05 * (B @ B.T) / 40 + 0.05 * np.eye(40)  # an SPD noise matrix (test_noise.py:52-66 uses a dense SPD block)
    V = rng.normal(size=(9, 4))
    return X, T, y, dense, V


def transform_cases(mod):
    """name -> kernel with an input transform, built from ANY module tree exposing `kernels` and `transforms`
    (the reference package under oracle/refshim, or tinygp_amd): reference transforms.py:39-162, the kernels of
    tests/test_transforms.py:10-49 plus anisotropic / full-matrix variants."""
    k, t = mod.kernels, mod.transforms
    chol = np.array([[1.5, 0.0, 0.0], [0.4, 0.9, 0.0], [-0.3, 0.2, 2.1]])
    mat = np.array([[0.5, 0.1, 0.0], [0.0, 1.3, -0.2]])  # 3-D -> 2-D
    return {
        "linear_scalar": t.Linear(1 / 4.5, k.Matern32()),
        "linear_vector": 1.3 * t.Linear(np.array([0.5, 2.0, 1.3]), k.ExpSquared()),
        "linear_matrix": t.Linear(mat, k.Matern52(distance=k.L2Distance())),
        "cholesky_scalar": t.Cholesky(4.5, k.Matern32()),
        "cholesky_vector": t.Cholesky(np.array([4.5, 0.8, 2.0]), k.ExpSquared()) + 0.2 * k.Exp(1.5),
        "cholesky_matrix": 0.7 * t.Cholesky(chol, k.ExpSquared()),
        "subspace_1": t.Subspace(1, k.Matern32(0.8)),
        "subspace_02": t.Subspace(np.array([0, 2]), k.RationalQuadratic(alpha=1.5)),
    }
