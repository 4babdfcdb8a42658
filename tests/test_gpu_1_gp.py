"""GPU parity, API layer: tinygp_amd.GaussianProcess / DirectSolver (HIP through the C ABI)
against the oracle and the committed golden vectors, written the way the reference's own
tests read (tests/test_gp.py, test_kernels.py, test_solver.py, test_george_compat.py).

Tolerances: log-likelihood 1e-8 relative (BASELINE.json north_star); posterior mean /
variance / covariance rtol = atol = 5e-7, the reference's fp64 tolerance
(src/tinygp/test_utils.py:16); fp32 5e-4 (:15).
"""
import os

import numpy as np
import pytest

import _cases
from oracle import tinygp_np as o
from tinygp_amd import GaussianProcess, kernels, means, noise
from tinygp_amd.solvers import DirectSolver

pytestmark = pytest.mark.gpu

LL_RTOL = 1e-8
TOL = dict(rtol=5e-7, atol=5e-7)


def _loaded_native():
    """The parity below must have run in native code: the in-tree .so is mapped."""
    with open("/proc/self/maps") as f:
        return any("libtgp_hip.so" in line for line in f)


@pytest.mark.parametrize("name", sorted(_cases.gp_cases(o, o.GaussianProcess)))
def test_gp_cases_match_oracle_and_golden(name, golden_dir):
    g = np.load(golden_dir / "gp.npz")
    gp, y, t = _cases.gp_cases(kernels, GaussianProcess)[name]
    ref, _, _ = _cases.gp_cases(o, o.GaussianProcess)[name]
    assert isinstance(gp.solver, DirectSolver) and gp.solver.info == 0

    np.testing.assert_allclose(gp.log_probability(y), ref.log_probability(y), rtol=LL_RTOL)
    np.testing.assert_allclose(gp.log_probability(y), g[f"{name}__logp"], rtol=LL_RTOL)
    np.testing.assert_allclose(gp.solver.normalization(), g[f"{name}__norm"], rtol=LL_RTOL)
    np.testing.assert_allclose(gp.variance, g[f"{name}__var"], **TOL)
    np.testing.assert_allclose(gp.covariance, ref.covariance, **TOL)
    np.testing.assert_allclose(gp.solver.scale_tril, ref.solver.scale_tril, rtol=1e-9, atol=1e-9)

    c0 = gp.condition(y)  # at the data (gp.py:342-346 fast path)
    np.testing.assert_allclose(c0.log_probability, g[f"{name}__logp"], rtol=LL_RTOL)
    np.testing.assert_allclose(c0.gp.loc, g[f"{name}__self_loc"], **TOL)
    np.testing.assert_allclose(c0.gp.variance, g[f"{name}__self_var"], **TOL)

    c1 = gp.condition(y, t)  # at test points
    np.testing.assert_allclose(c1.log_probability, g[f"{name}__test_logp"], rtol=LL_RTOL)
    np.testing.assert_allclose(c1.gp.loc, g[f"{name}__test_loc"], **TOL)
    np.testing.assert_allclose(c1.gp.variance, g[f"{name}__test_var"], **TOL)
    np.testing.assert_allclose(c1.gp.covariance, g[f"{name}__test_cov"], **TOL)
    assert _loaded_native()


@pytest.mark.parametrize("name", sorted(_cases.gp_cases(o, o.GaussianProcess)))
def test_gp_cases_match_reference_golden(name, golden_dir):
    """Against the reference's OWN outputs (tests/golden/ref_gp.npz: /root/reference's classes
    run unmodified under oracle/refshim), including the conditioned process used as a GP of
    its own: ``means.Conditioned`` at new points (means.py:58-86) and a second conditioning
    (gp.py:380-385)."""
    r = np.load(golden_dir / "ref_gp.npz")
    gp, y, t = _cases.gp_cases(kernels, GaussianProcess)[name]
    np.testing.assert_allclose(gp.log_probability(y), r[f"{name}__logp"], rtol=LL_RTOL)
    np.testing.assert_allclose(gp.solver.normalization(), r[f"{name}__norm"], rtol=LL_RTOL)
    np.testing.assert_allclose(gp.variance, r[f"{name}__var"], **TOL)
    c0 = gp.condition(y)
    np.testing.assert_allclose(c0.gp.loc, r[f"{name}__self_loc"], **TOL)
    np.testing.assert_allclose(c0.gp.variance, r[f"{name}__self_var"], **TOL)
    c1 = gp.condition(y, t)
    np.testing.assert_allclose(c1.log_probability, r[f"{name}__test_logp"], rtol=LL_RTOL)
    np.testing.assert_allclose(c1.gp.loc, r[f"{name}__test_loc"], **TOL)
    np.testing.assert_allclose(c1.gp.variance, r[f"{name}__test_var"], **TOL)
    np.testing.assert_allclose(c1.gp.covariance, r[f"{name}__test_cov"], **TOL)
    np.testing.assert_allclose(gp.predict(y, t, include_mean=False), r[f"{name}__predict_nomean"], **TOL)
    tn = t[:5] + 0.05
    assert isinstance(c1.gp.mean_function, means.Conditioned)
    np.testing.assert_allclose([c1.gp.mean_function(x) for x in tn], r[f"{name}__cmean_new"], **TOL)
    np.testing.assert_allclose(c1.gp.mean_function.batch(tn), r[f"{name}__cmean_new"], **TOL)
    y2 = np.asarray(c1.gp.loc) + 0.1 * np.cos(np.arange(len(t)))
    c2 = c1.gp.condition(y2, tn)  # conditioning a conditioned GP: host-evaluated Conditioned kernel
    np.testing.assert_allclose(c2.log_probability, r[f"{name}__recond_logp"], rtol=1e-6)
    np.testing.assert_allclose(c2.gp.loc, r[f"{name}__recond_loc"], **TOL)
    np.testing.assert_allclose(c2.gp.variance, r[f"{name}__recond_var"], **TOL)
    assert np.all(np.isfinite(c2.gp.covariance)) and _loaded_native()


def test_reference_golden_configs(golden_dir):
    """BASELINE config 1 and the mid-size scalars, from the reference's own code."""
    r = np.load(golden_dir / "ref_configs.npz")
    syn = _cases.synthetic
    for n in (1024, 4096):
        X, y = syn.make_inputs(n, 1)
        gp = GaussianProcess(syn.config_kernel(kernels, "expsq"), X, diag=0.01)
        np.testing.assert_allclose(gp.log_probability(y), r[f"expsq_n{n}__logp"], rtol=LL_RTOL)
        np.testing.assert_allclose(gp.solver.normalization(), r[f"expsq_n{n}__norm"], rtol=LL_RTOL)
        alpha = gp.solver.solve_triangular(y)
        np.testing.assert_allclose(alpha[:16], r[f"expsq_n{n}__alpha_head"], rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(alpha[-16:], r[f"expsq_n{n}__alpha_tail"], rtol=1e-6, atol=1e-8)
    cnd = GaussianProcess(syn.config_kernel(kernels, "expsq"), *syn.make_inputs(1024, 1)[:1],
                          diag=0.01).condition(syn.make_inputs(1024, 1)[1], np.linspace(0, 10.24, 64))
    np.testing.assert_allclose(cnd.gp.loc, r["expsq_n1024__test_loc"], **TOL)
    np.testing.assert_allclose(cnd.gp.variance, r["expsq_n1024__test_var"], **TOL)
    X3, y3 = syn.make_inputs(2048, 3)
    gp = GaussianProcess(syn.config_kernel(kernels, "matern52"), X3, diag=0.01)
    np.testing.assert_allclose(gp.log_probability(y3), r["m52_3d_n2048__logp"], rtol=LL_RTOL)
    xb, yb = _cases.data_benchmark(2000)
    gp = GaussianProcess(_cases.kernel_zoo(kernels)["bench_m32"], xb, diag=0.01)
    np.testing.assert_allclose(gp.log_probability(yb), r["bench_m32_n2000__logp"], rtol=LL_RTOL)
    X5, y5 = syn.make_inputs(1024, 1)  # config 5's kernel: posterior mean at test points
    gp = GaussianProcess(syn.config_kernel(kernels, "sum"), X5, diag=0.1)
    np.testing.assert_allclose(gp.log_probability(y5), r["sum_n1024__logp"], rtol=LL_RTOL)
    np.testing.assert_allclose(gp.predict(y5, np.linspace(0, 10.24, 128)), r["sum_n1024__test_loc"], **TOL)


def test_predict_variants_like_george_compat():
    # test_george_compat.py:122-154
    x, y, t, diag = _cases.data_george(1)
    k, ko = kernels.Matern32(1.5), o.Matern32(1.5)
    gp, ref = GaussianProcess(k, x, diag=diag), o.GaussianProcess(ko, x, diag=diag)
    np.testing.assert_allclose(gp.predict(y), ref.predict(y), **TOL)
    np.testing.assert_allclose(gp.predict(y, x), ref.predict(y, x), **TOL)
    np.testing.assert_allclose(gp.predict(y, t), ref.predict(y, t), **TOL)
    for kw in (dict(return_var=True), dict(return_cov=True)):
        a, b = gp.predict(y, **kw), ref.predict(y, **kw)
        np.testing.assert_allclose(a[0], b[0], **TOL), np.testing.assert_allclose(a[1], b[1], **TOL)
        a, b = gp.predict(y, t, **kw), ref.predict(y, t, **kw)
        np.testing.assert_allclose(a[0], b[0], **TOL), np.testing.assert_allclose(a[1], b[1], **TOL)


def test_condition_with_other_kernel():
    # test_solver.py:91-103: condition(y, kernel=kernel0) and condition(y, X_test=t, kernel=kernel0)
    x, y, t = _cases.data_solver()
    k, ko = 1.8**2 * kernels.Matern52(1.5), 1.8**2 * o.Matern52(1.5)
    k0, k0o = 3.8**2 * kernels.Matern32(4.5), 3.8**2 * o.Matern32(4.5)
    gp, ref = GaussianProcess(k, x, diag=0.1), o.GaussianProcess(ko, x, diag=0.1)
    for kwargs, kwo in ((dict(kernel=k0), dict(kernel=k0o)),
                        (dict(X_test=t, kernel=k0), dict(X_test=t, kernel=k0o)),
                        (dict(X_test=t, include_mean=False), dict(X_test=t, include_mean=False))):
        a, b = gp.condition(y, **kwargs), ref.condition(y, **kwo)
        np.testing.assert_allclose(a.log_probability, b.log_probability, rtol=LL_RTOL)
        np.testing.assert_allclose(a.gp.loc, b.gp.loc, **TOL)
        np.testing.assert_allclose(a.gp.variance, b.gp.variance, **TOL)
        np.testing.assert_allclose(a.gp.covariance, b.gp.covariance, **TOL)


def test_conditioned_kernel_like_test_kernels():
    # test_kernels.py:72-83 against np.linalg.solve
    x1, x2 = _cases.data_kernels()
    k1 = 1.5 * kernels.Matern32(2.5)
    k2 = 0.9 * kernels.ExpSineSquared(scale=1.5, gamma=0.3)
    K = k1(x1, x1) + 0.1 * np.eye(x1.shape[0])
    solver = DirectSolver.init(k1, x1, noise.Diagonal(np.full(x1.shape[0], 0.1)))
    cond = kernels.Conditioned(x1, solver, k2)
    np.testing.assert_allclose(cond(x1, x2), k2(x1, x2) - k2(x1, x1) @ np.linalg.solve(K, k2(x1, x2)),
                               **TOL)


def test_means_and_scalar_y_like_test_gp():
    # test_gp.py:14-21,41-51: y = rng.normal(len(X)) is a SCALAR with loc=50 that broadcasts
    rng = np.random.default_rng(1058390)
    X = rng.uniform(-3, 3, (50, 5))
    y = rng.normal(len(X))
    gp1 = GaussianProcess(kernels.Matern32(1.5), X, diag=0.01, mean=lambda x: 0.0)
    gp2 = GaussianProcess(kernels.Matern32(1.5), X, diag=0.01, mean=0.0)
    gp3 = GaussianProcess(kernels.Matern32(1.5), X, diag=0.01)
    np.testing.assert_allclose(gp1.mean, gp2.mean), np.testing.assert_allclose(gp1.mean, gp3.mean)
    np.testing.assert_allclose(gp1.log_probability(y), gp2.log_probability(y), **TOL)
    np.testing.assert_allclose(gp1.log_probability(y), gp3.log_probability(y), **TOL)
    ref = o.GaussianProcess(o.Matern32(1.5), X, diag=0.01)
    np.testing.assert_allclose(gp3.log_probability(y), ref.log_probability(y), rtol=LL_RTOL)
    gsum = GaussianProcess(kernels.Matern32(1.5), X, diag=0.01, mean=np.sum)
    np.testing.assert_allclose(gsum.loc, X.sum(axis=1))


def test_condition_shape_error_like_test_gp():
    # test_gp.py:54-76 (array branch)
    rng = np.random.default_rng(1058390)
    X = rng.uniform(-3, 3, (50, 5))
    y = rng.normal(size=50)
    gp = GaussianProcess(kernels.ExpSquared(distance=kernels.L2Distance()), X, diag=0.1)
    gp.condition(y, X[0][None])
    with pytest.raises(ValueError):
        gp.condition(y, X[0])
    with pytest.raises(ValueError):
        GaussianProcess(kernels.Exp(1.0), X, mean=lambda x: np.ones(2))  # gp.py:91-94


def test_solver_protocol_pieces():
    x, y, t = _cases.data_solver()
    k, ko = 1.8**2 * kernels.Matern32(1.5), 1.8**2 * o.Matern32(1.5)
    s = DirectSolver(k, x, noise.Diagonal(np.full(50, 0.1)))
    r = o.DirectSolver(ko, x, o.Diagonal(np.full(50, 0.1)))
    Y = np.random.default_rng(5).normal(size=(50, 7))
    for tr in (False, True):
        np.testing.assert_allclose(s.solve_triangular(y, transpose=tr), r.solve_triangular(y, transpose=tr), **TOL)
        np.testing.assert_allclose(s.solve_triangular(Y, transpose=tr), r.solve_triangular(Y, transpose=tr), **TOL)
    np.testing.assert_allclose(s.dot_triangular(y), r.dot_triangular(y), **TOL)
    np.testing.assert_allclose(s.dot_triangular(Y), r.dot_triangular(Y), **TOL)
    np.testing.assert_allclose(s.dot_triangular(Y.reshape(50, 7, 1)), r.dot_triangular(Y.reshape(50, 7, 1)), **TOL)
    nz = noise.Diagonal(np.full(10, 0.02))
    np.testing.assert_allclose(s.condition(k, t, nz), r.condition(ko, t, o.Diagonal(np.full(10, 0.02))), **TOL)
    np.testing.assert_allclose(s.condition(k, None, noise.Diagonal(np.full(50, 0.02))),
                               r.condition(ko, None, o.Diagonal(np.full(50, 0.02))), **TOL)


def test_covariance_argument_and_dense_noise():
    x, y, _ = _cases.data_solver()
    k, ko = kernels.Matern52(1.1), o.Matern52(1.1)
    rng = np.random.default_rng(2)
    R = rng.normal(size=(50, 50)) * 0.05
    M = R @ R.T + 0.1 * np.eye(50)
    gp = GaussianProcess(k, x, noise=noise.Dense(M))
    ref = o.GaussianProcess(ko, x, noise=o.Dense(M))
    np.testing.assert_allclose(gp.log_probability(y), ref.log_probability(y), rtol=LL_RTOL)
    np.testing.assert_allclose(gp.covariance, ref.covariance, **TOL)
    cov = ko(x, x) + 0.2 * np.eye(50)
    gp = GaussianProcess(k, x, diag=0.2, covariance_value=cov)
    np.testing.assert_allclose(gp.log_probability(y), o.GaussianProcess(ko, x, diag=0.2).log_probability(y),
                               rtol=LL_RTOL)


def test_refactor_keeps_dense_noise_and_drops_stale_covariance():
    """a new kernel must neither drop the off-diagonal noise (noise.Dense lives
    only in the host covariance) nor factor the previous kernel's matrix."""
    x, y, _ = _cases.data_solver()
    rng = np.random.default_rng(2)
    R = rng.normal(size=(50, 50)) * 0.05
    M = R @ R.T + 0.1 * np.eye(50)
    gp = GaussianProcess(kernels.Matern52(1.1), x, noise=noise.Dense(M))
    for k, ko in ((1.7 * kernels.Matern32(0.8), 1.7 * o.Matern32(0.8)),
                  (kernels.ExpSquared(0.6), o.ExpSquared(0.6))):
        gp.solver.refactor(k)
        want = o.GaussianProcess(ko, x, noise=o.Dense(M))
        np.testing.assert_allclose(gp.solver.log_probability(y), want.log_probability(y), rtol=LL_RTOL)
        np.testing.assert_allclose(gp.solver.covariance(), want.covariance, **TOL)
        np.testing.assert_allclose(gp.solver.factor_log_probability(y, k), want.log_probability(y),
                                   rtol=LL_RTOL)
    # user-supplied covariance with diagonal noise: a new (device) kernel re-assembles
    cov = o.Matern52(1.1)(x, x) + 0.2 * np.eye(50)
    gp = GaussianProcess(kernels.Matern52(1.1), x, diag=0.2, covariance_value=cov)
    gp.solver.refactor(kernels.Matern32(0.9))
    np.testing.assert_allclose(gp.solver.log_probability(y),
                               o.GaussianProcess(o.Matern32(0.9), x, diag=0.2).log_probability(y),
                               rtol=LL_RTOL)
    # ... and a host-evaluated kernel rebuilds the host matrix instead of keeping the old one
    gp.solver.refactor(kernels.Custom(lambda a, b: np.exp(-0.5 * np.sum(np.square(a - b)) / 0.49)))
    np.testing.assert_allclose(gp.solver.log_probability(y),
                               o.GaussianProcess(o.ExpSquared(0.7), x, diag=0.2).log_probability(y),
                               rtol=LL_RTOL)


def test_host_evaluated_kernels_factor_on_the_device():
    """Custom / DotProduct / Polynomial / a user subclass that only overrides evaluate()
    (reference kernels/base.py:38-57,156-256): the matrix comes from Python, the Cholesky,
    solves and conditioning from the HIP solver."""
    x1, _ = _cases.data_kernels()
    y = np.sin(x1[:, 0])
    t = x1[:9] + 0.2

    class MyKernel(kernels.Kernel):
        def evaluate(self, X1, X2):
            return np.exp(-np.sum(np.abs(X1 - X2)) / 1.5)

    for k, ko in ((MyKernel(), o.Exp(1.5)),
                  (kernels.Custom(lambda a, b: np.exp(-np.sum(np.abs(a - b)) / 1.5)), o.Exp(1.5)),
                  (kernels.Custom(lambda a, b: np.exp(-np.sum(np.abs(a - b)) / 1.5)) * 2.0
                   + kernels.Matern32(0.7), 2.0 * o.Exp(1.5) + o.Matern32(0.7))):
        gp, ref = GaussianProcess(k, x1, diag=0.05), o.GaussianProcess(ko, x1, diag=0.05)
        np.testing.assert_allclose(gp.log_probability(y), ref.log_probability(y), rtol=LL_RTOL)
        c, cr = gp.condition(y, t), ref.condition(y, t)
        np.testing.assert_allclose(c.gp.loc, cr.gp.loc, **TOL)
        np.testing.assert_allclose(c.gp.variance, cr.gp.variance, **TOL)
        np.testing.assert_allclose(c.gp.covariance, cr.gp.covariance, **TOL)
    K = x1 @ x1.T
    gp = GaussianProcess(kernels.DotProduct(), x1, diag=0.3)
    sign, logdet = np.linalg.slogdet(K + 0.3 * np.eye(50))
    want = -0.5 * y @ np.linalg.solve(K + 0.3 * np.eye(50), y) - 0.5 * logdet - 25 * np.log(2 * np.pi)
    np.testing.assert_allclose(gp.log_probability(y), want, rtol=LL_RTOL)
    Kp = ((x1 / 2.0) @ (x1 / 2.0).T + 0.5**2) ** 2
    gp = GaussianProcess(kernels.Polynomial(order=2, scale=2.0, sigma=0.5), x1, diag=0.3)
    sign, logdet = np.linalg.slogdet(Kp + 0.3 * np.eye(50))
    want = -0.5 * y @ np.linalg.solve(Kp + 0.3 * np.eye(50), y) - 0.5 * logdet - 25 * np.log(2 * np.pi)
    np.testing.assert_allclose(gp.log_probability(y), want, rtol=LL_RTOL)
    assert _loaded_native()


def test_inputs_beyond_the_device_evaluator_still_factor_on_the_device():
    """Drop-in edges (reference kernels/distance.py:41-59, kernels/base.py:84-103 have no such limits): D = 20 > 16
    input dimensions, and a kernel tree of 41 ops > 32.  The kernel MATRIX comes from the host route (for the
    long tree: sub-trees that fit run on the device, the combination on the host), the factorisation, the solves
    and the conditioning run on the device through covariance= -- results at the usual tolerances."""
    rng = np.random.default_rng(12)
    n = 300
    X = rng.uniform(-2, 2, (n, 20))
    y = np.sin(X[:, 0]) + 0.1 * rng.normal(size=n)
    Xt = rng.uniform(-2, 2, (17, 20))
    k, ko = 1.3 * kernels.Matern52(6.0) + 0.5 * kernels.ExpSquared(9.0), 1.3 * o.Matern52(6.0) + 0.5 * o.ExpSquared(9.0)
    gp, ref = GaussianProcess(k, X, diag=0.05), o.GaussianProcess(ko, X, diag=0.05)
    assert gp.solver._prog is None and _loaded_native()
    np.testing.assert_allclose(gp.log_probability(y), ref.log_probability(y), rtol=LL_RTOL)
    c, r = gp.condition(y, Xt), ref.condition(y, Xt)
    np.testing.assert_allclose(c.gp.loc, r.gp.loc, **TOL)
    np.testing.assert_allclose(c.gp.variance, r.gp.variance, **TOL)
    np.testing.assert_allclose(c.gp.covariance, r.gp.covariance, **TOL)
    # the optimiser step with new hyper-parameters: the host matrix is rebuilt, never a stale one
    k2, k2o = 1.1 * kernels.Matern52(5.0) + 0.5 * kernels.ExpSquared(9.0), 1.1 * o.Matern52(5.0) + 0.5 * o.ExpSquared(9.0)
    np.testing.assert_allclose(gp.solver.factor_log_probability(y, k2),
                               o.GaussianProcess(k2o, X, diag=0.05).log_probability(y), rtol=LL_RTOL)
    # 41 ops in 2-D
    x2 = rng.uniform(-2, 2, (200, 2))
    y2 = np.sin(x2[:, 0])
    kl, klo = kernels.Exp(1.0), o.Exp(1.0)
    for i in range(20):
        kl, klo = kl + (0.1 + 0.01 * i) * kernels.Matern32(1.0 + 0.1 * i), klo + (0.1 + 0.01 * i) * o.Matern32(1.0 + 0.1 * i)
    gpl, refl = GaussianProcess(kl, x2, diag=0.1), o.GaussianProcess(klo, x2, diag=0.1)
    np.testing.assert_allclose(gpl.log_probability(y2), refl.log_probability(y2), rtol=LL_RTOL)
    np.testing.assert_allclose(gpl.predict(y2, x2[:9] + 0.05), refl.predict(y2, x2[:9] + 0.05), **TOL)


def test_pytree_input_with_a_custom_kernel():
    """Reference gp.py:64-112, 176-191: X as a pytree (time, band label) with a `Custom` kernel -- the tutorials'
    multi-band pattern.  Kernel matrices on the host, factorisation / solves / conditioning on the device; checked
    against the textbook formulas in NumPy; X_test with another tree structure raises like the reference."""
    rng = np.random.default_rng(3)
    n = 120
    t, band = np.sort(rng.uniform(0, 6, n)), rng.integers(0, 2, n)
    y = np.sin(t) + 0.3 * band + 0.05 * rng.normal(size=n)
    tt, tb = np.linspace(0, 6, 11), np.ones(11, dtype=band.dtype)
    f = lambda a, b: 1.2 * np.exp(-0.5 * (a[0] - b[0]) ** 2 / 0.8**2) * (1.0 if a[1] == b[1] else 0.6)  # noqa: E731

    def dense(t1, b1, t2, b2):
        return 1.2 * np.exp(-0.5 * (t1[:, None] - t2[None]) ** 2 / 0.8**2) * np.where(b1[:, None] == b2[None], 1.0, 0.6)

    gp = GaussianProcess(kernels.Custom(f), (t, band), diag=0.01)
    assert gp.num_data == n and _loaded_native()
    K = dense(t, band, t, band) + 0.01 * np.eye(n)
    L = np.linalg.cholesky(K)
    alpha = np.linalg.solve(K, y)
    want_ll = -0.5 * y @ alpha - np.sum(np.log(np.diag(L))) - 0.5 * n * np.log(2 * np.pi)
    np.testing.assert_allclose(gp.log_probability(y), want_ll, rtol=LL_RTOL)
    c = gp.condition(y, (tt, tb))
    Ks = dense(t, band, tt, tb)
    np.testing.assert_allclose(c.gp.loc, Ks.T @ alpha, **TOL)
    A = np.linalg.solve(L, Ks)
    np.testing.assert_allclose(c.gp.variance, np.diag(dense(tt, tb, tt, tb)) - np.sum(A * A, axis=0)
                               + np.sqrt(np.finfo(np.float64).eps), **TOL)
    with pytest.raises(ValueError):
        gp.condition(y, (tt, tb, tb))
    with pytest.raises(ValueError):
        gp.condition(y, tt)


def test_numerical_failure_never_raises():
    # gp.py:316: non-finite log-likelihood -> -inf; the factor holds NaNs like jax's cholesky
    x = np.linspace(0, 1, 200)
    gp = GaussianProcess(kernels.ExpSquared(5.0), x, diag=-0.5)
    assert gp.solver.info > 0
    assert gp.log_probability(np.sin(x)) == -np.inf
    assert np.all(np.isnan(gp.solver.solve_triangular(np.sin(x))))
    assert np.isnan(gp.solver.normalization())
    assert o.GaussianProcess(o.ExpSquared(5.0), x, diag=-0.5).log_probability(np.sin(x)) == -np.inf


def test_default_jitter_and_fp32():
    x, y, t = _cases.data_solver()
    gp = GaussianProcess(kernels.Matern32(1.5), x)  # default diag = sqrt(eps) (gp.py:388-393)
    ref = o.GaussianProcess(o.Matern32(1.5), x)
    np.testing.assert_allclose(gp.noise.diagonal(), np.sqrt(np.finfo(np.float64).eps))
    np.testing.assert_allclose(gp.log_probability(y), ref.log_probability(y), rtol=1e-7)
    x32, y32, t32 = x.astype(np.float32), y.astype(np.float32), t.astype(np.float32)
    gp32 = GaussianProcess(1.8**2 * kernels.Matern32(1.5), x32, diag=np.float32(0.1))
    ref64 = o.GaussianProcess(1.8**2 * o.Matern32(1.5), x, diag=0.1)
    assert gp32.dtype == np.float32 and gp32.log_probability(y32).dtype == np.float32
    np.testing.assert_allclose(gp32.log_probability(y32), ref64.log_probability(y), rtol=5e-4)
    c = gp32.condition(y32, t32)
    assert c.gp.loc.dtype == np.float32
    np.testing.assert_allclose(c.gp.loc, ref64.condition(y, t).gp.loc, rtol=5e-4, atol=5e-4)
    np.testing.assert_allclose(c.gp.variance, ref64.condition(y, t).gp.variance, rtol=5e-4, atol=5e-4)


def test_ill_conditioned_default_jitter_n4096():
    """Worst-case conditioning the API allows by default: ExpSquared(2.5) on 4 096 points at
    100 points per unit length with the DEFAULT jitter sqrt(eps) (gp.py:388-393), cond(K) ~ 1e11,
    smallest pivot 1.3e-4.  The factorisation's pivot path uses Newton-refined rcp / rsq instead
    of IEEE division (chol.hip: fast_rcp / fast_rsqrt); this bounds it against LAPACK.  Two
    LAPACK-based factorisations with different blockings differ by 2e-8 (log-likelihood) and
    7e-8 (pivots) on this matrix, so the bars are 50x that; the backward error
    |K - L L^T| / |K| is conditioning-independent and must stay at the rounding level."""
    n = 4096
    X, y = _cases.synthetic.make_inputs(n, 1)
    gp = GaussianProcess(kernels.ExpSquared(2.5), X)
    ref = o.GaussianProcess(o.ExpSquared(2.5), X)
    assert gp.solver.info == 0
    L, Lr = gp.solver.scale_tril, ref.solver.scale_tril
    K = ref.covariance
    assert np.linalg.norm(K - L @ L.T) / np.linalg.norm(K) < 1e-14
    np.testing.assert_allclose(np.diag(L), np.diag(Lr), rtol=5e-6)
    assert np.min(np.diag(L)) > 1e-4
    np.testing.assert_allclose(gp.log_probability(y), ref.log_probability(y), rtol=1e-6)
    z = np.random.default_rng(0).normal(size=n)
    back = L @ gp.solver.solve_triangular(z)  # L (L^-1 z) = z to the solve's backward error
    assert np.linalg.norm(back - z) / np.linalg.norm(z) < 1e-10


def test_refactor_is_the_optimizer_step():
    X, y = _cases.synthetic.make_inputs(1024, 1)
    gp = GaussianProcess(1.5**2 * kernels.ExpSquared(2.5), X, diag=0.01)
    for amp, scale in ((1.2, 2.0), (0.8, 3.1)):
        gp.solver.refactor(amp**2 * kernels.ExpSquared(scale))
        want = o.GaussianProcess(amp**2 * o.ExpSquared(scale), X, diag=0.01).log_probability(y)
        np.testing.assert_allclose(gp.solver.log_probability(y), want, rtol=LL_RTOL)


def test_a_timed_out_hand_off_is_released_and_the_pass_repeated_on_the_per_block_path():
    """With `chain_polls = 3` the followers of a chain launch wait behind hipStreamWaitValue32 -- the runtime's own wait
    kernel, which has no timeout (the default poll kernel is bounded on the device by wall clock).  The
    host joins such a pass with a deadline; past it, it satisfies the pending waits itself, drains the streams and
    reports TGP_E_TIMEOUT, and the solver repeats the pass ONCE on the launch-per-block path.  The test hook
    `fault_inject = 1` lets the deadline pass while the device is still at work: the result must be the reference's,
    the retry must be counted, and the next evaluation must run on the default path again."""
    from tinygp_amd import _ffi

    ctx = _ffi.default_ctx()
    if ctx.get_option("chain_kernel") != 1:
        pytest.skip("the persistent chain is switched off")
    keep_polls = ctx.set_option("chain_polls", 3)  # followers behind stream wait-values: the mode that needs the host deadline
    keep_fwd = ctx.set_option("chain_fwd_tasks", 0)  # (round 6's default has no followers at all: the forward steps are chain tasks)
    n = 4096
    X, y = _cases.synthetic.make_inputs(n, 1)
    k = 1.5**2 * kernels.ExpSquared(2.5)
    want = float(o.GaussianProcess(1.5**2 * o.ExpSquared(2.5), X, diag=0.01).log_probability(y))
    gp = GaussianProcess(k, X, diag=0.01)
    before = ctx.get_option("timeout_retries")
    ctx.set_option("fault_inject", 1)
    try:
        got = float(gp.log_probability(y))
    finally:
        ctx.set_option("fault_inject", 0)
        ctx.set_option("chain_polls", keep_polls)
        ctx.set_option("chain_fwd_tasks", keep_fwd)
    np.testing.assert_allclose(got, want, rtol=LL_RTOL)
    assert ctx.get_option("timeout_retries") == before + 1
    assert ctx.get_option("chain_kernel") == 1
    gp.solver.refactor(1.2**2 * kernels.ExpSquared(2.0))
    want2 = float(o.GaussianProcess(1.2**2 * o.ExpSquared(2.0), X, diag=0.01).log_probability(y))
    np.testing.assert_allclose(gp.solver.log_probability(y), want2, rtol=LL_RTOL)
    assert ctx.get_option("timeout_retries") == before + 1


@pytest.mark.parametrize("env", [dict(AMD_SERIALIZE_KERNEL="3"), dict(TGP_SERIALIZED_KERNELS="1"), dict(TGP_HIP_OPTIONS="chain_merged=0")],
                         ids=["AMD_SERIALIZE_KERNEL=3", "serialising-tool-defaults", "depth-2-schedule"])
def test_kernel_serialising_environments_give_the_same_bits_without_a_timeout(env):
    """The default schedule hangs two kinds of work on device-side polls: the tasks of a chain launch (inside ONE launch)
    and -- round 6 -- the next chain behind the PREFIX of the merged trailing update (a one-wave poll kernel on the priority
    stream).  A runtime that serialises kernels (AMD_SERIALIZE_KERNEL=3: every launch completes before the next is
    submitted) or a tool that runs them one at a time in an order of its own (a context created under such a tool defaults
    to chain_polls = 0: the next chain behind the WHOLE update) must still give the same bits, and no wait may run into
    poll_timeout_ms.  Also: round 5's depth-2 schedule against the merged one (the same updates cut differently: 1e-12)."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    code = ("import time, numpy as np\n"
            "from tinygp_amd import kernels, noise, synthetic, _ffi\n"
            "from tinygp_amd.solvers import DirectSolver\n"
            "X, y = synthetic.make_inputs(8192, 1)\n"
            "s = DirectSolver(1.5**2 * kernels.ExpSquared(2.5), X, noise.Diagonal(np.full(8192, 0.01)))\n"
            "s.set_residual(y)\n"
            "s.factor_log_probability(None, 1.5**2 * kernels.ExpSquared(2.5))\n"
            "t0 = time.perf_counter()\n"
            "v = s.factor_log_probability(None, 1.4**2 * kernels.ExpSquared(2.2))\n"
            "print('RESULT', repr(float(v)), time.perf_counter() - t0, _ffi.default_ctx().get_option('timeout_retries'))\n")
    out = {}
    for name, e in (("plain", {}), ("env", env)):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(root), timeout=600,
                           env=dict(os.environ, PYTHONPATH=str(root), **e))
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")]
        assert r.returncode == 0 and line, r.stdout[-1000:] + r.stderr[-2000:]
        _, val, secs, retries = line[0].split()
        out[name] = (val, float(secs), int(retries))
    if "TGP_HIP_OPTIONS" in env:  # another cut of the same updates (64 x 64 tiles for two of three): equal to rounding
        np.testing.assert_allclose(float(out["env"][0]), float(out["plain"][0]), rtol=1e-12)
    else:
        assert out["env"][0] == out["plain"][0]      # the same launches in another order of execution: bit-identical
    assert out["env"][2] == 0 and out["env"][1] < 1.0  # no hand-off timed out (poll_timeout_ms is 4 s), no retry


def test_poll_timeout_option_round_trips_and_rejects_nonsense():
    from tinygp_amd import _ffi

    ctx = _ffi.default_ctx()
    old = ctx.set_option("poll_timeout_ms", 2500)
    try:
        assert ctx.get_option("poll_timeout_ms") == 2500
        with pytest.raises(ValueError):
            ctx.set_option("poll_timeout_ms", 0)
        # ONE value per process and device (it lives in a device global of the library): a second context reads and
        # sets the same one, and the first context's get_option follows (advisor r5: a copy per context went stale)
        other = _ffi.Ctx(device=0)
        try:
            assert other.get_option("poll_timeout_ms") == 2500
            other.set_option("poll_timeout_ms", 3100)
            assert ctx.get_option("poll_timeout_ms") == 3100
        finally:
            other.close() if hasattr(other, "close") else None
    finally:
        ctx.set_option("poll_timeout_ms", old)


# ---- BASELINE.json configs -----------------------------------------------------------------
def test_config1_n1024(golden_dir):
    g = np.load(golden_dir / "configs.npz")
    X, y, c = _cases.data_config("c1")
    gp = GaussianProcess(_cases.synthetic.config_kernel(kernels, c["kernel"]), X, diag=c["diag"])
    np.testing.assert_allclose(gp.log_probability(y), g["expsq_n1024__logp"], rtol=LL_RTOL)
    np.testing.assert_allclose(gp.solver.normalization(), g["expsq_n1024__norm"], rtol=LL_RTOL)
    alpha = gp.solver.solve_triangular(y)
    np.testing.assert_allclose(alpha[:16], g["expsq_n1024__alpha_head"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(alpha[-16:], g["expsq_n1024__alpha_tail"], rtol=1e-7, atol=1e-9)
    Ld = np.diag(gp.solver.scale_tril)
    np.testing.assert_allclose(Ld[:16], g["expsq_n1024__Ldiag_head"], rtol=1e-10)
    np.testing.assert_allclose(Ld[-16:], g["expsq_n1024__Ldiag_tail"], rtol=1e-9)
    cnd = gp.condition(y, np.linspace(0, 10.24, 64))
    np.testing.assert_allclose(cnd.gp.loc, g["expsq_n1024__test_loc"], **TOL)
    np.testing.assert_allclose(cnd.gp.variance, g["expsq_n1024__test_var"], **TOL)


def test_mid_sizes_against_golden(golden_dir):
    g = np.load(golden_dir / "configs.npz")
    X, y = _cases.synthetic.make_inputs(4096, 1)
    gp = GaussianProcess(_cases.synthetic.config_kernel(kernels, "expsq"), X, diag=0.01)
    np.testing.assert_allclose(gp.log_probability(y), g["expsq_n4096__logp"], rtol=LL_RTOL)
    X3, y3 = _cases.synthetic.make_inputs(2048, 3)
    gp = GaussianProcess(_cases.synthetic.config_kernel(kernels, "matern52"), X3, diag=0.01)
    np.testing.assert_allclose(gp.log_probability(y3), g["m52_3d_n2048__logp"], rtol=LL_RTOL)
    xb, yb = _cases.data_benchmark(2000)  # docs/benchmarks.ipynb recipe, non-multiple of 128
    gp = GaussianProcess(_cases.kernel_zoo(kernels)["bench_m32"], xb, diag=0.01)
    np.testing.assert_allclose(gp.log_probability(yb), g["bench_m32_n2000__logp"], rtol=LL_RTOL)


@pytest.mark.parametrize("n", [1100, 2500, 3333, 5000])
def test_ragged_multi_panel_sizes_against_oracle(n):
    """Sizes that are multiples of neither the 128 tile nor the 1024 outer panel: several outer
    panels with a short last one (look-ahead, early block-column share, side-stream assembly
    and the fused forward substitution all take their edge branches)."""
    X, y = _cases.synthetic.make_inputs(n, 1)
    gp = GaussianProcess(_cases.synthetic.config_kernel(kernels, "expsq"), X, diag=0.01)
    ref = o.GaussianProcess(_cases.synthetic.config_kernel(o, "expsq"), X, diag=0.01)
    np.testing.assert_allclose(gp.log_probability(y), ref.log_probability(y), rtol=LL_RTOL)
    t = np.linspace(X[0], X[-1], 37)
    np.testing.assert_allclose(gp.predict(y, t), ref.predict(y, t), **TOL)
    # the optimiser-step entry point (fused assembly + factorisation + solve) on the same solver
    k2 = 1.3**2 * kernels.ExpSquared(2.0)
    ll2 = gp.solver.factor_log_probability(y, k2)
    ref2 = o.GaussianProcess(1.3**2 * o.ExpSquared(2.0), X, diag=0.01).log_probability(y)
    np.testing.assert_allclose(ll2, ref2, rtol=LL_RTOL)
    X32 = X.astype(np.float32)
    gp32 = GaussianProcess(_cases.synthetic.config_kernel(kernels, "expsq"), X32, diag=np.float32(0.1))
    ref32 = o.GaussianProcess(_cases.synthetic.config_kernel(o, "expsq"), X32.astype(np.float64), diag=0.1)
    np.testing.assert_allclose(gp32.log_probability(y.astype(np.float32)), ref32.log_probability(y),
                               rtol=5e-4)


@pytest.mark.parametrize("fwd_tasks", [1, 0], ids=["forward-steps-as-chain-tasks", "forward-steps-as-followers"])
@pytest.mark.parametrize("n,reps", [(5000, 40), (4096, 40), (16384, 6)])
def test_multi_stream_schedule_is_deterministic(n, reps, fwd_tasks):
    """Race detector: the factorisation runs on five streams; its arithmetic has fixed
    reduction orders and no atomics, so repeated fused evaluations of the same inputs must
    be BIT-identical.  A missing stream dependency shows up as an occasional different bit
    pattern (scripts/stress_determinism.py is the long version).  Round 6: with the forward substitution as tasks of
    the chain launches (the default: per-group column order, tests/test_chain_tasks.py) and as round 5's followers."""
    from tinygp_amd import _ffi

    ctx = _ffi.default_ctx()
    keep = ctx.set_option("chain_fwd_tasks", fwd_tasks)
    try:
        _deterministic_evaluations(n, reps)
    finally:
        ctx.set_option("chain_fwd_tasks", keep)


@pytest.mark.parametrize("n", [100, 128, 1000, 4096, 5000, 9000])
def test_reductions_left_by_the_chain_match_the_reduction_kernels(n):
    """Round 6: the fused evaluation's two sums (z.z and sum log L_ii) come from the chain launches' forward-substitution
    tasks -- per-block partial sums in a fixed tree, added up by the task of the matrix's last block -- instead of two
    reduction launches (ctx option chain_reduce).  Both orders are fixed; they differ in rounding only, and both match the
    oracle (reference gp.py:313-320)."""
    from tinygp_amd import _ffi

    ctx = _ffi.default_ctx()
    X, y = _cases.synthetic.make_inputs(n, 1)
    k = 1.5**2 * kernels.ExpSquared(2.5)
    got = {}
    for mode in (1, 0):
        keep = ctx.set_option("chain_reduce", mode)
        try:
            solver = DirectSolver(k, X, noise.Diagonal(np.full(n, 0.01)))
            solver.set_residual(y)
            got[mode] = [float(solver.factor_log_probability(None, k)) for _ in range(3)]
            assert got[mode][0] == got[mode][1] == got[mode][2]
            # ... and the resident factor's own value (separate solve + reductions)
            np.testing.assert_allclose(float(solver.log_probability(y)), got[mode][0], rtol=1e-12)
        finally:
            ctx.set_option("chain_reduce", keep)
    np.testing.assert_allclose(got[1][0], got[0][0], rtol=1e-13)
    want = float(o.GaussianProcess(1.5**2 * o.ExpSquared(2.5), X, diag=0.01).log_probability(y))
    np.testing.assert_allclose(got[1][0], want, rtol=LL_RTOL)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n", [100, 128, 129, 1000, 4096, 9000])
def test_resident_residual_is_solved_in_place_with_the_same_bits(n, dtype):
    """`tgp_solver_logprob` with the RESIDENT residual (NULL host pointer; what bench.py's secondary roofline times) lets the
    streaming solve read it where it is -- no copy into the work vector, none inside the solve's preparation -- and gives the
    bits of the same call with the residual uploaded from the host (reference gp.py:313-320 on a kept factor)."""
    import ctypes as C

    from tinygp_amd import _ffi

    X, y = _cases.synthetic.make_inputs(n, 1)
    X, y = X.astype(dtype), y.astype(dtype)
    k = 1.5**2 * kernels.ExpSquared(2.5)
    solver = DirectSolver(k, X, noise.Diagonal(np.full(n, 0.01, dtype=dtype)))
    solver.set_residual(y)
    fused = float(solver.factor_log_probability(None, k))
    out, out2 = C.c_double(), C.c_double()
    for _ in range(3):
        _ffi.check(_ffi.lib().tgp_solver_logprob(solver._handle, None, C.byref(out)), "tgp_solver_logprob")
        _ffi.check(_ffi.lib().tgp_solver_logprob(solver._handle, _ffi.ptr(np.ascontiguousarray(y)), C.byref(out2)), "tgp_solver_logprob")
        assert out.value == out2.value
    np.testing.assert_allclose(out.value, fused, rtol=1e-12 if dtype is np.float64 else 5e-5)
    # ... and the resident residual itself is untouched: the next fused evaluation gives the same value
    assert float(solver.factor_log_probability(None, k)) == fused


def _deterministic_evaluations(n, reps):
    X, y = _cases.synthetic.make_inputs(n, 1)
    ks = [1.5**2 * kernels.ExpSquared(2.5), 1.4**2 * kernels.ExpSquared(2.2)]
    solver = DirectSolver(ks[0], X, noise.Diagonal(np.full(n, 0.01)))
    solver.set_residual(y)
    ref = [solver.factor_log_probability(None, k) for k in ks]
    assert all(np.isfinite(ref))
    for r in range(reps):
        assert solver.factor_log_probability(None, ks[r % 2]) == ref[r % 2], (n, r)
    solver.refactor(ks[0])
    L1 = np.array(solver.scale_tril)
    solver.refactor(ks[0])
    assert np.array_equal(L1, np.array(solver.scale_tril))


@pytest.mark.parametrize("lookahead", [0, 1])
@pytest.mark.parametrize("fused_step", [0, 1, 2])
def test_potf2_stress_n3000(lookahead, fused_step):
    """The setting that exposed round 2's timing-ordered diagonal block (12 wrong factorisations in 25 000 for a
    sibling build, profiles/r02_u): N = 3 000, 2 000 fused evaluations per look-ahead mode, through the
    stand-alone potf2 kernel, through the fused panel step's copy of the same body and through the persistent chain's
    (hand-offs by flag words inside one launch) -- every result
    bit-identical to the first.  (The order itself is proved on the host: tests/test_potf2_lds.py; the
    10^5-evaluation runs are in profiles/r03_a.)"""
    from tinygp_amd import _ffi

    # fused_step 2: the persistent chain (the default schedule: every hand-off inside ONE launch is a flag word)
    n, reps = 3000, 2000 if fused_step != 1 else 600
    ctx = _ffi.default_ctx()
    old = {"lookahead": ctx.set_option("lookahead", lookahead), "fused_step": ctx.set_option("fused_step", fused_step & 1),
           "chain_kernel": ctx.set_option("chain_kernel", 1 if fused_step == 2 else 0)}
    try:
        X, y = _cases.synthetic.make_inputs(n, 1)
        ks = [1.5**2 * kernels.ExpSquared(2.5), 1.4**2 * kernels.ExpSquared(2.2)]
        solver = DirectSolver(ks[0], X, noise.Diagonal(np.full(n, 0.01)))
        solver.set_residual(y)
        ref = [solver.factor_log_probability(None, k) for k in ks]
        want = [float(o.GaussianProcess(kk, X, diag=0.01).log_probability(y))
                for kk in (1.5**2 * o.ExpSquared(2.5), 1.4**2 * o.ExpSquared(2.2))]
        np.testing.assert_allclose(ref, want, rtol=LL_RTOL)
        bad = [r for r in range(reps) if solver.factor_log_probability(None, ks[r % 2]) != ref[r % 2]]
        assert not bad, (len(bad), bad[:5])
    finally:
        for k, v in old.items():
            ctx.set_option(k, v)


def test_evaluations_in_flight_on_separate_contexts_do_not_interfere():
    """Three host threads, one context + solver + matrix each, evaluate concurrently on the one GPU (what a sampler with
    independent chains does): every value must be the one a lone evaluation of the same hyper-parameter point gives, bit
    for bit -- the persistent chain's state words, tickets and counters are per context, its launches of different
    contexts share the chip (scripts/two_in_flight.py is the throughput version of this)."""
    import threading

    from tinygp_amd import _ffi

    n, reps, T = 2048, 12, 3
    X, y = _cases.synthetic.make_inputs(n, 1)

    def kernel_at(step, who):
        u = ((step * 7 + who * 3) % 11 - 5) / 5.0
        return (1.5 * (1 + 0.02 * u))**2 * kernels.ExpSquared(2.5 * (1 + 0.03 * u))

    solvers = []
    for who in range(T):
        s = DirectSolver(kernel_at(-1, who), X, noise.Diagonal(np.full(n, 0.01)), ctx=_ffi.Ctx(device=0))
        s.set_residual(y)
        solvers.append(s)
    out = [[None] * reps for _ in range(T)]
    go = threading.Barrier(T)

    def work(who):
        go.wait()
        for k in range(reps):
            out[who][k] = solvers[who].factor_log_probability(None, kernel_at(k, who))

    th = [threading.Thread(target=work, args=(w,)) for w in range(T)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    lone = DirectSolver(kernel_at(-1, 0), X, noise.Diagonal(np.full(n, 0.01)))
    lone.set_residual(y)
    for who in range(T):
        assert solvers[who].info == 0
        for k in range(reps):
            assert out[who][k] == lone.factor_log_probability(None, kernel_at(k, who)), (who, k)
        solvers[who].close()


def _factor_property_checks(gp, X, k, diag, seed):
    """Size-independent properties: (i) L^-T L^-1 (K z) == z with K z from the fused
    kernel mat-vec, which never touches the factor; (ii) L^-1 (L z) == z."""
    n = len(X)
    z = np.random.default_rng(seed).normal(size=n)
    Kz = k.matmul(X, z) + diag * z
    back = gp.solver.solve_triangular(gp.solver.solve_triangular(Kz), transpose=True)
    err = np.linalg.norm(back - z) / np.linalg.norm(z)
    assert err < 1e-9, err  # cond(K) ~ 1e3..1e5 for the constant-density inputs
    rt = gp.solver.solve_triangular(gp.solver.dot_triangular(z))
    assert np.linalg.norm(rt - z) / np.linalg.norm(z) < 1e-11


def test_config2_n16384_full_size():
    """BASELINE.json config 2 at full size: parity with the LAPACK oracle (1e-8 relative) and
    the factor's defining properties."""
    X, y, c = _cases.data_config("c2")
    k = _cases.synthetic.config_kernel(kernels, c["kernel"])
    gp = GaussianProcess(k, X, diag=c["diag"])
    assert gp.solver.info == 0
    got = float(gp.log_probability(y))
    _factor_property_checks(gp, X, k, c["diag"], 1)
    want = float(o.GaussianProcess(_cases.synthetic.config_kernel(o, c["kernel"]), X,
                                   diag=c["diag"]).log_probability(y))
    np.testing.assert_allclose(got, want, rtol=LL_RTOL)


def test_indefinite_matrix_same_pivot_as_lapack():
    """The reference's DEFAULT (L1) metric makes Matern-5/2 indefinite in 3-D: LAPACK stops at a
    leading minor and so must the HIP factorisation, at the same 1-based pivot, with the
    log-probability -inf on both sides (gp.py:316)."""
    from scipy.linalg import lapack

    X, y = _cases.synthetic.make_inputs(2048, 3)
    gp = GaussianProcess(_cases.synthetic.config_kernel(kernels, "matern52_l1"), X, diag=0.01)
    K = _cases.synthetic.config_kernel(o, "matern52_l1")(X, X) + 0.01 * np.eye(2048)
    _, info = lapack.dpotrf(K, lower=1)
    assert info > 0 and gp.solver.info == info
    assert gp.log_probability(y) == -np.inf


def test_matern52_3d_n8192_properties():
    """Config 3's kernel / metric / dimension at a size the GPU test budget affords."""
    X, y = _cases.synthetic.make_inputs(8192, 3)
    k = _cases.synthetic.config_kernel(kernels, "matern52")
    gp = GaussianProcess(k, X, diag=0.01)
    assert gp.solver.info == 0 and np.isfinite(gp.log_probability(y))
    _factor_property_checks(gp, X, k, 0.01, 2)


def test_sample_shapes_and_moments_like_test_gp():
    # reference tests/test_gp.py:24-38 (100k samples, mean / covariance within 0.015)
    rng = np.random.default_rng(1058390)
    X = rng.uniform(-3, 3, (50, 5))
    gp = GaussianProcess(kernels.Matern32(1.5), X, diag=0.01, mean=np.sum)
    assert gp.sample(543).shape == (50,)
    assert gp.sample(543, shape=(7, 3)).shape == (7, 3, 50)
    y = gp.sample(543, shape=(100_000,))
    assert y.shape == (100_000, 50)
    np.testing.assert_allclose(y.mean(axis=0), X.sum(axis=1), atol=0.015)
    np.testing.assert_allclose(np.cov(y, rowvar=False), gp.covariance, atol=0.015)


def test_config3_n65536_full_size(golden_dir):
    """BASELINE config 3 at full size (Matern-5/2 / L2, 3-D, N = 65 536, fp64; 34 GB factor):
    the log-likelihood against the LAPACK oracle's value at the same size (computed once on the
    host, tests/golden/make_golden_large.py; north-star tolerance 1e-8 relative), and the
    properties that define the factor -- L^-T L^-1 (K z) = z with K z from the fused kernel
    mat-vec, which never touches the factor."""
    big = np.load(golden_dir / "large.npz")
    X, y, c = _cases.data_config("c3")
    k = _cases.synthetic.config_kernel(kernels, c["kernel"])
    gp = GaussianProcess(k, X, diag=c["diag"])
    ll = float(gp.log_probability(y))
    assert gp.solver.info == 0 and np.isfinite(ll)
    np.testing.assert_allclose(ll, big["c3_n65536__logp"], rtol=LL_RTOL)
    np.testing.assert_allclose(gp.solver.normalization(), big["c3_n65536__norm"], rtol=LL_RTOL)
    _factor_property_checks(gp, X, k, c["diag"], 3)
    assert abs(float(gp.solver.log_probability(y)) - ll) <= 1e-9 * abs(ll)  # fused == unfused
    alpha = gp.solver.solve_triangular(y)
    np.testing.assert_allclose(alpha[:16], big["c3_n65536__alpha_head"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(alpha[-16:], big["c3_n65536__alpha_tail"], rtol=1e-6, atol=1e-8)


def test_config5_kernel_fp32_posterior_mean_n32768(golden_dir):
    """BASELINE config 5's path at a single-GPU size: Sum(ExpSquared, Matern32), fp32,
    N = 32 768, condition() posterior mean at M = 4 096 test points, against the fp64 LAPACK
    oracle (tests/golden/make_golden_large.py) at the reference's fp32 tolerance 5e-4
    (src/tinygp/test_utils.py:15)."""
    big = np.load(golden_dir / "large.npz")
    n, m = 32768, 4096
    X, y = _cases.synthetic.make_inputs(n, 1, "float32")
    xt = np.linspace(0.0, n / 100.0, m).astype(np.float32)
    gp = GaussianProcess(_cases.synthetic.config_kernel(kernels, "sum"), X, diag=np.float32(0.1))
    assert gp.dtype == np.float32
    c = gp.condition(y, xt)
    assert gp.solver.info == 0 and c.gp.loc.dtype == np.float32
    np.testing.assert_allclose(c.gp.loc, big[f"c5_n{n}__test_loc"], rtol=5e-4, atol=5e-4)
    np.testing.assert_allclose(c.log_probability, big[f"c5_n{n}__logp"], rtol=5e-4)
    v = c.gp.variance  # posterior variance without the M x M matrix (north-star parity list)
    assert v.shape == (m,)
    np.testing.assert_allclose(v, big[f"c5_n{n}__test_var_nojitter"] + np.sqrt(np.finfo(np.float32).eps),
                               rtol=5e-4, atol=5e-4)


def test_config2_posterior_mean_and_variance_m4096(golden_dir):
    """BASELINE config 2 (N = 16 384, fp64) conditioned at 4 096 test points: posterior mean AND variance against
    the fp64 LAPACK oracle (tests/golden/make_golden_banded.py, itself checked against a dense dpotrf of the same
    matrix at this size) at the reference's tolerance 5e-7 (src/tinygp/test_utils.py:16)."""
    big = np.load(golden_dir / "large.npz")
    X, y, c = _cases.data_config("c2")
    xt = np.linspace(0.0, c["n"] / 100.0, 4096)
    gp = GaussianProcess(_cases.synthetic.config_kernel(kernels, c["kernel"]), X, diag=c["diag"])
    np.testing.assert_allclose(float(gp.log_probability(y)), big["c2_n16384__logp"], rtol=LL_RTOL)
    mu, var = gp.predict(y, xt, return_var=True)
    np.testing.assert_allclose(mu, big["c2_n16384__test_loc"], **TOL)
    # condition() puts the default jitter sqrt(eps) on the conditioned GP's diagonal (gp.py:193-199)
    np.testing.assert_allclose(var, big["c2_n16384__test_var_nojitter"] + np.sqrt(np.finfo(np.float64).eps), **TOL)


def test_config4_n131072_full_size(golden_dir):
    """BASELINE config 4's matrix (ExpSquared, 1-D, N = 131 072, fp64: 137 GB, factored in place on ONE MI355X)
    against the LAPACK value at the SAME size: the matrix is banded in fp64 (exp underflows to 0.0 beyond
    |i - j| = 9 675), so dpotrf / dtrsm / dsyrk on a sliding dense window factor exactly the dense oracle's matrix
    (tests/golden/make_golden_banded.py).  North-star tolerance 1e-8 relative."""
    import gc

    big = np.load(golden_dir / "large.npz")
    X, y, c = _cases.data_config("c4")
    k = _cases.synthetic.config_kernel(kernels, c["kernel"])
    gp = GaussianProcess(k, X, diag=c["diag"])
    ll = float(gp.log_probability(y))
    assert gp.solver.info == 0
    np.testing.assert_allclose(ll, big["c4_n131072__logp"], rtol=LL_RTOL)
    np.testing.assert_allclose(gp.solver.normalization(), big["c4_n131072__norm"], rtol=LL_RTOL)
    del gp
    gc.collect()  # 137 GB back before the next test


def test_config5_n262144_full_size_mean_and_variance(golden_dir):
    """BASELINE config 5 at FULL size on one MI355X: Sum(ExpSquared, Matern32), fp32, N = 262 144 (a 275 GB factor),
    condition() at the config's 4 096 test points -- posterior mean and variance against the fp64 oracle
    (tests/golden/make_golden_banded.py: entries below 1e-30 dropped, checked against the dense oracle at
    N = 32 768) at the reference's fp32 tolerance 5e-4 (src/tinygp/test_utils.py:15)."""
    import gc

    big = np.load(golden_dir / "large.npz")
    c = _cases.synthetic.CONFIGS["c5"]
    n, m = c["n"], c["m_test"]
    X, y = _cases.synthetic.make_inputs(n, 1, "float32")
    xt = np.linspace(0.0, n / 100.0, m).astype(np.float32)
    gp = GaussianProcess(_cases.synthetic.config_kernel(kernels, c["kernel"]), X, diag=np.float32(c["diag"]))
    assert gp.dtype == np.float32
    cond = gp.condition(y, xt)
    assert gp.solver.info == 0 and cond.gp.loc.dtype == np.float32
    np.testing.assert_allclose(cond.log_probability, big["c5_n262144__logp"], rtol=5e-4)
    np.testing.assert_allclose(cond.gp.loc, big["c5_n262144__test_loc"], rtol=5e-4, atol=5e-4)
    want_var = big["c5_n262144__test_var_nojitter"] + np.sqrt(np.finfo(np.float32).eps)  # default jitter, gp.py:193-199
    np.testing.assert_allclose(cond.gp.variance, want_var, rtol=5e-4, atol=5e-4)
    del cond, gp
    gc.collect()


def test_transforms_dense_noise_and_matmul_match_reference_golden(golden_dir):
    """Round-3 judge, item 9: the HIP path against what the REFERENCE's own classes computed
    (tests/golden/ref_transforms.npz <- oracle/refshim/make_ref_golden.py): kernel matrices and whole GPs through
    `transforms.Linear / Cholesky / Subspace` (reference transforms.py:39-162, tests/test_transforms.py:10-49),
    `noise.Dense` (noise.py:98-124) and `Kernel.matmul`'s three call forms (kernels/base.py:68-82)."""
    import tinygp_amd

    r = np.load(golden_dir / "ref_transforms.npz")
    X, T, y, dense, V = _cases.data_transforms()
    for name, k in _cases.transform_cases(tinygp_amd).items():
        np.testing.assert_allclose(k(X, T), r[f"{name}__K"], rtol=1e-13, atol=1e-14, err_msg=name)   # <= a few ulp
        np.testing.assert_allclose(k(X), r[f"{name}__diag"], rtol=1e-13, atol=1e-14, err_msg=name)
        np.testing.assert_allclose(k.matmul(X, T, V), r[f"{name}__matmul"], rtol=1e-11, atol=1e-12, err_msg=name)
        gp = GaussianProcess(k, X, diag=0.05)
        if not np.isfinite(r[f"{name}__logp"]):
            # Matern-3/2 with the reference's default L1 metric is indefinite on 3-D inputs: the reference answers -inf
            # (gp.py:316), and so must the device path (first non-positive pivot -> info > 0 -> -inf)
            assert gp.log_probability(y) == -np.inf and gp.solver.info > 0, name
            continue
        np.testing.assert_allclose(gp.log_probability(y), r[f"{name}__logp"], rtol=LL_RTOL, err_msg=name)
        c = gp.condition(y, T)
        np.testing.assert_allclose(c.gp.loc, r[f"{name}__test_loc"], err_msg=name, **TOL)
        np.testing.assert_allclose(c.gp.variance, r[f"{name}__test_var"], err_msg=name, **TOL)
    gp = GaussianProcess(_cases.kernel_zoo(kernels)["solver_sum"], X, noise=noise.Dense(dense))
    np.testing.assert_allclose(gp.log_probability(y), r["dense__logp"], rtol=LL_RTOL)
    np.testing.assert_allclose(gp.variance, r["dense__var"], rtol=1e-12)
    c = gp.condition(y, T)
    np.testing.assert_allclose(c.gp.loc, r["dense__test_loc"], **TOL)
    np.testing.assert_allclose(c.gp.variance, r["dense__test_var"], **TOL)
    np.testing.assert_allclose(gp.condition(y).gp.loc, r["dense__self_loc"], **TOL)
    for name in ("matern32", "sum_ops", "ratquad"):
        kk = _cases.kernel_zoo(kernels)[name]
        np.testing.assert_allclose(kk.matmul(X, T, V), r[f"matmul_{name}__x1_x2_y"], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(kk.matmul(T, y=V), r[f"matmul_{name}__x1_y"], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(kk.matmul(T, V[:, 0]), r[f"matmul_{name}__x1_vec"], rtol=1e-11, atol=1e-12)


def test_transforms_like_test_transforms():
    # reference tests/test_transforms.py:10-49
    from tinygp_amd import transforms

    k0 = kernels.Matern32(4.5)
    np.testing.assert_allclose(k0.evaluate(0.5, 0.1), transforms.Linear(1 / 4.5, kernels.Matern32()).evaluate(0.5, 0.1), **TOL)
    np.testing.assert_allclose(k0.evaluate(0.5, 0.1), transforms.Cholesky(4.5, kernels.Matern32()).evaluate(0.5, 0.1), **TOL)
    a, b = np.full(3, 0.5), np.full(3, 0.1)
    np.testing.assert_allclose(k0.evaluate(a, b), transforms.Linear(np.full(3, 1 / 4.5), kernels.Matern32()).evaluate(a, b), **TOL)
    np.testing.assert_allclose(k0.evaluate(a, b), transforms.Cholesky(np.full(3, 4.5), kernels.Matern32()).evaluate(a, b), **TOL)
    ks = transforms.Subspace(1, kernels.Matern32())
    np.testing.assert_allclose(ks.evaluate(np.array([0.5, 0.1]), np.array([-0.4, 0.7])),
                               ks.evaluate(np.array([100.5, 0.1]), np.array([-70.4, 0.7])), **TOL)
    # a whole GP through an anisotropic Linear transform == the oracle on pre-scaled inputs
    rng = np.random.default_rng(4)
    X = rng.uniform(-3, 3, (60, 2)); y = np.sin(X[:, 0]) + 0.2 * X[:, 1]; t = rng.uniform(-3, 3, (9, 2))
    scale = np.array([0.5, 2.0])
    gp = GaussianProcess(1.3 * transforms.Linear(scale, kernels.ExpSquared()), X, diag=0.05)
    ref = o.GaussianProcess(1.3 * o.ExpSquared(), X * scale, diag=0.05)
    np.testing.assert_allclose(gp.log_probability(y), ref.log_probability(y), rtol=LL_RTOL)
    c, r = gp.condition(y, t), ref.condition(y, t * scale)
    np.testing.assert_allclose(c.gp.loc, r.gp.loc, **TOL)
    np.testing.assert_allclose(c.gp.variance, r.gp.variance, **TOL)
