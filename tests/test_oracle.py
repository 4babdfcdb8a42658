"""CPU tests of the oracle itself: regression against the committed golden vectors and the
pins the reference's own tests use (agreement between independent implementations)."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

import _cases
from oracle import tinygp_np as o

ROOT = Path(__file__).resolve().parent.parent


# ---- golden regression -------------------------------------------------------------------
def test_kernels_match_golden(golden_dir):
    g = np.load(golden_dir / "kernels.npz")
    x1, x2 = _cases.data_kernels()
    xs, _, ts = _cases.data_solver()
    for name, k in _cases.kernel_zoo(o).items():
        np.testing.assert_allclose(k(x1, x2), g[f"{name}__5d"], rtol=1e-14, atol=1e-15)
        np.testing.assert_allclose(k(xs, ts), g[f"{name}__1d"], rtol=1e-14, atol=1e-15)
        np.testing.assert_allclose(k(x1), g[f"{name}__diag"], rtol=1e-14, atol=1e-15)


def test_gp_matches_golden(golden_dir):
    g = np.load(golden_dir / "gp.npz")
    for name, (gp, y, t) in _cases.gp_cases(o, o.GaussianProcess).items():
        np.testing.assert_allclose(gp.log_probability(y), g[f"{name}__logp"], rtol=1e-12)
        c1 = gp.condition(y, t)
        np.testing.assert_allclose(c1.gp.loc, g[f"{name}__test_loc"], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(c1.gp.covariance, g[f"{name}__test_cov"], rtol=1e-8, atol=1e-10)


def test_config1_matches_golden_and_survey_datum(golden_dir):
    g = np.load(golden_dir / "configs.npz")
    X, y, c = _cases.data_config("c1")
    gp = o.GaussianProcess(_cases.synthetic.config_kernel(o, c["kernel"]), X, diag=c["diag"])
    lp = float(gp.log_probability(y))
    np.testing.assert_allclose(lp, g["expsq_n1024__logp"], rtol=1e-12)
    # SURVEY.md 8(c) sanity datum, measured independently during the survey
    np.testing.assert_allclose(lp, 853.7063780492, rtol=1e-11)


# ---- pins: closed forms and independent linear algebra -------------------------------------
def test_closed_form_expsquared():
    # reference tests/test_kernels/test_kernels.py:43-52
    x1, x2 = _cases.data_kernels()
    scale = 1.5
    want = np.exp(-0.5 * np.sum(np.square((x1[:, None, :] - x2[None, :, :]) / scale), axis=-1))
    o.assert_allclose(o.ExpSquared(scale)(x1, x2), want)


def test_closed_forms_1d():
    xs, _, ts = _cases.data_solver()
    r = np.abs(xs[:, None] - ts[None, :]) / 1.5
    o.assert_allclose(o.Exp(1.5)(xs, ts), np.exp(-r))
    o.assert_allclose(o.Matern32(1.5)(xs, ts), (1 + np.sqrt(3) * r) * np.exp(-np.sqrt(3) * r))
    o.assert_allclose(o.Matern52(1.5)(xs, ts),
                      (1 + np.sqrt(5) * r + 5 * r**2 / 3) * np.exp(-np.sqrt(5) * r))
    o.assert_allclose(o.Cosine(1.5)(xs, ts), np.cos(2 * np.pi * r))
    o.assert_allclose(o.ExpSineSquared(1.5, gamma=0.3)(xs, ts), np.exp(-0.3 * np.sin(np.pi * r) ** 2))
    o.assert_allclose(o.RationalQuadratic(1.5, alpha=0.7)(xs, ts), (1 + r**2 / 1.4) ** -0.7)


def test_metric_defaults():
    # stationary.py:56,102: L1 everywhere except ExpSquared; RationalQuadratic squares the L1 norm
    x1, x2 = _cases.data_kernels()
    d = x1[:, None, :] - x2[None, :, :]
    l1 = np.abs(d).sum(-1)
    o.assert_allclose(o.Matern32(1.5)(x1, x2), (1 + np.sqrt(3) * l1 / 1.5) * np.exp(-np.sqrt(3) * l1 / 1.5))
    o.assert_allclose(o.RationalQuadratic(alpha=1.5)(x1, x2), (1 + 0.5 * l1**2 / 1.5) ** -1.5)


def test_constant_and_ops():
    # test_kernels.py:23-40,62-69
    x1, x2 = _cases.data_kernels()
    k1 = o.Matern32(2.5)
    o.assert_allclose(2.5 * k1(x1, x2), (2.5 * k1)(x1, x2))
    ka = 1.5 * o.Matern32(2.5)
    kb = 0.9 * o.ExpSineSquared(scale=1.5, gamma=0.3)
    o.assert_allclose(ka(x1, x2) + kb(x1, x2), (ka + kb)(x1, x2))
    o.assert_allclose(ka(x1, x2) * kb(x1, x2), (ka * kb)(x1, x2))
    with pytest.raises(ValueError):
        o.Constant(np.ones(3)).evaluate(np.ones(3), np.ones(3))
    assert sum([k1, k1]) is not None


def test_conditioned_vs_dense_solve():
    # test_kernels.py:72-83
    x1, x2 = _cases.data_kernels()
    k1 = 1.5 * o.Matern32(2.5)
    k2 = 0.9 * o.ExpSineSquared(scale=1.5, gamma=0.3)
    K = k1(x1, x1) + 0.1 * np.eye(x1.shape[0])
    solver = o.DirectSolver.init(k1, x1, o.Diagonal(np.full(x1.shape[0], 0.1)))
    cond = o.Conditioned(x1, solver, k2)
    o.assert_allclose(cond(x1, x2), k2(x1, x2) - k2(x1, x1) @ np.linalg.solve(K, k2(x1, x2)))


def test_gp_vs_textbook_formulas():
    """The identities george implements (test_george_compat.py:116-154), by np.linalg."""
    for nd in (1, 3):
        x, y, t, diag = _cases.data_george(nd)
        for name in ["exp", "expsq", "matern32", "matern52", "ratquad"]:
            k = _cases.kernel_zoo(o)[name]
            gp = o.GaussianProcess(k, x, diag=diag)
            K = k(x, x) + np.diag(diag)
            sign, logdet = np.linalg.slogdet(K)
            want = -0.5 * y @ np.linalg.solve(K, y) - 0.5 * logdet - 0.5 * len(y) * np.log(2 * np.pi)
            o.assert_allclose(gp.log_probability(y), want)
            Ks = k(t, x)
            mu = Ks @ np.linalg.solve(K, y)
            cov = k(t, t) - Ks @ np.linalg.solve(K, Ks.T)
            loc, c = gp.predict(y, t, return_cov=True)
            o.assert_allclose(loc, mu)
            o.assert_allclose(c - np.sqrt(np.finfo(float).eps) * np.eye(len(t)), cov)
            o.assert_allclose(gp.predict(y), k(x, x) @ np.linalg.solve(K, y))
            o.assert_allclose(gp.predict(y, x), k(x, x) @ np.linalg.solve(K, y))


def test_oracle_against_60_digit_arithmetic():
    """Known answers in (near-)exact arithmetic: the formulas of kernels/stationary.py:76-153
    and gp.py:313-361, written out once more with mpmath at 60 digits on the reference's
    test_solver.py:16-57 data.  Pins the oracle's double-precision results (kernel entries,
    log-likelihood, conditional mean and variance) to the rounding error of the inputs."""
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 60
    x, y, t = _cases.data_solver()
    diag = 0.1
    X, Y, Tt = [mp.mpf(float(v)) for v in x], [mp.mpf(float(v)) for v in y], [mp.mpf(float(v)) for v in t]
    s3, s5, two_pi = mp.sqrt(3), mp.sqrt(5), 2 * mp.pi

    def m32(r):
        return (1 + s3 * r) * mp.e ** (-s3 * r)

    def m52(r):
        return (1 + s5 * r + 5 * r * r / 3) * mp.e ** (-s5 * r)

    # 1-D inputs: every metric is |dx|.  Hyper-parameters enter as the doubles the oracle sees.
    scal = {v: mp.mpf(float(v)) for v in ("1.8", "1.5", "0.9", "0.7")}
    for name in ("solver_m32", "solver_m52", "solver_exp", "solver_cos", "solver_sum", "expsq"):
        k = _cases.kernel_zoo(o)[name]

        def fd(d, name=name):
            a, l, a2, l2 = (scal[v] for v in ("1.8", "1.5", "0.9", "0.7"))
            if name == "solver_m32":
                return a * a * m32(d / l)
            if name == "solver_m52":
                return a * a * m52(d / l)
            if name == "solver_exp":
                return a * a * mp.e ** (-d / l)
            if name == "solver_cos":
                return a * a * mp.cos(two_pi * d / l)
            if name == "solver_sum":
                return a * a * m32(d / l) + a2 * a2 * m52(d / l2)
            return mp.e ** (-(d / l) ** 2 / 2)

        n, m = len(X), len(Tt)
        K = mp.matrix(n, n)
        for i in range(n):
            for j in range(n):
                K[i, j] = fd(abs(X[i] - X[j])) + (mp.mpf(diag) if i == j else 0)
        Ks = mp.matrix(n, m)
        for i in range(n):
            for j in range(m):
                Ks[i, j] = fd(abs(X[i] - Tt[j]))
        # kernel entries
        got = k(x, t)
        want = np.array([[float(Ks[i, j]) for j in range(m)] for i in range(n)])
        # (cos(2 pi r) at r ~ 4 amplifies the rounding of its argument: absolute 1e-14)
        np.testing.assert_allclose(got, want, rtol=2e-15, atol=5e-14)
        L = mp.cholesky(K)
        yv = mp.matrix(Y)
        alpha = mp.lu_solve(K, yv)
        logdet = 2 * sum(mp.log(L[i, i]) for i in range(n))
        ll = -(yv.T * alpha)[0, 0] / 2 - logdet / 2 - mp.mpf(n) / 2 * mp.log(two_pi)
        gp = o.GaussianProcess(k, x, diag=diag)
        np.testing.assert_allclose(float(gp.log_probability(y)), float(ll), rtol=1e-11)
        mu = Ks.T * alpha
        loc, var = gp.predict(y, t, return_var=True)
        np.testing.assert_allclose(loc, [float(mu[j]) for j in range(m)], rtol=1e-9, atol=1e-11)
        jitter = mp.sqrt(mp.mpf(float(np.finfo(np.float64).eps)))
        want_var = []
        for j in range(m):
            v = mp.lu_solve(K, Ks[:, j])
            want_var.append(float(fd(mp.mpf(0)) - sum(Ks[i, j] * v[i] for i in range(n)) + jitter))
        np.testing.assert_allclose(var, want_var, rtol=1e-8, atol=1e-10)


def test_oracle_metrics_against_60_digit_arithmetic():
    """The same in 3-D, where the metric matters (kernels/distance.py:30-59): the default L1
    distance of Matern-3/2, the L2 default of ExpSquared, RationalQuadratic's squared L1
    distance (stationary.py:232-235 with the base-class squared_distance) and an explicit L2."""
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 60
    x, y, t, diag = _cases.data_george(3)
    X = [[mp.mpf(float(v)) for v in row] for row in x]
    T3 = [[mp.mpf(float(v)) for v in row] for row in t]
    ell, s3 = mp.mpf(1.5), mp.sqrt(3)

    def l1(a, b):
        return sum(abs(p - q) for p, q in zip(a, b))

    def l2sq(a, b):
        return sum((p - q) ** 2 for p, q in zip(a, b))

    funcs = {
        "matern32": lambda a, b: (1 + s3 * l1(a, b) / ell) * mp.e ** (-s3 * l1(a, b) / ell),
        "expsq": lambda a, b: mp.e ** (-l2sq(a, b) / ell**2 / 2),
        "ratquad": lambda a, b: (1 + l1(a, b) ** 2 / (2 * mp.mpf(1.5))) ** (-mp.mpf(1.5)),
        "l2_m32": lambda a, b: ((1 + s3 * mp.sqrt(l2sq(a, b)) / ell)
                                * mp.e ** (-s3 * mp.sqrt(l2sq(a, b)) / ell)),
    }
    n, m = len(X), len(T3)
    for name, f in funcs.items():
        k = _cases.kernel_zoo(o)[name]
        want = np.array([[float(f(X[i], T3[j])) for j in range(m)] for i in range(n)])
        np.testing.assert_allclose(k(x, t), want, rtol=4e-15, atol=1e-17)
        K = mp.matrix(n, n)
        for i in range(n):
            for j in range(n):
                K[i, j] = f(X[i], X[j]) + (mp.mpf(float(diag[i])) if i == j else 0)
        L = mp.cholesky(K)
        yv = mp.matrix([mp.mpf(float(v)) for v in y])
        alpha = mp.lu_solve(K, yv)
        ll = (-(yv.T * alpha)[0, 0] / 2 - sum(mp.log(L[i, i]) for i in range(n))
              - mp.mpf(n) / 2 * mp.log(2 * mp.pi))
        got = float(o.GaussianProcess(k, x, diag=diag).log_probability(y))
        np.testing.assert_allclose(got, float(ll), rtol=1e-11)


def test_means_equivalent():
    # test_gp.py:41-51 (y is a scalar there: it broadcasts against loc)
    rng = np.random.default_rng(1058390)
    X = rng.uniform(-3, 3, (50, 5))
    y = rng.normal(len(X))
    gp1 = o.GaussianProcess(o.Matern32(1.5), X, diag=0.01, mean=lambda x: 0.0)
    gp2 = o.GaussianProcess(o.Matern32(1.5), X, diag=0.01, mean=0.0)
    gp3 = o.GaussianProcess(o.Matern32(1.5), X, diag=0.01)
    o.assert_allclose(gp1.log_probability(y), gp2.log_probability(y))
    o.assert_allclose(gp1.log_probability(y), gp3.log_probability(y))


def test_nonpd_gives_minus_inf():
    x = np.linspace(0, 1, 20)
    gp = o.GaussianProcess(o.ExpSquared(5.0), x, diag=-0.5)
    assert gp.log_probability(np.sin(x)) == -np.inf


# ---- independent plain-C restatement --------------------------------------------------------
@pytest.fixture(scope="module")
def refc():
    so = ROOT / "oracle" / "_build" / "libref_c.so"
    if not so.exists():
        subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)
    lib = C.CDLL(str(so))
    lib.ref_log_probability.restype = C.c_double
    return lib


def _prog(kernel):
    from tinygp_amd import _ffi
    return _ffi.as_kprog(kernel.program())


def test_c_oracle_agrees(refc):
    from tinygp_amd import kernels as pk

    dp = C.POINTER(C.c_double)
    zoo_o, zoo_p = _cases.kernel_zoo(o), _cases.kernel_zoo(pk)
    x1, x2 = _cases.data_kernels()
    for name in zoo_o:
        kp, nops = _prog(zoo_p[name])
        out = np.empty((50, 50))
        refc.ref_kmat(kp, nops, C.c_int64(50), C.c_int64(50), 5, x1.ctypes.data_as(dp),
                      x2.ctypes.data_as(dp), None, out.ctypes.data_as(dp))
        np.testing.assert_allclose(out, zoo_o[name](x1, x2), rtol=1e-13, atol=1e-15, err_msg=name)
    # whole log_probability through the textbook unblocked Cholesky
    X, y, c = _cases.data_config("c1")
    n = 512
    X, y = np.ascontiguousarray(X[:n]), np.ascontiguousarray(y[:n])
    kp, nops = _prog(_cases.synthetic.config_kernel(pk, "expsq"))
    diag = np.full(n, 0.01)
    work = np.empty((n, n))
    got = refc.ref_log_probability(kp, nops, C.c_int64(n), 1, X.ctypes.data_as(dp),
                                   diag.ctypes.data_as(dp), y.ctypes.data_as(dp),
                                   work.ctypes.data_as(dp))
    want = float(o.GaussianProcess(_cases.synthetic.config_kernel(o, "expsq"), X, diag=0.01).log_probability(y))
    np.testing.assert_allclose(got, want, rtol=1e-10)


def test_closed_form_kernel_derivatives_match_central_differences():
    """oracle/kernel_derivs_np.py (the closed forms the device evaluates, stated independently) against central
    differences of the oracle's own kernel matrices: every stationary leaf, both metrics, 1-D and 3-D inputs."""
    import numpy as np

    from oracle import kernel_derivs_np as kd
    from oracle import tinygp_np as o

    rng = np.random.default_rng(3)
    leaves = {
        "scale": [lambda s, m: o.Exp(s, distance=m), lambda s, m: o.ExpSquared(s, distance=m), lambda s, m: o.Matern32(s, distance=m),
                  lambda s, m: o.Matern52(s, distance=m), lambda s, m: o.Cosine(s, distance=m),
                  lambda s, m: o.ExpSineSquared(s, distance=m, gamma=0.7), lambda s, m: o.RationalQuadratic(s, distance=m, alpha=1.3)],
    }
    for d in (1, 3):
        X1, X2 = rng.normal(size=(40, d)) * 1.5, rng.normal(size=(35, d)) * 1.5
        X2[0] = X1[0]  # a zero distance
        for metric in (o.L1Distance, o.L2Distance):
            for make in leaves["scale"]:
                s0, h = 0.9, 1e-6
                want = (make(s0 + h, metric())(X1, X2) - make(s0 - h, metric())(X1, X2)) / (2 * h)
                got = kd.dleaf(make(s0, metric()), X1, X2, "scale")
                np.testing.assert_allclose(got, want, rtol=2e-7, atol=2e-8, err_msg=f"{type(make(s0, metric())).__name__} {metric.__name__} d={d}")
            for name, make, p0 in (("gamma", lambda v, m: o.ExpSineSquared(0.9, distance=m, gamma=v), 0.7),
                                   ("alpha", lambda v, m: o.RationalQuadratic(0.9, distance=m, alpha=v), 1.3)):
                h = 1e-6
                want = (make(p0 + h, metric())(X1, X2) - make(p0 - h, metric())(X1, X2)) / (2 * h)
                got = kd.dleaf(make(p0, metric()), X1, X2, name)
                np.testing.assert_allclose(got, want, rtol=2e-7, atol=2e-8, err_msg=f"{name} {metric.__name__} d={d}")


def test_trace_identity_with_closed_form_derivatives_matches_the_gradient_oracle():
    """d ll / d theta = 1/2 tr((alpha alpha^T - K^-1) dK/dtheta) with the CLOSED-FORM dK/dtheta against oracle/grad_np.py
    (the same identity with dK/dtheta by central differences): amplitude and scale of Matern-3/2 and ExpSquared."""
    import numpy as np

    from oracle import grad_np
    from oracle import kernel_derivs_np as kd
    from oracle import tinygp_np as o

    rng = np.random.default_rng(8)
    X = np.sort(rng.uniform(0, 10, size=(60, 1)), axis=0)
    y = np.sin(X[:, 0]) + 0.1 * rng.normal(size=60)
    diag = 0.05
    for leaf in (o.Matern32, o.ExpSquared):
        theta = np.array([1.7, 0.8])  # amplitude, scale
        build = lambda t: t[0] * leaf(t[1])  # noqa: E731
        _, g, _, alpha = grad_np.log_probability_and_grad(build, theta, X, diag, y)
        K = build(theta)(X, X) + diag * np.eye(60)
        G = np.outer(alpha, alpha) - np.linalg.inv(K)
        closed = np.array([0.5 * np.sum(G * leaf(theta[1])(X, X)), 0.5 * np.sum(G * theta[0] * kd.dleaf(leaf(theta[1]), X, X))])
        np.testing.assert_allclose(closed, g, rtol=1e-6, atol=1e-8)
