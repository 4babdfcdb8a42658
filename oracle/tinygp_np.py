"""CPU oracle for the tinygp dense ``DirectSolver`` hot path (NumPy / SciPy-LAPACK).

TEST INFRASTRUCTURE ONLY.  Nothing under ``tinygp_amd/`` may import this file;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and only as the checker / the reported CPU baseline.

PARITY PINNED to the reference's own execution (since round 3).  The reference (`/root/reference`, dfm/tinygp) needs
``jax`` + ``equinox`` and Python >= 3.11, none of which this image has, and ships no golden vectors for this path
(SURVEY.md section 8c); its arithmetic lives in the un-vendored third-party ``jax`` / ``jaxlib`` (``pyproject.toml:18``:
``jax.scipy.linalg.cholesky`` / ``solve_triangular``, LAPACK ``?potrf`` / BLAS ``?trsm`` on CPU).  ``oracle/refshim/``
therefore holds NumPy stand-ins for the dozen jax / equinox primitives the reference uses (``vmap`` = a Python loop,
``jit`` = identity, ``cholesky`` / ``solve_triangular`` = LAPACK) and ``oracle/refshim/make_ref_golden.py`` imports THE
UNMODIFIED REFERENCE PACKAGE as it lies under ``/root/reference`` and writes ``tests/golden/ref_{kernels,gp,configs,
transforms}.npz`` -- 19 kernels x 3 shapes, 16 GP cases, BASELINE config 1 in full, the transforms and ``noise.Dense``.
This file reproduces those kernel matrices bit for bit and the GP results to 1e-12 ... 1e-9 (``tests/test_oracle.py``),
and where the reference tree exists the generation is re-run and must reproduce the committed fixtures
(``tests/test_reference_pin.py``).  The full-size configs (N >= 16 384) are pinned to this oracle's LAPACK arithmetic, which
is chained to the reference at N <= 4 096 (the shim's looped ``vmap`` is N^2 Python calls).  Independent second
restatement: plain C, ``oracle/ref_c.c``.

Every function cites the reference ``file:line`` it follows (paths relative to
``/root/reference/src/tinygp``).
"""

from __future__ import annotations

import numpy as np
import scipy.linalg as sla

__all__ = [
    "L1Distance", "L2Distance", "Kernel", "Sum", "Product", "Constant", "Exp",
    "ExpSquared", "Matern32", "Matern52", "Cosine", "ExpSineSquared",
    "RationalQuadratic", "Conditioned", "Diagonal", "Dense", "DirectSolver",
    "GaussianProcess", "ConditionResult", "default_diag", "assert_allclose",
]


# --------------------------------------------------------------------------
# kernels/distance.py
# --------------------------------------------------------------------------
def _as_points(X):
    """(N,) -> (N,1); (N,D) stays.  A scalar coordinate behaves as D=1."""
    X = np.asarray(X)
    if X.ndim == 1:
        return X[:, None]
    if X.ndim != 2:
        raise ValueError("oracle supports X of shape (N,) or (N, D)")
    return X


class Distance:
    def distance(self, d):  # d = X1 - X2 with the feature axis last
        raise NotImplementedError

    def squared_distance(self, d):  # distance.py:30-38
        return np.square(self.distance(d))


class L1Distance(Distance):
    def distance(self, d):  # distance.py:41-45
        return np.sum(np.abs(d), axis=-1)


class L2Distance(Distance):
    def distance(self, d):  # distance.py:51-56 (zero-safe sqrt)
        r1 = np.sum(np.abs(d), axis=-1)
        r2 = self.squared_distance(d)
        zeros = r2 == 0
        r2 = np.where(zeros, np.ones_like(r2), r2)
        return np.where(zeros, r1, np.sqrt(r2))

    def squared_distance(self, d):  # distance.py:58-59
        return np.sum(np.square(d), axis=-1)


# --------------------------------------------------------------------------
# kernels/base.py
# --------------------------------------------------------------------------
class Kernel:
    """``evaluate`` takes broadcastable point arrays with the feature axis last
    (the numpy spelling of the reference's nested ``vmap``, base.py:94-96)."""

    def evaluate(self, X1, X2):
        raise NotImplementedError

    def evaluate_diag(self, X):  # base.py:59-66
        return self.evaluate(X, X)

    def __call__(self, X1, X2=None, *, block=2048):  # base.py:84-103
        P1 = _as_points(X1)
        if X2 is None:
            k = np.asarray(self.evaluate_diag(P1))
            k = np.broadcast_to(k, (P1.shape[0],)) if k.ndim == 0 else k
            if k.ndim != 1:
                raise ValueError("Invalid kernel diagonal shape")
            return np.array(k)
        P2 = _as_points(X2)
        out = np.empty((P1.shape[0], P2.shape[0]), dtype=np.result_type(P1, P2))
        block = max(1, min(block, (1 << 24) // max(1, P2.shape[0] * P2.shape[1])))
        for i0 in range(0, P1.shape[0], block):  # blockwise: no N^2 temporaries
            blk = self.evaluate(P1[i0:i0 + block, None, :], P2[None, :, :])
            out[i0:i0 + block] = np.broadcast_to(blk, out[i0:i0 + block].shape)
        return out

    def matmul(self, X1, X2=None, y=None):  # base.py:68-82
        if y is None:
            assert X2 is not None
            y = X2
            X2 = None
        if X2 is None:
            X2 = X1
        return np.dot(self(X1, X2), y)

    # operator overloads, base.py:105-126
    def __add__(self, other):
        return Sum(self, other if isinstance(other, Kernel) else Constant(other))

    def __radd__(self, other):
        if not isinstance(other, Kernel) and np.ndim(other) == 0 and other == 0:
            return self
        return Sum(other if isinstance(other, Kernel) else Constant(other), self)

    def __mul__(self, other):
        return Product(self, other if isinstance(other, Kernel) else Constant(other))

    def __rmul__(self, other):
        return Product(other if isinstance(other, Kernel) else Constant(other), self)


class Sum(Kernel):  # base.py:170-177
    def __init__(self, kernel1, kernel2):
        self.kernel1, self.kernel2 = kernel1, kernel2

    def evaluate(self, X1, X2):
        return self.kernel1.evaluate(X1, X2) + self.kernel2.evaluate(X1, X2)


class Product(Kernel):  # base.py:180-187
    def __init__(self, kernel1, kernel2):
        self.kernel1, self.kernel2 = kernel1, kernel2

    def evaluate(self, X1, X2):
        return self.kernel1.evaluate(X1, X2) * self.kernel2.evaluate(X1, X2)


class Constant(Kernel):  # base.py:190-209
    def __init__(self, value):
        self.value = value

    def evaluate(self, X1, X2):
        if np.ndim(self.value) != 0:
            raise ValueError("The value of a constant kernel must be a scalar")
        return np.asarray(self.value, dtype=np.result_type(np.asarray(X1).dtype, np.float32))


# --------------------------------------------------------------------------
# kernels/stationary.py
# --------------------------------------------------------------------------
class Stationary(Kernel):  # stationary.py:38-56 (default metric L1, scale 1)
    default_distance = L1Distance

    def __init__(self, scale=1.0, distance=None):
        self.scale = scale
        self.distance = self.default_distance() if distance is None else distance


class Exp(Stationary):  # stationary.py:76-82
    def evaluate(self, X1, X2):
        if np.ndim(self.scale):
            raise ValueError("Only scalar scales are permitted for stationary kernels")
        d = X1 - X2
        return np.exp(-self.distance.distance(d) / d.dtype.type(self.scale))


class ExpSquared(Stationary):  # stationary.py:102-106 (default metric L2)
    default_distance = L2Distance

    def evaluate(self, X1, X2):
        d = X1 - X2
        r2 = self.distance.squared_distance(d) / np.square(d.dtype.type(self.scale))
        return np.exp(d.dtype.type(-0.5) * r2)


class Matern32(Stationary):  # stationary.py:126-129
    def evaluate(self, X1, X2):
        d = X1 - X2
        r = self.distance.distance(d) / d.dtype.type(self.scale)
        arg = r.dtype.type(np.sqrt(3)) * r  # np.sqrt(3): an fp64 Python-side constant
        return (1 + arg) * np.exp(-arg)


class Matern52(Stationary):  # stationary.py:150-153
    def evaluate(self, X1, X2):
        d = X1 - X2
        r = self.distance.distance(d) / d.dtype.type(self.scale)
        arg = r.dtype.type(np.sqrt(5)) * r
        return (1 + arg + np.square(arg) / 3) * np.exp(-arg)


class Cosine(Stationary):  # stationary.py:173-175
    def evaluate(self, X1, X2):
        d = X1 - X2
        r = self.distance.distance(d) / d.dtype.type(self.scale)
        return np.cos(r.dtype.type(2 * np.pi) * r)


class ExpSineSquared(Stationary):  # stationary.py:196-205
    def __init__(self, scale=1.0, distance=None, *, gamma=None):
        super().__init__(scale, distance)
        if gamma is None:
            raise ValueError("Missing required argument 'gamma'")
        self.gamma = gamma

    def evaluate(self, X1, X2):
        d = X1 - X2
        r = self.distance.distance(d) / d.dtype.type(self.scale)
        t = r.dtype.type
        return np.exp(-t(self.gamma) * np.square(np.sin(t(np.pi) * r)))


class RationalQuadratic(Stationary):  # stationary.py:226-235 (inherits L1 default!)
    def __init__(self, scale=1.0, distance=None, *, alpha=None):
        super().__init__(scale, distance)
        if alpha is None:
            raise ValueError("Missing required argument 'alpha'")
        self.alpha = alpha

    def evaluate(self, X1, X2):
        d = X1 - X2
        r2 = self.distance.squared_distance(d) / np.square(d.dtype.type(self.scale))
        t = r2.dtype.type
        return (t(1.0) + t(0.5) * r2 / t(self.alpha)) ** -t(self.alpha)


# --------------------------------------------------------------------------
# noise.py
# --------------------------------------------------------------------------
class Diagonal:  # noise.py:55-95
    __array_priority__ = 2001

    def __init__(self, diag):
        if np.ndim(diag) != 1:
            raise ValueError("The diagonal for the noise model be the same shape as the data")
        self.diag = np.asarray(diag)

    def diagonal(self):
        return self.diag

    def _add(self, other):  # noise.py:77-78
        out = np.array(other)
        idx = np.diag_indices(out.shape[0])
        out[idx] += self.diag
        return out

    __add__ = __radd__ = lambda self, other: self._add(other)

    def __matmul__(self, other):  # noise.py:86-90
        if np.ndim(other) == 1:
            return self.diag * other
        return self.diag[:, None] * other


class Dense:  # noise.py:98-124
    __array_priority__ = 2001

    def __init__(self, value):
        self.value = np.asarray(value)

    def diagonal(self):
        return np.diag(self.value)

    def __add__(self, other):
        return self.value + other

    def __radd__(self, other):
        return other + self.value

    def __matmul__(self, other):
        return self.value @ other


# --------------------------------------------------------------------------
# solvers/direct.py  (the O(N^3) step: LAPACK potrf / trtrs through SciPy)
# --------------------------------------------------------------------------
def _cholesky_lower(K):
    """``jax.scipy.linalg.cholesky(K, lower=True)`` (direct.py:53): NaNs, not an
    exception, on a non positive-definite input; upper triangle zeroed."""
    try:
        return np.tril(sla.cholesky(K, lower=True, check_finite=False))
    except sla.LinAlgError:
        return np.full_like(K, np.nan)


class DirectSolver:
    def __init__(self, kernel, X, noise, *, covariance=None):  # direct.py:30-53
        self.X = X
        self.variance_value = kernel(X) + noise.diagonal()
        if covariance is None:
            covariance = kernel(X, X) + noise
        self.covariance_value = covariance
        self.scale_tril = _cholesky_lower(np.asarray(covariance))

    @classmethod
    def init(cls, kernel, X, noise, *, covariance=None):  # solver.py:29-38
        return cls(kernel, X, noise, covariance=covariance)

    def variance(self):  # direct.py:55-56
        return self.variance_value

    def covariance(self):  # direct.py:58-59
        return self.covariance_value

    def normalization(self):  # direct.py:61-64
        L = self.scale_tril
        return np.sum(np.log(np.diag(L))) + 0.5 * L.shape[0] * np.log(2 * np.pi)

    def solve_triangular(self, y, *, transpose=False):  # direct.py:66-70
        if not np.all(np.isfinite(np.diag(self.scale_tril))):
            return np.full(np.shape(y), np.nan)
        return sla.solve_triangular(self.scale_tril, y, lower=True,
                                    trans=1 if transpose else 0, check_finite=False)

    def dot_triangular(self, y):  # direct.py:72-73
        return np.einsum("ij,j...->i...", self.scale_tril, y)

    def condition(self, kernel, X_test, noise):  # direct.py:75-95
        if X_test is None:
            Ks = kernel(self.X, self.X)
            Kss = Ks + noise
        else:
            Ks = kernel(self.X, X_test)
            Kss = kernel(X_test, X_test) + noise
        A = self.solve_triangular(Ks)
        return Kss - A.transpose() @ A


class Conditioned(Kernel):  # kernels/base.py:129-153 (matrix form of the vmap)
    def __init__(self, X, solver, kernel):
        self.X, self.solver, self.kernel = X, solver, kernel

    def __call__(self, X1, X2=None):
        if X2 is None:
            K = self.solver.solve_triangular(self.kernel(self.X, X1))
            return self.kernel(X1) - np.sum(K * K, axis=0)
        K1 = self.solver.solve_triangular(self.kernel(self.X, X1))
        K2 = self.solver.solve_triangular(self.kernel(self.X, X2))
        return self.kernel(X1, X2) - K1.transpose() @ K2


# --------------------------------------------------------------------------
# means.py
# --------------------------------------------------------------------------
def _eval_mean(mean, X):  # means.py:31-55 + the vmap at gp.py:86-87
    n = _as_points(X).shape[0]
    if mean is None:
        return np.zeros(n, dtype=np.result_type(np.asarray(X).dtype, np.float32))
    if callable(mean):
        return np.asarray([mean(x) for x in np.asarray(X)])
    return np.broadcast_to(np.asarray(mean, dtype=np.asarray(X).dtype), (n,)).copy()


class ConditionedMean:  # means.py:58-86 (`means.Conditioned`): k(x, X) . alpha (+ mean(x))
    def __init__(self, X, alpha, kernel, include_mean, mean_function=None):
        self.X, self.alpha, self.kernel = X, alpha, kernel
        self.include_mean, self.mean_function = include_mean, mean_function

    def __call__(self, x):  # ONE point, like the reference (vmapped by its callers)
        xs = np.asarray(x)[None]
        mu = self.kernel(xs, self.X)[0] @ self.alpha  # means.py:81-82
        if self.include_mean and self.mean_function is not None:  # means.py:83-85
            mu = mu + _eval_mean(self.mean_function, xs)[0]
        return mu


# --------------------------------------------------------------------------
# gp.py
# --------------------------------------------------------------------------
def default_diag(reference):  # gp.py:388-393
    return np.sqrt(np.finfo(np.asarray(reference).dtype).eps)


class ConditionResult(tuple):  # gp.py:364-385
    def __new__(cls, log_probability, gp):
        return tuple.__new__(cls, (log_probability, gp))

    log_probability = property(lambda self: self[0])
    gp = property(lambda self: self[1])


class GaussianProcess:
    def __init__(self, kernel, X, *, diag=None, noise=None, mean=None, solver=None,
                 mean_value=None, covariance_value=None, **solver_kwargs):  # gp.py:64-112
        self.kernel = kernel
        self.X = X
        self.mean_function = mean
        if mean_value is None:
            mean_value = _eval_mean(mean, X)
        self.num_data = mean_value.shape[0]
        self.dtype = mean_value.dtype
        self.mean = mean_value
        if self.mean.ndim != 1:
            raise ValueError(f"Invalid mean shape: expected ndim = 1, got ndim={self.mean.ndim}")
        if noise is None:
            diag = default_diag(self.mean) if diag is None else diag
            noise = Diagonal(np.broadcast_to(np.asarray(diag, dtype=self.dtype), self.mean.shape))
        self.noise = noise
        solver = DirectSolver if solver is None else solver
        self.solver = solver(kernel, self.X, self.noise, covariance=covariance_value,
                             **solver_kwargs)

    loc = property(lambda self: self.mean)  # gp.py:114-116
    variance = property(lambda self: self.solver.variance())  # gp.py:118-120
    covariance = property(lambda self: self.solver.covariance())  # gp.py:122-124

    def _get_alpha(self, y):  # gp.py:318-320
        return self.solver.solve_triangular(y - self.loc)

    def _compute_log_prob(self, alpha):  # gp.py:313-316
        loglike = -0.5 * np.sum(np.square(alpha)) - self.solver.normalization()
        return np.where(np.isfinite(loglike), loglike, -np.inf)

    def log_probability(self, y):  # gp.py:126-138
        return self._compute_log_prob(self._get_alpha(y))

    def _condition(self, y, X_test, include_mean, kernel=None):  # gp.py:322-361
        alpha = self._get_alpha(y)
        log_prob = self._compute_log_prob(alpha)
        alpha = self.solver.solve_triangular(alpha, transpose=True)
        if X_test is None:
            X_test = self.X
            if kernel is None:
                delta = self.noise @ alpha
                mean_value = y - delta
                if not include_mean:
                    mean_value = mean_value - self.loc
            else:
                mean_value = kernel.matmul(self.X, y=alpha)
                if include_mean:
                    mean_value = mean_value + self.loc
        else:
            if kernel is None:
                kernel = self.kernel
            mean_value = kernel.matmul(X_test, self.X, alpha)
            if include_mean:
                mean_value = mean_value + _eval_mean(self.mean_function, X_test)
        return alpha, log_prob, mean_value

    def condition(self, y, X_test=None, *, diag=None, noise=None, include_mean=True,
                  kernel=None):  # gp.py:140-223
        if X_test is not None:
            a, b = np.asarray(self.X), np.asarray(X_test)
            if not (a.ndim == b.ndim and a.shape[1:] == b.shape[1:]):
                raise ValueError("`X_test` must have the same tree structure as the input `X`")
        alpha, log_prob, mean_value = self._condition(y, X_test, include_mean, kernel)
        if kernel is None:
            kernel = self.kernel
        if noise is None:
            diag = default_diag(mean_value) if diag is None else diag
            noise = Diagonal(np.broadcast_to(np.asarray(diag, dtype=mean_value.dtype),
                                             mean_value.shape))
        covariance_value = self.solver.condition(kernel, X_test, noise)
        if X_test is None:
            X_test = self.X
        gp = GaussianProcess(Conditioned(self.X, self.solver, kernel), X_test, noise=noise,
                             mean=ConditionedMean(self.X, alpha, kernel, include_mean=include_mean,
                                                  mean_function=self.mean_function),  # gp.py:210-216
                             mean_value=mean_value, covariance_value=covariance_value)
        return ConditionResult(log_prob, gp)

    def predict(self, y, X_test=None, *, kernel=None, include_mean=True, return_var=False,
                return_cov=False):  # gp.py:225-271
        _, cond = self.condition(y, X_test, kernel=kernel, include_mean=include_mean)
        if return_var:
            return cond.loc, cond.variance
        if return_cov:
            return cond.loc, cond.covariance
        return cond.loc


# --------------------------------------------------------------------------
# test_utils.py:9-26 -- the reference's dtype-keyed tolerances
# --------------------------------------------------------------------------
def assert_allclose(calculated, expected, *, atol=None, rtol=None):
    dt = np.result_type(np.asarray(calculated).dtype, np.asarray(expected).dtype)
    tol = 5e-4 if dt == np.float32 else 5e-7
    np.testing.assert_allclose(np.asarray(calculated), np.asarray(expected),
                               atol=tol if atol is None else atol,
                               rtol=tol if rtol is None else rtol)
