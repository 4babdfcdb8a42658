"""``GaussianProcess`` front-end (mirror of reference ``gp.py:30-393`` for the dense path).

Same constructor, properties, ``log_probability`` / ``condition`` / ``predict`` and error
conventions as the reference; every O(N^2) / O(N^3) step is delegated to the solver, which
runs it on the MI355X.  Two deliberate departures, both invisible in the results:

* ``log_probability`` uses the solver's fused solve+reduce entry point when it has one
  (the reference's two ``@jax.jit`` stages ``gp.py:313-320``);
* the GP returned by ``condition`` builds its covariance / M x M factor lazily -- the
  reference computes them eagerly (``gp.py:201-221``) and relies on XLA dead-code
  elimination when the caller only reads ``.loc``.

``sample`` draws its standard normals with NumPy on the host (JAX's threefry stream cannot be
reproduced bit for bit; moments match) and multiplies by L on the device;
``numpyro_dist`` is adjacent to the hot path and not provided.
"""

from __future__ import annotations

from collections.abc import Callable
from typing import Any, NamedTuple

import numpy as np

from tinygp_amd import _device, kernels, means
from tinygp_amd.noise import Diagonal, Noise
from tinygp_amd.solvers import DirectSolver

__all__ = ["GaussianProcess", "ConditionResult"]


def _default_diag(reference: np.ndarray):
    """sqrt(eps) of the mean's dtype (reference ``gp.py:388-393``)."""
    return np.sqrt(np.finfo(np.asarray(reference).dtype).eps)


class GaussianProcess:
    """Interface for designing a Gaussian Process regression model.

    Args:
        kernel: the kernel function.
        X: input coordinates, shape (N,) or (N, D).
        diag: value(s) added to the diagonal of the covariance (scalar or (N,)); defaults to
            ``sqrt(eps)`` like the reference.
        noise: a :mod:`tinygp_amd.noise` model (overrides ``diag``).
        mean: a scalar, a callable of one coordinate, or a :class:`means.MeanBase`.
        solver: the solver class, default :class:`solvers.DirectSolver`.
        mean_value / covariance_value: pre-computed mean vector / covariance matrix.
        **solver_kwargs: forwarded to the solver constructor (e.g. ``ctx=``).
    """

    def __init__(self, kernel: kernels.Kernel, X, *, diag=None, noise: Noise | None = None,
                 mean: means.MeanBase | Callable | Any | None = None, solver: Any | None = None,
                 mean_value=None, covariance_value: Any | None = None, _lazy: bool = False,
                 **solver_kwargs: Any):
        self.kernel = kernel
        self.X = X

        if isinstance(mean, means.MeanBase):
            self.mean_function = mean
        elif mean is None:
            self.mean_function = means.Mean(np.zeros(()))
        else:
            self.mean_function = means.Mean(mean)

        if mean_value is None:
            if _device.is_tree(X):  # pytree input (gp.py:64-112): host-evaluated kernels only
                n_pts = _device.num_points(X)
                dt = _device.common_dtype(*_device.tree_leaves(X))
            else:
                P = _device.points(X, limit=False)
                n_pts, dt = P.shape[0], P.dtype
            mean_value = means.evaluate_mean(self.mean_function, X, n_pts, dt)
        mean_value = np.asarray(mean_value)
        if mean_value.dtype.kind != "f":
            mean_value = mean_value.astype(np.float64)
        self.num_data = mean_value.shape[0] if mean_value.ndim else 0
        self.dtype = mean_value.dtype
        self.mean = mean_value
        if self.mean.ndim != 1:
            raise ValueError(f"Invalid mean shape: expected ndim = 1, got ndim={self.mean.ndim}")

        if noise is None:
            diag = _default_diag(self.mean) if diag is None else diag
            noise = Diagonal(diag=np.broadcast_to(np.asarray(diag, dtype=self.dtype),
                                                  self.mean.shape))
        self.noise = noise

        self._solver_cls = DirectSolver if solver is None else solver
        self._solver_args = (covariance_value, solver_kwargs)
        self._solver = None
        if not _lazy:
            _ = self.solver

    @property
    def solver(self):
        if self._solver is None:
            covariance_value, solver_kwargs = self._solver_args
            if callable(covariance_value):  # lazy conditional covariance
                covariance_value = covariance_value()
            self._solver = self._solver_cls(self.kernel, self.X, self.noise,
                                            covariance=covariance_value, **solver_kwargs)
        return self._solver

    # -- reference gp.py:114-124 ------------------------------------------------------
    @property
    def loc(self):
        return self.mean

    @property
    def variance(self):
        if self._solver is None and isinstance(self.kernel, kernels.Conditioned):
            # conditioned GP: diag(Kss - A^T A) + noise without the M x M factorisation
            return self.kernel(self.X) + np.asarray(self.noise.diagonal())
        return self.solver.variance()

    @property
    def covariance(self):
        if self._solver is None and callable(self._solver_args[0]):
            cov = self._solver_args[0]()
            self._solver_args = (cov, self._solver_args[1])
            return cov
        return self.solver.covariance()

    # -- reference gp.py:273-311 ---------------------------------------------------------
    def sample(self, key, shape=None):
        """Draw samples from the prior: ``mean + L z`` (reference ``gp.py:273-311``).

        ``key``: an ``int`` seed or a ``numpy.random.Generator`` (the reference takes a JAX
        PRNG key; the random stream necessarily differs, the distribution does not).
        Returns shape ``shape + (N,)`` like the reference.
        """
        rng = key if isinstance(key, np.random.Generator) else np.random.default_rng(key)
        full = (self.num_data,) if shape is None else (self.num_data,) + tuple(shape)
        z = rng.standard_normal(full).astype(self.dtype, copy=False)
        out = self.solver.dot_triangular(z)
        return self.mean + np.moveaxis(out, 0, -1)

    # -- reference gp.py:126-138, 313-320 ---------------------------------------------
    def log_probability(self, y):
        """Marginal log-probability of ``y`` under this multivariate normal."""
        resid = self._residual(y)
        fused = getattr(self.solver, "log_probability", None)
        if fused is not None:
            return fused(resid)
        return self._compute_log_prob(self.solver.solve_triangular(resid))

    def log_probability_and_grad(self, y):
        """``(log_probability, grads)``; see :meth:`solvers.DirectSolver.log_probability_and_grad`.
        ``grads["kernel"]`` follows ``self.kernel.parameters()``."""
        return self.solver.log_probability_and_grad(self._residual(y))

    def _residual(self, y):
        y = np.asarray(y)
        try:
            return np.broadcast_to(y - self.loc, self.loc.shape).astype(self.dtype, copy=False)
        except ValueError as e:
            raise ValueError(f"y must broadcast against the mean of shape {self.loc.shape}") from e

    def _get_alpha(self, y):
        return self.solver.solve_triangular(self._residual(y))

    def _compute_log_prob(self, alpha):
        loglike = -0.5 * np.sum(np.square(alpha)) - self.solver.normalization()
        return self.dtype.type(loglike if np.isfinite(loglike) else -np.inf)

    # -- reference gp.py:140-223, 322-361 ---------------------------------------------
    def _condition(self, y, X_test, include_mean: bool, kernel=None):
        resid = self._residual(y)
        two_solves = getattr(self.solver, "alpha", None)
        if two_solves is not None:
            alpha, log_prob = two_solves(resid)
        else:
            a = self.solver.solve_triangular(resid)
            log_prob = self._compute_log_prob(a)
            alpha = self.solver.solve_triangular(a, transpose=True)

        if X_test is None:
            if kernel is None:
                # predicting at the data with the original kernel: O(N) (gp.py:342-346)
                delta = self.noise @ alpha
                mean_value = np.asarray(y) - delta
                if not include_mean:
                    mean_value = mean_value - self.loc
            else:
                mean_value = self._kernel_matvec(kernel, self.X, alpha)
                if include_mean:
                    mean_value = mean_value + self.loc
        else:
            if kernel is None:
                kernel = self.kernel
            mean_value = self._kernel_matvec(kernel, X_test, alpha)
            if include_mean:
                mean_value = mean_value + means.evaluate_mean(self.mean_function, X_test,
                                                              _device.num_points(X_test), self.dtype)
        return alpha, log_prob, np.asarray(mean_value, dtype=self.dtype)

    def _kernel_matvec(self, kernel, X_out, alpha):
        fused = getattr(self.solver, "conditional_mean", None)
        if fused is not None:
            return fused(kernel, X_out, alpha)  # X resident on the device
        return kernel.matmul(X_out, self.X, alpha)

    def condition(self, y, X_test=None, *, diag=None, noise: Noise | None = None,
                  include_mean: bool = True, kernel: kernels.Kernel | None = None):
        """Condition the model on observed data ``y``; returns ``ConditionResult``."""
        if X_test is not None:
            if _device.is_tree(self.X) or _device.is_tree(X_test):
                same = _device.tree_structure(self.X) == _device.tree_structure(X_test)
            else:
                a, b = np.asarray(self.X), np.asarray(X_test)
                same = a.ndim == b.ndim and a.shape[1:] == b.shape[1:]
            if not same:
                raise ValueError(
                    "`X_test` must have the same tree structure as the input `X`, "
                    "and all but the leading dimension must have matching sizes")

        alpha, log_prob, mean_value = self._condition(y, X_test, include_mean, kernel)
        if kernel is None:
            kernel = self.kernel

        if noise is None:
            diag = _default_diag(mean_value) if diag is None else diag
            noise = Diagonal(diag=np.broadcast_to(np.asarray(diag, dtype=self.dtype),
                                                  mean_value.shape))

        solver, Xt = self.solver, X_test
        covariance_value = lambda: solver.condition(kernel, Xt, noise)  # noqa: E731 (lazy)
        if X_test is None:
            X_test = self.X

        gp = GaussianProcess(
            kernels.Conditioned(self.X, self.solver, kernel), X_test, noise=noise,
            mean=means.Conditioned(self.X, alpha, kernel, include_mean=include_mean,
                                   mean_function=self.mean_function),
            mean_value=mean_value, covariance_value=covariance_value, _lazy=True)
        return ConditionResult(log_prob, gp)

    def predict(self, y, X_test=None, *, kernel: kernels.Kernel | None = None,
                include_mean: bool = True, return_var: bool = False, return_cov: bool = False):
        """Reference ``gp.py:225-271``."""
        _, cond = self.condition(y, X_test, kernel=kernel, include_mean=include_mean)
        if return_var:
            return cond.loc, cond.variance
        if return_cov:
            return cond.loc, cond.covariance
        return cond.loc


class ConditionResult(NamedTuple):
    """``(log_probability, gp)`` (reference ``gp.py:364-385``)."""

    log_probability: Any
    gp: GaussianProcess
