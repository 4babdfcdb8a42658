"""Mean functions (mirror of ``tinygp.means``): O(N) host-side helpers."""

from __future__ import annotations

from collections.abc import Callable

import numpy as np

__all__ = ["MeanBase", "Mean", "Conditioned"]


class MeanBase:
    def __call__(self, X):
        raise NotImplementedError


class Mean(MeanBase):
    """A scalar constant, or a callable taking ONE coordinate and returning the scalar mean
    there (reference ``means.py:31-55``)."""

    def __init__(self, value):
        if callable(value):
            self.func: Callable | None = value
            self.value = np.zeros(())
        else:
            self.func = None
            self.value = value

    def __call__(self, X):
        if self.func is not None:
            return self.func(X)
        return self.value


def evaluate_mean(mean: MeanBase, X, n: int, dtype) -> np.ndarray:
    """The reference's ``jax.vmap(mean_function)(X)`` (``gp.py:86-87``)."""
    if isinstance(mean, Mean) and mean.func is None:
        return np.broadcast_to(np.asarray(mean.value, dtype=dtype), (n,) + np.shape(mean.value)).copy()
    if isinstance(mean, Conditioned):
        return mean.batch(X)
    from tinygp_amd import _device

    vals = [mean(x) for x in _device.iter_points(X)]
    return np.asarray(vals, dtype=np.result_type(dtype, np.asarray(vals).dtype) if vals else dtype)


class Conditioned(MeanBase):
    """Mean of a process conditioned on data (reference ``means.py:58-86``):
    ``k(x, X) . alpha (+ mean(x))``; ``alpha = K^-1 (y - mu)``."""

    def __init__(self, X, alpha, kernel, include_mean: bool, mean_function: MeanBase | None = None):
        self.X, self.alpha, self.kernel = X, alpha, kernel
        self.include_mean, self.mean_function = include_mean, mean_function

    def __call__(self, X):
        return self.batch(np.asarray(X)[None])[0]

    def batch(self, Xs):
        mu = self.kernel.matmul(Xs, self.X, self.alpha)
        if self.include_mean and self.mean_function is not None:
            mu = mu + evaluate_mean(self.mean_function, Xs, mu.shape[0], mu.dtype)
        return mu
