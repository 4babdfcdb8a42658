"""Input-space transforms (mirror of ``tinygp.transforms``, reference ``transforms.py:23-162``).

Every transform here is a pure pre-transform of the coordinates, so it is folded on the
host: ``Linear(scale, k)(X1, X2) == k(scale*X1, scale*X2)``.  The device still evaluates the
plain stationary kernel program, on the transformed points.  A general ``Transform`` takes
any callable mapping ONE coordinate to a transformed coordinate, like the reference.
"""

from __future__ import annotations

from collections.abc import Callable
from typing import Any

import numpy as np

from tinygp_amd.kernels.base import Kernel

__all__ = ["Transform", "Linear", "Cholesky", "Subspace"]


class _PreTransform(Kernel):
    """A kernel evaluated on host-transformed coordinates."""

    kernel: Kernel

    def _map_points(self, P: np.ndarray) -> np.ndarray:  # (N, D) -> (N, D')
        raise NotImplementedError

    def _lower(self, X):
        X = np.asarray(X)
        P = X[:, None] if X.ndim == 1 else X
        out = np.asarray(self._map_points(P))
        if out.ndim == 1:
            out = out[:, None]
        return self.kernel._lower(out)

    def _emit(self, ops):
        raise NotImplementedError(
            f"{type(self).__name__} carries an input transform: lower it with _lower(X)")

    def _slots(self, out):  # the inner kernel's parameters; the transform's own: _logscale_gradient
        self.kernel._slots(out)

    def _logscale_gradient(self, glog):
        """Gradient of log_probability with respect to THIS transform's parameter, from the device's
        ``d ll / d log s_q`` per dimension q of the transformed coordinates (``tgp_solver_grad``'s
        ``grad_logscale``).  Only transforms that scale each input dimension by its own factor have one."""
        raise NotImplementedError(f"{type(self).__name__} has no per-dimension scale to differentiate")


def find_transforms(kernel) -> list:
    """Every input transform in a kernel tree (depth first)."""
    out = []
    if isinstance(kernel, _PreTransform):
        out.append(kernel)
        out += find_transforms(kernel.kernel)
    else:
        for name in ("kernel1", "kernel2"):
            sub = getattr(kernel, name, None)
            if sub is not None:
                out += find_transforms(sub)
    return out


def covering_transform(kernel):
    """The ONE input transform that every coordinate-dependent leaf of ``kernel`` sits under, or None.

    The device differentiates with respect to the per-dimension log-scales of the coordinates it is given over EVERY
    leaf of the kernel program (no per-leaf mask).  That is the transform's gradient only if no leaf saw the raw
    coordinates: in ``Sum(Linear(s, k1), k2)`` -- which lowers to one device pass while ``s == 1`` -- k2's share would be
    attributed to ``s`` (advisor r4).  Constant-only siblings are fine: they do not depend on the coordinates."""
    from tinygp_amd.kernels.base import Constant

    def walk(k):  # -> (transforms found, a coordinate-dependent leaf outside any transform?)
        if isinstance(k, _PreTransform):
            return [k], False
        subs = [getattr(k, name, None) for name in ("kernel1", "kernel2")]
        subs = [x for x in subs if x is not None]
        if not subs:
            return [], not isinstance(k, Constant)
        found, bare = [], False
        for x in subs:
            f, b = walk(x)
            found += f
            bare = bare or b
        return found, bare

    found, bare = walk(kernel)
    if len(found) != 1 or bare or find_transforms(found[0].kernel):
        return None
    return found[0]


class Transform(_PreTransform):
    """Apply ``transform`` (one coordinate -> one coordinate) before ``kernel``
    (reference ``transforms.py:23-37``)."""

    def __init__(self, transform: Callable[[Any], Any], kernel: Kernel):
        self.transform, self.kernel = transform, kernel

    def _map_points(self, P):
        return np.stack([np.atleast_1d(np.asarray(self.transform(p if P.shape[1] > 1 else p[0])))
                         for p in P])


class Linear(_PreTransform):
    """``kernel(scale * x)`` / ``kernel(scale @ x)`` (reference ``transforms.py:39-72``)."""

    def __init__(self, scale, kernel: Kernel):
        self.scale, self.kernel = scale, kernel

    def _map_points(self, P):
        s = np.asarray(self.scale)
        if s.ndim < 2:
            return P * s
        if s.ndim == 2:
            return P @ s.T
        raise ValueError("'scale' must be 0-, 1-, or 2-dimensional")

    def _logscale_gradient(self, glog):
        # x'_q = s_q x_q:  d/d s_q = (d/d log s_q) / s_q;  a scalar scale moves every dimension at once
        s = np.asarray(self.scale, dtype=np.float64)
        if s.ndim == 0:
            return float(np.sum(glog) / s)
        if s.ndim == 1:
            return np.asarray(glog, dtype=np.float64) / s
        raise NotImplementedError("gradient with respect to a full (matrix) Linear scale is not available: the "
                                  "device holds the transformed coordinates only")


class Cholesky(_PreTransform):
    """``kernel(L^-1 x)`` for a lower-triangular (or diagonal / scalar) factor
    (reference ``transforms.py:75-133``)."""

    def __init__(self, factor, kernel: Kernel):
        self.factor, self.kernel = factor, kernel

    def _map_points(self, P):
        f = np.asarray(self.factor)
        if f.ndim < 2:
            return P * (1.0 / f)
        if f.ndim == 2:
            import scipy.linalg as sla

            return sla.solve_triangular(f, P.T, lower=True).T
        raise ValueError("'scale' must be 0-, 1-, or 2-dimensional")

    def _logscale_gradient(self, glog):
        # x'_q = x_q / f_q:  log s_q = -log f_q,  d/d f_q = -(d/d log s_q) / f_q
        f = np.asarray(self.factor, dtype=np.float64)
        if f.ndim == 0:
            return float(-np.sum(glog) / f)
        if f.ndim == 1:
            return -np.asarray(glog, dtype=np.float64) / f
        raise NotImplementedError("gradient with respect to a full (matrix) Cholesky factor is not available: the "
                                  "device holds the transformed coordinates only")

    @classmethod
    def from_parameters(cls, diagonal, off_diagonal, kernel: Kernel) -> "Cholesky":
        diagonal, off_diagonal = np.asarray(diagonal), np.asarray(off_diagonal)
        ndim = diagonal.size
        if off_diagonal.size != ((ndim - 1) * ndim) // 2:
            raise ValueError(
                "Dimension mismatch: expected "
                f"(ndim-1)*ndim/2 = {((ndim - 1) * ndim) // 2} elements in "
                f"'off_diagonal'; got {off_diagonal.size}")
        factor = np.zeros((ndim, ndim))
        factor[np.diag_indices(ndim)] += diagonal
        factor[np.tril_indices(ndim, -1)] += off_diagonal
        return cls(factor, kernel)


class Subspace(_PreTransform):
    """``kernel(x[axis])`` (reference ``transforms.py:136-162``)."""

    def __init__(self, axis, kernel: Kernel):
        self.axis, self.kernel = axis, kernel

    def _map_points(self, P):
        return P[:, self.axis]
