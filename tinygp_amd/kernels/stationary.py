"""Stationary kernels (mirror of ``tinygp.kernels.stationary``).

Each class is one leaf op of the device kernel program; the formulas (and the order
of floating-point operations the HIP evaluator uses) are those of reference
``kernels/stationary.py:59-235``.  As in the reference, every stationary kernel
defaults to the **L1** metric except :class:`ExpSquared`, which defaults to L2 --
including :class:`RationalQuadratic`, which inherits L1 and therefore evaluates
``(sum|d|)^2`` in more than one dimension (reference ``stationary.py:56,234``).
"""

from __future__ import annotations

import numpy as np

from tinygp_amd.kernels import base
from tinygp_amd.kernels.distance import Distance, L1Distance, L2Distance

__all__ = ["Stationary", "Exp", "ExpSquared", "Matern32", "Matern52", "Cosine",
           "ExpSineSquared", "RationalQuadratic"]


class Stationary(base.Kernel):
    """Isotropic kernel of a scalar distance (reference ``stationary.py:38-56``).

    Args:
        scale: the length scale; must be a scalar.
        distance: :class:`L1Distance` or :class:`L2Distance`.
    """

    _op: int = -1
    _default_distance: type = L1Distance

    def __init__(self, scale=1.0, distance: Distance | None = None):
        self.scale = scale
        self.distance = self._default_distance() if distance is None else distance

    _extra_name: str | None = None

    def _extra(self) -> float:
        return 0.0

    def _slots(self, out):
        out.append([(self, "scale"), (self, self._extra_name) if self._extra_name else None])

    def _emit(self, ops):
        if np.ndim(self.scale) != 0:
            raise ValueError(
                "Only scalar scales are permitted for stationary kernels; pre-scale the "
                "inputs for anisotropic length scales")
        code = getattr(self.distance, "metric_code", None)
        if code is None or type(self.distance) not in (L1Distance, L2Distance):
            raise NotImplementedError(
                "only L1Distance and L2Distance run in the HIP kernel evaluator; evaluate a "
                "custom metric on the host and pass covariance_value=")
        ops.append((self._op, code, float(self.scale), float(self._extra())))

    def __repr__(self):
        return f"{type(self).__name__}(scale={self.scale!r}, distance={self.distance!r})"

    # -- host route (inputs beyond the device evaluator's limits: D > 16, over-long trees) -------------------
    # The reference's own formulas (stationary.py:76-235) in NumPy, blocked over rows so that the
    # (n1, n2, D) difference tensor stays small.  Reached only through tinygp_amd._device.DeviceLimit.
    def _of_r(self, r, r2):
        """Kernel value from the scaled distance ``r`` / scaled squared distance ``r2`` (whichever the
        reference's evaluate() uses; the other one is None)."""
        raise NotImplementedError

    _uses_squared = False

    def _host_pairs(self, A, B):
        """(n1, n2) kernel values for point arrays A (n1, D), B (n2, D)."""
        if np.ndim(self.scale) != 0:
            raise ValueError(
                "Only scalar scales are permitted for stationary kernels; pre-scale the "
                "inputs for anisotropic length scales")
        dt = np.result_type(A, B)
        out = np.empty((A.shape[0], B.shape[0]), dtype=dt)
        scale = dt.type(self.scale)
        known = type(self.distance) in (L1Distance, L2Distance)
        step = max(1, (1 << 22) // max(1, B.shape[0] * B.shape[1]))
        for i0 in range(0, A.shape[0], step):
            a = A[i0:i0 + step]
            if known:
                d = a[:, None, :] - B[None, :, :]
                if isinstance(self.distance, L1Distance):  # distance.py:41-45; squared: the base-class square (:30-38)
                    r = np.sum(np.abs(d), axis=-1)
                    r2 = np.square(r)
                else:  # distance.py:48-59 (zero-safe square root)
                    r2 = np.sum(np.square(d), axis=-1)
                    zeros = r2 == 0
                    r = np.where(zeros, np.sum(np.abs(d), axis=-1), np.sqrt(np.where(zeros, np.ones_like(r2), r2)))
            else:  # a user-defined metric: its scalar protocol, pair by pair
                r = np.array([[self.distance.distance(x, y) for y in B] for x in a], dtype=dt)
                r2 = np.array([[self.distance.squared_distance(x, y) for y in B] for x in a], dtype=dt)
            if self._uses_squared:
                out[i0:i0 + step] = self._of_r(None, r2 / np.square(scale))
            else:
                out[i0:i0 + step] = self._of_r(r / scale, None)
        return out

    def _host_matrix(self, X1, X2):
        dt = _device_common(X1, X2)
        return self._host_pairs(_as_points(X1, dt), _as_points(X2, dt))

    def _host_diag(self, X):
        """k(x, x) per point, like the reference's evaluate_diag -> evaluate(x, x) (base.py:59-66): the same
        scale validation as ``_host_pairs`` and, for a user-defined metric, the metric's OWN value at (x, x)."""
        if np.ndim(self.scale) != 0:
            raise ValueError(
                "Only scalar scales are permitted for stationary kernels; pre-scale the "
                "inputs for anisotropic length scales")
        P = _as_points(X, _device_common(X))
        if type(self.distance) in (L1Distance, L2Distance):  # distance 0 at (x, x) for both built-in metrics
            zero = np.zeros((1, 1), dtype=P.dtype)
            v = self._of_r(None, zero)[0, 0] if self._uses_squared else self._of_r(zero, None)[0, 0]
            return np.full((P.shape[0],), v, dtype=P.dtype)
        scale = P.dtype.type(self.scale)
        if self._uses_squared:
            r2 = np.array([self.distance.squared_distance(x, x) for x in P], dtype=P.dtype)[:, None]
            return self._of_r(None, r2 / np.square(scale))[:, 0]
        r = np.array([self.distance.distance(x, x) for x in P], dtype=P.dtype)[:, None]
        return self._of_r(r / scale, None)[:, 0]


def _device_common(*arrays):
    from tinygp_amd import _device

    return _device.common_dtype(*[np.asarray(a) for a in arrays])


def _as_points(X, dt):
    from tinygp_amd import _device

    return _device.points(X, dt, limit=False)


class Exp(Stationary):
    """exp(-r), reference ``stationary.py:59-82``."""

    _op = base.K_EXP

    def _of_r(self, r, r2):  # stationary.py:76-82
        return np.exp(-r)


class ExpSquared(Stationary):
    """exp(-r^2/2) with the L2 metric by default, reference ``stationary.py:85-106``."""

    _op = base.K_EXPSQ
    _default_distance = L2Distance
    _uses_squared = True

    def _of_r(self, r, r2):  # stationary.py:104-106
        return np.exp(r2.dtype.type(-0.5) * r2)


class Matern32(Stationary):
    """(1 + sqrt3 r) exp(-sqrt3 r), reference ``stationary.py:109-129``."""

    _op = base.K_M32

    def _of_r(self, r, r2):  # stationary.py:126-129 (np.sqrt(3): an fp64 constant rounded to the inputs' dtype)
        arg = r.dtype.type(np.sqrt(3)) * r
        return (1 + arg) * np.exp(-arg)


class Matern52(Stationary):
    """(1 + sqrt5 r + 5 r^2/3) exp(-sqrt5 r), reference ``stationary.py:132-153``."""

    _op = base.K_M52

    def _of_r(self, r, r2):  # stationary.py:150-153
        arg = r.dtype.type(np.sqrt(5)) * r
        return (1 + arg + np.square(arg) / 3) * np.exp(-arg)


class Cosine(Stationary):
    """cos(2 pi r), reference ``stationary.py:156-175``."""

    _op = base.K_COS

    def _of_r(self, r, r2):  # stationary.py:173-175
        return np.cos(r.dtype.type(2 * np.pi) * r)


class ExpSineSquared(Stationary):
    """exp(-Gamma sin^2(pi r)), reference ``stationary.py:178-205``; ``gamma`` is required."""

    _op = base.K_ESS
    _extra_name = "gamma"

    def __init__(self, scale=1.0, distance: Distance | None = None, gamma=None):
        super().__init__(scale, distance)
        if gamma is None:
            raise ValueError("Missing required argument 'gamma'")
        self.gamma = gamma

    def _extra(self):
        if np.ndim(self.gamma) != 0:
            raise ValueError("'gamma' must be a scalar")
        return float(self.gamma)

    def _of_r(self, r, r2):  # stationary.py:202-205
        t = r.dtype.type
        return np.exp(-t(self._extra()) * np.square(np.sin(t(np.pi) * r)))


class RationalQuadratic(Stationary):
    """(1 + r^2 / 2 alpha)^-alpha, reference ``stationary.py:208-235``; ``alpha`` is required."""

    _op = base.K_RQ
    _extra_name = "alpha"

    def __init__(self, scale=1.0, distance: Distance | None = None, alpha=None):
        super().__init__(scale, distance)
        if alpha is None:
            raise ValueError("Missing required argument 'alpha'")
        self.alpha = alpha

    _uses_squared = True

    def _extra(self):
        if np.ndim(self.alpha) != 0:
            raise ValueError("'alpha' must be a scalar")
        return float(self.alpha)

    def _of_r(self, r, r2):  # stationary.py:232-235
        t = r2.dtype.type
        return (t(1.0) + t(0.5) * r2 / t(self._extra())) ** -t(self._extra())
