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


class Exp(Stationary):
    """exp(-r), reference ``stationary.py:59-82``."""

    _op = base.K_EXP


class ExpSquared(Stationary):
    """exp(-r^2/2) with the L2 metric by default, reference ``stationary.py:85-106``."""

    _op = base.K_EXPSQ
    _default_distance = L2Distance


class Matern32(Stationary):
    """(1 + sqrt3 r) exp(-sqrt3 r), reference ``stationary.py:109-129``."""

    _op = base.K_M32


class Matern52(Stationary):
    """(1 + sqrt5 r + 5 r^2/3) exp(-sqrt5 r), reference ``stationary.py:132-153``."""

    _op = base.K_M52


class Cosine(Stationary):
    """cos(2 pi r), reference ``stationary.py:156-175``."""

    _op = base.K_COS


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


class RationalQuadratic(Stationary):
    """(1 + r^2 / 2 alpha)^-alpha, reference ``stationary.py:208-235``; ``alpha`` is required."""

    _op = base.K_RQ
    _extra_name = "alpha"

    def __init__(self, scale=1.0, distance: Distance | None = None, alpha=None):
        super().__init__(scale, distance)
        if alpha is None:
            raise ValueError("Missing required argument 'alpha'")
        self.alpha = alpha

    def _extra(self):
        if np.ndim(self.alpha) != 0:
            raise ValueError("'alpha' must be a scalar")
        return float(self.alpha)
