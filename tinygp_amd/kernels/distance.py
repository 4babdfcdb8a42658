"""Distance metrics for stationary kernels (mirror of ``tinygp.kernels.distance``).

On the device the metric is just a tag in the kernel program (``TGP_METRIC_L1`` /
``TGP_METRIC_L2``); the two methods below are the reference's scalar protocol
(reference ``kernels/distance.py:22-59``), kept so user code that calls a metric
directly keeps working.  They operate on ONE pair of coordinates.
"""

from __future__ import annotations

import numpy as np

__all__ = ["Distance", "L1Distance", "L2Distance"]

METRIC_L1, METRIC_L2 = 0, 1


class Distance:
    """Abstract metric.  Only :class:`L1Distance` / :class:`L2Distance` can be
    lowered to the HIP kernel evaluator; a custom subclass raises at compile time."""

    metric_code: int | None = None

    def distance(self, X1, X2):
        raise NotImplementedError()

    def squared_distance(self, X1, X2):
        # reference distance.py:30-38: default = distance**2
        return np.square(self.distance(X1, X2))

    def __eq__(self, other):
        return type(self) is type(other)

    def __hash__(self):
        return hash(type(self))

    def __repr__(self):
        return f"{type(self).__name__}()"


class L1Distance(Distance):
    """Manhattan distance, reference ``distance.py:41-45``."""

    metric_code = METRIC_L1

    def distance(self, X1, X2):
        return np.sum(np.abs(np.asarray(X1) - np.asarray(X2)))


class L2Distance(Distance):
    """Euclidean distance with the zero-safe square root of reference
    ``distance.py:48-59`` (finite gradients at r = 0 in the reference)."""

    metric_code = METRIC_L2

    def distance(self, X1, X2):
        r2 = self.squared_distance(X1, X2)
        if r2 == 0:
            return L1Distance().distance(X1, X2)
        return np.sqrt(r2)

    def squared_distance(self, X1, X2):
        return np.sum(np.square(np.asarray(X1) - np.asarray(X2)))
