"""Observation-noise models (mirror of ``tinygp.noise`` for the dense path).

Noise objects are O(N) host-side value objects.  On the hot path only
``Diagonal.diagonal()`` matters: the solver uploads it once and the HIP assembly kernel
adds it to the diagonal while writing K (the fused form of reference
``noise.py:77-78``).  The array-facing operators are kept for API parity.
"""

from __future__ import annotations

import numpy as np

__all__ = ["Noise", "Diagonal", "Dense"]


class Noise:
    """Protocol of reference ``noise.py:27-52``."""

    __array_priority__ = 2001  # so that `ndarray + noise` dispatches to __radd__

    def diagonal(self):
        raise NotImplementedError

    def __add__(self, other):
        raise NotImplementedError

    def __radd__(self, other):
        raise NotImplementedError

    def __matmul__(self, other):
        raise NotImplementedError


class Diagonal(Noise):
    """Per-observation variances (reference ``noise.py:55-95``)."""

    def __init__(self, diag):
        if np.ndim(diag) != 1:
            raise ValueError(
                "The diagonal for the noise model be the same shape as the data; "
                "if passing a constant, it should be broadcasted first")
        self.diag = np.asarray(diag)

    def diagonal(self):
        return self.diag

    def _add(self, other):
        out = np.array(other, dtype=np.result_type(np.asarray(other).dtype, self.diag.dtype))
        if out.ndim != 2 or out.shape[0] != out.shape[1] or out.shape[0] != self.diag.shape[0]:
            raise ValueError("shape mismatch between the matrix and the noise diagonal")
        out[np.diag_indices(out.shape[0])] += self.diag
        return out

    def __add__(self, other):
        return self._add(other)

    def __radd__(self, other):
        return self._add(other)

    def __matmul__(self, other):
        other = np.asarray(other)
        if other.ndim == 1:
            return self.diag * other
        return self.diag[:, None] * other


class Dense(Noise):
    """Full-rank observation model (reference ``noise.py:98-124``).  The solver adds it on
    the host and uploads the sum through the ``covariance=`` channel."""

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
