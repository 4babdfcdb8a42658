"""The ``Solver`` plug-in contract (mirror of reference ``solvers/solver.py:15-82``)."""

from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any

__all__ = ["Solver"]


class Solver(ABC):
    """What ``GaussianProcess(..., solver=Cls, **solver_kwargs)`` expects of ``Cls``
    (reference ``gp.py:101-112``): a constructor ``Cls(kernel, X, noise, *, covariance=None)``
    and the six methods below."""

    def __init__(self, kernel, X, noise, *, covariance: Any | None = None):
        del kernel, X, noise, covariance
        raise NotImplementedError

    @classmethod
    def init(cls, kernel, X, noise, *, covariance: Any | None = None) -> "Solver":
        """Back-compat alias of the constructor (reference ``solver.py:29-38``)."""
        return cls(kernel, X, noise, covariance=covariance)

    @abstractmethod
    def variance(self):
        """The diagonal of the covariance matrix."""

    @abstractmethod
    def covariance(self):
        """The evaluated covariance matrix."""

    @abstractmethod
    def normalization(self):
        """``(log_det + n*log(2*pi))/2``."""

    @abstractmethod
    def solve_triangular(self, y, *, transpose: bool = False):
        """Solve ``L x = y`` (or ``L^T x = y``) for the lower Cholesky factor ``L``."""

    @abstractmethod
    def dot_triangular(self, y):
        """``L @ y``."""

    @abstractmethod
    def condition(self, kernel, X_test, noise) -> Any:
        """The covariance of the process conditioned on the data, at ``X_test``."""
