"""Kernel building blocks (mirror of ``tinygp.kernels`` for the dense hot path).

Kernels are built as sums and products of the stationary leaves below, exactly as in
the reference; calling a kernel evaluates it on the MI355X through the HIP tile
evaluator.  ``Custom``, ``DotProduct`` and ``Polynomial`` are arbitrary / non-stationary
functions: their matrix is evaluated on the host and handed to the solver through the
``covariance=`` channel (the factorisation still runs on the device).  The ``quasisep``
family is outside the hot path this package replaces.
"""

__all__ = [
    "Distance", "L1Distance", "L2Distance", "Kernel", "Conditioned", "Custom", "Sum",
    "Product", "Constant", "DotProduct", "Polynomial", "Stationary", "Exp", "ExpSquared", "Matern32", "Matern52", "Cosine",
    "ExpSineSquared", "RationalQuadratic",
]

from tinygp_amd.kernels.base import (
    Conditioned,
    Constant,
    Custom,
    DotProduct,
    Kernel,
    Polynomial,
    Product,
    Sum,
)
from tinygp_amd.kernels.distance import Distance, L1Distance, L2Distance
from tinygp_amd.kernels.stationary import (
    Cosine,
    Exp,
    ExpSineSquared,
    ExpSquared,
    Matern32,
    Matern52,
    RationalQuadratic,
    Stationary,
)
