"""Solvers (mirror of ``tinygp.solvers``): the dense :class:`DirectSolver` on MI355X.

``QuasisepSolver`` / ``KalmanSolver`` are a different algorithm family (O(N) state-space
recurrences) and are outside the hot path this package replaces.
"""

__all__ = ["Solver", "DirectSolver", "DistributedDirectSolver"]

from tinygp_amd.solvers.direct import DirectSolver
from tinygp_amd.solvers.distributed import DistributedDirectSolver
from tinygp_amd.solvers.solver import Solver
