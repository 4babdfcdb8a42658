"""tinygp_amd -- the dense ``DirectSolver`` hot path of dfm/tinygp, native to MI355X (gfx950).

Same ``GaussianProcess`` / ``kernels.*`` / ``noise.*`` / ``solvers.DirectSolver`` surface as
``tinygp`` for the dense path; kernel-matrix assembly, the blocked Cholesky, triangular
solves and the conditional products run in hand-written HIP kernels behind the C ABI of
``include/tgp_hip.h``.  There is no CPU fallback.
"""

from tinygp_amd import kernels as kernels
from tinygp_amd import means as means
from tinygp_amd import noise as noise
from tinygp_amd import solvers as solvers
from tinygp_amd import transforms as transforms
from tinygp_amd.gp import ConditionResult as ConditionResult
from tinygp_amd.gp import GaussianProcess as GaussianProcess

__version__ = "0.1.0"
