"""`jax.scipy.linalg` -> SciPy/LAPACK (refshim).

jaxlib's CPU kernels for these two calls are LAPACK ?potrf / BLAS ?trsm; SciPy reaches the
same routines through OpenBLAS.  A failed factorisation returns NaNs (JAX never raises)."""
import numpy as _np
import scipy.linalg as _sla

from .._array import wrap


def cholesky(a, lower=False, overwrite_a=False, check_finite=True):
    a = _np.asarray(a)
    try:
        return wrap(_sla.cholesky(a, lower=lower, check_finite=False))
    except _sla.LinAlgError:
        return wrap(_np.full_like(a, _np.nan))


def solve_triangular(a, b, trans=0, lower=False, unit_diagonal=False, overwrite_b=False,
                     debug=None, check_finite=True):
    return wrap(_sla.solve_triangular(_np.asarray(a), _np.asarray(b), trans=trans, lower=lower,
                                      unit_diagonal=unit_diagonal, check_finite=False))


def block_diag(*arrs):
    return wrap(_sla.block_diag(*[_np.asarray(x) for x in arrs]))
