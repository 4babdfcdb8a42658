"""NumPy-in / NumPy-out wrappers over the device-pointer C ABI.

Used by ``Kernel.__call__`` / ``Kernel.matmul`` (stand-alone kernel evaluation,
reference ``kernels/base.py:68-103``).  The solver keeps its own resident
buffers (``solvers/direct.py``) and does not go through here.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from tinygp_amd import _ffi

__all__ = ["points", "common_dtype", "kmat", "kdiag", "kmat_gemv", "DeviceLimit", "MAX_DIM"]

MAX_DIM = 16  # TGP_MAX_DIM of include/tgp_hip.h: input dimensions of the HIP kernel evaluator


class DeviceLimit(NotImplementedError):
    """An input the reference accepts but the device kernel evaluator does not hold (D > 16, a kernel tree of
    more than 32 ops / stack depth 8): such a kernel MATRIX is evaluated on the host and handed to the solver
    through the reference's own ``covariance=`` channel (solvers/direct.py:44-52), like ``kernels.Custom``; the
    O(N^3) / O(N^2) linear algebra still runs on the device."""


def common_dtype(*arrays) -> np.dtype:
    """float32 only if every floating input is float32, else float64 (the reference's
    dtype follows JAX's global x64 flag, docs/troubleshooting.md:21-25; here it follows
    the inputs)."""
    dts = [np.asarray(a).dtype for a in arrays if a is not None]
    if dts and all(dt == np.float32 for dt in dts):
        return np.dtype(np.float32)
    return np.dtype(np.float64)


# -- pytree inputs (reference gp.py:64-112: X may be any pytree whose leaves share the leading data axis) ---------
# The device evaluator takes (N,) / (N, D) arrays only; a tuple / list / dict of arrays is an input for kernels that
# are arbitrary Python anyway (kernels.Custom, subclasses overriding evaluate()) and goes the host-evaluated route.
def is_tree(X) -> bool:
    return isinstance(X, (dict, list, tuple))


def tree_leaves(X) -> list:
    if isinstance(X, dict):
        return [leaf for k in sorted(X) for leaf in tree_leaves(X[k])]
    if isinstance(X, (list, tuple)):
        return [leaf for x in X for leaf in tree_leaves(x)]
    return [np.asarray(X)]


def tree_map(fn, X):
    if isinstance(X, dict):
        return {k: tree_map(fn, v) for k, v in X.items()}
    if isinstance(X, (list, tuple)):
        return type(X)(tree_map(fn, x) for x in X)
    return fn(np.asarray(X))


def tree_structure(X):
    """Nested keys / lengths plus the trailing shapes of the leaves (what `condition` compares, gp.py:176-191)."""
    if isinstance(X, dict):
        return ("dict", tuple((k, tree_structure(X[k])) for k in sorted(X)))
    if isinstance(X, (list, tuple)):
        return (type(X).__name__, tuple(tree_structure(x) for x in X))
    a = np.asarray(X)
    return ("leaf", a.ndim, a.shape[1:])


def num_points(X) -> int:
    """Length of the leading data axis (the same for every leaf of a pytree)."""
    leaves = tree_leaves(X) if is_tree(X) else [np.asarray(X)]
    if not leaves or any(a.ndim == 0 for a in leaves):
        raise ValueError("expected coordinates with a leading data axis")
    n = {a.shape[0] for a in leaves}
    if len(n) != 1:
        raise ValueError("the leaves of a pytree input must share their leading dimension")
    return n.pop()


def iter_points(X):
    """One data point at a time: a row of an array, or the pytree of the leaves' rows."""
    if not is_tree(X):
        yield from np.asarray(X)
        return
    for i in range(num_points(X)):
        yield tree_map(lambda a: a[i], X)


def points(X, dtype=None, *, limit: bool = True) -> np.ndarray:
    """Coordinates as a C-contiguous (N, D) array: (N,) -> (N, 1).  ``limit``: D must fit the device evaluator
    (:class:`DeviceLimit` otherwise, which every caller answers with the host-evaluated route)."""
    if isinstance(X, (dict, list, tuple)):
        raise NotImplementedError(
            "pytree inputs need a custom kernel evaluated on the host; the HIP path takes "
            "X of shape (N,) or (N, D)")
    X = np.asarray(X)
    if X.ndim == 0:
        raise ValueError("expected an array of coordinates with a leading data axis")
    if X.ndim == 1:
        X = X[:, None]
    if X.ndim != 2:
        raise ValueError(f"coordinates must have shape (N,) or (N, D); got ndim={X.ndim}")
    if limit and X.shape[1] > MAX_DIM:
        raise DeviceLimit(f"the HIP kernel evaluator holds at most D = {MAX_DIM} input dimensions (got {X.shape[1]})")
    dtype = common_dtype(X) if dtype is None else dtype
    return np.ascontiguousarray(X, dtype=dtype)


def kmat(prog, X1, X2, *, ctx=None) -> np.ndarray:
    """K[i, j] = k(X1[i], X2[j]) as a row-major (n1, n2) host array (K1)."""
    ctx = _ffi.default_ctx() if ctx is None else ctx
    dt = common_dtype(X1, X2)
    P1, P2 = points(X1, dt), points(X2, dt)
    if P1.shape[1] != P2.shape[1]:
        raise ValueError("X1 and X2 must have the same number of input dimensions")
    n1, n2, d = P1.shape[0], P2.shape[0], P1.shape[1]
    if n1 == 0 or n2 == 0:
        return np.zeros((n1, n2), dtype=dt)
    kp, nops = _ffi.as_kprog(prog)
    lib = _ffi.lib()
    # The device writes column-major.  Evaluating k(X2[j], X1[i]) into a column-major
    # (n2 x n1) buffer is byte-identical to the row-major (n1, n2) result we want
    # (|d| and d^2 are exactly symmetric in floating point).
    d1, d2 = ctx.upload(P1), ctx.upload(P2)
    out = ctx.malloc(n1 * n2 * dt.itemsize)
    try:
        _ffi.check(lib.tgp_kmat(ctx.handle, _ffi.dtype_code(dt), kp, nops, n2, n1, d,
                                C.c_void_p(d2), C.c_void_p(d1), None, C.c_void_p(out), n2, n2, n1,
                                0), "tgp_kmat")
        return ctx.download(out, (n1, n2), dt)
    finally:
        ctx.free(d1), ctx.free(d2), ctx.free(out)


def kdiag(prog, X, *, ctx=None) -> np.ndarray:
    """k(X[i], X[i]) as an (n,) host array (K2)."""
    ctx = _ffi.default_ctx() if ctx is None else ctx
    P = points(X)
    dt = P.dtype
    n, d = P.shape
    if n == 0:
        return np.zeros((0,), dtype=dt)
    kp, nops = _ffi.as_kprog(prog)
    dX = ctx.upload(P)
    out = ctx.malloc(n * dt.itemsize)
    try:
        _ffi.check(_ffi.lib().tgp_kdiag(ctx.handle, _ffi.dtype_code(dt), kp, nops, n, d,
                                        C.c_void_p(dX), C.c_void_p(out)), "tgp_kdiag")
        return ctx.download(out, (n,), dt)
    finally:
        ctx.free(dX), ctx.free(out)


def kmat_gemv(prog, X1, X2, v, *, ctx=None) -> np.ndarray:
    """sum_j k(X1[i], X2[j]) v[j, ...] without materialising K (K9)."""
    ctx = _ffi.default_ctx() if ctx is None else ctx
    v = np.asarray(v)
    dt = common_dtype(X1, X2, v)
    P1, P2 = points(X1, dt), points(X2, dt)
    if P1.shape[1] != P2.shape[1]:
        raise ValueError("X1 and X2 must have the same number of input dimensions")
    if v.shape[0] != P2.shape[0]:
        raise ValueError("dimension mismatch between X2 and y in Kernel.matmul")
    n1, n2, d = P1.shape[0], P2.shape[0], P1.shape[1]
    cols = np.ascontiguousarray(v.reshape(n2, -1).T, dtype=dt)  # (R, n2)
    nv = cols.shape[0]
    if n1 == 0 or nv == 0:
        return np.zeros((n1,) + v.shape[1:], dtype=dt)
    kp, nops = _ffi.as_kprog(prog)
    d1, d2, dv = ctx.upload(P1), ctx.upload(P2), ctx.upload(cols)
    do = ctx.malloc(nv * n1 * dt.itemsize)
    try:  # all R columns in one call: every kernel value is evaluated once per group of 8 columns
        _ffi.check(_ffi.lib().tgp_kmat_gemv_multi(ctx.handle, _ffi.dtype_code(dt), kp, nops, n1, n2, d,
                                                  C.c_void_p(d1), C.c_void_p(d2), C.c_void_p(dv), nv,
                                                  C.c_void_p(do)), "tgp_kmat_gemv_multi")
        res = ctx.download(do, (nv, n1), dt)
    finally:
        ctx.free(d1), ctx.free(d2), ctx.free(dv), ctx.free(do)
    return res.T.reshape((n1,) + v.shape[1:])
