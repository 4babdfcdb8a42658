"""NumPy front-ends to the device-pointer C ABI, for the GPU parity tests only."""
import ctypes as C

import numpy as np

from tinygp_amd import _ffi


def _f(a, dt):
    return np.asfortranarray(a, dtype=dt)


def potrf(A, nb_outer=None, lookahead=None, **options):
    """In-place lower Cholesky of a symmetric (n,n) host matrix, n % 128 == 0. Returns (L, info).
    Further context options (fused_step, gate_split, chain_reserve, ...) by keyword."""
    ctx = _ffi.default_ctx()
    dt = A.dtype
    n = A.shape[0]
    old = {}
    if nb_outer is not None:
        old["nb_outer"] = ctx.set_option("nb_outer", nb_outer)
    if lookahead is not None:
        old["lookahead"] = ctx.set_option("lookahead", lookahead)
    for k, v in options.items():
        old[k] = ctx.set_option(k, v)
    dA = ctx.upload(_f(A, dt).ravel(order="K"))
    info = C.c_int32()
    try:
        _ffi.check(_ffi.lib().tgp_potrf(ctx.handle, _ffi.dtype_code(dt), n, C.c_void_p(dA), n,
                                        C.byref(info)), "tgp_potrf")
        out = ctx.download(dA, (n, n), dt).T  # column-major buffer read row-major = transpose
    finally:
        ctx.free(dA)
        for k, v in old.items():
            ctx.set_option(k, v)
    return np.tril(out), info.value


def trsv(L, y, transpose=False):
    ctx = _ffi.default_ctx()
    dt = L.dtype
    n = L.shape[0]
    dL = ctx.upload(_f(L, dt).ravel(order="K"))
    dy = ctx.upload(np.ascontiguousarray(y, dtype=dt))
    try:
        _ffi.check(_ffi.lib().tgp_trsv(ctx.handle, _ffi.dtype_code(dt), n, C.c_void_p(dL), n,
                                       int(transpose), C.c_void_p(dy)), "tgp_trsv")
        return ctx.download(dy, (n,), dt)
    finally:
        ctx.free(dL), ctx.free(dy)


def trsm_right_lt(L, B):
    """B (m, n) -> B L^-T."""
    ctx = _ffi.default_ctx()
    dt = L.dtype
    n = L.shape[0]
    m = B.shape[0]
    dL = ctx.upload(_f(L, dt).ravel(order="K"))
    dB = ctx.upload(_f(B, dt).ravel(order="K"))
    try:
        _ffi.check(_ffi.lib().tgp_trsm_right_lt(ctx.handle, _ffi.dtype_code(dt), m, n,
                                                C.c_void_p(dL), n, C.c_void_p(dB), m),
                   "tgp_trsm_right_lt")
        return ctx.download(dB, (n, m), dt).T
    finally:
        ctx.free(dL), ctx.free(dB)


def gemm_nt(A, B, Cm, alpha, beta, lower=False):
    """C <- beta C + alpha A B^T; returns the full C (entries outside `lower` tiles untouched)."""
    ctx = _ffi.default_ctx()
    dt = A.dtype
    m, k = A.shape
    n = B.shape[0]
    dA = ctx.upload(_f(A, dt).ravel(order="K"))
    dB = ctx.upload(_f(B, dt).ravel(order="K"))
    dC = ctx.upload(_f(Cm, dt).ravel(order="K"))
    try:
        _ffi.check(_ffi.lib().tgp_gemm_nt(ctx.handle, _ffi.dtype_code(dt), m, n, k, float(alpha),
                                          C.c_void_p(dA), m, C.c_void_p(dB), n, float(beta),
                                          C.c_void_p(dC), m, int(lower)), "tgp_gemm_nt")
        return ctx.download(dC, (n, m), dt).T
    finally:
        ctx.free(dA), ctx.free(dB), ctx.free(dC)


def ubench(dtype):
    ctx = _ffi.default_ctx()
    out = C.c_double()
    _ffi.check(_ffi.lib().tgp_ubench_mfma(ctx.handle, _ffi.dtype_code(dtype), C.byref(out)),
               "tgp_ubench_mfma")
    return out.value
