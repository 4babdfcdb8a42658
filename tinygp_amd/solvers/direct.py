"""``DirectSolver`` on MI355X (replaces reference ``solvers/direct.py:17-95``).

The reference builds ``K = kernel(X, X) + noise`` and calls
``jax.scipy.linalg.cholesky`` / ``solve_triangular`` (``direct.py:51-53,66-70``), keeping
both K and L resident (16 N^2 bytes).  Here a ``tgp_solver`` handle (C ABI,
``include/tgp_hip.h``) owns ONE padded N x N device matrix: the HIP tile evaluator writes
the lower triangle of K with the noise fused onto the diagonal, a right-looking blocked
LL^T (fp64/fp32 MFMA trailing updates) overwrites it with L, and the triangular solves,
reductions and conditional products run against that resident factor.  ``covariance()``
is recomputed on demand.

Numerical failure never raises (reference: NaN factor, ``gp.py:316`` -> ``-inf``): a
positive ``potrf`` info is kept in ``self.info`` and the affected results are NaN.
"""

from __future__ import annotations

import ctypes as C
from typing import Any

import numpy as np

from tinygp_amd import _device, _ffi
from tinygp_amd.noise import Diagonal, Noise
from tinygp_amd.solvers.solver import Solver

__all__ = ["DirectSolver"]


class DirectSolver(Solver):
    """Dense Cholesky solver whose O(N^3) / O(N^2) work runs in hand-written HIP kernels.

    Args:
        kernel: the kernel function (a tree of stationary kernels for on-device assembly).
        X: input coordinates, shape (N,) or (N, D).
        noise: the noise model.
        covariance: optional pre-computed (N, N) covariance; like the reference
            (``direct.py:44-46``) it is trusted to equal ``kernel(X, X) + noise``.
        ctx: optional :class:`tinygp_amd._ffi.Ctx` (device + stream); extra keyword of
            this implementation, reachable through ``GaussianProcess(**solver_kwargs)``.
    """

    def __init__(self, kernel, X, noise: Noise, *, covariance: Any | None = None, ctx=None):
        self.kernel, self.X, self.noise = kernel, X, noise
        self._ctx = _ffi.default_ctx() if ctx is None else ctx
        self._handle = None
        self._covariance_value = None
        self._scale_tril = None

        noise_diag = np.asarray(noise.diagonal())
        if covariance is not None:
            covariance = np.asarray(covariance)
        try:  # transforms fold their input map into the coordinates the device sees
            self._prog, Xdev = kernel._lower(X)
        except NotImplementedError:
            self._prog, Xdev = None, X  # e.g. kernels.Conditioned: needs covariance=
        if _device.is_tree(Xdev):  # pytree input: host-evaluated kernels only (kernel._lower raised above)
            assert self._prog is None
            dt = _device.common_dtype(*_device.tree_leaves(Xdev), noise_diag, covariance)
            P = np.zeros((_device.num_points(Xdev), 1), dtype=dt)
        else:
            dt = _device.common_dtype(np.asarray(X), noise_diag, covariance)
            P = _device.points(Xdev, dt, limit=False)
        if P.shape[1] > _device.MAX_DIM:
            # beyond the device evaluator (kernel._lower raised DeviceLimit above): every kernel matrix of this
            # solver comes from the host; the device never reads the coordinates -- it gets a 1-D placeholder
            assert self._prog is None
            P = np.zeros((P.shape[0], 1), dtype=dt)
        self.dtype = dt
        self._P = P
        self.n, self.d = P.shape
        if noise_diag.shape != (self.n,):
            raise ValueError("the noise model must have one entry per data point")
        self._noise_diag = np.ascontiguousarray(noise_diag, dtype=dt)

        if covariance is None and (self._prog is None or not isinstance(noise, Diagonal)):
            # noise.Dense, or a host-evaluated kernel (Custom / DotProduct / Conditioned ...):
            # K + noise is formed on the host exactly like reference direct.py:50-52 and
            # shipped through the covariance channel; the factorisation runs on the device
            covariance = kernel(X, X) + noise
        if covariance is not None:
            covariance = np.ascontiguousarray(covariance, dtype=dt)
            if covariance.shape != (self.n, self.n):
                raise ValueError("covariance must have shape (N, N)")
            self._covariance_value = covariance

        lib = _ffi.lib()
        h = C.c_void_p()
        _ffi.check(lib.tgp_solver_create(self._ctx.handle, _ffi.dtype_code(dt), self.n, self.d,
                                         _ffi.ptr(P), _ffi.ptr(self._noise_diag), C.byref(h)),
                   "tgp_solver_create")
        self._handle = h
        # The factorisation is deferred to the first use: when that first use is
        # log_probability (the optimiser / MCMC loop) assembly, Cholesky and the forward
        # solve run as ONE fused device pass.  Results are identical either way.
        self._info = 0
        self._factored = False

    # -- factorisation -------------------------------------------------------------
    def _set_kernel(self, kernel):
        """New hyper-parameters.  The host copy of ``K + noise`` (if any) belongs to the OLD
        kernel: it is dropped when the device can re-assemble the matrix (device program +
        diagonal noise) and REBUILT on the host otherwise (``noise.Dense``, host-evaluated
        kernels), so that neither off-diagonal noise nor a stale matrix is ever factored."""
        if kernel is None:
            return
        # validate first, commit kernel / program / host matrix together: a ValueError must leave the
        # solver exactly as it was (round-2 advisor finding)
        try:
            prog, Xdev = kernel._lower(self.X)
            if not np.array_equal(_device.points(Xdev, self.dtype, limit=False), self._P):
                raise ValueError("refactor() cannot change the kernel's input transform: the "
                                 "transformed coordinates are resident on the device")
        except NotImplementedError:
            prog = None
        if prog is not None and isinstance(self.noise, Diagonal):
            cov = None
        else:
            cov = np.ascontiguousarray(kernel(self.X, self.X) + self.noise, dtype=self.dtype)
        self.kernel, self._prog, self._covariance_value = kernel, prog, cov

    def refactor(self, kernel=None, *, covariance=None) -> int:
        """(Re-)assemble and (re-)factor in place with new hyper-parameters (same X / noise):
        the optimiser / MCMC step of SURVEY 3.4 without re-uploading anything."""
        self._set_kernel(kernel)
        if covariance is not None:  # trusted to equal kernel(X, X) + noise (direct.py:44-46)
            covariance = np.ascontiguousarray(covariance, dtype=self.dtype)
            if covariance.shape != (self.n, self.n):
                raise ValueError("covariance must have shape (N, N)")
            self._covariance_value = covariance
        covariance = self._covariance_value
        kp, nops = _ffi.as_kprog(self._prog or [])
        info = C.c_int32(0)
        _ffi.check(_ffi.lib().tgp_solver_factor(self._handle, kp, nops if self._prog else 0,
                                                _ffi.ptr(covariance), C.byref(info)),
                   "tgp_solver_factor")
        self._info = int(info.value)
        self._factored = True
        self._scale_tril = None
        return self._info

    def factor_log_probability(self, resid=None, kernel=None) -> float:
        """Fused step: assemble + factor + forward solve + reductions in one device pass
        (``tgp_solver_factor_logprob``).  ``resid=None`` re-uses the residual made resident
        by :meth:`set_residual`.  Returns the raw log-probability (NaN if not PD)."""
        self._set_kernel(kernel)
        kp, nops = _ffi.as_kprog(self._prog or [])
        r = None
        if resid is not None:
            r = np.ascontiguousarray(np.broadcast_to(resid, (self.n,)), dtype=self.dtype)
        info, out = C.c_int32(0), C.c_double()
        _ffi.check(_ffi.lib().tgp_solver_factor_logprob(
            self._handle, kp, nops if self._prog else 0, _ffi.ptr(self._covariance_value),
            _ffi.ptr(r), C.byref(info), C.byref(out)), "tgp_solver_factor_logprob")
        self._info = int(info.value)
        self._factored = True
        self._scale_tril = None
        return out.value

    def set_residual(self, resid):
        """Make ``y - loc`` resident on the device for repeated fused evaluations."""
        r = np.ascontiguousarray(np.broadcast_to(resid, (self.n,)), dtype=self.dtype)
        _ffi.check(_ffi.lib().tgp_solver_set_resid(self._handle, _ffi.ptr(r)), "tgp_solver_set_resid")

    def _ensure_factor(self):
        if not self._factored:
            self.refactor()

    @property
    def info(self) -> int:
        """LAPACK-style potrf info: 0, or the 1-based index of the first non-positive pivot."""
        self._ensure_factor()
        return self._info

    # -- Solver protocol -------------------------------------------------------------
    def variance(self):
        """Reference ``direct.py:49,55-56``: ``kernel(X) + noise.diagonal()``."""
        if self._prog is None:
            return self.kernel(self.X) + self._noise_diag
        self._ensure_factor()
        out = np.empty(self.n, dtype=self.dtype)
        _ffi.check(_ffi.lib().tgp_solver_variance(self._handle, _ffi.ptr(out)), "tgp_solver_variance")
        return out

    def covariance(self):
        """Reference ``direct.py:58-59``.  Recomputed: the factor overwrote K on the device."""
        if self._covariance_value is not None:
            return self._covariance_value
        self._ensure_factor()
        out = np.empty((self.n, self.n), dtype=self.dtype)
        _ffi.check(_ffi.lib().tgp_solver_covariance(self._handle, _ffi.ptr(out)),
                   "tgp_solver_covariance")
        return out

    @property
    def variance_value(self):
        return self.variance()

    @property
    def covariance_value(self):
        return self.covariance()

    @property
    def scale_tril(self):
        """The lower Cholesky factor as an (N, N) host array, upper triangle zero."""
        self._ensure_factor()
        if self._scale_tril is None:
            out = np.empty((self.n, self.n), dtype=self.dtype)
            _ffi.check(_ffi.lib().tgp_solver_get_factor(self._handle, _ffi.ptr(out)),
                       "tgp_solver_get_factor")
            if self.info:
                out[:] = np.nan  # jax.scipy.linalg.cholesky: all-NaN on failure
            self._scale_tril = out
        return self._scale_tril

    def normalization(self):
        """Reference ``direct.py:61-64``."""
        self._ensure_factor()
        out = C.c_double()
        _ffi.check(_ffi.lib().tgp_solver_normalization(self._handle, C.byref(out)),
                   "tgp_solver_normalization")
        return self.dtype.type(np.nan if self.info else out.value)

    def solve_triangular(self, y, *, transpose: bool = False):
        """Reference ``direct.py:66-70``: ``L x = y`` or ``L^T x = y``; y (N,) or (N, R)."""
        self._ensure_factor()
        y = np.asarray(y)
        if y.ndim not in (1, 2) or y.shape[0] != self.n:
            raise ValueError(f"y must have shape ({self.n},) or ({self.n}, R); got {y.shape}")
        dt = np.result_type(self.dtype, y.dtype) if y.dtype.kind == "f" else self.dtype
        yy = np.ascontiguousarray(y, dtype=self.dtype)
        nrhs = 1 if y.ndim == 1 else y.shape[1]
        out = np.empty_like(yy)
        if nrhs:
            _ffi.check(_ffi.lib().tgp_solver_solve_tri(self._handle, int(bool(transpose)), nrhs,
                                                       _ffi.ptr(yy), _ffi.ptr(out)),
                       "tgp_solver_solve_tri")
        if self.info:
            out[:] = np.nan
        return out.astype(dt, copy=False)

    def dot_triangular(self, y):
        """Reference ``direct.py:72-73``: ``einsum('ij,j...->i...', L, y)``."""
        self._ensure_factor()
        y = np.asarray(y)
        if y.ndim < 1 or y.shape[0] != self.n:
            raise ValueError(f"y must have leading dimension {self.n}")
        yy = np.ascontiguousarray(y.reshape(self.n, -1), dtype=self.dtype)
        out = np.empty_like(yy)
        if yy.shape[1]:
            _ffi.check(_ffi.lib().tgp_solver_dot_tri(self._handle, yy.shape[1], _ffi.ptr(yy),
                                                     _ffi.ptr(out)), "tgp_solver_dot_tri")
        if self.info:
            out[:] = np.nan
        return out.reshape(y.shape)

    def _cond_host_kernel(self, kernel, X_test, noise_diag, var_only: bool):
        """``Kss + noise - A^T A`` for a kernel without a device program (``Conditioned`` --
        conditioning a conditioned GP, reference gp.py:380-385 -- ``Custom``, ...): the cross
        covariances come from the kernel's own ``__call__``, the N x M triangular solve runs
        on the device (reference direct.py:87-95 line by line)."""
        Xt = self.X if X_test is None else X_test
        A = self.solve_triangular(np.asarray(kernel(self.X, Xt), dtype=self.dtype))
        if var_only:
            out = np.asarray(kernel(Xt), dtype=self.dtype) - np.sum(A * A, axis=0)
            out = out if noise_diag is None else out + noise_diag
        else:
            out = np.asarray(kernel(Xt, Xt), dtype=self.dtype) - A.T @ A
            if noise_diag is not None:
                out[np.diag_indices(out.shape[0])] += noise_diag
        if self.info:  # a failed factorisation poisons every result, as on the device path (gp.py:316)
            out = np.full_like(out, np.nan)
        return out

    def _cond(self, kernel, X_test, noise_diag, var_only: bool):
        self._ensure_factor()
        try:
            prog = self._lower_like_resident(kernel)
        except NotImplementedError:
            return self._cond_host_kernel(kernel, X_test, noise_diag, var_only)
        kp, nops = _ffi.as_kprog(prog)
        if X_test is None:
            Pt, m = None, self.n
        else:
            Pt = _device.points(kernel._lower(X_test)[1], self.dtype)
            if Pt.shape[1] != self.d:
                raise ValueError("X_test must have the same number of input dimensions as X")
            m = Pt.shape[0]
        nd = None
        if noise_diag is not None:
            nd = np.ascontiguousarray(np.broadcast_to(noise_diag, (m,)), dtype=self.dtype)
        out = np.empty((m,) if var_only else (m, m), dtype=self.dtype)
        if m:
            _ffi.check(_ffi.lib().tgp_solver_condition_cov(self._handle, kp, nops, m, _ffi.ptr(Pt),
                                                           _ffi.ptr(nd), int(var_only),
                                                           _ffi.ptr(out)),
                       "tgp_solver_condition_cov")
        if self.info:
            out[:] = np.nan
        return out

    def _lower_like_resident(self, kernel):
        """Program of ``kernel`` provided its input transform maps X onto the resident points."""
        prog, Xdev = kernel._lower(self.X)
        if Xdev is not self.X and not np.array_equal(_device.points(Xdev, self.dtype, limit=False), self._P):
            raise NotImplementedError(
                "conditioning with a kernel whose input transform differs from the GP's kernel "
                "needs a host-evaluated covariance")
        return prog

    def condition(self, kernel, X_test, noise):
        """Reference ``direct.py:75-95``: ``Kss + noise - A^T A`` with ``A = L^-1 Ks``.
        Assembly of Ks/Kss, the M-RHS triangular solve and the SYRK all stay on the device."""
        if isinstance(noise, Diagonal):
            return self._cond(kernel, X_test, noise.diagonal(), False)
        return self._cond(kernel, X_test, None, False) + noise  # e.g. noise.Dense

    def condition_variance(self, kernel, X_test):
        """diag of ``k(Xt,Xt) - A^T A`` without forming the (M, M) matrix
        (what ``kernels.Conditioned.evaluate_diag`` computes per point, reference
        ``kernels/base.py:150-153``)."""
        return self._cond(kernel, X_test, None, True)

    # -- fused hot path used by GaussianProcess ----------------------------------------
    def log_probability(self, resid):
        """``-0.5 |L^-1 r|^2 - normalization`` fused on the device (reference
        ``gp.py:313-320``); non-finite -> ``-inf`` (``gp.py:316``)."""
        if not self._factored:
            v = self.factor_log_probability(resid)  # first use: one fused device pass
        else:
            r = np.ascontiguousarray(np.broadcast_to(resid, (self.n,)), dtype=self.dtype)
            out = C.c_double()
            _ffi.check(_ffi.lib().tgp_solver_logprob(self._handle, _ffi.ptr(r), C.byref(out)),
                       "tgp_solver_logprob")
            v = out.value
        if self.info or not np.isfinite(v):
            v = -np.inf
        return self.dtype.type(v)

    def log_probability_and_grad(self, resid):
        """``(log_probability, grads)`` with ``grads = {"kernel": [...], "noise_diag": (N,),
        "mean": (N,), "transform": ...}``: derivatives with respect to ``kernel.parameters()`` (same order), to
        every noise variance, to every entry of the mean vector (= K^-1 r), and -- when the kernel tree holds ONE
        ``transforms.Linear`` / ``Cholesky`` with a scalar or per-dimension parameter -- to that ``scale`` /
        ``factor`` (same shape; ``None`` otherwise).  The
        reference's users get this from ``jax.value_and_grad`` around ``log_probability``
        (docs/tutorials/quickstart.ipynb); here it is
        ``1/2 tr((alpha alpha^T - K^-1) dK/dtheta)`` with ``K^-1 = L^-T L^-1`` formed by a
        clipped triangular sweep + one MFMA GEMM, and dK/dtheta evaluated tile by tile."""
        self._ensure_factor()
        if self._prog is None:
            raise NotImplementedError("gradients need an on-device kernel program")
        slots: list = []
        self.kernel._slots(slots)
        r = np.ascontiguousarray(np.broadcast_to(resid, (self.n,)), dtype=self.dtype)
        nops = len(self._prog)
        gp_ = (C.c_double * (2 * nops))()
        gnoise = np.empty(self.n, dtype=self.dtype)
        alpha = np.empty(self.n, dtype=self.dtype)
        out = C.c_double()
        # an input transform with per-dimension scales (transforms.Linear / Cholesky, reference transforms.py:39-133;
        # kernels/stationary.py:41-43 sends users there for anisotropic length scales): the device also returns
        # d ll / d log s_q per dimension of the transformed coordinates, one extra pass over K^-1 each
        from tinygp_amd.transforms import covering_transform

        tf = covering_transform(self.kernel)  # None unless ONE transform wraps every coordinate-dependent leaf
        tfs = [] if tf is None else [tf]
        glog = (C.c_double * self.d)() if len(tfs) == 1 else None
        _ffi.check(_ffi.lib().tgp_solver_grad(self._handle, _ffi.ptr(r), C.byref(out), gp_,
                                              _ffi.ptr(gnoise), _ffi.ptr(alpha), glog), "tgp_solver_grad")
        kgrad = [gp_[2 * i + q] for i, pair in enumerate(slots) for q in (0, 1) if pair[q] is not None]
        tgrad = None
        if glog is not None:
            try:
                tgrad = tfs[0]._logscale_gradient(np.array(list(glog)))
            except NotImplementedError:
                tgrad = None  # (Subspace, a Python callable, a full matrix: no per-dimension scale)
        ll = out.value
        if self.info or not np.isfinite(ll):
            ll = -np.inf
            kgrad = [np.nan] * len(kgrad)
            gnoise[:] = np.nan
            alpha[:] = np.nan
            tgrad = None if tgrad is None else np.full_like(np.asarray(tgrad, dtype=np.float64), np.nan)
        return self.dtype.type(ll), {"kernel": kgrad, "noise_diag": gnoise, "mean": alpha, "transform": tgrad}

    def alpha(self, resid):
        """``(K^-1 r, log_probability)`` -- the two solves of reference ``gp.py:330-334``."""
        self._ensure_factor()
        r = np.ascontiguousarray(np.broadcast_to(resid, (self.n,)), dtype=self.dtype)
        a = np.empty(self.n, dtype=self.dtype)
        out = C.c_double()
        _ffi.check(_ffi.lib().tgp_solver_alpha(self._handle, _ffi.ptr(r), _ffi.ptr(a), C.byref(out)),
                   "tgp_solver_alpha")
        v = out.value
        if self.info or not np.isfinite(v):
            v = -np.inf
        if self.info:
            a[:] = np.nan
        return a, self.dtype.type(v)

    def conditional_mean(self, kernel, X_test, alpha):
        """``K(X_test, X) @ alpha`` fused (reference ``gp.py:357`` via ``base.py:68-82``)."""
        try:
            prog = self._lower_like_resident(kernel)
            Pt = _device.points(kernel._lower(X_test)[1], self.dtype)
        except NotImplementedError:  # host-evaluated kernel: its own matmul (base.py:68-82)
            return np.asarray(kernel.matmul(X_test, self.X, alpha), dtype=self.dtype)
        if Pt.shape[1] != self.d:
            raise ValueError("X_test must have the same number of input dimensions as X")
        kp, nops = _ffi.as_kprog(prog)
        a = np.ascontiguousarray(alpha, dtype=self.dtype)
        out = np.empty(Pt.shape[0], dtype=self.dtype)
        if Pt.shape[0]:
            _ffi.check(_ffi.lib().tgp_solver_cond_mean(self._handle, kp, nops, Pt.shape[0],
                                                       _ffi.ptr(Pt), _ffi.ptr(a), _ffi.ptr(out)),
                       "tgp_solver_cond_mean")
        return out

    def timings(self) -> dict:
        ms = (C.c_double * 8)()
        _ffi.check(_ffi.lib().tgp_solver_timings(self._handle, ms, 8), "tgp_solver_timings")
        keys = ["assembly_ms", "potrf_ms", "syrk_ms", "syrk_launches", "trsv_ms", "host_submit_ms"]
        return {k: ms[i] for i, k in enumerate(keys)}

    # -- lifetime ----------------------------------------------------------------------
    def close(self):
        if getattr(self, "_handle", None):
            _ffi.lib().tgp_solver_destroy(self._handle)
            self._handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass
