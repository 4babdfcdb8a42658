"""Kernel base class and kernel algebra (mirror of ``tinygp.kernels.base``).

Where the reference evaluates ``vmap(vmap(evaluate))`` through XLA (reference
``kernels/base.py:84-103``), a kernel here *compiles* to a small postfix
"kernel program" (``tgp_kop`` array, ``include/tgp_hip.h``) that the HIP tile
evaluator runs for every pair of points.  ``Sum`` / ``Product`` / ``Constant``
(reference ``base.py:170-209``) are the ADD / MUL / CONST ops of that program.
"""

from __future__ import annotations

from typing import Any

import numpy as np

from tinygp_amd import _device

__all__ = ["Kernel", "Conditioned", "Custom", "Sum", "Product", "Constant", "DotProduct",
           "Polynomial"]

# op codes of include/tgp_hip.h
K_CONST, K_EXP, K_EXPSQ, K_M32, K_M52, K_COS, K_ESS, K_RQ, K_ADD, K_MUL = 0, 1, 2, 3, 4, 5, 6, 7, 16, 17
KPROG_MAX, KSTACK_MAX = 32, 8


class Kernel:
    """Base class of every kernel.

    Subclasses implement :meth:`_emit`, appending their postfix ops; the public
    protocol (``__call__``, ``evaluate``, ``evaluate_diag``, ``matmul`` and the
    ``+`` / ``*`` overloads) matches reference ``kernels/base.py:30-126``.
    """

    # ndarray (op) kernel must reach Kernel.__radd__/__rmul__ instead of broadcasting
    __array_ufunc__ = None

    # -- lowering --------------------------------------------------------------
    def _emit(self, ops: list) -> None:
        raise NotImplementedError(
            f"{type(self).__name__} cannot be lowered to the HIP kernel evaluator")

    def program(self) -> list[tuple[int, int, float, float]]:
        """The validated postfix program ``[(op, metric, p0, p1), ...]``."""
        ops: list = []
        self._emit(ops)
        if len(ops) > KPROG_MAX:
            raise _device.DeviceLimit(f"kernel expression too large for the device evaluator: {len(ops)} ops > {KPROG_MAX}")
        depth = peak = 0
        for op, *_ in ops:
            depth += -1 if op in (K_ADD, K_MUL) else 1
            peak = max(peak, depth)
        if peak > KSTACK_MAX:
            raise _device.DeviceLimit(f"kernel expression too deep for the device evaluator: stack {peak} > {KSTACK_MAX}")
        return ops

    def _slots(self, out: list) -> None:
        """Append, per program op, ``[(obj, attr) | None, (obj, attr) | None]`` naming the
        Python attributes behind the op's ``p0`` / ``p1`` (same order as :meth:`_emit`)."""
        raise NotImplementedError(f"{type(self).__name__} has no differentiable parameters")

    def parameters(self) -> list[tuple[Any, str]]:
        """The kernel's scalar hyper-parameters as ``(object, attribute)`` pairs, in program
        order -- the order of the ``"kernel"`` entry of ``log_probability_and_grad``."""
        slots: list = []
        self._slots(slots)
        return [s for pair in slots for s in pair if s is not None]

    def _lower(self, X):
        """``(program, coordinates)`` for evaluation at ``X``.  Plain kernels pass ``X``
        through; :mod:`tinygp_amd.transforms` fold their input map into the coordinates.
        Raises ``NotImplementedError`` (:class:`tinygp_amd._device.DeviceLimit` for inputs beyond the device
        evaluator's limits) when the kernel has to be evaluated on the host instead."""
        if _device.is_tree(X):
            raise _device.DeviceLimit("pytree inputs are evaluated on the host (kernels.Custom / evaluate())")
        prog = self.program()
        if np.ndim(X) == 2 and np.shape(X)[1] > _device.MAX_DIM:
            raise _device.DeviceLimit(f"D = {np.shape(X)[1]} input dimensions > {_device.MAX_DIM}")
        return prog, X

    # -- reference protocol ----------------------------------------------------
    def evaluate(self, X1, X2):
        """k(x1, x2) for ONE pair of points (reference ``base.py:38-57``).  Runs the same
        device evaluator as the matrix form on a 1x1 problem."""
        x1, x2 = np.asarray(X1), np.asarray(X2)
        if x1.ndim > 1 or x2.ndim > 1:
            raise ValueError("Kernel.evaluate takes single data points; call the kernel "
                             "instance to evaluate on arrays of points")
        dt = _device.common_dtype(x1, x2)
        return self(x1.reshape(1, -1).astype(dt), x2.reshape(1, -1).astype(dt))[0, 0]

    def evaluate_diag(self, X):
        """Reference ``base.py:59-66``."""
        return self.evaluate(X, X)

    def matmul(self, X1, X2=None, y=None):
        """``k(X1, X2) @ y`` with the reference's argument juggling (``base.py:68-82``);
        fused on the device -- the (N1, N2) matrix is never stored."""
        if y is None:
            assert X2 is not None
            y = X2
            X2 = None
        if X2 is None:
            X2 = X1
        try:
            prog, P1 = self._lower(X1)
            _, P2 = self._lower(X2)
        except NotImplementedError:
            return np.dot(self._host_matrix(X1, X2), y)  # host-evaluated kernel (see below)
        return _device.kmat_gemv(prog, P1, P2, y)

    def __call__(self, X1, X2=None):
        """Reference ``base.py:84-103``: diagonal (N,) when ``X2`` is None, else (N1, N2)."""
        try:
            prog, P1 = self._lower(X1)
            P2 = None if X2 is None else self._lower(X2)[1]
        except NotImplementedError:
            return self._host_diag(X1) if X2 is None else self._host_matrix(X1, X2)
        if X2 is None:
            return _device.kdiag(prog, P1)
        return _device.kmat(prog, P1, P2)

    # -- host-evaluated kernels ----------------------------------------------------
    # A kernel that is an arbitrary Python function of a pair of points (`Custom`, a user
    # subclass that only overrides `evaluate` -- the reference's extension point,
    # base.py:38-57) cannot run in a HIP kernel.  Its MATRIX is evaluated here on the host and
    # handed to the solver through the reference's own `covariance=` channel
    # (solvers/direct.py:44-52); the factorisation and every solve still run on the device.
    # Stationary trees never come this way: they lower to a device program or raise.
    def _host_matrix(self, X1, X2):
        if type(self).evaluate is Kernel.evaluate:
            raise NotImplementedError(
                f"{type(self).__name__} has neither a device program nor an evaluate() method")
        B = list(_device.iter_points(X2))
        K = np.asarray([[self.evaluate(a, b) for b in B] for a in _device.iter_points(X1)])
        if K.ndim != 2:
            raise ValueError(
                "Invalid kernel shape: "
                f"expected ndim = 2, got ndim={K.ndim} "
                "check the dimensions of parameters and custom kernels")
        return K

    def _host_diag(self, X):
        if type(self).evaluate is Kernel.evaluate and type(self).evaluate_diag is Kernel.evaluate_diag:
            raise NotImplementedError(
                f"{type(self).__name__} has neither a device program nor an evaluate() method")
        k = np.asarray([self.evaluate_diag(x) for x in _device.iter_points(X)])
        if k.ndim != 1:
            raise ValueError(
                "Invalid kernel diagonal shape: "
                f"expected ndim = 1, got ndim={k.ndim} "
                "check the dimensions of parameters and custom kernels")
        return k

    # -- algebra (reference base.py:105-126) -------------------------------------
    def __add__(self, other: Any) -> "Kernel":
        if isinstance(other, Kernel):
            return Sum(self, other)
        return Sum(self, Constant(other))

    def __radd__(self, other: Any) -> "Kernel":
        # `sum([...])` starts from the integer 0
        if not isinstance(other, Kernel) and np.ndim(other) == 0 and other == 0:
            return self
        if isinstance(other, Kernel):
            return Sum(other, self)
        return Sum(Constant(other), self)

    def __mul__(self, other: Any) -> "Kernel":
        if isinstance(other, Kernel):
            return Product(self, other)
        return Product(self, Constant(other))

    def __rmul__(self, other: Any) -> "Kernel":
        if isinstance(other, Kernel):
            return Product(other, self)
        return Product(Constant(other), self)


def _lower_binary(node, op, X):
    """Both operands must see the same device coordinates (one X per kernel matrix)."""
    p1, X1 = node.kernel1._lower(X)
    p2, X2 = node.kernel2._lower(X)
    const1 = all(o[0] in (K_CONST, K_ADD, K_MUL) for o in p1)
    const2 = all(o[0] in (K_CONST, K_ADD, K_MUL) for o in p2)
    if not (X1 is X2 or const1 or const2 or
            (np.shape(X1) == np.shape(X2) and np.array_equal(X1, X2))):
        raise NotImplementedError(
            "a Sum/Product whose operands use different input transforms cannot be evaluated "
            "in one device pass; evaluate on the host and pass covariance_value=")
    ops = p1 + p2 + [(op, 0, 0.0, 0.0)]
    if len(ops) > KPROG_MAX:
        raise _device.DeviceLimit(f"kernel expression too large for the device evaluator: {len(ops)} ops > {KPROG_MAX}")
    depth = peak = 0
    for o, *_ in ops:
        depth += -1 if o in (K_ADD, K_MUL) else 1
        peak = max(peak, depth)
    if peak > KSTACK_MAX:
        raise _device.DeviceLimit(f"kernel expression too deep for the device evaluator: stack {peak} > {KSTACK_MAX}")
    return ops, (X2 if const1 and not const2 else X1)


class Sum(Kernel):
    """k1 + k2 (reference ``base.py:170-177``)."""

    def __init__(self, kernel1: Kernel, kernel2: Kernel):
        self.kernel1, self.kernel2 = kernel1, kernel2

    def _emit(self, ops):
        self.kernel1._emit(ops)
        self.kernel2._emit(ops)
        ops.append((K_ADD, 0, 0.0, 0.0))

    def _lower(self, X):
        return _lower_binary(self, K_ADD, X)

    def _slots(self, out):
        self.kernel1._slots(out)
        self.kernel2._slots(out)
        out.append([None, None])

    def _host_matrix(self, X1, X2):  # an operand is host-evaluated: combine the two matrices
        return self.kernel1(X1, X2) + self.kernel2(X1, X2)

    def _host_diag(self, X):
        return self.kernel1(X) + self.kernel2(X)

    def __repr__(self):
        return f"Sum({self.kernel1!r}, {self.kernel2!r})"


class Product(Kernel):
    """k1 * k2 (reference ``base.py:180-187``)."""

    def __init__(self, kernel1: Kernel, kernel2: Kernel):
        self.kernel1, self.kernel2 = kernel1, kernel2

    def _emit(self, ops):
        self.kernel1._emit(ops)
        self.kernel2._emit(ops)
        ops.append((K_MUL, 0, 0.0, 0.0))

    def _lower(self, X):
        return _lower_binary(self, K_MUL, X)

    def _slots(self, out):
        self.kernel1._slots(out)
        self.kernel2._slots(out)
        out.append([None, None])

    def _host_matrix(self, X1, X2):  # an operand is host-evaluated: combine the two matrices
        return self.kernel1(X1, X2) * self.kernel2(X1, X2)

    def _host_diag(self, X):
        return self.kernel1(X) * self.kernel2(X)

    def __repr__(self):
        return f"Product({self.kernel1!r}, {self.kernel2!r})"


class Constant(Kernel):
    """k(x_i, x_j) = c (reference ``base.py:190-209``); a non-scalar value raises
    ``ValueError`` when the kernel is evaluated, as in the reference (``:207-208``)."""

    def __init__(self, value):
        self.value = value

    def _emit(self, ops):
        if np.ndim(self.value) != 0:
            raise ValueError("The value of a constant kernel must be a scalar")
        ops.append((K_CONST, 0, float(self.value), 0.0))

    def _slots(self, out):
        out.append([(self, "value"), None])

    def _host_matrix(self, X1, X2):  # only reached as an operand of a tree beyond the device limits
        if np.ndim(self.value) != 0:
            raise ValueError("The value of a constant kernel must be a scalar")
        leaves = lambda X: _device.tree_leaves(X) if _device.is_tree(X) else [np.asarray(X)]  # noqa: E731
        dt = _device.common_dtype(*leaves(X1), *leaves(X2))
        return np.full((_device.num_points(X1), _device.num_points(X2)), self.value, dtype=dt)

    def _host_diag(self, X):
        if np.ndim(self.value) != 0:
            raise ValueError("The value of a constant kernel must be a scalar")
        leaves = _device.tree_leaves(X) if _device.is_tree(X) else [np.asarray(X)]
        return np.full((_device.num_points(X),), self.value, dtype=_device.common_dtype(*leaves))

    def __repr__(self):
        return f"Constant({self.value!r})"


class Custom(Kernel):
    """A kernel given as a Python function of ONE pair of points (reference ``base.py:156-167``).
    Host-evaluated; see :meth:`Kernel._host_matrix`."""

    def __init__(self, function):
        self.function = function

    def evaluate(self, X1, X2):
        return self.function(X1, X2)


class DotProduct(Kernel):
    """``x_i . x_j`` (reference ``base.py:212-228``); host-evaluated (one GEMM)."""

    def evaluate(self, X1, X2):
        if np.ndim(X1) == 0:
            return X1 * X2
        return np.asarray(X1) @ np.asarray(X2)

    def _host_matrix(self, X1, X2):
        A, B = np.asarray(X1), np.asarray(X2)
        return np.multiply.outer(A, B) if A.ndim == 1 else A @ B.T

    def _host_diag(self, X):
        A = np.asarray(X)
        return A * A if A.ndim == 1 else np.einsum("ij,ij->i", A, A)


class Polynomial(Kernel):
    """``[(x_i / l) . (x_j / l) + sigma^2]^P`` (reference ``base.py:231-256``); host-evaluated."""

    def __init__(self, order, scale=1.0, sigma=0.0):
        self.order, self.scale, self.sigma = order, scale, sigma

    def evaluate(self, X1, X2):
        a, b = np.asarray(X1) / self.scale, np.asarray(X2) / self.scale
        dot = a * b if a.ndim == 0 else a @ b
        return (dot + np.square(self.sigma)) ** self.order

    def _host_matrix(self, X1, X2):
        A, B = np.asarray(X1) / self.scale, np.asarray(X2) / self.scale
        dot = np.multiply.outer(A, B) if A.ndim == 1 else A @ B.T
        return (dot + np.square(self.sigma)) ** self.order

    def _host_diag(self, X):
        A = np.asarray(X) / self.scale
        dot = A * A if A.ndim == 1 else np.einsum("ij,ij->i", A, A)
        return (dot + np.square(self.sigma)) ** self.order


def host_diag(kernel: "Kernel", X):
    """``kernel(X)`` -- the diagonal (N,) -- evaluated with the reference's formulas in NumPy, NEVER on the device
    (O(N) host logic: the variance a distributed solver reports identically on every rank).  Sums / products and
    host-side input transforms are walked here; the leaves carry the formulas (`_host_diag`)."""
    if isinstance(kernel, Sum):
        return host_diag(kernel.kernel1, X) + host_diag(kernel.kernel2, X)
    if isinstance(kernel, Product):
        return host_diag(kernel.kernel1, X) * host_diag(kernel.kernel2, X)
    mapper = getattr(kernel, "_map_points", None)
    if mapper is not None:  # transforms.*: the inner kernel on the transformed coordinates
        return host_diag(kernel.kernel, _host_mapped(kernel, X))
    return kernel._host_diag(X)


def host_matrix(kernel: "Kernel", X1, X2):
    """``kernel(X1, X2)`` (N1, N2) evaluated on the host only (see :func:`host_diag`)."""
    if isinstance(kernel, Sum):
        return host_matrix(kernel.kernel1, X1, X2) + host_matrix(kernel.kernel2, X1, X2)
    if isinstance(kernel, Product):
        return host_matrix(kernel.kernel1, X1, X2) * host_matrix(kernel.kernel2, X1, X2)
    if getattr(kernel, "_map_points", None) is not None:
        return host_matrix(kernel.kernel, _host_mapped(kernel, X1), _host_mapped(kernel, X2))
    return kernel._host_matrix(X1, X2)


def _host_mapped(kernel, X):
    X = np.asarray(X)
    out = np.asarray(kernel._map_points(X[:, None] if X.ndim == 1 else X))
    return out[:, None] if out.ndim == 1 else out


class Conditioned(Kernel):
    """The kernel of a process conditioned on data (reference ``base.py:129-153``):

        k*(x1, x2) = k(x1, x2) - (L^-1 k(X, x1))^T (L^-1 k(X, x2))

    Evaluated in matrix form through the solver's batched triangular solve; it is not a
    stationary expression, so it has no kernel program of its own.
    """

    def __init__(self, X, solver, kernel: Kernel):
        self.X, self.solver, self.kernel = X, solver, kernel

    def evaluate(self, X1, X2):
        x1 = np.asarray(X1)[None]
        x2 = np.asarray(X2)[None]
        return self(x1, x2)[0, 0]

    def evaluate_diag(self, X):
        return self(np.asarray(X)[None])[0]

    def matmul(self, X1, X2=None, y=None):
        if y is None:
            assert X2 is not None
            y = X2
            X2 = None
        if X2 is None:
            X2 = X1
        return np.dot(self(X1, X2), y)

    def __call__(self, X1, X2=None):
        if X2 is None:
            # diagonal: k(x,x) - |L^-1 k(X,x)|^2, one batched solve (base.py:150-153)
            return self.solver.condition_variance(self.kernel, X1)
        K1 = self.solver.solve_triangular(self.kernel(self.X, X1))
        K2 = K1 if X2 is X1 else self.solver.solve_triangular(self.kernel(self.X, X2))
        return self.kernel(X1, X2) - K1.T @ K2
