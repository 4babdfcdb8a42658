"""1-D block-cyclic column Cholesky across the GPUs of one node (BASELINE config 4).

The reference has no multi-device code at all; this is new design for MI355X + RCCL:

* the N x N matrix is split into block columns of width ``nb``; block column ``j`` (rows
  ``j*nb ..`` only -- the lower part) lives on rank ``j mod G`` as one column-major device
  buffer.  Cyclic ownership keeps the N^3/3 work balanced to within one block while staying
  "1-D block-column";
* step ``k``: the owner factors its panel (diagonal block Cholesky + triangular solve of
  the rows below, both through the C ABI), the ``(N - k nb) x nb`` panel is **broadcast**
  (``torch.distributed.broadcast`` = ``ncclBroadcast`` over xGMI), and every rank applies
  the MFMA trailing update to the block columns it owns;
* look-ahead: the owner of panel ``k+1`` updates and factors it first and its broadcast is
  posted asynchronously into the second receive buffer, so the transfer and the next
  panel's latency hide under the remaining updates of step ``k``;
* forward solve for ``log_probability``: per block one ``nb``-slice all-reduce of the
  per-rank partial sums, then the owner solves its diagonal block and folds its block
  column into its partial sum; two scalar all-reduces finish the job.

One process per GPU.  The schedule is written against a tiny block-operations interface:
:class:`HipBlockOps` (the product: device pointers into ``libtgp_hip.so``, on torch's
current stream so that RCCL orders with it) and, in ``tests/`` only, a NumPy stand-in that
lets the same schedule run under ``gloo`` on CPUs.
"""

from __future__ import annotations

import ctypes as C
import math

import numpy as np

from tinygp_amd import _ffi

__all__ = ["HipBlockOps", "BlockCyclicCholesky"]


class HipBlockOps:
    """Block operations on CUDA(=HIP) torch tensors through the device-pointer C ABI."""

    def __init__(self, device: int):
        import torch

        self.torch = torch
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        # One dedicated torch stream carries BOTH the HIP kernels (the C ABI launches on the
        # raw hipStream_t) and, through `context()`, the RCCL collectives torch enqueues, so
        # broadcasts order against the panel factorisation / trailing updates.
        self.stream = torch.cuda.Stream(device=self.device)
        self.ctx = _ffi.Ctx(device=device, stream=self.stream.cuda_stream)
        self.lib = _ffi.lib()

    def context(self):
        return self.torch.cuda.stream(self.stream)

    # -- buffers ---------------------------------------------------------------------
    def empty(self, nelem: int, dtype):
        return self.torch.empty(int(nelem), dtype=self._tdtype(dtype), device=self.device)

    def zeros(self, nelem: int, dtype):
        return self.torch.zeros(int(nelem), dtype=self._tdtype(dtype), device=self.device)

    def from_numpy(self, a: np.ndarray):
        return self.torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def _tdtype(self, dtype):
        return self.torch.float64 if np.dtype(dtype) == np.float64 else self.torch.float32

    @staticmethod
    def _p(t, offset_elems: int = 0):
        return C.c_void_p(t.data_ptr() + offset_elems * t.element_size())

    def _code(self, t):
        return _ffi.F64 if t.element_size() == 8 else _ffi.F32

    # -- block kernels ---------------------------------------------------------------
    def assemble(self, prog, X, diag, n, d, j0, nb, out, rows):
        """out (rows x nb, ld = rows) = K[j0:, j0:j0+nb] + noise on the diagonal, identity padding."""
        kp, nops = _ffi.as_kprog(prog)
        n1 = max(n - j0, 0)
        n2 = max(min(nb, n - j0), 0)
        _ffi.check(self.lib.tgp_kmat(self.ctx.handle, self._code(out), kp, nops, n1, n2, d,
                                     self._p(X, min(j0, n) * d), self._p(X, min(j0, n) * d),
                                     self._p(diag, min(j0, n)), self._p(out), rows, rows, nb, 0),
                   "tgp_kmat")

    def factor_panel(self, P, rows, nb) -> int:
        """Diagonal nb x nb block -> L_kk in place, rows below -> P L_kk^-T.  Returns potrf info."""
        info = C.c_int32()
        _ffi.check(self.lib.tgp_potrf(self.ctx.handle, self._code(P), nb, self._p(P), rows,
                                      C.byref(info)), "tgp_potrf")
        if rows > nb:
            _ffi.check(self.lib.tgp_trsm_right_lt(self.ctx.handle, self._code(P), rows - nb, nb,
                                                  self._p(P), rows, self._p(P, nb), rows),
                       "tgp_trsm_right_lt")
        return int(info.value)

    def update(self, P, prow, off, Cj, crow, nb):
        """C_j (crow x nb) -= P[off:, :] P[off:off+nb, :]^T on the lower trapezoid (MFMA)."""
        _ffi.check(self.lib.tgp_gemm_nt(self.ctx.handle, self._code(P), crow, nb, nb, -1.0,
                                        self._p(P, off), prow, self._p(P, off), prow, 1.0,
                                        self._p(Cj), crow, 1), "tgp_gemm_nt")

    def solve_diag(self, P, rows, nb, t):
        """t <- L_kk^-1 t for the nb x nb diagonal block at the top of the panel."""
        _ffi.check(self.lib.tgp_trsv(self.ctx.handle, self._code(P), nb, self._p(P), rows, 0,
                                     self._p(t)), "tgp_trsv")

    def gemv_sub(self, P, rows, nb, x, w_below):
        """w_below (rows - nb) -= P[nb:, :] x."""
        if rows > nb:
            _ffi.check(self.lib.tgp_gemv_sub(self.ctx.handle, self._code(P), rows - nb, nb,
                                             self._p(P, nb), rows, self._p(x), self._p(w_below)),
                       "tgp_gemv_sub")

    def sum_log_diag(self, P, rows, nb, nvalid) -> float:
        out = C.c_double()
        _ffi.check(self.lib.tgp_sum_log_diag(self.ctx.handle, self._code(P), nvalid, self._p(P),
                                             rows, C.byref(out)), "tgp_sum_log_diag")
        return out.value

    def sum_squares(self, x, nvalid) -> float:
        out = C.c_double()
        _ffi.check(self.lib.tgp_sum_squares(self.ctx.handle, self._code(x), nvalid, self._p(x),
                                            C.byref(out)), "tgp_sum_squares")
        return out.value


class BlockCyclicCholesky:
    """Distributed ``log_probability`` of the dense GP (assembly + Cholesky + forward solve).

    Args:
        kernel: a :mod:`tinygp_amd.kernels` tree.
        X: (N,) or (N, D) coordinates, replicated on every rank (<= a few MB).
        noise_diag: (N,) noise variances.
        nb: block-column width (multiple of 128).
        ops: block operations (default :class:`HipBlockOps` on ``LOCAL_RANK``).
        group: ``torch.distributed`` process group (default: the world).
    """

    def __init__(self, kernel, X, noise_diag, *, nb: int = 512, ops=None, group=None, dist=None):
        if dist is None:
            import torch.distributed as dist
        self.dist, self.group = dist, group
        self.rank = dist.get_rank(group)
        self.G = dist.get_world_size(group)
        if nb % 128:
            raise ValueError("nb must be a multiple of 128")
        X = np.asarray(X)
        P = np.ascontiguousarray(X[:, None] if X.ndim == 1 else X)
        self.dtype = np.dtype(np.float32 if P.dtype == np.float32 else np.float64)
        P = P.astype(self.dtype)
        self.n, self.d = P.shape
        self.nb = nb
        self.nblk = math.ceil(self.n / nb)
        self.npad = self.nblk * nb
        self.prog = kernel.program()
        if ops is None:
            import os

            ops = HipBlockOps(int(os.environ.get("LOCAL_RANK", "0")))
        self.ops = ops
        with ops.context():
            self._alloc(P, noise_diag)

    def _alloc(self, P, noise_diag):
        ops, nb = self.ops, self.nb
        self.X = ops.from_numpy(P.reshape(-1))
        self.diag = ops.from_numpy(np.ascontiguousarray(np.broadcast_to(noise_diag, (self.n,)),
                                                        dtype=self.dtype))
        self.owned = [j for j in range(self.nblk) if j % self.G == self.rank]
        self.cols = {j: ops.empty(self.rows(j) * nb, self.dtype) for j in self.owned}
        self.recv = [ops.empty(self.npad * nb, self.dtype) for _ in range(2)]
        self.info = 0
        self.factored = False

    def rows(self, j: int) -> int:
        return self.npad - j * self.nb

    def owner(self, j: int) -> int:
        return j % self.G

    def _global_rank(self, r: int) -> int:
        return r if self.group is None else self.dist.get_global_rank(self.group, r)

    # -- assembly + factorisation -------------------------------------------------------
    def assemble(self, kernel=None):
        with self.ops.context():
            self._assemble(kernel)

    def factor(self):
        with self.ops.context():
            return self._factor()

    def log_probability(self, resid) -> float:
        """``-0.5 |L^-1 r|^2 - sum log L_ii - n/2 log(2 pi)``; ``-inf`` when not finite."""
        with self.ops.context():
            return self._log_probability(resid)

    def _assemble(self, kernel=None):
        if kernel is not None:
            self.prog = kernel.program()
        for j in self.owned:
            self.ops.assemble(self.prog, self.X, self.diag, self.n, self.d, j * self.nb, self.nb,
                              self.cols[j], self.rows(j))
        self.factored = False

    def _panel(self, k):
        """The buffer holding panel k on this rank (own column or receive buffer)."""
        if self.owner(k) == self.rank:
            return self.cols[k]
        return self.recv[k % 2][: self.rows(k) * self.nb]

    def _factor_own(self, k):
        info = self.ops.factor_panel(self.cols[k], self.rows(k), self.nb)
        if info > 0 and self.info == 0:
            self.info = k * self.nb + info

    def _bcast(self, k):
        return self.dist.broadcast(self._panel(k), src=self._global_rank(self.owner(k)),
                                   group=self.group, async_op=True)

    def _factor(self):
        nb, ops = self.nb, self.ops
        self.info = 0
        if self.owner(0) == self.rank:
            self._factor_own(0)
        work = self._bcast(0)
        for k in range(self.nblk):
            work.wait()  # panel k has arrived (RCCL: a stream dependency, not a host block)
            Pk, prow = self._panel(k), self.rows(k)
            nxt = k + 1
            if nxt < self.nblk:
                if self.owner(nxt) == self.rank:  # look-ahead: next panel first
                    ops.update(Pk, prow, nb, self.cols[nxt], self.rows(nxt), nb)
                    self._factor_own(nxt)
                work = self._bcast(nxt)
            for j in self.owned:
                if j > nxt:
                    ops.update(Pk, prow, (j - k) * nb, self.cols[j], self.rows(j), nb)
        # agree on the first failing pivot (LAPACK convention), 0 if none
        t = self._scalar_tensor(float(self.info) if self.info else float(2**52))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        v = float(t.item())
        self.info = 0 if v >= 2**52 else int(v)
        self.factored = True
        return self.info

    def _scalar_tensor(self, v: float):
        return self.ops.from_numpy(np.array([v], dtype=np.float64))

    # -- forward solve + reductions -----------------------------------------------------
    def _log_probability(self, resid) -> float:
        if not self.factored:
            self._assemble()
            self._factor()
        nb, ops, dist = self.nb, self.ops, self.dist
        r = np.zeros(self.npad, dtype=self.dtype)
        r[: self.n] = np.broadcast_to(resid, (self.n,))
        y = ops.from_numpy(r)
        w = ops.zeros(self.npad, self.dtype)  # minus this rank's partial sums  -sum_j L[:, j] x_j
        ss = logdet = 0.0
        for k in range(self.nblk):
            sl = w[k * nb:(k + 1) * nb].clone()
            dist.all_reduce(sl, group=self.group)
            if self.owner(k) == self.rank:
                t = y[k * nb:(k + 1) * nb] + sl
                ops.solve_diag(self.cols[k], self.rows(k), nb, t)
                ops.gemv_sub(self.cols[k], self.rows(k), nb, t, w[(k + 1) * nb:])
                nvalid = max(min(nb, self.n - k * nb), 0)
                ss += ops.sum_squares(t, nvalid)
                logdet += ops.sum_log_diag(self.cols[k], self.rows(k), nb, nvalid)
        t2 = self.ops.from_numpy(np.array([ss, logdet], dtype=np.float64))
        dist.all_reduce(t2, group=self.group)
        ss, logdet = (float(v) for v in t2.cpu().numpy())
        ll = -0.5 * ss - (logdet + 0.5 * self.n * math.log(2.0 * math.pi))
        if self.info or not math.isfinite(ll):
            return -math.inf
        return ll
