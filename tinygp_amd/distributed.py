"""1-D block-cyclic column Cholesky across the GPUs of one node (BASELINE
configs 4 and 5): distributed ``log_probability`` and posterior mean of ``condition``.

The reference has no multi-device code at all; this is new design for MI355X + RCCL/xGMI:

* the N x N matrix is cut into block columns of width ``nb``; block column ``j`` lives on rank
  ``j mod G`` (cyclic ownership keeps the N^3/3 work balanced to within one block and stays
  "1-D block-column").  A rank holds its block columns side by side in one column-major device
  matrix (``csrc/dist.hip``);
* step ``k``: the owner factors panel ``k`` with the single-GPU panel chain on its priority
  stream and packs it into a ring slot; the slot is **broadcast** (``ncclBroadcast`` over xGMI, issued by the
  library itself on that stream: ``csrc/comm.hip``, :class:`tinygp_amd.comm.RcclComm`); every rank updates its own
  block columns with ONE MFMA launch over all of them;
* look-ahead of depth 2 (round 3): the CHAIN PIPELINE -- arrival of panel ``k`` -> gate (panel ``k`` applied to
  block column ``k+1``) -> chain of panel ``k+1`` -> pack -> broadcast -- lives on the owner's priority stream
  and depends on the main stream only through the small *pre-update* that brought block column ``k+1`` up to
  panel ``k-1``; the big updates of steps ``k-1`` and ``k`` may still be running (three ring slots: a slot is
  rewritten only behind the readers of the panel it held three steps earlier);
* a panel is packed and broadcast in column CHUNKS (contiguous in the column-major slot): a chunk is final
  while the chain still factors the columns to its right, so the transfer of a big panel overlaps its own
  factorisation instead of following it;
* ``log_probability`` needs no other exchange: every rank receives every panel, so the forward
  substitution of the (replicated) right-hand side and ``sum log L_ii`` run redundantly on each
  rank straight from the received panels, on a side stream, under the updates;
* ``condition`` mean (config 5): backward substitution block by block on the owners, each
  solved ``nb``-slice broadcast to all; then every rank evaluates ``K(X*, X_owned) alpha_owned``
  (fused, K* never formed) and ONE all-reduce of the (M,) vector finishes the job.

One process per GPU.  The schedule below is written against two small interfaces: the per-rank operations --
:class:`HipBlockOps` (the product: ``tgp_dist_*`` of ``libtgp_hip.so`` on plain device buffers of the library) and, in
``tests/`` only, a NumPy stand-in that lets the same schedule run under ``gloo`` on CPUs -- and the collectives
(:mod:`tinygp_amd.comm`: RCCL from the C ABI; torch is not on the data path and not needed at all with
``RcclComm.from_env`` / ``from_file``).

Stream contract: a collective is enqueued ON a stream of the library and is ordered like a kernel there; every rank
issues the broadcast of a chunk on its PANEL stream -- the owner behind the pack of that chunk, a receiver behind the last
readers of the slot (``slot_ready``); ``work.wait(stream)`` makes a stream wait for the broadcast (an event, no host
block): the owner of the NEXT panel waits on its PANEL stream (gate + chain), everyone on the MAIN stream (forward step,
updates).  Reductions of the resident-factor solves run on the MAIN stream, in order with the kernels around them.
"""

from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np

from tinygp_amd import _ffi
from tinygp_amd._device import MAX_DIM

__all__ = ["HipBlockOps", "BlockCyclicCholesky"]

MAIN, PANEL = 0, 1


class DevBuf:
    """A device buffer of the library (``tgp_malloc``) or a contiguous view into one: ``count`` elements of ``dtype`` at
    ``ptr``, logically ``shape`` (row-major).  Owning buffers go back to their :class:`HipBlockOps`' pool when the last
    reference dies; everything that touches them is ordered on the driver's MAIN stream, so re-use is stream-ordered."""

    __slots__ = ("ops", "ptr", "shape", "dtype", "code", "count", "_own", "_base")

    def __init__(self, ops, ptr, shape, dtype, own=False, base=None):
        self.ops, self.ptr, self.shape, self.dtype = ops, int(ptr), tuple(shape), np.dtype(dtype)
        self.code = _ffi.dtype_code(self.dtype)
        self.count = int(np.prod(self.shape)) if self.shape else 1
        self._own, self._base = own, base  # a view keeps its base alive

    @property
    def nbytes(self) -> int:
        return self.count * self.dtype.itemsize

    def rows(self, r0: int, r1: int) -> "DevBuf":
        """Rows [r0, r1) of the leading axis (contiguous in a row-major buffer)."""
        per = self.count // self.shape[0]
        return DevBuf(self.ops, self.ptr + r0 * per * self.dtype.itemsize, (r1 - r0,) + self.shape[1:], self.dtype,
                      base=self)

    def flat(self, e0: int, e1: int) -> "DevBuf":
        return DevBuf(self.ops, self.ptr + e0 * self.dtype.itemsize, (e1 - e0,), self.dtype, base=self)

    def __del__(self):  # pragma: no cover
        if self._own:
            try:
                self.ops._release(self.ptr, self.nbytes)
            except Exception:
                pass


class HipBlockOps:
    """One rank's device state: a ``tgp_dist`` handle + the plain device buffers its collectives send."""

    def __init__(self, device: int):
        self.ctx = _ffi.Ctx(device=device)
        self.lib = _ffi.lib()
        self.device = device
        self.h = None
        self._pool = {}   # nbytes -> [ptr, ...]: released buffers (every use is ordered on the MAIN stream)
        self._pool_bytes = 0

    # -- buffers ---------------------------------------------------------------------------
    POOL_LIMIT = 32 << 30  # (an MI355X has 288 GB; the gradient at N = 131 072 cycles through four 2-GB buffers per chunk)

    def _alloc(self, shape, dtype=None, zero=False) -> DevBuf:
        dtype = self.dtype if dtype is None else np.dtype(dtype)
        nbytes = max(int(np.prod(shape)) * dtype.itemsize, 8)
        free = self._pool.get(nbytes)
        if free:
            ptr = free.pop()
            self._pool_bytes -= nbytes
        else:
            ptr = self.ctx.malloc(nbytes)
        buf = DevBuf(self, ptr, shape, dtype, own=True)
        if zero:
            _ffi.check(self.lib.tgp_stream_memset(self.ctx.handle, MAIN, C.c_void_p(ptr), 0, buf.nbytes),
                       "tgp_stream_memset")
        return buf

    def _release(self, ptr: int, nbytes: int):
        nbytes = max(nbytes, 8)
        if self.ctx.handle is None:
            return
        if self._pool_bytes + nbytes > self.POOL_LIMIT:
            self.ctx.free(ptr)  # (joins the main stream first)
            return
        self._pool.setdefault(nbytes, []).append(ptr)
        self._pool_bytes += nbytes

    # -- set-up ------------------------------------------------------------------------
    def setup(self, P: np.ndarray, noise_diag: np.ndarray, nb: int, world: int, rank: int):
        lib = self.lib
        n, d = P.shape
        self.dtype = np.dtype(P.dtype)
        self.nb, self.n = nb, n
        nslot = lib.tgp_dist_slot_elems(n, nb)
        self.nd = (nb // 128) * 2048
        self.npad = -(-n // nb) * nb
        self.nloc = len(range(rank, self.npad // nb, world))
        self.ring = [self._alloc((nslot,)) for _ in range(3)]
        self.x = self._alloc((self.npad,), zero=True)
        _ffi.check(lib.tgp_stream_sync(self.ctx.handle, MAIN), "tgp_stream_sync")
        h = C.c_void_p()
        _ffi.check(lib.tgp_dist_create(self.ctx.handle, _ffi.dtype_code(P.dtype), n, d, _ffi.ptr(P),
                                       _ffi.ptr(noise_diag), nb, world, rank,
                                       C.c_void_p(self.ring[0].ptr), C.c_void_p(self.ring[1].ptr),
                                       C.c_void_p(self.ring[2].ptr), C.c_void_p(self.x.ptr), C.byref(h)), "tgp_dist_create")
        self.h = h

    def slot(self, k: int, rows: int):
        """The broadcast buffer of panel k: [rows x nb panel, ld = rows | dinv]."""
        return self.ring[k % 3].flat(0, rows * self.nb + self.nd)

    def slot_chunk(self, k: int, rows: int, c: int, nch: int):
        """Column chunk c of nch of that buffer (contiguous); the last one carries the inverses."""
        cw = self.nb // nch
        end = rows * self.nb + self.nd if c == nch - 1 else (c + 1) * cw * rows
        return self.ring[k % 3].flat(c * cw * rows, end)

    def nbytes_of(self, buf) -> int:
        return buf.nbytes

    def x_slice(self, k: int):
        return self.x.rows(k * self.nb, (k + 1) * self.nb)

    def empty_vec(self, m: int):
        return self._alloc((m,))

    # -- per-step calls (all asynchronous) ---------------------------------------------------
    def assemble(self, prog):
        kp, nops = _ffi.as_kprog(prog)
        _ffi.check(self.lib.tgp_dist_assemble(self.h, kp, nops), "tgp_dist_assemble")

    def load_matrix(self, K: np.ndarray):
        """This rank's block columns of a host matrix (N, N), symmetric, noise included -- instead of ``assemble``."""
        K = np.ascontiguousarray(K, dtype=self.dtype)
        if K.shape != (self.n, self.n):
            raise ValueError("covariance must have shape (N, N)")
        _ffi.check(self.lib.tgp_dist_load_matrix(self.h, _ffi.ptr(K)), "tgp_dist_load_matrix")

    def begin(self, resid):
        _ffi.check(self.lib.tgp_dist_begin(self.h, _ffi.ptr(resid)), "tgp_dist_begin")

    def first_panel(self):
        _ffi.check(self.lib.tgp_dist_first_panel(self.h), "tgp_dist_first_panel")

    def panel_chunk(self, k: int, c: int, nch: int):
        _ffi.check(self.lib.tgp_dist_panel_chunk(self.h, k, c, nch), "tgp_dist_panel_chunk")

    def slot_ready(self, k: int):
        _ffi.check(self.lib.tgp_dist_slot_ready(self.h, k), "tgp_dist_slot_ready")

    def lookahead(self, k: int):
        _ffi.check(self.lib.tgp_dist_lookahead(self.h, k), "tgp_dist_lookahead")

    def arrived(self, k: int):
        _ffi.check(self.lib.tgp_dist_arrived(self.h, k), "tgp_dist_arrived")

    def pre_update(self, k: int):
        _ffi.check(self.lib.tgp_dist_pre_update(self.h, k), "tgp_dist_pre_update")

    def fwd_step(self, k: int):
        _ffi.check(self.lib.tgp_dist_fwd_step(self.h, k), "tgp_dist_fwd_step")

    def rest(self, k: int):
        _ffi.check(self.lib.tgp_dist_rest(self.h, k), "tgp_dist_rest")

    def end(self):
        info, ss, ld = C.c_int32(), C.c_double(), C.c_double()
        _ffi.check(self.lib.tgp_dist_end(self.h, C.byref(info), C.byref(ss), C.byref(ld)), "tgp_dist_end")
        return int(info.value), ss.value, ld.value

    def bwd_step(self, k: int):
        _ffi.check(self.lib.tgp_dist_bwd_step(self.h, k), "tgp_dist_bwd_step")

    def cond_mean_partial(self, prog, Pt: np.ndarray):
        kp, nops = _ffi.as_kprog(prog)
        out = self.empty_vec(Pt.shape[0])
        _ffi.check(self.lib.tgp_dist_cond_mean_partial(self.h, kp, nops, Pt.shape[0], _ffi.ptr(Pt),
                                                       C.c_void_p(out.ptr)),
                   "tgp_dist_cond_mean_partial")
        return out

    # -- solves on the resident factor (buffers: device tensors, (npad,) or (npad, nrhs) row-major) ------------
    def rhs_zeros(self, nrhs: int):
        return self._alloc((self.npad,) if nrhs == 1 else (self.npad, nrhs), zero=True)

    def rhs_from_host(self, Y: np.ndarray, nrhs: int):
        """(n,) or (n, R) host array -> zero-padded device buffer of `nrhs` (1 or a multiple of 128) columns."""
        buf = np.zeros((self.npad,) if nrhs == 1 else (self.npad, nrhs), dtype=self.dtype)
        if nrhs == 1:
            buf[: self.n] = Y.reshape(self.n)
        else:
            buf[: self.n, : Y.shape[1]] = Y
        out = self._alloc(buf.shape)
        _ffi.check(self.lib.tgp_stream_h2d(self.ctx.handle, MAIN, C.c_void_p(out.ptr), _ffi.ptr(buf), buf.nbytes),
                   "tgp_stream_h2d")
        return out

    def rhs_to_host(self, buf) -> np.ndarray:
        """Behind everything on the MAIN stream (the collective that produced the buffer included)."""
        out = np.empty(buf.shape, dtype=buf.dtype)
        _ffi.check(self.lib.tgp_stream_d2h(self.ctx.handle, MAIN, _ffi.ptr(out), C.c_void_p(buf.ptr), out.nbytes),
                   "tgp_stream_d2h")
        return out

    def rhs_block(self, buf, k: int):
        return buf.rows(k * self.nb, (k + 1) * self.nb)

    def fwd_block(self, k: int, nrhs: int, y, acc, x):
        _ffi.check(self.lib.tgp_dist_fwd_block(self.h, k, nrhs, C.c_void_p(y.ptr), C.c_void_p(acc.ptr),
                                               C.c_void_p(x.ptr)), "tgp_dist_fwd_block")

    def bwd_block(self, k: int, x):
        _ffi.check(self.lib.tgp_dist_bwd_block(self.h, k, C.c_void_p(x.ptr)), "tgp_dist_bwd_block")

    def xloc_zeros(self, nrhs: int):
        """This rank's solved blocks side by side (local column l at rows l * nb ..): the left-looking forward solve's operand."""
        return self._alloc((max(self.nloc, 1) * self.nb, nrhs), zero=True)

    def fwd_partial(self, k: int, nrhs: int, xloc, acc, first: int):
        _ffi.check(self.lib.tgp_dist_fwd_partial(self.h, k, nrhs, C.c_void_p(xloc.ptr), C.c_void_p(acc.ptr), first),
                   "tgp_dist_fwd_partial")

    def fwd_solve_left(self, k: int, nrhs: int, y, acc, x, xloc):
        _ffi.check(self.lib.tgp_dist_fwd_solve_left(self.h, k, nrhs, C.c_void_p(y.ptr), C.c_void_p(acc.ptr),
                                                    C.c_void_p(x.ptr), C.c_void_p(xloc.ptr)), "tgp_dist_fwd_solve_left")

    def bwd_block_multi(self, k: int, nrhs: int, x, yloc):
        _ffi.check(self.lib.tgp_dist_bwd_block_multi(self.h, k, nrhs, C.c_void_p(x.ptr), C.c_void_p(yloc.ptr)),
                   "tgp_dist_bwd_block_multi")

    def bwd_update_multi(self, k: int, nrhs: int, x, yloc, stop: int):
        _ffi.check(self.lib.tgp_dist_bwd_update_multi(self.h, k, nrhs, C.c_void_p(x.ptr), C.c_void_p(yloc.ptr), stop),
                   "tgp_dist_bwd_update_multi")

    def gather_owned(self, x, nrhs: int, world: int):
        """This rank's blocks of a global (n_pad, nrhs) buffer side by side (world size 1: the buffer itself)."""
        if world == 1:
            return x
        yloc = self._alloc((max(self.nloc, 1) * self.nb, nrhs))
        _ffi.check(self.lib.tgp_dist_gather_owned(self.h, nrhs, C.c_void_p(x.ptr), C.c_void_p(yloc.ptr)),
                   "tgp_dist_gather_owned")
        return yloc

    def rhs_identity(self, c0: int, nrhs: int):
        out = self._alloc((self.npad, nrhs))
        _ffi.check(self.lib.tgp_dist_identity_cols(self.h, c0, nrhs, C.c_void_p(out.ptr)), "tgp_dist_identity_cols")
        return out

    # -- gradient accumulators (this rank's partial sums; the driver all-reduces them) ------------------------------
    def grad_begin(self, prog):
        kp, nops = _ffi.as_kprog(prog)
        self._grad_nops = nops
        _ffi.check(self.lib.tgp_dist_grad_begin(self.h, kp, nops), "tgp_dist_grad_begin")

    def grad_chunk(self, c0: int, nrhs: int, kcols, with_logscale: bool):
        _ffi.check(self.lib.tgp_dist_grad_chunk(self.h, c0, nrhs, C.c_void_p(kcols.ptr), int(with_logscale)),
                   "tgp_dist_grad_chunk")

    def grad_end(self, d: int):
        """(partial sums as a device buffer of 2 * nops + d float64 for the all-reduce, diag(K^-1) on the host)"""
        gp_ = (C.c_double * (2 * self._grad_nops))()
        # the library writes one double per INPUT DIMENSION of the data into grad_logscale whenever the pointer is
        # non-NULL -- whatever `d` the caller wants back (advisor r5: an array of max(d, 1) was overrun by D >= 2
        # inputs without a covering transform).  NULL when no log-scale gradient is wanted, TGP_MAX_DIM doubles else.
        gl = (C.c_double * MAX_DIM)() if d > 0 else None
        diag = np.empty(self.n, dtype=self.dtype)
        _ffi.check(self.lib.tgp_dist_grad_end(self.h, gp_, gl, _ffi.ptr(diag)), "tgp_dist_grad_end")
        part = np.array(list(gp_) + (list(gl)[:d] if d > 0 else []), dtype=np.float64)
        buf = self._alloc((part.size,), np.float64)
        _ffi.check(self.lib.tgp_stream_h2d(self.ctx.handle, MAIN, C.c_void_p(buf.ptr), _ffi.ptr(part), part.nbytes),
                   "tgp_stream_h2d")
        return buf, diag

    def trmv_partial(self, y):
        out = self.rhs_zeros(1)
        _ffi.check(self.lib.tgp_dist_trmv_partial(self.h, C.c_void_p(y.ptr), C.c_void_p(out.ptr)),
                   "tgp_dist_trmv_partial")
        return out

    def cross_cov(self, prog, Pt: np.ndarray, m_pad: int):
        kp, nops = _ffi.as_kprog(prog)
        out = self._alloc((self.npad, m_pad))
        _ffi.check(self.lib.tgp_dist_cross_cov(self.h, kp, nops, Pt.shape[0], _ffi.ptr(Pt), m_pad,
                                               C.c_void_p(out.ptr)), "tgp_dist_cross_cov")
        return out

    def colsumsq_owned(self, nrhs: int, x):
        out = self._alloc((nrhs,))
        _ffi.check(self.lib.tgp_dist_colsumsq_owned(self.h, nrhs, C.c_void_p(x.ptr), C.c_void_p(out.ptr)),
                   "tgp_dist_colsumsq_owned")
        return out

    def gram_owned(self, nrhs: int, x):
        out = self._alloc((nrhs, nrhs))
        _ffi.check(self.lib.tgp_dist_gram_owned(self.h, nrhs, C.c_void_p(x.ptr), C.c_void_p(out.ptr)),
                   "tgp_dist_gram_owned")
        return out

    def gram_pair_owned(self, ni: int, xi, nj: int, xj):
        """(nj, ni) host-readable view of the (ni x nj) column-major share of x_i^T x_j (rhs_to_host reads row-major)."""
        out = self._alloc((nj, ni))
        _ffi.check(self.lib.tgp_dist_gram_pair_owned(self.h, ni, C.c_void_p(xi.ptr), nj, C.c_void_p(xj.ptr),
                                                     C.c_void_p(out.ptr)), "tgp_dist_gram_pair_owned")
        return out

    def set_x(self, buf):
        """The handle's own replicated vector <- a solved vector (the backward substitution works in place there)."""
        _ffi.check(self.lib.tgp_stream_d2d(self.ctx.handle, MAIN, C.c_void_p(self.x.ptr), C.c_void_p(buf.ptr),
                                           self.x.nbytes), "tgp_stream_d2d")

    def abort(self):
        if self.h is not None:
            self.lib.tgp_dist_abort(self.h)

    def column(self, l: int, rows: int) -> np.ndarray:
        out = np.empty((self.nb, rows), dtype=self.dtype)  # column-major (rows x nb)
        _ffi.check(self.lib.tgp_dist_get_column(self.h, l, _ffi.ptr(out)), "tgp_dist_get_column")
        return out.T

    def close(self):
        if self.h is not None:
            self.lib.tgp_dist_destroy(self.h)  # (joins every stream of the context)
            self.h = None
            for ptrs in self._pool.values():
                for p in ptrs:
                    self.ctx.free(p)
            self._pool, self._pool_bytes = {}, 0

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


# One default communicator per (context, process group): ncclCommInitRank is a collective with a TCP id exchange and tens
# of MB of communicator memory -- a solver per hyper-parameter point must not pay (or leak) it every time (advisor r5).
_DEFAULT_COMMS: dict = {}


def _shared_default_comm(ops, dist, group):
    key = (id(ops.ctx), id(group) if group is not None else None, None if dist is None else id(dist))
    ent = _DEFAULT_COMMS.get(key)
    if ent is None:
        ent = [BlockCyclicCholesky._make_default_comm(ops, dist, group), 0, key]
        _DEFAULT_COMMS[key] = ent
    ent[1] += 1
    ent[0]._tgp_default_key = key
    return ent[0]


def _release_default_comm(comm):
    key = getattr(comm, "_tgp_default_key", None)
    ent = _DEFAULT_COMMS.get(key)
    if ent is None or ent[0] is not comm:  # (not from the cache: the CPU stand-in's TorchComm has nothing to release)
        if hasattr(comm, "close") and key is None and not isinstance(comm, type(None)):
            try:
                comm.close()
            except Exception:
                pass
        return
    ent[1] -= 1
    if ent[1] <= 0:
        del _DEFAULT_COMMS[key]
        if hasattr(comm, "close"):
            comm.close()


class BlockCyclicCholesky:
    """Distributed dense GP: ``log_probability`` and the posterior mean of ``condition``.

    Args:
        kernel: a :mod:`tinygp_amd.kernels` tree.
        X: (N,) or (N, D) coordinates, replicated on every rank (<= a few MB).
        noise_diag: (N,) noise variances (``noise.Diagonal``, reference noise.py:55-95).
        nb: block-column width (multiple of 128).
        ops: per-rank operations (default :class:`HipBlockOps` on ``LOCAL_RANK``).
        comm: the collectives (:mod:`tinygp_amd.comm`).  Default for the HIP operations: :class:`RcclComm` -- its
            128-byte id exchanged through ``dist`` / an initialised ``torch.distributed`` when there is one (that ONE
            message; a ``gloo`` group selects the host-staged test transport instead), else over TCP from the launcher's
            ``RANK`` / ``WORLD_SIZE`` / ``MASTER_ADDR`` / ``MASTER_PORT`` with no torch at all.  Default for stand-in
            operations: ``torch.distributed`` on their tensors.
        group, dist: ``torch.distributed`` process group / module for the above (default: the world).
    """

    def __init__(self, kernel, X, noise_diag, *, nb: int = 1024, ops=None, group=None, dist=None, comm=None,
                 covariance=None):
        if nb % 128 or nb <= 0:
            raise ValueError("nb must be a positive multiple of 128")
        self._own_ops = ops is None
        if ops is None:
            ops = HipBlockOps(int(os.environ.get("LOCAL_RANK", "0")))
        self._own_comm = comm is None  # a communicator this object made: close() releases it, BEFORE the operations
        if comm is None:
            comm = self._default_comm(ops, dist, group)
        self.comm = comm
        self.rank, self.G = comm.rank, comm.world
        X = np.asarray(X)
        P = np.ascontiguousarray(X[:, None] if X.ndim == 1 else X)
        self.dtype = np.dtype(np.float32 if P.dtype == np.float32 else np.float64)
        P = P.astype(self.dtype)
        self.n, self.d = P.shape
        self.nb = nb
        self.nblk = math.ceil(self.n / nb)
        self.npad = self.nblk * nb
        self.kernel = kernel
        self.prog = kernel.program()
        # the seam's `covariance=` argument (reference solvers/direct.py:36,50-52): a host matrix, noise included, the same
        # on every rank -- each uploads its own block columns instead of evaluating the kernel (round 6)
        self._cov = None
        if covariance is not None:
            cov = np.ascontiguousarray(covariance, dtype=self.dtype)
            if cov.shape != (self.n, self.n):
                raise ValueError("covariance must have shape (N, N)")
            self._cov = cov
        self.ops = ops
        self.owned = [j for j in range(self.nblk) if j % self.G == self.rank]
        diag = np.ascontiguousarray(np.broadcast_to(noise_diag, (self.n,)), dtype=self.dtype)
        ops.setup(P, diag, nb, self.G, self.rank)
        self.info = 0
        self.factored = self.solved = self.have_alpha = False
        self._resid = None
        self._err = None
        self.bytes_received = 0  # panel bytes this rank received in the last factorisation
        # A process group of one still sends its panels through the collective (the same code path as
        # with peers; 30.9 vs 30.8 ms at N = 16 384 since the driver keeps to three streams of its own
        # -- with five in use plus RCCL's it was 39.5 ms, profiles/r02_m_stream_count.txt).
        # TGP_DIST_SELF_BROADCAST=0 skips the call.
        self.self_broadcast = os.environ.get("TGP_DIST_SELF_BROADCAST", "1") != "0"

    def rows(self, j: int) -> int:
        return self.npad - j * self.nb

    def owner(self, j: int) -> int:
        return j % self.G

    def close(self, close_ops: bool | None = None):
        """Release what this object created, in dependency order: the communicator (its destroy drains the context's
        streams under the context's lock) first, then the operations -- never left to ``__del__`` order at interpreter
        exit, where the context could go first (advisor r5).  A default RCCL communicator is shared by every driver of
        the same context and process group and is released with the LAST of them.  Operations the CALLER passed in are
        closed only on request (``close_ops=True``)."""
        comm, self.comm = getattr(self, "comm", None), None
        if comm is not None and getattr(self, "_own_comm", False):
            _release_default_comm(comm)
        ops = getattr(self, "ops", None)
        if ops is not None and hasattr(ops, "close") and (getattr(self, "_own_ops", False) if close_ops is None else close_ops):
            ops.close()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _default_comm(ops, dist, group):
        from tinygp_amd import comm as _comm

        if not isinstance(ops, HipBlockOps):  # the CPU stand-in of the tests: torch tensors under gloo
            return _comm.TorchComm(dist, group)
        return _shared_default_comm(ops, dist, group)

    @staticmethod
    def _make_default_comm(ops, dist, group):
        from tinygp_amd import comm as _comm

        if dist is None:
            import sys

            td = sys.modules.get("torch.distributed")  # (never imported for this: a process without torch stays so)
            if td is not None and td.is_available() and td.is_initialized():
                dist = td
        if dist is None:
            return _comm.RcclComm.from_env(ops.ctx)
        if str(dist.get_backend(group)).lower() == "gloo":
            return _comm.HostStagedComm(ops.ctx, dist, group)
        return _comm.RcclComm.from_torch(ops.ctx, dist, group)

    # -- the schedule -----------------------------------------------------------------------
    CHUNK_MIN_BYTES = 32 << 20  # a panel is sent in pieces once a piece is at least this big

    def chunks(self, k: int) -> int:
        """Column chunks of panel k's broadcast: the same on every rank (a function of shapes only)."""
        per = self.rows(k) * self.nb * self.dtype.itemsize
        nblk_p = self.nb // 128
        nch = 1
        while nch * 2 <= min(4, nblk_p) and nblk_p % (nch * 2) == 0 and per // (nch * 2) >= self.CHUNK_MIN_BYTES:
            nch *= 2
        return nch

    def _guard(self, fn, *a):
        """A rank-local failure (a HIP error, the device-side hand-off timeout) must not leave the peers hanging
        in the next collective: it is remembered, this rank keeps issuing its collectives on whatever the buffers
        hold, and every rank raises together behind the agreed all-reduce at the end of the factorisation."""
        if self._err is None:
            try:
                return fn(*a)
            except Exception as e:  # noqa: BLE001
                self._err = e
        return None

    def _bcast_panel(self, k: int):
        """Panel k from its owner to everyone, chunk by chunk under the PANEL stream; the owner factors and packs
        each chunk right in front of its broadcast.  Returns the work handles."""
        own = self.owner(k) == self.rank
        nch = self.chunks(k)
        if not own:
            self._guard(self.ops.slot_ready, k)
            self.bytes_received += self.ops.nbytes_of(self.ops.slot(k, self.rows(k)))
        works = []
        for c in range(nch):
            if own:
                self._guard(self.ops.panel_chunk, k, c, nch)
            if self.G == 1 and not self.self_broadcast:
                works.append(self.comm.marker(PANEL))  # nobody to send to: only the stream dependency remains
                continue
            buf = self.ops.slot_chunk(k, self.rows(k), c, nch)
            works.append(self.comm.broadcast(buf, self.owner(k), PANEL))  # on the PANEL stream, behind the pack
        return works

    def factor(self, resid=None, kernel=None) -> int:
        """Assemble K + noise and factor it; with ``resid`` (= y - mean) also ``L^-1 resid``,
        panel by panel as the panels arrive.  Returns the potrf info agreed by all ranks."""
        if kernel is not None:
            if self._cov is not None:
                raise NotImplementedError("this driver factors the covariance matrix it was given: a new kernel needs a new "
                                          "matrix (build another solver)")
            self.kernel, self.prog = kernel, kernel.program()
        ops = self.ops
        r = None
        if resid is not None:
            r = np.ascontiguousarray(np.broadcast_to(resid, (self.n,)), dtype=self.dtype)
        self.bytes_received = 0
        self._err = None
        if self._cov is not None:
            self._guard(ops.load_matrix, self._cov)  # this rank's block columns of the host matrix (covariance=)
        else:
            self._guard(ops.assemble, self.prog)
        self._guard(ops.begin, r)
        works = self._bcast_panel(0)   # owner of 0: its chain branches off behind the assembly of block column 0
        self._guard(ops.first_panel)   # everyone: the other block columns, beside that chain
        for k in range(self.nblk):
            # -- priority stream: the chain pipeline.  Only the owner of k+1 needs panel k here.
            nxt = None
            if k + 1 < self.nblk:
                if self.owner(k + 1) == self.rank:
                    for w in works:
                        w.wait(PANEL)  # a stream dependency, not a host block
                    self._guard(ops.lookahead, k)  # gate: panel k -> block column k+1
                nxt = self._bcast_panel(k + 1)     # (owner: chain + pack per chunk, each right before its send)
            # -- main stream: forward step, then the updates; block column k+2 first on its owner
            for w in works:
                w.wait(MAIN)
            self._guard(ops.arrived, k)
            self._guard(ops.fwd_step, k)
            self._guard(ops.pre_update, k)
            self._guard(ops.rest, k)
            works = nxt
        out = self._guard(ops.end)
        info, self._sumsq, self._logdet = out if out is not None else (0, math.nan, math.nan)
        # agree on the first failing pivot (LAPACK convention), 0 if none; a rank-local error travels as -1
        v = self.comm.agree_min(-1.0 if self._err is not None else (float(info) if info else float(2**52)))
        if v < 0:
            # every rank: join the streams (broadcasts and kernels of the interrupted pass may still be in flight on
            # the ring slots and on x) and forget its markers before raising, so that a retry starts from a quiet
            # device (round-3 advisor finding)
            self.factored = self.solved = self.have_alpha = False
            self._resid = None
            abort = getattr(ops, "abort", None)
            if abort is not None:
                abort()
            if self._err is not None:
                raise self._err
            raise _ffi.TgpError("block-column driver: another rank failed during the factorisation")
        self.info = 0 if v >= 2**52 else int(v)
        self.factored, self.solved, self.have_alpha = True, resid is not None, False
        self._resid = None if r is None else r.copy()  # the right-hand side the cached solves belong to
        return self.info

    def log_probability(self, resid, kernel=None) -> float:
        """``-0.5 |L^-1 r|^2 - sum log L_ii - n/2 log(2 pi)`` (reference gp.py:313-320,
        solvers/direct.py:61-64); ``-inf`` when not finite (gp.py:316).  One fused pass:
        assembly, factorisation and forward solve."""
        self.factor(resid, kernel)
        ll = -0.5 * self._sumsq - (self._logdet + 0.5 * self.n * math.log(2.0 * math.pi))
        if self.info or not math.isfinite(ll):
            return -math.inf
        return ll

    # -- solves on the resident factor (reference solvers/direct.py:66-73, 75-95) -------------------------------
    def _reduce_to_owner(self, buf, k: int):
        if self.G == 1 and not self.self_broadcast:
            return
        self.comm.reduce(buf, self.owner(k), MAIN)

    def _all_reduce(self, buf):
        if self.G == 1 and not self.self_broadcast:
            return
        self.comm.all_reduce(buf, MAIN)

    def _forward(self, y_dev, nrhs: int, first: int = 0, want_loc: bool = False):
        """``L^-1 Y`` for device right-hand sides (``nrhs`` = 1 or a multiple of 128), fan-in: block by block the
        accumulators' slice is REDUCED to the block's owner (north_star's reduce of the solve RHS: nb x nrhs entries
        per block), the owner solves its block and turns it into updates of the rows below -- its own column is all it
        needs.  Returns the solved buffer, zero outside the OWNED blocks (``_all_reduce`` replicates it)."""
        ops = self.ops
        acc, x = ops.rhs_zeros(nrhs), ops.rhs_zeros(nrhs)
        if nrhs > 1 and self._forward_left():
            # LEFT-looking (round 5): every rank forms its share of block row k's sum from the blocks it has solved itself --
            # the work of a step is spread over the ranks, and only the owner's next product waits for the reduce chain
            xloc = ops.xloc_zeros(nrhs)
            for k in range(first, self.nblk):
                ops.fwd_partial(k, nrhs, xloc, acc, first)
                self._reduce_to_owner(ops.rhs_block(acc, k), k)
                ops.fwd_solve_left(k, nrhs, y_dev, acc, x, xloc)
            return (x, xloc) if want_loc else x
        for k in range(first, self.nblk):  # (`first`: right-hand sides that are zero above that block -- identity columns)
            self._reduce_to_owner(ops.rhs_block(acc, k), k)
            ops.fwd_block(k, nrhs, y_dev, acc, x)
        return (x, None) if want_loc else x

    FORWARD = os.environ.get("TGP_DIST_FORWARD", "auto")  # "left" | "right" | "auto" (left with peers, right alone)

    def _forward_left(self) -> bool:
        """Right-looking: the owner of block column k alone updates every row below (big products: the better form on ONE
        GPU, serial across ranks).  Left-looking: balanced over the ranks (small outputs, long k-ranges: split-k)."""
        mode = self.FORWARD
        return mode == "left" or (mode == "auto" and self.G > 1)

    def _backward(self, x, nrhs: int, stop: int = 0, yloc=None):
        """``L^-T Y`` in place for ``nrhs`` (a multiple of 128) right-hand sides, right-looking: block ``k`` from the last
        down to ``stop`` -- its owner solves ``X_k = L_kk^-T Y_k``, ONE ``nb x nrhs`` broadcast replicates it, and every
        rank subtracts ``L[k, i]^T X_k`` from the blocks ``i < k`` it owns (reference solvers/direct.py:66-68 with y (N, R);
        VERDICT r4: "trsm per block, with one nb x R broadcast").  On entry block ``k`` of ``x`` must be valid on ITS
        owner -- what :meth:`_forward` leaves, or any replicated buffer (``yloc``: the same blocks side by side, if the
        caller has them: the left-looking forward solve does); on return blocks ``>= stop`` of ``x`` are replicated.
        (``stop`` > 0: only those rows are wanted -- the lower triangle of a chunk of K^-1.)"""
        ops = self.ops
        if yloc is None:  # the rank's own blocks side by side: every step's update is ONE product on them
            yloc = ops.gather_owned(x, nrhs, self.G)
        for k in reversed(range(stop, self.nblk)):
            ops.bwd_block_multi(k, nrhs, x, yloc)
            if not (self.G == 1 and not self.self_broadcast):
                self.comm.broadcast(ops.rhs_block(x, k), self.owner(k), MAIN).wait(MAIN)
            ops.bwd_update_multi(k, nrhs, x, yloc, stop)
        return x

    def _need_factor(self):
        if not self.factored:
            self.factor()

    def solve_triangular(self, y, *, transpose: bool = False) -> np.ndarray:
        """``L x = y`` / ``L^T x = y`` for a NEW right-hand side, y (N,) or (N, R), on the resident factor: O(N^2 R),
        never a factorisation (reference solvers/direct.py:66-70).  Identical on every rank."""
        self._need_factor()
        y = np.asarray(y)
        if y.ndim not in (1, 2) or y.shape[0] != self.n:
            raise ValueError(f"y must have shape ({self.n},) or ({self.n}, R); got {y.shape}")
        Y = np.ascontiguousarray(y, dtype=self.dtype)
        ops = self.ops
        if transpose and Y.ndim == 2:
            # ONE blocked pass for all right-hand sides (round 5; until round 4: column by column, R x N/nb broadcasts)
            R = Y.shape[1]
            rp = -(-R // 128) * 128
            res = ops.rhs_to_host(self._backward(ops.rhs_from_host(Y, rp), rp))[: self.n, :R]
        elif transpose or Y.ndim == 1:
            cols = [Y] if Y.ndim == 1 else [np.ascontiguousarray(Y[:, r]) for r in range(Y.shape[1])]
            out = []
            for col in cols:  # (the backward substitution is a vector kernel: right-hand sides one by one)
                if transpose:
                    x = ops.rhs_from_host(col, 1)
                    for k in reversed(range(self.nblk)):
                        ops.bwd_block(k, x)
                        if not (self.G == 1 and not self.self_broadcast):
                            self.comm.broadcast(ops.rhs_block(x, k), self.owner(k), MAIN).wait(MAIN)
                else:
                    x = self._forward(ops.rhs_from_host(col, 1), 1)
                    self._all_reduce(x)
                out.append(ops.rhs_to_host(x)[: self.n])
            res = out[0] if Y.ndim == 1 else np.stack(out, axis=1)
        else:
            R = Y.shape[1]
            rp = -(-R // 128) * 128
            x = self._forward(ops.rhs_from_host(Y, rp), rp)
            self._all_reduce(x)
            res = ops.rhs_to_host(x)[: self.n, :R]
        if self.info:
            res = np.full_like(res, np.nan)
        return res

    def dot_triangular(self, y) -> np.ndarray:
        """``L @ y`` (reference solvers/direct.py:72-73): every rank multiplies its own block columns, one all-reduce."""
        self._need_factor()
        y = np.asarray(y)
        Y = np.ascontiguousarray(y.reshape(self.n, -1), dtype=self.dtype)
        out = np.empty_like(Y)
        for r in range(Y.shape[1]):
            part = self.ops.trmv_partial(self.ops.rhs_from_host(np.ascontiguousarray(Y[:, r]), 1))
            self._all_reduce(part)
            out[:, r] = self.ops.rhs_to_host(part)[: self.n]
        if self.info:
            out[:] = np.nan
        return out.reshape(y.shape)

    def _test_points(self, X_test):
        Xt = np.asarray(X_test)
        Pt = np.ascontiguousarray(Xt[:, None] if Xt.ndim == 1 else Xt, dtype=self.dtype)
        if Pt.shape[1] != self.d:
            raise ValueError("X_test must have the same number of input dimensions as X")
        return Pt

    GRAM_BYTES_LIMIT = 96 << 30  # device bytes condition_gram may ask for (a third of an MI355X's 288 GB)
    RHS_CHUNK = 2048  # test points per forward solve of the conditional variance (bounds the (n_pad, chunk) buffers)

    def condition_colsumsq(self, X_test, kernel=None) -> np.ndarray:
        """``colsum(A o A)`` with ``A = L^-1 K(X, X*)`` (reference solvers/direct.py:87-95 without the M x M product):
        the cross covariance is assembled on every rank, forward-solved in chunks of test points, each rank sums the
        squares over the block rows it owns, ONE all-reduce of the (M,) vector.  The posterior variance is
        ``k(x*, x*) - this`` (+ noise)."""
        self._need_factor()
        Pt = self._test_points(X_test)
        prog = self.prog if kernel is None else kernel.program()
        m = Pt.shape[0]
        out = np.empty(m, dtype=self.dtype)
        for m0 in range(0, m, self.RHS_CHUNK):
            part = Pt[m0:m0 + self.RHS_CHUNK]
            mp = -(-part.shape[0] // 128) * 128
            a = self._forward(self.ops.cross_cov(prog, part, mp), mp)
            s = self.ops.colsumsq_owned(mp, a)
            self._all_reduce(s)
            out[m0:m0 + part.shape[0]] = self.ops.rhs_to_host(s)[: part.shape[0]]
        if self.info:
            out[:] = np.nan
        return out

    GRAM_CHUNK = 4096  # test points per forward solve of the conditional covariance

    def condition_gram(self, X_test, kernel=None) -> np.ndarray:
        """``A^T A`` (M, M) with ``A = L^-1 K(X, X*)`` (reference solvers/direct.py:94-95): block rows of A stay on their
        owners, every rank forms its share of the product on the MFMAs, one all-reduce per block of the result.

        Round 6 (VERDICT r5 "missing" 3): in CHUNKS of test points.  Each chunk is forward-solved on the resident factor
        (three (n_pad, chunk) buffers while it runs), its solved columns stay on the device, and block (i, j) of the
        product is the share ``A_i^T A_j`` of every rank, all-reduced -- the call is bounded by ONE (n_pad, M) matrix of
        solved columns + two chunk buffers, a third of what the single pass held (three (n_pad, M) buffers at once),
        and says so instead of running out of memory inside a solve."""
        self._need_factor()
        Pt = self._test_points(X_test)
        prog = self.prog if kernel is None else kernel.program()
        m = Pt.shape[0]
        ch = max(128, self.GRAM_CHUNK // 128 * 128)
        parts = [Pt[m0:m0 + ch] for m0 in range(0, m, ch)]
        pads = [-(-p.shape[0] // 128) * 128 for p in parts]
        need = (self.npad * sum(pads) + 2 * self.npad * max(pads) + max(pads) ** 2) * self.dtype.itemsize
        if need > self.GRAM_BYTES_LIMIT:
            raise MemoryError(
                f"condition covariance at {m} test points needs {need / 2**30:.1f} GiB of device buffers per rank "
                f"(N = {self.n}: one (n_pad, M) matrix of solved columns); ask for the variance (condition_colsumsq / "
                f"predict(return_var=True)), use fewer test points per call, or raise BlockCyclicCholesky.GRAM_BYTES_LIMIT")
        solved = [self._forward(self.ops.cross_cov(prog, p, mp), mp) for p, mp in zip(parts, pads)]
        out = np.empty((m, m), dtype=self.dtype)
        offs = np.cumsum([0] + [p.shape[0] for p in parts])
        for i in range(len(parts)):
            for j in range(i, len(parts)):
                g = self.ops.gram_pair_owned(pads[i], solved[i], pads[j], solved[j])
                self._all_reduce(g)
                blk = self.ops.rhs_to_host(g)  # (pads[j], pads[i]) row-major = block (i, j) transposed
                ni, nj = parts[i].shape[0], parts[j].shape[0]
                out[offs[i]:offs[i] + ni, offs[j]:offs[j] + nj] = blk[:nj, :ni].T
                if j > i:
                    out[offs[j]:offs[j] + nj, offs[i]:offs[i] + ni] = blk[:nj, :ni]
        out = 0.5 * (out + out.T)
        if self.info:
            out = np.full_like(out, np.nan)
        return out

    def alpha(self, resid):
        """``K^-1 resid`` replicated on every rank (reference gp.py:330-334).  Right behind a fused pass over the SAME
        residual the forward solve is already there; any other right-hand side is forward-solved on the resident
        factor (fan-in, O(N^2)) -- never a new factorisation.  The backward substitution walks the block columns from
        the last to the first, the owner solves its ``nb`` slice and broadcasts it."""
        r = np.ascontiguousarray(np.broadcast_to(resid, (self.n,)), dtype=self.dtype)
        self._need_factor()
        same = self.solved and self._resid is not None and np.array_equal(self._resid, r)
        ops = self.ops
        if not same:
            x = self._forward(ops.rhs_from_host(r, 1), 1)
            self._all_reduce(x)
            ops.set_x(x)
            xs = ops.rhs_to_host(x)
            self._sumsq = float(np.sum(np.square(xs.astype(np.float64))))
            self._resid, self.solved, self.have_alpha = r.copy(), True, False
        if not self.have_alpha:
            for k in reversed(range(self.nblk)):
                ops.bwd_step(k)  # owner: x_k on the main stream (needs the slices below it: waited for underneath)
                if self.G == 1 and not self.self_broadcast:
                    continue
                # asynchronous: issued behind bwd_step on the main stream, the next step follows it there
                self.comm.broadcast(ops.x_slice(k), self.owner(k), MAIN).wait(MAIN)
            self.have_alpha = True
        return self.ops.x

    GRAD_CHUNK = 2048  # columns of K^-1 per solve of the gradient (three (n_pad, chunk) buffers in flight)

    def log_probability_and_grad(self, resid, kernel=None, with_logscale: bool = False):
        """``log_probability`` and its gradient on the block-column path -- what ``jax.value_and_grad`` of reference
        ``gp.py:126-138`` gives a caller at any size (VERDICT r4 "what's missing" 2):

            d ll / d theta = 1/2 tr((alpha alpha^T - K^-1) dK/dtheta),      alpha = K^-1 resid.

        One fused factorisation pass gives the value; ``K^-1`` is then visited ``GRAD_CHUNK`` columns at a time on the
        resident factor -- fan-in forward solve of the identity's columns from the chunk's first block, right-looking
        backward solve down to it (only the lower triangle of the chunk is needed: K^-1 is symmetric) --, the chunk ends
        up replicated and every rank contracts ITS block rows with ``dK/dtheta`` evaluated on the fly
        (``tgp_dist_grad_chunk``).  ONE all-reduce of the parameter vector at the end.  4/3 N^3 flops in the solves on
        top of the factorisation's N^3 / 3, spread over the ranks like the factorisation itself.

        Returns ``(ll, grads)`` with ``grads = {"kernel": per-op pairs as DirectSolver.log_probability_and_grad's flat
        list source (2 per op of the program), "noise_diag": (N,), "mean": alpha (N,), "logscale": (D,) or None}``,
        identical on every rank."""
        if self._cov is not None:
            raise NotImplementedError("gradients need the kernel the matrix came from: this driver was given a covariance matrix")
        ll = self.log_probability(resid, kernel)
        ops, n, nb = self.ops, self.n, self.nb
        nops = len(self.prog)
        if self.info or not math.isfinite(ll):
            nanv = np.full(n, np.nan, dtype=self.dtype)
            return -math.inf, {"kernel": [math.nan] * (2 * nops), "noise_diag": nanv, "mean": nanv.copy(),
                               "logscale": np.full(self.d, np.nan) if with_logscale else None}
        alpha = ops.rhs_to_host(self.alpha(resid))[:n].astype(self.dtype, copy=True)  # (K^-1 r stays in the handle's x)
        ops.grad_begin(self.prog)
        R = min(self.GRAD_CHUNK, self.npad)
        R = max(128, R // 128 * 128)
        for c0 in range(0, n, R):
            first = c0 // nb
            z, zloc = self._forward(ops.rhs_identity(c0, R), R, first=first, want_loc=True)  # L^-1 E: block k on owner(k)
            z = self._backward(z, R, stop=first, yloc=zloc)              # K^-1[c0 // nb * nb :, c0 : c0 + R], replicated
            ops.grad_chunk(c0, R, z, with_logscale)
            del z, zloc
        part, kdiag = ops.grad_end(self.d if with_logscale else 0)
        self._all_reduce(part)
        g = ops.rhs_to_host(part)
        gnoise = (0.5 * (alpha.astype(np.float64) ** 2 - kdiag.astype(np.float64))).astype(self.dtype)
        return ll, {"kernel": [float(v) for v in g[: 2 * nops]], "noise_diag": gnoise, "mean": alpha,
                    "logscale": np.asarray(g[2 * nops:], dtype=np.float64) if with_logscale else None}

    def resident_log_probability(self, resid) -> float:
        """``log_probability`` of a new residual on the resident factor: the fan-in forward solve, O(N^2)."""
        r = np.ascontiguousarray(np.broadcast_to(resid, (self.n,)), dtype=self.dtype)
        self._need_factor()
        if not (self.solved and self._resid is not None and np.array_equal(self._resid, r)):
            x = self._forward(self.ops.rhs_from_host(r, 1), 1)
            self._all_reduce(x)
            self.ops.set_x(x)
            xs = self.ops.rhs_to_host(x)
            self._sumsq = float(np.sum(np.square(xs.astype(np.float64))))
            self._resid, self.solved, self.have_alpha = r.copy(), True, False
        elif self.have_alpha:  # x holds K^-1 r by now: |L^-1 r|^2 was kept
            pass
        ll = -0.5 * self._sumsq - (self._logdet + 0.5 * self.n * math.log(2.0 * math.pi))
        if self.info or not math.isfinite(ll):
            return -math.inf
        return ll

    def normalization(self) -> float:
        """``sum log L_ii + N/2 log 2 pi`` (reference solvers/direct.py:61-64)."""
        self._need_factor()
        return math.nan if self.info else self._logdet + 0.5 * self.n * math.log(2.0 * math.pi)

    def cond_mean_from_alpha(self, alpha, Pt, kernel=None) -> np.ndarray:
        """``K(X*, X) alpha`` for a replicated host vector ``alpha`` (reference gp.py:357 via kernels/base.py:68-82)."""
        a = np.ascontiguousarray(np.broadcast_to(alpha, (self.n,)), dtype=self.dtype)
        self.ops.set_x(self.ops.rhs_from_host(a, 1))
        self.solved = self.have_alpha = False  # x no longer belongs to the cached residual
        self._resid = None
        prog = self.prog if kernel is None else kernel.program()
        part = self.ops.cond_mean_partial(prog, Pt)
        self._all_reduce(part)
        out = self.ops.rhs_to_host(part)
        if self.info:
            out = np.full_like(out, np.nan)
        return out

    def condition_mean(self, resid, X_test, kernel=None) -> np.ndarray:
        """Posterior mean ``K(X*, X) K^-1 resid`` at the test points (reference gp.py:353-359 with
        ``include_mean=False``; add ``mean(X*)`` on the host), identical on every rank."""
        self.alpha(resid)
        Pt = self._test_points(X_test)
        prog = self.prog if kernel is None else kernel.program()
        part = self.ops.cond_mean_partial(prog, Pt)
        self.comm.all_reduce(part, MAIN)  # (no self_broadcast shortcut: the product's one collective per call)
        out = self.ops.rhs_to_host(part)
        if self.info:
            out = np.full_like(out, np.nan)
        return out
