"""TEST INFRASTRUCTURE: a NumPy/SciPy stand-in for tinygp_amd.distributed.HipBlockOps so the
block-cyclic schedule (ownership, look-ahead order, ring slots, panel broadcasts, replicated
forward solve, slice broadcasts of the backward solve, the (M,) all-reduce) can run under `gloo`
on CPUs.  Same methods and the same buffer layout as csrc/dist.hip -- local block columns side
by side in one column-major matrix with GLOBAL rows, ring slot = [rows x nb panel | dinv] --
torch CPU tensors as the buffers the collectives see, the oracle's kernel evaluation for
assembly.  Never imported by the product."""
import contextlib
import re

import numpy as np
import scipy.linalg as sla
import torch

from oracle import ref_prog


class NumpyBlockOps:
    def setup(self, P, noise_diag, nb, world, rank):
        self.P, self.diag = P, noise_diag
        self.n, self.d = P.shape
        self.nb, self.G, self.rank = nb, world, rank
        self.dtype = P.dtype
        self.nblk = -(-self.n // nb)
        self.npad = self.nblk * nb
        self.nloc = len(range(rank, self.nblk, world))
        self.nd = (nb // 128) * 2048
        tdt = torch.float64 if P.dtype == np.float64 else torch.float32
        self.ring = [torch.zeros(self.npad * nb + self.nd, dtype=tdt) for _ in range(3)]
        self.x = torch.zeros(self.npad, dtype=tdt)
        self.xn = self.x.numpy()  # shares memory
        self.A = np.zeros((self.npad, max(self.nloc, 1) * nb), dtype=P.dtype, order="F")
        self.calls = []  # host-order log of the schedule (asserted by the tests)

    def stream(self, which):
        return contextlib.nullcontext()

    def slot(self, k, rows):
        return self.ring[k % 3][: rows * self.nb + self.nd]

    def slot_chunk(self, k, rows, c, nch):
        cw = self.nb // nch
        end = rows * self.nb + self.nd if c == nch - 1 else (c + 1) * cw * rows
        return self.ring[k % 3][c * cw * rows: end]

    def nbytes_of(self, buf):
        return buf.numel() * buf.element_size()

    def x_slice(self, k):
        return self.x[k * self.nb:(k + 1) * self.nb]

    def scalar(self, v):
        return torch.tensor([v], dtype=torch.float64)

    def _panel_view(self, k):
        rows = self.npad - k * self.nb
        return self.ring[k % 3].numpy()[: rows * self.nb].reshape((rows, self.nb), order="F")

    def _col(self, l):
        return self.A[:, l * self.nb:(l + 1) * self.nb]

    def assemble(self, prog):
        self.calls.append(("assemble",))
        nb, n = self.nb, self.n
        for l in range(self.nloc):
            j0 = (l * self.G + self.rank) * nb
            C = self._col(l)
            C[:] = 0.0
            n1, n2 = max(n - j0, 0), max(min(nb, n - j0), 0)
            if n1 and n2:
                C[j0:j0 + n1, :n2] = ref_prog.eval_matrix(prog, self.P[j0:j0 + n1], self.P[j0:j0 + n2])
                idx = np.arange(n2)
                C[j0 + idx, idx] += self.diag[j0:j0 + n2]
            for i in range(nb):  # identity padding
                if i >= n2:
                    C[j0 + i, i] = 1.0

    def load_matrix(self, K):
        self.calls.append(("load_matrix",))
        nb, n = self.nb, self.n
        K = np.asarray(K, dtype=self.dtype)
        for l in range(self.nloc):
            j0 = (l * self.G + self.rank) * nb
            Cc = self._col(l)
            Cc[:] = 0.0
            n1, n2 = max(n - j0, 0), max(min(nb, n - j0), 0)
            if n1 and n2:
                Cc[j0:j0 + n1, :n2] = K[j0:j0 + n1, j0:j0 + n2]
            for i in range(nb):  # identity padding
                if i >= n2:
                    Cc[j0 + i, i] = 1.0

    def begin(self, resid):
        self.calls.append(("begin",))
        self.info = 0
        self.logdet = np.zeros(self.nblk)
        self.solving = resid is not None
        self.xn[:] = 0.0
        if self.solving:
            self.xn[: self.n] = resid

    def _factor_and_pack(self, k):
        self.calls.append(("panel", k))
        nb, l = self.nb, k // self.G
        M = self._col(l)[k * nb:]
        if not np.all(np.isfinite(np.tril(M[:nb]))):  # poisoned by an earlier failing pivot
            M[:] = np.nan
        else:
            try:
                L = sla.cholesky(M[:nb], lower=True, check_finite=False)
                M[:nb] = L
                if M.shape[0] > nb:
                    M[nb:] = sla.solve_triangular(L, M[nb:].T, lower=True, check_finite=False).T
            except sla.LinAlgError as e:
                minor = int(re.search(r"(\d+)-th leading minor", str(e)).group(1))
                if self.info == 0:
                    self.info = k * nb + minor
                M[:] = np.nan
        self._panel_view(k)[:] = M

    def first_panel(self):
        self.calls.append(("first_panel",))

    def panel_chunk(self, k, c, nch):
        """The whole panel is factored and packed with its first chunk (a synchronous stand-in has nothing to
        overlap); the later chunks are already in place when the host sends them."""
        assert k % self.G == self.rank and 0 <= c < nch and (self.nb // 128) % nch == 0
        self.calls.append(("chunk", k, c))
        if c == 0:
            self._factor_and_pack(k)

    def slot_ready(self, k):
        assert k % self.G != self.rank
        self.calls.append(("slot_ready", k))

    def lookahead(self, k):
        self.calls.append(("lookahead", k))
        nb = self.nb
        P = self._panel_view(k)
        k1 = k + 1
        assert k1 < self.nblk and k1 % self.G == self.rank
        # the gate must find block column k+1 updated by every panel < k: pre_update(k-1) ran on this rank
        assert k == 0 or ("pre_update*", k - 1) in self.calls
        C = self._col(k1 // self.G)[k1 * nb:]
        C -= P[nb:] @ P[nb:2 * nb].T

    def arrived(self, k):
        self.calls.append(("arrived", k))

    def pre_update(self, k):
        k2 = k + 2
        if k2 >= self.nblk or k2 % self.G != self.rank:
            return
        self.calls.append(("pre_update*", k))
        nb = self.nb
        P = self._panel_view(k)
        off = 2 * nb
        self._col(k2 // self.G)[k2 * nb:] -= P[off:] @ P[off:off + nb].T

    def fwd_step(self, k):
        self.calls.append(("fwd_step", k))
        nb = self.nb
        P = self._panel_view(k)
        if self.solving:
            with np.errstate(all="ignore"):
                xk = sla.solve_triangular(P[:nb], self.xn[k * nb:(k + 1) * nb], lower=True, check_finite=False)
                self.xn[k * nb:(k + 1) * nb] = xk
                self.xn[(k + 1) * nb:] -= P[nb:] @ xk
        with np.errstate(all="ignore"):
            self.logdet[k] = np.sum(np.log(np.diag(P[:nb])))

    def rest(self, k):
        self.calls.append(("rest", k))
        nb = self.nb
        P = self._panel_view(k)
        for l in range(self.nloc):
            j = l * self.G + self.rank
            if j <= k + 2:  # k+1: the gate (lookahead); k+2: pre_update
                continue
            off = (j - k) * nb
            self._col(l)[j * nb:] -= P[off:] @ P[off:off + nb].T

    def end(self):
        self.calls.append(("end",))
        with np.errstate(all="ignore"):
            return self.info, float(np.sum(self.xn ** 2)) if self.solving else 0.0, float(np.sum(self.logdet))

    def bwd_step(self, k):
        if k % self.G != self.rank:
            return
        self.calls.append(("bwd", k))
        nb = self.nb
        C = self._col(k // self.G)
        xk = self.xn[k * nb:(k + 1) * nb] - C[(k + 1) * nb:].T @ self.xn[(k + 1) * nb:]
        self.xn[k * nb:(k + 1) * nb] = sla.solve_triangular(C[k * nb:(k + 1) * nb], xk, lower=True, trans=1,
                                                            check_finite=False)

    def cond_mean_partial(self, prog, Pt):
        out = np.zeros(Pt.shape[0], dtype=self.dtype)
        for l in range(self.nloc):
            j0 = (l * self.G + self.rank) * self.nb
            cnt = min(self.nb, self.n - j0)
            if cnt <= 0:
                break
            out += ref_prog.eval_matrix(prog, Pt, self.P[j0:j0 + cnt]) @ self.xn[j0:j0 + cnt]
        return torch.from_numpy(out)

    # -- solves on the resident factor: same buffer layout as csrc/dist.hip ((npad,) or (npad, nrhs) row-major) ----
    def rhs_zeros(self, nrhs):
        tdt = torch.float64 if self.dtype == np.float64 else torch.float32
        return torch.zeros((self.npad,) if nrhs == 1 else (self.npad, nrhs), dtype=tdt)

    def rhs_from_host(self, Y, nrhs):
        buf = self.rhs_zeros(nrhs)
        if nrhs == 1:
            buf.numpy()[: self.n] = Y.reshape(self.n)
        else:
            buf.numpy()[: self.n, : Y.shape[1]] = Y
        return buf

    def rhs_to_host(self, buf):
        return buf.numpy().copy()

    def rhs_block(self, buf, k):
        return buf[k * self.nb:(k + 1) * self.nb]

    def fwd_block(self, k, nrhs, y, acc, x):
        self.calls.append(("fwd_block", k))
        if k % self.G != self.rank:
            return
        nb = self.nb
        C = self._col(k // self.G)
        sl = slice(k * nb, (k + 1) * nb)
        with np.errstate(all="ignore"):
            xk = sla.solve_triangular(C[sl], y.numpy()[sl] + acc.numpy()[sl], lower=True, check_finite=False)
            x.numpy()[sl] = xk
            acc.numpy()[(k + 1) * nb:] -= C[(k + 1) * nb:] @ xk

    def bwd_block(self, k, x):
        if k % self.G != self.rank:
            return
        nb = self.nb
        C = self._col(k // self.G)
        xn = x.numpy()
        with np.errstate(all="ignore"):
            xk = xn[k * nb:(k + 1) * nb] - C[(k + 1) * nb:].T @ xn[(k + 1) * nb:]
            xn[k * nb:(k + 1) * nb] = sla.solve_triangular(C[k * nb:(k + 1) * nb], xk, lower=True, trans=1,
                                                           check_finite=False)

    def xloc_zeros(self, nrhs):
        tdt = torch.float64 if self.dtype == np.float64 else torch.float32
        return torch.zeros((max(self.nloc, 1) * self.nb, nrhs), dtype=tdt)

    def fwd_partial(self, k, nrhs, xloc, acc, first):
        self.calls.append(("fwd_partial", k))
        nb = self.nb
        cols = [l for l in range(self.nloc) if first <= l * self.G + self.rank < k]
        if not cols:
            return
        a = acc.numpy()
        xl = xloc.numpy()
        for l in cols:  # (one product on the device; block by block here)
            a[k * nb:(k + 1) * nb] -= self._col(l)[k * nb:(k + 1) * nb] @ xl[l * nb:(l + 1) * nb]

    def fwd_solve_left(self, k, nrhs, y, acc, x, xloc):
        self.calls.append(("fwd_block", k))  # (counted like the right-looking step: one reduce per block column)
        if k % self.G != self.rank:
            return
        nb, l = self.nb, k // self.G
        sl = slice(k * nb, (k + 1) * nb)
        Lkk = np.tril(self._col(l)[sl])
        xk = sla.solve_triangular(Lkk, y.numpy()[sl] + acc.numpy()[sl], lower=True, check_finite=False)
        x.numpy()[sl] = xk
        xloc.numpy()[l * nb:(l + 1) * nb] = xk

    def gather_owned(self, x, nrhs, world):
        nb = self.nb
        yloc = self.xloc_zeros(nrhs)
        for l in range(self.nloc):
            i = l * self.G + self.rank
            yloc.numpy()[l * nb:(l + 1) * nb] = x.numpy()[i * nb:(i + 1) * nb]
        return yloc

    def bwd_block_multi(self, k, nrhs, x, yloc):
        self.calls.append(("bwd_block_multi", k))
        if k % self.G != self.rank:
            return
        nb, l = self.nb, k // self.G
        Lkk = np.tril(self._col(l)[k * nb:(k + 1) * nb])
        x.numpy()[k * nb:(k + 1) * nb] = sla.solve_triangular(Lkk, yloc.numpy()[l * nb:(l + 1) * nb], lower=True, trans=1,
                                                              check_finite=False)

    def bwd_update_multi(self, k, nrhs, x, yloc, stop):
        self.calls.append(("bwd_update_multi", k, stop))
        nb = self.nb
        xk = x.numpy()[k * nb:(k + 1) * nb]
        yl = yloc.numpy()
        for l in range(self.nloc):
            i = l * self.G + self.rank
            if stop <= i < k:
                yl[l * nb:(l + 1) * nb] -= self._col(l)[k * nb:(k + 1) * nb].T @ xk

    def rhs_identity(self, c0, nrhs):
        buf = self.rhs_zeros(nrhs)
        v = buf.numpy()
        for r in range(nrhs):
            if c0 + r < self.npad:
                v[c0 + r, r] = 1.0
        return buf

    # gradient accumulators: the device kernel's sums restated with central differences of the program's matrix
    def grad_begin(self, prog):
        self._gprog = [tuple(p) for p in prog]
        self._gacc = np.zeros(2 * len(prog) + self.d)
        self._kdiag = np.zeros(self.n, dtype=self.dtype)

    def _dK(self, which, q, rows, cols):
        """d K[rows, cols] / d (parameter q of op `which`) -- or, which < 0, d / d log-scale of input dimension -1 - which --
        by central differences of the oracle's evaluation of the program."""
        P = self.P.astype(np.float64)
        if which >= 0:
            op, metric, p0, p1 = self._gprog[which]
            base = p0 if q == 0 else p1
            h = 1e-6 * max(1.0, abs(base))

            def at(v):
                prog = list(self._gprog)
                prog[which] = (op, metric, v, p1) if q == 0 else (op, metric, p0, v)
                return ref_prog.eval_matrix(prog, P[rows], P[cols])
            return (at(base + h) - at(base - h)) / (2 * h)
        dim, h = -1 - which, 1e-6

        def at_s(ls):
            S = np.ones(self.d)
            S[dim] = np.exp(ls)
            return ref_prog.eval_matrix(self._gprog, P[rows] * S, P[cols] * S)
        return (at_s(h) - at_s(-h)) / (2 * h)

    def grad_chunk(self, c0, nrhs, kcols, with_logscale):
        self.calls.append(("grad_chunk", c0))
        Kc = kcols.numpy().astype(np.float64)
        alpha = self.xn.astype(np.float64)
        cols = np.arange(c0, min(c0 + nrhs, self.n))
        rows = np.array([i for i in range(self.n) if (i // self.nb) % self.G == self.rank and i >= c0], dtype=int)
        self._kdiag[cols] = Kc[cols, cols - c0]
        if not len(rows) or not len(cols):
            return
        Gm = np.outer(alpha[rows], alpha[cols]) - Kc[rows][:, cols - c0]
        W = np.where(rows[:, None] > cols[None, :], 1.0, np.where(rows[:, None] == cols[None, :], 0.5, 0.0))
        for i, (op, _m, _p0, _p1) in enumerate(self._gprog):
            if op >= 16:
                continue
            for q in range(2 if op in (6, 7) else 1):
                self._gacc[2 * i + q] += np.sum(W * Gm * self._dK(i, q, rows, cols))
        if with_logscale:
            for dim in range(self.d):
                self._gacc[2 * len(self._gprog) + dim] += np.sum(W * Gm * self._dK(-1 - dim, 0, rows, cols))

    def grad_end(self, d):
        part = np.concatenate([self._gacc[: 2 * len(self._gprog)], self._gacc[2 * len(self._gprog): 2 * len(self._gprog) + d]])
        return torch.from_numpy(part.copy()), self._kdiag.copy()

    def trmv_partial(self, y):
        out = self.rhs_zeros(1)
        for l in range(self.nloc):
            k = l * self.G + self.rank
            Lk = self._col(l)[k * self.nb:].copy()
            Lk[: self.nb] = np.tril(Lk[: self.nb])
            out.numpy()[k * self.nb:] += Lk @ y.numpy()[k * self.nb:(k + 1) * self.nb]
        return out

    def cross_cov(self, prog, Pt, m_pad):
        out = self.rhs_zeros(m_pad)
        out.numpy()[: self.n, : Pt.shape[0]] = ref_prog.eval_matrix(prog, self.P, Pt)
        return out

    def _owned_rows(self):
        rows = np.zeros(self.npad, dtype=bool)
        for l in range(self.nloc):
            k = l * self.G + self.rank
            rows[k * self.nb:(k + 1) * self.nb] = True
        return rows

    def colsumsq_owned(self, nrhs, x):
        return torch.from_numpy(np.sum(np.square(x.numpy()[self._owned_rows()]), axis=0))

    def gram_owned(self, nrhs, x):
        a = x.numpy()[self._owned_rows()]
        return torch.from_numpy(np.ascontiguousarray(a.T @ a))

    def gram_pair_owned(self, ni, xi, nj, xj):
        a, b = xi.numpy()[self._owned_rows()], xj.numpy()[self._owned_rows()]
        return torch.from_numpy(np.ascontiguousarray((a.T @ b).T))  # (nj, ni): the device's column-major (ni x nj) block

    def set_x(self, buf):
        self.xn[:] = buf.numpy()

    def abort(self):
        self.calls.append(("abort",))

    def column(self, l, rows):
        j0 = (l * self.G + self.rank) * self.nb
        return self._col(l)[j0:].copy()
