"""TEST INFRASTRUCTURE: a NumPy/SciPy stand-in for tinygp_amd.distributed.HipBlockOps so the
block-cyclic schedule (ownership, look-ahead, broadcasts, slice all-reduces) can run under
`gloo` on CPUs.  Same method signatures, torch CPU tensors as buffers, the oracle's kernel
evaluation for assembly.  Never imported by the product."""
import numpy as np
import scipy.linalg as sla
import torch

from oracle import ref_prog


class NumpyBlockOps:
    def context(self):
        import contextlib

        return contextlib.nullcontext()

    def empty(self, nelem, dtype):
        return torch.empty(int(nelem), dtype=torch.float64 if np.dtype(dtype) == np.float64 else torch.float32)

    def zeros(self, nelem, dtype):
        return torch.zeros(int(nelem), dtype=torch.float64 if np.dtype(dtype) == np.float64 else torch.float32)

    def from_numpy(self, a):
        return torch.from_numpy(np.ascontiguousarray(a).copy())

    @staticmethod
    def _mat(t, rows, cols, off=0):
        """column-major (rows x cols) view with leading dimension = the panel's row count."""
        a = t.numpy()
        return a[off:].reshape(-1)  # flat; callers index explicitly

    def assemble(self, prog, X, diag, n, d, j0, nb, out, rows):
        Xn = X.numpy().reshape(n, d)
        dg = diag.numpy()
        M = np.zeros((rows, nb), dtype=Xn.dtype)
        n1, n2 = max(n - j0, 0), max(min(nb, n - j0), 0)
        if n1 and n2:
            M[:n1, :n2] = ref_prog.eval_matrix(prog, Xn[j0:j0 + n1], Xn[j0:j0 + n2])
            idx = np.arange(n2)
            M[idx, idx] += dg[j0:j0 + n2]
        for i in range(min(rows, nb)):  # identity padding
            if i >= n1 or i >= n2:
                M[i, i] = 1.0
        out.numpy()[: rows * nb] = M.reshape(-1, order="F")

    def factor_panel(self, P, rows, nb):
        M = P.numpy()[: rows * nb].reshape((rows, nb), order="F")
        if not np.all(np.isfinite(np.tril(M[:nb]))):  # poisoned by an earlier failing pivot
            M[:] = np.nan
            P.numpy()[: rows * nb] = M.reshape(-1, order="F")
            return 0
        try:
            L = sla.cholesky(M[:nb], lower=True, check_finite=False)
        except sla.LinAlgError as e:
            import re

            k = int(re.search(r"(\d+)-th leading minor", str(e)).group(1))
            M[:] = np.nan
            P.numpy()[: rows * nb] = M.reshape(-1, order="F")
            return k
        M[:nb] = L
        if rows > nb:
            M[nb:] = sla.solve_triangular(L, M[nb:].T, lower=True, check_finite=False).T
        P.numpy()[: rows * nb] = M.reshape(-1, order="F")
        return 0

    def update(self, P, prow, off, Cj, crow, nb):
        Pm = P.numpy()[: prow * nb].reshape((prow, nb), order="F")
        Cm = Cj.numpy()[: crow * nb].reshape((crow, nb), order="F")
        A = Pm[off:off + crow]
        Cm -= A @ Pm[off:off + nb].T
        Cj.numpy()[: crow * nb] = Cm.reshape(-1, order="F")

    def solve_diag(self, P, rows, nb, t):
        L = P.numpy()[: rows * nb].reshape((rows, nb), order="F")[:nb]
        t.numpy()[:] = sla.solve_triangular(L, t.numpy(), lower=True, check_finite=False)

    def gemv_sub(self, P, rows, nb, x, w_below):
        if rows > nb:
            M = P.numpy()[: rows * nb].reshape((rows, nb), order="F")
            w_below.numpy()[: rows - nb] -= M[nb:] @ x.numpy()

    def sum_log_diag(self, P, rows, nb, nvalid):
        L = P.numpy()[: rows * nb].reshape((rows, nb), order="F")
        return float(np.sum(np.log(np.diag(L[:nb])[:nvalid])))

    def sum_squares(self, x, nvalid):
        return float(np.sum(x.numpy()[:nvalid] ** 2))
