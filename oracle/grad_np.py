"""TEST INFRASTRUCTURE: gradient of the GP log-probability in NumPy.

The reference has no explicit gradient code -- its users differentiate ``log_probability``
with JAX autodiff (docs/tutorials/quickstart.ipynb cell 4) -- so this oracle states the
textbook identity
    d ll / d theta = 1/2 alpha^T (dK/dtheta) alpha - 1/2 tr(K^-1 dK/dtheta),  alpha = K^-1 r,
with dK/dtheta taken by central differences of the ORACLE's own kernel matrix, and
cross-checks it against central differences of the oracle's log-likelihood itself.

HOW THIS IS PINNED: not to the reference's autodiff (JAX cannot run in this image) but THROUGH FINITE DIFFERENCES OF A
PINNED SCALAR -- the log-likelihood of ``oracle/tinygp_np.py`` is pinned to the reference's own execution
(``tests/golden/ref_gp.npz``, ``tests/test_reference_pin.py``), this gradient must agree with central differences of that
scalar to 1e-4 (``tests/test_oracle.py``), and the closed forms dk/dtheta the device evaluates are restated a second time in
NumPy (``oracle/kernel_derivs_np.py``) and held against central differences of the oracle's kernel matrices at 2e-7.
"""
import numpy as np

from oracle import tinygp_np as o


def log_probability_and_grad(build, theta, X, diag, y, *, rel_step=1e-6):
    """build(theta) -> oracle kernel.  Returns (ll, grad_theta, grad_noise_diag, grad_mean)."""
    theta = np.asarray(theta, dtype=np.float64)
    diag = np.broadcast_to(np.asarray(diag, dtype=np.float64), (len(y),))
    K = build(theta)(X, X) + np.diag(diag)
    Kinv = np.linalg.inv(K)
    alpha = Kinv @ y
    ll = float(o.GaussianProcess(build(theta), X, diag=diag).log_probability(y))
    G = np.outer(alpha, alpha) - Kinv
    g = np.empty_like(theta)
    for p in range(len(theta)):
        h = rel_step * max(1.0, abs(theta[p]))
        tp, tm = theta.copy(), theta.copy()
        tp[p] += h
        tm[p] -= h
        dK = (build(tp)(X, X) - build(tm)(X, X)) / (2 * h)
        g[p] = 0.5 * np.sum(G * dK)
    return ll, g, 0.5 * np.diag(G), alpha


def finite_difference_grad(build, theta, X, diag, y, *, rel_step=1e-5):
    theta = np.asarray(theta, dtype=np.float64)
    g = np.empty_like(theta)
    for p in range(len(theta)):
        h = rel_step * max(1.0, abs(theta[p]))
        tp, tm = theta.copy(), theta.copy()
        tp[p] += h
        tm[p] -= h
        lp = float(o.GaussianProcess(build(tp), X, diag=diag).log_probability(y))
        lm = float(o.GaussianProcess(build(tm), X, diag=diag).log_probability(y))
        g[p] = (lp - lm) / (2 * h)
    return g


class Scaled:
    """Oracle kernel on per-dimension scaled inputs, ``k(s * x1, s * x2)``: what the reference's
    ``transforms.Linear(scale, kernel)`` evaluates for a 0- or 1-dimensional scale (transforms.py:39-72) -- the oracle
    restatement has no transforms module, and this is all the gradient tests need of one."""

    def __init__(self, scale, kernel):
        self.scale, self.kernel = np.asarray(scale, dtype=np.float64), kernel

    def __call__(self, X1, X2=None):
        X1 = np.asarray(X1) * self.scale
        return self.kernel(X1) if X2 is None else self.kernel(X1, np.asarray(X2) * self.scale)

    def __rmul__(self, c):
        return Scaled(self.scale, c * self.kernel)
