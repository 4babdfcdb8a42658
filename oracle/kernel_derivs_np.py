"""TEST INFRASTRUCTURE: closed-form parameter derivatives of the stationary kernels in NumPy.

The reference has no derivative code of its own (its users differentiate with JAX autodiff, docs/tutorials/
quickstart.ipynb cell 4); the device evaluates d k / d theta in closed form (tinygp_amd/csrc/kmat.hip, leaf_deriv).  This
file states those closed forms a second time, independently, from the kernels' definitions
(/root/reference/src/tinygp/kernels/stationary.py:76-235, restated in oracle/tinygp_np.py), and tests/test_oracle.py holds
them against central differences of the oracle's own kernel matrices -- so that the gradient oracle (oracle/grad_np.py:
the trace identity with dK/dtheta by central differences) and the device's formulas are pinned by something other than
each other.  With r the kernel's distance, l its scale:

  Exp                k = exp(-r / l)                          dk/dl = k r / l^2
  ExpSquared         k = exp(-r^2 / (2 l^2))                  dk/dl = k r^2 / l^3
  Matern32           a = sqrt(3) r / l, k = (1 + a) e^-a      dk/dl = a^2 e^-a / l
  Matern52           a = sqrt(5) r / l, k = (1+a+a^2/3) e^-a  dk/dl = (a^2 / 3)(1 + a) e^-a / l
  Cosine             u = 2 pi r / l, k = cos u                dk/dl = u sin(u) / l
  ExpSineSquared     u = pi r / l, k = exp(-g sin^2 u)        dk/dl = k g 2 sin(u) cos(u) u / l,  dk/dg = -sin^2(u) k
  RationalQuadratic  q = r^2 / (2 a l^2), k = (1 + q)^-a      dk/dl = k / (1 + q) r^2 / l^3,  dk/da = k (q / (1 + q) - ln(1 + q))
"""
import numpy as np

from oracle import tinygp_np as o


def _r(kernel, X1, X2, squared=False):
    d = np.asarray(X1)[:, None, :] - np.asarray(X2)[None, :, :]
    return kernel.distance.squared_distance(d) if squared else kernel.distance.distance(d)


def dleaf(kernel, X1, X2, param="scale"):
    """d k(X1, X2) / d param for a stationary leaf of oracle/tinygp_np.py; X (n, d)."""
    l = float(kernel.scale)
    if isinstance(kernel, o.Exp):
        assert param == "scale"
        r = _r(kernel, X1, X2)
        return np.exp(-r / l) * r / l**2
    if isinstance(kernel, o.ExpSquared):
        assert param == "scale"
        r2 = _r(kernel, X1, X2, squared=True)
        return np.exp(-0.5 * r2 / l**2) * r2 / l**3
    if isinstance(kernel, o.Matern32):
        assert param == "scale"
        a = np.sqrt(3) * _r(kernel, X1, X2) / l
        return a * a * np.exp(-a) / l
    if isinstance(kernel, o.Matern52):
        assert param == "scale"
        a = np.sqrt(5) * _r(kernel, X1, X2) / l
        return (a * a / 3) * (1 + a) * np.exp(-a) / l
    if isinstance(kernel, o.Cosine):
        assert param == "scale"
        u = 2 * np.pi * _r(kernel, X1, X2) / l
        return u * np.sin(u) / l
    if isinstance(kernel, o.ExpSineSquared):
        u = np.pi * _r(kernel, X1, X2) / l
        g = float(kernel.gamma)
        k = np.exp(-g * np.sin(u) ** 2)
        if param == "scale":
            return k * g * 2 * np.sin(u) * np.cos(u) * u / l
        assert param == "gamma"
        return -np.sin(u) ** 2 * k
    if isinstance(kernel, o.RationalQuadratic):
        a = float(kernel.alpha)
        r2 = _r(kernel, X1, X2, squared=True)
        q = 0.5 * r2 / (a * l**2)
        k = (1 + q) ** -a
        if param == "scale":
            return k / (1 + q) * r2 / l**3
        assert param == "alpha"
        return k * (q / (1 + q) - np.log1p(q))
    raise TypeError(f"no closed form for {type(kernel).__name__}")
