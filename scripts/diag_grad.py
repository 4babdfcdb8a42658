import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from tinygp_amd import GaussianProcess, kernels
from oracle import tinygp_np as o, grad_np
rng = np.random.default_rng(11)
n = 300
X = rng.uniform(0, 3, (n, 3))
y = np.sin(X[:, 0]) + 0.1 * rng.normal(size=n)
diag = rng.uniform(0.05, 0.15, n)
for nm, build in (("m32", lambda k, t: t[0] * k.Matern32(t[1])), ("expsq", lambda k, t: t[0] * k.ExpSquared(t[1]))):
    th = [1.8, 1.5]
    gp = GaussianProcess(build(kernels, th), X, diag=diag)
    ll, g = gp.log_probability_and_grad(y)
    print(nm, "info", gp.solver.info, "ll", ll, "kernel grad", g["kernel"], "noise nan", np.isnan(g["noise_diag"]).sum(), "alpha nan", np.isnan(g["mean"]).sum())
    K = build(o, th)(X, X) + np.diag(diag)
    w = np.linalg.eigvalsh(K)
    print("   eig min/max", w.min(), w.max())
    print("   oracle", grad_np.log_probability_and_grad(lambda t: build(o, t), th, X, diag, y)[1])
