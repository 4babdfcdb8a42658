"""TEST INFRASTRUCTURE: NumPy evaluator of a `tgp_kop` postfix kernel program (the same
encoding libtgp_hip consumes), built on the oracle's distance / kernel formulas.  Lets
tests check the host-side kernel-tree lowering without a GPU."""
import numpy as np

from oracle import tinygp_np as o

_LEAF = {1: o.Exp, 2: o.ExpSquared, 3: o.Matern32, 4: o.Matern52, 5: o.Cosine}


def eval_matrix(prog, X1, X2):
    X1 = X1[:, None] if X1.ndim == 1 else X1
    X2 = X2[:, None] if X2.ndim == 1 else X2
    stack = []
    for op, metric, p0, p1 in prog:
        if op == 16:
            b, a = stack.pop(), stack.pop()
            stack.append(a + b)
        elif op == 17:
            b, a = stack.pop(), stack.pop()
            stack.append(a * b)
        elif op == 0:
            stack.append(np.full((X1.shape[0], X2.shape[0]), p0, dtype=X1.dtype))
        else:
            dist = o.L2Distance() if metric == 1 else o.L1Distance()
            if op in _LEAF:
                k = _LEAF[op](p0, distance=dist)
            elif op == 6:
                k = o.ExpSineSquared(p0, distance=dist, gamma=p1)
            elif op == 7:
                k = o.RationalQuadratic(p0, distance=dist, alpha=p1)
            else:
                raise ValueError(op)
            stack.append(k(X1, X2))
    assert len(stack) == 1
    return stack[0]
