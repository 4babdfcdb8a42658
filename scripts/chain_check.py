"""Persistent panel chain (ctx option chain_kernel = 1) against LAPACK and against the per-block launches:
factor error per size / dtype / schedule, then evaluation times of both chains.  Every GPU call sits under the
caller's `timeout`; a device-side hand-off timeout surfaces as a TgpError, never as a hang."""
import sys
import time
from pathlib import Path

import numpy as np
import scipy.linalg as sla

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import _lowlevel as ll  # noqa: E402
from tinygp_amd import _ffi, kernels, noise, synthetic  # noqa: E402
from tinygp_amd.solvers import DirectSolver  # noqa: E402


QUICK = "--quick" in sys.argv


def spd(n, dt, seed):
    rng = np.random.default_rng(seed)
    B = rng.normal(size=(n, 96))
    A = B @ B.T / 96 + np.eye(n) * (1.0 + rng.uniform(size=n))
    return A.astype(dt)


bad = 0
for dt, tol in ((np.float64, 1e-11), (np.float32, 2e-3)):
    for n in (128, 384, 1152, 2560, 5248, 8192) if QUICK else (128, 256, 384, 1024, 1152, 2176, 2560, 4224, 5248, 8192):
        A = spd(n, dt, n)
        Lref = sla.cholesky(A.astype(np.float64), lower=True)
        for opts in ((dict(lookahead=1), dict(lookahead=1, chain_full_rows=0), dict(lookahead=0, chain_full_rows=0)) if QUICK else
                     (dict(lookahead=1), dict(lookahead=0), dict(lookahead=1, nb_outer=512, first_split=3),
                      dict(lookahead=1, first_split=0), dict(lookahead=1, chain_full_rows=0),
                      dict(lookahead=0, chain_full_rows=0), dict(lookahead=1, chain_full_rows=2048, nb_outer=512))):
            try:
                L, info = ll.potrf(A.copy(), chain_kernel=1, **opts)
                err = np.abs(L - Lref).max() / np.abs(Lref).max()
                L0, _ = ll.potrf(A.copy(), chain_kernel=0, **opts)
                err0 = np.abs(L0 - Lref).max() / np.abs(Lref).max()
                L2, _ = ll.potrf(A.copy(), chain_kernel=1, **opts)
                same = np.array_equal(L, L2)
            except Exception as e:  # noqa: BLE001
                print(f"{dt.__name__} n={n} {opts}: EXCEPTION {e}", flush=True)
                bad += 1
                continue
            ok = info == 0 and err < tol and same
            bad += 0 if ok else 1
            print(f"{dt.__name__} n={n} {opts}: info={info} err={err:.2e} (per-block chain {err0:.2e}) "
                  f"repeat-identical={same} {'ok' if ok else 'BAD'}", flush=True)
    if bad:
        break
# a non-positive pivot inside a chained block: same info as LAPACK's
A = spd(1152, np.float64, 7)
A[700, 700] = -1.0
L, info = ll.potrf(A.copy(), chain_kernel=1)
print("bad pivot info", info, "(expected 701)")
bad += 0 if info == 701 else 1

ctx = _ffi.default_ctx()
for n in (1024, 2048, 4096, 8192, 16384):
    X, y = synthetic.make_inputs(n, 1, "float64")
    k = 1.5**2 * kernels.ExpSquared(2.5)
    solver = DirectSolver(k, X, noise.Diagonal(np.full(n, 0.01)))
    solver.set_residual(y)
    res = {}
    for ck in (0, 1, 0, 1):
        ctx.set_option("chain_kernel", ck)
        v = solver.factor_log_probability(None, k)
        reps = 20 if n <= 8192 else 10
        t0 = time.perf_counter()
        for _ in range(reps):
            v = solver.factor_log_probability(None, k)
        res.setdefault(ck, []).append(((time.perf_counter() - t0) / reps * 1e3, v))
    ctx.set_option("chain_kernel", 0)
    t0s = min(t for t, _ in res[0])
    t1s = min(t for t, _ in res[1])
    v0, v1 = res[0][0][1], res[1][0][1]
    print(f"N={n}: per-block chain {t0s:.3f} ms, persistent chain {t1s:.3f} ms ({t0s / t1s:.2f}x); "
          f"loglik {v0!r} vs {v1!r} rel {abs(v0 - v1) / abs(v0):.1e}", flush=True)
    solver.close()
print("CHAIN", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
