"""GPU diagnostics, one stage per process (a hang in one stage must not eat the budget)."""
import faulthandler
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
faulthandler.dump_traceback_later(50, exit=True)

import numpy as np  # noqa: E402

T0 = time.time()


def log(*a):
    print(f"[{time.time() - T0:7.2f}s]", *a, flush=True)


stage = sys.argv[1]
log("stage", stage)
import _lowlevel as ll  # noqa: E402
from tinygp_amd import GaussianProcess, _ffi, kernels, synthetic  # noqa: E402

if stage == "ubench":
    import ctypes as C
    ctx = _ffi.default_ctx()
    names = {0: "mfma_f64_16x16x4", 1: "mfma_f32_16x16x4", 2: "valu_fma_f64", 3: "mfma_f64+4xfma_f64", 4: "mfma_f64_4x4x4_4b"}
    for kind in (0, 1, 2, 3, 4):
        for bpc in (1, 2, 4, 8):
            tf, cyc = C.c_double(), C.c_double()
            _ffi.check(_ffi.lib().tgp_ubench(ctx.handle, kind, bpc, C.byref(tf), C.byref(cyc)), "ubench")
            log(f"{names[kind]:22s} waves/SIMD={bpc}  {tf.value:8.2f} TFLOP/s   {cyc.value:8.1f} cycles/wave-op")
elif stage == "kmat":
    from oracle import tinygp_np as o
    import _cases
    x1, x2 = _cases.data_kernels()
    for name, k in _cases.kernel_zoo(kernels).items():
        got = k(x1, x2)
        want = _cases.kernel_zoo(o)[name](x1, x2)
        log(name, "maxabs", np.abs(got - want).max())
elif stage == "gemm":
    rng = np.random.default_rng(0)
    for m, n, k, lower in [(128, 128, 16, False), (256, 384, 64, False), (512, 512, 128, True)]:
        A, B, C0 = rng.normal(size=(m, k)), rng.normal(size=(n, k)), rng.normal(size=(m, n))
        got = ll.gemm_nt(A, B, C0, -1.0, 1.0, lower)
        want = C0 - A @ B.T
        if lower:
            mask = (np.arange(m)[:, None] // 128) >= (np.arange(n)[None, :] // 128)
            log(m, n, k, "lower err", np.abs(got - want)[mask].max(), "untouched", np.abs(got - C0)[~mask].max())
        else:
            log(m, n, k, "err", np.abs(got - want).max())
elif stage == "potrf":
    import scipy.linalg as sla
    from oracle import tinygp_np as o
    for n in (128, 256, 1024):
        X, _ = synthetic.make_inputs(n, 1)
        K = (2.25 * o.ExpSquared(2.5))(X, X) + 0.05 * np.eye(n)
        for la in (0, 1):
            L, info = ll.potrf(K, lookahead=la)
            Lr = sla.cholesky(K, lower=True)
            log("n", n, "la", la, "info", info, "err", np.abs(L - Lr).max())
elif stage in ("trsv0", "trsv1"):
    import scipy.linalg as sla
    from oracle import tinygp_np as o
    tr = stage == "trsv1"
    for n in (128, 384):
        X, _ = synthetic.make_inputs(n, 1)
        K = (2.25 * o.ExpSquared(2.5))(X, X) + 0.05 * np.eye(n)
        L = sla.cholesky(K, lower=True)
        y = np.random.default_rng(1).normal(size=n)
        got = ll.trsv(L, y, tr)
        log("n", n, "transpose", tr, "err", np.abs(got - sla.solve_triangular(L, y, lower=True, trans=int(tr))).max())
elif stage == "trsm":
    import scipy.linalg as sla
    from oracle import tinygp_np as o
    for m, n in [(128, 128), (256, 640)]:
        X, _ = synthetic.make_inputs(n, 1)
        K = (2.25 * o.ExpSquared(2.5))(X, X) + 0.05 * np.eye(n)
        L = sla.cholesky(K, lower=True)
        B = np.random.default_rng(1).normal(size=(m, n))
        got = ll.trsm_right_lt(L, B)
        log(m, n, "err", np.abs(got - sla.solve_triangular(L, B.T, lower=True).T).max())
elif stage == "smoke":
    X, y = synthetic.make_inputs(1024, 1)
    log("construct")
    gp = GaussianProcess(synthetic.config_kernel(kernels, "expsq"), X, diag=0.01)
    log("info", gp.solver.info)
    log("logp", float(gp.log_probability(y)))
    log("norm", float(gp.solver.normalization()))
    a = gp.solver.solve_triangular(y)
    log("fwd solve", a[:3])
    a = gp.solver.solve_triangular(y, transpose=True)
    log("bwd solve", a[:3])
    al, lp = gp.solver.alpha(y)
    log("alpha", al[:3], lp)
    xt = np.linspace(0, 10.24, 32)
    m = gp.solver.conditional_mean(gp.kernel, xt, al)
    log("cond mean", m[:3])
    v = gp.solver.condition_variance(gp.kernel, xt)
    log("cond var", v[:3])
    c = gp.solver.condition(gp.kernel, xt, __import__("tinygp_amd").noise.Diagonal(np.full(32, 1e-8)))
    log("cond cov", c[0, :3])
    log("variance", gp.variance[:3]); log("covariance", gp.covariance[0, :3]); log("tril", gp.solver.scale_tril[1, :3])
    from oracle import tinygp_np as o
    log("oracle start")
    ref = o.GaussianProcess(synthetic.config_kernel(o, "expsq"), X, diag=0.01)
    log("oracle logp", float(ref.log_probability(y)))
    r = ref.condition(y, xt)
    log("oracle cond", r.gp.loc[:3], r.gp.variance[:3])
elif stage == "oracle":
    from oracle import tinygp_np as o
    import _cases
    log("start")
    for name, (gp, y, t) in _cases.gp_cases(o, o.GaussianProcess).items():
        log(name, float(gp.log_probability(y)))
elif stage == "small":
    import _cases
    for name, (gp, y, t) in _cases.gp_cases(kernels, GaussianProcess).items():
        log(name, "info", gp.solver.info, "logp", float(gp.log_probability(y)))
        c = gp.condition(y, t)
        log("   loc", c.gp.loc[:2], "var", c.gp.variance[:2])
log("done")
