"""GPU parity, device-pointer layer: each HIP kernel through the C ABI against the oracle
(NumPy/SciPy) on the same seeded inputs.  fp64 tolerances are written next to each check."""
import numpy as np
import pytest
import scipy.linalg as sla

import _cases
import _lowlevel as ll
from conftest import ulp_diff
from oracle import tinygp_np as o
from tinygp_amd import kernels

pytestmark = pytest.mark.gpu


def test_mfma_issue_rate_reported():
    f64 = ll.ubench(np.float64)
    f32 = ll.ubench(np.float32)
    print(f"\n[ubench] v_mfma_f64_16x16x4_f64: {f64:.1f} TFLOP/s   v_mfma_f32_16x16x4_f32: {f32:.1f} TFLOP/s")
    assert 20 < f64 < 200 and 40 < f32 < 400


@pytest.mark.parametrize("name", sorted(_cases.kernel_zoo(kernels)))
def test_kernel_matrix_matches_oracle(name, golden_dir):
    """K1: per-entry agreement <= 4 ulp with NumPy (same operation order, ocml vs libm
    exp/sin/cos/pow) on the 5-D and 1-D fixtures of the reference's kernel tests."""
    g = np.load(golden_dir / "kernels.npz")
    ref = np.load(golden_dir / "ref_kernels.npz")  # the reference's own outputs (oracle/refshim)
    x1, x2 = _cases.data_kernels()
    xs, _, ts = _cases.data_solver()
    kp, ko = _cases.kernel_zoo(kernels)[name], _cases.kernel_zoo(o)[name]
    for X1, X2, key in ((x1, x2, "5d"), (xs, ts, "1d")):
        got = kp(X1, X2)
        want = ko(X1, X2)
        assert got.shape == want.shape and got.dtype == want.dtype
        # cos/sin of large arguments lose relative accuracy near zeros of the function:
        # bound those by absolute error instead
        if name in ("cosine", "solver_cos", "expsine2", "sum_ops", "prod_ops"):
            np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-14)
        else:
            assert ulp_diff(got, want) <= 4, (name, key, ulp_diff(got, want))
        np.testing.assert_allclose(got, g[f"{name}__{key}"], rtol=1e-13, atol=1e-14)
        np.testing.assert_allclose(got, ref[f"{name}__{key}"], rtol=1e-13, atol=1e-14)
        if name not in ("cosine", "solver_cos", "expsine2", "sum_ops", "prod_ops"):
            assert ulp_diff(got, ref[f"{name}__{key}"]) <= 4, (name, key)
    np.testing.assert_allclose(kp(x1), ko(x1), rtol=1e-15)
    np.testing.assert_allclose(kp(x1), ref[f"{name}__diag"], rtol=1e-15)
    # fp32 path: reference tolerance 5e-4 (test_utils.py:15)
    got32 = kp(x1.astype(np.float32), x2.astype(np.float32))
    assert got32.dtype == np.float32
    np.testing.assert_allclose(got32, ko(x1, x2), rtol=5e-4, atol=5e-4)


def test_division_free_quotients_are_bit_identical_to_the_division():
    """kmat_fast_kernel takes r / l and r^2 / l^2 as one multiply + four FMAs on the reciprocal (Markstein; kmat.hip,
    UDiv) instead of the division sequence: every entry must be the division's, bit for bit -- over scales with short,
    long and all-ones significands (the excluded case: the kernel divides), tiny and huge scales, and coordinates whose
    distances leave the range the fast route is proven for (the kernel redoes those lanes with the division)."""
    from tinygp_amd import _ffi

    ctx = _ffi.default_ctx()
    rng = np.random.default_rng(5)
    all_ones = float(np.frombuffer(np.uint64(0x3FFFFFFFFFFFFFFF).tobytes(), dtype=np.float64)[0])  # 2 - 2^-52
    scales = [0.9, 3.0, 1.0 / 3.0, 2.5, all_ones, float(np.sqrt(all_ones)), 1e-40, 1e-25, 1e25, 1e40,
              float(rng.uniform(0.1, 10))]
    for d in (1, 3):
        a, b = rng.normal(size=(256, d)) * 3.0, rng.normal(size=(256, d)) * 3.0
        sets = [("plain", a, b)]
        big = a.copy()
        big[7] = 1e160                      # distances ~1e160, squared ~1e320 = inf: beyond 2^300
        big[100, 0] = np.inf
        sets.append(("huge", big, b))
        tiny = a * 1e-170                   # squared distances underflow to subnormals / zero
        sets.append(("tiny", tiny, b * 1e-170))
        for s in scales:
            zoo = [kernels.ExpSquared(s), 1.7 * kernels.ExpSquared(s, distance=kernels.L1Distance()),
                   kernels.Matern32(s) * 0.6, 2.5 * kernels.Matern52(s, distance=kernels.L2Distance()),
                   kernels.Matern52(s), kernels.Exp(s)]
            for k in zoo:
                for tag, x1, x2 in sets:
                    with np.errstate(all="ignore"):
                        ctx.set_option("kmat_plain_div", 0)
                        fast = k(x1, x2)
                        ctx.set_option("kmat_plain_div", 1)
                        try:
                            plain = k(x1, x2)
                        finally:
                            ctx.set_option("kmat_plain_div", 0)
                    assert np.array_equal(fast, plain, equal_nan=True), (type(k), s, d, tag)


def test_kernel_matrix_ragged_shapes():
    """Edge shapes: single point, non-multiples of the 128 tile, D > 4 (dynamic-D path)."""
    rng = np.random.default_rng(7)
    k, ko = kernels.Matern52(0.7) * 1.3, o.Matern52(0.7) * 1.3
    for n1, n2, d in [(1, 1, 1), (1, 300, 2), (129, 127, 3), (257, 1, 4), (130, 140, 7), (5, 9, 16)]:
        a, b = rng.normal(size=(n1, d)), rng.normal(size=(n2, d))
        if d < 8:
            assert ulp_diff(k(a, b), ko(a, b)) <= 4, (n1, n2, d)
        else:
            # NumPy sums >= 8 terms pairwise (different rounding of sum|d|), and exp(-a) at
            # a ~ 60 amplifies one ulp of `a` sixty-fold: bound the relative error instead
            np.testing.assert_allclose(k(a, b), ko(a, b), rtol=1e-12, atol=0)
    a = rng.normal(size=(0, 2))
    assert k(a, rng.normal(size=(3, 2))).shape == (0, 3)
    # D > 16 is beyond the device evaluator: the matrix comes from the host route (reference formulas in NumPy)
    a17, b17 = rng.normal(size=(4, 17)), rng.normal(size=(6, 17))
    np.testing.assert_allclose(k(a17, b17), ko(a17, b17), rtol=1e-14)
    with pytest.raises(ValueError):
        k(rng.normal(size=(4, 2)), rng.normal(size=(4, 3)))


@pytest.mark.parametrize("d", [1, 2, 3])
def test_straight_line_evaluator_is_bit_identical_to_the_general_one(d):
    """"leaf" / "amp * leaf" programs with an exp-family leaf run full 128 x 128 tiles through
    kmat_fast_kernel (template on op and metric) and ragged tiles through the general evaluator:
    the same entries computed both ways must agree to the last bit, and with the oracle."""
    rng = np.random.default_rng(11 + d)
    a, b = rng.normal(size=(300, d)) * 2.0, rng.normal(size=(300, d)) * 2.0
    zoo = [
        (kernels.ExpSquared(0.9), o.ExpSquared(0.9)),
        (1.7 * kernels.ExpSquared(1.3, distance=kernels.L1Distance()),
         1.7 * o.ExpSquared(1.3, distance=o.L1Distance())),
        (kernels.Matern32(0.8) * 0.6, o.Matern32(0.8) * 0.6),
        (2.5 * kernels.Matern52(1.1, distance=kernels.L2Distance()),
         2.5 * o.Matern52(1.1, distance=o.L2Distance())),
        (kernels.Exp(0.7), o.Exp(0.7)),
    ]
    for k, ko in zoo:
        full = k(a[:256], b[:256])          # 2 x 2 full tiles: straight-line kernel only
        ragged = k(a[:255], b[:255])        # tile (0, 0) straight-line, the other three general
        assert np.array_equal(full[:255, :255], ragged), type(k)
        assert ulp_diff(k(a, b), ko(a, b)) <= 4, type(k)
        # K(X, X) with the fused diagonal, lower + upper, through the GaussianProcess path is covered
        # by the factorisation tests; here: the symmetric full-tile case against the oracle
        assert ulp_diff(k(a[:256], a[:256]), ko(a[:256], a[:256])) <= 4
        v = rng.normal(size=(300, 3))
        np.testing.assert_allclose(k.matmul(a, b, v), ko(a, b) @ v, rtol=1e-12, atol=1e-12)


def test_kernel_scalar_protocol_and_matmul():
    x1, x2 = _cases.data_kernels()
    k, ko = 1.5 * kernels.Matern32(2.5), 1.5 * o.Matern32(2.5)
    np.testing.assert_allclose(k.evaluate(x1[0], x2[3]), ko(x1[:1], x2[3:4])[0, 0], rtol=1e-14)
    y = np.random.default_rng(3).normal(size=(50, 3))
    np.testing.assert_allclose(k.matmul(x1, x2, y), ko(x1, x2) @ y, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(k.matmul(x1, y[:, 0]), ko(x1, x1) @ y[:, 0], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(k.matmul(x1, y=y[:, 1]), ko(x1, x1) @ y[:, 1], rtol=1e-12, atol=1e-12)
    # more right-hand sides than one pass of the fused kernel carries (8), ragged sizes, 3-D y
    rng = np.random.default_rng(11)
    a, b = rng.normal(size=(300, 2)), rng.normal(size=(517, 2))
    Y = rng.normal(size=(517, 11))
    k2, k2o = kernels.Matern52(0.9) + 0.3 * kernels.ExpSquared(1.7), o.Matern52(0.9) + 0.3 * o.ExpSquared(1.7)
    np.testing.assert_allclose(k2.matmul(a, b, Y), k2o(a, b) @ Y, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(k2.matmul(a, b, Y.reshape(517, 11, 1)), (k2o(a, b) @ Y).reshape(300, 11, 1),
                               rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-13), (np.float32, 2e-5)])
def test_gemm_nt_vs_numpy(dtype, tol):
    """MFMA building block, asymmetric operands (catches operand / C-layout transposes)."""
    rng = np.random.default_rng(11)
    for m, n, k, lower in [(128, 128, 16, False), (256, 384, 64, False), (512, 512, 128, True),
                           (640, 256, 48, True), (384, 384, 512, True)]:
        A = rng.normal(size=(m, k)).astype(dtype)
        B = rng.normal(size=(n, k)).astype(dtype)
        C0 = rng.normal(size=(m, n)).astype(dtype)
        scale = np.sqrt(k)
        got = ll.gemm_nt(A, B, C0, -1.0, 1.0, lower)
        want = C0.astype(np.float64) - A.astype(np.float64) @ B.astype(np.float64).T
        if lower:
            # tiles on/below the diagonal are updated (tile size is an implementation detail:
            # 64 for short K, 128 otherwise); everything on/below the diagonal must be right,
            # anything above it is either updated or untouched
            tri = np.arange(m)[:, None] >= np.arange(n)[None, :]
            np.testing.assert_allclose(got[tri], want[tri], rtol=0, atol=tol * scale * 10)
            up = ~tri
            ok = np.isclose(got[up], want[up], rtol=0, atol=tol * scale * 10) | (got[up] == C0[up])
            assert ok.all()
            far = (np.arange(m)[:, None] // 128) < (np.arange(n)[None, :] // 128)
            np.testing.assert_array_equal(got[far], C0[far])
        else:
            np.testing.assert_allclose(got, want, rtol=0, atol=tol * scale * 10)
            got2 = ll.gemm_nt(A, B, C0, 1.0, 0.0, False)
            np.testing.assert_allclose(got2, A.astype(np.float64) @ B.astype(np.float64).T,
                                       rtol=0, atol=tol * scale * 10)


def _spd(n, dtype, seed=0, cond_diag=0.05):
    X, _ = _cases.synthetic.make_inputs(n, 1, seed=_cases.synthetic.SEED + seed)
    K = (1.5**2 * o.ExpSquared(2.5))(X, X) + cond_diag * np.eye(n)
    return K.astype(dtype)


@pytest.mark.parametrize("n", [128, 256, 640, 1024, 2048])
@pytest.mark.parametrize("lookahead", [0, 1])
def test_potrf_vs_lapack(n, lookahead):
    """K4: factor agrees with LAPACK dpotrf to a backward-error bound, and reconstructs K."""
    K = _spd(n, np.float64)
    L, info = ll.potrf(K, lookahead=lookahead)
    assert info == 0
    Lref = sla.cholesky(K, lower=True)
    # forward agreement scaled by cond(K) ~ 1e3 for these inputs: stay within 1e-11 relative
    np.testing.assert_allclose(L, Lref, rtol=0, atol=1e-11 * np.abs(Lref).max())
    resid = np.abs(L @ L.T - K).max() / np.abs(K).max()
    assert resid < 50 * n * np.finfo(np.float64).eps / 8, resid
    assert np.all(np.triu(L, 1) == 0)


@pytest.mark.parametrize("nb_outer", [128, 256, 512, 1024])
def test_potrf_block_sizes_agree(nb_outer):
    K = _spd(1536, np.float64, seed=1)
    L, info = ll.potrf(K, nb_outer=nb_outer)
    assert info == 0
    Lref = sla.cholesky(K, lower=True)
    np.testing.assert_allclose(L, Lref, rtol=0, atol=1e-11 * np.abs(Lref).max())


@pytest.mark.parametrize("n", [1152, 2560, 5120])
@pytest.mark.parametrize("opts", [dict(chain_kernel=0, fused_step=1, gate_split=0), dict(chain_kernel=0, fused_step=1, gate_split=1),
                                  dict(chain_kernel=0, fused_step=1, gate_split=1, lookahead=0),
                                  dict(chain_kernel=0, fused_step=0, chain_reserve=0), dict(chain_kernel=0, fused_step=1, chain_reserve=256),
                                  dict(chain_kernel=0, sub_panel=512), dict(chain_kernel=0, sub_panel=256, first_split=4),
                                  dict(chain_kernel=0, nb_first=256), dict(chain_kernel=0, nb_first=512, sub_panel=512, lookahead=0),
                                  dict(chain_kernel=0, split_tail=1), dict(chain_kernel=0, split_tail=1, sub_panel=512, nb_first=768),
                                  dict(chain_kernel=0),
                                  # the persistent chain (the default): panel by panel only, without look-ahead, two
                                  # chain workgroups per compute unit, narrow panels + early share, a narrow first panel
                                  dict(chain_full_rows=0), dict(lookahead=0), dict(lookahead=0, chain_full_rows=0),
                                  dict(chain_lds_pad=0), dict(nb_outer=512, first_split=3, chain_full_rows=2048),
                                  dict(nb_first=256, chain_full_rows=1024), dict(first_split=0, chain_reserve=0),
                                  # round 5: update tasks on the 4x4x4 MFMA form with LDS-direct operands (measured, not the default); the
                                  # followers of a chain launch behind a stream wait-value / behind the whole launch
                                  # (default: the wall-clock-bounded one-wave poll kernel)
                                  dict(chain_fast_update=1), dict(chain_polls=3), dict(chain_polls=0),
                                  dict(chain_fast_update=1, chain_polls=3, chain_full_rows=0),
                                  # round 6 (measured, not the defaults): K-batched update tasks, everywhere and in the
                                  # one-launch tail only; the next panel's first diagonal block beside the gate
                                  dict(chain_batch=4, chain_batch_minrows=0), dict(chain_batch=4, chain_full_rows=8192),
                                  dict(chain_batch=8, chain_batch_lag=2, chain_batch_rowlag=3, chain_batch_minrows=0),
                                  dict(chain_gate_split=1), dict(chain_gate_split=1, chain_full_rows=0),
                                  # the merged schedule's prefix poll as a kernel of its own (default: inside the panel's potf2)
                                  dict(chain_polls=2), dict(chain_polls=2, chain_full_rows=0),
                                  # the chain of a panel sub-panel by sub-panel (measured, not the default: profiles/r06_i)
                                  dict(chain_sub_panel=512), dict(chain_sub_panel=256, chain_sub_role=0, chain_full_rows=0)])
def test_panel_chain_variants_agree(n, opts):
    """The schedules of the panel chain -- the default persistent chain (chain_kernel: tile tasks behind a ticket
    counter, the whole rest of the matrix in one launch once few rows are left), one fused launch per 128-column
    block (panel_step_kernel), the separate launches per block with their options -- give the default schedule's
    factor (the same arithmetic per entry up to the order of the in-panel updates: 1e-12) and LAPACK's."""
    K = _spd(n, np.float64, seed=3)
    Lref, info0 = ll.potrf(K)
    L, info = ll.potrf(K, **opts)
    assert info == 0 and info0 == 0
    np.testing.assert_allclose(L, Lref, rtol=0, atol=1e-12 * np.abs(Lref).max())
    Llap = sla.cholesky(K, lower=True)
    np.testing.assert_allclose(L, Llap, rtol=0, atol=1e-11 * np.abs(Llap).max())


def test_split_tail_and_two_level_at_n6144_deterministic():
    """Round-3 schedule options at a size where the trailing updates have several rounds of tiles: the split tail
    (last round of 128 x 128 tiles cut along k, partial products combined in slice order by whichever workgroup
    arrives last) and the two-level panel agree with the default schedule to rounding and are bit-reproducible."""
    K = _spd(6144, np.float64, seed=5)
    Lref, info0 = ll.potrf(K)
    assert info0 == 0
    for opts in (dict(split_tail=1, chain_reserve=0), dict(chain_kernel=0, split_tail=1, sub_panel=512), dict(chain_full_rows=0),
                 dict(chain_kernel=0)):
        L, info = ll.potrf(K, **opts)
        assert info == 0
        np.testing.assert_allclose(L, Lref, rtol=0, atol=1e-12 * np.abs(Lref).max())
        L2, _ = ll.potrf(K, **opts)
        assert np.array_equal(L, L2), opts


def test_panel_step_fp32_and_bad_pivot():
    K = _spd(1536, np.float32, cond_diag=0.5)
    Lref = sla.cholesky(K.astype(np.float64), lower=True)
    for opts in (dict(chain_kernel=0, fused_step=1), dict(chain_kernel=1), dict(chain_kernel=1, chain_full_rows=0)):
        L, info = ll.potrf(K, **opts)
        assert info == 0
        np.testing.assert_allclose(L, Lref, rtol=5e-4, atol=5e-4)
    # first non-positive pivot is reported from inside the fused step / the persistent chain like from potf2
    Kb = _spd(1024, np.float64, seed=2)
    Kb[700, 700] = -1.0
    _, info = ll.potrf(Kb, chain_kernel=0, fused_step=1)
    _, info_chain = ll.potrf(Kb, chain_kernel=1)
    _, info_ref = ll.potrf(Kb, chain_kernel=0)
    assert info == info_chain == info_ref == 701


def test_potrf_fp32():
    K = _spd(1024, np.float32, cond_diag=0.5)
    L, info = ll.potrf(K)
    assert info == 0
    Lref = sla.cholesky(K.astype(np.float64), lower=True)
    np.testing.assert_allclose(L, Lref, rtol=5e-4, atol=5e-4)


def test_potrf_not_positive_definite_reports_pivot():
    K = _spd(512, np.float64)
    K[300, 300] = -1.0
    L, info = ll.potrf(K)
    assert info == 301  # LAPACK convention: 1-based index of the failing pivot
    assert np.isnan(L[300, 300]) and np.all(np.isfinite(L[:300, :300]))


@pytest.mark.parametrize("transpose", [False, True])
def test_trsv_vs_lapack(transpose):
    for n in (128, 384, 1280):
        K = _spd(n, np.float64, seed=2)
        L = sla.cholesky(K, lower=True)
        y = np.random.default_rng(n).normal(size=n)
        got = ll.trsv(L, y, transpose)
        want = sla.solve_triangular(L, y, lower=True, trans=1 if transpose else 0)
        np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-10 * np.abs(want).max())


def test_trsm_right_lt_vs_lapack():
    for m, n in [(128, 128), (256, 640), (384, 1152)]:
        K = _spd(n, np.float64, seed=3)
        L = sla.cholesky(K, lower=True)
        B = np.random.default_rng(m + n).normal(size=(m, n))
        got = ll.trsm_right_lt(L, B)
        want = sla.solve_triangular(L, B.T, lower=True).T
        np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-10 * np.abs(want).max())


@pytest.mark.parametrize("n,dtype", [(100, np.float64), (300, np.float64), (1100, np.float64), (5000, np.float64),
                                     (16384, np.float64), (3000, np.float32)])
def test_streaming_solves_match_lapack_and_the_stepwise_path(n, dtype):
    """The single-launch forward and backward substitutions (one workgroup per block, data-tagged
    hand-off of the solved blocks, explicit inverses of the 128 x 128 diagonal blocks) against
    LAPACK dtrtrs and against the one-launch-pair-per-block path, and bit-reproducible."""
    from tinygp_amd import GaussianProcess, _ffi

    X, y = _cases.synthetic.make_inputs(n, 1)
    k = _cases.synthetic.config_kernel(kernels, "expsq")
    diag = 0.01 if dtype == np.float64 else 0.1
    gp = GaussianProcess(k, X.astype(dtype), diag=dtype(diag))
    s = gp.solver
    L = s.scale_tril.astype(np.float64)
    want = sla.solve_triangular(L, y.astype(dtype).astype(np.float64), lower=True, check_finite=False)
    want_t = sla.solve_triangular(L, y.astype(dtype).astype(np.float64), lower=True, trans=1, check_finite=False)
    Y3 = np.stack([y, np.cos(X), np.ones(n)], axis=1).astype(dtype)
    want_t3 = sla.solve_triangular(L, Y3.astype(np.float64), lower=True, trans=1, check_finite=False)
    ctx = _ffi.default_ctx()
    got, got_t = {}, {}
    for mode in (1, 0):
        old = ctx.set_option("stream_trsv", mode)
        try:
            got[mode] = s.solve_triangular(y.astype(dtype))
            lp = float(s.log_probability(y.astype(dtype)))
            again = [s.solve_triangular(y.astype(dtype)) for _ in range(3)]
            got_t[mode] = s.solve_triangular(y.astype(dtype), transpose=True)
            again_t = [s.solve_triangular(y.astype(dtype), transpose=True) for _ in range(3)]
            t3 = s.solve_triangular(Y3, transpose=True)
        finally:
            ctx.set_option("stream_trsv", old)
        assert all(np.array_equal(a, got[mode]) for a in again)
        assert all(np.array_equal(a, got_t[mode]) for a in again_t)
        tol = 1e-9 if dtype == np.float64 else 2e-3
        scale = np.max(np.abs(want))
        assert np.max(np.abs(got[mode] - want)) <= tol * scale, (mode, np.max(np.abs(got[mode] - want)) / scale)
        assert np.max(np.abs(got_t[mode] - want_t)) <= tol * np.max(np.abs(want_t)), mode
        assert np.max(np.abs(t3 - want_t3)) <= tol * np.max(np.abs(want_t3)), mode
        ref = -0.5 * want @ want - np.sum(np.log(np.diag(L))) - 0.5 * n * np.log(2 * np.pi)
        np.testing.assert_allclose(lp, ref, rtol=1e-9 if dtype == np.float64 else 5e-4)
    np.testing.assert_allclose(got[1], got[0], rtol=0, atol=(1e-10 if dtype == np.float64 else 1e-3) * np.max(np.abs(want)))
    np.testing.assert_allclose(got_t[1], got_t[0], rtol=0, atol=(1e-10 if dtype == np.float64 else 1e-3) * np.max(np.abs(want_t)))
