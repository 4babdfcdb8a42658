"""Several ranks of the block-column driver on the ONE GPU a test box has.

RCCL refuses two ranks on the same device, so the collectives of this test go through `gloo`
(which accepts device tensors); everything else is the product: HipBlockOps, csrc/dist.hip through
the C ABI, ring slots in device memory, the driver's streams.  This is the only place where the
RECEIVER side of the HIP path runs (a rank that does not own the panel: arrival -> forward step ->
update of its own block columns from the received slot), and where owners alternate between
processes.  Results must match the oracle and be bit-identical across ranks."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent


def _k(mod):
    return 1.5**2 * mod.ExpSquared(2.5) + 0.3 * mod.Matern32(1.2)


def _worker(rank, world, port, n, nb, dtype_name, m_test, q):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0")
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tinygp_amd import kernels, synthetic
        from tinygp_amd.distributed import BlockCyclicCholesky, HipBlockOps

        dt = np.dtype(dtype_name)
        X, y = synthetic.make_inputs(n, 1)
        diag = 0.01 if dt == np.float64 else 0.1
        s = BlockCyclicCholesky(_k(kernels), X.astype(dt), np.full(n, diag, dtype=dt), nb=nb,
                                ops=HipBlockOps(0), dist=dist)
        ll = s.log_probability(y.astype(dt))
        xt = np.linspace(X[0], X[-1], m_test)
        mean = s.condition_mean(y.astype(dt), xt.astype(dt))
        # solves on the RESIDENT factor with peers: fan-in forward solve (one reduce per block column), the (M,) all-reduce
        Y = np.random.default_rng(3).normal(size=(n, 3)).astype(dt)
        res = (s.solve_triangular(Y), s.solve_triangular(y.astype(dt), transpose=True), s.condition_colsumsq(xt.astype(dt)))
        # round 5 with peers: ONE blocked pass for (N, R) transposed; value-and-gradient (chunked K^-1 solves: every rank
        # contracts its own block rows, one all-reduce)
        bwdR = s.solve_triangular(Y, transpose=True)
        s.GRAD_CHUNK = 512
        gll, grad = s.log_probability_and_grad(y.astype(dt))
        res = res + (bwdR, gll, np.array(grad["kernel"]), grad["noise_diag"])
        ll2 = s.log_probability(y.astype(dt), kernel=1.1 * _k(kernels))  # the optimiser's next step
        # this rank's block columns of the factor (lower part), for the LAPACK comparison
        cols = [(j, s.ops.column(l, s.rows(j))) for l, j in enumerate(s.owned)]
        q.put((rank, ll, s.info, mean, ll2, s.bytes_received, cols, res))
        s.close(close_ops=True)  # (the communicator this driver made first, then the operations)
    finally:
        dist.destroy_process_group()


def _run(world, n, nb, dtype_name, m_test):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, nb, dtype_name, m_test, q)) for r in range(world)]
    [p.start() for p in procs]
    out = sorted((q.get(timeout=300 if world <= 3 else 900) for _ in range(world)), key=lambda t: t[0])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return out


@pytest.mark.parametrize("world,n,nb,dtype_name,rtol", [(2, 3000, 512, "float64", 1e-8),
                                                         (3, 2500, 256, "float64", 1e-8),
                                                         (2, 5000, 1024, "float64", 1e-8),
                                                         # round 6: EIGHT ranks (two block columns each, seven peers per
                                                         # panel, a reduce chain over eight owners in the forward solve)
                                                         (8, 4000, 256, "float64", 1e-8),
                                                         (2, 2000, 256, "float32", 5e-4)])
def test_block_column_driver_with_peers_on_one_gpu(world, n, nb, dtype_name, rtol):
    from oracle import tinygp_np as o
    from tinygp_amd import synthetic

    X, y = synthetic.make_inputs(n, 1)
    diag = 0.01 if dtype_name == "float64" else 0.1
    gp = o.GaussianProcess(_k(o), X, diag=diag)
    want = float(gp.log_probability(y))
    m_test = 41
    xt = np.linspace(X[0], X[-1], m_test)
    want_mean = gp.predict(y, xt)
    want2 = float(o.GaussianProcess(1.1 * _k(o), X, diag=diag).log_probability(y))
    from oracle import grad_np

    # _k = 1.5^2 ExpSquared(2.5) + 0.3 Matern32(1.2): program [const, expsq, mul, const, m32, mul, add]
    want_grad = grad_np.log_probability_and_grad(lambda t: t[0] * o.ExpSquared(t[1]) + t[2] * o.Matern32(t[3]),
                                                 np.array([1.5**2, 2.5, 0.3, 1.2]), X, diag, y) if dtype_name == "float64" else None
    out = _run(world, n, nb, dtype_name, m_test)
    tol = dict(rtol=5e-7, atol=5e-7) if dtype_name == "float64" else dict(rtol=5e-4, atol=5e-4)
    L = gp.solver.scale_tril
    nblk = -(-n // nb)
    npad = nblk * nb
    es = 8 if dtype_name == "float64" else 4
    seen = set()
    import scipy.linalg as sla

    Y = np.random.default_rng(3).normal(size=(n, 3))
    Aw = sla.solve_triangular(L, _k(o)(X, xt), lower=True)
    for rank, ll, info, mean, ll2, nbytes, cols, res in out:
        assert info == 0
        np.testing.assert_allclose(res[0], sla.solve_triangular(L, Y, lower=True), rtol=rtol * 10, atol=1e-7 if dtype_name == "float64" else 5e-3)
        np.testing.assert_allclose(res[1], sla.solve_triangular(L, y, lower=True, trans=1), rtol=rtol * 10,
                                   atol=1e-7 if dtype_name == "float64" else 5e-2)
        np.testing.assert_allclose(res[2], np.sum(Aw * Aw, axis=0), **tol)
        np.testing.assert_allclose(res[3], sla.solve_triangular(L, Y, lower=True, trans=1), rtol=rtol * 10,
                                   atol=1e-7 if dtype_name == "float64" else 5e-2)
        np.testing.assert_allclose(res[4], want, rtol=rtol)
        if dtype_name == "float64":
            np.testing.assert_allclose([res[5][2 * i] for i in (0, 1, 3, 4)], want_grad[1], rtol=2e-6,
                                       atol=2e-6 * np.abs(want_grad[1]).max())
            np.testing.assert_allclose(res[6], want_grad[2], rtol=1e-6, atol=1e-6 * np.abs(want_grad[2]).max())
        np.testing.assert_allclose(ll, want, rtol=rtol)
        np.testing.assert_allclose(ll2, want2, rtol=rtol)
        np.testing.assert_allclose(mean, want_mean, **tol)
        expect = sum(((npad - k * nb) * nb + (nb // 128) * 2048) * es for k in range(nblk) if k % world != rank)
        assert nbytes == expect
        for j, col in cols:  # (rows, nb) from the diagonal block down
            seen.add(j)
            if dtype_name != "float64":
                continue
            j0 = j * nb
            rows, ncol = min(n - j0, col.shape[0]), min(nb, n - j0)
            # the second factorisation (1.1 x kernel) is what sits in memory now
            idx = np.tril_indices(rows, 0, ncol)
            L2 = o.GaussianProcess(1.1 * _k(o), X, diag=diag).solver.scale_tril
            np.testing.assert_allclose(col[:rows, :ncol][idx], L2[j0:j0 + rows, j0:j0 + ncol][idx],
                                       rtol=1e-9, atol=1e-9)
    assert seen == set(range(nblk))          # every block column has exactly one owner
    assert len({t[1] for t in out}) == 1     # the log-likelihood is bit-identical on every rank
    assert len({t[4] for t in out}) == 1
    for t in out[1:]:
        assert np.array_equal(t[3], out[0][3])
        assert np.array_equal(t[7][0], out[0][7][0]) and np.array_equal(t[7][2], out[0][7][2])  # replicated solves too
    del L
