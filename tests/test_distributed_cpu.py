"""The block-cyclic multi-GPU schedule under `gloo`, world size 2, on CPUs: ownership,
look-ahead order, panel broadcasts and the slice all-reduces of the forward solve, with a
NumPy stand-in for the per-block HIP kernels (tests/_numpy_blockops.py)."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, n, nb, bad, q):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1",
                      OPENBLAS_NUM_THREADS="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from _numpy_blockops import NumpyBlockOps
        from tinygp_amd import kernels, synthetic
        from tinygp_amd.distributed import BlockCyclicCholesky

        X, y = synthetic.make_inputs(n, 1)
        diag = np.full(n, 0.01)
        if bad:
            diag[bad] = -5.0
        k = 1.5**2 * kernels.ExpSquared(2.5) + 0.3 * kernels.Matern32(1.2)
        s = BlockCyclicCholesky(k, X, diag, nb=nb, ops=NumpyBlockOps(), dist=dist)
        assert s.owned == [j for j in range(s.nblk) if j % world == rank]
        ll = s.log_probability(y)
        q.put((rank, ll, s.info))
    finally:
        dist.destroy_process_group()


def _run(world, n, nb, bad=0):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, nb, bad, q)) for r in range(world)]
    [p.start() for p in procs]
    out = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    return out


@pytest.mark.parametrize("world,n,nb", [(2, 450, 128), (2, 512, 256), (3, 600, 128)])
def test_block_cyclic_log_probability_matches_oracle(world, n, nb):
    from oracle import tinygp_np as o
    from tinygp_amd import synthetic

    X, y = synthetic.make_inputs(n, 1)
    want = float(o.GaussianProcess(1.5**2 * o.ExpSquared(2.5) + 0.3 * o.Matern32(1.2), X,
                                   diag=0.01).log_probability(y))
    out = _run(world, n, nb)
    for rank, ll, info in out:  # every rank ends with the same scalar
        assert info == 0
        np.testing.assert_allclose(ll, want, rtol=1e-9)


def test_block_cyclic_reports_first_bad_pivot_on_every_rank():
    out = _run(2, 500, 128, bad=300)
    for rank, ll, info in out:
        assert info == 301 and ll == -np.inf


def test_kernel_program_lowering_matches_oracle_classes():
    """The host-side postfix lowering, evaluated by oracle/ref_prog.py, equals the oracle's
    class-based evaluation for every kernel of the zoo (no GPU involved)."""
    import _cases
    from oracle import ref_prog
    from oracle import tinygp_np as o
    from tinygp_amd import kernels

    x1, x2 = _cases.data_kernels()
    for name, k in _cases.kernel_zoo(kernels).items():
        got = ref_prog.eval_matrix(k.program(), x1, x2)
        np.testing.assert_allclose(got, _cases.kernel_zoo(o)[name](x1, x2), rtol=1e-14, atol=1e-15, err_msg=name)
