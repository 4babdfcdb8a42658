"""The block-cyclic multi-GPU schedule under `gloo` on CPUs, world sizes 2, 3 and 4: ownership,
look-ahead order, ring-slot reuse, panel broadcasts, the replicated forward solve, the slice
broadcasts of the backward solve and the (M,) all-reduce of the conditional mean -- with a NumPy
stand-in for the per-rank device operations (tests/_numpy_blockops.py)."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _kernels(mod):
    return 1.5**2 * mod.ExpSquared(2.5) + 0.3 * mod.Matern32(1.2)


def _worker(rank, world, port, n, nb, bad, m_test, q, chunk_min=None, fail_at=None):
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
        ops = NumpyBlockOps()
        if fail_at is not None and rank == fail_at[0]:  # a rank-local failure in the middle of the step loop
            real_rest = ops.rest

            def failing_rest(k):
                if k == fail_at[1]:
                    raise RuntimeError("injected failure")
                real_rest(k)

            ops.rest = failing_rest
        s = BlockCyclicCholesky(_kernels(kernels), X, diag, nb=nb, ops=ops, dist=dist)
        if chunk_min is not None:
            s.CHUNK_MIN_BYTES = chunk_min
        if fail_at is not None:
            try:
                s.log_probability(y)
                q.put((rank, "no error"))
            except Exception as e:  # noqa: BLE001
                q.put((rank, type(e).__name__ + ": " + str(e)))
            return
        assert s.owned == [j for j in range(s.nblk) if j % world == rank]
        ll = s.log_probability(y)
        mean = None
        if m_test:
            xt = np.linspace(X[0], X[-1], m_test)
            mean = s.condition_mean(y, xt)
            # a DIFFERENT right-hand side must not reuse the cached solve of y (round-2 advisor finding)
            mean_other = s.condition_mean(3.0 * y + 1.0, xt)
            # a second evaluation (the optimiser step) re-assembles and re-factors in place
            ll2 = s.log_probability(y, kernel=1.1 * _kernels(kernels))
        else:
            ll2 = mean_other = None
        # host order of the schedule (the LAST factorisation's calls): on the owner of k+1 the gate and the chain
        # of panel k+1 are enqueued before this rank's forward step / updates of step k -- the chain pipeline
        # never queues behind a big update --, block column k+2 is brought up to date first on its owner, and
        # a receiver makes the slot wait for its last readers before the collective is issued
        calls = ops.calls[len(ops.calls) - 1 - ops.calls[::-1].index(("assemble",)):]
        for k in range(s.nblk - 1):
            if (k + 1) % world == rank:
                assert calls.index(("lookahead", k)) < calls.index(("panel", k + 1)) < calls.index(("arrived", k))
            else:
                assert calls.index(("slot_ready", k + 1)) < calls.index(("arrived", k))
            if k + 2 < s.nblk and (k + 2) % world == rank:
                assert calls.index(("fwd_step", k)) < calls.index(("pre_update*", k)) < calls.index(("rest", k))
        q.put((rank, ll, s.info, mean, ll2, s.bytes_received, mean_other))
    finally:
        dist.destroy_process_group()


def _run(world, n, nb, bad=0, m_test=0, chunk_min=None, fail_at=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, nb, bad, m_test, q, chunk_min, fail_at))
             for r in range(world)]
    [p.start() for p in procs]
    out = sorted((q.get(timeout=180) for _ in range(world)), key=lambda t: t[0])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    return out


@pytest.mark.parametrize("world,n,nb,chunk_min", [(2, 450, 128, None), (2, 512, 256, None), (3, 600, 128, None),
                                                  (4, 1100, 128, None),
                                                  (2, 1300, 512, 0), (3, 1500, 256, 0)])  # 0: every panel in chunks
def test_block_cyclic_log_probability_and_condition_mean_match_oracle(world, n, nb, chunk_min):
    from oracle import tinygp_np as o
    from tinygp_amd import synthetic

    X, y = synthetic.make_inputs(n, 1)
    gp = o.GaussianProcess(_kernels(o), X, diag=0.01)
    want = float(gp.log_probability(y))
    xt = np.linspace(X[0], X[-1], 37)
    want_mean = gp.predict(y, xt)
    want2 = float(o.GaussianProcess(1.1 * _kernels(o), X, diag=0.01).log_probability(y))
    out = _run(world, n, nb, m_test=37, chunk_min=chunk_min)
    nblk = -(-n // nb)
    npad = nblk * nb
    want_other = gp.predict(3.0 * y + 1.0, xt)
    for rank, ll, info, mean, ll2, nbytes, mean_other in out:  # every rank ends with the same results
        assert info == 0
        np.testing.assert_allclose(ll, want, rtol=1e-9)
        np.testing.assert_allclose(mean, want_mean, rtol=5e-7, atol=5e-7)
        np.testing.assert_allclose(mean_other, want_other, rtol=5e-7, atol=5e-7)
        np.testing.assert_allclose(ll2, want2, rtol=1e-9)
        # broadcast volume: every panel this rank does not own, rows x nb (+ dinv)
        expect = sum(((npad - k * nb) * nb + (nb // 128) * 2048) * 8 for k in range(nblk) if k % world != rank)
        assert nbytes == expect  # whole panels, in however many chunks they travel
    assert len({o_[1] for o_ in out}) == 1  # bit-identical across ranks (replicated solve)


def test_block_cyclic_reports_first_bad_pivot_on_every_rank():
    out = _run(2, 500, 128, bad=300)
    for rank, ll, info, *_ in out:
        assert info == 301 and ll == -np.inf


def test_a_rank_local_failure_raises_on_every_rank_instead_of_hanging():
    """Round-2 advisor finding: one rank failing between two collectives left its peers blocked in the next one.
    Now the failing rank keeps issuing its collectives, the error travels in the agreed all-reduce, and every
    rank raises."""
    out = _run(3, 900, 128, fail_at=(1, 2))
    msgs = dict(out)
    assert "injected failure" in msgs[1]
    # the peers either see the agreed error flag or trip over the garbage panels the failing rank kept sending --
    # what matters: nobody hangs (the queue above would time out) and nobody reports a result
    assert all(msgs[r] != "no error" for r in (0, 2)), msgs


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
