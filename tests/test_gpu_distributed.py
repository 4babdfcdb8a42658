"""The block-cyclic driver with the REAL block operations (HipBlockOps: torch CUDA buffers,
C ABI on a torch stream, RCCL collectives) on the one GPU a test box has: world size 1, so
every panel is 'broadcast' to itself.  Multi-rank schedule logic is covered on CPU under
gloo (tests/test_distributed_cpu.py); 8-GPU runs are the driver's."""
import os

import numpy as np
import pytest

import _cases
from oracle import tinygp_np as o

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pg():
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("n,nb,dtype,rtol", [(2000, 256, np.float64, 1e-8), (3000, 512, np.float64, 1e-8),
                                              (1500, 128, np.float32, 5e-4)])
def test_block_cyclic_hip_single_rank(pg, n, nb, dtype, rtol):
    from tinygp_amd import kernels
    from tinygp_amd.distributed import BlockCyclicCholesky

    X, y = _cases.synthetic.make_inputs(n, 1)
    diag = 0.01 if dtype == np.float64 else 0.1
    k = 1.5**2 * kernels.ExpSquared(2.5) + 0.3 * kernels.Matern32(1.2)
    s = BlockCyclicCholesky(k, X.astype(dtype), np.full(n, diag, dtype=dtype), nb=nb, dist=pg)
    got = s.log_probability(y.astype(dtype))
    want = float(o.GaussianProcess(1.5**2 * o.ExpSquared(2.5) + 0.3 * o.Matern32(1.2), X,
                                   diag=diag).log_probability(y))
    assert s.info == 0
    np.testing.assert_allclose(got, want, rtol=rtol)
