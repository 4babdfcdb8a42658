"""Host logic of tinygp_amd.comm without a GPU: the communicator-id exchange (TCP and file), the buffer views of the
block-column driver, and which transport the driver picks by default."""
import os
import sys
import threading
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world", [1, 2, 5])
def test_tcp_exchange_hands_rank_zeros_payload_to_every_rank(world):
    from tinygp_amd.comm import _exchange_tcp

    payload = bytes(range(128))
    port = _free_port()
    got = [None] * world
    errs = []

    def run(r):
        try:
            got[r] = _exchange_tcp(r, world, payload if r == 0 else None, "127.0.0.1", port, timeout=30.0)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    # ranks 1.. start FIRST: they must retry until rank 0 listens
    th = [threading.Thread(target=run, args=(r,)) for r in list(range(1, world)) + [0]]
    for t in th:
        t.start()
    for t in th:
        t.join(60)
    assert not errs, errs
    assert all(g == payload for g in got)


def test_tcp_exchange_times_out_loudly_without_rank_zero():
    from tinygp_amd import _ffi
    from tinygp_amd.comm import _exchange_tcp

    with pytest.raises(_ffi.TgpError, match="no communicator id"):
        _exchange_tcp(1, 2, None, "127.0.0.1", _free_port(), timeout=0.5)


def test_devbuf_views_are_contiguous_slices():
    from tinygp_amd.distributed import DevBuf

    class Ops:  # (views never release anything)
        pass

    b = DevBuf(Ops(), 4096, (10, 3), np.float64)
    assert (b.count, b.nbytes) == (30, 240)
    r = b.rows(2, 5)
    assert (r.ptr, r.shape, r.count) == (4096 + 2 * 3 * 8, (3, 3), 9)
    f = b.flat(4, 10)
    assert (f.ptr, f.shape) == (4096 + 32, (6,))
    v = DevBuf(Ops(), 64, (256,), np.float32).rows(128, 256)
    assert (v.ptr, v.count, v.code) == (64 + 512, 128, 0)


def test_default_transport_for_stand_in_operations_is_torch_distributed():
    """(the HIP operations pick RCCL: tests/test_gpu_5_distributed.py)"""
    from tinygp_amd.comm import TorchComm
    from tinygp_amd.distributed import BlockCyclicCholesky

    class FakeDist:
        ReduceOp = None

        @staticmethod
        def get_rank(group=None):
            return 1

        @staticmethod
        def get_world_size(group=None):
            return 4

    c = BlockCyclicCholesky._default_comm(object(), FakeDist, None)
    assert isinstance(c, TorchComm) and (c.rank, c.world) == (1, 4)


def test_rccl_loads_lazily_and_the_single_gpu_path_never_maps_it():
    """librccl is dlopen'ed by the first tgp_comm_* call only."""
    import subprocess

    code = ("from tinygp_amd import _ffi; _ffi.lib()\n"
            "print('RCCL', any('librccl' in l for l in open('/proc/self/maps')))\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(ROOT), timeout=300,
                       env=dict(os.environ, PYTHONPATH=str(ROOT)))
    assert "RCCL False" in r.stdout, r.stdout + r.stderr[-500:]
