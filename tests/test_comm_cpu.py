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


def test_tcp_exchange_ignores_strangers_and_counts_distinct_ranks(monkeypatch):
    """A stray connection (port scanner, health check, a rank of ANOTHER job) gets no id and uses up no slot."""
    import socket
    import struct
    import time

    from tinygp_amd import comm as comm_mod

    monkeypatch.setenv("TGP_COMM_NONCE", "job-a")
    payload = bytes(range(128))
    port = _free_port()
    got, errs, stray = {}, [], {}

    def run(r):
        try:
            got[r] = comm_mod._exchange_tcp(r, 3, payload if r == 0 else None, "127.0.0.1", port, timeout=30.0)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    t0 = threading.Thread(target=run, args=(0,))
    t0.start()
    deadline = time.monotonic() + 10
    while time.monotonic() < deadline:  # a stranger that says nothing, and one with another job's nonce
        try:
            with socket.create_connection(("127.0.0.1", port), timeout=2.0) as s:
                s.sendall(b"GET / HTTP/1.0\r\n\r\n")
                s.settimeout(10.0)
                stray["http"] = s.recv(256)
            break
        except OSError:
            time.sleep(0.02)
    with socket.create_connection(("127.0.0.1", port), timeout=2.0) as s:
        s.sendall(b"TGPC" + struct.pack("<I", 1) + b"x" * 16)
        s.settimeout(10.0)
        stray["other_job"] = s.recv(256)
    assert stray == {"http": b"", "other_job": b""}  # closed without a byte of the id
    th = [threading.Thread(target=run, args=(r,)) for r in (1, 2)]
    for t in th:
        t.start()
    for t in th + [t0]:
        t.join(60)
    assert not errs, errs
    assert got == {0: payload, 1: payload, 2: payload}


def test_tcp_exchange_rejects_a_port_out_of_range():
    from tinygp_amd.comm import _exchange_tcp

    with pytest.raises(ValueError, match="outside 1..65535"):
        _exchange_tcp(1, 2, None, "127.0.0.1", 65536 + 29, timeout=0.5)


def test_id_file_of_an_earlier_run_is_not_joined(tmp_path, monkeypatch):
    """RcclComm.from_file: a reader skips a stale record (old stamp or another job's nonce) and takes the fresh one."""
    import struct
    import time

    from tinygp_amd import comm as comm_mod

    made = []
    joined = threading.Barrier(2)  # ncclCommInitRank is a collective: it returns when every rank has called it

    class Fake(comm_mod.RcclComm):
        def __init__(self, ctx, world, rank, uid):  # (no library, no GPU: record what would be joined)
            made.append((rank, uid))
            joined.wait(20)

        def __del__(self):
            pass

        @staticmethod
        def unique_id():
            return b"F" * comm_mod.ID_BYTES

    monkeypatch.setenv("TGP_COMM_NONCE", "job-b")
    path = tmp_path / "id"
    nonce = comm_mod._job_nonce()
    path.write_bytes(b"TGPC" + nonce + struct.pack("<d", time.time() - 3600.0) + b"S" * comm_mod.ID_BYTES)  # stale stamp
    errs = []

    def reader():
        try:
            Fake.from_file(None, path, 2, 1, timeout=20.0)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    t = threading.Thread(target=reader)
    t.start()
    time.sleep(0.3)
    assert not made  # still waiting: the stale record was not joined
    path.write_bytes(b"TGPC" + b"y" * 16 + struct.pack("<d", time.time()) + b"O" * comm_mod.ID_BYTES)  # another job's
    time.sleep(0.3)
    assert not made
    Fake.from_file(None, path, 2, 0, timeout=20.0)  # rank 0 of THIS job: unlinks, writes, "joins", removes the file
    t.join(30)
    assert not errs, errs
    assert sorted(made) == [(0, b"F" * comm_mod.ID_BYTES), (1, b"F" * comm_mod.ID_BYTES)]
    assert not path.exists()


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
