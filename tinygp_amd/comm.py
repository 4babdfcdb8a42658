"""Collectives of the block-column driver (:mod:`tinygp_amd.distributed`).

The product path is :class:`RcclComm`: RCCL called by ``libtgp_hip.so`` itself (``csrc/comm.hip``: ``ncclBroadcast`` /
``ncclReduce`` / ``ncclAllReduce`` on the library's own streams, on plain device pointers) -- north_star's "RCCL
broadcast of the current panel and reduce of the solve RHS over xGMI" with no ``torch.distributed`` process group
between the panel chain and the wire.  One process per GPU; the 128-byte ``ncclUniqueId`` that rank 0 creates travels
through whatever side channel the launcher offers:

* :meth:`RcclComm.from_env` -- ``RANK`` / ``WORLD_SIZE`` / ``LOCAL_RANK`` / ``MASTER_ADDR`` / ``MASTER_PORT`` as
  ``torchrun`` (or any launcher) sets them; the id goes over one TCP connection per rank to ``MASTER_ADDR`` at
  ``TGP_COMM_PORT`` (default ``MASTER_PORT + 29``).  No torch anywhere.
* :meth:`RcclComm.from_file` -- a path on a file system all ranks see.
* :meth:`RcclComm.from_torch` -- an initialised ``torch.distributed`` group carries that ONE message
  (``broadcast_object_list``); the data path does not touch torch.

Two more implementations of the same five methods exist for tests and are never the default on a multi-GPU node:
:class:`HostStagedComm` (device buffers through host memory and a CPU ``gloo`` group: several ranks sharing the ONE GPU
of a test box, which RCCL refuses) and :class:`TorchComm` (the CPU stand-in of the per-rank operations under ``gloo``).

Interface (``stream``: 0 = the driver's main stream, 1 = its priority stream; buffers are what the rank's ``ops``
hand out):

``broadcast(buf, root, stream) -> work``   asynchronous; ``work.wait(stream)`` makes ``stream`` wait for it (no host block)
``reduce(buf, root, stream)``              sum to ``root``, in place, ordered on ``stream``
``all_reduce(buf, stream)``                sum, in place
``agree_min(value) -> float``              host scalar, the minimum over the ranks (the agreed potrf info)
``marker(stream) -> work``                 "everything enqueued on ``stream`` so far"
"""

from __future__ import annotations

import ctypes as C
import os
import socket
import struct
import time

import numpy as np

from tinygp_amd import _ffi

__all__ = ["RcclComm", "HostStagedComm", "TorchComm", "MAIN", "PANEL", "ID_BYTES"]

MAIN, PANEL = 0, 1
ID_BYTES = 128


class _Ticket:
    """Everything enqueued on a stream of the library up to a point; ``wait(stream)`` orders another stream behind it."""

    def __init__(self, comm, ticket):
        self.comm, self.ticket = comm, ticket

    def wait(self, stream: int = MAIN):
        _ffi.check(_ffi.lib().tgp_comm_wait(self.comm.handle, stream, self.ticket), "tgp_comm_wait")


class _Complete:
    """A collective that had completed on the host when the call returned."""

    def wait(self, stream: int = MAIN):
        return None


def _job_nonce() -> bytes:
    """16 bytes that every rank of ONE job derives the same way and another job does not: ``TGP_COMM_NONCE`` if set,
    else the launcher's run id (``TORCHELASTIC_RUN_ID``) with the rendezvous endpoint, else zeros (no check)."""
    import hashlib

    key = os.environ.get("TGP_COMM_NONCE")
    if key is None and os.environ.get("TORCHELASTIC_RUN_ID"):
        key = "|".join(os.environ.get(k, "") for k in ("TORCHELASTIC_RUN_ID", "MASTER_ADDR", "MASTER_PORT"))
    return hashlib.sha256(key.encode()).digest()[:16] if key else b"\0" * 16


_HELLO = b"TGPC"


def _recv_exact(s, n: int) -> bytes:
    data = b""
    while len(data) < n:
        chunk = s.recv(n - len(data))
        if not chunk:
            raise ConnectionError("closed")
        data += chunk
    return data


def _exchange_tcp(rank: int, world: int, payload: bytes | None, addr: str, port: int, timeout: float) -> bytes:
    """Rank 0 serves ``payload`` to the world - 1 OTHER RANKS of this job; the others fetch it (retrying until rank 0
    listens).  A client says who it is first -- magic, rank, job nonce -- and rank 0 counts distinct valid ranks, not raw
    connections: a port scanner, a health check or a leftover rank of another job is closed without using up a slot
    and without seeing the id (advisor r5).  Rank 0 listens on ``addr`` (MASTER_ADDR), not on every interface."""
    if world == 1:
        return payload
    if not 0 < port < 65536:
        raise ValueError(f"communicator id port {port} is outside 1..65535 (set TGP_COMM_PORT)")
    nonce = _job_nonce()
    if rank == 0:
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        try:
            srv.bind((addr, port))
        except OSError:  # MASTER_ADDR names another interface of this host (or a name that does not resolve here)
            srv.bind(("", port))
        srv.listen(world + 8)
        deadline = time.monotonic() + timeout
        served = set()
        try:
            while len(served) < world - 1:
                left = deadline - time.monotonic()
                if left <= 0:
                    raise _ffi.TgpError(f"rank 0: only {len(served)} of {world - 1} ranks fetched the communicator id "
                                        f"within {timeout:.0f} s")
                srv.settimeout(left)
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    continue
                with conn:
                    try:
                        conn.settimeout(5.0)
                        hello = _recv_exact(conn, 4 + 4 + 16)
                        (peer,) = struct.unpack("<I", hello[4:8])
                        if hello[:4] != _HELLO or hello[8:] != nonce or not 0 < peer < world:
                            continue  # not a rank of this job: closed, no id, no slot
                        conn.sendall(struct.pack("<I", len(payload)) + payload)
                        served.add(peer)
                    except (ConnectionError, OSError):
                        continue
        finally:
            srv.close()
        return payload
    deadline = time.monotonic() + timeout
    last = None
    while time.monotonic() < deadline:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as s:
                s.settimeout(timeout)
                s.sendall(_HELLO + struct.pack("<I", rank) + nonce)
                (n,) = struct.unpack("<I", _recv_exact(s, 4))
                return _recv_exact(s, n)
        except (ConnectionError, OSError) as e:  # rank 0 is not listening yet (or belongs to another job)
            last = e
            time.sleep(0.05)
    raise _ffi.TgpError(f"rank {rank}: no communicator id from {addr}:{port} within {timeout:.0f} s ({last})")


class RcclComm:
    """One rank of an RCCL communicator owned by the library (``tgp_comm``)."""

    def __init__(self, ctx: _ffi.Ctx, world: int, rank: int, uid: bytes):
        if len(uid) != ID_BYTES:
            raise ValueError(f"an ncclUniqueId is {ID_BYTES} bytes, got {len(uid)}")
        self.ctx, self.world, self.rank = ctx, int(world), int(rank)
        h = C.c_void_p()
        _ffi.check(_ffi.lib().tgp_comm_create(ctx.handle, self.world, self.rank, C.c_char_p(uid), C.byref(h)),
                   "tgp_comm_create")
        self.handle = h
        self._scratch = ctx.malloc(64)

    # -- construction -----------------------------------------------------------------------------------------------
    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(ID_BYTES)
        _ffi.check(_ffi.lib().tgp_comm_unique_id(buf, None), "tgp_comm_unique_id")
        return buf.raw

    @classmethod
    def from_env(cls, ctx: _ffi.Ctx, timeout: float = 600.0) -> "RcclComm":
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(os.environ.get("TGP_COMM_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 29))
        uid = _exchange_tcp(rank, world, cls.unique_id() if rank == 0 else None, addr, port, timeout)
        return cls(ctx, world, rank, uid)

    @classmethod
    def from_file(cls, ctx: _ffi.Ctx, path, world: int, rank: int, timeout: float = 600.0,
                  stale_after: float = 120.0) -> "RcclComm":
        """The id travels through a file on a shared path.  The file is ``magic | job nonce | wall-clock stamp | id``:
        rank 0 REMOVES whatever is at the path before it writes, a reader accepts only a file of ITS job (nonce, see
        ``_job_nonce``) that is not older than ``stale_after`` seconds before the reader started waiting -- a file an
        earlier run or an earlier communicator left behind is ignored, not joined (advisor r5) -- and rank 0 removes the
        file once the communicator is up (``ncclCommInitRank`` returns when every rank has joined, i.e. has read it)."""
        path = os.fspath(path)
        nonce = _job_nonce()
        t_start = time.time()
        if rank == 0:
            try:
                os.unlink(path)
            except FileNotFoundError:
                pass
            tmp = f"{path}.tmp.{os.getpid()}"
            with open(tmp, "wb") as f:
                f.write(_HELLO + nonce + struct.pack("<d", time.time()) + cls.unique_id())
            os.replace(tmp, path)  # atomic: a reader sees the whole record or no file
        deadline = time.monotonic() + timeout
        uid = None
        while uid is None:
            try:
                with open(path, "rb") as f:
                    rec = f.read()
            except FileNotFoundError:
                rec = b""
            if len(rec) == 4 + 16 + 8 + ID_BYTES and rec[:4] == _HELLO and rec[4:20] == nonce:
                (stamp,) = struct.unpack("<d", rec[20:28])
                if stamp >= t_start - stale_after:
                    uid = rec[28:]
                    break
            if time.monotonic() > deadline:
                raise _ffi.TgpError(f"rank {rank}: no fresh communicator id at {path} within {timeout:.0f} s")
            time.sleep(0.02)
        comm = cls(ctx, world, rank, uid)
        if rank == 0:
            try:
                os.unlink(path)
            except OSError:
                pass
        return comm

    @classmethod
    def from_torch(cls, ctx: _ffi.Ctx, dist=None, group=None) -> "RcclComm":
        """``torch.distributed`` for ONE message -- the communicator id -- and nothing else."""
        if dist is None:
            import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        if world > 1:
            src = 0 if group is None else dist.get_global_rank(group, 0)
            dist.broadcast_object_list(box, src=src, group=group)
        return cls(ctx, world, rank, box[0])

    # -- collectives --------------------------------------------------------------------------------------------------
    def marker(self, stream: int = MAIN) -> _Ticket:
        t = C.c_int64()
        _ffi.check(_ffi.lib().tgp_comm_record(self.handle, stream, C.byref(t)), "tgp_comm_record")
        return _Ticket(self, t.value)

    def broadcast(self, buf, root: int, stream: int = MAIN) -> _Ticket:
        _ffi.check(_ffi.lib().tgp_comm_broadcast(self.handle, stream, C.c_void_p(buf.ptr), buf.count, buf.code, int(root)),
                   "tgp_comm_broadcast")
        return self.marker(stream)

    def reduce(self, buf, root: int, stream: int = MAIN):
        _ffi.check(_ffi.lib().tgp_comm_reduce(self.handle, stream, C.c_void_p(buf.ptr), buf.count, buf.code, int(root)),
                   "tgp_comm_reduce")

    def all_reduce(self, buf, stream: int = MAIN):
        _ffi.check(_ffi.lib().tgp_comm_all_reduce(self.handle, stream, C.c_void_p(buf.ptr), buf.count, buf.code, 0),
                   "tgp_comm_all_reduce")

    def agree_min(self, value: float) -> float:
        lib = _ffi.lib()
        v = np.array([value], dtype=np.float64)
        _ffi.check(lib.tgp_stream_h2d(self.ctx.handle, MAIN, C.c_void_p(self._scratch), _ffi.ptr(v), 8), "tgp_stream_h2d")
        _ffi.check(lib.tgp_comm_all_reduce(self.handle, MAIN, C.c_void_p(self._scratch), 1, _ffi.F64, 1),
                   "tgp_comm_all_reduce")
        _ffi.check(lib.tgp_stream_d2h(self.ctx.handle, MAIN, _ffi.ptr(v), C.c_void_p(self._scratch), 8), "tgp_stream_d2h")
        return float(v[0])

    def close(self):
        if getattr(self, "handle", None):
            _ffi.lib().tgp_comm_destroy(self.handle)
            self.handle = None
            self.ctx.free(self._scratch)

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


class HostStagedComm:
    """TEST transport: device buffers through host memory and a CPU (``gloo``) process group -- several ranks on the
    one GPU of a test box exercise the receiver side of the HIP path, which RCCL cannot do (one rank per device).
    Every call completes on the host before it returns."""

    def __init__(self, ctx: _ffi.Ctx, dist, group=None):
        import torch

        self.ctx, self.dist, self.group, self.torch = ctx, dist, group, torch
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def _global(self, r: int) -> int:
        return r if self.group is None else self.dist.get_global_rank(self.group, r)

    def _down(self, buf, stream):
        host = np.empty(buf.count, dtype=buf.dtype)
        _ffi.check(_ffi.lib().tgp_stream_d2h(self.ctx.handle, stream, _ffi.ptr(host), C.c_void_p(buf.ptr), host.nbytes),
                   "tgp_stream_d2h")
        return host

    def _up(self, buf, host, stream):
        _ffi.check(_ffi.lib().tgp_stream_h2d(self.ctx.handle, stream, C.c_void_p(buf.ptr), _ffi.ptr(host), host.nbytes),
                   "tgp_stream_h2d")

    def marker(self, stream: int = MAIN):
        _ffi.check(_ffi.lib().tgp_stream_sync(self.ctx.handle, stream), "tgp_stream_sync")
        return _Complete()

    def broadcast(self, buf, root: int, stream: int = MAIN):
        host = self._down(buf, stream)
        t = self.torch.from_numpy(host)
        self.dist.broadcast(t, src=self._global(root), group=self.group)
        if self.rank != root:
            self._up(buf, host, stream)
        return _Complete()

    def reduce(self, buf, root: int, stream: int = MAIN):
        host = self._down(buf, stream)
        t = self.torch.from_numpy(host)
        self.dist.reduce(t, dst=self._global(root), op=self.dist.ReduceOp.SUM, group=self.group)
        if self.rank == root:
            self._up(buf, host, stream)

    def all_reduce(self, buf, stream: int = MAIN):
        host = self._down(buf, stream)
        t = self.torch.from_numpy(host)
        self.dist.all_reduce(t, group=self.group)
        self._up(buf, host, stream)

    def agree_min(self, value: float) -> float:
        t = self.torch.tensor([value], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return float(t.item())

    def close(self):
        pass


class _TorchWork:
    def __init__(self, work):
        self.work = work

    def wait(self, stream: int = MAIN):
        if self.work is not None:
            self.work.wait()


class TorchComm:
    """``torch.distributed`` on torch tensors: the CPU stand-in of the per-rank operations under ``gloo``
    (tests/_numpy_blockops.py).  Streams mean nothing here."""

    def __init__(self, dist=None, group=None):
        if dist is None:
            import torch.distributed as dist
        import torch

        self.dist, self.group, self.torch = dist, group, torch
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def _global(self, r: int) -> int:
        return r if self.group is None else self.dist.get_global_rank(self.group, r)

    def marker(self, stream: int = MAIN):
        return _Complete()

    def broadcast(self, buf, root: int, stream: int = MAIN):
        return _TorchWork(self.dist.broadcast(buf, src=self._global(root), group=self.group, async_op=True))

    def reduce(self, buf, root: int, stream: int = MAIN):
        self.dist.reduce(buf, dst=self._global(root), op=self.dist.ReduceOp.SUM, group=self.group)

    def all_reduce(self, buf, stream: int = MAIN):
        self.dist.all_reduce(buf, group=self.group)

    def agree_min(self, value: float) -> float:
        t = self.torch.tensor([value], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return float(t.item())

    def close(self):
        pass
