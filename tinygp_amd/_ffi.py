"""ctypes binding of ``libtgp_hip.so`` (the C ABI declared in ``include/tgp_hip.h``).

This is the only place the Python host touches native code.  There is NO CPU
fallback: if the shared library is missing or no MI355X is visible, every
entry point raises.  (The NumPy oracle under ``oracle/`` is test infrastructure
and is never imported from here.)
"""

from __future__ import annotations

import ctypes as C
import os
import threading
from pathlib import Path

import numpy as np

__all__ = [
    "lib", "library_path", "default_ctx", "Ctx", "KOp", "check", "TgpError",
    "dtype_code", "F32", "F64", "TILE",
]

F32, F64 = 0, 1
TILE = 128
E_ARG = -1

_LIB_NAME = "libtgp_hip.so"
_lock = threading.RLock()
_lib = None
_default_ctx = None


class TgpError(RuntimeError):
    pass


class KOp(C.Structure):
    """``tgp_kop`` -- one postfix op of a kernel program."""

    _fields_ = [("op", C.c_int32), ("metric", C.c_int32), ("p0", C.c_double), ("p1", C.c_double)]


def library_path() -> Path:
    env = os.environ.get("TGP_HIP_LIBRARY")
    if env:
        return Path(env)
    return Path(__file__).resolve().parent / "lib" / _LIB_NAME


_i32, _i64, _dbl, _vp, _int = C.c_int32, C.c_int64, C.c_double, C.c_void_p, C.c_int
_pi32, _pi64, _pdbl, _pvp = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_void_p)
_pkop = C.POINTER(KOp)

# name -> argtypes; every symbol include/tgp_hip.h declares (checked by tests/test_abi.py)
SIGNATURES = {
    "tgp_abi_version": [],
    "tgp_last_error": [],
    "tgp_ctx_create": [_int, _vp, _pvp],
    "tgp_ctx_destroy": [_vp],
    "tgp_ctx_sync": [_vp],
    "tgp_ctx_set_option": [_vp, C.c_char_p, _i64, _pi64],
    "tgp_ctx_get_option": [_vp, C.c_char_p, _pi64],
    "tgp_ctx_device_info": [_vp, C.c_char_p, _int, _pi32, _pi64, _pi32],
    "tgp_malloc": [_vp, C.c_size_t, _pvp],
    "tgp_free": [_vp, _vp],
    "tgp_memcpy_h2d": [_vp, _vp, _vp, C.c_size_t],
    "tgp_memcpy_d2h": [_vp, _vp, _vp, C.c_size_t],
    "tgp_memset": [_vp, _vp, _int, C.c_size_t],
    "tgp_kmat": [_vp, _int, _pkop, _int, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _int],
    "tgp_kdiag": [_vp, _int, _pkop, _int, _i64, _i32, _vp, _vp],
    "tgp_kmat_gemv": [_vp, _int, _pkop, _int, _i64, _i64, _i32, _vp, _vp, _vp, _vp],
    "tgp_kmat_gemv_multi": [_vp, _int, _pkop, _int, _i64, _i64, _i32, _vp, _vp, _vp, _i64, _vp],
    "tgp_potrf": [_vp, _int, _i64, _vp, _i64, _pi32],
    "tgp_trsv": [_vp, _int, _i64, _vp, _i64, _int, _vp],
    "tgp_trsm_right_lt": [_vp, _int, _i64, _i64, _vp, _i64, _vp, _i64],
    "tgp_gemm_nt": [_vp, _int, _i64, _i64, _i64, _dbl, _vp, _i64, _vp, _i64, _dbl, _vp, _i64, _int],
    "tgp_gemv_sub": [_vp, _int, _i64, _i64, _vp, _i64, _vp, _vp],
    "tgp_sum_log_diag": [_vp, _int, _i64, _vp, _i64, _pdbl],
    "tgp_sum_squares": [_vp, _int, _i64, _vp, _pdbl],
    "tgp_ubench_mfma": [_vp, _int, _pdbl],
    "tgp_ubench": [_vp, _int, _int, _pdbl, _pdbl],
    "tgp_solver_create": [_vp, _int, _i64, _i32, _vp, _vp, _pvp],
    "tgp_solver_destroy": [_vp],
    "tgp_solver_factor": [_vp, _pkop, _int, _vp, _pi32],
    "tgp_solver_factor_logprob": [_vp, _pkop, _int, _vp, _vp, _pi32, _pdbl],
    "tgp_solver_set_noise": [_vp, _vp],
    "tgp_solver_normalization": [_vp, _pdbl],
    "tgp_solver_solve_tri": [_vp, _int, _i64, _vp, _vp],
    "tgp_solver_dot_tri": [_vp, _i64, _vp, _vp],
    "tgp_solver_set_resid": [_vp, _vp],
    "tgp_solver_logprob": [_vp, _vp, _pdbl],
    "tgp_solver_alpha": [_vp, _vp, _vp, _pdbl],
    "tgp_solver_grad": [_vp, _vp, _pdbl, _pdbl, _vp, _vp, _pdbl],
    "tgp_solver_cond_mean": [_vp, _pkop, _int, _i64, _vp, _vp, _vp],
    "tgp_solver_condition_cov": [_vp, _pkop, _int, _i64, _vp, _vp, _int, _vp],
    "tgp_solver_covariance": [_vp, _vp],
    "tgp_solver_variance": [_vp, _vp],
    "tgp_solver_get_factor": [_vp, _vp],
    "tgp_solver_device_factor": [_vp, _pvp, _pi64],
    "tgp_solver_timings": [_vp, _pdbl, _int],
    "tgp_trace_factor": [_i64, C.c_char_p, _i32, _pi64, _i64, _pi64],
    "tgp_chain_stamps": [_vp, _pi64, _i64, _pi64],
    "tgp_chain_task": [_i64, _i64, _i64, _i64, _i64, _pi32, _pi64],
    "tgp_chain_tasks": [_i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _pi32, _i64, _pi64],
    "tgp_tile_order": [_i64, _i64, _i32, _i32, _i64, _pi32, _pi32, _pi64],
    "tgp_dist_slot_elems": [_i64, _i64],
    "tgp_dist_create": [_vp, _int, _i64, _i32, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _pvp],
    "tgp_dist_destroy": [_vp],
    "tgp_dist_stream": [_vp, _int, _pvp],
    "tgp_dist_assemble": [_vp, _pkop, _int],
    "tgp_dist_begin": [_vp, _vp],
    "tgp_dist_first_panel": [_vp],
    "tgp_dist_fwd_step": [_vp, _i64],
    "tgp_dist_panel_chunk": [_vp, _i64, _i64, _i64],
    "tgp_dist_slot_ready": [_vp, _i64],
    "tgp_dist_lookahead": [_vp, _i64],
    "tgp_dist_arrived": [_vp, _i64],
    "tgp_dist_pre_update": [_vp, _i64],
    "tgp_dist_rest": [_vp, _i64],
    "tgp_dist_end": [_vp, _pi32, _pdbl, _pdbl],
    "tgp_dist_bwd_step": [_vp, _i64],
    "tgp_dist_cond_mean_partial": [_vp, _pkop, _int, _i64, _vp, _vp],
    "tgp_dist_get_column": [_vp, _i64, _vp],
    "tgp_dist_fwd_block": [_vp, _i64, _i64, _vp, _vp, _vp],
    "tgp_dist_bwd_block": [_vp, _i64, _vp],
    "tgp_dist_trmv_partial": [_vp, _vp, _vp],
    "tgp_dist_cross_cov": [_vp, _pkop, _int, _i64, _vp, _i64, _vp],
    "tgp_dist_colsumsq_owned": [_vp, _i64, _vp, _vp],
    "tgp_dist_gram_owned": [_vp, _i64, _vp, _vp],
    "tgp_dist_gram_pair_owned": [_vp, _i64, _vp, _i64, _vp, _vp],
    "tgp_dist_load_matrix": [_vp, _vp],
    "tgp_dist_abort": [_vp],
    "tgp_dist_fwd_partial": [_vp, _i64, _i64, _vp, _vp, _i64],
    "tgp_dist_fwd_solve_left": [_vp, _i64, _i64, _vp, _vp, _vp, _vp],
    "tgp_dist_bwd_block_multi": [_vp, _i64, _i64, _vp, _vp],
    "tgp_dist_bwd_update_multi": [_vp, _i64, _i64, _vp, _vp, _i64],
    "tgp_dist_gather_owned": [_vp, _i64, _vp, _vp],
    "tgp_dist_identity_cols": [_vp, _i64, _i64, _vp],
    "tgp_dist_grad_begin": [_vp, _pkop, _int],
    "tgp_dist_grad_chunk": [_vp, _i64, _i64, _vp, _i32],
    "tgp_dist_grad_end": [_vp, _pdbl, _pdbl, _vp],
    "tgp_comm_unique_id": [_vp, _pi32],
    "tgp_comm_create": [_vp, _i32, _i32, _vp, _pvp],
    "tgp_comm_destroy": [_vp],
    "tgp_comm_info": [_vp, _pi32, _pi32],
    "tgp_comm_broadcast": [_vp, _int, _vp, _i64, _int, _i32],
    "tgp_comm_reduce": [_vp, _int, _vp, _i64, _int, _i32],
    "tgp_comm_all_reduce": [_vp, _int, _vp, _i64, _int, _i32],
    "tgp_comm_record": [_vp, _int, _pi64],
    "tgp_comm_wait": [_vp, _int, _i64],
    "tgp_stream_h2d": [_vp, _int, _vp, _vp, _i64],
    "tgp_stream_d2h": [_vp, _int, _vp, _vp, _i64],
    "tgp_stream_d2d": [_vp, _int, _vp, _vp, _i64],
    "tgp_stream_memset": [_vp, _int, _vp, _int, _i64],
    "tgp_stream_sync": [_vp, _int],
}


ABI_VERSION = 6  # TGP_ABI_VERSION of include/tgp_hip.h


def _mapped_hip_runtimes() -> list[str]:
    """Distinct ``libamdhip64`` images mapped into this process (Linux: /proc/self/maps)."""
    seen = []
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.split(None, 5)[-1].strip() if line.count("/") else ""
                if "libamdhip64" in os.path.basename(path):
                    real = os.path.realpath(path)
                    if real not in seen:
                        seen.append(real)
    except OSError:  # pragma: no cover - not Linux
        pass
    return seen


def _torch_hip_runtime() -> Path | None:
    """PyTorch-ROCm wheels bundle their own HIP runtime under torch/lib; located WITHOUT importing torch."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return None
    if spec is None or not spec.origin:
        return None
    cand = Path(spec.origin).resolve().parent / "lib" / "libamdhip64.so"
    return cand if cand.exists() else None


def share_hip_runtime() -> str | None:
    """Make sure ONE HIP runtime serves this process, whoever comes first.

    ``libtgp_hip.so`` needs ``libamdhip64.so.7`` (no RUNPATH); PyTorch-ROCm loads the copy bundled in its wheel through
    ``$ORIGIN``.  Two copies mean two HSA runtimes, and the second one sees no device
    (``tgp_ctx_create: no HIP device visible`` right after ``init_process_group``, GPUTEST_r04).  Loaded after torch
    the SONAME already resolves to torch's copy; loaded BEFORE torch -- the single-GPU path, then
    ``DistributedDirectSolver`` -- the system copy used to win and torch mapped a second one.  So: when no runtime is
    mapped yet and a torch wheel with a bundled runtime is installed, that copy is loaded first (RTLD_GLOBAL, by the
    very path torch will ask for: the loader then re-uses the image by device/inode).  ``TGP_HIP_RUNTIME=system`` keeps
    the loader's default (ROCm's ld.so.conf entry), ``TGP_HIP_RUNTIME=/path/libamdhip64.so`` names one.
    Returns the path preloaded, or None."""
    if _mapped_hip_runtimes():
        return None
    choice = os.environ.get("TGP_HIP_RUNTIME", "auto")
    if choice == "system":
        return None
    cand = _torch_hip_runtime() if choice in ("auto", "torch") else Path(choice)
    if cand is None:
        if choice == "torch":
            raise TgpError("TGP_HIP_RUNTIME=torch, but no torch wheel with a bundled libamdhip64.so is installed")
        return None
    if not cand.exists():
        raise TgpError(f"TGP_HIP_RUNTIME={choice}: no such file")
    C.CDLL(str(cand), mode=C.RTLD_GLOBAL)
    return str(cand)


def assert_single_hip_runtime(where: str = "") -> None:
    """Two HIP runtimes in one process = a dead device for whichever came second: say so instead of failing later."""
    got = _mapped_hip_runtimes()
    if len(got) > 1:
        raise TgpError(
            f"{where + ': ' if where else ''}two HIP runtimes are mapped into this process ({', '.join(got)}); the "
            "second one cannot see the GPU. Import tinygp_amd (or call tinygp_amd._ffi.lib()) before anything that loads "
            "ROCm's system libamdhip64, or set TGP_HIP_RUNTIME to the copy the rest of the process uses.")


def load_library(path: Path | None = None) -> C.CDLL:
    """dlopen the library and attach signatures.  Needs no GPU (used by the ABI test)."""
    path = library_path() if path is None else Path(path)
    if not path.exists():
        raise TgpError(
            f"{path} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C tinygp_amd/csrc`). "
            "tinygp_amd has no CPU fallback."
        )
    share_hip_runtime()
    lib_ = C.CDLL(str(path))
    assert_single_hip_runtime(str(path))
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib_, name)  # AttributeError = ABI drift, deliberately loud
        fn.argtypes = argtypes
        fn.restype = (C.c_char_p if name == "tgp_last_error"
                      else C.c_int64 if name == "tgp_dist_slot_elems" else C.c_int)
    got = lib_.tgp_abi_version()
    if got != ABI_VERSION:  # a stale libtgp_hip.so would mis-parse re-ordered arguments silently
        raise TgpError(f"{path} has ABI version {got}, this binding needs {ABI_VERSION}: rebuild it "
                       "(`make -C tinygp_amd/csrc`)")
    return lib_


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                _lib = load_library()
    return _lib


def check(status: int, what: str = "") -> int:
    """Raise on a negative status; positive values (potrf info) pass through."""
    if status >= 0:
        return status
    msg = lib().tgp_last_error()
    msg = msg.decode() if msg else "unknown error"
    if status == E_ARG:
        raise ValueError(f"{what}: {msg}" if what else msg)
    raise TgpError(f"{what}: {msg} (status {status})" if what else f"{msg} (status {status})")


def dtype_code(dtype) -> int:
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return F64
    if dtype == np.float32:
        return F32
    raise ValueError(f"tinygp_amd computes in float32 or float64, got {dtype}")


class Ctx:
    """A ``tgp_ctx``: one HIP device + one stream."""

    def __init__(self, device: int | None = None, stream: int | None = None):
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device
        h = C.c_void_p()
        lib()
        assert_single_hip_runtime("tgp_ctx_create")
        check(lib().tgp_ctx_create(device, C.c_void_p(stream or 0), C.byref(h)), "tgp_ctx_create")
        self.handle = h
        # tuning overrides, e.g. TGP_HIP_OPTIONS="nb_outer=512,lookahead=0"
        for item in filter(None, os.environ.get("TGP_HIP_OPTIONS", "").split(",")):
            key, _, value = item.partition("=")
            self.set_option(key.strip(), int(value))

    def sync(self):
        check(lib().tgp_ctx_sync(self.handle), "tgp_ctx_sync")

    def set_option(self, key: str, value: int) -> int:
        old = C.c_int64()
        check(lib().tgp_ctx_set_option(self.handle, key.encode(), int(value), C.byref(old)),
              "tgp_ctx_set_option")
        return old.value

    # the options that shape the factorisation's schedule (tgp_trace_factor takes the same names)
    SCHEDULE_OPTIONS = ("nb_outer", "lookahead", "first_split", "first_small_tiles", "nb_wide_rows", "fused_step", "chain_kernel", "chain_fast_update", "chain_batch", "chain_batch_lag", "chain_batch_rowlag", "chain_batch_minrows", "tile_band", "chain_full_rows", "chain_lds_pad", "chain_depth2", "chain_merged", "chain_sub_panel", "chain_sub_min_rows", "chain_sub_role", "chain_pre_wait", "chain_polls", "chain_fwd_tasks", "chain_reduce",
                        "gate_split", "chain_reserve", "reserve_max_tiles", "sub_panel", "sub_panel_min_rows",
                        "nb_first", "split_tail", "solve_on_update")

    def get_option(self, key: str) -> int:
        v = C.c_int64()
        check(lib().tgp_ctx_get_option(self.handle, key.encode(), C.byref(v)), "tgp_ctx_get_option")
        return v.value

    def schedule_options(self) -> dict:
        return {k: self.get_option(k) for k in self.SCHEDULE_OPTIONS}

    def device_info(self) -> dict:
        name = C.create_string_buffer(256)
        cus, mem, clk = C.c_int32(), C.c_int64(), C.c_int32()
        check(lib().tgp_ctx_device_info(self.handle, name, 256, C.byref(cus), C.byref(mem),
                                        C.byref(clk)), "tgp_ctx_device_info")
        return {"name": name.value.decode(), "cus": cus.value, "mem_bytes": mem.value,
                "clock_khz": clk.value}

    # raw buffers ------------------------------------------------------------
    def malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        check(lib().tgp_malloc(self.handle, max(int(nbytes), 8), C.byref(p)), "tgp_malloc")
        return p.value

    def free(self, ptr: int):
        if ptr:
            lib().tgp_free(self.handle, C.c_void_p(ptr))

    def upload(self, arr: np.ndarray) -> int:
        arr = np.ascontiguousarray(arr)
        p = self.malloc(arr.nbytes)
        if arr.nbytes:
            check(lib().tgp_memcpy_h2d(self.handle, C.c_void_p(p), arr.ctypes.data_as(C.c_void_p),
                                       arr.nbytes), "tgp_memcpy_h2d")
        return p

    def download(self, ptr: int, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        if out.nbytes:
            check(lib().tgp_memcpy_d2h(self.handle, out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr),
                                       out.nbytes), "tgp_memcpy_d2h")
        return out

    def close(self):
        if getattr(self, "handle", None):
            lib().tgp_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def default_ctx() -> Ctx:
    global _default_ctx
    if _default_ctx is None:
        with _lock:
            if _default_ctx is None:
                _default_ctx = Ctx()
    return _default_ctx


def as_kprog(ops):
    """list of (op, metric, p0, p1) -> (ctypes array, n)."""
    arr = (KOp * max(len(ops), 1))()
    for i, (op, metric, p0, p1) in enumerate(ops):
        arr[i] = KOp(int(op), int(metric), float(p0), float(p1))
    return arr, len(ops)


def ptr(a: np.ndarray | None):
    return C.c_void_p(0) if a is None else a.ctypes.data_as(C.c_void_p)
