"""Launches potf2 (tgp_potrf on a 128x128 SPD tile) / small potrf repeatedly; run under
rocprofv3 --kernel-trace --stats to read uncontended kernel durations."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tinygp_amd import _ffi  # noqa: E402

ctx = _ffi.default_ctx()
lib = _ffi.lib()
for n in (128, 512):
    rng = np.random.default_rng(0)
    B = rng.normal(size=(n, n))
    K = B @ B.T + n * np.eye(n)
    host = np.asfortranarray(K).ravel(order="K")
    d = ctx.malloc(host.nbytes)
    info = C.c_int32()
    for _ in range(50):
        lib.tgp_memcpy_h2d(ctx.handle, C.c_void_p(d), host.ctypes.data_as(C.c_void_p), host.nbytes)
        _ffi.check(lib.tgp_potrf(ctx.handle, _ffi.F64, n, C.c_void_p(d), n, C.byref(info)), "potrf")
    ctx.free(d)
print("done")
