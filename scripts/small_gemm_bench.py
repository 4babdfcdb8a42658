"""The shapes the factorisation gives the 64x64-tile GEMM (in-panel rank-128 updates, gates), timed alone on resident
buffers, next to the 128x128-tile kernel on the same shape: TFLOP/s per launch.  usage: small_gemm_bench.py [reps]"""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tinygp_amd import _ffi  # noqa: E402

ctx = _ffi.default_ctx()
lib = _ffi.lib()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dt, code = np.float64, _ffi.dtype_code(np.float64)
rng = np.random.default_rng(0)
shapes = [(16256, 896, 128), (12288, 896, 128), (8192, 896, 128), (8192, 384, 128), (4096, 896, 128),
          (12288, 1024, 384), (12288, 1024, 640), (12288, 1024, 1024), (6144, 1024, 384), (6144, 1024, 1024),
          (12288, 512, 512), (6144, 512, 512)]
for m, n, k in shapes:
    dA = ctx.upload(rng.normal(size=m * k).astype(dt))
    dC = ctx.upload(rng.normal(size=m * n).astype(dt))
    line = f"m={m:6d} n={n:5d} k={k:5d} lower:"
    for role, name in ((4, "64x64"), (0, "128x128")):
        ctx.set_option("gemm_role", role)

        def run(r):
            for _ in range(r):
                _ffi.check(lib.tgp_gemm_nt(ctx.handle, code, m, n, k, -1.0, C.c_void_p(dA), m, C.c_void_p(dA), m, 1.0,
                                           C.c_void_p(dC), m, 1), "gemm")
            ctx.sync()
        run(3)
        t0 = time.perf_counter()
        run(reps)
        t = (time.perf_counter() - t0) / reps
        entries = n * m - n * (n - 1) / 2
        line += f"  {name}: {t * 1e6:8.1f} us {2 * entries * k / t / 1e12:6.1f} TF"
    ctx.set_option("gemm_role", 1)
    print(line, flush=True)
    ctx.free(dA), ctx.free(dC)
