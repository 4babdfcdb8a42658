"""Times tgp_gemm_nt on resident buffers: TFLOP/s vs K (prologue/epilogue amortisation)."""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tinygp_amd import _ffi  # noqa: E402

ctx = _ffi.default_ctx()
lib = _ffi.lib()
dt = np.float64 if len(sys.argv) < 2 or sys.argv[1] == "f64" else np.float32
es = np.dtype(dt).itemsize
code = _ffi.dtype_code(dt)
M = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
ONE = len(sys.argv) > 3 and sys.argv[3] == "one"  # one shape (for counter passes)
for mode, lower in ((((-1.0, 1.0), 0),) if ONE else (((-1.0, 1.0), 0), ((-1.0, 1.0), 1))):
  for K in ((1024,) if ONE else (16, 128, 512, 1024, 2048)):
    if True:
        rng = np.random.default_rng(0)
        dA = ctx.upload(rng.normal(size=M * K).astype(dt))
        dC = ctx.upload(rng.normal(size=M * M).astype(dt))
        def run(reps):
            for _ in range(reps):
                _ffi.check(lib.tgp_gemm_nt(ctx.handle, code, M, M, K, mode[0], C.c_void_p(dA), M, C.c_void_p(dA), M,
                                           mode[1], C.c_void_p(dC), M, lower), "gemm")
            ctx.sync()
        run(2)
        t0 = time.perf_counter(); reps = 8; run(reps); dtm = (time.perf_counter() - t0) / reps
        tiles = (M // 128) * (M // 128 + 1) // 2 if lower else (M // 128) ** 2
        flops = tiles * 128 * 128 * K * 2.0
        print(f"mode={mode} {np.dtype(dt).name} lower={lower} M={M} K={K:5d}: {dtm*1e3:8.3f} ms  {flops/dtm/1e12:7.2f} TFLOP/s", flush=True)
        ctx.free(dA); ctx.free(dC)
