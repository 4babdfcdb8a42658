"""The C-ABI library loads (no GPU needed) and exports every symbol include/tgp_hip.h declares."""
import re
from pathlib import Path

import pytest

from tinygp_amd import _ffi

ROOT = Path(__file__).resolve().parent.parent


def header_symbols():
    text = (ROOT / "include" / "tgp_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tgp_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert header_symbols() == sorted(_ffi.SIGNATURES)


def test_library_exports_every_symbol():
    lib = _ffi.load_library()  # raises if the .so is missing or a symbol is absent
    for name in header_symbols():
        assert hasattr(lib, name), name
    assert lib.tgp_abi_version() == _ffi.ABI_VERSION == 6


def test_missing_library_is_loud(tmp_path):
    with pytest.raises(_ffi.TgpError, match="no CPU fallback"):
        _ffi.load_library(tmp_path / "libtgp_hip.so")


def test_no_gpu_is_loud():
    """Without a HIP device every compute entry point must fail, never fall back."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises((_ffi.TgpError, ValueError)):
        _ffi.Ctx(0)


def test_product_never_imports_the_oracle():
    for py in (ROOT / "tinygp_amd").rglob("*.py"):
        src = py.read_text()
        assert "import oracle" not in src and "from oracle" not in src, py


def test_default_ctx_does_not_deadlock():
    """Regression: default_ctx() took a non-reentrant lock and then called lib()."""
    import threading

    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    res = []

    def run():
        try:
            _ffi.default_ctx()
        except Exception as e:  # loud failure expected on a CPU-only box
            res.append(e)

    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(30)
    assert not t.is_alive(), "default_ctx() deadlocked"
    assert res and isinstance(res[0], _ffi.TgpError)


# --- ONE HIP runtime per process (GPUTEST_r04: lib-then-torch mapped two, the second saw no device) ---------------------

_ORDER_SNIPPET = r"""
import sys
for what in sys.argv[1].split(","):
    if what == "lib":
        from tinygp_amd import _ffi
        _ffi.lib()
    elif what == "torch":
        import torch
    elif what == "dist":
        import tinygp_amd.distributed  # the sharded path's module (imports torch lazily or not at all)
from tinygp_amd import _ffi
hip = _ffi._mapped_hip_runtimes()
hsa = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libhsa-runtime64" in l})
print("HIP", len(hip), "HSA", len(hsa), hip, hsa)
"""


def _run_order(order, env_extra=None):
    import os
    import subprocess
    import sys

    env = dict(os.environ, PYTHONPATH=str(ROOT))
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, "-c", _ORDER_SNIPPET, order], capture_output=True, text=True, env=env,
                       cwd=str(ROOT), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout.strip().splitlines()[-1]


@pytest.mark.parametrize("order", ["lib,torch", "torch,lib", "lib,dist,torch", "lib"])
def test_one_hip_runtime_whatever_the_import_order(order):
    line = _run_order(order)
    assert line.startswith("HIP 1 HSA 1 "), line


def test_two_hip_runtimes_are_reported_not_suffered():
    """With the system runtime forced first and torch's bundled copy second, creating a context says what is wrong."""
    if _ffi._torch_hip_runtime() is None:
        pytest.skip("torch has no bundled HIP runtime here")
    import os
    import subprocess
    import sys

    code = ("from tinygp_amd import _ffi; _ffi.lib(); import torch\n"
            "try:\n    _ffi.Ctx(0)\nexcept _ffi.TgpError as e:\n    print('TGPERR', e)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(ROOT), timeout=600,
                       env=dict(os.environ, PYTHONPATH=str(ROOT), TGP_HIP_RUNTIME="system"))
    assert "TGPERR" in r.stdout and "two HIP runtimes" in r.stdout, r.stdout + r.stderr[-1000:]


def test_library_has_no_runpath():
    import subprocess

    out = subprocess.run(["readelf", "-d", str(_ffi.library_path())], capture_output=True, text=True).stdout
    assert "RUNPATH" not in out and "RPATH" not in out, out
    assert "libamdhip64.so.7" in out


def test_collecting_the_suite_does_not_load_the_library():
    """pytest imports every test module at collection (deselected ones too): none of them may dlopen the library."""
    import os
    import subprocess
    import sys

    code = ("import sys, pytest\n"
            "class P:\n"
            "    def pytest_collection_finish(self, session):\n"
            "        maps = open('/proc/self/maps').read()\n"
            "        print('COLLECTED', len(session.items), 'TGP', 'libtgp_hip' in maps)\n"
            "sys.exit(pytest.main(['tests', '--collect-only', '-q', '-p', 'no:cacheprovider'], plugins=[P()]))\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(ROOT), timeout=600,
                       env=dict(os.environ, PYTHONPATH=str(ROOT)))
    line = [l for l in r.stdout.splitlines() if l.startswith("COLLECTED")]
    assert line and line[0].endswith(" TGP False"), (line, r.stdout[-500:], r.stderr[-500:])


def test_makefile_tracks_header_dependencies(tmp_path):
    """chol.o and capi.o both include chain_tasks.h; a stale one of the two = kernel and launcher disagree about the grid."""
    import os
    import subprocess

    src = ROOT / "tinygp_amd" / "csrc"
    if not (src / "chol.d").exists():
        pytest.skip("objects were not built by this Makefile (no .d files)")
    hdr = src / "chain_tasks.h"
    st = hdr.stat()
    try:
        os.utime(hdr, None)
        out = subprocess.run(["make", "-n", "-C", str(src)], capture_output=True, text=True).stdout
    finally:
        os.utime(hdr, ns=(st.st_atime_ns, st.st_mtime_ns))
    assert "chol.hip -o chol.o" in out and "capi.hip -o capi.o" in out, out
    assert "kmat.hip" not in out and "gemm.hip" not in out, out
