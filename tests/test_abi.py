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
    assert lib.tgp_abi_version() == _ffi.ABI_VERSION == 4


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
