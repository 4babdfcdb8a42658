import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _gpu_available() -> bool:
    try:
        import torch

        return bool(torch.cuda.is_available())
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not silently skip: only skip gpu
    # tests when they were not explicitly selected.
    selected = "gpu" in (config.getoption("-m") or "")
    if selected or _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (run with -m gpu on MI355X)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built_artefacts():
    """The .so files are git-ignored: on a fresh checkout build them once (hipcc cross-compiles
    gfx950 without a GPU; ~2 minutes) -- the same thing ``__graft_entry__.build()`` does."""
    import subprocess

    if not (ROOT / "tinygp_amd" / "lib" / "libtgp_hip.so").exists():
        subprocess.run(["make", "-C", str(ROOT / "tinygp_amd" / "csrc")], check=True, capture_output=True)
    if not (ROOT / "oracle" / "_build" / "libref_c.so").exists():
        subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"


def ulp_diff(a, b):
    """max |a-b| in units of the last place of b's dtype."""
    a, b = np.asarray(a), np.asarray(b)
    spacing = np.spacing(np.maximum(np.abs(a), np.abs(b)).astype(b.dtype))
    spacing = np.where(spacing == 0, np.finfo(b.dtype).tiny, spacing)
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)) / spacing)) if a.size else 0.0
