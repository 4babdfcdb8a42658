"""potf2's LDS hand-offs are ordered by barriers, not by timing -- checked WITHOUT a GPU.

``tests/potf2_lds_check.cpp`` #includes the device text ``tinygp_amd/csrc/potf2_body.inc`` unchanged, with a
tracking scalar in place of double / float, runs its 512 threads on the host and reports every LDS element that
one wave writes while another wave reads or writes it inside one ``__syncthreads()`` phase.  Round 2 shipped
such a pair (every eliminating wave read the diagonal block that wave 0 overwrote ~5 k cycles later,
``profiles/r02_u_potf2_function_form.txt``); the negative case below is exactly that order and must be reported,
so a future edit (or a compiler-motivated restructuring) cannot re-open it silently."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "tests" / "potf2_lds_check.cpp"


def _build_and_run(tmp_path, *defs):
    exe = tmp_path / ("chk_" + "_".join(d.replace("=", "") for d in defs).replace("-D", ""))
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wno-unknown-pragmas", *defs, str(SRC), "-o", str(exe)], check=True,
                   capture_output=True)
    return subprocess.run([str(exe)], capture_output=True, text=True)


@pytest.mark.parametrize("fold", [0, 1])
@pytest.mark.parametrize("is_float", [0, 1])
def test_every_lds_hand_off_between_waves_is_barrier_ordered(tmp_path, fold, is_float):
    r = _build_and_run(tmp_path, f"-DCHK_FOLD={fold}", f"-DCHK_FLOAT={is_float}")
    assert r.returncode == 0, r.stdout[-2000:]
    assert "conflicts=0" in r.stdout
    # the replay really ran the whole body: 8 column blocks x 2 barriers + tile staging (+ 5 for the fold)
    assert f"phases={22 if fold else 17}" in r.stdout


def test_the_round2_write_back_order_is_reported(tmp_path):
    r = _build_and_run(tmp_path, "-DCHK_FOLD=1", "-DTGP_POTF2_UNORDERED_WRITEBACK")
    assert r.returncode == 1
    # the diagonal block of steps with more than one eliminating wave: read by waves 1 / 2, written by wave 0
    assert "block (0, 0)" in r.stdout and "written by wave(s) 0, read by wave(s) 0 1 2" in r.stdout
    n = int(r.stdout.rsplit("conflicts=", 1)[1])
    assert n == 4 * 256  # column blocks 0..3 have 3, 2, 2, 2 eliminating waves
