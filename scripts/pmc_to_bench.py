"""profiles/pmc_traffic.json from the two PMC passes of the evidence batch: fabric bytes per launch of the trailing-update
kernel at workload c2 = (FETCH_SIZE x 2 + WRITE_SIZE) KB summed over its launches / launches (MI355X_MICROARCH.md, HBM
section: separate --pmc passes, KB units, FETCH_SIZE doubled for wide coalesced reads), stamped with the hash of
gemm.hip and the schedule options it was collected with -- bench.py reports `roofline.traffic` only while both match.

The passes run with chain_polls=0 (rocprofv3 --pmc runs kernels one at a time, and a poller that waits for a chain
launch behind it would time out): the same trailing-update launches, the forward steps behind the chain launch instead of
beside it.  The stamp carries the DEFAULT options plus `collected_with`.

usage (on the GPU box): pmc_to_bench.py <FETCH_SIZE db> <WRITE_SIZE db> <evidence file name>"""
import json
import sqlite3
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from tinygp_amd import _ffi  # noqa: E402


def total(db, counter):
    c = sqlite3.connect(db)
    tot, cnt = 0.0, 0
    for k, v in c.execute("select kernel_name,value from counters_collection where counter_name=?", (counter,)):
        if "gemm_nt_kernel<double, 0>" in k or "gemm_nt_kernel<double, 2>" in k:
            tot += v
            cnt += 1
    return tot * 1e3, cnt  # KB -> bytes


fetch, nf = total(sys.argv[1], "FETCH_SIZE")
write, nw = total(sys.argv[2], "WRITE_SIZE")
assert nf == nw and nf > 0, (nf, nw)
out = {"file": sys.argv[3], "bytes_per_launch": (2 * fetch + write) / nf, "launches": nf,
       "fetch_bytes_x2": 2 * fetch, "write_bytes": write, "gemm_hip_sha256_16": bench.gemm_source_hash(),
       "options": {k: int(v) for k, v in _ffi.Ctx().schedule_options().items()},
       "collected_with": {"chain_polls": 0}}
(ROOT / "gpurun_out" / "pmc_traffic.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out))
