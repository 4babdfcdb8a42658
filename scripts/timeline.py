"""Dump the kernel timeline of the last bench step from a rocprofv3 rocpd database.

usage: timeline.py <db> <out.csv> [n_last_rows]
Prints the schema of the `kernels` view first (column names differ between rocprofv3 builds).
"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
out = sys.argv[2]
nlast = int(sys.argv[3]) if len(sys.argv) > 3 else 2500
names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
print("objects:", [n for n in names if "kernel" in n.lower() or "dispatch" in n.lower()])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print("kernels columns:", cols)
want = [x for x in ("start", "end", "name", "queue_id", "stream_id", "stream", "queue", "grid_x",
                    "grid_size_x", "workgroup_x", "workgroup_size_x", "lds_size", "lds_block_size")
        if x in cols]
rows = list(c.execute("select %s from kernels order by start" % ",".join(want)))
rows = rows[-nlast:]
t0 = rows[0][0]
with open(out, "w") as f:
    f.write(",".join(want) + "\n")
    for r in rows:
        r = list(r)
        r[0] = (r[0] - t0) / 1e3
        r[1] = (r[1] - t0) / 1e3
        r[2] = r[2].replace("void tgp::(anonymous namespace)::", "").replace("void (anonymous namespace)::", "").split("(")[0][:48].replace(", ", "_")
        f.write(",".join(str(x) for x in r) + "\n")
print("wrote", len(rows), "rows")
