"""Per-panel view of a timeline CSV written by timeline.py (last bench step):
first-part update, rest update, chain length, and which of them gated the next panel."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
idx = [i for i, r in enumerate(rows) if "kmat_kernel" in r["name"]]
step = rows[idx[-1]:]
t0 = float(step[0]["start"])
print("step %.0f us, %d kernels" % (float(step[-1]["end"]) - t0, len(step)))
dur = defaultdict(list)
for r in step:
    dur[r["name"][:26]].append(float(r["end"]) - float(r["start"]))
for k, v in dur.items():
    v.sort()
    print(f"  {k:28s} n={len(v):4d} sum={sum(v)/1e3:8.2f} ms  med={v[len(v)//2]:7.1f} min={v[0]:7.1f} max={v[-1]:7.1f}")
big = [(float(r["start"]) - t0, float(r["end"]) - t0) for r in step if r["name"].startswith("gemm_nt_kernel")]
chain = [(float(r["start"]) - t0, float(r["end"]) - t0) for r in step
         if ("potf2" in r["name"] or "trsm" in r["name"])]
prev_end = 0
for i in range(0, len(big) - 1, 2):
    fp, rest = sorted(big[i:i + 2])
    nxt = big[i + 2][0] if i + 2 < len(big) else 1e12
    ch = [c for c in chain if c[0] >= fp[1] - 1 and c[1] <= nxt + 1]
    cend = max(c[1] for c in ch) if ch else 0
    print(f"panel {i//2+1:2d}: first {fp[1]-fp[0]:5.0f} | rest {rest[1]-rest[0]:6.0f} (start+{rest[0]-fp[1]:4.0f}) "
          f"chain {cend-fp[1]:6.0f} n={len(ch):2d} | {'chain' if cend > rest[1] else 'gemm '} by {abs(cend-rest[1]):5.0f}")
