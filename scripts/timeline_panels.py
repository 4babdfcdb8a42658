"""Per-panel view of a timeline CSV written by timeline.py (last bench step): `gate` = the last
share of the look-ahead block-column update (64x64 tiles) before the chain starts, `rest` =
the big trailing update beside the chain, the length of the potf2/trsm chain of the next
panel, and which of the two finished last."""
import sys
from collections import defaultdict


def read(path):
    rows = []
    with open(path) as f:
        f.readline()
        for line in f:
            a = line.rstrip("\n").split(",", 2)
            b = a[2].rsplit(",", 7)
            rows.append({"start": float(a[0]), "end": float(a[1]), "name": b[0].replace(", ", "_"),
                         "queue": b[1], "grid": int(b[5])})
    return rows


rows = read(sys.argv[1])
idx = [i for i, r in enumerate(rows) if r["name"].startswith(("kmat_kernel", "kmat_fast_kernel"))]
# the last step starts at the last kmat launch that follows a reduction kernel (two kmat launches per step)
starts = [i for i in idx if i == 0 or not rows[i - 1]["name"].startswith(("kmat_kernel", "kmat_fast_kernel"))]
step = rows[starts[-1]:]
t0 = step[0]["start"]
print("step %.0f us, %d kernels" % (step[-1]["end"] - t0, len(step)))
dur = defaultdict(list)
for r in step:
    dur[r["name"][:30]].append(r["end"] - r["start"])
for k, v in dur.items():
    v.sort()
    print(f"  {k:30s} n={len(v):4d} sum={sum(v)/1e3:8.2f} ms  med={v[len(v)//2]:7.1f} min={v[0]:7.1f} max={v[-1]:7.1f}")
big = [r for r in step if r["name"].startswith("gemm_nt_kernel")]
mainq = big[0]["queue"] if big else None
firsts = [r for r in step if r["name"].startswith("gemm_nt_small") and r["queue"] == mainq]
chain = [r for r in step if r["name"].startswith(("potf2", "trsm", "panel_step"))]
for i, rest in enumerate(big):
    # the panel's own first potf2 runs on the main stream just before `rest`
    lo = rest["start"] - 80
    hi = (big[i + 1]["start"] - 80) if i + 1 < len(big) else 1e18
    ch = [c for c in chain if lo <= c["start"] < hi]
    fp = [f for f in firsts if lo - 1 <= f["start"] < hi]       # early + final shares issued in this window
    fin = [f for f in firsts if f["end"] <= rest["start"] + 1]  # the share that gated this chain
    gate = fin[-1] if fin else None
    cend = max(c["end"] for c in ch) if ch else 0
    cstart = min(c["start"] for c in ch) if ch else 0
    print(f"panel {i+1:2d}: gate {(gate['end']-gate['start']) if gate else 0:5.0f} | rest {rest['end']-rest['start']:6.0f} "
          f"chain {cend-cstart:6.0f} n={len(ch):2d} early+final shares {len(fp)} | "
          f"{'chain' if cend > rest['end'] else 'gemm '} by {abs(cend-rest['end']):5.0f}")

# optional: `dump <first panel> <n panels>` lists every kernel of those panels' windows (start, duration, queue)
if len(sys.argv) > 4 and sys.argv[2] == "dump" and big:
    p0, npan = int(sys.argv[3]) - 1, int(sys.argv[4])
    lo = big[min(p0, len(big) - 1)]["start"] - 100
    hi = big[min(p0 + npan, len(big) - 1)]["start"] if p0 + npan < len(big) else step[-1]["end"]
    for r in step:
        if lo <= r["start"] < hi:
            print(f"  {r['start']-t0:9.1f} +{r['end']-r['start']:7.1f}  q{r['queue']} grid {r['grid']:6d}  {r['name'][:44]}")
