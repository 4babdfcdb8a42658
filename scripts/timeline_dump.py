"""Every kernel of the last evaluation in a timeline CSV written by timeline.py: start (us from the step's first
kernel), duration, queue, grid, name -- the forward-substitution steps and one-wave pollers collapsed per run.

usage: timeline_dump.py <timeline.csv>"""
import sys


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
starts = [i for i in idx if i == 0 or not rows[i - 1]["name"].startswith(("kmat_kernel", "kmat_fast_kernel"))]
step = rows[starts[-1]:]
t0 = step[0]["start"]
print("step %.0f us, %d kernels" % (step[-1]["end"] - t0, len(step)))
small = ("trsv_", "chain_poll")
run = None
for r in step:
    if r["name"].startswith(small):
        if run is None:
            run = [r["start"], r["end"], 1, r["queue"]]
        else:
            run[1], run[2] = r["end"], run[2] + 1
        continue
    if run is not None:
        print(f"  {run[0]-t0:9.1f} +{run[1]-run[0]:8.1f}  q{run[3]}  [{run[2]} forward-step / poll kernels]")
        run = None
    print(f"  {r['start']-t0:9.1f} +{r['end']-r['start']:8.1f}  q{r['queue']} grid {r['grid']:7d}  {r['name'][:60]}")
if run is not None:
    print(f"  {run[0]-t0:9.1f} +{run[1]-run[0]:8.1f}  q{run[3]}  [{run[2]} forward-step / poll kernels]")
