"""Sum one PMC counter per kernel from a rocprofv3 (rocpd) database.

usage: pmc_summary.py <db> <COUNTER>
FETCH_SIZE / WRITE_SIZE are reported in KB; read the HBM section of MI355X_MICROARCH.md before
interpreting FETCH_SIZE (x2 for wide coalesced loads)."""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
name = sys.argv[2]
tot = defaultdict(float)
cnt = defaultdict(int)
for k, v in c.execute("select kernel_name,value from counters_collection where counter_name=?", (name,)):
    k = k.replace("void tgp::(anonymous namespace)::", "").split("(")[0]
    tot[k] += v
    cnt[k] += 1
print(f"# {name}: kernel | launches | total (GB if KB counter) | per launch (MB)")
for k in sorted(tot, key=lambda k: -tot[k]):
    print(f"{k} | {cnt[k]} | {tot[k]/1e6:.3f} | {tot[k]/1e3/cnt[k]:.2f}")
