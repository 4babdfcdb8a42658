"""Whole-evaluation counters: every kernel of a rocprofv3 --pmc pass summed, per kernel and in total.

usage: pmc_whole.py <db> COUNTER [COUNTER ...]

What the totals mean on MI355X (256 CUs, 4 SIMDs each, 8 XCDs; see scripts/pmc_multi.py for the units):
  SQ_BUSY_CU_CYCLES / SQ_BUSY_CYCLES            share of the time a CU had a wave while its shader engine was busy
  SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 * 4)   aggregate MFMA utilisation of the kernels as the
                                                 collector ran them (ONE AT A TIME: the sum of the kernels' own
                                                 durations is the denominator, not the overlapped evaluation)
The collector serialises kernels, so these are per-kernel figures weighted by kernel time -- an upper bound of what the
overlapped schedule can reach per kernel, and the honest statement of how busy the matrix pipes are while each runs."""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
want = sys.argv[2:]
tot = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for k, cn, v, d in c.execute("select kernel_name,counter_name,value,dispatch_id from counters_collection"):
    k = k.replace("void tgp::(anonymous namespace)::", "").split("(")[0]
    tot[k][cn] += v
    disp[k].add(d)
names = [n for n in want if any(n in t for t in tot.values())] or sorted({n for t in tot.values() for n in t})
allk = defaultdict(float)
print("# kernel | dispatches | " + " | ".join(names))
for k in sorted(tot, key=lambda k: -tot[k].get(names[0], 0)):
    print(f"{k[:60]} | {len(disp[k])} | " + " | ".join(f"{tot[k].get(n, 0):.5g}" for n in names))
    for n in names:
        allk[n] += tot[k].get(n, 0)
print("ALL KERNELS | - | " + " | ".join(f"{allk[n]:.6g}" for n in names))
if "SQ_VALU_MFMA_BUSY_CYCLES" in allk and "SQ_BUSY_CU_CYCLES" in allk and allk["SQ_BUSY_CU_CYCLES"]:
    # both are summed over the chip: MFMA-busy SIMD cycles per busy CU cycle (4 SIMDs per CU)
    print("MFMA busy / (4 x CU busy) over all kernels = %.3f" % (allk["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * allk["SQ_BUSY_CU_CYCLES"])))
    for k in sorted(tot, key=lambda k: -tot[k].get("SQ_BUSY_CU_CYCLES", 0))[:6]:
        t = tot[k]
        if t.get("SQ_BUSY_CU_CYCLES"):
            print(f"   {k[:50]}: {t['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * t['SQ_BUSY_CU_CYCLES']):.3f}")
