"""Per-kernel sums of several PMC counters from one rocprofv3 (rocpd) database, plus the MFMA
utilisation the gfx94x derived-metric formula gives (ROCm 7.2 ships no gfx950 section):

    MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / XCDs x CUs x 4 SIMDs)

rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (one GRBM each: a 9.6 ms dispatch at
~2.1 GHz shows 1.8e8 = 8 x 2.2e7 cycles) and SQ_VALU_MFMA_BUSY_CYCLES summed over every SIMD of
the chip, so the busy cycles of ONE SIMD are the sum / (CUs x 4) and the active cycles of the
dispatch are GRBM_GUI_ACTIVE / 8.

usage: pmc_multi.py <db> [CUs=256] [XCDs=8]"""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
cus = int(sys.argv[2]) if len(sys.argv) > 2 else 256
xcds = int(sys.argv[3]) if len(sys.argv) > 3 else 8
tot = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
names = set()
for k, cn, v in c.execute("select kernel_name,counter_name,value from counters_collection"):
    k = k.replace("void tgp::(anonymous namespace)::", "").split("(")[0]
    tot[k][cn] += v
    names.add(cn)
for k, n in c.execute("select kernel_name,count(*) from counters_collection group by kernel_name,counter_name"):
    k = k.replace("void tgp::(anonymous namespace)::", "").split("(")[0]
    cnt[k] = max(cnt[k], n)
names = sorted(names)
print("# kernel | dispatches | " + " | ".join(names) + " | MfmaUtil")
key = "GRBM_GUI_ACTIVE" if "GRBM_GUI_ACTIVE" in names else names[0]
for k in sorted(tot, key=lambda k: -tot[k][key]):
    t = tot[k]
    util = ""
    if t.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in t:
        util = f"{t['SQ_VALU_MFMA_BUSY_CYCLES'] / (t['GRBM_GUI_ACTIVE'] / xcds * cus * 4):.3f}"
    print(f"{k} | {cnt[k]} | " + " | ".join(f"{t.get(n, 0):.4g}" for n in names) + f" | {util}")
