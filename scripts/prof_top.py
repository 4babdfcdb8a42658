import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit %d" % (int(sys.argv[2]) if len(sys.argv) > 2 else 10)):
    nm = r[0].replace("void tgp::(anonymous namespace)::", "")[:72]
    print(f"{nm:72s} {r[1]:6d} tot {r[2]/1e3:9.2f} ms  avg {r[3]:9.2f} us {r[4]:6.2f}%")
