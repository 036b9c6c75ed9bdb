#!/usr/bin/env python3
"""Per-kernel summary (calls, avg/min/max us, share) from a rocprofv3 rocpd sqlite file:
python tools/rocpd_stats.py results.db [--csv out.csv] [--skip N first dispatches of each kernel]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
skip = int(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else 0
rows = db.execute("select name, start, end from kernels order by start").fetchall()
seen, agg = {}, {}
for name, s, e in rows:
    seen[name] = seen.get(name, 0) + 1
    if seen[name] <= skip:
        continue
    a = agg.setdefault(name, [0, 0, 1 << 62, 0])
    d = e - s
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values()) or 1
lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append(f"\"{name}\",{a[0]},{a[1]},{a[1] / a[0]:.1f},{100 * a[1] / tot:.2f},{a[2]},{a[3]}")
    print(f"{name[:90]:90s} {a[0]:6d} {a[1] / a[0] / 1e3:10.1f} us {100 * a[1] / tot:6.2f}%")
if "--csv" in sys.argv:
    open(sys.argv[sys.argv.index("--csv") + 1], "w").write("\n".join(lines) + "\n")
