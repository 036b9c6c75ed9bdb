#!/usr/bin/env python3
"""Per-kernel average of a PMC counter from a rocprofv3 rocpd sqlite file.
FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B?  rocprofv3 reports them in kilobytes (KB = 1024 B).
usage: python tools/rocpd_pmc.py results.db [--json out.json]"""
import json
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, counter_name, counter_value from pmc_events").fetchall()
agg = {}
for name, cn, val in rows:
    a = agg.setdefault((name, cn), [0, 0.0])
    a[0] += 1
    a[1] += val
out = {}
for (name, cn), (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{name[:80]:80s} {cn:12s} calls {n:5d} avg {tot / n:16.1f}")
    out.setdefault(name, {})[cn] = {"calls": n, "avg": tot / n}
if "--json" in sys.argv:
    json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
