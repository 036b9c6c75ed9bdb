#!/usr/bin/env python3
"""Effective clock per kernel from a rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace run (MI355X_MICROARCH.md, DVFS give-back):
clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time (reads high on dispatches under ~0.3 ms).
python tools/rocpd_clock.py results.db [min_us]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
cols = [r[1] for r in db.execute("pragma table_info(pmc_events)")]
rows = db.execute("select name, counter_name, counter_value, start, end from pmc_events").fetchall() if "start" in cols else None
if rows is None:
    print("pmc_events columns:", cols)
    sys.exit(1)
agg = {}
for name, cn, val, s, e in rows:
    if cn != "GRBM_GUI_ACTIVE" or e <= s:
        continue
    a = agg.setdefault(name, [0, 0.0, 0.0])
    a[0] += 1; a[1] += val; a[2] += (e - s)
for name, (n, cyc, ns) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    if ns / n / 1e3 >= min_us:
        print(f"{name[:70]:70s} calls {n:4d} avg {ns / n / 1e3:8.1f} us  clock {cyc / 8 / ns * 1e3:7.0f} MHz")
