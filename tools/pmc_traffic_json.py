#!/usr/bin/env python3
"""profiles/rNN_pmc_traffic_<shape>.json from the two rocprofv3 --pmc passes of tools/gpu_profile.sh:
python tools/pmc_traffic_json.py <FETCH_SIZE results.db> <WRITE_SIZE results.db> <out.json> [shape]
Corrections (MI355X_MICROARCH.md, HBM section): both counters are in units of 1024 B; on gfx950 FETCH_SIZE tallies
128-B requests at 64 B, so wide coalesced reads are doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores."""
import json
import sqlite3
import sys

KEYS = {"enc_fwd": "pvk_enc_fwd<", "dec": "pvk_dec2<4, 0>", "enc_bwd": "pvk_enc_bwd<4, 0, 1,", "dec_eval": "pvk_dec<4, 0, 0>"}
SHAPE = sys.argv[4] if len(sys.argv) > 4 else "C4"


def avg(dbf, counter):
    out = {}
    for name, cn, val in sqlite3.connect(dbf).execute("select name, counter_name, counter_value from pmc_events"):
        if cn != counter:
            continue
        for k, pat in KEYS.items():
            if pat in name:
                a = out.setdefault(k, [0, 0.0])
                a[0] += 1
                a[1] += val
    return {k: (n, tot / n) for k, (n, tot) in out.items()}


f, w = avg(sys.argv[1], "FETCH_SIZE"), avg(sys.argv[2], "WRITE_SIZE")
res = {"_about": f"HBM traffic per launch at {SHAPE} (width 128): rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in "
                 f"separate passes over `tools/kbench.py {SHAPE} 12` (tools/gpu_profile.sh, tools/pmc_traffic_json.py). Counters are "
                 "in units of 1024 B; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so wide coalesced reads are doubled "
                 "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-B-per-lane streaming stores.",
       "kernels": {}}
for k in KEYS:
    if k in f and k in w:
        fb, wb = f[k][1] * 1024 * 2, w[k][1] * 1024
        res["kernels"][k] = {"fetch_bytes": fb, "fetch_counter_avg_raw": f[k][1], "launches": f[k][0], "write_bytes": wb,
                             "write_counter_avg_raw": w[k][1], "traffic_bytes": fb + wb}
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v["traffic_bytes"] for k, v in res["kernels"].items()}))
