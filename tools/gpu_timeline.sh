#!/bin/bash
# kernel timeline of a few C2 train steps (rocprofv3 kernel trace): durations and the gaps between consecutive kernels
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/tl; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace -d $OUT -o tl -- python3 tools/kbench.py ${1:-C2} 40 > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - <<'PY'
import sqlite3, glob, os
db = sqlite3.connect(glob.glob(os.environ.get("OUT", "gpurun_out/tl") + "/*.db")[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
# last 3 train steps: find the last pvk_enc_bwd occurrences
idx = [i for i, r in enumerate(rows) if "pvk_enc_bwd" in r[0]]
a, b = idx[-12], idx[-9]
prev_end = None
for name, s, e in rows[a + 1:b + 1]:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{name.split('(')[0][-40:]:42s} dur {(e - s) / 1e3:8.1f} us   gap before {gap:6.1f} us")
    prev_end = e
PY
