#!/bin/bash
# quick iteration check: step-level parity tests, then kernel timings at C2 and C4 (tools/kbench.py)
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/quick
mkdir -p $OUT
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q --timeout 200 --timeout-method thread > $OUT/tests.log 2>&1; rc=$?
tail -3 $OUT/tests.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $OUT/tests.log | head -20; exit $rc; }
for wl in C2 C4 "$@"; do
  timeout -k 10 200 python3 tools/kbench.py $wl 71 > $OUT/kb_$wl.txt 2>&1 || { tail -5 $OUT/kb_$wl.txt; exit 1; }
  NOTIMING=1 timeout -k 10 200 python3 tools/kbench.py $wl 71 2>&1 | head -1 >> $OUT/kb_$wl.txt
  cat $OUT/kb_$wl.txt
done
