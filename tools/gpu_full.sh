#!/bin/bash
# full GPU suite + kernel timings at C3 + bench lines (C4 default, C2, C3) into gpurun_out/full
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/full
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/tests.log 2>&1; rc=$?
tail -4 $OUT/tests.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $OUT/tests.log | head -20; exit $rc; }
timeout -k 10 200 python3 tools/kbench.py C3 54 > $OUT/kb_C3.txt 2>&1 || { tail -5 $OUT/kb_C3.txt; exit 1; }
cat $OUT/kb_C3.txt
for wl in C4 C2 C3; do
  timeout -k 10 400 python3 bench.py --workload $wl > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err || { tail -5 $OUT/bench_$wl.err; exit 1; }
  cut -c1-400 $OUT/bench_$wl.json
done
