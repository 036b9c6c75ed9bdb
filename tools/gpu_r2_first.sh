#!/bin/bash
# round 2, first GPU call: the whole GPU suite, then C2 timing + kernel trace as the baseline for the shard-regime work
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2a
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/tests.log 2>&1; rc=$?
tail -15 $OUT/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python3 tools/kbench.py C2 200 > $OUT/kbench_C2.txt 2>&1 && cat $OUT/kbench_C2.txt || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/kt_c2 -o kt -- python3 tools/kbench.py C2 200 > $OUT/kt_c2.log 2>&1 || { tail -5 $OUT/kt_c2.log; exit 1; }
find $OUT/kt_c2 -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -20 {}'
