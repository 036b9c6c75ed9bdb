#!/bin/bash
# after pvk_dec3b: full GPU suite, batch sweep, kernel stats at C4 batch 128
set -o pipefail
export TMPDIR=/tmp
O=$PWD/gpurun_out/final2; rm -rf $O; mkdir -p $O
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1; rc=$?
tail -3 $O/tests.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $O/tests.log | head -20; exit $rc; }
./tools/gpu_batch_sweep.sh > $O/batch_sweep.log 2>&1 || { tail -5 $O/batch_sweep.log; exit 1; }
cp gpurun_out/sweep/sweep.txt $O/batch_sweep.txt; cat $O/batch_sweep.txt
NOTIMING=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 tools/kbench.py C4 20 128 > $O/kt.log 2>&1 || { tail -5 $O/kt.log; exit 1; }
python3 tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) --csv $O/kernel_stats_C4_B128.csv --skip 2 > $O/stats.txt; sed -n 1,12p $O/stats.txt
