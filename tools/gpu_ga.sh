#!/bin/bash
# next batch gathered on a second stream (PVAE_GATHER_AHEAD=1) against the head gathering its own batch (default): parity, then step times and the timeline
set -o pipefail
export TMPDIR=/tmp
O=$PWD/gpurun_out/ga; rm -rf $O; mkdir -p $O
timeout -k 10 700 python -u -m pytest tests/test_gpu_optins.py tests/test_gpu_fit.py tests/test_gpu_parity.py tests/test_gpu_comm.py tests/test_gpu_fullsize.py -m gpu -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1
rc=$?; tail -3 $O/tests.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $O/tests.log | head; exit $rc; }
for wl in C2 C4; do
  for ga in 0 1; do
    if [ $ga = 1 ]; then export PVAE_GATHER_AHEAD=1; else unset PVAE_GATHER_AHEAD; fi
    for i in 1 2; do
      NOTIMING=1 timeout -k 10 120 python tools/kbench.py $wl 71 > $O/kb.log 2>&1 || { tail -5 $O/kb.log; exit 1; }
      echo "gather_ahead=$ga $(sed -n 1p $O/kb.log)"
    done
  done
done
export PVAE_GATHER_AHEAD=1
NOTIMING=1 ./tools/gpu_timeline.sh C2 | tail -8
