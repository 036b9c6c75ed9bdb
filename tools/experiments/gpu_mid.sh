#!/bin/bash
# grid-synchronised chain at widths 256 / 512: parity (workers on one die, then on consecutive workgroups), timings vs per-layer
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/mid; mkdir -p $OUT
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fit.py -m gpu -x -q --timeout 200 --timeout-method thread -k "256 or 512 or 200 or fit or epoch or staged or large" > $OUT/tests.log 2>&1; rc=$?
tail -3 $OUT/tests.log; [ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $OUT/tests.log | head; exit $rc; }
PVAE_MID_SPREAD=1 timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 200 --timeout-method thread -k "256 or 512 or 200" > $OUT/tests_spread.log 2>&1; rc=$?
tail -2 $OUT/tests_spread.log; [ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $OUT/tests_spread.log | head; exit $rc; }
for sh in C3 C5a; do
  echo "== $sh chain"; timeout -k 10 200 python3 tools/kbench.py $sh 54 | grep -E "ms/step|narrow" || exit 1
  echo "== $sh chain, any die"; PVAE_MID_SPREAD=1 timeout -k 10 200 python3 tools/kbench.py $sh 54 | grep -E "ms/step|narrow" || exit 1
  echo "== $sh per-layer"; PVAE_NO_MID_CHAIN=1 timeout -k 10 200 python3 tools/kbench.py $sh 54 | grep -E "ms/step|narrow" || exit 1
done
