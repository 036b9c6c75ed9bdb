#!/bin/bash
# decoder pass in two launches (PVAE_SPLIT_DEC=1): parity of the epoch loop, then step time with and without at shard sizes
set -o pipefail
mkdir -p gpurun_out/split
O=gpurun_out/split
PVAE_SPLIT_DEC=1 timeout -k 10 400 python -u -m pytest tests/test_gpu_fit.py tests/test_gpu_parity.py -m gpu -x -q --timeout 200 --timeout-method thread > $O/tests.log 2>&1
rc=$?; tail -3 $O/tests.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $O/tests.log | head; exit $rc; }
for wl in ${WLS:-C2}; do
  for sd in 0 1; do
    PVAE_SPLIT_DEC=$sd NOTIMING=1 timeout -k 10 120 python tools/kbench.py $wl 71 > $O/kb_nt_${wl}_$sd.log 2>&1 || { tail -5 $O/kb_nt_${wl}_$sd.log; exit 1; }
    echo "split=$sd $(sed -n 1p $O/kb_nt_${wl}_$sd.log)"
  done
  PVAE_SPLIT_DEC=1 timeout -k 10 120 python tools/kbench.py $wl 71 > $O/kb_${wl}_1.log 2>&1 || exit 1
  cat $O/kb_${wl}_1.log
  for r in 0 1 2; do
    PVAE_SPLIT_RESERVE=$r PVAE_SPLIT_DEC=1 NOTIMING=1 timeout -k 10 120 python tools/kbench.py $wl 71 > $O/kb_r.log 2>&1 || exit 1
    echo "reserve=$r $(sed -n 1p $O/kb_r.log)"
  done
done
