#!/bin/bash
# pvk_dec3b (four row tiles, logits and dV on the bf16 pipe): large-batch parity, then step times against pvk_dec3<4>
set -o pipefail
O=gpurun_out/dec3b; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fit.py -m gpu -x -q -k "large_batch or weight_pass or batch" --timeout 300 --timeout-method thread > $O/tests.log 2>&1
rc=$?; tail -4 $O/tests.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $O/tests.log | head -20; exit $rc; }
for B in 128 256; do
  for nb in 0 1; do
    if [ $nb = 1 ]; then export PVAE_NO_DEC3B=1; else unset PVAE_NO_DEC3B; fi
    NOTIMING=1 timeout -k 10 200 python3 tools/kbench.py C4 ${STEPS:-30} $B > $O/kb_${B}_$nb.log 2>&1 || { tail -5 $O/kb_${B}_$nb.log; exit 1; }
    echo "no_dec3b=$nb $(sed -n 1p $O/kb_${B}_$nb.log)"
  done
done
unset PVAE_NO_DEC3B
timeout -k 10 200 python3 tools/kbench.py C4 30 128 > $O/kb_128_ev.log 2>&1 || exit 1
cat $O/kb_128_ev.log
