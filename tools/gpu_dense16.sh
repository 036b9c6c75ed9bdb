#!/bin/bash
# per-layer narrow kernels at widths 256 / 512 (pvk_dense_fwd16 / bwd16): parity at those widths, then C3 / C5a kernel timings
set -o pipefail
O=gpurun_out/d16; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_random_shapes.py tests/test_gpu_comm.py -m gpu -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1
rc=$?; tail -3 $O/tests.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $O/tests.log | head; exit $rc; }
for wl in C3 C5a; do
  for nd in 1 0; do
    if [ $nd = 1 ]; then export PVAE_NO_DENSE16=1; else unset PVAE_NO_DENSE16; fi
    timeout -k 10 200 python3 tools/kbench.py $wl 54 > $O/kb_${wl}_$nd.txt 2>&1 || { tail -5 $O/kb_${wl}_$nd.txt; exit 1; }
    echo "no_dense16=$nd $(sed -n 1p $O/kb_${wl}_$nd.txt) | $(grep narrow_fwd $O/kb_${wl}_$nd.txt) | $(grep narrow_bwd $O/kb_${wl}_$nd.txt)"
  done
done
