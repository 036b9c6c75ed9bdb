#!/bin/bash
# parity tests that exercise the narrow stack, then the per-op trace and the C2 / C4 kernel timings
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fit.py tests/test_gpu_random_shapes.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/gputests.log 2>&1
rc=$?; tail -5 gpurun_out/gputests.log
[ $rc -eq 0 ] || exit $rc
PVAE_LIB=tools/_build/libpvae_trace.so timeout -k 10 100 python tools/trace_narrow.py > gpurun_out/trace.log 2>&1 || { cat gpurun_out/trace.log; exit 1; }
grep -v cycles gpurun_out/trace.log
for wl in C2 C4; do
  timeout -k 10 120 python tools/kbench.py $wl 71 > gpurun_out/kbench_$wl.log 2>&1 || exit 1
  NOTIMING=1 timeout -k 10 120 python tools/kbench.py $wl 71 2>&1 > gpurun_out/kbench_nt_$wl.log || exit 1
  cat gpurun_out/kbench_$wl.log; sed -n 1p gpurun_out/kbench_nt_$wl.log
done
