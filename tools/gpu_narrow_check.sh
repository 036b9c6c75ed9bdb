#!/bin/bash
# parity tests that exercise the narrow stack, then the per-op trace and the C4 kernel timings
set -o pipefail
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fit.py -x -q --timeout 200 --timeout-method thread > gpurun_out/gputests.log 2>&1
rc=$?; tail -5 gpurun_out/gputests.log
[ $rc -eq 0 ] || exit $rc
PVAE_LIB=tools/_build/libpvae_trace.so timeout -k 10 100 python tools/trace_narrow.py > gpurun_out/trace.log 2>&1 || { cat gpurun_out/trace.log; exit 1; }
cat gpurun_out/trace.log
timeout -k 10 120 python tools/kbench.py C4 40 > gpurun_out/kbench.log 2>&1
head -8 gpurun_out/kbench.log
