#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py::test_epoch_loop_equals_stepwise_at_full_size tests/test_gpu_fit.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/gputests.log 2>&1
rc=$?; tail -3 gpurun_out/gputests.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" gpurun_out/gputests.log | head -20; exit $rc; }
./tools/gpu_profile.sh
