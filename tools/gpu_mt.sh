#!/bin/bash
# multi-tile passes (batches above 32 rows, validation four at a time, encoder.predict): parity, then timings at C4 / C2
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/mt
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fit.py tests/test_gpu_random_shapes.py "tests/test_gpu_fullsize.py::test_full_size_batches_in_one_weight_pass" -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/mt/tests.log 2>&1
rc=$?; tail -3 gpurun_out/mt/tests.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" gpurun_out/mt/tests.log | head -20; exit $rc; }
for b in 64 128; do timeout -k 10 200 python3 tools/kbench.py C4 40 $b || exit 1; done
timeout -k 10 200 python3 tools/kbench.py C2 40 128 || exit 1
timeout -k 10 200 python3 tools/encode_time.py || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/mt/kt -o kt -- python3 tools/kbench.py C4 20 128 > gpurun_out/mt/kt.log 2>&1 || { tail -5 gpurun_out/mt/kt.log; exit 1; }
python3 tools/rocpd_stats.py $(find gpurun_out/mt/kt -name "*.db" | head -1) --csv gpurun_out/mt/kernel_stats_C4_B128.csv --skip 2  > gpurun_out/mt/stats.txt; sed -n 1,12p gpurun_out/mt/stats.txt
