#!/bin/bash
# round 3, first call: the new sharded-width / default-lr / p2p tests, then the stream ceiling finder
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3first
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -u -m pytest tests/test_gpu_fullsize.py::test_snp_shards_add_up_at_full_size tests/test_gpu_fullsize.py::test_headline_shape_at_keras_default_lr tests/test_gpu_comm.py tests/test_gpu_dist.py tests/test_gpu_pabu.py::test_test_flag_runs_in_this_tree -m gpu -x -v -s --timeout 300 --timeout-method thread > $OUT/tests.log 2>&1; rc=$?
tail -15 $OUT/tests.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $OUT/tests.log | head -30; exit $rc; }
timeout -k 10 300 ./tools/stream_bench > $OUT/stream_bench.txt 2>&1 || { tail -5 $OUT/stream_bench.txt; exit 1; }
cat $OUT/stream_bench.txt
