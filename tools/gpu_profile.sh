#!/bin/bash
# rocprofv3 evidence for the bench line: kernel-trace summary of bench.py, then HBM traffic counters (separate --pmc passes)
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python3 bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/kt.err || { tail -5 $OUT/kt.err; exit 1; }
tail -1 $OUT/bench_under_rocprof.json | cut -c1-200
ls -R $OUT/kt | head -20
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c -d $OUT/pmc_$c -o pmc -- python3 tools/kbench.py C4 12 > $OUT/pmc_$c.log 2>&1 || { tail -5 $OUT/pmc_$c.log; exit 1; }
done
find $OUT -name "*.db" -o -name "*.csv" | head -20
