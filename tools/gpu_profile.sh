#!/bin/bash
# rocprofv3 evidence for the bench lines: kernel-trace summary of bench.py at C4 and C2, then HBM traffic counters at both
# shapes (separate --pmc passes); summaries land in gpurun_out/prof, copy what is to be judged into profiles/
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
for wl in C4 C2; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_$wl -o kt -- python3 bench.py --no-cpu-baseline --workload $wl > $OUT/bench_${wl}_under_rocprof.json 2> $OUT/kt_$wl.err || { tail -5 $OUT/kt_$wl.err; exit 1; }
  tail -1 $OUT/bench_${wl}_under_rocprof.json | cut -c1-200
  db=$(find $OUT/kt_$wl -name "*.db" | head -1)
  [ -n "$db" ] && python3 tools/rocpd_stats.py $db --csv $OUT/kernel_stats_$wl.csv --skip 5
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 200 rocprofv3 --pmc $c -d $OUT/pmc_${wl}_$c -o pmc -- python3 tools/kbench.py $wl 12 > $OUT/pmc_${wl}_$c.log 2>&1 || { tail -5 $OUT/pmc_${wl}_$c.log; exit 1; }
  done
  python3 tools/pmc_traffic_json.py $(find $OUT/pmc_${wl}_FETCH_SIZE -name "*.db" | head -1) $(find $OUT/pmc_${wl}_WRITE_SIZE -name "*.db" | head -1) $OUT/pmc_traffic_$wl.json $wl || exit 1
done
ls $OUT
