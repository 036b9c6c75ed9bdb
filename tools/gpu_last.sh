#!/bin/bash
# smoke(), the bench lines at HEAD, then the rocprofv3 evidence for them (tools/gpu_profile.sh)
set -o pipefail
export TMPDIR=/tmp
O=$PWD/gpurun_out/last; rm -rf $O; mkdir -p $O
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
for wl in C4 C2 C3; do
  timeout -k 10 400 python3 bench.py --workload $wl > $O/bench_$wl.json 2> $O/bench_$wl.err || { tail -5 $O/bench_$wl.err; exit 1; }
  cut -c1-260 $O/bench_$wl.json
done
timeout -k 10 200 python3 tools/kbench.py C3 54 > $O/kb_C3.txt 2>&1 || { tail -5 $O/kb_C3.txt; exit 1; }
./tools/gpu_profile.sh > $O/profile.log 2>&1 || { tail -8 $O/profile.log; exit 1; }
tail -3 $O/profile.log
