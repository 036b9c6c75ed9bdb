#!/bin/bash
# round 2 check: GPU suite, bench lines (C4 default, C2, C3), one-rank RCCL communicator line, 2-rank gloo rehearsal
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2b
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/tests.log 2>&1; rc=$?
tail -5 $OUT/tests.log
[ $rc -eq 0 ] || { grep -n "Error\|assert" $OUT/tests.log | head -20; exit $rc; }
for wl in C4 C2 C3; do
  timeout -k 10 400 python3 bench.py --workload $wl > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err || { tail -5 $OUT/bench_$wl.err; exit 1; }
  cut -c1-330 $OUT/bench_$wl.json
done
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --force-sharded --no-cpu-baseline > $OUT/bench_rccl1.json 2> $OUT/bench_rccl1.err || { tail -5 $OUT/bench_rccl1.err; exit 1; }
python3 -c "import json,sys; d=json.loads(open('$OUT/bench_rccl1.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['comm'], {k:v['ms'] for k,v in d['kernels'].items() if k.startswith('xch')})"
PVAE_BENCH_BACKEND=gloo timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 20 --repeats 2 --workload C2 > $OUT/bench_gloo2.json 2> $OUT/bench_gloo2.err || { tail -5 $OUT/bench_gloo2.err; exit 1; }
python3 -c "import json,sys; d=json.loads(open('$OUT/bench_gloo2.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['comm'], d['data'])"
