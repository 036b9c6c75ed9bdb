#!/bin/bash
# end-of-round measurements beside the profiles: shard-size sweep, batch-size sweep, BASELINE config 5 as a run
set -o pipefail
export TMPDIR=/tmp
O=$PWD/gpurun_out/final; mkdir -p $O
timeout -k 10 300 python3 tools/shard_sweep.py $O/shard_sweep.json > $O/shard_sweep.log 2>&1 || { tail -5 $O/shard_sweep.log; exit 1; }
tail -3 $O/shard_sweep.log
./tools/gpu_batch_sweep.sh > $O/batch_sweep.log 2>&1 || { tail -5 $O/batch_sweep.log; exit 1; }
cp gpurun_out/sweep/sweep.txt $O/batch_sweep.txt; cat $O/batch_sweep.txt
timeout -k 10 500 python3 tools/grid_c5.py > $O/grid_c5.json 2> $O/grid_c5.err || { tail -5 $O/grid_c5.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/grid_c5.json')); print(d['wall_seconds_whole_command']); [print(c) for c in d['cells']]; print('\n'.join(d['param_grid_csv']))"
