#!/bin/bash
# rehearsal of bench.py --gpus N on a one-GPU box: one rank through the engine-owned RCCL communicator (--force-sharded), two ranks
# sharing the GPU over gloo (host-staged exchanges) and through the one-shot exchange
set -o pipefail
export TMPDIR=/tmp
O=$PWD/gpurun_out/rehearse; rm -rf $O; mkdir -p $O
timeout -k 10 300 python3 bench.py --workload C2 --force-sharded --no-cpu-baseline --steps 30 --warmup 3 > $O/n1_rccl.json 2> $O/n1_rccl.err || { tail -5 $O/n1_rccl.err; exit 1; }
python3 -c "import json,sys; d=json.loads(open('$O/n1_rccl.json').read().strip().splitlines()[-1]); print('n1 forced', round(d['ms_per_step'],4), d['config']['parallelism'], d['comm'])"
PVAE_BENCH_BACKEND=gloo timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --workload C2 --no-cpu-baseline --steps 20 --warmup 3 > $O/n2_gloo.json 2> $O/n2_gloo.err || { tail -8 $O/n2_gloo.err; exit 1; }
python3 -c "import json,sys; d=json.loads(open('$O/n2_gloo.json').read().strip().splitlines()[-1]); print('n2 gloo', round(d['ms_per_step'],4), d['n_gpus'], d['config']['parallelism'], d['comm'])"
PVAE_COMM=p2p PVAE_BENCH_BACKEND=gloo timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --workload C2 --no-cpu-baseline --steps 20 --warmup 3 > $O/n2_p2p.json 2> $O/n2_p2p.err || { tail -8 $O/n2_p2p.err; exit 1; }
python3 -c "import json,sys; d=json.loads(open('$O/n2_p2p.json').read().strip().splitlines()[-1]); print('n2 p2p', round(d['ms_per_step'],4), d['n_gpus'], d['config']['parallelism'], d['comm'])"
