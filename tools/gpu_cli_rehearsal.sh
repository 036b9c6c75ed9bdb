#!/bin/bash
# popvae.py end to end on the pabu fixture: one rank, then two ranks sharing the GPU (SNP-axis sharding over gloo), same seed
set -o pipefail
O=$PWD/gpurun_out/cli; rm -rf $O; mkdir -p $O
VCF=tests/golden/pabu_test_genotypes.vcf.gz
timeout -k 10 120 python popvae.py --infile $VCF --out $O/one --seed 42 --max_epochs 60 --patience 20 > $O/one.log 2>&1 || { tail -5 $O/one.log; exit 1; }
PVAE_DIST_BACKEND=gloo timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 \
  popvae.py --infile $VCF --out $O/two --seed 42 --max_epochs 60 --patience 20 --PCA > $O/two.log 2>&1 || { tail -15 $O/two.log; exit 1; }
PVAE_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29532 \
  popvae.py --infile $VCF --out $O/grid --seed 42 --max_epochs 15 --patience 5 --search_network_sizes --width_range 32,64 --depth_range 2,3 > $O/grid.log 2>&1 || { tail -15 $O/grid.log; exit 1; }
ls $O
python - <<'PY'
import numpy as np, os
O = os.environ.get("O", "gpurun_out/cli")
def load(p):
    rows = [l.rstrip("\n").split("\t") for l in open(p)]
    return rows[0], rows[1:]
h1, a = load(f"{O}/one_latent_coords.txt"); h2, b = load(f"{O}/two_latent_coords.txt")
assert h1 == h2 and len(a) == len(b), (h1, h2, len(a), len(b))
A = np.array([[float(x) for x in r[:2]] for r in a]); B = np.array([[float(x) for x in r[:2]] for r in b])
print("samples", len(a), "ids equal", [r[-1] for r in a] == [r[-1] for r in b])
print("max |mean one - mean two|", np.abs(A - B).max(), "scale", np.abs(A).max())
H1 = open(f"{O}/one_history.txt").read().splitlines(); H2 = open(f"{O}/two_history.txt").read().splitlines()
print("epochs", len(H1) - 1, len(H2) - 1); print(H1[1]); print(H2[1]); print(H1[-1]); print(H2[-1])
print(open(f"{O}/grid_param_grid.csv").read())
PY
