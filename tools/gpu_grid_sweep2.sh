#!/bin/bash
# grid sweep of the two wide kernels at a small shape (per-CU balance): tools/gpu_grid_sweep2.sh [shape] [lib]
set -o pipefail
export TMPDIR=/tmp
SH=${1:-C2}
for G in 128 256 320 391 512 640 768; do
  echo "== grid $G"
  PVAE_DEC_WGS=$G PVAE_ENCB_WGS=$G timeout -k 10 100 python3 tools/kbench.py $SH 71 | grep -E "ms/step|dec |enc_bwd" || exit 1
done
