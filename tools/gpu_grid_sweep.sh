#!/bin/bash
# wide-kernel grid sizes: does a balanced grid (every wave the same number of subtiles) beat the full one?
# usage: gpu_grid_sweep.sh SHAPE WGS...
sh=$1; shift
for g in "$@"; do
  echo "== WGS $g"
  PVAE_DEC_WGS=$g PVAE_ENCB_WGS=$g timeout -k 10 100 python tools/kbench.py $sh 71 | grep -E "ms/step| dec | enc_bwd " || exit 1
done
