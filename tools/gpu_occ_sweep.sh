#!/bin/bash
# occupancy sweep of the wide kernels (grid = CUs x blocks/CU) at C4 and C3
for cfg in C4 C3; do
  echo "== $cfg default"; PVAE_VERBOSE=1 timeout -k 10 120 python tools/kbench.py $cfg 30 2>&1 | grep -E "pvae\]|dec |enc_bwd|ms/step" || exit 1
  for o in 1 2 3 4; do
    echo "== $cfg DEC_OCC=$o"; PVAE_DEC_OCC=$o timeout -k 10 120 python tools/kbench.py $cfg 30 2>&1 | grep -E "dec " || exit 1
  done
  for o in 1 2 3; do
    echo "== $cfg ENCB_OCC=$o"; PVAE_ENCB_OCC=$o timeout -k 10 120 python tools/kbench.py $cfg 30 2>&1 | grep -E "enc_bwd" || exit 1
  done
done
