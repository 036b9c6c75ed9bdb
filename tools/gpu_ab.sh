#!/bin/bash
# A/B of two engine builds in one call: tools/gpu_ab.sh libA.so libB.so [shapes...]; NOTIMING kbench, alternating
export TMPDIR=/tmp NOTIMING=1
A=$1; B=$2; shift 2
for sh in "$@"; do for rep in 1 2; do for v in $A $B; do
  echo "$sh $(basename $v): $(PVAE_LIB=$PWD/$v timeout -k 10 100 python3 tools/kbench.py $sh 71 2>/dev/null | sed -n 1p)"
done; done; done
