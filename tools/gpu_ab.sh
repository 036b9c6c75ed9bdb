#!/bin/bash
# A/B of engine builds in one process sequence on one box: tools/gpu_ab.sh libA.so libB.so ...  (kbench without events, 3 rounds interleaved)
set -o pipefail
mkdir -p gpurun_out/ab
for round in 1 2 3; do
  for lib in "$@"; do
    for wl in C2 C4; do
      echo -n "$(basename $lib) r$round: "
      PVAE_LIB=$lib NOTIMING=1 timeout -k 10 120 python3 tools/kbench.py $wl 142 | sed -n '1p;$p' | tr '\n' ' ' || exit 1
      echo
    done
    echo -n "$(basename $lib) r$round: "
    PVAE_LIB=$lib NOTIMING=1 timeout -k 10 120 python3 tools/kbench.py C4 40 128 | sed -n '1p;$p' | tr '\n' ' ' || exit 1
    echo
  done
done
