#!/bin/bash
# A/B of engine builds on one box: tools/gpu_ab.sh libA.so libB.so ...  -- per-kernel HIP-event times (kbench) and the step
# time without events, C4 and C2, three interleaved rounds
set -o pipefail
mkdir -p gpurun_out/ab
for round in 1 2 3; do
  for lib in "$@"; do
    for wl in C4 C2; do
      PVAE_LIB=$lib timeout -k 10 120 python3 tools/kbench.py $wl 71 > gpurun_out/ab/k.txt || exit 1
      nt=$(PVAE_LIB=$lib NOTIMING=1 timeout -k 10 120 python3 tools/kbench.py $wl 142 | sed -n '1p' | cut -d: -f2 | cut -c1-16) || exit 1
      echo "$(basename $lib) r$round $wl: no-events$nt | $(grep -E '^  (dec|enc_bwd) ' gpurun_out/ab/k.txt | awk '{printf "%s %s us  ", $1, $2}')"
    done
  done
done
