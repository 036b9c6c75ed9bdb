#!/bin/bash
# batch-size sweep (SURVEY 8d secondary): ms/step without events at C4 and C2, B = 32 / 64 / 128, one weight pass vs 32-row tiles
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/sweep
mkdir -p $OUT
: > $OUT/sweep.txt
run() {  # label, env assignments..., -- shape steps B
  local label=$1; shift
  env "$@" NOTIMING=1 timeout -k 10 200 python3 tools/kbench.py $SHAPE $STEPS $B > $OUT/last.txt 2>&1 || { tail -5 $OUT/last.txt; exit 1; }
  echo "$label$(head -1 $OUT/last.txt)" | tee -a $OUT/sweep.txt
}
STEPS=30
for SHAPE in C4 C2; do
  for B in 32 64 128; do
    run "" X=1
    [ $B -gt 32 ] && run "   tiles: " PVAE_NO_MULTITILE=1
  done
done
for B in 256 512; do
  SHAPE=C4 STEPS=10; run "" X=1
  run "   tiles: " PVAE_NO_MULTITILE=1
done
SHAPE=C2 STEPS=10 B=512; run "" X=1
