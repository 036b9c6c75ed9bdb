#!/bin/bash
# ms/step of the C-side epoch loop at every BASELINE shape (no HIP events)
for s in C1 C2 C5b C5a C3 C4; do
  NOTIMING=1 timeout -k 10 120 python tools/kbench.py $s 71 | grep -E "ms/step" || exit 1
done
