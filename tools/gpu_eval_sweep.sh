#!/bin/bash
# blocks per CU of the read-only passes (validation steps): usage gpu_eval_sweep.sh SHAPE VAR values...
sh=$1; var=$2; shift 2
for o in "$@"; do
  echo "== $var $o"
  env $var=$o NOTIMING=1 timeout -k 10 100 python tools/kbench.py $sh 20 | grep -E "eval step" || exit 1
done
