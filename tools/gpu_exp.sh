#!/bin/bash
# kernel-variant experiments: every tools/_build/libpvae_x*.so / named libs through kbench at a shape (per-kernel HIP events)
sh=${1:-C4}; shift
for f in "$@"; do
  echo "== $f"
  PVAE_LIB=$f timeout -k 10 100 python tools/kbench.py $sh 40 | grep -E "ms/step| dec | enc_bwd " || exit 1
done
