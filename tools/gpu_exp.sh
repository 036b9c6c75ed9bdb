#!/bin/bash
for f in tools/_build/libpvae_x*.so; do
  echo "== $f"
  PVAE_LIB=$f timeout -k 10 100 python tools/trace_narrow.py 2>&1 | grep -v cycles | sed -n '2,5p;10,11p' || exit 1
done
