#!/bin/bash
# grids of the wide kernels at 100 k SNPs (3125 SNP blocks): does an even share per workgroup beat the full co-resident grid?
set -o pipefail
O=gpurun_out/gridc2; mkdir -p $O
for d in 512 447 391 348; do
  PVAE_DEC2_WGS=$d timeout -k 10 120 python tools/kbench.py C2 71 > $O/d_$d.log 2>&1 || { tail -3 $O/d_$d.log; exit 1; }
  echo "dec2 grid $d: $(grep ' dec ' $O/d_$d.log)"
done
for b in 0 447 391 1042 782; do
  if [ $b = 0 ]; then unset PVAE_ENCB_WGS; else export PVAE_ENCB_WGS=$b; fi
  PVAE_VERBOSE=1 timeout -k 10 120 python tools/kbench.py C2 71 > $O/e_$b.log 2>&1 || { tail -3 $O/e_$b.log; exit 1; }
  echo "enc_bwd grid $b: $(grep ' enc_bwd ' $O/e_$b.log) $(grep -o 'enc_bwd [0-9]*' $O/e_$b.log | head -1)"
done
