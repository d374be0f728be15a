#!/bin/bash
# k_fin at the C2 / C3 stand-in shapes (long contigs, 3 sub-fragments per bin): fragments y per mass unit x grid size
out=gpurun_out/fin_seg_ab.log; : > $out
for shape in "1086 120000" "3500 600000"; do
  set -- $shape
  for seg in 0 4 2 1; do for blk in 512 768 2048; do
    echo "== n_bins $1 nnz $2  GRAAL_FIN_SEG=$seg GRAAL_FIN_BLOCKS=$blk" >> $out
    GRAAL_FIN_SEG=$seg GRAAL_FIN_BLOCKS=$blk timeout -k 10 120 python tools/step_breakdown.py --n-bins $1 --nnz $2 --n-sub 3 --original --steps 1500 2>/dev/null | grep -E "full MCMC step|scoring" >> $out || exit 1
  done; done
done
cat $out
