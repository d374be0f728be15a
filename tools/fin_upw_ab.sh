#!/bin/bash
# k_fin: mass units per wave the unit size aims at (GRAAL_FIN_UPW) on the C3 shape and the late stage of C5
out=gpurun_out/fin_upw_ab.log; : > $out
for upw in 2 4 8 16; do
  echo "== GRAAL_FIN_UPW=$upw" >> $out
  GRAAL_FIN_UPW=$upw timeout -k 10 120 python tools/step_breakdown.py --n-bins 3500 --nnz 600000 --n-sub 3 --original --steps 1500 2>&1 | grep -E "scoring" >> $out || exit 1
  GRAAL_FIN_UPW=$upw timeout -k 10 120 python tools/step_breakdown.py --n-bins 1086 --nnz 120000 --n-sub 3 --original --steps 1500 2>&1 | grep -E "scoring" >> $out || exit 1
  GRAAL_FIN_UPW=$upw timeout -k 10 300 python bench.py --layout original --steps 30 --warmup 5 --mcmc-warmup 0 --no-cpu-baseline --no-late-stage --no-hbm-control 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('late stage: us/step %.1f' % (1e3 * j['ms_per_step']))
" >> $out || exit 1
done
cat $out
