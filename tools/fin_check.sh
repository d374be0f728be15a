#!/bin/bash
# after a change to k_fin: the step at the C2 / C3 stand-in shapes, the C4 shape mid-run, and the late stage of C5 (defaults)
out=gpurun_out/fin_check.log; : > $out
for shape in "1086 120000 3" "3500 600000 3"; do
  set -- $shape
  echo "== n_bins $1 nnz $2 n_sub $3 (original contigs)" >> $out
  timeout -k 10 120 python tools/step_breakdown.py --n-bins $1 --nnz $2 --n-sub $3 --original --steps 1500 2>&1 | grep -E "full MCMC step|scoring|Error|error" >> $out || exit 1
done
echo "== C5, 7 original contigs (late stage)" >> $out
timeout -k 10 300 python bench.py --layout original --steps 30 --warmup 5 --mcmc-warmup 0 --no-cpu-baseline --no-late-stage --no-hbm-control 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('late stage: us/step %.1f  cand/s %.0f  k_scan %.1f us' % (1e3 * j['ms_per_step'], j['value'], 1e3 * j['roofline']['avg_launch_ms']))
" >> $out || exit 1
cat $out
