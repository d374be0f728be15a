#!/bin/bash
out=gpurun_out/late_blocks.log; : > $out
for cfg in "2048 0" "768 0" "1024 0" "2048 64"; do
set -- $cfg
echo "== GRAAL_FIN_BLOCKS=$1 GRAAL_FIN_SEG=$2" >> $out
GRAAL_FIN_BLOCKS=$1 GRAAL_FIN_SEG=$2 timeout -k 10 300 python bench.py --layout original --steps 30 --warmup 5 --mcmc-warmup 0 --no-cpu-baseline --no-late-stage --no-hbm-control 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('late stage: us/step %.1f  cand/s %.0f  k_scan %.1f us  mass items %d' % (1e3 * j['ms_per_step'], j['value'], 1e3 * j['roofline']['avg_launch_ms'], j['mass_items_last_step']))
" >> $out || exit 1
done
cat $out
