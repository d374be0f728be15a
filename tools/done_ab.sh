#!/bin/bash
# completion counters of k_scan: one word against N words on lines of their own
out=gpurun_out/done_ab.log; : > $out
for n in 1 16 8 4 16 1; do
  echo "== GRAAL_SCAN_DONE_N=$n" >> $out
  GRAAL_SCAN_DONE_N=$n timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-late-stage --no-hbm-control 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('us/step %.1f  cand/s %.0f  k_scan %.2f us frac %.3f full_step %.1f us' % (1e3*j['ms_per_step'], j['value'], 1e3*r['avg_launch_ms'], r['frac'], 1e3*j['full_mcmc_step_ms']))
" >> $out || exit 1
done
cat $out
