#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd $REPO
OUT=gpurun_out/ev_ab.log
: > $OUT
run() {
  echo "== $*" >> $OUT
  env "$@" timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-late-stage --no-hbm-control 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('us/step %.1f  k_scan mean %.2f us  samples(us): %s' % (1e3*j['ms_per_step'], 1e3*r['avg_launch_ms'], [round(1e3*x,1) for x in r['launch_ms_samples']]))
" >> $OUT
}
run GRAAL_BENCH_EVENT_EVERY=4
run GRAAL_BENCH_EVENT_EVERY=4
run GRAAL_BENCH_EVENT_EVERY=8
run GRAAL_BENCH_EVENT_EVERY=2
run GRAAL_BENCH_EVENT_EVERY=1
cat $OUT
