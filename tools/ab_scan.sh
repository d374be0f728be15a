#!/bin/bash
# GPU box: A/B of k_scan variants on the bench workload (200 steps, event pair every 8th step): default build, rows through
# kernel arguments (GRAAL_HOST_ROWS=1), and builds with fewer row words requested above the barrier (-DGRAAL_SCAN_PRE=0).
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd $REPO
OUT=gpurun_out/ab_scan.log
: > $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DGRAAL_SCAN_PRE=0 -o /tmp/libgraal_pre0.so graal_amd/csrc/graal_hip.hip 2>>$OUT
run() {
  echo "== $*" >> $OUT
  env "$@" timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-late-stage --no-hbm-control 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('cand/s %.0f  us/step %.1f  k_scan in-step %.2f us (frac %.3f)  back-to-back %.2f us  isolated replay %.2f us  full step %.1f us' % (j['value'], 1e3*j['ms_per_step'], 1e3*r['avg_launch_ms'], r['frac'], 1e3*r['back_to_back_replay_ms'], 1e3*r['isolated_replay_ms'], 1e3*j['full_mcmc_step_ms']))
" >> $OUT
}
run X=1
run GRAAL_HOST_ROWS=1
run GRAAL_HIP_LIB=/tmp/libgraal_pre0.so
run GRAAL_HIP_LIB=/tmp/libgraal_pre0.so GRAAL_HOST_ROWS=1
run X=2
cat $OUT
