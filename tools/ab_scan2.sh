#!/bin/bash
# GPU box: bisecting builds of k_scan (what makes it slower than tools/scan_micro's loop?)
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd $REPO
OUT=gpurun_out/ab_scan2.log
: > $OUT
B="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off"
$B -DGRAAL_EXP_NOHITS -o /tmp/lib_nohits.so graal_amd/csrc/graal_hip.hip 2>>$OUT &
wait
run() {
  echo "== $*" >> $OUT
  env "$@" timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-late-stage --no-hbm-control 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('cand/s %.0f  us/step %.1f  k_scan in-step %.2f us (frac %.3f)  back-to-back %.2f us  isolated replay %.2f us' % (j['value'], 1e3*j['ms_per_step'], 1e3*r['avg_launch_ms'], r['frac'], 1e3*r['back_to_back_replay_ms'], 1e3*r['isolated_replay_ms']))
" >> $OUT
}
run X=1
run GRAAL_SCAN_G=2
run GRAAL_SCAN_G=2 GRAAL_SCAN_BLOCKS=992
run GRAAL_HIP_LIB=/tmp/lib_nohits.so
run GRAAL_HIP_LIB=/tmp/lib_nohits.so GRAAL_SCAN_G=2
run GRAAL_HIP_LIB=/tmp/lib_nohits.so GRAAL_SCAN_G=2 GRAAL_SCAN_BLOCKS=992
cat $OUT
