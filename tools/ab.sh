#!/bin/bash
# GPU box: A/B of engine variants on the bench workload.  Each argument is  NAME[:ENV=VAL,ENV=VAL][:-DFLAG -DFLAG]  -- a variant with
# compile flags is built into /tmp/lib_NAME.so first.  Two rounds, so that drift of the box shows.
# usage: tools/ab.sh base "prio::-DGRAAL_SCAN_SETPRIO" "b480:GRAAL_SCAN_BLOCKS=480"
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd $REPO
OUT=gpurun_out/ab.log
: > $OUT
for spec in "$@"; do
  name=${spec%%:*}; rest=${spec#*:}; [ "$rest" = "$spec" ] && rest=":"
  flags=${rest#*:}; [ "$flags" = "$rest" ] && flags=""
  if [ -n "$flags" ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off $flags -o /tmp/lib_$name.so graal_amd/csrc/graal_hip.hip 2>>$OUT || echo "build of $name failed" >> $OUT; fi
done
for round in 1 2; do for spec in "$@"; do
  name=${spec%%:*}; rest=${spec#*:}; [ "$rest" = "$spec" ] && rest=":"
  envs=${rest%%:*}; flags=${rest#*:}; [ "$flags" = "$rest" ] && flags=""
  E=""; [ -n "$envs" ] && E=$(echo $envs | tr ',' ' ')
  [ -n "$flags" ] && E="$E GRAAL_HIP_LIB=/tmp/lib_$name.so"
  env $E timeout -k 10 300 python bench.py --steps ${AB_STEPS:-300} --warmup 20 --long-steps 0 --no-cpu-baseline ${AB_LATE:---no-late-stage} --no-hbm-control 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('%-10s cand/s %.0f  us/step %.1f  k_scan in-step median %.2f us (frac %.3f, mean %.2f)  back-to-back %.2f  isolated %.2f  full step %.1f us  exact %.1f us/step' % ('$name', j['value'], 1e3*j['ms_per_step'], 1e3*r['launch_ms_median'], r['frac'], 1e3*r['launch_ms_mean_all_samples'], 1e3*r['back_to_back_replay_ms'], 1e3*r['isolated_replay_ms'], 1e3*j['full_mcmc_step_ms'], 1e3*j['other_arithmetic']['ms_per_step']))
        ls = j.get('late_stage')
        if ls: print('%-10s late stage: strict %.2f ms/step, exact %.2f, full step %.2f, full eval %.3f | exploded full eval %.4f ms' % ('$name', ls['ms_per_step'], ls['other_arithmetic']['ms_per_step'], ls['full_mcmc_step_ms'], ls['full_eval_ms'], j['full_eval_ms']))
" >> $OUT
done; done
cat $OUT
