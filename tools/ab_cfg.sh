#!/bin/bash
# GPU box: A/B of environment switches / library builds on the BASELINE stand-ins, same box, interleaved.
# usage: tools/ab_cfg.sh CFG CYCLES "NAME ENV=VAL ..." "NAME ENV=VAL ..."
CFG=$1; CYC=$2; shift 2
for round in 1 2; do for spec in "$@"; do
  name=${spec%% *}; envs=${spec#* }; [ "$envs" = "$spec" ] && envs="X=1"
  env $envs timeout -k 10 400 python tools/run_configs.py $CFG --cycles $CYC 2>&1 | grep 'us/step' | sed "s/^/$name /" | sed 's/: .*steps in/:/' | cut -c1-80
done; done
