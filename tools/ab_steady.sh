#!/bin/bash
# GPU box: A/B of library builds / switches on full MCMC steps at the late-stage state of the C2 / C3 stand-ins (7 contigs), interleaved.
# usage: tools/ab_steady.sh "NAME ENV=VAL ..." ...
for round in 1 2; do for spec in "$@"; do
  name=${spec%% *}; envs=${spec#* }; [ "$envs" = "$spec" ] && envs="X=1"
  for shape in "1086 120000" "3500 600000"; do set -- $shape "$@"; nb=$1; nnz=$2; shift 2
    env $envs timeout -k 10 300 python tools/step_breakdown.py --n-bins $nb --nnz $nnz --n-sub 3 --original --steps 1500 2>&1 | grep 'full MCMC step' | sed "s/^/$name /" | cut -c1-70
  done
done; done
