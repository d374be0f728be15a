#!/bin/bash
# GPU box: the C4 stand-in (explode + 2 cycles, reference arithmetic) under k_strict launch variants.  usage: tools/c4_sweep.sh "ENV=VAL,ENV=VAL" ...
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd $REPO
OUT=gpurun_out/c4_sweep.log
: > $OUT
for spec in "$@"; do
  E=$(echo $spec | tr ',' ' ')
  [ "$spec" = "base" ] && E=""
  echo "== $spec" >> $OUT
  env $E timeout -k 10 300 python tools/run_configs.py ${C4_WHAT:-C4} --cycles ${C4_CYCLES:-2} 2>/dev/null | sed -e 's/.*MCMC steps in/   steps in/' -e 's/carried.*//' >> $OUT
done
cat $OUT
