#!/bin/bash
# GPU box: the bench workload's scoring steps under several library builds, interleaved.  usage: tools/ab_bench.sh NAME=path.so|- ...
for round in 1 2 3; do for spec in "$@"; do
  name=${spec%%=*}; lib=${spec#*=}
  if [ "$lib" = "-" ]; then unset GRAAL_HIP_LIB; else export GRAAL_HIP_LIB=$lib; fi
  timeout -k 10 300 python bench.py --steps 300 --warmup 20 --long-steps 1000 --no-cpu-baseline --no-late-stage --no-hbm-control 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('%-8s %.3f M (%.1f us/step), 1000 steps %.3f M, exact %.3f M, full step %.1f us' % ('$name', j['value'] / 1e6, 1e3 * j['ms_per_step'], j['value_1000'] / 1e6, j['other_arithmetic']['value'] / 1e6, 1e3 * j['full_mcmc_step_ms']))
        break
"
done; done
