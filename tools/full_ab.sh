#!/bin/bash
# GPU box: full-evaluation kernels (k_full_nnz variants) on the C5 state: wall time of graal_eval_full_q
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd $REPO
OUT=gpurun_out/full_ab.log
: > $OUT
run() {
  echo "== $*" >> $OUT
  env "$@" timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-late-stage --no-hbm-control 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('full_eval %.1f us  full step %.1f us  with sample_param %.1f us' % (1e3*j['full_eval_ms'], 1e3*j['full_mcmc_step_ms'], 1e3*j['full_mcmc_step_sample_param_ms']))
" >> $OUT
}
run GRAAL_FULL_G=2
run GRAAL_FULL_G=1
run GRAAL_FULL_G=1 GRAAL_FULL_BPC=16
run GRAAL_FULL_G=4
run GRAAL_FULL_G=2 GRAAL_FULL_BPC=4
cat $OUT
