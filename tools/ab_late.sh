#!/bin/bash
# GPU box: the late stage's scoring step (bench.py --late-only) under several library builds, interleaved.  usage: tools/ab_late.sh NAME=path.so ...
for round in 1 2; do for spec in "$@"; do
  name=${spec%%=*}; lib=${spec#*=}
  if [ "$lib" = "-" ]; then unset GRAAL_HIP_LIB; else export GRAAL_HIP_LIB=$lib; fi
  timeout -k 10 200 python bench.py --late-only --late-repeats 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); ls = j.get('late_stage', j)
        print('%-12s late ms/step %.3f   k_strict2 %.1f us' % ('$name', ls['ms_per_step'], 1e3 * ls['roofline']['avg_launch_ms']))
"
done; done
