#!/bin/bash
# GPU box: late-stage regime (C5 on its 7 original contigs): per-step time with k_fin's mass units / queued contacts switched off
# (GRAAL_FIN_SKIP: wrong sums, timing only) and the kernel trace of the complete step.
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd $REPO
export TMPDIR=/tmp
OUT=gpurun_out/late_ab.log
: > $OUT
run() {
  echo "== $*" >> $OUT
  env "$@" timeout -k 10 200 python bench.py --layout original --steps 30 --warmup 5 --mcmc-warmup 0 --no-cpu-baseline --no-late-stage --no-hbm-control 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('late stage: us/step %.1f  cand/s %.0f  k_scan %.1f us  queued %d  mass items %d' % (1e3*j['ms_per_step'], j['value'], 1e3*r['avg_launch_ms'], j['queued_contacts_last_step'], j['mass_items_last_step']))
" >> $OUT
}
run X=1
run GRAAL_FIN_SKIP=1
run GRAAL_FIN_SKIP=2
run GRAAL_FIN_SKIP=3
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/late_trace -- python3 $REPO/bench.py --layout original --steps 30 --warmup 5 --mcmc-warmup 0 --no-cpu-baseline --no-late-stage --no-hbm-control > /tmp/late_trace.log 2>&1 )
python3 - /tmp/late_trace >> $OUT <<'PY'
import sys, glob, csv, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    for key in ("k_scan", "k_tm", "k_fin", "k_apply", "k_incr", "k_full_nnz", "k_full_mass"):
        if key in n and "lookback" not in n:
            d[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("kernel trace (count, mean us of the last 30):", {k: (len(v), round(sum(v[-30:]) / len(v[-30:]) / 1e3, 1)) for k, v in d.items()})
PY
cat $OUT
