#!/bin/bash
# GPU box: rocprofv3 kernel-trace of the bench (200 steps), per-kernel averages of the last launches.  Usage: tools/trace_only.sh <tag> [env...]
TAG=$1; shift
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=/tmp/trace_$TAG; rm -rf $OUT
( cd /tmp && env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $REPO/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-late-stage --no-hbm-control > /tmp/trace_$TAG.log 2>&1 )
python3 - $OUT "$TAG $*" /tmp/trace_$TAG.log <<'PY'
import sys, glob, csv, collections, json
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    for key in ("k_scan", "k_tm", "k_fin", "k_apply", "k_incr", "k_full_nnz", "k_full_mass", "k_subrec"):
        if key in n and "lookback" not in n:
            d[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
val = None
for line in open(sys.argv[3]):
    if line.startswith("{"):
        j = json.loads(line); val = (round(j["value"]), round(j["ms_per_step"] * 1e3, 1), round(j["roofline"]["avg_launch_ms"] * 1e3, 2))
print(sys.argv[2], {k: (len(v), round(sum(v[-400:]) / len(v[-400:]) / 1e3, 2)) for k, v in d.items()}, "cand/s, us/step, event us:", val, flush=True)
PY
