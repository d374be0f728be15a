#!/bin/bash
# GPU box: kernel trace of the C4 stand-in run (tools/run_configs.py C4): how many steps need k_fin, kernel durations
REPO=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/c4_trace -- python3 $REPO/tools/run_configs.py C4 > /tmp/c4_trace.log 2>&1 )
tail -2 /tmp/c4_trace.log
python3 - /tmp/c4_trace <<'PY'
import sys, glob, csv, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    for key in ("k_scan", "k_tm", "k_fin", "k_apply", "k_incr", "k_full_nnz", "k_full_mass", "k_stats"):
        if key in n and "lookback" not in n:
            d[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in d.items():
    h = len(v) // 2
    print("%-12s calls %6d  mean %7.1f us  (first half %7.1f, second half %7.1f)" % (k, len(v), sum(v) / len(v) / 1e3, sum(v[:h]) / max(h, 1) / 1e3, sum(v[h:]) / max(len(v) - h, 1) / 1e3))
PY
