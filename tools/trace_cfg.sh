#!/bin/bash
# GPU box: rocprofv3 kernel trace of a tools/run_configs.py run; prints the kernels by total time.  Usage: tools/trace_cfg.sh <tag> <args to run_configs.py ...>
TAG=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp
PYTHONPATH=$R timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/run_configs.py "$@" > $OUT/run.log 2>&1
cd $R
tail -1 $OUT/run.log | cut -c1-200
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:12]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print("%-40s calls %7s  avg %9.1f us  total %8.1f ms  %5.1f %%" % (n[:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
find $OUT -name "*.csv" ! -name "*kernel_stats*" -delete
