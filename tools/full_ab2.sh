#!/bin/bash
# GPU box: k_full_nnz variants (groups per lane, blocks per CU, compact records on/off): rocprofv3 kernel-trace averages
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd $REPO
OUT=gpurun_out/full_ab2.log
: > $OUT
run() {
  echo "== $*" >> $OUT
  ( cd /tmp && env TMPDIR=/tmp "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$$ -- python3 $REPO/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-late-stage --no-hbm-control > /tmp/tr.log 2>&1 )
  python3 - /tmp/tr_$$ >> $OUT <<'PY'
import sys, glob, csv, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    for key in ("k_full_nnz", "k_full_mass", "k_subrec"):
        if key in r["Kernel_Name"]:
            d[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print({k: (len(v), round(sum(v) / len(v) / 1e3, 1)) for k, v in d.items()})
PY
  rm -rf /tmp/tr_$$
}
run GRAAL_FULL_G=2
run GRAAL_FULL_G=4
run GRAAL_FULL_G=4 GRAAL_FULL_BPC=3
run GRAAL_FULL_G=1 GRAAL_FULL_BPC=6
run GRAAL_FULL_G=2 GRAAL_FULL_BPC=4
run GRAAL_FULL_G=2 GRAAL_FULL_NO_COMPACT=1
cat $OUT
