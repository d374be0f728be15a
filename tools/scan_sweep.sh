#!/bin/bash
# GPU box: rocprofv3 kernel-trace of the bench for several launch configurations; prints per-kernel averages + bench value
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
i=0
while read -r cfg; do
  [ -z "$cfg" ] && continue
  i=$((i+1)); OUT=/tmp/sweep_$i; rm -rf $OUT
  ( cd /tmp && env $cfg timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $REPO/bench.py --steps 200 --warmup 20 --no-cpu-baseline > /tmp/sweep_$i.log 2>&1 )
  python3 - $OUT "$cfg" /tmp/sweep_$i.log <<'PY'
import sys, glob, csv, collections, json
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    for key in ("k_scan", "k_tm", "k_fin", "k_side"):
        if key in n and "lookback" not in n:
            d[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
val = None
for line in open(sys.argv[3]):
    if line.startswith("{"):
        j = json.loads(line); val = (round(j["value"]), round(j["ms_per_step"] * 1e3, 1), round(j["roofline"]["avg_launch_ms"] * 1e3, 2))
print(sys.argv[2], {k: round(sum(v[-200:]) / len(v[-200:]) / 1e3, 2) for k, v in d.items()}, "cand/s, us/step, replay us:", val, flush=True)
PY
done <<CFGS
X=1
GRAAL_SCAN_BLOCKS=504
GRAAL_SCAN_BLOCKS=496
GRAAL_TM_SERIAL=1
GRAAL_SCAN_BLOCKS=1008 GRAAL_SCAN_THREADS=512
GRAAL_FIN_BLOCKS=4
CFGS
