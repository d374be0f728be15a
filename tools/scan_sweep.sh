#!/bin/bash
# GPU box: rocprofv3 kernel-trace of the bench for several scan grid sizes; prints k_scan / k_prep / k_post averages
export TMPDIR=/tmp
for cfg in "2048 256" "1024 512" "512 1024" "2048 512" "1024 1024"; do set -- $cfg; b=$1; t=$2
  OUT=/tmp/sweep_$b; rm -rf $OUT
  GRAAL_SCAN_BLOCKS=$b GRAAL_SCAN_THREADS=$t timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > /tmp/sweep_$b.log 2>&1
  python3 - $OUT "$b x $t" <<'PY'
import sys, glob, csv, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    for key in ("k_scan", "k_prep", "k_post"):
        if key in n and "lookback" not in n:
            d[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("blocks x threads", sys.argv[2], {k: round(sum(v[-100:]) / len(v[-100:]) / 1e3, 2) for k, v in d.items()})
PY
done
