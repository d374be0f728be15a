#!/usr/bin/env python3
"""Average GPU timeline of a full MCMC step from a rocprofv3 --kernel-trace CSV: for the last N steps (a step = the kernels
between two k_apply launches), start / end of every kernel relative to the end of the previous k_apply.
Usage: python tools/timeline.py <dir with *_kernel_trace.csv> [n_steps]"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 500
path = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm))
rows.sort()
steps, cur = [], []
for a, b, nm in rows:
    cur.append((a, b, nm))
    if nm == "k_apply":
        steps.append(cur); cur = []
steps = [s for s in steps[1:] if len(s) >= 4][-n_last:]
acc = defaultdict(lambda: [0.0, 0.0, 0])
period = []
for i in range(1, len(steps)):
    t0 = steps[i - 1][-1][1]          # end of the previous k_apply
    for a, b, nm in steps[i]:
        e = acc[nm]; e[0] += (a - t0) / 1e3; e[1] += (b - t0) / 1e3; e[2] += 1
    period.append((steps[i][-1][1] - t0) / 1e3)
print("steps analysed: %d; mean period (end of k_apply -> end of next k_apply): %.1f us" % (len(period), sum(period) / len(period)))
for nm, (s, e, c) in sorted(acc.items(), key=lambda kv: kv[1][0] / kv[1][2]):
    print("  %-16s calls/step %.2f  start %6.1f  end %6.1f  (dur %5.1f us)" % (nm, c / len(period), s / c, e / c, (e - s) / c))
