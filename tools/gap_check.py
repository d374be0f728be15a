#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace csv: when does k_strict2 start relative to the end of the kernels it follows (k_scan in its stream,
k_gprep through the event)?  usage: gap_check.py <trace dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    for key in ("k_tm", "k_scan", "k_gprep", "k_strict2", "k_apply", "k_incr"):
        if key in n:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), key)); break
ev.sort()
last = {}
g_scan, g_prep, g_tm_prep, d = [], [], [], {}
for s, e, k in ev:
    if k == "k_strict2" and "k_scan" in last and "k_gprep" in last:
        g_scan.append(s - last["k_scan"][1]); g_prep.append(s - last["k_gprep"][1])
    if k == "k_gprep" and "k_tm" in last:
        g_tm_prep.append(s - last["k_tm"][1])
    last[k] = (s, e)
    d.setdefault(k, []).append(e - s)
def m(v): v = sorted(v); return sum(v) / max(1, len(v)) / 1e3, v[len(v) // 2] / 1e3 if v else 0.0
print("k_strict2 start - k_scan end:  mean %.2f us, median %.2f   (n %d)" % (*m(g_scan), len(g_scan)))
print("k_strict2 start - k_gprep end: mean %.2f us, median %.2f" % m(g_prep))
print("k_gprep start - k_tm end:      mean %.2f us, median %.2f" % m(g_tm_prep))
for k, v in d.items():
    print("%-10s duration mean %.2f us median %.2f (n %d)" % (k, *m(v), len(v)))
