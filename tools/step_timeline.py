#!/usr/bin/env python3
"""Per-step GPU timeline from a rocprofv3 --kernel-trace csv of bench.py: where a scoring step's time goes."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    for key in ("k_tm", "k_scan", "k_fin"):
        if key in n:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), key))
ev.sort()
rep = [e[1] - e[0] for i, e in enumerate(ev) if e[2] == "k_scan" and i > 0 and ev[i - 1][2] == "k_scan"]
if rep:
    print("k_scan replays (back-to-back, dry): n %d avg %.2f us" % (len(rep), sum(rep) / len(rep) / 1e3))
steps = []
i = 0
while i + 2 < len(ev):
    names = sorted(e[2] for e in ev[i:i + 3])
    if names == ["k_fin", "k_scan", "k_tm"]:
        d = {e[2]: e for e in ev[i:i + 3]}
        steps.append(d); i += 3
    elif sorted(e[2] for e in ev[i:i + 2]) == ["k_scan", "k_tm"] and (i + 2 >= len(ev) or ev[i + 2][2] != "k_scan"):
        d = {e[2]: e for e in ev[i:i + 2]}
        d["k_fin"] = (d["k_tm"][1], d["k_tm"][1], "k_fin")  # finished by k_tm's last block: no k_fin launch
        steps.append(d); i += 2
    else:
        i += 1
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 2030
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 2280
steps = steps[lo:hi]  # default: inside the timed scoring region of bench.py (after the 2000 MCMC warm-up steps)
def avg(fn):
    v = [fn(s) for s in steps]
    return sum(v) / len(v) / 1e3
print("steps", len(steps))
print("k_tm dur %.2f  k_scan dur %.2f  k_fin dur %.2f" % (avg(lambda s: s["k_tm"][1] - s["k_tm"][0]), avg(lambda s: s["k_scan"][1] - s["k_scan"][0]), avg(lambda s: s["k_fin"][1] - s["k_fin"][0])))
print("scan start - tm start %.2f   fin start - scan end %.2f   fin end - tm end %.2f" % (avg(lambda s: s["k_scan"][0] - s["k_tm"][0]), avg(lambda s: s["k_fin"][0] - s["k_scan"][1]), avg(lambda s: s["k_fin"][1] - s["k_tm"][1])))
print("gpu span (first start .. fin end) %.2f" % avg(lambda s: s["k_fin"][1] - min(s["k_tm"][0], s["k_scan"][0])))
per = [steps[j + 1]["k_tm"][0] - steps[j]["k_tm"][0] for j in range(len(steps) - 1)]
per = [p for p in per if p < 1e6]
print("period %.2f   idle between steps (next first start - fin end) %.2f" % (sum(per) / len(per) / 1e3, sum(min(steps[j + 1]["k_tm"][0], steps[j + 1]["k_scan"][0]) - steps[j]["k_fin"][1] for j in range(len(steps) - 1)) / (len(steps) - 1) / 1e3))
