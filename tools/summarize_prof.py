#!/usr/bin/env python3
"""Reduce a rocprofv3 output tree (tools/profile_gpu.sh) to small summaries that can be committed under profiles/.

  <dir>/trace : --kernel-trace --stats      -> kernel_stats (copied) + per-kernel duration stats from the trace
  <dir>/fetch : --pmc FETCH_SIZE            -> per-kernel counter stats
  <dir>/write : --pmc WRITE_SIZE            -> per-kernel counter stats

HBM bytes per launch follow MI355X_MICROARCH.md section HBM: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 on gfx950
(FETCH_SIZE reports half of a wide coalesced streaming read; both counters are in KiB).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    if "rocprim" in name or "hipcub" in name:
        for key in ("merge_sort_block_merge", "merge_sort_block_sort", "radix_sort", "onesweep", "lookback_scan_state", "scan_impl", "histogram"):
            if key in name:
                return "rocprim::" + key
        return "rocprim::other"
    return name.split("(")[0].split("<")[0].strip() + ("<true>" if "k_scan<true>" in name else "<false>" if "k_scan<false>" in name else "")


HIST_EDGES_US = [0, 25, 50, 100, 200, 400, 800, 1600, 3200, 1e9]


def trace_stats(path, last_n=100):
    per = defaultdict(list)
    rows = []
    with open(path) as f:
        for row in csv.DictReader(f):
            rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), short(row["Kernel_Name"])))
    rows.sort()
    prev = ""
    for a, b, name in rows:
        per[name].append(b - a)
        if name.startswith("k_scan"):
            # bench.py replays the scan back to back (counting only) after the timed region; the other launches are steps
            per[name + (" [replay]" if prev.startswith("k_scan") else " [in a step]")].append(b - a)
        prev = name
    out = {}
    for k, v in per.items():
        tail = v[-last_n:]
        out[k] = {"calls": len(v), "total_ms": sum(v) / 1e6, "avg_us": sum(v) / len(v) / 1e3, "min_us": min(v) / 1e3,
                  "max_us": max(v) / 1e3, "avg_us_last_%d" % last_n: sum(tail) / len(tail) / 1e3}
        if len(v) >= 1000:   # where a kernel's time goes by the duration of its launches (a run's launches differ by orders of magnitude)
            hist = []
            for lo, hi in zip(HIST_EDGES_US[:-1], HIST_EDGES_US[1:]):
                sel = [x for x in v if lo * 1e3 <= x < hi * 1e3]
                if sel:
                    hist.append({"from_us": lo, "to_us": hi, "calls": len(sel), "total_ms": sum(sel) / 1e6})
            out[k]["by_duration"] = hist
    return out


def counter_stats(path, last_n=100):
    per = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            per[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for k, cs in per.items():
        out[k] = {}
        for c, v in cs.items():
            tail = v[-last_n:]
            out[k][c] = {"n": len(v), "mean": sum(v) / len(v), "mean_last_%d" % last_n: sum(tail) / len(tail),
                         "min": min(v), "max": max(v)}
    return out


def main():
    d, tag = sys.argv[1], sys.argv[2]
    res = {"tag": tag}
    t = find(os.path.join(d, "trace"), "*kernel_trace.csv")
    if t:
        res["kernel_trace"] = trace_stats(t)
    ks = find(os.path.join(d, "trace"), "*kernel_stats.csv")
    if ks:
        res["kernel_stats_csv"] = open(ks).read()
    for name in ("fetch", "write", "sq", "mix64", "mix32", "mixint"):
        c = find(os.path.join(d, name), "*counter_collection.csv")
        if c:
            res["pmc_" + name] = counter_stats(c)
    # HBM bytes per launch of every kernel that has both TCC counters: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 correction)
    hbm = {}
    for k, cs in res.get("pmc_fetch", {}).items():
        if k in res.get("pmc_write", {}) and "FETCH_SIZE" in cs and "WRITE_SIZE" in res["pmc_write"][k]:
            hbm[k] = (2.0 * cs["FETCH_SIZE"]["mean_last_100"] + res["pmc_write"][k]["WRITE_SIZE"]["mean_last_100"]) * 1024.0
    res["hbm_bytes_per_launch"] = hbm
    scan = [k for k in res.get("pmc_fetch", {}) if k.startswith("k_scan")]
    if scan and scan[0] in res.get("pmc_write", {}):
        k = scan[0]
        fetch = res["pmc_fetch"][k]["FETCH_SIZE"]["mean_last_100"]
        write = res["pmc_write"][k]["WRITE_SIZE"]["mean_last_100"]
        res["k_scan_hbm_bytes_per_launch"] = (2.0 * fetch + write) * 1024.0
        res["k_scan_fetch_size_kib"] = fetch
        res["k_scan_write_size_kib"] = write
    # VALU roofline of every kernel with an SQ_INSTS_VALU count: wave instructions per dispatch / kernel-trace duration against
    # 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction (bench.py: VALU_PEAK_WAVE_INSTR)
    peak = 256 * 4 * 2.4e9 / 4
    valu = {}
    for k, cs in res.get("pmc_sq", {}).items():
        if "SQ_INSTS_VALU" in cs and k in res.get("kernel_trace", {}):
            instr, us = cs["SQ_INSTS_VALU"]["mean"], res["kernel_trace"][k]["avg_us"]
            if us > 0:
                valu[k] = {"wave_instr_per_launch": instr, "kernel_avg_us_rocprof": us, "launches": cs["SQ_INSTS_VALU"]["n"],
                           "achieved_wave_instr_per_s": instr / (us * 1e-6), "peak": peak, "frac": instr / (us * 1e-6) / peak,
                           "wait_frac": (cs["SQ_WAIT_ANY"]["mean"] / cs["SQ_WAVE_CYCLES"]["mean"]) if "SQ_WAIT_ANY" in cs and "SQ_WAVE_CYCLES" in cs and cs["SQ_WAVE_CYCLES"]["mean"] > 0 else None}
    res["valu_roofline"] = valu
    # The ceiling for the kernel's instruction MIX.  Sustained chip-wide issue rates by instruction class, wave instructions per second at 4 waves
    # per SIMD (what k_strict2 runs), measured by tools/valu_issue_micro.hip on this pool (profiles/r05_valu_issue.log: the clock the chip holds under
    # such a stream -- 1.6-2.1 GHz, not the 2.4 GHz of the data sheet -- is in these rates).  The remainder of SQ_INSTS_VALU that no per-type counter
    # claims (moves, compares, selects, lane reads) and all of INT32 are priced at the fastest class: the ceiling errs high, the fraction low.
    RATE = {"ADD_F64": 0.5265e12, "MUL_F64": 0.5240e12, "FMA_F64": 0.4987e12, "TRANS_F64": 0.1509e12, "ADD_F32": 0.9369e12, "MUL_F32": 0.9307e12,
            "FMA_F32": 0.5857e12, "TRANS_F32": 0.2984e12, "INT32": 1.0266e12, "INT64": 0.5588e12, "CVT": 0.5588e12, "OTHER": 1.0266e12}
    mix = {}
    for k, v in valu.items():
        counts = {}
        for grp in ("pmc_mix64", "pmc_mix32", "pmc_mixint"):
            for c, st in res.get(grp, {}).get(k, {}).items():
                if c.startswith("SQ_INSTS_VALU_"):
                    counts[c[len("SQ_INSTS_VALU_"):]] = st["mean"]
        if len(counts) < 11:
            continue
        total = v["wave_instr_per_launch"]
        other = max(0.0, total - sum(counts.values()))
        secs = sum(n / RATE[c] for c, n in counts.items()) + other / RATE["OTHER"]
        peak_mix = total / secs if secs > 0 else None
        act = res.get("pmc_mixint", {}).get(k, {}).get("SQ_ACTIVE_INST_VALU", {}).get("mean")
        mix[k] = {"counts_per_launch": dict(counts, OTHER=other), "wave_instr_per_launch": total, "ceiling_s_per_launch": secs,
                  "peak_for_the_mix_wave_instr_per_s": peak_mix, "achieved_wave_instr_per_s": v["achieved_wave_instr_per_s"],
                  "frac_of_the_mix_ceiling": v["achieved_wave_instr_per_s"] / peak_mix if peak_mix else None,
                  "f64_share": sum(counts.get(c, 0.0) for c in ("ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64")) / total if total else None,
                  "SQ_ACTIVE_INST_VALU_per_launch": act, "rates_source": "tools/valu_issue_micro.hip at 4 waves per SIMD (profiles/r05_valu_issue.log)"}
    res["valu_mix"] = mix
    with open(os.path.join(d, "summary_%s.json" % tag), "w") as f:
        json.dump(res, f, indent=1)
    # human readable
    with open(os.path.join(d, "summary_%s.md" % tag), "w") as f:
        f.write("# rocprofv3 summary %s\n\n## kernel trace (durations from --kernel-trace)\n\n" % tag)
        f.write("| kernel | calls | total ms | avg us | min us | max us | avg us (last 100) |\n|---|---|---|---|---|---|---|\n")
        for k, v in sorted(res.get("kernel_trace", {}).items(), key=lambda kv: -kv[1]["total_ms"]):
            f.write("| %s | %d | %.3f | %.2f | %.2f | %.2f | %.2f |\n" % (k[:70], v["calls"], v["total_ms"], v["avg_us"], v["min_us"],
                                                                  v["max_us"], v["avg_us_last_100"]))
        top = sorted(res.get("kernel_trace", {}).items(), key=lambda kv: -kv[1]["total_ms"])
        for k, v in top[:3]:
            if v.get("by_duration"):
                f.write("\n## %s: launches and time by launch duration\n\n| duration (us) | launches | total ms | share of the kernel's time |\n|---|---|---|---|\n" % k[:70])
                for h in v["by_duration"]:
                    f.write("| %g - %s | %d | %.1f | %.1f %% |\n" % (h["from_us"], ("%g" % h["to_us"]) if h["to_us"] < 1e8 else "", h["calls"], h["total_ms"],
                                                                 100.0 * h["total_ms"] / v["total_ms"]))
        for name in ("fetch", "write", "sq", "mix64", "mix32", "mixint"):
            if "pmc_" + name in res:
                f.write("\n## PMC pass: %s (KiB per dispatch for the TCC counters)\n\n| kernel | counter | n | mean | mean last 100 | min | max |\n|---|---|---|---|---|---|---|\n" % name)
                for k, cs in res["pmc_" + name].items():
                    for c, v in cs.items():
                        f.write("| %s | %s | %d | %.1f | %.1f | %.1f | %.1f |\n" % (k[:70], c, v["n"], v["mean"], v["mean_last_100"], v["min"], v["max"]))
        if res.get("hbm_bytes_per_launch"):
            f.write("\n## HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (mean of the last 100 launches)\n\n| kernel | MB |\n|---|---|\n")
            for k, v in sorted(res["hbm_bytes_per_launch"].items(), key=lambda kv: -kv[1]):
                f.write("| %s | %.3f |\n" % (k[:70], v / 1e6))
        if res.get("valu_roofline"):
            f.write("\n## VALU roofline = SQ_INSTS_VALU per dispatch / kernel-trace duration / (1,024 SIMDs x 2.4 GHz / 4) (mean over all launches)\n\n"
                    "| kernel | launches | wave instr / launch | avg us | wave instr / s | fraction of 6.14e11 | SQ_WAIT_ANY / SQ_WAVE_CYCLES |\n|---|---|---|---|---|---|---|\n")
            for k, v in sorted(res["valu_roofline"].items(), key=lambda kv: -kv[1]["wave_instr_per_launch"] * kv[1]["launches"]):
                f.write("| %s | %d | %.4g | %.2f | %.3g | %.3f | %s |\n" % (k[:70], v["launches"], v["wave_instr_per_launch"], v["kernel_avg_us_rocprof"],
                                                                        v["achieved_wave_instr_per_s"], v["frac"], "%.2f" % v["wait_frac"] if v["wait_frac"] is not None else "-"))
        if res.get("valu_mix"):
            f.write("\n## VALU ceiling for the instruction MIX (per-type SQ_INSTS_VALU_* counters x the sustained issue rates of tools/valu_issue_micro.hip, 4 waves per SIMD)\n\n"
                    "| kernel | wave instr / launch | float64 share | ADD/MUL/FMA/TRANS F64 | ADD/MUL/FMA/TRANS F32 | INT32 / INT64 / CVT / other | ceiling for the mix (wave instr / s) | achieved | fraction |\n|---|---|---|---|---|---|---|---|---|\n")
            for k, v in sorted(res["valu_mix"].items(), key=lambda kv: -kv[1]["wave_instr_per_launch"]):
                c = v["counts_per_launch"]
                f.write("| %s | %.4g | %.2f | %.3g / %.3g / %.3g / %.3g | %.3g / %.3g / %.3g / %.3g | %.3g / %.3g / %.3g / %.3g | %.3g | %.3g | %.3f |\n" % (
                    k[:60], v["wave_instr_per_launch"], v["f64_share"], c["ADD_F64"], c["MUL_F64"], c["FMA_F64"], c["TRANS_F64"], c["ADD_F32"], c["MUL_F32"],
                    c["FMA_F32"], c["TRANS_F32"], c["INT32"], c["INT64"], c["CVT"], c["OTHER"], v["peak_for_the_mix_wave_instr_per_s"],
                    v["achieved_wave_instr_per_s"], v["frac_of_the_mix_ceiling"]))
        if "k_scan_hbm_bytes_per_launch" in res:
            f.write("\nk_scan HBM bytes / launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 = %.0f\n" % res["k_scan_hbm_bytes_per_launch"])
        if "kernel_stats_csv" in res:
            f.write("\n## rocprofv3 --stats (kernel_stats.csv)\n\n```\n%s```\n" % res["kernel_stats_csv"])
    print(open(os.path.join(d, "summary_%s.md" % tag)).read()[:6000])


if __name__ == "__main__":
    main()
