#!/usr/bin/env python3
"""From the summaries of tools/profile_gpu.sh (profiles/rNN_rocprof_{c5,late}.json) to the two small files bench.py reads for the parts of its
`roofline` objects that come from COMMITTED profiler passes rather than from the run itself:

  profiles/traffic.json    k_scan: HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 correction) and the kernel-trace duration
  profiles/valu_late.json  k_strict2 in the late stage: wave instructions per launch (SQ_INSTS_VALU), the per-type mix, the ceiling for that mix

usage: python tools/make_roofline_json.py r05"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
P = os.path.join(ROOT, "profiles")

c5 = os.path.join(P, "%s_rocprof_c5.json" % tag)
if os.path.exists(c5):
    j = json.load(open(c5))
    scan = [k for k in j.get("kernel_trace", {}) if k.startswith("k_scan") and k.endswith("[in a step]")]
    out = {"kernel": "k_scan", "workload": "bench.py --steps 100 --warmup 10 (C5 exploded + 2,000 MCMC warm-up steps)", "n_frags": 50000, "nnz": 20000000,
           "hbm_bytes_per_launch": j.get("k_scan_hbm_bytes_per_launch"), "fetch_size_kib": j.get("k_scan_fetch_size_kib"),
           "write_size_kib": j.get("k_scan_write_size_kib"),
           "kernel_avg_us_rocprof_in_a_step": j["kernel_trace"][scan[0]]["avg_us"] if scan else None,
           "source": "profiles/%s_rocprof_c5.md (tools/profile_gpu.sh %s_c5 c5: --kernel-trace pass + separate --pmc FETCH_SIZE / WRITE_SIZE passes)" % (tag, tag)}
    json.dump(out, open(os.path.join(P, "traffic.json"), "w"), indent=1)
    print("traffic.json:", out)

late = os.path.join(P, "%s_rocprof_late.json" % tag)
if os.path.exists(late):
    j = json.load(open(late))
    v = j["valu_roofline"]["k_strict2"]
    m = j.get("valu_mix", {}).get("k_strict2")
    out = {"kernel": "k_strict2", "workload": "bench.py --late-only --late-repeats 3: C5 on its 7 original contigs, reference arithmetic, 3 warm-up + 3 x 12 scoring steps",
           "n_frags": 50000, "nnz": 20000000, "wave_instr_per_launch": v["wave_instr_per_launch"], "kernel_avg_us_rocprof": v["kernel_avg_us_rocprof"],
           "launches": v["launches"], "frac_of_6.144e11": v["frac"], "wait_frac": v["wait_frac"],
           "source": "profiles/%s_rocprof_late.md (tools/profile_gpu.sh %s_late late with PROF_VALU_MIX=1)" % (tag, tag)}
    if m:
        sq = j["pmc_sq"]["k_strict2"]
        out.update({"mix_counts_per_launch": m["counts_per_launch"], "float64_share": m["f64_share"],
                    "peak_for_the_mix_wave_instr_per_s": m["peak_for_the_mix_wave_instr_per_s"],
                    "frac_of_the_mix_ceiling_rocprof": m["frac_of_the_mix_ceiling"],
                    "SQ_ACTIVE_INST_VALU_per_launch": m["SQ_ACTIVE_INST_VALU_per_launch"],
                    # SQ_ACTIVE_INST_VALU counts quad-cycles (MI355X_MICROARCH.md): x 4 = SIMD-cycles in which a vector instruction was executing, summed
                    # over the chip's 1,024 SIMDs; divided by SQ_INSTS_VALU = cycles a SIMD spends per vector instruction of this kernel
                    "simd_cycles_per_valu_instr": 4.0 * m["SQ_ACTIVE_INST_VALU_per_launch"] / v["wave_instr_per_launch"] if m["SQ_ACTIVE_INST_VALU_per_launch"] else None,
                    "rates_source": m["rates_source"], "SQ_BUSY_CYCLES_per_launch": sq.get("SQ_BUSY_CYCLES", {}).get("mean")})
    json.dump(out, open(os.path.join(P, "valu_late.json"), "w"), indent=1)
    print("valu_late.json:", {k: out[k] for k in out if k != "mix_counts_per_launch"})
