#!/usr/bin/env python3
"""GPU box, debug build (-DGRAAL_STAMPS): in-kernel wall-clock stamps of MCMC steps in the MIDDLE of a run -- the C4 stand-in
(40,000 bins, 8 M contacts) after its first cycle, contigs of ~20-100 bins, reference arithmetic: where a step that needs
k_strict_cull + k_strict spends its time."""
import ctypes, os, sys, subprocess, time
import torch
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "graal_amd", "libgraal_hip_stamps.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                       "-DGRAAL_STAMPS", "-o", so, os.path.join(ROOT, "graal_amd", "csrc", "graal_hip.hip")])
from graal_amd import build
build.HIP_LIB = so
from graal_amd import lib, synth
import bench
P = synth.make_problem(n_bins=40000, nnz=8_000_000, n_sub=1, seed=2014, contig_weights=(6.8, 6.2, 5.3, 4.6, 4.0, 3.4, 2.7), mean_len_bp=660.0, accu=1)
P["S_o_A_frags"] = bench.exploded_layout(P)
rng = np.random.RandomState(1)
smp = bench.build_sampler(P, rng, None, 0)
smp.init_likelihood()
n = P["n_frags"]
order = np.arange(n, dtype=np.int32); rng.shuffle(order)
for i in order:
    smp.step_max_likelihood(int(i), 5)
L = lib.load()
L.graal_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
N = 600
S = np.zeros((N, 32)); host = np.zeros(N); C = np.zeros((N, 4))
rng.shuffle(order)
for j, i in enumerate(order[:N]):
    h0 = time.perf_counter()
    smp.step_max_likelihood(int(i), 5)
    host[j] = (time.perf_counter() - h0) * 1e6
    st = np.zeros(32, dtype=np.uint64)
    assert L.graal_debug_stamps(smp.engine._h, st.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))) == 0   # (synchronises the device)
    S[j] = st.astype(np.float64) * 0.01      # us
    C[j] = smp.engine.last_counters()
strict = S[:, 16] > S[:, 8]          # k_strict ran in this step (its stamp is younger than the step's scan start)
print("%d of %d steps needed k_strict; host time per step: those %.1f us, the others %.1f us" % (strict.sum(), N, host[strict].mean(), host[~strict].mean()))
print("queued contacts per strict step: mean %.0f median %.0f max %.0f; work units: mean %.0f median %.0f max %.0f" % (C[strict][:, 2].mean(), np.median(C[strict][:, 2]), C[strict][:, 2].max(), C[strict][:, 3].mean(), np.median(C[strict][:, 3]), C[strict][:, 3].max()))
names = [(22, "k_incr start"), (0, "k_tm start"), (8, "k_scan start"), (24, "k_tm: fA / fB records loaded"), (25, "k_tm: piece representatives loaded"), (26, "k_tm: transforms"), (1, "k_tm tables done"), (10, "k_scan block 0 loop done"), (29, "k_strict_cull start (tiled path only)"),
         (3, "k_tm: last neighbour's tables released"), (16, "k_strict[_flat] start (block 0)"), (21, "k_strict_flat: tables seen"), (17, "k_strict block 0 past its prologue"), (18, "k_strict units done (latest wave)"),
         (19, "k_strict queued contacts done (latest wave)"), (20, "k_strict published")]
ref = S[strict][:, 8]
for idx, nm in names:
    d = S[strict][:, idx] - ref
    ok = np.abs(d) < 1e5
    print("  %-46s %7.2f us (median %7.2f)" % (nm, d[ok].mean(), np.median(d[ok])))
