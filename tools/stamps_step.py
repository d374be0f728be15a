#!/usr/bin/env python3
"""GPU box, debug build (-DGRAAL_STAMPS): in-kernel wall-clock stamps of FULL MCMC steps through graal_step (C5 exploded + 2,000
steps), consecutive steps chained: where the GPU timeline of a step goes, including the gaps between its kernels."""
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
P = synth.make_problem(n_bins=50000, nnz=20_000_000, n_sub=1, seed=20141217)
P["S_o_A_frags"] = bench.exploded_layout(P)
rng = np.random.RandomState(20141217)
smp = bench.build_sampler(P, rng, None, 0)
smp.init_likelihood()
order = np.arange(P["n_frags"], dtype=np.int32); rng.shuffle(order)
for i in order[:2000]:
    smp.step_max_likelihood(int(i), 5)
L = lib.load()
L.graal_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
N = 300
S = np.zeros((N, 32))
host = np.zeros((N, 2))
for j, i in enumerate(order[2000:2000 + N]):
    h0 = time.perf_counter()
    smp.step_max_likelihood(int(i), 5)
    h1 = time.perf_counter()
    st = np.zeros(32, dtype=np.uint64)
    assert L.graal_debug_stamps(smp.engine._h, st.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))) == 0   # (synchronises the device)
    S[j] = st.astype(np.float64) * 0.01      # us
    host[j] = (h0 * 1e6, h1 * 1e6)
names = [(22, "k_incr start"), (23, "k_incr end (a late block)"), (31, "statistics published"), (0, "k_tm start"), (8, "k_scan start"), (24, "k_tm: fA / fB records loaded"), (25, "k_tm: piece representatives loaded"), (26, "k_tm: transforms"), (1, "k_tm tables done"),
         (2, "k_tm mass done"), (10, "k_scan block 0 loop done"), (4, "finisher: scan seen complete"), (5, "finisher: contacts priced"), (6, "finisher: published"),
         (7, "k_apply start"), (15, "k_apply end (last block)")]
ref = S[:, 22]
print("within a step, relative to k_incr's start (mean over %d steps, each followed by a device synchronise -> no overlap between steps):" % N)
for idx, nm in names:
    d = S[:, idx] - ref
    ok = np.abs(d) < 1e5
    print("  %-34s %7.2f us" % (nm, d[ok].mean()))
print("host: graal_step call %.1f us" % (host[:, 1] - host[:, 0]).mean())
