#!/usr/bin/env python3
"""GPU box, debug build: how long the third stage of k_scan (doubly-affected contacts) takes in the waves that run it, and when the
last of them ends relative to the scan's start / the last block's loop end -- per scoring step of the bench workload."""
import ctypes, os, sys, subprocess
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
max_id = smp.modify_gl_cuda_buffer(0)
L = lib.load()
u64p = ctypes.POINTER(ctypes.c_uint64)
L.graal_debug_stamps.argtypes = [ctypes.c_void_p, u64p]
L.graal_debug_block_stamps.argtypes = [ctypes.c_void_p, u64p]
L.graal_debug_hitstat.argtypes = [ctypes.c_void_p, u64p, ctypes.c_int]
hs = np.zeros(8, dtype=np.uint64)
L.graal_debug_hitstat(smp.engine._h, hs.ctypes.data_as(u64p), 1)
rows = []
for f in rng.randint(0, P["n_frags"], size=200):
    nb = smp.return_neighbours(int(f), 5); nb.sort()
    smp._candidate_deltas(int(f), nb, max_id)
    st = np.zeros(32, dtype=np.uint64); L.graal_debug_stamps(smp.engine._h, st.ctypes.data_as(u64p))
    bs = np.zeros(4096 * 4, dtype=np.uint64); L.graal_debug_block_stamps(smp.engine._h, bs.ctypes.data_as(u64p))
    L.graal_debug_hitstat(smp.engine._h, hs.ctypes.data_as(u64p), 1)
    b = bs.reshape(4096, 4)[:496].astype(np.float64)
    t0 = b[:, 0].min()
    loop_end = b[:, 2].max()
    rows.append(((loop_end - t0) * 0.01, float(hs[0]) * 0.01, (float(hs[1]) - t0) * 0.01 if hs[1] else np.nan, int(hs[2]), float(hs[3]) * 0.01 / max(int(hs[2]), 1)))
r = np.array(rows)
print("per step (200 steps), us after the first scan block started:")
print("  last block's loop done (wave 0 of each block): mean %.2f" % np.nanmean(r[:, 0]))
print("  third stage: waves per step %.1f, mean duration %.2f us, longest %.2f us (mean over steps of the max), last one ends at %.2f" % (
    r[:, 3].mean(), np.nanmean(r[:, 4]), np.nanmean(r[:, 1]), np.nanmean(r[:, 2])))
print("  steps in which the last third stage ends after every block's wave-0 loop end: %d of %d" % (int(np.sum(r[:, 2] > r[:, 0])), len(r)))
