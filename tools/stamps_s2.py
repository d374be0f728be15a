#!/usr/bin/env python3
"""GPU box, debug build (-DGRAAL_STAMPS): in-kernel timeline and work counters of the reference-arithmetic kernels over the union set
(k_gprep, k_strict2) on the C2 / C3 stand-ins' 7 contigs or on C5's (STAMPS_SHAPE=c2|c3|c5late), averaged over proposals."""
import ctypes, os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "graal_amd", "libgraal_hip_stamps.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                       "-DGRAAL_STAMPS"] + (["-DGRAAL_S2_COUNTS"] if os.environ.get("STAMPS_COUNTS") else []) + ["-o", so, os.path.join(ROOT, "graal_amd", "csrc", "graal_hip.hip")])
from graal_amd import build
build.HIP_LIB = so
from graal_amd import lib, synth
import bench
shape = os.environ.get("STAMPS_SHAPE", "c2")
if shape in ("c2", "c3"):
    P = synth.make_problem(n_bins=3500 if shape == "c3" else 1086, nnz=600000 if shape == "c3" else 120000, n_sub=3, seed=2014,
                           contig_weights=(6.8, 6.2, 5.3, 4.6, 4.0, 3.4, 2.7), mean_len_bp=660.0 * 27 / 3, accu=9)
    K = 3
elif shape == "c4":   # the C4 stand-in after some cycles from the exploded genome (STAMPS_CYCLES, default 3): contigs of tens to hundreds of bins
    P = synth.make_problem(n_bins=40000, nnz=8_000_000, n_sub=1, seed=2014, contig_weights=(6.8, 6.2, 5.3, 4.6, 4.0, 3.4, 2.7), mean_len_bp=660.0, accu=1)
    K = 5
else:
    P = synth.make_problem(n_bins=50000, nnz=20_000_000, n_sub=1, seed=20141217)
    K = 5
NB = P["n_frags"]
rng = np.random.RandomState(20141217)
smp = bench.build_sampler(P, rng, None, 0, "strict")
if shape == "c4":
    from graal_amd import em
    em.run_em(smp, int(os.environ.get("STAMPS_CYCLES", 3)), K, rng=rng)
smp.init_likelihood()
max_id = smp.modify_gl_cuda_buffer(0)
L = lib.load()
L.graal_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
L.graal_debug_hitstat.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
names = {0: "k_tm start", 3: "k_tm released its tables", 8: "k_scan start", 10: "k_scan block 0 loop done", 11: "k_gprep start", 12: "k_gprep: union set built", 13: "k_gprep: classes done (last wave)", 21: "k_gprep: ends and transforms loaded", 22: "k_gprep: geometry built (thread 0)", 23: "k_strict2: block 0 past its wait for k_gprep's word",
         24: "k_gprep: completion word stored", 25: "k_gprep: last block started", 28: "k_strict2: last unit's fragments and classes loaded", 29: "k_strict2: last unit's current-layout pass done", 26: "k_gprep: last class wave has its keys",
         14: "k_gprep: unit list done (last block)", 16: "k_strict2 start", 17: "k_strict2 prologue done", 18: "k_strict2 units done (last wave)", 19: "k_strict2 contacts done (last wave)",
         20: "k_strict2 sums handed out"}
acc = np.zeros(32); cnt = np.zeros(32); hs = np.zeros(8); n = 0
blk_mean = np.zeros(4); blk_max = np.zeros(4); blk_n = np.zeros(4)
L.graal_debug_block_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
import time
smp.engine.set_timing(0)
wall = 0.0
for f in rng.randint(0, NB, size=(12 if shape == "c5late" else (400 if shape == "c4" else 120))):
    nb = smp.return_neighbours(int(f), K); nb.sort()
    z = np.zeros(8, dtype=np.uint64)
    L.graal_debug_hitstat(smp.engine._h, z.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), 1)
    t0w = time.perf_counter()
    smp._candidate_deltas(int(f), nb, max_id)
    wall += time.perf_counter() - t0w
    st = np.zeros(32, dtype=np.uint64)
    assert L.graal_debug_stamps(smp.engine._h, st.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))) == 0
    L.graal_debug_hitstat(smp.engine._h, z.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), 1)
    st = st.astype(np.float64)
    if st[16] == 0:
        continue
    bk = np.zeros(4096 * 4, dtype=np.uint64)
    assert L.graal_debug_block_stamps(smp.engine._h, bk.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))) == 0
    bk = bk.reshape(4096, 4)[2048:].astype(np.float64)
    live = bk[:, 0] >= min(st[0], st[8])        # the blocks of THIS step's k_strict2
    if live.any():
        d_ = (bk[live] - min(st[0], st[8])) * 0.01
        ok_ = bk[live] > 0
        for j in range(4):
            col = d_[:, j][ok_[:, j] & (d_[:, j] > 0) & (d_[:, j] < 1e6)]
            if len(col):
                blk_mean[j] += col.mean(); blk_max[j] += col.max(); blk_n[j] += 1
    t0 = min(st[0], st[8])
    d = (st - t0) * 0.01
    ok = (st > 0) & (np.abs(d) < 1e6)
    acc[ok] += d[ok]; cnt[ok] += 1; n += 1
    zz = z.astype(np.float64)
    hs[5] += zz[5]; hs[6] += zz[6]; hs[4] += zz[7] // 1000000; hs[3] += (zz[7] % 1000000) // 100; hs[2] += zz[7] % 100
a = acc / np.maximum(cnt, 1)
print("shape %s: %d proposals through k_strict2, %.1f us per scoring call (host clock)" % (shape, n, 1e6 * wall / max(n, 1)))
for i in sorted(names):
    if cnt[i] > 0:
        print("%-44s %8.2f us" % (names[i], a[i]))
for j, nm in enumerate(("past the prologue", "wave 0, first unit: list entries read", "wave 0, first unit: fragments and class records in", "wave 0, first unit: at the class loop")):
    if blk_n[j] > 0:
        print("k_strict2 blocks, %-52s mean %8.2f us   last %8.2f us" % (nm, blk_mean[j] / blk_n[j], blk_max[j] / blk_n[j]))
print("per step: list entries %.0f, merged x%.2f, shared by %.2f waves, (unit, layout) passes %.0f, fragment pairs priced in them %.3g" % (hs[4] / n, hs[3] / n, hs[2] / n, hs[5] / n, hs[6] / n))
