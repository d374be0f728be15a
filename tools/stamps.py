#!/usr/bin/env python3
"""GPU box, debug build (-DGRAAL_STAMPS): in-kernel wall-clock stamps of one scoring step, averaged over many steps."""
import ctypes, os, sys, subprocess
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
C3 = os.environ.get("STAMPS_SHAPE") == "c3"
C2 = os.environ.get("STAMPS_SHAPE") == "c2" or C3   # the C2 / C3 stand-ins with their 7 contigs (k_fin finishes every step) instead of C5 exploded
if C2:
    P = synth.make_problem(n_bins=3500 if C3 else 1086, nnz=600000 if C3 else 120000, n_sub=3, seed=2014, contig_weights=(6.8, 6.2, 5.3, 4.6, 4.0, 3.4, 2.7),
                           mean_len_bp=660.0 * 27 / 3, accu=9)
else:
    P = synth.make_problem(n_bins=50000, nnz=20_000_000, n_sub=1, seed=20141217)
    if os.environ.get("STAMPS_SHAPE") != "c5late":   # c5late: the 7 original contigs
        P["S_o_A_frags"] = bench.exploded_layout(P)
NB = P["n_frags"]
rng = np.random.RandomState(20141217)
smp = bench.build_sampler(P, rng, None, 0)
smp.init_likelihood()
order = np.arange(NB, dtype=np.int32); rng.shuffle(order)
order = np.concatenate([order] * (1 + 2000 // NB))
LATE = os.environ.get("STAMPS_SHAPE") == "c5late"
for i in order[:(20 if LATE else 2000)]:
    smp.step_max_likelihood(int(i), 5)
max_id = smp.modify_gl_cuda_buffer(0)
L = lib.load()
L.graal_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
L.graal_debug_block_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
blk_acc = np.zeros(6)
acc = np.zeros(32); n = 0; cnt32 = np.zeros(32)
for f in rng.randint(0, NB, size=(40 if LATE else 300)):
    nb = smp.return_neighbours(int(f), 5); nb.sort()
    smp._candidate_deltas(int(f), nb, max_id)
    st = np.zeros(32, dtype=np.uint64)
    assert L.graal_debug_stamps(smp.engine._h, st.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))) == 0
    st = st.astype(np.float64)
    t0 = min(st[0], st[8])
    st[st == 0] = np.nan
    d = (st - t0) * 0.01; ok = np.isfinite(d) & (np.abs(d) < 1e6); acc[ok] += d[ok]; cnt32[ok] += 1; n += 1   # 100 MHz -> us
    bs = np.zeros(4096 * 4, dtype=np.uint64)
    assert L.graal_debug_block_stamps(smp.engine._h, bs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))) == 0
    nblk = int(os.environ.get("GRAAL_SCAN_BLOCKS", 512))
    fb = bs.reshape(4096, 4)[2048:2048 + (2048 if LATE else 768)].astype(np.float64)
    if (C2 or LATE) and n == (30 if LATE else 250) and fb[:, 0].max() > 0:   # k_fin's blocks: start, unit list built, wave 0 done, whole block done
        fb = (fb - t0) * 0.01
        for j, name in enumerate(("start", "unit list built", "wave 0: units + contacts done", "block done")):
            v = np.sort(fb[:, j])
            print("k_fin blocks, %-30s min %7.1f  median %7.1f  p90 %7.1f  max %7.1f us" % (name, v[0], v[len(v) // 2], v[int(0.9 * len(v))], v[-1]))
        late = np.argsort(fb[:, 3])[-8:]
        print("the 8 latest blocks:", late.tolist(), "done at", np.round(fb[late, 3], 1).tolist(), "their wave 0 at", np.round(fb[late, 2], 1).tolist())
    bs = bs.reshape(4096, 4)[:nblk].astype(np.float64)
    if n == 250:
        st0 = np.sort((bs[:, 0] - t0) * 0.01)
        print("block start times (us), sorted, every 32nd:", np.round(st0[::32], 1))
        en0 = np.sort((bs[:, 2] - t0) * 0.01)
        print("block loop-done times (us), sorted, every 32nd:", np.round(en0[::32], 1))
    blk_acc += np.array([bs[:, 0].min() - t0, bs[:, 0].max() - t0, bs[:, 1].min() - t0, bs[:, 1].max() - t0, bs[:, 2].min() - t0, bs[:, 2].max() - t0]) * 0.01
a = acc / np.maximum(cnt32, 1)
st_mask = None
names = {0: "tm start", 1: "tm tables done", 2: "tm mass done", 3: "tm released", 4: "tm finisher: scan seen", 5: "tm finisher: contacts priced", 6: "tm finisher: published", 8: "scan start", 9: "scan prologue done", 10: "scan block0 loop done",
         24: "tm: A0/B0 loaded", 25: "tm: representatives loaded", 26: "tm: transforms", 27: "tm: relations", 28: "tm: dedupe", 29: "tm: slots", 30: "tm: tasks written",
         11: "fin: unit list built (block 0)", 12: "fin: mass units done (block 0, wave 0)", 13: "fin: contacts priced (block 0, wave 0)", 14: "fin: block 0 past its barrier",
         16: "fin start", 17: "fin tables seen", 18: "fin last-indexed block at ticket", 19: "fin last block past ticket", 20: "fin sums handed out", 21: "fin seq published"}
for i in sorted(names):
    if abs(a[i]) < 1e6:
        print("%-36s %7.2f us" % (names[i], a[i]))

for name, v in zip(("first block start", "last block start", "first block past prologue", "last block past prologue", "first block loop done", "last block loop done"), blk_acc / n):
    print("scan: %-31s %7.2f us" % (name, v))
