#!/usr/bin/env python3
"""GPU box, counting build (-DGRAAL_STAMPS -DGRAAL_S2_COUNTS): of k_strict2's (fragment pair, class) evaluations in the late stage (C5 on its 7
original contigs, bench.py's `late_stage` proposals), how many hand the contact model the float32 inputs of the CURRENT layout -- the same
distance in the same circular model -- so that their term is exactly zero and the evaluation could be skipped?  (round-4 review item 6:
the one lever on the number of evaluations that does not touch the anchor.)  Prints the counts; STAMPS_SHAPE=c4 for the C4 stand-in after 3 cycles."""
import ctypes, os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "graal_amd", "libgraal_hip_stamps.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                       "-DGRAAL_STAMPS", "-DGRAAL_S2_COUNTS", "-o", so, os.path.join(ROOT, "graal_amd", "csrc", "graal_hip.hip")])
from graal_amd import build
build.HIP_LIB = so
from graal_amd import lib, synth
import bench
shape = os.environ.get("STAMPS_SHAPE", "c5late")
K = 5
if shape == "c4":
    P = synth.make_problem(n_bins=40000, nnz=8_000_000, n_sub=1, seed=2014, contig_weights=synth.C5_CONTIG_WEIGHTS, mean_len_bp=660.0, accu=1)
    P["S_o_A_frags"] = bench.exploded_layout(P)
else:
    P = synth.make_problem(n_bins=50000, nnz=20_000_000, n_sub=1, seed=20141217)
rng = np.random.RandomState(20141217)
smp = bench.build_sampler(P, rng, None, 0, "strict")
if shape == "c4":
    from graal_amd import em
    em.run_em(smp, int(os.environ.get("STAMPS_CYCLES", 3)), K, rng=rng, scrambled=False)
smp.init_likelihood()
max_id = smp.modify_gl_cuda_buffer(0)
L = lib.load()
L.graal_debug_s2eq.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
z = np.zeros(8, dtype=np.uint64)
L.graal_debug_s2eq(smp.engine._h, z.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), 1)
n = int(smp.n_new_frags)
props = 0
for f in rng.randint(0, n, size=12):
    nb = smp.return_neighbours(int(f), K); nb.sort()
    smp._candidate_deltas(int(f), nb, max_id)
    props += 1
assert L.graal_debug_s2eq(smp.engine._h, z.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), 0) == 0
tot, same, wave_all, half, passes, trans_old = (float(v) for v in z[:6])
st = smp.engine.layout_stats()
print("shape %s (%d contigs, longest %d), %d proposals x K = %d through k_strict2 (single sub-fragment, uniform RF counts: the cis branch)" % (shape, int(st[0]), int(st[4]), props, K))
print("(pair, class) evaluations of cis classes: %.4g per proposal; in (wave, segment fragment, class) passes: %.4g per proposal" % (tot / props, passes / props))
print("  with the float32 inputs of the current layout (term exactly zero): %.4g = %.2f %%" % (same / props, 100.0 * same / max(tot, 1)))
print("  ... in passes where EVERY lane agrees (a wave could skip the pass): %.4g = %.2f %%" % (wave_all / props, 100.0 * wave_all / max(tot, 1)))
print("  ... in passes where at least half the lanes agree (lane compaction could drop them): %.4g = %.2f %%" % (half / props, 100.0 * half / max(tot, 1)))
print("  pairs that are trans in the current layout (no distance to agree with): %.2f %%" % (100.0 * trans_old / max(tot, 1)))
smp.free_gpu()
