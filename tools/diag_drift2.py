#!/usr/bin/env python3
"""GPU box diagnostic: run a BASELINE stand-in for some cycles, then check EVERY step: accepted delta vs full(after) - full(before);
print the steps where they disagree, with the move, the records of fA / fB before the move and the contig statistics."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from graal_amd import synth
from tools.run_configs import CONFIGS
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 11
n_check = int(sys.argv[3]) if len(sys.argv) > 3 else 600
n_bins, nnz, n_sub, _, K, accu = CONFIGS[name]
P = synth.make_problem(n_bins=n_bins, nnz=nnz, n_sub=n_sub, seed=2014, contig_weights=(6.8, 6.2, 5.3, 4.6, 4.0, 3.4, 2.7),
                       mean_len_bp=660.0 * (27 if n_sub > 1 else 1) / max(n_sub, 1), accu=accu)
rng = np.random.RandomState(1)
smp = bench.build_sampler(P, rng, None, 0)
smp.init_likelihood(); smp.modify_gl_cuda_buffer(0); smp.explode_genome()
n = int(smp.n_new_frags)
frags = np.arange(n, dtype=np.int32)
for c in range(cycles):
    rng.shuffle(frags)
    for i in frags:
        smp.step_max_likelihood(int(i), K)
rng.shuffle(frags)
bad = 0
for t, i in enumerate(frags[:n_check]):
    before = smp.eval_likelihood()
    smp.gpu_vect_frags.copy_from_gpu()
    g = smp.gpu_vect_frags
    snap = {k: np.copy(getattr(g, k)) for k in ("id_c", "pos", "l_cont", "circ", "ori", "l_cont_bp", "start_bp", "len_bp")}
    smp.likelihood_t = before
    r = smp.step_max_likelihood(int(i), K)
    o, op, fB = r[0], int(r[5]), int(r[6])
    after = smp.eval_likelihood()
    d_acc, d_full = o - before, after - before
    if abs(d_acc - d_full) > 1e-6 * abs(before):
        bad += 1
        fA = int(i)
        print("step %d: fA %d fB %d op %d: accepted delta %.4f, full diff %.4f (mismatch %.4f) | fA: contig %d pos %d/%d circ %d ori %d | fB: contig %d pos %d/%d circ %d ori %d | n_circ_frags %d"
              % (t, fA, fB, op, d_acc, d_full, d_acc - d_full, snap["id_c"][fA], snap["pos"][fA], snap["l_cont"][fA], snap["circ"][fA], snap["ori"][fA],
                 snap["id_c"][fB], snap["pos"][fB], snap["l_cont"][fB], snap["circ"][fB], snap["ori"][fB], int((snap["circ"] == 1).sum())), flush=True)
        if bad >= 12:
            break
print("checked", t + 1, "steps,", bad, "mismatches above 1e-6 relative")
