#!/usr/bin/env python3
"""GPU box: wall time of one full likelihood evaluation (graal_eval_full_q: k_full_nnz + k_full_mass, what the
nuisance-parameter step and every resync pay) on the C5 map, early (exploded + 2,000 steps) and late (7 contigs) layout."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from graal_amd import dist as gdist, synth
for layout in ("exploded", "original"):
    P = synth.make_problem(n_bins=50000, nnz=20_000_000, n_sub=1, seed=20141217)
    if layout == "exploded":
        P["S_o_A_frags"] = bench.exploded_layout(P)
    rng = np.random.RandomState(1)
    smp = bench.build_sampler(P, rng, gdist.Group(0, 1), 0)
    smp.init_likelihood()
    if layout == "exploded":
        order = np.arange(50000); rng.shuffle(order)
        for i in order[:2000]:
            smp.step_max_likelihood(int(i), 5)
    smp.modify_gl_cuda_buffer(0)
    v = smp._full_likelihood()
    t0 = time.perf_counter()
    for _ in range(20):
        smp._full_likelihood()
    print("%s: full evaluation %.1f us, logL %.6f" % (layout, 1e6 * (time.perf_counter() - t0) / 20, v))
    smp.free_gpu()
