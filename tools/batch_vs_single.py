#!/usr/bin/env python3
"""GPU box: wall time per MCMC step through sampler.steps_max_likelihood (graal_steps: runs of steps in one C call) against one
step_max_likelihood call per step -- C5 exploded (+2,000 steps) and the C4 stand-in after its first cycle."""
import os, sys, time
import torch
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from graal_amd import synth
for name, kw, warm in (("C5 exploded", dict(n_bins=50000, nnz=20_000_000, n_sub=1, seed=20141217), 2000),
                       ("C4 after one cycle", dict(n_bins=40000, nnz=8_000_000, n_sub=1, seed=2014, contig_weights=(6.8, 6.2, 5.3, 4.6, 4.0, 3.4, 2.7), mean_len_bp=660.0, accu=1), 40000)):
    P = synth.make_problem(**kw)
    P["S_o_A_frags"] = bench.exploded_layout(P)
    rng = np.random.RandomState(1)
    smp = bench.build_sampler(P, rng, None, 0)
    smp.init_likelihood()
    n = P["n_frags"]
    order = np.arange(n, dtype=np.int32); rng.shuffle(order)
    smp.steps_max_likelihood(order[:warm], 5)
    rng.shuffle(order)
    N = 4000
    t0 = time.perf_counter(); smp.steps_max_likelihood(order[:N], 5); t1 = time.perf_counter()
    for i in order[N:2 * N]:
        smp.step_max_likelihood(int(i), 5)
    t2 = time.perf_counter()
    smp.steps_max_likelihood(order[2 * N:3 * N], 5); t3 = time.perf_counter()
    print("%s: runs of steps in one call %.1f us/step, one call per step %.1f us/step, runs again %.1f us/step" % (name, 1e6 * (t1 - t0) / N, 1e6 * (t2 - t1) / N, 1e6 * (t3 - t2) / N), flush=True)
    smp.free_gpu()
