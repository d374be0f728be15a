#!/usr/bin/env python3
"""GPU box diagnostic: drift of the carried-over likelihood against full re-evaluations during a headless run of a BASELINE
stand-in (tools/run_configs.py shapes): every `resync` steps the carried total is compared with a full evaluation; prints the
largest discrepancies with the step, the move and the layout statistics at that point.
Usage: python tools/diag_drift.py C3 [cycles] [resync]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from graal_amd import synth
from tools.run_configs import CONFIGS

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 20
resync = int(sys.argv[3]) if len(sys.argv) > 3 else 64
n_bins, nnz, n_sub, _, K, accu = CONFIGS[name]
P = synth.make_problem(n_bins=n_bins, nnz=nnz, n_sub=n_sub, seed=2014, contig_weights=(6.8, 6.2, 5.3, 4.6, 4.0, 3.4, 2.7),
                       mean_len_bp=660.0 * (27 if n_sub > 1 else 1) / max(n_sub, 1), accu=accu)
rng = np.random.RandomState(1)
smp = bench.build_sampler(P, rng, None, 0)
smp.resync_every = 10 ** 9          # no automatic resync: this script compares and resyncs itself
smp.init_likelihood()
smp.modify_gl_cuda_buffer(0)
smp.explode_genome()
n = int(smp.n_new_frags)
frags = np.arange(n, dtype=np.int32)
worst = []
step = 0
last_moves = []
t0 = time.time()
for c in range(cycles):
    rng.shuffle(frags)
    for i in frags:
        r = smp.step_max_likelihood(int(i), K)
        last_moves.append((step, int(i), int(r[6]), int(r[5]), int(r[1]), int(r[4])))
        last_moves = last_moves[-resync:]
        step += 1
        if step % resync == 0:
            carried = smp.likelihood_t
            full = smp.eval_likelihood()
            d = abs(carried - full)
            worst.append((d, d / abs(full), step, c, r[1], int(r[4])))
            if d > 1e-6 * abs(full):
                print("step %d cycle %d: carried %.6f full %.6f diff %.4f rel %.2e  n_contigs %d max_len %d; moves since the last check with op >= 9 or circular: %s"
                      % (step, c, carried, full, carried - full, d / abs(full), r[1], int(r[4]),
                         [m for m in last_moves if m[3] >= 9][:8]), flush=True)
            smp.likelihood_t = full
worst.sort(reverse=True)
print("%s: %d steps in %.0f s; discrepancies above 1e-6 relative: %d of %d checks; five largest (abs, rel, step, cycle, contigs, max len):"
      % (name, step, time.time() - t0, sum(1 for w in worst if w[1] > 1e-6), len(worst)))
for w in worst[:5]:
    print("   ", w)
