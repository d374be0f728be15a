"""GPU box: how many units does k_strict2 get per step late in a run of a stand-in?  usage: python tools/units_per_step.py [C2|C3|C4] [cycles]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from graal_amd import em, synth  # noqa: E402
from tools.run_configs import CONFIGS  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 12
n_bins, nnz, n_sub, _, K, accu = CONFIGS[name]
P = synth.make_problem(n_bins=n_bins, nnz=nnz, n_sub=n_sub, seed=2014, contig_weights=(6.8, 6.2, 5.3, 4.6, 4.0, 3.4, 2.7),
                       mean_len_bp=660.0 * (27 if n_sub > 1 else 1) / max(n_sub, 1), accu=accu)
rng = np.random.RandomState(1)
smp = bench.build_sampler(P, rng, None, 0, "strict")
em.run_em(smp, cycles, K, rng=rng)
units = []
for f in rng.permutation(n_bins)[:400]:
    smp.step_max_likelihood(int(f), K)
    units.append(int(smp.engine.last_counters()[3]))
u = np.array(units)
print("%s after %d cycles, 400 steps: units per step min %d, quartiles %s, max %d; <= 1,024: %.0f %%, <= 2,048: %.0f %%, <= 4,096: %.0f %%, <= 32,768: %.0f %%"
      % (name, cycles, u.min(), np.percentile(u, [25, 50, 75]).astype(int), u.max(), 100 * (u <= 1024).mean(), 100 * (u <= 2048).mean(),
         100 * (u <= 4096).mean(), 100 * (u <= 32768).mean()))
