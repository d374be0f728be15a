"""GPU box: the carried total (commits' own-pixel corrections) at 40,000 bins x 3 sub-fragments -- k_own_corr's grid-stride path, contigs of thousands of bins:
carried + correction against a full evaluation every 50 steps of 600.  usage: python tools/carry_big.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from graal_amd import synth
from tests.test_sampler_gpu import make_gpu_sampler
from tests.test_carried_total_gpu import current_total
par = synth.make_param_simu(fact=200.0, v_inter=0.02)
P = synth.make_problem(n_bins=40000, nnz=3_000_000, n_sub=3, seed=5, contig_weights=(6.8, 6.2, 5.3, 4.6, 4.0, 3.4, 2.7), mean_len_bp=1800.0, accu=9, param=par)
rng = np.random.RandomState(4)
g = make_gpu_sampler(P, rng, reference_arithmetic="strict")
print("own", g._own_corr, "n", P["n_frags"])
g.init_likelihood()
order = rng.permutation(P["n_frags"])[:600]
worst = 0.0; big = 0.0
t0 = time.perf_counter()
for i, f in enumerate(order):
    g.step_max_likelihood(int(f), 3)
    if (i + 1) % 50 == 0:
        carried, full, corr = current_total(g)
        worst = max(worst, abs(carried - full) / abs(full)); big = max(big, abs(corr))
print("600 steps from the 7 original contigs of ~5,700 bins x 3 sub-fragments: carried + correction vs full, worst rel %.2e, largest correction %.3e, repaired %d, %.0f us/step incl. 12 checks"
      % (worst, big, g.engine.run_counters()["carried_totals_repaired"], 1e6 * (time.perf_counter() - t0) / 600))
assert worst < 1e-10
