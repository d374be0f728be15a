"""Host-side profile (cProfile) of sampler.step_max_likelihood on the C5 headline state."""
import cProfile, pstats, io, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from graal_amd import synth, dist as gdist
P = synth.make_problem(n_bins=50000, nnz=20_000_000, n_sub=1, seed=20141217)
P["S_o_A_frags"] = bench.exploded_layout(P)
rng = np.random.RandomState(1)
smp = bench.build_sampler(P, rng, gdist.Group(0, 1), 0)
smp.init_likelihood()
order = np.arange(50000); rng.shuffle(order)
for i in order[:2000]:
    smp.step_max_likelihood(int(i), 5)
t = time.perf_counter()
for i in order[2000:4000]:
    smp.step_max_likelihood(int(i), 5)
print("full MCMC step %.1f us" % ((time.perf_counter() - t) / 2000 * 1e6))
pr = cProfile.Profile(); pr.enable()
for i in order[4000:6000]:
    smp.step_max_likelihood(int(i), 5)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue())
