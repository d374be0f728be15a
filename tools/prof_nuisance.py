"""Where does a nuisance-parameter step (sampler.step_nuisance_parameters) spend its time on the C5 state?  cProfile of the host side."""
import cProfile, pstats, io, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from graal_amd import synth, dist as gdist
import torch

n_bins = int(os.environ.get("NB", 50000)); nnz = int(os.environ.get("NNZ", 20_000_000))
P = synth.make_problem(n_bins=n_bins, nnz=nnz, n_sub=1, seed=20141217)
P["S_o_A_frags"] = bench.exploded_layout(P)
rng = np.random.RandomState(1)
smp = bench.build_sampler(P, rng, gdist.Group(0, 1), 0)
smp.init_likelihood()
order = np.arange(n_bins); rng.shuffle(order)
for i in order[:500]:
    smp.step_max_likelihood(int(i), 5)
smp.bins = np.arange(1.0, 41.0, 1.0)
for i in order[500:520]:
    smp.step_max_likelihood(int(i), 5); smp.step_nuisance_parameters(0, 0, 1)
torch.cuda.synchronize()
t = time.perf_counter()
for i in order[520:620]:
    smp.step_max_likelihood(int(i), 5)
t1 = time.perf_counter()
for i in order[620:720]:
    smp.step_nuisance_parameters(0, 0, 1)
t2 = time.perf_counter()
print("step_max_likelihood %.1f us, step_nuisance_parameters %.1f us" % ((t1 - t) * 1e4, (t2 - t1) * 1e4))
pr = cProfile.Profile(); pr.enable()
for i in order[720:820]:
    smp.step_max_likelihood(int(i), 5); smp.step_nuisance_parameters(0, 0, 1)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue())
