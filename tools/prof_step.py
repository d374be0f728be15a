"""Full MCMC step timing at C5 (exploded + 2,000 steps): graal_step (C host logic) vs GRAAL_PY_STEP=1 (Python host logic),
optionally a host profile.  Usage: python tools/prof_step.py [n_steps] [--profile]"""
import os, sys, time
import torch  # (before the engine initialises HIP: torch carries its own ROCm runtime)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from graal_amd import synth, dist as gdist
n_steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2000
P = synth.make_problem(n_bins=50000, nnz=20_000_000, n_sub=1, seed=20141217)
P["S_o_A_frags"] = bench.exploded_layout(P)
rng = np.random.RandomState(20141217)
smp = bench.build_sampler(P, rng, gdist.Group(0, 1), 0)
smp.init_likelihood()
order = np.arange(int(smp.n_new_frags), dtype=np.int32); rng.shuffle(order)
for i in order[:2000]:
    smp.step_max_likelihood(int(i), 5)

def run(tag, lo, hi):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in order[lo:hi]:
        smp.step_max_likelihood(int(i), 5)
    torch.cuda.synchronize()
    print("%-22s %.1f us per full MCMC step (%d steps)" % (tag, 1e6 * (time.perf_counter() - t0) / (hi - lo), hi - lo), flush=True)
run("c step" if smp._c_step else "python step", 2000, 2000 + n_steps)
smp.engine.set_timing(0)
run("  (no scan events)", 2000 + n_steps, 2000 + 2 * n_steps)
smp._c_step = False
run("python host logic", 2000 + 2 * n_steps, 2000 + 3 * n_steps)
smp._c_step = True
if "--profile" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for i in order[2000 + 3 * n_steps:2000 + 4 * n_steps]:
        smp.step_max_likelihood(int(i), 5)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
