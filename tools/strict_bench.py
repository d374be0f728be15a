"""Scoring-phase timing of the C5 headline state and of the late stage in both arithmetics (default = exact deltas,
strict = reference arithmetic), same engine, same proposals.  Usage: python tools/strict_bench.py [steps]"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from graal_amd import synth, dist as gdist

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
P = synth.make_problem(n_bins=50000, nnz=20_000_000, n_sub=1, seed=20141217)
soa_original = P["S_o_A_frags"]
P["S_o_A_frags"] = bench.exploded_layout(P)
rng = np.random.RandomState(20141217)
smp = bench.build_sampler(P, rng, gdist.Group(0, 1), 0)
n, K = int(smp.n_new_frags), 5
smp.init_likelihood()
order = np.arange(n, dtype=np.int32); rng.shuffle(order)
for i in order[:2000]:
    smp.step_max_likelihood(int(i), K)
max_id = smp.modify_gl_cuda_buffer(0)
props = []
for f in rng.randint(0, n, size=steps + 30):
    nb = smp.return_neighbours(int(f), K); nb.sort(); props.append((int(f), nb))
smp.engine.set_timing(0)
def run(tag):
    for f, nb in props[:30]: smp._candidate_deltas(f, nb, max_id)
    t0 = time.perf_counter()
    for f, nb in props[30:]: smp._candidate_deltas(f, nb, max_id)
    dt = (time.perf_counter() - t0) / steps
    print("%-28s %.1f us/step  %.3f M cand/s" % (tag, dt * 1e6, 65 / dt / 1e6), flush=True)
    return np.stack([smp._candidate_deltas(f, nb, max_id) for f, nb in props[30:60]])
d0 = run("headline default")
smp.engine.set_mode(ref_trans_accu=True, strict=True)
d1 = run("headline strict")
smp.engine.set_mode()
print("max |default - strict| on the headline state: %.3e (logL %.4e)" % (np.abs(d0 - d1).max(), smp.likelihood_t))
smp.free_gpu()
# late stage
P2 = dict(P); P2["S_o_A_frags"] = soa_original
rng2 = np.random.RandomState(20141217)
s2 = bench.build_sampler(P2, rng2, gdist.Group(0, 1), 0)
s2.init_likelihood()
mid2 = s2.modify_gl_cuda_buffer(0)
props2 = []
for f in rng2.randint(0, n, size=3 + 8):
    nb = s2.return_neighbours(int(f), K); nb.sort(); props2.append((int(f), nb))
s2.engine.set_timing(0)
def run2(tag):
    for f, nb in props2[:3]: s2._candidate_deltas(f, nb, mid2)
    t0 = time.perf_counter()
    out = [s2._candidate_deltas(f, nb, mid2) for f, nb in props2[3:]]
    dt = (time.perf_counter() - t0) / len(props2[3:])
    c = s2.engine.last_counters()
    print("%-28s %.3f ms/step  queued %d items %d" % (tag, dt * 1e3, c[2], c[3]), flush=True)
    return np.stack(out)
e0 = run2("late default")
s2.engine.set_mode(ref_trans_accu=True, strict=True)
e1 = run2("late strict")
print("max |default - strict| late: %.3e  rel to logL %.3e" % (np.abs(e0 - e1).max(), np.abs(e0 - e1).max() / abs(s2.likelihood_t)))
