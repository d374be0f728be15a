"""GPU box: scoring phase of the LATE STAGE (C5 on its 7 original contigs) in reference arithmetic, the union-set kernels
(strict2.h) against round 3's per-neighbour kernels (GRAAL_STRICT_V1=1), same proposals, results compared bit for bit.
Usage: python tools/late_ab.py [steps] [child-tag]"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
if len(sys.argv) > 2:   # child: time this build's path, store the sums
    import bench
    from graal_amd import synth, dist as gdist
    P = synth.make_problem(n_bins=50000, nnz=20_000_000, n_sub=1, seed=20141217)
    rng = np.random.RandomState(20141217)
    smp = bench.build_sampler(P, rng, gdist.Group(0, 1), 0, "strict")
    smp.init_likelihood()
    n, K = int(smp.n_new_frags), 5
    max_id = smp.modify_gl_cuda_buffer(0)
    props = []
    for f in rng.randint(0, n, size=steps + 3):
        nb = smp.return_neighbours(int(f), K); nb.sort(); props.append((int(f), nb))
    smp.engine.set_timing(0)
    for f, nb in props[:3]:
        smp._candidate_deltas(f, nb, max_id)
    t0 = time.perf_counter()
    out = [smp._candidate_deltas(f, nb, max_id) for f, nb in props[3:]]
    dt = (time.perf_counter() - t0) / steps
    c = smp.engine.last_counters()
    print("%-10s %.3f ms/step   (queued %d, units %d)" % (sys.argv[2], dt * 1e3, c[2], c[1]), flush=True)
    np.save(os.path.join(ROOT, "gpurun_out", "late_ab_%s.npy" % sys.argv[2]), np.stack(out))
    smp.free_gpu()
    sys.exit(0)
res = {}
for tag, env in (("v2", {}), ("v1", {"GRAAL_STRICT_V1": "1"})):
    e = dict(os.environ); e.update(env)
    subprocess.check_call([sys.executable, os.path.abspath(__file__), str(steps), tag], env=e, timeout=900)
    res[tag] = np.load(os.path.join(ROOT, "gpurun_out", "late_ab_%s.npy" % tag))
d = np.abs(res["v1"] - res["v2"])
print("max |v1 - v2| = %.3e  (bit-identical: %s; NaN pattern equal: %s)" % (np.nanmax(d), bool(np.array_equal(res["v1"], res["v2"], equal_nan=True)),
      bool(np.array_equal(np.isnan(res["v1"]), np.isnan(res["v2"])))))
