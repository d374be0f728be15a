#!/usr/bin/env python3
"""GPU box diagnostic: C2-shape start_EM from the exploded genome, oracle and engine side by side; first step after which the
contig LABELS differ (the layouts agree), with the move and the labels around it."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O
from tests.test_shapes_gpu import c2_problem
from tests.test_sampler_gpu import make_gpu_sampler

P = c2_problem()
seed = 77
ora = O.OracleSampler(P, np.random.RandomState(seed), fix_trans_accu=True)
rng = np.random.RandomState(seed)
g = make_gpu_sampler(P, rng)
for s in (ora, g):
    s.init_likelihood(); s.modify_gl_cuda_buffer(0, 0)
def labels(s):
    if s is g:
        s.gpu_vect_frags.copy_from_gpu(); return np.copy(s.gpu_vect_frags.id_c), np.copy(s.gpu_vect_frags.l_cont)
    return np.copy(s.gpu_vect_frags["id_c"]), np.copy(s.gpu_vect_frags["l_cont"])
n = int(g.n_new_frags)
for i in range(n):          # explode, step by step
    for s in (ora, g):
        m = s.modify_gl_cuda_buffer(i, 0)
        s.test_copy_struct(i, 0, 0, m if s is g else s.gpu_vect_frags["id_c"].max())
    a, la = labels(ora); b, lb = labels(g)
    if not np.array_equal(a, b):
        print("labels differ after exploding fragment", i, "n differing", int((a != b).sum()), "oracle", a[:12], "engine", b[:12], "l_cont", la[:12]); break
else:
    print("explode: labels identical")
lf_o = np.arange(n, dtype=np.int32); lf_g = np.arange(n, dtype=np.int32)
ora.rng.shuffle(lf_o); rng.shuffle(lf_g)
for step, (io, ig) in enumerate(zip(lf_o[:400], lf_g[:400])):
    assert io == ig
    ro = ora.step_max_likelihood(io, 3, 512, 0, np.float32(0), np.float32(1))
    rg = g.step_max_likelihood(ig, 3, 512, 0, np.float32(0), np.float32(1))
    assert (ro[5], ro[6]) == (rg[5], rg[6]), (step, ro, rg)
    # labels AFTER the relabel of the next step are what matters; compare the committed layouts after relabeling both
    a, la = labels(ora); b, lb = labels(g)
    if not np.array_equal(a, b):
        d = np.nonzero(a != b)[0]
        print("labels differ after step", step, "move", (int(io), int(ro[6]), int(ro[5])), "n differing", len(d), "first", d[:10], "oracle", a[d[:10]], "engine", b[d[:10]], "l_cont", la[d[:10]], lb[d[:10]])
        print("max label oracle", a.max(), "engine", b.max(), "n_contigs", len(np.unique(a)), len(np.unique(b)))
        break
else:
    print("400 steps: labels identical")
