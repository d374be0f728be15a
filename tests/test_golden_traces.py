"""Committed golden vectors (tests/golden/traces.json, written by tests/golden/make_traces.py from the oracle):
* CPU: the oracle still reproduces one of them (guards the fixture against silent changes of the oracle or the generator);
* GPU: the engine reproduces ALL of them without the oracle in the loop -- accepted-move trace, contig counts, genome
  distance and final layout bit-exact; log-likelihood series within 1e-6 relative."""
import json
import os

import numpy as np
import pytest

from graal_amd import em
from tests.golden.make_traces import problem

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "traces.json")))


@pytest.mark.parametrize("name", ["repeats", "generic_coordinates"])
def test_oracle_reproduces_the_committed_trace(name):
    from oracle import oracle as O
    g = GOLD[name]
    P = problem(g["n_sub"], g["seed"], g["n_bins"], g["nnz"], g["blacklist"], g["repeats"], g.get("generic", False))
    ora = O.OracleSampler(P, np.random.RandomState(g["seed"]), fix_trans_accu=not g.get("generic", False))
    t = em.run_em(ora, g["cycles"], g["neighbours"], rng=ora.rng)
    assert np.asarray(t.mutations()).tolist() == g["mutations"]
    assert np.allclose(t.likelihood, g["likelihood"], rtol=1e-9, atol=0)
    assert ora.gpu_vect_frags["pos"].tolist() == g["final_pos"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(GOLD))
def test_engine_reproduces_the_committed_trace(name):
    from tests.test_sampler_gpu import make_gpu_sampler
    g = GOLD[name]
    P = problem(g["n_sub"], g["seed"], g["n_bins"], g["nnz"], g["blacklist"], g["repeats"], g.get("generic", False))
    rng = np.random.RandomState(g["seed"])
    s = make_gpu_sampler(P, rng)
    t = em.run_em(s, g["cycles"], g["neighbours"], rng=rng)
    assert np.asarray(t.mutations()).tolist() == g["mutations"]
    assert [int(x) for x in t.n_contigs] == g["n_contigs"]
    assert [float(x) for x in t.dist] == g["dist"]
    assert np.allclose(t.likelihood, g["likelihood"], rtol=1e-6, atol=0)
    s.gpu_vect_frags.copy_from_gpu()
    assert s.gpu_vect_frags.id_c.tolist() == g["final_id_c"]
    assert s.gpu_vect_frags.pos.tolist() == g["final_pos"]
    assert s.gpu_vect_frags.ori.tolist() == g["final_ori"]
    assert s.gpu_vect_frags.activ.tolist() == g["final_activ"]
    s.free_gpu()
