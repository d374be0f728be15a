"""GPU parity at the BASELINE shapes and on LONG contigs (VERDICT r01 "next" #1).

The C oracle (dense restatement of kernels3.cu) is fast enough for the C2 stand-in (1,086 bins x 3 sub-fragments, 120 k
contacts, 7 original contigs of <= 227 bins) and for a C3-like 3,500 x 3 shape, so the code paths only long contigs reach are
checked against it here: pieces of more than 64 fragments (multi-chunk mass items), more than 256 affected fragments
(block-wide bitmap marking), work lists beyond the direct item table, the large k_fin grids, and -- in a child process with
GRAAL_SCAN_THREADS / GRAAL_SCAN_BLOCKS set -- the 1,024-thread multi-block streaming pass on a small list.

* candidate deltas: <= 1e-8 x |logL| against the oracle on grid coordinates (float32 kb values exact), circular contigs included;
* accepted-move traces of headless start_EM runs, bit-exact, from the exploded genome AND from the 7 original contigs
  (the late-stage regime, where every step is expected-mass work between whole contigs)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from graal_amd import em, synth
from oracle import oracle as O
from tests import util
from tests.test_engine_gpu import dense_for, engine_for, oracle_deltas, random_state_for, relabel_ref

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def shape_problem(n_bins, nnz, seed=2014, n_sub=3, accu=9, grid_bp=2000, mean_len_bp=2000.0, fact=200.0, v_inter=0.02):
    par = synth.make_param_simu(fact=fact, v_inter=v_inter)
    P = synth.make_problem(n_bins=n_bins, nnz=nnz, n_sub=n_sub, seed=seed, contig_weights=synth.C5_CONTIG_WEIGHTS,
                           mean_len_bp=mean_len_bp, accu=accu, param=par, grid_bp=grid_bp)
    return synth.with_dense(P)


def c2_problem():
    """BASELINE config 2 stand-in: S. cerevisiae S1 level 3 shape (N0 ~ 1.1k bins, 3 sub-fragments per bin)."""
    return shape_problem(1086, 120_000)


def zero_based(P):
    s = O.copy_state(P["S_o_A_frags"])
    s["id_c"][:] -= 1
    return s


def check_deltas(P, states, n_props, K, seed, tol_rel=1e-8, strict=False):
    """13*K deltas of n_props proposals per layout against the dense oracle; returns the worst |error| / |logL|."""
    dense = dense_for(P)
    rng = np.random.RandomState(seed)
    n = P["n_frags"]
    worst = 0.0
    for s in states:
        max_id = relabel_ref(s)
        e = engine_for(P, s)
        if strict:
            e.set_mode(strict=True)
        assert e.relabel_contigs() == max_id
        for _ in range(n_props):
            fA = int(rng.randint(n))
            fBs = sorted(int(v) for v in rng.choice(np.setdiff1d(np.arange(n), [fA]), K, replace=False))
            base, want = oracle_deltas(P, dense, s, fA, fBs, max_id)
            got = e.eval_candidates(fA, fBs, max_id)
            err = np.abs(got - want).max() / abs(base)
            assert err <= tol_rel, (fA, fBs, err)
            worst = max(worst, err)
        c = e.last_counters()
        e.close()
    return worst, c


def c2_states(P, seed):
    """The 7 original contigs, and two random 7-contig layouts with circular contigs and reversed bins."""
    rng = np.random.RandomState(seed)
    return [zero_based(P)] + [random_state_for(P, rng, n_contigs=7, p_circ=0.4) for _ in range(2)]


def test_c2_shape_candidate_deltas_long_contigs():
    P = c2_problem()
    states = c2_states(P, 1)
    assert max(int(s["l_cont"].max()) for s in states) > 200 and any((s["circ"] == 1).any() for s in states)
    worst, counters = check_deltas(P, states, n_props=10, K=3, seed=2)   # 30 proposals x 39 candidates
    assert worst <= 1e-8
    assert counters[2] > 64 and counters[3] > 0   # the last step queued contacts / mass items for k_fin: the long-contig path ran


def test_c2_shape_ten_neighbours_in_one_pass():
    """K = 10 (the reference's n_neighbors cap, cuda_lib_gl.py:444) is one scan pass."""
    P = c2_problem()
    worst, _ = check_deltas(P, c2_states(P, 3)[:2], n_props=2, K=10, seed=4)
    assert worst <= 1e-8


def test_c3_like_shape_candidate_deltas():
    """T. reesei level 3 stand-in (3,500 bins x 3 sub-fragments; the real N0 is unknown without the dataset):
    contigs of up to ~730 bins -- more than 256 affected fragments, pieces of several hundred fragments."""
    P = shape_problem(3500, 600_000, seed=2015)
    rng = np.random.RandomState(5)
    states = [zero_based(P), random_state_for(P, rng, n_contigs=7, p_circ=0.4)]
    worst, counters = check_deltas(P, states, n_props=3, K=3, seed=6)
    assert worst <= 1e-8
    assert counters[2] > 1000


def _trace_pair(P, seed, n_steps, delta, scrambled):
    from tests.test_sampler_gpu import make_gpu_sampler

    class Stop(Exception):
        pass

    def run(smp, rng):
        box = {}

        def on_step(j, i, tr):
            box["t"] = tr
            if len(tr.id_fA) >= n_steps:
                raise Stop()
        try:
            em.run_em(smp, 1, delta, rng=rng, scrambled=scrambled, on_step=on_step)
        except Stop:
            pass
        return box["t"]
    ora = O.OracleSampler(P, np.random.RandomState(seed), fix_trans_accu=True)
    t_ref = run(ora, ora.rng)
    gpu_rng = np.random.RandomState(seed)
    g = make_gpu_sampler(P, gpu_rng)
    t_gpu = run(g, gpu_rng)
    return ora, t_ref, g, t_gpu


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("scrambled,n_steps", [(True, 400), (False, 400)])
def test_c2_shape_trace_is_bit_exact(scrambled, n_steps):
    """start_EM (main_gl.py:210-283) on the C2 stand-in, K = 3: from the exploded genome, and from the 7 original contigs --
    there every step affects two contigs of ~150 bins (the regime a real run reaches after a few cycles)."""
    P = c2_problem()
    ora, t_ref, g, t_gpu = _trace_pair(P, 77, n_steps, 3, scrambled)
    assert len(t_gpu.id_fA) == len(t_ref.id_fA) == n_steps
    assert np.array_equal(t_gpu.mutations(), t_ref.mutations())          # accepted-move trace, bit-exact
    assert t_gpu.n_contigs == t_ref.n_contigs and t_gpu.mean_len == t_ref.mean_len and t_gpu.dist == t_ref.dist
    assert np.allclose(t_gpu.likelihood, t_ref.likelihood, rtol=1e-8, atol=0)
    g.gpu_vect_frags.copy_from_gpu()
    bad = {k: int((getattr(g.gpu_vect_frags, k) != ora.gpu_vect_frags[k]).sum()) for k in O.FIELDS
           if not np.array_equal(getattr(g.gpu_vect_frags, k), ora.gpu_vect_frags[k])}
    assert not bad, (bad, t_gpu.mutations()[-3:], g.gpu_vect_frags.id_c[:12], ora.gpu_vect_frags["id_c"][:12],
                     g.gpu_vect_frags.l_cont[:12], ora.gpu_vect_frags["l_cont"][:12])   # fragment ordering, bit-exact
    if not scrambled:
        assert max(t_gpu.n_contigs) < 120         # stayed in the long-contig regime (1,086 bins)
    assert ora.n_stale_paste == 0 and g.n_stale_paste == 0
    g.free_gpu()


def _child_deltas():
    """Body of the forced-configuration child process: C2-shape deltas on the original layout and one circular layout."""
    P = c2_problem()
    worst, counters = check_deltas(P, c2_states(P, 1)[:2], n_props=3, K=3, seed=8)
    print("child ok worst %.3e queued %d" % (worst, counters[2]))


@pytest.mark.timeout(900)
@pytest.mark.parametrize("env", [{"GRAAL_SCAN_THREADS": "1024", "GRAAL_SCAN_BLOCKS": "4"},
                                 {"GRAAL_SCAN_THREADS": "1024", "GRAAL_SCAN_BLOCKS": "4", "GRAAL_SCAN_G": "8"},
                                 {"GRAAL_SCAN_DONE": "flags", "GRAAL_FIN_BLOCKS": "2048"},
                                 {"GRAAL_SCAN_FOLD_BITS": "10"}])   # (3,258 ids folded onto 1,024 bits: most queued contacts are false positives)
def test_forced_scan_configurations_agree_with_the_oracle(env):
    """A 120 k-contact list takes the 256-thread scan by default; the tuning knobs are read once per process, so a child
    process runs the 1,024-thread, multi-block configuration (what a 20 M-contact list uses) against the oracle."""
    e = dict(os.environ)
    e.update(env)
    e["PYTHONPATH"] = ROOT + os.pathsep + e.get("PYTHONPATH", "")
    out = subprocess.run([sys.executable, "-c", "import tests.test_shapes_gpu as t; t._child_deltas()"], cwd=ROOT, env=e,
                         capture_output=True, text=True, timeout=800)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "child ok" in out.stdout


def test_generic_coordinates_c2_shape_default_bounded_strict_exact():
    """Arbitrary bp lengths at the C2 shape (660 bp fragments, contigs of ~0.4 Mb): the dense float32 reference carries coordinate
    rounding noise on pairs whose geometry a move does not change; the default engine treats those as exactly unchanged, so its
    candidate scores differ from the reference's by that noise (measured here: a few 1e-5 of logL -- beyond north_star's 1e-5,
    hence GRAAL_MODE_STRICT, which re-prices those pairs like the reference and agrees to libm ulps)."""
    P = shape_problem(1086, 120_000, seed=2016, grid_bp=None, mean_len_bp=660.0)
    worst, _ = check_deltas(P, [zero_based(P)], n_props=4, K=3, seed=9, tol_rel=1e-4)
    print("generic coordinates, C2 shape, default mode: worst |delta error| / |logL| = %.3e" % worst)
    worst_s, _ = check_deltas(P, [zero_based(P)], n_props=4, K=3, seed=9, tol_rel=1e-8, strict=True)
    print("generic coordinates, C2 shape, strict mode:  worst |delta error| / |logL| = %.3e" % worst_s)
    assert worst_s <= 1e-8 < worst
