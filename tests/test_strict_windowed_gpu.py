"""GPU: the windowed reference-arithmetic candidate kernels (k_tm's own pricing of small sets, k_strict_flat for sets of up to
~280 fragments, k_strict_cull + k_strict beyond, the finishing block's contacts) -- the path `reference_arithmetic="strict"` runs and bench.py measures.

* equal, BIT FOR BIT, to k_strict_dense -- the O(m^2) kernel that prices every pixel of contig(fA) u contig(fB) under every
  candidate the way sub_compute_likelihood does (kernels3.cu:3259-3718) -- run in a child process (GRAAL_STRICT_DENSE=1);
* against the oracle's reference-arithmetic restatement (fix_trans_accu=False) with repeated bins: deltas, full likelihood and
  full start_EM traces with activity swaps (the modes used to refuse repeats);
* the total carried from step to step in strict mode == a full evaluation (the reference re-evaluates every step)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from graal_amd import em, synth
from oracle import oracle as O
from tests import strict_cases
from tests.test_engine_gpu import relabel_ref
from tests.test_repeats_gpu import engine_with_repeats, oracle_deltas_with_repeats, random_state_with_repeats

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(1500)
def test_windowed_kernels_equal_the_dense_validation_kernel_bit_for_bit(tmp_path):
    out = str(tmp_path / "dense.npz")
    env = dict(os.environ, GRAAL_STRICT_DENSE="1")
    subprocess.check_call([sys.executable, "-m", "tests.strict_cases", out], cwd=ROOT, env=env, timeout=1200)
    dense = np.load(out)
    got = strict_cases.run_cases()
    assert os.environ.get("GRAAL_STRICT_DENSE") in (None, "0")
    # the tiled kernels (k_strict_cull + k_strict) on the small sets too, which k_strict_flat takes by default
    out_t = str(tmp_path / "tiled.npz")
    subprocess.check_call([sys.executable, "-m", "tests.strict_cases", out_t, "0,1,2,3"], cwd=ROOT, env=dict(os.environ, GRAAL_NO_FLAT="1"), timeout=1200)
    tiled = np.load(out_t)
    for i, (name, *_rest) in enumerate(strict_cases.CASES):
        if "case%d" % i in tiled:
            assert np.array_equal(tiled["case%d" % i], got[name]), name
    for i, (name, *_rest) in enumerate(strict_cases.CASES):
        want = dense["case%d" % i]
        assert got[name].shape == want.shape
        assert np.abs(want).max() > 0
        bad = np.argwhere(got[name] != want)
        assert len(bad) == 0, (name, len(bad), bad[:5], (got[name] - want)[tuple(bad[0])])


@pytest.mark.timeout(1500)
def test_terms_beyond_the_fixed_point_range_stay_finite_like_the_reference():
    """The reference accumulates in float64 (kernels3.cu:191-210, 3703-3717): a pair priced at > 2^31 log-likelihood units -- a circular
    contig closing over bp-sized fragments under C5's `fact` -- gives a hopeless but FINITE score (-8.6e8 here), and the sampler must see
    that score, not NaN.  The engine's Q30 sums cannot hold such a term; it goes to the candidate's coarse sum (whole units).  Against the
    oracle run the reference's way on the case of tests/strict_cases.py that used to flag: finite wherever the oracle is finite,
    within 1e-5 relative on the large scores (north_star's tolerance) and 2e-9 |logL| on the others."""
    from tests.test_strict_gpu import ref_deltas
    name, pk, cfg, n_props, K = strict_cases.CASES[5]     # (one contig of 900 bins: four of the 13 x 5 candidates close it into a circle)
    assert pk["fact"] == 1e4
    P = synth.with_dense(strict_cases._problem(**pk))
    dense = O.DenseOracle(P["hic_matrix"], P["np_sub_frags_id"], P["np_sub_frags_len_bp"], P["np_sub_frags_accu"],
                          P["frag_dispatcher"], P["collector_id_repeats"], P["n_frags"], P["mean_squared_frags_per_bin"],
                          P["param_simu"], fix_trans_accu=not cfg["quirk"])
    from graal_amd.lib import Engine
    n_large = 0
    for s, max_id, props in strict_cases.layouts_and_proposals(P, cfg, n_props, K, 1000 + 5):
        e = Engine(0)
        e.upload_subfrags(P["np_sub_frags_id"], P["np_sub_frags_len_bp"], P["np_sub_frags_accu"], P["init_n_sub_frags"], P["mean_squared_frags_per_bin"])
        e.upload_contacts(P["coo_row"], P["coo_col"], P["coo_val"])
        e.set_params(P["param_simu"])
        e.upload_frags(s)
        e.set_mode(ref_trans_accu=cfg["quirk"], strict=True)
        assert e.relabel_contigs() == max_id
        for fA, fBs in props:
            got = e.eval_candidates(fA, fBs, max_id)
            base, want = ref_deltas(P, dense, s, fA, fBs, max_id)
            assert np.array_equal(np.isfinite(got), np.isfinite(want)), (fA, fBs, got[~np.isfinite(got)], want[~np.isfinite(got)])
            big = np.abs(want) >= 1.0e8
            n_large += int(big.sum())
            if big.any():
                assert np.all(np.abs(got[big] - want[big]) <= 1e-5 * np.abs(want[big])), (fA, fBs, got[big], want[big])
            small = ~big & np.isfinite(want)
            assert np.all(np.abs(got[small] - want[small]) <= 2e-9 * abs(base) + 1e-5 * np.abs(want[small]) * (np.abs(want[small]) > 1e6)), (fA, fBs)
        e.close()
    assert n_large > 0, "no candidate with a score of -1e8 and beyond: the case no longer tests the coarse sums"


def rep_problem_ref(n_sub, seed, n_bins=40, nnz=900, dup=(7, 21), n_copies=2):
    """Repeated bins on generic coordinates with non-uniform RF counts: what the reference arithmetic is sensitive to."""
    par = synth.make_param_simu(fact=300.0, v_inter=0.03)
    P = synth.make_problem(n_bins=n_bins, nnz=nnz, n_sub=n_sub, seed=seed, contig_weights=(5, 3, 2), mean_len_bp=1500.0,
                           accu=("random", 1, 9) if n_sub > 1 else 1, param=par)
    return synth.add_repeats(synth.with_dense(P), dup, n_copies)


def ref_dense(P):
    return O.DenseOracle(P["hic_matrix"], P["np_sub_frags_id"], P["np_sub_frags_len_bp"], P["np_sub_frags_accu"],
                         P["frag_dispatcher"], P["collector_id_repeats"], P["n_frags"], P["mean_squared_frags_per_bin"],
                         P["param_simu"], fix_trans_accu=False)


@pytest.mark.parametrize("n_sub,seed,p_circ", [(1, 181, 0.0), (3, 182, 0.0), (3, 183, 0.3)])
def test_reference_arithmetic_with_repeats_full_and_deltas(n_sub, seed, p_circ):
    P = rep_problem_ref(n_sub, seed)
    dense = ref_dense(P)
    rng = np.random.RandomState(seed)
    n = int(P["n_new_frags"])
    copies = np.nonzero(P["S_o_A_frags"]["rep"] == 1)[0]
    for trial in range(3):
        s = random_state_with_repeats(P, rng, n_contigs=int(rng.randint(8, 16)), p_circ=p_circ)
        max_id = relabel_ref(s)
        e = engine_with_repeats(P, s)
        e.set_mode(ref_trans_accu=True, strict=True)
        assert e.relabel_contigs() == max_id
        assert e.eval_full() == pytest.approx(dense.evaluate(s), rel=1e-8)
        for fA in (int(rng.randint(n)), int(copies[trial % len(copies)]), 7):
            fBs = [int(v) for v in rng.choice(np.setdiff1d(np.arange(n), [fA]), 3, replace=False)]
            base, want = oracle_deltas_with_repeats(P, dense, s, fA, fBs, max_id)
            got = e.eval_candidates(fA, fBs, max_id)
            # (every pixel is priced twice from slightly different float32 coordinates: last-place differences between libms do
            # not cancel -- tests/test_strict_gpu.py; 1.5e-10 measured with the correctly rounded model_math.h)
            assert np.all(np.abs(got - want) <= 2e-9 * abs(base)), (trial, fA, fBs, np.abs(got - want).max() / abs(base))
        e.close()


@pytest.mark.parametrize("n_sub,seed,dup", [(1, 191, (7, 21)), (3, 192, (5, 18, 30))])
def test_strict_trace_with_repeats_matches_the_reference_arithmetic_oracle(n_sub, seed, dup):
    from tests.test_sampler_gpu import make_gpu_sampler
    par = synth.make_param_simu(fact=200.0, v_inter=0.02)
    P = synth.make_problem(n_bins=45, nnz=900, n_sub=n_sub, seed=seed, contig_weights=(5, 4, 3), mean_len_bp=2000.0,
                           accu=("random", 1, 9) if n_sub > 1 else 1, param=par)
    P = synth.add_repeats(synth.with_dense(P), dup, 2)
    ora = O.OracleSampler(P, np.random.RandomState(seed), fix_trans_accu=False)
    t_ref = em.run_em(ora, 2, 3, rng=ora.rng)
    rng = np.random.RandomState(seed)
    g = make_gpu_sampler(P, rng, reference_arithmetic="strict")
    t_gpu = em.run_em(g, 2, 3, rng=rng)
    m_ref = np.asarray(t_ref.mutations())
    assert np.array_equal(t_gpu.mutations(), m_ref)
    assert (m_ref[:, 2] == 8).any()                                     # activity swaps occurred
    assert t_gpu.n_contigs == t_ref.n_contigs and t_gpu.dist == t_ref.dist
    assert np.allclose(t_gpu.likelihood, t_ref.likelihood, rtol=1e-8, atol=0)
    g.gpu_vect_frags.copy_from_gpu()
    for k in O.FIELDS:
        assert np.array_equal(getattr(g.gpu_vect_frags, k), ora.gpu_vect_frags[k]), k
    g.free_gpu()


def test_strict_total_carried_over_equals_a_full_evaluation():
    """In reference arithmetic the candidate delta IS full(after) - full(before) pixel by pixel (pixels outside the set get the
    same inputs), so at one sub-fragment per bin -- no bin has a pixel of its own -- the carried-over total needs no per-step
    re-evaluation: after 150 steps on generic coordinates it equals a full pass to summation rounding.  (With sub-fragments the
    sampler adds each commit's own-pixel correction, or re-evaluates every step like the reference: sampler.resync_every.)"""
    from tests.test_sampler_gpu import make_gpu_sampler
    par = synth.make_param_simu(fact=200.0, v_inter=0.02)
    P = synth.with_dense(synth.make_problem(n_bins=150, nnz=5000, n_sub=1, seed=77, contig_weights=(5, 4, 3), mean_len_bp=1800.0,
                                            accu=1, param=par))
    rng = np.random.RandomState(5)
    g = make_gpu_sampler(P, rng, reference_arithmetic="strict")
    assert g.resync_every > 150
    g.init_likelihood()
    order = rng.permutation(P["n_frags"])
    for i in order[:150]:
        g.step_max_likelihood(int(i), 3)
    carried = g.likelihood_t
    full = g.eval_likelihood()
    assert carried == pytest.approx(full, rel=1e-10), (carried, full)
    g.free_gpu()
    P3 = synth.with_dense(synth.make_problem(n_bins=60, nnz=900, n_sub=3, seed=78, contig_weights=(5, 4, 3), mean_len_bp=1800.0, accu=9, param=par))
    g3 = make_gpu_sampler(P3, np.random.RandomState(5), reference_arithmetic="strict")
    assert g3.resync_every == 512 and g3._own_corr      # (one RF count per bin: the commit's own-pixel correction, tests/test_carried_total_gpu.py)
    g3.free_gpu()
    P3 = synth.with_dense(synth.make_problem(n_bins=60, nnz=900, n_sub=3, seed=78, contig_weights=(5, 4, 3), mean_len_bp=1800.0, accu=("random", 1, 9), param=par))
    g3 = make_gpu_sampler(P3, np.random.RandomState(5), reference_arithmetic="strict")
    assert g3.resync_every == 1 and not g3._own_corr     # (mixed RF counts in most bins: the reference's per-step evaluation)
    g3.free_gpu()


@pytest.mark.timeout(1500)
def test_hand_off_behind_k_gprep_survives_its_time_out(monkeypatch):
    """k_strict2 follows k_gprep through a word in memory instead of an event (grids of <= 512 blocks, one rank: launch_strict,
    strict2.h).  The same 150 MCMC steps at the C2 stand-in's shape from its 7 original contigs (every step through k_gprep + k_strict2)
    (a) behind the event (GRAAL_STRICT_GWAIT=0: the anchor), (b) through the word (the default), (c) with the wait made to run out at
    once (GRAAL_GP_WAIT_TICKS=1: a block that does not find the word gives up, the step is flagged failed, repeated behind events and the
    engine stays with events -- asserted through graal_run_counters), (d) with the acquire only in blocks that had to wait (GRAAL_GP_ACQUIRE=0,
    round 4's form):
    every step's 13 x K scores, the accepted-move trace, the likelihood series and the generator state must be the same."""
    from tests.test_sampler_gpu import make_gpu_sampler
    from tests.test_strict_gpu import ref_problem
    P = ref_problem(1086, 120_000, 2016, contig_weights=synth.C5_CONTIG_WEIGHTS, mean_len_bp=660.0, fact=200.0, v_inter=0.02)
    n_steps = 150

    def go(env):
        for k in ("GRAAL_STRICT_GWAIT", "GRAAL_GP_WAIT_TICKS", "GRAAL_GP_ACQUIRE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        rng = np.random.RandomState(31)
        g = make_gpu_sampler(P, rng, reference_arithmetic="strict")
        g.init_likelihood()
        order = np.arange(int(g.n_new_frags), dtype=np.int32)
        rng.shuffle(order)
        scores, trace = [], []
        for i in order[:n_steps]:
            r = g.step_max_likelihood(int(i), 3)
            scores.append(np.array(g.score, dtype=np.float64, copy=True))
            trace.append((int(i), int(r[6]), int(r[5]), float(r[0]), int(r[1])))
        st = rng.get_state(legacy=False)["state"]
        rc = g.engine.run_counters()
        g.free_gpu()
        return scores, trace, int(st["pos"]), st["key"].copy(), rc
    want = go({"GRAAL_STRICT_GWAIT": "0"})
    assert want[4]["strict2_behind_the_word"] == 0 and want[4]["strict2_behind_the_event"] >= n_steps // 2 and want[4]["fallbacks"] == 0
    for env in ({}, {"GRAAL_GP_WAIT_TICKS": "1"}, {"GRAAL_GP_ACQUIRE": "0"}):
        got = go(env)
        rc = got[4]
        if env.get("GRAAL_GP_WAIT_TICKS"):
            assert rc["fallbacks"] >= 1 and rc["in_kernel_waits_in_use"] == 0, rc      # the time-out did fire, the engine went back to events
        else:
            assert rc["fallbacks"] == 0 and rc["strict2_behind_the_word"] >= n_steps // 2, rc
        assert got[1] == want[1], env
        assert got[2] == want[2] and np.array_equal(got[3], want[3]), env
        for a, b in zip(got[0], want[0]):
            assert np.array_equal(a, b), env


@pytest.mark.timeout(900)
def test_unit_list_grows_when_a_step_overflows_it(monkeypatch):
    """The tiled kernel's unit list starts at a soft cap instead of its quadratic worst case (launch_strict; advisor r04): with the cap
    forced to 64 entries the first steps overflow it, k_gprep flags the step, eval_sync grows the list and repeats -- the candidates'
    int64 sums must equal those of a run whose list never overflowed, and the repeats must show in graal_run_counters."""
    name, pk, cfg, n_props, K = strict_cases.CASES[4]     # single sub-fragment, contigs of hundreds of bins, K = 10: hundreds of units per step
    monkeypatch.delenv("GRAAL_SLIST_SOFT_CAP", raising=False)
    want = strict_cases.run_cases(only=[4])[name]
    monkeypatch.setenv("GRAAL_SLIST_SOFT_CAP", "64")
    P = strict_cases._problem(**pk)
    grown = 0
    rows = []
    for s, max_id, props in strict_cases.layouts_and_proposals(P, cfg, n_props, K, 1000 + 4):
        from graal_amd.lib import Engine, Q_SCALE
        e = Engine(0)
        e.upload_subfrags(P["np_sub_frags_id"], P["np_sub_frags_len_bp"], P["np_sub_frags_accu"], P["init_n_sub_frags"], P["mean_squared_frags_per_bin"])
        e.upload_contacts(P["coo_row"], P["coo_col"], P["coo_val"])
        e.set_params(P["param_simu"])
        e.upload_frags(s)
        e.set_mode(ref_trans_accu=cfg["quirk"], strict=True)
        assert e.relabel_contigs() == max_id
        row = []
        for fA, fBs in props:
            d = e.eval_candidates(fA, fBs, max_id)
            row.append(np.where(np.isfinite(d), np.rint(np.nan_to_num(d) * Q_SCALE), -2.0 ** 62).astype(np.int64))
        rc = e.run_counters()
        grown += rc["unit_list_grown"]
        assert rc["fallbacks"] == 0
        rows.append(np.stack(row))
        e.close()
    assert grown >= 1, "the forced soft cap never overflowed: the case no longer exercises the growth"
    assert np.array_equal(np.stack(rows), want)
