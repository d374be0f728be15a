"""The carried total at sub-fragment shapes (several sub-fragments per bin, reference arithmetic).

The reference evaluates the full likelihood at the top of EVERY step (cuda_lib_gl.py:1828-1848) and adds the candidates' deltas to it.
A delta (sub_compute_likelihood, kernels3.cu:3259-3718) covers the pixels between DIFFERENT bins of contig(A) u contig(B); what a
commit does to the pixels of the bins themselves (a bin's sub-fragment pairs: evaluate_likelihood's diagonal pixels, kernels3.cu:3213)
is in no delta.  The engine's commit kernel computes exactly that remainder for the bins it moves (graal_take_carry_correction,
include/graal_hip.h), and graal_step (flag 16) adds it to the total carried from the last step: the step starts from the full
likelihood of its layout without evaluating it.  These tests pin that equality -- carried + correction == a full evaluation -- along
MCMC runs that go through circular contigs and mirrored bins, and that the run itself (every score, every accepted move) is the one
with the per-step evaluation."""
import numpy as np
import pytest

from graal_amd import synth

pytestmark = pytest.mark.gpu

REL = 1e-10     # float64 sums of ~1e4 float32-rounded terms, re-associated: what a carried total can differ from a full pass by


def problem(n_bins=90, nnz=2600, seed=91, accu=9, n_sub=3):
    par = synth.make_param_simu(fact=200.0, v_inter=0.02)
    return synth.with_dense(synth.make_problem(n_bins=n_bins, nnz=nnz, n_sub=n_sub, seed=seed, contig_weights=(5, 4, 3), mean_len_bp=1800.0,
                                               accu=accu, param=par))


def close_into_circles(P, labels):
    """The initial contigs `labels` as circular contigs, as the reference's paste_contigs leaves them (kernels3.cu:1977-2033)."""
    S = P["S_o_A_frags"]
    for c in labels:
        m = np.nonzero(S["id_c"] == c)[0]
        g = m[np.argsort(S["pos"][m])]
        S["circ"][g] = 1
        S["prev"][g[0]] = g[-1]
        S["next"][g[-1]] = g[0]
    return P


def current_total(g):
    """The sampler's carried total brought up to the committed layout: + the pending own-pixel correction (what the next step adds)."""
    full = g.eval_likelihood()                      # (relabels: collects the last commit's statistics, the correction with them)
    corr, ok = g.engine.take_carry_correction()
    if not ok:                                      # (a mirrored bin of mixed RF counts: the next step would have evaluated)
        g._set_total_from_full(full)
        return full, full, 0.0
    g.likelihood_t += corr                          # (taken: hand it to the total, as the next step would have)
    return g.likelihood_t, full, corr


def run(P, seed, n_steps, check_every, monkeypatch=None, env=None, delta=4, explode=False):
    from tests.test_sampler_gpu import make_gpu_sampler
    if monkeypatch is not None:
        for k in ("GRAAL_NO_OWN_PIXEL_CARRY", "GRAAL_PY_STEP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in (env or {}).items():
            monkeypatch.setenv(k, v)
    rng = np.random.RandomState(seed)
    g = make_gpu_sampler(P, rng, reference_arithmetic="strict")
    if explode:                                     # (every fragment its own contig: the run has everything to rebuild, flips included)
        g.explode_genome()
    g.init_likelihood()
    order = np.concatenate([rng.permutation(P["n_frags"]) for _ in range(1 + n_steps // P["n_frags"])])[:n_steps]
    trace, seen_circ, worst, corr_abs = [], 0, 0.0, 0.0
    for i, f in enumerate(order):
        r = g.step_max_likelihood(int(f), delta)
        trace.append((r[0], r[5], r[6], r[1]))
        seen_circ += int(g._n_circ_prev > 0)
        if check_every and (i + 1) % check_every == 0 and g._own_corr:
            carried, full, corr = current_total(g)
            assert carried == pytest.approx(full, rel=REL), (i, carried, full, corr)
            worst = max(worst, abs(carried - full) / abs(full))
            corr_abs = max(corr_abs, abs(corr))
    out = dict(trace=trace, circ=seen_circ, worst=worst, corr=corr_abs, since_full=g._steps_since_full, resync=g.resync_every,
               own=g._own_corr, counters=g.engine.run_counters(), state=rng.get_state()[1].copy())
    g.free_gpu()
    return out


def test_carried_total_plus_the_commits_own_pixels_is_the_full_likelihood(monkeypatch):
    """600 steps from three initial contigs, two of them circular (joins, splits, flips, circles opened), checked every 7 steps: the carried total + the
    correction equals a full evaluation to 1e-10 -- the delta alone does not (the correction is not zero) --, and nothing was
    re-evaluated in between (the sampler's own counter)."""
    P = close_into_circles(problem(), (1, 3))
    a = run(P, 5, 600, 7, monkeypatch)
    assert a["own"] and a["resync"] == 512
    assert a["circ"] > 0, "the run never held a circular contig: pick another seed"
    assert a["corr"] > 1e-6, "no commit changed an own pixel: the test did not test anything"
    assert a["since_full"] >= 80 and a["counters"]["carried_totals_repaired"] == 0
    print("carried vs full: worst relative difference %.2e, largest single correction %.3e" % (a["worst"], a["corr"]))


def test_the_run_is_the_one_with_the_per_step_evaluation(monkeypatch):
    """Same 400 steps with the carry (default) and with the reference's per-step full evaluation (GRAAL_NO_OWN_PIXEL_CARRY=1): the same
    accepted moves, the same generator state at the end, the likelihood series equal to 1e-10 relative; and through the Python step
    (GRAAL_PY_STEP=1), which takes the correction itself."""
    P = close_into_circles(problem(seed=92), (2,))
    a = run(P, 6, 400, 0, monkeypatch)
    b = run(P, 6, 400, 0, monkeypatch, {"GRAAL_NO_OWN_PIXEL_CARRY": "1"})
    c = run(P, 6, 400, 0, monkeypatch, {"GRAAL_PY_STEP": "1"})
    assert a["own"] and not b["own"] and b["resync"] == 1 and c["own"]
    for other in (b, c):
        assert [t[1:] for t in a["trace"]] == [t[1:] for t in other["trace"]]
        assert np.array_equal(a["state"], other["state"])
        assert np.allclose([t[0] for t in a["trace"]], [t[0] for t in other["trace"]], rtol=REL, atol=0)


def test_a_mirrored_bin_of_mixed_rf_counts_makes_the_next_step_evaluate(monkeypatch):
    """With the reference's trans-branch indexing (kernels3.cu:3155) a mirrored bin whose sub-fragments carry different RF counts also
    changes its trans pixels with every bin outside the two contigs: the commit reports its correction as unknown and the next step
    evaluates in full.  Nine such bins among 90 (the ragged last bins of a pyramid's contigs): the run equals the per-step-evaluation run."""
    P = problem(seed=93)
    acc = P["np_sub_frags_accu"].copy()
    for b in (3, 11, 17, 29, 37, 44, 61, 73, 88):
        acc[b, 0] = 5
    P["np_sub_frags_accu"] = acc
    a = run(P, 7, 400, 0, monkeypatch, explode=True)
    b = run(P, 7, 400, 0, monkeypatch, {"GRAAL_NO_OWN_PIXEL_CARRY": "1"}, explode=True)
    assert a["own"] and not b["own"]
    assert a["counters"]["carried_totals_repaired"] > 0, "no commit mirrored one of the nine bins: pick another seed"
    assert [t[1:] for t in a["trace"]] == [t[1:] for t in b["trace"]]
    assert np.allclose([t[0] for t in a["trace"]], [t[0] for t in b["trace"]], rtol=REL, atol=0)


@pytest.mark.parametrize("seed,n_bins,nnz,delta,circles", [(101, 60, 1500, 3, (1,)), (102, 140, 5000, 5, (2, 3)), (103, 75, 2200, 10, ())])
def test_carried_total_along_other_runs(seed, n_bins, nnz, delta, circles, monkeypatch):
    """More layouts and neighbour counts (3, 5, 10), ragged bins of one to three sub-fragments, from circular and from exploded starts:
    carried + correction == a full evaluation every 11 steps, and no step needed a repair."""
    P = close_into_circles(problem(n_bins=n_bins, nnz=nnz, seed=seed), circles)
    for explode in (False, True):
        a = run(P, seed, 330, 11, monkeypatch, delta=delta, explode=explode)
        assert a["own"] and a["counters"]["carried_totals_repaired"] == 0
        assert a["worst"] < REL


@pytest.mark.parametrize("sample_param", [False, True])
def test_headless_loop_equals_the_per_step_evaluation_run(sample_param, monkeypatch):
    """The outer loop as start_EM runs it (em.run_em: init_likelihood, THEN explode_genome -- its commits' corrections pile up in front of the
    first step, which starts from a full evaluation and must void them --, then cycles of steps in runs behind graal_steps): the carried run
    and the per-step-evaluation run give the same moves and the same likelihood series (1e-10), 1,800 steps."""
    from graal_amd import em
    from tests.test_sampler_gpu import make_gpu_sampler
    P = problem(n_bins=300, nnz=20000, seed=7)

    def go(carry):
        monkeypatch.delenv("GRAAL_NO_OWN_PIXEL_CARRY", raising=False)
        if not carry:
            monkeypatch.setenv("GRAAL_NO_OWN_PIXEL_CARRY", "1")
        rng = np.random.RandomState(3)
        g = make_gpu_sampler(P, rng, reference_arithmetic="strict")
        assert g._own_corr == carry
        # (sample_param: a nuisance-parameter step behind every MCMC step, the reference GUI's default -- it compares a full evaluation under
        # test parameters with the carried score and, when it rejects, leaves the commit's correction for the next step)
        tr = em.run_em(g, 2 if sample_param else 6, 3, rng=rng, sample_param=sample_param)
        out = (np.asarray(tr.likelihood), tr.mutations(), g.engine.run_counters()["carried_totals_repaired"], np.asarray(tr.success), np.asarray(tr.fact))
        g.free_gpu()
        return out

    a, b = go(True), go(False)
    assert np.array_equal(a[1], b[1])
    assert np.allclose(a[0], b[0], rtol=REL, atol=0), float(np.max(np.abs(a[0] - b[0]) / np.abs(b[0])))
    assert a[2] == 0
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])       # (the parameter walk: accepted at the same steps, the same values)


@pytest.mark.parametrize("scrambled", [True, False])
def test_carried_run_against_the_oracle(scrambled):
    """The carried total against an anchor that is not the engine: 250 steps of start_EM (from the exploded genome / from three contigs, two of
    them circular) at three sub-fragments per bin, one RF count, generic coordinates, against the oracle run the reference's way (a full
    dense evaluation at the top of every step): accepted moves and layout bit for bit, the likelihood series to 1e-8."""
    from oracle import oracle as O
    from tests.test_strict_gpu import _run, _samplers
    P = problem(n_bins=75, nnz=2400, seed=95)
    if not scrambled:
        close_into_circles(P, (1, 3))
    ora, g, gpu_rng = _samplers(P, 9, "strict")
    assert g._own_corr
    t_ref = _run(ora, ora.rng, 4, 250, scrambled=scrambled)
    t_gpu = _run(g, gpu_rng, 4, 250, scrambled=scrambled)
    assert np.array_equal(t_gpu.mutations(), t_ref.mutations())
    assert np.allclose(t_gpu.likelihood, t_ref.likelihood, rtol=1e-8, atol=0), float(np.max(np.abs(np.asarray(t_gpu.likelihood) / np.asarray(t_ref.likelihood) - 1)))
    g.gpu_vect_frags.copy_from_gpu()
    for k in O.FIELDS:
        assert np.array_equal(getattr(g.gpu_vect_frags, k), ora.gpu_vect_frags[k]), k
    assert g.engine.run_counters()["carried_totals_repaired"] == 0
    g.free_gpu()
