"""GPU: the reference-arithmetic switches (include/graal_hip.h: GRAAL_MODE_REF_TRANS_ACCU, GRAAL_MODE_STRICT) against the oracle run
with the reference's OWN arithmetic -- `fix_trans_accu=False` (the RF-count indexing of reversed bins in the trans branch,
kernels3.cu:3155 / 3638), sub-fragments with DIFFERENT RF counts, arbitrary bp lengths (float32 kb coordinates are not exact, so
the dense reference re-prices pairs whose geometry a move leaves unchanged with rounding noise: DESIGN.md section 2).

* full likelihood with GRAAL_MODE_REF_TRANS_ACCU == dense evaluate_likelihood restatement, <= 1e-8 relative;
* strict candidate deltas == the oracle's sub_compute_likelihood restatement, <= 2e-9 x |logL| (see the test);
* strict accepted-move traces bit-exact, small maps and the C2 shape;
* the EXACT mode (reference_arithmetic="exact") on the same inputs: bound on its candidate scores and the step at which its trace departs (recorded)."""
import json
import os

import numpy as np
import pytest

from graal_amd import em, synth
from oracle import oracle as O
from tests import util
from tests.test_engine_gpu import engine_for, random_state_for, relabel_ref

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ref_problem(n_bins, nnz, seed, n_sub=3, contig_weights=(5, 3, 2), mean_len_bp=1500.0, fact=300.0, v_inter=0.03, grid_bp=None):
    """Non-uniform RF counts (1..9 per sub-fragment), generic bp lengths."""
    par = synth.make_param_simu(fact=fact, v_inter=v_inter)
    P = synth.make_problem(n_bins=n_bins, nnz=nnz, n_sub=n_sub, seed=seed, contig_weights=contig_weights, mean_len_bp=mean_len_bp,
                           accu=("random", 1, 9), param=par, grid_bp=grid_bp)
    return synth.with_dense(P)


def ref_dense(P):
    return O.DenseOracle(P["hic_matrix"], P["np_sub_frags_id"], P["np_sub_frags_len_bp"], P["np_sub_frags_accu"],
                         P["frag_dispatcher"], P["collector_id_repeats"], P["n_frags"], P["mean_squared_frags_per_bin"],
                         P["param_simu"], fix_trans_accu=False)


def ref_deltas(P, dense, s, fA, fBs, max_id):
    per_pix = np.zeros(dense.n_pix)
    base = dense.evaluate(s, per_pix)
    n = P["n_frags"]
    out = np.zeros((len(fBs), 13))
    for k, fB in enumerate(fBs):
        sub = np.nonzero((s["id_c"] == s["id_c"][fA]) | (s["id_c"] == s["id_c"][fB]))[0]
        for op in range(13):
            cand, stale = util.oracle_candidate(s, fA, fB, op, max_id)
            assert not stale
            out[k, op] = dense.sub_compute(cand, np.sort(sub), [], np.arange(n, dtype=np.int32), per_pix)
    return base, out


@pytest.mark.parametrize("seed", [3, 4])
def test_full_likelihood_with_the_reference_trans_accu_indexing(seed):
    P = ref_problem(80, 1500, seed)
    assert len(np.unique(P["np_sub_frags_accu"][P["np_sub_frags_accu"] > 0])) > 3
    dense = ref_dense(P)
    plain = O.DenseOracle(P["hic_matrix"], P["np_sub_frags_id"], P["np_sub_frags_len_bp"], P["np_sub_frags_accu"],
                          P["frag_dispatcher"], P["collector_id_repeats"], P["n_frags"], P["mean_squared_frags_per_bin"],
                          P["param_simu"], fix_trans_accu=True)
    rng = np.random.RandomState(seed)
    differs = 0
    for _ in range(4):
        s = random_state_for(P, rng, p_circ=0.4)      # (p_rev = 0.4: plenty of reversed bins)
        e = engine_for(P, s)
        e.relabel_contigs()
        want, want_plain = dense.evaluate(s), plain.evaluate(s)
        assert e.eval_full() == pytest.approx(want_plain, rel=1e-8)
        e.set_mode(ref_trans_accu=True)
        assert e.eval_full() == pytest.approx(want, rel=1e-8)
        differs += abs(want - want_plain) > 1e-5 * abs(want)
        e.close()
    assert differs >= 2          # the two arithmetics are measurably different on these inputs


@pytest.mark.parametrize("n_sub,seed,p_circ", [(3, 11, 0.0), (3, 12, 0.5), (2, 13, 0.3), (1, 14, 0.3)])
def test_strict_deltas_match_the_reference_arithmetic(n_sub, seed, p_circ):
    P = ref_problem(70, 1500, seed, n_sub=n_sub) if n_sub > 1 else \
        synth.with_dense(synth.make_problem(n_bins=70, nnz=1500, n_sub=1, seed=seed, contig_weights=(5, 3, 2), mean_len_bp=1500.0,
                                            accu=1, param=synth.make_param_simu(fact=300.0, v_inter=0.03)))
    dense = ref_dense(P)
    rng = np.random.RandomState(seed)
    worst = 0.0
    for _ in range(3):
        s = random_state_for(P, rng, n_contigs=int(rng.randint(2, 8)), p_circ=p_circ)
        max_id = relabel_ref(s)
        e = engine_for(P, s)
        e.set_mode(ref_trans_accu=True, strict=True)
        assert e.relabel_contigs() == max_id
        for _ in range(3):
            fA = int(rng.randint(P["n_frags"]))
            fBs = [int(v) for v in rng.choice(np.setdiff1d(np.arange(P["n_frags"]), [fA]), 3, replace=False)]
            base, want = ref_deltas(P, dense, s, fA, fBs, max_id)
            got = e.eval_candidates(fA, fBs, max_id)
            err = np.abs(got - want).max() / abs(base)
            # (every pixel of the affected contigs is priced twice, old and new, from slightly different float32 coordinates, so
            # last-place differences between two libms do not cancel: with the device library's powf / expf the error was up to
            # 1.5e-6 of logL.  The engine's own correctly rounded functions, model_math.h, agree with glibc's on 99.94 % of the
            # arguments: 1.5e-10 measured)
            assert err <= 2e-9, (fA, fBs, err, (got - want)[0])
            worst = max(worst, err)
        e.close()
    assert worst <= 2e-9


def _strict_deltas_on(P, states, n_props, K, seed, tol=2e-9):
    """Strict candidate deltas of n_props proposals per layout against the reference-arithmetic oracle; the worst |error| / |logL|
    and the last step's counters."""
    dense = ref_dense(P)
    rng = np.random.RandomState(seed)
    n = P["n_frags"]
    worst, c = 0.0, None
    for s in states:
        max_id = relabel_ref(s)
        e = engine_for(P, s)
        e.set_mode(ref_trans_accu=True, strict=True)
        assert e.relabel_contigs() == max_id
        for _ in range(n_props):
            fA = int(rng.randint(n))
            fBs = sorted(int(v) for v in rng.choice(np.setdiff1d(np.arange(n), [fA]), K, replace=False))
            base, want = ref_deltas(P, dense, s, fA, fBs, max_id)
            got = e.eval_candidates(fA, fBs, max_id)
            err = np.abs(got - want).max() / abs(base)
            assert err <= tol, (fA, fBs, err)
            worst = max(worst, err)
        c = e.last_counters()
        e.close()
    return worst, c


def _zero_based(P):
    s = O.copy_state(P["S_o_A_frags"])
    s["id_c"][:] -= 1
    return s


@pytest.mark.timeout(1500)
def test_strict_deltas_at_the_c3_like_shape():
    """Reference arithmetic -- the sampler's default -- against the oracle run the reference's way (fix_trans_accu=False) at the
    T. reesei level-3 stand-in's shape: 3,500 bins x 3 sub-fragments on GENERIC bp lengths with RF counts 1..9, contigs of up to
    ~730 bins (the union-set kernels: several tiles per global piece, hundreds of units per step; kernels3.cu:3259-3718)."""
    P = ref_problem(3500, 600_000, 2015, contig_weights=synth.C5_CONTIG_WEIGHTS, mean_len_bp=660.0, fact=200.0, v_inter=0.02)
    rng = np.random.RandomState(5)
    states = [_zero_based(P), random_state_for(P, rng, n_contigs=7, p_circ=0.4)]
    worst, counters = _strict_deltas_on(P, states, n_props=2, K=3, seed=6)
    print("strict deltas, C3-like shape: worst |error| / |logL| = %.3e, queued contacts %d, units %d" % (worst, counters[2], counters[1]))
    assert counters[2] > 1000 and counters[1] > 0


@pytest.mark.timeout(1500)
def test_strict_deltas_with_ten_neighbours_and_sub_fragments():
    """K = 10 (the reference's n_neighbors cap, cuda_lib_gl.py:444) in one pass, sub-fragments with RF counts 1..9, generic coordinates,
    the C2 stand-in's shape: the union of up to 11 contigs, up to 33 global pieces, 130 candidates per piece pair."""
    P = ref_problem(1086, 120_000, 2017, contig_weights=synth.C5_CONTIG_WEIGHTS, mean_len_bp=660.0, fact=200.0, v_inter=0.02)
    rng = np.random.RandomState(7)
    states = [_zero_based(P), random_state_for(P, rng, n_contigs=12, p_circ=0.3)]
    worst, counters = _strict_deltas_on(P, states, n_props=2, K=10, seed=8)
    print("strict deltas, K = 10 at the C2 shape: worst |error| / |logL| = %.3e" % worst)
    assert counters[1] > 0


def _samplers(P, seed, mode):
    from tests.test_sampler_gpu import make_gpu_sampler
    ora = O.OracleSampler(P, np.random.RandomState(seed), fix_trans_accu=False)
    gpu_rng = np.random.RandomState(seed)
    g = make_gpu_sampler(P, gpu_rng, reference_arithmetic=mode)
    return ora, g, gpu_rng


def _run(smp, rng, delta, n_steps, scrambled):
    class Stop(Exception):
        pass
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


@pytest.mark.parametrize("n_sub,seed,n_bins,nnz,delta", [(3, 21, 60, 1500, 4), (3, 22, 50, 900, 3)])
def test_strict_trace_is_bit_exact_on_generic_coordinates(n_sub, seed, n_bins, nnz, delta):
    P = ref_problem(n_bins, nnz, seed, n_sub=n_sub, contig_weights=(5, 4, 3), mean_len_bp=2000.0, fact=200.0, v_inter=0.02)
    ora, g, gpu_rng = _samplers(P, seed, "strict")
    t_ref = em.run_em(ora, 2, delta, rng=ora.rng)
    t_gpu = em.run_em(g, 2, delta, rng=gpu_rng)
    assert np.array_equal(t_gpu.mutations(), t_ref.mutations())
    assert t_gpu.n_contigs == t_ref.n_contigs and t_gpu.dist == t_ref.dist
    assert np.allclose(t_gpu.likelihood, t_ref.likelihood, rtol=1e-8, atol=0)
    g.gpu_vect_frags.copy_from_gpu()
    for k in O.FIELDS:
        assert np.array_equal(getattr(g.gpu_vect_frags, k), ora.gpu_vect_frags[k]), k
    g.free_gpu()


@pytest.mark.timeout(1500)
def test_c2_shape_strict_trace_from_the_exploded_genome():
    """400 steps of start_EM at the C2 stand-in's shape from the EXPLODED genome (main_gl.py:219: every run starts there), generic bp
    lengths and RF counts 1..9, reference arithmetic against the oracle run the reference's way: accepted moves, contig counts, genome
    distance and layout bit for bit.  (From its 7 original contigs: the next test.)"""
    P = ref_problem(1086, 120_000, 2016, contig_weights=synth.C5_CONTIG_WEIGHTS, mean_len_bp=660.0, fact=200.0, v_inter=0.02)
    n_steps = 400
    ora, g, gpu_rng = _samplers(P, 33, "strict")
    t_ref = _run(ora, ora.rng, 3, n_steps, scrambled=True)
    t_gpu = _run(g, gpu_rng, 3, n_steps, scrambled=True)
    assert len(t_gpu.id_fA) == len(t_ref.id_fA) == n_steps
    assert np.array_equal(t_gpu.mutations(), t_ref.mutations())
    assert t_gpu.n_contigs == t_ref.n_contigs and t_gpu.dist == t_ref.dist
    assert np.allclose(t_gpu.likelihood, t_ref.likelihood, rtol=1e-8, atol=0)
    g.gpu_vect_frags.copy_from_gpu()
    for k in O.FIELDS:
        assert np.array_equal(getattr(g.gpu_vect_frags, k), ora.gpu_vect_frags[k]), k
    g.free_gpu()


@pytest.mark.timeout(1500)
def test_c2_shape_reference_arithmetic_strict_exact_default_bounded():
    """C2 stand-in with generic bp lengths (660 bp fragments, contigs of ~0.4 Mb) and non-uniform RF counts, from the 7 original
    contigs (the long-contig regime, where the coordinate noise of the reference's float32 geometry is largest):
    strict mode reproduces the reference-arithmetic oracle's accepted-move trace bit for bit; the default mode's scores differ
    from it by the reference's own coordinate noise (~2e-4 of logL here -- more than the 30 logL units of the sampling window, so
    the sampled traces part after a handful of steps); both numbers are recorded (profiles/r02_default_mode_departure.json)."""
    P = ref_problem(1086, 120_000, 2016, contig_weights=synth.C5_CONTIG_WEIGHTS, mean_len_bp=660.0, fact=200.0, v_inter=0.02)
    n_steps = 300
    ora, g, gpu_rng = _samplers(P, 31, "strict")
    t_ref = _run(ora, ora.rng, 3, n_steps, scrambled=False)
    t_gpu = _run(g, gpu_rng, 3, n_steps, scrambled=False)
    assert np.array_equal(t_gpu.mutations(), t_ref.mutations())
    assert np.allclose(t_gpu.likelihood, t_ref.likelihood, rtol=1e-8, atol=0)
    g.free_gpu()
    # ---- default mode on the same inputs
    from tests.test_sampler_gpu import make_gpu_sampler
    rng_d = np.random.RandomState(31)
    d = make_gpu_sampler(P, rng_d, reference_arithmetic="exact")
    t_def = _run(d, rng_d, 3, n_steps, scrambled=False)
    mut_d, mut_r = t_def.mutations(), t_ref.mutations()
    same = np.all(mut_d == mut_r, axis=1)
    first = int(np.argmin(same)) if not same.all() else n_steps
    lik_d, lik_r = np.asarray(t_def.likelihood[:max(first, 1)]), np.asarray(t_ref.likelihood[:max(first, 1)])
    worst = float(np.max(np.abs(lik_d - lik_r) / np.abs(lik_r)))
    rec = {"shape": "C2 stand-in, 1086 bins x 3 sub-fragments, generic bp lengths (mean 660), RF counts 1..9, 7 original contigs",
           "steps": n_steps, "default_mode_first_departure_step": first, "likelihood_rel_diff_until_departure": worst}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "default_mode_departure.json"), "w") as f:
        json.dump(rec, f)
    print("default mode vs reference arithmetic:", rec)
    assert worst <= 1e-3 and first >= 1
    d.free_gpu()
