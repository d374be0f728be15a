"""GPU, at the sizes bench.py runs: the engine's numbers against an INDEPENDENT checker -- the numpy sparse re-score
(oracle/sparse_numpy.py: its own float32 model on numpy's pow / exp, its own geometry, its own enumeration of the window; pinned
to the dense restatement of the reference's kernels at small sizes by tests/test_sparse_reformulation.py) applied to candidate
layouts produced by the oracle's restatement of the reference's MUTATION kernels (tests/util.oracle_candidate).  Nothing of the
engine is in the checker: not its records, not its relabel, not model_math.h.

The reference's own cross-check, debug_step_max_likelihood (cuda_lib_gl.py:2196-2220), scores each of the 13 candidates of a
neighbour with a FULL evaluation of the candidate layout; at one sub-fragment per bin the reference's candidate delta IS that
difference pixel by pixel.  Here, for every one of the 13 x K candidates of >= 3 proposals:

    strict delta (GPU)  ==  re-score(candidate) - re-score(current)          within 1e-8 x |logL|

and the engine's full evaluation == the checker's, on

* C5 (50,000 fragments / 20,000,000 contacts) in the headline state of bench.py (exploded + MCMC steps),
* C5 on its 7 original contigs (the late stage: k_gprep + k_strict2, millions of queued contacts),
* the C4 stand-in (40,000 / 8,000,000) after one cycle from the exploded genome (k_strict_flat's regime),

all on GENERIC coordinates (bp lengths 1 + Exp(660): the reference's float32 noise is part of every number compared).
The re-score is restricted to the pixels of contig(A) u contig(B) -- the set sub_compute_likelihood revisits
(kernels3.cu:3356-3380); that the restriction changes nothing is checked against two whole-genome re-scores per state."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-8          # x |logL|


def _scorer(P):
    from oracle.sparse_numpy import SparseScorer
    return SparseScorer(P["coo_row"], P["coo_col"], P["coo_val"], P["np_sub_frags_id"], P["np_sub_frags_len_bp"],
                        P["np_sub_frags_accu"], P["mean_squared_frags_per_bin"], P["param_simu"])


def _layout(smp):
    from oracle import oracle as O
    smp.gpu_vect_frags.copy_from_gpu()
    return {k: np.array(getattr(smp.gpu_vect_frags, k), dtype=np.int32, copy=True) for k in O.FIELDS}


def check_against_the_numpy_rescore(P, smp, proposals, n_whole=2, threads=None):
    """Returns (worst |delta error| / |logL|, |full error| / |logL|, candidates checked)."""
    from tests import util
    sc = _scorer(P)
    max_id = int(smp.modify_gl_cuda_buffer(0))
    cur = _layout(smp)
    assert int(cur["id_c"].max()) == max_id
    # ---- the full evaluation
    full_gpu = smp._full_likelihood()
    full_np = sc.full(cur, windowed=True)
    full_err = abs(full_gpu - full_np) / abs(full_np)
    assert full_err <= TOL, (full_gpu, full_np)
    logl = abs(full_np)
    threads = threads or max(2, min(8, (os.cpu_count() or 4) - 2))
    worst, n_checked, whole_left = 0.0, 0, n_whole
    base_whole = None
    with ThreadPoolExecutor(max_workers=threads) as pool:   # (numpy releases the GIL inside its large vector operations)
        for fA, nb in proposals:
            got = smp._candidate_deltas(fA, nb, max_id)                     # [K, 13], reference arithmetic
            assert got.shape == (len(nb), 13) and np.isfinite(got).all()
            jobs = []
            for k, fB in enumerate(nb):
                in_set = (cur["id_c"] == cur["id_c"][fA]) | (cur["id_c"] == cur["id_c"][fB])
                idx = sc.set_contacts(in_set)
                base = pool.submit(sc.restricted, cur, in_set, idx)
                for op in range(13):
                    cand, stale = util.oracle_candidate(cur, fA, int(fB), op, max_id)
                    if stale:
                        continue
                    jobs.append((k, op, base, pool.submit(sc.restricted, cand, in_set, idx), cand))
            for k, op, base, fut, cand in jobs:
                want = fut.result() - base.result()
                err = abs(got[k, op] - want) / logl
                assert err <= TOL, (fA, int(nb[k]), op, got[k, op], want, err)
                worst = max(worst, err)
                n_checked += 1
                if whole_left > 0 and op in (4, 10) and want != 0.0:
                    # the restriction itself: two whole-genome re-scores give the same difference (to their own summation noise)
                    if base_whole is None:
                        base_whole = sc.full(cur, same_bin=False, windowed=True)
                    whole = sc.full(cand, same_bin=False, windowed=True) - base_whole
                    assert abs(whole - want) <= 2e-9 * logl, (fA, int(nb[k]), op, whole, want)
                    whole_left -= 1
    assert whole_left == 0
    return worst, full_err, n_checked


def _proposals(smp, n_props, seed, K=5, pick=None):
    rng = np.random.RandomState(seed)
    n = int(smp.n_new_frags)
    props = []
    while len(props) < n_props:
        f = int(rng.randint(0, n)) if pick is None else int(pick(rng))
        nb = smp.return_neighbours(f, K)
        nb.sort()
        if len(nb):
            props.append((f, [int(x) for x in nb]))
    return props


@pytest.fixture(scope="module")
def c5_problem():
    from graal_amd import synth
    return synth.make_problem(n_bins=50000, nnz=20_000_000, n_sub=1, seed=20141217)


@pytest.mark.timeout(900)
def test_c5_headline_state_against_the_numpy_rescore(c5_problem):
    """bench.py's state: exploded genome + MCMC steps (contigs of a few fragments; k_tm prices the sets itself)."""
    import bench
    P = dict(c5_problem)
    P["S_o_A_frags"] = bench.exploded_layout(c5_problem)
    rng = np.random.RandomState(7)
    smp = bench.build_sampler(P, rng, None, 0)                  # (reference arithmetic: the default)
    assert smp.reference_arithmetic == "strict"
    smp.init_likelihood()
    order = np.arange(int(smp.n_new_frags), dtype=np.int32)
    rng.shuffle(order)
    for i in order[:2000]:
        smp.step_max_likelihood(int(i), 5)
    st = smp.engine.layout_stats()
    assert int(st[4]) >= 3                                       # contigs have begun to grow
    # proposals whose fragment already sits in a contig of several fragments (a singleton's candidates are mostly trivial)
    smp.gpu_vect_frags.copy_from_gpu()
    grown = np.flatnonzero(smp.gpu_vect_frags.l_cont >= 3)
    worst, full_err, n_c = check_against_the_numpy_rescore(P, smp, _proposals(smp, 4, 21, pick=lambda r: grown[r.randint(len(grown))]))
    print("C5 headline state: %d candidates, worst |strict delta - numpy re-score| / |logL| = %.2e, full evaluation %.2e" % (n_c, worst, full_err))
    assert n_c >= 3 * 65
    smp.free_gpu()


@pytest.mark.timeout(1500)
def test_c5_original_contigs_against_the_numpy_rescore(c5_problem):
    """bench.py's `late_stage`: the 7 original contigs of 4-10 thousand fragments (k_gprep + k_strict2 price ~1e8 pair-class
    evaluations per step, k_scan queues millions of contacts).  Every one of the 13 x 5 candidates of 3 proposals."""
    import bench
    smp = bench.build_sampler(c5_problem, np.random.RandomState(11), None, 0)
    smp.init_likelihood()
    st = smp.engine.layout_stats()
    assert int(st[0]) == 7 and int(st[4]) > 5000
    worst, full_err, n_c = check_against_the_numpy_rescore(c5_problem, smp, _proposals(smp, 3, 12))
    c = smp.engine.last_counters()
    assert c[2] > 100000                                         # the late-stage work did occur: queued contacts
    print("C5 original contigs: %d candidates, worst |strict delta - numpy re-score| / |logL| = %.2e, full evaluation %.2e" % (n_c, worst, full_err))
    assert n_c >= 3 * 65
    smp.free_gpu()


@pytest.mark.timeout(1500)
def test_c4_stand_in_mid_run_against_the_numpy_rescore():
    """BASELINE config 4's shape (40,000 bins x 1 sub-fragment, 8,000,000 contacts, generic bp lengths) after one cycle from the
    exploded genome: contigs of a few to a few hundred bins -- the regime of k_strict_flat and of the small k_strict2 grids."""
    import bench
    from graal_amd import synth
    P = synth.make_problem(n_bins=40000, nnz=8_000_000, n_sub=1, seed=2014, contig_weights=synth.C5_CONTIG_WEIGHTS)
    P["S_o_A_frags"] = bench.exploded_layout(P)
    rng = np.random.RandomState(41)
    smp = bench.build_sampler(P, rng, None, 0)
    smp.init_likelihood()
    order = np.arange(int(smp.n_new_frags), dtype=np.int32)
    rng.shuffle(order)
    for i in order:
        smp.step_max_likelihood(int(i), 5)
    st = smp.engine.layout_stats()
    assert 500 < int(st[0]) < 20000 and int(st[4]) > 16
    smp.gpu_vect_frags.copy_from_gpu()
    grown = np.flatnonzero(smp.gpu_vect_frags.l_cont >= 12)
    worst, full_err, n_c = check_against_the_numpy_rescore(P, smp, _proposals(smp, 4, 43, pick=lambda r: grown[r.randint(len(grown))]))
    print("C4 stand-in mid-run: %d candidates, worst |strict delta - numpy re-score| / |logL| = %.2e, full evaluation %.2e" % (n_c, worst, full_err))
    assert n_c >= 3 * 65
    smp.free_gpu()
