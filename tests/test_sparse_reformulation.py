"""CPU: the sparse (COO) reformulation of the likelihood, as restated in numpy (oracle/sparse_numpy.py), equals the
dense reference algorithm (oracle/graal_oracle.c) -- SURVEY.md H1.  Covers n_sub in {1, 3}, ragged bins, reversed
bins, circular contigs, and candidate deltas."""
import numpy as np
import pytest

from graal_amd import synth
from oracle import oracle as O
from oracle.sparse_numpy import SparseScorer, lf_term
from tests import util


def small_problem(n_sub, seed, n_bins=60, nnz=700, accu=1):
    par = synth.make_param_simu(fact=300.0, v_inter=0.03)
    P = synth.make_problem(n_bins=n_bins, nnz=nnz, n_sub=n_sub, seed=seed, contig_weights=(5, 3, 2), mean_len_bp=1500.0,
                           accu=accu, param=par)
    return synth.with_dense(P)


def scorers(P):
    dense = O.DenseOracle(P["hic_matrix"], P["np_sub_frags_id"], P["np_sub_frags_len_bp"], P["np_sub_frags_accu"],
                          P["frag_dispatcher"], P["collector_id_repeats"], P["n_frags"], P["mean_squared_frags_per_bin"],
                          P["param_simu"], fix_trans_accu=True)
    sparse = SparseScorer(P["coo_row"], P["coo_col"], P["coo_val"], P["np_sub_frags_id"], P["np_sub_frags_len_bp"],
                          P["np_sub_frags_accu"], P["mean_squared_frags_per_bin"], P["param_simu"])
    return dense, sparse


def test_lf_term_matches_oracle_scalar():
    for ob in [1, 2, 5, 9, 10, 11, 14, 15, 16, 40, 1000]:
        want = -(O.lik(1.0, float(ob)) + 1.0)   # lik(1, ob) = ob*log(1) - 1 - lf(ob)
        assert lf_term(np.array([ob]))[0] == pytest.approx(want, abs=3e-7)  # float32 Stirling for 10..14: libm ulps


@pytest.mark.parametrize("n_sub,seed", [(1, 3), (3, 4), (3, 5)])
def test_full_likelihood_sparse_equals_dense(n_sub, seed):
    P = small_problem(n_sub, seed, accu=1 if n_sub == 1 else 9)
    dense, sparse = scorers(P)
    rng = np.random.RandomState(seed)
    n = P["n_frags"]
    layouts = [{k: np.array(P["S_o_A_frags"][k], np.int32) for k in O.FIELDS}]
    for _ in range(4):
        s = util.random_layout(rng, n, p_circ=0.5)
        s["len_bp"][:] = P["S_o_A_frags"]["len_bp"]        # keep the real bin lengths: rebuild coordinates
        for lab in np.unique(s["id_c"]):
            m = np.nonzero(s["id_c"] == lab)[0]
            order = m[np.argsort(s["pos"][m])]
            s["start_bp"][order] = np.cumsum(s["len_bp"][order]) - s["len_bp"][order]
            s["l_cont_bp"][order] = s["len_bp"][order].sum()
        util.check_invariants(s)
        layouts.append(s)
    for s in layouts:
        want = dense.evaluate(s)
        got = sparse.full(s)
        assert got == pytest.approx(want, rel=2e-7), (n_sub, seed)  # numpy float32 pow/exp vs glibc: ulp-level differences per term


@pytest.mark.parametrize("n_sub,seed", [(1, 7), (3, 8)])
def test_candidate_delta_sparse_equals_dense(n_sub, seed):
    P = small_problem(n_sub, seed, accu=1 if n_sub == 1 else 9)
    dense, sparse = scorers(P)
    rng = np.random.RandomState(seed)
    n = P["n_frags"]
    s = util.random_layout(rng, n, n_contigs=5, p_circ=0.4)
    s["len_bp"][:] = P["S_o_A_frags"]["len_bp"]
    for lab in np.unique(s["id_c"]):
        m = np.nonzero(s["id_c"] == lab)[0]
        order = m[np.argsort(s["pos"][m])]
        s["start_bp"][order] = np.cumsum(s["len_bp"][order]) - s["len_bp"][order]
        s["l_cont_bp"][order] = s["len_bp"][order].sum()
    max_id = int(s["id_c"].max())
    base = sparse.full(s, same_bin=False)
    per_pix = np.zeros(dense.n_pix)
    dense.evaluate(s, per_pix)
    for _ in range(4):
        fA, fB = rng.choice(n, 2, replace=False)
        sub = np.nonzero((s["id_c"] == s["id_c"][fA]) | (s["id_c"] == s["id_c"][fB]))[0]
        for op in range(13):
            cand, _ = util.oracle_candidate(s, fA, fB, op, max_id)
            want = dense.sub_compute(cand, np.sort(sub), [], np.arange(n, dtype=np.int32), per_pix)
            got = sparse.full(cand, same_bin=False) - base
            assert got == pytest.approx(want, abs=1e-7 * abs(base)), (fA, fB, op)  # float32 libm noise of two full re-scores


def _layout_with_real_lengths(P, rng, n_contigs=5, p_circ=0.4):
    n = P["n_frags"]
    s = util.random_layout(rng, n, n_contigs=n_contigs, p_circ=p_circ)
    s["len_bp"][:] = P["S_o_A_frags"]["len_bp"]
    for lab in np.unique(s["id_c"]):
        m = np.nonzero(s["id_c"] == lab)[0]
        order = m[np.argsort(s["pos"][m])]
        s["start_bp"][order] = np.cumsum(s["len_bp"][order]) - s["len_bp"][order]
        s["l_cont_bp"][order] = s["len_bp"][order].sum()
    return s


@pytest.mark.parametrize("n_sub,seed", [(1, 11), (3, 12), (3, 13)])
def test_windowed_enumeration_of_the_cis_mass_equals_the_offset_loop(n_sub, seed):
    """mass_cis_windowed (one binary search per sub-fragment, pairs priced in blocks: the form the full-size checker of the GPU
    suite can afford) against mass_cis (the offset loop the dense oracle pinned above), also with a tiny block size, a window
    shorter than a bin and both same-bin settings."""
    rng = np.random.RandomState(seed)
    for d_max in (None, 2.5):
        par = synth.make_param_simu(fact=300.0, v_inter=0.03, d_max=d_max)
        P = synth.with_dense(synth.make_problem(n_bins=80, nnz=900, n_sub=n_sub, seed=seed, contig_weights=(5, 3, 2), mean_len_bp=1500.0,
                                                accu=1 if n_sub == 1 else 9, param=par))
        _, sparse = scorers(P)
        for _ in range(3):
            s = _layout_with_real_lengths(P, rng, n_contigs=int(rng.randint(1, 7)), p_circ=0.5)
            for same_bin in (True, False):
                want = sparse.mass_cis(s, same_bin=same_bin)
                for block in (1 << 22, 37):
                    got = sparse.mass_cis_windowed(s, same_bin=same_bin, block=block)
                    assert got == pytest.approx(want, rel=1e-12, abs=1e-9), (n_sub, seed, d_max, same_bin, block)
            assert sparse.full(s, windowed=True) == pytest.approx(sparse.full(s), rel=1e-13)


@pytest.mark.parametrize("n_sub,seed", [(1, 17), (3, 18)])
def test_restricted_rescore_gives_the_candidate_deltas(n_sub, seed):
    """SparseScorer.restricted: the difference of two re-scores of contig(A) u contig(B) only == the difference of two FULL re-scores
    == the dense oracle's sub_compute_likelihood, for all 13 candidates of a few proposals (circular contigs, reversed bins)."""
    P = small_problem(n_sub, seed, n_bins=70, nnz=900, accu=1 if n_sub == 1 else 9)
    dense, sparse = scorers(P)
    rng = np.random.RandomState(seed)
    n = P["n_frags"]
    s = _layout_with_real_lengths(P, rng)
    max_id = int(s["id_c"].max())
    base = sparse.full(s, same_bin=False)
    per_pix = np.zeros(dense.n_pix)
    dense.evaluate(s, per_pix)
    for _ in range(4):
        fA, fB = rng.choice(n, 2, replace=False)
        in_set = (s["id_c"] == s["id_c"][fA]) | (s["id_c"] == s["id_c"][fB])
        idx = sparse.set_contacts(in_set)
        base_r = sparse.restricted(s, in_set, idx)
        for op in range(13):
            cand, _ = util.oracle_candidate(s, fA, fB, op, max_id)
            got = sparse.restricted(cand, in_set, idx) - base_r
            assert got == pytest.approx(sparse.full(cand, same_bin=False) - base, abs=1e-11 * abs(base)), (fA, fB, op)
            want = dense.sub_compute(cand, np.sort(np.nonzero(in_set)[0]), [], np.arange(n, dtype=np.int32), per_pix)
            assert got == pytest.approx(want, abs=1e-7 * abs(base)), (fA, fB, op)


def test_own_pair_counts_are_the_diagonal_pixels_contacts():
    """graal_amd.sampler.own_pair_counts (the table graal_upload_own_obs takes: the observed contacts between the sub-fragments of ONE bin,
    by data-slot pair) against a loop over the contact list."""
    from graal_amd import synth
    from graal_amd.sampler import own_pair_counts
    P = synth.make_problem(n_bins=40, nnz=3000, n_sub=3, seed=5, contig_weights=(3, 2), mean_len_bp=1500.0, accu=9)
    ids = P["np_sub_frags_id"]
    got = own_pair_counts((P["coo_row"], P["coo_col"], P["coo_val"]), ids, P["init_n_sub_frags"])
    want = np.zeros((len(ids), 3), dtype=np.float32)
    where = {}
    for b in range(len(ids)):
        for k in range(ids[b, 3]):
            where[int(ids[b, k])] = (b, k)
    for r, c, v in zip(P["coo_row"], P["coo_col"], P["coo_val"]):
        (b1, k1), (b2, k2) = where[int(r)], where[int(c)]
        if b1 == b2 and k1 != k2:
            lo, hi = min(k1, k2), max(k1, k2)
            want[b1, hi - 1 if lo == 0 else 2] = v
    assert want.sum() > 0
    assert np.array_equal(got, want)
