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
