"""GPU parity tests of the raw engine (C ABI via graal_amd.lib.Engine) against the oracle.

Run on the MI355X box with ``pytest -m gpu``.  Everything goes through libgraal_hip.so; the oracle is only the checker.
"""
import os

import numpy as np
import pytest

from graal_amd import synth
from oracle import oracle as O
from oracle.sparse_numpy import SparseScorer
from tests import util

pytestmark = pytest.mark.gpu


def make(n_sub, seed, n_bins=80, nnz=1500, accu=None, d_max=None, weights=(5, 3, 2), grid_bp=None):
    accu = (1 if n_sub == 1 else 9) if accu is None else accu
    par = synth.make_param_simu(fact=300.0, v_inter=0.03, d_max=d_max)
    P = synth.make_problem(n_bins=n_bins, nnz=nnz, n_sub=n_sub, seed=seed, contig_weights=weights, mean_len_bp=1500.0,
                           accu=accu, param=par, grid_bp=grid_bp)
    return synth.with_dense(P)


def engine_for(P, state=None):
    from graal_amd.lib import Engine
    e = Engine(0)
    e.upload_subfrags(P["np_sub_frags_id"], P["np_sub_frags_len_bp"], P["np_sub_frags_accu"], P["init_n_sub_frags"],
                      P["mean_squared_frags_per_bin"])
    e.upload_contacts(P["coo_row"], P["coo_col"], P["coo_val"])
    e.set_params(P["param_simu"])
    e.upload_frags(state if state is not None else P["S_o_A_frags"])
    return e


def dense_for(P):
    return O.DenseOracle(P["hic_matrix"], P["np_sub_frags_id"], P["np_sub_frags_len_bp"], P["np_sub_frags_accu"],
                         P["frag_dispatcher"], P["collector_id_repeats"], P["n_frags"], P["mean_squared_frags_per_bin"],
                         P["param_simu"], fix_trans_accu=True)


def relabel_ref(state):
    """cuda_lib_gl.py:1697-1722 with a stable argsort (oracle.OracleSampler.modify_gl_cuda_buffer)."""
    idc_un, idx_un = np.unique(state["id_c"], return_index=True)
    ordl = np.argsort(state["l_cont"][idx_un], kind="stable")
    o2n = np.zeros(idc_un.max() + 1, np.int32)
    o2n[idc_un[ordl]] = np.arange(len(idc_un), dtype=np.int32)
    state["id_c"][:] = o2n[state["id_c"]]
    return len(idc_un) - 1


def random_state_for(P, rng, **kw):
    n = P["n_frags"]
    s = util.random_layout(rng, n, **kw)
    s["len_bp"][:] = P["S_o_A_frags"]["len_bp"]
    for lab in np.unique(s["id_c"]):
        m = np.nonzero(s["id_c"] == lab)[0]
        order = m[np.argsort(s["pos"][m])]
        s["start_bp"][order] = np.cumsum(s["len_bp"][order]) - s["len_bp"][order]
        s["l_cont_bp"][order] = s["len_bp"][order].sum()
    return s


def test_library_loads_on_gpu():
    from graal_amd import lib
    assert lib.load().graal_abi_version() == lib.ABI_VERSION


@pytest.mark.parametrize("n_sub,seed", [(1, 1), (3, 2)])
def test_upload_download_relabel_stats(n_sub, seed):
    P = make(n_sub, seed)
    rng = np.random.RandomState(seed)
    s = random_state_for(P, rng, p_circ=0.3)
    s["id_c"][:] = s["id_c"] * 2 + 1  # sparse, non-dense labels
    e = engine_for(P, s)
    got = e.download_frags()
    for k in O.FIELDS:
        assert np.array_equal(got[k], s[k])
    st = e.layout_stats()
    heads = s["start_bp"] == 0
    assert list(st) == [len(np.unique(s["id_c"])), s["l_cont"].sum(), heads.sum(), s["l_cont_bp"][heads].sum(),
                        s["l_cont"].max(), s["l_cont"].min(), (s["circ"] == 1).sum(), 0]
    max_id = e.relabel_contigs()
    want = O.copy_state(s)
    assert max_id == relabel_ref(want)
    got = e.download_frags()
    for k in O.FIELDS:
        assert np.array_equal(got[k], want[k]), k


@pytest.mark.parametrize("n_sub,seed", [(1, 3), (3, 4), (3, 5)])
def test_full_likelihood(n_sub, seed):
    P = make(n_sub, seed)
    dense = dense_for(P)
    sparse = SparseScorer(P["coo_row"], P["coo_col"], P["coo_val"], P["np_sub_frags_id"], P["np_sub_frags_len_bp"],
                          P["np_sub_frags_accu"], P["mean_squared_frags_per_bin"], P["param_simu"])
    rng = np.random.RandomState(seed)
    states = [O.copy_state(P["S_o_A_frags"])] + [random_state_for(P, rng, p_circ=0.5) for _ in range(3)]
    for s in states:
        e = engine_for(P, s)
        e.relabel_contigs()
        got = e.eval_full()
        want = dense.evaluate(s)
        assert got == pytest.approx(want, rel=1e-8)          # north_star tolerance is 1e-5 relative
        assert got == pytest.approx(sparse.full(s), rel=1e-6)   # (numpy's own float32 power / exp: last-place differences, 4e-8 measured)
        e.close()


@pytest.mark.parametrize("n_sub,seed", [(1, 6), (3, 7), (3, 8), (1, 9)])
def test_apply_move_bit_exact(n_sub, seed):
    P = make(n_sub, seed, n_bins=40, nnz=300)
    rng = np.random.RandomState(seed)
    for _ in range(6):
        s = random_state_for(P, rng, p_circ=0.4)
        max_id = relabel_ref(s)
        for _ in range(3):
            fA, fB = [int(v) for v in rng.choice(P["n_frags"], 2, replace=False)]
            for op in range(13):
                e = engine_for(P, s)
                assert e.apply_move(fA, fB, op, max_id) == 0
                got = e.download_frags()
                want, stale = util.oracle_candidate(s, fA, fB, op, max_id)
                assert not stale
                for k in O.FIELDS:
                    assert np.array_equal(got[k], want[k]), (fA, fB, op, k)
                e.close()


def oracle_deltas(P, dense, s, fA, fBs, max_id):
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


@pytest.mark.parametrize("n_sub,seed,p_circ", [(1, 11, 0.0), (1, 12, 0.5), (3, 13, 0.0), (3, 14, 0.5), (1, 15, 0.3)])
def test_candidate_deltas(n_sub, seed, p_circ):
    # grid_bp=2000: float32 kb coordinates are exact, so the dense oracle's values are shift invariant and the
    # comparison is limited only by libm ulps on the pairs the engine actually re-evaluates
    P = make(n_sub, seed, n_bins=70, nnz=1500, grid_bp=2000)
    dense = dense_for(P)
    rng = np.random.RandomState(seed)
    worst = 0.0
    for _ in range(4):
        s = random_state_for(P, rng, n_contigs=int(rng.randint(2, 8)), p_circ=p_circ, len_grid=1)
        max_id = relabel_ref(s)
        e = engine_for(P, s)
        assert e.relabel_contigs() == max_id
        for _ in range(3):
            fA = int(rng.randint(P["n_frags"]))
            fBs = [int(v) for v in rng.choice(np.setdiff1d(np.arange(P["n_frags"]), [fA]), 3, replace=False)]
            base, want = oracle_deltas(P, dense, s, fA, fBs, max_id)
            got = e.eval_candidates(fA, fBs, max_id)
            tol = 1e-8 * abs(base)
            assert np.all(np.abs(got - want) <= tol), (fA, fBs, np.abs(got - want).max(), tol, got - want)
            worst = max(worst, np.abs(got - want).max() / abs(base))
        e.close()
    assert worst < 1e-8


def test_candidates_generic_coordinates_within_tolerance():
    """Arbitrary bp lengths: the dense float32 reference carries coordinate rounding noise on pairs whose geometry a
    move does not change; the engine treats those as exactly unchanged.  Agreement stays far inside 1e-5 of logL."""
    P = make(3, 21, n_bins=70, nnz=1500)
    dense = dense_for(P)
    rng = np.random.RandomState(21)
    s = random_state_for(P, rng, n_contigs=4, p_circ=0.2)
    max_id = relabel_ref(s)
    e = engine_for(P, s)
    e.relabel_contigs()
    for _ in range(4):
        fA = int(rng.randint(P["n_frags"]))
        fBs = [int(v) for v in rng.choice(np.setdiff1d(np.arange(P["n_frags"]), [fA]), 4, replace=False)]
        base, want = oracle_deltas(P, dense, s, fA, fBs, max_id)
        got = e.eval_candidates(fA, fBs, max_id)
        assert np.all(np.abs(got - want) <= 1e-5 * abs(base))


def test_engine_errors_are_loud():
    from graal_amd.lib import Engine, GraalError
    e = Engine(0)
    with pytest.raises(GraalError):
        e.eval_full()
    with pytest.raises(GraalError):
        e.set_params([1, 9.6, 0.1, -1.5, 3, 100, 10, 0.0])  # v_inter must be > 0
    P = make(1, 1, n_bins=20, nnz=50)
    bad = O.copy_state(P["S_o_A_frags"])
    bad["rep"][3] = 1
    e.upload_subfrags(P["np_sub_frags_id"], P["np_sub_frags_len_bp"], P["np_sub_frags_accu"], P["init_n_sub_frags"], 1.0)
    with pytest.raises(GraalError):
        e.upload_frags(bad)
    with pytest.raises(GraalError):
        e.upload_contacts([3], [3], [1])  # row < col required


def test_unsorted_contact_list_gives_the_same_deltas():
    """The wave-uniform row-range shortcut of the scan is only taken for a row-sorted list; any order must work."""
    P = make(1, 31, n_bins=70, nnz=1500, grid_bp=2000)
    rng = np.random.RandomState(31)
    s = random_state_for(P, rng, n_contigs=5, p_circ=0.2)
    max_id = relabel_ref(s)
    e1 = engine_for(P, s)
    e1.relabel_contigs()
    perm = rng.permutation(len(P["coo_row"]))
    P2 = dict(P)
    P2["coo_row"], P2["coo_col"], P2["coo_val"] = P["coo_row"][perm], P["coo_col"][perm], P["coo_val"][perm]
    e2 = engine_for(P2, s)
    e2.relabel_contigs()
    for _ in range(5):
        fA = int(rng.randint(P["n_frags"]))
        fBs = [int(v) for v in rng.choice(np.setdiff1d(np.arange(P["n_frags"]), [fA]), 4, replace=False)]
        assert np.array_equal(e1.eval_candidates(fA, fBs, max_id), e2.eval_candidates(fA, fBs, max_id))
    assert e1.eval_full() == e2.eval_full()


# ------------------------------------------------------------------------------------------------ edge cases
def _deltas_match(P, s, fA, fBs, tol_rel=1e-8):
    dense = dense_for(P)
    want_state = O.copy_state(s)
    max_id = relabel_ref(want_state)
    e = engine_for(P, s)
    assert e.relabel_contigs() == max_id
    base, want = oracle_deltas(P, dense, want_state, fA, fBs, max_id)
    got = e.eval_candidates(fA, fBs, max_id)
    e.close()
    assert got.shape == (len(fBs), 13)
    assert np.all(np.abs(got - want) <= tol_rel * max(abs(base), 1.0)), np.abs(got - want).max()
    return got, want


def test_more_than_eight_neighbours_take_two_scan_passes():
    P = make(1, 51, n_bins=60, nnz=900, grid_bp=2000)
    s = random_state_for(P, np.random.RandomState(51), n_contigs=6, p_circ=0.2)
    _deltas_match(P, s, 7, [1, 2, 3, 10, 11, 20, 30, 31, 40, 41, 50])


def test_empty_contact_list():
    P = make(1, 52, n_bins=30, nnz=200, grid_bp=2000)
    P["coo_row"], P["coo_col"], P["coo_val"] = P["coo_row"][:0], P["coo_col"][:0], P["coo_val"][:0]
    P["hic_matrix"] = np.zeros_like(P["hic_matrix"])
    s = random_state_for(P, np.random.RandomState(52), n_contigs=3, p_circ=0.0)
    got, want = _deltas_match(P, s, 4, [5, 17, 22])
    e = engine_for(P, s)
    e.relabel_contigs()
    assert e.eval_full() == pytest.approx(dense_for(P).evaluate(s), rel=1e-8)   # pure expected mass
    assert np.any(np.abs(want) > 0)


def test_identical_fragments_score_a_no_op():
    P = make(1, 53, n_bins=30, nnz=200, grid_bp=2000)
    s = random_state_for(P, np.random.RandomState(53), n_contigs=3)
    max_id = relabel_ref(s)
    e = engine_for(P, s)
    e.relabel_contigs()
    got = e.eval_candidates(9, [9, 12], max_id)
    assert np.all(got[0] == 0) and np.any(got[1] != 0)
    before = e.download_frags()
    assert e.apply_move(9, 9, 6, max_id) == 0
    after = e.download_frags()
    for k in O.FIELDS:
        assert np.array_equal(before[k], after[k])


def test_window_smaller_than_a_bin_and_single_contig():
    # d_max = 0.5 kb < every bin: every cis pair is priced at the trans level; one single contig
    P = make(3, 54, n_bins=40, nnz=600, d_max=0.5, weights=(1,), grid_bp=2000)
    s = O.copy_state(P["S_o_A_frags"])
    _deltas_match(P, s, 10, [11, 30, 3])


def test_large_counts_use_the_stirling_branches():
    P = make(1, 55, n_bins=40, nnz=500, grid_bp=2000)
    v = P["coo_val"].copy()
    v[::3] = 15 + (np.arange(len(v[::3])) % 200)      # ob >= 15: Stirling in double
    v[1::3] = 10 + (np.arange(len(v[1::3])) % 5)      # 10..14: float32 Stirling (kernels3.cu:80-93)
    P["coo_val"] = v
    P["hic_matrix"] = synth.dense_from_coo(P["coo_row"], P["coo_col"], v, P["init_n_sub_frags"])
    s = random_state_for(P, np.random.RandomState(55), n_contigs=4, p_circ=0.3)
    e = engine_for(P, s)
    e.relabel_contigs()
    assert e.eval_full() == pytest.approx(dense_for(P).evaluate(s), rel=1e-8)
    _deltas_match(P, s, 3, [4, 20, 33])


def test_two_sub_fragments_per_bin():
    par = synth.make_param_simu(fact=300.0, v_inter=0.03)
    P = synth.with_dense(synth.make_problem(n_bins=50, nnz=900, n_sub=2, seed=56, contig_weights=(5, 3, 2), mean_len_bp=1500.0,
                                            accu=3, param=par, grid_bp=2000))
    s = random_state_for(P, np.random.RandomState(56), n_contigs=5, p_circ=0.3)
    e = engine_for(P, s)
    e.relabel_contigs()
    assert e.eval_full() == pytest.approx(dense_for(P).evaluate(s), rel=1e-8)
    _deltas_match(P, s, 8, [9, 25, 41, 2])


def test_every_fragment_a_singleton():
    P = make(3, 57, n_bins=40, nnz=700, grid_bp=2000)
    n = P["n_frags"]
    s = O.copy_state(P["S_o_A_frags"])
    s["pos"][:] = 0; s["id_c"][:] = np.arange(n); s["start_bp"][:] = 0; s["prev"][:] = -1; s["next"][:] = -1
    s["l_cont"][:] = 1; s["l_cont_bp"][:] = s["len_bp"]
    _deltas_match(P, s, 5, [6, 7, 30])


@pytest.mark.parametrize("n_sub,seed,p_circ", [(1, 61, 0.0), (1, 62, 0.4), (3, 63, 0.3)])
def test_short_contigs_finished_by_the_table_kernel(n_sub, seed, p_circ):
    """Many short contigs (<= 8 fragments: the regime of an exploded genome): the step is finished by the last block of the
    table kernel, from LDS-resident geometry.  Against the oracle, and bit-identical to the path through the finishing
    kernel (graal_set_finisher(0))."""
    P = make(n_sub, seed, n_bins=90, nnz=2500, grid_bp=2000)
    rng = np.random.RandomState(seed)
    for _ in range(50):
        s = random_state_for(P, rng, n_contigs=36, p_circ=p_circ)
        if s["l_cont"].max() <= 8:
            break
    assert s["l_cont"].max() <= 8
    max_id = relabel_ref(s)
    e1, e2 = engine_for(P, s), engine_for(P, s)
    e2.set_finisher(False)
    assert e1.relabel_contigs() == max_id and e2.relabel_contigs() == max_id
    dense = dense_for(P)
    for _ in range(6):
        fA = int(rng.randint(P["n_frags"]))
        fBs = sorted(int(v) for v in rng.choice(np.setdiff1d(np.arange(P["n_frags"]), [fA]), 5, replace=False))
        got = e1.eval_candidates(fA, fBs, max_id)
        assert np.array_equal(got, e2.eval_candidates(fA, fBs, max_id))
        base, want = oracle_deltas(P, dense, s, fA, fBs, max_id)
        assert np.all(np.abs(got - want) <= 1e-8 * max(abs(base), 1.0)), np.abs(got - want).max()
    e1.close(); e2.close()


@pytest.mark.parametrize("n_sub,seed,p_circ,n_bins", [(1, 31, 0.0, 60), (1, 32, 0.5, 60), (3, 33, 0.3, 60), (1, 34, 0.1, 1300)])
def test_incremental_relabel_matches_the_sort(n_sub, seed, p_circ, n_bins):
    """graal_begin_step after one commit derives the new ranking by counting (k_incr); it must give
    the labels of the stable sort (cuda_lib_gl.py:1697-1722) and the same position index as a fresh upload."""
    P = make(n_sub, seed, n_bins=n_bins, nnz=800)
    rng = np.random.RandomState(seed)
    s = random_state_for(P, rng, p_circ=p_circ, n_contigs=None if n_bins < 1000 else 1150)   # (> 1,024 contigs: mostly singletons)
    n = P["n_frags"]
    e = engine_for(P, s)
    ref = O.copy_state(s)
    _, max_id = e.begin_step()
    assert max_id == relabel_ref(ref)
    for step in range(120):
        fA, fB = [int(v) for v in rng.choice(n, 2, replace=False)]
        op = int(rng.randint(13))
        e.apply_move(fA, fB, op, max_id, wait=False)
        ref, stale = util.oracle_candidate(ref, fA, fB, op, max_id)
        assert not stale
        stats, max_id = e.begin_step()
        assert max_id == relabel_ref(ref), step
        got = e.download_frags()
        for k in O.FIELDS:
            assert np.array_equal(got[k], ref[k]), (step, op, k)
        heads = ref["start_bp"] == 0
        assert list(stats[:6]) == [max_id + 1, ref["l_cont"].sum(), heads.sum(), ref["l_cont_bp"][heads].sum(),
                                   ref["l_cont"].max(), ref["l_cont"].min()]
        if step % 10 == 0:  # the position index / offsets feed the candidate tables: compare with a fresh engine
            fBs = np.array([int(v) for v in rng.choice(np.setdiff1d(np.arange(n), [fA]), 4, replace=False)], np.int32)
            e2 = engine_for(P, ref)
            _, m2 = e2.begin_step()
            assert m2 == max_id
            assert np.array_equal(e.eval_candidates(fA, fBs, max_id), e2.eval_candidates(fA, fBs, max_id))
            assert np.array_equal(e.eval_full_q(), e2.eval_full_q())
            e2.close()
    e.close()


def test_layout_stats_behind_a_commit_does_not_disturb_the_relabel():
    """graal_layout_stats between a commit and the next graal_begin_step reports the statistics of the layout BEHIND the commit; it used
    to take their contig count over as the ranked layout's, and the incremental relabel that followed counted from the wrong number
    whenever the commit had changed the number of contigs (bench.py's pair behind its MCMC warm-up: misplaced entries of the position
    index, and -- where the index's last buffer ended a mapped region -- the GPU fault of its two-rank rehearsal)."""
    P = make(1, 77, n_bins=200, nnz=2000)
    rng = np.random.RandomState(77)
    s = random_state_for(P, rng, p_circ=0.2, n_contigs=60)
    n = P["n_frags"]
    e = engine_for(P, s)
    ref = O.copy_state(s)
    _, max_id = e.begin_step()
    assert max_id == relabel_ref(ref)
    changed = 0
    for step in range(80):
        fA, fB = [int(v) for v in rng.choice(n, 2, replace=False)]
        op = int(rng.randint(13))
        n_before = max_id + 1
        e.apply_move(fA, fB, op, max_id, wait=False)
        ref, stale = util.oracle_candidate(ref, fA, fB, op, max_id)
        assert not stale
        st = e.layout_stats()                              # (the committed, not yet relabelled layout)
        stats, max_id = e.begin_step()
        assert max_id == relabel_ref(ref), step
        assert int(st[0]) == max_id + 1
        changed += int(max_id + 1 != n_before)
        got = e.download_frags()
        for k in O.FIELDS:
            assert np.array_equal(got[k], ref[k]), (step, op, k)
        if step % 8 == 0:  # the position index feeds the candidate tables: compare with a fresh engine
            fBs = np.array([int(v) for v in rng.choice(np.setdiff1d(np.arange(n), [fA]), 4, replace=False)], np.int32)
            e2 = engine_for(P, ref)
            _, m2 = e2.begin_step()
            assert np.array_equal(e.eval_candidates(fA, fBs, max_id), e2.eval_candidates(fA, fBs, max_id))
            e2.close()
    assert changed > 10, "few commits changed the number of contigs: the test did not test much"
    e.close()


def test_two_commits_between_begin_steps_fall_back_to_the_sort():
    P = make(1, 35, n_bins=50, nnz=500)
    rng = np.random.RandomState(35)
    s = random_state_for(P, rng, p_circ=0.2)
    e = engine_for(P, s)
    ref = O.copy_state(s)
    _, max_id = e.begin_step()
    relabel_ref(ref)
    for op, (fA, fB) in [(2, (3, 17)), (10, (8, 30))]:
        e.apply_move(fA, fB, op, max_id, wait=False)
        ref, _ = util.oracle_candidate(ref, fA, fB, op, max_id)
        max_id += 2  # labels of the second commit must not collide with the first one's fresh labels
    _, max_id = e.begin_step()
    assert max_id == relabel_ref(ref)
    got = e.download_frags()
    for k in O.FIELDS:
        assert np.array_equal(got[k], ref[k]), k
    e.close()


def test_model_math_is_correctly_rounded(tmp_path):
    """The contact model's powf / expf / log (graal_amd/csrc/model_math.h: self-contained, double precision inside) return the
    CORRECTLY ROUNDED float32 -- the one value every libm should agree on; glibc's powf, which the oracle calls, misses it for
    0.06 % of the arguments, the ROCm device library's for 25 %.  tools/model_math_check.hip compares them with the device
    library's double-precision pow / exp / log over every float32 of the model's ranges."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = str(tmp_path / "model_math_check")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value", "-o", exe,
                           os.path.join(root, "tools", "model_math_check.hip")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600, check=True).stdout
    print(out)
    pw = [int(re.search(r"mismatches (\d+)", l).group(1)) for l in out.splitlines() if l.startswith("pow y=")]
    ex = [int(re.search(r"mismatches (\d+)", l).group(1)) for l in out.splitlines() if l.startswith("exp:")]
    ln = [int(re.search(r"difference (\d+) ulp", l).group(1)) for l in out.splitlines() if l.startswith("ln:")]
    # (2.6e8 arguments per exponent; a handful within 1e-7 ulp of a rounding tie may fall either way in BOTH implementations)
    assert len(pw) == 8 and max(pw) <= 40, out
    assert len(ex) == 1 and ex[0] <= 200, out
    assert len(ln) == 1 and ln[0] <= 16, out


def test_non_finite_terms_surface_as_nan_not_as_arbitrary_numbers():
    """An expected value that overflows float32 makes the reference's evaluate_likelihood_double return +-inf / NaN
    (kernels3.cu:191-210).  The engine's int64 sums cannot hold that: such a term flags its candidate (or the full
    evaluation) and the host reports NaN -- never a saturated, finite-looking number."""
    P = make(1, 58, n_bins=40, nnz=500, grid_bp=2000)
    s = random_state_for(P, np.random.RandomState(58), n_contigs=3, p_circ=0.0)
    max_id = relabel_ref(s)
    e = engine_for(P, s)
    e.relabel_contigs()
    ok = e.eval_candidates(3, [4, 20, 33], max_id)
    assert np.isfinite(ok).all() and np.isfinite(e.eval_full())
    par = np.array(P["param_simu"], dtype=np.float32)
    par[6] = 3.0e38                      # fact: every cis expected value inside the window overflows to +inf
    e.set_params(par)
    assert np.isnan(e.eval_full())
    got = e.eval_candidates(3, [4, 20, 33], max_id)
    assert np.isnan(got).any()
    e.set_params(P["param_simu"])        # and the flags do not stick
    assert np.array_equal(e.eval_candidates(3, [4, 20, 33], max_id), ok) and np.isfinite(e.eval_full())
    e.close()


@pytest.mark.parametrize("n_sub,n_bins,K", [(1, 1500, 3), (1, 1500, 5), (3, 700, 3)])
def test_window_of_hundreds_of_fragments_long_pieces(n_sub, n_bins, K):
    """A contact model whose window (d_max) spans hundreds of fragments on contigs of ~500-750 bins: mass tasks between pieces
    of several hundred fragments, walks of several 64-fragment tiles and several segments per work item (128 fragments per
    segment with one sub-fragment per bin) -- the shape of C5 on its original contigs, at a size the dense oracle handles."""
    par = synth.make_param_simu(fact=1.0e4, v_inter=1.0e-3)          # d_max ~ 3,166 kb: the window is the whole contig
    P = synth.with_dense(synth.make_problem(n_bins=n_bins, nnz=40000, n_sub=n_sub, seed=71, contig_weights=(1, 1), mean_len_bp=1500.0,
                                            accu=1 if n_sub == 1 else 9, param=par, grid_bp=2000))
    dense = dense_for(P)
    s = O.copy_state(P["S_o_A_frags"]); s["id_c"][:] -= 1
    max_id = relabel_ref(s)
    e = engine_for(P, s)
    assert e.relabel_contigs() == max_id
    rng = np.random.RandomState(72)
    n = P["n_frags"]
    for trial in range(3):
        fA = int(rng.randint(n))
        if trial == 0:   # neighbours next to fA in its own contig (what the proposal usually draws)
            fBs = sorted(set(int(v) for v in np.clip(fA + np.array([-2, -1, 1, 2, 3, 5, 8][:K]), 0, n - 1)) - {fA})
        else:
            fBs = sorted(int(v) for v in rng.choice(np.setdiff1d(np.arange(n), [fA]), K, replace=False))
        base, want = oracle_deltas(P, dense, s, fA, fBs, max_id)
        got = e.eval_candidates(fA, fBs, max_id)
        err = np.abs(got - want).max()
        assert err <= 1e-8 * abs(base), (trial, fA, fBs, err / abs(base), got[0], want[0])
    e.close()


@pytest.mark.parametrize("op", [0, 1, 8])
def test_ops_that_ignore_the_neighbour_are_applied_with_identical_fragments(op):
    """explode_genome commits (i, 0, op 0) for every fragment -- also (0, 0, 0) (cuda_lib_gl.py:1539-1544): eject, flip and swap
    activity never look at fB, so fA == fB is no reason to skip them (ops that do involve fB stay a no-op there)."""
    P = make(1, 59, n_bins=30, nnz=200, grid_bp=2000)
    s = random_state_for(P, np.random.RandomState(59), n_contigs=3)
    max_id = relabel_ref(s)
    f = int(np.nonzero(s["l_cont"] > 2)[0][0])
    e = engine_for(P, s)
    e.relabel_contigs()
    assert e.apply_move(f, f, op, max_id) == 0
    got = e.download_frags()
    want, _ = util.oracle_candidate(s, f, f, op, max_id)
    for k in O.FIELDS:
        assert np.array_equal(got[k], want[k]), k
    e.close()


def _full_q_of_fixture(n_sub, seed):
    P = make(n_sub, seed, n_bins=300, nnz=20000)
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(3):
        s = random_state_for(P, rng, n_contigs=12, p_circ=0.4)
        e = engine_for(P, s)
        e.relabel_contigs()
        out.append([int(v) for v in e.eval_full_q()])
        e.close()
    return out


def _print_full_q():   # body of the child process below
    import json
    print("FULLQ " + json.dumps([_full_q_of_fixture(1, 91), _full_q_of_fixture(3, 92)]))


@pytest.mark.timeout(600)
def test_compact_records_of_the_full_evaluation_are_bit_identical():
    """k_full_nnz_u (8-byte records, used when every sub-fragment has the same RF count) against k_full_nnz (16-byte records;
    GRAAL_FULL_NO_COMPACT=1 in a child process -- the switch is read once per process): the same int64 sums, circular contigs
    included."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GRAAL_FULL_NO_COMPACT="1", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = subprocess.run([sys.executable, "-c", "import tests.test_engine_gpu as t; t._print_full_q()"], cwd=root, env=env,
                         capture_output=True, text=True, timeout=500)
    assert out.returncode == 0, out.stderr[-3000:]
    want = json.loads([l for l in out.stdout.splitlines() if l.startswith("FULLQ ")][-1][6:])
    got = [_full_q_of_fixture(1, 91), _full_q_of_fixture(3, 92)]
    assert got == want
