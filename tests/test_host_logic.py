"""CPU: the product's host-side pure functions (graal_amd/sampler.py) against the oracle's literal restatement of
cuda_lib_gl.py (oracle/oracle.py), plus the multi-GPU glue.  No engine, no GPU."""
import numpy as np
import pytest

from graal_amd import dist as gdist
from graal_amd import sampler as S
from graal_amd import synth
from oracle import oracle as O
from tests import util


def problem(n_sub=3, seed=5, n_bins=90, nnz=1200):
    par = synth.make_param_simu(fact=200.0, v_inter=0.02)
    P = synth.make_problem(n_bins=n_bins, nnz=nnz, n_sub=n_sub, seed=seed, contig_weights=(5, 4, 3), mean_len_bp=2000.0,
                           accu=9 if n_sub > 1 else 1, param=par)
    return synth.with_dense(P)


def test_as_coo_upper_accepts_dense_sparse_and_triples():
    import scipy.sparse as sp
    P = problem(n_sub=1, nnz=400)
    want = (P["coo_row"], P["coo_col"], P["coo_val"])
    dense = P["hic_matrix"]
    sym = sp.csr_matrix(dense)                      # csr + csr.T as simulation_loader.py:81-82 builds it
    upper = sp.coo_matrix((P["coo_val"], (P["coo_row"], P["coo_col"])), shape=dense.shape)
    for src in (dense, sym, upper, want, (P["coo_col"], P["coo_row"], P["coo_val"])[::1] and want):
        r, c, v = S.as_coo_upper(src)
        assert np.array_equal(r, want[0]) and np.array_equal(c, want[1]) and np.array_equal(v, want[2])
    # a diagonal entry and an explicit zero are dropped like cuda_lib_gl.py:157-160 does
    r, c, v = S.as_coo_upper((np.array([2, 1, 0]), np.array([2, 3, 5]), np.array([7, 0, 4])))
    assert list(r) == [0] and list(c) == [5] and list(v) == [4]


@pytest.mark.parametrize("n_sub,seed", [(1, 3), (3, 4)])
def test_neighbour_distributions_match_dense_reference_logic(n_sub, seed):
    P = problem(n_sub=n_sub, seed=seed, n_bins=60, nnz=500)
    ora = O.OracleSampler(P, np.random.RandomState(0))
    xk, pk = S.neighbour_distributions(P["bin_coo_row"], P["bin_coo_col"], P["bin_coo_val"], P["n_frags"])
    for i in range(P["n_frags"]):
        assert np.array_equal(xk[i], ora.distri_frags[i]["xk"]), i
        assert np.array_equal(pk[i], ora.distri_frags[i]["pk"]), i
        assert pk[i].dtype == np.float32


def test_neighbour_distributions_short_and_empty_rows():
    # bin 0 has 2 contacts, bin 5 none: rows are padded with zero columns, largest index first
    n = 14
    row = np.array([0, 0, 1]); col = np.array([3, 7, 2]); val = np.array([4.0, 4.0, 1.0], np.float32)
    dense = synth.dense_from_coo(row, col, val, n)
    xk, pk = S.neighbour_distributions(row, col, val, n)
    for i in (0, 5, 7):
        order = np.argsort(dense[i], kind="stable")[::-1][:10]
        assert np.array_equal(xk[i], order)
    assert pk[5] == pytest.approx(np.full(10, 0.1))
    assert pk[0][:2] == pytest.approx([0.5, 0.5]) and np.all(pk[0][2:] == 0)


def test_select_move_matches_literal_reference_logic():
    rng = np.random.RandomState(3)
    P = problem(n_sub=1, n_bins=30, nnz=100)
    ora = O.OracleSampler(P, None)
    for trial in range(300):
        k = int(rng.randint(1, 6))
        score = -1000.0 + rng.standard_normal(13 * k) * rng.choice([0.1, 5.0, 50.0])
        if trial % 7 == 0:
            score[:] = score[0]                      # all equal -> nothing is > 0 after the shift -> argmax
        seed = int(rng.randint(1 << 30))
        if trial % 11 == 3:
            score[rng.randint(len(score))] += 500.0  # one candidate far ahead: everything else is cut off -> argmax
        if trial % 13 == 5:
            score[:] = np.round(score)               # ties
        seed = int(rng.randint(1 << 30))
        mine = np.random.RandomState(seed)
        got = S.select_move(score, 13, mine)
        # literal copy of cuda_lib_gl.py:1898-1947 as restated in the oracle
        ora.rng = np.random.RandomState(seed)
        ora.score = np.copy(score)
        want = _oracle_select(ora)
        assert got[0] == want[0] and got[1] == want[1]
        # ... and the generator is left in the same state (the next draw of the run depends on it)
        assert mine.random_sample() == ora.rng.random_sample()


def test_neighbour_draw_is_numpys_legacy_choice_without_replacement():
    """legacy_choice_without_replacement == RandomState.choice(a, size, replace=False, p=p): values, order and stream."""
    rng = np.random.RandomState(11)
    for trial in range(2000):
        n = int(rng.randint(2, 11))
        p = rng.random_sample(n) ** rng.choice([1.0, 3.0, 8.0])
        if trial % 3 == 0:
            p[rng.random_sample(n) < 0.4] = 0.0              # padded rows: zero entries
        if not p.any():
            p[rng.randint(n)] = 1.0
        p = (p / p.sum()).astype(np.float32) if trial % 2 else p / p.sum()   # pk rows are float32
        a = rng.permutation(200)[:n].astype(np.int64)
        size = int(min(rng.randint(1, 11), np.count_nonzero(p)))
        seed = int(rng.randint(1 << 30))
        r1, r2 = np.random.RandomState(seed), np.random.RandomState(seed)
        want = r1.choice(a, size, p=p, replace=False)
        got = S.legacy_choice_without_replacement(r2, a, size, p)
        assert np.array_equal(got, want) and got.dtype == want.dtype, trial
        assert r1.random_sample() == r2.random_sample(), trial
    # numpy's own errors stay numpy's
    with pytest.raises(ValueError):
        S.legacy_choice_without_replacement(np.random.RandomState(0), np.arange(4), 2, np.array([0.5, 0.2, 0.2, 0.2]))
    with pytest.raises(ValueError):
        S.legacy_choice_without_replacement(np.random.RandomState(0), np.arange(4), 3, np.array([0.5, 0.5, 0.0, 0.0]))


def _oracle_select(ora):
    n_tmp = ora.n_tmp_struct
    scores_2_remove = []
    scores_2_remove.extend(range(n_tmp, len(ora.score), n_tmp))
    scores_2_remove.extend(range(n_tmp + 1, len(ora.score), n_tmp))
    id_max = ora.score.argmax()
    or_score = np.copy(ora.score)
    filtered_score = ora.score - ora.score.min()
    filtered_score[scores_2_remove] = 0
    max_score = filtered_score.max()
    filtered_score = filtered_score - (max_score - 30)
    filtered_score[filtered_score < 0] = 0
    ok = np.ix_(filtered_score > 0)
    sub = filtered_score[ok]
    sub = sub / sub.sum()
    sub[sub > 0] = np.power(sub[sub > 0], 1.0)
    sub = sub / sub.sum()
    if len(ok[0]) in (0, 1):
        s = id_max
    else:
        s = ora.rng.choice(ok[0], 1, p=sub)[0]
    return int(s), float(or_score[s])


@pytest.mark.parametrize("n_sub", [1, 3])
def test_dist_inter_genome_matches_reference_loop(n_sub):
    P = problem(n_sub=n_sub, n_bins=50, nnz=300)
    ora = O.OracleSampler(P, np.random.RandomState(1))
    n = P["n_frags"]
    rng = np.random.RandomState(9)
    orientable = (np.asarray(P["np_sub_frags_id"])[:, 3] > 1).astype(np.int32)
    assert np.array_equal(orientable, ora.np_init_orientable)
    states = [O.copy_state(ora.gpu_vect_frags)]
    for _ in range(12):
        s = util.random_layout(rng, n, p_circ=0.2)
        states.append(s)
    # also a lightly perturbed version of the initial genome (most neighbours still correct)
    s = O.copy_state(ora.gpu_vect_frags)
    out, _ = util.oracle_candidate(s, 7, 20, 6, int(s["id_c"].max()))
    states.append(out)
    for s in states:
        want = ora.dist_inter_genome(s)
        got = S.dist_inter_genome(s["prev"], s["next"], s["ori"], s["id_d"], ora.np_init_prev, ora.np_init_next,
                                  ora.np_init_ori, orientable, np.ones(n, bool), ora.n_frags_4_dist)
        assert got == want


def test_shard_ranges_partition_the_contact_list():
    for nnz in (0, 1, 7, 1000, 20_000_003):
        for world in (1, 2, 3, 8):
            edges = [gdist.shard_range(nnz, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == nnz
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1


def test_shard_take_deals_blocks_round_robin():
    for nnz in (0, 1, 4095, 4096, 4097, 100_003):
        for world in (1, 2, 3, 8):
            parts = [np.arange(nnz)[gdist.shard_take(nnz, r, world)] for r in range(world)]
            allidx = np.concatenate(parts) if parts else np.zeros(0, int)
            assert len(allidx) == nnz and len(np.unique(allidx)) == nnz          # a partition
            for p in parts:
                assert np.all(np.diff(p) > 0)                                    # each shard keeps the list's order
            sizes = [len(p) for p in parts]
            assert max(sizes) - min(sizes) <= 4096
            if world > 1 and nnz > 8 * 4096:
                assert parts[1][0] == 4096 and parts[0][4096] == world * 4096    # blocks of 4096, dealt round robin


def test_blacklist_fill_equals_the_dense_fill():
    """blacklist_fill on COO lists == the reference's dense overwrite (cuda_lib_gl.py:161-172)."""
    from graal_amd.sampler import as_coo_upper, blacklist_fill
    rng = np.random.RandomState(5)
    n_bins, S = 12, 30
    sub_ids = np.zeros((n_bins, 4), dtype=np.int32)
    k = 0
    for b in range(n_bins):
        ns = 3 if b < 9 else 1
        sub_ids[b, :ns] = np.arange(k, k + ns); sub_ids[b, 3] = ns; k += ns
    assert k == S
    dense = np.triu(rng.poisson(0.8, size=(S, S)).astype(np.float32), 1); dense = dense + dense.T
    bdense = np.triu(rng.poisson(2.0, size=(n_bins, n_bins)).astype(np.float32), 1); bdense = bdense + bdense.T
    black, v = [2, 10], np.float32(0.037)
    want, bwant = dense.copy(), bdense.copy()
    for b in black:
        bwant[b, :] = 0; bwant[:, b] = 0
        for i in range(sub_ids[b, 3]):
            want[sub_ids[b, i], :] = v; want[:, sub_ids[b, i]] = v
    (r, c, val), (br, bc, bv) = blacklist_fill(as_coo_upper(dense), as_coo_upper(bdense), sub_ids, black, v, S)
    got = np.zeros((S, S), dtype=np.float32); got[r, c] = val; got[c, r] = val
    np.fill_diagonal(want, 0)
    assert np.array_equal(got, want)
    assert np.all(r < c) and np.all(np.diff(r.astype(np.int64) * S + c) > 0)      # upper, sorted, no duplicates
    bgot = np.zeros((n_bins, n_bins), dtype=np.float32); bgot[br, bc] = bv; bgot[bc, br] = bv
    assert np.array_equal(bgot, bwant)


# ---------------------------------------------------------------------------------------------------------------------------
# The same host logic behind the C ABI (graal_amd/csrc/host_step.h: graal_step): numpy's MT19937 state advanced in place,
# numpy's pairwise float64 sum, the two legacy `choice` algorithms -- values AND generator state against the Python path above
# (which tests above hold to numpy itself and to the oracle's literal restatement).  No device call is made.
def _mt_addr(rs):
    return int(rs._bit_generator.ctypes.state_address)


def test_mt19937_state_layout_is_what_the_c_side_assumes():
    import ctypes
    rs = np.random.RandomState(123)
    rs.random_sample(7)
    st = rs.get_state(legacy=False)["state"]
    raw = (ctypes.c_uint32 * 625).from_address(_mt_addr(rs))
    assert np.array_equal(np.frombuffer(raw, dtype=np.uint32, count=624), st["key"])
    assert int(np.frombuffer(raw, dtype=np.int32, count=625)[624]) == int(st["pos"])


def test_c_np_sum_is_numpys_sum():
    import ctypes
    from graal_amd import lib
    L = lib.load()
    rng = np.random.RandomState(9)
    for n in list(range(1, 40)) + [63, 64, 65, 127, 128, 129, 130, 200, 257, 1000]:
        for scale in (1.0, 1e-8, 1e12):
            a = np.ascontiguousarray(rng.standard_normal(n) * scale * rng.choice([1.0, 1e-6], size=n))
            got = L.graal_host_np_sum(a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), n)
            assert got == a.sum(), (n, scale)


def test_c_select_move_equals_the_python_path_values_and_generator_state():
    import ctypes
    from graal_amd import lib
    L = lib.load()
    rng = np.random.RandomState(31)
    n_sampled = 0
    for trial in range(600):
        k = int(rng.randint(1, 11))
        score = -1000.0 + rng.standard_normal(13 * k) * rng.choice([0.1, 5.0, 50.0])
        if trial % 7 == 0:
            score[:] = score[0]
        if trial % 11 == 3:
            score[rng.randint(len(score))] += 500.0
        if trial % 13 == 5:
            score[:] = np.round(score)
        if trial % 17 == 9:
            score[rng.randint(len(score))] = np.nan
        seed = int(rng.randint(1 << 30))
        a, b = np.random.RandomState(seed), np.random.RandomState(seed)
        a.random_sample(int(rng.randint(0, 700)) or 1); b.set_state(a.get_state())    # anywhere in the 624-word block
        want = S.select_move(score, 13, a)
        sc = np.ascontiguousarray(score)
        got = L.graal_host_select_move(ctypes.c_void_p(_mt_addr(b)), sc.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), len(sc), 13)
        assert got == want[0], (trial, got, want)
        sa, sb = a.get_state(legacy=False)["state"], b.get_state(legacy=False)["state"]
        assert np.array_equal(sa["key"], sb["key"]) and sa["pos"] == sb["pos"], trial
        n_sampled += 1
    assert n_sampled == 600


def test_c_neighbour_proposal_equals_the_python_path_values_and_generator_state():
    import ctypes
    from graal_amd import lib
    L = lib.load()
    P = synth.add_repeats(problem(n_sub=1, seed=8, n_bins=70, nnz=900), (5, 18), 2)
    n, nb = int(P["n_new_frags"]), int(P["n_frags"])
    xk, pk = S.neighbour_distributions(P["bin_coo_row"], P["bin_coo_col"], P["bin_coo_val"], nb)
    # a handle without a device: graal_create fails (no GPU here / or succeeds on the GPU box) but hands the context out either way
    h = ctypes.c_void_p()
    L.graal_create(0, ctypes.byref(h))
    assert h
    disp = np.ascontiguousarray(np.asarray(P["frag_dispatcher"]).reshape(-1, 2), dtype=np.int32)
    coll = np.ascontiguousarray(P["collector_id_repeats"], dtype=np.int32)
    id_d = np.ascontiguousarray(P["S_o_A_frags"]["id_d"], dtype=np.int32)
    dup = np.zeros(nb, np.uint8); dup[list(P["id_frag_duplicated"])] = 1
    black = np.zeros(n, np.uint8); black[[3, 40]] = 1
    i32, u8 = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_uint8)
    xk_c, pk_c = np.ascontiguousarray(xk, np.int32), np.ascontiguousarray(pk, np.float32)
    assert L.graal_upload_proposal_tables(h, xk_c.ctypes.data_as(i32), pk_c.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), nb, xk_c.shape[1],
                                          id_d.ctypes.data_as(i32), n, disp.ctypes.data_as(i32), coll.ctypes.data_as(i32), len(coll),
                                          dup.ctypes.data_as(u8), black.ctypes.data_as(u8)) == 0

    class Fake(object):      # the Python path's return_neighbours, without an engine
        return_neighbours = S.sampler.return_neighbours
    f = Fake()
    f.id_d = id_d; f.n_neighbors = 10; f.distri_frags = {"xk": xk, "pk": pk}; f._row_cache, f._copies_cache = {}, {}
    f._dup_set = set(int(x) for x in P["id_frag_duplicated"]); f.frag_dispatcher = disp; f.collector_id_repeats = coll
    f._black_set = {3, 40}
    rng = np.random.RandomState(2)
    out = np.zeros(128, np.int32)
    for trial in range(400):
        fA, delta = int(rng.randint(n)), int(rng.randint(1, 13))
        seed = int(rng.randint(1 << 30))
        a, b = np.random.RandomState(seed), np.random.RandomState(seed)
        a.random_sample(int(rng.randint(1, 700))); b.set_state(a.get_state())
        f.rng = a
        want = f.return_neighbours(fA, delta)
        k = L.graal_host_neighbours(h, ctypes.c_void_p(_mt_addr(b)), fA, delta, out.ctypes.data_as(i32), 128)
        assert k == len(want) and list(out[:k]) == [int(x) for x in want], (trial, fA, delta)
        sa, sb = a.get_state(legacy=False)["state"], b.get_state(legacy=False)["state"]
        assert np.array_equal(sa["key"], sb["key"]) and sa["pos"] == sb["pos"], trial
    L.graal_destroy(h)


def test_packed_class_keys_are_equivalent_to_same_inputs():
    """The table kernel sorts the 13 candidates of a piece pair into classes of equal contact-model inputs by comparing 128-bit keys
    (frag_ops.h: inputs_key); the definition of "equal inputs" is same_inputs.  On two million random quadruples of transforms drawn
    from small ranges (so that equal inputs do occur), with and without the trans-branch RF-count indexing: key equality == same_inputs."""
    from tests import util
    hc = util.hostcheck()
    for quirk in (0, 1):
        assert hc.hc_inputs_key_disagreements(1_000_000, 7 + quirk, quirk) == 0
