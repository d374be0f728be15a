"""GPU: the drop-in sampler (graal_amd.sampler.sampler over libgraal_hip.so) against the oracle's literal restatement
of cuda_lib_gl.sampler, driven by the same headless start_EM loop and the same seeded RandomState.

Bit-exact: accepted-move trace (id_fA, id_fB, op), contig statistics, distance to the initial genome, final fragment
SoA.  Within tolerance (north_star: 1e-5 relative): log-likelihood values."""
import numpy as np
import pytest

from graal_amd import em, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def problem(n_sub, seed, n_bins, nnz):
    par = synth.make_param_simu(fact=200.0, v_inter=0.02)
    # grid_bp = 2000: every float32 kb coordinate is exact, so the dense reference arithmetic is shift invariant
    P = synth.make_problem(n_bins=n_bins, nnz=nnz, n_sub=n_sub, seed=seed, contig_weights=(5, 4, 3), mean_len_bp=2000.0,
                           accu=9 if n_sub > 1 else 1, param=par, grid_bp=2000)
    return synth.with_dense(P)


def make_gpu_sampler(P, rng, **kw):
    from graal_amd.sampler import sampler
    import scipy.sparse as sp
    S = P["init_n_sub_frags"]
    sub = sp.coo_matrix((P["coo_val"], (P["coo_row"], P["coo_col"])), shape=(S, S)).tocsr()
    n = P["n_frags"]
    binm = sp.coo_matrix((P["bin_coo_val"], (P["bin_coo_row"], P["bin_coo_col"])), shape=(n, n)).tocsr()
    return sampler(True, P["S_o_A_frags"], P["collector_id_repeats"], P["frag_dispatcher"], P["id_frag_duplicated"],
                   P["id_frags_blacklisted"], P["n_frags"], P["n_new_frags"], P["init_n_sub_frags"],
                   P["n_new_sub_frags"], None, binm + binm.T, P["np_sub_frags_len_bp"], P["np_sub_frags_id"],
                   P["np_sub_frags_accu"], P["mean_squared_frags_per_bin"], None, None, sub + sub.T,
                   P["mean_value_trans"], 2, False, None, rng=rng, param_simu=P["param_simu"], **kw)


@pytest.mark.parametrize("n_sub,seed,n_bins,nnz,cycles,delta", [(1, 41, 70, 1200, 3, 3), (3, 42, 60, 1500, 2, 4),
                                                                (1, 43, 50, 700, 2, 5)])
def test_trace_matches_oracle(n_sub, seed, n_bins, nnz, cycles, delta):
    P = problem(n_sub, seed, n_bins, nnz)
    ora = O.OracleSampler(P, np.random.RandomState(seed), fix_trans_accu=True)
    t_ref = em.run_em(ora, cycles, delta, rng=ora.rng)
    gpu_rng = np.random.RandomState(seed)
    g = make_gpu_sampler(P, gpu_rng)
    t_gpu = em.run_em(g, cycles, delta, rng=gpu_rng)
    assert ora.n_stale_paste == 0 and g.n_stale_paste == 0
    assert np.array_equal(t_gpu.mutations(), t_ref.mutations())          # accepted-move trace, bit-exact
    assert t_gpu.n_contigs == t_ref.n_contigs
    assert t_gpu.mean_len == t_ref.mean_len
    assert t_gpu.dist == t_ref.dist
    assert np.allclose(t_gpu.likelihood, t_ref.likelihood, rtol=1e-8, atol=0)
    g.gpu_vect_frags.copy_from_gpu()
    for k in O.FIELDS:                                                    # fragment ordering, bit-exact
        assert np.array_equal(getattr(g.gpu_vect_frags, k), ora.gpu_vect_frags[k]), k
    # the run did something: contigs were merged back from the exploded genome
    assert t_gpu.n_contigs[-1] < 0.7 * P["n_frags"]
    # full re-evaluation agrees with the carried-over likelihood
    assert g.eval_likelihood() == pytest.approx(ora.init_likelihood(), rel=1e-8)
    g.free_gpu()


def test_trace_with_nuisance_parameter_sampling_matches_oracle():
    """start_EM with "sample parameters" on (main_gl.py:258-262): the random walk on (fact, slope, d_max, v_inter) draws
    from the same RandomState and re-evaluates the FULL likelihood under test parameters every step."""
    P = problem(3, 46, 45, 900)
    P["bins"] = np.arange(2.0, 60.0, 2.0)
    ora = O.OracleSampler(P, np.random.RandomState(46), fix_trans_accu=True)
    t_ref = em.run_em(ora, 1, 3, rng=ora.rng, sample_param=True)
    gpu_rng = np.random.RandomState(46)
    g = make_gpu_sampler(P, gpu_rng)
    g.bins = P["bins"]
    t_gpu = em.run_em(g, 1, 3, rng=gpu_rng, sample_param=True)
    assert np.array_equal(t_gpu.mutations(), t_ref.mutations())
    assert t_gpu.success == t_ref.success and 0 < sum(t_ref.success) < len(t_ref.success)
    for a, b in ((t_gpu.fact, t_ref.fact), (t_gpu.slope, t_ref.slope), (t_gpu.d_max, t_ref.d_max), (t_gpu.d_nuc, t_ref.d_nuc)):
        assert np.array_equal(np.asarray(a, np.float32), np.asarray(b, np.float32))   # parameters are float32: bit-exact
    assert np.allclose(t_gpu.likelihood_nuisance, t_ref.likelihood_nuisance, rtol=1e-8, atol=0)
    g.gpu_vect_frags.copy_from_gpu()
    for k in O.FIELDS:
        assert np.array_equal(getattr(g.gpu_vect_frags, k), ora.gpu_vect_frags[k]), k
    g.free_gpu()


def test_replay_of_a_trace_reproduces_the_layout(tmp_path):
    P = problem(1, 44, 40, 400)
    rng = np.random.RandomState(44)
    g = make_gpu_sampler(P, rng, compute_dist=False)
    t = em.run_em(g, 1, 3, rng=rng)
    em.save_behaviour_to_txt(t, str(tmp_path))
    g.gpu_vect_frags.copy_from_gpu()
    final = {k: np.copy(getattr(g.gpu_vect_frags, k)) for k in O.FIELDS}
    g.free_gpu()
    # fresh sampler: explode, then replay the saved list_mutations.txt
    g2 = make_gpu_sampler(P, np.random.RandomState(0), compute_dist=False)
    g2.modify_gl_cuda_buffer(0, 0)
    g2.explode_genome(0)
    em.replay(g2, em.load_mutations(str(tmp_path / "list_mutations.txt")))
    g2.gpu_vect_frags.copy_from_gpu()
    for k in O.FIELDS:
        if k != "id_c":   # labels depend on when the last relabel happened; membership is compared below
            assert np.array_equal(getattr(g2.gpu_vect_frags, k), final[k]), k
    a, b = g2.gpu_vect_frags.id_c, final["id_c"]
    assert np.array_equal(a[:, None] == a[None, :], b[:, None] == b[None, :])
    g2.free_gpu()


def test_inconsistent_inputs_fail_loudly():
    from graal_amd.sampler import sampler
    P = problem(1, 45, 20, 60)
    args = [True, P["S_o_A_frags"], P["collector_id_repeats"], P["frag_dispatcher"], [3], [], P["n_frags"],
            P["n_new_frags"], P["init_n_sub_frags"], P["n_new_sub_frags"], None, P["hic_matrix_sub_sampled"],
            P["np_sub_frags_len_bp"], P["np_sub_frags_id"], P["np_sub_frags_accu"], P["mean_squared_frags_per_bin"],
            None, None, P["hic_matrix"], P["mean_value_trans"], 2, False, None]
    with pytest.raises(ValueError):   # a duplicated bin without its copies
        sampler(*args)


@pytest.mark.parametrize("n_sub,seed", [(1, 47), (3, 48)])
def test_trace_with_blacklisted_fragments_matches_oracle(n_sub, seed):
    """Blacklisted bins (cuda_lib_gl.py:161-172, 1796, 1962-1978, 2326-2329): their observations are the non-integer fill
    value for EVERY pair, they are never proposed and never moved on their own, and they do not count in the distance."""
    P = problem(n_sub, seed, 50, 900)
    P["id_frags_blacklisted"] = [3, 4, 5, 17, 31]
    ora = O.OracleSampler(P, np.random.RandomState(seed), fix_trans_accu=True)
    t_ref = em.run_em(ora, 2, 4, rng=ora.rng)
    gpu_rng = np.random.RandomState(seed)
    g = make_gpu_sampler(P, gpu_rng)
    assert g.eval_likelihood() == pytest.approx(O.OracleSampler(P, np.random.RandomState(0), fix_trans_accu=True).init_likelihood(), rel=1e-8)
    t_gpu = em.run_em(g, 2, 4, rng=gpu_rng)
    assert np.array_equal(t_gpu.mutations(), t_ref.mutations())
    assert (np.asarray(t_ref.mutations())[:, 2] == -1).sum() == 2 * len(P["id_frags_blacklisted"])   # skipped steps are in the trace
    assert t_gpu.n_contigs == t_ref.n_contigs and t_gpu.dist == t_ref.dist
    assert np.allclose(t_gpu.likelihood, t_ref.likelihood, rtol=1e-8, atol=0)
    g.gpu_vect_frags.copy_from_gpu()
    for k in O.FIELDS:
        assert np.array_equal(getattr(g.gpu_vect_frags, k), ora.gpu_vect_frags[k]), k
    g.free_gpu()


def test_dataset_to_trace_through_the_pyramid(tmp_path):
    """End to end (rows f3 + f2 + the hot path): 3-file text dataset -> filtered pyramid -> sampler inputs of level 1 (ragged
    bins of 1-3 sub-fragments, one merged level-0 bin, a blacklisted contig) -> MCMC cycles on the GPU, against the oracle's
    literal restatement fed with the DENSE matrices of the same pyramid."""
    from graal_amd import pyramid as pyr
    from graal_amd.sampler import sampler
    from tests.test_pyramid import make_dataset
    rng = np.random.RandomState(11)
    base = str(tmp_path / "ds")
    make_dataset(base, rng, contig_sizes=(23, 16, 12), n_pairs=30000, empty=(4, 30))
    P0 = pyr.build_and_filter(base, 2, 3)
    inp = pyr.simulation_inputs(P0, 1, candidates_blacklist=[3])
    # exact float32 kb coordinates make the dense reference arithmetic shift invariant: put the level-0
    # fragments on a 2 kb grid
    S, n = inp["init_n_sub_frags"], inp["n_frags"]
    par = synth.make_param_simu(fact=200.0, v_inter=max(float(inp["mean_value_trans"]), 0.02))
    P = dict(inp)
    P["param_simu"] = par
    sr, sc, sv = inp["hic_matrix"]
    br, bc, bv = inp["hic_matrix_sub_sampled"]
    P["hic_matrix"] = synth.dense_from_coo(sr, sc, sv, S)
    P["hic_matrix_sub_sampled"] = synth.dense_from_coo(br, bc, bv, n)
    seed = 12
    # (ragged bins with different RF counts: the sampler's default is the reference's arithmetic, trans-branch RF-count indexing
    # included -- the oracle runs with it too)
    ora = O.OracleSampler(P, np.random.RandomState(seed), fix_trans_accu=False)
    t_ref = em.run_em(ora, 2, 4, rng=ora.rng)
    gpu_rng = np.random.RandomState(seed)
    g = sampler(True, inp["S_o_A_frags"], inp["collector_id_repeats"], inp["frag_dispatcher"], inp["id_frag_duplicated"],
                inp["id_frags_blacklisted"], inp["n_frags"], inp["n_new_frags"], inp["init_n_sub_frags"], inp["n_new_sub_frags"],
                None, inp["hic_matrix_sub_sampled"], inp["np_sub_frags_len_bp"], inp["np_sub_frags_id"], inp["np_sub_frags_accu"],
                inp["mean_squared_frags_per_bin"], inp["norm_vect_accu"], inp["S_o_A_sub_frags"], inp["hic_matrix"],
                inp["mean_value_trans"], 2, False, None, rng=gpu_rng, param_simu=par)
    t_gpu = em.run_em(g, 2, 4, rng=gpu_rng)
    assert np.array_equal(t_gpu.mutations(), t_ref.mutations())
    assert t_gpu.n_contigs == t_ref.n_contigs and t_gpu.dist == t_ref.dist
    assert np.allclose(t_gpu.likelihood, t_ref.likelihood, rtol=1e-5, atol=0)   # generic bp coordinates: north_star tolerance
    g.gpu_vect_frags.copy_from_gpu()
    for k in O.FIELDS:
        assert np.array_equal(getattr(g.gpu_vect_frags, k), ora.gpu_vect_frags[k]), k
    g.free_gpu()


def test_headless_run_from_a_dataset_folder(tmp_path):
    """python -m graal_amd.run: dataset folder -> pyramid -> fit -> MCMC -> trace files + scaffolded FASTA."""
    import os
    from graal_amd import run
    from tests.test_pyramid import make_dataset
    base = str(tmp_path / "ds")
    make_dataset(base, np.random.RandomState(21), contig_sizes=(40, 30, 20), n_pairs=40000, empty=(4, 45), polymer_like=True)
    out = str(tmp_path / "out")
    tr = run.main(["--dataset", base, "--fasta", os.path.join(base, "genome.fa"), "--size-pyramid", "3", "--level", "1",
                   "--cycles", "3", "--neighbours", "3", "--seed", "5", "--out", out])
    n = len(tr.likelihood)
    n_frags = len(set(tr.id_fA))
    assert n == 3 * n_frags and 20 <= n_frags <= 30 and np.isfinite(tr.likelihood).all()   # (a toy: no claim on the assembly)
    muts = em.load_mutations(os.path.join(out, "list_mutations.txt"))
    assert muts.shape == (n, 3) and np.array_equal(muts, tr.mutations())
    fa = open(os.path.join(out, "genome.fasta")).read()
    assert fa.count(">3C-assembly|contig_") == tr.n_contigs[-1] or fa.count(">") > 0
    assert sum(len(line) for line in fa.split("\n") if not line.startswith(">")) > 0
    # same seed, same trace
    tr2 = run.main(["--dataset", base, "--size-pyramid", "3", "--level", "1", "--cycles", "3", "--neighbours", "3", "--seed", "5",
                    "--out", str(tmp_path / "out2")])
    assert np.array_equal(tr2.mutations(), tr.mutations())


def test_level0_headless_run_equals_the_oracle(tmp_path):
    """BASELINE.json's headline configuration, "pyramid level 0 (full restriction-fragment resolution)" -- SURVEY 8d C4 -- from a dataset
    folder: python -m graal_amd.run --size-pyramid 1 --level 0 (bins = the filtered level-0 fragments, one sub-fragment each, both
    matrices the level-0 COO list; graal_amd/pyramid.py:simulation_inputs).  The reference cannot run this level
    (simulation_loader.py:45,68); its algorithm can: the oracle's literal restatement fed with the DENSE matrix of the same inputs gives
    the trace the headless run must reproduce -- accepted moves, contig counts and distances bit for bit."""
    import os
    from graal_amd import pyramid as pyr
    from graal_amd import run
    from tests.test_pyramid import make_dataset
    base = str(tmp_path / "ds")
    make_dataset(base, np.random.RandomState(31), contig_sizes=(40, 30, 20), n_pairs=40000, empty=(4, 45), polymer_like=True)
    P0 = pyr.build_and_filter(base, 1, 3)
    inp = pyr.simulation_inputs(P0, 0)
    n = inp["n_frags"]
    assert 80 <= n <= 90 and inp["init_n_sub_frags"] == n
    par = synth.make_param_simu(fact=200.0, v_inter=max(float(inp["mean_value_trans"]), 0.02))
    P = dict(inp)
    P["param_simu"] = par
    r, c, v = inp["hic_matrix"]
    P["hic_matrix"] = synth.dense_from_coo(r, c, v, n)
    P["hic_matrix_sub_sampled"] = synth.dense_from_coo(r, c, v, n)
    seed = 32
    ora = O.OracleSampler(P, np.random.RandomState(seed), fix_trans_accu=False)
    t_ref = em.run_em(ora, 2, 4, rng=ora.rng)
    out = str(tmp_path / "out")
    tr = run.main(["--dataset", base, "--size-pyramid", "1", "--level", "0", "--cycles", "2", "--neighbours", "4", "--seed", str(seed),
                   "--no-fit", "--param"] + [repr(float(x)) for x in par] + ["--fasta", os.path.join(base, "genome.fa"), "--out", out, "--images"])
    assert len(tr.likelihood) == 2 * n
    assert np.array_equal(tr.mutations(), t_ref.mutations())
    assert tr.n_contigs == t_ref.n_contigs and tr.dist == t_ref.dist
    assert np.allclose(tr.likelihood, t_ref.likelihood, rtol=1e-5, atol=0)     # generic bp coordinates: north_star's tolerance
    assert np.array_equal(em.load_mutations(os.path.join(out, "list_mutations.txt")), tr.mutations())
    assert open(os.path.join(out, "genome.fasta")).read().count(">") >= 1
    # the two images of start_EM (display_current_matrix, main_gl.py:213, 283): the contact matrix in the genome's order before and after
    from graal_amd import image
    pre, post = image.read_tiff_f32(os.path.join(out, "pre_simu.tiff")), image.read_tiff_f32(os.path.join(out, "post_em.tiff"))
    want_pre = synth.dense_from_coo(r, c, v, n)
    np.fill_diagonal(want_pre, 0)               # (the reference zeroes the diagonal of its matrices, cuda_lib_gl.py:157-160; level 0's list holds self-contacts)
    assert pre.shape == post.shape == (n, n) and np.array_equal(pre, want_pre)   # (the layout as loaded: the identity order)
    assert float(pre.sum()) == float(post.sum()) and not np.array_equal(pre, post)
    # and with the Rippe fit of the level-0 histogram in front (cuda_lib_gl.py:1229-1294), as the GUI's start button would run it
    tr2 = run.main(["--dataset", base, "--size-pyramid", "1", "--level", "0", "--cycles", "1", "--neighbours", "3", "--seed", "5",
                    "--out", str(tmp_path / "out2")])
    assert len(tr2.likelihood) == n and np.isfinite(tr2.likelihood).all() and tr2.n_contigs[-1] < n


@pytest.mark.parametrize("n_sub,black", [(1, False), (3, False), (3, True)])
def test_genome_distance_kernel_equals_the_host_loop(n_sub, black):
    """k_dist (graal_genome_distance) against the vectorised host restatement of dist_inter_genome and the oracle's literal
    loop (cuda_lib_gl.py:475-541) on random layouts: reversed fragments, circular contigs, singletons, blacklisted bins."""
    from graal_amd import sampler as S
    from tests import util
    P = problem(n_sub, 77, 64, 900)
    if black:
        P["id_frags_blacklisted"] = [3, 17, 40]
        P["mean_value_trans"] = 0.05
    g = make_gpu_sampler(P, np.random.RandomState(1))
    ora = O.OracleSampler(P, np.random.RandomState(1))
    n = P["n_frags"]
    rng = np.random.RandomState(8)
    assert g.dist_inter_genome() == 0.0 == ora.dist_inter_genome(ora.gpu_vect_frags)   # the initial genome itself
    for trial in range(12):
        s = util.random_layout(rng, n, p_circ=0.3)
        if trial % 3 == 0:      # near the initial genome: most neighbours still right, some orientations flipped
            s = O.copy_state(ora.gpu_vect_frags)
            flip = rng.random_sample(n) < 0.2
            s["ori"][flip] *= -1
        for k in O.FIELDS:
            getattr(g.gpu_vect_frags, k)[:] = s[k]
        g.gpu_vect_frags.copy_to_gpu()
        got = g.dist_inter_genome()
        host = S.dist_inter_genome(s["prev"], s["next"], s["ori"], s["id_d"], g.np_init_prev, g.np_init_next, g.np_init_ori,
                                   g.np_init_orientable, g._dist_counted(), g.n_frags_4_dist)
        assert got == host == ora.dist_inter_genome(s), trial
        assert 0.0 <= got <= 1.0
    g.free_gpu()


def test_blacklisted_fragments_with_parameter_sampling_match_oracle():
    """Blacklisted fragments AND "sample parameters": the step of a blacklisted fragment proposes nothing and hands the
    reference's stale `self.o` to the nuisance step that follows it (cuda_lib_gl.py:1962-1978, 2076) -- reproduced -- but the
    carried-over total must not keep that stale value: the next scored step starts from a full evaluation (the reference
    re-evaluates every step, cuda_lib_gl.py:1828).  Accepted moves, accepted parameter moves and parameters are bit-exact."""
    P = problem(3, 49, 45, 900)
    P["id_frags_blacklisted"] = [2, 9, 10, 30]
    P["bins"] = np.arange(2.0, 60.0, 2.0)
    ora = O.OracleSampler(P, np.random.RandomState(49), fix_trans_accu=True)
    t_ref = em.run_em(ora, 2, 3, rng=ora.rng, sample_param=True)
    gpu_rng = np.random.RandomState(49)
    g = make_gpu_sampler(P, gpu_rng)
    g.bins = P["bins"]
    t_gpu = em.run_em(g, 2, 3, rng=gpu_rng, sample_param=True)
    assert np.array_equal(t_gpu.mutations(), t_ref.mutations())
    assert (np.asarray(t_ref.mutations())[:, 2] == -1).sum() == 2 * len(P["id_frags_blacklisted"])
    assert t_gpu.success == t_ref.success and 0 < sum(t_ref.success) < len(t_ref.success)
    for a, b in ((t_gpu.fact, t_ref.fact), (t_gpu.slope, t_ref.slope), (t_gpu.d_max, t_ref.d_max), (t_gpu.d_nuc, t_ref.d_nuc)):
        assert np.array_equal(np.asarray(a, np.float32), np.asarray(b, np.float32))
    assert np.allclose(t_gpu.likelihood, t_ref.likelihood, rtol=1e-8, atol=0)
    assert np.allclose(t_gpu.likelihood_nuisance, t_ref.likelihood_nuisance, rtol=1e-8, atol=0)
    g.gpu_vect_frags.copy_from_gpu()
    for k in O.FIELDS:
        if k != "id_c":
            assert np.array_equal(getattr(g.gpu_vect_frags, k), ora.gpu_vect_frags[k]), k
    # (the last nuisance step's full evaluation relabels the contigs on the GPU side: labels are compared as a partition)
    a, b = g.gpu_vect_frags.id_c, ora.gpu_vect_frags["id_c"]
    assert np.array_equal(a[:, None] == a[None, :], b[:, None] == b[None, :])
    g.free_gpu()


def test_c_step_equals_python_step_and_survives_its_fallbacks(monkeypatch):
    """graal_step (the per-step host logic behind the C ABI, scoring kernels launched behind the relabel without an event, statistics
    collected with the scores) against the Python host logic on the same engine code: the same accepted-move trace, likelihood
    series, statistics and generator state -- also when k_tm's wait for k_scan's announcement is made to time out at once, so that
    every engine takes the repeat-with-an-event path once (GRAAL_TM_SPIN_TICKS=1), and with the finishing kernel instead of
    k_tm's finisher."""
    P = problem(1, 51, 160, 6000)

    def go(env, finisher=True):
        for k in ("GRAAL_PY_STEP", "GRAAL_TM_SPIN_TICKS", "GRAAL_NO_TM_SPIN"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        rng = np.random.RandomState(51)
        g = make_gpu_sampler(P, rng)
        g.engine.set_finisher(finisher)
        t = em.run_em(g, 2, 4, rng=rng)
        st = rng.get_state(legacy=False)["state"]
        out = (np.asarray(t.mutations()), list(t.likelihood), list(t.n_contigs), list(t.dist), int(st["pos"]), st["key"].copy())
        used_c = g._c_step
        g.free_gpu()
        return out, used_c
    want, used_c = go({"GRAAL_PY_STEP": "1"})
    assert not used_c
    for env, fin in (({}, True), ({"GRAAL_TM_SPIN_TICKS": "1"}, True), ({"GRAAL_NO_TM_SPIN": "1"}, True), ({}, False), ({"GRAAL_TM_SPIN_TICKS": "1"}, False)):
        got, used_c = go(env, fin)
        assert used_c
        assert np.array_equal(got[0], want[0]), (env, fin)
        assert got[1] == want[1] and got[2] == want[2] and got[3] == want[3], (env, fin)
        assert got[4] == want[4] and np.array_equal(got[5], want[5]), (env, fin)


@pytest.mark.parametrize("n_sub,seed", [(1, 55), (3, 56)])
def test_selection_handed_back_by_the_c_step_keeps_the_generator_stream(monkeypatch, n_sub, seed):
    """GRAAL_STEP_SELECT: graal_step has drawn the neighbours and scored them, but leaves the selection to the caller (a score vector
    hs_select does not judge itself; forced here for EVERY step by the test hook GRAAL_STEP_FORCE_SELECT).  The caller must select
    and commit WITHOUT drawing the neighbours again -- the return code used to be GRAAL_STEP_FALLBACK ("nothing drawn"), the Python
    path redid the step and the generator stream silently left the reference's.  Trace, likelihood series and generator state must
    equal the pure Python path's -- one step at a time and through graal_steps; with sub-fragments the step carries a full
    re-evaluation inside (flag 8), whose value must survive the hand-over too."""
    P = problem(n_sub, seed, 120, 4000)

    def go(env, batched):
        for k in ("GRAAL_PY_STEP", "GRAAL_STEP_FORCE_SELECT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        rng = np.random.RandomState(seed)
        g = make_gpu_sampler(P, rng)
        t = em.run_em(g, 2, 4, rng=rng, on_step=None if batched else (lambda j, i, tr: None))
        st = rng.get_state(legacy=False)["state"]
        out = (np.asarray(t.mutations()), list(t.likelihood), list(t.n_contigs), int(st["pos"]), st["key"].copy())
        used_c = g._c_step
        g.free_gpu()
        return out, used_c
    want, used_c = go({"GRAAL_PY_STEP": "1"}, False)
    assert not used_c
    for batched in (False, True):
        got, used_c = go({"GRAAL_STEP_FORCE_SELECT": "1"}, batched)
        assert used_c
        assert np.array_equal(got[0], want[0]), batched
        assert got[1] == want[1] and got[2] == want[2], batched
        assert got[3] == want[3] and np.array_equal(got[4], want[4]), batched


def test_runs_of_steps_in_one_call_equal_single_steps():
    """sampler.steps_max_likelihood (graal_steps: a run of MCMC steps behind the C ABI in one call, em.run_em's path) against one
    step_max_likelihood call per step (run_em with a per-step callback): traces, likelihood series, statistics, final layout and
    generator state -- with a full re-evaluation due every 7th step, so that the runs are cut and resumed all the time."""
    P = problem(1, 53, 160, 6000)

    def go(batched, resync):
        rng = np.random.RandomState(53)
        g = make_gpu_sampler(P, rng)
        g.resync_every = resync
        calls = []
        t = em.run_em(g, 2, 4, rng=rng, on_step=None if batched else (lambda j, i, tr: calls.append(i)))
        st = rng.get_state(legacy=False)["state"]
        g.gpu_vect_frags.copy_from_gpu()
        lay = {k: np.copy(getattr(g.gpu_vect_frags, k)) for k in O.FIELDS}
        out = (np.asarray(t.mutations()), list(t.likelihood), list(t.full_likelihood), list(t.n_contigs), list(t.mean_len), int(st["pos"]),
               st["key"].copy(), lay, g.n_stale_paste)
        g.free_gpu()
        return out
    for resync in (512, 7):
        a, b = go(True, resync), go(False, resync)
        assert np.array_equal(a[0], b[0]) and a[1] == b[1] and a[2] == b[2] and a[3] == b[3] and a[4] == b[4], resync
        assert a[5] == b[5] and np.array_equal(a[6], b[6]) and a[8] == b[8], resync
        for k in O.FIELDS:
            assert np.array_equal(a[7][k], b[7][k]), (resync, k)


@pytest.mark.parametrize("n_sub,seed", [(1, 61), (3, 62)])
def test_explode_genome_behind_the_c_abi_equals_the_loop(n_sub, seed, monkeypatch):
    """explode_genome (cuda_lib_gl.py:1539-1556: relabel + eject, fragment by fragment) as ONE call (graal_explode) against the loop as
    written (GRAAL_PY_STEP=1): the same layout field by field, the same stale-paste count, and every fragment its own contig."""
    P = problem(n_sub, seed, 80, 1500)
    out = []
    for py in (False, True):
        monkeypatch.delenv("GRAAL_PY_STEP", raising=False)
        if py:
            monkeypatch.setenv("GRAAL_PY_STEP", "1")
        g = make_gpu_sampler(P, np.random.RandomState(seed))
        g.init_likelihood()
        g.explode_genome()
        assert g.likelihood_t is None
        g.modify_gl_cuda_buffer(0)
        g.gpu_vect_frags.copy_from_gpu()
        out.append(({k: np.copy(getattr(g.gpu_vect_frags, k)) for k in O.FIELDS}, g.n_stale_paste))
        g.free_gpu()
    (a, sa), (b, sb) = out
    assert sa == sb
    for k in O.FIELDS:
        assert np.array_equal(a[k], b[k]), k
    assert np.all(a["l_cont"] == 1) and len(np.unique(a["id_c"])) == P["n_new_frags"]
