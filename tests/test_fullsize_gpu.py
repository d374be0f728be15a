"""GPU, BASELINE.json's full size (C5: 50,000 fragments / 20,000,000 contacts, the bench.py workload): properties that do not
need the dense oracle (which cannot run at this size).

* the reference's implied invariant (cuda_lib_gl.py:2196-2220): candidate delta == full likelihood after the move - before;
* the step finished inside k_tm == the step finished by k_fin, bit for bit;
* determinism: the same seed gives the same accepted-move trace and the same final layout twice;
* structural invariants of the layout after 1,500 real MCMC steps (cuda_lib_gl.py:1530-1537);
* 1,500 incremental relabels (k_incr: counting, link-walked mates rows) == one full relabel (sort) of the same layout in a
  fresh engine: labels, statistics, candidate deltas and the full likelihood bit for bit;
* the full evaluation with the labels in LDS (k_full_nnz_l) == the one with 8-byte record gathers (k_full_nnz_u), bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c5():
    import bench
    from graal_amd import synth
    P = synth.make_problem(n_bins=50000, nnz=20_000_000, n_sub=1, seed=20141217)
    P["S_o_A_frags"] = bench.exploded_layout(P)
    return P


def run(P, seed, n_steps, finisher=True):
    import bench
    rng = np.random.RandomState(seed)
    smp = bench.build_sampler(P, rng, None, 0)
    smp.engine.set_finisher(finisher)
    smp.init_likelihood()
    order = np.arange(int(smp.n_new_frags), dtype=np.int32)
    rng.shuffle(order)
    trace = []
    for i in order[:n_steps]:
        o, n_contigs, _, _, _, op, fB, _, _ = smp.step_max_likelihood(int(i), 5)
        trace.append((int(i), int(fB), int(op), float(o), int(n_contigs)))
    return smp, trace


def test_c5_properties(c5):
    smp, trace = run(c5, 7, 1500)
    n = int(smp.n_new_frags)
    # ---- delta of the accepted move == full(after) - full(before), on the carried-over total
    carried = trace[-1][3]
    full = smp.eval_likelihood()
    assert carried == pytest.approx(full, rel=1e-9)      # 1,500 accumulated deltas vs one full evaluation (|logL| ~ 1e8)
    for fA in (11, 2222, 33333):
        max_id = smp.modify_gl_cuda_buffer(0)
        before = smp._full_likelihood()
        nb = smp.return_neighbours(fA, 5); nb.sort()
        d = smp._candidate_deltas(fA, nb, max_id)
        k, op = np.unravel_index(np.argmax(np.abs(d)), d.shape)
        smp.test_copy_struct(fA, nb[k], int(op), max_id)
        after = smp.eval_likelihood()
        assert d[k, op] == pytest.approx(after - before, rel=1e-6, abs=2e-6 * abs(before) * 1e-3)
    # ---- layout invariants
    smp.gpu_vect_frags.copy_from_gpu()
    g = smp.gpu_vect_frags
    heads = g.pos == 0
    assert g.l_cont[heads].sum() == n and (g.prev[heads & (g.circ == 0)] == -1).all()
    assert (g.start_bp[heads] == 0).all() and (g.activ == 1).all()
    for c in np.unique(g.id_c)[:200]:
        m = np.nonzero(g.id_c == c)[0]
        o = m[np.argsort(g.pos[m])]
        assert np.array_equal(g.pos[o], np.arange(len(o))) and (g.l_cont[o] == len(o)).all()
        assert np.array_equal(g.start_bp[o], np.cumsum(g.len_bp[o]) - g.len_bp[o]) and (g.l_cont_bp[o] == g.len_bp[o].sum()).all()
    # ---- the incrementally maintained labels / index / geometry / mates == a fresh engine's full relabel of this layout
    from graal_amd.lib import Engine
    state = {k: np.copy(getattr(g, k)) for k in ("pos", "id_c", "start_bp", "len_bp", "circ", "id", "prev", "next", "l_cont",
                                                 "l_cont_bp", "ori", "rep", "activ", "id_d")}
    e2 = Engine(0)
    e2.upload_subfrags(c5["np_sub_frags_id"], c5["np_sub_frags_len_bp"], c5["np_sub_frags_accu"], c5["init_n_sub_frags"],
                       c5["mean_squared_frags_per_bin"])
    e2.upload_contacts(c5["coo_row"], c5["coo_col"], c5["coo_val"])
    e2.set_params(c5["param_simu"])
    e2.upload_frags(state)
    e2.set_mode(ref_trans_accu=True, strict=True)        # (the sampler's arithmetic: bench.build_sampler's default)
    st2, max2 = e2.begin_step()
    st1, max1 = smp.engine.begin_step()
    assert max1 == max2 and list(st1[:7]) == list(st2[:7])
    got2 = e2.download_frags()
    for k in state:
        assert np.array_equal(got2[k], state[k]), k          # already ranked: the sort leaves the labels alone
    rng = np.random.RandomState(3)
    for fA in rng.randint(0, n, size=6):
        nb = smp.return_neighbours(int(fA), 5); nb.sort()
        assert np.array_equal(smp.engine.eval_candidates(int(fA), nb, max1), e2.eval_candidates(int(fA), nb, max2))
    assert np.array_equal(smp.engine.eval_full_q(), e2.eval_full_q())
    e2.close()
    # ---- determinism and finisher == k_fin at full size
    smp2, trace2 = run(c5, 7, 300, finisher=False)
    assert trace2 == trace[:300]
    smp.free_gpu(); smp2.free_gpu()


# ------------------------------------------------------------------------------------------------ late-stage regime
# C5 on its 7 ORIGINAL contigs (2.7-6.8k fragments each): what bench.py reports as `late_stage`.  Every step queues millions of
# contacts and thousands of mass work items; the dense oracle cannot run at this size, so the checks are the reference's implied
# invariant (delta == full after - full before, cuda_lib_gl.py:2196-2220), finisher setting irrelevant, and 2 ranks == 1 rank.
#
# The invariant is checked on a GRID variant of the map (every fragment a multiple of 1,000 bp: float32 kb coordinates are exact,
# so shifting a piece does not re-round anybody's coordinates).  On C5's own fragments (1 + Exp(660) bp, some of a few bp, at
# coordinates of several thousand kb where one float32 ulp is 0.5 bp) a FULL evaluation of the moved layout re-rounds the
# centres of every downstream fragment, and the expected values of neighbouring few-bp fragments -- ~1e6 contacts at the C5
# parameters -- jump by that rounding: full(after) - full(before) then differs from the geometric delta by ~1e-4 of logL
# (measured with the numpy re-score, which agrees with the engine's full evaluation to 1e-9: tools/diag_late.py).  The
# reference's own candidate scores carry that noise; GRAAL_MODE_STRICT reproduces it, the default path is free of it.
def _original_sampler(P, group=None, seed=11):
    import bench
    rng = np.random.RandomState(seed)
    smp = bench.build_sampler(P, rng, group, 0)
    smp.init_likelihood()
    return smp


def _late_proposals(smp, n_props, seed=12):
    rng = np.random.RandomState(seed)
    props = []
    for f in rng.randint(0, int(smp.n_new_frags), size=n_props):
        nb = smp.return_neighbours(int(f), 5)
        nb.sort()
        props.append((int(f), nb))
    return props


@pytest.fixture(scope="module")
def c5_original():
    from graal_amd import synth
    return synth.make_problem(n_bins=50000, nnz=20_000_000, n_sub=1, seed=20141217)


@pytest.fixture(scope="module")
def c5_grid():
    from graal_amd import synth
    return synth.make_problem(n_bins=50000, nnz=20_000_000, n_sub=1, seed=20141217, grid_bp=1000)


def test_c5_original_layout_delta_is_full_after_minus_before(c5_grid):
    smp = _original_sampler(c5_grid)
    st = smp.engine.layout_stats()
    assert int(st[0]) == 7 and int(st[4]) > 5000
    worst = 0.0
    for fA, nb in _late_proposals(smp, 6):
        max_id = smp.modify_gl_cuda_buffer(0)
        before = smp._full_likelihood()
        d = smp._candidate_deltas(fA, nb, max_id)
        c = smp.engine.last_counters()
        assert c[2] > 100000 and c[3] > 100            # the late-stage work: queued contacts and mass items for k_fin
        smp.engine.set_finisher(False)
        assert np.array_equal(d, smp._candidate_deltas(fA, nb, max_id))   # finisher setting: same sums, bit for bit
        smp.engine.set_finisher(True)
        # the move with the largest |delta| among the non-degenerate ones, and the best one
        for k, op in (np.unravel_index(np.argmax(np.abs(d)), d.shape), np.unravel_index(np.argmax(d), d.shape)):
            smp.test_copy_struct(fA, nb[k], int(op), max_id)
            after = smp.eval_likelihood()
            err = abs(d[k, op] - (after - before)) / abs(before)
            worst = max(worst, err)
            assert err < 1e-8, (fA, nb[k], op, d[k, op], after - before)   # (grid coordinates: exact geometry; Q30 rounding only)
            # undo is not defined for every op: continue from the moved layout
            max_id = smp.modify_gl_cuda_buffer(0)
            before = after
            d = smp._candidate_deltas(fA, nb, max_id)
    print("C5 original layout: worst |delta - (after - before)| / |logL| = %.3e" % worst)
    smp.free_gpu()


def test_c5_reference_arithmetic_at_full_size(c5_original, c5):
    """GRAAL_MODE_STRICT on C5's OWN fragments (generic bp lengths at coordinates of thousands of kb -- where the float32 noise of
    the reference's geometry is largest, see above).  At one sub-fragment per bin the reference's candidate delta is
    full(after) - full(before) pixel by pixel, so -- unlike the default mode, which needs the grid variant for this -- the
    windowed strict kernels (k_strict_cull + k_strict) must reproduce the difference of two FULL evaluations (independent
    kernels: k_full_nnz / k_full_mass) on the 7 original contigs; and the default mode's distance from the reference arithmetic
    is recorded for the late stage and for the headline state bench.py measures."""
    import json
    import os
    smp = _original_sampler(c5_original)
    smp.engine.set_mode(ref_trans_accu=True, strict=True)
    worst = worst_default = 0.0
    for fA, nb in _late_proposals(smp, 3):
        max_id = smp.modify_gl_cuda_buffer(0)
        before = smp._full_likelihood()
        d = smp._candidate_deltas(fA, nb, max_id)
        smp.engine.set_mode()
        d_def = smp._candidate_deltas(fA, nb, max_id)
        smp.engine.set_mode(ref_trans_accu=True, strict=True)
        worst_default = max(worst_default, float(np.nanmax(np.abs(d - d_def))) / abs(before))
        for k, op in (np.unravel_index(np.nanargmax(np.abs(d)), d.shape), np.unravel_index(np.nanargmax(d), d.shape)):
            smp.test_copy_struct(fA, nb[k], int(op), max_id)
            after = smp.eval_likelihood()
            err = abs(d[k, op] - (after - before)) / abs(before)
            worst = max(worst, err)
            assert err < 1e-9, (fA, nb[k], op, d[k, op], after - before)
            max_id = smp.modify_gl_cuda_buffer(0)
            before = after
            d = smp._candidate_deltas(fA, nb, max_id)
    smp.free_gpu()
    # the headline state (exploded + MCMC steps): the same proposals in both arithmetics
    smp, _ = run(c5, 7, 600)
    max_id = smp.modify_gl_cuda_buffer(0)
    logl = abs(smp._full_likelihood())
    rng = np.random.RandomState(5)
    head = 0.0
    for fA in rng.randint(0, int(smp.n_new_frags), size=40):
        nb = smp.return_neighbours(int(fA), 5); nb.sort()
        d_def = smp._candidate_deltas(int(fA), nb, max_id)
        smp.engine.set_mode(ref_trans_accu=True, strict=True)
        d_ref = smp._candidate_deltas(int(fA), nb, max_id)
        smp.engine.set_mode()
        head = max(head, float(np.nanmax(np.abs(d_ref - d_def))))
    smp.free_gpu()
    rec = {"late_stage_strict_delta_vs_full_difference_rel": worst, "late_stage_default_vs_reference_arithmetic_rel": worst_default,
           "headline_state_default_vs_reference_arithmetic_abs_logL_units": head, "headline_state_abs_logL": logl}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "c5_reference_arithmetic.json"), "w") as f:
        json.dump(rec, f)
    print("C5 reference arithmetic:", rec)


def _late_worker(rank, world, port, q):
    import os
    import torch.distributed as td
    from graal_amd import dist as gdist
    from graal_amd import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = synth.make_problem(n_bins=50000, nnz=20_000_000, n_sub=1, seed=20141217)
        smp = _original_sampler(P, gdist.Group(rank, world))
        max_id = smp.modify_gl_cuda_buffer(0)
        out = [smp._candidate_deltas(fA, nb, max_id) for fA, nb in _late_proposals(smp, 3)]
        full = smp._full_likelihood()
        q.put((rank, out, full, smp.exchange))
        smp.free_gpu()
    finally:
        td.destroy_process_group()


@pytest.mark.timeout(900)
def test_c5_original_layout_two_ranks_equal_one_rank(c5_original):
    import socket
    import torch.multiprocessing as mp
    smp = _original_sampler(c5_original)
    max_id = smp.modify_gl_cuda_buffer(0)
    want = [smp._candidate_deltas(fA, nb, max_id) for fA, nb in _late_proposals(smp, 3)]
    want_full = smp._full_likelihood()
    smp.free_gpu()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_late_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=800) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, out, full, exchange in res:
        assert exchange == "host"
        for a, b in zip(out, want):
            assert np.array_equal(a, b), rank          # sharded contacts + sharded mass units: the same int64 sums
        assert full == want_full


@pytest.mark.timeout(1200)
def test_full_evaluation_with_labels_in_lds_is_bit_identical():
    """k_full_nnz_l (lists of >= 2 M contacts with uniform RF counts: 16-bit labels in LDS, record gathers only for contacts
    inside one contig) against k_full_nnz_u (GRAAL_FULL_NO_LDS=1; the switch is read once per process, hence child processes):
    both int64 sums of graal_eval_full_q, on the exploded layout after 600 MCMC steps (nearly every contact joins two
    contigs) and on the 7 original contigs (nearly every contact inside one).  tools/full_l_check.py asserts the equality."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "full_l_check.py")], cwd=root, capture_output=True, text=True, timeout=1100)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    assert "bit-identical sums" in out.stdout


@pytest.mark.timeout(1500)
def test_tiled_windowed_mass_of_the_full_evaluation_is_bit_identical():
    """k_full_mass_t (long contigs: a wave's lanes are 64 fragments x, the fragments y behind them staged in LDS with their centres computed
    once) against k_full_mass (one lane per fragment pair, records and centres per pair): both int64 sums of graal_eval_full_q on C5's two
    layouts and on circular / reversed / 1-3 sub-fragment layouts in both RF-count indexings -- forced on and off in child processes
    (tools/full_mass_check.py asserts the equality).  evaluate_likelihood, kernels3.cu:2802-3222."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "full_mass_check.py")], cwd=root, capture_output=True, text=True, timeout=1400)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    assert "bit-identical sums" in out.stdout


# ------------------------------------------------------------------------------------------------ BASELINE config 4 shape, mid-run
# 40,000 bins x 1 sub-fragment, 8,000,000 contacts (tools/run_configs.py C4), on grid coordinates, after one full cycle from
# the exploded genome: ~3,000 contigs of a few to ~100 bins.  Here most steps leave k_tm's thresholds and are finished by
# k_fin after its verdict (tools/c4_trace.sh: 69 % of the steps) -- the path neither the exploded C5 state (finisher) nor its 7
# original contigs (k_fin launched unconditionally) take.  The dense oracle cannot run at this size: properties.
@pytest.mark.timeout(900)
def test_c4_shape_mid_run_properties():
    import bench
    from graal_amd import synth
    P = synth.make_problem(n_bins=40000, nnz=8_000_000, n_sub=1, seed=2014, contig_weights=(6.8, 6.2, 5.3, 4.6, 4.0, 3.4, 2.7),
                           grid_bp=1000)
    P["S_o_A_frags"] = bench.exploded_layout(P)
    rng = np.random.RandomState(41)
    smp = bench.build_sampler(P, rng, None, 0)
    smp.init_likelihood()
    n = int(smp.n_new_frags)
    order = np.arange(n, dtype=np.int32)
    rng.shuffle(order)
    for i in order:                                   # one cycle: 40,000 MCMC steps
        smp.step_max_likelihood(int(i), 5)
    st = smp.engine.layout_stats()
    assert 500 < int(st[0]) < 20000 and int(st[4]) > 16          # contigs have grown, the longest beyond the finisher's regime
    carried = smp.likelihood_t
    full = smp.eval_likelihood()
    assert carried == pytest.approx(full, rel=1e-8)              # 40,000 accumulated deltas vs one full evaluation
    needed_fin = 0
    for fA in rng.randint(0, n, size=40):
        fA = int(fA)
        max_id = smp.modify_gl_cuda_buffer(0)
        before = smp._full_likelihood()
        nb = smp.return_neighbours(fA, 5); nb.sort()
        d = smp._candidate_deltas(fA, nb, max_id)
        c = smp.engine.last_counters()
        needed_fin += (c[2] > 64) or (c[3] > 0)
        smp.engine.set_finisher(False)
        assert np.array_equal(d, smp._candidate_deltas(fA, nb, max_id))      # k_tm's finisher == k_fin, bit for bit
        smp.engine.set_finisher(True)
        k, op = np.unravel_index(np.argmax(np.abs(d)), d.shape)
        smp.test_copy_struct(fA, nb[k], int(op), max_id)
        after = smp.eval_likelihood()
        assert abs(d[k, op] - (after - before)) / abs(before) < 1e-8, (fA, nb[k], op, d[k, op], after - before)
    assert needed_fin >= 10                                       # the regime this test is about did occur
    smp.free_gpu()


# More contact-list ids than the scan's LDS bitmap has bits (393,216): the ids are folded onto 2^18 bits (launch_scan), a folded
# bit set by another id queues a contact for nothing and the consumers drop it.  450,000 sub-fragments, properties as above.
@pytest.mark.timeout(900)
def test_more_sub_fragments_than_bitmap_bits():
    import bench
    from graal_amd import synth
    P = synth.make_problem(n_bins=150000, nnz=3_000_000, n_sub=3, seed=77, accu=4, grid_bp=1000)
    assert int(P["init_n_sub_frags"]) > 393216
    P["S_o_A_frags"] = bench.exploded_layout(P)
    rng = np.random.RandomState(43)
    smp = bench.build_sampler(P, rng, None, 0, arithmetic="exact")
    smp.init_likelihood()
    n = int(smp.n_new_frags)
    for i in rng.randint(0, n, size=400):             # grow some contigs first
        smp.step_max_likelihood(int(i), 5)
    carried = smp.likelihood_t
    assert carried == pytest.approx(smp.eval_likelihood(), rel=1e-9)
    queued = 0
    for fA in rng.randint(0, n, size=12):
        fA = int(fA)
        max_id = smp.modify_gl_cuda_buffer(0)
        before = smp._full_likelihood()
        nb = smp.return_neighbours(fA, 5); nb.sort()
        d = smp._candidate_deltas(fA, nb, max_id)
        queued += int(smp.engine.last_counters()[2])
        smp.engine.set_finisher(False)
        assert np.array_equal(d, smp._candidate_deltas(fA, nb, max_id))
        smp.engine.set_finisher(True)
        k, op = np.unravel_index(np.argmax(np.abs(d)), d.shape)
        smp.test_copy_struct(fA, nb[k], int(op), max_id)
        after = smp.eval_likelihood()
        assert abs(d[k, op] - (after - before)) / abs(before) < 1e-8, (fA, nb[k], op, d[k, op], after - before)
    assert queued > 0
    smp.free_gpu()
