"""GPU: repeated fragments (allow_repeats) -- bins with several fragment copies, activity swaps (op 8) -- against the dense
oracle (kernels3.cu:2915-2930 copy loops, 3356-3380 repeat pixel ranges)."""
import numpy as np
import pytest

from graal_amd import synth
from oracle import oracle as O
from tests import util
from tests.test_engine_gpu import dense_for, relabel_ref

pytestmark = pytest.mark.gpu


def rep_problem(n_sub, seed, n_bins=40, nnz=900, dup=(7, 21), n_copies=2):
    par = synth.make_param_simu(fact=300.0, v_inter=0.03)
    P = synth.make_problem(n_bins=n_bins, nnz=nnz, n_sub=n_sub, seed=seed, contig_weights=(5, 3, 2), mean_len_bp=1500.0,
                           accu=1 if n_sub == 1 else 9, param=par, grid_bp=2000)
    return synth.add_repeats(synth.with_dense(P), dup, n_copies)


def split_observations(P):
    """What the sampler hands to the engine: contacts without the repeated bins + the repeated bins' observation rows."""
    dup = np.asarray(P["id_frag_duplicated"], dtype=np.int64)
    ids = np.asarray(P["np_sub_frags_id"]).reshape(-1, 4)
    S = int(P["init_n_sub_frags"])
    dup_sub = np.zeros(S, dtype=bool)
    for b in dup:
        dup_sub[ids[b, :ids[b, 3]]] = True
    r, c, v = P["coo_row"], P["coo_col"], P["coo_val"]
    keep = ~(dup_sub[r] | dup_sub[c])
    dense = np.array(P["hic_matrix"], dtype=np.float32)
    np.fill_diagonal(dense, 0)
    obs = np.zeros((len(dup), 3, S), dtype=np.float32)
    for i, b in enumerate(dup):
        for a in range(ids[b, 3]):
            obs[i, a] = dense[ids[b, a]]
    return (r[keep], c[keep], v[keep]), obs


def engine_with_repeats(P, state):
    from graal_amd.lib import Engine
    e = Engine(0)
    e.upload_subfrags(P["np_sub_frags_id"], P["np_sub_frags_len_bp"], P["np_sub_frags_accu"], P["init_n_sub_frags"],
                      P["mean_squared_frags_per_bin"])
    (r, c, v), obs = split_observations(P)
    e.upload_repeats(P["id_frag_duplicated"], P["frag_dispatcher"], P["collector_id_repeats"], obs)
    e.upload_contacts(r, c, v)
    e.set_params(P["param_simu"])
    e.upload_frags(state)
    return e


def random_state_with_repeats(P, rng, n_contigs, p_circ=0.0, p_inactive=0.3):
    """A random valid layout of ALL fragments (copies anywhere); some copies inactive (singleton contigs, as
    swap_activity_frag leaves them)."""
    n = int(P["n_new_frags"])
    S0 = P["S_o_A_frags"]
    s = util.random_layout(rng, n, n_contigs=n_contigs, p_circ=p_circ, len_grid=1)
    s["len_bp"][:] = S0["len_bp"]
    s["rep"][:] = S0["rep"]; s["id_d"][:] = S0["id_d"]; s["activ"][:] = 1
    for f in np.nonzero(S0["rep"] == 1)[0]:
        if s["l_cont"][f] == 1 and rng.random_sample() < p_inactive:
            s["activ"][f] = 0
    for lab in np.unique(s["id_c"]):
        m = np.nonzero(s["id_c"] == lab)[0]
        order = m[np.argsort(s["pos"][m])]
        s["start_bp"][order] = np.cumsum(s["len_bp"][order]) - s["len_bp"][order]
        s["l_cont_bp"][order] = s["len_bp"][order].sum()
    return s


@pytest.mark.parametrize("n_sub,seed", [(1, 71), (3, 72), (3, 73)])
def test_full_likelihood_with_repeats(n_sub, seed):
    P = rep_problem(n_sub, seed)
    dense = dense_for(P)
    rng = np.random.RandomState(seed)
    for trial in range(4):
        s = random_state_with_repeats(P, rng, n_contigs=int(rng.randint(8, 20)), p_circ=0.2 if trial else 0.0)
        relabel_ref(s)
        e = engine_with_repeats(P, s)
        e.relabel_contigs()
        want = dense.evaluate(s)
        got = e.eval_full()
        assert got == pytest.approx(want, rel=1e-6), (trial, got, want)
        e.close()
