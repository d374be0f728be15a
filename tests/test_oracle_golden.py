"""Pin the oracle (oracle/graal_oracle.c) to the reference's known answers (SURVEY.md Appendix E,
produced by the reference's own kernels) and to the reference's implied invariants
(cuda_lib_gl.py:1530-1537, 2196-2220).  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "appendix_e.json")))


def state_from_contigs(contigs):
    """contigs: {label: {"bins": [...], "len_bp": [...]}} -> fragment SoA (pyramid_sparse.py:1316-1327)."""
    n = sum(len(c["bins"]) for c in contigs.values())
    s = O.new_state(n)
    for label, c in contigs.items():
        start = 0
        L = sum(c["len_bp"])
        for k, (b, l) in enumerate(zip(c["bins"], c["len_bp"])):
            s["pos"][b] = k
            s["id_c"][b] = int(label)
            s["start_bp"][b] = start
            s["len_bp"][b] = l
            s["prev"][b] = c["bins"][k - 1] if k > 0 else -1
            s["next"][b] = c["bins"][k + 1] if k + 1 < len(c["bins"]) else -1
            s["l_cont"][b] = len(c["bins"])
            s["l_cont_bp"][b] = L
            start += l
    s["id"][:] = np.arange(n)
    s["id_d"][:] = np.arange(n)
    return s


def check_contigs(state, expected):
    n = len(state["pos"])
    seen = 0
    for label, c in expected.items():
        members = np.nonzero(state["id_c"] == int(label))[0]
        order = members[np.argsort(state["pos"][members])]
        got = [[int(b), int(state["ori"][b]), int(state["start_bp"][b])] for b in order]
        assert got == c["frags"], (label, got, c["frags"])
        assert list(state["pos"][order]) == list(range(len(order)))
        if c["n"] is not None:
            assert np.all(state["l_cont"][order] == c["n"])
            assert np.all(state["l_cont_bp"][order] == c["L"])
        # links follow the order (linear contigs)
        for k, b in enumerate(order):
            assert state["prev"][b] == (order[k - 1] if k > 0 else -1)
            assert state["next"][b] == (order[k + 1] if k + 1 < len(order) else -1)
        seen += len(order)
    assert seen == n
    assert np.all(state["id"] == np.arange(n))


def run_op(name):
    g = GOLD["G"]
    cur = state_from_contigs(g["contigs"])
    n = len(cur["pos"])
    fA, fB, max_id = g["fA"], g["fB"], g["max_id"]
    D = O.DenseOracle
    pop, out = O.new_state(n), O.new_state(n)
    ids = np.zeros(n, np.int32)
    if name in ("op0", "op2", "op5", "op6", "pop_in_4"):
        D.pop_out(pop, cur, ids, fA, max_id)
        m2 = ids.max()
        if name == "op0":
            D.copy(out, pop)
        elif name == "op2":
            D.pop_in(1, out, pop, fA, fB, m2, 1)
        elif name == "op5":
            D.pop_in(2, out, pop, fA, fB, m2, -1)
        elif name == "op6":
            D.pop_in(3, out, pop, fA, fB, m2, 1)
        else:
            D.pop_in(4, out, pop, fA, fB, m2, 1)
    elif name.startswith("split"):
        D.split(out, cur, ids, fA, 0 if name.endswith("up0") else 1, max_id)
    else:
        upA, upB = {"op9": (0, 0), "op10": (0, 1), "op11": (1, 0), "op12": (1, 1)}[name]
        t1, t2 = O.new_state(n), O.new_state(n)
        D.split(t1, cur, ids, fA, upA, max_id)
        m1 = ids.max()
        D.split(t2, t1, ids, fB, upB, m1)
        m2 = ids.max()
        assert D.paste(out, t2, fA, fB, m2) == 0
    return out


@pytest.mark.parametrize("name", sorted(GOLD["ops"].keys()))
def test_mutation_known_answers(name):
    check_contigs(run_op(name), GOLD["ops"][name]["contigs"])


def test_stale_paste_slot():
    g = GOLD["G"]
    cur = state_from_contigs(g["contigs"])
    n = len(cur["pos"])
    out = O.new_state(n)
    for k in O.FIELDS:
        out[k][:] = -7
    n_stale = O.DenseOracle.paste(out, cur, 1, 2, g["max_id"])
    written = [b for b in range(n) if out["pos"][b] != -7]
    assert written == GOLD["stale_paste"]["bins_written"]
    assert n_stale == n - len(written)


def toy_likelihood_problem():
    t = GOLD["likelihood_toy"]
    nb, ns = t["n_bins"], t["n_sub"]
    contigs = {}
    b = 0
    for ci, k in enumerate(t["bins_per_contig"]):
        contigs[str(ci + 1)] = {"bins": list(range(b, b + k)), "len_bp": [t["len_bp"]] * k}
        b += k
    state = state_from_contigs(contigs)
    S = nb * ns
    obs = np.zeros((S, S), np.float32)
    lcg = t["obs_lcg"]
    s = lcg["seed"]
    for i in range(S):
        for j in range(i + 1, S):
            s = (s * lcg["a"] + lcg["c"]) % (1 << 32)
            m = lcg["mod_same_block_of_9"] if i // 9 == j // 9 else lcg["mod_other"]
            obs[i, j] = obs[j, i] = (s >> 24) % m
    sub_id = np.array([[3 * i, 3 * i + 1, 3 * i + 2, 3] for i in range(nb)], np.int32)
    sub_len = np.full((nb, 3), t["sub_len_kb"], np.float32)
    sub_accu = np.full((nb, 3), t["accu"], np.int32)
    disp = np.array([[i, i + 1] for i in range(nb)], np.int32)
    coll = np.arange(nb, dtype=np.int32)
    p = t["param_simu"]
    c1 = np.float32(0.53 * 9.6 ** -1.5)
    param = np.array([p["kuhn"], p["lm"], c1, p["slope"], p["d"], p["d_max"], p["fact"], p["v_inter"]], np.float32)
    dev = O.DenseOracle(obs, sub_id, sub_len, sub_accu, disp, coll, nb, t["n_frags_per_bins"], param)
    return dev, state


def test_likelihood_known_answers():
    t = GOLD["likelihood_toy"]
    dev, state = toy_likelihood_problem()
    per_pix = np.zeros(dev.n_pix)
    full = dev.evaluate(state, per_pix)
    assert full == pytest.approx(t["full"], rel=t["rel_tol"])
    assert per_pix.sum() == pytest.approx(full, rel=1e-12)
    n = len(state["pos"])
    pop = O.new_state(n)
    ids = np.zeros(n, np.int32)
    dev.pop_out(pop, state, ids, 1, 2)
    after = dev.evaluate(pop)
    assert after == pytest.approx(t["full_after_eject_1"], rel=t["rel_tol"])
    delta = dev.sub_compute(pop, [0, 1, 2], [], np.arange(6, dtype=np.int32), per_pix)
    assert delta == pytest.approx(t["delta_kernel"], rel=t["rel_tol"])
    # the reference's own implied invariant (cuda_lib_gl.py:2196-2220)
    assert delta == pytest.approx(after - full, abs=1e-9)


def test_scalar_model_functions():
    p = np.array([1.0, 9.6, np.float32(0.53 * 9.6 ** -1.5), -1.5, 3.0, 500.0, 50.0, 0.05], np.float32)
    assert O.rippe(0.0, p) == pytest.approx(0.05)          # s == 0 -> clamp (kernels3.cu:125,128)
    assert O.rippe(600.0, p) == pytest.approx(0.05)        # s >= d_max -> v_inter
    s = 2.0
    want = 0.53 * 9.6 ** -1.5 * s ** -1.5 * np.exp(1.0 / ((s * 9.6) ** 2 + 3.0)) * 50.0
    assert O.rippe(s, p) == pytest.approx(want, rel=1e-5)
    assert O.lik(0.0, 3.0) == 0.0                          # ex == 0 guard (kernels3.cu:197)
    assert O.lik(2.5, 0.0) == -2.5
    assert O.lik(2.0, 3.0) == pytest.approx(3 * np.log(2.0) - 2.0 - np.log(6.0), rel=1e-12)
    ob = 20.0
    want = ob * np.log(7.0) - 7.0 - (ob * np.log(ob) - ob + np.log(np.sqrt(ob * 2 * np.pi)))
    assert O.lik(7.0, ob) == pytest.approx(want, rel=1e-12)
    assert O.rippe_circ(3.0, 40.0, p) >= 0.05
