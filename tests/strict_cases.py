"""Shared by tests/test_strict_windowed_gpu.py and its child process: a fixed list of (problem, layouts, proposals) cases for the
reference-arithmetic candidate kernels.  The parent runs them through k_tm / k_strict_cull + k_strict (the product path), the
child -- ``python -m tests.strict_cases out.npz`` with ``GRAAL_STRICT_DENSE=1`` in its environment -- through k_strict_dense, the
O(m^2) validation kernel that prices every pixel of contig(fA) u contig(fB) under every candidate (kernels3.cu:3259-3718 as written);
both store the candidates' int64 fixed-point sums, which must be EQUAL."""
import sys

import numpy as np

from graal_amd import synth
from graal_amd.lib import Q_SCALE, Engine


def _problem(n_bins, nnz, seed, n_sub, weights, mean_len_bp, accu, fact=300.0, v_inter=0.03, d_max=None):
    par = synth.make_param_simu(fact=fact, v_inter=v_inter, d_max=d_max)
    return synth.make_problem(n_bins=n_bins, nnz=nnz, n_sub=n_sub, seed=seed, contig_weights=weights, mean_len_bp=mean_len_bp,
                              accu=accu, param=par)


CASES = [
    # name, problem kwargs, (mode: quirk, n layouts, n_contigs range, p_circ), proposals per layout, K
    ("small sets, 3 sub-fragments, RF counts 1..9 (priced by k_tm)", dict(n_bins=70, nnz=1500, seed=11, n_sub=3, weights=(5, 3, 2), mean_len_bp=1500.0, accu=("random", 1, 9)),
     dict(quirk=True, layouts=2, contigs=(12, 25), p_circ=0.3), 4, 3),
    ("contigs of ~100 bins, 3 sub-fragments, uniform RF counts, window of ~20 bins", dict(n_bins=300, nnz=20000, seed=12, n_sub=3, weights=(5, 3, 2), mean_len_bp=700.0, accu=9, d_max=40.0),
     dict(quirk=True, layouts=2, contigs=(2, 5), p_circ=0.3), 3, 4),
    ("contigs of ~100 bins, 3 sub-fragments, RF counts 1..9 (no window with the trans-branch indexing)", dict(n_bins=260, nnz=15000, seed=13, n_sub=3, weights=(5, 3, 2), mean_len_bp=700.0, accu=("random", 1, 9), d_max=40.0),
     dict(quirk=True, layouts=1, contigs=(2, 4), p_circ=0.3), 3, 3),
    ("the same without the trans-branch indexing (window)", dict(n_bins=260, nnz=15000, seed=13, n_sub=3, weights=(5, 3, 2), mean_len_bp=700.0, accu=("random", 1, 9), d_max=40.0),
     dict(quirk=False, layouts=1, contigs=(2, 4), p_circ=0.3), 3, 3),
    ("single sub-fragment, contigs of hundreds of bins, K = 10, window of ~60 bins", dict(n_bins=1500, nnz=60000, seed=14, n_sub=1, weights=(6, 5, 3, 1), mean_len_bp=660.0, accu=1, fact=1e4, v_inter=1e-3, d_max=40.0),
     dict(quirk=True, layouts=2, contigs=(3, 6), p_circ=0.2), 3, 10),
    ("single sub-fragment, one contig of 900 bins and short ones", dict(n_bins=1000, nnz=40000, seed=15, n_sub=1, weights=(90, 2, 2, 2, 2, 2), mean_len_bp=660.0, accu=1, fact=1e4, v_inter=1e-3, d_max=25.0),
     dict(quirk=False, layouts=1, contigs=None, p_circ=0.0), 4, 5),
    # (round 5: with sub-fragments the union set is tiled by 32 fragments, and once a contig may hold more than 512 bins the unit list's entries
    # are 4 fragments long -- the halves of a wave walk them two at a time)
    ("contigs of 500-900 bins, 3 sub-fragments, uniform RF counts (tiles of 32, entries of 4)", dict(n_bins=1400, nnz=60000, seed=16, n_sub=3, weights=(5, 3), mean_len_bp=700.0, accu=9, d_max=60.0),
     dict(quirk=True, layouts=1, contigs=(2, 3), p_circ=0.3), 2, 3),
]


def layouts_and_proposals(P, cfg, n_props, K, seed):
    from tests.test_engine_gpu import random_state_for, relabel_ref
    from oracle import oracle as O
    rng = np.random.RandomState(seed)
    n = P["n_frags"]
    out = []
    for _ in range(cfg["layouts"]):
        if cfg["contigs"] is None:
            s = O.copy_state(P["S_o_A_frags"])
            s["id_c"][:] -= 1
            rev = rng.random_sample(n) < 0.3
            s["ori"][rev] = -1
        else:
            s = random_state_for(P, rng, n_contigs=int(rng.randint(*cfg["contigs"])), p_circ=cfg["p_circ"])
        max_id = relabel_ref(s)
        props = []
        for _ in range(n_props):
            fA = int(rng.randint(n))
            fBs = sorted(int(v) for v in rng.choice(np.setdiff1d(np.arange(n), [fA]), K, replace=False))
            props.append((fA, fBs))
        out.append((s, max_id, props))
    return out


def run_cases(only=None):
    """{case name: int64 [layouts, proposals, K, 13]} through whatever kernels the environment selects."""
    res = {}
    for ci, (name, pk, cfg, n_props, K) in enumerate(CASES):
        if only is not None and ci not in only:
            continue
        P = _problem(**pk)
        rows = []
        for s, max_id, props in layouts_and_proposals(P, cfg, n_props, K, 1000 + ci):
            e = Engine(0)
            e.upload_subfrags(P["np_sub_frags_id"], P["np_sub_frags_len_bp"], P["np_sub_frags_accu"], P["init_n_sub_frags"],
                              P["mean_squared_frags_per_bin"])
            e.upload_contacts(P["coo_row"], P["coo_col"], P["coo_val"])
            e.set_params(P["param_simu"])
            e.upload_frags(s)
            e.set_mode(ref_trans_accu=cfg["quirk"], strict=True)
            assert e.relabel_contigs() == max_id
            row = []
            for fA, fBs in props:
                d = e.eval_candidates(fA, fBs, max_id)
                # (NaN = a term beyond the fixed-point range: a circular contig closing over bp-sized fragments prices pairs at
                # > 2^31 under the circular model, kernels3.cu:135-166 -- both paths must flag the same candidates)
                q = np.where(np.isfinite(d), np.rint(np.nan_to_num(d) * Q_SCALE), -2.0 ** 62).astype(np.int64)
                row.append(q)   # (the float64 results are Q / 2^30 exactly)
            rows.append(np.stack(row))
            e.close()
        res[name] = np.stack(rows)
    return res


if __name__ == "__main__":
    only = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else None
    r = run_cases(only)
    np.savez(sys.argv[1], **{"case%d" % i: r[name] for i, (name, *_rest) in enumerate(CASES) if name in r})
