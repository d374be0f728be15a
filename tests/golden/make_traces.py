#!/usr/bin/env python3
"""Generates tests/golden/traces.json: accepted-move traces and likelihood series of full start_EM runs produced by the
ORACLE (the literal CPU restatement of cuda_lib_gl.sampler over the dense C restatement of kernels3.cu, oracle/), for a few
seeded synthetic problems.  The problems themselves are regenerated from their seeds by graal_amd/synth.py, so the fixture
only stores what to expect.  Run from the repo root:  python tests/golden/make_traces.py
(The reference itself cannot run here -- PyCUDA + an NVIDIA GPU -- so these vectors pin the oracle/engine pair to each
other across rounds; the oracle in turn is pinned to the reference's known answers in appendix_e.json.)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graal_amd import em, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASES = [  # name, n_sub, seed, n_bins, nnz, cycles, neighbours, blacklist, repeated bins[, generic]
    ("single_sub", 1, 41, 70, 1200, 2, 3, [], []),
    ("three_sub", 3, 42, 60, 1500, 2, 4, [], []),
    ("blacklist", 3, 48, 50, 900, 2, 4, [3, 4, 5, 17, 31], []),
    ("repeats", 3, 92, 45, 900, 2, 3, [], [5, 18, 30]),
    # generic bp lengths (float32 kb coordinates inexact) and RF counts 1..9 per sub-fragment: what the reference's own arithmetic is
    # sensitive to (coordinate rounding noise of unchanged pairs, the trans-branch RF-count indexing: kernels3.cu:2997-3078, 3155 / 3638);
    # the oracle runs with fix_trans_accu=False, the engine in its default mode (reference arithmetic)
    ("generic_coordinates", 3, 57, 60, 1500, 2, 4, [], [], True),
]


def problem(n_sub, seed, n_bins, nnz, blacklist, repeats=(), generic=False):
    par = synth.make_param_simu(fact=200.0, v_inter=0.02)
    P = synth.make_problem(n_bins=n_bins, nnz=nnz, n_sub=n_sub, seed=seed, contig_weights=(5, 4, 3), mean_len_bp=2000.0,
                           accu=(("random", 1, 9) if generic else 9) if n_sub > 1 else 1, param=par, grid_bp=None if generic else 2000)
    P = synth.with_dense(P)
    if len(repeats):
        P = synth.add_repeats(P, repeats, 2)   # two extra copies of every repeated bin
    P["id_frags_blacklisted"] = list(blacklist)
    return P


def main():
    out = {}
    for name, n_sub, seed, n_bins, nnz, cycles, delta, black, reps, *rest in CASES:
        generic = bool(rest and rest[0])
        P = problem(n_sub, seed, n_bins, nnz, black, reps, generic)
        ora = O.OracleSampler(P, np.random.RandomState(seed), fix_trans_accu=not generic)
        t = em.run_em(ora, cycles, delta, rng=ora.rng)
        out[name] = {"n_sub": n_sub, "seed": seed, "n_bins": n_bins, "nnz": nnz, "cycles": cycles, "neighbours": delta,
                     "blacklist": black, "repeats": reps, "generic": generic, "mutations": np.asarray(t.mutations()).tolist(),
                     "likelihood": [float(x) for x in t.likelihood], "n_contigs": [int(x) for x in t.n_contigs],
                     "dist": [float(x) for x in t.dist],
                     "final_id_c": ora.gpu_vect_frags["id_c"].tolist(), "final_pos": ora.gpu_vect_frags["pos"].tolist(),
                     "final_ori": ora.gpu_vect_frags["ori"].tolist(), "final_activ": ora.gpu_vect_frags["activ"].tolist()}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "traces.json"), "w") as f:
        json.dump(out, f)
    print({k: len(v["mutations"]) for k, v in out.items()})


if __name__ == "__main__":
    main()
