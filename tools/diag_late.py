#!/usr/bin/env python3
"""GPU box diagnostic: C5 on its 7 original contigs -- one move, engine delta vs engine full(after) - full(before) vs the numpy
re-score (contacts part and mass part separately)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from graal_amd import synth  # noqa: E402
from graal_amd.lib import Q_SCALE  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle.sparse_numpy import SparseScorer  # noqa: E402
from tests import util  # noqa: E402


def main():
    n_bins = int(os.environ.get("DIAG_NBINS", 50000))
    nnz = int(os.environ.get("DIAG_NNZ", 20_000_000))
    P = synth.make_problem(n_bins=n_bins, nnz=nnz, n_sub=1, seed=20141217)
    if os.environ.get("DIAG_DMAX"):      # a narrower window: the same model cut off earlier
        par = np.array(P["param_simu"], dtype=np.float32)
        par[5] = float(os.environ["DIAG_DMAX"])
        P["param_simu"] = par
    rng = np.random.RandomState(11)
    smp = bench.build_sampler(P, rng, None, 0)
    smp.init_likelihood()
    sc = SparseScorer(P["coo_row"], P["coo_col"], P["coo_val"], P["np_sub_frags_id"], P["np_sub_frags_len_bp"],
                      P["np_sub_frags_accu"], P["mean_squared_frags_per_bin"], P["param_simu"])
    rng2 = np.random.RandomState(12)
    fA = int(os.environ.get("DIAG_FA", rng2.randint(0, n_bins)))
    nb = smp.return_neighbours(fA, 5); nb.sort()
    max_id = smp.modify_gl_cuda_buffer(0)
    smp.gpu_vect_frags.copy_from_gpu()
    s0 = {k: np.copy(v) for k, v in smp.gpu_vect_frags.as_dict().items()}
    q_before = smp.engine.eval_full_q()
    d = smp._candidate_deltas(fA, nb, max_id)
    print("fA", fA, "nb", nb, "max_id", max_id, "counters", smp.engine.last_counters())
    t0 = time.time()
    c0 = sc.centres(s0)
    np_nnz0, np_mass0 = sc.nnz_part(s0, c0), sc.mass_cis(s0, c0)
    print("numpy before: nnz %.6f mass %.6f (%.1f s)" % (np_nnz0, np_mass0, time.time() - t0))
    print("engine before: nnz %.6f mass %.6f" % (q_before[0] / Q_SCALE + sc.c_lf, -(q_before[1] / Q_SCALE) - sc.t_all()))
    ops = [int(x) for x in os.environ.get("DIAG_OPS", "11,0,6,9").split(",")]
    for k in range(min(int(os.environ.get("DIAG_K", 1)), len(nb))):
        for op in ops:
            s1, stale = util.oracle_candidate(s0, fA, nb[k], op, max_id)
            c1 = sc.centres(s1)
            np_nnz1, np_mass1 = sc.nnz_part(s1, c1), sc.mass_cis(s1, c1)
            e2 = bench.build_sampler(dict(P, S_o_A_frags=s1), np.random.RandomState(1), None, 0)
            e2.modify_gl_cuda_buffer(0)
            q_after = e2.engine.eval_full_q()
            e2.free_gpu()
            eng_nnz = (q_after[0] - q_before[0]) / Q_SCALE
            eng_mass = -(q_after[1] - q_before[1]) / Q_SCALE
            print("k %d fB %d op %2d: delta %.4f | engine full diff %.4f (nnz %.4f, -mass %.4f) | numpy full diff %.4f (nnz %.4f, -mass %.4f)"
                  % (k, nb[k], op, d[k, op], eng_nnz - eng_mass, eng_nnz, -eng_mass, (np_nnz1 - np_nnz0) - (np_mass1 - np_mass0),
                     np_nnz1 - np_nnz0, -(np_mass1 - np_mass0)), flush=True)
    smp.free_gpu()


if __name__ == "__main__":
    main()
