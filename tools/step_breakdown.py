#!/usr/bin/env python3
"""Host wall-clock breakdown of FULL MCMC steps (relabel + statistics, proposal, scoring, sampling, commit) on the bench
workload: which part of sampler.step_max_likelihood the time goes to.  Run on the GPU box:

    python tools/step_breakdown.py [--steps 2000] [--dist]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from graal_amd import dist as gdist, synth  # noqa: E402
from graal_amd import sampler as S  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--n-bins", type=int, default=50000)
    ap.add_argument("--nnz", type=int, default=20_000_000)
    ap.add_argument("--dist", action="store_true", help="with the genome distance every step (compute_dist=True)")
    ap.add_argument("--n-sub", type=int, default=1)
    ap.add_argument("--counters", action="store_true", help="also average the engine's per-step counters (adds a copy per step)")
    ap.add_argument("--original", action="store_true", help="start from the 7 reference contigs (late-stage regime), no explode")
    a = ap.parse_args()
    if a.n_sub > 1:   # the shapes of tools/run_configs.py (C2 / C3)
        P = synth.make_problem(n_bins=a.n_bins, nnz=a.nnz, n_sub=a.n_sub, seed=2014, contig_weights=(6.8, 6.2, 5.3, 4.6, 4.0, 3.4, 2.7),
                               mean_len_bp=660.0 * 27 / a.n_sub, accu=9)
    else:
        P = synth.make_problem(n_bins=a.n_bins, nnz=a.nnz, n_sub=1, seed=20141217)
    if not a.original:
        P["S_o_A_frags"] = bench.exploded_layout(P)
    rng = np.random.RandomState(20141217)
    smp = bench.build_sampler(P, rng, gdist.Group(0, 1), 0)
    smp.compute_dist = a.dist
    smp.init_likelihood()
    n = int(smp.n_new_frags)
    order = np.arange(n, dtype=np.int32)
    rng.shuffle(order)
    order = np.concatenate([order] * (1 + (2000 + a.steps) // n))
    for i in order[:2000]:
        smp.step_max_likelihood(int(i), 5)
    acc = {}

    def wrap(obj, name, label):
        fn = getattr(obj, name)

        def w(*args, **kw):
            t = time.perf_counter()
            try:
                return fn(*args, **kw)
            finally:
                acc[label] = acc.get(label, 0.0) + time.perf_counter() - t
        setattr(obj, name, w)

    wrap(smp.engine, "begin_step", "begin_step (k_stats + relabel launches, wait for the statistics)")
    wrap(smp, "return_neighbours", "return_neighbours (host)")
    wrap(smp, "_candidate_deltas", "scoring (k_tm || k_scan, wait)")
    wrap(S, "select_move", "select_move (host)")
    wrap(smp, "test_copy_struct", "commit launch (k_apply)")
    wrap(smp, "dist_inter_genome", "genome distance (k_dist, wait)")
    wrap(smp, "_full_likelihood", "full re-evaluation (circular contigs around / every 512 steps)")
    cnt = np.zeros(4)
    orig = smp._candidate_deltas

    def counted(*args, **kw):
        r = orig(*args, **kw)
        cnt[:] += smp.engine.last_counters()
        return r
    if a.counters:
        smp._candidate_deltas = counted
    t0 = time.perf_counter()
    for i in order[2000:2000 + a.steps]:
        smp.step_max_likelihood(int(i), 5)
    total = time.perf_counter() - t0
    if a.counters:
        print("per step: relevant (contact, neighbour) pairs %.0f, queued contacts %.0f, mass work items %.0f" % tuple(cnt[1:4] / a.steps))
    st = smp.engine.layout_stats()
    print("layout: %d contigs, longest %d fragments" % (st[0], st[4]))
    print("full MCMC step: %.1f us  (n=%d, nnz=%d, %d steps, compute_dist=%s)" % (1e6 * total / a.steps, n, a.nnz, a.steps, a.dist))
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        print("  %-75s %7.1f us" % (k, 1e6 * v / a.steps))
    print("  %-75s %7.1f us" % ("rest of step_max_likelihood (python)", 1e6 * (total - sum(acc.values())) / a.steps))
    smp.free_gpu()


if __name__ == "__main__":
    main()
