#!/usr/bin/env python3
"""GPU box: soak of k_strict2's hand-off behind k_gprep (strict2.h: a word in memory instead of an event; round-4 review item 3).

Two samplers on the same problem, seed and arithmetic run the same MCMC side by side -- A with the default hand-off (the completion
word; GRAAL_STRICT_GWAIT unset), B ordered by the event (GRAAL_STRICT_GWAIT=0, the anchor) -- and EVERY ONE of the 13 x K candidate scores
of EVERY step is compared (float64 values of the int64 sums: bit equality), next to the accepted move.  Usage:

    python tools/handoff_soak.py [C2|C3|C4] [--cycles N] [--no-acquire]      (--no-acquire: A with GRAAL_GP_ACQUIRE=0, round 4's form)

Prints one line per shape: steps, steps whose tiled kernel followed the word (A) / the event (B), differing steps (must be 0), fallbacks,
us per step of each."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from graal_amd import synth  # noqa: E402

SHAPES = {  # name: (n_bins, nnz, n_sub, cycles, neighbours, accu)
    "C2": (1086, 120_000, 3, 20, 3, 9),
    "C3": (3500, 600_000, 3, 8, 3, 9),
    "C4": (40000, 8_000_000, 1, 3, 5, 1),
}


def build(P, seed, env):
    for k in ("GRAAL_STRICT_GWAIT", "GRAAL_GP_ACQUIRE", "GRAAL_GP_WAIT_TICKS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    rng = np.random.RandomState(seed)
    smp = bench.build_sampler(P, rng, None, 0, "strict")      # (the switches are read when the handle is created)
    return smp, rng


def main():
    args = sys.argv[1:]
    cyc = int(args[args.index("--cycles") + 1]) if "--cycles" in args else None
    env_a = {"GRAAL_GP_ACQUIRE": "0"} if "--no-acquire" in args else {}
    names = [a for a in args if a in SHAPES] or ["C2", "C4"]
    for name in names:
        n_bins, nnz, n_sub, cycles, K, accu = SHAPES[name]
        cycles = cyc or cycles
        P = synth.make_problem(n_bins=n_bins, nnz=nnz, n_sub=n_sub, seed=2014, contig_weights=synth.C5_CONTIG_WEIGHTS,
                               mean_len_bp=660.0 * (27 if n_sub > 1 else 1) / max(n_sub, 1), accu=accu)
        P["S_o_A_frags"] = bench.exploded_layout(P)
        A, rng_a = build(P, 1, env_a)
        B, rng_b = build(P, 1, {"GRAAL_STRICT_GWAIT": "0"})
        for k in ("GRAAL_STRICT_GWAIT", "GRAAL_GP_ACQUIRE"):
            os.environ.pop(k, None)
        A.init_likelihood(); B.init_likelihood()
        n = int(A.n_new_frags)
        order_a, order_b = np.arange(n, dtype=np.int32), np.arange(n, dtype=np.int32)
        steps = differing = moves_differ = 0
        ta = tb = 0.0
        first_bad = None
        for c in range(cycles):
            rng_a.shuffle(order_a); rng_b.shuffle(order_b)
            assert np.array_equal(order_a, order_b)
            for i in order_a:
                t0 = time.perf_counter()
                ra = A.step_max_likelihood(int(i), K)
                t1 = time.perf_counter()
                rb = B.step_max_likelihood(int(i), K)
                t2 = time.perf_counter()
                ta += t1 - t0; tb += t2 - t1
                sa, sb = np.asarray(A.score), np.asarray(B.score)
                same = sa.shape == sb.shape and np.array_equal(sa, sb, equal_nan=True)
                if not same:
                    differing += 1
                    if first_bad is None:
                        first_bad = (steps, int(i), sa.tolist(), sb.tolist())
                if (int(ra[5]), int(ra[6])) != (int(rb[5]), int(rb[6])):
                    moves_differ += 1
                steps += 1
            st = A.engine.layout_stats()
            print("  %s cycle %d: %d steps so far, %d contigs, longest %d, differing steps %d" % (name, c + 1, steps, int(st[0]), int(st[4]), differing), flush=True)
        ca, cb = A.engine.run_counters(), B.engine.run_counters()
        print("%s: %d bins x %d sub, %d contacts, K = %d, %d cycles = %d steps x %d scores: steps with differing scores %d, differing moves %d | "
              "A (%s): tiled kernel behind the word %d / the event %d, flat %d, fallbacks %d, %.1f us/step | B (event): behind the word %d / the event %d, "
              "flat %d, fallbacks %d, %.1f us/step" % (name, n_bins, n_sub, nnz, K, cycles, steps, 13 * K, differing, moves_differ,
                                                       "word, acquire only after a wait" if env_a else "word + acquire", ca["strict2_behind_the_word"],
                                                       ca["strict2_behind_the_event"], ca["flat_launches"], ca["fallbacks"], 1e6 * ta / steps,
                                                       cb["strict2_behind_the_word"], cb["strict2_behind_the_event"], cb["flat_launches"],
                                                       cb["fallbacks"], 1e6 * tb / steps), flush=True)
        if first_bad:
            print("  first differing step:", first_bad[:2], flush=True)
        A.free_gpu(); B.free_gpu()
        if differing or moves_differ:
            raise SystemExit(1)


if __name__ == "__main__":
    main()
