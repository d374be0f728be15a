"""GPU box: the same run at a stand-in's shape with the commits' own-pixel corrections carried (default) and with the reference's per-step full
evaluation (GRAAL_NO_OWN_PIXEL_CARRY=1): accepted moves identical, likelihood series equal to 1e-10 relative, generator state identical.
usage: python tools/carry_vs_per_step.py [C2|C3] [cycles]   ->  one summary line (profiles/r05_carry_vs_per_step.log)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from graal_amd import em, synth  # noqa: E402
from tools.run_configs import CONFIGS  # noqa: E402


def run(name, cycles, carry):
    if carry:
        os.environ.pop("GRAAL_NO_OWN_PIXEL_CARRY", None)
    else:
        os.environ["GRAAL_NO_OWN_PIXEL_CARRY"] = "1"
    n_bins, nnz, n_sub, _, K, accu = CONFIGS[name]
    P = synth.make_problem(n_bins=n_bins, nnz=nnz, n_sub=n_sub, seed=2014, contig_weights=(6.8, 6.2, 5.3, 4.6, 4.0, 3.4, 2.7),
                           mean_len_bp=660.0 * (27 if n_sub > 1 else 1) / max(n_sub, 1), accu=accu)
    rng = np.random.RandomState(1)
    smp = bench.build_sampler(P, rng, None, 0, "strict")
    assert smp._own_corr == carry
    t0 = time.perf_counter()
    tr = em.run_em(smp, cycles, K, rng=rng)
    dt = time.perf_counter() - t0
    out = (tr.mutations(), np.asarray(tr.likelihood), rng.get_state()[1].copy(), dt, smp.engine.run_counters())
    smp.free_gpu()
    return out


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C2"
    cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    a = run(name, cycles, True)
    b = run(name, cycles, False)
    same_moves = bool(np.array_equal(a[0], b[0]))
    same_state = bool(np.array_equal(a[2], b[2]))
    rel = float(np.max(np.abs(a[1] - b[1]) / np.abs(b[1])))
    print("%s, %d cycles = %d MCMC steps: accepted moves identical %s, generator state identical %s, likelihood series max relative difference %.2e; "
          "%.0f us/step carried (%d steps repaired by an evaluation), %.0f us/step with the per-step evaluation"
          % (name, cycles, len(a[1]), same_moves, same_state, rel, 1e6 * a[3] / len(a[1]), a[4]["carried_totals_repaired"], 1e6 * b[3] / len(b[1])), flush=True)
    assert same_moves and same_state and rel < 1e-9


if __name__ == "__main__":
    main()
