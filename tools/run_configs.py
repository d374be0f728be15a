#!/usr/bin/env python3
"""GPU box: the BASELINE.json configurations 2-4 on synthetic stand-ins of the documented shapes (the
S1 / T. reesei tarballs are not in this image): full headless start_EM runs (explode + n cycles) through the drop-in
sampler, wall time per MCMC step, contigs left, final log-likelihood.  Usage: python tools/run_configs.py [C2 C3 C4 C4D] [--arithmetic strict|exact] [--cycles N]

C4D = BASELINE config 4 through its DATA PATH: the C4 stand-in written as the reference's 3-file text dataset (synth.write_dataset: one line
per read pair), then `python -m graal_amd.run --dataset ... --size-pyramid 1 --level 0` -- pyramid builder with its sparsity filter, level-0
sampler inputs (graal_amd/pyramid.py:simulation_inputs), Rippe fit of the level-0 histogram, explode + MCMC cycles, trace files."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from graal_amd import em, synth  # noqa: E402

CONFIGS = {  # name: (n_bins, nnz, n_sub, cycles, neighbours, accu)
    "C2": (1086, 120_000, 3, 100, 3, 9),      # S. cerevisiae S1, pyramid level 3, 100 MCMC cycles
    "C3": (3500, 600_000, 3, 500, 3, 9),      # T. reesei QM6a, level 3, 500 cycles (N0 = #RF / 27, assumed 3,500)
    "C4": (40000, 8_000_000, 1, 2, 5, 1),     # T. reesei level 0: bins = restriction fragments (2 cycles timed here)
}


def run_c4_from_a_dataset(arith, cycles):
    import shutil
    import tempfile
    from graal_amd import run
    n_bins, nnz, n_sub, _, K, accu = CONFIGS["C4"]
    t0 = time.perf_counter()
    P = synth.make_problem(n_bins=n_bins, nnz=nnz, n_sub=1, seed=2014, contig_weights=(6.8, 6.2, 5.3, 4.6, 4.0, 3.4, 2.7), mean_len_bp=660.0, accu=1)
    t_gen = time.perf_counter() - t0
    base = tempfile.mkdtemp(prefix="graal_c4_dataset_")
    try:
        t0 = time.perf_counter()
        n_reads = synth.write_dataset(P, base)
        t_write = time.perf_counter() - t0
        size_mb = os.path.getsize(os.path.join(base, "abs_fragments_contacts_weighted.txt")) / 1e6
        print("C4D: stand-in generated in %.1f s, written as a 3-file dataset in %.1f s: %d fragments, %d contacts = %d read lines (%.0f MB)"
              % (t_gen, t_write, n_bins, nnz, n_reads, size_mb), flush=True)
        t0 = time.perf_counter()
        # (the stand-in's contacts stop at the generator's d_max while the fit's histogram runs to the mean contig length: its last bins are
        # empty and the reference's log-space least squares does not converge on them -- the run takes the generator's parameters, the fit still runs)
        tr = run.main(["--dataset", base, "--size-pyramid", "1", "--level", "0", "--cycles", str(cycles), "--neighbours", str(K), "--seed", "1",
                       "--arithmetic", arith, "--out", os.path.join(base, "out"), "--param"] + [repr(float(x)) for x in P["param_simu"]])
        print("C4D [%s]: dataset -> pyramid -> level-0 inputs -> fit -> explode + %d cycles -> traces: %.1f s in all, %d MCMC steps, %d contigs left"
              % (arith, cycles, time.perf_counter() - t0, len(tr.likelihood), tr.n_contigs[-1]), flush=True)
    finally:
        shutil.rmtree(base, ignore_errors=True)


def main():
    args = sys.argv[1:]
    arith = args[args.index("--arithmetic") + 1] if "--arithmetic" in args else "strict"
    cyc = int(args[args.index("--cycles") + 1]) if "--cycles" in args else None
    names = [a for a in args if a in CONFIGS or a == "C4D"] or ["C2", "C3", "C4"]
    for name in names:
        if name == "C4D":
            run_c4_from_a_dataset(arith, cyc or 2)
            continue
        n_bins, nnz, n_sub, cycles, K, accu = CONFIGS[name]
        cycles = cyc or cycles
        P = synth.make_problem(n_bins=n_bins, nnz=nnz, n_sub=n_sub, seed=2014, contig_weights=(6.8, 6.2, 5.3, 4.6, 4.0, 3.4, 2.7),
                               mean_len_bp=660.0 * (27 if n_sub > 1 else 1) / max(n_sub, 1), accu=accu)
        rng = np.random.RandomState(1)
        t0 = time.perf_counter()
        smp = bench.build_sampler(P, rng, None, 0, arith)
        t_setup = time.perf_counter() - t0
        t_explode = [0.0]
        explode = smp.explode_genome

        def timed_explode(*a, **k):      # (start_EM's explode_genome -- one relabel + one commit per fragment, cuda_lib_gl.py:1539-1556 -- timed on its own)
            te = time.perf_counter()
            r = explode(*a, **k)
            smp.engine.begin_step()
            t_explode[0] = time.perf_counter() - te
            return r
        smp.explode_genome = timed_explode
        t0 = time.perf_counter()
        tr = em.run_em(smp, cycles, K, rng=rng)
        dt = time.perf_counter() - t0
        n_steps = len(tr.likelihood)
        full = smp.eval_likelihood()
        carried, how = tr.likelihood[-1], "carried logL"
        if getattr(smp, "_own_corr", False):     # the last commit's own-pixel correction is still pending (the next step would add it)
            corr, ok = smp.engine.take_carry_correction()
            carried, how = carried + (corr if ok else float("nan")), "carried logL + the last commit's own pixels (%.3e)" % corr
        print("%s [%s]: %d bins x %d sub, %d contacts, %d cycles x %d neighbours: setup %.1f s, explode + %d MCMC steps in %.1f s = %.0f us/step (explode_genome alone %.2f s: %.0f us per MCMC step without it), "
              "%d contigs left (started exploded: %d), %s %.6e vs full re-evaluation %.6e (rel %.1e), %d steps repaired by an evaluation"
              % (name, arith, n_bins, n_sub, nnz, cycles, K, t_setup, n_steps, dt, 1e6 * dt / n_steps, t_explode[0], 1e6 * (dt - t_explode[0]) / n_steps, tr.n_contigs[-1], n_bins,
                 how, carried, full, abs(full - carried) / abs(full), smp.engine.run_counters()["carried_totals_repaired"]), flush=True)
        smp.free_gpu()


if __name__ == "__main__":
    main()
