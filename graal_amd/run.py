"""Headless run from a dataset folder -- what pressing "start" in the reference's GUI does (``main_window.py:617-650`` ->
``simulation_loader.simulation.__init__`` -> ``main_gl.window.start_EM``), without wx / GLUT:

    python -m graal_amd.run --dataset DIR [--fasta genome.fa] --level 3 --cycles 100 --neighbours 3 --out OUT
    python -m graal_amd.run --dataset DIR --size-pyramid 1 --level 0 ...      (restriction-fragment resolution, BASELINE config 4)

DIR holds ``info_contigs.txt``, ``fragments_list.txt``, ``abs_fragments_contacts_weighted.txt`` (``README.md:111-113``).
Steps: build (or reuse) the pyramid, build the sampler inputs of the chosen level, fit the Rippe contact model, run the
MCMC cycles on the MI355X, write the trace files of ``main_gl.py:321-342`` and -- if a FASTA file is given -- the
scaffolded genome (``pyramid_sparse.py:1430-1488``)."""
import argparse
import os
import time

import numpy as np

from . import em
from . import pyramid as pyr


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--dataset", required=True)
    ap.add_argument("--fasta", default=None)
    ap.add_argument("--size-pyramid", type=int, default=4)
    ap.add_argument("--factor", type=int, default=3)
    ap.add_argument("--level", type=int, default=3, help="pyramid level of the bins: 1 .. size-pyramid - 1 as in the reference's GUI (main_window.py:452), "
                                                         "or 0 = full restriction-fragment resolution (bins = the filtered level-0 fragments, one "
                                                         "sub-fragment each: what the dense reference cannot run, BASELINE config 4)")
    ap.add_argument("--cycles", type=int, default=10)
    ap.add_argument("--neighbours", type=int, default=3)
    ap.add_argument("--blacklist", type=int, nargs="*", default=[0], help="contig ids to blacklist (0 = none)")
    ap.add_argument("--allow-repeats", action="store_true")
    ap.add_argument("--sample-params", action="store_true")
    ap.add_argument("--no-explode", action="store_true")
    ap.add_argument("--seed", type=int, default=None, help="seed of the numpy RandomState (the reference never seeds)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--arithmetic", choices=("strict", "trans_accu", "exact"), default="strict",
                    help="strict = the reference's float32 pixel arithmetic (default: traces are the reference's); exact = "
                         "mathematically exact candidate deltas (faster on long contigs)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--images", action="store_true", help="write pre_simu.tiff / post_em.tiff (the contact matrix in the genome's order before and after, "
                                                           "main_gl.py:213, 283) into the output folder")
    ap.add_argument("--no-fit", action="store_true", help="skip the Rippe fit; needs --param (8 floats: kuhn lm c1 slope d d_max fact v_inter)")
    ap.add_argument("--param", type=float, nargs=8, default=None, help="param_simu to run with instead of the fit's")
    args = ap.parse_args(argv)
    if not 0 <= args.level < args.size_pyramid:
        raise SystemExit("--level must be in 0 .. size-pyramid - 1 (levels >= 1: the level below holds the observations; 0: the level itself)")
    from .sampler import sampler
    root = os.path.join(args.dataset, "pyramids", "pyramid_%d_thresh_auto" % args.size_pyramid)
    if os.path.exists(os.path.join(root, "pyramid.npz")):
        P = pyr.Pyramid(root, args.size_pyramid)
    else:
        P = pyr.build_and_filter(args.dataset, args.size_pyramid, args.factor)
    inp = pyr.simulation_inputs(P, args.level, candidates_blacklist=args.blacklist, allow_repeats=args.allow_repeats)
    rng = np.random.RandomState(args.seed) if args.seed is not None else None
    smp = sampler(True, inp["S_o_A_frags"], inp["collector_id_repeats"], inp["frag_dispatcher"], inp["id_frag_duplicated"],
                  inp["id_frags_blacklisted"], inp["n_frags"], inp["n_new_frags"], inp["init_n_sub_frags"], inp["n_new_sub_frags"],
                  None, inp["hic_matrix_sub_sampled"], inp["np_sub_frags_len_bp"], inp["np_sub_frags_id"], inp["np_sub_frags_accu"],
                  inp["mean_squared_frags_per_bin"], inp["norm_vect_accu"], inp["S_o_A_sub_frags"], inp["hic_matrix"],
                  inp["mean_value_trans"], args.cycles, False, None, device=args.device, rng=rng,
                  reference_arithmetic=args.arithmetic)
    # simulation_loader.py:109-123: window and bin size of the fit from the initial layout
    g = smp.gpu_vect_frags
    g.copy_from_gpu()
    mean_dist_kb = float(g.l_cont_bp[g.start_bp == 0].mean()) / 1000.0
    size_bin_kb = float(g.len_bp.mean()) / 1000.0
    if args.no_fit:
        if args.param is None:
            raise SystemExit("--no-fit needs --param")
        smp.set_param_simu(np.asarray(args.param, dtype=np.float32))
    else:
        fit_ok = True
        try:
            with np.errstate(all="ignore"):
                smp.estimate_parameters(mean_dist_kb, size_bin_kb)
            fit_ok = bool(np.all(np.isfinite(smp._param_flat)))
        except Exception as e:      # (graal_set_params refuses parameters that are not finite / out of range)
            fit_ok = False
            print("graal_amd.run: the Rippe fit of this dataset gave no usable parameters (%s)" % e)
        if args.param is not None:
            smp.set_param_simu(np.asarray(args.param, dtype=np.float32))
        elif not fit_ok:
            # the reference would carry on with whatever leastsq returned (optim_rippe_curve_update.py:93-115) and score NaN from the first
            # step on; a headless run says so instead
            raise SystemExit("graal_amd.run: the Rippe fit did not converge to finite parameters on this dataset (histogram of %d distance bins, "
                             "%d of them without contacts): pass --param kuhn lm c1 slope d d_max fact v_inter" % (
                                 len(smp.bins), int(np.sum(np.asarray(smp.mean_contacts) <= 1e-10))))
    out = args.out or os.path.join(args.dataset, "graal_out")
    os.makedirs(out, exist_ok=True)
    images = (os.path.join(out, "pre_simu.tiff"), os.path.join(out, "post_em.tiff")) if args.images else None
    t0 = time.perf_counter()
    trace = em.run_em(smp, args.cycles, args.neighbours, rng=rng, sample_param=args.sample_params, scrambled=not args.no_explode,
                      matrix_files=images)
    dt = time.perf_counter() - t0
    em.save_behaviour_to_txt(trace, out)
    lev = P.get_level(args.level)
    if args.fasta:
        P.load_reference_sequence(args.fasta)
        g.copy_from_gpu()
        lev.generate_new_fasta(g, os.path.join(out, "genome.fasta"), os.path.join(out, "info_frags.txt"))
    n_steps = len(trace.likelihood)
    print("%d bins (%d fragments, %d sub-fragments), %d MCMC steps in %.1f s (%.0f us/step): %d contigs, logL %.6e, "
          "distance to the initial genome %.4f; traces in %s" % (inp["n_frags"], inp["n_new_frags"], inp["init_n_sub_frags"], n_steps,
                                                                 dt, 1e6 * dt / max(n_steps, 1), trace.n_contigs[-1],
                                                                 trace.likelihood[-1], trace.dist[-1], out))
    smp.free_gpu()
    return trace


if __name__ == "__main__":
    main()
