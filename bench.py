#!/usr/bin/env python3
"""bench.py -- candidate logL evaluations per second of the per-move likelihood scan (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1 with WORLD_SIZE unset starts the N ranks itself: a child
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`
is spawned BEFORE anything here touches the GPU, its output is relayed and its exit code returned; under an external
torchrun (WORLD_SIZE set) the process is one of the ranks.

Workload (BASELINE.json configs[4], SURVEY.md section 8d "C5"): synthetic 50,000-fragment / 20,000,000-contact map,
generator seed 20141217; layout = exploded genome + 2,000 real MCMC warm-up steps (K = 5 neighbours).  One timed
"step" = the scoring phase of one MCMC step: ONE fused pass over the contact list for the 13 x 5 = 65 candidates of a
(fA, 5 neighbours) proposal, their expected-mass tasks, the exchange of the ranks' 65 int64 values (N > 1) and the
device->host hand-over of the result (N > 1: each rank's GPU publishes its 65 int64 sums to pinned host memory shared by
the ranks of the node and every host adds them up; the same region is then timed once more with an all-reduce of a device
buffer instead, reported as `exchange_alt`).  Inputs are resident in HBM; proposals are drawn beforehand.  With N GPUs the SAME
contact list is sharded N ways (strong scaling).

`value` is measured in REFERENCE ARITHMETIC (`config.reference_arithmetic` = "strict": every pixel of contig(A) u contig(B) re-priced
from float32 kb coordinates like sub_compute_likelihood, kernels3.cu:3259-3718 -- the sampler's default, whose traces are the
reference's); the same proposals are then timed in the exact arithmetic (`other_arithmetic`).  `--arithmetic exact` swaps the two.

Extra fields: `value_1000` (SURVEY 8d's 1,000-step region), `full_mcmc_step_ms` (graal_step: relabel + proposal + scoring + sampling +
commit + statistics), `full_eval_ms` (one full likelihood evaluation: what a nuisance-parameter step adds to every MCMC step),
`full_mcmc_step_sample_param_ms`, `late_stage` (the same map with its 7 original contigs -- millions of queued contacts and thousands
of work units per step, sharded over the ranks -- in both arithmetics, with its own full-step / full-evaluation figures),
`exchange_alt` and `distributed` (N > 1).

Output: one JSON line on rank 0 (contract in the task statement) with `roofline` (fused scan kernel k_scan: algorithmic bytes
= 4 B x contacts (row words) + n/8 B (bitmap) + 16 B x queued contacts per launch; duration = MEDIAN of HIP event pairs around the
launches of the timed region (every 8th step) and of an untimed repeat of the same steps with a pair on every step; plus the flat
`hbm_control_*` fields: the same kernel, same layout, same proposals over a list whose row array (480 MB) cannot stay in the
256 MiB Infinity Cache, so HBM-vs-cache is settled by measurement), `cpu_baseline` (numpy re-score of the same sparse likelihood on
the host) and `cpu_baseline_dense` (the C restatement of the reference's dense sub_compute_likelihood on the C2 stand-in), N = 1 only.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
# VALU issue ceiling the late stage's kernel is priced against: 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction -- what one
# wave's stream sustains on a SIMD (MI355X_MICROARCH.md, constants table: `v_fma_f32` 4 cycles for one wave alone, 2 with several waves
# per SIMD for single-rate float32 / integer instructions; the kernel's mix is half float64 FMAs)
VALU_PEAK_WAVE_INSTR = 256 * 4 * 2.4e9 / 4


def build_sampler(P, rng, group, device, arithmetic="strict"):
    from graal_amd.sampler import sampler
    return sampler(True, P["S_o_A_frags"], P["collector_id_repeats"], P["frag_dispatcher"], [], [], P["n_frags"],
                   P["n_new_frags"], P["init_n_sub_frags"], P["n_new_sub_frags"], None,
                   (P["bin_coo_row"], P["bin_coo_col"], P["bin_coo_val"]), P["np_sub_frags_len_bp"],
                   P["np_sub_frags_id"], P["np_sub_frags_accu"], P["mean_squared_frags_per_bin"], None, None,
                   (P["coo_row"], P["coo_col"], P["coo_val"]), P["mean_value_trans"], 1, False, None,
                   device=device, rng=rng, group=group, param_simu=P["param_simu"], compute_dist=False,
                   reference_arithmetic=arithmetic)


def exploded_layout(P):
    """Every fragment its own contig (what explode_genome produces, cuda_lib_gl.py:1539-1556), built directly."""
    n = P["n_frags"]
    s = {k: np.array(P["S_o_A_frags"][k], dtype=np.int32, copy=True) for k in P["S_o_A_frags"]}
    s["pos"][:] = 0
    s["id_c"][:] = np.arange(n)
    s["start_bp"][:] = 0
    s["circ"][:] = 0
    s["prev"][:] = -1
    s["next"][:] = -1
    s["l_cont"][:] = 1
    s["l_cont_bp"][:] = s["len_bp"]
    s["ori"][:] = 1
    return s


CPU_BASELINE_PAIRS = ((1234, 4321), (777, 31337), (20141, 217), (4242, 12345), (9001, 40404), (27182, 31415))
CPU_BASELINE_OPS = (0, 6, 3, 4)   # eject; insert right of fB; split-insert @ left, reversed; split-insert @ right


def cpu_baseline(P, state, budget_s=24.0, param=None, gpu_deltas=None):
    """numpy re-score (oracle/sparse_numpy.py) of whole candidates on the host: the checker, timed, never shipped.
    The candidates are the reference's mutation kernels (the oracle's restatement) applied to the layout the GPU holds; with
    `gpu_deltas` -- the engine's candidate deltas for the same (fA, fB, op), scored OUTSIDE every timed region -- the values do not go to
    waste: re-score(candidate) - re-score(current) is held against them (`max_rel_diff_vs_gpu`, in units of |logL|; the reference's own
    cross-check, cuda_lib_gl.py:2196-2220)."""
    from oracle import oracle as O
    from oracle.sparse_numpy import SparseScorer
    sc = SparseScorer(P["coo_row"], P["coo_col"], P["coo_val"], P["np_sub_frags_id"], P["np_sub_frags_len_bp"],
                      P["np_sub_frags_accu"], P["mean_squared_frags_per_bin"], P["param_simu"] if param is None else param)
    n = P["n_frags"]
    max_id = int(state["id_c"].max())
    pop, ids = O.new_state(n), np.zeros(n, np.int32)
    cands = []
    for fA, fB in CPU_BASELINE_PAIRS:
        fA, fB = fA % n, fB % n
        O.DenseOracle.pop_out(pop, state, ids, fA, max_id)
        cands.append(((fA, fB, 0), O.copy_state(pop)))                     # op 0: eject
        for op, which, ori in ((6, 3, 1), (3, 1, -1), (4, 2, 1)):
            out = O.new_state(n)
            O.DenseOracle.pop_in(which, out, pop, fA, fB, int(ids.max()), ori)
            cands.append(((fA, fB, op), out))
    base = sc.full(state, same_bin=False) if gpu_deltas is not None else None   # (the current layout: untimed, not a candidate)
    done, t0 = 0, time.perf_counter()
    values = []
    for key, c in cands:                                                   # ~0.7 s each: 24 candidates, ~17 s of CPU work (>= 20)
        values.append((key, sc.full(c, same_bin=False)))
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    out = {"value": done / dt, "unit": "candidate logL evals/s", "cores": 1, "kind": "port",
           "sample": "%d whole-candidate numpy float32/float64 re-scores of the same %d-contact state (%.1f s)" % (
               done, len(P["coo_row"]), dt)}
    if gpu_deltas is not None:
        diffs = [abs((v - base) - gpu_deltas[key]) for key, v in values if key in gpu_deltas]
        out["candidates_compared_with_gpu"] = len(diffs)
        out["max_abs_diff_vs_gpu_logL_units"] = max(diffs) if diffs else None
        out["max_rel_diff_vs_gpu"] = max(diffs) / abs(base) if diffs else None
        out["max_rel_diff_note"] = ("max over the candidates of |(re-score(candidate) - re-score(current)) - GPU strict delta| / |logL|; "
                                    "the GPU values were computed outside the timed regions for the same (fA, fB, op)")
        out["largest_abs_delta_compared"] = max(abs(gpu_deltas[key]) for key, _ in values if key in gpu_deltas) if diffs else None
    return out


def cpu_baseline_dense(budget_s=8.0):
    """SURVEY 8d(i) "reference arithmetic on CPU": the C restatement of the reference's dense kernels (oracle/graal_oracle.c:
    sub_compute_likelihood over every pixel of contig(A) u contig(B), kernels3.cu:3259-3718) on the C2 stand-in (1,086 bins x 3
    sub-fragments, its 7 original contigs), one host thread.  The checker, timed -- never shipped."""
    from graal_amd import synth
    from oracle import oracle as O
    from tests import util
    par = synth.make_param_simu(fact=200.0, v_inter=0.02)
    P = synth.with_dense(synth.make_problem(n_bins=1086, nnz=120_000, n_sub=3, seed=2016, contig_weights=synth.C5_CONTIG_WEIGHTS,
                                            mean_len_bp=660.0, accu=("random", 1, 9), param=par))
    dense = O.DenseOracle(P["hic_matrix"], P["np_sub_frags_id"], P["np_sub_frags_len_bp"], P["np_sub_frags_accu"],
                          P["frag_dispatcher"], P["collector_id_repeats"], P["n_frags"], P["mean_squared_frags_per_bin"],
                          P["param_simu"], fix_trans_accu=False)
    s = O.copy_state(P["S_o_A_frags"])
    s["id_c"][:] -= 1
    per_pix = np.zeros(dense.n_pix)
    dense.evaluate(s, per_pix)
    n = P["n_frags"]
    max_id = int(s["id_c"].max())
    rng = np.random.RandomState(1)
    done, pixels, t0 = 0, 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        fA, fB = (int(v) for v in rng.choice(n, 2, replace=False))
        sub = np.sort(np.nonzero((s["id_c"] == s["id_c"][fA]) | (s["id_c"] == s["id_c"][fB]))[0])
        for op in range(13):
            cand, _ = util.oracle_candidate(s, fA, fB, op, max_id)
            dense.sub_compute(cand, sub, [], np.arange(n, dtype=np.int32), per_pix)
            done += 1
            pixels += len(sub) * (len(sub) - 1) // 2
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "candidate logL evals/s", "cores": 1, "kind": "port",
            "workload": "C2 stand-in (1,086 bins x 3 sub-fragments, 120,000 contacts, 7 contigs)",
            "sample": "%d candidates = %d dense pixels re-priced by the C restatement of sub_compute_likelihood (%.1f s)" % (done, pixels, dt)}


def hbm_control(P, smp, props, max_id, n, repeat):
    """The Infinity-Cache control of the roofline figure: the SAME kernel, layout and proposals over the contact list with every
    contact listed `repeat` times (row-sorted still; 6 x 20 M contacts = a 480 MB row array, which cannot stay in the 256 MiB
    cache between two launches).  Returns the in-step (event pair per launch) and back-to-back durations."""
    from graal_amd.lib import Engine
    e = Engine(smp.engine.device)
    try:
        e.upload_subfrags(P["np_sub_frags_id"], P["np_sub_frags_len_bp"], P["np_sub_frags_accu"], P["init_n_sub_frags"],
                          P["mean_squared_frags_per_bin"])
        e.upload_contacts(np.repeat(P["coo_row"], repeat), np.repeat(P["coo_col"], repeat), np.repeat(P["coo_val"], repeat))
        e.set_params(P["param_simu"])
        smp.gpu_vect_frags.copy_from_gpu()
        e.upload_frags(smp.gpu_vect_frags.as_dict())
        _, mid = e.begin_step()
        assert mid == int(max_id)
        e.set_timing(1)
        for f, nb in props[:4]:
            e.eval_candidates(f, nb, mid)
        use = props[4:4 + 24]
        t0 = time.perf_counter()
        for f, nb in use:
            e.eval_candidates(f, nb, mid)
        wall = (time.perf_counter() - t0) / max(1, len(use))
        ms = e.scan_times(len(use))
        c = e.last_counters()
        replay_ms = e.time_scan(len(use[-1][1]), reps=20)
        nnz = int(e.nnz)
        bytes_per_launch = 4.0 * nnz + n / 8.0 + 16.0 * float(c[2])
        scan_s = float(np.mean(ms)) * 1e-3
        return {"contacts": nnz, "row_array_MB": 4.0 * nnz / 1e6, "how": "every contact of the C5 list listed %d times" % repeat,
                "bytes_per_launch": bytes_per_launch, "avg_launch_ms": scan_s * 1e3, "launches_timed": int(len(ms)),
                "achieved": bytes_per_launch / scan_s / 1e9, "frac": bytes_per_launch / scan_s / 1e9 / HBM_PEAK_GBS,
                "back_to_back_replay_ms": replay_ms, "frac_back_to_back_replays": bytes_per_launch / (replay_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "host_wall_per_step_ms": wall * 1e3}
    finally:
        e.close()


def launch_ranks(args, argv):
    """--gpus N without a torchrun environment: start the N ranks as a child (this process has not touched the GPU)."""
    # (--standalone: the launcher binds port 0 itself and keeps it -- a port picked here by bind / close and handed over could be taken in between)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "2")
    p = subprocess.run(cmd, env=env)
    return p.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--n-bins", type=int, default=int(os.environ.get("GRAAL_BENCH_NBINS", 50000)))
    ap.add_argument("--nnz", type=int, default=int(os.environ.get("GRAAL_BENCH_NNZ", 20_000_000)))
    ap.add_argument("--mcmc-warmup", type=int, default=int(os.environ.get("GRAAL_BENCH_MCMC_WARMUP", 2000)))
    ap.add_argument("--neighbours", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--arithmetic", choices=("strict", "exact"), default="strict",
                    help="arithmetic of the headline `value`: strict = the reference's (kernels3.cu:3259-3718; the sampler's default), "
                         "exact = mathematically exact deltas; the other one is reported next to it")
    ap.add_argument("--long-steps", type=int, default=1000, help="steps of the second timed region (`value_1000`, SURVEY 8d); 0 = skip")
    ap.add_argument("--no-late-stage", action="store_true", help="skip the extra measurement on the map's 7 original contigs")
    ap.add_argument("--no-hbm-control", action="store_true", help="skip the Infinity-Cache control of the roofline figure")
    ap.add_argument("--control-repeat", type=int, default=6, help="hbm_control: list every contact this many times")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse "
                                                      "the multi-rank code path with several ranks on ONE GPU)")
    ap.add_argument("--layout", choices=("exploded", "original"), default="exploded",
                    help="exploded (+ MCMC warm-up) is the BASELINE workload; original = the 7 reference contigs (late-stage regime)")
    ap.add_argument("--late-only", action="store_true", help="only the late stage's reference-arithmetic scoring steps (what profiles/valu_late.json is profiled on)")
    ap.add_argument("--late-repeats", type=int, default=1, help="--late-only: run the timed steps this many times")
    ap.add_argument("--dry-run", action="store_true", help="launcher / rendezvous / sharding check without a GPU (CPU test-suite)")
    args = ap.parse_args()

    from graal_amd import dist as gdist
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args, sys.argv[1:]))
    rank, world, local = gdist.env_world()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world))

    if args.dry_run:   # no GPU, no engine: only what the launcher, the rendezvous and the sharding do
        import torch
        import torch.distributed as td
        if world > 1:
            td.init_process_group("gloo")
        take = gdist.shard_take(args.nnz, rank, world)
        mine = (take.stop - take.start) if isinstance(take, slice) else len(take)
        t = torch.tensor([mine], dtype=torch.int64)
        if world > 1:
            td.all_reduce(t)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "contacts": int(t[0]), "contacts_rank0": int(mine)}), flush=True)
        if world > 1:
            td.barrier()
            td.destroy_process_group()
        return

    from graal_amd import synth
    import torch
    td = None
    if world > 1:
        import datetime
        import torch.distributed as td
        # (a collective that one rank never enters -- e.g. it failed inside the optional late-stage extra -- must not hang
        # the others for ever: they time out, record the error and still let rank 0 print the headline)
        tmo = datetime.timedelta(seconds=300)
        if args.backend == "nccl":
            torch.cuda.set_device(local)
            td.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=tmo)
        else:
            local = 0
            torch.cuda.set_device(0)
            td.init_process_group(args.backend, timeout=tmo)
    group = gdist.Group(rank, world)

    def sync_all():
        group.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        # (a device tensor only with RCCL: gloo's handling of device tensors is a side path of a CPU backend -- two GPU memory faults at the base of
        # torch's small-tensor pool, in the 2-rank rehearsal of this file, both between two calls of this function, went away with it)
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if td.get_backend() == "nccl" else "cpu")
        td.all_reduce(t, op=td.ReduceOp.MAX)
        return float(t.cpu()[0])

    def timed_region(sm, pr, mid, n_steps):
        nc = 0
        sync_all()
        tt = time.perf_counter()
        for f, nb in pr[:n_steps]:
            sm._candidate_deltas(f, nb, mid)
            nc += 13 * len(nb)
        sync_all()
        return nc, max_over_ranks(time.perf_counter() - tt)

    def phase(msg):   # (diagnostics: GRAAL_BENCH_PHASES=1 prints where the run is)
        if os.environ.get("GRAAL_BENCH_PHASES"):
            print("[bench rank %d, %.2f s] %s" % (rank, time.perf_counter(), msg), file=sys.stderr, flush=True)
    phase("start")
    t_gen = time.perf_counter()
    P = synth.make_problem(n_bins=args.n_bins, nnz=args.nnz, n_sub=1, seed=20141217)
    soa_original = P["S_o_A_frags"]
    if args.layout == "exploded":
        P["S_o_A_frags"] = exploded_layout(P)
    t_gen = time.perf_counter() - t_gen
    n = args.n_bins
    K = args.neighbours
    other = "exact" if args.arithmetic == "strict" else "strict"

    def set_arithmetic(sm, which):
        sm.engine.set_mode(ref_trans_accu=which == "strict", strict=which == "strict")

    def run_late(strict_only=False, repeats=1):
        """The same map in its LATE stage (the 7 original contigs of 2.7-6.8k fragments): scoring phase in both arithmetics, the VALU
        roofline of the reference-arithmetic kernel (k_strict2), a full MCMC step, a full evaluation."""
        P2 = dict(P)
        P2["S_o_A_frags"] = soa_original
        rng2 = np.random.RandomState(20141217)
        smp2 = build_sampler(P2, rng2, group, local if world > 1 else 0, args.arithmetic)
        phase("late stage: sampler built")
        smp2.init_likelihood()
        max_id2 = smp2.modify_gl_cuda_buffer(0)
        props2 = []
        for f in rng2.randint(0, n, size=3 + 12):
            nb = smp2.return_neighbours(int(f), K)
            nb.sort()
            props2.append((int(f), nb))
        st2 = smp2.engine.layout_stats()
        late = {"workload": "same map, its %d original contigs (longest %d fragments)" % (int(st2[0]), int(st2[4]))}
        smp2.engine.set_timing(1)     # (an event pair around k_scan and around k_strict2 on every step: a few us of a 1.5 ms step)
        for which in ((args.arithmetic,) if strict_only else (args.arithmetic, other)):
            set_arithmetic(smp2, which)
            for f, nb in props2[:3]:
                smp2._candidate_deltas(f, nb, max_id2)
            n_timed = len(props2[3:])
            for _ in range(max(1, repeats)):
                n_cand2, tl = timed_region(smp2, props2[3:], max_id2, n_timed)
            c2 = smp2.engine.last_counters()
            blk = {"value": n_cand2 / tl, "unit": "candidate logL evals/s", "ms_per_step": 1e3 * tl / n_timed, "steps": n_timed,
                   "queued_contacts_last_step_this_rank": int(c2[2]), "work_units_last_step_this_rank": int(c2[3])}
            if rank == 0:
                # what of the step shards over the ranks (the streaming pass and the tiled pricing kernel: their durations by HIP events on this
                # rank) and what does not (the rest of the step: tables, union set and classes, hand-out, host) -- the line the driver's multi-GPU
                # run can be read against
                sc_us = 1e3 * float(np.median(smp2.engine.scan_times(n_timed)))
                st_us = 1e3 * float(np.mean(smp2.engine.strict_times(n_timed))) if which == "strict" else None
                blk["sharded_kernels_us_this_rank"] = {"k_scan": sc_us, "k_strict2": st_us}
                blk["non_sharded_us_per_step"] = 1e3 * blk["ms_per_step"] - sc_us - (st_us or 0.0) if which == "strict" else None
            if which == "strict" and rank == 0:
                blk["roofline"] = valu_roofline(smp2, n_timed)
            if which == args.arithmetic:
                late.update(blk)
                late["arithmetic"] = which
            else:
                late["other_arithmetic"] = dict(blk, arithmetic=which)
        if strict_only:
            smp2.free_gpu()
            return late
        # a full MCMC step (+ the nuisance-parameter step of the reference GUI's default) in this regime
        phase("late stage: scoring regions done")
        set_arithmetic(smp2, args.arithmetic)
        smp2.bins = np.arange(1.0, 41.0, 1.0)
        smp2.step_nuisance_parameters(0, 0, 1)
        order2 = np.arange(n, dtype=np.int32)
        rng2.shuffle(order2)
        torch.cuda.synchronize()
        tl = time.perf_counter()
        for i in order2[:8]:
            smp2.step_max_likelihood(int(i), K)
        torch.cuda.synchronize()
        late["full_mcmc_step_ms"] = 1e3 * (time.perf_counter() - tl) / 8
        tl = time.perf_counter()
        for i in order2[8:16]:
            smp2.step_max_likelihood(int(i), K)
            smp2.step_nuisance_parameters(0, 0, 1)
        torch.cuda.synchronize()
        late["full_mcmc_step_sample_param_ms"] = 1e3 * (time.perf_counter() - tl) / 8
        smp2.engine.eval_full_q()
        tl = time.perf_counter()
        for _ in range(5):
            smp2.engine.eval_full_q()
        late["full_eval_ms"] = 1e3 * (time.perf_counter() - tl) / 5
        late["fallbacks"] = smp2.engine.run_counters()["fallbacks"]
        smp2.free_gpu()
        return late

    def valu_roofline(sm, n_timed):
        """VALU roofline of the reference-arithmetic kernel over the late stage's timed steps: wave instructions per launch from a committed
        rocprofv3 --pmc SQ_INSTS_VALU pass of exactly these launches (profiles/valu_late.json, written by tools/summarize_prof.py from
        `bench.py --late-only`) divided by the launch duration measured HERE (HIP events around the kernel on its stream); peak = 1,024
        SIMDs x 2.4 GHz / 4 cycles per wave instruction (MI355X_MICROARCH.md)."""
        ms = sm.engine.strict_times(n_timed)
        dur_s = float(np.mean(ms)) * 1e-3
        r = {"bound": "valu", "kernel": "k_strict2", "peak": VALU_PEAK_WAVE_INSTR, "unit": "wave instructions/s", "avg_launch_ms": dur_s * 1e3,
             "launches_timed": int(len(ms)), "achieved": None, "frac": None, "wave_instr_per_launch": None}
        vpath = os.path.join(ROOT, "profiles", "valu_late.json")
        if os.path.exists(vpath):
            try:
                vj = json.load(open(vpath))
                if vj.get("n_frags") == n and vj.get("nnz") == len(P["coo_row"]):
                    r["wave_instr_per_launch"] = vj["wave_instr_per_launch"]
                    r["achieved"] = vj["wave_instr_per_launch"] / dur_s
                    r["frac"] = r["achieved"] / VALU_PEAK_WAVE_INSTR
                    r["kernel_trace_avg_ms"] = vj.get("kernel_avg_us_rocprof", 0.0) * 1e-3
                    r["frac_kernel_trace"] = vj["wave_instr_per_launch"] / (vj["kernel_avg_us_rocprof"] * 1e-6) / VALU_PEAK_WAVE_INSTR
                    r["instr_source"] = "committed rocprofv3 --pmc SQ_INSTS_VALU pass of `bench.py --late-only` (profiles/valu_late.json, " + str(vj.get("source", "profiles/")) + "): mean over the same launches, not measured in this run"
                    if vj.get("peak_for_the_mix_wave_instr_per_s"):
                        # the ceiling for THIS kernel's instruction mix (round-4 review item 5): per-type SQ_INSTS_VALU_* counts of the same launches priced
                        # with the sustained issue rates measured by tools/valu_issue_micro.hip at 4 waves per SIMD (profiles/r05_valu_issue.log; what
                        # no per-type counter claims is priced at the fastest class, so the ceiling errs high).  `peak` / `frac` are against it;
                        # the round-4 figure -- 1,024 SIMDs x 2.4 GHz / 4 cycles, one float64 instruction stream of one wave per SIMD -- stays next to it
                        r["peak_one_wave_f64_stream"] = VALU_PEAK_WAVE_INSTR
                        r["frac_of_the_one_wave_f64_stream"] = r["frac"]
                        r["peak"] = vj["peak_for_the_mix_wave_instr_per_s"]
                        r["frac"] = r["achieved"] / r["peak"]
                        r["frac_kernel_trace"] = vj["wave_instr_per_launch"] / (vj["kernel_avg_us_rocprof"] * 1e-6) / r["peak"]
                        r["peak_basis"] = ("ceiling for the kernel's measured instruction mix: sum over instruction classes of (wave instructions per launch, rocprofv3 "
                                           "--pmc SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_{F32,F64}, _INT32, _INT64, _CVT) / (sustained chip-wide issue rate of that class at 4 "
                                           "waves per SIMD, tools/valu_issue_micro.hip: 0.50-0.53e12/s float64 add / mul / fma, 0.93e12 float32 add / mul and int32, "
                                           "0.59e12 float32 fma, 0.56e12 conversions and 64-bit integer, 0.30e12 float32 transcendental); %.0f %% of the instructions are float64"
                                           % (100.0 * vj.get("float64_share", 0.0)))
                        r["simd_cycles_per_valu_instr"] = vj.get("simd_cycles_per_valu_instr")
            except Exception:
                pass
        return r

    if args.late_only:   # (what tools/profile_gpu.sh profiles for late_stage.roofline: nothing but the late stage's strict scoring steps)
        late = run_late(strict_only=True, repeats=args.late_repeats)
        if rank == 0:
            print(json.dumps({"late_stage": late}), flush=True)
        return

    rng = np.random.RandomState(20141217)
    t_setup = time.perf_counter()
    phase("problem generated")
    smp = build_sampler(P, rng, group, local if world > 1 else 0, args.arithmetic)
    t_setup = time.perf_counter() - t_setup
    phase("sampler built")

    assert n == int(smp.n_new_frags)

    # ---- proposals for the timed regions, drawn BEFORE the MCMC warm-up (the neighbour distribution is the contact map's, not the
    # layout's): drawn right in front of the timed region -- ~1,000 Python calls, 30 ms -- they left the GPU idle, and the first timed
    # steps were 10 us slower than the rest (r03: 1.645 M in the driver's 20-step run against 1.754 M over 1,000).  The generator is put back
    # where it was: the warm-up steps draw what they always drew.
    total = args.warmup + max(args.steps, args.long_steps)
    rng_state = rng.get_state()
    frags = rng.randint(0, n, size=total)
    props = []
    for f in frags:
        nb = smp.return_neighbours(int(f), K)
        nb.sort()
        props.append((int(f), nb))
    rng.set_state(rng_state)

    # ---- layout: exploded genome + real MCMC warm-up steps (every rank runs the same, sharded, steps) --------
    smp.init_likelihood()
    phase("first full evaluation done")
    t_mcmc = time.perf_counter()
    order = np.arange(n, dtype=np.int32)
    rng.shuffle(order)
    for j_, i in enumerate(order[:args.mcmc_warmup]):
        smp.step_max_likelihood(int(i), K)
        if j_ < 4 or j_ % 500 == 0:
            phase("MCMC warm-up step %d done" % j_)
    torch.cuda.synchronize()
    phase("MCMC warm-up done")
    t_mcmc = time.perf_counter() - t_mcmc
    stats = smp.engine.layout_stats()
    max_id = smp.modify_gl_cuda_buffer(0)

    # The timed region carries a HIP event pair around k_scan on every 8th step only (a pair costs the step ~10-25 us of
    # command-processor marker gaps: 64 us per step with a pair on every step against 40 without, profiles/r02_event_sampling.log);
    # right behind it the same steps run once more, untimed, with a pair on EVERY step (at most 64): those samples -- in-step
    # launches of the same proposals -- price the roofline, so that a 20-step run has 20 of them rather than 2.
    # (a run of fewer than 64 steps -- the driver's 20 -- carries NO pair in its timed region: the one pair it used to carry cost a single step
    # 100 us, a tenth of the region; its roofline samples are those of the untimed repeat, one per step)
    EVENT_EVERY = int(os.environ.get("GRAAL_BENCH_EVENT_EVERY", 8 if args.steps >= 64 else 0))
    smp.engine.set_timing(EVENT_EVERY)
    # (set-up, untimed: the GPU has been busy for a tenth of a second -- 2,000 warm-up steps -- since it sat idle through the problem's
    # generation, and its clocks are still on their way up: the first 6-16 steps of a 20-step region took 50-70 us, the rest 35
    # (GRAAL_BENCH_STEP_TIMES=1).  A quarter of a second of the same scoring steps first; then the W warm-up steps and the K timed ones.)
    t_settle = time.perf_counter()
    settle_s = float(os.environ.get("GRAAL_BENCH_SETTLE_S", 0.25))
    settle_t = [] if os.environ.get("GRAAL_BENCH_STEP_TIMES") else None
    settle_sync = int(os.environ.get("GRAAL_BENCH_SETTLE_SYNC", 0))   # (diagnostics: a device synchronize every n-th step of the settling phase)
    # (several ranks: a scoring step is a collective -- every rank must run the same number of them, so the ranks agree on every round of 64:
    # one more while ANY rank's clock says so)
    while max_over_ranks(1.0 if time.perf_counter() - t_settle < settle_s else 0.0) > 0.0:
        for f, nb in props[args.warmup:args.warmup + 64]:
            smp._candidate_deltas(f, nb, max_id)
            if settle_t is not None:
                settle_t.append(time.perf_counter())
                if settle_sync and len(settle_t) % settle_sync == 0:
                    torch.cuda.synchronize()
    # (the HIP runtime keeps the commands of a stream until a synchronize -- or, every ~1,000 launches without one, a marker of its own -- lets it
    # release them, and the release is felt: 3-12 steps of 60 us and, a millisecond later, one of 110-150 us, every 1,019 steps of this loop
    # = every 35.8 ms (GRAAL_BENCH_STEP_TIMES=1 lists them; 0.4 % of a long run, and value_1000 carries its share).  A quarter of a second is 7.0
    # of those periods: the tail of the 7th fell into the K timed steps of about a third of the driver-style runs -- a 115 us step and a
    # 180-210 us closing synchronize, 1.1-1.4 M where the other runs gave 1.70-1.77 M.  So the settling phase ends with a synchronize (the
    # release of what it left), 64 more steps (2 ms: that release is over) and a second synchronize (128 commands: nothing to feel); the W
    # warm-up steps and the K timed ones then run between two of the runtime's releases rather than, by accident, across one.)
    sync_all()
    for f, nb in props[args.warmup:args.warmup + 64]:
        smp._candidate_deltas(f, nb, max_id)
    sync_all()
    for f, nb in props[:args.warmup]:
        smp._candidate_deltas(f, nb, max_id)
    n_cand = 0
    step_t = [] if os.environ.get("GRAAL_BENCH_STEP_TIMES") else None   # (diagnostics: host clock after every step of the timed region)
    timed_props = props[args.warmup:args.warmup + args.steps]
    sync_all()
    t0 = time.perf_counter()
    for f, nb in timed_props:
        smp._candidate_deltas(f, nb, max_id)   # records a HIP event pair around k_scan on the stream it runs on
        n_cand += 13 * len(nb)
        if step_t is not None:
            step_t.append(time.perf_counter())
    sync_all()
    elapsed = time.perf_counter() - t0
    if step_t is not None and rank == 0:
        print("per-step us:", " ".join("%.1f" % (1e6 * (b - a)) for a, b in zip([t0] + step_t[:-1], step_t)),
              "| closing barrier + synchronize: %.1f" % (1e6 * (t0 + elapsed - step_t[-1])), file=sys.stderr)
    if settle_t and rank == 0:   # (diagnostics: when, behind the start of the settling phase, steps took more than 1.3x the median)
        d_ = np.diff(np.asarray(settle_t))
        med = float(np.median(d_))
        slow = np.flatnonzero(d_ > 1.3 * med)
        eps, j = [], 0
        while j < len(slow):
            k = j
            while k + 1 < len(slow) and slow[k + 1] - slow[k] <= 4:
                k += 1
            eps.append("%.1f ms: %d steps, mean %.0f us" % (1e3 * (settle_t[slow[j]] - t_settle), slow[k] - slow[j] + 1, 1e6 * float(d_[slow[j]:slow[k] + 1].mean())))
            j = k + 1
        print("settling: %d steps, median %.1f us, slow episodes: %s" % (len(d_), 1e6 * med, "; ".join(eps[:60])), file=sys.stderr)
    if step_t is not None and rank == 0:
        # what the bracket itself costs (diagnostics): a synchronize with nothing pending, the first step behind one, the first step behind a pause without one
        def one(i):
            ta = time.perf_counter()
            smp._candidate_deltas(props[args.warmup + i][0], props[args.warmup + i][1], max_id)
            return 1e6 * (time.perf_counter() - ta)
        def sync_us():
            ta = time.perf_counter()
            torch.cuda.synchronize()
            return 1e6 * (time.perf_counter() - ta)
        for rep in range(3):
            a = [one(i) for i in range(6)]
            s1, s2 = sync_us(), sync_us()
            b = [one(i) for i in range(6)]
            ta = time.perf_counter()
            while time.perf_counter() - ta < 200e-6:
                pass
            c = [one(i) for i in range(6)]
            time.sleep(0.002)
            d = [one(i) for i in range(6)]
            print("bracket: steps %s | synchronize %.1f, again %.1f | steps behind it %s | behind a 200 us spin %s | behind a 2 ms sleep %s" % (
                " ".join("%.0f" % x for x in a), s1, s2, " ".join("%.0f" % x for x in b), " ".join("%.0f" % x for x in c), " ".join("%.0f" % x for x in d)), file=sys.stderr)
    n_timed_pairs = min(args.steps // EVENT_EVERY, 1024) if EVENT_EVERY > 0 else 0
    scan_ms_timed = smp.engine.scan_times(n_timed_pairs) if n_timed_pairs else np.zeros(0, np.float32)   # pairs of the timed region
    elapsed = max_over_ranks(elapsed)
    n2 = min(args.steps, 64)
    smp.engine.set_timing(1)
    for f, nb in props[args.warmup:args.warmup + n2]:
        smp._candidate_deltas(f, nb, max_id)
    sync_all()
    scan_ms = np.concatenate([smp.engine.scan_times(n2), scan_ms_timed])
    smp.engine.set_timing(EVENT_EVERY if EVENT_EVERY > 0 else 8)
    counters = smp.engine.last_counters()
    # SURVEY 8d's region: 1,000 steps (the driver's 20 steps last under a millisecond)
    phase("timed region and its event-pair repeat done")
    long_region = None
    if args.long_steps > 0:
        nc, tl_ = timed_region(smp, props[args.warmup:], max_id, args.long_steps)
        long_region = {"value_1000": nc / tl_, "ms_per_step_1000": 1e3 * tl_ / args.long_steps, "steps_1000": args.long_steps}
    phase("long region done")
    # the same steps in the OTHER arithmetic (same engine, same proposals)
    set_arithmetic(smp, other)
    for f, nb in props[:args.warmup]:
        smp._candidate_deltas(f, nb, max_id)
    n_other = max(args.steps, min(args.long_steps, 300))
    nc, to_ = timed_region(smp, props[args.warmup:], max_id, n_other)
    other_block = {"arithmetic": other, "value": nc / to_, "unit": "candidate logL evals/s", "ms_per_step": 1e3 * to_ / n_other, "steps": n_other}
    set_arithmetic(smp, args.arithmetic)
    phase("other arithmetic done")
    alt = None
    # for reference: back-to-back replays of the last step's scan between two events (per-launch event overhead amortised)
    scan_replay_ms = smp.engine.time_scan(len(props[-1][1]), reps=100)
    scan_isolated_ms = smp.engine.time_scan(len(props[-1][1]), reps=-40)   # median of isolated replays (device idle in between)
    phase("scan replays done")

    # ---- Infinity-Cache control of the roofline figure (rank 0's GPU, N = 1) ---------------------------------------
    control = None
    if world == 1 and not args.no_hbm_control and args.control_repeat > 1:
        try:
            control = hbm_control(P, smp, props, max_id, n, args.control_repeat)
        except Exception as e:   # an extra: never costs the headline
            control = {"error": repr(e)}

    # ---- one full likelihood evaluation (every nuisance-parameter step pays one: cuda_lib_gl.py:1986-2017) ---------
    smp.engine.eval_full_q()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    n_fe = 20
    for _ in range(n_fe):
        smp.engine.eval_full_q()
    full_eval_s = (time.perf_counter() - t1) / n_fe
    phase("full evaluations done")

    # ---- full MCMC steps (scoring + sampling + commit + relabel), reported as an extra ----------------------------
    t1 = time.perf_counter()
    n_full = 500   # (an extra next to the headline: always enough steps for a stable figure, 30 ms)
    for j, i in enumerate(order[args.mcmc_warmup:args.mcmc_warmup + n_full]):
        smp.step_max_likelihood(int(i), K)
        if os.environ.get("GRAAL_BENCH_PHASES") and j % 25 == 24:     # (diagnostics of the open two-ranks-on-one-GPU fault: DESIGN.md section 9)
            phase("full MCMC step %d done (%d since the last full evaluation)" % (j, smp._steps_since_full))
    torch.cuda.synchronize()
    full_step_s = (time.perf_counter() - t1) / n_full
    phase("full MCMC steps done")
    # the same with the reference GUI's default "sample parameters" (main_gl.py:258-262): one nuisance-parameter Metropolis
    # step -- one full evaluation under test parameters, plus a scipy fsolve for d_max on the host -- after every MCMC step
    smp.bins = np.arange(1.0, 41.0, 1.0)
    smp.step_nuisance_parameters(0, 0, 1)   # (untimed: the first call imports scipy.optimize, 0.17 s)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    n_sp = min(50, max(1, args.steps))
    for i in order[args.mcmc_warmup + n_full:args.mcmc_warmup + n_full + n_sp]:
        smp.step_max_likelihood(int(i), K)
        smp.step_nuisance_parameters(0, 0, 1)
    torch.cuda.synchronize()
    full_step_sp_s = (time.perf_counter() - t1) / n_sp

    # ---- extra: the same map in its LATE stage (the 7 original contigs of 2.7-6.8k fragments; every step prices thousands of
    # expected-mass work items and millions of queued contacts, all of it sharded over the ranks) -- reported next to the
    # headline, which is the short-contig regime SURVEY 8d defines ----------------------------------------------------
    late = None
    phase("headline, long region, other arithmetic, full evaluation, full steps and nuisance steps done")
    if args.layout == "exploded" and not args.no_late_stage:
        try:
            late = run_late()
            phase("late stage done")
        except Exception as e:   # an extra must not cost the headline line (a rank that fails alone makes the others' next
            late = {"error": repr(e)}   # collective time out after 300 s: they land here too)

    def run_exchange_alt():
        """N > 1: the timed region once more with the OTHER way of summing the ranks' 13*K int64 values -- north_star's wording: ONE RCCL
        all-reduce of the per-shard logL vector per step.  With torch.distributed on RCCL the library drives it itself (graal_attach_rccl:
        ncclAllReduce on the engine's stream, the total published behind it); with the gloo rehearsal it is torch's all-reduce of the device
        buffer.  Runs LAST, behind the headline line: nothing multi-GPU can be rehearsed on the one-GPU boxes this was built on, and a
        collective that hangs must not cost the line."""
        if not (world > 1 and smp.exchange == "host") or os.environ.get("GRAAL_BENCH_NO_ALT"):
            return None
        smp.engine.detach_exchange()
        smp.exchange = "rccl"
        smp._attach_rccl_c(opt_in=True)   # (the library-driven all-reduce is opt-in -- GRAAL_RCCL_C -- until a multi-GPU run has covered it: this IS that run, last in the job)
        for f, nb in props[:args.warmup]:
            smp._candidate_deltas(f, nb, max_id)
        sync_all()
        ta = time.perf_counter()
        for f, nb in props[args.warmup:args.warmup + args.steps]:
            smp._candidate_deltas(f, nb, max_id)
        sync_all()
        ta = max_over_ranks(time.perf_counter() - ta)
        return {"exchange": ("one ncclAllReduce(3 x 130 int64) per step on the engine's stream, driven by the library (graal_attach_rccl)" if smp._rccl_c else
                             "%s all-reduce of a device buffer (graal_eval_candidates_q + torch.distributed)" % args.backend),
                "driven_by_the_library": bool(smp._rccl_c), "value": n_cand / ta, "ms_per_step": 1e3 * ta / args.steps}

    out = None
    if rank == 0:
        nnz_local = smp.engine.nnz
        # what the streaming pass must read: the row word of every contact (4 B), the affected-fragment bitmap
        # (n/8 B) and, for the queued contacts only, the 16-byte queue entry (SURVEY 8d priced a naive pass
        # at 12 B per contact; col words of affected rows that fail the second test are not counted -> conservative)
        bytes_per_launch = 4.0 * nnz_local + n / 8.0 + 16.0 * float(counters[2])
        # kernel duration: HIP events on the stream the kernel runs on.  (a) an event pair around the launches of the timed
        # region: the duration of a launch as the sampler experiences it -- block launch ramp, the prologue that builds the
        # affected-fragment bitmap, the stream, the drain.  This prices the roofline; rocprofv3's kernel-trace average
        # (profiles/) agrees with it.  (b) 100 back-to-back replays of the last step's scan between two events, reported next
        # to it: there the ramp and the prologue of one launch overlap the tail of the previous one, so it measures the
        # streaming phase alone.
        # (an event pair also spans whatever the HOST does between recording the first event and submitting the kernel: now and
        # then a sample is ten times the others -- 250 us instead of 23, always among the first pairs after a synchronisation.
        # Those are host stalls, not kernel time (rocprofv3's per-kernel durations never show them): samples above twice the
        # median are left out of the mean, and counted.)
        med = float(np.median(scan_ms))
        kept = scan_ms[scan_ms <= 2.0 * med]
        scan_s = med * 1e-3                       # the MEDIAN launch prices the roofline (the trimmed and the plain mean are reported)
        achieved = bytes_per_launch / scan_s / 1e9
        replay_s = scan_replay_ms * 1e-3
        traffic = None
        trace_us = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")  # HBM bytes / launch from a committed rocprofv3 --pmc run
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("nnz") == int(nnz_local) and tj.get("n_frags") == n:
                    traffic = tj.get("hbm_bytes_per_launch")
                    trace_us = tj.get("kernel_avg_us_rocprof_in_a_step")
            except Exception:
                traffic = None
        out = {
            "metric": "candidate logL evals/sec", "value": n_cand / elapsed, "unit": "candidate logL evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32 model / f64 log / int64 Q30 sums",
            "data": "synthetic",
            "config": {"workload": "C5 synthetic %d-fragment / %d-contact map, %s + %d MCMC warm-up steps"
                                   % (n, len(P["coo_row"]), args.layout, args.mcmc_warmup),
                       "reference_arithmetic": args.arithmetic,
                       "reference_arithmetic_note": "strict = every pixel of contig(A) u contig(B) re-priced from float32 kb coordinates like "
                                                    "sub_compute_likelihood (kernels3.cu:3259-3718) + the trans-branch RF-count indexing: the "
                                                    "sampler's default, traces are the reference's; exact = mathematically exact deltas",
                       "neighbours_per_step": K, "candidates_per_step": 13 * K, "contacts_per_gpu": int(nnz_local),
                       "n_contigs": int(stats[0]), "max_contig_len": int(stats[4]),
                       "parallelism": "contacts sharded x%d, %s" % (world, {
                           "none": "single rank", "rccl": "1 all-reduce(65 x int64)/step (%s)" % args.backend,
                           "host": "65 x int64 per rank and step published to pinned host memory shared by the ranks, summed by every host"
                       }[smp.exchange])},
            "roofline": dict({"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "committed rocprofv3 --pmc run of this workload (profiles/traffic.json), not measured in this run",
                         "kernel": "k_scan", "bytes_per_launch": bytes_per_launch, "avg_launch_ms": scan_s * 1e3,
                         "launch_ms_median": med, "launch_ms_mean_all_samples": float(np.mean(scan_ms)),
                         "launch_ms_mean_without_host_stalls": float(np.mean(kept)),
                         "launches_timed": int(len(scan_ms)), "host_stall_samples": int(len(scan_ms) - len(kept)),
                         "launch_ms_samples": [round(float(x), 5) for x in scan_ms[:96]],
                         # (an event pair around ONE kernel also spans the command processor's handling of the two markers: the
                         # kernel-trace duration of the same kernel in the same flow is ~2 us shorter.  `frac` stays with the events.)
                         "kernel_trace_avg_ms": None if trace_us is None else trace_us * 1e-3,
                         "frac_kernel_trace": None if trace_us is None else bytes_per_launch / (trace_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "kernel_trace_source": "committed rocprofv3 --kernel-trace run of this workload (profiles/r04_rocprof_c5.md, k_scan [in a step]), not measured in this run",
                         "back_to_back_replay_ms": replay_s * 1e3,
                         "frac_back_to_back_replays": bytes_per_launch / replay_s / 1e9 / HBM_PEAK_GBS,
                         "isolated_replay_ms": scan_isolated_ms,
                         "working_set_note": "the 80 MB row array is re-read every step and fits the 256 MiB Infinity Cache: "
                                             "the hbm_control_* fields time the same kernel on a 480 MB row array"},
                        **({} if not control else {"hbm_control_" + k_: v_ for k_, v_ in control.items()})),
            "phase_ms": {"k_scan": scan_s * 1e3, "host_wall_per_step": 1e3 * elapsed / args.steps},
            "relevant_pairs_last_step": int(counters[1]), "queued_contacts_last_step": int(counters[2]),
            "mass_items_last_step": int(counters[3]),
            "full_mcmc_step_ms": 1e3 * full_step_s, "full_eval_ms": 1e3 * full_eval_s,
            "full_mcmc_step_sample_param_ms": 1e3 * full_step_sp_s,
            "setup_s": {"generate": t_gen, "sampler": t_setup, "mcmc_warmup": t_mcmc},
        }
        # steps the engine had to REPEAT behind events because an in-kernel wait between two kernels of a step ran out (k_tm waiting for the
        # scan's announcement, k_strict2 for k_gprep's completion word: include/graal_hip.h, graal_run_counters) over this handle's whole
        # life -- warm-up, settling, timed regions and extras; 0 unless a tool serialises the dispatches
        rc_ = smp.engine.run_counters()
        out["fallbacks"] = rc_["fallbacks"]
        out["engine_counters"] = rc_
        out["other_arithmetic"] = other_block
        if long_region:
            out.update(long_region)
        # per regime: the step, the part of it that shards over the ranks (kernel durations by HIP events on this rank) and the part that does not,
        # next to what DESIGN.md section 6 expects of 8 GPUs -- so that the driver's scaling run can be read against the estimate
        out["regimes"] = {
            "headline (exploded + MCMC warm-up: contigs of a few fragments)": {
                "ms_per_step": 1e3 * elapsed / args.steps, "sharded_kernels_us_this_rank": {"k_scan": scan_s * 1e6},
                "non_sharded_us_per_step": 1e6 * elapsed / args.steps - scan_s * 1e6,
                "expected_at_8_gpus": "the pass over an 8th of the list keeps its launch floor (~5 us), tables + finish + host round trip stay: <= 1.5x"},
            "late stage (the map's 7 original contigs)": None if not late or "error" in late else {
                "ms_per_step": late.get("ms_per_step"), "sharded_kernels_us_this_rank": late.get("sharded_kernels_us_this_rank"),
                "non_sharded_us_per_step": late.get("non_sharded_us_per_step"),
                "expected_at_8_gpus": "units dealt (ti + tj) % world, contacts by shard: ~0.23 ms per step, ~6x -- the regime that can meet north_star's >= 6x"}}
        if world > 1:
            out["distributed"] = {"backend": td.get_backend(), "ranks": int(td.get_world_size()), "exchange": smp.exchange,
                                  "rccl_ranks": int(td.get_world_size()) if td.get_backend() == "nccl" else 0,
                                  "value_produced_by_exchange": smp.exchange,
                                  "north_star_exchange": "one RCCL all-reduce of the per-shard candidate vector per MCMC step over xGMI: timed as `exchange_alt` "
                                                         "(last in the job); `value` is produced by the exchange named in value_produced_by_exchange -- 'host' = "
                                                         "the ranks' 65 int64 sums through pinned host memory of the node (bit-identical sums, nothing added to the "
                                                         "GPU timeline), DESIGN.md section 6"}
        if alt is not None:
            out["exchange_alt"] = alt
        if late is not None:
            out["late_stage"] = late
        if world == 1 and not args.no_cpu_baseline:
            # the engine's deltas of the candidates the numpy re-score is about to price (untimed; K = 1 per pair), then the re-score
            mid_b = smp.modify_gl_cuda_buffer(0)
            smp.gpu_vect_frags.copy_from_gpu()
            gpu_d = {}
            for fA_, fB_ in CPU_BASELINE_PAIRS:
                d_ = smp._candidate_deltas(fA_ % n, [fB_ % n], mid_b)
                for op_ in CPU_BASELINE_OPS:
                    gpu_d[(fA_ % n, fB_ % n, op_)] = float(d_[0, op_])
            out["cpu_baseline"] = cpu_baseline(P, smp.gpu_vect_frags.as_dict(), budget_s=16.0, param=smp._param_flat, gpu_deltas=gpu_d)
            out["cpu_baseline"]["host_cpus"] = os.cpu_count()
            try:
                out["cpu_baseline_dense"] = cpu_baseline_dense()
            except Exception as e:
                out["cpu_baseline_dense"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        phase("headline line printed; exchange_alt next")
        try:
            alt = run_exchange_alt()
            phase("exchange_alt done")
        except Exception as e:
            alt = {"error": repr(e)}
        if rank == 0 and alt is not None:
            out["exchange_alt"] = alt
            print(json.dumps(out), flush=True)     # (the same line again, with the extra: the last line is the complete one)
    smp.free_gpu()
    if world > 1:
        try:
            td.barrier()
            td.destroy_process_group()
        except Exception:
            pass


if __name__ == "__main__":
    main()
