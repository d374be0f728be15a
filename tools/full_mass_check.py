#!/usr/bin/env python3
"""GPU box: the tiled windowed-mass kernel of the full evaluation (k_full_mass_t: long contigs) against k_full_mass (GRAAL_FULL_MASS_TILED=0 /
=1 in child processes -- the switch is read once per process): the int64 sums of graal_eval_full_q must be EQUAL, on C5's 7 original
contigs and its exploded layout, and on layouts with circular contigs, reversed bins and 1-3 sub-fragments with RF counts 1..9 (both RF-count
indexings); and the time of a full evaluation."""
import json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure():
    import bench
    from graal_amd import synth, dist as gdist
    from graal_amd.lib import Engine
    from tests.test_engine_gpu import random_state_for, relabel_ref
    import torch
    out = {}
    for layout in ("exploded", "original"):
        P = synth.make_problem(n_bins=50000, nnz=20_000_000, n_sub=1, seed=20141217)
        if layout == "exploded":
            P["S_o_A_frags"] = bench.exploded_layout(P)
        rng = np.random.RandomState(3)
        smp = bench.build_sampler(P, rng, gdist.Group(0, 1), 0)
        smp.init_likelihood()
        order = np.arange(50000); rng.shuffle(order)
        for i in order[:(300 if layout == "exploded" else 6)]:
            smp.step_max_likelihood(int(i), 5)
        smp.modify_gl_cuda_buffer(0)
        q = [int(v) for v in smp.engine.eval_full_q()]
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            smp.engine.eval_full_q()
        out["c5 " + layout] = {"q": q, "full_eval_us": (time.perf_counter() - t) / 20 * 1e6}
        smp.free_gpu()
    for n_sub, n_bins, seed in ((3, 1200, 5), (1, 3000, 6), (2, 900, 7)):
        par = synth.make_param_simu(fact=300.0, v_inter=0.03)
        P = synth.make_problem(n_bins=n_bins, nnz=30000, n_sub=n_sub, seed=seed, contig_weights=(6, 3, 1), mean_len_bp=900.0,
                               accu=("random", 1, 9) if n_sub > 1 else 1, param=par)
        rng = np.random.RandomState(seed)
        for trial in range(2):
            s = random_state_for(P, rng, n_contigs=int(rng.randint(2, 5)), p_circ=0.5)
            relabel_ref(s)
            for quirk in (False, True):
                e = Engine(0)
                e.upload_subfrags(P["np_sub_frags_id"], P["np_sub_frags_len_bp"], P["np_sub_frags_accu"], P["init_n_sub_frags"], P["mean_squared_frags_per_bin"])
                e.upload_contacts(P["coo_row"], P["coo_col"], P["coo_val"])
                e.set_params(P["param_simu"])
                e.upload_frags(s)
                e.set_mode(ref_trans_accu=quirk)
                e.relabel_contigs()
                out["%d sub %d bins layout %d quirk %d" % (n_sub, n_bins, trial, quirk)] = {"q": [int(v) for v in e.eval_full_q()], "full_eval_us": 0.0}
                e.close()
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        print("RESULT " + json.dumps(measure()))
        sys.exit(0)
    res = {}
    for name, env in (("default", {}), ("untiled", {"GRAAL_FULL_MASS_TILED": "0"}), ("tiled", {"GRAAL_FULL_MASS_TILED": "1"})):
        o = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert o.returncode == 0, o.stderr[-3000:]
        res[name] = json.loads([l for l in o.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
        print(name, {k: round(v["full_eval_us"], 1) for k, v in res[name].items() if v["full_eval_us"]}, flush=True)
    for key in res["default"]:
        assert res["default"][key]["q"] == res["untiled"][key]["q"] == res["tiled"][key]["q"], (key, [res[n][key]["q"] for n in res])
    print("bit-identical sums on %d layouts" % len(res["default"]))
