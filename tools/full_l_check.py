#!/usr/bin/env python3
"""GPU box: k_full_nnz_l (labels in LDS) against k_full_nnz_u (GRAAL_FULL_NO_LDS=1 in a child process): the two int64 sums of the
full evaluation on the C5 map, exploded layout after some MCMC steps and on its 7 original contigs, and the time of a full evaluation."""
import json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure():
    import bench
    from graal_amd import synth, dist as gdist
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
        for i in order[:(600 if layout == "exploded" else 10)]:
            smp.step_max_likelihood(int(i), 5)
        smp.modify_gl_cuda_buffer(0)
        q = [int(v) for v in smp.engine.eval_full_q()]
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(30):
            smp.engine.eval_full_q()
        out[layout] = {"q": q, "full_eval_us": (time.perf_counter() - t) / 30 * 1e6}
        smp.free_gpu()
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        print("RESULT " + json.dumps(measure()))
        sys.exit(0)
    res = {}
    for name, env in (("lds", {}), ("lds_g4", {"GRAAL_FULL_G": "4"}), ("no_lds", {"GRAAL_FULL_NO_LDS": "1"})):
        o = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env), capture_output=True, text=True, timeout=500)
        assert o.returncode == 0, o.stderr[-3000:]
        res[name] = json.loads([l for l in o.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
        print(name, {k: round(v["full_eval_us"], 1) for k, v in res[name].items()}, flush=True)
    for layout in ("exploded", "original"):
        assert res["lds"][layout]["q"] == res["no_lds"][layout]["q"] == res["lds_g4"][layout]["q"], (layout, res)
    print("bit-identical sums:", {k: v["q"] for k, v in res["lds"].items()})
