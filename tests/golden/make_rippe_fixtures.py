#!/usr/bin/env python3
"""Generate tests/golden/rippe_fit.json from the REFERENCE's own optim_rippe_curve_update.py.

Build container only: a temporary copy of /root/reference/optim_rippe_curve_update.py (+ leastsqbound.py, which it star-imports)
is converted from Python 2 with lib2to3, imported from the temporary directory and run; only inputs and outputs are written.
The reference's source never enters the repository or the GPU box -- the JSON (numbers) does.

    python tests/golden/make_rippe_fixtures.py

Recorded: numpy / scipy versions of the generating run (the reference pins scipy 1.0.0 / numpy 1.13.3; MINPACK's leastsq and
fsolve are what the fitted numbers depend on).
"""
import importlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rippe_fit.json")


def load_reference_module():
    tmp = tempfile.mkdtemp(prefix="graal_ref_fit_")
    for name in ("optim_rippe_curve_update.py", "leastsqbound.py"):
        shutil.copy(os.path.join(REF, name), os.path.join(tmp, name))
    subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n", tmp], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sys.path.insert(0, tmp)
    try:
        return importlib.import_module("optim_rippe_curve_update"), tmp
    finally:
        sys.path.remove(tmp)


def main():
    import scipy
    import warnings
    warnings.simplefilter("ignore")
    ref, tmp = load_reference_module()
    rng = np.random.RandomState(20141217)
    fx = {"generated_by": "tests/golden/make_rippe_fixtures.py from /root/reference/optim_rippe_curve_update.py (lib2to3 copy)",
          "numpy": np.__version__, "scipy": scipy.__version__, "d_module_constant": int(ref.d),
          "peval": [], "estimate_max_dist_intra": [], "estimate_param_rippe": []}
    # ---- peval (optim_rippe_curve_update.py:22-28): 4-lists (the fit) and the 5-lists the nuisance step passes
    # (cuda_lib_gl.py:1982-1984, 2062: [kuhn, lm, slope, d, fact] -> param[3] = d acts as the amplitude)
    xs = [np.arange(2.0, 60.0, 2.0), np.array([0.5, 1.0, 7.25, 133.0, 2500.0]), np.array([417.3])]
    for i in range(24):
        kuhn = float(rng.choice([1.0, 1.0, 0.7, 2.5]))
        lm = float(rng.uniform(5.0, 15.0))
        slope = float(rng.uniform(-2.2, -0.6))
        amp = float(10 ** rng.uniform(0, 5))
        par = [kuhn, lm, slope, amp] if i % 2 == 0 else [kuhn, lm, slope, 3.0, amp]
        x = xs[i % len(xs)]
        y = ref.peval(x, par)
        fx["peval"].append({"x": [float(v) for v in x], "param": par, "y": [float(v) for v in np.atleast_1d(y)]})
    # scalar x (the nuisance step calls peval(new_d_max, ...) with a numpy float32 scalar, cuda_lib_gl.py:2062)
    for xv in (31.66, 500.0):
        par = [1.0, 9.6, -1.5, 3.0, 200.0]
        fx["peval"].append({"x": float(np.float32(xv)), "param": par, "y": float(ref.peval(np.float32(xv), par))})
    # ---- estimate_max_dist_intra (:117-135): fsolve from s0 = 500, including parameter sets where MINPACK wanders off and
    # the start value (or something near it) comes back
    cases = [([1.0, 9.6, -1.5, 3, 200.0], 0.02), ([1.0, 9.6, -1.5, 3, 1.0e4], 1.0e-3), ([1.0, 9.6, -1.5, 3, 300.0], 0.03),
             ([1.0, 9.6, -1.5, 3, 50.0], 0.05), ([1.0, 9.6, -1.5, 3, 1.0], 5.0), ([1.0, 9.6, -1.5, 3, 1.0e-3], 10.0),
             ([1.0, 9.6, -0.2, 3, 10.0], 1.0e-6), ([1.0, 9.6, -1.5, 3, 200.0], -1.0)]
    for _ in range(20):
        cases.append(([float(rng.choice([1.0, 0.8, 1.7])), float(rng.uniform(6, 14)), float(rng.uniform(-2.0, -0.8)), 3,
                       float(10 ** rng.uniform(0, 4.5))], float(10 ** rng.uniform(-4, 0))))
    for p, v in cases:
        x = ref.estimate_max_dist_intra(p, v)
        fx["estimate_max_dist_intra"].append({"p": p, "val_inter": v, "x": float(x)})
    # float32 inputs as the sampler passes them (param_simu fields are numpy float32 scalars)
    p32 = [np.float32(1.0), np.float32(9.6), np.float32(-1.5), np.float32(3.0), np.float32(200.0)]
    fx["estimate_max_dist_intra"].append({"p": [float(v) for v in p32], "val_inter": float(np.float32(0.02)), "float32_inputs": True,
                                          "x": float(ref.estimate_max_dist_intra(p32, np.float32(0.02)))})
    # ---- estimate_param_rippe (:73-115): exact curves, noisy curves, float32 histograms with the 1e-10 floor
    x_bins = np.arange(2.0, 200.0, 2.0)
    for i in range(22):
        truth = [1.0, float(rng.uniform(7, 12)), float(rng.uniform(-1.9, -1.1)), float(10 ** rng.uniform(1, 4))]
        y = ref.peval(x_bins, truth)
        kind = "exact"
        if i % 3 == 1:
            y = y * np.exp(rng.normal(0.0, 0.15, size=len(y)))
            kind = "lognormal noise"
        elif i % 3 == 2:
            y = np.float32(y * np.exp(rng.normal(0.0, 0.1, size=len(y))))
            y[rng.choice(len(y), 5, replace=False)] = np.float32(1e-10)   # empty distance bins (cuda_lib_gl.py:1266-1270)
            kind = "float32, noise, 1e-10 floor"
        p, y_est = ref.estimate_param_rippe(y, x_bins)
        fx["estimate_param_rippe"].append({"kind": kind, "x_bins": [float(v) for v in x_bins], "y_meas": [float(v) for v in y],
                                           "y_is_float32": bool(np.asarray(y).dtype == np.float32),
                                           "p": [float(v) for v in p], "y_estim": [float(v) for v in y_est]})
    with open(OUT, "w") as f:
        json.dump(fx, f)
    shutil.rmtree(tmp, ignore_errors=True)
    print("wrote", OUT, {k: len(v) for k, v in fx.items() if isinstance(v, list)})


if __name__ == "__main__":
    main()
