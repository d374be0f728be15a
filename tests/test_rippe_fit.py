"""CPU: the COO-based contacts-vs-distance histogram equals the reference's dense O(S^2) double loop
(cuda_lib_gl.py:1236-1270), and the fit recovers the generating parameters' order of magnitude."""
import numpy as np

from graal_amd import rippe_fit, synth


def dense_reference_histogram(S, hic, bins, max_dist_kb, size_bin_kb):
    collect = {k: [] for k in range(len(bins))}
    n = len(S["id_c"])
    for i in range(n - 1):
        for j in range(i + 1, n):
            if S["id_c"][i] == S["id_c"][j]:
                if S["pos"][i] < S["pos"][j]:
                    d = ((S["start_bp"][j] - S["start_bp"][i] - S["len_bp"][i]) + (S["len_bp"][i] + S["len_bp"][j]) / 2.) / 1000.
                else:
                    d = ((S["start_bp"][i] - S["start_bp"][j] - S["len_bp"][j]) + (S["len_bp"][j] + S["len_bp"][i]) / 2.) / 1000.
                if d < max_dist_kb:
                    collect[int(d / size_bin_kb)].append(hic[i, j])
    out = np.zeros(len(bins), np.float32)
    for k in range(len(bins)):
        tmp = np.mean(collect[k]) if collect[k] else np.nan
        out[k] = 1e-10 if (np.isnan(tmp) or tmp == 0) else tmp
    return out


def sub_level_soa(P):
    """Sub-level fragment list (one entry per sub-fragment) of a synthetic problem."""
    S = P["init_n_sub_frags"]
    b = P["bin_of_sub"]
    id_c = P["S_o_A_frags"]["id_c"][b]
    len_bp = P["sub_len_bp"].astype(np.int64)
    pos = np.zeros(S, np.int64)
    start = np.zeros(S, np.int64)
    for c in np.unique(id_c):
        m = np.nonzero(id_c == c)[0]
        pos[m] = np.arange(len(m))
        start[m] = np.cumsum(len_bp[m]) - len_bp[m]
    return dict(id_c=id_c, pos=pos, start_bp=start, len_bp=len_bp)


def test_histogram_matches_dense_double_loop():
    par = synth.make_param_simu(fact=300.0, v_inter=0.03)
    P = synth.with_dense(synth.make_problem(n_bins=40, nnz=700, n_sub=3, seed=3, contig_weights=(5, 3), mean_len_bp=1500.0,
                                            accu=9, param=par))
    S = sub_level_soa(P)
    size_bin_kb, max_dist_kb = 2.0, 40.0
    bins = np.arange(size_bin_kb, max_dist_kb + size_bin_kb, size_bin_kb)
    want = dense_reference_histogram(S, P["hic_matrix"], bins, max_dist_kb, size_bin_kb)
    got = rippe_fit.mean_contacts_per_bin(S, (P["coo_row"], P["coo_col"], P["coo_val"]), bins, max_dist_kb, size_bin_kb)
    assert np.allclose(got, want, rtol=1e-6)


def test_fit_runs_and_is_sane():
    x = np.arange(2.0, 200.0, 2.0)
    truth = [1.0, 9.6, -1.5, 3, 5000.0]
    y = rippe_fit.peval(x, [truth[0], truth[1], truth[2], truth[4]])
    p, y_est = rippe_fit.estimate_param_rippe(y, x)
    assert np.allclose(y_est, y, rtol=1e-3)
    d_max = rippe_fit.estimate_max_dist_intra(p, 0.05)
    assert rippe_fit.peval(d_max, [p[0], p[1], p[2], p[4]]) == np.float64(0.05) or abs(
        rippe_fit.peval(d_max, [p[0], p[1], p[2], p[4]]) - 0.05) < 1e-6 or d_max == 500


# ---- pinned by the reference itself: tests/golden/rippe_fit.json holds inputs and outputs of the REFERENCE's
# optim_rippe_curve_update.py (peval, estimate_max_dist_intra, estimate_param_rippe), generated in the build container by
# tests/golden/make_rippe_fixtures.py from a lib2to3 copy of /root/reference/optim_rippe_curve_update.py.
def _golden():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rippe_fit.json")) as f:
        return json.load(f)


def _same_stack(fx):
    import scipy
    return fx["numpy"] == np.__version__ and fx["scipy"] == scipy.__version__


def _close(got, want, fx):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    if _same_stack(fx):      # same numpy / scipy (MINPACK) as the generating run: the very same numbers
        return np.array_equal(got, want, equal_nan=True)
    return np.allclose(got, want, rtol=1e-6, atol=0, equal_nan=True)


def test_peval_matches_the_reference_module():
    from oracle import optim_ref
    fx = _golden()
    assert fx["d_module_constant"] == rippe_fit.D_RIPPE == optim_ref.d
    for c in fx["peval"]:
        x = np.asarray(c["x"], dtype=np.float64) if isinstance(c["x"], list) else np.float32(c["x"])
        for mod in (rippe_fit, optim_ref):
            assert _close(mod.peval(x, c["param"]), c["y"], fx), c["param"]


def test_estimate_max_dist_intra_matches_the_reference_module():
    from oracle import optim_ref
    fx = _golden()
    xs = [c["x"] for c in fx["estimate_max_dist_intra"]]
    assert any(abs(x - 500.0) < 1e-6 for x in xs)          # MINPACK gives the start value back: kept as is
    assert any(0 < x < 400 for x in xs)
    for c in fx["estimate_max_dist_intra"]:
        p, v = c["p"], c["val_inter"]
        if c.get("float32_inputs"):
            p, v = [np.float32(a) for a in p], np.float32(v)
        for mod in (rippe_fit, optim_ref):
            assert _close(mod.estimate_max_dist_intra(p, v), c["x"], fx), c


def test_estimate_param_rippe_matches_the_reference_module():
    fx = _golden()
    kinds = set()
    for c in fx["estimate_param_rippe"]:
        y = np.asarray(c["y_meas"], dtype=np.float32 if c["y_is_float32"] else np.float64)
        p, y_est = rippe_fit.estimate_param_rippe(y, np.asarray(c["x_bins"]))
        assert _close(p, c["p"], fx), c["kind"]
        assert _close(y_est, c["y_estim"], fx), c["kind"]
        kinds.add(c["kind"])
    assert len(kinds) == 3


def test_c_restatement_of_the_scalar_root_solve_matches_the_reference_module():
    """graal_host_max_dist_intra (graal_amd/csrc/host_fit.h: MINPACK's hybrd restated for one unknown, the residual in C) -- what
    step_nuisance_parameters calls per step instead of scipy's fsolve -- against the outputs of the REFERENCE's
    optim_rippe_curve_update.estimate_max_dist_intra (tests/golden/rippe_fit.json), the returns-500 case included: the float64 root to
    1e-13 relative (the residual's pow / exp are glibc's here and numpy's there: last-place differences) and the SAME float32, the
    precision in which d_max enters param_simu."""
    fx = _golden()
    n500 = 0
    for c in fx["estimate_max_dist_intra"]:
        p, v = c["p"], c["val_inter"]
        if c.get("float32_inputs"):
            p, v = [np.float32(a) for a in p], np.float32(v)
        else:
            p, v = [float(a) for a in p], float(v)
        x = rippe_fit.estimate_max_dist_intra_step(p, v)
        assert np.float32(x) == np.float32(c["x"]), c
        assert abs(x - c["x"]) <= 1e-13 * abs(c["x"]), c
        n500 += abs(c["x"] - 500.0) < 1e-6
    assert n500 >= 1


def test_c_restatement_of_the_scalar_root_solve_matches_scipy_on_random_parameters():
    """... and against scipy's fsolve on 1,500 random parameter sets in the ranges the nuisance-parameter walk visits (float32 scalars as
    the step passes them, and plain floats): same float32, <= 1e-13 relative, the same give-ups (x = 500 exactly)."""
    import warnings
    rng = np.random.RandomState(7)
    n500 = n_conv = 0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i in range(1500):
            kuhn = np.float32(rng.choice([1.0, rng.uniform(0.5, 2.0)]))
            lm, slope = np.float32(rng.uniform(5, 15)), np.float32(rng.uniform(-2.2, -0.8))
            d = np.float32(3.0 if rng.rand() < 0.7 else rng.uniform(2.2, 5))
            A, v = np.float32(10 ** rng.uniform(0.5, 5)), np.float32(10 ** rng.uniform(-4, 0.5))
            p = [kuhn, lm, slope, d, A]
            if i % 3 == 0:
                p, v = [float(a) for a in p], float(v)
            want, got = rippe_fit.estimate_max_dist_intra(p, v), rippe_fit.estimate_max_dist_intra_step(p, v)
            assert np.float32(got) == np.float32(want), (p, v, want, got)
            assert abs(got - want) <= 1e-13 * abs(want), (p, v, want, got)
            n500 += want == 500.0
            n_conv += want != 500.0
    assert n500 > 100 and n_conv > 100
    # argument types the C side does not model exactly take the scipy path (same numbers as estimate_max_dist_intra)
    mixed = [np.float32(1.0), 9.6, np.float32(-1.5), 3, np.float32(5000.0)]
    assert rippe_fit.estimate_max_dist_intra_step(mixed, np.float32(0.01)) == rippe_fit.estimate_max_dist_intra(mixed, np.float32(0.01))
