"""Initial fit of the Rippe contact model: ``cuda_lib_gl.sampler.estimate_parameters``
(``cuda_lib_gl.py:1229-1294``) and ``optim_rippe_curve_update.py`` (``estimate_param_rippe`` :73-115,
``estimate_max_dist_intra`` :125-135, ``peval`` :22-28).

The reference builds the contacts-vs-distance histogram with an O(S^2) Python double loop over the DENSE sub-level
matrix.  Here the same histogram comes from the COO list (sum of counts per distance bin) plus the number of cis
sub-fragment pairs per distance bin (so that zero pairs weigh in the mean exactly as they do in the dense loop).
The least-squares / fsolve calls are scipy's MINPACK wrappers as in the reference (scipy version differs from the
reference's pin: parity of the fitted numbers is "unpinned"; fixtures pass param_simu in).
"""
import numpy as np
from scipy.optimize import fsolve, leastsq

D_RIPPE = 3  # optim_rippe_curve_update.py:9


def peval(x, param):
    return param[3] * (0.53 * (param[0] ** -3.) * np.power((param[1] * x / param[0]), (param[2])) *
                       np.exp((D_RIPPE - 2) / ((np.power((param[1] * x / param[0]), 2) + D_RIPPE))))


def log_residuals(p, y, x):
    kuhn, lm, slope, A = p
    rippe = np.log(A) + np.log(0.53) - 3 * np.log(kuhn) + slope * (np.log(lm * x) - np.log(kuhn)) + \
        (D_RIPPE - 2) / ((np.power((lm * x / kuhn), 2) + D_RIPPE))
    return y - rippe


def estimate_param_rippe(y_meas, x_bins):
    kuhn, lm, slope = 1, 9.6, -1.5
    A = np.sum(y_meas)
    p0 = [kuhn, lm, slope, A]
    with np.errstate(all="ignore"):
        plsq = leastsq(log_residuals, p0, args=(np.log(y_meas), x_bins))
    y_estim = peval(x_bins, plsq[0])
    kuhn_x, lm_x, slope_x, A_x = plsq[0]
    plsq_out = [kuhn_x, lm_x, slope_x, D_RIPPE, A_x]
    if np.any(np.isnan(np.array(plsq_out))) or slope >= 0:
        plsq_out = [kuhn, lm, slope, D_RIPPE, A]
    return plsq_out, y_estim


def _residual_4_max_dist(x, p):
    kuhn, lm, slope, d, A, y = p
    rippe = A * (0.53 * (kuhn ** -3.) * np.power((lm * x / kuhn), slope) *
                 np.exp((d - 2) / ((np.power((lm * x / kuhn), 2) + d))))
    return y - rippe


def estimate_max_dist_intra(p, val_inter):
    """``optim_rippe_curve_update.py:117-135``: distance at which the fitted curve meets the trans level, MINPACK
    ``fsolve`` from s0 = 500 (silently returns ~500 when the solver wanders off, SURVEY.md H6 -- kept)."""
    kuhn, lm, slope, d, A = p
    with np.errstate(all="ignore"):
        x = fsolve(_residual_4_max_dist, 500, args=([kuhn, lm, slope, d, A, val_inter]))
    return x[0]


def estimate_max_dist_intra_step(p, val_inter):
    """The same root for the PER-STEP call of step_nuisance_parameters (cuda_lib_gl.py:2053, 2060, 2071): MINPACK's hybrd restated for
    one unknown behind the C ABI (graal_amd/csrc/host_fit.h, graal_host_max_dist_intra) -- 3 us instead of 0.12-0.2 ms of fsolve
    driving a numpy expression on a one-element array.  Same algorithm, same start value, same give-up behaviour; the residual's
    pow / exp are glibc's instead of numpy's, so the float64 root may differ in its last bits (never, on the golden cases and 20,000
    random ones, in the float32 it becomes in param_simu: tests/test_rippe_fit.py).  Argument types the C side does not model exactly
    -- anything but five numpy float32 scalars, or five Python / float64 numbers -- take the scipy path above."""
    vals = list(p) + [val_inter]
    if all(isinstance(v, np.float32) for v in vals):
        f32 = True
    elif all(isinstance(v, (float, int, np.float64)) and not isinstance(v, bool) for v in vals):
        f32 = False
    else:
        return estimate_max_dist_intra(p, val_inter)
    from . import lib
    x, _ = lib.host_max_dist_intra(vals[:5], vals[5], f32)
    return np.float64(x)


def mean_contacts_per_bin(S_o_A_sub_frags, sub_coo, bins, max_dist_kb, size_bin_kb):
    """Mean number of contacts of cis sub-fragment pairs per genomic-distance bin (cuda_lib_gl.py:1236-1270):
    d = ((start_j - start_i - len_i) + (len_i + len_j) / 2) / 1000 for the pair ordered by position; bins of width
    size_bin_kb up to max_dist_kb; empty or all-zero bins -> 1e-10."""
    id_c = np.asarray(S_o_A_sub_frags["id_c"])
    start = np.asarray(S_o_A_sub_frags["start_bp"], dtype=np.float64)
    length = np.asarray(S_o_A_sub_frags["len_bp"], dtype=np.float64)
    pos = np.asarray(S_o_A_sub_frags["pos"])
    n_bins = len(bins)

    def dist(i, j):
        first_i = pos[i] < pos[j]
        a = np.where(first_i, i, j)
        b = np.where(first_i, j, i)
        return ((start[b] - start[a] - length[a]) + (length[a] + length[b]) / 2.) / 1000.

    sums = np.zeros(n_bins)
    row, col, val = sub_coo
    cis = id_c[row] == id_c[col]
    d = dist(row[cis], col[cis])
    ok = d < max_dist_kb
    idx = (d[ok] / size_bin_kb).astype(np.int64)
    # (bincount accumulates in list order like np.add.at: the same float64 sums, ~30x faster -- level 0 holds 1e7 contacts)
    sums += np.bincount(np.clip(idx, 0, n_bins - 1), weights=np.asarray(val, dtype=np.float64)[cis][ok], minlength=n_bins)
    # number of cis pairs per distance bin: per contig, pairs at position offset k while the distance is in range
    counts = np.zeros(n_bins)
    for c in np.unique(id_c):
        m = np.nonzero(id_c == c)[0]
        m = m[np.argsort(pos[m], kind="stable")]
        for k in range(1, len(m)):
            dk = dist(m[:-k], m[k:])
            okk = dk < max_dist_kb
            if not okk.any():
                break
            counts += np.bincount(np.clip((dk[okk] / size_bin_kb).astype(np.int64), 0, n_bins - 1), minlength=n_bins)
    mean = np.full(n_bins, 1e-10, dtype=np.float32)
    good = (counts > 0) & (sums > 0)
    mean[good] = (sums[good] / counts[good]).astype(np.float32)
    return mean
