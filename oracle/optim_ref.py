"""TEST INFRASTRUCTURE -- restatement of the two functions of optim_rippe_curve_update.py that the nuisance-parameter
step calls (``peval`` :22-28, ``estimate_max_dist_intra`` :117-135), Python 3, same scipy MINPACK wrapper (fsolve)."""
import numpy as np
from scipy.optimize import fsolve

d = 3  # optim_rippe_curve_update.py:9 (module-level constant, NOT the parameter of the same name)


def peval(x, param):
    rippe = param[3] * (0.53 * (param[0] ** -3.) * np.power((param[1] * x / param[0]), (param[2])) *
                        np.exp((d - 2) / ((np.power((param[1] * x / param[0]), 2) + d))))
    return rippe


def residual_4_max_dist(x, p):
    kuhn, lm, slope, d_, A, y = p
    rippe = A * (0.53 * (kuhn ** -3.) * np.power((lm * x / kuhn), slope) *
                 np.exp((d_ - 2) / ((np.power((lm * x / kuhn), 2) + d_))))
    return y - rippe


def estimate_max_dist_intra(p, val_inter):
    s0 = 500
    kuhn, lm, slope, d_, A = p
    p0 = [kuhn, lm, slope, d_, A, val_inter]
    with np.errstate(all="ignore"):
        x = fsolve(residual_4_max_dist, s0, args=(p0))
    return x[0]
