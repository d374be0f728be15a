// host_fit.h -- host-only: the scalar root solve of the nuisance-parameter step behind the C ABI (included by graal_hip.hip inside extern "C").
//
// cuda_lib_gl.sampler.step_nuisance_parameters (cuda_lib_gl.py:2022-2107) calls opti.estimate_max_dist_intra
// (optim_rippe_curve_update.py:117-135) after three of its four perturbations: the distance at which the Rippe curve meets the trans level,
//     scipy.optimize.fsolve(residual_4_max_dist, 500, args=(kuhn, lm, slope, d, A, y)),
//     residual(x) = y - A * (0.53 * kuhn**-3. * power(lm * x / kuhn, slope) * exp((d - 2) / (power(lm * x / kuhn, 2) + d)))
// i.e. MINPACK's hybrd (Powell's hybrid method: forward-difference Jacobian, dogleg step, rank-one updates) on ONE unknown, driven from
// Python: 16 evaluations of a numpy expression on a one-element array, 0.12 ms per MCMC step with "sample parameters" on (the reference
// GUI's default, main_window.py:544) -- as much as the step's GPU work.  Here: hybrd restated for n = 1, statement by statement (every
// vector loop has one element, every norm is an absolute value, the QR factorisation of a 1 x 1 matrix is its sign; the order of the
// floating-point operations is MINPACK's), with the residual in C.  Settings as scipy's fsolve passes them: xtol = 1.49012e-8,
// maxfev = 200 (n + 1) = 400, epsfcn = machine epsilon, factor = 100, mode 1 (internal scaling); the machine epsilon is DBL_EPSILON
// (what this image's scipy 1.15.3 uses: probed through the finite-difference step of its second evaluation, tools/hybrd_probe.py; the
// Fortran original's dpmpar(1) = 2.22044604926e-16 differs from it in the 12th digit).
//
// What is NOT the same as the Python path: the residual's pow / exp are glibc's here and numpy's (SVML) there, which differ in the last
// place on ~5 % of the arguments, so x may differ in its last bits (tests/test_rippe_fit.py: <= 1e-13 relative and the SAME float32 -- the
// precision in which d_max enters param_simu -- on every case of tests/golden/rippe_fit.json, the returns-500 case included, and on random
// parameter sets against scipy).  graal_amd/rippe_fit.py keeps the scipy path for the initial fit (pinned by the reference's own module);
// the per-step call goes through here.

#include <cfloat>

namespace {

struct MaxDistFn {
    double kuhn, lm, slope, d, A, y, c0, dm2;
    double operator()(double x) const
    {
        const double t = lm * x / kuhn;                                   // (lm * x / kuhn), twice in the expression: the same value
        const double r = A * (c0 * pow(t, slope) * exp(dm2 / (pow(t, 2.0) + d)));
        return y - r;
    }
};

// MINPACK hybrd, n = 1.  Returns info (1 = converged, 2 = maxfev, 3 = xtol too small, 4 / 5 = not making progress); *x in / out.
static int hybrd_scalar(const MaxDistFn& fcn, double* x_io, int* nfev_out)
{
    const double xtol = 1.49012e-08, factor = 100.0, epsmch = DBL_EPSILON, epsfcn = DBL_EPSILON;
    const int maxfev = 400;
    const double p1 = 0.1, p5 = 0.5, p001 = 1.0e-3, p0001 = 1.0e-4;
    double x = *x_io;
    int info = 0, nfev = 0;
    // evaluate the function at the starting point and calculate its norm
    double fvec = fcn(x);
    nfev = 1;
    double fnorm = fabs(fvec);
    int iter = 1, ncsuc = 0, ncfail = 0, nslow1 = 0, nslow2 = 0;
    double diag = 0.0, delta = 0.0, xnorm = 0.0;
    double q = 1.0, r = 0.0, qtf = 0.0;       // the orthogonal factor (1 x 1: +-1), the triangular factor, (q transpose) * fvec
    for (;;) {                                // ---- outer loop
        bool jeval = true;
        // fdjac1: forward-difference approximation
        {
            const double eps = sqrt(fmax(epsfcn, epsmch));
            const double temp = x;
            double h = eps * fabs(temp);
            if (h == 0.0) h = eps;
            const double wa1 = fcn(temp + h);
            r = (wa1 - fvec) / h;             // fjac(1,1)
        }
        nfev += 1;
        // qrfac of the 1 x 1 matrix: acnorm = |J|; the Householder vector is J / ajnorm + 1 = 2 (J != 0), rdiag = -ajnorm = -J
        const double acnorm = fabs(r);
        double fjac = r, rdiag;
        {
            double ajnorm = fabs(fjac);
            if (ajnorm == 0.0) rdiag = -ajnorm;
            else {
                if (fjac < 0.0) ajnorm = -ajnorm;
                fjac = fjac / ajnorm;
                fjac = fjac + 1.0;
                rdiag = -ajnorm;
            }
        }
        // on the first iteration, scale according to the norm of the column of the initial jacobian
        if (iter == 1) {
            diag = acnorm;
            if (acnorm == 0.0) diag = 1.0;
            const double wa3 = diag * x;
            xnorm = fabs(wa3);
            delta = factor * xnorm;
            if (delta == 0.0) delta = factor;
        }
        // form (q transpose) * fvec and store in qtf
        qtf = fvec;
        if (fjac != 0.0) {
            const double sum = fjac * qtf;
            const double temp = -sum / fjac;
            qtf = qtf + fjac * temp;
        }
        // copy the triangular factor of the qr factorization into r
        r = rdiag;
        // accumulate the orthogonal factor in fjac (qform)
        {
            const double wa = fjac;
            q = 1.0;
            if (wa != 0.0) {
                const double sum = q * wa;
                const double temp = sum / wa;
                q = q - temp * wa;
            }
        }
        // rescale if necessary
        diag = fmax(diag, acnorm);
        for (;;) {                            // ---- inner loop
            // dogleg: determine the direction p
            double px;
            {
                double temp = r;
                if (temp == 0.0) {
                    temp = fmax(temp, fabs(r));
                    temp = epsmch * temp;
                    if (temp == 0.0) temp = epsmch;
                }
                px = (qtf - 0.0) / temp;                      // the gauss-newton direction
                double wa1 = 0.0;
                double wa2 = diag * px;
                const double qnorm = fabs(wa2);
                if (!(qnorm <= delta)) {
                    // the gauss-newton direction is not acceptable: the scaled gradient direction
                    wa1 = wa1 + r * qtf;
                    wa1 = wa1 / diag;
                    const double gnorm = fabs(wa1);
                    double sgnorm = 0.0;
                    double alpha = delta / qnorm;
                    if (gnorm != 0.0) {
                        wa1 = (wa1 / gnorm) / diag;
                        wa2 = r * wa1;
                        double t2 = fabs(wa2);
                        sgnorm = (gnorm / t2) / t2;
                        alpha = 0.0;
                        if (!(sgnorm >= delta)) {
                            // the scaled gradient direction is not acceptable either: the point along the dogleg
                            const double bnorm = fabs(qtf);
                            double t3 = (bnorm / gnorm) * (bnorm / qnorm) * (sgnorm / delta);
                            const double dq = delta / qnorm, sd = sgnorm / delta;
                            t3 = t3 - dq * (sd * sd) + sqrt((t3 - dq) * (t3 - dq) + (1.0 - dq * dq) * (1.0 - sd * sd));
                            alpha = (dq * (1.0 - sd * sd)) / t3;
                        }
                    }
                    const double t4 = (1.0 - alpha) * fmin(sgnorm, delta);
                    px = t4 * wa1 + alpha * px;
                }
            }
            // store the direction p and x + p; calculate the norm of p
            const double wa1 = -px;
            const double wa2 = x + wa1;
            double wa3 = diag * wa1;
            const double pnorm = fabs(wa3);
            // on the first iteration, adjust the initial step bound
            if (iter == 1) delta = fmin(delta, pnorm);
            // evaluate the function at x + p and calculate its norm
            const double wa4 = fcn(wa2);
            nfev += 1;
            const double fnorm1 = fabs(wa4);
            // compute the scaled actual reduction
            double actred = -1.0;
            if (fnorm1 < fnorm) { const double t = fnorm1 / fnorm; actred = 1.0 - t * t; }
            // compute the scaled predicted reduction
            wa3 = qtf + r * wa1;
            const double temp = fabs(wa3);
            double prered = 0.0;
            if (temp < fnorm) { const double t = temp / fnorm; prered = 1.0 - t * t; }
            // compute the ratio of the actual to the predicted reduction
            double ratio = 0.0;
            if (prered > 0.0) ratio = actred / prered;
            // update the step bound
            if (!(ratio >= p1)) {
                ncsuc = 0;
                ncfail += 1;
                delta = p5 * delta;
            } else {
                ncfail = 0;
                ncsuc += 1;
                if (ratio >= p5 || ncsuc > 1) delta = fmax(delta, pnorm / p5);
                if (fabs(ratio - 1.0) <= p1) delta = pnorm / p5;
            }
            // test for successful iteration
            if (!(ratio < p0001)) {
                x = wa2;
                xnorm = fabs(diag * x);
                fvec = wa4;
                fnorm = fnorm1;
                iter += 1;
            }
            // determine the progress of the iteration
            nslow1 += 1;
            if (actred >= p001) nslow1 = 0;
            if (jeval) nslow2 += 1;
            if (actred >= p1) nslow2 = 0;
            // test for convergence
            if (delta <= xtol * xnorm || fnorm == 0.0) info = 1;
            if (info != 0) goto done;
            // tests for termination and stringent tolerances
            if (nfev >= maxfev) info = 2;
            if (p1 * fmax(p1 * delta, pnorm) <= epsmch * xnorm) info = 3;
            if (nslow2 == 5) info = 4;
            if (nslow1 == 10) info = 5;
            if (info != 0) goto done;
            // criterion for recalculating the jacobian approximation by forward differences
            if (ncfail == 2) break;
            // the rank one modification to the jacobian; update qtf if necessary
            {
                const double sum = q * wa4;
                const double v = (sum - wa3) / pnorm;            // wa2
                const double u = diag * ((diag * wa1) / pnorm);  // wa1
                if (ratio >= p0001) qtf = sum;
                // r1updt of the 1 x 1 factor: r = r + v * u; r1mpyq: nothing to rotate
                r = r + v * u;
            }
            jeval = false;
        }
    }
done:
    *x_io = x;
    if (nfev_out) *nfev_out = nfev;
    return info;
}

} // namespace

// estimate_max_dist_intra (optim_rippe_curve_update.py:117-135).  p5 = kuhn, lm, slope, d, A; f32 != 0: the parameters are numpy float32
// scalars as step_nuisance_parameters passes them (cuda_lib_gl.py:2053) -- then `0.53 * kuhn ** -3.` and `d - 2` are float32 operations in
// the reference's expression (the other operands meet the float64 array x and are promoted exactly).  *info = MINPACK's code; no device.
int graal_host_max_dist_intra(const double* p5, double val_inter, int32_t f32, double* x_out, int32_t* info_out)
{
    if (!p5 || !x_out) return GRAAL_E_ARG;
    MaxDistFn f;
    f.kuhn = p5[0]; f.lm = p5[1]; f.slope = p5[2]; f.d = p5[3]; f.A = p5[4]; f.y = val_inter;
    if (f32) {
        f.c0 = (double)(0.53f * powf((float)p5[0], -3.0f));
        f.dm2 = (double)((float)p5[3] - 2.0f);
    } else {
        f.c0 = 0.53 * pow(p5[0], -3.0);
        f.dm2 = p5[3] - 2.0;
    }
    double x = 500.0;
    int nfev = 0;
    const int info = hybrd_scalar(f, &x, &nfev);
    *x_out = x;
    if (info_out) *info_out = info;
    return GRAAL_OK;
}
