// Elementary functions of the contact model, self-contained (no ROCm-internal symbols).
//
// The reference prices a contact with float32 powf / expf (kernels3.cu:120-166) and takes the float64 log of the result
// (kernels3.cu:191-210).  Every libm rounds those differently in the last place: CUDA's, glibc's (the oracle's) and the ROCm device
// library's powf disagree on a quarter of all arguments.  The only implementation-independent choice is the CORRECTLY ROUNDED
// value, so that is what these functions return: x^y and e^t are computed in double precision (relative error ~1e-14, so the
// final rounding to float32 is the correct one except within ~1e-7 ulp of a tie) -- which on this chip is also CHEAPER than the
// float-float arithmetic of the device library's powf: v_fma_f64 issues at the rate of v_fma_f32, and the double-precision
// version needs half the instructions (tools/powf_dp_micro.hip: 1.50 ps vs 3.19 ps per evaluation, whole chip; agreement with
// glibc's powf 99.94 % -- glibc's own misroundings -- against 75 % for the device library's).
// tools/model_math_check.hip checks them over every float32 in the model's range (tests/test_engine_gpu.py runs it).
#pragma once
#include <hip/hip_runtime.h>

// ln x for a finite NORMAL double x > 0 (every positive float32, subnormals included, converts to one):
// x = m 2^e with m in [sqrt(1/2), sqrt(2)), ln m = 2 atanh(s), s = (m - 1) / (m + 1), |s| <= 0.1716.  Accurate to ~4 ulp of DOUBLE (not
// correctly rounded in double: the reference takes the float64 log of a float32 expected value, kernels3.cu:191-210, and this agrees with
// any libm's to a few 1e-16 relative -- far inside what a float64 sum over contacts resolves).
__device__ __forceinline__ double mm_ln_pos(double x)
{
    const long long bits = __double_as_longlong(x);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    double m = __longlong_as_double((bits & 0x000fffffffffffffLL) | 0x3ff0000000000000LL);
    if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
    const double num = m - 1.0, den = m + 1.0;        // (both exact)
    double r = (double)__builtin_amdgcn_rcpf((float)den);
    r = fma(fma(-den, r, 1.0), r, r);                 // 1 / den to ~2^-45
    double s = num * r;
    s = fma(fma(-den, s, num), r, s);                 // the quotient's residual, corrected: s to ~1 ulp
    const double z = s * s;                           // <= 0.02944: the series' next term, z^9 / 19, is below 1e-15
    double p = 1.0 / 17.0;
    p = fma(p, z, 1.0 / 15.0);
    p = fma(p, z, 1.0 / 13.0);
    p = fma(p, z, 1.0 / 11.0);
    p = fma(p, z, 1.0 / 9.0);
    p = fma(p, z, 1.0 / 7.0);
    p = fma(p, z, 1.0 / 5.0);
    p = fma(p, z, 1.0 / 3.0);
    const double lnm = fma(s * z, 2.0 * p, 2.0 * s);
    return fma((double)e, 0.6931471805599453094, lnm);
}

// e^t for |t| <= 800 (the callers clamp to that): t = n ln 2 + r, |r| <= 0.3466, Taylor polynomial of degree 11 (next term below 2e-16),
// scaled by 2^n.  Beyond |t| ~ 709 the double result saturates through ldexp -- +inf, or 0 / a subnormal -- which is what the callers want:
// they round to float32, where e^800 is +inf and e^-800 is 0 anyway.
__device__ __forceinline__ double mm_exp(double t)
{
    const double n = rint(t * 1.4426950408889634074);
    double r = fma(-n, 6.93147180369123816490e-01, t);    // ln 2 in two parts: n * high part is exact
    r = fma(-n, 1.90821492927058770002e-10, r);
    double q = 1.0 / 39916800.0;
    q = fma(q, r, 1.0 / 3628800.0);
    q = fma(q, r, 1.0 / 362880.0);
    q = fma(q, r, 1.0 / 40320.0);
    q = fma(q, r, 1.0 / 5040.0);
    q = fma(q, r, 1.0 / 720.0);
    q = fma(q, r, 1.0 / 120.0);
    q = fma(q, r, 1.0 / 24.0);
    q = fma(q, r, 1.0 / 6.0);
    q = fma(q, r, 0.5);
    q = fma(q, r, 1.0);
    q = fma(q, r, 1.0);
    return ldexp(q, (int)n);
}

#ifdef GRAAL_EXP_OCML_MATH   // (A/B build, tools/ab.sh: the device library's float-float powf internals, expf and log -- what the engine called until round 3)
typedef float mm_v2f __attribute__((ext_vector_type(2)));
extern "C" __device__ mm_v2f __ocmlpriv_epln_f32(float);
extern "C" __device__ float __ocmlpriv_expep_f32(mm_v2f);
__device__ __forceinline__ float mm_powf(float x, float y)
{
    const mm_v2f ln = __ocmlpriv_epln_f32(x);
    const float yh = y * ln.y;
    const float err = fmaf(y, ln.y, -yh);
    const float t = fmaf(y, ln.x, err);
    const float hi = yh + t;
    const float lo = t - (hi - yh);
    mm_v2f a; a.x = lo; a.y = hi;
    return __ocmlpriv_expep_f32(a);
}
__device__ __forceinline__ float mm_powf_pos(float x, float y) { return mm_powf(x, y); }
__device__ __forceinline__ float mm_expf(float t) { return expf(t); }
__device__ __forceinline__ double mm_ln(float x) { return log((double)x); }
#else
// The special cases are resolved by selects behind the main computation -- no second implementation is inlined next to it (the
// kernels that price contacts hold this code a dozen times over; with the device library's powf / log as fallbacks they grew
// past the instruction cache and ran 1.5x slower than before).

// powf(x, y), correctly rounded, for finite x > 0 and y (no special cases: a distance inside the model's window)
__device__ __forceinline__ float mm_powf_pos(float x, float y)
{
    const double t = (double)y * mm_ln_pos((double)x);
    return (float)mm_exp(fmin(fmax(t, -800.0), 800.0));          // (e^800 rounds to +inf in float32, e^-800 to 0)
}

// powf(x, y), correctly rounded, for x > 0.  x = 0 and x = +inf give powf's limits; a negative base or a NaN gives NaN (powf
// allows negative bases with integer exponents: the model has none -- distances, the Kuhn length).  Overflow -> +inf, underflow -> 0.
// One deviation from C's powf: powf(1, NaN) is 1 there and NaN here (y != y wins) -- the model's exponents are parameters that
// graal_set_params has checked to be finite, so it cannot occur.
__device__ __forceinline__ float mm_powf(float x, float y)
{
    const bool ok = x > 0.0f && x < __builtin_inff();
    double t = (double)y * mm_ln_pos((double)(ok ? x : 1.0f));
    t = fmin(fmax(t, -800.0), 800.0);          // (e^800 rounds to +inf in float32, e^-800 to 0)
    float r = (float)mm_exp(t);
    if (!ok) {
        const float big = y < 0.0f ? 0.0f : __builtin_inff(), small = y < 0.0f ? __builtin_inff() : 0.0f;
        r = x == 0.0f ? small : (x == __builtin_inff() ? big : __builtin_nanf(""));
        if (y == 0.0f) r = 1.0f;
    }
    return y != y ? y : r;
}

// expf(t), correctly rounded
__device__ __forceinline__ float mm_expf(float t)
{
    const float r = (float)mm_exp(fmin(fmax((double)t, -800.0), 800.0));
    return t != t ? t : r;
}

// log((double)x) of a float32 (the float64 logarithm of an expected value): ln 0 = -inf, ln(+inf) = +inf, NaN below 0
__device__ __forceinline__ double mm_ln(float x)
{
    const bool ok = x > 0.0f && x < __builtin_inff();
    double r = mm_ln_pos((double)(ok ? x : 1.0f));
    if (!ok) r = x == 0.0f ? -(double)__builtin_inff() : (x == __builtin_inff() ? (double)__builtin_inff() : (double)__builtin_nanf(""));
    return r;
}
#endif
