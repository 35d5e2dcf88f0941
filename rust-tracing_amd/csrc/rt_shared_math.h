// rt_shared_math.h — fixed scalar algorithms that BOTH sides of the boundary run: the render kernel (hipcc, device) and the
// host library (g++, rust-tracing_amd/host/).  Built only from + - * / comparisons and bit moves, compiled without FMA
// contraction on either side, so device and host results agree bit for bit (include/rt_amd.h "Normative definitions").
//   rt_log   natural logarithm (ConstantMedium's free path, src/constant_medium.rs:48; and gamma below)
//   rt_exp   exponential (gamma below)
//   rt_gamma_encode   x^(1/2.2), the output stage's linear_to_gamma (src/color.rs:3-6)
//   rt_quantise       (256 * clamp(g, 0, 0.999)) as u8 (src/color.rs:12-19)
// Coefficients: FreeBSD msun e_log.c / e_exp.c (public constants of the published algorithms).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rtm {

RT_HD uint64_t f2u(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
RT_HD double u2f(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }

// x = 2^k * m, m in [sqrt(1/2), sqrt(2)); s = f / (2 + f), f = m - 1; degree-7 even/odd split polynomial; <= 1 ulp
RT_HD double rt_log(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t b = f2u(x);
    if ((b << 1) == 0) return -__builtin_inf();
    if (b >> 63) return __builtin_nan("");
    if ((b >> 52) == 0x7ff) return x;
    int64_t e = (int64_t)(b >> 52);
    if (e == 0) {
        x *= 18014398509481984.0;
        b = f2u(x);
        e = (int64_t)(b >> 52) - 54;
    }
    uint64_t mant = b & 0x000fffffffffffffull;
    int64_t k;
    double m;
    if (mant >= 0x6a09e667f3bcdull) { m = u2f(mant | 0x3fe0000000000000ull); k = e - 1022; }
    else                            { m = u2f(mant | 0x3ff0000000000000ull); k = e - 1023; }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double R = t2 + t1;
    double hfsq = 0.5 * f * f;
    double dk = (double)k;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

// exp(x) for |x| <= 700 (beyond: +inf / 0; NaN stays NaN): k = round(x / ln2), r = x - k ln2 in two pieces, degree-5 Remez
// polynomial in r^2 for r (e^r + 1) / (e^r - 1), scaled by 2^k through the exponent field; <= 1 ulp
RT_HD double rt_exp(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10, inv_ln2 = 1.44269504088896338700e+00,
                 P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > 700.0) return __builtin_inf();
    if (x < -700.0) return 0.0;
    const double ax = x < 0.0 ? -x : x;
    double hi = x, lo = 0.0;
    int64_t k = 0;
    if (ax > 0.34657359027997264) { // 0.5 ln2
        const double t = (double)(int64_t)(inv_ln2 * x + (x < 0.0 ? -0.5 : 0.5));
        k = (int64_t)t;
        hi = x - t * ln2_hi;
        lo = t * ln2_lo;
        x = hi - lo;
    } else if (ax < 3.725290298461914e-09) { // 2^-28
        return 1.0 + x;
    }
    const double t = x * x;
    const double c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
    const double y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
    return u2f(f2u(y) + ((uint64_t)k << 52)); // y in [0.5, 2], |k| <= 1010: the exponent field neither over- nor underflows
}

// linear_to_gamma: x.powf(1.0 / 2.2) (src/color.rs:3-6) as exp(ln(x) * (1 / 2.2)); powf's special cases for this exponent:
// +-0 -> 0, x < 0 -> NaN, +inf -> +inf, NaN -> NaN.  Relative error <= 2^-52 (1 + |ln x| / 2.2) of the exact power.
RT_HD double rt_gamma_encode(double x) {
    if (x != x) return x;
    if (x == 0.0) return 0.0;
    if (x < 0.0) return __builtin_nan("");
    if (x == __builtin_inf()) return x;
    return rt_exp(rt_log(x) * (1.0 / 2.2));
}

// `(256.0 * intensity.clamp(g)) as u8` with intensity = [0, 0.999] (src/color.rs:12-19): f64::clamp keeps NaN, and Rust's
// `NaN as u8` is 0
RT_HD uint8_t rt_quantise(double g) {
    if (g != g) return 0;
    const double c = g < 0.0 ? 0.0 : (g > 0.999 ? 0.999 : g);
    return (uint8_t)(256.0 * c);
}

} // namespace rtm
