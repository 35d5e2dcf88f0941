// Device-side f64 arithmetic for the render kernel: Vec3 in the reference's operation order, the ABI's
// seeded per-path RNG (a stream keyed by a hash of the (seed, pixel, sample) counters) and the fixed transcendental algorithms (include/rt_amd.h, "Normative definitions").
// Compiled with -ffp-contract=off: a*b+c is two roundings, exactly as the Rust reference computes it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_shared_math.h"

namespace rtk {

#define RT_DEV __device__ __forceinline__

// ---- Vec3 (reference: src/vec3.rs) ----------------------------------------------------------------------
struct V3 {
    double x, y, z;
};
RT_DEV V3 v3(double x, double y, double z) { return V3{x, y, z}; }
RT_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_DEV V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
RT_DEV V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
RT_DEV V3 operator*(V3 a, double s) { return V3{a.x * s, a.y * s, a.z * s}; }
RT_DEV double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }          // src/vec3.rs:104-106
RT_DEV double len2(V3 a) { return dot(a, a); }
RT_DEV V3 cross(V3 a, V3 b) {                                                          // src/vec3.rs:133-139
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
RT_DEV V3 normalize(V3 a) { return a * (1.0 / __builtin_sqrt(dot(a, a))); }            // src/vec3.rs:124-131
RT_DEV V3 div(V3 a, double s) { return a * (1.0 / s); }                                // src/vec3.rs:244-249
RT_DEV bool near_zero(V3 a) {                                                          // src/vec3.rs:113-116
    const double EPS = 1e-8;
    return __builtin_fabs(a.x) < EPS && __builtin_fabs(a.y) < EPS && __builtin_fabs(a.z) < EPS;
}
RT_DEV V3 reflect(V3 v, V3 n) { return v - n * (2.0 * dot(v, n)); }                    // src/vec3.rs:91-93
RT_DEV V3 refract(V3 v, V3 n, double etai_over_etat) {                                 // src/vec3.rs:96-101
    double cos_theta = __builtin_fmin(dot(-v, n), 1.0);
    V3 r_out_perp = (v + n * cos_theta) * etai_over_etat;
    V3 r_out_parallel = n * (-__builtin_sqrt(__builtin_fabs(1.0 - len2(r_out_perp))));
    return r_out_perp + r_out_parallel;
}

// A float not below (f32_above) / not above (f32_below) the double x, at most 1.5 ulps away: x rounded to nearest, then moved by more
// than half an ulp and at most one — 2^-24 (1 + 2^-23) |f|; exactly half would be lost to the tie at a power of two — and by the
// smallest normal float on top (which only shows below 2^-102: an x that rounds to zero).  The rounded value is first clamped to
// +-FLT_MAX, so that an x beyond the float range, infinite or not, ends at the right infinity on its outer side and at a finite float
// on its inner side (inf x 2^-24 - inf would be a NaN; -inf is not above -1e300).  Four instructions where the exact directed rounding
// (__double2float_ru / _rd: a conversion and an integer next-after with its special cases) takes sixteen; the ordered walk's box test
// only needs an interval that CONTAINS (cur_tmin, cur_tmax), not the tightest one (tests/test_gpu_parity.py checks both properties).
RT_DEV float f32_above(double x) {
    const float f = __builtin_amdgcn_fmed3f((float)x, -0x1.fffffep+127f, 0x1.fffffep+127f);
    return __builtin_fmaf(__builtin_fabsf(f), 0x1.000002p-24f, f) + 0x1p-126f;
}
RT_DEV float f32_below(double x) {
    const float f = __builtin_amdgcn_fmed3f((float)x, -0x1.fffffep+127f, 0x1.fffffep+127f);
    return __builtin_fmaf(__builtin_fabsf(f), -0x1.000002p-24f, f) - 0x1p-126f;
}
RT_DEV uint64_t f2u(double x) { return (uint64_t)__double_as_longlong(x); }
RT_DEV double u2f(uint64_t u) { return __longlong_as_double((long long)u); }

// ---- RNG (rt_amd.h "RNG") -------------------------------------------------------------------------------
RT_DEV uint64_t mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
constexpr uint64_t RNG_GAMMA = 0x9E3779B97F4A7C15ull;

// The stream of one camera path: RomuDuoJr (Overton 2020: two 64-bit words, one multiply, one subtract, one rotate per
// draw — 9 VALU instructions against splitmix64's 20), started from the path's key.
struct Rng {
    uint64_t x, y;
    RT_DEV void start_key(uint64_t key) {
        x = key;
        y = mix64(key + RNG_GAMMA);
    }
    RT_DEV void start(uint64_t seed_mixed, uint32_t pixel, uint32_t sample) {
        start_key(mix64(seed_mixed ^ (((uint64_t)pixel << 32) | (uint64_t)sample)));
    }
    RT_DEV uint64_t next() {
        const uint64_t xp = x;
        x = 0xD3833E804F4C574Bull * y;
        y = y - xp;
        y = (y << 27) | (y >> 37);
        return xp;
    }
    // the stream one draw back (next() is a bijection of the state: x' = C y, y' = rotl(y - x, 27); C^-1 mod 2^64 below)
    RT_DEV void unnext() {
        const uint64_t y_old = 0x43D68ED20CD1FA63ull * x;
        const uint64_t diff = (y >> 27) | (y << 37);
        x = y_old - diff;
        y = y_old;
    }
    RT_DEV double random() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    RT_DEV double range(double lo, double hi) {
        double value1_2 = u2f((next() >> 12) | 0x3ff0000000000000ull);
        return (value1_2 - 1.0) * (hi - lo) + lo;
    }
};

// ---- fixed transcendental algorithms (DESIGN.md "Device math") ------------------------------------------
// (ln lives in rt_shared_math.h: the host library's output stage runs the same code)
using rtm::rt_log;

RT_DEV double k_sin(double x, double y) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = x * x;
    double v = z * x;
    double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
RT_DEV double k_cos(double x, double y) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = x * x;
    double w = z * z;
    double r = z * (C1 + z * (C2 + z * C3)) + (w * w) * (C4 + z * (C5 + z * C6));
    double hz = 0.5 * z;
    w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - x * y));
}
RT_DEV double rt_sin(double x) {
    const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
                 pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21;
    if (!(__builtin_fabs(x) <= 1.0e6)) return (x - x) / (x - x);
    double fn = __builtin_rint(x * invpio2);
    double t = x - fn * pio2_1;
    double w = fn * pio2_2;
    double r = t - w;
    w = fn * pio2_2t - ((t - r) - w);
    double y0 = r - w;
    double y1 = (r - y0) - w;
    int n = (int)fn;
    double s = k_sin(y0, y1), c = k_cos(y0, y1);
    double v = (n & 1) ? c : s;
    return (n & 2) ? -v : v;
}

RT_DEV double acos_ratio(double z) {
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
                 pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
                 qS4 = 7.70381505559019352791e-02;
    double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    return p / q;
}
RT_DEV double rt_acos(double x) {
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17,
                 pi = 3.14159265358979311600e+00;
    double ax = __builtin_fabs(x);
    if (!(ax < 1.0)) {
        if (x == 1.0) return 0.0;
        if (x == -1.0) return pi + 2.0 * pio2_lo;
        return (x - x) / (x - x);
    }
    if (ax < 0.5) {
        if (ax < 6.938893903907228e-18) return pio2_hi + pio2_lo;
        double r = acos_ratio(x * x);
        return pio2_hi - (x - (pio2_lo - x * r));
    } else if (x < 0.0) {
        double z = (1.0 + x) * 0.5;
        double r = acos_ratio(z);
        double s = __builtin_sqrt(z);
        double w = r * s - pio2_lo;
        return pi - 2.0 * (s + w);
    } else {
        double z = (1.0 - x) * 0.5;
        double s = __builtin_sqrt(z);
        double df = u2f(f2u(s) & 0xffffffff00000000ull);
        double c = (z - df * df) / (s + df);
        double r = acos_ratio(z);
        double w = r * s + c;
        return 2.0 * (df + w);
    }
}

RT_DEV double rt_atan_pos(double x) {
    const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01, aT2 = 1.42857142725034663711e-01,
                 aT3 = -1.11111104054623557880e-01, aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
                 aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02, aT8 = 4.97687799461593236017e-02,
                 aT9 = -3.65315727442169155270e-02, aT10 = 1.62858201153657823623e-02;
    int id;
    double hi = 0.0, lo = 0.0;
    if (x >= 73786976294838206464.0) return 1.57079632679489655800e+00 + 6.12323399573676603587e-17;
    if (x < 0.4375) {
        if (x < 1.862645149230957e-09) return x;
        id = -1;
    } else if (x < 1.1875) {
        if (x < 0.6875) { id = 0; x = (2.0 * x - 1.0) / (2.0 + x); hi = 4.63647609000806093515e-01; lo = 2.26987774529616870924e-17; }
        else            { id = 1; x = (x - 1.0) / (x + 1.0);       hi = 7.85398163397448278999e-01; lo = 3.06161699786838301793e-17; }
    } else {
        if (x < 2.4375) { id = 2; x = (x - 1.5) / (1.0 + 1.5 * x); hi = 9.82793723247329054082e-01; lo = 1.39033110312309984516e-17; }
        else            { id = 3; x = -1.0 / x;                    hi = 1.57079632679489655800e+00; lo = 6.12323399573676603587e-17; }
    }
    double z = x * x;
    double w = z * z;
    double s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    double s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    return hi - ((x * (s1 + s2) - lo) - x);
}
RT_DEV double rt_atan2(double y, double x) {
    const double pi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16, pi_o_2 = 1.5707963267948965580E+00;
    if (x != x || y != y) return x + y;
    int sx = (int)(f2u(x) >> 63), sy = (int)(f2u(y) >> 63);
    int m = sy + 2 * sx;
    if (y == 0.0) {
        if (m < 2) return y;
        return m == 2 ? pi : -pi;
    }
    if (x == 0.0) return sy ? -pi_o_2 : pi_o_2;
    double z = rt_atan_pos(__builtin_fabs(y / x));
    switch (m) {
    case 0: return z;
    case 1: return -z;
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}

RT_DEV double rt_pow5(double x) {
    double x2 = x * x;
    double x4 = x2 * x2;
    return x4 * x;
}

// Rust `f64 as i32` / `as u32` (saturating, NaN -> 0)
RT_DEV int32_t f64_as_i32(double x) {
    if (x != x) return 0;
    if (x <= -2147483648.0) return INT32_MIN;
    if (x >= 2147483647.0) return INT32_MAX;
    return (int32_t)x;
}
RT_DEV uint32_t f64_as_u32(double x) {
    if (!(x > 0.0)) return 0;
    if (x >= 4294967295.0) return UINT32_MAX;
    return (uint32_t)x;
}

} // namespace rtk
