// rt_qfilt.hpp — host side of the conservative f32 quad filter (rt_layout.h QFiltPair, rt_device_scene.h quad_pair_keep): the
// per-quad constants and their error terms.  The filter may only err towards "keep": a quad it drops is one the exact f64
// Quad::hit (src/quad.rs:96-133) rejects, for every ray and every interval contained in the one the filter is given.
//
// What the device computes for a ray (o, d) — everything in f32, fmas allowed, inputs rounded to nearest:
//     dn = n . d            nm = D - n . o          t~ = nm * rcp(dn)
//     p  = o + t~ d         a~ = A . p - (A . q + 1/2)      b~ = B . p - (B . q + 1/2)
// and, next to each, a bound on how far it can be from what the f64 test computes (eps = 2^-24):
//     |dn - n.d| <= 5 eps |n|_1 max|d|,   |nm - (D - n.o)| <= 5 eps (|D| + |n|_1 max|o|)     (inputs, products, three additions);
//     the f64 values are within 2^-29 of those bounds of the real ones.  With Ed2 := 14 eps |n|_1 max|d| (= 2 x 7 eps ...) and
//     En2 := 14 eps (|D| + |n|_1 max|o|):   if |dn| > Ed2 then  |t~ - t| <= (En2 + |t~| Ed2) |rcp| + 2^-21 |t~| =: Et
//     (quotient of two perturbed numbers with the denominator off by less than half; rcp and the product: 2^-22 |t~|);
//     |p_i - (o + t d)_i| <= Et max|d| + 2^-21 (max|o| + |t~| max|d|) =: Ep  (share of it for the three roundings of the dot products)
//     |a~ - (alpha - 1/2)| <= |A|_1 Ep + K_alpha,   K_alpha = 2^-20 (|A.q| + 1/2) + 2^-40 |A|_1 |q|_1  (+ the f64 side's own rounding)
// The quad is dropped iff |dn| > Ed2 and one of  t~ + Et < tmin,  t~ - Et > tmax,  |a~| - |A|_1 Ep > 1/2 + K_alpha,  the same for b~
// holds as a FINITE positive excess (an overflow anywhere gives inf - inf = NaN or an infinite bound: never a drop).
// The error terms |n|_1, |D|, |A|_1, K are those of the PAIR (the larger of its two quads', one value each): a flat leaf's quads are of
// one size, the bounds stay a few 1e-6 of the unit square, and a record is 112 bytes instead of 144.
// The same bounds also tell when alpha and beta are CERTAINLY inside [0, 1] (|a~| + bound <= 1/2): the exact test of such a survivor
// need not evaluate them (its verdict on them is known), only its plane distance — two thirds of its arithmetic.
// Quads whose constants leave [2^-40, 2^40] are never filtered (a NaN normal makes dn a NaN).
#pragma once
#include "rt_layout.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace rtd {

inline float qfilt_up(double x) { // a float not below x (x >= 0)
    float f = (float)x;
    if (!((double)f >= x)) f = std::nextafterf(f, INFINITY);
    return f;
}

// the quad's record values into `slot` of a pair record (zeroed by the caller), its error terms folded into the pair's (the larger
// of the two); false (and a NaN normal in the slot: never filtered) for a quad whose constants leave the range the bounds are derived for
inline bool qfilt_fill(const Quad &q, QFiltPair &out, int slot) {
    const double *u = q.u, *v = q.v, *w = q.w, *n = q.normal, *Q = q.q;
    const double A[3] = {v[1] * w[2] - v[2] * w[1], v[2] * w[0] - v[0] * w[2], v[0] * w[1] - v[1] * w[0]};
    const double B[3] = {w[1] * u[2] - w[2] * u[1], w[2] * u[0] - w[0] * u[2], w[0] * u[1] - w[1] * u[0]};
    const double AQ = A[0] * Q[0] + A[1] * Q[1] + A[2] * Q[2], BQ = B[0] * Q[0] + B[1] * Q[1] + B[2] * Q[2];
    const double n1 = std::fabs(n[0]) + std::fabs(n[1]) + std::fabs(n[2]), A1 = std::fabs(A[0]) + std::fabs(A[1]) + std::fabs(A[2]),
                 B1 = std::fabs(B[0]) + std::fabs(B[1]) + std::fabs(B[2]), Q1 = std::fabs(Q[0]) + std::fabs(Q[1]) + std::fabs(Q[2]);
    const double big = 0x1p40, small = 0x1p-40, eps = 0x1p-24;
    const bool ok = n1 >= small && n1 <= big && A1 >= small && A1 <= big && B1 >= small && B1 <= big && std::fabs(q.d) <= big &&
                    std::fabs(AQ) <= big && std::fabs(BQ) <= big && Q1 <= big; // (false for a NaN anywhere)
    float r[12] = {};
    if (ok) {
        for (int k = 0; k < 3; ++k) { r[k] = (float)n[k]; r[4 + k] = (float)A[k]; r[8 + k] = (float)B[k]; }
        r[3] = (float)q.d;
        r[7] = (float)(AQ + 0.5);
        r[11] = (float)(BQ + 0.5);
        out.n1c = std::fmax(out.n1c, qfilt_up(14.0 * eps * n1));
        out.dc = std::fmax(out.dc, qfilt_up(14.0 * eps * std::fabs(q.d) + 0x1p-100));
        out.a1 = std::fmax(out.a1, qfilt_up(std::fmax(A1, B1) * (1.0 + 0x1p-20)));
        out.ka = std::fmax(out.ka, qfilt_up(0.5 + 0x1p-20 * (std::fmax(std::fabs(AQ), std::fabs(BQ)) + 0.5) + 0x1p-40 * std::fmax(A1, B1) * Q1 + 0x1p-100));
    } else {
        r[0] = r[1] = r[2] = NAN; // dn is a NaN: the guard |dn| > Ed2 fails, the quad is always kept
    }
    for (int k = 0; k < 12; ++k) out.v[k][slot] = r[k];
    return ok;
}

// record i = the pair (quad i, quad i + 1); the last record's second slot is a quad that is never filtered (and never looked at:
// the quad stage masks the pair's bits with its leaf's count)
inline std::vector<QFiltPair> qfilt_table(const std::vector<Quad> &quads) {
    std::vector<QFiltPair> t(quads.size());
    for (size_t i = 0; i < quads.size(); ++i) {
        std::memset(&t[i], 0, sizeof t[i]);
        qfilt_fill(quads[i], t[i], 0);
        if (i + 1 < quads.size()) qfilt_fill(quads[i + 1], t[i], 1);
        else t[i].v[0][1] = t[i].v[1][1] = t[i].v[2][1] = NAN;
    }
    return t;
}

} // namespace rtd
