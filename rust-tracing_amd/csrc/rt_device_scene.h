// rt_device_scene.h — device-side pieces of the render kernel (rt_kernel.hip): frame changes, textures and Perlin noise, record
// loads, the conservative f32 box and quad tests.  Everything is file-local (anonymous namespace).
#pragma once
#include "rt_kernels.h"
#include "rt_amd_debug.h"
#include "rt_device_math.h"

#include <hip/hip_runtime.h>

using namespace rtd;
using namespace rtk;

namespace {

struct Counts {
    uint32_t samples, rays, node_visits, sphere_tests, quad_tests, medium_visits, rng_draws, noise_evals,
        image_lookups, instance_enters;
};

RT_DEV V3 ld3(const double *p) { return V3{p[0], p[1], p[2]}; }
RT_DEV V3 from(const rt_vec3 &a) { return V3{a.x, a.y, a.z}; }

// ---- frame changes (Translate::hit then RotateY::hit, src/hittable.rs:96-106,:159-188) -------------------
RT_DEV void apply_instance(const Instance &in, V3 &o, V3 &d) {
    // all fields are read up front (one memory round trip, not one per `if (flags & ...)`)
    const uint32_t flags = in.flags;
    const V3 offset = ld3(in.offset);
    const double c = in.cos_theta, s = in.sin_theta;
    if (flags & INST_TRANSLATE) o = o - offset;
    if (flags & INST_ROTATE) {
        const double ox = c * o.x - s * o.z, oz = s * o.x + c * o.z;
        const double dx = c * d.x - s * d.z, dz = s * d.x + c * d.z;
        o.x = ox; o.z = oz; d.x = dx; d.z = dz;
    }
}
// world ray -> the frame of instance `inst` (outermost ancestor first); inst < 0: world frame
RT_DEV void ray_to_frame(const Instance *insts, int32_t inst, V3 &o, V3 &d) {
    if (inst < 0) return;
    const uint32_t depth = insts[inst].depth;
    for (uint32_t lv = 0; lv <= depth; ++lv) {
        int32_t a = inst;
        for (uint32_t k = depth; k > lv; --k) a = insts[a].parent;
        apply_instance(insts[a], o, d);
    }
}
// hit point and normal from the frame of `inst` back to the world (innermost first):
// RotateY's output step (src/hittable.rs:173-182) then Translate's (src/hittable.rs:101)
RT_DEV void hit_to_world(const Instance *insts, int32_t inst, V3 &p, V3 &n) {
    while (inst >= 0) {
        const Instance &in = insts[inst];
        if (in.flags & INST_ROTATE) {
            const double c = in.cos_theta, s = in.sin_theta;
            const double px = c * p.x + s * p.z, pz = -s * p.x + c * p.z;
            const double nx = c * n.x + s * n.z, nz = -s * n.x + c * n.z;
            p.x = px; p.z = pz; n.x = nx; n.z = nz;
        }
        if (in.flags & INST_TRANSLATE) p = p + ld3(in.offset);
        inst = in.parent;
    }
}

// ---- random vectors (src/vec3.rs:54-88) ------------------------------------------------------------------
template <bool COUNT> RT_DEV V3 random_in_unit_sphere(Rng &rng, Counts &cn) {
    for (;;) {
        const double x = rng.range(-1.0, 1.0);
        const double y = rng.range(-1.0, 1.0);
        const double z = rng.range(-1.0, 1.0);
        if (COUNT) cn.rng_draws += 3;
        const V3 p = v3(x, y, z);
        if (len2(p) < 1.0) return p;
    }
}
template <bool COUNT> RT_DEV V3 random_unit_vector(Rng &rng, Counts &cn) {
    return normalize(random_in_unit_sphere<COUNT>(rng, cn));
}

// ---- Perlin (src/perlin.rs:27-64,:81-100) ----------------------------------------------------------------
// Rare and register-hungry: kept as rolled loops (one corner of the lattice cell per iteration) so that its
// temporaries do not inflate the register allocation of the whole kernel.
RT_DEV double perlin_turbulence(const rt_perlin *pn, V3 p, int depth) {
    double acc = 0.0;
    double wgt = 1.0;
#pragma unroll 1
    for (int dpt = 0; dpt < depth; ++dpt) {
        // Perlin::noise (src/perlin.rs:27-50)
        const int32_t i = f64_as_i32(__builtin_floor(p.x));
        const int32_t j = f64_as_i32(__builtin_floor(p.y));
        const int32_t k = f64_as_i32(__builtin_floor(p.z));
        const double u = p.x - (double)i;
        const double v = p.y - (double)j;
        const double w = p.z - (double)k;
        // trilinear_interpolation (src/perlin.rs:81-100)
        const double uu = u * u * (3.0 - 2.0 * u);
        const double vv = v * v * (3.0 - 2.0 * v);
        const double ww = w * w * (3.0 - 2.0 * w);
        double noise = 0.0;
#pragma unroll 1
        for (int corner = 0; corner < 8; ++corner) { // (di, dj, dk) in the reference's loop order: dk fastest
            const int di = corner >> 2, dj = (corner >> 1) & 1, dk = corner & 1;
            const int32_t idx = pn->perm_x[(uint32_t)(i + di) & 255u] ^ pn->perm_y[(uint32_t)(j + dj) & 255u] ^
                                pn->perm_z[(uint32_t)(k + dk) & 255u];
            const V3 c = from(pn->ranvec[idx]);
            // `i as FP * uu + (1 - i) as FP * (1 - uu)` is exactly uu (i = 1) or 1 - uu (i = 0)
            const double fi = di ? uu : 1.0 - uu;
            const double fj = dj ? vv : 1.0 - vv;
            const double fk = dk ? ww : 1.0 - ww;
            const V3 weight_v = v3(u - (double)di, v - (double)dj, w - (double)dk);
            noise += fi * fj * fk * dot(c, weight_v);
        }
        acc += wgt * noise;
        wgt *= 0.5;
        p = p * 2.0;
    }
    return __builtin_fabs(acc);
}

RT_DEV double clamp01(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }

// ---- Texture::value (src/texture.rs) ---------------------------------------------------------------------
template <bool COUNT> RT_DEV V3 texture_value(const KParams &P, const rt_texture *texs, const rt_perlin *perlins, uint32_t tex, double u, double v, V3 p,
                                              Counts &cn) {
    const rt_texture *t = &texs[tex];
    while (t->kind == RT_TEXTURE_CHECKER) { // src/texture.rs:59-69
        const int32_t x = f64_as_i32(__builtin_floor(t->inv_scale * p.x));
        const int32_t y = f64_as_i32(__builtin_floor(t->inv_scale * p.y));
        const int32_t z = f64_as_i32(__builtin_floor(t->inv_scale * p.z));
        const int32_t s = (int32_t)((uint32_t)x + (uint32_t)y + (uint32_t)z);
        t = &texs[(s % 2 == 0) ? t->even : t->odd];
    }
    if (t->kind == RT_TEXTURE_SOLID) return from(t->color); // src/texture.rs:32-36
    if (t->kind == RT_TEXTURE_IMAGE) {                      // src/texture.rs:82-92
        if (COUNT) cn.image_lookups++;
        const ImageRef im = P.images[t->image];
        const double uc = clamp01(u);
        const double vc = 1.0 - clamp01(v);
        const uint32_t i = f64_as_u32(uc * (double)(im.width - 1u));
        const uint32_t j = f64_as_u32(vc * (double)(im.height - 1u));
        // (8x8 texel tiles, rt_layout.h ImageRef: the same texel the reference's row-major index names)
        const uint8_t *px = P.texels + im.offset + (((size_t)(j >> 3) * im.tiles_x + (i >> 3)) * 64u + ((j & 7u) << 3) + (i & 7u)) * 3u;
        return v3(P.srgb_lut[px[0]], P.srgb_lut[px[1]], P.srgb_lut[px[2]]);
    }
    // RT_TEXTURE_NOISE, src/texture.rs:107-110
    if (COUNT) cn.noise_evals++;
    const double turb = perlin_turbulence(&perlins[t->perlin], p, 7);
    const double s = rt_sin(t->scale * p.z + 10.0 * turb) * 0.5 + 0.5;
    return v3(s, s, s);
}

// What one box-stage round needs of a record
struct NodeData {
    float lo[3], hi[3];
    uint32_t skip, packed;
};

// SCENE_IN_LDS: the node, sphere and quad tables are copied into the CU's LDS once per workgroup and every lane
// gathers from there (divergent 16-byte reads: ~10x lower latency than L1 and no tag-lookup serialisation).
// LDS image: node_a[N] = (x.lo, x.hi, y.lo, y.hi) | node_b[N] = (z.lo, z.hi, skip, packed) | spheres (64 B) |
// quads (144 B).  The two halves of a node record are separate tables so that lanes reading the same half of
// different records spread over all 16 four-bank slots (a 32-byte record stride would use only 8).
template <int LDS> RT_DEV NodeData load_node(const KParams &P, const unsigned char *lds, uint32_t id) {
    float4 a, b;
    if constexpr (LDS != 0) {
        a = reinterpret_cast<const float4 *>(lds)[id];
        b = reinterpret_cast<const float4 *>(lds + P.lds_off_node_b)[id];
    } else {
        const float4 *np = reinterpret_cast<const float4 *>(&P.nodes[id]);
        a = np[0];
        b = np[1];
    }
    NodeData n;
    n.lo[0] = a.x; n.hi[0] = a.y; n.lo[1] = a.z; n.hi[1] = a.w; n.lo[2] = b.x; n.hi[2] = b.y;
    n.skip = __float_as_uint(b.z);
    n.packed = __float_as_uint(b.w);
    return n;
}

// Ordered layout, as the device holds it (global memory and LDS alike): seven tables indexed by record —
//   X+ | X- | Y+ | Y- | Z+ | Z-   16 bytes each: (child 0, child 1) planes the ray ENTERS through on that axis, then the
//                                  (child 0, child 1) planes it LEAVES through; "+" for rays with 1/d >= 0 on the axis
//                                  (enter = lo, leave = hi), "-" the same four values swapped (enter = hi, leave = lo)
//   R                              8 bytes: the two child references
// A lane reads ONE of each axis pair, chosen by the sign of its ray's 1/d — the choice is a per-ray byte offset, so
// the loaded registers already hold (near, near, far, far) pairs ready for the packed fmas: no per-visit selects.
// Separate tables keep divergent 16-byte reads spread over all LDS banks (as for the threaded records above).
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct OPair {
    f32x2 nx, fx, ny, fy, nz, fz; // entering / leaving planes of (child 0, child 1) per axis
    uint32_t c0, c1;
};
template <int LDS> RT_DEV OPair load_opair(const KParams &P, const unsigned char *lds, uint32_t id, uint32_t offx, uint32_t offy, uint32_t offz) {
    float4 qx, qy, qz;
    uint2 r;
    if constexpr (LDS != 0) { // seven tables
        qx = *reinterpret_cast<const float4 *>(lds + offx + id * 16u);
        qy = *reinterpret_cast<const float4 *>(lds + offy + id * 16u);
        qz = *reinterpret_cast<const float4 *>(lds + offz + id * 16u);
        r = *reinterpret_cast<const uint2 *>(lds + 6u * P.lds_off_node_b + id * 8u);
    } else { // global memory: the same seven pieces side by side in one 128-byte line per record (offsets 0 .. 80, 96)
        const unsigned char *rec = reinterpret_cast<const unsigned char *>(P.oimage) + (size_t)id * 128u;
        qx = *reinterpret_cast<const float4 *>(rec + offx);
        qy = *reinterpret_cast<const float4 *>(rec + offy);
        qz = *reinterpret_cast<const float4 *>(rec + offz);
        r = *reinterpret_cast<const uint2 *>(rec + 96);
    }
    OPair n;
    n.nx = f32x2{qx.x, qx.y}; n.fx = f32x2{qx.z, qx.w};
    n.ny = f32x2{qy.x, qy.y}; n.fy = f32x2{qy.z, qy.w};
    n.nz = f32x2{qz.x, qz.y}; n.fz = f32x2{qz.z, qz.w};
    n.c0 = r.x;
    n.c1 = r.y;
    return n;
}

// f32 copies of a ray for the conservative box test: origin, 1/d, and a bound E on how far rounding the origin to
// f32 can move a slab distance on each axis; `degenerate`: some 1/d or E is not finite -> every box is entered.
struct Ray32 {
    float ox, oy, oz, ix, iy, iz, ex, ey, ez;
    bool degenerate;
};
RT_DEV Ray32 make_ray32(V3 o, V3 d) {
    Ray32 r;
    r.ox = (float)o.x; r.oy = (float)o.y; r.oz = (float)o.z;
    // 1/d (the reference's per-visit quotient, src/aabb.rs:66) to f32 accuracy: v_rcp_f32 of the rounded d is within
    // 2 ulp of it, which the test's slack absorbs (below)
    r.ix = __builtin_amdgcn_rcpf((float)d.x); r.iy = __builtin_amdgcn_rcpf((float)d.y); r.iz = __builtin_amdgcn_rcpf((float)d.z);
    // |o - o32| <= 2^-24 |o|, i.e. at most 2^-24 |o| |1/d| in t; E carries a 4x margin
    r.ex = __builtin_fabsf(r.ox * r.ix) * 0x1p-22f;
    r.ey = __builtin_fabsf(r.oy * r.iy) * 0x1p-22f;
    r.ez = __builtin_fabsf(r.oz * r.iz) * 0x1p-22f;
    const float fsum = (r.ex + r.ey + r.ez) + (__builtin_fabsf(r.ix) + __builtin_fabsf(r.iy) + __builtin_fabsf(r.iz));
    r.degenerate = !(fsum < __builtin_inff()); // an inf or a NaN anywhere
    return r;
}
// Conservative f32 slab test.  The exact test (f64, interval narrowed axis by axis; equivalent to the reference's
// un-narrowed one, DESIGN.md "Box test") passes iff max(near) < min(far) over the three slabs and (tmin, tmax).
// Here: boxes are rounded outward; E bounds the effect of rounding the origin; every computed slab distance
// (b - o) * (1/d) carries at most 2^-24 (subtraction) + 2^-24 (product) + 2^-22 (1/d: conversion of d, v_rcp_f32)
// relative error, and tmin / tmax 2^-24 from their conversion: under 2^-21.5 in all.  Each slab's near / far distance
// is moved outward by its E, and the test declares a miss only if enter still exceeds leave by more than
// 2^-20 (|enter| + |leave|) — so it passes whenever the exact one does;
// when it passes although the exact one would not, the visit finds nothing (primitives are intersected in f64).
// A record without a box carries (-inf, +inf): always passes (inf - inf = NaN compares false).
RT_DEV bool box_miss_f32(const float lo[3], const float hi[3], const Ray32 &r, float tmin32, float tmax32) {
    const float t0x = (lo[0] - r.ox) * r.ix, t1x = (hi[0] - r.ox) * r.ix;
    const float t0y = (lo[1] - r.oy) * r.iy, t1y = (hi[1] - r.oy) * r.iy;
    const float t0z = (lo[2] - r.oz) * r.iz, t1z = (hi[2] - r.oz) * r.iz;
    const float enter = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x) - r.ex, __builtin_fminf(t0y, t1y) - r.ey),
                                        __builtin_fmaxf(__builtin_fminf(t0z, t1z) - r.ez, tmin32));
    const float leave = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t0x, t1x) + r.ex, __builtin_fmaxf(t0y, t1y) + r.ey),
                                        __builtin_fminf(__builtin_fmaxf(t0z, t1z) + r.ez, tmax32));
    const float gap = enter - leave;
    const float tol = (__builtin_fabsf(enter) + __builtin_fabsf(leave)) * 0x1p-20f;
    return !r.degenerate && gap > tol; // a NaN compares false: pass
}
// The ordered walk tests the two boxes of a record at once, in a form with a third of the instructions: per axis the plane
// the ray meets first is known from the sign of 1/d (load_opair reads the table laid out for that sign), each slab
// distance is ONE fused multiply-add
//     t = b * (1/d) - (o * (1/d) +- E)          (explicit fma: this f32 filter is not part of the f64 arithmetic contract)
// with the product o * (1/d) and an outward shift E folded into two per-ray constants per axis (`n` for entering planes,
// `f` for leaving ones), and the verdict is a plain comparison enter > leave.  E is an ABSOLUTE bound on everything that
// rounding can do to t on that axis, for any plane of the scene: with |b| <= B (the largest coordinate of any box, from the
// scene compiler) and i = 1/d,
//     |b i' - b i| <= 2^-22 B |i|           (1/d: conversion of d to f32 + v_rcp_f32)
//     |p - o i|    <= 1.6 * 2^-22 |o i|      (o -> f32, the same 1/d, the product's rounding)
//     rounding of p +- E and of the fma: <= 2^-23 (B |i| + |o i|)
// together under 2^-21 (B |i| + |o i|); E = 2^-20 (B |i| + |o i|) leaves a factor two.  Entering distances are therefore
// never over-, leaving distances never under-estimated; the interval ends are rounded outward when they are converted
// (tmin32 down, tmax32 up).  So the test passes whenever the exact one does (tests: rt_debug_box_tests).  In world
// units E is 2^-20 (B + |o|): random-spheres 2 mm (B = 2000, the ground sphere), Cornell 0.5 mm.
// The per-ray constants are kept as five register pairs, already negated where the fma subtracts, and each packed fma
// picks the half it needs for BOTH boxes through op_sel — so nothing has to be duplicated or negated per visit.
struct RayPair32 {
    f32x2 ixy;  // (1/d.x, 1/d.y)
    f32x2 izs;  // (1/d.z, sign bits as float bits: unused half)
    f32x2 nxy;  // -(o/d + E) on x, y: entering planes
    f32x2 fxy;  // -(o/d - E) on x, y: leaving planes
    f32x2 nfz;  // the same two constants on z
    uint32_t offx, offy, offz; // byte offsets of the tables this ray reads (load_opair): X+ or X-, Y+ or Y-, Z+ or Z-
    bool degenerate;
};
RT_DEV RayPair32 make_ray_pair32(V3 o, V3 d, uint32_t table_bytes, float extent) {
    RayPair32 r;
    const float ix = __builtin_amdgcn_rcpf((float)d.x), iy = __builtin_amdgcn_rcpf((float)d.y), iz = __builtin_amdgcn_rcpf((float)d.z);
    const float px = (float)o.x * ix, py = (float)o.y * iy, pz = (float)o.z * iz;
    const float ex = (extent * __builtin_fabsf(ix) + __builtin_fabsf(px)) * 0x1p-20f;
    const float ey = (extent * __builtin_fabsf(iy) + __builtin_fabsf(py)) * 0x1p-20f;
    const float ez = (extent * __builtin_fabsf(iz) + __builtin_fabsf(pz)) * 0x1p-20f;
    r.ixy = f32x2{ix, iy};
    r.izs = f32x2{iz, 0.0f};
    r.nxy = f32x2{-(px + ex), -(py + ey)};
    r.fxy = f32x2{-(px - ex), -(py - ey)};
    r.nfz = f32x2{-(pz + ez), -(pz - ez)};
    r.offx = (ix < 0.0f ? 1u : 0u) * table_bytes;
    r.offy = (iy < 0.0f ? 3u : 2u) * table_bytes;
    r.offz = (iz < 0.0f ? 5u : 4u) * table_bytes;
    const float fsum = (ex + ey + ez) + (__builtin_fabsf(ix) + __builtin_fabsf(iy) + __builtin_fabsf(iz));
    // an inf or a NaN anywhere, or a zero 1/d (an infinite direction component: plane * 0 is a NaN for an infinite plane): every box is entered
    r.degenerate = !(fsum < __builtin_inff()) || ix == 0.0f || iy == 0.0f || iz == 0.0f;
    return r;
}
// planes * (the chosen half of inv, for both boxes) + (the chosen half of off, for both boxes)
#define RT_PK_FMA_SEL(out, planes, inv, off, SEL)                                                             \
    asm("v_pk_fma_f32 %0, %1, %2, %3 " SEL : "=v"(out) : "v"(planes), "v"(inv), "v"(off))
#define RT_SEL_LO_LO "op_sel:[0,0,0] op_sel_hi:[1,0,0]" /* inv.lo, off.lo */
#define RT_SEL_HI_HI "op_sel:[0,1,1] op_sel_hi:[1,1,1]" /* inv.hi, off.hi */
#define RT_SEL_LO_HI "op_sel:[0,0,1] op_sel_hi:[1,0,1]" /* inv.lo, off.hi */
// enter0 / enter1: where the ray enters each box (for choosing which child to walk first: any choice is correct, the
// nearer one prunes more).  tmin32 / tmax32: the interval, rounded outward.
RT_DEV void box_pair_f32(const OPair &b, const RayPair32 &r, float tmin32, float tmax32, bool &miss0, bool &miss1, float &enter0,
                         float &enter1) {
    f32x2 tnx, tny, tnz, tfx, tfy, tfz;
    RT_PK_FMA_SEL(tnx, b.nx, r.ixy, r.nxy, RT_SEL_LO_LO);
    RT_PK_FMA_SEL(tny, b.ny, r.ixy, r.nxy, RT_SEL_HI_HI);
    RT_PK_FMA_SEL(tnz, b.nz, r.izs, r.nfz, RT_SEL_LO_LO);
    RT_PK_FMA_SEL(tfx, b.fx, r.ixy, r.fxy, RT_SEL_LO_LO);
    RT_PK_FMA_SEL(tfy, b.fy, r.ixy, r.fxy, RT_SEL_HI_HI);
    RT_PK_FMA_SEL(tfz, b.fz, r.izs, r.nfz, RT_SEL_LO_HI);
    // (v_max3 / v_min3 spelled out: through fmaxf the compiler first "quiets" each operand it cannot prove is no signalling
    // NaN — one extra instruction per operand, five per visit; the instructions themselves return the non-NaN operand)
    float en0, en1, le0, le1;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(en0) : "v"(tnx.x), "v"(tny.x), "v"(tnz.x));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(en1) : "v"(tnx.y), "v"(tny.y), "v"(tnz.y));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(le0) : "v"(tfx.x), "v"(tfy.x), "v"(tfz.x));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(le1) : "v"(tfx.y), "v"(tfy.y), "v"(tfz.y));
    asm("v_max_f32 %0, %1, %2" : "=v"(en0) : "v"(en0), "v"(tmin32));
    asm("v_max_f32 %0, %1, %2" : "=v"(en1) : "v"(en1), "v"(tmin32));
    asm("v_min_f32 %0, %1, %2" : "=v"(le0) : "v"(le0), "v"(tmax32));
    asm("v_min_f32 %0, %1, %2" : "=v"(le1) : "v"(le1), "v"(tmax32));
    miss0 = !r.degenerate && en0 > le0; // a NaN compares false: pass
    miss1 = !r.degenerate && en1 > le1;
    enter0 = en0;
    enter1 = en1;
}
// ---- wide records (rt_layout.h ONode4): four children per record -------------------------------------------------------------
// Tables as for the pairs, 32 bytes per record each: X+ | X- | Y+ | Y- | Z+ | Z-  = (near planes of children 0..3, far planes of
// children 0..3) for a ray of that sign on that axis; then R, 16 bytes: the four references.  In global memory the seven pieces of a
// record sit side by side in 256 bytes (offsets 0 .. 160, 192).
struct OQuad {
    f32x2 nx[2], fx[2], ny[2], fy[2], nz[2], fz[2]; // [0]: children 0, 1; [1]: children 2, 3
    uint32_t c[4];
};
template <int LDS> RT_DEV OQuad load_oquad(const KParams &P, const unsigned char *lds, uint32_t id, uint32_t offx, uint32_t offy, uint32_t offz) {
    float4 q[6];
    uint4 r;
    if constexpr (LDS != 0) {
        const unsigned char *bx = lds + offx + id * 32u, *by = lds + offy + id * 32u, *bz = lds + offz + id * 32u;
        q[0] = *reinterpret_cast<const float4 *>(bx); q[1] = *reinterpret_cast<const float4 *>(bx + 16);
        q[2] = *reinterpret_cast<const float4 *>(by); q[3] = *reinterpret_cast<const float4 *>(by + 16);
        q[4] = *reinterpret_cast<const float4 *>(bz); q[5] = *reinterpret_cast<const float4 *>(bz + 16);
        r = *reinterpret_cast<const uint4 *>(lds + 6u * P.lds_off_node_b + id * 16u);
    } else {
        const unsigned char *rec = reinterpret_cast<const unsigned char *>(P.oimage) + (size_t)id * 256u;
        q[0] = *reinterpret_cast<const float4 *>(rec + offx); q[1] = *reinterpret_cast<const float4 *>(rec + offx + 16);
        q[2] = *reinterpret_cast<const float4 *>(rec + offy); q[3] = *reinterpret_cast<const float4 *>(rec + offy + 16);
        q[4] = *reinterpret_cast<const float4 *>(rec + offz); q[5] = *reinterpret_cast<const float4 *>(rec + offz + 16);
        r = *reinterpret_cast<const uint4 *>(rec + 192);
    }
    OQuad n;
    n.nx[0] = f32x2{q[0].x, q[0].y}; n.nx[1] = f32x2{q[0].z, q[0].w}; n.fx[0] = f32x2{q[1].x, q[1].y}; n.fx[1] = f32x2{q[1].z, q[1].w};
    n.ny[0] = f32x2{q[2].x, q[2].y}; n.ny[1] = f32x2{q[2].z, q[2].w}; n.fy[0] = f32x2{q[3].x, q[3].y}; n.fy[1] = f32x2{q[3].z, q[3].w};
    n.nz[0] = f32x2{q[4].x, q[4].y}; n.nz[1] = f32x2{q[4].z, q[4].w}; n.fz[0] = f32x2{q[5].x, q[5].y}; n.fz[1] = f32x2{q[5].z, q[5].w};
    n.c[0] = r.x; n.c[1] = r.y; n.c[2] = r.z; n.c[3] = r.w;
    return n;
}
// the conservative test of box_pair_f32 for the four boxes of a record: where the ray enters each (clamped to the interval's start)
// and leaves it (clamped to its end); the ray misses box k iff enter[k] > leave[k]
RT_DEV void box_quad_f32(const OQuad &b, const RayPair32 &r, float tmin32, float tmax32, float enter[4], float leave[4]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f32x2 tnx, tny, tnz, tfx, tfy, tfz;
        RT_PK_FMA_SEL(tnx, b.nx[h], r.ixy, r.nxy, RT_SEL_LO_LO);
        RT_PK_FMA_SEL(tny, b.ny[h], r.ixy, r.nxy, RT_SEL_HI_HI);
        RT_PK_FMA_SEL(tnz, b.nz[h], r.izs, r.nfz, RT_SEL_LO_LO);
        RT_PK_FMA_SEL(tfx, b.fx[h], r.ixy, r.fxy, RT_SEL_LO_LO);
        RT_PK_FMA_SEL(tfy, b.fy[h], r.ixy, r.fxy, RT_SEL_HI_HI);
        RT_PK_FMA_SEL(tfz, b.fz[h], r.izs, r.nfz, RT_SEL_LO_HI);
        float en0, en1, le0, le1;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(en0) : "v"(tnx.x), "v"(tny.x), "v"(tnz.x));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(en1) : "v"(tnx.y), "v"(tny.y), "v"(tnz.y));
        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(le0) : "v"(tfx.x), "v"(tfy.x), "v"(tfz.x));
        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(le1) : "v"(tfx.y), "v"(tfy.y), "v"(tfz.y));
        asm("v_max_f32 %0, %1, %2" : "=v"(en0) : "v"(en0), "v"(tmin32));
        asm("v_max_f32 %0, %1, %2" : "=v"(en1) : "v"(en1), "v"(tmin32));
        asm("v_min_f32 %0, %1, %2" : "=v"(le0) : "v"(le0), "v"(tmax32));
        asm("v_min_f32 %0, %1, %2" : "=v"(le1) : "v"(le1), "v"(tmax32));
        enter[2 * h] = en0; enter[2 * h + 1] = en1;
        leave[2 * h] = le0; leave[2 * h + 1] = le1;
    }
}
// one box given as (x.lo, x.hi, y.lo, y.hi, z.lo, z.hi) in both slots of a pair, as the "+" / "-" tables would hold it
RT_DEV OPair opair_of_box(const float b[6], const RayPair32 &r) {
    OPair p;
    const bool sx = r.ixy.x < 0.0f, sy = r.ixy.y < 0.0f, sz = r.izs.x < 0.0f;
    p.nx = f32x2{sx ? b[1] : b[0], sx ? b[1] : b[0]}; p.fx = f32x2{sx ? b[0] : b[1], sx ? b[0] : b[1]};
    p.ny = f32x2{sy ? b[3] : b[2], sy ? b[3] : b[2]}; p.fy = f32x2{sy ? b[2] : b[3], sy ? b[2] : b[3]};
    p.nz = f32x2{sz ? b[5] : b[4], sz ? b[5] : b[4]}; p.fz = f32x2{sz ? b[4] : b[5], sz ? b[4] : b[5]};
    p.c0 = p.c1 = 0;
    return p;
}
// ---- conservative f32 filter for the quads of a flat leaf (rt_layout.h QFiltPair; bounds derived in rt_qfilt.hpp) -------------
// Quad::hit (src/quad.rs:96-133) is a plane distance (one f64 division) and two oblique coordinates (two cross products): ~85
// instructions a quad, and a flat leaf (Cornell's walls, a box's faces) holds six of which a ray hits one.  The filter computes t,
// alpha, beta for TWO quads at a time in packed f32 with an absolute error bound beside each, and drops a quad only when one of
// them is outside its range by more than its bound; the exact test then runs over the survivors alone.  It can only err towards
// "keep" (tests/test_gpu_parity.py: the exact test never accepts what the filter dropped).
struct QRay32 {
    float ox, oy, oz, dx, dy, dz; // the ray, rounded to nearest
    float om, dm;                 // max |o_i|, max |d_i|
};
RT_DEV QRay32 make_qray32(V3 o, V3 d) {
    QRay32 r;
    r.ox = (float)o.x; r.oy = (float)o.y; r.oz = (float)o.z;
    r.dx = (float)d.x; r.dy = (float)d.y; r.dz = (float)d.z;
    asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r.om) : "v"(r.ox), "v"(r.oy), "v"(r.oz));
    asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r.dm) : "v"(r.dx), "v"(r.dy), "v"(r.dz));
    return r;
}
RT_DEV f32x2 pk_splat(float x) { return f32x2{x, x}; }
RT_DEV f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); } // (v_pk_fma_f32; a splat folds into op_sel)
// bit 0 / bit 1: the first / second quad of the pair may be hit within (tmin32, tmax32) — an interval that CONTAINS the exact one
// Bits 0 / 1: the first / second quad of the pair may be hit; bits 8 / 9: ... and if its plane distance passes the exact test, so do
// alpha and beta (both certainly inside [0, 1]: the exact test need not evaluate them).
RT_DEV uint32_t quad_pair_keep(const QFiltPair *rec, const QRay32 &r, float tmin32, float tmax32) {
    const float4 *q4 = reinterpret_cast<const float4 *>(rec);
    const float4 c0 = q4[0], c1 = q4[1], bd = q4[6]; // bd: (n1c, dc, a1, ka)
    const f32x2 nx{c0.x, c0.y}, ny{c0.z, c0.w}, nz{c1.x, c1.y}, D{c1.z, c1.w};
    f32x2 dn = nx * pk_splat(r.dx);
    dn = pk_fma(ny, pk_splat(r.dy), dn);
    dn = pk_fma(nz, pk_splat(r.dz), dn);
    f32x2 nm = pk_fma(-nx, pk_splat(r.ox), D);
    nm = pk_fma(-ny, pk_splat(r.oy), nm);
    nm = pk_fma(-nz, pk_splat(r.oz), nm);
    const float ed2 = bd.x * r.dm, en2 = __builtin_fmaf(bd.x, r.om, bd.y); // (the same for both quads: the pair's bounds)
    const f32x2 rc{__builtin_amdgcn_rcpf(dn.x), __builtin_amdgcn_rcpf(dn.y)};
    const f32x2 t = nm * rc;
    const f32x2 at = __builtin_elementwise_abs(t);
    const f32x2 et = pk_fma(at, pk_splat(0x1p-21f), pk_fma(at, pk_splat(ed2), pk_splat(en2)) * __builtin_elementwise_abs(rc));
    // by how much each value is outside its range beyond its bound (a positive FINITE excess drops the quad; NaN and inf never do)
    const f32x2 s_lo = (pk_splat(tmin32) - t) - et, s_hi = (t - pk_splat(tmax32)) - et;
    // the hit point and how far off it may be
    const f32x2 px = pk_fma(t, pk_splat(r.dx), pk_splat(r.ox)), py = pk_fma(t, pk_splat(r.dy), pk_splat(r.oy)), pz = pk_fma(t, pk_splat(r.dz), pk_splat(r.oz));
    const f32x2 ep = pk_fma(et, pk_splat(r.dm), pk_fma(at, pk_splat(r.dm), pk_splat(r.om)) * pk_splat(0x1p-21f));
    const f32x2 eab = pk_fma(pk_splat(bd.z), ep, pk_splat(bd.w)); // 1/2 + the bound on alpha - 1/2 (and on beta - 1/2)
    // (the plane's rows are consumed before alpha's and beta's are loaded: with all seven in flight the allocator spills path state —
    // measured: Cornell 2697 Msamples/s without this line, 2710 with it)
    __builtin_amdgcn_sched_barrier(0);
    const float4 c2 = q4[2], c3 = q4[3], c4 = q4[4], c5 = q4[5];
    const f32x2 ax{c2.x, c2.y}, ay{c2.z, c2.w}, az{c3.x, c3.y}, aq{c3.z, c3.w}, bx{c4.x, c4.y}, by{c4.z, c4.w}, bz{c5.x, c5.y}, bq{c5.z, c5.w};
    f32x2 al = pk_fma(ax, px, -aq);
    al = pk_fma(ay, py, al);
    al = pk_fma(az, pz, al);
    f32x2 be = pk_fma(bx, px, -bq);
    be = pk_fma(by, py, be);
    be = pk_fma(bz, pz, be);
    const f32x2 aal = __builtin_elementwise_abs(al), abe = __builtin_elementwise_abs(be);
    const f32x2 s_a = aal - eab, s_b = abe - eab;
    // certainly inside: |a~| + (bound) <= 1/2, i.e. |a~| + eab <= 1 (eab = 1/2 + bound; the sum's own rounding is inside K's slack)
    const f32x2 in_a = aal + eab, in_b = abe + eab;
    uint32_t bits = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float m3, m;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m3) : "v"(s_lo[k]), "v"(s_hi[k]), "v"(s_a[k]));
        asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(m3), "v"(s_b[k]));
        // (the guard: |dn| > Ed2, false for a NaN: keep; the excess: positive denormal | positive normal)
        const bool guard = __builtin_fabsf(dn[k]) > ed2;
        const bool drop = guard && __builtin_amdgcn_classf(m, 0x180);
        const bool certain = guard && in_a[k] <= 1.0f && in_b[k] <= 1.0f; // (a NaN on either side: false, not certain)
        bits |= (drop ? 0u : (1u << k)) | (certain ? (0x100u << k) : 0u);
    }
    return bits;
}

// The exact f64 test the kernel used before (and the oracle's tight mode): kept as the yardstick for the test hook
RT_DEV bool box_miss_f64(const double lo[3], const double hi[3], V3 o, V3 d, double tmin, double tmax) {
    const double od[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
    for (int ax = 0; ax < 3; ++ax) {
        const double inv = 1.0 / dd[ax];
        double t0 = (lo[ax] - od[ax]) * inv, t1 = (hi[ax] - od[ax]) * inv;
        if (inv < 0.0) { const double tt = t0; t0 = t1; t1 = tt; }
        tmin = __builtin_fmax(t0, tmin);
        tmax = __builtin_fmin(t1, tmax);
        if (tmax <= tmin) return true;
    }
    return false;
}

} // namespace
