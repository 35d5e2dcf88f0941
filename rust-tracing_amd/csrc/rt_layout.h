// Device-side scene layout (private to librt_amd; the ABI only sees rt_scene_desc).
//
// The reference walks a pointer tree recursively, always left child first (src/bvh.rs:97-108), list items in
// order (src/hittable.rs:66-71).  Because that visiting order never depends on the ray, the whole object graph
// can be laid out once in that order ("preorder") and walked without a stack: every record carries the index
// of the record to continue with when its subtree is finished or its box is missed (`skip`); on a box hit the
// walk simply continues with the next record.  Nested BVHs are spliced in place, a HittableList of quads or
// spheres becomes one leaf record with a count, Translate/RotateY pairs become ENTER/EXIT records around their
// subtree, and a ConstantMedium becomes ENTER/EXIT records around its boundary's subtree (walked twice).
#pragma once
#include "rt_amd.h"
#include <cstdint>

namespace rtd {

enum NodeKind : uint32_t {
    NK_INNER = 0,        // bounding box only
    NK_SPHERES = 1,      // leaf: spheres [a, a+b)
    NK_QUADS = 2,        // leaf: quads   [a, a+b)
    NK_INST_ENTER = 3,   // a = instance index
    NK_INST_EXIT = 4,    // a = instance index
    NK_MEDIUM_ENTER = 5, // a = medium index
    NK_MEDIUM_EXIT = 6,  // a = medium index, b = first record of the boundary subtree
    NK_MEDIUM_SPHERE = 7, // a = medium index; the boundary is one Sphere: both boundary queries are solved in place
};
constexpr uint32_t NODE_KIND_MASK = 0xffu;
constexpr uint32_t NODE_NO_BBOX = 0x100u; // the reference performs no box test here (object inside a list /
                                          // instance / medium): always enter

// Build-time record (f64 box, as the reference holds it)
struct Node {
    double lo[3];
    double hi[3];
    uint32_t skip;
    uint32_t kind; // NodeKind | flags
    uint32_t a;
    uint32_t b;
};

// Device record, 32 bytes = two 16-byte loads per lane.  The box is held in f32, rounded OUTWARD (lo down, hi up),
// and tested with an error-bounded f32 slab test that can only err towards "hit" (rt_kernel.hip, box stage):
// a false positive costs a wasted visit and can never change a result (DESIGN.md "Box test"), a false negative
// cannot happen.  Primitives are always intersected in f64.
struct alignas(32) Node32 {
    float bx[2], by[2], bz[2]; // (lo, hi) per axis
    uint32_t skip;
    uint32_t packed;           // kind (4 bits) | NO_BBOX (bit 4) | count (7 bits, from bit 5) | a (20 bits, from bit 12)
};
static_assert(sizeof(Node32) == 32, "Node32 must be 32 bytes");
constexpr uint32_t N32_KIND_MASK = 0xfu, N32_NO_BBOX = 0x10u, N32_COUNT_SHIFT = 5, N32_COUNT_MASK = 0x7fu, N32_A_SHIFT = 12;
constexpr uint32_t N32_MAX_COUNT = 127, N32_MAX_A = (1u << 20) - 1u;

// ---- ordered layout (closest-hit queries whose result does not depend on the visiting order) ---------------
// A scene without a ConstantMedium consumes no random numbers inside hit(): the closest hit is then a pure
// minimum over the primitives (ties settled by the reference's scan order, rt_kernel.hip "ties"), so any tree
// and any visiting order give the reference's result.  For those scenes the compiler builds its own SAH tree
// per frame (world, and one per Translate/RotateY instance) and the kernel walks it nearest child first with
// a short per-lane stack.  One record = one inner node holding BOTH children's boxes and references.
enum OrderedKind : uint32_t { OK_INNER = 0, OK_SPHERES = 1, OK_QUADS = 2, OK_INSTANCE = 3, OK_EMPTY = 7 };
constexpr uint32_t OREF_KIND_SHIFT = 29, OREF_COUNT_SHIFT = 26, OREF_COUNT_MASK = 7u, OREF_INDEX_MASK = (1u << 26) - 1u;
constexpr uint32_t OREF_MAX_LEAF = 8;
struct alignas(64) ONode {
    float b0[6]; // child 0: x.lo, x.hi, y.lo, y.hi, z.lo, z.hi (f32, rounded outward)
    float b1[6]; // child 1
    uint32_t c[2]; // kind (3 bits) | count - 1 (3 bits) | index (26 bits): inner record / first primitive / instance
    uint32_t _pad[2];
};
static_assert(sizeof(ONode) == 64, "ONode must be 64 bytes");
// Wide records (rt_scene_options.wide): one record = up to FOUR children, made by pulling the grandchildren of a binary record up
// (rt_ordered.hpp widen).  A walk enters the nearest child whose box it hits and sets the RECORD aside with a mask of the children
// still to look at; when the entry's turn comes the record's boxes are tested again, against the interval as it has shrunk by then.
// An unused slot holds an empty reference and a box no ray enters (lo = +inf, hi = -inf).
struct alignas(128) ONode4 {
    float b[4][6]; // child k: x.lo, x.hi, y.lo, y.hi, z.lo, z.hi (f32, rounded outward)
    uint32_t c[4]; // as ONode::c
    uint32_t _pad[4];
};
static_assert(sizeof(ONode4) == 128, "ONode4 must be 128 bytes");
constexpr uint32_t WIDE_MAX_LDS_RECORDS = 4094; // 2-byte stack entries of the LDS kernels: 12 bits of record index + 4 bits of child mask
constexpr uint32_t ORDERED_MAX_STACK = 32; // per-lane stack entries the kernel provides at most
// The world frame of an ordered scene is a sequence of these, walked in order (rt_ordered.hpp): a tree over a run of
// medium-free objects, or a ConstantMedium — bounded by one Sphere (solved in place) or by a subtree with its own tree.
enum OrderedSeqKind : uint32_t { OSEQ_TREE = 0, OSEQ_MEDIUM_SPHERE = 1, OSEQ_MEDIUM = 2 };
struct alignas(16) OSeq {
    uint32_t kind;
    uint32_t a;      // TREE: root record; media: medium index
    uint32_t b;      // MEDIUM: root record of the boundary's tree
    uint32_t moving; // MEDIUM_SPHERE: the boundary sphere moves
    float box[6];    // (x.lo, x.hi, y.lo, y.hi, z.lo, z.hi), f32 rounded outward: what the step can touch
    uint32_t _pad[2];
    // everything a medium step needs, so that taking it touches no other table (the steps sit in the LDS)
    double center[3], radius;  // MEDIUM_SPHERE: the boundary
    double center_vec[3];
    double neg_inv_density;    // media
};
static_assert(sizeof(OSeq) == 112, "OSeq must be 112 bytes");
constexpr uint32_t ORDERED_MAX_STEPS = 64; // steps of a world frame's sequence the kernel keeps in the LDS

// 64 bytes
struct alignas(64) Sphere {
    double center[3];
    double radius;
    double center_vec[3];
    uint32_t material;
    uint32_t seq_moving; // bit 0: the sphere moves; bits 1..31: seq, the primitive's position in the reference's scan order
};
static_assert(sizeof(Sphere) == 64, "Sphere must be 64 bytes");

// 144 bytes; the plane (normal, d) comes first: most tests end after reading only that
struct alignas(16) Quad {
    double normal[3];
    double d;
    double q[3];
    double w[3];
    double u[3];
    double v[3];
    uint32_t material;
    uint32_t seq; // the primitive's position in the reference's scan order (spheres and quads share one numbering)
    uint32_t _pad[2];
};
static_assert(sizeof(Quad) == 144, "Quad must be 144 bytes");

// Conservative f32 filter records for the quads of a flat leaf (rt_device_scene.h quad_pair_keep): one record per quad index i, holding
// the pair (i, i + 1) interleaved — (value of quad i, value of quad i + 1) — so that one packed f32 instruction serves both; the
// quad stage reads the records of its leaf's first, third, fifth ... quad.  Built by rt_qfilt.hpp (host), LDS-resident scenes only.
//   v[0..2] normal  v[3] d  v[4..6] A = v x w  v[7] A.q + 1/2  v[8..10] B = w x u  v[11] B.q + 1/2
// so that alpha - 1/2 = A . p - v[7] and beta - 1/2 = B . p - v[11] at the hit point p (src/quad.rs:117-127: alpha = w . (php x v) = php . (v x w)).
// A quad that is never to be filtered has a NaN normal (every comparison against it is false: kept).  The error terms that go
// with these values are the pair's: the larger of its two quads' (a leaf's quads are of one size).
struct alignas(16) QFiltPair {
    float v[12][2];
    float n1c;  // 14 * 2^-24 * |normal|_1
    float dc;   // 14 * 2^-24 * |d|
    float a1;   // max(|A|_1, |B|_1)
    float ka;   // 1/2 + K_alpha (resp. K_beta)
};
static_assert(sizeof(QFiltPair) == 112, "QFiltPair must be 112 bytes");

// One frame change: optional Translate (applied to the ray first) then optional RotateY, i.e. the reference's
// Translate::hit -> RotateY::hit nesting (src/hittable.rs:96-106,:159-188).  parent = enclosing instance or -1.
struct alignas(64) Instance {
    double offset[3];
    double sin_theta, cos_theta;
    int32_t parent;
    uint32_t flags; // bit 0: has translate, bit 1: has rotate
    uint32_t depth; // 0 for an instance in the world frame
    uint32_t root;  // ordered layout: the root record of the tree over this frame's contents
    uint32_t start_ref; // ordered layout: != 0: that root holds nothing but ONE leaf (a box's six faces): its reference — a walk that enters
                        // the frame starts in the leaf's primitive stage, not with a visit of a record that has one thing to say
    uint32_t _pad;
    float start_box[6]; // ... and that leaf's box in the frame's own coordinates (x.lo, x.hi, y.lo, y.hi, z.lo, z.hi; f32, rounded outward): the
                        // walk met the instance through its box in the PARENT's frame — for a rotated box a fifth wider than the box itself
    uint32_t _pad2[10];
};
static_assert(sizeof(Instance) == 128, "Instance must be 128 bytes");
constexpr uint32_t INST_TRANSLATE = 1u, INST_ROTATE = 2u;

struct alignas(16) Medium {
    double neg_inv_density;
    uint32_t phase_material;
    uint32_t first_node; // first record of the boundary subtree (the second boundary query restarts there);
                         // for NK_MEDIUM_SPHERE: index of the boundary sphere
};

// Texels are stored in 8x8 tiles (a tile = 192 consecutive bytes, tiles row-major, the image padded to whole tiles): the
// lookups of neighbouring hit points — the reference indexes a row-major image (src/texture.rs:83-92) — land in one or two
// cache lines instead of one line per row.  texel (i, j) lives at offset + ((j / 8 * tiles_x + i / 8) * 64 + (j % 8) * 8 + i % 8) * 3.
constexpr uint32_t TEXEL_TILE = 8;
struct alignas(16) ImageRef {
    uint32_t width, height;
    uint32_t tiles_x; // tiles per tile row
    uint32_t _pad;
    uint64_t offset;  // byte offset of the first tile in the texel pool
    uint64_t _pad2;
};
static_assert(sizeof(ImageRef) == 32, "ImageRef must be 32 bytes");
inline uint64_t texel_index(uint32_t tiles_x, uint32_t i, uint32_t j) {
    return ((uint64_t)(j / TEXEL_TILE) * tiles_x + i / TEXEL_TILE) * (TEXEL_TILE * TEXEL_TILE) + (j % TEXEL_TILE) * TEXEL_TILE + i % TEXEL_TILE;
}

// prim id stored with the closest hit: kind in the top 2 bits
constexpr uint32_t PRIM_NONE = 0xffffffffu;
constexpr uint32_t PRIM_SPHERE = 0u << 30, PRIM_QUAD = 1u << 30, PRIM_MEDIUM = 2u << 30;
constexpr uint32_t PRIM_KIND_MASK = 3u << 30, PRIM_INDEX_MASK = ~PRIM_KIND_MASK;

constexpr uint32_t MAX_INSTANCE_DEPTH = 4;

} // namespace rtd
