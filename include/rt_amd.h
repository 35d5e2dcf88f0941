/*
 * rt_amd.h — C ABI of the MI355X path-tracing renderer (librt_amd.so).
 *
 * This is the drop-in boundary for the reference's per-pixel render loop.  The reference
 * (Husenap/rust-tracing) has no FFI; its seam is
 *
 *     pub fn render(camera: Arc<Camera>, world: Arc<dyn Hittable>, output_file_name: String)
 *                                                              (src/renderer.rs:12, called at src/main.rs:665)
 *
 * and, beneath it, the trait objects dyn Hittable (src/hittable.rs:45-48), dyn Material
 * (src/material.rs:11-16) and dyn Texture (src/texture.rs:12-14).  Trait objects cannot cross to a GPU,
 * so the boundary carries the same object graph as plain-old-data: every Rust struct that implements one
 * of the three traits becomes one POD record below (same fields, same meaning), and every
 * Arc<dyn Hittable/Material/Texture> becomes an index.  A Rust host produces these records by walking its
 * own objects (one `describe()` method per trait); nothing here is device-specific.
 * How the library lays the scene out in HBM/LDS is private to the library.
 *
 * Everything the renderer computes is f64 (reference: `pub type FP = f64`, src/common.rs:1).
 *
 * Conventions
 *   - all functions return 0 on success and a negative rt_status on failure; they never abort or unwind;
 *     rt_last_error() returns a thread-local message for the last failure on the calling thread.
 *   - every pointer passed in is borrowed for the duration of the call only; the library copies.
 *   - rt_scene handles are immutable after creation and may be rendered from concurrently; two
 *     concurrent renders must not share an output buffer.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 2 /* 2: per-path RomuDuoJr streams (the normative RNG changed: frames differ from abi 1), scene options, gather */

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID_ARGUMENT = -1, /* null pointer, bad index, unsupported graph shape */
    RT_ERR_NO_DEVICE = -2,        /* no HIP device / device index out of range          */
    RT_ERR_HIP = -3,              /* a HIP runtime call failed (message has the HIP error string) */
    RT_ERR_OUT_OF_MEMORY = -4,
    RT_ERR_UNSUPPORTED = -5,      /* graph is valid but outside what the device path handles */
    RT_ERR_COMM = -6              /* RCCL is not available or one of its calls failed (message has its error string) */
} rt_status;

/* Vec3 / Point3 / Color (src/vec3.rs:8-16). */
typedef struct rt_vec3 { double x, y, z; } rt_vec3;

/* AABB = three Intervals (src/aabb.rs:9-14, src/interval.rs:5-9). */
typedef struct rt_aabb { double lo[3]; double hi[3]; } rt_aabb;

/* ---- Hittable graph -------------------------------------------------------------------------------
 * rt_ref stands in for Arc<dyn Hittable>: (kind, index into the array of that kind). */
typedef enum rt_hittable_kind {
    RT_HITTABLE_NONE = 0,
    RT_HITTABLE_SPHERE = 1,          /* src/sphere.rs:13-20            */
    RT_HITTABLE_QUAD = 2,            /* src/quad.rs:11-20              */
    RT_HITTABLE_LIST = 3,            /* HittableList, src/hittable.rs:50-54 */
    RT_HITTABLE_TRANSLATE = 4,       /* src/hittable.rs:81-85          */
    RT_HITTABLE_ROTATE_Y = 5,        /* src/hittable.rs:113-118        */
    RT_HITTABLE_BVH = 6,             /* BVHNode, src/bvh.rs:12-14      */
    RT_HITTABLE_CONSTANT_MEDIUM = 7  /* src/constant_medium.rs:14-18   */
} rt_hittable_kind;

typedef struct rt_ref { int32_t kind; int32_t index; } rt_ref;

/* Sphere (src/sphere.rs:13-20).  `center_vec`/`is_moving` as set by with_target (src/sphere.rs:34-46).
 * The bounding box lives in the BVH leaf that holds the sphere (src/bvh.rs:44), not here. */
typedef struct rt_sphere {
    rt_vec3 center;
    double radius;
    rt_vec3 center_vec;
    int32_t is_moving;
    int32_t material;
} rt_sphere;

/* Quad (src/quad.rs:11-20) with the derived fields exactly as Quad::new computes them
 * (src/quad.rs:24-27): normal = normalize(u x v), d = normal . q, w = n / (n . n). */
typedef struct rt_quad {
    rt_vec3 q, u, v, w, normal;
    double d;
    int32_t material;
    int32_t _pad;
} rt_quad;

/* HittableList (src/hittable.rs:50-54): objects = list_items[first .. first+count). */
typedef struct rt_list { int32_t first; int32_t count; } rt_list;

/* Translate (src/hittable.rs:81-85) and RotateY (src/hittable.rs:113-118; sin/cos of the angle as
 * RotateY::new stores them, src/hittable.rs:121-123). */
typedef struct rt_translate { rt_ref object; rt_vec3 offset; } rt_translate;
typedef struct rt_rotate_y { rt_ref object; double sin_theta; double cos_theta; } rt_rotate_y;

/* One `(Node, AABB)` pair of the BVH (src/bvh.rs:16-19).  is_leaf: Node::Leaf(object) else
 * Node::Branch(left, right) with left/right indexing bvh_nodes. */
typedef struct rt_bvh_node {
    rt_aabb bbox;
    int32_t is_leaf;
    int32_t left, right;
    rt_ref object;
    int32_t _pad;
} rt_bvh_node;

/* BVHNode (src/bvh.rs:12-14): root indexes bvh_nodes. */
typedef struct rt_bvh { int32_t root; int32_t _pad; } rt_bvh;

/* ConstantMedium (src/constant_medium.rs:14-18); phase_material indexes materials (an ISOTROPIC one). */
typedef struct rt_constant_medium {
    rt_ref boundary;
    double neg_inv_density;
    int32_t phase_material;
    int32_t _pad;
} rt_constant_medium;

/* ---- Materials (src/material.rs) ------------------------------------------------------------------ */
typedef enum rt_material_kind {
    RT_MATERIAL_LAMBERTIAN = 1,    /* :18-42   texture = albedo                 */
    RT_MATERIAL_METAL = 2,         /* :44-64   albedo, fuzz (not clamped)        */
    RT_MATERIAL_DIELECTRIC = 3,    /* :66-104  ir                                */
    RT_MATERIAL_DIFFUSE_LIGHT = 4, /* :106-122 texture = emit                    */
    RT_MATERIAL_ISOTROPIC = 5      /* :124-138 texture = albedo                  */
} rt_material_kind;

typedef struct rt_material {
    int32_t kind;
    int32_t texture; /* index into textures, or -1 */
    rt_vec3 albedo;  /* Metal */
    double fuzz;     /* Metal */
    double ir;       /* Dielectric */
} rt_material;

/* ---- Textures (src/texture.rs) -------------------------------------------------------------------- */
typedef enum rt_texture_kind {
    RT_TEXTURE_SOLID = 1,   /* :17-37   color                                   */
    RT_TEXTURE_CHECKER = 2, /* :39-70   inv_scale, even, odd (texture indices)  */
    RT_TEXTURE_IMAGE = 3,   /* :72-93   image index                             */
    RT_TEXTURE_NOISE = 4    /* :95-111  perlin index, scale                     */
} rt_texture_kind;

typedef struct rt_texture {
    int32_t kind;
    int32_t even, odd; /* Checker */
    int32_t image;     /* Image   */
    int32_t perlin;    /* Noise   */
    int32_t _pad;
    rt_vec3 color;     /* Solid   */
    double inv_scale;  /* Checker: 1/scale as CheckerTexture::new stores it (src/texture.rs:46) */
    double scale;      /* Noise   */
} rt_texture;

/* Perlin tables (src/perlin.rs:7-13): 256 un-normalised gradient vectors and three permutations. */
#define RT_PERLIN_POINTS 256
typedef struct rt_perlin {
    rt_vec3 ranvec[RT_PERLIN_POINTS];
    int32_t perm_x[RT_PERLIN_POINTS];
    int32_t perm_y[RT_PERLIN_POINTS];
    int32_t perm_z[RT_PERLIN_POINTS];
} rt_perlin;

/* Decoded image of an ImageTexture (src/texture.rs:72-81): RGB8, row-major, row 0 at the top, as
 * image::DynamicImage::get_pixel(i, j) addresses it (src/texture.rs:89). */
typedef struct rt_image {
    int32_t width, height;
    const uint8_t *rgb;
} rt_image;

/* ---- Scene ---------------------------------------------------------------------------------------- */
typedef struct rt_scene_desc {
    uint32_t abi_version; /* RT_ABI_VERSION */
    uint32_t _pad;
    rt_ref world;         /* what main hands to render(): the top-level BVHNode (src/main.rs:659-665) */

    int32_t n_spheres, n_quads, n_lists, n_list_items, n_translates, n_rotates, n_bvh_nodes, n_bvhs,
        n_media, n_materials, n_textures, n_perlins, n_images;
    int32_t _pad2;

    const rt_sphere *spheres;
    const rt_quad *quads;
    const rt_list *lists;
    const rt_ref *list_items;
    const rt_translate *translates;
    const rt_rotate_y *rotates;
    const rt_bvh_node *bvh_nodes;
    const rt_bvh *bvhs;
    const rt_constant_medium *media;
    const rt_material *materials;
    const rt_texture *textures;
    const rt_perlin *perlins;
    const rt_image *images;
} rt_scene_desc;

/* Camera: the twelve fields of `pub struct Camera` after Camera::new (src/camera.rs:38-51, :54-110). */
typedef struct rt_camera {
    int32_t image_width, image_height;
    int32_t samples_per_pixel, max_depth;
    rt_vec3 background;
    rt_vec3 center;
    rt_vec3 pixel00_loc;
    rt_vec3 pixel_delta_u, pixel_delta_v;
    double defocus_angle;
    rt_vec3 defocus_disk_u, defocus_disk_v;
} rt_camera;

/* Output layouts. */
typedef enum rt_out_layout {
    /* out[(j*w + i)*3 + c]: exactly the reference's Vec<Color> (src/renderer.rs:32-33,:49).  Only the
     * pixels of this shard's tiles are written (all pixels when shard_count == 1). */
    RT_OUT_FRAME = 0,
    /* out[((lt*tile_h + ty)*tile_w + tx)*3 + c], lt = local tile number: tile k of the frame
     * (k = tile_row*tiles_per_row + tile_col) belongs to shard k % shard_count and is that shard's local
     * tile k / shard_count.  Edge tiles are padded to tile_w x tile_h; padding is written as 0.
     * This is the buffer each GPU contributes to the frame-end gather. */
    RT_OUT_TILES = 1
} rt_out_layout;

#define RT_TILE_W 8
#define RT_TILE_H 8

typedef struct rt_render_params {
    uint64_t seed;        /* render seed: keys the per-(pixel, sample) random stream (see RNG below) */
    int32_t sample_begin; /* samples [sample_begin, sample_end) of every pixel are traced and summed in order */
    int32_t sample_end;   /* <= 0: camera.samples_per_pixel */
    int32_t max_depth;    /* <= 0: camera.max_depth */
    int32_t accumulate;   /* 0: out = sum over the range; 1: out += (continues a previous range bit-exactly:
                             the per-pixel sum is a sequential f64 += over samples, src/renderer.rs:35-40) */
    int32_t shard_index;  /* this caller renders the tiles k with k % shard_count == shard_index */
    int32_t shard_count;  /* <= 0: 1 */
    int32_t out_layout;   /* rt_out_layout */
    int32_t device;       /* HIP device ordinal for rt_render (host-buffer form); ignored by rt_render_device,
                             which runs on the scene's device */
} rt_render_params;

/* Work counters of one render call (instrumented kernel; summed over all traced samples). */
typedef struct rt_counters {
    uint64_t samples;       /* camera paths traced                                          */
    uint64_t rays;          /* closest-hit queries issued by ray_color (src/renderer.rs:144) */
    uint64_t node_visits;   /* bounding-box tests; ordered walk: records visited (two boxes each) */
    uint64_t sphere_tests;
    uint64_t quad_tests;
    uint64_t medium_visits; /* ConstantMedium::hit entries (src/constant_medium.rs:34)       */
    uint64_t rng_draws;
    uint64_t noise_evals;   /* NoiseTexture::value calls                                     */
    uint64_t image_lookups; /* ImageTexture::value calls                                     */
    uint64_t instance_enters;
} rt_counters;

typedef struct rt_scene rt_scene; /* opaque; owned by the library */

/* Number of HIP devices visible to the library (0 if none; never an error). */
int rt_device_count(void);

/* Validates `desc`, compiles it into the device layout and uploads it to HIP device `device`.
 * Replaces: the Arc<dyn Hittable> world + Arc<Camera> hand-off at src/main.rs:659-665. */
int rt_scene_create(const rt_scene_desc *desc, int device, rt_scene **out_scene);
void rt_scene_destroy(rt_scene *scene);

/* Per-scene options (all optional: rt_scene_options_init fills the defaults; results never depend on them, only speed and
 * memory use do).  They replace the process-wide setters that round 1 used for this. */
typedef enum rt_walk {
    RT_WALK_DEFAULT = -1,        /* the library's default (RT_WALK_AUTO unless RT_ORDERED is set in the environment) */
    RT_WALK_REFERENCE_ORDER = 0, /* the reference's tree in the reference's order (src/bvh.rs:97-108), stackless */
    RT_WALK_AUTO = 1,            /* the library's own trees where they measured faster (DESIGN.md "Ordered layout") */
    RT_WALK_OWN_TREES = 2        /* the library's own trees wherever the scene allows it */
} rt_walk;
typedef struct rt_scene_options {
    uint32_t struct_size;        /* sizeof(rt_scene_options) as the CALLER was compiled: lets the struct grow compatibly — the library
                                    reads that many bytes and takes its defaults for every field beyond them */
    int32_t walk;                /* rt_walk */
    int32_t leaf_max;            /* own trees: primitives per leaf at most (<= 0: default) */
    int32_t refit;               /* -1 default (on); 0: keep the reference's boxes; 1: shrink them to the geometry */
    int32_t use_lds;             /* -1 default (on); 0: gather the scene from global memory even if it fits the LDS */
    int32_t th_prim, th_other, th_shade, th_box, th_new; /* scheduler thresholds in 64ths of a wave's live lanes; -1: preset */
    int64_t sample_buffer_bytes; /* per-(scene, stream) sample buffer at most; <= 0: default (2 GiB).  A frame that needs
                                    more is rendered in several launches over sample sub-ranges (same result). */
    int32_t reserved_pool;       /* ignored (rounds 2-3: the opt-in pool kernel, measured at 0.55x and removed — DESIGN_HISTORY.md); keeps
                                    the offsets of the fields below */
    int32_t flat_max;            /* own trees: a frame of at most this many primitives of one kind keeps them in one leaf under its root
                                    (0: off; -1: default, 8) */
    /* how the walk of the library's own trees starts and ends its queries (each -1: default) */
    int32_t start_shortcut;      /* 1: a query starts with the primitives of a leaf under the root that spans the scene */
    int32_t defer_instances;     /* 1: the world frame's Translate / RotateY subtrees are walked after the world's own tree */
    int32_t seq_lookahead;       /* 1: scenes with media: a query that cannot reach a later step of the world's sequence ends it early */
    int32_t slow_min, slow_age;  /* hits on a noise texture wait in the shade stage for slow_min of their kind, at most slow_age shade
                                    rounds (slow_min 1: nobody waits) */
    int32_t wide;                /* own trees: 1: records of four children, 0: of two; -1: default — four for scenes of 64 primitives or
                                    more (DESIGN.md "Wide records") */
    int32_t quad_filter;         /* -1 default (on); 0: every quad of a multi-quad leaf gets the exact test at once, without the conservative
                                    f32 filter in front of it (DESIGN.md "Quad filter") */
    int32_t medium_first;        /* -1 default (on); 0: off.  Own trees, scenes with media: a ray that starts inside the ball of a sphere-bounded
                                    medium makes that medium's draw before the tree in front of it is walked, and walks the tree no further
                                    than the draw's candidate (same results) */
} rt_scene_options;
/* Fills the defaults.  rt_scene_options_init writes sizeof(rt_scene_options) of THIS header: caller and library must have been built
 * from the same header.  A caller that may meet a newer library calls rt_scene_options_init_sized(&o, sizeof o) instead: only
 * that many bytes are written, struct_size is set to it, and rt_scene_create_ex treats the fields beyond it as default. */
void rt_scene_options_init(rt_scene_options *options);
int rt_scene_options_init_sized(rt_scene_options *options, uint32_t struct_size);
int rt_scene_create_ex(const rt_scene_desc *desc, int device, const rt_scene_options *options /* NULL: defaults */,
                       rt_scene **out_scene);

/* Bytes of device memory the compiled scene occupies, by part (for DESIGN.md's layout table). */
typedef struct rt_scene_stats {
    uint64_t node_bytes, sphere_bytes, quad_bytes, instance_bytes, medium_bytes, material_bytes,
        texture_bytes, perlin_bytes, image_bytes;
    uint32_t n_nodes, n_spheres, n_quads, n_instances, n_media, max_instance_depth;
    uint32_t lds_nodes, lds_bytes;      /* records resident in the LDS; LDS bytes a workgroup uses (image + stacks) */
    uint32_t ordered, stack_entries;    /* 1: the scene is walked through the library's own trees (DESIGN.md "Ordered walk") */
} rt_scene_stats;
int rt_scene_get_stats(const rt_scene *scene, rt_scene_stats *out);

/* The render loop (replaces src/renderer.rs:26-49, ray_color :139-155 and everything they call).
 * `out_rgb_sum` is a HOST buffer of rt_out_size() doubles holding per-pixel SUMS over the sample range —
 * the caller divides by spp and applies color_to_rgb exactly as src/renderer.rs:55-58 does.
 * Blocking; includes the device->host copy. */
int rt_render(const rt_scene *scene, const rt_camera *camera, const rt_render_params *params,
              double *out_rgb_sum);

/* Same, but `d_out_rgb_sum` is DEVICE memory on the scene's device and the work is enqueued on
 * `hip_stream` (a hipStream_t; NULL = the null stream) without synchronising. */
int rt_render_device(const rt_scene *scene, const rt_camera *camera, const rt_render_params *params,
                     double *d_out_rgb_sum, void *hip_stream);

/* As rt_render_device with the instrumented kernel; blocks until done and fills `out_counters`
 * (results in d_out_rgb_sum are identical to the un-instrumented kernel's). */
int rt_render_device_counted(const rt_scene *scene, const rt_camera *camera, const rt_render_params *params,
                             double *d_out_rgb_sum, void *hip_stream, rt_counters *out_counters);

/* Number of doubles an output buffer needs for (width, height, layout, shard_index, shard_count). */
int64_t rt_out_size(int32_t width, int32_t height, int32_t out_layout, int32_t shard_index,
                    int32_t shard_count);

/* Frame-end reassembly on the device: `d_gathered` holds shard 0's RT_OUT_TILES buffer, then shard 1's, ...
 * each padded to rt_out_size(w, h, RT_OUT_TILES, 0, shard_count) doubles (shard 0 has the most tiles);
 * writes the RT_OUT_FRAME image into d_frame.  Enqueued on hip_stream. */
int rt_tiles_to_frame_device(int32_t width, int32_t height, int32_t shard_count, const double *d_gathered,
                             double *d_frame, void *hip_stream);

/* Output stage on the device (reference: color_to_rgb(c / spp), src/renderer.rs:55-58, src/color.rs:12-19):
 * rgb8[k] = (u8)(256 * clamp(gamma(sum[k] * (1/spp)), 0, 0.999)) for each of the n_values channel sums, with gamma(x) =
 * x^(1/2.2) by the fixed algorithm of "Normative definitions" below — the same code the host library runs
 * (rth_resolve_rgb8), so device and host bytes are identical.  Elementwise: works on a frame (3 * w * h values) and on a
 * shard's RT_OUT_TILES buffer alike.  Enqueued on hip_stream. */
int rt_resolve_rgb8_device(int32_t width, int32_t height, int32_t spp, const double *d_frame_sum,
                           uint8_t *d_rgb8, void *hip_stream);
int rt_resolve_rgb8_values_device(int64_t n_values, int32_t spp, const double *d_sum, uint8_t *d_rgb8, void *hip_stream);
/* rt_tiles_to_frame_device for gathered RGB8 tile buffers (one byte per value instead of one double). */
int rt_tiles_to_frame_rgb8_device(int32_t width, int32_t height, int32_t shard_count, const uint8_t *d_gathered,
                                  uint8_t *d_frame, void *hip_stream);

/* Device memory for hosts that do not link the HIP runtime themselves (the Rust binding, host/renderer.cpp): the buffers
 * rt_render_device, the gather and the frame-end kernels work on.  rt_device_download copies to host memory and returns when
 * the copy — and everything enqueued on hip_stream before it — is done. */
int rt_device_malloc(int device, int64_t bytes, void **out_device_ptr);
int rt_device_free(int device, void *device_ptr);
int rt_device_download(int device, void *dst_host, const void *src_device, int64_t bytes, void *hip_stream);

/* ---- frame-end gather over RCCL / xGMI ----------------------------------------------------------
 * One communicator handle per rank (= per GPU).  Create it either
 *   - one process per GPU: rank 0 calls rt_comm_get_unique_id and hands the 128 bytes to the other ranks by whatever
 *     channel the host program has (MPI, a file, a socket); every rank then calls rt_comm_create;
 *   - one process, several GPUs: rt_comm_create_all (ncclCommInitAll), then one thread per device;
 *   - or wrap a communicator the host already owns: rt_comm_adopt(ncclComm_t).
 * rt_gather_tiles_device brings every rank's RT_OUT_TILES buffer (f64 sums: elem_bytes 8; resolved RGB8: elem_bytes 1)
 * to `root` in ONE grouped exchange (N - 1 receives on the root, one send per other rank, each on its own xGMI link) into
 * d_gathered = [rank 0's tiles | rank 1's | ...], every slot rt_out_size(w, h, RT_OUT_TILES, 0, N) elements long — the
 * layout rt_tiles_to_frame_device / rt_tiles_to_frame_rgb8_device take.  d_gathered is only read on the root.
 * Enqueued on hip_stream; RCCL is loaded on first use (RT_ERR_COMM if it is not installed). */
#define RT_COMM_ID_BYTES 128
typedef struct rt_comm rt_comm; /* opaque */
int rt_comm_get_unique_id(uint8_t out_id[RT_COMM_ID_BYTES]);
int rt_comm_create(const uint8_t id[RT_COMM_ID_BYTES], int rank, int n_ranks, int device, rt_comm **out_comm);
int rt_comm_create_all(int n_devices, const int *devices /* NULL: 0 .. n-1 */, rt_comm **out_comms /* [n_devices] */);
int rt_comm_adopt(void *nccl_comm, int device, rt_comm **out_comm);
void rt_comm_destroy(rt_comm *comm);
int rt_comm_rank(const rt_comm *comm);
int rt_comm_size(const rt_comm *comm);
int rt_gather_tiles_device(rt_comm *comm, int32_t width, int32_t height, int32_t elem_bytes, const void *d_tiles,
                           void *d_gathered, int root, void *hip_stream);

const char *rt_last_error(void);
const char *rt_version(void);

/* ---- Normative definitions shared by every implementation of this ABI ------------------------------
 *
 * RNG.  The reference draws from rand 0.8.5's thread_rng(), which is OS-seeded and cannot be reproduced
 * (Cargo.toml:10; call sites src/vec3.rs:43-50,:80-81, src/camera.rs:123,:134-135, src/material.rs:94,
 * src/constant_medium.rs:48).  This ABI replaces it by a seeded generator with one stream per camera path, so
 * that an image is a pure function of (scene, camera, seed):
 *
 *     mix64(z): z ^= z >> 30; z *= 0xBF58476D1CE4E5B9; z ^= z >> 27; z *= 0x94D049BB133111EB; z ^= z >> 31
 *     key(seed, pixel, sample) = mix64( mix64(seed + 0x9E3779B97F4A7C15) ^ ((u64)pixel << 32 | (u32)sample) )
 *     stream of that path (RomuDuoJr): x = key, y = mix64(key + 0x9E3779B97F4A7C15); every draw returns x and then
 *         (x, y) <- (0xD3833E804F4C574B * y,  rotl64(y - x, 27))                  (all arithmetic mod 2^64)
 *     (round 1 used one splitmix64 output per draw: two 64-bit multiplies; this form needs one — DESIGN.md "RNG")
 *
 * pixel = j * image_width + i (src/renderer.rs:32-33); sample counts from 0.  The draws of one camera path
 * are consumed in the reference's program order (camera px, py, [disk x, y]*, time; then per bounce the
 * draws of ConstantMedium::hit in traversal order, then the material's).  From a 64-bit draw x:
 *
 *     random::<f64>()        = (x >> 11) * 2^-53                               (rand 0.8 Standard for f64)
 *     gen_range(lo..hi)      = (f64::from_bits((x >> 12) | 0x3FF0000000000000) - 1.0) * (hi - lo) + lo
 *                                                                               (rand 0.8 UniformFloat::sample_single)
 *
 * Arithmetic.  IEEE-754 binary64, round-to-nearest-even, no fused multiply-add contraction, operations in
 * the reference's source order.  Square root and division are correctly rounded.  The five transcendental
 * functions on the path — ln (src/constant_medium.rs:48), sin (src/texture.rs:109), acos and atan2
 * (src/sphere.rs:49-50), x^5 (src/material.rs:77) — are computed by the fixed algorithms documented in
 * DESIGN.md ("Device math") so that CPU and GPU implementations agree bit for bit; they are within 2 ulp
 * of a correctly rounded result (x^5: 3 ulp).  gamma_to_linear on texels (src/color.rs:8-10,:21-26) is the
 * host libm's pow(c/255, 2.2).
 */

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
