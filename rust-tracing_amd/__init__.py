"""ctypes bindings of the two shared libraries of this package.

  librt_amd.so   csrc/  — the HIP renderer behind the C ABI of include/rt_amd.h (the product)
  librt_host.so  host/  — the reference's host-side surface (scene builders, BVH build, camera, output stage)
                          behind include/rt_host.h

Python is plumbing here: tests and bench.py use it to hold device memory (torch), to launch one process per
GPU (torch.distributed) and to call the C ABI.  Nothing in this module computes pixels, and there is no
fallback: if librt_amd.so is missing or has no GPU to run on, calls raise.

The package directory is named ``rust-tracing_amd`` (not importable with a plain ``import`` statement); load it
with ``importlib.import_module("rust-tracing_amd")``.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_PKG_DIR = Path(__file__).resolve().parent
LIB_DIR = _PKG_DIR / "lib"

RT_ABI_VERSION = 2
RT_TILE_W = 8
RT_TILE_H = 8
RT_OUT_FRAME = 0
RT_OUT_TILES = 1

# rt_hittable_kind
RT_HITTABLE_NONE, RT_HITTABLE_SPHERE, RT_HITTABLE_QUAD, RT_HITTABLE_LIST, RT_HITTABLE_TRANSLATE, \
    RT_HITTABLE_ROTATE_Y, RT_HITTABLE_BVH, RT_HITTABLE_CONSTANT_MEDIUM = range(8)
# rt_material_kind
RT_MATERIAL_LAMBERTIAN, RT_MATERIAL_METAL, RT_MATERIAL_DIELECTRIC, RT_MATERIAL_DIFFUSE_LIGHT, \
    RT_MATERIAL_ISOTROPIC = range(1, 6)
# rt_texture_kind
RT_TEXTURE_SOLID, RT_TEXTURE_CHECKER, RT_TEXTURE_IMAGE, RT_TEXTURE_NOISE = range(1, 5)


class Vec3(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double)]

    def tuple(self):
        return (self.x, self.y, self.z)


class Aabb(C.Structure):
    _fields_ = [("lo", C.c_double * 3), ("hi", C.c_double * 3)]


class Ref(C.Structure):
    _fields_ = [("kind", C.c_int32), ("index", C.c_int32)]


class Sphere(C.Structure):
    _fields_ = [("center", Vec3), ("radius", C.c_double), ("center_vec", Vec3), ("is_moving", C.c_int32),
                ("material", C.c_int32)]


class Quad(C.Structure):
    _fields_ = [("q", Vec3), ("u", Vec3), ("v", Vec3), ("w", Vec3), ("normal", Vec3), ("d", C.c_double),
                ("material", C.c_int32), ("_pad", C.c_int32)]


class List(C.Structure):
    _fields_ = [("first", C.c_int32), ("count", C.c_int32)]


class Translate(C.Structure):
    _fields_ = [("object", Ref), ("offset", Vec3)]


class RotateY(C.Structure):
    _fields_ = [("object", Ref), ("sin_theta", C.c_double), ("cos_theta", C.c_double)]


class BvhNode(C.Structure):
    _fields_ = [("bbox", Aabb), ("is_leaf", C.c_int32), ("left", C.c_int32), ("right", C.c_int32),
                ("object", Ref), ("_pad", C.c_int32)]


class Bvh(C.Structure):
    _fields_ = [("root", C.c_int32), ("_pad", C.c_int32)]


class ConstantMedium(C.Structure):
    _fields_ = [("boundary", Ref), ("neg_inv_density", C.c_double), ("phase_material", C.c_int32),
                ("_pad", C.c_int32)]


class Material(C.Structure):
    _fields_ = [("kind", C.c_int32), ("texture", C.c_int32), ("albedo", Vec3), ("fuzz", C.c_double),
                ("ir", C.c_double)]


class Texture(C.Structure):
    _fields_ = [("kind", C.c_int32), ("even", C.c_int32), ("odd", C.c_int32), ("image", C.c_int32),
                ("perlin", C.c_int32), ("_pad", C.c_int32), ("color", Vec3), ("inv_scale", C.c_double),
                ("scale", C.c_double)]


class Perlin(C.Structure):
    _fields_ = [("ranvec", Vec3 * 256), ("perm_x", C.c_int32 * 256), ("perm_y", C.c_int32 * 256),
                ("perm_z", C.c_int32 * 256)]


class Image(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgb", C.POINTER(C.c_uint8))]


class SceneDesc(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("_pad", C.c_uint32), ("world", Ref),
                ("n_spheres", C.c_int32), ("n_quads", C.c_int32), ("n_lists", C.c_int32),
                ("n_list_items", C.c_int32), ("n_translates", C.c_int32), ("n_rotates", C.c_int32),
                ("n_bvh_nodes", C.c_int32), ("n_bvhs", C.c_int32), ("n_media", C.c_int32),
                ("n_materials", C.c_int32), ("n_textures", C.c_int32), ("n_perlins", C.c_int32),
                ("n_images", C.c_int32), ("_pad2", C.c_int32),
                ("spheres", C.POINTER(Sphere)), ("quads", C.POINTER(Quad)), ("lists", C.POINTER(List)),
                ("list_items", C.POINTER(Ref)), ("translates", C.POINTER(Translate)),
                ("rotates", C.POINTER(RotateY)), ("bvh_nodes", C.POINTER(BvhNode)), ("bvhs", C.POINTER(Bvh)),
                ("media", C.POINTER(ConstantMedium)), ("materials", C.POINTER(Material)),
                ("textures", C.POINTER(Texture)), ("perlins", C.POINTER(Perlin)), ("images", C.POINTER(Image))]


class Camera(C.Structure):
    _fields_ = [("image_width", C.c_int32), ("image_height", C.c_int32), ("samples_per_pixel", C.c_int32),
                ("max_depth", C.c_int32), ("background", Vec3), ("center", Vec3), ("pixel00_loc", Vec3),
                ("pixel_delta_u", Vec3), ("pixel_delta_v", Vec3), ("defocus_angle", C.c_double),
                ("defocus_disk_u", Vec3), ("defocus_disk_v", Vec3)]


class RenderParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("sample_begin", C.c_int32), ("sample_end", C.c_int32),
                ("max_depth", C.c_int32), ("accumulate", C.c_int32), ("shard_index", C.c_int32),
                ("shard_count", C.c_int32), ("out_layout", C.c_int32), ("device", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples", "rays", "node_visits", "sphere_tests", "quad_tests",
                                          "medium_visits", "rng_draws", "noise_evals", "image_lookups",
                                          "instance_enters")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class SceneStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("node_bytes", "sphere_bytes", "quad_bytes", "instance_bytes",
                                          "medium_bytes", "material_bytes", "texture_bytes", "perlin_bytes",
                                          "image_bytes")] + \
               [(n, C.c_uint32) for n in ("n_nodes", "n_spheres", "n_quads", "n_instances", "n_media",
                                          "max_instance_depth", "lds_nodes", "lds_bytes", "ordered",
                                          "stack_entries")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class SceneOptions(C.Structure):
    _fields_ = [("scene", C.c_int32), ("bvh_policy", C.c_int32), ("scene_seed", C.c_uint64),
                ("image_width", C.c_int64), ("aspect_ratio", C.c_double), ("samples_per_pixel", C.c_int32),
                ("max_depth", C.c_int32), ("earth_image", C.c_char_p)]


RT_WALK_DEFAULT, RT_WALK_REFERENCE_ORDER, RT_WALK_AUTO, RT_WALK_OWN_TREES = -1, 0, 1, 2
RT_COMM_ID_BYTES = 128


class SceneCreateOptions(C.Structure):
    """rt_scene_options (per-scene options of rt_scene_create_ex)."""
    _fields_ = [("struct_size", C.c_uint32), ("walk", C.c_int32), ("leaf_max", C.c_int32), ("refit", C.c_int32),
                ("use_lds", C.c_int32), ("th_prim", C.c_int32), ("th_other", C.c_int32), ("th_shade", C.c_int32),
                ("th_box", C.c_int32), ("th_new", C.c_int32), ("sample_buffer_bytes", C.c_int64), ("reserved_pool", C.c_int32),
                ("flat_max", C.c_int32), ("start_shortcut", C.c_int32), ("defer_instances", C.c_int32),
                ("seq_lookahead", C.c_int32), ("slow_min", C.c_int32), ("slow_age", C.c_int32), ("wide", C.c_int32),
                ("quad_filter", C.c_int32), ("medium_first", C.c_int32)]


def scene_options(**kw) -> "SceneCreateOptions":
    """rt_scene_options_init, then the given fields (walk=, leaf_max=, flat_max=, refit=, use_lds=, th_*=, sample_buffer_bytes=,
    start_shortcut=, defer_instances=, seq_lookahead=, slow_min=, slow_age=, quad_filter=, medium_first=)."""
    o = SceneCreateOptions()
    amd_lib().rt_scene_options_init(C.byref(o))
    for k, v in kw.items():
        if k not in dict(SceneCreateOptions._fields_):
            raise TypeError(f"rt_scene_options has no field {k}")
        setattr(o, k, v)
    return o


class DebugNode(C.Structure):
    _fields_ = [("lo", C.c_double * 3), ("hi", C.c_double * 3), ("lo32", C.c_float * 3), ("hi32", C.c_float * 3),
                ("prim_lo", C.c_double * 3), ("prim_hi", C.c_double * 3), ("skip", C.c_uint32), ("kind", C.c_uint32),
                ("no_bbox", C.c_uint32), ("a", C.c_uint32), ("b", C.c_uint32), ("_pad", C.c_uint32)]


class RtError(RuntimeError):
    pass


# every symbol include/rt_amd.h declares: name -> (restype, argtypes)
RT_AMD_SYMBOLS = {
    "rt_device_count": (C.c_int, []),
    "rt_scene_create": (C.c_int, [C.POINTER(SceneDesc), C.c_int, C.POINTER(C.c_void_p)]),
    "rt_scene_destroy": (None, [C.c_void_p]),
    "rt_scene_get_stats": (C.c_int, [C.c_void_p, C.POINTER(SceneStats)]),
    "rt_render": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams), C.POINTER(C.c_double)]),
    "rt_render_device": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams), C.c_void_p, C.c_void_p]),
    "rt_render_device_counted": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams), C.c_void_p,
                                           C.c_void_p, C.POINTER(Counters)]),
    "rt_out_size": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "rt_tiles_to_frame_device": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_resolve_rgb8_device": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_scene_options_init": (None, [C.c_void_p]),
    "rt_scene_options_init_sized": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rt_scene_create_ex": (C.c_int, [C.POINTER(SceneDesc), C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "rt_resolve_rgb8_values_device": (C.c_int, [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_tiles_to_frame_rgb8_device": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_device_malloc": (C.c_int, [C.c_int, C.c_int64, C.POINTER(C.c_void_p)]),
    "rt_device_free": (C.c_int, [C.c_int, C.c_void_p]),
    "rt_device_download": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "rt_comm_get_unique_id": (C.c_int, [C.c_void_p]),
    "rt_comm_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "rt_comm_create_all": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]),
    "rt_comm_adopt": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "rt_comm_destroy": (None, [C.c_void_p]),
    "rt_comm_rank": (C.c_int, [C.c_void_p]),
    "rt_comm_size": (C.c_int, [C.c_void_p]),
    "rt_gather_tiles_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "rt_last_error": (C.c_char_p, []),
    "rt_version": (C.c_char_p, []),
}

# every symbol include/rt_amd_debug.h declares (test and tuning hooks; not part of the drop-in boundary)
RT_AMD_DEBUG_SYMBOLS = {
    "rt_debug_eval": (C.c_int, [C.c_int32, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                C.POINTER(C.c_double), C.c_int]),
    "rt_debug_box_tests": (C.c_int, [C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_double,
                                     C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_int]),
    "rt_debug_quad_filter_tests": (C.c_int, [C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_double,
                                             C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_int]),
    "rt_debug_compiled_nodes": (C.c_int, [C.POINTER(SceneDesc), C.c_int32, C.POINTER(DebugNode), C.c_int64,
                                          C.POINTER(C.c_int64)]),
    "rt_debug_stage_profile": (C.c_int, [C.POINTER(C.c_uint64)]),
    "rt_debug_last_launch": (C.c_int, [C.POINTER(C.c_uint32)]),
    "rt_debug_wide_layout": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]),
    "rt_debug_set_traversal": (C.c_int, [C.c_int32, C.c_int32]),
    "rt_debug_set_walk_shortcuts": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "rt_debug_ordered_layout": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_debug_ordered_layout_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_debug_set_tuning": (C.c_int, [C.c_int32] * 6),
}

RT_DEBUG_LOG, RT_DEBUG_SIN, RT_DEBUG_ACOS, RT_DEBUG_ATAN2, RT_DEBUG_POW5, RT_DEBUG_SQRT, RT_DEBUG_DIV, \
    RT_DEBUG_MUL_ADD, RT_DEBUG_RNG_RANDOM, RT_DEBUG_RNG_RANGE, RT_DEBUG_F32_ABOVE, RT_DEBUG_F32_BELOW, RT_DEBUG_RNG_UNNEXT = range(1, 14)


def debug_box_tests(rays, boxes, tmin, tmax, device=0):
    """rt_debug_box_tests: per (ray, box) pair: does the exact f64 test enter? the conservative f32 test of the
    reference-order walk? the packed pair test of the ordered walk (both slots must agree)?"""
    import numpy as np
    rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
    boxes = np.ascontiguousarray(boxes, dtype=np.float64).reshape(-1, 6)
    n = rays.shape[0]
    assert boxes.shape[0] == n
    exact = np.zeros(n, dtype=np.uint8); f32 = np.zeros(n, dtype=np.uint8)
    _check(amd_lib().rt_debug_box_tests(n, rays.ctypes.data_as(C.POINTER(C.c_double)),
                                        boxes.ctypes.data_as(C.POINTER(C.c_double)), tmin, tmax,
                                        exact.ctypes.data_as(C.POINTER(C.c_uint8)), f32.ctypes.data_as(C.POINTER(C.c_uint8)),
                                        device), "rt_debug_box_tests")
    assert (((f32 >> 1) & 1) == ((f32 >> 2) & 1)).all(), "the two slots of the pair test disagree on the same box"
    return exact.astype(bool), (f32 & 1).astype(bool), ((f32 >> 1) & 1).astype(bool)


def debug_quad_filter_tests(rays, quads, tmin, tmax, device=0):
    """rt_debug_quad_filter_tests: per (ray, quad = Q, u, v) pair: does the exact f64 Quad::hit accept within [tmin, tmax]? does the
    quad stage's conservative f32 filter keep the quad (both slots of the pair record must agree)?  does it claim alpha and beta are
    certainly inside [0, 1] — and are they, where the exact test evaluates them?"""
    import numpy as np
    rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
    quads = np.ascontiguousarray(quads, dtype=np.float64).reshape(-1, 9)
    n = rays.shape[0]
    assert quads.shape[0] == n
    exact = np.zeros(n, dtype=np.uint8); keep = np.zeros(n, dtype=np.uint8)
    _check(amd_lib().rt_debug_quad_filter_tests(n, rays.ctypes.data_as(C.POINTER(C.c_double)), quads.ctypes.data_as(C.POINTER(C.c_double)),
                                                tmin, tmax, exact.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                keep.ctypes.data_as(C.POINTER(C.c_uint8)), device), "rt_debug_quad_filter_tests")
    assert ((keep & 1) == ((keep >> 1) & 1)).all() and ((keep >> 2 & 1) == (keep >> 3 & 1)).all(), "the two slots of the pair record disagree on the same quad"
    # exact verdict | the filter keeps the quad | the filter says alpha, beta are certainly inside | the exact alpha, beta are inside
    # (True where the exact test did not get that far)
    return exact.astype(bool), (keep & 1).astype(bool), (keep >> 2 & 1).astype(bool), (keep >> 4 & 1).astype(bool)


def debug_compiled_nodes(host_scene, refit=True):
    """rt_debug_compiled_nodes: the records the device walks for this scene (runs on the CPU)."""
    n = C.c_int64()
    _check(amd_lib().rt_debug_compiled_nodes(C.byref(host_scene.desc), 1 if refit else 0, None, 0, C.byref(n)),
           "rt_debug_compiled_nodes")
    buf = (DebugNode * n.value)()
    _check(amd_lib().rt_debug_compiled_nodes(C.byref(host_scene.desc), 1 if refit else 0, buf, n.value, C.byref(n)),
           "rt_debug_compiled_nodes")
    return buf


class DebugOrdered(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("cap_nodes", "cap_spheres", "cap_quads", "cap_instances", "cap_steps", "cap_media",
                                         "n_nodes", "n_spheres", "n_quads", "n_instances", "n_steps", "n_media")] + \
               [(n, C.c_uint32) for n in ("ordered", "root", "stack_entries", "_pad")] + \
               [("nodes", C.c_void_p), ("spheres", C.c_void_p), ("quads", C.c_void_p), ("instances", C.c_void_p),
                ("steps", C.c_void_p), ("media", C.c_void_p)]


def debug_ordered_layout(host_scene, **option_fields) -> dict:
    """rt_debug_ordered_layout_ex: the scene compiler's ordered layout (numpy arrays), no device needed; leaf_max= / flat_max= as
    rt_scene_options has them."""
    import numpy as np
    io = DebugOrdered()
    opts = scene_options(**option_fields) if option_fields else None
    call = lambda: amd_lib().rt_debug_ordered_layout_ex(C.addressof(host_scene.desc), C.addressof(opts) if opts is not None else None, C.addressof(io))
    _check(call(), "rt_debug_ordered_layout")
    nodes = np.zeros((max(io.n_nodes, 1), 16), dtype=np.uint32)
    spheres = np.zeros((max(io.n_spheres, 1), 9)); quads = np.zeros((max(io.n_quads, 1), 10))
    insts = np.zeros((max(io.n_instances, 1), 8))
    steps = np.zeros((max(io.n_steps, 1), 28), dtype=np.uint32); media = np.zeros(max(io.n_media, 1), dtype=np.uint32)
    io.cap_nodes, io.cap_spheres, io.cap_quads, io.cap_instances = len(nodes), len(spheres), len(quads), len(insts)
    io.cap_steps, io.cap_media = len(steps), len(media)
    io.nodes, io.spheres, io.quads, io.instances = (a.ctypes.data for a in (nodes, spheres, quads, insts))
    io.steps, io.media = steps.ctypes.data, media.ctypes.data
    _check(call(), "rt_debug_ordered_layout")
    return {"ordered": bool(io.ordered), "root": int(io.root), "stack_entries": int(io.stack_entries),
            "nodes": nodes[:io.n_nodes], "spheres": spheres[:io.n_spheres], "quads": quads[:io.n_quads],
            "instances": insts[:io.n_instances], "steps": steps[:io.n_steps], "media": media[:io.n_media]}


def debug_wide_layout(host_scene, **option_fields) -> dict:
    """rt_debug_wide_layout: structure check of the four-child layout (CPU)."""
    buf = (C.c_uint64 * 6)()
    opts = scene_options(**option_fields) if option_fields else None
    _check(amd_lib().rt_debug_wide_layout(C.addressof(host_scene.desc), C.addressof(opts) if opts is not None else None, buf), "rt_debug_wide_layout")
    return dict(zip(("records", "found", "primitives", "stack_entries", "violations", "deepest"), (int(x) for x in buf)))


def debug_last_launch() -> dict:
    """rt_debug_last_launch: how this thread's last render was launched (LDS level, workgroup threads, workgroups)."""
    buf = (C.c_uint32 * 4)()
    _check(amd_lib().rt_debug_last_launch(buf), "rt_debug_last_launch")
    return {"lds_level": int(buf[1]), "threads": int(buf[2]), "grid": int(buf[3])}


def debug_stage_profile() -> dict:
    """rt_debug_stage_profile: per stage {rounds, lanes (mean active per round), cycles} of the last counted render."""
    buf = (C.c_uint64 * 36)()
    _check(amd_lib().rt_debug_stage_profile(buf), "rt_debug_stage_profile")
    out = {}
    for i, name in enumerate(("box", "sphere", "quad", "other", "shade", "newjob", "shade.rebuild", "shade.sample", "shade.texture",
                              "shade.material", "end.products", "end.newjob")):
        rounds, lanes, cycles = int(buf[3 * i]), int(buf[3 * i + 1]), int(buf[3 * i + 2])
        out[name] = {"rounds": rounds, "mean_active_lanes": lanes / rounds if rounds else 0.0, "cycles": cycles}
    return out


def debug_eval(op, a, b=None, device=0):
    """rt_debug_eval: one device-side scalar function over arrays (test hook)."""
    import numpy as np
    a = np.ascontiguousarray(a, dtype=np.float64)
    out = np.empty_like(a)
    bp = None
    if b is not None:
        b = np.ascontiguousarray(b, dtype=np.float64)
        assert b.shape == a.shape
        bp = b.ctypes.data_as(C.POINTER(C.c_double))
    _check(amd_lib().rt_debug_eval(op, a.size, a.ctypes.data_as(C.POINTER(C.c_double)), bp,
                                   out.ctypes.data_as(C.POINTER(C.c_double)), device), "rt_debug_eval")
    return out

# every symbol include/rt_host.h declares
RT_HOST_SYMBOLS = {
    "rth_scene_build": (C.c_int, [C.POINTER(SceneOptions), C.POINTER(C.c_void_p)]),
    "rth_scene_destroy": (None, [C.c_void_p]),
    "rth_scene_desc": (C.POINTER(SceneDesc), [C.c_void_p]),
    "rth_scene_camera": (C.POINTER(Camera), [C.c_void_p]),
    "rth_resolve_rgb8": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_uint8)]),
    "rth_write_png": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_uint8)]),
    "rth_synthetic_earth": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_uint8)]),
    "rth_load_image": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_uint8),
                                 C.c_int64]),
    "rth_last_error": (C.c_char_p, []),
}


def _bind(lib, table):
    for name, (restype, argtypes) in table.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


_host_lib = None
_amd_lib = None


def host_lib():
    """librt_host.so (CPU only)."""
    global _host_lib
    if _host_lib is None:
        path = LIB_DIR / "librt_host.so"
        if not path.exists():
            raise RtError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _host_lib = _bind(C.CDLL(str(path)), RT_HOST_SYMBOLS)
    return _host_lib


def amd_lib():
    """librt_amd.so (HIP).  Loading works without a GPU; rendering does not."""
    global _amd_lib
    if _amd_lib is None:
        path = Path(os.environ.get("RT_AMD_LIB", LIB_DIR / "librt_amd.so"))  # RT_AMD_LIB: tuning builds only
        if not path.exists():
            raise RtError(f"{path} is missing: the HIP renderer is not built and there is no fallback; "
                          "run `python -c 'import __graft_entry__ as g; g.build()'`")
        _amd_lib = _bind(_bind(C.CDLL(str(path), mode=C.RTLD_GLOBAL), RT_AMD_SYMBOLS), RT_AMD_DEBUG_SYMBOLS)
    return _amd_lib


def _check(rc, where):
    if rc != 0:
        raise RtError(f"{where} failed ({rc}): {amd_lib().rt_last_error().decode(errors='replace')}")


class HostScene:
    """A scene built by the host library: the reference's `main` up to the call of render()
    (scene function -> BVHNode::new -> Camera), described as rt_scene_desc + rt_camera."""

    def __init__(self, scene: int, *, scene_seed: int = 1, width: int = 0, aspect: float = 0.0, spp: int = 0,
                 depth: int = 0, earth_image: str | None = None, bvh: str = "reference"):
        lib = host_lib()
        self._earth = earth_image.encode() if earth_image else None
        opts = SceneOptions(scene=scene, bvh_policy=1 if bvh == "sah" else 0, scene_seed=scene_seed,
                            image_width=width, aspect_ratio=aspect, samples_per_pixel=spp, max_depth=depth,
                            earth_image=self._earth)
        handle = C.c_void_p()
        if lib.rth_scene_build(C.byref(opts), C.byref(handle)) != 0:
            raise RtError(lib.rth_last_error().decode(errors="replace"))
        self._handle = handle
        self.desc = lib.rth_scene_desc(handle).contents
        self.camera = lib.rth_scene_camera(handle).contents
        # the records live in the C object: keep it alive for as long as either view is referenced
        self.desc._owner = self
        self.camera._owner = self

    @property
    def width(self):
        return self.camera.image_width

    @property
    def height(self):
        return self.camera.image_height

    def close(self):
        if getattr(self, "_handle", None):
            host_lib().rth_scene_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def render_params(*, seed=1, sample_begin=0, sample_end=0, max_depth=0, accumulate=False, shard_index=0,
                  shard_count=1, out_layout=RT_OUT_FRAME, device=0) -> RenderParams:
    return RenderParams(seed=seed, sample_begin=sample_begin, sample_end=sample_end, max_depth=max_depth,
                        accumulate=1 if accumulate else 0, shard_index=shard_index, shard_count=shard_count,
                        out_layout=out_layout, device=device)


def out_size(width, height, out_layout=RT_OUT_FRAME, shard_index=0, shard_count=1) -> int:
    n = int(amd_lib().rt_out_size(width, height, out_layout, shard_index, shard_count))
    if n < 0:
        raise RtError(f"rt_out_size: invalid arguments (size {width}x{height}, layout {out_layout}, "
                      f"shard {shard_index} of {shard_count})")
    return n


class DeviceScene:
    """rt_scene handle: the compiled scene resident in one GPU's HBM."""

    def __init__(self, host_scene: HostScene, device: int = 0, options: "SceneCreateOptions | None" = None, **option_fields):
        lib = amd_lib()
        self.host_scene = host_scene  # keeps desc memory alive during create
        self.device = device
        handle = C.c_void_p()
        if option_fields:
            assert options is None
            options = scene_options(**option_fields)
        if options is None:
            _check(lib.rt_scene_create(C.byref(host_scene.desc), device, C.byref(handle)), "rt_scene_create")
        else:
            _check(lib.rt_scene_create_ex(C.byref(host_scene.desc), device, C.byref(options), C.byref(handle)), "rt_scene_create_ex")
        self._handle = handle

    def stats(self) -> dict:
        st = SceneStats()
        _check(amd_lib().rt_scene_get_stats(self._handle, C.byref(st)), "rt_scene_get_stats")
        return st.as_dict()

    def render(self, params: RenderParams, camera: Camera | None = None):
        """rt_render: host buffer out (numpy float64)."""
        import numpy as np
        cam = camera if camera is not None else self.host_scene.camera
        n = out_size(cam.image_width, cam.image_height, params.out_layout, params.shard_index, params.shard_count)
        out = np.zeros(n, dtype=np.float64)
        _check(amd_lib().rt_render(self._handle, C.byref(cam), C.byref(params),
                                   out.ctypes.data_as(C.POINTER(C.c_double))), "rt_render")
        return out

    def render_device(self, params: RenderParams, d_out_ptr: int, stream: int = 0, camera: Camera | None = None):
        """rt_render_device: `d_out_ptr` is a device pointer (e.g. torch tensor .data_ptr()), `stream` a hipStream_t."""
        cam = camera if camera is not None else self.host_scene.camera
        _check(amd_lib().rt_render_device(self._handle, C.byref(cam), C.byref(params), C.c_void_p(d_out_ptr),
                                          C.c_void_p(stream)), "rt_render_device")

    def render_device_counted(self, params: RenderParams, d_out_ptr: int, stream: int = 0,
                              camera: Camera | None = None) -> dict:
        cam = camera if camera is not None else self.host_scene.camera
        cnt = Counters()
        _check(amd_lib().rt_render_device_counted(self._handle, C.byref(cam), C.byref(params),
                                                  C.c_void_p(d_out_ptr), C.c_void_p(stream), C.byref(cnt)),
               "rt_render_device_counted")
        return cnt.as_dict()

    def close(self):
        if getattr(self, "_handle", None):
            amd_lib().rt_scene_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def tiles_to_frame_device(width, height, shard_count, d_gathered_ptr: int, d_frame_ptr: int, stream: int = 0):
    _check(amd_lib().rt_tiles_to_frame_device(width, height, shard_count, C.c_void_p(d_gathered_ptr),
                                              C.c_void_p(d_frame_ptr), C.c_void_p(stream)),
           "rt_tiles_to_frame_device")


def resolve_rgb8_device(width, height, spp, d_frame_ptr: int, d_rgb8_ptr: int, stream: int = 0):
    _check(amd_lib().rt_resolve_rgb8_device(width, height, spp, C.c_void_p(d_frame_ptr), C.c_void_p(d_rgb8_ptr),
                                            C.c_void_p(stream)), "rt_resolve_rgb8_device")


def resolve_rgb8_values_device(n_values, spp, d_sum_ptr: int, d_rgb8_ptr: int, stream: int = 0):
    _check(amd_lib().rt_resolve_rgb8_values_device(n_values, spp, C.c_void_p(d_sum_ptr), C.c_void_p(d_rgb8_ptr),
                                                   C.c_void_p(stream)), "rt_resolve_rgb8_values_device")


def tiles_to_frame_rgb8_device(width, height, shard_count, d_gathered_ptr: int, d_frame_ptr: int, stream: int = 0):
    _check(amd_lib().rt_tiles_to_frame_rgb8_device(width, height, shard_count, C.c_void_p(d_gathered_ptr),
                                                   C.c_void_p(d_frame_ptr), C.c_void_p(stream)),
           "rt_tiles_to_frame_rgb8_device")


class Comm:
    """rt_comm: one RCCL communicator handle of this rank (include/rt_amd.h "frame-end gather")."""

    def __init__(self, handle):
        self._handle = handle

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * RT_COMM_ID_BYTES)()
        _check(amd_lib().rt_comm_get_unique_id(buf), "rt_comm_get_unique_id")
        return bytes(buf)

    @classmethod
    def create(cls, unique_id: bytes, rank: int, n_ranks: int, device: int) -> "Comm":
        assert len(unique_id) == RT_COMM_ID_BYTES
        buf = (C.c_uint8 * RT_COMM_ID_BYTES).from_buffer_copy(unique_id)
        h = C.c_void_p()
        _check(amd_lib().rt_comm_create(buf, rank, n_ranks, device, C.byref(h)), "rt_comm_create")
        return cls(h)

    rank = property(lambda self: amd_lib().rt_comm_rank(self._handle))
    size = property(lambda self: amd_lib().rt_comm_size(self._handle))

    def gather_tiles(self, width, height, elem_bytes, d_tiles_ptr: int, d_gathered_ptr: int, root: int = 0, stream: int = 0):
        _check(amd_lib().rt_gather_tiles_device(self._handle, width, height, elem_bytes, C.c_void_p(d_tiles_ptr),
                                                C.c_void_p(d_gathered_ptr), root, C.c_void_p(stream)), "rt_gather_tiles_device")

    def close(self):
        if getattr(self, "_handle", None):
            amd_lib().rt_comm_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def resolve_rgb8_host(width, height, spp, sums):
    """color_to_rgb(sum / spp) on the host (librt_host; src/renderer.rs:55-58)."""
    import numpy as np
    sums = np.ascontiguousarray(sums, dtype=np.float64)
    out = np.zeros(width * height * 3, dtype=np.uint8)
    if host_lib().rth_resolve_rgb8(width, height, spp, sums.ctypes.data_as(C.POINTER(C.c_double)),
                                   out.ctypes.data_as(C.POINTER(C.c_uint8))) != 0:
        raise RtError(host_lib().rth_last_error().decode(errors="replace"))
    return out.reshape(height, width, 3)


def write_png(path, rgb8):
    import numpy as np
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w, _ = rgb8.shape
    if host_lib().rth_write_png(os.fsencode(path), w, h, rgb8.ctypes.data_as(C.POINTER(C.c_uint8))) != 0:
        raise RtError(host_lib().rth_last_error().decode(errors="replace"))
