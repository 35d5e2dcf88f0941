"""ctypes binding of the CPU oracle (oracle/librt_oracle.so) — for tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg only.  The product (rust-tracing_amd/) never imports this."""
from __future__ import annotations

import ctypes as C
import importlib
import os
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
rt = importlib.import_module("rust-tracing_amd")

ORC_AABB_REFERENCE = 0
ORC_AABB_TIGHT = 1


class OrcOptions(C.Structure):
    _fields_ = [("aabb_mode", C.c_int32), ("threads", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = ROOT / "oracle" / "librt_oracle.so"
        if not path.exists():
            raise RuntimeError(f"{path} missing: run `make -C oracle`")
        l = C.CDLL(str(path))
        l.orc_render.restype = C.c_int
        l.orc_render.argtypes = [C.POINTER(rt.SceneDesc), C.POINTER(rt.Camera), C.POINTER(rt.RenderParams),
                                 C.POINTER(OrcOptions), C.POINTER(C.c_double), C.POINTER(rt.Counters)]
        l.orc_out_size.restype = C.c_int64
        l.orc_out_size.argtypes = [C.c_int32] * 5
        for name in ("orc_log", "orc_sin", "orc_acos", "orc_pow5"):
            getattr(l, name).restype = C.c_double
            getattr(l, name).argtypes = [C.c_double]
        l.orc_atan2.restype = C.c_double
        l.orc_atan2.argtypes = [C.c_double, C.c_double]
        l.orc_rng_key.restype = C.c_uint64
        l.orc_rng_key.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        l.orc_rng_draw.restype = C.c_uint64
        l.orc_rng_draw.argtypes = [C.c_uint64, C.c_uint64]
        l.orc_kat_random.restype = C.c_double
        l.orc_kat_random.argtypes = [C.c_uint64, C.c_uint64]
        l.orc_kat_gen_range.restype = C.c_double
        l.orc_kat_gen_range.argtypes = [C.c_uint64, C.c_uint64, C.c_double, C.c_double]
        d3 = C.c_double * 3
        l.orc_kat_sphere_hit.restype = C.c_int
        l.orc_kat_sphere_hit.argtypes = [C.POINTER(rt.Sphere), d3, d3, C.c_double, C.c_double, C.c_double,
                                         C.c_double * 10]
        l.orc_kat_quad_hit.restype = C.c_int
        l.orc_kat_quad_hit.argtypes = [C.POINTER(rt.Quad), d3, d3, C.c_double, C.c_double, C.c_double * 10]
        l.orc_kat_aabb_hit.restype = C.c_int
        l.orc_kat_aabb_hit.argtypes = [C.POINTER(rt.Aabb), d3, d3, C.c_double, C.c_double, C.c_int]
        l.orc_kat_reflect.restype = None
        l.orc_kat_reflect.argtypes = [d3, d3, d3]
        l.orc_kat_refract.restype = None
        l.orc_kat_refract.argtypes = [d3, d3, C.c_double, d3]
        l.orc_kat_reflectance.restype = C.c_double
        l.orc_kat_reflectance.argtypes = [C.c_double, C.c_double]
        l.orc_kat_perlin_noise.restype = C.c_double
        l.orc_kat_perlin_noise.argtypes = [C.POINTER(rt.Perlin), d3]
        l.orc_kat_perlin_turbulence.restype = C.c_double
        l.orc_kat_perlin_turbulence.argtypes = [C.POINTER(rt.Perlin), d3, C.c_int]
        l.orc_kat_texture_value.restype = None
        l.orc_kat_texture_value.argtypes = [C.POINTER(rt.SceneDesc), C.c_int32, C.c_double, C.c_double, d3, d3]
        l.orc_kat_camera_ray.restype = C.c_int
        l.orc_kat_camera_ray.argtypes = [C.POINTER(rt.Camera), C.c_uint64, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_double * 7]
        _lib = l
    return _lib


def default_threads() -> int:
    """Threads for the CPU baseline: the cores this process may actually use (affinity mask, capped by the
    cgroup CPU quota when there is one)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    env = os.environ.get("RT_ORACLE_THREADS")
    if env:
        n = int(env)
    return max(1, min(n, 256))


def render(host_scene, params, *, aabb_mode=ORC_AABB_REFERENCE, threads=None, camera=None, out=None,
           want_counters=False):
    """orc_render: same contract as rt_render, on the CPU.  Returns float64 array (and counters dict)."""
    cam = camera if camera is not None else host_scene.camera
    n = int(lib().orc_out_size(cam.image_width, cam.image_height, params.out_layout, params.shard_index,
                               params.shard_count if params.shard_count > 0 else 1))
    if out is None:
        out = np.zeros(n, dtype=np.float64)
    assert out.size == n and out.dtype == np.float64 and out.flags.c_contiguous
    opts = OrcOptions(aabb_mode=aabb_mode, threads=threads or default_threads())
    cnt = rt.Counters()
    rc = lib().orc_render(C.byref(host_scene.desc), C.byref(cam), C.byref(params), C.byref(opts),
                          out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(cnt))
    if rc != 0:
        raise RuntimeError(f"orc_render failed ({rc})")
    return (out, cnt.as_dict()) if want_counters else out
