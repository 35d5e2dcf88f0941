"""The C-ABI libraries load (no GPU needed), export every symbol the headers declare, and the ctypes mirrors of
the ABI structs have the C compiler's layout."""
import ctypes as C
import re
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
DECL = re.compile(r"^\s*(?:const\s+)?(?:int|void|int64_t|char|rt_\w+|rth_\w+)\s*\*?\s*\b((?:rt|rth)_\w+)\s*\(", re.M)


def declared(header):
    text = (ROOT / "include" / header).read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(DECL.findall(text)))


def exported(lib):
    out = subprocess.run(["nm", "-D", "--defined-only", str(lib)], check=True, capture_output=True, text=True).stdout
    return {line.split()[-1] for line in out.splitlines() if line.strip()}


def test_librt_amd_exports_everything_rt_amd_h_declares(rt):
    names = declared("rt_amd.h")
    assert "rt_render" in names and "rt_scene_create" in names and len(names) >= 12
    have = exported(rt.LIB_DIR / "librt_amd.so")
    assert not [n for n in names if n not in have]
    assert sorted(rt.RT_AMD_SYMBOLS) == names, "python binding table out of sync with rt_amd.h"
    assert not [n for n in names if n.startswith("rt_debug")], "test hooks belong in rt_amd_debug.h"
    debug_names = declared("rt_amd_debug.h")
    assert not [n for n in debug_names if n not in have]
    assert sorted(rt.RT_AMD_DEBUG_SYMBOLS) == debug_names, "python binding table out of sync with rt_amd_debug.h"
    lib = rt.amd_lib()  # binds every symbol; loading must not need a GPU
    assert lib.rt_version().startswith(b"rt_amd")
    assert lib.rt_device_count() >= 0


def test_librt_host_exports_everything_rt_host_h_declares(rt):
    names = declared("rt_host.h")
    have = exported(rt.LIB_DIR / "librt_host.so")
    assert not [n for n in names if n not in have]
    assert sorted(rt.RT_HOST_SYMBOLS) == names
    rt.host_lib()


def test_product_does_not_link_or_mention_the_oracle(rt):
    """The oracle is a checker: the product must not depend on it."""
    for lib in ("librt_amd.so", "librt_host.so"):
        out = subprocess.run(["ldd", str(rt.LIB_DIR / lib)], capture_output=True, text=True).stdout
        assert "oracle" not in out
    for path in (ROOT / "rust-tracing_amd").rglob("*"):
        if path.is_file() and path.suffix in (".py", ".hip", ".h", ".hpp", ".cpp") or path.name == "Makefile":
            text = path.read_text(errors="ignore")
            assert "oracle/" not in text and "oracle_lib" not in text and "librt_oracle" not in text, path


def test_no_gpu_means_an_error_not_a_fallback(rt):
    lib = rt.amd_lib()
    if lib.rt_device_count() > 0:
        return  # covered by the gpu tests
    hs = rt.HostScene(4, width=16, spp=1)
    handle = C.c_void_p()
    assert lib.rt_scene_create(C.byref(hs.desc), 0, C.byref(handle)) == -2  # RT_ERR_NO_DEVICE
    assert b"no CPU path" in lib.rt_last_error()


def test_scene_options_are_checked_before_anything_else(rt):
    """rt_scene_options: defaults from rt_scene_options_init; an unknown struct size or walk is an argument error (no GPU
    needed to find out); a shorter struct of an older caller is accepted."""
    lib = rt.amd_lib()
    o = rt.scene_options()
    assert (o.struct_size, o.walk, o.refit, o.use_lds, o.th_prim, o.sample_buffer_bytes, o.reserved_pool) == (C.sizeof(rt.SceneCreateOptions), -1, -1, -1, -1, 0, -1)
    hs = rt.HostScene(4, width=16, spp=1)
    handle = C.c_void_p()
    for bad in (dict(struct_size=4), dict(struct_size=4096), dict(walk=7)):
        opts = rt.scene_options(**bad)
        assert lib.rt_scene_create_ex(C.byref(hs.desc), 0, C.byref(opts), C.byref(handle)) == -1, bad  # RT_ERR_INVALID_ARGUMENT
    older = rt.scene_options(struct_size=48, walk=rt.RT_WALK_REFERENCE_ORDER)  # (the struct as it was in round 1)
    rc = lib.rt_scene_create_ex(C.byref(hs.desc), 0, C.byref(older), C.byref(handle))
    assert rc == (0 if lib.rt_device_count() > 0 else -2)
    if rc == 0:
        lib.rt_scene_destroy(handle)


def test_an_older_callers_shorter_options_struct_keeps_the_defaults_it_never_knew(rt):
    """Round 2's rt_scene_options was 56 bytes and ended in a reserved word that its init zeroed — where flat_max (0: off) is
    now.  A caller built against it must get the default flat leaves (Cornell: 4 records, not 17), and the sized init must
    not write past the caller's struct."""
    lib = rt.amd_lib()
    hs = rt.HostScene(6, width=16, spp=1)

    def layout(opts):
        io = rt.DebugOrdered()
        assert lib.rt_debug_ordered_layout_ex(C.addressof(hs.desc), C.addressof(opts) if opts is not None else None, C.addressof(io)) == 0, lib.rt_last_error()
        return int(io.n_nodes)

    default_records = layout(None)
    old = rt.scene_options()
    C.memset(C.addressof(old), 0xAB, C.sizeof(old))
    assert lib.rt_scene_options_init_sized(C.addressof(old), 56) == 0
    raw = bytes(old)
    assert raw[56:] == b"\xab" * (C.sizeof(old) - 56), "the sized init wrote past the caller's 56 bytes"
    assert old.struct_size == 56
    C.memset(C.addressof(old) + 52, 0, 4)  # what round 2's own init left in its reserved word
    assert layout(old) == default_records
    off = rt.scene_options(flat_max=0)
    assert layout(off) > default_records  # (the switch itself still works for a caller that knows it)
    for bad in (4, 58, 4096):
        assert lib.rt_scene_options_init_sized(C.addressof(old), bad) == -1
    short = rt.scene_options(struct_size=4)
    io = rt.DebugOrdered()
    assert lib.rt_debug_ordered_layout_ex(C.addressof(hs.desc), C.addressof(short), C.addressof(io)) == -1


def test_ctypes_structs_match_the_c_layout(rt, tmp_path):
    structs = {"rt_vec3": rt.Vec3, "rt_aabb": rt.Aabb, "rt_ref": rt.Ref, "rt_sphere": rt.Sphere, "rt_quad": rt.Quad,
               "rt_list": rt.List, "rt_translate": rt.Translate, "rt_rotate_y": rt.RotateY, "rt_bvh_node": rt.BvhNode,
               "rt_bvh": rt.Bvh, "rt_constant_medium": rt.ConstantMedium, "rt_material": rt.Material,
               "rt_texture": rt.Texture, "rt_perlin": rt.Perlin, "rt_image": rt.Image, "rt_scene_desc": rt.SceneDesc,
               "rt_camera": rt.Camera, "rt_render_params": rt.RenderParams, "rt_counters": rt.Counters,
               "rt_scene_stats": rt.SceneStats, "rt_scene_options": rt.SceneCreateOptions, "rth_scene_options": rt.SceneOptions}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "rt_host.h"', "int main(void){"]
    for cname, cls in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append("return 0;}")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", str(ROOT / "include"), "-o", str(exe), str(src)], check=True)
    got = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"


def test_the_rust_binding_in_integration_md_covers_rt_amd_h(rt):
    """INTEGRATION.md's `extern "C"` block is the binding a maintainer of the reference would paste into src/gpu.rs (the seam is
    render(), /root/reference src/renderer.rs:12 — no Rust toolchain here to compile it): every function include/rt_amd.h declares must
    be in it with the same number of arguments, and every #[repr(C)] struct must list its C counterpart's fields in order."""
    header = re.sub(r"/\*.*?\*/", "", (ROOT / "include" / "rt_amd.h").read_text(), flags=re.S)
    c_arity = {}
    for m in re.finditer(r"\b((?:rt)_\w+)\s*\(([^;{}]*?)\)\s*;", header):
        name, args = m.group(1), m.group(2).strip()
        if name in rt.RT_AMD_SYMBOLS:
            c_arity[name] = 0 if args in ("", "void") else args.count(",") + 1
    assert sorted(c_arity) == sorted(rt.RT_AMD_SYMBOLS)
    text = (ROOT / "INTEGRATION.md").read_text()
    block = text[text.index('extern "C" {'):]
    block = block[:block.index("\n}")]
    block = re.sub(r"//[^\n]*", "", block)
    rust_arity = {}
    for m in re.finditer(r"pub fn (rt_\w+)\s*\(([^)]*)\)", block):
        args = m.group(2).strip()
        rust_arity[m.group(1)] = 0 if not args else len([a for a in args.split(",") if a.strip()])
    assert sorted(rust_arity) == sorted(c_arity), (sorted(set(c_arity) - set(rust_arity)), sorted(set(rust_arity) - set(c_arity)))
    assert rust_arity == c_arity, {n: (rust_arity[n], c_arity[n]) for n in c_arity if rust_arity[n] != c_arity[n]}
    # struct fields, in order (names only: the types are checked against gcc through the ctypes mirrors above)
    mirrors = {"rt_vec3": rt.Vec3, "rt_aabb": rt.Aabb, "rt_ref": rt.Ref, "rt_sphere": rt.Sphere, "rt_quad": rt.Quad, "rt_list": rt.List,
               "rt_translate": rt.Translate, "rt_rotate_y": rt.RotateY, "rt_bvh_node": rt.BvhNode, "rt_bvh": rt.Bvh,
               "rt_constant_medium": rt.ConstantMedium, "rt_material": rt.Material, "rt_texture": rt.Texture, "rt_perlin": rt.Perlin,
               "rt_image": rt.Image, "rt_scene_desc": rt.SceneDesc, "rt_camera": rt.Camera, "rt_render_params": rt.RenderParams,
               "rt_scene_options": rt.SceneCreateOptions, "rt_counters": rt.Counters, "rt_scene_stats": rt.SceneStats}
    rust = re.sub(r"/\*.*?\*/", "", re.sub(r"//[^\n]*", "", text), flags=re.S)
    for cname, cls in mirrors.items():
        m = re.search(r"pub struct " + cname + r"\s*\{(.*?)\}", rust, flags=re.S)
        assert m, f"INTEGRATION.md declares no struct {cname}"
        fields = re.findall(r"pub (\w+)\s*:", m.group(1))
        assert fields == [f for f, _ in cls._fields_], (cname, fields)


def test_the_rccl_stand_in_of_the_test_suite_exports_what_the_gather_binds():
    """tests/rccl_stub (test infrastructure: RT_RCCL_LIB in tests/test_gpu_gather_stub.py) must define every RCCL entry point that
    rust-tracing_amd/csrc/rt_gather.cpp looks up — and the product must not know the stub exists."""
    stub = ROOT / "tests" / "rccl_stub" / "librccl_stub.so"
    assert stub.exists(), "run __graft_entry__.build()"
    wanted = set(re.findall(r'sym\("(nccl\w+)"\)', (ROOT / "rust-tracing_amd" / "csrc" / "rt_gather.cpp").read_text()))
    assert len(wanted) >= 11
    exported = {line.split()[-1] for line in subprocess.run(["nm", "-D", "--defined-only", str(stub)], check=True, capture_output=True, text=True).stdout.splitlines()}
    assert wanted <= exported, wanted - exported
    for src in (ROOT / "rust-tracing_amd").rglob("*"):
        if src.is_file() and src.suffix in (".cpp", ".hpp", ".h", ".hip", ".py"):
            assert "rccl_stub" not in src.read_text(errors="ignore"), src
