"""Host-side logic (librt_host): the reference's scene builders, BVH build, camera and output stage."""
import ctypes as C
import math

import numpy as np
import pytest


def kinds(d, arr, n):
    return [getattr(d, arr)[i] for i in range(getattr(d, n))]


def desc_bytes(rt, d):
    parts = []
    for arr, n in (("spheres", "n_spheres"), ("quads", "n_quads"), ("bvh_nodes", "n_bvh_nodes"), ("materials", "n_materials"),
                   ("textures", "n_textures"), ("perlins", "n_perlins"), ("translates", "n_translates"), ("rotates", "n_rotates")):
        k = getattr(d, n)
        if k:
            parts.append(C.string_at(getattr(d, arr), k * C.sizeof(getattr(d, arr)._type_)))
    return b"".join(parts)


def test_scene_contents_follow_the_reference(rt):
    # random_balls (src/main.rs:56-138): ground + up to 22*22 small + 3 big
    d = rt.HostScene(0, spp=1).desc
    assert 440 <= d.n_spheres <= 488 and d.n_quads == 0
    sp = kinds(d, "spheres", "n_spheres")
    assert sum(1 for s in sp if s.radius == 1000.0) == 1 and sum(1 for s in sp if s.radius == 1.0) == 3
    small = [s for s in sp if s.radius == 0.2]
    mats = kinds(d, "materials", "n_materials")
    frac_lambert = sum(1 for s in small if mats[s.material].kind == rt.RT_MATERIAL_LAMBERTIAN) / len(small)
    assert 0.7 < frac_lambert < 0.9                        # choose_mat < 0.8
    assert all(bool(s.is_moving) == (mats[s.material].kind == rt.RT_MATERIAL_LAMBERTIAN) for s in small)  # :79-84
    assert all(0 <= s.center_vec.y < 0.5 and s.center_vec.x == 0 for s in small if s.is_moving)
    assert all(math.dist((s.center.x, s.center.y, s.center.z), (4, 0.2, 0)) > 0.9 for s in small)
    # cornell_box (src/main.rs:344-421): 6 quads + 2 x Translate(RotateY(cube))
    c = rt.HostScene(6, spp=1).desc
    assert (c.n_quads, c.n_lists, c.n_translates, c.n_rotates, c.n_spheres, c.n_media) == (18, 2, 2, 2, 0, 0)
    # records are emitted in BVH order, not in scene order: find box1 by its offset
    t1 = [t for t in kinds(c, "translates", "n_translates") if t.offset.tuple() == (265.0, 0.0, 295.0)][0]
    assert t1.object.kind == rt.RT_HITTABLE_ROTATE_Y
    r = c.rotates[t1.object.index]
    assert r.sin_theta == math.sin(15 * math.pi / 180) and r.cos_theta == math.cos(15 * math.pi / 180)
    assert r.object.kind == rt.RT_HITTABLE_LIST and c.lists[r.object.index].count == 6
    lights = [m for m in kinds(c, "materials", "n_materials") if m.kind == rt.RT_MATERIAL_DIFFUSE_LIGHT]
    assert len(lights) == 1 and c.textures[lights[0].texture].color.tuple() == (15.0, 15.0, 15.0)
    # cornell_smoke (:423-506): the boxes become media with Isotropic phase functions
    s = rt.HostScene(7, spp=1).desc
    assert s.n_media == 2 and s.media[0].neg_inv_density == -1.0 / 0.01
    assert s.materials[s.media[0].phase_material].kind == rt.RT_MATERIAL_ISOTROPIC
    assert s.media[0].boundary.kind == rt.RT_HITTABLE_TRANSLATE
    # final_scene (:508-639)
    f = rt.HostScene(8, spp=1, earth_image="synthetic:64x32").desc
    assert f.n_quads == 2400 + 1 and f.n_spheres == 1000 + 7 and f.n_media == 2 and f.n_bvhs == 3 and f.n_perlins == 1
    assert f.n_images == 1 and (f.images[0].width, f.images[0].height) == (64, 32)
    # the r=70 glass sphere is shared by the world and by the first medium (one record, two references)
    med = {f.spheres[m.boundary.index].radius: m for m in kinds(f, "media", "n_media") if m.boundary.kind == rt.RT_HITTABLE_SPHERE}
    assert sorted(med) == [70.0, 5000.0]
    assert sum(1 for s_ in kinds(f, "spheres", "n_spheres") if s_.radius == 70.0) == 1
    assert med[70.0].neg_inv_density == -1.0 / 0.2 and med[5000.0].neg_inv_density == -1.0 / 0.0001


def test_scene_build_is_a_pure_function_of_the_seed(rt):
    a = rt.HostScene(8, scene_seed=5, spp=1, earth_image="synthetic:16x8")
    b = rt.HostScene(8, scene_seed=5, spp=1, earth_image="synthetic:16x8")
    c = rt.HostScene(8, scene_seed=6, spp=1, earth_image="synthetic:16x8")
    assert desc_bytes(rt, a.desc) == desc_bytes(rt, b.desc) != desc_bytes(rt, c.desc)


@pytest.mark.parametrize("scene", range(9))
@pytest.mark.parametrize("bvh", ["reference", "sah"])
def test_bvh_invariants(rt, scene, bvh):
    hs = rt.HostScene(scene, spp=1, earth_image="synthetic:16x8", bvh=bvh)
    d = hs.desc
    nodes = kinds(d, "bvh_nodes", "n_bvh_nodes")
    assert d.world.kind == rt.RT_HITTABLE_BVH
    seen = set()

    def walk(i):
        assert i not in seen
        seen.add(i)
        n = nodes[i]
        if n.is_leaf:
            return 1
        for child in (n.left, n.right):
            cb = nodes[child].bbox
            assert all(n.bbox.lo[k] <= cb.lo[k] and cb.hi[k] <= n.bbox.hi[k] for k in range(3))
        return walk(n.left) + walk(n.right)

    leaves = sum(walk(d.bvhs[b].root) for b in range(d.n_bvhs))
    assert len(seen) == d.n_bvh_nodes and leaves * 2 - d.n_bvhs == d.n_bvh_nodes


def test_bounding_boxes(rt):
    c = rt.HostScene(6, spp=1).desc
    nodes = kinds(c, "bvh_nodes", "n_bvh_nodes")
    leaf_boxes = {(n.object.kind, n.object.index): n.bbox for n in nodes if n.is_leaf}
    # a quad thinner than 1e-4 on an axis is padded by 5e-5 on both sides (src/aabb.rs:35-53)
    floor = [b for (k, i), b in leaf_boxes.items() if k == rt.RT_HITTABLE_QUAD and b.hi[1] - b.lo[1] < 1e-3 and b.lo[1] < 1]
    assert floor and floor[0].lo[1] == -0.00005 and floor[0].hi[1] == 0.00005
    # Translate(RotateY(cube)): hull of the eight rotated corners, then shifted; the cube list's own box starts at
    # the all-zero default (src/hittable.rs:50-57), so it always contains the origin
    i1 = [i for i in range(c.n_translates) if c.translates[i].offset.tuple() == (265.0, 0.0, 295.0)][0]
    b1 = leaf_boxes[(rt.RT_HITTABLE_TRANSLATE, i1)]
    th = math.radians(15)
    xs = [math.cos(th) * x + math.sin(th) * z for x in (0, 165) for z in (0, 165)]
    assert b1.lo[0] == pytest.approx(265 + min(xs)) and b1.hi[0] == pytest.approx(265 + max(xs)) and b1.hi[1] == pytest.approx(330)
    # moving sphere: union of the boxes at both ends (src/sphere.rs:34-46)
    r = rt.HostScene(0, spp=1).desc
    nodes = kinds(r, "bvh_nodes", "n_bvh_nodes")
    for n in nodes:
        if n.is_leaf and n.object.kind == rt.RT_HITTABLE_SPHERE:
            s = r.spheres[n.object.index]
            assert n.bbox.lo[1] == s.center.y - s.radius and n.bbox.hi[1] == s.center.y + s.center_vec.y + s.radius


def test_camera_new(rt):
    cam = rt.HostScene(6, spp=1).camera       # cornell: vfov 40, from (278,278,-800) at (278,278,0), 600x600
    assert (cam.image_width, cam.image_height, cam.max_depth, cam.samples_per_pixel) == (600, 600, 8, 1)
    vh = 2.0 * math.tan(math.radians(40) / 2) * 10.0
    assert cam.pixel_delta_u.x == pytest.approx(-vh / 600, rel=1e-15) and cam.pixel_delta_u.y == 0 and cam.pixel_delta_u.z == 0
    assert cam.pixel_delta_v.y == pytest.approx(-vh / 600, rel=1e-15)
    # centre of the image plane lies focus_dist along the view direction
    centre = np.array(cam.pixel00_loc.tuple()) + 299.5 * np.array(cam.pixel_delta_u.tuple()) + 299.5 * np.array(cam.pixel_delta_v.tuple())
    assert centre == pytest.approx([278, 278, -790], abs=1e-9)
    assert cam.defocus_angle == 0.0 and cam.background.tuple() == (0.0, 0.0, 0.0)
    c2 = rt.HostScene(0, width=1200, aspect=1.5, spp=500, depth=50).camera   # BASELINE.json configs[1]
    assert (c2.image_width, c2.image_height, c2.samples_per_pixel, c2.max_depth) == (1200, 800, 500, 50)
    assert c2.defocus_angle == 0.6 and c2.background.tuple() == (0.7, 0.8, 1.0)


def test_output_stage(rt, tmp_path):
    # color_to_rgb (src/color.rs:12-19): gamma 1/2.2, clamp to [0, 0.999], *256, truncating cast; NaN -> 0
    vals = np.array([[0.0, 1.0, 4.0], [0.5, 0.25, 1e-9], [float("nan"), -1.0, float("inf")], [0.999 ** 2.2, 0.2, 0.7]])
    sums = (vals * 10.0).reshape(-1)        # spp = 10
    rgb = rt.resolve_rgb8_host(2, 2, 10, sums).reshape(-1, 3)
    assert rgb[0].tolist() == [0, 255, 255]
    assert rgb[1].tolist() == [int(256 * 0.5 ** (1 / 2.2)), int(256 * 0.25 ** (1 / 2.2)), int(256 * 1e-9 ** (1 / 2.2))]
    assert rgb[2].tolist() == [0, 0, 255]
    assert rgb[3].tolist()[1:] == [int(256 * 0.2 ** (1 / 2.2)), 217]
    # the fixed-algorithm x^(1/2.2) (rt_shared_math.h, shared with the device) quantises like libm's pow
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.uniform(0, 1.2, 300000), 10.0 ** rng.uniform(-12, 3, 100000)])
    x = np.resize(x, (x.size // 3) * 3)
    got = rt.resolve_rgb8_host(x.size // 3, 1, 1, x).reshape(-1)
    want = (256.0 * np.clip(x ** (1 / 2.2), 0.0, 0.999)).astype(np.uint8)
    assert (got != want).sum() == 0  # (only a value within an ulp or two of a quantisation step could land on the other side)
    # PNG writer round trip
    from PIL import Image
    img = np.random.default_rng(0).integers(0, 256, (37, 53, 3), dtype=np.uint8)
    img[5:20, 5:30] = 128
    path = tmp_path / "x.png"
    rt.write_png(path, img)
    back = np.asarray(Image.open(path).convert("RGB"))
    assert back.shape == img.shape and np.array_equal(back, img)


def test_image_ingest(rt, tmp_path):
    lib = rt.host_lib()
    w, h = C.c_int32(), C.c_int32()
    a = np.zeros(64 * 32 * 3, dtype=np.uint8); b = np.zeros_like(a)
    assert lib.rth_synthetic_earth(64, 32, a.ctypes.data_as(C.POINTER(C.c_uint8))) == 0
    assert lib.rth_load_image(b"synthetic:64x32", C.byref(w), C.byref(h), b.ctypes.data_as(C.POINTER(C.c_uint8)), b.size) == 0
    assert (w.value, h.value) == (64, 32) and np.array_equal(a, b) and len(np.unique(a)) > 20
    ppm = tmp_path / "t.ppm"
    ppm.write_bytes(b"P6\n# comment\n3 2\n255\n" + bytes(range(18)))
    c = np.zeros(18, dtype=np.uint8)
    assert lib.rth_load_image(str(ppm).encode(), C.byref(w), C.byref(h), c.ctypes.data_as(C.POINTER(C.c_uint8)), 18) == 0
    assert (w.value, h.value) == (3, 2) and c.tolist() == list(range(18))
    assert lib.rth_load_image(b"/nonexistent.jpg", C.byref(w), C.byref(h), None, 0) != 0
    assert b"cannot open" in lib.rth_last_error()
    with pytest.raises(rt.RtError):
        rt.HostScene(2, earth_image="/nonexistent.jpg")


def test_jpeg_decoder_matches_pillow(rt, tmp_path):
    """ImageTexture ingest (reference: image::io::Reader::open(path).decode(), src/texture.rs:78).  The decoder follows
    libjpeg's arithmetic (islow IDCT, fancy chroma upsampling, fixed-point colour conversion), so it must agree with
    Pillow's libjpeg decode of the same file — exactly, on every sampling mode."""
    from PIL import Image
    lib = rt.host_lib()
    rng = np.random.default_rng(0)
    H, W = 123, 211                                     # not a multiple of any MCU size
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([128 + 100 * np.sin(xx / 17.0) * np.cos(yy / 11.0), 128 + 90 * np.cos(xx / 23.0 + yy / 19.0),
                    (xx * 3 + yy * 5) % 256], -1)
    img = np.clip(img + rng.normal(0, 12, img.shape), 0, 255).astype(np.uint8)
    cases = {"444": dict(subsampling=0, quality=92), "420": dict(subsampling=2, quality=85),
             "422": dict(subsampling=1, quality=60), "420_restart": dict(subsampling=2, quality=85, restart_marker_rows=1),
             "gray": dict(quality=80), "tiny": dict(subsampling=2, quality=90)}
    for name, kw in cases.items():
        path = tmp_path / f"{name}.jpg"
        src = img[:, :, 0] if name == "gray" else (img[:5, :3] if name == "tiny" else img)
        try:
            Image.fromarray(src).save(path, "JPEG", **kw)
        except TypeError:                               # older Pillow: no restart_marker_rows
            kw.pop("restart_marker_rows", None)
            Image.fromarray(src).save(path, "JPEG", **kw)
        want = np.asarray(Image.open(path).convert("RGB"))
        w, h = C.c_int32(), C.c_int32()
        assert lib.rth_load_image(str(path).encode(), C.byref(w), C.byref(h), None, 0) == 0, lib.rth_last_error()
        got = np.zeros(w.value * h.value * 3, dtype=np.uint8)
        assert lib.rth_load_image(str(path).encode(), C.byref(w), C.byref(h), got.ctypes.data_as(C.POINTER(C.c_uint8)), got.size) == 0
        assert (h.value, w.value) == want.shape[:2], name
        assert np.array_equal(got.reshape(want.shape), want), name
    # progressive files (spectral selection + successive approximation, T.81 Annex G; Pillow writes libjpeg's standard ten-scan
    # script): the same coefficients arrive in another order, so the decode must equal Pillow's here too
    for name, src, kw in [("prog_420", img, dict(quality=85)), ("prog_444", img, dict(subsampling=0, quality=95)),
                          ("prog_gray", img[:, :, 1], dict(quality=70)), ("prog_tiny", img[:9, :17], dict(quality=90)),
                          ("prog_422_lowq", img, dict(subsampling=1, quality=15))]:
        path = tmp_path / f"{name}.jpg"
        Image.fromarray(src).save(path, "JPEG", progressive=True, **kw)
        assert b"\xff\xc2" in path.read_bytes()           # really a progressive frame
        want = np.asarray(Image.open(path).convert("RGB"))
        w, h = C.c_int32(), C.c_int32()
        assert lib.rth_load_image(str(path).encode(), C.byref(w), C.byref(h), None, 0) == 0, lib.rth_last_error()
        got = np.zeros(w.value * h.value * 3, dtype=np.uint8)
        assert lib.rth_load_image(str(path).encode(), C.byref(w), C.byref(h), got.ctypes.data_as(C.POINTER(C.c_uint8)), got.size) == 0
        assert np.array_equal(got.reshape(want.shape), want), name
    # a truncated progressive file is an error, not a crash; an arithmetic-coded frame is refused by name
    data = (tmp_path / "prog_420.jpg").read_bytes()
    (tmp_path / "cut.jpg").write_bytes(data[:len(data) // 20])
    w, h = C.c_int32(), C.c_int32()
    assert lib.rth_load_image(str(tmp_path / "cut.jpg").encode(), C.byref(w), C.byref(h), None, 0) != 0
    at = data.index(b"\xff\xc4")                      # the first Huffman table: claim 255 codes of length one
    (tmp_path / "huff.jpg").write_bytes(data[:at + 5] + b"\xff" + data[at + 6:])
    assert lib.rth_load_image(str(tmp_path / "huff.jpg").encode(), C.byref(w), C.byref(h), None, 0) != 0
    assert b"Huffman" in lib.rth_last_error()
    (tmp_path / "arith.jpg").write_bytes(data.replace(b"\xff\xc2", b"\xff\xca", 1))
    assert lib.rth_load_image(str(tmp_path / "arith.jpg").encode(), C.byref(w), C.byref(h), None, 0) != 0
    assert b"unsupported JPEG process" in lib.rth_last_error()
    # and a texture built from a JPEG reaches the scene description
    Image.fromarray(img).save(tmp_path / "earth.jpg", "JPEG", quality=90)
    hs = rt.HostScene(2, spp=1, earth_image=str(tmp_path / "earth.jpg"))
    assert (hs.desc.images[0].width, hs.desc.images[0].height) == (W, H)


def test_png_reader_matches_pillow(rt, tmp_path):
    """ImageTexture ingest of PNG files (the reference opens any format the `image` crate knows, src/texture.rs:78, and reads texels as
    RGBA8 through get_pixel: grey replicated, palettes looked up, alpha dropped): grey, grey + alpha, RGB, RGBA and palette images come
    out as Pillow's convert("RGB") has them; so does every bit depth, plain or Adam7-interlaced; the library's own PNG writer round-trips;
    malformed files are refused by name."""
    from PIL import Image
    lib = rt.host_lib()
    rng = np.random.default_rng(5)
    H, W = 37, 53
    rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    rgb[5:20, 7:30] = (200, 30, 90)  # a flat patch: rows where the Sub / Up / Paeth filters get picked

    def load(path):
        w, h = C.c_int32(), C.c_int32()
        assert lib.rth_load_image(str(path).encode(), C.byref(w), C.byref(h), None, 0) == 0, lib.rth_last_error()
        got = np.zeros(w.value * h.value * 3, dtype=np.uint8)
        assert lib.rth_load_image(str(path).encode(), C.byref(w), C.byref(h), got.ctypes.data_as(C.POINTER(C.c_uint8)), got.size) == 0
        return got.reshape(h.value, w.value, 3)

    images = {"rgb": Image.fromarray(rgb), "rgba": Image.fromarray(np.dstack([rgb, rng.integers(0, 256, (H, W), dtype=np.uint8)])),
              "grey": Image.fromarray(rgb[:, :, 0]), "grey_alpha": Image.fromarray(np.dstack([rgb[:, :, 0], rgb[:, :, 1]]), "LA"),
              "palette": Image.fromarray(rgb).quantize(64)}
    for name, im in images.items():
        path = tmp_path / f"{name}.png"
        im.save(path, "PNG", optimize=(name == "rgb"))
        assert np.array_equal(load(path), np.asarray(Image.open(path).convert("RGB"))), name
    # the library's own writer (adaptive filters) and its reader are inverses
    own = tmp_path / "own.png"
    assert lib.rth_write_png(str(own).encode(), W, H, np.ascontiguousarray(rgb).ctypes.data_as(C.POINTER(C.c_uint8))) == 0
    assert np.array_equal(load(own), rgb) and np.array_equal(np.asarray(Image.open(own).convert("RGB")), rgb)
    # every other form the standard allows — 1 / 2 / 4-bit grey and palette, 16-bit samples, Adam7 interlacing, all five row filters —
    # written by a small PNG writer of the test's own, so that the expected texels follow from the source arrays: low-depth grey scaled
    # to 0..255, a 16-bit sample v as (v + 128) // 257 (the `image` crate's u16 -> u8)
    import struct, zlib

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body))

    def paeth(a, b, c):
        pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
        return a if pa <= pb and pa <= pc else (b if pb <= pc else c)

    def pack_rows(samples, depth):  # samples: (h, w, channels) integers -> one bytes object per row
        rows = []
        for line in samples.reshape(samples.shape[0], -1):
            if depth == 16:
                rows.append(b"".join(struct.pack(">H", int(v)) for v in line))
            elif depth == 8:
                rows.append(bytes(int(v) for v in line))
            else:
                bits = "".join(format(int(v), f"0{depth}b") for v in line)
                bits += "0" * (-len(bits) % 8)
                rows.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
        return rows

    def filtered(rows, fbpp, first_filter):
        out, prev = b"", bytes(len(rows[0])) if rows else b""
        for y, cur in enumerate(rows):
            f = (first_filter + y) % 5
            line = bytearray()
            for x, v in enumerate(cur):
                a = cur[x - fbpp] if x >= fbpp else 0
                c = prev[x - fbpp] if x >= fbpp else 0
                pred = (0, a, prev[x], (a + prev[x]) >> 1, paeth(a, prev[x], c))[f]
                line.append((v - pred) & 255)
            out += bytes([f]) + bytes(line)
            prev = cur
        return out

    def make_png(samples, depth, colour, interlace, palette=None):
        h, w, ch = samples.shape
        fbpp = max(1, ch * depth // 8)
        body = b""
        if interlace:
            for k, (x0, y0, dx, dy) in enumerate([(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]):
                sub = samples[y0::dy, x0::dx]
                if sub.size:
                    body += filtered(pack_rows(sub, depth), fbpp, k)
        else:
            body = filtered(pack_rows(samples, depth), fbpp, 1)
        png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, colour, 0, 0, 1 if interlace else 0))
        if palette is not None:
            png += chunk(b"PLTE", bytes(int(v) for v in palette.reshape(-1)))
        half = len(zlib.compress(body)) // 2
        z = zlib.compress(body)
        return png + chunk(b"IDAT", z[:half]) + chunk(b"IDAT", z[half:]) + chunk(b"IEND", b"")

    H2, W2 = 19, 21  # (not a multiple of 8: the Adam7 passes are ragged, two of them one column wide at the edge)
    for depth, colour, ch in [(1, 0, 1), (2, 0, 1), (4, 0, 1), (16, 0, 1), (16, 2, 3), (16, 4, 2), (16, 6, 4), (8, 2, 3), (8, 6, 4),
                              (1, 3, 1), (2, 3, 1), (4, 3, 1), (8, 3, 1)]:
        for interlace in (False, True):
            smp = rng.integers(0, 1 << depth, (H2, W2, ch))
            pal = rng.integers(0, 256, (1 << depth, 3)) if colour == 3 else None
            path = tmp_path / f"d{depth}_c{colour}_{int(interlace)}.png"
            path.write_bytes(make_png(smp, depth, colour, interlace, pal))
            if colour == 3:
                want = pal[smp[:, :, 0]]
            else:
                to8 = {1: lambda v: v * 255, 2: lambda v: v * 85, 4: lambda v: v * 17, 8: lambda v: v, 16: lambda v: (v + 128) // 257}[depth]
                want = to8(smp[:, :, :3]) if ch >= 3 else np.repeat(to8(smp[:, :, :1]), 3, axis=2)
            assert np.array_equal(load(path), want.astype(np.uint8)), path.name
            if depth <= 8:  # (the test's writer itself, against Pillow)
                assert np.array_equal(np.asarray(Image.open(path).convert("RGB")), want.astype(np.uint8)), path.name
    # a one-pixel interlaced image has six empty passes
    path = tmp_path / "one.png"
    path.write_bytes(make_png(np.array([[[7, 200, 31]]]), 8, 2, True))
    assert load(path).tolist() == [[[7, 200, 31]]]
    # a header that announces an absurd size is refused before anything is allocated for it
    (tmp_path / "huge.png").write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 60000, 60000, 8, 2, 0, 0, 0))
                                        + chunk(b"IDAT", zlib.compress(b"\0")) + chunk(b"IEND", b""))
    w, h = C.c_int32(), C.c_int32()
    assert lib.rth_load_image(str(tmp_path / "huge.png").encode(), C.byref(w), C.byref(h), None, 0) != 0 and b"too large" in lib.rth_last_error()
    # refused, with the reason: a depth the colour type does not have, a damaged file
    w, h = C.c_int32(), C.c_int32()
    (tmp_path / "bad_depth.png").write_bytes(make_png(rng.integers(0, 16, (4, 4, 3)), 4, 2, False))
    assert lib.rth_load_image(str(tmp_path / "bad_depth.png").encode(), C.byref(w), C.byref(h), None, 0) != 0 and b"bit depth" in lib.rth_last_error()
    damaged = bytearray((tmp_path / "rgb.png").read_bytes()); damaged[60] ^= 0x40
    (tmp_path / "damaged.png").write_bytes(bytes(damaged))
    assert lib.rth_load_image(str(tmp_path / "damaged.png").encode(), C.byref(w), C.byref(h), None, 0) != 0 and b"PNG" in lib.rth_last_error()
    # ... and a PNG texture reaches the scene description like a JPEG does
    hs = rt.HostScene(2, spp=1, earth_image=str(tmp_path / "rgb.png"))
    assert (hs.desc.images[0].width, hs.desc.images[0].height) == (W, H)
