"""Known-answer tests of the oracle's building blocks from closed-form geometry — the reference has no tests of
its own, so these pin the restatement to the mathematics the reference implements."""
import ctypes as C
import math

import numpy as np
import pytest

D3 = C.c_double * 3
D7 = C.c_double * 7
D10 = C.c_double * 10


def v(*x):
    return D3(*map(float, x))


def make_sphere(rt, c, r, moving=None):
    s = rt.Sphere()
    s.center = rt.Vec3(*c); s.radius = r
    if moving:
        s.center_vec = rt.Vec3(*(m - a for m, a in zip(moving, c))); s.is_moving = 1
    s.material = 0
    return s


def make_quad(rt, q, u, vv):
    q, u, vv = map(np.array, (q, u, vv))
    n = np.cross(u, vv)
    normal = n * (1.0 / math.sqrt(n @ n))
    w = n * (1.0 / (n @ n))
    r = rt.Quad()
    r.q = rt.Vec3(*q); r.u = rt.Vec3(*u); r.v = rt.Vec3(*vv); r.w = rt.Vec3(*w); r.normal = rt.Vec3(*normal)
    r.d = float(normal @ q); r.material = 0
    return r


def test_sphere_hit(rt, oracle):
    L = oracle.lib()
    s = make_sphere(rt, (0, 0, -5), 1.0)
    out = D10()
    # head-on from the origin: roots at t = 4 and 6 (src/sphere.rs:58-83)
    assert L.orc_kat_sphere_hit(C.byref(s), v(0, 0, 0), v(0, 0, -1), 0.0, 0.001, math.inf, out)
    assert out[0] == 4.0 and tuple(out[1:4]) == (0.0, 0.0, -4.0) and tuple(out[4:7]) == (0.0, 0.0, 1.0) and out[9] == 1.0
    # un-normalised direction: t scales inversely (src/camera.rs:122 never normalises)
    assert L.orc_kat_sphere_hit(C.byref(s), v(0, 0, 0), v(0, 0, -2), 0.0, 0.001, math.inf, out) and out[0] == 2.0
    # first root excluded by tmin -> second root, seen from inside: normal flipped, front_face false
    assert L.orc_kat_sphere_hit(C.byref(s), v(0, 0, 0), v(0, 0, -1), 0.0, 4.5, math.inf, out)
    assert out[0] == 6.0 and tuple(out[4:7]) == (0.0, 0.0, 1.0) and out[9] == 0.0
    # the interval is open at both ends (Interval::surrounds, src/interval.rs:44-46)
    assert not L.orc_kat_sphere_hit(C.byref(s), v(0, 0, 0), v(0, 0, -1), 0.0, 4.0, 6.0, out)
    assert not L.orc_kat_sphere_hit(C.byref(s), v(0, 2, 0), v(0, 0, -1), 0.0, 0.001, math.inf, out)  # misses
    # uv: theta = acos(-y), phi = atan2(-z, x) + pi (src/sphere.rs:48-52)
    s0 = make_sphere(rt, (0, 0, 0), 1.0)
    for p, (u, vv) in {(1, 0, 0): (0.5, 0.5), (0, 1, 0): (0.5, 1.0), (0, 0, 1): (0.25, 0.5), (-1, 0, 0): (0.0, 0.5),
                       (0, 0, -1): (0.75, 0.5), (0, -1, 0): (0.5, 0.0)}.items():
        o = tuple(2.0 * c for c in p); d = tuple(-c for c in p)
        assert L.orc_kat_sphere_hit(C.byref(s0), v(*o), v(*d), 0.0, 0.001, math.inf, out)
        assert out[7] == pytest.approx(u, abs=1e-15) and out[8] == pytest.approx(vv, abs=1e-15), p
    # moving centre: center + center_vec * time (src/sphere.rs:53-55)
    m = make_sphere(rt, (0, 0, -5), 1.0, moving=(0, 2, -5))
    assert not L.orc_kat_sphere_hit(C.byref(m), v(0, 0, 0), v(0, 0, -1), 1.0, 0.001, math.inf, out)
    assert L.orc_kat_sphere_hit(C.byref(m), v(0, 0, 0), v(0, 0, -1), 0.25, 0.001, math.inf, out)
    assert out[0] == pytest.approx(5 - math.sqrt(0.75))


def test_quad_hit(rt, oracle):
    L = oracle.lib()
    q = make_quad(rt, (-1, -1, -3), (2, 0, 0), (0, 2, 0))  # normal +z
    out = D10()
    assert L.orc_kat_quad_hit(C.byref(q), v(0, 0, 0), v(0, 0, -1), 0.001, math.inf, out)
    assert out[0] == 3.0 and (out[7], out[8]) == (0.5, 0.5) and tuple(out[4:7]) == (0.0, 0.0, 1.0) and out[9] == 1.0
    # two-sided: from behind the normal is flipped (src/hittable.rs:22-31)
    assert L.orc_kat_quad_hit(C.byref(q), v(0, 0, -6), v(0, 0, 1), 0.001, math.inf, out)
    assert tuple(out[4:7]) == (0.0, 0.0, -1.0) and out[9] == 0.0
    # alpha, beta in [0, 1] inclusive (src/quad.rs:122-127); t interval inclusive (Interval::contains)
    assert L.orc_kat_quad_hit(C.byref(q), v(1, 1, 0), v(0, 0, -1), 0.001, math.inf, out) and (out[7], out[8]) == (1.0, 1.0)
    assert not L.orc_kat_quad_hit(C.byref(q), v(1.0000001, 0, 0), v(0, 0, -1), 0.001, math.inf, out)
    assert L.orc_kat_quad_hit(C.byref(q), v(0, 0, 0), v(0, 0, -1), 0.001, 3.0, out)
    assert not L.orc_kat_quad_hit(C.byref(q), v(0, 0, 0), v(0, 0, -1), 0.001, 2.9999, out)
    # |denom| < 1e-8: parallel rays never hit (src/quad.rs:110)
    assert not L.orc_kat_quad_hit(C.byref(q), v(0, 0, 0), v(1, 0, -1e-9), 0.001, math.inf, out)


def test_aabb_hit_modes(rt, oracle):
    L = oracle.lib()
    b = rt.Aabb((C.c_double * 3)(0, 0, 0), (C.c_double * 3)(1, 1, 1))
    inf = math.inf
    for mode in (oracle.ORC_AABB_REFERENCE, oracle.ORC_AABB_TIGHT):
        assert L.orc_kat_aabb_hit(C.byref(b), v(0.5, 0.5, -1), v(0, 0, 1), 0.001, inf, mode)       # axis-parallel: 1/0 = inf
        assert not L.orc_kat_aabb_hit(C.byref(b), v(0.5, 0.5, -1), v(0, 0, 1), 0.001, 0.9, mode)   # box beyond tmax
        assert not L.orc_kat_aabb_hit(C.byref(b), v(0.5, 0.5, 2), v(0, 0, 1), 0.001, inf, mode)    # box behind
        assert L.orc_kat_aabb_hit(C.byref(b), v(2, 2, 2), v(-1, -1, -1), 0.001, inf, mode)         # negative direction
    # the reference never narrows the interval across axes (src/aabb.rs:64-84): a ray whose LINE misses the box
    # can pass the reference test (each slab alone overlaps (tmin, tmax)) and fail the textbook one
    o, d = v(-1, 1.5, 0.5), v(1, -0.1, 0)   # x-slab t in [1,2], y-slab t in [5,15]: disjoint
    assert L.orc_kat_aabb_hit(C.byref(b), o, d, 0.001, inf, oracle.ORC_AABB_REFERENCE)
    assert not L.orc_kat_aabb_hit(C.byref(b), o, d, 0.001, inf, oracle.ORC_AABB_TIGHT)


def test_reflect_refract_schlick(rt, oracle):
    L = oracle.lib()
    out = D3()
    L.orc_kat_reflect(v(1, -1, 0), v(0, 1, 0), out); assert tuple(out) == (1.0, 1.0, 0.0)
    # Snell: sin(theta_t) = eta * sin(theta_i)
    th = 0.7; eta = 1 / 1.5
    L.orc_kat_refract(v(math.sin(th), -math.cos(th), 0), v(0, 1, 0), eta, out)
    assert out[0] == pytest.approx(eta * math.sin(th), rel=1e-15)
    assert math.hypot(out[0], out[1]) == pytest.approx(1.0, rel=1e-15) and out[1] < 0
    r0 = ((1 - 1.5) / (1 + 1.5)) ** 2
    assert L.orc_kat_reflectance(1.0, 1.5) == pytest.approx(r0, rel=1e-15)       # normal incidence
    assert L.orc_kat_reflectance(0.0, 1.5) == pytest.approx(1.0, rel=1e-15)       # grazing
    assert L.orc_kat_reflectance(0.5, 1.5) == pytest.approx(r0 + (1 - r0) * 0.5 ** 5, rel=1e-15)


def test_camera_ray(rt, oracle):
    L = oracle.lib()
    out = D7()
    hs = rt.HostScene(6, width=100, spp=1)          # Cornell: no defocus
    cam = hs.camera
    n = L.orc_kat_camera_ray(C.byref(cam), 5, 10, 20, 0, out)
    assert n == 3                                    # px, py, time (src/camera.rs:113-123)
    assert tuple(out[0:3]) == cam.center.tuple()
    # the sample lies within half a pixel of the pixel centre
    pc = np.array(cam.pixel00_loc.tuple()) + 10 * np.array(cam.pixel_delta_u.tuple()) + 20 * np.array(cam.pixel_delta_v.tuple())
    off = np.array(out[0:3]) + np.array(out[3:6]) - pc
    du = np.array(cam.pixel_delta_u.tuple()); dv = np.array(cam.pixel_delta_v.tuple())
    assert abs(off @ du / (du @ du)) <= 0.5 and abs(off @ dv / (dv @ dv)) <= 0.5
    assert 0.0 <= out[6] < 1.0
    hs0 = rt.HostScene(0, width=100, spp=1)         # random_balls: defocus_angle 0.6
    n = L.orc_kat_camera_ray(C.byref(hs0.camera), 5, 10, 20, 0, out)
    assert n >= 5 and (n - 3) % 2 == 0              # + 2 draws per disk-rejection round
    c = np.array(hs0.camera.center.tuple())
    assert 0 < np.linalg.norm(np.array(out[0:3]) - c) < 0.06   # on the lens disk: radius 10*tan(0.3 deg) = 0.052
    # image_height = (w / aspect) as usize: truncation (src/camera.rs:69)
    assert rt.HostScene(5, spp=1).height == 337 and rt.HostScene(0, width=400, spp=1).height == 225


def perlin_tables(rt, seed=3):
    rng = np.random.default_rng(seed)
    p = rt.Perlin()
    for i in range(256):
        p.ranvec[i] = rt.Vec3(*rng.uniform(-1, 1, 3))
    for name in ("perm_x", "perm_y", "perm_z"):
        perm = rng.permutation(256)
        for i in range(256):
            getattr(p, name)[i] = int(perm[i])
    return p


def test_perlin(rt, oracle):
    L = oracle.lib()
    p = perlin_tables(rt)
    # gradient noise vanishes on the integer lattice (every weight vector there is zero or has zero weight)
    for q in [(0, 0, 0), (3, -2, 7), (-255, 256, 1000)]:
        assert L.orc_kat_perlin_noise(C.byref(p), v(*q)) == 0.0
    # independent restatement of src/perlin.rs:27-50,:81-100 in numpy
    def noise(pt):
        i, j, k = (math.floor(c) for c in pt)
        u, vv, w = pt[0] - i, pt[1] - j, pt[2] - k
        uu, vw, ww = (t * t * (3 - 2 * t) for t in (u, vv, w))
        acc = 0.0
        for di in range(2):
            for dj in range(2):
                for dk in range(2):
                    g = p.ranvec[p.perm_x[(i + di) & 255] ^ p.perm_y[(j + dj) & 255] ^ p.perm_z[(k + dk) & 255]]
                    acc += ((di * uu + (1 - di) * (1 - uu)) * (dj * vw + (1 - dj) * (1 - vw)) * (dk * ww + (1 - dk) * (1 - ww))
                            * (g.x * (u - di) + g.y * (vv - dj) + g.z * (w - dk)))
        return acc
    rng = np.random.default_rng(4)
    for pt in rng.uniform(-300, 300, (200, 3)):
        assert L.orc_kat_perlin_noise(C.byref(p), v(*pt)) == pytest.approx(noise(pt), rel=1e-13, abs=1e-15)
        t = 0.0; wgt = 1.0; q = np.array(pt)
        for _ in range(7):
            t += wgt * noise(q); wgt *= 0.5; q = q * 2
        assert L.orc_kat_perlin_turbulence(C.byref(p), v(*pt), 7) == pytest.approx(abs(t), rel=1e-12, abs=1e-15)


def test_textures(rt, oracle):
    L = oracle.lib()
    out = D3()
    # checker is spatial, parity of floor(p/scale) sums, negative odd sums are "odd" (src/texture.rs:59-69)
    hs = rt.HostScene(1, width=16, spp=1)
    d = hs.desc
    chk = [i for i in range(d.n_textures) if d.textures[i].kind == rt.RT_TEXTURE_CHECKER][0]
    even, odd = (0.2, 0.3, 0.1), (0.9, 0.9, 0.9)
    for pt, want in [((0.1, 0.1, 0.1), even), ((0.4, 0.1, 0.1), odd), ((-0.1, 0.1, 0.1), odd), ((-0.1, -0.1, 0.1), even),
                     ((-0.1, -0.1, -0.1), odd), ((0.33, 0.33, 0.0), even)]:
        L.orc_kat_texture_value(C.byref(d), chk, 0.0, 0.0, v(*pt), out)
        assert tuple(out) == want, pt
    # image: u clamped, v flipped, nearest texel by truncation, (c/255)^2.2 (src/texture.rs:82-92, src/color.rs:21-26)
    he = rt.HostScene(2, width=16, spp=1, earth_image="synthetic:8x4")
    de = he.desc
    img = [i for i in range(de.n_textures) if de.textures[i].kind == rt.RT_TEXTURE_IMAGE][0]
    im = de.images[0]
    assert (im.width, im.height) == (8, 4)
    px = np.ctypeslib.as_array(im.rgb, shape=(4, 8, 3))
    for (u, vv), (i, j) in {(0.0, 1.0): (0, 0), (1.0, 0.0): (7, 3), (0.5, 0.5): (3, 1), (2.0, -1.0): (7, 3), (0.999, 0.001): (6, 2)}.items():
        L.orc_kat_texture_value(C.byref(de), img, u, vv, v(0, 0, 0), out)
        want = tuple((float(c) / 255.0) ** 2.2 for c in px[j, i])
        assert tuple(out) == pytest.approx(want, rel=1e-15), (u, vv)


def test_furnace(rt, oracle):
    """A white (albedo 1) Lambertian world under a constant sky returns exactly the sky for every path that
    escapes: attenuation (1,1,1) multiplies exactly."""
    hs = rt.HostScene(1, width=32, spp=4, depth=50)  # two big spheres
    d = hs.desc
    for i in range(d.n_textures):
        if d.textures[i].kind == rt.RT_TEXTURE_SOLID:
            d.textures[i].color = rt.Vec3(1.0, 1.0, 1.0)
    out = oracle.render(hs, rt.render_params(seed=2)).reshape(-1, 3)
    bg = np.array(hs.camera.background.tuple())
    k = out[:, 2] / bg[2]                            # number of escaped paths of the pixel (blue = 1.0)
    assert np.all((k == np.round(k)) & (k >= 0) & (k <= 4))
    # 0.7 and 0.8 are not dyadic: the in-order sum of k copies differs from k * x by rounding only
    assert np.allclose(out, k[:, None] * bg[None, :], rtol=1e-15, atol=0)
    assert k.mean() > 3.5


def test_single_object_furnaces(rt, oracle):
    """One convex object under a white sky (1, 1, 1): what each material returns is known in closed form from the reference's
    scatter functions — a convex Lambertian or mirror sphere is left after exactly one bounce (src/material.rs:26-64: the
    scattered ray points away from the surface), so every sample that hits it IS its albedo; glass attenuates by Color::ONE
    (src/material.rs:99), so it returns the sky unless the depth runs out; a light returns its emission, whatever the sky
    (src/renderer.rs:148-150); a white Isotropic medium (src/material.rs:132-138) changes nothing."""
    import custom_scenes
    cam = rt.HostScene(4, width=48, spp=6, depth=50)  # the `quads` camera: looks down -z from (0, 0, 9)
    spp = 6

    def frame(build):
        s = custom_scenes.CustomScene(cam, spp=spp, depth=50, background=(1.0, 1.0, 1.0))
        return oracle.render(s.finish(build(s)), rt.render_params(seed=9)).reshape(-1, 3)

    def summed(x):  # the in-order sum of spp copies of x (src/renderer.rs:35-40)
        acc = 0.0
        for _ in range(spp):
            acc += x
        return acc

    sky = summed(1.0)

    def check_flat(out, colour):
        """Pixels whose samples all hit the object are exactly `colour`; silhouette pixels (some samples miss) lie between it and the sky."""
        hit = out[:, 0] != sky
        assert 0.05 < hit.mean() < 0.9
        for c in range(3):
            full = out[hit, c] == summed(colour[c])
            assert full.mean() > 0.6  # (the rest are silhouette pixels)
            lo, hi = sorted((summed(colour[c]), sky))
            assert np.all((out[hit, c] >= lo - 1e-12) & (out[hit, c] <= hi + 1e-12))

    check_flat(frame(lambda s: s.sphere((0, 0, 0), 3.0, s.lambertian(0.8, 0.3, 0.55))), (0.8, 0.3, 0.55))
    check_flat(frame(lambda s: s.sphere((0, 0, 0), 3.0, s.metal(0.9, 0.6, 0.2, 0.0))), (0.9, 0.6, 0.2))
    out = frame(lambda s: s.sphere((0, 0, 0), 3.0, s.dielectric(1.5)))
    assert out.max() == sky and (out == sky).mean() > 0.999 and out.min() >= summed(1.0) - 1.0  # (a path may run out of depth inside)
    check_flat(frame(lambda s: s.quad((-2, -2, 0), (4, 0, 0), (0, 4, 0), s.light(4.0, 2.0, 0.5))), (4.0, 2.0, 0.5))
    out = frame(lambda s: s.list([s.medium(s.sphere((0, 0, 0), 3.0, s.dielectric(1.5)), 0.8, 1.0, 1.0, 1.0)]))
    assert np.all(out == sky)
