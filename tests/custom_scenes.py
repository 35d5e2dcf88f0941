"""Hand-made rt_scene_desc graphs for cases the reference's nine scenes never produce: primitives hit at exactly
the same t (coincident copies, overlapping coplanar quads), single-primitive worlds, empty frames.

A CustomScene quacks like rt.HostScene (desc, camera, width, height) for rt.DeviceScene and the oracle."""
from __future__ import annotations

import ctypes as C
import importlib
import math

rt = importlib.import_module("rust-tracing_amd")


def _v(x, y, z):
    return rt.Vec3(float(x), float(y), float(z))


class CustomScene:
    def __init__(self, camera_from: "rt.HostScene", spp=4, depth=6, background=None):
        self.spheres, self.quads, self.lists, self.items = [], [], [], []
        self.translates, self.rotates, self.materials, self.textures = [], [], [], []
        self.media = []
        self.camera = rt.Camera.from_buffer_copy(camera_from.camera)
        self.camera.samples_per_pixel = spp
        self.camera.max_depth = depth
        if background is not None:
            self.camera.background = _v(*background)
        self.desc = None

    width = property(lambda self: self.camera.image_width)
    height = property(lambda self: self.camera.image_height)

    # ---- materials ----
    def solid(self, r, g, b):
        self.textures.append(rt.Texture(kind=rt.RT_TEXTURE_SOLID, even=-1, odd=-1, image=-1, perlin=-1, color=_v(r, g, b)))
        return len(self.textures) - 1

    def lambertian(self, r, g, b):
        self.materials.append(rt.Material(kind=rt.RT_MATERIAL_LAMBERTIAN, texture=self.solid(r, g, b)))
        return len(self.materials) - 1

    def light(self, r, g, b):
        self.materials.append(rt.Material(kind=rt.RT_MATERIAL_DIFFUSE_LIGHT, texture=self.solid(r, g, b)))
        return len(self.materials) - 1

    def metal(self, r, g, b, fuzz):
        self.materials.append(rt.Material(kind=rt.RT_MATERIAL_METAL, texture=-1, albedo=_v(r, g, b), fuzz=fuzz))
        return len(self.materials) - 1

    def isotropic(self, r, g, b):
        self.materials.append(rt.Material(kind=rt.RT_MATERIAL_ISOTROPIC, texture=self.solid(r, g, b)))
        return len(self.materials) - 1

    def dielectric(self, ir):
        self.materials.append(rt.Material(kind=rt.RT_MATERIAL_DIELECTRIC, texture=-1, ir=ir))
        return len(self.materials) - 1

    # ---- hittables: each returns an rt.Ref ----
    def sphere(self, center, radius, material):
        self.spheres.append(rt.Sphere(center=_v(*center), radius=radius, center_vec=_v(0, 0, 0), is_moving=0, material=material))
        return rt.Ref(rt.RT_HITTABLE_SPHERE, len(self.spheres) - 1)

    def quad(self, q, u, v, material):
        # Quad::new (src/quad.rs:23-43), in Python floats = f64, same operation order as the host mirror
        n = (u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0])
        n2 = n[0] * n[0] + n[1] * n[1] + n[2] * n[2]
        rl = 1.0 / math.sqrt(n2)
        normal = (n[0] * rl, n[1] * rl, n[2] * rl)
        d = normal[0] * q[0] + normal[1] * q[1] + normal[2] * q[2]
        w = (n[0] / n2, n[1] / n2, n[2] / n2)
        self.quads.append(rt.Quad(q=_v(*q), u=_v(*u), v=_v(*v), w=_v(*w), normal=_v(*normal), d=d, material=material))
        return rt.Ref(rt.RT_HITTABLE_QUAD, len(self.quads) - 1)

    def list(self, refs):
        first = len(self.items)
        self.items.extend(refs)
        self.lists.append(rt.List(first, len(refs)))
        return rt.Ref(rt.RT_HITTABLE_LIST, len(self.lists) - 1)

    def translate(self, ref, offset):
        self.translates.append(rt.Translate(object=ref, offset=_v(*offset)))
        return rt.Ref(rt.RT_HITTABLE_TRANSLATE, len(self.translates) - 1)

    def rotate_y(self, ref, degrees):
        th = math.radians(degrees)
        self.rotates.append(rt.RotateY(object=ref, sin_theta=math.sin(th), cos_theta=math.cos(th)))
        return rt.Ref(rt.RT_HITTABLE_ROTATE_Y, len(self.rotates) - 1)

    def medium(self, boundary, density, r, g, b):
        # ConstantMedium::new (src/constant_medium.rs:21-30): neg_inv_density = -1 / density, phase = Isotropic(colour)
        self.media.append(rt.ConstantMedium(boundary=boundary, neg_inv_density=-1.0 / density, phase_material=self.isotropic(r, g, b)))
        return rt.Ref(rt.RT_HITTABLE_CONSTANT_MEDIUM, len(self.media) - 1)

    def finish(self, world: "rt.Ref"):
        def arr(kind, values):
            a = (kind * max(1, len(values)))(*values)
            return a

        self._keep = dict(spheres=arr(rt.Sphere, self.spheres), quads=arr(rt.Quad, self.quads), lists=arr(rt.List, self.lists),
                          items=arr(rt.Ref, self.items), translates=arr(rt.Translate, self.translates),
                          rotates=arr(rt.RotateY, self.rotates), media=arr(rt.ConstantMedium, self.media),
                          materials=arr(rt.Material, self.materials),
                          textures=arr(rt.Texture, self.textures))
        k = self._keep
        d = rt.SceneDesc()
        d.abi_version = rt.RT_ABI_VERSION
        d.world = world
        d.n_spheres, d.spheres = len(self.spheres), k["spheres"]
        d.n_quads, d.quads = len(self.quads), k["quads"]
        d.n_lists, d.lists = len(self.lists), k["lists"]
        d.n_list_items, d.list_items = len(self.items), k["items"]
        d.n_translates, d.translates = len(self.translates), k["translates"]
        d.n_rotates, d.rotates = len(self.rotates), k["rotates"]
        d.n_media, d.media = len(self.media), k["media"]
        d.n_materials, d.materials = len(self.materials), k["materials"]
        d.n_textures, d.textures = len(self.textures), k["textures"]
        self.desc = d
        return self


def tie_scene(camera_from, order=0):
    """Every primitive has coincident copies with different materials, so every hit is an exact tie; `order` permutes the
    scan order of the world list (the winner of a tie depends on it)."""
    s = CustomScene(camera_from, spp=4, depth=6, background=(0.7, 0.8, 1.0))
    red, green, blue, white = s.lambertian(0.9, 0.1, 0.1), s.lambertian(0.1, 0.9, 0.1), s.lambertian(0.1, 0.1, 0.9), s.lambertian(0.8, 0.8, 0.8)
    mirror, glass, lamp = s.metal(0.8, 0.8, 0.3, 0.0), s.dielectric(1.5), s.light(4, 4, 4)
    back = [s.quad((-3, -3, -2), (6, 0, 0), (0, 6, 0), m) for m in (red, green, blue)]       # three coincident walls
    half = s.quad((0, -3, -2), (3, 0, 0), (0, 6, 0), mirror)                                  # coplanar, covers half of them
    floor = [s.quad((-3, -3, -2), (6, 0, 0), (0, 0, 4), m) for m in (white, red)]
    balls = [s.sphere((-1.2, -0.5, 0.0), 1.0, m) for m in (green, glass, mirror)]             # three coincident spheres
    ball2 = [s.sphere((1.3, -1.0, 0.5), 0.8, m) for m in (glass, blue)]
    top = [s.quad((-1, 2.9, -1), (2, 0, 0), (0, 0, 2), m) for m in (lamp, white)]             # a light and its opaque copy
    cube = s.translate(s.rotate_y(s.list([s.quad((-0.5, -0.5, 0.5), (1, 0, 0), (0, 1, 0), m) for m in (blue, mirror)] +
                                         [s.quad((-0.5, -0.5, -0.5), (0, 0, 1), (0, 1, 0), m) for m in (red, green)] +
                                         [s.sphere((0, 0, 0), 0.6, m) for m in (white, lamp)]), 30.0), (0.4, 1.2, 0.8))
    items = back + [half] + floor + balls + ball2 + top + [cube]
    if order == 1:
        items = items[::-1]
    elif order == 2:
        items = items[1::2] + items[0::2]
    return s.finish(s.list(items))


def single_sphere_scene(camera_from):
    s = CustomScene(camera_from, spp=4, depth=6, background=(0.7, 0.8, 1.0))
    return s.finish(s.sphere((0, 0, 0), 2.0, s.lambertian(0.5, 0.6, 0.7)))


def empty_frame_scene(camera_from):
    """A Translate around an empty list next to real geometry: the frame can never be hit."""
    s = CustomScene(camera_from, spp=4, depth=6, background=(0.7, 0.8, 1.0))
    nothing = s.translate(s.list([]), (1, 0, 0))
    return s.finish(s.list([nothing, s.sphere((0, 0, 0), 1.5, s.metal(0.7, 0.7, 0.7, 0.1)),
                            s.quad((-3, -2, -3), (6, 0, 0), (0, 0, 6), s.lambertian(0.3, 0.7, 0.3))]))


def many_spheres_scene(camera_from, n, seed=5):
    """n small spheres (Lambertian / Metal / Dielectric) scattered over a ground sphere, as one flat HittableList: sized
    by the caller so that the compiled scene does or does not fit the LDS."""
    import random
    rnd = random.Random(seed)
    s = CustomScene(camera_from, spp=4, depth=6, background=(0.7, 0.8, 1.0))
    mats = [s.lambertian(rnd.random(), rnd.random(), rnd.random()) for _ in range(12)]
    mats += [s.metal(0.5 + 0.5 * rnd.random(), 0.5 + 0.5 * rnd.random(), 0.5 + 0.5 * rnd.random(), 0.3 * rnd.random()) for _ in range(4)]
    mats += [s.dielectric(1.5)]
    items = [s.sphere((0.0, -1003.0, 0.0), 1000.0, mats[0])]
    for _ in range(n):
        x, z = rnd.uniform(-6, 6), rnd.uniform(-6, 6)
        r = rnd.uniform(0.05, 0.25)
        y = -3.0 + r + rnd.choice([0.0, 0.0, rnd.uniform(0, 3)])
        items.append(s.sphere((x, y, z), r, rnd.choice(mats)))
    return s.finish(s.list(items))


def media_scene(camera_from, order=0, nested=False):
    """Participating media in every position of the scan: bounded by a sphere, by a rotated and shifted cube, by a bare
    list of quads; first, last, next to each other; overlapping each other and solid objects.  nested: one of them sits
    inside a Translate (a case the ordered layout leaves to the reference-order walk)."""
    s = CustomScene(camera_from, spp=4, depth=8, background=(0.7, 0.8, 1.0))
    red, green, white = s.lambertian(0.9, 0.2, 0.2), s.lambertian(0.2, 0.9, 0.2), s.lambertian(0.8, 0.8, 0.8)
    mirror, glass, lamp = s.metal(0.8, 0.8, 0.6, 0.05), s.dielectric(1.5), s.light(5, 5, 5)

    def cube(lo, hi, m):
        (x0, y0, z0), (x1, y1, z1) = lo, hi
        dx, dy, dz = x1 - x0, y1 - y0, z1 - z0
        return s.list([s.quad((x0, y0, z1), (dx, 0, 0), (0, dy, 0), m), s.quad((x1, y0, z1), (0, 0, -dz), (0, dy, 0), m),
                       s.quad((x1, y0, z0), (-dx, 0, 0), (0, dy, 0), m), s.quad((x0, y0, z0), (0, 0, dz), (0, dy, 0), m),
                       s.quad((x0, y1, z1), (dx, 0, 0), (0, 0, -dz), m), s.quad((x0, y0, z0), (dx, 0, 0), (0, 0, dz), m)])

    floor = s.quad((-4, -2, -4), (8, 0, 0), (0, 0, 8), white)
    wall = s.quad((-4, -2, -3), (8, 0, 0), (0, 6, 0), red)
    light = s.quad((-1, 3.5, -1), (2, 0, 0), (0, 0, 2), lamp)
    ball, ball2 = s.sphere((-1.5, -1.0, 0.5), 1.0, glass), s.sphere((1.8, -1.2, 1.0), 0.8, mirror)
    fog_ball = s.medium(s.sphere((-1.5, -1.0, 0.5), 0.95, glass), 2.0, 0.2, 0.4, 0.9)        # inside the glass ball
    smoke = s.medium(s.translate(s.rotate_y(cube((-0.7, -0.7, -0.7), (0.7, 0.7, 0.7), white), 25.0), (0.5, -1.0, -1.0)), 1.5, 0.05, 0.05, 0.05)
    haze = s.medium(cube((-3.5, -2.0, -2.5), (3.5, 0.0, 3.0), white), 0.15, 1.0, 1.0, 1.0)   # a bare list as the boundary
    mist = s.medium(s.sphere((0, 0, 0), 30.0, white), 0.01, 0.9, 0.9, 0.9)                    # everything is inside it
    if nested:
        smoke = s.translate(s.list([smoke]), (0.0, 0.5, 0.0))
    items = [mist, floor, ball, fog_ball, wall, smoke, haze, light, ball2, s.sphere((1.8, -1.2, 1.0), 0.8, green)]
    if order == 1:
        items = items[::-1]
    elif order == 2:
        items = items[1::2] + items[0::2]
    return s.finish(s.list(items))


def nested_frames_scene(camera_from):
    """Frames inside frames (RotateY inside Translate inside RotateY inside Translate ...), a sphere and a cube at every
    level, plus a medium whose boundary is such a nest: hits are rebuilt through up to three enclosing frames."""
    s = CustomScene(camera_from, spp=4, depth=8, background=(0.7, 0.8, 1.0))
    mats = [s.lambertian(0.8, 0.3, 0.3), s.lambertian(0.3, 0.8, 0.3), s.metal(0.7, 0.7, 0.9, 0.1), s.dielectric(1.5), s.light(3, 3, 3)]

    def cube(lo, hi, m):
        (x0, y0, z0), (x1, y1, z1) = lo, hi
        dx, dy, dz = x1 - x0, y1 - y0, z1 - z0
        return [s.quad((x0, y0, z1), (dx, 0, 0), (0, dy, 0), m), s.quad((x1, y0, z1), (0, 0, -dz), (0, dy, 0), m),
                s.quad((x1, y0, z0), (-dx, 0, 0), (0, dy, 0), m), s.quad((x0, y0, z0), (0, 0, dz), (0, dy, 0), m),
                s.quad((x0, y1, z1), (dx, 0, 0), (0, 0, -dz), m), s.quad((x0, y0, z0), (dx, 0, 0), (0, 0, dz), m)]

    inner = s.list(cube((-0.3, -0.3, -0.3), (0.3, 0.3, 0.3), mats[2]) + [s.sphere((0.0, 0.6, 0.0), 0.25, mats[3])])
    level2 = s.list([s.translate(s.rotate_y(inner, 40.0), (0.8, 0.2, 0.0)), s.sphere((-0.5, 0.0, 0.3), 0.35, mats[0])] +
                    cube((-0.2, -0.9, -0.2), (0.2, -0.5, 0.2), mats[1]))
    level1 = s.list([s.rotate_y(s.translate(level2, (0.0, 0.5, -0.5)), -25.0), s.sphere((1.5, -0.5, 0.5), 0.4, mats[4])])
    outer = s.translate(s.rotate_y(level1, 15.0), (-0.5, 0.0, 0.0))
    smoke_nest = s.translate(s.rotate_y(s.list([s.translate(s.list(cube((-0.5, -0.5, -0.5), (0.5, 0.5, 0.5), mats[0])), (0.2, 0.0, 0.0))]), 35.0),
                             (-2.0, -1.0, 0.5))
    floor = s.quad((-4, -2, -4), (8, 0, 0), (0, 0, 8), mats[1])
    return s.finish(s.list([floor, outer, s.medium(smoke_nest, 1.2, 0.1, 0.1, 0.1), s.sphere((2.2, -1.3, -0.5), 0.7, mats[2])]))


def random_scene(camera_from, seed):
    """A random object graph for fuzzing the two walks against the oracle: spheres, quads and cubes, alone or grouped
    in (nested) Translate / RotateY frames, media bounded by spheres, cubes and framed cubes at random places of the
    world list, some objects duplicated exactly (ties), random materials."""
    import random
    rnd = random.Random(seed)
    s = CustomScene(camera_from, spp=2, depth=6, background=(0.7, 0.8, 1.0) if rnd.random() < 0.7 else (0.0, 0.0, 0.0))
    mats = [s.lambertian(rnd.random(), rnd.random(), rnd.random()) for _ in range(4)]
    mats += [s.metal(rnd.random(), rnd.random(), rnd.random(), rnd.choice([0.0, 0.3, 1.5])), s.dielectric(rnd.choice([1.5, 1.0 / 1.5, 2.4])),
             s.light(rnd.uniform(1, 8), rnd.uniform(1, 8), rnd.uniform(1, 8))]

    def pos(scale=2.5):
        return (rnd.uniform(-scale, scale), rnd.uniform(-scale, scale), rnd.uniform(-scale, scale))

    def cube(m):
        x0, y0, z0 = pos(2.0)
        dx, dy, dz = rnd.uniform(0.2, 1.2), rnd.uniform(0.2, 1.2), rnd.uniform(0.2, 1.2)
        x1, y1, z1 = x0 + dx, y0 + dy, z0 + dz
        return [s.quad((x0, y0, z1), (dx, 0, 0), (0, dy, 0), m), s.quad((x1, y0, z1), (0, 0, -dz), (0, dy, 0), m),
                s.quad((x1, y0, z0), (-dx, 0, 0), (0, dy, 0), m), s.quad((x0, y0, z0), (0, 0, dz), (0, dy, 0), m),
                s.quad((x0, y1, z1), (dx, 0, 0), (0, 0, -dz), m), s.quad((x0, y0, z0), (dx, 0, 0), (0, 0, dz), m)]

    def solid(depth):
        kind = rnd.choice(["sphere", "sphere", "quad", "cube", "group"] if depth < 2 else ["sphere", "quad", "cube"])
        m = rnd.choice(mats)
        if kind == "sphere":
            return [s.sphere(pos(), rnd.uniform(0.15, 0.9), m)]
        if kind == "quad":
            q = pos()
            u, v = (rnd.uniform(-1.5, 1.5), rnd.uniform(-0.3, 0.3), rnd.uniform(-1.5, 1.5)), (rnd.uniform(-0.3, 0.3), rnd.uniform(0.3, 1.5), rnd.uniform(-0.3, 0.3))
            return [s.quad(q, u, v, m)]
        if kind == "cube":
            return [s.list(cube(m))] if rnd.random() < 0.5 else cube(m)
        items = [r for _ in range(rnd.randint(0, 3)) for r in solid(depth + 1)]
        g = s.list(items)
        wrap = rnd.choice(["t", "r", "tr", "rt"])
        for w in wrap:
            g = s.translate(g, pos(1.0)) if w == "t" else s.rotate_y(g, rnd.uniform(-80, 80))
        return [g]

    items = []
    for _ in range(rnd.randint(1, 9)):
        roll = rnd.random()
        if roll < 0.2:
            boundary = rnd.choice(["sphere", "cube", "framed"])
            if boundary == "sphere":
                b = s.sphere(pos(), rnd.uniform(0.5, 3.0), mats[0])
            elif boundary == "cube":
                b = s.list(cube(mats[0]))
            else:
                b = s.translate(s.rotate_y(s.list(cube(mats[0])), rnd.uniform(-60, 60)), pos(1.0))
            items.append(s.medium(b, rnd.choice([0.05, 0.5, 3.0]), rnd.random(), rnd.random(), rnd.random()))
        else:
            new = solid(0)
            items.extend(new)
            if rnd.random() < 0.15 and new:  # an exact copy right after, or at the end: ties
                dup = new[0]
                if dup.kind == rt.RT_HITTABLE_SPHERE:
                    sp = s.spheres[dup.index]
                    items.insert(rnd.randint(0, len(items)), s.sphere((sp.center.x, sp.center.y, sp.center.z), sp.radius, rnd.choice(mats)))
                elif dup.kind == rt.RT_HITTABLE_QUAD:
                    qd = s.quads[dup.index]
                    items.insert(rnd.randint(0, len(items)), s.quad(qd.q.tuple(), qd.u.tuple(), qd.v.tuple(), rnd.choice(mats)))
    return s.finish(s.list(items))


def many_media_scene(camera_from, n=40, seed=11):
    """n solid spheres alternating with n sphere-bounded media in one flat list: 2n + steps of the world sequence, more than
    the ordered layout's kernel keeps (ORDERED_MAX_STEPS = 64), so the scene compiler turns the ordered layout down and the
    reference-order walk must find every medium's boundary sphere where the threaded layout put it."""
    import random
    rnd = random.Random(seed)
    s = CustomScene(camera_from, spp=4, depth=6, background=(0.7, 0.8, 1.0))
    mats = [s.lambertian(rnd.random(), rnd.random(), rnd.random()) for _ in range(5)] + [s.metal(0.8, 0.8, 0.8, 0.1), s.dielectric(1.5)]
    items = []
    for _ in range(n):
        items.append(s.sphere((rnd.uniform(-4, 4), rnd.uniform(-2, 2), rnd.uniform(-4, 4)), rnd.uniform(0.2, 0.6), rnd.choice(mats)))
        boundary = s.sphere((rnd.uniform(-4, 4), rnd.uniform(-2, 2), rnd.uniform(-4, 4)), rnd.uniform(0.3, 1.2), mats[0])
        items.append(s.medium(boundary, rnd.choice([0.3, 1.0, 4.0]), rnd.random(), rnd.random(), rnd.random()))
    return s.finish(s.list(items))


def many_instances_scene(camera_from, n, seed=3, spheres=True):
    """n rotated and shifted cubes (each a Translate(RotateY(list of six quads))) beside each other over a floor, some
    overlapping, two of them nested once more — the walk meets many instances per ray.  n <= 32: the world frame's
    instances are walked after its own tree (one bit each); n > 32: entered where the walk meets them.
    spheres=False: quads only, so that the quads + frames kernel renders it (the one that defers instances)."""
    import random
    rnd = random.Random(seed)
    s = CustomScene(camera_from, spp=4, depth=6, background=(0.7, 0.8, 1.0))
    mats = [s.lambertian(0.8, 0.3, 0.3), s.lambertian(0.3, 0.8, 0.3), s.metal(0.7, 0.7, 0.9, 0.2), s.light(2, 2, 2)]

    def cube(h, m):
        return [s.quad((-h, -h, h), (2 * h, 0, 0), (0, 2 * h, 0), m), s.quad((h, -h, h), (0, 0, -2 * h), (0, 2 * h, 0), m),
                s.quad((h, -h, -h), (-2 * h, 0, 0), (0, 2 * h, 0), m), s.quad((-h, -h, -h), (0, 0, 2 * h), (0, 2 * h, 0), m),
                s.quad((-h, h, h), (2 * h, 0, 0), (0, 0, -2 * h), m), s.quad((-h, -h, -h), (2 * h, 0, 0), (0, 0, 2 * h), m)]

    side = max(1, int(math.ceil(math.sqrt(n))))
    world = [s.quad((-6, -1.5, -6), (12, 0, 0), (0, 0, 12), mats[1]),
             s.sphere((0.0, 2.5, 0.0), 0.6, mats[3]) if spheres else s.quad((-0.6, 2.5, -0.6), (1.2, 0, 0), (0, 0, 1.2), mats[3])]
    for k in range(n):
        gx, gz = k % side, k // side
        h = rnd.uniform(0.15, 0.45)
        inner = s.list(cube(h, mats[k % 3]))
        if k % 7 == 3:  # one more frame inside
            extra = s.sphere((0.0, -0.4, 0.0), 0.2, mats[2]) if spheres else s.quad((-0.2, -0.4, -0.2), (0.4, 0, 0), (0, 0, 0.4), mats[2])
            inner = s.list([s.translate(s.rotate_y(inner, rnd.uniform(-60, 60)), (0.1, 0.2, 0.0)), extra])
        pos = ((gx - side / 2) * 0.9 + rnd.uniform(-0.3, 0.3), rnd.uniform(-1.0, 0.5), (gz - side / 2) * 0.9 + rnd.uniform(-0.3, 0.3))
        world.append(s.translate(s.rotate_y(inner, rnd.uniform(-90, 90)), pos))
    return s.finish(s.list(world))
