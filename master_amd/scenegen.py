"""Procedural scenes, built through the public Scene.from_arrays the way loader.cpp builds meshes
(de-indexed triangles, per-corner frames): small ones for closed-form tests and seeded stand-ins
for the BASELINE scenes whose .blend files are missing from the reference tree
(.MISSING_LARGE_BLOBS) — committed as code, never as blobs.

  atrium(n)    C4' (CrytekSponza stand-in): colonnaded hall, displaced walls/floor, one area light through a
               ceiling opening, mirror + glass spheres; >= 262 144 triangles by default.
  clutter(n)   C5' (BreakfastRoom stand-in): room with table-top clutter, two area lights, Phong + mirror + glass.
"""
import ctypes as C

import numpy as np

import master_amd as ma  # noqa: E402 (this module is part of the package; `ma` is the binding)


def _f(n, vals):
    return (C.c_float * n)(*[float(v) for v in vals])


def material(kind, diffuse=(0, 0, 0), specular=(0, 0, 0), power=0.0, ior=1.0, light_id=0):
    return ma.Material(kind, _f(3, diffuse), _f(3, specular), power, ior, 1.0, light_id, 0)


def camera(position, direction, up, fovx):
    d = np.asarray(direction, np.float64); d = d / np.linalg.norm(d)
    return ma.Camera(_f(3, position), _f(3, d), _f(3, up), fovx)


class Builder:
    """Accumulates meshes the way loader.cpp does: de-indexed triangles, per-corner frames."""

    def __init__(self):
        self.pos, self.tan, self.idx, self.off, self.mesh_mat = [], [], [], [0], []
        self.materials, self.lights, self.cameras = [], [], []

    def add_camera(self, position, direction, up=(0, 0, 1), fovx=0.6435):
        self.cameras.append(camera(position, direction, up, fovx))
        self.materials.append(material(ma.BSDF_CAMERA))  # loader.cpp:304-305

    def add_material(self, m):
        self.materials.append(m)
        return len(self.materials) - 1

    def add_mesh(self, tris, material_index, normals=None):
        """tris: [n][3][3] positions.  Frames as loader.cpp:317-342 (flat normal unless given)."""
        tris = np.asarray(tris, np.float32)
        for t_i, t in enumerate(tris):
            n = np.cross(t[1] - t[0], t[2] - t[0]); n = n / np.linalg.norm(n)
            edge = t[1] - t[0]
            for k in range(3):
                nk = n if normals is None else np.asarray(normals[t_i][k], np.float64)
                tg = edge - np.dot(nk, edge) * nk; tg = tg / np.linalg.norm(tg)
                bt = np.cross(nk, tg); bt = bt / np.linalg.norm(bt)
                self.idx.append(len(self.pos)); self.pos.append(t[k]); self.tan.append(np.concatenate([tg, nk, bt]))
        self.off.append(len(self.idx) // 3)
        self.mesh_mat.append((material_index << 2) | ma.ENTITY_MESH)

    def add_quad(self, p0, p1, p2, p3, material_index):
        self.add_mesh([[p0, p1, p2], [p0, p2, p3]], material_index)

    def add_light(self, position, direction, up, size, exitance, diffuse=True):
        """AreaLights::addLight + AreaLight::create_mesh (AreaLights.cpp:38-97)."""
        d = np.asarray(direction, np.float64); d /= np.linalg.norm(d)
        u = np.asarray(up, np.float64); u /= np.linalg.norm(u)
        t0 = np.cross(u, d); t0 /= np.linalg.norm(t0)
        mat_index = len(self.materials)
        light_id = len(self.lights)
        mid = (mat_index << 2) | ma.ENTITY_LIGHT
        self.lights.append(ma.Light(_f(3, position), _f(9, np.concatenate([t0, d, u])), _f(2, size), _f(3, exitance), 1 if diffuse else 0, mid, 0))
        self.materials.append(material(ma.BSDF_LIGHT if diffuse else ma.BSDF_SUN, light_id=light_id))
        p = np.asarray(position, np.float64); left, upv = t0 * 0.5, u * 0.5
        q = [p - size[0] * left - size[1] * upv, p + size[0] * left - size[1] * upv, p + size[0] * left + size[1] * upv, p - size[0] * left + size[1] * upv]
        base = len(self.pos)
        for v in q:
            self.pos.append(np.asarray(v, np.float32)); self.tan.append(np.concatenate([t0, d, u]))
        self.idx += [base + 0, base + 1, base + 2, base + 2, base + 3, base + 0]
        self.off.append(len(self.idx) // 3)
        self.mesh_mat.append(mid)

    def build(self):
        return ma.Scene.from_arrays(np.array(self.pos, np.float32), np.array(self.tan, np.float32), np.array(self.idx, np.uint32).reshape(-1, 3),
                                    self.off, self.mesh_mat, self.materials, self.lights, self.cameras)


def rect_irradiance(x, n, corners, radiance):
    """Lambert's formula: irradiance at point x (normal n) from a uniform Lambertian polygon."""
    x = np.asarray(x, np.float64); n = np.asarray(n, np.float64)
    v = [np.asarray(c, np.float64) - x for c in corners]
    v = [a / np.linalg.norm(a) for a in v]
    e = 0.0
    for i in range(len(v)):
        a, b = v[i], v[(i + 1) % len(v)]
        gamma = np.arccos(np.clip(np.dot(a, b), -1, 1))
        c = np.cross(a, b); c /= np.linalg.norm(c)
        e += gamma * np.dot(c, n)
    return abs(e) * 0.5 * radiance


def random_soup(n_tris, seed=0, extent=4.0, size=0.4, with_light=True):
    """Seeded triangle soup with all four surface BSDF kinds + one area light: BVH / traversal stress."""
    rng = np.random.default_rng(seed)
    b = Builder()
    b.add_camera((0, -extent * 1.5, 0), (0, 1, 0))
    mats = [b.add_material(material(ma.BSDF_DIFFUSE, diffuse=(0.7, 0.6, 0.5))),
            b.add_material(material(ma.BSDF_PHONG, diffuse=(0.3, 0.3, 0.3), specular=(0.4, 0.4, 0.4), power=20.0)),
            b.add_material(material(ma.BSDF_REFLECTION)),
            b.add_material(material(ma.BSDF_TRANSMISSION, ior=1.5))]
    c = rng.uniform(-extent, extent, (n_tris, 1, 3))
    tris = c + rng.normal(scale=size, size=(n_tris, 3, 3))
    which = rng.integers(0, 4, n_tris)
    for k in range(4):
        sel = tris[which == k]
        if len(sel):
            b.add_mesh(sel, mats[k])
    if with_light:
        b.add_light((0, 0, extent * 1.2), (0, 0, -1), (0, 1, 0), (2.0, 1.0), (10, 9, 8))
    return b.build()


# ---------------------------------------------------------------------------------------------
# large seeded stand-ins (vectorised)

def _frames(tris, normals):
    """tris [n,3,3], normals [n,3,3] -> positions [3n,3], tangents [3n,9] (loader.cpp:332-339)."""
    edge = tris[:, 1] - tris[:, 0]
    n = normals / np.linalg.norm(normals, axis=2, keepdims=True)
    e = edge[:, None, :]
    t = e - (n * e).sum(2, keepdims=True) * n
    t /= np.linalg.norm(t, axis=2, keepdims=True)
    b = np.cross(n, t)
    b /= np.linalg.norm(b, axis=2, keepdims=True)
    return tris.reshape(-1, 3).astype(np.float32), np.concatenate([t, n, b], axis=2).reshape(-1, 9).astype(np.float32)


def grid(nu, nv, fn):
    """Tessellates a parametric surface fn(u, v) -> (pos[...,3], normal[...,3]) on [0,1]^2 into 2*nu*nv triangles."""
    u, v = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing="ij")
    p, n = fn(u, v)
    a, b, c, d = (0, 0), (1, 0), (1, 1), (0, 1)

    def corner(k, arr):
        return arr[k[0]:k[0] + nu, k[1]:k[1] + nv].reshape(-1, 3)
    t1 = np.stack([corner(a, p), corner(b, p), corner(c, p)], 1); n1 = np.stack([corner(a, n), corner(b, n), corner(c, n)], 1)
    t2 = np.stack([corner(a, p), corner(c, p), corner(d, p)], 1); n2 = np.stack([corner(a, n), corner(c, n), corner(d, n)], 1)
    return np.concatenate([t1, t2]), np.concatenate([n1, n2])


def heightfield(origin, ax_u, ax_v, normal, nu, nv, amp, rng, freq=6.0):
    """Displaced rectangle: origin + u*ax_u + v*ax_v + h(u,v)*normal with a smooth random height."""
    origin, ax_u, ax_v, normal = [np.asarray(x, np.float64) for x in (origin, ax_u, ax_v, normal)]
    ph = rng.uniform(0, 2 * np.pi, (4, 2)); fr = rng.uniform(0.5 * freq, 1.5 * freq, (4, 2)); am = rng.uniform(0.3, 1.0, 4)

    def fn(u, v):
        h = sum(am[k] * np.sin(fr[k, 0] * 2 * np.pi * u + ph[k, 0]) * np.sin(fr[k, 1] * 2 * np.pi * v + ph[k, 1]) for k in range(4)) * amp / 4
        hu = sum(am[k] * fr[k, 0] * 2 * np.pi * np.cos(fr[k, 0] * 2 * np.pi * u + ph[k, 0]) * np.sin(fr[k, 1] * 2 * np.pi * v + ph[k, 1]) for k in range(4)) * amp / 4
        hv = sum(am[k] * np.sin(fr[k, 0] * 2 * np.pi * u + ph[k, 0]) * fr[k, 1] * 2 * np.pi * np.cos(fr[k, 1] * 2 * np.pi * v + ph[k, 1]) for k in range(4)) * amp / 4
        p = origin + u[..., None] * ax_u + v[..., None] * ax_v + h[..., None] * normal
        du = ax_u + hu[..., None] * normal; dv = ax_v + hv[..., None] * normal
        n = np.cross(du, dv)
        n *= np.sign((n * normal).sum(-1, keepdims=True))
        return p, n
    return grid(nu, nv, fn)


def cylinder(base, radius, height, nu, nv, flute=0.0, flutes=16):
    base = np.asarray(base, np.float64)

    def fn(u, v):
        ang = 2 * np.pi * u
        r = radius * (1 + flute * np.cos(flutes * ang))
        p = np.stack([base[0] + r * np.cos(ang), base[1] + r * np.sin(ang), base[2] + height * v], -1)
        n = np.stack([np.cos(ang), np.sin(ang), np.zeros_like(ang)], -1)
        return p, n
    return grid(nu, nv, fn)


def sphere(center, radius, nu, nv):
    center = np.asarray(center, np.float64)

    def fn(u, v):
        th = np.pi * (0.001 + 0.998 * v); ph = 2 * np.pi * u
        n = np.stack([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)], -1)
        return center + radius * n, n
    return grid(nu, nv, fn)


class FastBuilder(Builder):
    def add_tris(self, tris, normals, material_index):
        pos, tan = _frames(np.asarray(tris, np.float64), np.asarray(normals, np.float64))
        base = sum(len(p) for p in self._pos_chunks) if hasattr(self, "_pos_chunks") else 0
        if not hasattr(self, "_pos_chunks"):
            self._pos_chunks, self._tan_chunks, self._ntri = [], [], 0
        self._pos_chunks.append(pos); self._tan_chunks.append(tan)
        self._ntri += len(tris)
        self.off.append(self._ntri)
        self.mesh_mat.append((material_index << 2) | ma.ENTITY_MESH)
        return base

    def build(self):
        # lights appended by Builder.add_light live in self.pos/self.tan/self.idx; surface meshes in the chunks
        n_surface = sum(len(p) for p in self._pos_chunks)
        lpos = np.array(self.pos, np.float32).reshape(-1, 3); ltan = np.array(self.tan, np.float32).reshape(-1, 9)
        pos = np.concatenate(self._pos_chunks + [lpos]); tan = np.concatenate(self._tan_chunks + [ltan])
        idx = np.concatenate([np.arange(n_surface, dtype=np.uint32), np.array(self.idx, np.uint32) + n_surface]).reshape(-1, 3)
        return ma.Scene.from_arrays(pos, tan, idx, self.off, self.mesh_mat, self.materials, self.lights, self.cameras)

    def add_light(self, *a, **k):
        # keep offsets consistent: lights are added after all surface meshes
        n_before = self._ntri
        super().add_light(*a, **k)
        self._ntri = n_before + 2
        self.off[-1] = self._ntri


def atrium(n_target=270000, seed=1):
    rng = np.random.default_rng(seed)
    b = FastBuilder()
    b.add_camera((-14.0, -15.0, 3.0), (0.75, 0.8, -0.05), up=(0, 0, 1), fovx=1.1)
    stone = b.add_material(material(ma.BSDF_DIFFUSE, diffuse=(0.62, 0.58, 0.5)))
    floor = b.add_material(material(ma.BSDF_PHONG, diffuse=(0.35, 0.33, 0.3), specular=(0.25, 0.25, 0.25), power=40.0))
    brick = b.add_material(material(ma.BSDF_DIFFUSE, diffuse=(0.55, 0.3, 0.22)))
    mirror = b.add_material(material(ma.BSDF_REFLECTION))
    glass = b.add_material(material(ma.BSDF_TRANSMISSION, ior=1.5))
    L, H = 18.0, 9.0
    scale = (n_target / 151000.0) ** 0.5  # 151k triangles at scale 1
    g = max(4, int(96 * scale))
    b.add_tris(*heightfield((-L, -L, 0), (2 * L, 0, 0), (0, 2 * L, 0), (0, 0, 1), g, g, 0.05, rng), floor)
    for (o, au, av, nn) in (((-L, -L, 0), (2 * L, 0, 0), (0, 0, H), (0, 1, 0)), ((-L, L, 0), (2 * L, 0, 0), (0, 0, H), (0, -1, 0)),
                            ((-L, -L, 0), (0, 2 * L, 0), (0, 0, H), (1, 0, 0)), ((L, -L, 0), (0, 2 * L, 0), (0, 0, H), (-1, 0, 0))):
        b.add_tris(*heightfield(o, au, av, nn, g, max(2, g // 2), 0.25, rng), brick)
    # ceiling: four slabs around a central opening
    op = 5.0
    for (o, au, av) in (((-L, -L, H), (2 * L, 0, 0), (0, L - op, 0)), ((-L, op, H), (2 * L, 0, 0), (0, L - op, 0)),
                        ((-L, -op, H), (L - op, 0, 0), (0, 2 * op, 0)), ((op, -op, H), (L - op, 0, 0), (0, 2 * op, 0))):
        b.add_tris(*heightfield(o, au, av, (0, 0, -1), max(2, g // 2), max(2, g // 4), 0.1, rng), stone)
    cu, cv = max(6, int(64 * scale)), max(2, int(24 * scale))
    for ix in range(-2, 3):
        for iy in range(-2, 3):
            if ix == 0 and iy == 0:
                continue
            b.add_tris(*cylinder((ix * 6.0, iy * 6.0, 0.0), 0.6, H, cu, cv, flute=0.06), stone)
    b.add_tris(*sphere((0.0, 0.0, 1.6), 1.5, max(8, int(96 * scale)), max(4, int(48 * scale))), mirror)
    b.add_tris(*sphere((4.0, -3.0, 1.1), 1.0, max(8, int(64 * scale)), max(4, int(32 * scale))), glass)
    b.add_light((0.0, 0.0, H + 1.5), (0, 0, -1), (0, 1, 0), (8.0, 8.0), (60.0, 55.0, 45.0))
    return b.build()


def clutter(n_target=150000, seed=2):
    rng = np.random.default_rng(seed)
    b = FastBuilder()
    b.add_camera((-3.5, -4.2, 2.0), (0.6, 0.75, -0.25), up=(0, 0, 1), fovx=1.0)
    wall = b.add_material(material(ma.BSDF_DIFFUSE, diffuse=(0.7, 0.68, 0.62)))
    wood = b.add_material(material(ma.BSDF_PHONG, diffuse=(0.4, 0.25, 0.12), specular=(0.2, 0.2, 0.2), power=60.0))
    china = b.add_material(material(ma.BSDF_PHONG, diffuse=(0.6, 0.6, 0.65), specular=(0.3, 0.3, 0.3), power=200.0))
    mirror = b.add_material(material(ma.BSDF_REFLECTION))
    glass = b.add_material(material(ma.BSDF_TRANSMISSION, ior=1.45))
    S, H = 5.0, 3.2
    scale = (n_target / 114000.0) ** 0.5  # 114k triangles at scale 1
    g = max(4, int(48 * scale))
    b.add_tris(*heightfield((-S, -S, 0), (2 * S, 0, 0), (0, 2 * S, 0), (0, 0, 1), g, g, 0.01, rng), wood)
    b.add_tris(*heightfield((-S, -S, H), (2 * S, 0, 0), (0, 2 * S, 0), (0, 0, -1), g, g, 0.01, rng), wall)
    for (o, au, av, nn) in (((-S, -S, 0), (2 * S, 0, 0), (0, 0, H), (0, 1, 0)), ((-S, S, 0), (2 * S, 0, 0), (0, 0, H), (0, -1, 0)),
                            ((-S, -S, 0), (0, 2 * S, 0), (0, 0, H), (1, 0, 0)), ((S, -S, 0), (0, 2 * S, 0), (0, 0, H), (-1, 0, 0))):
        b.add_tris(*heightfield(o, au, av, nn, g, g // 2, 0.02, rng), wall)
    b.add_tris(*heightfield((-1.6, -0.9, 0.9), (3.2, 0, 0), (0, 1.8, 0), (0, 0, 1), g, g // 2, 0.002, rng), wood)  # table top
    for (x, y) in ((-1.5, -0.8), (1.5, -0.8), (-1.5, 0.8), (1.5, 0.8)):
        b.add_tris(*cylinder((x, y, 0.0), 0.05, 0.9, int(24 * scale), 4), wood)
    n_items = int(40 * scale)
    for k in range(n_items):
        x, y = rng.uniform(-1.4, 1.4), rng.uniform(-0.75, 0.75)
        r = rng.uniform(0.05, 0.14)
        m = [china, mirror, glass, china][k % 4]
        su, sv = int(48 * scale) + 8, int(24 * scale) + 4
        if k % 3 == 0:
            b.add_tris(*cylinder((x, y, 0.905), r, rng.uniform(0.08, 0.25), su, max(2, sv // 4), flute=0.02, flutes=12), m)
        else:
            b.add_tris(*sphere((x, y, 0.905 + r), r, su, sv), m)
    b.add_light((0.0, 0.0, H - 0.05), (0, 0, -1), (0, 1, 0), (1.2, 0.6), (55.0, 50.0, 42.0))
    b.add_light((-S + 0.05, 0.0, 1.8), (1, 0, 0), (0, 0, 1), (1.5, 1.0), (12.0, 14.0, 18.0))
    return b.build()


SCENES = {"atrium": atrium, "clutter": clutter}


def thin_slab(n_side=48, thickness=0.002, camera_height=3.0, seed=7):
    """A scene that is thin on one axis — a gently displaced n x n floor (2 n^2 triangles) with a low emitter lying just above it, everything within
    `thickness` of z = 0 — seen by a camera `camera_height` above it: the camera is camera_height / thickness scene extents (x 65 536 grid cells) outside
    the box on z.  The case ADVICE r03 names for the quantised node walks (16-bit grid over the scene box only)."""
    rng = np.random.default_rng(seed)
    b = FastBuilder()
    b.add_camera((0.3, -0.2, camera_height), (-0.05, 0.04, -1.0), up=(0, 1, 0), fovx=0.9)
    grey = b.add_material(material(ma.BSDF_DIFFUSE, diffuse=(0.7, 0.7, 0.65)))
    shiny = b.add_material(material(ma.BSDF_PHONG, diffuse=(0.3, 0.3, 0.3), specular=(0.4, 0.4, 0.4), power=30.0))
    half = n_side // 2
    b.add_tris(*heightfield((-2, -2, 0), (4, 0, 0), (0, 2, 0), (0, 0, 1), n_side, half, 0.4 * thickness, rng), grey)
    b.add_tris(*heightfield((-2, 0, 0), (4, 0, 0), (0, 2, 0), (0, 0, 1), n_side, half, 0.4 * thickness, rng), shiny)
    b.add_light((0.2, 0.1, 0.9 * thickness), (0, 0, -1), (0, 1, 0), (1.5, 1.0), (30, 28, 25))
    return b.build()


def load(spec):
    """'atrium', 'atrium:1000000', 'clutter:40000' -> Scene."""
    name, _, n = spec.partition(":")
    fn = SCENES[name]
    return fn(int(n)) if n else fn()


