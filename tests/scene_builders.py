"""Small procedural scenes for closed-form tests (built through the public Scene.from_arrays)."""
import ctypes as C

import numpy as np

import master_amd as ma


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
