"""ctypes binding of the CPU oracle (oracle/libpt_oracle.so).  TEST INFRASTRUCTURE: imported only
by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by master_amd/."""
import ctypes as C
import os
import subprocess

import numpy as np

import master_amd as ma

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "libpt_oracle.so")

_lib = None


def build():
    srcs = [os.path.join(ORACLE_DIR, "pt_oracle.c"), os.path.join(ORACLE_DIR, "bpt_oracle.inc"), os.path.join(ROOT, "include", "mi_pt.h")]
    if not os.path.exists(ORACLE_LIB) or os.path.getmtime(ORACLE_LIB) < max(os.path.getmtime(f) for f in srcs):
        subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)
    return ORACLE_LIB


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(ORACLE_LIB)
        vp, u32, u64, f32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float
        L.orc_create.argtypes = [C.POINTER(ma.SceneDesc), C.POINTER(ma.PtParams), C.c_int]
        L.orc_create.restype = vp
        L.orc_destroy.argtypes = [vp]
        L.orc_destroy.restype = None
        L.orc_set_use_bvh.argtypes = [vp, C.c_int]
        L.orc_bvh_info.argtypes = [vp, C.POINTER(ma.BvhInfo)]
        L.orc_bvh_download.argtypes = [vp, vp, vp, vp]
        L.orc_intersect.argtypes = [vp, u32, vp, vp, vp, vp, vp]
        L.orc_occluded.argtypes = [vp, u32, vp, vp, vp]
        L.orc_trace_paths.argtypes = [vp, u32, u32, u32, u32, vp, vp, u64, vp, vp]
        L.orc_render.argtypes = [vp, u32, u32, u32, ma.Window, u32, u64, u64, vp, C.POINTER(ma.PtStats), C.c_int]
        L.orc_bpt_set_sky.argtypes = [vp, C.POINTER(f32), C.POINTER(f32)]
        L.orc_bpt_set_sky.restype = None
        L.orc_bpt_trace_paths.argtypes = [vp, u32, u32, u32, u32, vp, vp, u64, vp, vp, vp]
        L.orc_bpt_render.argtypes = [vp, u32, u32, u32, ma.Window, u32, u64, u64, vp, C.POINTER(ma.PtStats), C.c_int]
        L.orc_camera_setup.argtypes = [C.POINTER(ma.Camera), f32, C.POINTER(ma.CameraFrame)]
        L.orc_ray_direction.argtypes = [f32, f32, f32, f32, f32, C.POINTER(f32)]
        L.orc_pixel_position.argtypes = [C.POINTER(f32), f32, f32, f32, C.POINTER(f32)]
        L.orc_rng_floats.argtypes = [u64, u32, u64, u32, vp]
        L.orc_bsdf_query.argtypes = [vp, vp, C.POINTER(f32), C.POINTER(f32), C.POINTER(f32), C.POINTER(f32), C.POINTER(f32), C.POINTER(C.c_int)]
        L.orc_bsdf_sample.argtypes = [vp, vp, C.POINTER(f32), u64, u32, u64, C.POINTER(f32), C.POINTER(f32), C.POINTER(f32), C.POINTER(f32), C.POINTER(C.c_int)]
        L.orc_light_sample.argtypes = [vp, u64, u32, u64, vp, C.POINTER(f32), C.POINTER(f32), C.POINTER(f32)]
        L.orc_light_table.argtypes = [vp, vp, vp]
        L.orc_rms_abs_errors.argtypes = [vp, vp, u32, u32, C.POINTER(f32), C.POINTER(f32)]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


class Oracle:
    """CPU restatement of PathTracing over a master_amd.Scene (same constructor arguments)."""

    def __init__(self, scene, lights=1.0, roulette=0.9, beta=1.0, max_path=ma.PTRDIFF_MAX, use_bvh=True, min_subpath=3):
        self.scene = scene
        self.params = ma.PtParams(int(max_path), float(beta), float(roulette), float(lights), int(min_subpath))
        self._h = C.c_void_p(lib().orc_create(C.byref(scene.desc), C.byref(self.params), 1 if use_bvh else 0))
        self.last_stats = None

    def __del__(self):
        try:
            if self._h:
                lib().orc_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def set_use_bvh(self, flag):
        lib().orc_set_use_bvh(self._h, 1 if flag else 0)

    def render_rgbn(self, width, height, spp=1, seed=0, sample_offset=0, camera_id=0, window=None, threads=None):
        out = np.zeros((height, width, 4), np.float32)
        st = ma.PtStats()
        win = ma.Window(*window) if window else ma.Window(0, 0, 0, 0)
        threads = threads or (os.cpu_count() or 1)
        rc = lib().orc_render(self._h, camera_id, width, height, win, spp, seed, sample_offset, _ptr(out), C.byref(st), threads)
        if rc != 0:
            raise ma.MiError(rc, "orc_render failed")
        self.last_stats = st
        return out

    def intersect(self, origins, directions):
        origins = np.ascontiguousarray(origins, ma.SURFACE_DTYPE)
        directions = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        n = len(origins)
        hits = np.zeros(n, ma.SURFACE_DTYPE)
        t = np.zeros(n, np.float32)
        prim = np.zeros(n, np.uint32)
        lib().orc_intersect(self._h, n, _ptr(origins), _ptr(directions), _ptr(hits), _ptr(t), _ptr(prim))
        return hits, t, prim

    def occluded(self, origins, targets):
        origins = np.ascontiguousarray(origins, ma.SURFACE_DTYPE)
        targets = np.ascontiguousarray(targets, ma.SURFACE_DTYPE)
        out = np.zeros(len(origins), np.float32)
        lib().orc_occluded(self._h, len(origins), _ptr(origins), _ptr(targets), _ptr(out))
        return out

    def trace_paths(self, width, height, pixel_xy, sample_index, seed=0, camera_id=0):
        pixel_xy = np.ascontiguousarray(pixel_xy, np.uint32).reshape(-1, 2)
        sample_index = np.ascontiguousarray(sample_index, np.uint64)
        n = len(pixel_xy)
        rad = np.zeros((n, 3), np.float32)
        cnt = np.zeros((n, 2), np.uint32)
        lib().orc_trace_paths(self._h, camera_id, width, height, n, _ptr(pixel_xy), _ptr(sample_index), seed, _ptr(rad), _ptr(cnt))
        return rad, cnt

    def bpt_set_sky(self, horizon, zenith):
        lib().orc_bpt_set_sky(self._h, _f3(horizon), _f3(zenith))

    def bpt_trace_paths(self, width, height, pixel_xy, sample_index, seed=0, camera_id=0):
        """BPT (BPT.cpp): per path the eye-image radiance, the float sum of its light-image splats, and (closest, shadow, splats) counts."""
        pixel_xy = np.ascontiguousarray(pixel_xy, np.uint32).reshape(-1, 2)
        sample_index = np.ascontiguousarray(sample_index, np.uint64)
        n = len(pixel_xy)
        rad = np.zeros((n, 3), np.float32); spl = np.zeros((n, 3), np.float32); cnt = np.zeros((n, 3), np.uint32)
        lib().orc_bpt_trace_paths(self._h, camera_id, width, height, n, _ptr(pixel_xy), _ptr(sample_index), C.c_uint64(seed), _ptr(rad), _ptr(spl), _ptr(cnt))
        return rad, spl, cnt

    def bpt_render_rgbn(self, width, height, spp=1, seed=0, sample_offset=0, camera_id=0, threads=None, window=None):
        out = np.zeros((height, width, 4), np.float32)
        st = ma.PtStats()
        win = ma.Window(*window) if window else ma.Window(0, 0, 0, 0)
        rc = lib().orc_bpt_render(self._h, camera_id, width, height, win, spp, C.c_uint64(seed), C.c_uint64(sample_offset), _ptr(out), C.byref(st),
                                  threads or (os.cpu_count() or 1))
        assert rc == 0, rc
        self.last_stats = st
        return out

    def bvh_info(self):
        info = ma.BvhInfo()
        lib().orc_bvh_info(self._h, C.byref(info))
        return info

    def bvh(self):
        info = self.bvh_info()
        nodes = np.zeros(info.n_nodes, ma.NODE_DTYPE)
        sorted_tri = np.zeros(info.n_triangles, np.uint32)
        morton = np.zeros(info.n_triangles, np.uint64)
        lib().orc_bvh_download(self._h, _ptr(nodes), _ptr(sorted_tri), _ptr(morton))
        return nodes, sorted_tri, morton

    def light_table(self):
        n = len(self.scene.lights)
        w, cdf = np.zeros(n, np.float32), np.zeros(n + 1, np.float32)
        lib().orc_light_table(self._h, _ptr(w), _ptr(cdf))
        return w, cdf

    def bsdf_query(self, surface, incident, outgoing):
        sp = np.ascontiguousarray(surface, ma.SURFACE_DTYPE).reshape(1)
        tp = (C.c_float * 3)()
        d, dr, fin = C.c_float(), C.c_float(), C.c_int()
        lib().orc_bsdf_query(self._h, _ptr(sp), _f3(incident), _f3(outgoing), tp, C.byref(d), C.byref(dr), C.byref(fin))
        return np.array(list(tp), np.float32), d.value, dr.value, fin.value

    def bsdf_sample(self, surface, omega, seed=0, pixel=0, sample=0):
        sp = np.ascontiguousarray(surface, ma.SURFACE_DTYPE).reshape(1)
        om, tp = (C.c_float * 3)(), (C.c_float * 3)()
        d, dr, fin = C.c_float(), C.c_float(), C.c_int()
        lib().orc_bsdf_sample(self._h, _ptr(sp), _f3(omega), seed, pixel, sample, om, tp, C.byref(d), C.byref(dr), C.byref(fin))
        return np.array(list(om), np.float32), np.array(list(tp), np.float32), d.value, dr.value, fin.value

    def light_sample(self, seed=0, pixel=0, sample=0):
        sp = np.zeros(1, ma.SURFACE_DTYPE)
        rad = (C.c_float * 3)()
        ad, ld = C.c_float(), C.c_float()
        lib().orc_light_sample(self._h, seed, pixel, sample, _ptr(sp), rad, C.byref(ad), C.byref(ld))
        return sp[0], np.array(list(rad), np.float32), ad.value, ld.value


def camera_setup(camera, aspect):
    out = ma.CameraFrame()
    lib().orc_camera_setup(C.byref(camera), aspect, C.byref(out))
    return out


def ray_direction(px, py, rx, ry, fl):
    out = (C.c_float * 3)()
    lib().orc_ray_direction(px, py, rx, ry, fl, out)
    return np.array(list(out), np.float32)


def pixel_position(direction, rx, ry, fl):
    out = (C.c_float * 2)()
    lib().orc_pixel_position(_f3(direction), rx, ry, fl, out)
    return np.array(list(out), np.float32)


def rng_floats(seed, pixel, sample, n):
    out = np.zeros(n, np.float32)
    lib().orc_rng_floats(seed, pixel, sample, n, _ptr(out))
    return out


def rms_abs_errors(rgbn, ref):
    rgbn = np.ascontiguousarray(rgbn, np.float32)
    ref = np.ascontiguousarray(ref, np.float32)
    h, w = rgbn.shape[:2]
    r, a = C.c_float(), C.c_float()
    lib().orc_rms_abs_errors(_ptr(rgbn), _ptr(ref), w, h, C.byref(r), C.byref(a))
    return r.value, a.value


def powf(x, y):
    """The build's own pow(x, y) as the oracle defines it (mi_powf in oracle/pt_oracle.c)."""
    import ctypes as C
    x = np.ascontiguousarray(x, np.float32); y = np.ascontiguousarray(y, np.float32); out = np.empty_like(x)
    L = lib()
    L.orc_powf.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_powf(x.size, x.ctypes.data, y.ctypes.data, out.ctypes.data)
    return out


def sincos_2pi(u):
    """(sin, cos)(2 pi u) as the build defines them (sincos_2pi in oracle/pt_oracle.c, the same statement as device/vecmath.h)."""
    import ctypes as C
    L = lib()
    L.orc_sincos_2pi.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    s, c = C.c_float(), C.c_float()
    L.orc_sincos_2pi(float(u), C.byref(s), C.byref(c))
    return s.value, c.value


def asinf(x):
    import ctypes as C
    L = lib()
    L.orc_asinf.argtypes = [C.c_float]; L.orc_asinf.restype = C.c_float
    return L.orc_asinf(float(x))
