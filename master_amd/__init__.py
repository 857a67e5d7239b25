"""master_amd — MI355X-native unidirectional path tracer behind the `Technique` interface of
ciechowoj/master (PT path only).

The product is `libmi_pt.so` (HIP kernels for gfx950 + a C ABI, see include/mi_pt.h).  This
module is only the thin ctypes binding used by tests and bench.py, shaped after the reference's
host classes so parity tests read like the reference's own code:

    scene = Scene.load("scenes/CornellBoxDiffuse.miscene")        # loadScene (loader.cpp:458)
    pt = PathTracing(scene, lights=1.0, roulette=0.9, beta=1.0, max_path=8)   # PT.cpp:5-13
    pt.render(view, seed, camera_id)                              # Technique::render (Technique.cpp:15)

There is no CPU fallback: every compute call fails loudly (MiError) when libmi_pt.so or a HIP
device is missing.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI_PT_LIB") or os.path.join(_HERE, "libmi_pt.so")  # MI_PT_LIB: A/B builds of the same library

MI_OK = 0
ERR_NAMES = {-1: "INVALID_ARGUMENT", -2: "NO_DEVICE", -3: "OUT_OF_MEMORY", -4: "IO", -5: "UNSUPPORTED", -6: "INTERNAL"}

BSDF_CAMERA, BSDF_DIFFUSE, BSDF_PHONG, BSDF_REFLECTION, BSDF_TRANSMISSION, BSDF_LIGHT, BSDF_SUN = range(7)
ENTITY_CAMERA, ENTITY_MESH, ENTITY_LIGHT, ENTITY_EMPTY = range(4)
KERNEL_AUTO, KERNEL_MEGA_LDS, KERNEL_MEGA_GLOBAL, KERNEL_WAVEFRONT = range(4)
UINT32_MAX = 0xFFFFFFFF
PTRDIFF_MAX = (1 << 63) - 1


class MiError(RuntimeError):
    """The C ABI's error return, raised the way the reference throws std::runtime_error."""

    def __init__(self, code, message):
        super().__init__("[%s] %s" % (ERR_NAMES.get(code, code), message))
        self.code = code


# ---- C structs (include/mi_pt.h) ----
class Material(C.Structure):
    _fields_ = [("type", C.c_uint32), ("diffuse", C.c_float * 3), ("specular", C.c_float * 3), ("power", C.c_float),
                ("ior_internal", C.c_float), ("ior_external", C.c_float), ("light_id", C.c_uint32), ("reserved", C.c_uint32)]


class Light(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("tangent", C.c_float * 9), ("size", C.c_float * 2), ("exitance", C.c_float * 3),
                ("diffuse", C.c_uint32), ("material_id", C.c_uint32), ("reserved", C.c_uint32)]


class Camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("direction", C.c_float * 3), ("up", C.c_float * 3), ("fovx", C.c_float)]


class SceneDesc(C.Structure):
    _fields_ = [("n_vertices", C.c_uint32), ("n_triangles", C.c_uint32), ("n_meshes", C.c_uint32), ("n_materials", C.c_uint32),
                ("n_lights", C.c_uint32), ("n_cameras", C.c_uint32),
                ("positions", C.POINTER(C.c_float)), ("tangents", C.POINTER(C.c_float)), ("indices", C.POINTER(C.c_uint32)),
                ("mesh_tri_offset", C.POINTER(C.c_uint32)), ("mesh_material_id", C.POINTER(C.c_uint32)),
                ("materials", C.POINTER(Material)), ("lights", C.POINTER(Light)), ("cameras", C.POINTER(Camera)),
                ("bounding_sphere", C.c_float * 4)]


class PtParams(C.Structure):
    _fields_ = [("max_path", C.c_uint64), ("beta", C.c_float), ("roulette", C.c_float), ("lights", C.c_float), ("min_subpath", C.c_uint32)]


class Window(C.Structure):
    _fields_ = [("x0", C.c_uint32), ("y0", C.c_uint32), ("w", C.c_uint32), ("h", C.c_uint32)]


class PtStats(C.Structure):
    _fields_ = [("num_paths", C.c_uint64), ("num_basic_rays", C.c_uint64), ("num_shadow_rays", C.c_uint64), ("numeric_errors", C.c_uint64),
                ("gpu_ms", C.c_double), ("trace_ms", C.c_double),
                ("nodes_closest", C.c_uint64), ("tris_closest", C.c_uint64), ("nodes_shadow", C.c_uint64), ("tris_shadow", C.c_uint64),
                ("num_hits", C.c_uint64), ("wave_steps_closest", C.c_uint64), ("wave_steps_shadow", C.c_uint64),
                ("phase_cycles", C.c_uint64 * 8), ("wave_loop_bodies", C.c_uint64 * 4)]


class LaunchInfo(C.Structure):
    _fields_ = [("kernel", C.c_uint32), ("n_blocks", C.c_uint32), ("n_chunks", C.c_uint32), ("chunk_spp", C.c_uint32),
                ("lds_bytes", C.c_uint32), ("wide_nodes", C.c_uint32), ("features", C.c_uint32), ("lds_tables", C.c_uint32),
                ("frame_tiles_per_wave", C.c_uint32), ("frames", C.c_uint32), ("dynamic_fetch", C.c_uint32), ("flat_leaves", C.c_uint32), ("partial_bytes", C.c_uint64), ("scene_bytes", C.c_uint64)]


MAX_FRAMES_PER_BATCH, BATCHES_IN_FLIGHT = 16, 3  # MI_PT_MAX_FRAMES_PER_BATCH, MI_PT_BATCHES_IN_FLIGHT


class SurfacePoint(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("gnormal", C.c_float * 3), ("tangent", C.c_float * 9), ("material_id", C.c_uint32)]


class BvhNode(C.Structure):
    _fields_ = [("lo0", C.c_float * 3), ("link0", C.c_int32), ("hi0", C.c_float * 3), ("link1", C.c_int32),
                ("lo1", C.c_float * 3), ("parent", C.c_uint32), ("hi1", C.c_float * 3), ("reserved", C.c_uint32)]


class BvhInfo(C.Structure):
    _fields_ = [("n_triangles", C.c_uint32), ("n_nodes", C.c_uint32), ("max_depth", C.c_uint32), ("stack_entries", C.c_uint32),
                ("scene_lo", C.c_float * 3), ("scene_hi", C.c_float * 3), ("build_ms", C.c_double),
                ("builder", C.c_uint32), ("build_rounds", C.c_uint32)]


class CameraFrame(C.Structure):
    _fields_ = [("view_to_world", C.c_float * 9), ("world_to_view", C.c_float * 9), ("position", C.c_float * 3),
                ("focal_length_y", C.c_float), ("fovy", C.c_float)]


class BlendOptions(C.Structure):
    _fields_ = [("diffuse_scale_by_ref", C.c_float), ("specular_scale_by_spec", C.c_float), ("lamp_energy_scale", C.c_float), ("reserved", C.c_uint32)]


SURFACE_DTYPE = np.dtype([("position", "<f4", 3), ("gnormal", "<f4", 3), ("tangent", "<f4", 9), ("material_id", "<u4")])
NODE_DTYPE = np.dtype([("lo0", "<f4", 3), ("link0", "<i4"), ("hi0", "<f4", 3), ("link1", "<i4"), ("lo1", "<f4", 3), ("parent", "<u4"),
                       ("hi1", "<f4", 3), ("reserved", "<u4")])
assert SURFACE_DTYPE.itemsize == C.sizeof(SurfacePoint) == 64 and NODE_DTYPE.itemsize == C.sizeof(BvhNode) == 64

# every symbol include/mi_pt.h declares; tests/test_abi.py checks the library exports them all
ABI_SYMBOLS = [
    "mi_pt_create", "mi_pt_destroy", "mi_pt_render", "mi_pt_render_device", "mi_pt_last_error", "mi_pt_abi_version", "mi_pt_build_id",
    "mi_pt_render_frames_async", "mi_pt_render_async", "mi_pt_wait", "mi_pt_wait_add", "mi_view_add_frame", "mi_pt_last_launch",
    "mi_pt_set_kernel", "mi_pt_get_kernel", "mi_pt_set_tile_shard", "mi_pt_render_multi", "mi_pt_last_multi_merge", "mi_pt_reduce_available", "mi_pt_reduce_unique_id", "mi_pt_reduce_init", "mi_pt_reduce_rgbn", "mi_pt_reduce_finalize", "mi_pt_device_count", "mi_pt_set_instrumented", "mi_pt_intersect", "mi_pt_occluded", "mi_pt_trace_paths", "mi_pt_bvh_info",
    "mi_bpt_render", "mi_bpt_trace_paths", "mi_bpt_set_sky",
    "mi_pt_bvh_download", "mi_pt_blob_download", "mi_camera_setup", "mi_camera_ray_direction", "mi_camera_pixel_position", "mi_scene_load_blend",
    "mi_scene_load", "mi_scene_save", "mi_scene_from_desc", "mi_scene_get_desc", "mi_scene_material_name", "mi_scene_mesh_name",
    "mi_scene_free", "mi_exr_save_rgbn", "mi_exr_load_rgbn", "mi_free", "mi_rms_abs_errors", "mi_rms_abs_errors_view",
]

_lib = None


def lib():
    """Loads libmi_pt.so (built in-tree by master_amd/build.py).  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MiError(-2, "libmi_pt.so is not built (run `python -m master_amd.build`); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, f32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float
    L.mi_pt_last_error.restype = C.c_char_p
    L.mi_pt_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(PtParams), C.c_int, C.POINTER(vp)]
    L.mi_pt_destroy.argtypes = [vp]
    L.mi_pt_destroy.restype = None
    L.mi_pt_render.argtypes = [vp, u32, u32, u32, Window, u32, u64, u64, vp, C.POINTER(PtStats)]
    L.mi_pt_render_device.argtypes = [vp, u32, u32, u32, Window, u32, u64, u64, vp, vp, C.POINTER(PtStats)]
    L.mi_pt_render_async.argtypes = [vp, u32, u32, u32, Window, u32, u64, u64, C.POINTER(u64)]
    L.mi_pt_render_frames_async.argtypes = [vp, u32, u32, u32, Window, u32, u64, u64, C.POINTER(u64)]
    L.mi_pt_wait.argtypes = [vp, u64, C.POINTER(C.POINTER(f32)), C.POINTER(PtStats)]
    L.mi_pt_wait_add.argtypes = [vp, u64, vp, C.POINTER(PtStats)]
    L.mi_view_add_frame.argtypes = [vp, vp, u32, u32, Window]
    L.mi_pt_last_launch.argtypes = [vp, C.POINTER(LaunchInfo)]
    L.mi_pt_set_kernel.argtypes = [vp, C.c_int]
    L.mi_pt_set_tile_shard.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.mi_pt_render_multi.argtypes = [C.POINTER(vp), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, Window, C.c_uint32, C.c_uint64, C.c_uint64, vp, C.POINTER(PtStats)]
    L.mi_pt_device_count.restype = C.c_int
    L.mi_pt_reduce_unique_id.argtypes = [vp]
    L.mi_pt_reduce_init.argtypes = [vp, vp, u32, u32]
    L.mi_pt_reduce_rgbn.argtypes = [vp, vp, u32, u32, C.c_int, vp]
    L.mi_pt_reduce_finalize.argtypes = [vp]
    L.mi_pt_get_kernel.argtypes = [vp]
    L.mi_pt_set_instrumented.argtypes = [vp, C.c_int]
    L.mi_pt_intersect.argtypes = [vp, u32, vp, vp, vp, vp, vp]
    L.mi_pt_occluded.argtypes = [vp, u32, vp, vp, vp]
    L.mi_pt_trace_paths.argtypes = [vp, u32, u32, u32, u32, vp, vp, u64, vp, vp]
    L.mi_bpt_set_sky.argtypes = [vp, C.POINTER(f32), C.POINTER(f32)]
    L.mi_bpt_render.argtypes = [vp, u32, u32, u32, Window, u32, u64, u64, vp, C.POINTER(PtStats)]
    L.mi_bpt_trace_paths.argtypes = [vp, u32, u32, u32, u32, vp, vp, u64, vp, vp, vp]
    L.mi_pt_bvh_info.argtypes = [vp, C.POINTER(BvhInfo)]
    L.mi_pt_bvh_download.argtypes = [vp, vp, vp, vp]
    L.mi_pt_blob_download.argtypes = [vp, C.POINTER(u32), vp, C.c_size_t]
    L.mi_camera_setup.argtypes = [C.POINTER(Camera), f32, C.POINTER(CameraFrame)]
    L.mi_camera_ray_direction.argtypes = [f32, f32, f32, f32, f32, C.POINTER(f32)]
    L.mi_camera_ray_direction.restype = None
    L.mi_camera_pixel_position.argtypes = [C.POINTER(f32), f32, f32, f32, C.POINTER(f32)]
    L.mi_camera_pixel_position.restype = None
    L.mi_scene_load_blend.argtypes = [C.c_char_p, C.POINTER(BlendOptions), C.POINTER(vp)]
    L.mi_scene_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.mi_scene_save.argtypes = [vp, C.c_char_p]
    L.mi_scene_from_desc.argtypes = [C.POINTER(SceneDesc), C.POINTER(vp)]
    L.mi_scene_get_desc.argtypes = [vp]
    L.mi_scene_get_desc.restype = C.POINTER(SceneDesc)
    L.mi_scene_material_name.argtypes = [vp, u32]
    L.mi_scene_material_name.restype = C.c_char_p
    L.mi_scene_mesh_name.argtypes = [vp, u32]
    L.mi_scene_mesh_name.restype = C.c_char_p
    L.mi_scene_free.argtypes = [vp]
    L.mi_scene_free.restype = None
    L.mi_exr_save_rgbn.argtypes = [C.c_char_p, u32, u32, vp, u32, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]
    L.mi_exr_load_rgbn.argtypes = [C.c_char_p, C.POINTER(u32), C.POINTER(u32), C.POINTER(C.POINTER(f32))]
    L.mi_free.argtypes = [vp]
    L.mi_free.restype = None
    L.mi_rms_abs_errors.argtypes = [vp, vp, u32, u32, C.POINTER(f32), C.POINTER(f32)]
    L.mi_rms_abs_errors_view.argtypes = [vp, vp, u32, u32, C.POINTER(f32), C.POINTER(f32)]
    _lib = L
    return L


def _check(rc):
    if rc != MI_OK:
        raise MiError(rc, lib().mi_pt_last_error().decode(errors="replace"))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Scene:
    """haste::Scene flattened (Scene.hpp:27-43).  Owns a `mi_scene*`; arrays are numpy views/copies."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)
        d = lib().mi_scene_get_desc(self._h).contents
        self.desc = d
        nv, nt, nm = d.n_vertices, d.n_triangles, d.n_meshes
        self.positions = np.ctypeslib.as_array(d.positions, (nv, 3)).copy()
        self.tangents = np.ctypeslib.as_array(d.tangents, (nv, 9)).copy()
        self.indices = np.ctypeslib.as_array(d.indices, (nt, 3)).copy()
        self.mesh_tri_offset = np.ctypeslib.as_array(d.mesh_tri_offset, (nm + 1,)).copy()
        self.mesh_material_id = np.ctypeslib.as_array(d.mesh_material_id, (nm,)).copy() if nm else np.zeros(0, np.uint32)
        # copies: d.materials[i] would alias memory owned by the C scene
        self.materials = [Material.from_buffer_copy(d.materials[i]) for i in range(d.n_materials)]
        self.lights = [Light.from_buffer_copy(d.lights[i]) for i in range(d.n_lights)]
        self.cameras = [Camera.from_buffer_copy(d.cameras[i]) for i in range(d.n_cameras)]
        self.material_names = [lib().mi_scene_material_name(self._h, i).decode() for i in range(d.n_materials)]
        self.mesh_names = [lib().mi_scene_mesh_name(self._h, i).decode() for i in range(nm)]
        self.bounding_sphere = np.array(list(d.bounding_sphere), np.float32)

    n_triangles = property(lambda s: s.desc.n_triangles)

    @property
    def tri_material(self):
        out = np.zeros(self.desc.n_triangles, np.uint32)
        for m in range(self.desc.n_meshes):
            out[self.mesh_tri_offset[m]:self.mesh_tri_offset[m + 1]] = self.mesh_material_id[m]
        return out

    @classmethod
    def load_blend(cls, path, diffuse_scale_by_ref=0.0, specular_scale_by_spec=0.0, lamp_energy_scale=1.0):
        """loadScene (loader.cpp:458-487) through the build's own .blend reader."""
        o = BlendOptions(diffuse_scale_by_ref, specular_scale_by_spec, lamp_energy_scale, 0)
        h = C.c_void_p()
        _check(lib().mi_scene_load_blend(os.fsencode(path), C.byref(o), C.byref(h)))
        return cls(h.value)

    @classmethod
    def load(cls, path):
        h = C.c_void_p()
        _check(lib().mi_scene_load(os.fsencode(path), C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_arrays(cls, positions, tangents, indices, mesh_tri_offset, mesh_material_id, materials, lights, cameras,
                    bounding_sphere=(0, 0, 0, 0)):
        positions = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        tangents = np.ascontiguousarray(tangents, np.float32).reshape(-1, 9)
        indices = np.ascontiguousarray(indices, np.uint32).reshape(-1, 3)
        mto = np.ascontiguousarray(mesh_tri_offset, np.uint32)
        mmi = np.ascontiguousarray(mesh_material_id, np.uint32)
        d = SceneDesc()
        d.n_vertices, d.n_triangles, d.n_meshes = len(positions), len(indices), len(mmi)
        d.n_materials, d.n_lights, d.n_cameras = len(materials), len(lights), len(cameras)
        d.positions = positions.ctypes.data_as(C.POINTER(C.c_float))
        d.tangents = tangents.ctypes.data_as(C.POINTER(C.c_float))
        d.indices = indices.ctypes.data_as(C.POINTER(C.c_uint32))
        d.mesh_tri_offset = mto.ctypes.data_as(C.POINTER(C.c_uint32))
        d.mesh_material_id = mmi.ctypes.data_as(C.POINTER(C.c_uint32))
        ma = (Material * max(1, len(materials)))(*materials)
        la = (Light * max(1, len(lights)))(*lights)
        ca = (Camera * max(1, len(cameras)))(*cameras)
        d.materials, d.lights, d.cameras = ma, la, ca
        d.bounding_sphere = (C.c_float * 4)(*bounding_sphere)
        h = C.c_void_p()
        _check(lib().mi_scene_from_desc(C.byref(d), C.byref(h)))
        return cls(h.value)

    def save(self, path):
        _check(lib().mi_scene_save(self._h, os.fsencode(path)))

    def __del__(self):
        try:
            if self._h:
                lib().mi_scene_free(self._h)
                self._h = None
        except Exception:
            pass


class Statistics:
    """The fields of statistics_t (statistics.hpp:15-57) that Technique::render fills (Technique.cpp:55-76)."""

    def __init__(self):
        self.num_samples = 0
        self.num_basic_rays = 0
        self.num_shadow_rays = 0
        self.total_time = 0.0
        self.records = []  # dicts: sample_index, rms_error, abs_error, clock_time, frame_duration, numeric_errors

    def to_dict(self):
        """statistics_t::to_dict (statistics.cpp:118-175): the string map the reference stores in the EXR header
        (same keys, std::to_string formatting), so `master continue / merge / statistics` can read our files."""
        f = lambda v: "%f" % float(v)  # std::to_string(double)
        d = {"statistics.num_samples": str(int(self.num_samples)), "statistics.num_basic_rays": str(int(self.num_basic_rays)),
             "statistics.num_shadow_rays": str(int(self.num_shadow_rays)), "statistics.num_tentative_rays": "0",
             "statistics.num_photons": "0", "statistics.num_scattered": "0", "statistics.total_time": f(self.total_time)}
        for k in ("scatter_time", "build_time", "gather_time", "merge_time", "density_time", "intersect_time", "trace_eye_time", "trace_light_time"):
            d["statistics." + k] = f(0.0)
        for r in self.records:
            i = int(r["sample_index"])
            d["records[%d].rms_error" % i] = f(r["rms_error"]); d["records[%d].abs_error" % i] = f(r["abs_error"])
            d["records[%d].clock_time" % i] = f(r["clock_time"]); d["records[%d].frame_duration" % i] = f(r["frame_duration"])
            d["records[%d].numeric_errors" % i] = str(int(r["numeric_errors"]))
        return d


def camera_setup(camera, aspect):
    out = CameraFrame()
    _check(lib().mi_camera_setup(C.byref(camera), aspect, C.byref(out)))
    return out


def ray_direction(px, py, res_x, res_y, focal_length_y):
    out = (C.c_float * 3)()
    lib().mi_camera_ray_direction(px, py, res_x, res_y, focal_length_y, out)
    return np.array(list(out), np.float32)


def pixel_position(direction, res_x, res_y, focal_length_y):
    d = (C.c_float * 3)(*[float(x) for x in direction])
    out = (C.c_float * 2)()
    lib().mi_camera_pixel_position(d, res_x, res_y, focal_length_y, out)
    return np.array(list(out), np.float32)


def rms_abs_errors(rgbn, reference_rgb):
    """rms_abs_errors (ImageView.cpp:60-85)."""
    rgbn = np.ascontiguousarray(rgbn, np.float32)
    ref = np.ascontiguousarray(reference_rgb, np.float32)
    h, w = rgbn.shape[:2]
    r, a = C.c_float(), C.c_float()
    _check(lib().mi_rms_abs_errors(_ptr(rgbn), _ptr(ref), w, h, C.byref(r), C.byref(a)))
    return r.value, a.value


def rms_abs_errors_view(view, reference_rgb):
    """rms_abs_errors (ImageView.cpp:60-85) over the float64 [H][W][4] view itself (the reference's image_view_t<dvec4>)."""
    assert view.dtype == np.float64 and view.flags["C_CONTIGUOUS"]
    ref = np.ascontiguousarray(reference_rgb, np.float32)
    h, w = view.shape[:2]
    r, a = C.c_float(), C.c_float()
    _check(lib().mi_rms_abs_errors_view(_ptr(view), _ptr(ref), w, h, C.byref(r), C.byref(a)))
    return r.value, a.value


def view_add_frame(view, rgbn, window=None):
    """Technique::_commit_images for the PT path (Technique.cpp:215-236): view += dvec4(rgbn) over the window, on the library's host threads."""
    assert view.dtype == np.float64 and rgbn.dtype == np.float32 and view.shape == rgbn.shape and view.flags["C_CONTIGUOUS"] and rgbn.flags["C_CONTIGUOUS"]
    h, w = view.shape[:2]
    _check(lib().mi_view_add_frame(_ptr(rgbn), _ptr(view), w, h, Window(*window) if window else Window(0, 0, 0, 0)))


def save_exr(path, rgbn, metadata=None):
    """save_exr (exr.cpp:177-232): channels R,G,B,denom + string attributes."""
    rgbn = np.ascontiguousarray(rgbn, np.float32)
    h, w = rgbn.shape[:2]
    md = metadata or {}
    keys = (C.c_char_p * max(1, len(md)))(*[k.encode() for k in md])
    vals = (C.c_char_p * max(1, len(md)))(*[str(v).encode() for v in md.values()])
    _check(lib().mi_exr_save_rgbn(os.fsencode(path), w, h, _ptr(rgbn), len(md), keys, vals))


def load_exr(path):
    w, h, p = C.c_uint32(), C.c_uint32(), C.POINTER(C.c_float)()
    _check(lib().mi_exr_load_rgbn(os.fsencode(path), C.byref(w), C.byref(h), C.byref(p)))
    try:
        return np.ctypeslib.as_array(p, (h.value, w.value, 4)).copy()
    finally:
        lib().mi_free(p)


REDUCE_ID_BYTES = 128


def reduce_available():
    return bool(lib().mi_pt_reduce_available())


def reduce_unique_id():
    """mi_pt_reduce_unique_id: the 128 bytes rank 0 hands to the other ranks."""
    buf = (C.c_ubyte * REDUCE_ID_BYTES)()
    _check(lib().mi_pt_reduce_unique_id(buf))
    return bytes(buf)


def device_count():
    return lib().mi_pt_device_count()


def render_multi(techniques, width, height, spp=1, seed=0, sample_offset=0, camera_id=0, window=None):
    """mi_pt_render_multi: one frame set rendered by several PathTracing handles (one per GPU) of ONE process; the 32x32
    tiles of Technique::_trace_paths (Technique.cpp:167) are dealt round-robin to the handles.  Returns (rgbn, stats)."""
    out = np.zeros((height, width, 4), dtype=np.float32)
    st = PtStats()
    win = Window(*window) if window else Window(0, 0, 0, 0)
    hs = (C.c_void_p * len(techniques))(*[t._h.value for t in techniques])
    _check(lib().mi_pt_render_multi(hs, len(techniques), camera_id, width, height, win, spp, seed, sample_offset, _ptr(out), C.byref(st)))
    return out, st


class PathTracing:
    """haste::PathTracing (PT.hpp:6-29) on the GPU.  Constructor arguments follow PT.cpp:5-13;
    `num_threads` has no meaning here and is replaced by `device`."""

    def __init__(self, scene, lights=1.0, roulette=0.9, beta=1.0, max_path=PTRDIFF_MAX, device=0, min_subpath=3):
        self.scene = scene
        self.params = PtParams(int(max_path), float(beta), float(roulette), float(lights), int(min_subpath))
        self._h = C.c_void_p()
        _check(lib().mi_pt_create(C.byref(scene.desc), C.byref(self.params), int(device), C.byref(self._h)))
        self._statistics = Statistics()
        self.last_stats = None

    def close(self):
        if self._h:
            lib().mi_pt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def options_dict(self, width, height, camera_id=0, input0="", output=""):
        """The PT-relevant part of Options::to_dict (Options.cpp:1186-1232) for the EXR header."""
        f = lambda v: "%f" % float(v)
        mp = self.params.max_path
        return {"options.input0": input0, "options.output": output, "options.technique": "PT",
                "options.max_path": str(PTRDIFF_MAX if mp >= PTRDIFF_MAX else int(mp)), "options.beta": f(self.params.beta),
                "options.roulette": f(self.params.roulette), "options.lights": f(self.params.lights),
                "options.num_samples": str(int(self._statistics.num_samples)), "options.camera_id": str(int(camera_id)),
                "options.width": str(int(width)), "options.height": str(int(height))}

    # -- Technique API ----------------------------------------------------------------------
    def statistics(self):
        return self._statistics

    def set_statistics(self, s):
        self._statistics = s

    def render_rgbn(self, width, height, spp=1, seed=0, sample_offset=0, camera_id=0, window=None):
        """mi_pt_render: float32 [H][W][4] sums of `spp` samples per pixel, row 0 = bottom."""
        out = np.empty((height, width, 4), np.float32)
        st = PtStats()
        win = Window(*window) if window else Window(0, 0, 0, 0)
        _check(lib().mi_pt_render(self._h, camera_id, width, height, win, spp, seed, sample_offset, _ptr(out), C.byref(st)))
        self.last_stats = st
        return out

    def render_device(self, device_ptr, width, height, spp=1, seed=0, sample_offset=0, camera_id=0, window=None, stream=None, want_stats=True):
        """mi_pt_render_device: result stays in device memory (e.g. a torch CUDA tensor's data_ptr())."""
        st = PtStats()
        win = Window(*window) if window else Window(0, 0, 0, 0)
        _check(lib().mi_pt_render_device(self._h, camera_id, width, height, win, spp, seed, sample_offset, C.c_void_p(device_ptr),
                                         C.c_void_p(stream) if stream else None, C.byref(st) if want_stats else None))
        self.last_stats = st if want_stats else None
        return self.last_stats

    def render_async(self, width, height, spp=1, seed=0, sample_offset=0, camera_id=0, window=None):
        """mi_pt_render_async: enqueue ONE frame of `spp` samples, returns its ticket (up to BATCHES_IN_FLIGHT calls may be pending)."""
        t = C.c_uint64()
        win = Window(*window) if window else Window(0, 0, 0, 0)
        _check(lib().mi_pt_render_async(self._h, camera_id, width, height, win, spp, seed, sample_offset, C.byref(t)))
        self._frame_shape = (height, width, 4)
        return t.value

    def render_frames_async(self, width, height, n_frames, seed=0, first_sample=0, camera_id=0, window=None):
        """mi_pt_render_frames_async: enqueue `n_frames` consecutive one-sample frames as ONE launch, returns their tickets."""
        t = (C.c_uint64 * n_frames)()
        win = Window(*window) if window else Window(0, 0, 0, 0)
        _check(lib().mi_pt_render_frames_async(self._h, camera_id, width, height, win, n_frames, seed, first_sample, t))
        self._frame_shape = (height, width, 4)
        return list(t)

    def wait(self, ticket, copy=True):
        """mi_pt_wait: the frame's [H][W][4] sums and its statistics.  copy=False returns a VIEW of the handle's pinned buffer: it is only valid until
        the batch slot is enqueued again (BATCHES_IN_FLIGHT launches later) and is overwritten then without notice — copy it or consume it at once."""
        p, st = C.POINTER(C.c_float)(), PtStats()
        _check(lib().mi_pt_wait(self._h, ticket, C.byref(p), C.byref(st)))
        self.last_stats = st
        a = np.ctypeslib.as_array(p, self._frame_shape)
        return a.copy() if copy else a

    def wait_add(self, ticket, view):
        """mi_pt_wait_add: wait for the frame and add it to the float64 [H][W][4] view (Technique::_commit_images) on the library's host threads."""
        assert view.dtype == np.float64 and view.flags["C_CONTIGUOUS"] and view.shape == self._frame_shape
        st = PtStats()
        _check(lib().mi_pt_wait_add(self._h, ticket, _ptr(view), C.byref(st)))
        self.last_stats = st

    def last_launch(self):
        li = LaunchInfo()
        _check(lib().mi_pt_last_launch(self._h, C.byref(li)))
        return li

    def render_frames(self, view, n_frames, seed=0, camera_id=0, window=None, batch=8):
        """The adapter's frame loop (integration/GpuPathTracing.cpp): `n_frames` calls of Technique::render at one sample per call, with
        batches of `batch` frames in flight — the next batches render while the host adds the frames of this one to the dvec4 view."""
        h, w = view.shape[:2]
        st = self._statistics
        first = st.num_samples
        n_batches = (n_frames + batch - 1) // batch
        tickets = {}

        def enqueue(j):
            if j < n_batches:
                k0 = j * batch
                tickets[j] = self.render_frames_async(w, h, min(batch, n_frames - k0), seed, first + k0, camera_id, window)

        import time
        t_last = time.perf_counter()
        done = {}  # tickets already waited for, per batch
        try:
            for j in range(BATCHES_IN_FLIGHT - 1):
                enqueue(j)
            for k in range(n_frames):
                j, f = divmod(k, batch)
                if f == 0:
                    enqueue(j + BATCHES_IN_FLIGHT - 1)
                self.wait_add(tickets[j][f], view)  # _commit_images (Technique.cpp:222-226)
                done[j] = f + 1
                now = time.perf_counter()
                st.num_samples += 1
                st.num_basic_rays += self.last_stats.num_basic_rays
                st.num_shadow_rays += self.last_stats.num_shadow_rays
                st.total_time += now - t_last
                # one record per frame, like render() (Technique.cpp:61-76); rms / abs errors need a reference image and stay 0 here
                st.records.append(dict(sample_index=st.num_samples - 1, rms_error=0.0, abs_error=0.0, clock_time=st.total_time, frame_duration=now - t_last,
                                       numeric_errors=int(self.last_stats.numeric_errors)))
                t_last = now
        finally:
            # drain: a wait or an enqueue that raised must not leave batches pending on the handle (every later render_frames_async would then
            # fail with 'batches are pending' because the tickets are lost) — the C++ adapter's _drain()
            for j, ts in tickets.items():
                for t in ts[done.get(j, 0):]:
                    try:
                        self.wait(t, copy=False)
                    except MiError:
                        pass

    def render(self, view, seed=0, camera_id=0, reference=None, window=None, spp=1):
        """Technique::render (Technique.cpp:15-77): adds `spp` frames (default 1, as the reference) to
        `view`, a float64 [H][W][4] array of (R,G,B sums, denom) — subimage_view_t's dvec4 data."""
        import time
        t0 = time.perf_counter()
        h, w = view.shape[:2]
        st = self._statistics
        rgbn = self.render_rgbn(w, h, spp=spp, seed=seed, sample_offset=st.num_samples, camera_id=camera_id, window=window)
        if view.dtype == np.float64 and view.flags["C_CONTIGUOUS"]:
            view_add_frame(view, rgbn)  # _commit_images (Technique.cpp:222-226) on the library's host threads; non-finite samples were dropped on the device
        else:
            view += rgbn.astype(np.float64)
        elapsed = time.perf_counter() - t0
        st.num_samples += spp
        st.num_basic_rays += self.last_stats.num_basic_rays
        st.num_shadow_rays += self.last_stats.num_shadow_rays
        st.total_time += elapsed
        rec = dict(sample_index=st.num_samples - 1, rms_error=0.0, abs_error=0.0, clock_time=st.total_time, frame_duration=elapsed,
                   numeric_errors=int(self.last_stats.numeric_errors))
        if reference is not None:
            if view.dtype == np.float64 and view.flags["C_CONTIGUOUS"]:
                rec["rms_error"], rec["abs_error"] = rms_abs_errors_view(view, reference)  # over the dvec4 sums themselves, like the reference
            else:
                rec["rms_error"], rec["abs_error"] = rms_abs_errors(view.astype(np.float32), reference)
        st.records.append(rec)
        return rec

    # -- scene services (parity hooks) -------------------------------------------------------
    def set_kernel(self, kernel):
        _check(lib().mi_pt_set_kernel(self._h, kernel))

    # -- one process per GPU: RCCL sum-reduce of the device framebuffers (merge_exr, Options.cpp:1340-1409) --
    def reduce_init(self, unique_id, rank, world):
        buf = (C.c_ubyte * REDUCE_ID_BYTES).from_buffer_copy(bytes(unique_id))
        _check(lib().mi_pt_reduce_init(self._h, buf, rank, world))

    def reduce_rgbn(self, device_ptr, width, height, root=-1, stream=None):
        _check(lib().mi_pt_reduce_rgbn(self._h, C.c_void_p(device_ptr), width, height, int(root), C.c_void_p(stream) if stream else None))

    def reduce_finalize(self):
        _check(lib().mi_pt_reduce_finalize(self._h))

    def set_tile_shard(self, rank, world):
        """Render only the 32x32 tiles {t : t mod world == rank} of the window (Technique.cpp:167); world <= 1 = off."""
        _check(lib().mi_pt_set_tile_shard(self._h, rank, world))

    def get_kernel(self):
        return lib().mi_pt_get_kernel(self._h)

    def set_instrumented(self, on):
        _check(lib().mi_pt_set_instrumented(self._h, 1 if on else 0))

    def intersect(self, origins, directions):
        """Scene::intersect + querySurface for n rays.  origins: SURFACE_DTYPE array."""
        origins = np.ascontiguousarray(origins, SURFACE_DTYPE)
        directions = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        n = len(origins)
        hits = np.zeros(n, SURFACE_DTYPE)
        t = np.zeros(n, np.float32)
        prim = np.zeros(n, np.uint32)
        _check(lib().mi_pt_intersect(self._h, n, _ptr(origins), _ptr(directions), _ptr(hits), _ptr(t), _ptr(prim)))
        return hits, t, prim

    def occluded(self, origins, targets):
        origins = np.ascontiguousarray(origins, SURFACE_DTYPE)
        targets = np.ascontiguousarray(targets, SURFACE_DTYPE)
        out = np.zeros(len(origins), np.float32)
        _check(lib().mi_pt_occluded(self._h, len(origins), _ptr(origins), _ptr(targets), _ptr(out)))
        return out

    def trace_paths(self, width, height, pixel_xy, sample_index, seed=0, camera_id=0):
        pixel_xy = np.ascontiguousarray(pixel_xy, np.uint32).reshape(-1, 2)
        sample_index = np.ascontiguousarray(sample_index, np.uint64)
        n = len(pixel_xy)
        rad = np.zeros((n, 3), np.float32)
        cnt = np.zeros((n, 2), np.uint32)
        _check(lib().mi_pt_trace_paths(self._h, camera_id, width, height, n, _ptr(pixel_xy), _ptr(sample_index), seed, _ptr(rad), _ptr(cnt)))
        return rad, cnt

    def bpt_set_sky(self, horizon, zenith):
        """Technique::set_sky_gradient: colour of camera rays that leave the scene (BPT only)."""
        _check(lib().mi_bpt_set_sky(self._h, (C.c_float * 3)(*horizon), (C.c_float * 3)(*zenith)))

    def bpt_trace_paths(self, width, height, pixel_xy, sample_index, seed=0, camera_id=0):
        """BPT (BPT.cpp) per path: eye-image radiance, sum of light-image splats, (closest rays, shadow rays, splats)."""
        pixel_xy = np.ascontiguousarray(pixel_xy, np.uint32).reshape(-1, 2)
        sample_index = np.ascontiguousarray(sample_index, np.uint64)
        n = len(pixel_xy)
        rad = np.zeros((n, 3), np.float32); spl = np.zeros((n, 3), np.float32); cnt = np.zeros((n, 3), np.uint32)
        _check(lib().mi_bpt_trace_paths(self._h, camera_id, width, height, n, _ptr(pixel_xy), _ptr(sample_index), seed, _ptr(rad), _ptr(spl), _ptr(cnt)))
        return rad, spl, cnt

    def bpt_render_rgbn(self, width, height, spp=1, seed=0, sample_offset=0, camera_id=0, window=None):
        """BPT: `spp` frames of the view window (default: whole image) -> [H][W][4] (R, G, B sums, denom)."""
        out = np.zeros((height, width, 4), np.float32)
        st = PtStats()
        win = Window(*window) if window else Window(0, 0, 0, 0)
        _check(lib().mi_bpt_render(self._h, camera_id, width, height, win, spp, seed, sample_offset, _ptr(out), C.byref(st)))
        self.last_stats = st
        return out

    def blob(self):
        """Device scene blob as ([n][4] float32, dict of section offsets in float4 units)."""
        off = (C.c_uint32 * 7)()
        _check(lib().mi_pt_blob_download(self._h, off, None, 0))
        data = np.zeros((off[6], 4), np.float32)
        _check(lib().mi_pt_blob_download(self._h, off, _ptr(data), off[6]))
        names = ["nodes", "tris", "shade", "materials", "lights", "cdf", "end"]
        return data, dict(zip(names, list(off)))

    def bvh_info(self):
        info = BvhInfo()
        _check(lib().mi_pt_bvh_info(self._h, C.byref(info)))
        return info

    def bvh(self):
        info = self.bvh_info()
        nodes = np.zeros(info.n_nodes, NODE_DTYPE)
        sorted_tri = np.zeros(info.n_triangles, np.uint32)
        morton = np.zeros(info.n_triangles, np.uint64)
        _check(lib().mi_pt_bvh_download(self._h, _ptr(nodes), _ptr(sorted_tri), _ptr(morton)))
        return nodes, sorted_tri, morton
