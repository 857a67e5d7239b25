"""integration/GpuPathTracing.{hpp,cpp} cannot be compiled here (glm / Embree headers are absent and stand-ins are not allowed),
so this test pins the adapter against the reference's own headers as text: every member, signature and type of the reference
the adapter touches must exist there with the shape the adapter assumes, and every symbol of the C ABI it calls must be declared
in include/mi_pt.h.  Build-container only (the reference tree does not travel to the GPU box)."""
import os
import re

import pytest

from conftest import REFERENCE, ROOT

pytestmark = pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="needs /root/reference (build container only)")


def ref(name):
    return open(os.path.join(REFERENCE, name), errors="replace").read()


def adapter():
    return open(os.path.join(ROOT, "integration", "GpuPathTracing.cpp")).read() + open(os.path.join(ROOT, "integration", "GpuPathTracing.hpp")).read()


# (what the adapter writes, reference file, regex that must match there)
USED = [
    # Technique: base class, members render() fills, helpers
    ("public Technique", "Technique.hpp", r"class Technique\s*\{"),
    ("Technique(scene, 1)", "Technique.hpp", r"Technique\(const shared<const Scene>& scene, size_t num_threads\);"),
    ("void render(subimage_view_t& view, RandomEngine& engine, size_t cameraId,", "Technique.hpp",
     r"virtual void render\(\s*subimage_view_t& view,\s*RandomEngine& engine,\s*size_t cameraId,\s*const vector<vec3>& reference,\s*const vector<ivec3>& trace_points\);"),
    ("_statistics", "Technique.hpp", r"statistics_t _statistics;"),
    ("_start_time", "Technique.hpp", r"double _start_time = NAN;"),
    ("_sky_horizon", "Technique.hpp", r"vec3 _sky_horizon"),
    ("_sky_zenith", "Technique.hpp", r"vec3 _sky_zenith"),
    ("_make_measurements(trace_points, a, b)", "Technique.hpp", r"void _make_measurements\(\s*const vector<ivec3>& trace_points,\s*image_view_t<dvec4> a,\s*image_view_t<vec3> b\);"),
    ("high_resolution_time()", "utility.hpp", r"double high_resolution_time\(\);"),
    # statistics_t
    ("_statistics.num_samples", "statistics.hpp", r"size_t num_samples = 0;"),
    ("_statistics.num_basic_rays", "statistics.hpp", r"size_t num_basic_rays = 0;"),
    ("_statistics.num_shadow_rays", "statistics.hpp", r"size_t num_shadow_rays = 0;"),
    ("_statistics.total_time", "statistics.hpp", r"double total_time = 0.0;"),
    ("statistics_t::record_t", "statistics.hpp", r"struct record_t \{"),
    ("record.sample_index", "statistics.hpp", r"size_t sample_index = 0;"),
    ("record.rms_error", "statistics.hpp", r"float rms_error;"),
    ("record.abs_error", "statistics.hpp", r"float abs_error;"),
    ("record.clock_time", "statistics.hpp", r"float clock_time;"),
    ("record.frame_duration", "statistics.hpp", r"float frame_duration;"),
    ("record.numeric_errors", "statistics.hpp", r"size_t numeric_errors;"),
    ("_statistics.records", "statistics.hpp", r"vector<record_t> records;"),
    # subimage_view_t / image_view_t / rms_abs_errors
    ("view.width()", "ImageView.hpp", r"const size_t width\(\) const"),
    ("view.height()", "ImageView.hpp", r"const size_t height\(\) const"),
    ("view.xBegin()", "ImageView.hpp", r"const size_t xBegin\(\) const"),
    ("view.yBegin()", "ImageView.hpp", r"const size_t yBegin\(\) const"),
    ("view.xWindow()", "ImageView.hpp", r"const size_t xWindow\(\) const"),
    ("view.yWindow()", "ImageView.hpp", r"const size_t yWindow\(\) const"),
    ("view.data()", "ImageView.hpp", r"dvec4\* data\(\) \{ return _data; \}"),
    ("image_view_t<dvec4>(view)", "ImageView.hpp", r"image_view_t\(const subimage_view_t& view\)"),
    ("image_view_t<vec3>(reference, view.width(), view.height())", "ImageView.hpp", r"image_view_t\(\s*const vector<T>& data,\s*size_t width,\s*size_t height\)"),
    ("rms_abs_errors(record.rms_error, record.abs_error, a, b)", "ImageView.hpp", r"void rms_abs_errors\("),
    # Scene
    ("scene->meshes", "Scene.hpp", r"const vector<Mesh> meshes;"),
    ("scene->materials.bsdfs", "Scene.hpp", r"const Materials materials;"),
    ("scene->materials.bsdfs", "Materials.hpp", r"vector<unique<BSDF>> bsdfs;"),
    ("scene->lights", "Scene.hpp", r"AreaLights lights;"),
    ("scene->cameras()", "Scene.hpp", r"const Cameras& cameras\(\) const"),
    # Mesh / AreaLight / AreaLights
    ("mesh.vertices", "AreaLights.hpp", r"vector<vec3> vertices;"),
    ("mesh.tangents", "AreaLights.hpp", r"vector<mat3> tangents;"),
    ("mesh.indices", "AreaLights.hpp", r"vector<int> indices;"),
    ("mesh.material_id", "AreaLights.hpp", r"uint32_t material_id;"),
    ("scene->lights.num_lights()", "AreaLights.hpp", r"const size_t num_lights\(\) const;"),
    ("scene->lights.light(i)", "AreaLights.hpp", r"const AreaLight& light\(size_t light_id\) const;"),
    ("l.position", "AreaLights.hpp", r"struct AreaLight \{\s*vec3 position;"),
    ("l.tangent", "AreaLights.hpp", r"mat3 tangent;"),
    ("l.size", "AreaLights.hpp", r"vec2 size;"),
    ("l.exitance", "AreaLights.hpp", r"vec3 exitance;"),
    ("l.diffuse", "AreaLights.hpp", r"float diffuse;"),
    ("l.material_id", "AreaLights.hpp", r"uint32_t material_id;"),
    # Cameras
    ("cameras.numCameras()", "Cameras.hpp", r"const size_t numCameras\(\) const;"),
    ("cameras.position(i)", "Cameras.hpp", r"const vec3& position\(size_t cameraId\) const;"),
    ("cameras.direction(i)", "Cameras.hpp", r"const vec3& direction\(size_t cameraId\) const;"),
    ("cameras.up(i)", "Cameras.hpp", r"const vec3& up\(size_t cameraId\) const;"),
    ("cameras.fovx(i, 1.0f)", "Cameras.hpp", r"const float fovx\(size_t cameraId, float aspect\) const;"),
    # BSDF classes the adapter dispatches on
    ("DiffuseBSDF", "BSDF.hpp", r"class DiffuseBSDF : public BSDF"),
    ("PhongBSDF", "BSDF.hpp", r"class PhongBSDF : public BSDF"),
    ("ReflectionBSDF", "BSDF.hpp", r"class ReflectionBSDF : public DeltaBSDF"),
    ("TransmissionBSDF", "BSDF.hpp", r"class TransmissionBSDF : public DeltaBSDF"),
    ("LightBSDF", "BSDF.hpp", r"class LightBSDF : public BSDF"),
    ("sun_light_bsdf", "BSDF.hpp", r"class sun_light_bsdf : public BSDF"),
    ("CameraBSDF", "BSDF.hpp", r"class CameraBSDF : public BSDF"),
    ("bsdf.light_id()", "BSDF.hpp", r"virtual uint32_t light_id\(\) const;"),
    # registration site
    ("make_technique.cpp:112-130", "make_technique.cpp", r"case Options::PT:\s*result = std::make_shared<PathTracing>\("),
]

# private members the BSDF.hpp accessor patch of INTEGRATION.md exposes: (accessor the adapter calls, member it returns, owning class)
ACCESSORS = [
    ("p->diffuse()", "vec3 _diffuse;", "DiffuseBSDF"),
    ("p->specular()", "vec3 _specular;", "PhongBSDF"),
    ("p->power()", "float _power;", "PhongBSDF"),
    ("p->external_over_internal_ior()", "float externalOverInternalIOR;", "TransmissionBSDF"),
]


@pytest.mark.parametrize("used,header,pattern", USED, ids=[u[0] for u in USED])
def test_reference_has_what_the_adapter_uses(used, header, pattern):
    assert used in adapter(), "the adapter no longer uses %r: update this list" % used
    assert re.search(pattern, ref(header)), "%s: no match for %r" % (header, pattern)


def test_accessor_patch_matches_the_private_members():
    hdr = ref("BSDF.hpp")
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for call, member, cls in ACCESSORS:
        assert call in adapter()
        body = hdr[hdr.index("class %s " % cls):]
        body = body[:body.index("};")]
        assert member in body and body.index("private:") < body.index(member), "%s is expected to be a private member of %s" % (member, cls)
        name = call.split("->")[1].rstrip("()")
        assert re.search(r"\+\s+.*\b%s\(\) const \{ return %s; \}" % (name, member.split()[-1].rstrip(";")), integ), "INTEGRATION.md's BSDF.hpp patch lacks %s()" % name


def test_every_abi_call_of_the_adapter_is_declared():
    src = adapter()
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mi_pt.h")).read(), flags=re.S)
    calls = sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", src)))
    assert {"mi_pt_create", "mi_pt_destroy", "mi_pt_render_frames_async", "mi_pt_wait_add", "mi_pt_wait", "mi_view_add_frame", "mi_pt_render_multi", "mi_bpt_render"} <= set(calls)
    for c in calls:
        assert re.search(r"\b%s\s*\(" % c, hdr), c
    for macro in re.findall(r"\bMI_[A-Z_]+\b", src):
        assert macro in hdr, macro


def test_flatten_covers_every_bsdf_class_of_the_reference():
    classes = re.findall(r"class (\w+) : public (?:BSDF|DeltaBSDF)", ref("BSDF.hpp"))
    src = adapter()
    for c in classes:
        if c == "DeltaBSDF":  # abstract base of the two delta materials
            continue
        assert "dynamic_cast<const %s*>" % c in src, c


def test_reference_patch_applies_cleanly():
    """the whole change to the reference — the BSDF accessors, `--gpu[=<device>]` in Options, the `case Options::PT` of
    make_technique.cpp, the Makefile lines — is integration/reference.patch.  A dry run (`git apply --check`, nothing is written, nothing of the
    reference is copied) against the reference tree next to this repository must succeed, so the patch cannot rot against the files it edits."""
    import shutil
    import subprocess
    if not shutil.which("git"):
        pytest.skip("git not available")
    patch = os.path.join(ROOT, "integration", "reference.patch")
    r = subprocess.run(["git", "apply", "--check", "--verbose", patch], cwd=REFERENCE, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    text = open(patch).read()
    touched = sorted(set(re.findall(r"^\+\+\+ b/(\S+)", text, flags=re.M)))
    assert touched == ["BSDF.hpp", "Makefile", "Options.cpp", "Options.hpp", "make_technique.cpp"]
    # the patch wires up exactly what the adapter offers: constructor arguments in order (+ the device), accessors by name
    assert "std::make_shared<GpuPathTracing>(" in text and "options.gpu_device);" in text and "#include <GpuPathTracing.hpp>" in text
    for call, _, _ in ACCESSORS:
        assert "+  " in text and call.split("->")[1] in text
    ctor = re.search(r"GpuPathTracing\(([^)]*)\)", open(os.path.join(ROOT, "integration", "GpuPathTracing.hpp")).read()).group(1)
    names = [a.split("=")[0].strip().split()[-1] for a in ctor.split(",")]
    assert names[:7] == ["scene", "lights", "roulette", "beta", "max_path", "num_threads", "device"], names  # the order make_technique.cpp passes them in
