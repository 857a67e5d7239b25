// GpuPathTracing.cpp — flattens haste::Scene into mi_scene_desc, calls the C ABI once per frame and
// fills statistics_t exactly as Technique::render does (Technique.cpp:15-77).
#include <GpuPathTracing.hpp>

#include <random>
#include <stdexcept>

namespace haste {

namespace {

void check(int rc) {
  if (rc != MI_OK) throw std::runtime_error(mi_pt_last_error());  // runtime_assert.cpp:7-11 analogue
}

mi_material flatten(const BSDF& bsdf) {
  mi_material m = {};
  if (auto p = dynamic_cast<const DiffuseBSDF*>(&bsdf)) { m.type = MI_BSDF_DIFFUSE; /* copy p->_diffuse (needs a getter or friend) */ (void)p; }
  else if (dynamic_cast<const PhongBSDF*>(&bsdf)) m.type = MI_BSDF_PHONG;         // _diffuse, _specular, _power
  else if (dynamic_cast<const ReflectionBSDF*>(&bsdf)) m.type = MI_BSDF_REFLECTION;
  else if (dynamic_cast<const TransmissionBSDF*>(&bsdf)) m.type = MI_BSDF_TRANSMISSION;  // internalIOR, externalOverInternalIOR
  else if (dynamic_cast<const LightBSDF*>(&bsdf)) { m.type = MI_BSDF_LIGHT; m.light_id = bsdf.light_id(); }
  else if (dynamic_cast<const sun_light_bsdf*>(&bsdf)) { m.type = MI_BSDF_SUN; m.light_id = bsdf.light_id(); }
  else m.type = MI_BSDF_CAMERA;
  return m;
}

}  // namespace

GpuPathTracing::GpuPathTracing(const shared<const Scene>& scene, float lights, float roulette, float beta,
                               size_t max_path, size_t num_threads, int device, bool bidirectional)
    : Technique(scene, 1), _bidirectional(bidirectional) {
  (void)num_threads;
  std::vector<float> positions, tangents;
  std::vector<uint32_t> indices, offsets{0}, mesh_material;
  for (const Mesh& mesh : scene->meshes) {              // incl. the light quads appended by load_lights (loader.cpp:448)
    const uint32_t base = uint32_t(positions.size() / 3);
    for (const vec3& v : mesh.vertices) { positions.push_back(v.x); positions.push_back(v.y); positions.push_back(v.z); }
    for (const mat3& t : mesh.tangents) for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) tangents.push_back(t[c][r]);
    for (int i : mesh.indices) indices.push_back(base + uint32_t(i));
    offsets.push_back(uint32_t(indices.size() / 3));
    mesh_material.push_back(mesh.material_id);
  }
  std::vector<mi_material> materials;
  for (auto& bsdf : scene->materials.bsdfs) materials.push_back(flatten(*bsdf));
  std::vector<mi_light> mlights;
  for (size_t i = 0; i < scene->lights.num_lights(); ++i) {
    const AreaLight& l = scene->lights.light(i);
    mi_light m = {};
    for (int k = 0; k < 3; ++k) { m.position[k] = l.position[k]; m.exitance[k] = l.exitance[k]; }
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) m.tangent[3 * c + r] = l.tangent[c][r];
    m.size[0] = l.size.x; m.size[1] = l.size.y; m.diffuse = l.diffuse != 0.0f; m.material_id = l.material_id;
    mlights.push_back(m);
  }
  std::vector<mi_camera> cams;
  const Cameras& cameras = scene->cameras();
  for (size_t i = 0; i < cameras.numCameras(); ++i) {
    mi_camera c = {};
    for (int k = 0; k < 3; ++k) { c.position[k] = cameras.position(i)[k]; c.direction[k] = cameras.direction(i)[k]; c.up[k] = cameras.up(i)[k]; }
    c.fovx = cameras.fovx(i, 1.0f);
    cams.push_back(c);
  }
  mi_scene_desc d = {};
  d.n_vertices = uint32_t(positions.size() / 3); d.n_triangles = uint32_t(indices.size() / 3);
  d.n_meshes = uint32_t(mesh_material.size()); d.n_materials = uint32_t(materials.size());
  d.n_lights = uint32_t(mlights.size()); d.n_cameras = uint32_t(cams.size());
  d.positions = positions.data(); d.tangents = tangents.data(); d.indices = indices.data();
  d.mesh_tri_offset = offsets.data(); d.mesh_material_id = mesh_material.data();
  d.materials = materials.data(); d.lights = mlights.data(); d.cameras = cams.data();
  mi_pt_params p = {uint64_t(max_path), beta, roulette, lights, 3};
  // copies the scene, builds the BVH on the GPU; a zero bounding_sphere is computed as loader.cpp:408-432
  const int first = device < 0 ? 0 : device, count = device < 0 && !bidirectional ? mi_pt_device_count() : 1;
  for (int k = 0; k < (count > 0 ? count : 1); ++k) {
    mi_pt_handle* h = nullptr;
    check(mi_pt_create(&d, &p, first + k, &h));
    _handles.push_back(h);
  }
  _handle = _handles[0];
  _seed = std::random_device()();                          // like Sample.inl:249-252: PT is not seedable
}

GpuPathTracing::~GpuPathTracing() { for (mi_pt_handle* h : _handles) mi_pt_destroy(h); }

void GpuPathTracing::render(subimage_view_t& view, RandomEngine&, size_t cameraId, const vector<vec3>& reference,
                            const vector<ivec3>& trace_points) {
  if (!std::isfinite(_start_time)) {
    double offset = _statistics.records.empty() ? 0.0 : _statistics.records.back().clock_time;
    _start_time = high_resolution_time() - offset;      // Technique.cpp:24-30
  }
  const double start_time = high_resolution_time();
  _rgbn.resize(view.width() * view.height() * 4);
  mi_window win = {uint32_t(view.xBegin()), uint32_t(view.yBegin()), uint32_t(view.xWindow()), uint32_t(view.yWindow())};
  mi_pt_stats st = {};
  if (_bidirectional) check(mi_bpt_set_sky(_handle, &_sky_horizon.x, &_sky_zenith.x));  // Technique::set_sky_gradient (Technique.cpp:90-93)
  if (_bidirectional)  // light-image splats land anywhere; light + eye are committed per frame inside, for the window only
    check(mi_bpt_render(_handle, uint32_t(cameraId), uint32_t(view.width()), uint32_t(view.height()), win,
                        /*spp=*/1, _seed, /*sample_offset=*/_statistics.num_samples, _rgbn.data(), &st));
  else if (_handles.size() > 1)  // one frame, its tiles dealt to all GPUs of this process; bit-identical to one GPU
    check(mi_pt_render_multi(_handles.data(), uint32_t(_handles.size()), uint32_t(cameraId), uint32_t(view.width()), uint32_t(view.height()), win,
                             /*spp=*/1, _seed, /*sample_offset=*/_statistics.num_samples, _rgbn.data(), &st));
  else
    check(mi_pt_render(_handle, uint32_t(cameraId), uint32_t(view.width()), uint32_t(view.height()), win,
                       /*spp=*/1, _seed, /*sample_offset=*/_statistics.num_samples, _rgbn.data(), &st));
  for (size_t y = view.yBegin(); y < view.yEnd(); ++y)    // _commit_images (Technique.cpp:215-236)
    for (size_t x = view.xBegin(); x < view.xEnd(); ++x) {
      const float* s = &_rgbn[(y * view.width() + x) * 4];
      view.absAt(x, y) += dvec4(s[0], s[1], s[2], s[3]);
    }
  const double now = high_resolution_time();
  ++_statistics.num_samples;                               // Technique.cpp:55-67
  _statistics.num_basic_rays += st.num_basic_rays;
  _statistics.num_shadow_rays += st.num_shadow_rays;
  _statistics.total_time = now - _start_time;
  statistics_t::record_t record;
  record.sample_index = _statistics.num_samples - 1;
  record.rms_error = record.abs_error = 0.0f;
  record.clock_time = float(_statistics.total_time);
  record.frame_duration = float(now - start_time);
  record.numeric_errors = st.numeric_errors;
  if (!reference.empty()) {
    auto a = image_view_t<dvec4>(view);
    auto b = image_view_t<vec3>(reference, view.width(), view.height());
    rms_abs_errors(record.rms_error, record.abs_error, a, b);
    _make_measurements(trace_points, a, b);
  }
  _statistics.records.push_back(record);
}

}  // namespace haste
