// GpuPathTracing.cpp — flattens haste::Scene into mi_scene_desc, keeps frames in flight through the C ABI and
// fills statistics_t exactly as Technique::render does (Technique.cpp:15-77).
#include <GpuPathTracing.hpp>

#include <random>
#include <stdexcept>

namespace haste {

namespace {

constexpr unsigned kBatchesAhead = MI_PT_BATCHES_IN_FLIGHT - 1;     // one slot is being consumed, the others render
// frames of one launch (<= MI_PT_MAX_FRAMES_PER_BATCH): a 512 x 512 frame is 1.3 rounds of waves, so small frames go eight to a launch (0.131 ms per
// frame against 0.152 with four); from a megapixel on the frame fills the chip by itself and the copy (16 B per pixel) is the limit
unsigned frames_per_batch(size_t width, size_t height) { return width * height <= (size_t(1) << 20) ? 8u : 4u; }

void check(int rc) {
  if (rc != MI_OK) throw std::runtime_error(mi_pt_last_error());  // runtime_assert.cpp:7-11 analogue
}

// One entry of Materials::bsdfs -> mi_material.  Needs the accessors INTEGRATION.md adds to BSDF.hpp (the parameters are
// private there): DiffuseBSDF::diffuse(), PhongBSDF::diffuse() / specular() / power(), TransmissionBSDF::external_over_internal_ior().
// An unknown BSDF class throws: a material silently rendered black would be a wrong image, not an error.
mi_material flatten(const BSDF& bsdf) {
  mi_material m = {};
  if (auto p = dynamic_cast<const DiffuseBSDF*>(&bsdf)) {
    m.type = MI_BSDF_DIFFUSE;
    const vec3 d = p->diffuse();
    m.diffuse[0] = d.x; m.diffuse[1] = d.y; m.diffuse[2] = d.z;
  } else if (auto p = dynamic_cast<const PhongBSDF*>(&bsdf)) {
    m.type = MI_BSDF_PHONG;
    const vec3 d = p->diffuse(), s = p->specular();
    m.diffuse[0] = d.x; m.diffuse[1] = d.y; m.diffuse[2] = d.z;
    m.specular[0] = s.x; m.specular[1] = s.y; m.specular[2] = s.z;
    m.power = p->power();                      // the diffuse-lobe probability (BSDF.cpp:306-315) is recomputed by mi_pt_create
  } else if (dynamic_cast<const ReflectionBSDF*>(&bsdf)) {
    m.type = MI_BSDF_REFLECTION;
  } else if (auto p = dynamic_cast<const TransmissionBSDF*>(&bsdf)) {
    m.type = MI_BSDF_TRANSMISSION;
    // the integrator only ever uses externalIOR / internalIOR (BSDF.cpp:467-470,480,487): hand over that quotient exactly
    m.ior_internal = 1.0f;
    m.ior_external = p->external_over_internal_ior();
  } else if (dynamic_cast<const LightBSDF*>(&bsdf)) {
    m.type = MI_BSDF_LIGHT; m.light_id = bsdf.light_id();
  } else if (dynamic_cast<const sun_light_bsdf*>(&bsdf)) {
    m.type = MI_BSDF_SUN; m.light_id = bsdf.light_id();
  } else if (dynamic_cast<const CameraBSDF*>(&bsdf)) {
    m.type = MI_BSDF_CAMERA;
  } else {
    throw std::runtime_error("GpuPathTracing: a BSDF class without a mi_material mapping");
  }
  return m;
}

// compute_bounding_sphere (loader.cpp:408-432) over the surface meshes — what load_lights hands to the emitters' BSDFs
// (loader.cpp:436,452) before it appends the light quads; same float arithmetic
void bounding_sphere(const vector<Mesh>& meshes, float out[4]) {
  vec3 center = vec3(0.0f);
  float radius = 0.0f;
  size_t num_vertices = 0;
  for (const Mesh& mesh : meshes) {
    if ((mesh.material_id & 3u) != MI_ENTITY_MESH) continue;   // light quads (entity_type::light, SurfacePoint.hpp:8-21)
    for (const vec3& vertex : mesh.vertices) center += vertex;
    num_vertices += mesh.vertices.size();
  }
  if (num_vertices == 0) { out[0] = out[1] = out[2] = out[3] = 0.0f; return; }
  center /= static_cast<float>(num_vertices);
  for (const Mesh& mesh : meshes) {
    if ((mesh.material_id & 3u) != MI_ENTITY_MESH) continue;
    for (const vec3& vertex : mesh.vertices) radius = glm::max(radius, glm::distance2(center, vertex));
  }
  out[0] = center.x; out[1] = center.y; out[2] = center.z; out[3] = glm::sqrt(radius);
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
    c.fovx = cameras.fovx(i, 1.0f);                      // Desc::fovx as given to addCameraFovX (Cameras.cpp:72-79)
    cams.push_back(c);
  }
  mi_scene_desc d = {};
  d.n_vertices = uint32_t(positions.size() / 3); d.n_triangles = uint32_t(indices.size() / 3);
  d.n_meshes = uint32_t(mesh_material.size()); d.n_materials = uint32_t(materials.size());
  d.n_lights = uint32_t(mlights.size()); d.n_cameras = uint32_t(cams.size());
  d.positions = positions.data(); d.tangents = tangents.data(); d.indices = indices.data();
  d.mesh_tri_offset = offsets.data(); d.mesh_material_id = mesh_material.data();
  d.materials = materials.data(); d.lights = mlights.data(); d.cameras = cams.data();
  bounding_sphere(scene->meshes, d.bounding_sphere);
  mi_pt_params p = {uint64_t(max_path), beta, roulette, lights, 3};
  // copies the scene, builds the BVH on the GPU
  const int first = device < 0 ? 0 : device, count = device < 0 && !bidirectional ? mi_pt_device_count() : 1;
  for (int k = 0; k < (count > 0 ? count : 1); ++k) {
    mi_pt_handle* h = nullptr;
    check(mi_pt_create(&d, &p, first + k, &h));
    _handles.push_back(h);
  }
  _handle = _handles[0];
  _seed = std::random_device()();                          // like Sample.inl:249-252: PT is not seedable
}

GpuPathTracing::~GpuPathTracing() {
  try { _drain(); } catch (...) { }
  for (mi_pt_handle* h : _handles) mi_pt_destroy(h);
}

// frames that were rendered ahead for another camera / resolution / window / sample index: wait for them and drop them
void GpuPathTracing::_drain() {
  while (!_frames.empty()) {
    const float* unused = nullptr;
    check(mi_pt_wait(_handle, _frames.front().ticket, &unused, nullptr));
    _frames.pop_front();
  }
}

void GpuPathTracing::_render_ahead(const FrameKey& key, mi_window win) {
  const unsigned batch = frames_per_batch(key.width, key.height);
  while (_frames.size() + batch <= kBatchesAhead * batch || _frames.empty()) {
    uint64_t tickets[MI_PT_MAX_FRAMES_PER_BATCH];
    check(mi_pt_render_frames_async(_handle, uint32_t(key.camera), uint32_t(key.width), uint32_t(key.height), win, batch, _seed,
                                    _next_sample, tickets));
    for (unsigned f = 0; f < batch; ++f) _frames.push_back(Frame{tickets[f], _next_sample + f});
    _next_sample += batch;
  }
}

void GpuPathTracing::render(subimage_view_t& view, RandomEngine&, size_t cameraId, const vector<vec3>& reference,
                            const vector<ivec3>& trace_points) {
  if (!std::isfinite(_start_time)) {
    double offset = _statistics.records.empty() ? 0.0 : _statistics.records.back().clock_time;
    _start_time = high_resolution_time() - offset;      // Technique.cpp:24-30
  }
  const double start_time = high_resolution_time();
  mi_window win = {uint32_t(view.xBegin()), uint32_t(view.yBegin()), uint32_t(view.xWindow()), uint32_t(view.yWindow())};
  double* const sums = reinterpret_cast<double*>(view.data());   // dvec4 = (R, G, B sums, denom), row 0 = bottom (ImageView.hpp:10-67)
  mi_pt_stats st = {};
  if (_bidirectional || _handles.size() > 1) {
    _rgbn.resize(view.width() * view.height() * 4);
    if (_bidirectional) {  // light-image splats land anywhere; light + eye are committed per frame inside, for the window only
      check(mi_bpt_set_sky(_handle, &_sky_horizon.x, &_sky_zenith.x));  // Technique::set_sky_gradient (Technique.cpp:90-93)
      check(mi_bpt_render(_handle, uint32_t(cameraId), uint32_t(view.width()), uint32_t(view.height()), win,
                          /*spp=*/1, _seed, /*sample_offset=*/_statistics.num_samples, _rgbn.data(), &st));
    } else {               // one frame, its tiles dealt to all GPUs of this process; bit-identical to one GPU
      check(mi_pt_render_multi(_handles.data(), uint32_t(_handles.size()), uint32_t(cameraId), uint32_t(view.width()), uint32_t(view.height()), win,
                               /*spp=*/1, _seed, /*sample_offset=*/_statistics.num_samples, _rgbn.data(), &st));
    }
    check(mi_view_add_frame(_rgbn.data(), sums, uint32_t(view.width()), uint32_t(view.height()), win));   // _commit_images (Technique.cpp:215-236)
  } else {
    // PT on one GPU: the frame of this call was enqueued by an earlier call (or is enqueued now); the batches after it render and
    // cross PCIe while the host adds this one.  After the call the view holds exactly the frames 0 .. num_samples.
    const FrameKey key{cameraId, view.width(), view.height(), view.xBegin(), view.yBegin(), view.xWindow(), view.yWindow()};
    if (!(key == _key) || (!_frames.empty() && _frames.front().sample_index != _statistics.num_samples)) {
      _drain();                                         // another camera / size / window, or set_statistics() moved the sample index (`continue`)
      _key = key;
      _next_sample = _statistics.num_samples;
    }
    if (_frames.empty()) _next_sample = _statistics.num_samples;
    _render_ahead(key, win);
    check(mi_pt_wait_add(_handle, _frames.front().ticket, sums, &st));   // wait + _commit_images (Technique.cpp:215-236)
    _frames.pop_front();
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
