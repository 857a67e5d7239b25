// scene_host.cpp — scene container, .miscene files, camera helpers, error reporting.
// Host-only C++; part of libmi_pt.so.
#include "scene_host.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

namespace mi {

static thread_local std::string g_last_error;

void set_last_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

mi_scene_desc SceneData::desc() const {
  mi_scene_desc d;
  std::memset(&d, 0, sizeof d);
  d.n_vertices = uint32_t(positions.size() / 3);
  d.n_triangles = uint32_t(indices.size() / 3);
  d.n_meshes = uint32_t(mesh_material_id.size());
  d.n_materials = uint32_t(materials.size());
  d.n_lights = uint32_t(lights.size());
  d.n_cameras = uint32_t(cameras.size());
  d.positions = positions.data();
  d.tangents = tangents.data();
  d.indices = indices.data();
  d.mesh_tri_offset = mesh_tri_offset.data();
  d.mesh_material_id = mesh_material_id.data();
  d.materials = materials.data();
  d.lights = lights.data();
  d.cameras = cameras.data();
  std::memcpy(d.bounding_sphere, bounding_sphere, sizeof bounding_sphere);
  return d;
}

// The checks the reference performs lazily with runtime_assert (Scene.cpp:76,84,136,145;
// AreaLights.cpp:104-110,217) are done once, up front.
std::string SceneData::validate() const {
  const size_t nv = positions.size() / 3, nt = indices.size() / 3, nm = mesh_material_id.size();
  if (positions.size() % 3 || indices.size() % 3) return "positions/indices size is not a multiple of 3";
  if (tangents.size() != nv * 9) return "tangents must hold one mat3 per vertex";
  if (nt == 0) return "scene has no triangles";
  if (mesh_tri_offset.size() != nm + 1) return "mesh_tri_offset must have n_meshes + 1 entries";
  if (mesh_tri_offset.front() != 0 || mesh_tri_offset.back() != nt) return "mesh_tri_offset must span [0, n_triangles]";
  for (size_t m = 0; m < nm; ++m) {
    if (mesh_tri_offset[m] > mesh_tri_offset[m + 1]) return "mesh_tri_offset is not monotonic";
    uint32_t id = mesh_material_id[m];
    if ((id >> 2) >= materials.size()) return "mesh material index out of range";
    uint32_t ent = id & 3u;
    if (ent != MI_ENTITY_MESH && ent != MI_ENTITY_LIGHT) return "mesh entity type must be mesh or light";
    uint32_t mt = materials[id >> 2].type;
    if (ent == MI_ENTITY_LIGHT) {
      if (mt != MI_BSDF_LIGHT && mt != MI_BSDF_SUN) return "light mesh must reference a light material";
      if (materials[id >> 2].light_id >= lights.size()) return "light material has light_id out of range";
    } else if (mt == MI_BSDF_LIGHT || mt == MI_BSDF_SUN || mt == MI_BSDF_CAMERA) {
      return "surface mesh must reference a surface material";
    }
  }
  for (uint32_t i : indices)
    if (i >= nv) return "vertex index out of range";
  for (const mi_material& m : materials) {
    if (m.type > MI_BSDF_SUN) return "unknown material type";
    // parameters a kernel would turn into NaNs or a black surface without saying so (an adapter that forgot to copy them)
    for (int k = 0; k < 3; ++k)
      if (!std::isfinite(m.diffuse[k]) || !std::isfinite(m.specular[k])) return "non-finite material colour";
    // the reference accepts an exponent of 0 (loader.cpp:220-224 defaults AI_MATKEY_SHININESS to 0; PhongBSDF, BSDF.cpp:306-315: 2 pi specular / 1)
    if (m.type == MI_BSDF_PHONG && !(m.power >= 0.0f && std::isfinite(m.power))) return "Phong material needs a non-negative, finite exponent (PhongBSDF, BSDF.cpp:306-315)";
    if (m.type == MI_BSDF_TRANSMISSION && !(std::isfinite(m.ior_internal) && std::isfinite(m.ior_external) && m.ior_internal != 0.0f && m.ior_external != 0.0f))
      return "transmission material needs finite, non-zero indices of refraction (TransmissionBSDF, BSDF.cpp:467-470)";
  }
  for (const mi_light& l : lights) {
    if ((l.material_id >> 2) >= materials.size()) return "light material_id out of range";
    if (!(l.size[0] > 0.0f) || !(l.size[1] > 0.0f)) return "light size must be positive";
  }
  float total_power = 0.0f;
  for (const mi_light& l : lights) {
    const float e = std::fabs(l.exitance[0]) + std::fabs(l.exitance[1]) + std::fabs(l.exitance[2]);
    if (!std::isfinite(e)) return "non-finite light exitance";
    total_power += l.size[0] * l.size[1] * e;
  }
  if (!lights.empty() && !(total_power > 0.0f)) return "total light power must be positive (AreaLights.cpp:199-209 divides by it)";
  for (float v : positions)
    if (!std::isfinite(v)) return "non-finite vertex position";
  // tangent frames are not checked: a zero-area triangle has no frame (loader.cpp:336 normalises a zero vector) and can never be hit
  return "";
}

SceneData SceneData::from_desc(const mi_scene_desc& d) {
  SceneData s;
  s.positions.assign(d.positions, d.positions + size_t(d.n_vertices) * 3);
  s.tangents.assign(d.tangents, d.tangents + size_t(d.n_vertices) * 9);
  s.indices.assign(d.indices, d.indices + size_t(d.n_triangles) * 3);
  s.mesh_tri_offset.assign(d.mesh_tri_offset, d.mesh_tri_offset + d.n_meshes + 1);
  s.mesh_material_id.assign(d.mesh_material_id, d.mesh_material_id + d.n_meshes);
  s.materials.assign(d.materials, d.materials + d.n_materials);
  s.lights.assign(d.lights, d.lights + d.n_lights);
  s.cameras.assign(d.cameras, d.cameras + d.n_cameras);
  std::memcpy(s.bounding_sphere, d.bounding_sphere, sizeof s.bounding_sphere);
  s.material_names.resize(d.n_materials);
  s.mesh_names.resize(d.n_meshes);
  return s;
}

// ---- .miscene: "MISCENE1" + counts + raw arrays + name table (little endian) ----
namespace {
struct FileHeader {
  char magic[8];
  uint32_t n_vertices, n_triangles, n_meshes, n_materials, n_lights, n_cameras;
  float bounding_sphere[4];
};
template <class T> bool wr(FILE* f, const std::vector<T>& v) { return v.empty() || std::fwrite(v.data(), sizeof(T), v.size(), f) == v.size(); }
template <class T> bool rd(FILE* f, std::vector<T>& v, size_t n) { v.resize(n); return n == 0 || std::fread(v.data(), sizeof(T), n, f) == n; }
bool wr_names(FILE* f, const std::vector<std::string>& names) {
  for (const std::string& s : names) {
    uint32_t n = uint32_t(s.size());
    if (std::fwrite(&n, 4, 1, f) != 1) return false;
    if (n && std::fwrite(s.data(), 1, n, f) != n) return false;
  }
  return true;
}
bool rd_names(FILE* f, std::vector<std::string>& names, size_t count) {
  names.resize(count);
  for (std::string& s : names) {
    uint32_t n;
    if (std::fread(&n, 4, 1, f) != 1 || n > 4096) return false;
    s.resize(n);
    if (n && std::fread(&s[0], 1, n, f) != n) return false;
  }
  return true;
}
}  // namespace

int save_miscene(const SceneData& s, const char* path) {
  FILE* f = std::fopen(path, "wb");
  if (!f) return fail(MI_ERR_IO, std::string("Cannot write \"") + path + "\".");
  FileHeader h;
  std::memcpy(h.magic, "MISCENE1", 8);
  mi_scene_desc d = s.desc();
  h.n_vertices = d.n_vertices; h.n_triangles = d.n_triangles; h.n_meshes = d.n_meshes;
  h.n_materials = d.n_materials; h.n_lights = d.n_lights; h.n_cameras = d.n_cameras;
  std::memcpy(h.bounding_sphere, s.bounding_sphere, sizeof h.bounding_sphere);
  std::vector<std::string> mat_names = s.material_names, mesh_names = s.mesh_names;
  mat_names.resize(d.n_materials); mesh_names.resize(d.n_meshes);
  bool ok = std::fwrite(&h, sizeof h, 1, f) == 1 && wr(f, s.positions) && wr(f, s.tangents) && wr(f, s.indices) &&
            wr(f, s.mesh_tri_offset) && wr(f, s.mesh_material_id) && wr(f, s.materials) && wr(f, s.lights) &&
            wr(f, s.cameras) && wr_names(f, mat_names) && wr_names(f, mesh_names);
  std::fclose(f);
  return ok ? MI_OK : fail(MI_ERR_IO, std::string("Short write to \"") + path + "\".");
}

int load_miscene(const char* path, SceneData& s) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return fail(MI_ERR_IO, std::string("Cannot load \"") + path + "\" scene.");
  FileHeader h;
  bool ok = std::fread(&h, sizeof h, 1, f) == 1 && std::memcmp(h.magic, "MISCENE1", 8) == 0;
  ok = ok && h.n_vertices < (1u << 30) && h.n_triangles < (1u << 30) && h.n_meshes < (1u << 24) &&
       h.n_materials < (1u << 24) && h.n_lights < (1u << 20) && h.n_cameras < (1u << 16);
  ok = ok && rd(f, s.positions, size_t(h.n_vertices) * 3) && rd(f, s.tangents, size_t(h.n_vertices) * 9) &&
       rd(f, s.indices, size_t(h.n_triangles) * 3) && rd(f, s.mesh_tri_offset, size_t(h.n_meshes) + 1) &&
       rd(f, s.mesh_material_id, h.n_meshes) && rd(f, s.materials, h.n_materials) && rd(f, s.lights, h.n_lights) &&
       rd(f, s.cameras, h.n_cameras) && rd_names(f, s.material_names, h.n_materials) &&
       rd_names(f, s.mesh_names, h.n_meshes);
  std::fclose(f);
  if (!ok) return fail(MI_ERR_IO, std::string("Cannot load \"") + path + "\" scene (bad or truncated .miscene).");
  std::memcpy(s.bounding_sphere, h.bounding_sphere, sizeof h.bounding_sphere);
  return MI_OK;
}

// ---- Cameras.cpp restated with glm's formulas (host, float) ----
namespace {
struct V3 { float x, y, z; };
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
// dot / cross as the arithmetic contract defines them: fused exactly where spelled
inline float dot(V3 a, V3 b) { return std::fmaf(a.z, b.z, std::fmaf(a.y, b.y, a.x * b.x)); }
inline V3 cross(V3 a, V3 b) { return {std::fmaf(a.y, b.z, -(b.y * a.z)), std::fmaf(a.z, b.x, -(b.z * a.x)), std::fmaf(a.x, b.y, -(b.x * a.y))}; }
inline V3 normalize(V3 a) { float s = 1.0f / std::sqrt(dot(a, a)); return {a.x * s, a.y * s, a.z * s}; }
// column-major 3x3: m[3*c + r]
void inverse3(const float* m, float* r) {
  float a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
  float inv = 1.0f / (a * (e * i - h * f) - d * (b * i - h * c) + g * (b * f - e * c));
  r[0] = (e * i - h * f) * inv; r[1] = -(b * i - h * c) * inv; r[2] = (b * f - e * c) * inv;
  r[3] = -(d * i - g * f) * inv; r[4] = (a * i - g * c) * inv; r[5] = -(a * f - d * c) * inv;
  r[6] = (d * h - g * e) * inv; r[7] = -(a * h - g * b) * inv; r[8] = (a * e - d * b) * inv;
}
void transpose3(const float* m, float* r) {
  r[0] = m[0]; r[1] = m[3]; r[2] = m[6]; r[3] = m[1]; r[4] = m[4]; r[5] = m[7]; r[6] = m[2]; r[7] = m[5]; r[8] = m[8];
}
}  // namespace

void camera_setup(const mi_camera& c, float aspect, mi_camera_frame& out) {
  V3 eye{c.position[0], c.position[1], c.position[2]};
  V3 dir{c.direction[0], c.direction[1], c.direction[2]};
  V3 up{c.up[0], c.up[1], c.up[2]};
  // glm::lookAt (right handed), Cameras.cpp:99-102
  V3 f = normalize(sub(add(eye, dir), eye));
  V3 s = normalize(cross(f, up));
  V3 u = cross(s, f);
  float view3[9] = {s.x, u.x, -f.x, s.y, u.y, -f.y, s.z, u.z, -f.z};
  float inv[9], w2v[9], v2w[9];
  inverse3(view3, inv);
  transpose3(inv, w2v);  // Cameras.cpp:108-110
  inverse3(w2v, v2w);    // Cameras.cpp:104-106
  std::memcpy(out.world_to_view, w2v, sizeof w2v);
  std::memcpy(out.view_to_world, v2w, sizeof v2w);
  out.position[0] = eye.x; out.position[1] = eye.y; out.position[2] = eye.z;
  float focal = 1.0f / std::tan(c.fovx * 0.5f);                 // Cameras.cpp:23-25
  out.fovy = 2.0f * std::atan2(1.0f / aspect, focal);           // Cameras.cpp:85
  out.focal_length_y = 1.0f / std::tan(out.fovy * 0.5f);        // Cameras.cpp:116
}

}  // namespace mi

// ------------------------------------------------------------------------- C ABI

namespace mi {
namespace {
// A small persistent pool for the per-frame host work of the frame cadence (the dvec4 add of 16 B per pixel): at 512 x 512 the
// add is memory-bound at ~0.27 ms on one core, as long as the frame's kernel.  Workers sleep on a condition variable between frames.
class RowPool {
 public:
  RowPool() {
    unsigned n = std::thread::hardware_concurrency();
    if (const char* e = std::getenv("MI_PT_HOST_THREADS")) { const int v = std::atoi(e); if (v > 0) n = unsigned(v); }
    if (n > 8) n = 8;
    if (n < 1) n = 1;
    for (unsigned i = 1; i < n; ++i) workers_.emplace_back([this] { loop(); });
  }
  ~RowPool() {
    { std::lock_guard<std::mutex> g(m_); stop_ = true; ++generation_; }
    cv_.notify_all();
    for (std::thread& t : workers_) t.join();
  }
  // fn(row) for every row in [0, rows), chunks of rows handed out by an atomic counter; the caller works too
  template <class F> void run(uint32_t rows, F fn) {
    if (workers_.empty() || rows < 64) { for (uint32_t r = 0; r < rows; ++r) fn(r); return; }
    // one job at a time: job_ / rows_ / next_ / busy_ describe THE job.  A second caller (another thread adding another handle's frame — ctypes
    // releases the GIL, the adapter may run one thread per GPU) does its rows itself instead of waiting for the pool.
    std::unique_lock<std::mutex> owner(run_m_, std::try_to_lock);
    if (!owner.owns_lock()) { for (uint32_t r = 0; r < rows; ++r) fn(r); return; }
    const std::function<void(uint32_t)> f = fn;
    {
      std::lock_guard<std::mutex> g(m_);
      job_ = &f; rows_ = rows; next_.store(0); busy_ = unsigned(workers_.size()); ++generation_;
    }
    cv_.notify_all();
    work(f, rows);
    std::unique_lock<std::mutex> g(m_);
    done_.wait(g, [this] { return busy_ == 0; });
    job_ = nullptr;
  }

 private:
  void work(const std::function<void(uint32_t)>& f, uint32_t rows) {
    for (;;) {
      const uint32_t r0 = next_.fetch_add(16);
      if (r0 >= rows) return;
      const uint32_t r1 = r0 + 16 < rows ? r0 + 16 : rows;
      for (uint32_t r = r0; r < r1; ++r) f(r);
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(uint32_t)>* f; uint32_t rows;
      {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [&] { return generation_ != seen; });
        seen = generation_;
        if (stop_) return;
        f = job_; rows = rows_;
      }
      if (f) work(*f, rows);
      { std::lock_guard<std::mutex> g(m_); if (--busy_ == 0) done_.notify_one(); }
    }
  }
  std::vector<std::thread> workers_;
  std::mutex m_, run_m_;  // run_m_: held by the caller whose job the pool is running
  std::condition_variable cv_, done_;
  const std::function<void(uint32_t)>* job_ = nullptr;
  uint32_t rows_ = 0;
  std::atomic<uint32_t> next_{0};
  unsigned busy_ = 0;
  uint64_t generation_ = 0;
  bool stop_ = false;
};
}  // namespace

void add_frame_to_view(const float* rgbn, double* view, uint32_t width, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h) {
  static RowPool pool;
  pool.run(h, [=](uint32_t r) {
    const size_t o = (size_t(y0 + r) * width + x0) * 4;
    const float* s = rgbn + o;
    double* d = view + o;
    for (size_t i = 0; i < size_t(w) * 4; ++i) d[i] += double(s[i]);
  });
}
}  // namespace mi

extern "C" {

const char* mi_pt_last_error(void) { return mi::g_last_error.c_str(); }
int mi_pt_abi_version(void) { return MI_PT_ABI_VERSION; }
#ifndef MI_PT_BUILD_ID_STRING
#define MI_PT_BUILD_ID_STRING "MI_PT_BUILD_ID=unknown"
#endif
// "MI_PT_BUILD_ID=<hash>": the tag is in the binary so that master_amd/build.py can read the id without loading the library
const char* mi_pt_build_id(void) { static const char id[] = MI_PT_BUILD_ID_STRING; return id + 15; }

static int finish_scene(mi_scene* s, mi_scene** out) {
  std::string err = s->data.validate();
  if (!err.empty()) {
    delete s;
    return mi::fail(MI_ERR_INVALID_ARGUMENT, "Invalid scene: " + err);
  }
  s->cached_desc = s->data.desc();
  *out = s;
  return MI_OK;
}

int mi_scene_load_blend(const char* path, const mi_blend_options* opts, mi_scene** out) {
  if (!path || !out) return mi::fail(MI_ERR_INVALID_ARGUMENT, "mi_scene_load_blend: null argument");
  mi_scene* s = new mi_scene();
  int rc = mi::load_blend(path, opts, s->data);
  if (rc != MI_OK) { delete s; return rc; }
  return finish_scene(s, out);
}
int mi_scene_load(const char* path, mi_scene** out) {
  if (!path || !out) return mi::fail(MI_ERR_INVALID_ARGUMENT, "mi_scene_load: null argument");
  mi_scene* s = new mi_scene();
  int rc = mi::load_miscene(path, s->data);
  if (rc != MI_OK) { delete s; return rc; }
  return finish_scene(s, out);
}
int mi_scene_save(const mi_scene* scene, const char* path) {
  if (!scene || !path) return mi::fail(MI_ERR_INVALID_ARGUMENT, "mi_scene_save: null argument");
  return mi::save_miscene(scene->data, path);
}
int mi_scene_from_desc(const mi_scene_desc* desc, mi_scene** out) {
  if (!desc || !out) return mi::fail(MI_ERR_INVALID_ARGUMENT, "mi_scene_from_desc: null argument");
  if ((desc->n_vertices && (!desc->positions || !desc->tangents)) || (desc->n_triangles && !desc->indices) ||
      !desc->mesh_tri_offset || (desc->n_meshes && !desc->mesh_material_id) || (desc->n_materials && !desc->materials) ||
      (desc->n_lights && !desc->lights) || (desc->n_cameras && !desc->cameras))
    return mi::fail(MI_ERR_INVALID_ARGUMENT, "mi_scene_from_desc: null array with non-zero count");
  mi_scene* s = new mi_scene();
  s->data = mi::SceneData::from_desc(*desc);
  return finish_scene(s, out);
}
const mi_scene_desc* mi_scene_get_desc(const mi_scene* scene) { return scene ? &scene->cached_desc : nullptr; }
const char* mi_scene_material_name(const mi_scene* scene, uint32_t i) {
  return (scene && i < scene->data.material_names.size()) ? scene->data.material_names[i].c_str() : "";
}
const char* mi_scene_mesh_name(const mi_scene* scene, uint32_t i) {
  return (scene && i < scene->data.mesh_names.size()) ? scene->data.mesh_names[i].c_str() : "";
}
void mi_scene_free(mi_scene* scene) { delete scene; }
void mi_free(void* p) { std::free(p); }

int mi_camera_setup(const mi_camera* cam, float aspect, mi_camera_frame* out) {
  if (!cam || !out || !(aspect > 0.0f)) return mi::fail(MI_ERR_INVALID_ARGUMENT, "mi_camera_setup: bad argument");
  mi::camera_setup(*cam, aspect, *out);
  return MI_OK;
}
// ray_direction, Cameras.cpp:120-127
void mi_camera_ray_direction(float px, float py, float res_x, float res_y, float fl, float out_dir[3]) {
  float ry_inv = 1.0f / res_y;
  float x = px * ry_inv * 2.0f - res_x * ry_inv;
  float y = py * ry_inv * 2.0f - 1.0f;
  float s = 1.0f / std::sqrt(x * x + y * y + fl * fl);
  out_dir[0] = x * s; out_dir[1] = y * s; out_dir[2] = -fl * s;
}
// pixel_position, Cameras.cpp:134-144
void mi_camera_pixel_position(const float dir[3], float res_x, float res_y, float fl, float out_xy[2]) {
  float ry_inv = 1.0f / res_y;
  float factor = fl / -dir[2];
  float x = dir[0] * factor, y = dir[1] * factor;
  out_xy[1] = (y + 1.0f) * res_y * 0.5f;
  out_xy[0] = (x + res_x * ry_inv) * res_y * 0.5f;
}

// rms_abs_errors, ImageView.cpp:60-85 (float accumulators, like the reference)
int mi_view_add_frame(const float* rgbn, double* view, uint32_t width, uint32_t height, mi_window win) {
  if (!rgbn || !view) return mi::fail(MI_ERR_INVALID_ARGUMENT, "mi_view_add_frame: null argument");
  if (win.w == 0 || win.h == 0) { win.x0 = 0; win.y0 = 0; win.w = width; win.h = height; }
  if (uint64_t(win.x0) + win.w > width || uint64_t(win.y0) + win.h > height) return mi::fail(MI_ERR_INVALID_ARGUMENT, "window exceeds the image");
  mi::add_frame_to_view(rgbn, view, width, win.x0, win.y0, win.w, win.h);
  return MI_OK;
}

int mi_rms_abs_errors(const float* rgbn, const float* ref, uint32_t w, uint32_t h, float* rms, float* abs_err) {
  if (!rgbn || !ref || !rms || !abs_err) return mi::fail(MI_ERR_INVALID_ARGUMENT, "mi_rms_abs_errors: null argument");
  float r = 0.0f, a = 0.0f;
  for (size_t i = 0; i < size_t(w) * h; ++i) {
    double ww = rgbn[4 * i + 3];
    float d0 = std::fabs(float(rgbn[4 * i] / ww) - ref[3 * i]);
    float d1 = std::fabs(float(rgbn[4 * i + 1] / ww) - ref[3 * i + 1]);
    float d2 = std::fabs(float(rgbn[4 * i + 2] / ww) - ref[3 * i + 2]);
    a += d0 + d1 + d2;
    r += d0 * d0 + d1 * d1 + d2 * d2;
  }
  float num = float(size_t(w) * h * 3);
  *rms = std::sqrt(r / num);
  *abs_err = a / num;
  return MI_OK;
}

int mi_rms_abs_errors_view(const double* view, const float* ref, uint32_t w, uint32_t h, float* rms, float* abs_err) {
  if (!view || !ref || !rms || !abs_err) return mi::fail(MI_ERR_INVALID_ARGUMENT, "mi_rms_abs_errors_view: null argument");
  float r = 0.0f, a = 0.0f;  // float accumulators in pixel order, as the reference's loop (ImageView.cpp:72-78)
  for (size_t i = 0; i < size_t(w) * h; ++i) {
    const double ww = view[4 * i + 3];
    const float d0 = std::fabs(float(view[4 * i] / ww) - ref[3 * i]);
    const float d1 = std::fabs(float(view[4 * i + 1] / ww) - ref[3 * i + 1]);
    const float d2 = std::fabs(float(view[4 * i + 2] / ww) - ref[3 * i + 2]);
    a += d0 + d1 + d2;
    r += d0 * d0 + d1 * d1 + d2 * d2;
  }
  const float num = float(size_t(w) * h * 3);
  *rms = std::sqrt(r / num);
  *abs_err = a / num;
  return MI_OK;
}

}  // extern "C"
