// scene_host.hpp — owning container behind `mi_scene` (host side, no GPU code).
// Mirrors the data haste::Scene owns (Scene.hpp:27-43): meshes, materials, lights, cameras.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/mi_pt.h"

namespace mi {

struct SceneData {
  std::vector<float> positions;          // [n_vertices][3]
  std::vector<float> tangents;           // [n_vertices][9]
  std::vector<uint32_t> indices;         // [n_triangles][3]
  std::vector<uint32_t> mesh_tri_offset; // [n_meshes+1]
  std::vector<uint32_t> mesh_material_id;
  std::vector<mi_material> materials;
  std::vector<mi_light> lights;
  std::vector<mi_camera> cameras;
  std::vector<std::string> material_names;
  std::vector<std::string> mesh_names;
  float bounding_sphere[4] = {0, 0, 0, 0};

  mi_scene_desc desc() const;
  // runtime_assert analogue: returns "" when the description is well-formed.
  std::string validate() const;
  static SceneData from_desc(const mi_scene_desc& d);
};

void set_last_error(const std::string& msg);
int fail(int code, const std::string& msg);

int load_blend(const char* path, const mi_blend_options* opts, SceneData& out);  // blend_reader.cpp
int save_miscene(const SceneData& s, const char* path);
int load_miscene(const char* path, SceneData& s);
void camera_setup(const mi_camera& c, float aspect, mi_camera_frame& out);
// Technique::_commit_images for the PT path (Technique.cpp:215-236): view[p] += dvec4(rgbn[p]) over the window, rows dealt to the
// library's host threads (the reference deals its frame to a thread pool too, threadpool.cpp:190-233)
void add_frame_to_view(const float* rgbn, double* view, uint32_t width, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h);

}  // namespace mi

struct mi_scene {
  mi::SceneData data;
  mi_scene_desc cached_desc;
};
