// GpuPathTracing.hpp — drop-in `haste::Technique` that forwards the PT path to libmi_pt.so.
// Add next to PT.hpp in the reference tree; needs only <mi_pt.h> besides the reference's headers.
// (Not compiled in this repository: the reference's glm / Embree headers are absent here.)
#pragma once
#include <Technique.hpp>
#include <mi_pt.h>

namespace haste {

class GpuPathTracing : public Technique {
 public:
  // same arguments as PathTracing (PT.cpp:5-13); num_threads is ignored, `device` selects the GPU; devices = -1 uses every
  // visible GPU (mi_pt_render_multi deals the 32x32 tiles of _trace_paths to them; PT only)
  // `bidirectional` = the BPT0/1/2/b techniques (make_technique.cpp:112-130) through mi_bpt_render: beta 0 / 1 / 2 / other
  GpuPathTracing(const shared<const Scene>& scene, float lights, float roulette, float beta,
                 size_t max_path, size_t num_threads, int device = 0, bool bidirectional = false);
  ~GpuPathTracing() override;

  // Technique::render replaced wholesale (precedent: Viewer::render, Viewer.cpp:14-23)
  void render(subimage_view_t& view, RandomEngine& engine, size_t cameraId,
              const vector<vec3>& reference, const vector<ivec3>& trace_points) override;

 private:
  mi_pt_handle* _handle = nullptr;          // first device
  std::vector<mi_pt_handle*> _handles;      // all devices (PT)
  std::vector<float> _rgbn;
  uint64_t _seed;
  bool _bidirectional = false;
};

}  // namespace haste
