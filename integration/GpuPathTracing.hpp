// GpuPathTracing.hpp — drop-in `haste::Technique` that forwards the PT path to libmi_pt.so.
// Add next to PT.hpp in the reference tree; needs only <mi_pt.h> besides the reference's headers and the
// accessor patch to BSDF.hpp given in INTEGRATION.md (the BSDF parameters are private there, BSDF.hpp:113-170).
// (Not compiled in this repository: the reference's glm / Embree headers are absent here;
// tests/test_integration_adapter.py checks every reference member used below against the reference's headers.)
#pragma once
#include <Technique.hpp>
#include <mi_pt.h>

#include <deque>

namespace haste {

class GpuPathTracing : public Technique {
 public:
  // same arguments as PathTracing (PT.cpp:5-13); num_threads is ignored, `device` selects the GPU; device = -1 uses every
  // visible GPU (mi_pt_render_multi deals the 32x32 tiles of _trace_paths to them; PT only)
  // `bidirectional` = the BPT0/1/2/b techniques (make_technique.cpp:112-130) through mi_bpt_render: beta 0 / 1 / 2 / other
  GpuPathTracing(const shared<const Scene>& scene, float lights, float roulette, float beta,
                 size_t max_path, size_t num_threads, int device = 0, bool bidirectional = false);
  ~GpuPathTracing() override;

  // Technique::render replaced wholesale (precedent: Viewer::render, Viewer.cpp:14-23)
  void render(subimage_view_t& view, RandomEngine& engine, size_t cameraId,
              const vector<vec3>& reference, const vector<ivec3>& trace_points) override;

 private:
  // frames in flight: Application::render asks for ONE sample per call (Application.cpp:66); the frames of the next calls are
  // rendered ahead in batches (mi_pt_render_frames_async) and handed out in order
  struct Frame { uint64_t ticket; size_t sample_index; };
  struct FrameKey {
    size_t camera = size_t(-1), width = 0, height = 0, x0 = 0, y0 = 0, w = 0, h = 0;
    bool operator==(const FrameKey& o) const { return camera == o.camera && width == o.width && height == o.height && x0 == o.x0 && y0 == o.y0 && w == o.w && h == o.h; }
  };
  void _drain();
  void _render_ahead(const FrameKey& key, mi_window win);

  mi_pt_handle* _handle = nullptr;          // first device
  std::vector<mi_pt_handle*> _handles;      // all devices (PT)
  std::vector<float> _rgbn;                 // synchronous paths (BPT, several devices)
  std::deque<Frame> _frames;                // enqueued, not yet added to the view
  FrameKey _key;
  size_t _next_sample = 0;                  // sample index of the next frame to enqueue
  uint64_t _seed;
  bool _bidirectional = false;
};

}  // namespace haste
