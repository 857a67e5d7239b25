// bpt.h — state of the bidirectional path tracer (bpt_kernels.hip), shared with the host API.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "layout.h"

namespace mi {

constexpr uint32_t kBptStepF4 = 16;  // float4 per path of BptState::step_state

struct BptState {
  uint32_t lanes;          // paths of this launch (one lane each)
  uint32_t first;          // image mode: first lane's index in tile order; list mode: first list item
  uint32_t frame;          // image mode: sample index of the first frame of this batch (added to RenderParams::sample_offset)
  uint32_t frames;         // image mode: frames in this batch (each with its own eye / light image)
  uint32_t max_vertices;   // capacity of a lane's light sub-path (BPT.hpp:30 allows 1024)
  const uint32_t* path_ids;  // image mode: lane i holds path path_ids[i] of the batch instead of first + i (the launch of the paths that outgrew their slab share)
  uint32_t* over_ids;      // image mode, r04: a path whose sub-path outgrows max_vertices is set aside — its index (first + i) is appended here, it contributes nothing
  uint32_t* over_count;    // in this launch — and is traced again at the reference's capacity when the batch's launches are done.  nullptr: the launch counts
                           // the overflow (counters[15]) and the host redoes it in slices
  uint32_t persist_hint;   // tracing stage of a scene read from HBM below 16 384 triangles: 0 / 1 = one lane per path, 2 = path regeneration (the host has seen >= 6.5 closest-hit rays per path)
  uint32_t async_total;    // 1: bpt_stage_trace leaves the item count's copy to the (pinned) host word in flight instead of waiting for it (launches overlapped on several streams)
  float4* slab;            // one-kernel form: [max_vertices][7][lanes] light sub-path vertices
  // staged form: path-major records (7 float4 per vertex), emission terms, per-path info, item offsets and values
  float4* lslab; float4* eslab; float4* nslab;   // [lanes][max_vertices][7]: light vertices, eye vertices, NEE samples of the eye vertices
  float4* emission;        // [lanes][max_vertices]: emission terms of the eye sub-path (rgb | index of the eye vertex they follow)
  uint2* evinfo;           // [lanes][max_vertices]: (NEE kind, first item) of every eye vertex — what bpt_gather and the item decode need of an eye vertex, 8 B
                           // instead of two words in two 128-byte lines of its 112-byte record (r02: 0.9 GB -> 28 MB per launch on the Cornell box)
  uint4* info;             // [lanes][2]: (L, E, items, emission terms), (closest-hit rays of the tracing stage, directional NEE, py<<16|px, frame<<1|ok)
  uint32_t* item_offset;   // [lanes + 1]: first connection item of every path (exclusive scan of the counts)
  uint32_t* scan_tmp;      // [ceil((lanes + 1) / 2048)]: tile totals of that scan
  float4* values;          // [items]: value of a connection | flags
  // visibility stage (r02): the shadow rays of a launch's items as a compacted list, walked by persistent waves that refill idle lanes from it
  float4* rays;            // [items][2]: (origin | 1 if the item casts a shadow ray), (direction | 0); t in (0, 1], mesh geometry only
  uint8_t* occl;           // [items]: 1 = the item's shadow ray is blocked (written for items that cast one)
  uint32_t* pool;          // [64] chunk cursors of the walking waves (bpt_visibility)
  uint32_t dyn_vis;        // 1: bpt_items reads `occl` instead of walking the item's ray itself
  uint32_t vis_th;         // idle lanes of a walking wave that trigger a refill from the ray list
  uint32_t vis_wide;       // node records of the visibility walk: 1 = 64-byte wide quantised, 0 = 32-byte binary quantised (scenes read from HBM)
  // tracing stage as uniform steps (r04, scenes read from HBM): a path's coroutine state between two closest-hit rays, its pending ray and that ray's hit,
  // the lists of paths that have a ray in flight (ping-pong) with their counts
  float4* step_state;      // [lanes][kBptStepF4]
  float4* step_rays;       // [lanes][2]: (origin | geometry mask), (direction | -)
  float4* step_hits;       // [lanes]: (t, u, v, Morton position or 0xFFFFFFFF = miss)
  uint32_t* step_active[2];  // [lanes] path indices
  uint32_t* step_count;    // [2] entries of step_active[k]
  float* eye;              // [frames][H][W][3] eye images (Technique::_eye_image)
  double* light;           // [frames][H][W][3] light images (Technique::_light_image)
  float sphere[4];         // scene bounding sphere (loader.cpp:408-432) for the emitters' bounded cosine sampling
  float w2v[9];            // world_to_view_mat3 (Technique.cpp:40)
  float sky_horizon[3], sky_zenith[3];  // Technique::set_sky_gradient: what a camera ray that leaves the scene returns (BPT.cpp:49-51)
  float* list_splat_sum; uint32_t* list_counts3;  // list mode outputs (mi_bpt_trace_paths)
};

// launchers of one compiled feature set (bpt_kernels.hip is built once per set)
#define MI_BPT_DECLARE(ns)                                                                                                                        \
  namespace ns {                                                                                                                                  \
  hipError_t bpt_launch_frame(const RenderParams& p, const BptState& w, bool list, hipStream_t stream);                                            \
  hipError_t bpt_stage_trace(const RenderParams& p, const BptState& w, bool list, bool lds_scene, hipStream_t stream, uint32_t* total_items);      \
  hipError_t bpt_stage_trace_steps(const RenderParams& p, const BptState& w, bool list, hipStream_t stream, uint32_t* total_items, uint32_t* rounds); \
  hipError_t bpt_stage_trace_passes(const RenderParams& p, const BptState& w, bool list, hipStream_t stream, uint32_t* total_items, uint32_t* rounds); \
  hipError_t bpt_stage_connect(const RenderParams& p, const BptState& w, bool list, bool lds_scene, uint32_t total_items, hipStream_t stream);     \
  hipError_t bpt_launch_commit(const RenderParams& p, const BptState& w, hipStream_t stream);                                                      \
  }
MI_BPT_DECLARE(bpt_all)    // every BSDF, any beta
MI_BPT_DECLARE(bpt_fixed)  // every BSDF, beta in {0, 1, 2}
MI_BPT_DECLARE(bpt_plain)  // no Phong lobes, no mirrors / glass, beta in {0, 1, 2}
#undef MI_BPT_DECLARE

struct BptLaunchers {
  hipError_t (*frame)(const RenderParams&, const BptState&, bool, hipStream_t);
  hipError_t (*trace)(const RenderParams&, const BptState&, bool, bool, hipStream_t, uint32_t*);
  hipError_t (*trace_steps)(const RenderParams&, const BptState&, bool, hipStream_t, uint32_t*, uint32_t*);
  hipError_t (*trace_passes)(const RenderParams&, const BptState&, bool, hipStream_t, uint32_t*, uint32_t*);
  hipError_t (*connect)(const RenderParams&, const BptState&, bool, bool, uint32_t, hipStream_t);
  hipError_t (*commit)(const RenderParams&, const BptState&, hipStream_t);
};
// the set compiled for `features` (RenderParams::features, kFeat* bits; for BPT kFeatPow means beta not in {0, 1, 2})
inline BptLaunchers bpt_launchers(uint32_t features) {
  features &= 7u;  // the BPT kernels are not specialised on the number of lights
  if (features == 0u) return {bpt_plain::bpt_launch_frame, bpt_plain::bpt_stage_trace, bpt_plain::bpt_stage_trace_steps, bpt_plain::bpt_stage_trace_passes, bpt_plain::bpt_stage_connect, bpt_plain::bpt_launch_commit};
  if ((features & 4u) == 0u) return {bpt_fixed::bpt_launch_frame, bpt_fixed::bpt_stage_trace, bpt_fixed::bpt_stage_trace_steps, bpt_fixed::bpt_stage_trace_passes, bpt_fixed::bpt_stage_connect, bpt_fixed::bpt_launch_commit};
  return {bpt_all::bpt_launch_frame, bpt_all::bpt_stage_trace, bpt_all::bpt_stage_trace_steps, bpt_all::bpt_stage_trace_passes, bpt_all::bpt_stage_connect, bpt_all::bpt_launch_commit};
}

}  // namespace mi
