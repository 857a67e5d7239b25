// wavefront.h — state of the wavefront pipeline (wf_kernels.hip), shared with the host API.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "layout.h"

#ifndef MI_WF_QUANT
#define MI_WF_QUANT 1  // traversal kernels walk the 32-byte quantised nodes
#endif

namespace mi {

struct WfState {
  uint32_t P;                  // path slots in flight
  uint32_t list;               // 1: work items come from RenderParams::list_* (mi_pt_trace_paths)
  uint64_t n_items;            // work items of this batch
  uint64_t batch_sample0;      // first sample index of this batch (added to RenderParams::sample_offset)
  // per slot (SoA)
  uint64_t* rng;               // PCG state
  float4* ray_o; float4* ray_d;  // ray origin / direction of the next closest-hit cast
  float4* st_a;                // eye[prv].position | bsdf.density
  float4* st_b;                // throughput numerator (PT.cpp:59-60) | flags: bounce, bsdf.finite, path_size
  float4* st_c;                // radiance | work item
  uint2* cnt;                  // rays cast by this path (closest, shadow) — reported by mi_pt_trace_paths
  float4* hit;                 // t, u, v | Morton position of the triangle (0xFFFFFFFF = miss)
  float4* sh_o; float4* sh_d; float* sh_z;  // shadow ray origin | nee.r, direction (to the target) | nee.g, nee.b
  // queues of slot indices
  uint32_t* qc[2];             // closest-hit queue, ping-pong
  uint32_t* qs;                // shadow queue
  uint32_t* qf;                // finished paths
  uint32_t* n;                 // [4]: sizes of qc[0], qc[1], qs, qf
  unsigned long long* work_next;  // next unclaimed work item
  float4* results;             // [n_items] per-path radiance | valid (image mode)
};

constexpr size_t kWfBytesPerSlot = 8 + 16 * 8 + 8 + 4 + 4 * 4;  // rng, 8 float4 arrays, cnt, sh_z, 4 queue entries

hipError_t wf_run_batch(const RenderParams& p, const WfState& w, bool count, uint32_t batch_spp, hipStream_t stream, uint32_t* iterations_out);

}  // namespace mi
