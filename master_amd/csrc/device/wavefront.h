// wavefront.h — state of the wavefront pipeline (wf_kernels.hip), shared with the host API.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "layout.h"

#ifndef MI_WF_QUANT
#define MI_WF_QUANT 1  // traversal kernels: 1 = 32-byte quantised binary nodes, 2 = 64-byte wide nodes
#endif

namespace mi {

struct WfState {
  uint32_t P;                  // path slots = per_sample * R (image mode)
  uint32_t per_sample;         // pixel slots: tiles * 64 (image mode); P in list mode
  uint32_t R;                  // sample replicas in flight per pixel
  uint32_t list;               // 1: work items come from RenderParams::list_* (mi_pt_trace_paths), slot i takes items i + j P
  uint64_t n_items;            // list mode: number of items
  // per slot (SoA, thread == slot: coalesced)
  uint64_t* rng;               // PCG state
  float4* ray_o; float4* ray_d;  // next closest-hit ray: origin | slot flags, direction | paths started by this slot
  float4* st_a;                // eye[prv].position | bsdf.density
  float4* st_b;                // throughput numerator (PT.cpp:59-60) | flags: bounce, bsdf.finite, path_size
  float4* st_c;                // radiance of the path under way
  uint2* cnt;                  // rays cast by this path (closest, shadow) — reported by mi_pt_trace_paths
  float4* hit;                 // t, u, v | Morton position of the triangle (0xFFFFFFFF = miss)
  float4* sh_o; float4* sh_d; float* sh_z;  // shadow ray origin | nee.r, direction (to the target) | nee.g, nee.b
  double4* acc;                // image mode: the slot's FP64 sum over its finished paths (r, g, b, count)
  unsigned long long* n_active;  // slots still active after wf_shade (read by the host every fourth round)
};

constexpr size_t kWfBytesPerSlot = 8 + 16 * 8 + 8 + 4 + 32;  // rng, 8 float4 arrays, cnt, sh_z, acc

hipError_t wf_run(const RenderParams& p, const WfState& w, bool count, hipStream_t stream, uint32_t* rounds_out);

}  // namespace mi
