// launch.h — host-callable entry points of the device translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../../include/mi_pt.h"
#include "layout.h"

namespace mi {

hipError_t build_bvh(int builder, uint32_t nt, const float* pos, const float* tan, const uint32_t* idx, const uint32_t* tri_material,
                     mi_bvh_node* nodes, float4* tri_isect, float4* tri_shade, uint32_t* sorted_tri, uint64_t* morton,
                     float scene_lo[3], float scene_hi[3], uint32_t* max_depth_out, float* build_ms, uint32_t* rounds_out, int2* plain_links, bool pairs,
                     hipStream_t stream);

// pad_cells: padding of every quantised child box in grid cells (1; more for cameras far outside the grid).  area_collapse: wide records by the area-guided
// collapse (r04) instead of two BVH2 levels per record; wide_stack_need (host): the stack entries the wide walk can need at most
hipError_t quantize_nodes(uint32_t n_nodes, const mi_bvh_node* nodes, uint4* qnodes, uint4* qnodes4, const float lo[3], const float inv_step[3],
                          uint32_t pad_cells, uint32_t depth, bool area_collapse, uint32_t* wide_stack_need, hipStream_t stream);

hipError_t ce_nodes(uint32_t n_nodes, const float4* nodes, float4* out, hipStream_t stream);  // centre / half-extent copy of the full-precision nodes

namespace fastmath {  // the megakernel built with -DMI_PT_FAST: approximate reciprocal / square root / sin / cos (vecmath.h); opt-in, never the default
size_t pt_lds_bytes(const RenderParams& p, bool lds_scene);
hipError_t launch_megakernel(const RenderParams& p, bool lds_scene, int mode, bool count, uint32_t n_blocks, hipStream_t stream);
}
size_t pt_lds_bytes(const RenderParams& p, bool lds_scene);
hipError_t launch_megakernel(const RenderParams& p, bool lds_scene, int mode, bool count, uint32_t n_blocks, hipStream_t stream);  // mode: 0 image, 1 list, 2 frame (one sample per pixel)
hipError_t launch_finalize(const double* partial, float* rgbn, uint32_t width, uint32_t height, uint32_t x0, uint32_t y0, uint32_t w,
                           uint32_t h, uint32_t n_chunks, hipStream_t stream);
hipError_t launch_intersect(const SceneView& sv, bool wide, uint32_t stack_entries, uint32_t n, const mi_surface_point* origins, const float* dirs,
                            mi_surface_point* out_hits, float* out_t, uint32_t* out_prim, hipStream_t stream);
hipError_t launch_occluded(const SceneView& sv, bool wide, uint32_t stack_entries, uint32_t n, const mi_surface_point* a, const mi_surface_point* b,
                           float* out, hipStream_t stream);

// the parity hooks through the flat leaf list (scenes with a leaf table; MI_PT_INTERSECT_FLAT=1)
hipError_t launch_intersect_flat(const SceneView& sv, const float* table, uint32_t K, uint32_t n, const mi_surface_point* origins, const float* dirs,
                                 mi_surface_point* out_hits, float* out_t, uint32_t* out_prim, hipStream_t stream);
hipError_t launch_occluded_flat(const SceneView& sv, const float* table, uint32_t K, uint32_t k_mesh, uint32_t n, const mi_surface_point* a, const mi_surface_point* b,
                                float* out, hipStream_t stream);

}  // namespace mi
