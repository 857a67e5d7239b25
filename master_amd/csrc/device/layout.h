// layout.h — device-side data layout shared by the host API and the kernels.
//
// One contiguous "scene blob" of float4 records lives in HBM (and, for scenes that fit, is
// staged whole into LDS by every workgroup):
//   nodes      n_nodes  x 4 float4   BVH2 node = two child boxes + two links (mi_bvh_node, 64 B)
//   tri_isect  n_tris   x 3 float4   Morton order: v0, e1 = v0-v1, e2 = v2-v0, id, mask (48 B)
//   tri_shade  n_tris   x 8 float4   Morton order: 3 vertex frames (3 x mat3), material_id, unit geometric normal (128 B)
//   materials  n_mats   x 3 float4   mi_material with `reserved` = Phong diffuse probability
//   lights     n_lights x 6 float4   DevLight
//   light_cdf  ceil((n_lights+1)/4) float4
// Traversal only ever touches nodes + tri_isect; shading data is a separate stream so it
// does not compete for cache lines with the intersection data.
#pragma once
#include <stdint.h>

namespace mi {

constexpr int kBlock = 256;  // threads per workgroup of the path kernels; measured: 128 -> -16 % on C2, +-0 on HBM-resident scenes; 512 -> -7 % on C2, -4 ... -10 % elsewhere
constexpr int kWavesPerBlock = kBlock / 64;


// Leaf links.  A negative link is a leaf: ~link = position in the intersection / shading streams, with kLeafPairBit set when the triangle
// at position + 1 is to be tested under the same box as well (pair leaves, bvh_build.hip).
constexpr uint32_t kLeafPairBit = 0x40000000u, kLeafPosMask = 0x3FFFFFFFu;

// Flat leaf list (traverse_flat, pt_device.h): at most kFlatMaxLeaves leaf links (one mask bit each); an LDS leaf record is kFlatLeafF4 float4.
constexpr uint32_t kFlatMaxLeaves = 32u, kFlatLeafF4 = 7u;

struct SceneView {
  const float4* blob;  // HBM
  uint32_t off_nodes, off_tris, off_shade, off_mats, off_lights, off_cdf;  // in float4 units
  uint32_t blob_f4;                                                        // total float4 count
  uint32_t n_tris, n_nodes, n_mats, n_lights;
  // quantised copy of the BVH2 nodes for HBM-resident scenes: 32 B per node (two child boxes as 12 x u16 on a
  // 65536^3 grid over the scene box, rounded outward by one cell, + two links) — half the bytes per visited node
  const uint4* qnodes;
  // wide quantised nodes, indexed like the BVH2 nodes (only even-depth entries are valid): 4 x (child box as 6 x u16 + link),
  // the grandchildren of the BVH2 node — 64 B per visited node, half as many dependent fetches per ray
  const uint4* qnodes4;
  // full-precision nodes as centre + half extent (scenes the 16-bit grid is too coarse for; same layout as the blob's nodes: 4 float4, links in n0.w / n1.w)
  const float4* ce_nodes;
  float grid_lo[3];
  float grid_inv_step[3];  // cells per world unit
  uint32_t dyn_uni;        // traverse_dyn: 1 = one fetch per loop iteration whatever the lane is at (node or triangle record into the same registers), 0 = a load in each of
                           // the two branches.  Measured (profiles/r04/ab_hbm_walk.txt): +3.5..6 % where the node records fit the L2 of an XCD (4 MB), -3..-10 % where they do not
  float box_pad;           // 2^-20 of the largest |coordinate| of the scene box and the cameras: absolute padding of centre / half-extent boxes (LDS node copy, flat leaf table)
};

struct DevLight {  // 6 float4
  float position[3]; float weight;       // AreaLights::_weights[i] (AreaLights.cpp:199-209)
  float t0[3]; float inv_cd;             // tangent[0]; 1 / (area_density * weight): the reciprocal _connect divides the radiance by (PT.cpp:117-119), formed once on the
                                         // host (same IEEE division).  In this slot because connect_prepare reads t0 anyway: a fourth word of l5 cost the 80-register
                                         // kernels of HBM-resident scenes four more spilled dwords (profiles/r04/ab_c2_instruction_cuts.txt)
  float t1[3]; float size_x;             // tangent[1] = emission normal
  float t2[3]; float size_y;             // tangent[2]
  float radiance[3]; uint32_t material_id;  // exitance / pi (AreaLights.hpp:54)
  float lsdf_density;                    // weight / area        (AreaLights.cpp:152)
  float area_density;                    // 1 / area             (AreaLights.cpp:135)
  uint32_t diffuse; uint32_t pad;
};

struct RenderParams {
  SceneView sv;
  // render_context_t (Technique.hpp:14-27)
  float v2w[9];
  float cam_pos[3];
  float focal_length_y;
  float res_x, res_y, res_y_inv;
  float res_x_res_y_inv;   // res_x * res_y_inv as one FP32 product (Cameras.cpp:123), formed on the host: a loop-invariant VALU result would cost a VGPR
  uint32_t width, height;
  uint32_t win_x0, win_y0, win_w, win_h;
  uint32_t tiles_x, tiles_y;
  uint32_t shard_rank, shard_world, shard_mtx;  // pixel-tile sharding (mi_pt_set_tile_shard): world > 1 = on, mtx = 32x32 tiles per window row
  uint32_t stack_entries;  // per-lane traversal stack capacity (LDS), >= BVH depth
  uint32_t stack_in_lds;   // the binary walk never needs more than stack_entries: kernels may drop the private spill path
  uint32_t wide_nodes;     // HBM-resident kernels: 0 = 32-byte quantised binary nodes, 1 = 64-byte quantised wide nodes with the 7-wave
                           // register budget (large scenes), 2 = full-precision 64-byte binary nodes (grid too coarse for the scene)
  uint32_t features;       // kFeat* bits the scene and parameters need (pt_device.h): selects the kernel variant compiled without the rest
  uint32_t dyn_traverse;   // LDS-resident kernels: closest-hit and shadow rays share one traversal loop with dynamic fetch (traverse_dyn)
  uint32_t lds_tables;     // HBM-resident kernels: materials, lights and the light CDF are staged into LDS by every workgroup (they fit kLdsTablesMaxF4)
  // LDS-resident kernels, flat leaf list (traverse_flat, pt_device.h): flat_k > 0 selects it.  flat_table: flat_k x { lo.xyz, hi.xyz, link, - } in global
  // memory, the flat_k_mesh leaves that hold a mesh triangle first
  const float* flat_table; uint32_t flat_k, flat_k_mesh;
  // sample range
  uint32_t spp, n_chunks, chunk_spp;
  uint64_t seed, sample_offset;
  // PathTracing members (PT.hpp:24-28)
  uint32_t max_path, min_subpath;
  float beta, roulette, lights;
  float inv_roulette;      // 1 / roulette as one IEEE division on the host: throughput / roulette = throughput * (1 / roulette) by the contract (PT.cpp:92)
  // outputs
  double* partial;        // [n_chunks][height*width][4] (r, g, b sums, count)
  unsigned long long* counters;  // [9]: basic rays, shadow rays, numeric errors, paths, + instrumented: nodes/tris visited by closest-hit rays, by shadow rays, closest-hit rays that hit
  // list mode (mi_pt_trace_paths)
  const uint32_t* list_xy; const uint64_t* list_sample; uint32_t list_n;
  float* list_radiance; uint32_t* list_counts;
  // frame mode (one sample per pixel): the FP32 framebuffer the paths write to, and the 8x8 tiles one wave regenerates over
  float4* frame_rgbn; uint32_t frame_tiles_per_wave;
  uint32_t frame_count;    // frames of the launch (consecutive samples sample_offset ..), each into its own framebuffer; <= kMaxFramesPerLaunch
  uint32_t frame_stride;   // float4 elements between the framebuffers of consecutive frames (width * height)
  uint32_t frame_chunk;    // consecutive frames of the launch one wave owns (of its tiles): pool of a wave = 64 * frame_tiles_per_wave * frame_chunk paths
};
constexpr uint32_t kMaxFramesPerLaunch = 16;  // per-wave frame counts live in 64 LDS words
constexpr uint32_t kCounterWords = 32 + 4 * kMaxFramesPerLaunch;  // counters buffer: totals / instrumentation [0, 32), per-frame counts [32 + 4 f ..]

// Origin of wave tile `tile` (8x8 pixels).  Unsharded: row-major over the window.  Sharded: the rank's k-th 32x32 tile
// (Technique.cpp:167) is tile k * world + rank of the window, and holds 16 wave tiles.
__device__ __forceinline__ void tile_origin(const RenderParams& p, uint32_t tile, uint32_t& x0, uint32_t& y0) {
  if (p.shard_world > 1u) {
    const uint32_t m = (tile >> 4) * p.shard_world + p.shard_rank, micro = tile & 15u;
    const uint32_t my = m / p.shard_mtx, mx = m - my * p.shard_mtx;
    x0 = p.win_x0 + mx * 32u + (micro & 3u) * 8u;
    y0 = p.win_y0 + my * 32u + (micro >> 2) * 8u;
  } else {
    const uint32_t ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    x0 = p.win_x0 + tx * 8u;
    y0 = p.win_y0 + ty * 8u;
  }
}

}  // namespace mi
