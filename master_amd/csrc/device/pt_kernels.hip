// pt_kernels.hip — the path-tracing megakernel for gfx950 and its satellites.
//
// pt_megakernel<LDS_SCENE, LIST>: one wave owns an 8x8 pixel tile x a chunk of samples =
// a pool of 64 * chunk_spp camera samples.  Every lane runs the PT state machine
// (PathTracing::_traceEye, PT.cpp:15-98) one path SEGMENT per loop trip: one closest-hit
// traversal, then — if the hit is a surface vertex — next-event estimation with its shadow
// ray, BSDF sampling and Russian roulette.  A lane whose path ends pulls the next sample of
// the pool at once (wave ballot + prefix count), so the wave stays full until the pool is
// drained; per-pixel sums live in LDS as FP64 (Technique.cpp:338) and leave the wave once.
//
//   LDS_SCENE  the whole scene blob (BVH nodes, triangles, frames, materials, lights) is
//              staged into LDS by the workgroup; otherwise it is read from HBM/L2.
//   LIST       mi_pt_trace_paths: samples come from an explicit (pixel, sample) list and
//              per-path radiance is written out — same state machine, same code.
//
// LDS: [scene blob (LDS_SCENE)] [traversal stacks: stack_entries x 256 dwords] [4 x per-wave
// pixel accumulators: 3 x 64 doubles + 64 counts].
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "pt_device.h"

// MI_PT_FAST (build.py compiles this file a second time with it): the megakernel with the hardware's approximate reciprocal / square root / sin / cos
// (vecmath.h) in namespace mi::fastmath; only launch_megakernel and pt_lds_bytes exist there.  Opt-in at run time (MI_PT_FAST=1), never the default.
#ifdef MI_PT_FAST
#define MI_PT_NS_BEGIN namespace mi { namespace fastmath {
#define MI_PT_NS_END } }
#else
#define MI_PT_NS_BEGIN namespace mi {
#define MI_PT_NS_END }
#endif
MI_PT_NS_BEGIN

MI_DEV uint32_t rank_in(uint64_t mask) {  // number of set bits of `mask` below this lane
  return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
}
MI_DEV uint32_t wave_max(uint32_t v) {
  for (int o = 32; o > 0; o >>= 1) { const uint32_t w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
  return v;
}
MI_DEV uint32_t wave_sum(uint32_t v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

#ifdef MI_PHASE_TIMING
#define MI_STAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); phase_t[k] += now_ - last_t; last_t = now_; } while (0)
#else
#define MI_STAMP(k) do { } while (0)
#endif

constexpr uint32_t kAccBytesPerWave = 3 * 64 * 8 + 64 * 4 + 32;  // r, g, b sums, counts, 8 instrumentation words

// Measured and removed (profiles/r01): the shadow ray of vertex k riding in the closest-hit loop of trip k+1 (-18 %, ab_fused.txt);
// the shadow ray traversed after the BSDF sample instead of inside the NEE step (-7 %, ab_launch_bounds.txt).
// Second __launch_bounds__ argument = minimum waves per SIMD the register budget must allow.  Measured on
// CornellBoxDiffuse (LDS-resident scene): 3 waves 6.1, 4 waves 7.1, 5 waves 7.7, 6 waves 8.1, 8 waves 6.7 Gsamples/s
// (profiles/r01/ab_launch_bounds.txt); the 6-wave build spills 136 B/lane to scratch and still wins on latency hiding.
// Kernels that read the scene from HBM walk quantised nodes: QN = 1 the 32-byte binary nodes, QN = 2 the 64-byte wide nodes
// (four grandchildren per record).  Wide nodes halve the dependent fetches per ray: +9 % on 270 k triangles, +5 % on 158 k,
// but -3..-12 % on scenes of 2-44 k triangles whose nodes sit in L2 (profiles/r01/ab_bvh4.txt) — chosen per scene
// (RenderParams::wide_nodes; default: scenes of >= 100 000 triangles, together with the 7-wave register budget).
#ifndef MI_WAVES_LDS
#define MI_WAVES_LDS 6
#endif
#ifndef MI_WAVES_FLAT
#define MI_WAVES_FLAT 6  // flat leaf list (LDS-resident scenes of <= kFlatMaxLeaves leaves)
#endif
#ifndef MI_DYN_W_HI
#define MI_DYN_W_HI 6  // waves per SIMD the dynamic-fetch variants are compiled for where that many workgroups' LDS fit a CU (else 5)
#endif
#ifndef MI_DYN_TH
#define MI_DYN_TH 8  // idle lanes that trigger a refill of the unified traversal loop
#endif
// dynamic-fetch variants of the HBM-resident kernels: the parked rays cost 6.5 KB of LDS per workgroup.  Where six workgroups still fit a CU
// (<= 27 306 B each) the 6-wave register budget wins (+2..5 %); where only five fit (LivingRoom: 3.2 KB of tables) the 5-wave budget
// (96 VGPRs, no spills) wins by 9 % (profiles/r02/ab_dynamic_fetch.txt)
#ifndef MI_WAVES_HBM
#define MI_WAVES_HBM 6  // HBM-resident scenes (profiles/r01/ab_launch_bounds.txt): 6 waves = 5 waves +-1 % on small scenes, +3..9 % on 150-270 k triangles
#endif
#ifndef MI_WAVES_HBM_LARGE
#define MI_WAVES_HBM_LARGE 7  // scenes of >= kLargeSceneTris triangles: latency-bound gathers want occupancy (atrium +12 %, clutter +4 % over 5 waves); -11 % on a 2 k-triangle scene
#endif

// MODE: 0 = image (tile x sample chunk per wave, per-pixel FP64 sums in LDS), 1 = list (mi_pt_trace_paths), 2 = frame: ONE sample per
// pixel, the cadence of the reference (Application.cpp:66: one Technique::render per sample).  A frame has nothing to accumulate, so
// a path's radiance goes straight into the FP32 framebuffer as (r, g, b, 1) — no LDS sums, no partial buffer, no pt_finalize — and a
// wave owns its 8x8 tiles (RenderParams::frame_tiles_per_wave) in ALL the frames of the launch (frame_count consecutive samples, each
// into its own framebuffer), so that dead lanes regenerate onto the same pixels of the next frame: with one frame of one tile per wave
// the waves run as long as their longest path while their lanes die (0.29 ms per 512 x 512 frame against 0.10 ms per frame inside a
// 1024-spp launch).  Ray / error counts are kept per frame (LDS atomics at path end; Technique::render fills statistics per frame).
// TBL (kernels that read the scene from HBM): the small tables every vertex touches — materials, lights, light CDF — are staged into LDS by the
// workgroup (LivingRoom: 65 materials = 3.2 KB), so the shading block's dependent reads (triangle -> material -> light) stop at the triangle.
// DYN: closest-hit and shadow rays of a trip share one traversal loop with dynamic fetch (traverse_dyn, pt_device.h);
// the shadow ray of vertex k is resolved at the start of trip k + 1.
// FLAT: the flat leaf list instead of the tree walk (traverse_flat, pt_device.h): no nodes in LDS, no traversal stack.
// UNI (DYN, wide records read from HBM): one fetch per iteration of the unified loop whatever the lane is at (traverse_dyn)
template <bool LDS_SCENE, int MODE, bool COUNT, int WAVES, int QN, int FEAT = kFeatAll, bool SPILL = true, bool TBL = false, bool DYN = false, bool FLAT = false, bool UNI = false>
__global__ __launch_bounds__(kBlock, WAVES) void pt_megakernel(const RenderParams p) {
  constexpr bool LIST = MODE == 1, FRAME = MODE == 2, IMAGE = MODE == 0;
  extern __shared__ float4 smem[];
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: keeps everything derived from it in SGPRs
  SceneView sv = p.sv;

  // LDS-resident variant: padded copy of the scene blob (stage_scene_to_lds, pt_device.h)
  constexpr int NS = LDS_SCENE ? 5 : 4, SS = LDS_SCENE ? 9 : 8;
  const uint32_t blob_f4 = LDS_SCENE ? (FLAT ? flat_scene_f4(sv, p.flat_k) : lds_scene_f4(sv)) : (TBL ? sv.blob_f4 - sv.off_mats : 0u);
  const float4* sb = sv.blob;
  const float4* __restrict__ light0 = p.sv.blob + p.sv.off_lights;  // global copy of the first light record (one-light variants read it through scalar loads)
  if (LDS_SCENE) { if (FLAT) stage_scene_flat(smem, sv, p.flat_table, p.flat_k, tid); else stage_scene_to_lds(smem, sv, tid); sb = smem; }
  cfloat* flat_table = (cfloat*)p.flat_table;  // constant address space: wave-uniform reads become scalar loads
  // tb / tv: where the tables are read from — the LDS copy of the whole scene, the LDS copy of the tables alone, or the blob in HBM
  const float4* tb = sb;
  SceneView tv = sv;
  if (!LDS_SCENE && TBL) {
    for (uint32_t i = tid; i < blob_f4; i += kBlock) smem[i] = sv.blob[sv.off_mats + i];
    tv.off_lights = sv.off_lights - sv.off_mats; tv.off_cdf = sv.off_cdf - sv.off_mats; tv.off_mats = 0u;
    tb = smem;
  }
  TravStackT<SPILL> stack;
  stack.lds = (lds_u32*)(reinterpret_cast<uint32_t*>(smem + blob_f4) + tid);
  stack.cap = p.stack_entries;
  char* acc_base = reinterpret_cast<char*>(smem + blob_f4) + size_t(p.stack_entries) * kBlock * 4 + size_t(wave) * kAccBytesPerWave;
  double* acc_r = reinterpret_cast<double*>(acc_base);
  double* acc_g = acc_r + 64;
  double* acc_b = acc_g + 64;
  uint32_t* acc_n = reinterpret_cast<uint32_t*>(acc_b + 64);
  DynLds dyn;  // DYN: this wave's parked shadow rays, mailbox and occlusion mask, behind the accumulators of the workgroup
  {
    char* db = reinterpret_cast<char*>(smem + blob_f4) + size_t(p.stack_entries) * kBlock * 4 + size_t(kWavesPerBlock) * kAccBytesPerWave + size_t(wave) * kDynBytesPerWave;
    dyn.ray = (lds_f32*)reinterpret_cast<float*>(db);
    dyn.mailbox = (lds_u8*)reinterpret_cast<uint8_t*>(db + kDynRayBytes);
    dyn.occl = (lds_u32*)reinterpret_cast<uint32_t*>(db + kDynRayBytes + 64);
  }
#ifdef MI_DYN_STATS
  uint32_t dyn_stats[6] = {0, 0, 0, 0, 0, 0};
#endif
  bool pend = false;          // DYN: this lane parked a shadow ray at its last vertex; nee_saved = its contribution if unoccluded
  f3 nee_saved = F3(0, 0, 0);
  if (IMAGE) { acc_r[lane] = 0.0; acc_g[lane] = 0.0; acc_b[lane] = 0.0; acc_n[lane] = 0u; }
  if (FRAME) acc_n[lane] = 0u;  // per-frame counts of this wave: [frame][closest-hit rays, shadow rays, numeric errors, paths]
  __syncthreads();

  // ---- which pool does this wave own?  XCD-aware: workgroups are dealt round-robin over the 8
  // XCDs (b % 8), so give each XCD a contiguous range of logical work and with it a contiguous
  // screen region (its L2 then sees one part of the BVH).  Speed only, never correctness.
  const uint32_t nb = gridDim.x, b = blockIdx.x;
  const uint32_t q = nb >> 3, rr = nb & 7u, xcd = b & 7u, slot = b >> 3;
  const uint32_t logical_block = xcd * q + (xcd < rr ? xcd : rr) + slot;
  const uint32_t logical_wave = logical_block * uint32_t(kWavesPerBlock) + wave;

  uint32_t pool_next, pool_end, tile_x0 = 0, tile_y0 = 0, chunk = 0;
  uint64_t chunk_sample0 = 0;
  if (LIST) {
    const uint32_t per_wave = 64u * 16u;
    pool_next = logical_wave * per_wave;
    pool_end = pool_next + per_wave < p.list_n ? pool_next + per_wave : p.list_n;
    if (pool_next > pool_end) pool_next = pool_end;
  } else if (FRAME) {
    // wave = (group of frame_tiles_per_wave tiles) x (chunk of frame_chunk consecutive frames of the launch); consecutive waves share the tiles.
    // Measured on C2 (tools/sessions/spp_scaling.py, profiles/r02/cadence.txt): a launch wants >= 2-3 rounds of waves AND >= 2 paths per lane.
    const uint32_t n_fc = (p.frame_count + p.frame_chunk - 1u) / p.frame_chunk, n_tiles = p.tiles_x * p.tiles_y;
    const uint32_t group = logical_wave / n_fc, fc = logical_wave - group * n_fc;
    chunk = fc * p.frame_chunk;                                                              // first frame of the wave
    tile_y0 = p.frame_count - chunk < p.frame_chunk ? p.frame_count - chunk : p.frame_chunk;  // its frames
    tile_x0 = group * p.frame_tiles_per_wave;                                                // first tile of the wave
    const uint32_t tcount = tile_x0 < n_tiles ? (n_tiles - tile_x0 < p.frame_tiles_per_wave ? n_tiles - tile_x0 : p.frame_tiles_per_wave) : 0u;
    pool_next = 0;
    pool_end = 64u * tcount * tile_y0;
  } else {
    const uint32_t n_tiles = p.tiles_x * p.tiles_y;
    const uint32_t tile = logical_wave / p.n_chunks;
    chunk = logical_wave - tile * p.n_chunks;
    const bool valid = tile < n_tiles;
    tile_origin(p, valid ? tile : 0u, tile_x0, tile_y0);
    const uint32_t s0 = chunk * p.chunk_spp;
    const uint32_t s1 = s0 + p.chunk_spp < p.spp ? s0 + p.chunk_spp : p.spp;
    chunk_sample0 = p.sample_offset + s0;
    pool_next = 0;
    pool_end = (valid && s1 > s0) ? (s1 - s0) * 64u : 0u;
  }

  const m33 v2w = {F3(p.v2w[0], p.v2w[1], p.v2w[2]), F3(p.v2w[3], p.v2w[4], p.v2w[5]), F3(p.v2w[6], p.v2w[7], p.v2w[8])};
  const f3 cam_pos = F3(p.cam_pos[0], p.cam_pos[1], p.cam_pos[2]);
  const f3 cam_gnormal = -v2w.c2;  // Technique::_camera_surface (Technique.cpp:107-116)

  // ---- per-lane path state ----
  bool alive = false;
  bool bounce = false;  // false = before the first surface vertex (PT.cpp:20-26), true = after a BSDF sample (PT.cpp:46-82)
  f3 org = F3(0, 0, 0), dir = F3(0, 0, 1);
  f3 xpos = F3(0, 0, 0);   // eye[prv].surface.position
  f3 tnum = F3(0, 0, 0);   // eye[prv].throughput * bsdf.throughput * edge.bCosTheta (PT.cpp:59-60)
  float bs_density = 1.0f; // bsdf.density
  bool bs_finite = true;   // bsdf.finite
  f3 radiance = F3(0, 0, 0);
  // path_size (PT.cpp:38) in the low 26 bits, pixel-in-tile in the high 6: one register.  A path of 2^26 edges
  // cannot occur (roulette survival 0.9^n); the counter saturates there and max_path >= 2^26 means unlimited.
  uint32_t ps_pix = 0;
  Rng rng; rng.state = 0;
  uint32_t item_id = 0;
  uint32_t n_basic = 0, n_shadow = 0, n_err = 0, n_paths = 0;  // wave-uniform (ballot popcounts): live in SGPRs
  uint32_t path_basic = 0, path_shadow = 0;  // LIST mode per-path counts
  Visits vis_c = {0u, 0u, nullptr}, vis_s = {0u, 0u, nullptr};  // instrumented variant only
  if (COUNT) {  // wave-level loop-body counters (node-loop bodies, leaf-phase bodies) in the tail of the wave's LDS block
    uint32_t* wi = acc_n + 64;
    if (lane < 8) wi[lane] = 0u;
    vis_c.wave_iters = wi;
    vis_s.wave_iters = wi + 2;
  }
  uint32_t trips_c = 0, trips_s = 0;           // instrumented: sum over loop trips of the slowest lane's traversal steps
  uint32_t n_hits = 0;

#ifdef MI_PHASE_TIMING
  unsigned long long phase_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_t = __builtin_amdgcn_s_memtime();
#endif
  for (;;) {
    MI_STAMP(7);
    bool t_started = false, t_shadow = false, t_err = false;  // per-trip events, counted wave-wide at the end of the trip
    // ---- path regeneration: dead lanes take the next samples of the wave's pool ----
    {
      const bool need = !alive;
      const uint64_t m = __ballot(need);
      if (m != 0ull && pool_next < pool_end) {
        const uint32_t item = pool_next + rank_in(m);
        pool_next += uint32_t(__popcll(m));
        if (need && item < pool_end) {
          uint32_t px, py; uint64_t sample; bool ok = true;
          if (LIST) {
            px = p.list_xy[2 * item]; py = p.list_xy[2 * item + 1]; sample = p.list_sample[item];
            item_id = item;
          } else if (FRAME) {
            // 64 consecutive items = one tile in one frame; the next 64 = the same tile in the next frame
            const uint32_t slab = item >> 6, tl = slab / tile_y0, frame = chunk + (slab - tl * tile_y0);
            uint32_t fx0, fy0;
            tile_origin(p, tile_x0 + tl, fx0, fy0);
            px = fx0 + (item & 7u); py = fy0 + ((item >> 3) & 7u);
            sample = p.sample_offset + frame;
            ok = px < p.win_x0 + p.win_w && py < p.win_y0 + p.win_h;
            item_id = frame * p.frame_stride + py * p.width + px;
            ps_pix = frame << 26;
          } else {
            const uint32_t pix = item & 63u;
            ps_pix = pix << 26;
            px = tile_x0 + (pix & 7u); py = tile_y0 + (pix >> 3);
            sample = chunk_sample0 + (item >> 6);
            ok = px < p.win_x0 + p.win_w && py < p.win_y0 + p.win_h;
          }
          if (ok) {
            // shoot() (Technique.cpp:321-331) + ray_direction (Cameras.cpp:120-127)
            rng = rng_seed(p.seed, py * p.width + px, sample);
            const float u0 = rng_f(rng), u1 = rng_f(rng);
            const float fx = float(px) + u0, fy = float(py) + u1;
            const float vx = fx * p.res_y_inv * 2.0f - p.res_x_res_y_inv;
            const float vy = fy * p.res_y_inv * 2.0f - 1.0f;
            dir = mulmv(v2w, normalize(F3(vx, vy, -p.focal_length_y)));
            org = nudge(cam_pos, cam_gnormal, dir);
            bounce = false; radiance = F3(0, 0, 0); ps_pix &= 0xFC000000u; alive = true;
            path_basic = 0; path_shadow = 0;
            t_started = true;
          }
        }
      }
    }
    n_paths += uint32_t(__popcll(__ballot(t_started)));
    {
      const uint64_t am = __ballot(alive);
      if (am == 0ull) {
        if (pool_next >= pool_end) break;
        continue;
      }
      n_basic += uint32_t(__popcll(am));  // every live lane casts exactly one closest-hit ray per trip
    }

    MI_STAMP(0);  // regeneration
    uint32_t steps_mine_c = 0, steps_mine_s = 0;
    Hit h;
    h.t = __builtin_inff(); h.u = h.v = 0.0f; h.id = 0xFFFFFFFFu; h.pos = 0;
    if (DYN) {
      // every lane of the wave takes part: its own closest-hit ray first, then unstarted shadow rays of the wave (Scene.cpp:151-203)
#ifdef MI_DYN_STATS
      const float visible = traverse_dyn<QN, NS, MI_DYN_TH, false, UNI>(sb, sv, stack, dyn, lane, alive, org, dir, pend, h, nullptr, nullptr, nullptr, dyn_stats);
      ++dyn_stats[5];
#else
      uint32_t trips = 0;
      const float visible = traverse_dyn<QN, NS, MI_DYN_TH, COUNT, UNI>(sb, sv, stack, dyn, lane, alive, org, dir, pend, h, &vis_c, &vis_s, &trips);
      if (COUNT) trips_c += trips;  // wave-uniform: the trips of the unified loop (closest-hit and shadow rays together)
#endif
      if (pend) { radiance = radiance + nee_saved * visible; pend = false; }  // PT.cpp:41: radiance += _connect(...) of the previous vertex
      MI_STAMP(1);
    }
    if (alive) {
      // ---- Scene::intersect (Scene.cpp:182-203) ----
      const uint32_t steps0 = vis_c.nodes + vis_c.tris;
      if (FLAT) { traverse_flat<false, COUNT>(sb, flat_table, (p.flat_k + 3u) >> 2, 0xFFFFFFFFu, org, dir, h, &vis_c); finish_hit(h); }
      else if (!DYN) traverse<false, COUNT, QN, NS, false>(sb, sv, stack, org, dir, 0xFFFFFFFFu, h, &vis_c);  // unmasked: PT's closest-hit rays see every geometry
      MI_STAMP(1);  // closest-hit traversal
      ++path_basic;
      if (COUNT && h.id != 0xFFFFFFFFu) ++n_hits;
      if (COUNT) steps_mine_c = vis_c.nodes + vis_c.tris - steps0;

      bool terminate = false, do_vertex = false;
      Surf sp;
      f3 x_throughput = F3(1, 1, 1);
      if (h.id == 0xFFFFFFFFu) {
        terminate = true;  // PT.cpp:28,49-51: a miss ends the path (PT ignores the sky)
      } else {
        sp = query_surface<SS>(sb, sv, org, dir, h);
        const bool is_light = surf_is_light(sp);
        if (!bounce) {
          if (is_light && p.max_path > 0u) {  // PT.cpp:23-26: directly visible light, continue through it
            const Material lm = load_material(tb, tv, sp.material_id);
            f3 le; float dens;
            query_lsdf<FEAT>(tb, tv, light0, lm.light_id, -dir, le, dens);
            radiance = radiance + le * p.lights;
            org = nudge(sp.position, sp.gnormal, dir);
          } else if (p.max_path < 2u) {
            terminate = true;  // PT.cpp:28-30
          } else {
            ps_pix = (ps_pix & 0xFC000000u) | 2u; do_vertex = true;  // PT.cpp:32-38: path_size = 2
          }
        } else {
          // new vertex z = hit (PT.cpp:53-68); Edge(eye[prv], eye[itr], -dir)
          const f3 omega = -dir;
          const f3 d = xpos - sp.position;
          const float distSqInv = mi_rcp(dot(d, d));
          const float fCos = fabsf(dot(omega, sp.tangent.c1));
          const float fG = distSqInv * fCos;
          if (l1norm(tnum) < MI_FLT_EPSILON) {
            terminate = true;  // PT.cpp:62-64
          } else {
            const f3 ztp = tnum / bs_density;  // PT.cpp:66
            if (is_light) {  // PT.cpp:70-79: MIS-weighted emission, then continue through the light
              const Material lm = load_material(tb, tv, sp.material_id);
              f3 le; float dens;
              query_lsdf<FEAT>(tb, tv, light0, lm.light_id, omega, le, dens);
              float wInv = mi_div(powb<FEAT>(dens, p.beta), powb<FEAT>(fG * bs_density, p.beta)) + 1.0f;
              if (!bs_finite) wInv = 1.0f;
              radiance = radiance + (le * ztp) / wInv;
              org = nudge(sp.position, sp.gnormal, dir);
            } else {
              // Russian roulette (PT.cpp:86-94)
              const uint32_t path_size = ps_pix & 0x03FFFFFFu;
              const bool free_ride = path_size < p.min_subpath;
              const float roul = free_ride ? 1.0f : p.roulette;
              const float uu = rng_f(rng);
              if (roul < uu) {
                terminate = true;
              } else {
                x_throughput = ztp * (free_ride ? 1.0f : p.inv_roulette);  // ztp / roul = ztp * (1 / roul): the reciprocal is a launch constant (host, same IEEE division)
                if (path_size != 0x03FFFFFFu) ++ps_pix;
                if (path_size + 1u > p.max_path) terminate = true; else do_vertex = true;  // PT.cpp:40
              }
            }
          }
        }
      }

      MI_STAMP(2);  // querySurface + path logic
      if (do_vertex) {
        // ---- vertex x = sp: NEE (PT.cpp:41) then BSDF sample (PT.cpp:43-44) ----
        const Material mat = load_material(tb, tv, sp.material_id);
        const f3 x_omega = -dir;
        // the vertex's shadow ray and its contribution if unoccluded (PT.cpp:117-119 without the visibility) live inside this block only:
        // declared outside the loop they were loop-carried for the register allocator (10 VGPRs it then spilled around)
        bool pending = false;
        ShadowRay sray; sray.org = F3(0, 0, 0); sray.dir = F3(0, 0, 1);
        const f3 nee = connect_prepare<FEAT>(tb, tv, light0, rng, mat, sp, x_omega, x_throughput, p.beta, pending, sray);
        if (pending) { t_shadow = true; ++path_shadow; }
        MI_STAMP(3);  // NEE set-up
        // a contribution that is exactly zero (delta BSDF at x, black surface) cannot change the sum whatever the
        // visibility: the ray the reference would cast (and count, Scene.cpp:177) is counted but not traversed
        if (pending && !(nee.x != 0.0f || nee.y != 0.0f || nee.z != 0.0f)) pending = false;
        if (DYN) {
          if (pending) { dyn_park_shadow_ray(dyn, lane, sray.org, sray.dir); pend = true; nee_saved = nee; }
        } else if (pending) {
          Hit sh; sh.t = 1.0f; sh.u = sh.v = 0.0f; sh.id = 0xFFFFFFFFu; sh.pos = 0;
          const uint32_t s0 = vis_s.nodes + vis_s.tris;
          if (FLAT) traverse_flat<true, COUNT>(sb, flat_table, (p.flat_k_mesh + 3u) >> 2, p.flat_k_mesh >= 32u ? 0xFFFFFFFFu : (1u << p.flat_k_mesh) - 1u, sray.org, sray.dir, sh, &vis_s);
          else traverse<true, COUNT, QN, NS>(sb, sv, stack, sray.org, sray.dir, 1u << MI_ENTITY_MESH, sh, &vis_s);
          if (COUNT) steps_mine_s = vis_s.nodes + vis_s.tris - s0;
          radiance = radiance + nee * (sh.id != 0xFFFFFFFFu ? 0.f : 1.f);  // PT.cpp:41: radiance += _connect(...)
        }
        MI_STAMP(4);  // shadow traversal
        const f3 x_position = sp.position, x_gnormal = sp.gnormal;
        const BSample bs = bsdf_sample<FEAT>(mat, rng, sp, x_omega);
        const float bCos = fabsf(dot(-bs.omega, sp.tangent.c1));  // Edge::bCosTheta with omega = -bsdf.omega
        tnum = (x_throughput * bs.q.throughput) * bCos;
        bs_density = bs.q.density; bs_finite = bs.q.finite != 0;
        xpos = x_position;
        dir = bs.omega;
        org = nudge(x_position, x_gnormal, dir);
        bounce = true;
      }

      MI_STAMP(5);  // BSDF sample
      if (terminate) {
        // ---- _eye_image += radiance; finite filter of _commit_images (Technique.cpp:222-230,338) ----
        alive = false;
        if (LIST) {
          p.list_radiance[3 * item_id] = radiance.x; p.list_radiance[3 * item_id + 1] = radiance.y; p.list_radiance[3 * item_id + 2] = radiance.z;
          if (p.list_counts) { p.list_counts[2 * item_id] = path_basic; p.list_counts[2 * item_id + 1] = path_shadow; }
        } else if (FRAME) {  // the pixel's only sample of this frame: (radiance, 1), or nothing counted if it is not finite
          const bool fin = isfinite(l1norm(radiance));
          p.frame_rgbn[item_id] = fin ? make_float4(radiance.x, radiance.y, radiance.z, 1.0f) : make_float4(0.f, 0.f, 0.f, 0.f);
          if (!fin) t_err = true;
          uint32_t* fc = acc_n + 4u * (ps_pix >> 26);  // this wave's per-frame counts: closest-hit rays, shadow rays, numeric errors, paths
          atomicAdd(&fc[0], path_basic); atomicAdd(&fc[1], path_shadow); atomicAdd(&fc[2], fin ? 0u : 1u); atomicAdd(&fc[3], 1u);
        } else if (isfinite(l1norm(radiance))) {
          const uint32_t pix = ps_pix >> 26;
          atomicAdd(&acc_r[pix], double(radiance.x));
          atomicAdd(&acc_g[pix], double(radiance.y));
          atomicAdd(&acc_b[pix], double(radiance.z));
          atomicAdd(&acc_n[pix], 1u);
        } else {
          t_err = true;
        }
      }
    }
    n_shadow += uint32_t(__popcll(__ballot(t_shadow)));
    n_err += uint32_t(__popcll(__ballot(t_err)));
    MI_STAMP(6);  // commit
    if (COUNT) {  // wave-level trip counts: what the wave pays is the slowest lane of each traversal
      trips_c += wave_max(steps_mine_c);
      trips_s += wave_max(steps_mine_s);
    }
  }

  // ---- the wave's sums leave LDS once: partial[chunk][pixel] = (r, g, b, count) ----
  if (IMAGE && pool_end != 0u) {
    // the lane index and the accumulator addresses are derived afresh here (mbcnt, not threadIdx): kept from the prologue they were four VGPRs
    // that lived — in scratch — across the whole path loop
    const uint32_t le = rank_in(~0ull);
    const uint32_t px = tile_x0 + (le & 7u), py = tile_y0 + (le >> 3);
    if (px < p.win_x0 + p.win_w && py < p.win_y0 + p.win_h) {
      char* ab = reinterpret_cast<char*>(smem + blob_f4) + size_t(p.stack_entries) * kBlock * 4 + size_t(wave) * kAccBytesPerWave;
      const double* er = reinterpret_cast<const double*>(ab);
      const uint32_t* en = reinterpret_cast<const uint32_t*>(er + 192);
      double* o = p.partial + (size_t(chunk) * p.width * p.height + size_t(py) * p.width + px) * 4;
      reinterpret_cast<double2*>(o)[0] = make_double2(er[le], er[64 + le]);
      reinterpret_cast<double2*>(o)[1] = make_double2(er[128 + le], double(en[le]));
    }
  }
  if (FRAME && p.counters && lane < 4u * p.frame_count && acc_n[lane]) atomicAdd(&p.counters[32u + lane], (unsigned long long)acc_n[lane]);
  const uint32_t sb_ = n_basic, ss_ = n_shadow, se_ = n_err, sp_ = n_paths;
  if (lane == 0 && p.counters) {
    if (sb_) atomicAdd(&p.counters[0], (unsigned long long)sb_);
    if (ss_) atomicAdd(&p.counters[1], (unsigned long long)ss_);
    if (se_) atomicAdd(&p.counters[2], (unsigned long long)se_);
    if (sp_) atomicAdd(&p.counters[3], (unsigned long long)sp_);
  }
#ifdef MI_PHASE_TIMING
  if (lane == 0 && p.counters) for (int k = 0; k < 8; ++k) atomicAdd(&p.counters[16 + k], phase_t[k]);
#endif
#ifdef MI_DYN_STATS
  if (DYN && lane == 0 && p.counters) for (int k = 0; k < 6; ++k) atomicAdd(&p.counters[16 + k], (unsigned long long)dyn_stats[k]);  // reported as phase_cycles[k]
#endif
  if (COUNT && p.counters) {
    const uint32_t v0 = wave_sum(vis_c.nodes), v1 = wave_sum(vis_c.tris), v2 = wave_sum(vis_s.nodes), v3 = wave_sum(vis_s.tris), v4 = wave_sum(n_hits);
    if (lane == 0) {
      atomicAdd(&p.counters[4], (unsigned long long)v0); atomicAdd(&p.counters[5], (unsigned long long)v1);
      atomicAdd(&p.counters[6], (unsigned long long)v2); atomicAdd(&p.counters[7], (unsigned long long)v3);
      atomicAdd(&p.counters[8], (unsigned long long)v4);
      atomicAdd(&p.counters[9], (unsigned long long)trips_c);
      atomicAdd(&p.counters[10], (unsigned long long)trips_s);
      for (int k = 0; k < 4; ++k) atomicAdd(&p.counters[11 + k], (unsigned long long)vis_c.wave_iters[k]);
    }
  }
}

#ifndef MI_PT_FAST
// rgbn[pixel] = sum over chunks (fixed order) of the FP64 partials, cast to FP32 — the
// payload handed to the caller / to the RCCL reduce ([H][W][4], row 0 = bottom).
__global__ __launch_bounds__(256) void pt_finalize(const double* __restrict__ partial, float4* __restrict__ rgbn, uint32_t width,
                                                  uint32_t height, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                                                  uint32_t n_chunks) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= width * height) return;
  const uint32_t y = i / width, x = i - y * width;
  float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
  if (x >= x0 && x < x0 + w && y >= y0 && y < y0 + h) {
    double r = 0.0, g = 0.0, b = 0.0, n = 0.0;
    for (uint32_t c = 0; c < n_chunks; ++c) {
      const double2* s = reinterpret_cast<const double2*>(partial + (size_t(c) * width * height + i) * 4);
      const double2 a = s[0], bb = s[1];
      r += a.x; g += a.y; b += bb.x; n += bb.y;
    }
    out = make_float4(float(r), float(g), float(b), float(n));
  }
  rgbn[i] = out;
}

// Batched Scene::intersect + querySurface (parity hook, mi_pt_intersect).
template <int QN>
__global__ __launch_bounds__(kBlock) void k_intersect(SceneView sv, uint32_t stack_entries, uint32_t n, const mi_surface_point* __restrict__ origins,
                                                     const float* __restrict__ dirs, mi_surface_point* __restrict__ out_hits,
                                                     float* __restrict__ out_t, uint32_t* __restrict__ out_prim) {
  extern __shared__ float4 smem[];
  TravStack stack;
  stack.lds = (lds_u32*)(reinterpret_cast<uint32_t*>(smem) + threadIdx.x);
  stack.cap = stack_entries;
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const f3 pos = F3(origins[i].position[0], origins[i].position[1], origins[i].position[2]);
  const f3 gn = F3(origins[i].gnormal[0], origins[i].gnormal[1], origins[i].gnormal[2]);
  const f3 dir = F3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
  const f3 org = nudge(pos, gn, dir);
  Hit h; h.t = __builtin_inff(); h.u = h.v = 0.0f; h.id = 0xFFFFFFFFu; h.pos = 0;
  traverse<false, false, QN, 4, true, true>(sv.blob, sv, stack, org, dir, 0xFFFFFFFFu, h);  // FAR: a caller's point may lie anywhere
  if (out_t) out_t[i] = h.t;
  if (out_prim) out_prim[i] = h.id;
  if (out_hits) {
    mi_surface_point o;
    if (h.id == 0xFFFFFFFFu) {
      for (int k = 0; k < 3; ++k) { o.position[k] = 0; o.gnormal[k] = 0; }
      for (int k = 0; k < 9; ++k) o.tangent[k] = 0;
      o.material_id = 0xFFFFFFFFu;
    } else {
      const Surf s = query_surface(sv.blob, sv, org, dir, h);
      o.position[0] = s.position.x; o.position[1] = s.position.y; o.position[2] = s.position.z;
      o.gnormal[0] = s.gnormal.x; o.gnormal[1] = s.gnormal.y; o.gnormal[2] = s.gnormal.z;
      o.tangent[0] = s.tangent.c0.x; o.tangent[1] = s.tangent.c0.y; o.tangent[2] = s.tangent.c0.z;
      o.tangent[3] = s.tangent.c1.x; o.tangent[4] = s.tangent.c1.y; o.tangent[5] = s.tangent.c1.z;
      o.tangent[6] = s.tangent.c2.x; o.tangent[7] = s.tangent.c2.y; o.tangent[8] = s.tangent.c2.z;
      o.material_id = s.material_id;
    }
    out_hits[i] = o;
  }
}

// Batched Scene::occluded (parity hook, mi_pt_occluded).
template <int QN>
__global__ __launch_bounds__(kBlock) void k_occluded(SceneView sv, uint32_t stack_entries, uint32_t n, const mi_surface_point* __restrict__ a,
                                                    const mi_surface_point* __restrict__ b, float* __restrict__ out) {
  extern __shared__ float4 smem[];
  TravStack stack;
  stack.lds = (lds_u32*)(reinterpret_cast<uint32_t*>(smem) + threadIdx.x);
  stack.cap = stack_entries;
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  out[i] = occluded<false, QN, 4, true>(sv.blob, sv, stack, F3(a[i].position[0], a[i].position[1], a[i].position[2]),
                    F3(a[i].gnormal[0], a[i].gnormal[1], a[i].gnormal[2]), F3(b[i].position[0], b[i].position[1], b[i].position[2]),
                    F3(b[i].gnormal[0], b[i].gnormal[1], b[i].gnormal[2]));
}

// The parity hooks through the flat leaf list (MI_PT_INTERSECT_FLAT=1, scenes that have a leaf table): the same staging and the same
// traverse_flat the megakernel's FLAT variants run, so adversarial rays (axis-parallel, grazing, on box faces) reach it directly.
__global__ __launch_bounds__(kBlock) void k_intersect_flat(SceneView sv, const float* __restrict__ table, uint32_t K, uint32_t n, const mi_surface_point* __restrict__ origins,
                                                          const float* __restrict__ dirs, mi_surface_point* __restrict__ out_hits, float* __restrict__ out_t,
                                                          uint32_t* __restrict__ out_prim) {
  extern __shared__ float4 smem[];
  stage_scene_flat(smem, sv, table, K, threadIdx.x);
  __syncthreads();
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const f3 pos = F3(origins[i].position[0], origins[i].position[1], origins[i].position[2]);
  const f3 gn = F3(origins[i].gnormal[0], origins[i].gnormal[1], origins[i].gnormal[2]);
  const f3 dir = F3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
  const f3 org = nudge(pos, gn, dir);
  Hit h; h.t = __builtin_inff(); h.u = h.v = 0.0f; h.id = 0xFFFFFFFFu; h.pos = 0;
  traverse_flat<false>(smem, (cfloat*)table, (K + 3u) >> 2, 0xFFFFFFFFu, org, dir, h, nullptr);
  finish_hit(h);
  if (out_t) out_t[i] = h.t;
  if (out_prim) out_prim[i] = h.id;
  if (out_hits) {
    mi_surface_point o;
    if (h.id == 0xFFFFFFFFu) {
      for (int k = 0; k < 3; ++k) { o.position[k] = 0; o.gnormal[k] = 0; }
      for (int k = 0; k < 9; ++k) o.tangent[k] = 0;
      o.material_id = 0xFFFFFFFFu;
    } else {
      const Surf s = query_surface<9>(smem, sv, org, dir, h);
      o.position[0] = s.position.x; o.position[1] = s.position.y; o.position[2] = s.position.z;
      o.gnormal[0] = s.gnormal.x; o.gnormal[1] = s.gnormal.y; o.gnormal[2] = s.gnormal.z;
      o.tangent[0] = s.tangent.c0.x; o.tangent[1] = s.tangent.c0.y; o.tangent[2] = s.tangent.c0.z;
      o.tangent[3] = s.tangent.c1.x; o.tangent[4] = s.tangent.c1.y; o.tangent[5] = s.tangent.c1.z;
      o.tangent[6] = s.tangent.c2.x; o.tangent[7] = s.tangent.c2.y; o.tangent[8] = s.tangent.c2.z;
      o.material_id = s.material_id;
    }
    out_hits[i] = o;
  }
}

__global__ __launch_bounds__(kBlock) void k_occluded_flat(SceneView sv, const float* __restrict__ table, uint32_t K, uint32_t k_mesh, uint32_t n,
                                                         const mi_surface_point* __restrict__ a, const mi_surface_point* __restrict__ b, float* __restrict__ out) {
  extern __shared__ float4 smem[];
  stage_scene_flat(smem, sv, table, K, threadIdx.x);
  __syncthreads();
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  // Scene::occluded (Scene.cpp:151-180), as occluded() in pt_device.h
  const f3 opos = F3(a[i].position[0], a[i].position[1], a[i].position[2]), ogn = F3(a[i].gnormal[0], a[i].gnormal[1], a[i].gnormal[2]);
  const f3 tpos = F3(b[i].position[0], b[i].position[1], b[i].position[2]), tgn = F3(b[i].gnormal[0], b[i].gnormal[1], b[i].gnormal[2]);
  const f3 direction = tpos - opos;
  const f3 ao = opos + (ogn * (dot(ogn, direction) > 0.0f ? 1.0f : -1.0f)) * 0.0001f;
  const f3 at = tpos + (tgn * (dot(tgn, direction) < 0.0f ? 1.0f : -1.0f)) * 0.0001f;
  Hit h; h.t = 1.0f; h.u = h.v = 0.0f; h.id = 0xFFFFFFFFu; h.pos = 0;
  traverse_flat<true>(smem, (cfloat*)table, (k_mesh + 3u) >> 2, k_mesh >= 32u ? 0xFFFFFFFFu : (1u << k_mesh) - 1u, ao, at - ao, h, nullptr);
  out[i] = h.id != 0xFFFFFFFFu ? 0.f : 1.f;
}

#endif  // !MI_PT_FAST
#ifdef MI_ONE_KERNEL
// register-budget experiments (tools/one_kernel.sh): compile ONE instantiation and stop — seconds instead of minutes
template __global__ void pt_megakernel<MI_ONE_KERNEL>(const RenderParams p);
#else
// ---- host-callable launchers (declared in launch.h) ----
size_t pt_lds_bytes(const RenderParams& p, bool lds_scene) {
  return (lds_scene ? (p.flat_k ? size_t(kFlatLeafF4 * p.flat_k + 18u * p.flat_k + (p.sv.blob_f4 - p.sv.off_mats)) * 16 : size_t(p.sv.blob_f4 + p.sv.n_nodes + p.sv.n_tris) * 16)
                    : (p.lds_tables ? size_t(p.sv.blob_f4 - p.sv.off_mats) * 16 : 0)) +
         size_t(p.stack_entries) * kBlock * 4 + kWavesPerBlock * kAccBytesPerWave + (p.dyn_traverse ? kWavesPerBlock * kDynBytesPerWave : 0);
}

hipError_t launch_megakernel(const RenderParams& p, bool lds_scene, int mode, bool count, uint32_t n_blocks, hipStream_t stream) {
  size_t lds = pt_lds_bytes(p, lds_scene);
  if (const char* e = std::getenv("MI_PT_LDS_PAD")) lds += size_t(std::atol(e));  // measurement only: more LDS per workgroup = fewer workgroups per CU
  void (*fn)(const RenderParams) = nullptr;
  const bool large = p.wide_nodes == 1u;
  const bool list = mode == 1;
  const bool six = lds <= (160u * 1024u) / MI_DYN_W_HI;  // six workgroups of this LDS size fit a CU
#ifdef MI_PT_FAST
  if (count || list || mode != 0) return hipErrorInvalidValue;  // the fast-arithmetic build holds the image-mode variants only
#define MI_MODE2(a, b) (b)
#else
#define MI_MODE2(a, b) (mode == 2 ? (a) : (b))
  if (!lds_scene && p.wide_nodes == 2u && ((count || list) && !(p.dyn_traverse && p.lds_tables))) {  // full-precision 64-byte nodes from HBM: scenes whose triangles are small against the 16-bit grid
    if (count) fn = pt_megakernel<false, 0, true, MI_WAVES_HBM, 0>;
    else fn = pt_megakernel<false, 1, false, MI_WAVES_HBM, 0>;
  } else
  if (count && p.dyn_traverse && !lds_scene && p.lds_tables)  // instrumented dynamic-fetch variants (5-wave budget)
    fn = p.wide_nodes == 2u ? pt_megakernel<false, 0, true, 5, 0, kFeatAll, true, true, true>
                            : (large ? pt_megakernel<false, 0, true, 5, 2, kFeatAll, true, true, true> : pt_megakernel<false, 0, true, 5, 1, kFeatAll, true, true, true>);
  else if (count && lds_scene && p.flat_k) fn = pt_megakernel<true, 0, true, MI_WAVES_FLAT, 0, kFeatAll, false, false, false, true>;
  else if (list && lds_scene && p.flat_k) fn = pt_megakernel<true, 1, false, MI_WAVES_FLAT, 0, kFeatAll, false, false, false, true>;
  else if (count && p.dyn_traverse && lds_scene && p.stack_in_lds) fn = pt_megakernel<true, 0, true, MI_WAVES_LDS, 0, kFeatAll, false, false, true>;
  else if (count) fn = lds_scene ? pt_megakernel<true, 0, true, MI_WAVES_LDS, 0> : (large ? pt_megakernel<false, 0, true, MI_WAVES_HBM_LARGE, 2> : pt_megakernel<false, 0, true, MI_WAVES_HBM, 1>);
  else if (list && lds_scene && p.dyn_traverse && p.stack_in_lds) fn = pt_megakernel<true, 1, false, MI_WAVES_LDS, 0, kFeatAll, false, false, true>;  // per-path parity hook of the dynamic-fetch variant
  else if (list && !lds_scene && p.dyn_traverse && p.lds_tables)
    fn = p.wide_nodes == 2u ? pt_megakernel<false, 1, false, 5, 0, kFeatAll, true, true, true>
                            : (large ? pt_megakernel<false, 1, false, 5, 2, kFeatAll, true, true, true> : pt_megakernel<false, 1, false, 5, 1, kFeatAll, true, true, true>);
  else if (list) fn = lds_scene ? pt_megakernel<true, 1, false, MI_WAVES_LDS, 0> : (large ? pt_megakernel<false, 1, false, MI_WAVES_HBM_LARGE, 2> : pt_megakernel<false, 1, false, MI_WAVES_HBM, 1>);
  else
#endif
  {
    // the compiled feature set that covers the scene: any combination of Phong lobes and mirrors / glass with beta in {1, 2}, or everything
#define MI_PICK4(L, M, W, Q, S, B) (f2 == 0 ? pt_megakernel<L, M, false, W, Q, (B) | 0, S> : f2 == 1 ? pt_megakernel<L, M, false, W, Q, (B) | 1, S> : \
                                    f2 == 2 ? pt_megakernel<L, M, false, W, Q, (B) | 2, S> : pt_megakernel<L, M, false, W, Q, (B) | 3, S>)
#define MI_PICK(L, M, W, Q, S) (feat == kFeatAll ? pt_megakernel<L, M, false, W, Q, kFeatAll, S> : (feat & kFeatLights) ? MI_PICK4(L, M, W, Q, S, kFeatLights) : MI_PICK4(L, M, W, Q, S, 0))
#define MI_PICK_MODE(L, W, Q, S) MI_MODE2(MI_PICK(L, 2, W, Q, S), MI_PICK(L, 0, W, Q, S))
#define MI_PICK4T(M, W, Q, B) (f2 == 0 ? pt_megakernel<false, M, false, W, Q, (B) | 0, true, true> : f2 == 1 ? pt_megakernel<false, M, false, W, Q, (B) | 1, true, true> : \
                               f2 == 2 ? pt_megakernel<false, M, false, W, Q, (B) | 2, true, true> : pt_megakernel<false, M, false, W, Q, (B) | 3, true, true>)
#define MI_PICKT(M, W, Q) (feat == kFeatAll ? pt_megakernel<false, M, false, W, Q, kFeatAll, true, true> : (feat & kFeatLights) ? MI_PICK4T(M, W, Q, kFeatLights) : MI_PICK4T(M, W, Q, 0))
#define MI_PICK4TD(M, W, Q, B) (f2 == 0 ? pt_megakernel<false, M, false, W, Q, (B) | 0, true, true, true> : f2 == 1 ? pt_megakernel<false, M, false, W, Q, (B) | 1, true, true, true> : \
                                f2 == 2 ? pt_megakernel<false, M, false, W, Q, (B) | 2, true, true, true> : pt_megakernel<false, M, false, W, Q, (B) | 3, true, true, true>)
#define MI_PICKTD(M, W, Q) (feat == kFeatAll ? pt_megakernel<false, M, false, W, Q, kFeatAll, true, true, true> : (feat & kFeatLights) ? MI_PICK4TD(M, W, Q, kFeatLights) : MI_PICK4TD(M, W, Q, 0))
#define MI_PICK4TDU(W, B) (f2 == 0 ? pt_megakernel<false, 0, false, W, 2, (B) | 0, true, true, true, false, true> : f2 == 1 ? pt_megakernel<false, 0, false, W, 2, (B) | 1, true, true, true, false, true> : \
                           f2 == 2 ? pt_megakernel<false, 0, false, W, 2, (B) | 2, true, true, true, false, true> : pt_megakernel<false, 0, false, W, 2, (B) | 3, true, true, true, false, true>)
#define MI_PICKTDU(W) (feat == kFeatAll ? pt_megakernel<false, 0, false, W, 2, kFeatAll, true, true, true, false, true> : (feat & kFeatLights) ? MI_PICK4TDU(W, kFeatLights) : MI_PICK4TDU(W, 0))
#define MI_PICK_HBM(W, Q) (p.lds_tables ? (p.dyn_traverse ? (six ? MI_MODE2(MI_PICKTD(2, MI_DYN_W_HI, Q), MI_PICKTD(0, MI_DYN_W_HI, Q)) : MI_MODE2(MI_PICKTD(2, 5, Q), MI_PICKTD(0, 5, Q))) : MI_MODE2(MI_PICKT(2, W, Q), MI_PICKT(0, W, Q))) : MI_PICK_MODE(false, W, Q, true))
    const int feat = (p.features & uint32_t(kFeatPow)) ? kFeatAll : int(p.features);  // a general beta is rare: only the general variant has pow
    const int f2 = feat & 3;
#define MI_PICK4D(M, B) (f2 == 0 ? pt_megakernel<true, M, false, MI_WAVES_LDS, 0, (B) | 0, false, false, true> : f2 == 1 ? pt_megakernel<true, M, false, MI_WAVES_LDS, 0, (B) | 1, false, false, true> : \
                         f2 == 2 ? pt_megakernel<true, M, false, MI_WAVES_LDS, 0, (B) | 2, false, false, true> : pt_megakernel<true, M, false, MI_WAVES_LDS, 0, (B) | 3, false, false, true>)
#define MI_PICKD(M) (feat == kFeatAll ? pt_megakernel<true, M, false, MI_WAVES_LDS, 0, kFeatAll, false, false, true> : (feat & kFeatLights) ? MI_PICK4D(M, kFeatLights) : MI_PICK4D(M, 0))
#define MI_PICK4F(M, B) (f2 == 0 ? pt_megakernel<true, M, false, MI_WAVES_FLAT, 0, (B) | 0, false, false, false, true> : f2 == 1 ? pt_megakernel<true, M, false, MI_WAVES_FLAT, 0, (B) | 1, false, false, false, true> : \
                         f2 == 2 ? pt_megakernel<true, M, false, MI_WAVES_FLAT, 0, (B) | 2, false, false, false, true> : pt_megakernel<true, M, false, MI_WAVES_FLAT, 0, (B) | 3, false, false, false, true>)
#define MI_PICKF(M) (feat == kFeatAll ? pt_megakernel<true, M, false, MI_WAVES_FLAT, 0, kFeatAll, false, false, false, true> : (feat & kFeatLights) ? MI_PICK4F(M, kFeatLights) : MI_PICK4F(M, 0))
    if (lds_scene && p.flat_k) fn = MI_MODE2(MI_PICKF(2), MI_PICKF(0));  // flat leaf list
    else
    if (lds_scene && p.dyn_traverse && p.stack_in_lds) fn = MI_MODE2(MI_PICKD(2), MI_PICKD(0));  // unified traversal with dynamic fetch
    else
    if (lds_scene) fn = p.stack_in_lds ? MI_PICK_MODE(true, MI_WAVES_LDS, 0, false) : MI_PICK_MODE(true, MI_WAVES_LDS, 0, true);  // shallow tree: stack without the spill path
    else if (p.wide_nodes == 2u) fn = MI_PICK_HBM(MI_WAVES_HBM, 0);
    else if (large && mode == 0 && p.lds_tables && p.dyn_traverse && p.sv.dyn_uni) fn = six ? MI_PICKTDU(MI_DYN_W_HI) : MI_PICKTDU(5);  // wide records + one fetch per iteration
    else if (large) fn = MI_PICK_HBM(MI_WAVES_HBM_LARGE, 2);
    else fn = MI_PICK_HBM(MI_WAVES_HBM, 1);
#undef MI_PICK_HBM
#undef MI_PICKTDU
#undef MI_PICK4TDU
#undef MI_MODE2
#undef MI_PICKF
#undef MI_PICK4F
#undef MI_PICKTD
#undef MI_PICK4TD
#undef MI_PICKD
#undef MI_PICK4D
#undef MI_PICKT
#undef MI_PICK4T
#undef MI_PICK_MODE
#undef MI_PICK
#undef MI_PICK4
  }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fn, dim3(n_blocks), dim3(kBlock), lds, stream, p);
  return hipGetLastError();
}

#ifndef MI_PT_FAST
hipError_t launch_finalize(const double* partial, float* rgbn, uint32_t width, uint32_t height, uint32_t x0, uint32_t y0, uint32_t w,
                           uint32_t h, uint32_t n_chunks, hipStream_t stream) {
  const uint32_t n = width * height;
  hipLaunchKernelGGL(pt_finalize, dim3((n + 255) / 256), dim3(256), 0, stream, partial, reinterpret_cast<float4*>(rgbn), width, height,
                     x0, y0, w, h, n_chunks);
  return hipGetLastError();
}

hipError_t launch_intersect(const SceneView& sv, bool wide, uint32_t stack_entries, uint32_t n, const mi_surface_point* origins, const float* dirs,
                            mi_surface_point* out_hits, float* out_t, uint32_t* out_prim, hipStream_t stream) {
  if (wide) hipLaunchKernelGGL(k_intersect<2>, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), size_t(stack_entries) * kBlock * 4, stream, sv, stack_entries, n,
                               origins, dirs, out_hits, out_t, out_prim);
  else hipLaunchKernelGGL(k_intersect<1>, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), size_t(stack_entries) * kBlock * 4, stream, sv, stack_entries, n,
                     origins, dirs, out_hits, out_t, out_prim);
  return hipGetLastError();
}

hipError_t launch_occluded(const SceneView& sv, bool wide, uint32_t stack_entries, uint32_t n, const mi_surface_point* a, const mi_surface_point* b,
                           float* out, hipStream_t stream) {
  if (wide) hipLaunchKernelGGL(k_occluded<2>, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), size_t(stack_entries) * kBlock * 4, stream, sv, stack_entries, n, a, b,
                               out);
  else hipLaunchKernelGGL(k_occluded<1>, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), size_t(stack_entries) * kBlock * 4, stream, sv, stack_entries, n, a, b,
                     out);
  return hipGetLastError();
}

hipError_t launch_intersect_flat(const SceneView& sv, const float* table, uint32_t K, uint32_t n, const mi_surface_point* origins, const float* dirs,
                                 mi_surface_point* out_hits, float* out_t, uint32_t* out_prim, hipStream_t stream) {
  const size_t lds = size_t(kFlatLeafF4 * K + 18u * K + (sv.blob_f4 - sv.off_mats)) * 16;
  hipLaunchKernelGGL(k_intersect_flat, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), lds, stream, sv, table, K, n, origins, dirs, out_hits, out_t, out_prim);
  return hipGetLastError();
}

hipError_t launch_occluded_flat(const SceneView& sv, const float* table, uint32_t K, uint32_t k_mesh, uint32_t n, const mi_surface_point* a, const mi_surface_point* b,
                                float* out, hipStream_t stream) {
  const size_t lds = size_t(kFlatLeafF4 * K + 18u * K + (sv.blob_f4 - sv.off_mats)) * 16;
  hipLaunchKernelGGL(k_occluded_flat, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), lds, stream, sv, table, K, k_mesh, n, a, b, out);
  return hipGetLastError();
}

#endif  // !MI_PT_FAST
#endif  // MI_ONE_KERNEL

MI_PT_NS_END  // namespace mi (:: fastmath)
