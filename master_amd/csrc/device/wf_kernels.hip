// wf_kernels.hip — wavefront form of the PT integrator: path state in HBM, one kernel per ray cast / vertex (gfx950).
//
// The megakernel keeps a path in one lane from camera to termination and carries ~100 registers of shading state
// through the BVH loops.  Here the same per-path arithmetic (PathTracing::_traceEye, PT.cpp:15-98 — identical device
// functions, identical RNG draw order, so results are bit-identical per path) is cut at the two ray casts:
//
//   wf_init     every slot shoots its first camera ray (Technique.cpp:321-331)
//   wf_extend   Scene::intersect's traversal for every active slot                          -> hit record
//   wf_shade    querySurface, light pass-through / MIS emission, roulette, NEE set-up, BSDF sample (PT.cpp:20-94);
//               a path that ends is committed (Technique.cpp:222-230) and the slot starts its next path at once
//   wf_shadow   Scene::occluded's traversal for slots with a pending shadow ray; radiance += nee * visibility (PT.cpp:41)
//   wf_reduce   per pixel: the replicas' FP64 sums, in replica order, into the partial (Technique.cpp:338)
//
// Slots are pixel-affine: slot i owns pixel slot i % per_sample (8x8 tiles, so a wave is one screen tile, as in the
// megakernel) and renders samples r, r + R, ... with r = i / per_sample.  thread == slot in every kernel: all state
// traffic is coalesced SoA (~400 B per segment), there are no queues and no atomics on the data path, and the order
// of every sum is fixed.  (A first version compacted live paths into queues: scattered 16-byte state accesses and the
// loss of tile coherence in the traversal kernels made it 2-10x slower than the megakernel; profiles/r01/ab_wavefront.txt.)
// Traversal kernels keep only ray state in registers (46-48 VGPRs), so they run at 8+ waves per SIMD.
#include <hip/hip_runtime.h>

#include "pt_device.h"
#include "wavefront.h"

namespace mi {

namespace {

MI_DEV uint32_t lane_rank(uint64_t mask) { return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u)); }

MI_DEV void count_add(unsigned long long* __restrict__ c, bool pred) {
  const uint64_t m = __ballot(pred);
  if (m != 0ull && int(threadIdx.x & 63u) == __ffsll((long long)m) - 1) atomicAdd(c, (unsigned long long)__popcll(m));
}
MI_DEV uint32_t wf_wave_sum(uint32_t v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

constexpr uint32_t kFlagBounce = 1u, kFlagFinite = 2u;  // st_b.w: bit 0 bounce, bit 1 bs_finite, bits 2.. path_size (saturating)
constexpr uint32_t kPathSizeMax = 0x03FFFFFFu;
// XCD-aware block -> slot-range map: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8); give every XCD a
// contiguous range of slots = a contiguous band of screen tiles, so its L2 sees one part of the BVH (as the megakernel does).
MI_DEV uint32_t xcd_block(uint32_t b, uint32_t nb) {
  const uint32_t q = nb >> 3, rr = nb & 7u, xcd = b & 7u, k = b >> 3;
  return xcd * q + (xcd < rr ? xcd : rr) + k;
}

constexpr uint32_t kSlotActive = 1u, kSlotShadow = 2u, kSlotDone = 4u;  // ray_o.w: the slot's place in the round


// ---- a slot's next path: commit the one that just ended, shoot the next camera ray ----
MI_DEV void slot_next_path(const RenderParams& p, const WfState& w, uint32_t slot, bool commit, f3 radiance, uint2 cnt, uint32_t& k,
                           bool& started, bool& err, uint32_t& flags, f3& org, f3& dir, Rng& rng) {
  // work of slot i: image mode — pixel slot i % per_sample, samples r + j R (r = i / per_sample); list mode — items i + j P
  const uint32_t ps = slot % w.per_sample, r = slot / w.per_sample;
  if (commit) {
    // _eye_image += radiance; finite filter of _commit_images (Technique.cpp:222-230,338)
    if (w.list) {
      const size_t item = size_t(slot) + size_t(k - 1u) * w.P;
      p.list_radiance[3 * item] = radiance.x; p.list_radiance[3 * item + 1] = radiance.y; p.list_radiance[3 * item + 2] = radiance.z;
      if (p.list_counts) { p.list_counts[2 * item] = cnt.x; p.list_counts[2 * item + 1] = cnt.y; }
    } else if (isfinite(l1norm(radiance))) {
      double4 a = w.acc[slot];
      a.x += double(radiance.x); a.y += double(radiance.y); a.z += double(radiance.z); a.w += 1.0;
      w.acc[slot] = a;
    } else {
      err = true;
    }
  }
  uint32_t px, py; uint64_t sample; bool have;
  if (w.list) {
    const uint64_t item = uint64_t(slot) + uint64_t(k) * w.P;
    have = item < w.n_items;
    if (have) { px = p.list_xy[2 * item]; py = p.list_xy[2 * item + 1]; sample = p.list_sample[item]; }
  } else {
    const uint32_t tile = ps >> 6, pix = ps & 63u;
    uint32_t tx0, ty0;
    tile_origin(p, tile, tx0, ty0);
    px = tx0 + (pix & 7u); py = ty0 + (pix >> 3);
    const uint64_t s = uint64_t(r) + uint64_t(k) * w.R;
    have = s < p.spp && px < p.win_x0 + p.win_w && py < p.win_y0 + p.win_h;
    sample = p.sample_offset + s;
  }
  if (!have) { flags = kSlotDone; return; }
  ++k;
  // shoot() (Technique.cpp:321-331) + ray_direction (Cameras.cpp:120-127)
  const m33 v2w = {F3(p.v2w[0], p.v2w[1], p.v2w[2]), F3(p.v2w[3], p.v2w[4], p.v2w[5]), F3(p.v2w[6], p.v2w[7], p.v2w[8])};
  const f3 cam_pos = F3(p.cam_pos[0], p.cam_pos[1], p.cam_pos[2]);
  const f3 cam_gnormal = -v2w.c2;  // Technique::_camera_surface (Technique.cpp:107-116)
  rng = rng_seed(p.seed, py * p.width + px, sample);
  const float u0 = rng_f(rng), u1 = rng_f(rng);
  const float fx = float(px) + u0, fy = float(py) + u1;
  const float vx = fx * p.res_y_inv * 2.0f - p.res_x * p.res_y_inv;
  const float vy = fy * p.res_y_inv * 2.0f - 1.0f;
  dir = mulmv(v2w, normalize(F3(vx, vy, -p.focal_length_y)));
  org = nudge(cam_pos, cam_gnormal, dir);
  flags = kSlotActive;
  started = true;
}

}  // namespace

// first launch: every slot takes its first path
__global__ __launch_bounds__(256) void wf_init(const RenderParams p, const WfState w) {
  const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
  bool started = false, err = false;
  if (slot < w.P) {
    uint32_t k = 0, flags = 0; f3 org = F3(0, 0, 0), dir = F3(0, 0, 1); Rng rng; rng.state = 0;
    if (!w.list) w.acc[slot] = make_double4(0.0, 0.0, 0.0, 0.0);
    slot_next_path(p, w, slot, false, F3(0, 0, 0), make_uint2(0u, 0u), k, started, err, flags, org, dir, rng);
    w.rng[slot] = rng_pack(rng);
    w.ray_o[slot] = make_float4(org.x, org.y, org.z, __uint_as_float(flags));
    w.ray_d[slot] = make_float4(dir.x, dir.y, dir.z, __uint_as_float(k));
    w.st_a[slot] = make_float4(0.f, 0.f, 0.f, 1.0f);
    w.st_b[slot] = make_float4(0.f, 0.f, 0.f, __uint_as_float(kFlagFinite));
    w.st_c[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
    w.cnt[slot] = make_uint2(0u, 0u);
  }
  count_add(&p.counters[3], started);
  count_add(reinterpret_cast<unsigned long long*>(w.n_active), started);
}

// ---- Scene::intersect (Scene.cpp:182-203): closest-hit traversal of every active slot ----
template <bool COUNT>
__global__ __launch_bounds__(kBlock, 8) void wf_extend(const RenderParams p, const WfState w) {
  extern __shared__ float4 smem[];
  TravStack stack;
  stack.lds = (lds_u32*)(reinterpret_cast<uint32_t*>(smem) + threadIdx.x);
  stack.cap = p.stack_entries;
  const uint32_t slot = xcd_block(blockIdx.x, gridDim.x) * kBlock + threadIdx.x;
  Visits vis = {0u, 0u, nullptr};
  bool active = false;
  if (slot < w.P) {
    const float4 o = w.ray_o[slot];
    active = (__float_as_uint(o.w) & kSlotActive) != 0u;
    if (active) {
      const float4 d = w.ray_d[slot];
      Hit h; h.t = __builtin_inff(); h.u = h.v = 0.0f; h.id = 0xFFFFFFFFu; h.pos = 0;
      traverse<false, COUNT, MI_WF_QUANT>(p.sv.blob, p.sv, stack, F3(o.x, o.y, o.z), F3(d.x, d.y, d.z), 0xFFFFFFFFu, h, &vis);
      w.hit[slot] = make_float4(h.t, h.u, h.v, __uint_as_float(h.id == 0xFFFFFFFFu ? 0xFFFFFFFFu : h.pos));
    }
  }
  count_add(&p.counters[0], active);  // every active slot casts exactly one closest-hit ray per round
  if (COUNT) {
    const uint32_t a = wf_wave_sum(vis.nodes), b = wf_wave_sum(vis.tris);
    if ((threadIdx.x & 63u) == 0 && (a | b)) { atomicAdd(&p.counters[4], (unsigned long long)a); atomicAdd(&p.counters[5], (unsigned long long)b); }
  }
}

// ---- the vertex: PT.cpp:20-94 between the two ray casts; a path that ends here is committed and the slot starts its next one ----
__global__ __launch_bounds__(256) void wf_shade(const RenderParams p, const WfState w) {
  const uint32_t slot = xcd_block(blockIdx.x, gridDim.x) * 256u + threadIdx.x;
  bool t_shadow = false, t_hit = false, started = false, err = false, still = false;
  if (slot < w.P) {
    const float4 ro = w.ray_o[slot];
    uint32_t sflags = __float_as_uint(ro.w);
    if (sflags & kSlotActive) {
      const SceneView& sv = p.sv;
      const float4* sb = sv.blob;
      const float4 rd = w.ray_d[slot], hv = w.hit[slot];
      const float4 sa = w.st_a[slot], sbv = w.st_b[slot], sc = w.st_c[slot];
      f3 org = F3(ro.x, ro.y, ro.z), dir = F3(rd.x, rd.y, rd.z);
      uint32_t k = __float_as_uint(rd.w);
      f3 xpos = F3(sa.x, sa.y, sa.z); float bs_density = sa.w;
      f3 tnum = F3(sbv.x, sbv.y, sbv.z); uint32_t flags = __float_as_uint(sbv.w);
      f3 radiance = F3(sc.x, sc.y, sc.z);
      const bool bounce = (flags & kFlagBounce) != 0u; bool bs_finite = (flags & kFlagFinite) != 0u;
      uint32_t path_size = flags >> 2;
      Rng rng = rng_unpack(w.rng[slot]);
      uint2 cnt = w.cnt[slot];
      ++cnt.x;
      Hit h; h.t = hv.x; h.u = hv.y; h.v = hv.z; h.pos = __float_as_uint(hv.w); h.id = h.pos == 0xFFFFFFFFu ? 0xFFFFFFFFu : 0u;
      t_hit = h.id != 0xFFFFFFFFu;

      bool terminate = false, do_vertex = false, new_bounce = bounce;
      Surf sp;
      f3 x_throughput = F3(1, 1, 1);
      if (h.id == 0xFFFFFFFFu) {
        terminate = true;  // PT.cpp:28,49-51: a miss ends the path (PT ignores the sky)
      } else {
        sp = query_surface(sb, sv, org, dir, h);
        const bool is_light = surf_is_light(sp);
        if (!bounce) {
          if (is_light && p.max_path > 0u) {  // PT.cpp:23-26: directly visible light, continue through it
            const Material lm = load_material(sb, sv, sp.material_id);
            f3 le; float dens;
            query_lsdf(sb, sv, nullptr, lm.light_id, -dir, le, dens);
            radiance = radiance + le * p.lights;
            org = nudge(sp.position, sp.gnormal, dir);
          } else if (p.max_path < 2u) {
            terminate = true;  // PT.cpp:28-30
          } else {
            path_size = 2u; do_vertex = true;  // PT.cpp:32-38
          }
        } else {
          // new vertex z = hit (PT.cpp:53-68); Edge(eye[prv], eye[itr], -dir)
          const f3 omega = -dir;
          const f3 d = xpos - sp.position;
          const float distSqInv = 1.0f / dot(d, d);
          const float fCos = fabsf(dot(omega, sp.tangent.c1));
          const float fG = distSqInv * fCos;
          if (l1norm(tnum) < MI_FLT_EPSILON) {
            terminate = true;  // PT.cpp:62-64
          } else {
            const f3 ztp = tnum / bs_density;  // PT.cpp:66
            if (is_light) {  // PT.cpp:70-79: MIS-weighted emission, then continue through the light
              const Material lm = load_material(sb, sv, sp.material_id);
              f3 le; float dens;
              query_lsdf(sb, sv, nullptr, lm.light_id, omega, le, dens);
              float wInv = powb(dens, p.beta) / powb(fG * bs_density, p.beta) + 1.0f;
              if (!bs_finite) wInv = 1.0f;
              radiance = radiance + (le * ztp) / wInv;
              org = nudge(sp.position, sp.gnormal, dir);
            } else {
              // Russian roulette (PT.cpp:86-94)
              const float roul = path_size < p.min_subpath ? 1.0f : p.roulette;
              const float uu = rng_f(rng);
              if (roul < uu) {
                terminate = true;
              } else {
                x_throughput = ztp / roul;
                const uint32_t before = path_size;
                if (path_size != kPathSizeMax) ++path_size;
                if (before + 1u > p.max_path) terminate = true; else do_vertex = true;  // PT.cpp:40
              }
            }
          }
        }
      }

      sflags = kSlotActive;
      if (do_vertex) {
        // ---- vertex x = sp: NEE (PT.cpp:41) then BSDF sample (PT.cpp:43-44) ----
        const Material mat = load_material(sb, sv, sp.material_id);
        const f3 x_omega = -dir;
        bool pending = false; ShadowRay sray; sray.org = F3(0, 0, 0); sray.dir = F3(0, 0, 1);
        const f3 nee = connect_prepare(sb, sv, nullptr, rng, mat, sp, x_omega, x_throughput, p.beta, pending, sray);
        if (pending) { t_shadow = true; ++cnt.y; }
        // a contribution that is exactly zero cannot change the sum whatever the visibility: counted, not traversed
        if (pending && !(nee.x != 0.0f || nee.y != 0.0f || nee.z != 0.0f)) pending = false;
        if (pending) {
          w.sh_o[slot] = make_float4(sray.org.x, sray.org.y, sray.org.z, nee.x);
          w.sh_d[slot] = make_float4(sray.dir.x, sray.dir.y, sray.dir.z, nee.y);
          w.sh_z[slot] = nee.z;
          sflags |= kSlotShadow;
        }
        const f3 x_position = sp.position, x_gnormal = sp.gnormal;
        const BSample bs = bsdf_sample(mat, rng, sp, x_omega);
        const float bCos = fabsf(dot(-bs.omega, sp.tangent.c1));  // Edge::bCosTheta with omega = -bsdf.omega
        tnum = (x_throughput * bs.q.throughput) * bCos;
        bs_density = bs.q.density; bs_finite = bs.q.finite != 0;
        xpos = x_position;
        dir = bs.omega;
        org = nudge(x_position, x_gnormal, dir);
        new_bounce = true;
      }

      if (terminate) {  // never together with a pending shadow ray (do_vertex and terminate exclude each other)
        slot_next_path(p, w, slot, true, radiance, cnt, k, started, err, sflags, org, dir, rng);
        radiance = F3(0, 0, 0); cnt = make_uint2(0u, 0u);
        xpos = F3(0, 0, 0); bs_density = 1.0f; tnum = F3(0, 0, 0); new_bounce = false; bs_finite = true; path_size = 0u;
      }
      still = (sflags & kSlotActive) != 0u;
      w.rng[slot] = rng_pack(rng);
      w.ray_o[slot] = make_float4(org.x, org.y, org.z, __uint_as_float(sflags));
      w.ray_d[slot] = make_float4(dir.x, dir.y, dir.z, __uint_as_float(k));
      w.st_a[slot] = make_float4(xpos.x, xpos.y, xpos.z, bs_density);
      w.st_b[slot] = make_float4(tnum.x, tnum.y, tnum.z, __uint_as_float((new_bounce ? kFlagBounce : 0u) | (bs_finite ? kFlagFinite : 0u) | (path_size << 2)));
      w.st_c[slot] = make_float4(radiance.x, radiance.y, radiance.z, 0.f);
      w.cnt[slot] = cnt;
    }
  }
  count_add(&p.counters[1], t_shadow);
  count_add(&p.counters[8], t_hit);
  count_add(&p.counters[2], err);
  count_add(&p.counters[3], started);
  count_add(reinterpret_cast<unsigned long long*>(w.n_active), still);
}

// ---- Scene::occluded (Scene.cpp:151-180) for slots with a pending shadow ray; PT.cpp:41: radiance += _connect(...) ----
template <bool COUNT>
__global__ __launch_bounds__(kBlock, 8) void wf_shadow(const RenderParams p, const WfState w) {
  extern __shared__ float4 smem[];
  TravStack stack;
  stack.lds = (lds_u32*)(reinterpret_cast<uint32_t*>(smem) + threadIdx.x);
  stack.cap = p.stack_entries;
  const uint32_t slot = xcd_block(blockIdx.x, gridDim.x) * kBlock + threadIdx.x;
  Visits vis = {0u, 0u, nullptr};
  if (slot < w.P) {
    const float4 ro = w.ray_o[slot];
    const uint32_t sflags = __float_as_uint(ro.w);
    if (sflags & kSlotShadow) {
      const float4 o = w.sh_o[slot], d = w.sh_d[slot];
      const f3 nee = F3(o.w, d.w, w.sh_z[slot]);
      Hit sh; sh.t = 1.0f; sh.u = sh.v = 0.0f; sh.id = 0xFFFFFFFFu; sh.pos = 0;
      traverse<true, COUNT, MI_WF_QUANT>(p.sv.blob, p.sv, stack, F3(o.x, o.y, o.z), F3(d.x, d.y, d.z), 1u << MI_ENTITY_MESH, sh, &vis);
      const float4 c = w.st_c[slot];
      const f3 r = F3(c.x, c.y, c.z) + nee * (sh.id != 0xFFFFFFFFu ? 0.f : 1.f);
      w.st_c[slot] = make_float4(r.x, r.y, r.z, c.w);
      w.ray_o[slot] = make_float4(ro.x, ro.y, ro.z, __uint_as_float(sflags & ~kSlotShadow));
    }
  }
  if (COUNT) {
    const uint32_t a = wf_wave_sum(vis.nodes), b = wf_wave_sum(vis.tris);
    if ((threadIdx.x & 63u) == 0 && (a | b)) { atomicAdd(&p.counters[6], (unsigned long long)a); atomicAdd(&p.counters[7], (unsigned long long)b); }
  }
}

// ---- per pixel: the replicas' FP64 sums in replica order into the partial (Technique.cpp:338, _commit_images) ----
__global__ __launch_bounds__(256) void wf_reduce(const RenderParams p, const WfState w) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= w.per_sample) return;
  const uint32_t tile = i >> 6, pix = i & 63u;
  uint32_t tx0, ty0;
  tile_origin(p, tile, tx0, ty0);
  const uint32_t px = tx0 + (pix & 7u), py = ty0 + (pix >> 3);
  if (px >= p.win_x0 + p.win_w || py >= p.win_y0 + p.win_h) return;
  double r = 0.0, g = 0.0, b = 0.0, n = 0.0;
  for (uint32_t k = 0; k < w.R; ++k) {
    const double4 a = w.acc[size_t(k) * w.per_sample + i];
    r += a.x; g += a.y; b += a.z; n += a.w;
  }
  double* o = p.partial + (size_t(py) * p.width + px) * 4;
  o[0] = r; o[1] = g; o[2] = b; o[3] = n;
}

// ---- host driver ----
#define WF_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

hipError_t wf_run(const RenderParams& p, const WfState& w, bool count, hipStream_t stream, uint32_t* rounds_out) {
  const uint32_t gP = (w.P + 255u) / 256u;
  const size_t lds = size_t(p.stack_entries) * kBlock * 4;
  WF_CHECK(hipMemsetAsync(w.n_active, 0, sizeof(unsigned long long), stream));
  hipLaunchKernelGGL(wf_init, dim3(gP), dim3(256), 0, stream, p, w);
  uint32_t it = 0;
  for (;;) {
    if (count) hipLaunchKernelGGL(wf_extend<true>, dim3(gP), dim3(kBlock), lds, stream, p, w);
    else hipLaunchKernelGGL(wf_extend<false>, dim3(gP), dim3(kBlock), lds, stream, p, w);
    const bool look = (it & 3u) == 3u;  // the host looks at the number of active slots every fourth round
    if (look) WF_CHECK(hipMemsetAsync(w.n_active, 0, sizeof(unsigned long long), stream));
    hipLaunchKernelGGL(wf_shade, dim3(gP), dim3(256), 0, stream, p, w);
    if (count) hipLaunchKernelGGL(wf_shadow<true>, dim3(gP), dim3(kBlock), lds, stream, p, w);
    else hipLaunchKernelGGL(wf_shadow<false>, dim3(gP), dim3(kBlock), lds, stream, p, w);
    ++it;
    if (look) {
      unsigned long long left = 0;
      WF_CHECK(hipMemcpyAsync(&left, w.n_active, sizeof left, hipMemcpyDeviceToHost, stream));
      WF_CHECK(hipStreamSynchronize(stream));
      if (left == 0ull) break;
    }
    if (it > (1u << 26)) return hipErrorLaunchTimeOut;
  }
  WF_CHECK(hipGetLastError());
  if (!w.list) hipLaunchKernelGGL(wf_reduce, dim3((w.per_sample + 255u) / 256u), dim3(256), 0, stream, p, w);
  if (rounds_out) *rounds_out = it;
  return hipGetLastError();
}

}  // namespace mi
