// wf_kernels.hip — wavefront form of the PT integrator for scenes that live in HBM (gfx950).
//
// The megakernel keeps a path in one lane from camera to termination; its wave pays for the slowest
// lane of every traversal and carries ~100 registers of shading state through the BVH loops.  Here the same
// per-path arithmetic (PathTracing::_traceEye, PT.cpp:15-98 — identical device functions, identical RNG
// draw order, so results are bit-identical per path) is cut at the two ray casts:
//
//   wf_regen    finished / empty slots: commit the path's radiance, take the next (pixel, sample) of the work
//               pool, shoot the camera ray (Technique.cpp:321-331)                         -> closest queue
//   wf_extend   Scene::intersect's traversal for every slot of the closest queue            -> hit record
//   wf_shade    querySurface, light pass-through / MIS emission, roulette, NEE set-up, BSDF sample
//               (PT.cpp:20-94)                                  -> shadow queue, closest queue or finished queue
//   wf_shadow   Scene::occluded's traversal for the shadow queue; radiance += nee * visibility (PT.cpp:41)
//   wf_reduce   per pixel: the batch's samples summed in sample order into the FP64 partial (Technique.cpp:338)
//
// Path state is SoA in HBM (156 B per slot, ~400 B of traffic per segment); queues are index lists filled
// with wave-aggregated atomics.  Traversal kernels keep only ray state in registers, so they run at high
// occupancy, which is what hides the L2 / Infinity-Cache latency of incoherent node fetches.
#include <hip/hip_runtime.h>

#include "pt_device.h"
#include "wavefront.h"

namespace mi {

namespace {

MI_DEV uint32_t lane_rank(uint64_t mask) { return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u)); }

// slot index appended to a queue; one atomic per wave
MI_DEV void queue_push(uint32_t* __restrict__ q, uint32_t* __restrict__ counter, bool pred, uint32_t value) {
  const uint64_t m = __ballot(pred);
  if (m == 0ull) return;
  const int leader = __ffsll((long long)m) - 1;
  uint32_t base = 0;
  if (int(threadIdx.x & 63u) == leader) base = atomicAdd(counter, uint32_t(__popcll(m)));
  base = __shfl(base, leader, 64);
  if (pred) q[base + lane_rank(m)] = value;
}
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

// work item -> pixel / sample.  Image mode: item = s * (tiles * 64) + tile * 64 + pixel-in-tile, so 64 consecutive
// items are one 8x8 tile at one sample index (coherent camera rays in a wave).
struct Item { uint32_t px, py; uint64_t sample; bool ok; };
MI_DEV Item decode_item(const RenderParams& p, const WfState& w, uint64_t item) {
  Item it;
  if (w.list) {
    it.px = p.list_xy[2 * item]; it.py = p.list_xy[2 * item + 1]; it.sample = p.list_sample[item]; it.ok = true;
  } else {
    const uint32_t per_sample = p.tiles_x * p.tiles_y * 64u;
    const uint32_t s = uint32_t(item / per_sample), rem = uint32_t(item - uint64_t(s) * per_sample);
    const uint32_t tile = rem >> 6, pix = rem & 63u;
    const uint32_t ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    it.px = p.win_x0 + tx * 8u + (pix & 7u); it.py = p.win_y0 + ty * 8u + (pix >> 3);
    it.sample = p.sample_offset + w.batch_sample0 + s;
    it.ok = it.px < p.win_x0 + p.win_w && it.py < p.win_y0 + p.win_h;
  }
  return it;
}

}  // namespace

// ---- commit finished paths, refill their slots from the work pool ----
// init != 0: every slot is empty (first launch of a batch), nothing to commit.
__global__ __launch_bounds__(256) void wf_regen(const RenderParams p, const WfState w, uint32_t init, uint32_t next) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const uint32_t n = init ? w.P : w.n[3];
  if (blockIdx.x * 256u >= n) return;
  const bool active = i < n;
  uint32_t slot = 0;
  bool err = false;
  if (active) {
    slot = init ? i : w.qf[i];
    if (!init) {
      // _eye_image += radiance; finite filter of _commit_images (Technique.cpp:222-230,338)
      const float4 c = w.st_c[slot];
      const uint32_t item = __float_as_uint(c.w);
      if (w.list) {
        p.list_radiance[3 * size_t(item)] = c.x; p.list_radiance[3 * size_t(item) + 1] = c.y; p.list_radiance[3 * size_t(item) + 2] = c.z;
        if (p.list_counts) { const uint2 k = w.cnt[slot]; p.list_counts[2 * size_t(item)] = k.x; p.list_counts[2 * size_t(item) + 1] = k.y; }
      } else {
        const bool fin = isfinite(fabsf(c.x) + fabsf(c.y) + fabsf(c.z));
        w.results[item] = fin ? make_float4(c.x, c.y, c.z, 1.0f) : make_float4(0.f, 0.f, 0.f, 0.f);
        err = !fin;
      }
    }
  }
  count_add(&p.counters[2], err);
  // next work item (skipping pixels of edge tiles that lie outside the window)
  bool want = active, started = false;
  for (;;) {
    const uint64_t m = __ballot(want);
    if (m == 0ull) break;
    const int leader = __ffsll((long long)m) - 1;
    unsigned long long base = 0;
    if (int(threadIdx.x & 63u) == leader) base = atomicAdd(w.work_next, (unsigned long long)__popcll(m));
    base = __shfl(base, leader, 64);
    if (want) {
      const uint64_t item = base + lane_rank(m);
      if (item >= w.n_items) { want = false; }
      else {
        const Item it = decode_item(p, w, item);
        if (!it.ok) { w.results[item] = make_float4(0.f, 0.f, 0.f, 0.f); }  // outside the window: contributes nothing, try again
        else {
          // shoot() (Technique.cpp:321-331) + ray_direction (Cameras.cpp:120-127)
          const m33 v2w = {F3(p.v2w[0], p.v2w[1], p.v2w[2]), F3(p.v2w[3], p.v2w[4], p.v2w[5]), F3(p.v2w[6], p.v2w[7], p.v2w[8])};
          const f3 cam_pos = F3(p.cam_pos[0], p.cam_pos[1], p.cam_pos[2]);
          const f3 cam_gnormal = -v2w.c2;  // Technique::_camera_surface (Technique.cpp:107-116)
          Rng rng = rng_seed(p.seed, it.py * p.width + it.px, it.sample);
          const float u0 = rng_f(rng), u1 = rng_f(rng);
          const float fx = float(it.px) + u0, fy = float(it.py) + u1;
          const float vx = fx * p.res_y_inv * 2.0f - p.res_x * p.res_y_inv;
          const float vy = fy * p.res_y_inv * 2.0f - 1.0f;
          const f3 dir = mulmv(v2w, normalize(F3(vx, vy, -p.focal_length_y)));
          const f3 org = nudge(cam_pos, cam_gnormal, dir);
          w.rng[slot] = rng.state;
          w.ray_o[slot] = make_float4(org.x, org.y, org.z, 0.f);
          w.ray_d[slot] = make_float4(dir.x, dir.y, dir.z, 0.f);
          w.st_a[slot] = make_float4(0.f, 0.f, 0.f, 1.0f);
          w.st_b[slot] = make_float4(0.f, 0.f, 0.f, __uint_as_float(kFlagFinite));
          w.st_c[slot] = make_float4(0.f, 0.f, 0.f, __uint_as_float(uint32_t(item)));
          w.cnt[slot] = make_uint2(0u, 0u);
          want = false; started = true;
        }
      }
    }
  }
  count_add(&p.counters[3], started);
  queue_push(w.qc[next], &w.n[next], started, slot);
}

// ---- Scene::intersect (Scene.cpp:182-203): closest-hit traversal for the closest queue ----
template <bool COUNT>
__global__ __launch_bounds__(kBlock, 8) void wf_extend(const RenderParams p, const WfState w, uint32_t cur) {
  extern __shared__ float4 smem[];
  TravStack stack;
  stack.lds = (lds_u32*)(reinterpret_cast<uint32_t*>(smem) + threadIdx.x);
  stack.cap = p.stack_entries;
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  const uint32_t n = w.n[cur];
  if (blockIdx.x * kBlock >= n) return;
  Visits vis = {0u, 0u, nullptr};
  if (i < n) {
    const uint32_t slot = w.qc[cur][i];
    const float4 o = w.ray_o[slot], d = w.ray_d[slot];
    Hit h; h.t = __builtin_inff(); h.u = h.v = 0.0f; h.id = 0xFFFFFFFFu; h.pos = 0;
    traverse<false, COUNT, MI_WF_QUANT != 0>(p.sv.blob, p.sv, stack, F3(o.x, o.y, o.z), F3(d.x, d.y, d.z), 0xFFFFFFFFu, h, &vis);
    w.hit[slot] = make_float4(h.t, h.u, h.v, __uint_as_float(h.id == 0xFFFFFFFFu ? 0xFFFFFFFFu : h.pos));
  }
  count_add(&p.counters[0], i < n);  // every queued slot casts exactly one closest-hit ray
  if (COUNT) {
    const uint32_t a = wf_wave_sum(vis.nodes), b = wf_wave_sum(vis.tris);
    if ((threadIdx.x & 63u) == 0) { atomicAdd(&p.counters[4], (unsigned long long)a); atomicAdd(&p.counters[5], (unsigned long long)b); }
  }
}

// ---- the vertex: PT.cpp:20-94 between the two ray casts ----
__global__ __launch_bounds__(256) void wf_shade(const RenderParams p, const WfState w, uint32_t cur, uint32_t next) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const uint32_t n = w.n[cur];
  if (blockIdx.x * 256u >= n) return;
  const bool active = i < n;
  bool push_next = false, push_fin = false, push_shadow = false, t_shadow = false, t_hit = false;
  uint32_t slot = 0;
  if (active) {
    slot = w.qc[cur][i];
    const SceneView& sv = p.sv;
    const float4* sb = sv.blob;
    const float4 ro = w.ray_o[slot], rd = w.ray_d[slot], hv = w.hit[slot];
    const float4 sa = w.st_a[slot], sbv = w.st_b[slot], sc = w.st_c[slot];
    f3 org = F3(ro.x, ro.y, ro.z), dir = F3(rd.x, rd.y, rd.z);
    f3 xpos = F3(sa.x, sa.y, sa.z); float bs_density = sa.w;
    f3 tnum = F3(sbv.x, sbv.y, sbv.z); uint32_t flags = __float_as_uint(sbv.w);
    f3 radiance = F3(sc.x, sc.y, sc.z);
    const bool bounce = (flags & kFlagBounce) != 0u; bool bs_finite = (flags & kFlagFinite) != 0u;
    uint32_t path_size = flags >> 2;
    Rng rng; rng.state = w.rng[slot];
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
          query_lsdf(sb, sv, lm.light_id, -dir, le, dens);
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
            query_lsdf(sb, sv, lm.light_id, omega, le, dens);
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

    if (do_vertex) {
      // ---- vertex x = sp: NEE (PT.cpp:41) then BSDF sample (PT.cpp:43-44) ----
      const Material mat = load_material(sb, sv, sp.material_id);
      const f3 x_omega = -dir;
      bool pending = false; ShadowRay sray; sray.org = F3(0, 0, 0); sray.dir = F3(0, 0, 1);
      const f3 nee = connect_prepare(sb, sv, rng, mat, sp, x_omega, x_throughput, p.beta, pending, sray);
      if (pending) { t_shadow = true; ++cnt.y; }
      // a contribution that is exactly zero cannot change the sum whatever the visibility: counted, not traversed
      if (pending && !(nee.x != 0.0f || nee.y != 0.0f || nee.z != 0.0f)) pending = false;
      if (pending) {
        w.sh_o[slot] = make_float4(sray.org.x, sray.org.y, sray.org.z, nee.x);
        w.sh_d[slot] = make_float4(sray.dir.x, sray.dir.y, sray.dir.z, nee.y);
        w.sh_z[slot] = nee.z;
        push_shadow = true;
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

    // state back to HBM (radiance also for finished paths: wf_shadow may still add the last NEE term before the commit)
    w.st_c[slot] = make_float4(radiance.x, radiance.y, radiance.z, sc.w);
    w.cnt[slot] = cnt;
    if (terminate) {
      push_fin = true;
    } else {
      w.rng[slot] = rng.state;
      w.ray_o[slot] = make_float4(org.x, org.y, org.z, 0.f);
      w.ray_d[slot] = make_float4(dir.x, dir.y, dir.z, 0.f);
      w.st_a[slot] = make_float4(xpos.x, xpos.y, xpos.z, bs_density);
      w.st_b[slot] = make_float4(tnum.x, tnum.y, tnum.z, __uint_as_float((new_bounce ? kFlagBounce : 0u) | (bs_finite ? kFlagFinite : 0u) | (path_size << 2)));
      push_next = true;
    }
  }
  count_add(&p.counters[1], t_shadow);
  count_add(&p.counters[8], t_hit);
  queue_push(w.qs, &w.n[2], push_shadow, slot);
  queue_push(w.qf, &w.n[3], push_fin, slot);
  queue_push(w.qc[next], &w.n[next], push_next, slot);
}

// ---- Scene::occluded (Scene.cpp:151-180) for the shadow queue; PT.cpp:41: radiance += _connect(...) ----
template <bool COUNT>
__global__ __launch_bounds__(kBlock, 8) void wf_shadow(const RenderParams p, const WfState w) {
  extern __shared__ float4 smem[];
  TravStack stack;
  stack.lds = (lds_u32*)(reinterpret_cast<uint32_t*>(smem) + threadIdx.x);
  stack.cap = p.stack_entries;
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  const uint32_t n = w.n[2];
  if (blockIdx.x * kBlock >= n) return;
  Visits vis = {0u, 0u, nullptr};
  if (i < n) {
    const uint32_t slot = w.qs[i];
    const float4 o = w.sh_o[slot], d = w.sh_d[slot];
    const f3 nee = F3(o.w, d.w, w.sh_z[slot]);
    Hit sh; sh.t = 1.0f; sh.u = sh.v = 0.0f; sh.id = 0xFFFFFFFFu; sh.pos = 0;
    traverse<true, COUNT, MI_WF_QUANT != 0>(p.sv.blob, p.sv, stack, F3(o.x, o.y, o.z), F3(d.x, d.y, d.z), 1u << MI_ENTITY_MESH, sh, &vis);
    const float4 c = w.st_c[slot];
    const f3 r = F3(c.x, c.y, c.z) + nee * (sh.id != 0xFFFFFFFFu ? 0.f : 1.f);
    w.st_c[slot] = make_float4(r.x, r.y, r.z, c.w);
  }
  if (COUNT) {
    const uint32_t a = wf_wave_sum(vis.nodes), b = wf_wave_sum(vis.tris);
    if ((threadIdx.x & 63u) == 0) { atomicAdd(&p.counters[6], (unsigned long long)a); atomicAdd(&p.counters[7], (unsigned long long)b); }
  }
}

// queue counters for the next iteration: the queue just consumed, the shadow queue and the finished queue are empty again
__global__ void wf_reset(const WfState w, uint32_t cur) { w.n[cur] = 0u; w.n[2] = 0u; w.n[3] = 0u; }

// ---- per pixel: this batch's samples in sample order into the FP64 partial (Technique.cpp:338, _commit_images) ----
__global__ __launch_bounds__(256) void wf_reduce(const RenderParams p, const WfState w, uint32_t batch_spp) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const uint32_t per_sample = p.tiles_x * p.tiles_y * 64u;
  if (i >= per_sample) return;
  const uint32_t tile = i >> 6, pix = i & 63u;
  const uint32_t ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
  const uint32_t px = p.win_x0 + tx * 8u + (pix & 7u), py = p.win_y0 + ty * 8u + (pix >> 3);
  if (px >= p.win_x0 + p.win_w || py >= p.win_y0 + p.win_h) return;
  double r = 0.0, g = 0.0, b = 0.0, n = 0.0;
  for (uint32_t s = 0; s < batch_spp; ++s) {
    const float4 v = w.results[size_t(s) * per_sample + i];
    if (v.w != 0.0f) { r += double(v.x); g += double(v.y); b += double(v.z); n += 1.0; }
  }
  double* o = p.partial + (size_t(py) * p.width + px) * 4;
  o[0] += r; o[1] += g; o[2] += b; o[3] += n;
}

// ---- host driver ----
#define WF_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

hipError_t wf_run_batch(const RenderParams& p, const WfState& w, bool count, uint32_t batch_spp, hipStream_t stream, uint32_t* iterations_out) {
  const uint32_t gP = (w.P + 255u) / 256u;
  const size_t lds = size_t(p.stack_entries) * kBlock * 4;
  WF_CHECK(hipMemsetAsync(w.n, 0, 4 * sizeof(uint32_t), stream));
  WF_CHECK(hipMemsetAsync(w.work_next, 0, sizeof(unsigned long long), stream));
  hipLaunchKernelGGL(wf_regen, dim3(gP), dim3(256), 0, stream, p, w, 1u, 0u);
  uint32_t cur = 0, it = 0;
  for (;;) {
    const uint32_t next = cur ^ 1u;
    if (count) hipLaunchKernelGGL(wf_extend<true>, dim3(gP), dim3(kBlock), lds, stream, p, w, cur);
    else hipLaunchKernelGGL(wf_extend<false>, dim3(gP), dim3(kBlock), lds, stream, p, w, cur);
    hipLaunchKernelGGL(wf_shade, dim3(gP), dim3(256), 0, stream, p, w, cur, next);
    if (count) hipLaunchKernelGGL(wf_shadow<true>, dim3(gP), dim3(kBlock), lds, stream, p, w);
    else hipLaunchKernelGGL(wf_shadow<false>, dim3(gP), dim3(kBlock), lds, stream, p, w);
    hipLaunchKernelGGL(wf_regen, dim3(gP), dim3(256), 0, stream, p, w, 0u, next);
    hipLaunchKernelGGL(wf_reset, dim3(1), dim3(1), 0, stream, w, cur);
    cur = next; ++it;
    if ((it & 3u) == 0u || w.n_items <= w.P) {  // look at the queue every few rounds (a small batch drains quickly)
      uint32_t left = 0;
      WF_CHECK(hipMemcpyAsync(&left, w.n + cur, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
      WF_CHECK(hipStreamSynchronize(stream));
      if (left == 0u) break;  // nothing to extend: every slot is empty and the pool is drained (regen refills while work remains)
    }
    if (it > (1u << 26)) return hipErrorLaunchTimeOut;
  }
  WF_CHECK(hipGetLastError());
  if (!w.list) {
    const uint32_t per_sample = p.tiles_x * p.tiles_y * 64u;
    hipLaunchKernelGGL(wf_reduce, dim3((per_sample + 255u) / 256u), dim3(256), 0, stream, p, w, batch_spp);
  }
  if (iterations_out) *iterations_out += it;
  return hipGetLastError();
}

}  // namespace mi
