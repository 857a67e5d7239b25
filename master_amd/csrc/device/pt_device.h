// pt_device.h — device functions of the PT hot path (gfx950): BVH traversal, surface query,
// BSDFs, light sampling, next-event estimation.  Every function names the reference code it
// computes; the expression structure follows DESIGN.md's arithmetic contract.
#pragma once
#include <hip/hip_runtime.h>

#include "../../../include/mi_pt.h"
#include "layout.h"
#include "rng.h"
#include "vecmath.h"

namespace mi {

// Scene features a kernel variant is compiled for (RenderParams::features): a scene without Phong lobes, mirrors / glass and with beta in {1, 2}
// runs the variant that has none of that code (C2: 10 860 -> 11 135 Msamples/s, 172 B/lane less scratch).
constexpr int kFeatPhong = 1, kFeatDelta = 2, kFeatPow = 4, kFeatLights = 8, kFeatAll = 15;  // kFeatLights: more (or fewer) than one light




// id = global triangle index, pos = Morton position.  While a closest-hit traversal runs, u and v hold Embree's UNdivided U and V and `den` their
// divisor |den|; finish_hit() divides once per ray when the walk is over (the same two correctly rounded divisions the per-hit form made for every
// accepted candidate — two IEEE divisions, ~24 instructions, taken out of a leaf body that runs for 2.8 lanes of 64 on average).
struct Hit { float t, u, v; uint32_t id, pos; float den; };
__device__ __forceinline__ void finish_hit(Hit& h) {
  if (h.id != 0xFFFFFFFFu) { h.u = mi_div(h.u, h.den); h.v = mi_div(h.v, h.den); }
}

// SurfacePoint (SurfacePoint.hpp:37-63)
struct Surf { f3 position, gnormal; m33 tangent; uint32_t material_id; };

MI_DEV bool surf_is_light(const Surf& s) { return (s.material_id & 3u) == MI_ENTITY_LIGHT; }
MI_DEV f3 to_world(const Surf& s, f3 v) { return mulmv(s.tangent, v); }
MI_DEV f3 to_surface(const Surf& s, f3 v) { return mulvm(v, s.tangent); }
MI_DEV f3 xyz(float4 v) { return F3(v.x, v.y, v.z); }
MI_DEV float4 as_f4(uint4 v) { return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)); }

// ---------------------------------------------------------------------------------------------
// Ray / triangle: Embree 2 single-ray Moeller–Trumbore (rtcIntersect / rtcOccluded at
// Scene.cpp:175,198), mask test of Scene.cpp:42.  ANY = rtcOccluded (t in (0, h.t]).
// Closest-hit ties: smaller t, then smaller global triangle id (order independent).
// MASKED = false: the ray's mask is 0xFFFFFFFF (every geometry mask is one non-zero bit): no test.
template <bool ANY, bool MASKED = true>
MI_DEV bool tri_test(const float4* __restrict__ tris, uint32_t pos, f3 org, f3 dir, uint32_t ray_mask, Hit& h) {
  const float4 a = tris[3 * pos], b = tris[3 * pos + 1], c = tris[3 * pos + 2];
  const uint32_t mask = __float_as_uint(c.z);
  if (MASKED && !(mask & ray_mask)) return false;
  const f3 v0 = F3(a.x, a.y, a.z), e1 = F3(a.w, b.x, b.y), e2 = F3(b.z, b.w, c.x);
  const uint32_t id = __float_as_uint(c.y);
  const f3 ng = cross(e2, e1);
  const f3 C = v0 - org;
  const f3 R = cross(C, dir);
  const float den = dot(ng, dir);
  const float absden = fabsf(den);
  const float sgn = den < 0.0f ? -1.0f : 1.0f;
  const float U = dot(R, e2) * sgn;
  const float V = dot(R, e1) * sgn;
  if (den == 0.0f) return false;
  if (!(U >= 0.0f) || !(V >= 0.0f) || !(U + V <= absden)) return false;
  const float T = dot(ng, C) * sgn;
  if (!(absden * 0.0f < T)) return false;
  if (ANY) {  // rtcOccluded on the segment of Scene::occluded (Scene.cpp:165-175: tfar = 1, exactly): RN(T / |den|) <= 1 <=> T <= |den| for floats — no division
    if (T <= absden) { h.id = id; return true; }
    return false;
  }
  const float t = mi_div(T, absden);
  if (t < h.t || (t == h.t && id < h.id)) {
    h.t = t; h.u = U; h.v = V; h.den = absden; h.id = id; h.pos = pos;
    return true;
  }
  return false;
}

// LDS copy of the scene blob with padded records (nodes 64 -> 80 B, shading records 128 -> 144 B: power-of-two strides put the same
// field of every record into the same few banks).  Every thread of the workgroup calls this; `sv` is rewritten to the LDS layout
// (strides 5 / 9 float4: traverse<.., NS = 5>, query_surface<9>).  The caller synchronises the workgroup afterwards.
MI_DEV uint32_t lds_scene_f4(const SceneView& sv) { return sv.blob_f4 + sv.n_nodes + sv.n_tris; }
MI_DEV void stage_scene_to_lds(float4* __restrict__ smem, SceneView& sv, uint32_t tid) {
  const uint32_t o_tris = sv.n_nodes * 5u, o_shade = o_tris + sv.n_tris * 3u, o_rest = o_shade + sv.n_tris * 9u;
  // nodes: each child box becomes centre + half extent (ce_box_test below: three fmas per axis, no min / max); the half extent covers the box from the
  // rounded centre, + 2^-20 relative + sv.box_pad (2^-20 of the largest coordinate of the scene and its cameras: the roundings of the test for rays
  // that start within those bounds — every path ray does).  The links keep their places (n0.w, n1.w); parent / reserved ride along in n2.w, n3.w.
  for (uint32_t i = tid; i < sv.n_nodes * 2u; i += kBlock) {
    const uint32_t n = i >> 1, c = i & 1u;
    const float4 lo = sv.blob[sv.off_nodes + 4u * n + 2u * c], hi = sv.blob[sv.off_nodes + 4u * n + 2u * c + 1u];
    const f3 ctr = (xyz(lo) + xyz(hi)) * 0.5f;
    const f3 up = xyz(hi) - ctr, dn = ctr - xyz(lo);
    const f3 ext = F3(fmaf(fmaxf(up.x, dn.x), 1.000001f, sv.box_pad), fmaf(fmaxf(up.y, dn.y), 1.000001f, sv.box_pad), fmaf(fmaxf(up.z, dn.z), 1.000001f, sv.box_pad));
    smem[n * 5u + 2u * c] = make_float4(ctr.x, ctr.y, ctr.z, lo.w);
    smem[n * 5u + 2u * c + 1u] = make_float4(ext.x, ext.y, ext.z, hi.w);
  }
  for (uint32_t i = tid; i < sv.n_tris * 3u; i += kBlock) smem[o_tris + i] = sv.blob[sv.off_tris + i];
  for (uint32_t i = tid; i < sv.n_tris * 8u; i += kBlock) smem[o_shade + (i >> 3) * 9u + (i & 7u)] = sv.blob[sv.off_shade + i];
  for (uint32_t i = tid; i < sv.blob_f4 - sv.off_mats; i += kBlock) smem[o_rest + i] = sv.blob[sv.off_mats + i];
  sv.off_lights = o_rest + (sv.off_lights - sv.off_mats); sv.off_cdf = o_rest + (sv.off_cdf - sv.off_mats);
  sv.off_nodes = 0u; sv.off_tris = o_tris; sv.off_shade = o_shade; sv.off_mats = o_rest;
}

// Slab test.  Not part of the bit-exact contract: it only has to be conservative (leaf boxes are
// padded, the interval is widened), because the closest hit is chosen by (t, id) and does not
// depend on which boxes were opened.  t = lo * inv - org * inv is one fma per plane; its absolute
// error is <= 2^-24 * |org * inv| per axis (plus 1 ulp of v_rcp_f32), covered by the per-ray slack.
struct RayBox { f3 inv, oi; float slack; };

MI_DEV RayBox make_raybox(f3 org, f3 dir) {
  RayBox r;
  const float big = 1e18f;  // direction component == 0: a finite stand-in keeps 0 * inf out of the slabs
  // v_rcp_f32(+-0) = +-inf keeps the sign of the zero, so the median with +-big is the round-3 "fabsf(i) < big ? i : copysignf(big, d)" in one instruction
  const float ix = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(dir.x), -big, big), iy = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(dir.y), -big, big),
              iz = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(dir.z), -big, big);
  r.inv = F3(ix, iy, iz);
  r.oi = org * r.inv;
  r.slack = (fabsf(r.oi.x) + fabsf(r.oi.y) + fabsf(r.oi.z)) * 2.5e-7f;
  return r;
}

// centre / half-extent form for the node copies: in LDS (stage_scene_to_lds) the padding of the half extent replaces the slack term; the copy read
// from HBM (sv.ce_nodes, SLACK) pads each box by 2^-21 of its own coordinates only — a scene whose far-away lights stretch its box (MetalRings: 400
// units around 0.1-unit triangles) must not pay for them in every leaf box — and keeps the per-ray slack for the origin's share of the rounding.
template <bool SLACK = false>
MI_DEV bool ce_box_test(f3 c, f3 e, const RayBox& rb, float tmax, float& tnear) {
  const float mx = fmaf(c.x, rb.inv.x, -rb.oi.x), my = fmaf(c.y, rb.inv.y, -rb.oi.y), mz = fmaf(c.z, rb.inv.z, -rb.oi.z);
  const float tn = fmaxf(fmaxf(fmaxf(fmaf(-e.x, fabsf(rb.inv.x), mx), fmaf(-e.y, fabsf(rb.inv.y), my)), fmaf(-e.z, fabsf(rb.inv.z), mz)), 0.0f);
  const float tf = fminf(fminf(fminf(fmaf(e.x, fabsf(rb.inv.x), mx), fmaf(e.y, fabsf(rb.inv.y), my)), fmaf(e.z, fabsf(rb.inv.z), mz)), tmax);
  tnear = tn;
  return SLACK ? tn <= fmaf(tf, 1.000002f, rb.slack) : tn <= tf;
}

MI_DEV bool box_test(f3 lo, f3 hi, const RayBox& rb, float tmax, float& tnear) {
  const float t0x = fmaf(lo.x, rb.inv.x, -rb.oi.x), t1x = fmaf(hi.x, rb.inv.x, -rb.oi.x);
  const float t0y = fmaf(lo.y, rb.inv.y, -rb.oi.y), t1y = fmaf(hi.y, rb.inv.y, -rb.oi.y);
  const float t0z = fmaf(lo.z, rb.inv.z, -rb.oi.z), t1z = fmaf(hi.z, rb.inv.z, -rb.oi.z);
  const float tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), 0.0f));
  const float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tmax));
  tnear = tn;
  return tn <= fmaf(tf, 1.000002f, rb.slack);
}

// A child of a wide quantised record (r03): centre and half extent on the 16-bit grid (x = cx | cy << 16, y = cz | ex << 16, z = ey | ez << 16; the
// half extent rounded up over the quantised box, + 1 cell).  Per axis m = c * inv - org * inv, tnear = m -/+ e * |inv| — three fmas (|inv| is an
// operand modifier) instead of two fmas, a min and a max: min / max / cmp issue at half the fma rate on gfx950, and the walk of the large scenes is
// bound by its box tests (profiles/r03/ab_wide8.txt).  No per-ray slack: the extra cell covers the roundings (coordinates <= 65535: every term
// below 0.04 cell).  Only has to be conservative; hits are the (t, id) minimum whatever is opened.
// FAR (the parity hooks, whose callers may pass a point anywhere): + the per-ray slack, which grows with |org * inv| and covers origins far outside the grid.
template <bool FAR = false>
MI_DEV bool wide_child_test(const uint4 a, const RayBox& rb, float tmax, float& tnear) {
  const float cx = float(a.x & 0xFFFFu), cy = float(a.x >> 16), cz = float(a.y & 0xFFFFu);
  const float ex = float(a.y >> 16), ey = float(a.z & 0xFFFFu), ez = float(a.z >> 16);
  const float mx = fmaf(cx, rb.inv.x, -rb.oi.x), my = fmaf(cy, rb.inv.y, -rb.oi.y), mz = fmaf(cz, rb.inv.z, -rb.oi.z);
  const float tn = fmaxf(fmaxf(fmaxf(fmaf(-ex, fabsf(rb.inv.x), mx), fmaf(-ey, fabsf(rb.inv.y), my)), fmaf(-ez, fabsf(rb.inv.z), mz)), 0.0f);
  const float tf = fminf(fminf(fminf(fmaf(ex, fabsf(rb.inv.x), mx), fmaf(ey, fabsf(rb.inv.y), my)), fmaf(ez, fabsf(rb.inv.z), mz)), tmax);
  tnear = tn;
  return FAR ? tn <= fmaf(tf, 1.000002f, rb.slack) : tn <= tf;
}

// Per-lane traversal stack: the first `cap` levels live in LDS (stack[level * kBlock + tid]: consecutive
// lanes hit consecutive banks); deeper levels — rare, traversal keeps few far children pending — go to a
// private (scratch) array, so LDS per workgroup stays at `cap` KB however deep the LBVH is.
#ifndef MI_STACK_SPILL
#define MI_STACK_SPILL 128
#endif
constexpr uint32_t kStackSpill = MI_STACK_SPILL;
// unused child slots of a wide node carry a box nothing enters and the link of leaf 0 (bvh_build.hip k_collapse4): no sentinel to test
#ifndef MI_WIDE_SORT_FULL
#define MI_WIDE_SORT_FULL 1  // 1: the four children fully sorted by entry distance; 0: the nearest first, the other three in the order of two pair swaps — two
                            // compare-exchanges fewer per visit, measured +-0.3 % on five scenes (what the order loses in visits the network saves): profiles/r04/ab_hbm_walk.txt
#endif
// The LDS part is addressed through an address-space-3 pointer: with a generic pointer the compiler cannot prove the
// target is LDS next to the private spill array and falls back to FLAT loads/stores with 64-bit address arithmetic
// on the pop -> next-node critical path; this way push/pop are ds_write_b32 / ds_read_b32.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
template <bool SPILL>
struct TravStackT {
  lds_u32* lds;
  uint32_t cap;
  uint32_t spill[kStackSpill];
  MI_DEV void push(int sp, uint32_t v) {
    if (uint32_t(sp) < cap) lds[sp * kBlock] = v; else spill[(uint32_t(sp) - cap) & (kStackSpill - 1)] = v;
  }
  MI_DEV uint32_t pop(int sp) const {
    return uint32_t(sp) < cap ? lds[sp * kBlock] : spill[(uint32_t(sp) - cap) & (kStackSpill - 1)];
  }
};
// the whole stack fits its LDS rows (host: depth - 1 <= 12, RenderParams::stack_in_lds): no bounds test on push / pop, no private array
template <>
struct TravStackT<false> {
  lds_u32* lds;
  uint32_t cap;
  MI_DEV void push(int sp, uint32_t v) { lds[sp * kBlock] = v; }
  MI_DEV uint32_t pop(int sp) const { return lds[sp * kBlock]; }
};
typedef TravStackT<true> TravStack;

// BVH2 traversal.  `sb` = scene blob base (LDS or HBM).
// Visit counters (nodes fetched, triangles tested) feed the roofline's algorithmic-bytes figure; they
// are only live in the instrumented kernel variant (COUNT) and compile away otherwise.
struct Visits { uint32_t nodes, tris; uint32_t* wave_iters; };  // wave_iters: LDS [2] per wave: node-loop bodies, leaf-phase bodies (instrumented)

// NS = node stride in float4 units: 4 in HBM; the LDS copy pads nodes to 5 (80 B) so that lanes reading the same field of
// different nodes spread over all 32 banks instead of 2 groups of 4 (64 B = 16 banks: every other node collides).
template <bool ANY, bool COUNT = false, int QUANT = 0, int NS = 4, bool MASKED = true, bool FAR = false, class Stack = TravStack>
MI_DEV void traverse_raw(const float4* __restrict__ sb, const SceneView& sv, Stack& stack, f3 org, f3 dir,
                         uint32_t ray_mask, Hit& h, Visits* vis) {
  // full-precision nodes read from HBM (NS == 4, scenes the 16-bit grid is too coarse for): the centre / half-extent copy of the tree (sv.ce_nodes)
  const float4* nodes = (QUANT == 0 && NS == 4) ? sv.ce_nodes : sb + sv.off_nodes;
  const float4* tris = sb + sv.off_tris;
  if (sv.n_nodes == 0) {
    if (COUNT) ++vis->tris;
    tri_test<ANY, MASKED>(tris, 0, org, dir, ray_mask, h);
    return;
  }
  // QUANT (HBM-resident scenes): boxes are read as 16-bit grid coordinates (32 B per node instead of 64).  The
  // ray is moved into grid space once — per axis (x - grid_lo) * cells_per_unit, which keeps the ray parameter t —
  // and the slab test runs on the integers converted to float.  Triangles are always tested in world space.
  const f3 glo = F3(sv.grid_lo[0], sv.grid_lo[1], sv.grid_lo[2]), gis = F3(sv.grid_inv_step[0], sv.grid_inv_step[1], sv.grid_inv_step[2]);
  const RayBox rb = QUANT ? make_raybox((org - glo) * gis, dir * gis) : make_raybox(org, dir);
  const uint4* __restrict__ qn = sv.qnodes;
  int sp = 0;
  int node = 0;
  if (QUANT == 2) {
    // ---- wide quantised nodes: one 64-byte record = the (up to) four grandchildren of an even-depth BVH2 node.  Half as
    // many dependent fetches per ray as the binary walk; the far children are pushed farthest first. ----
    const uint4* __restrict__ q4 = sv.qnodes4;
    for (;;) {
      if (node >= 0) {
        float t[4]; int l[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint4 a = q4[4 * node + k];
          float tn;
          const bool hk = wide_child_test<FAR>(a, rb, h.t, tn);
          t[k] = hk ? tn : __builtin_inff();
          l[k] = int(a.w);
        }
        if (COUNT) { ++vis->nodes; if (vis->wave_iters && __builtin_amdgcn_mbcnt_hi(uint32_t(__ballot(1) >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(__ballot(1)), 0u)) == 0) atomicAdd(&vis->wave_iters[0], 1u); }
#define MI_CSWAP(a, b) do { const bool s_ = t[b] < t[a]; const float ta_ = s_ ? t[b] : t[a], tb_ = s_ ? t[a] : t[b]; \
                            const int la_ = s_ ? l[b] : l[a], lb_ = s_ ? l[a] : l[b]; t[a] = ta_; t[b] = tb_; l[a] = la_; l[b] = lb_; } while (0)
        MI_CSWAP(0, 1); MI_CSWAP(2, 3); MI_CSWAP(0, 2);  // t[0] is the nearest; t[1] and t[3] keep the order of their pairs
        if (MI_WIDE_SORT_FULL) { MI_CSWAP(1, 3); MI_CSWAP(1, 2); }
#undef MI_CSWAP
        if (t[0] < __builtin_inff()) {
          if (t[3] < __builtin_inff()) { stack.push(sp, uint32_t(l[3])); ++sp; }
          if (t[2] < __builtin_inff()) { stack.push(sp, uint32_t(l[2])); ++sp; }
          if (t[1] < __builtin_inff()) { stack.push(sp, uint32_t(l[1])); ++sp; }
          node = l[0];
          continue;
        }
      } else {
        if (COUNT) { ++vis->tris; if (vis->wave_iters && __builtin_amdgcn_mbcnt_hi(uint32_t(__ballot(1) >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(__ballot(1)), 0u)) == 0) atomicAdd(&vis->wave_iters[1], 1u); }
        const uint32_t lp = uint32_t(~node), pos = lp & kLeafPosMask;
        const bool hit = tri_test<ANY, MASKED>(tris, pos, org, dir, ray_mask, h);
        if (ANY && hit) return;
        if (lp & kLeafPairBit) {  // pair leaf: the next triangle of the stream lies under the same box
          if (COUNT) ++vis->tris;
          const bool hit_b = tri_test<ANY, MASKED>(tris, pos + 1u, org, dir, ray_mask, h);
          if (ANY && hit_b) return;
        }
      }
      if (sp == 0) return;
      --sp;
      node = int(stack.pop(sp));
    }
  }
  for (;;) {
    if (node >= 0) {
      int l0, l1;
      float tn0, tn1;
      bool h0, h1;
      if (QUANT) {  // quantised binary node: two children as centre + half extent on the grid (wide_child_test)
        const uint4 a = qn[2 * node], b = qn[2 * node + 1];
        l0 = int(a.w); l1 = int(b.w);
        h0 = wide_child_test<FAR>(a, rb, h.t, tn0);
        h1 = wide_child_test<FAR>(b, rb, h.t, tn1);
      } else {
        const float4 n0 = nodes[NS * node], n1 = nodes[NS * node + 1], n2 = nodes[NS * node + 2], n3 = nodes[NS * node + 3];
        l0 = __float_as_int(n0.w); l1 = __float_as_int(n1.w);
        h0 = ce_box_test<NS == 4>(xyz(n0), xyz(n1), rb, h.t, tn0); h1 = ce_box_test<NS == 4>(xyz(n2), xyz(n3), rb, h.t, tn1);  // LDS copy or sv.ce_nodes: centre + half extent
      }
      if (COUNT) { ++vis->nodes; if (vis->wave_iters && __builtin_amdgcn_mbcnt_hi(uint32_t(__ballot(1) >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(__ballot(1)), 0u)) == 0) atomicAdd(&vis->wave_iters[0], 1u); }
      if (h0 && h1) {
        const bool sw = tn1 < tn0;
        stack.push(sp, uint32_t(sw ? l0 : l1));
        ++sp;
        node = sw ? l1 : l0;
        continue;
      }
      if (h0 || h1) {
        node = h0 ? l0 : l1;
        continue;
      }
    } else {
      if (COUNT) { ++vis->tris; if (vis->wave_iters && __builtin_amdgcn_mbcnt_hi(uint32_t(__ballot(1) >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(__ballot(1)), 0u)) == 0) atomicAdd(&vis->wave_iters[1], 1u); }
      const uint32_t lp = uint32_t(~node), pos = lp & kLeafPosMask;
      const bool hit = tri_test<ANY, MASKED>(tris, pos, org, dir, ray_mask, h);
      if (ANY && hit) return;
      if (lp & kLeafPairBit) {  // pair leaf: the next triangle of the stream lies under the same box
        if (COUNT) ++vis->tris;
        const bool hit_b = tri_test<ANY, MASKED>(tris, pos + 1u, org, dir, ray_mask, h);
        if (ANY && hit_b) return;
      }
    }
    if (sp == 0) return;
    --sp;
    node = int(stack.pop(sp));
  }
}

// ---------------------------------------------------------------------------------------------
// Unified traversal with dynamic fetch (LDS-resident scenes, full-precision binary nodes).
//
// In the plain loop trip every lane walks its own closest-hit ray, then — in a second loop — its own shadow ray: the wave pays the slowest
// lane of each loop while 62 % of the lanes have a shadow ray at all (C2: 36.7 + 19.8 node bodies per trip for 11.0 + 4.4 node visits per
// lane, profiles/r01/ab_postpone.txt).  Here the shadow ray of vertex k rides in the loop of trip k + 1 together with the closest-hit
// rays of that trip, and it need not be walked by its owner: a lane that has finished its own closest-hit ray takes the next unstarted
// shadow ray of the WAVE (rays are parked in LDS by their owners, results come back as one bit per owner in an LDS mask).  Closest-hit
// rays never migrate — a live lane always starts with its own — so the hit record stays in the owner's registers.
// Results do not depend on who walks a ray: the closest hit is the (t, id) minimum and occlusion is a boolean.
constexpr uint32_t kDynRayBytes = 6u * 64u * 4u;                  // org.xyz, dir.xyz per owner lane, [field][lane]
constexpr uint32_t kDynBytesPerWave = kDynRayBytes + 64u + 16u;  // + mailbox (owner lane per pool rank, one byte each) + occlusion mask
typedef __attribute__((address_space(3))) float lds_f32;

typedef __attribute__((address_space(3))) uint8_t lds_u8;
struct DynLds { lds_f32* ray; lds_u8* mailbox; lds_u32* occl; };  // this wave's block

MI_DEV void dyn_park_shadow_ray(const DynLds& d, uint32_t lane, f3 org, f3 dir) {
  d.ray[lane] = org.x; d.ray[64 + lane] = org.y; d.ray[128 + lane] = org.z;
  d.ray[192 + lane] = dir.x; d.ray[256 + lane] = dir.y; d.ray[320 + lane] = dir.z;
}

// alive: this lane has a closest-hit ray (org, dir) -> h.  pend: this lane parked a shadow ray (t in (0, 1], mesh mask) -> returns 1 if it is
// unoccluded (meaningful for pend lanes only).  TH: idle lanes that trigger a refill from the pool.
#ifdef MI_DYN_STATS
#define MI_DYN_STAT(k) do { ++dyn_stats[k]; } while (0)
#else
#define MI_DYN_STAT(k) do { } while (0)
#endif
// COUNT (instrumented variant): per-lane node / triangle visits by ray kind (whoever walks the ray) and the loop trips of the wave
template <int QUANT, int NS, int TH, bool COUNT = false, bool UNI_T = false, class Stack>
MI_DEV float traverse_dyn(const float4* __restrict__ sb, const SceneView& sv, Stack& stack, const DynLds& d, uint32_t lane, bool alive, f3 org, f3 dir, bool pend, Hit& h,
                          Visits* vis_c = nullptr, Visits* vis_s = nullptr, uint32_t* wave_trips = nullptr
#ifdef MI_DYN_STATS
                          , uint32_t* dyn_stats  // wave-uniform: [0] loop iterations, [1] node bodies, [2] leaf bodies, [3] refills, [4] rays fetched
#endif
) {
  const float4* nodes = (QUANT == 0 && NS == 4) ? sv.ce_nodes : sb + sv.off_nodes;  // traverse_raw
  const float4* tris = sb + sv.off_tris;
  const uint4* __restrict__ qn = sv.qnodes;
  const uint4* __restrict__ q4 = sv.qnodes4;
  // QUANT: the boxes are 16-bit grid coordinates; the ray is moved into grid space once per ray (traverse() above)
  const f3 glo = F3(sv.grid_lo[0], sv.grid_lo[1], sv.grid_lo[2]), gis = F3(sv.grid_inv_step[0], sv.grid_inv_step[1], sv.grid_inv_step[2]);
  uint64_t pool = __ballot(pend);  // shadow rays nobody has started
  if (lane < 2u) d.occl[lane] = 0u;
  uint32_t mode = alive ? 1u : 0u;  // 0 idle, 1 own closest-hit ray, 2 a shadow ray of `owner`
  uint32_t owner = lane;
  // Register diet (r03: the 6-wave budget spilled 18 dwords of path state around this loop, 100 B per lane and trip of scratch traffic through L2):
  // the walked ray itself is not kept — a leaf re-reads it (own ray: the caller's org / dir; a fetched shadow ray: its parked copy in LDS), the
  // far end is h.t or 1 by `mode`, and of a closest hit only (t, id, pos) live in the loop: U, V and |den| are formed again for the winner after it.
  RayBox rb = QUANT ? make_raybox((org - glo) * gis, dir * gis) : make_raybox(org, dir);
  h.pos = 0xFFFFFFFFu;  // nothing hit yet (h.id is formed after the loop)
  int sp = 0, node = 0;
  for (;;) {
    const uint64_t busy = __ballot(mode != 0u);
    if (pool != 0ull && (busy == 0ull || __popcll(~busy) >= TH)) {
      // ---- refill: the k-th idle lane takes the k-th shadow ray of the pool ----
      const uint64_t idle = ~busy;
      const uint32_t n_idle = uint32_t(__popcll(idle));
      const bool in_pool = (pool >> lane) & 1ull;
      const uint32_t rank_p = __builtin_amdgcn_mbcnt_hi(uint32_t(pool >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(pool), 0u));
      if (in_pool) d.mailbox[rank_p] = uint8_t(lane);
      const uint32_t n_pool = uint32_t(__popcll(pool));
      pool &= ~__ballot(in_pool && rank_p < n_idle);
      const uint32_t rank_i = __builtin_amdgcn_mbcnt_hi(uint32_t(idle >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(idle), 0u));
      MI_DYN_STAT(3);
#ifdef MI_DYN_STATS
      dyn_stats[4] += n_idle < n_pool ? n_idle : n_pool;
#endif
      if (mode == 0u && rank_i < n_pool) {
        owner = d.mailbox[rank_i];
        const f3 so = F3(d.ray[owner], d.ray[64 + owner], d.ray[128 + owner]);
        const f3 sd = F3(d.ray[192 + owner], d.ray[256 + owner], d.ray[320 + owner]);
        rb = QUANT ? make_raybox((so - glo) * gis, sd * gis) : make_raybox(so, sd);
        mode = 2u; sp = 0; node = 0;
      }
      continue;
    }
    if (busy == 0ull) break;
    if (COUNT) {
      ++*wave_trips;
      if (mode == 1u) { if (node >= 0) ++vis_c->nodes; else ++vis_c->tris; }
      if (mode == 2u) { if (node >= 0) ++vis_s->nodes; else ++vis_s->tris; }
    }
    MI_DYN_STAT(0);
#ifdef MI_DYN_STATS
    if (__ballot(mode != 0u && node >= 0)) ++dyn_stats[1];
    if (__ballot(mode != 0u && node < 0)) ++dyn_stats[2];
#endif
    if (mode != 0u) {
      bool pop = false;
      const float tmax = mode == 2u ? 1.0f : h.t;
      // r04 — ONE fetch per iteration whatever the lane is at.  An iteration of a wave nearly always has lanes at nodes AND lanes at leaves (75 % of them, r02
      // census), and the two branches below are serialised: with a load in each, the wave paid two dependent memory latencies per iteration.  Now every lane
      // reads ITS record before the branch — the 64-byte wide (or f32 binary) node, or the 48-byte triangle at the leaf — into the same sixteen registers, and the branches only compute.  (r02 requested BOTH record sets per lane and
      // lost 9-45 % to the registers; here the two kinds share one set.)  UNI: scenes read from HBM through 64-byte records.
      constexpr bool UNI = UNI_T && NS == 4 && (QUANT == 2 || QUANT == 0);  // a kernel variant of its own (as a run-time flag the two forms in one loop halved the throughput)
      uint4 u0 = make_uint4(0u, 0u, 0u, 0u), u1 = u0, u2 = u0, u3 = u0;
      if (UNI) {
        const uint4* __restrict__ rec = node >= 0 ? (QUANT == 2 ? q4 + 4 * node : reinterpret_cast<const uint4*>(nodes + 4 * node))
                                                   : reinterpret_cast<const uint4*>(tris + 3 * (uint32_t(~node) & kLeafPosMask));
        u0 = rec[0]; u1 = rec[1]; u2 = rec[2];
        if (node >= 0) u3 = rec[3];  // a triangle is 48 bytes: its lanes sit this load out (the 2 M-triangle scene is bound by 128-byte requests: -10 % with 64 bytes per triangle)
      }
      if (node >= 0) {
        if (QUANT == 2) {
          // wide quantised node: the (up to) four grandchildren, nearest first, the others pushed farthest first (traverse() above)
          float t[4]; int l[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint4 a = UNI ? (k == 0 ? u0 : k == 1 ? u1 : k == 2 ? u2 : u3) : q4[4 * node + k];
            float tn;
            const bool hk = wide_child_test(a, rb, tmax, tn);
            t[k] = hk ? tn : __builtin_inff();
            l[k] = int(a.w);
          }
#define MI_CSWAP(a, b) do { const bool s_ = t[b] < t[a]; const float ta_ = s_ ? t[b] : t[a], tb_ = s_ ? t[a] : t[b]; \
                            const int la_ = s_ ? l[b] : l[a], lb_ = s_ ? l[a] : l[b]; t[a] = ta_; t[b] = tb_; l[a] = la_; l[b] = lb_; } while (0)
          MI_CSWAP(0, 1); MI_CSWAP(2, 3); MI_CSWAP(0, 2);  // t[0] is the nearest; the rest is pushed in the order two pair swaps leave (r04: -2 of 5 compare-exchanges)
          if (MI_WIDE_SORT_FULL) { MI_CSWAP(1, 3); MI_CSWAP(1, 2); }
#undef MI_CSWAP
          if (t[0] < __builtin_inff()) {
            if (t[3] < __builtin_inff()) { stack.push(sp, uint32_t(l[3])); ++sp; }
            if (t[2] < __builtin_inff()) { stack.push(sp, uint32_t(l[2])); ++sp; }
            if (t[1] < __builtin_inff()) { stack.push(sp, uint32_t(l[1])); ++sp; }
            node = l[0];
          } else {
            pop = true;
          }
        } else {
          int l0, l1;
          float tn0, tn1;
          bool h0, h1;
          if (QUANT) {
            const uint4 a = qn[2 * node], b = qn[2 * node + 1];
            l0 = int(a.w); l1 = int(b.w);
            h0 = wide_child_test(a, rb, tmax, tn0);
            h1 = wide_child_test(b, rb, tmax, tn1);
          } else {
            const float4 n0 = UNI ? as_f4(u0) : nodes[NS * node], n1 = UNI ? as_f4(u1) : nodes[NS * node + 1], n2 = UNI ? as_f4(u2) : nodes[NS * node + 2],
                         n3 = UNI ? as_f4(u3) : nodes[NS * node + 3];
            l0 = __float_as_int(n0.w); l1 = __float_as_int(n1.w);
            h0 = ce_box_test<NS == 4>(xyz(n0), xyz(n1), rb, tmax, tn0); h1 = ce_box_test<NS == 4>(xyz(n2), xyz(n3), rb, tmax, tn1);  // LDS copy or sv.ce_nodes
          }
          if (h0 && h1) {
            const bool sw = tn1 < tn0;
            stack.push(sp, uint32_t(sw ? l0 : l1));
            ++sp;
            node = sw ? l1 : l0;
          } else if (h0 || h1) {
            node = h0 ? l0 : l1;
          } else {
            pop = true;
          }
        }
      } else {
        // Embree single-ray Moeller-Trumbore (tri_test above), both ray kinds: a closest-hit ray sees every geometry and keeps the (t, id)
        // minimum; a shadow ray sees mesh geometry only and ends at the first hit with t <= 1
        const uint32_t pos = uint32_t(~node) & kLeafPosMask;  // bit 30 of ~node: pair leaf, the triangle at pos + 1 is this lane's next iteration
        f3 co = org, cd = dir;
        if (mode == 2u) { co = F3(d.ray[owner], d.ray[64 + owner], d.ray[128 + owner]); cd = F3(d.ray[192 + owner], d.ray[256 + owner], d.ray[320 + owner]); }
        const float4 a = UNI ? as_f4(u0) : tris[3 * pos], b = UNI ? as_f4(u1) : tris[3 * pos + 1], c = UNI ? as_f4(u2) : tris[3 * pos + 2];
        const f3 v0 = F3(a.x, a.y, a.z), e1 = F3(a.w, b.x, b.y), e2 = F3(b.z, b.w, c.x);
        const uint32_t id = __float_as_uint(c.y), gmask = __float_as_uint(c.z);
        const f3 ng = cross(e2, e1);
        const f3 C = v0 - co;
        const f3 R = cross(C, cd);
        const float den = dot(ng, cd);
        const float absden = fabsf(den);
        const float sgn = den < 0.0f ? -1.0f : 1.0f;
        const float U = dot(R, e2) * sgn;
        const float V = dot(R, e1) * sgn;
        const float T = dot(ng, C) * sgn;
        const bool seen = mode == 1u || (gmask & (1u << MI_ENTITY_MESH)) != 0u;
        pop = true;
        if (seen && den != 0.0f && U >= 0.0f && V >= 0.0f && U + V <= absden && absden * 0.0f < T) {
          if (mode == 2u) {  // t = T / |den| <= 1 <=> T <= |den| (tri_test): the shadow rays of the loop need no division
            if (T <= absden) { atomicOr((unsigned int*)&d.occl[owner >> 5], 1u << (owner & 31u)); mode = 0u; pop = false; }
          } else {
            const float t = mi_div(T, absden);
            if (t < h.t || (t == h.t && (h.pos == 0xFFFFFFFFu || id < __float_as_uint(tris[3 * h.pos + 2].y)))) {
              // (t, id) minimum; on an exact tie (rare) the best hit's id is read back.  t == h.t == infinity with nothing hit yet (|den| so
              // small that T / |den| overflows) is a hit, as in tri_test: its id is smaller than the initial 0xFFFFFFFF
              h.t = t; h.pos = uint32_t(~node) & kLeafPosMask;
            }
          }
        }
        if ((uint32_t(node) & kLeafPairBit) == 0u && mode != 0u) { node = int((uint32_t(node) | kLeafPairBit) - 1u); pop = false; }  // ~(pos + 1), a single leaf
      }
      if (pop) {
        if (sp == 0) mode = 0u;
        else { --sp; node = int(stack.pop(sp)); }
      }
    }
  }
  if (alive && h.pos != 0xFFFFFFFFu) {  // id, U, V, |den| of the winner: the leaf's own expressions on the same operands, so the same bits
    const float4 a = tris[3 * h.pos], b = tris[3 * h.pos + 1], c = tris[3 * h.pos + 2];
    h.id = __float_as_uint(c.y);
    const f3 v0 = F3(a.x, a.y, a.z), e1 = F3(a.w, b.x, b.y), e2 = F3(b.z, b.w, c.x);
    const f3 ng = cross(e2, e1);
    const f3 R = cross(v0 - org, dir);
    const float den = dot(ng, dir);
    const float sgn = den < 0.0f ? -1.0f : 1.0f;
    h.u = dot(R, e2) * sgn; h.v = dot(R, e1) * sgn; h.den = fabsf(den);
  }
  else h.pos = 0u;
  finish_hit(h);
  const uint32_t word = d.occl[lane >> 5];
  return (word >> (lane & 31u)) & 1u ? 0.0f : 1.0f;
}

// ---------------------------------------------------------------------------------------------
// Flat leaf list (LDS-resident scenes of at most kFlatMaxLeaves leaf links; r03).
//
// A tree walk over a 32-triangle scene is all divergence: a node body serves a third of the wave, a leaf body three lanes of 64
// (profiles/r02/ab_c2_leaf_bodies.txt).  Here the wave runs ONE instruction stream over the K leaf boxes of the tree — the box of every
// leaf link, pair leaves counting once — with the boxes as SCALAR operands (s_load from the leaf table in global memory through the
// scalar cache: no LDS read, no bank conflict, no VGPR), every lane testing its own ray and keeping a K-bit mask of the boxes it enters.
// Real path segments of the Cornell box enter 1.6 of its 16 leaf boxes on average (tests/lab/flat_lab.py).  Then each lane tests the
// leaves of its mask, two triangles per leaf record, straight from LDS: no stack, no node records, no pop.  What is tested never
// changes the answer — the closest hit is the (t, id) minimum, occlusion a boolean — so everything stays bit-identical to the tree walk.
//
// Leaf table in global memory (scalar loads): K x { c.xyz, e.xyz, link, - } (32 B), leaves that hold a mesh triangle first (a shadow
// ray, mesh mask, tests only those: k_mesh), padded to a multiple of 4 entries.  LDS leaf record (7 float4): triangle A, triangle B, each { v0, e1, e2, ng = cross(e2, e1) }
// (ng is the per-test cross product of tri_test, formed once at staging by the same function), then idA, idB with the entity tag in bits
// 30-31.  A leaf of one triangle holds it twice (same id: never replaces itself).  Slot 2k + j also indexes the shading records.
typedef __attribute__((address_space(4))) const float cfloat;

MI_DEV uint32_t flat_scene_f4(const SceneView& sv, uint32_t K) { return kFlatLeafF4 * K + 18u * K + (sv.blob_f4 - sv.off_mats); }
MI_DEV void stage_scene_flat(float4* __restrict__ smem, SceneView& sv, const float* __restrict__ table, uint32_t K, uint32_t tid) {
  const uint32_t o_shade = kFlatLeafF4 * K, o_rest = o_shade + 18u * K;
  for (uint32_t s = tid; s < 2u * K; s += kBlock) {
    const uint32_t k = s >> 1, j = s & 1u, link = __float_as_uint(table[8u * k + 6u]);
    const uint32_t pos = (link & kLeafPosMask) + ((link & kLeafPairBit) ? j : 0u);
    const float4 a = sv.blob[sv.off_tris + 3u * pos], b = sv.blob[sv.off_tris + 3u * pos + 1u], c = sv.blob[sv.off_tris + 3u * pos + 2u];
    const f3 ng = cross(F3(b.z, b.w, c.x), F3(a.w, b.x, b.y));
    float4* rec = smem + kFlatLeafF4 * k + 3u * j;
    rec[0] = a; rec[1] = b; rec[2] = make_float4(c.x, ng.x, ng.y, ng.z);
    const uint32_t ent = uint32_t(__builtin_ctz(__float_as_uint(c.z) | 0x8u));
    reinterpret_cast<uint32_t*>(smem + kFlatLeafF4 * k + 6u)[j] = __float_as_uint(c.y) | (ent << 30);
    for (uint32_t q = 0; q < 8u; ++q) smem[o_shade + 9u * s + q] = sv.blob[sv.off_shade + 8u * pos + q];
  }
  for (uint32_t i = tid; i < sv.blob_f4 - sv.off_mats; i += kBlock) smem[o_rest + i] = sv.blob[sv.off_mats + i];
  sv.off_lights = o_rest + (sv.off_lights - sv.off_mats); sv.off_cdf = o_rest + (sv.off_cdf - sv.off_mats);
  sv.off_nodes = 0u; sv.off_tris = 0u; sv.off_shade = o_shade; sv.off_mats = o_rest;
}

// one triangle of a leaf record: tri_test with ng read instead of formed (same bits)
// MASKED (BPT's Scene::intersectMesh): a closest-hit ray that sees only the entities of ray_mask (bit = entity tag)
template <bool ANY, bool MASKED = false>
MI_DEV bool flat_tri(const float4 a, const float4 b, const float4 c, uint32_t idw, uint32_t slot, f3 org, f3 dir, Hit& h, uint32_t ray_mask = 0xFFFFFFFFu) {
  const f3 v0 = F3(a.x, a.y, a.z), e1 = F3(a.w, b.x, b.y), e2 = F3(b.z, b.w, c.x), ng = F3(c.y, c.z, c.w);
  const f3 C = v0 - org;
  const f3 R = cross(C, dir);
  const float den = dot(ng, dir);
  const float absden = fabsf(den);
  const float sgn = den < 0.0f ? -1.0f : 1.0f;
  const float U = dot(R, e2) * sgn;
  const float V = dot(R, e1) * sgn;
  const float T = dot(ng, C) * sgn;
  if (ANY && (idw >> 30) != uint32_t(MI_ENTITY_MESH)) return false;  // Scene.cpp:42,173: shadow rays see mesh geometry only
  if (MASKED && !((1u << (idw >> 30)) & ray_mask)) return false;
  if (!(den != 0.0f && U >= 0.0f && V >= 0.0f && U + V <= absden && absden * 0.0f < T)) return false;
  const uint32_t id = idw & 0x3FFFFFFFu;
  if (ANY) {  // tfar of a shadow ray is exactly 1 (Scene.cpp:165-175): RN(T / |den|) <= 1 <=> T <= |den| (tri_test above; tests/test_box_forms.py)
    if (T <= absden) { h.id = id; return true; }
    return false;
  }
  const float t = mi_div(T, absden);
  if (t < h.t || (t == h.t && id < h.id)) {
    h.t = t; h.u = U; h.v = V; h.den = absden; h.id = id; h.pos = slot;
    return true;
  }
  return false;
}

// The box loop.  Table entry k = { c.xyz, e.xyz, link, - }: centre and half extent of the leaf box, the half extent rounded up and padded on the
// host by 2^-20 of the largest coordinate of the scene (cameras included), which covers every rounding of the three fmas below and of v_rcp_f32 for
// rays that start inside the scene's bounds (mi_pt_create: flat table).  Per axis  m = c * inv - org * inv,  tnear = m - e * |inv|,  tfar = m + e * |inv|:
// three fmas instead of two fmas, a min and a max — min / max / cmp issue at half the fma rate on gfx950 (tools/micro/valu_rate.hip).  The test only
// has to be conservative.  Entries are padded to a multiple of 4 with boxes nothing enters (e = -1e30); the loop runs downwards and shifts the
// mask left, so bit k of the mask is entry k.
// K4: groups of four entries to test (all of them for a closest-hit ray, those holding the mesh leaves for a shadow ray, whose mask is then cut to
// keep_mask).  h.t = the ray's tfar on entry (closest-hit rays: infinity, the clamp compiles away).
template <bool ANY, bool COUNT = false, bool MASKED = false>
MI_DEV void traverse_flat(const float4* __restrict__ leaves, cfloat* __restrict__ table, uint32_t K4, uint32_t keep_mask, f3 org, f3 dir, Hit& h, Visits* vis,
                          uint32_t ray_mask = 0xFFFFFFFFu) {
  const RayBox rb = make_raybox(org, dir);
  const f3 ainv = F3(fabsf(rb.inv.x), fabsf(rb.inv.y), fabsf(rb.inv.z));
  // r04: the verdict of a box is kept as a SIGN BIT, not a compare.  The ray enters the box iff tnear <= tfar and 0 <= tfar, i.e. iff neither
  // tfar - tnear nor tfar is negative: one subtraction, one OR of the two words, and v_alignbit_b32 shifts that sign into the lane's mask —
  // three full-rate instructions where max(tnear, 0), v_cmp, v_cndmask and v_lshl_or (all half rate on gfx950) stood: 16 -> 6 issue cycles of the
  // 42 a box cost, for the 16 + ~10 boxes of every trip.  (tfar = -0.0 counts as a miss: the padding of the half extents puts the exit of a
  // box that holds a hit strictly in front of the origin.)  The mask collects MISS bits; it starts as all ones so that bits above the table stay misses.
  // Measured and removed (profiles/r04/ab_c2_instruction_cuts.txt): a box around every group of four entries, tested first, with a wave-wide skip of a
  // group no lane enters — C2 -5 % (24 390 -> 23 180 Msamples/s, VALU per segment 2 200 -> 2 310): the lanes of a wave hold paths at every bounce, some
  // lane enters every group on nearly every trip, and the group tests are pure addition.
  uint32_t miss = 0xFFFFFFFFu;
  for (uint32_t g = K4; g-- != 0u;) {  // wave-uniform: four boxes per trip, scalar operands (two groups per trip, i.e. more scalar loads in flight: -0.7 %, r04)
    cfloat* t = table + 32u * g;
#pragma unroll
    for (int j = 3; j >= 0; --j) {
      const float mx = fmaf(t[8 * j], rb.inv.x, -rb.oi.x), my = fmaf(t[8 * j + 1], rb.inv.y, -rb.oi.y), mz = fmaf(t[8 * j + 2], rb.inv.z, -rb.oi.z);
      const float ex = t[8 * j + 3], ey = t[8 * j + 4], ez = t[8 * j + 5];
      const float tn = fmaxf(fmaxf(fmaf(-ex, ainv.x, mx), fmaf(-ey, ainv.y, my)), fmaf(-ez, ainv.z, mz));
      float tf = fminf(fminf(fmaf(ex, ainv.x, mx), fmaf(ey, ainv.y, my)), fmaf(ez, ainv.z, mz));
      if (ANY) tf = fminf(tf, h.t);
      miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(tf - tn) | __float_as_uint(tf), 31u);  // (miss << 1) | sign
    }
  }
  uint32_t mask = ~miss;
  if (ANY) mask &= keep_mask;
  if (COUNT) vis->nodes += 4u * K4;
  while (mask != 0u) {
    const uint32_t k = uint32_t(__builtin_ctz(mask));
    mask &= mask - 1u;
    const float4* r = leaves + kFlatLeafF4 * k;
    const float4 a0 = r[0], a1 = r[1], a2 = r[2], b0 = r[3], b1 = r[4], b2 = r[5];
    const uint2 ids = *reinterpret_cast<const uint2*>(r + 6);
    if (COUNT) vis->tris += 2u;
    const bool ha = flat_tri<ANY, MASKED>(a0, a1, a2, ids.x, 2u * k, org, dir, h, ray_mask);
    const bool hb = flat_tri<ANY, MASKED>(b0, b1, b2, ids.y, 2u * k + 1u, org, dir, h, ray_mask);
    if (ANY && (ha || hb)) return;
  }
}

template <bool ANY, bool COUNT = false, int QUANT = 0, int NS = 4, bool MASKED = true, bool FAR = false, class Stack = TravStack>
MI_DEV void traverse(const float4* __restrict__ sb, const SceneView& sv, Stack& stack, f3 org, f3 dir,
                     uint32_t ray_mask, Hit& h, Visits* vis = nullptr) {
  traverse_raw<ANY, COUNT, QUANT, NS, MASKED, FAR>(sb, sv, stack, org, dir, ray_mask, h, vis);
  if (!ANY) finish_hit(h);
}

// Scene::querySurface (Scene.cpp:80-126).  SS = shading-record stride in float4 units (8 in HBM, 9 in the padded LDS copy:
// with 128-byte records every lane's k-th float4 falls into the same 4 LDS banks).
template <int SS = 8>
MI_DEV Surf query_surface(const float4* __restrict__ sb, const SceneView& sv, f3 org, f3 dir, const Hit& h) {
  Surf p;
  const float4* sh = sb + sv.off_shade + SS * h.pos;
  const float4 q0 = sh[0], q1 = sh[1], q2 = sh[2], q3 = sh[3], q4 = sh[4], q5 = sh[5], q6 = sh[6], q7 = sh[7];
  const float w = 1.f - h.u - h.v;
  const float u = h.u, v = h.v;
  // vertex frames: t0 = q0.xyzw q1.xyzw q2.x ; t1 = q2.yzw q3.xyzw q4.xy ; t2 = q4.zw q5.xyzw q6.xyz
  const f3 a0 = F3(q0.x, q0.y, q0.z), a1 = F3(q0.w, q1.x, q1.y), a2 = F3(q1.z, q1.w, q2.x);
  const f3 b0 = F3(q2.y, q2.z, q2.w), b1 = F3(q3.x, q3.y, q3.z), b2 = F3(q3.w, q4.x, q4.y);
  const f3 c0 = F3(q4.z, q4.w, q5.x), c1 = F3(q5.y, q5.z, q5.w), c2 = F3(q6.x, q6.y, q6.z);
  p.material_id = __float_as_uint(q6.w);
  p.position = madd(org, dir, h.t);
  p.tangent.c0 = (a0 * w + b0 * u) + c0 * v;
  p.tangent.c1 = (a1 * w + b1 * u) + c1 * v;
  p.tangent.c2 = (a2 * w + b2 * u) + c2 * v;
  p.tangent.c1 = normalize(p.tangent.c1);
  p.tangent.c0 = p.tangent.c0 - p.tangent.c1 * dot(p.tangent.c0, p.tangent.c1);
  p.tangent.c0 = normalize(p.tangent.c0);
  p.tangent.c2 = (p.tangent.c2 - p.tangent.c1 * dot(p.tangent.c2, p.tangent.c1)) - p.tangent.c0 * dot(p.tangent.c2, p.tangent.c0);
  p.tangent.c2 = normalize(p.tangent.c2);
  // RayIsect::gnormal = normalize(-Ng) (RayIsect.hpp:24) is a per-triangle constant: the LBVH build stores it
  // in the shading record (k_emit).  Flip toward the ray origin (Scene.cpp:119-120); the sign of
  // dot(normalize(-dir), g) is taken from dot(-dir, g).
  const f3 g = F3(q7.x, q7.y, q7.z);
  p.gnormal = g * (dot(-dir, g) < 0.0f ? -1.0f : 1.0f);
  return p;
}

// origin offset of Scene::intersect (Scene.cpp:185-188)
MI_DEV f3 nudge(f3 position, f3 gnormal, f3 dir) {
  return position + (gnormal * (dot(gnormal, dir) > 0.0f ? 1.0f : -1.0f)) * 0.0001f;
}

// Scene::occluded (Scene.cpp:151-180): 1 = visible.
template <bool COUNT = false, int QUANT = 0, int NS = 4, bool FAR = false, class Stack = TravStack>
MI_DEV float occluded(const float4* __restrict__ sb, const SceneView& sv, Stack& stack, f3 opos, f3 ognormal, f3 tpos,
                      f3 tgnormal, Visits* vis = nullptr) {
  const f3 direction = tpos - opos;  // Scene.cpp:153 normalises; only signs are used
  const f3 ao = opos + (ognormal * (dot(ognormal, direction) > 0.0f ? 1.0f : -1.0f)) * 0.0001f;
  const f3 at = tpos + (tgnormal * (dot(tgnormal, direction) < 0.0f ? 1.0f : -1.0f)) * 0.0001f;
  Hit h; h.t = 1.0f; h.u = h.v = 0.0f; h.id = 0xFFFFFFFFu; h.pos = 0;
  traverse<true, COUNT, QUANT, NS, true, FAR>(sb, sv, stack, ao, at - ao, 1u << MI_ENTITY_MESH, h, vis);
  return h.id != 0xFFFFFFFFu ? 0.f : 1.f;
}

// ---------------------------------------------------------------------------------------------
// BSDFs
struct BQuery { f3 throughput; float density, densityRev; int finite; };
struct BSample { BQuery q; f3 omega; };

struct Material { uint32_t type; f3 diffuse, specular; float power, ior_internal, ior_external; uint32_t light_id; float phong_pd; };

MI_DEV Material load_material(const float4* __restrict__ sb, const SceneView& sv, uint32_t material_id) {
  const float4* m = sb + sv.off_mats + 3 * (material_id >> 2);
  const float4 a = m[0], b = m[1], c = m[2];
  Material r;
  r.type = __float_as_uint(a.x); r.diffuse = F3(a.y, a.z, a.w); r.specular = F3(b.x, b.y, b.z); r.power = b.w;
  r.ior_internal = c.x; r.ior_external = c.y; r.light_id = __float_as_uint(c.z); r.phong_pd = c.w;
  return r;
}

// DiffuseBSDF::_query (BSDF.cpp:291-304)
MI_DEV BQuery diffuse_query_local(const Material& m, f3 gn, f3 incident, f3 outgoing) {
  const float same_side = dot(incident, gn) * dot(outgoing, gn) > 0.0f ? 1.0f : 0.0f;
  BQuery q;
  q.throughput = (m.diffuse * MI_ONE_OVER_PI) * same_side;
  q.density = fabsf(outgoing.y * MI_ONE_OVER_PI) * same_side;
  q.densityRev = fabsf(incident.y * MI_ONE_OVER_PI) * same_side;
  q.finite = 1;
  return q;
}
// PhongBSDF::_query (BSDF.cpp:354-391)
MI_DEV BQuery phong_query_local(const Material& m, f3 incident, f3 outgoing, float same_side) {
  const float pd = m.phong_pd, ps = 1.0f - pd;
  const float dd = fabsf(outgoing.y * MI_ONE_OVER_PI), ddr = fabsf(incident.y * MI_ONE_OVER_PI);
  const f3 diffuse = m.diffuse * MI_ONE_OVER_PI;
  const float half_over_pi = 0.5f * MI_ONE_OVER_PI;
  const f3 reflected = F3(-incident.x, incident.y, -incident.z);
  float ca = dot(outgoing, reflected);
  ca = ca < 0.0f ? 0.0f : (ca > 1.0f ? 1.0f : ca);
  const float cap = mi_powf(ca, m.power);
  const float sd = (m.power + 1.0f) * half_over_pi * cap;
  const f3 specular = ((m.specular * (m.power + 2.0f)) * half_over_pi) * cap;
  BQuery q;
  q.density = same_side * (sd * ps + dd * pd);
  q.densityRev = same_side * (sd * ps + ddr * pd);
  q.throughput = (diffuse + specular) * same_side;
  q.finite = 1;
  return q;
}
MI_DEV BQuery bq_zero() { BQuery q; q.throughput = F3(0, 0, 0); q.density = 0; q.densityRev = 0; q.finite = 1; return q; }

// Scene::queryBSDF(surface, incident, outgoing) (Scene.cpp:142-149) for surface materials.
template <int FEAT = kFeatAll>
MI_DEV BQuery bsdf_query(const Material& m, const Surf& sf, f3 incident, f3 outgoing) {
  if (m.type == MI_BSDF_DIFFUSE)  // BSDF.cpp:239-243
    return diffuse_query_local(m, to_surface(sf, sf.gnormal), to_surface(sf, incident), to_surface(sf, outgoing));
  if ((FEAT & kFeatPhong) && m.type == MI_BSDF_PHONG) {  // BSDF.cpp:317-326
    const float same_side = dot(incident, sf.gnormal) * dot(outgoing, sf.gnormal) > 0.0f ? 1.0f : 0.0f;
    return phong_query_local(m, to_surface(sf, incident), to_surface(sf, outgoing), same_side);
  }
  BQuery q = bq_zero();  // DeltaBSDF::query (BSDF.cpp:438-448)
  q.finite = 0;
  return q;
}

// sample_lambert (Sample.inl:52-60)
MI_DEV f3 sample_lambert(Rng& g, f3 omega) {
  const float y = mi_sqrt(rng_f(g)) * gsign(omega.y);
  const float r = mi_sqrt(1.0f - y * y);
  float sn, cs;
  sincos_2pi(rng_f(g), &sn, &cs);
  return F3(r * cs, y, r * sn);
}
// reflection_to_surface (Sample.inl:43-50) + sample_phong (Sample.inl:139-151)
MI_DEV f3 sample_phong(Rng& g, f3 omega, float power) {
  m33 m;
  m.c1 = F3(-omega.x, omega.y, -omega.z);
  m.c2 = normalize(F3(0.0f, 1.0f, 0.0f) - m.c1 * m.c1.y);
  m.c0 = normalize(cross(m.c1, m.c2));
  const float y = mi_powf(rng_f(g), mi_rcp(power + 1.0f));
  const float r = mi_sqrt(1.0f - y * y);
  float sn, cs;
  sincos_2pi(rng_f(g), &sn, &cs);
  return mulmv(m, F3(r * cs, y, r * sn));
}
// Scene::sampleBSDF (Scene.cpp:133-140)
template <int FEAT = kFeatAll>
MI_DEV BSample bsdf_sample(const Material& m, Rng& g, const Surf& sf, f3 omega) {
  BSample r;
  r.q = bq_zero();
  r.omega = F3(0, 0, 0);
  const f3 lo = to_surface(sf, omega);
  if (m.type == MI_BSDF_DIFFUSE) {  // BSDF.cpp:245-262
    const f3 d = sample_lambert(g, lo);
    r.q = diffuse_query_local(m, to_surface(sf, sf.gnormal), lo, d);
    r.omega = to_world(sf, d);
  } else if ((FEAT & kFeatPhong) && m.type == MI_BSDF_PHONG) {  // BSDF.cpp:328-352
    f3 d;
    if (rng_f(g) < m.phong_pd) d = sample_lambert(g, lo); else d = sample_phong(g, lo, m.power);
    r.omega = to_world(sf, d);
    const float same_side = dot(omega, sf.gnormal) * dot(r.omega, sf.gnormal) > 0.0f ? 1.0f : 0.0f;
    r.q = phong_query_local(m, lo, d, same_side);
  } else if ((FEAT & kFeatDelta) && m.type == MI_BSDF_REFLECTION) {  // BSDF.cpp:450-465
    const float v = mi_rcp(lo.y);
    r.q.throughput = F3(v, v, v);
    r.omega = to_world(sf, F3(-lo.x, lo.y, -lo.z));
    r.q.density = 1.0f; r.q.densityRev = 1.0f; r.q.finite = 0;
  } else if ((FEAT & kFeatDelta) && m.type == MI_BSDF_TRANSMISSION) {  // BSDF.cpp:467-504
    const float ext_over_int = mi_div(m.ior_external, m.ior_internal);
    f3 o;
    if (lo.y > 0.f) {
      const float eta = ext_over_int;
      const float yy = mi_sqrt(1 - eta * eta * (1 - lo.y * lo.y));
      o = ((lo - F3(0.0f, lo.y, 0.0f)) * -eta) - F3(0.0f, yy, 0.0f);
    } else {
      const float eta = mi_rcp(ext_over_int);
      const float yy = mi_sqrt(1 - eta * eta * (1 - lo.y * lo.y));
      o = ((lo - F3(0.0f, lo.y, 0.0f)) * -eta) + F3(0.0f, yy, 0.0f);
    }
    const float v = mi_rcp(fabsf(o.y));
    r.q.throughput = F3(v, v, v);
    r.omega = to_world(sf, o);
    r.q.density = 1.0f; r.q.densityRev = 1.0f; r.q.finite = 0;
  } else {  // light / camera materials are never sampled by PT (lights are passed through)
    r.omega = -omega; r.q.density = 1.0f;
  }
  return r;
}

// ---------------------------------------------------------------------------------------------
// Lights
MI_DEV const float4* light_rec(const float4* __restrict__ sb, const SceneView& sv, uint32_t id) { return sb + sv.off_lights + 6 * id; }

// Scene::queryLSDF -> AreaLights::queryLSDF (Scene.cpp:128-131, AreaLights.cpp:142-155)
// `light0`: the first light's record in GLOBAL memory.  A scene with exactly one light (FEAT without kFeatLights) reads it from there: the address is
// wave-uniform, so the record arrives through scalar loads in SGPRs instead of 6 float4 = 24 VGPRs per lane at the point of highest pressure.
template <int FEAT = kFeatAll>
MI_DEV void query_lsdf(const float4* __restrict__ sb, const SceneView& sv, const float4* __restrict__ light0, uint32_t light_id, f3 omega, f3& radiance, float& density) {
  const float4* L = (FEAT & kFeatLights) ? light_rec(sb, sv, light_id) : light0;
  const float4 l2 = L[2], l4 = L[4], l5 = L[5];
  const float c = dot(omega, xyz(l2));
  radiance = xyz(l4) * (c > 0.0f ? 1.0f : 0.0f);
  density = l5.x;
}

template <int FEAT = kFeatAll>
MI_DEV float powb(float x, float beta) { return beta == 1.0f ? x : (beta == 2.0f || !(FEAT & kFeatPow) ? x * x : mi_powf(x, beta)); }

// PathTracing::_connect (PT.cpp:100-120) incl. AreaLights::sample (AreaLights.cpp:121-140,216-231),
// LightBSDF::query / sun_light_bsdf::query (BSDF.cpp:95-114,181-191; only .throughput is used) and
// Edge (SurfacePoint.hpp:65-83) — everything except the visibility test.  Returns the contribution
// for an unoccluded light sample and the end points of the shadow ray (Scene::occluded's
// adjusted origin/target, Scene.cpp:153-167); the caller multiplies by the visibility (0 or 1)
// once the ray has been traversed.  has_shadow = false: the reference returned before casting.
struct ShadowRay { f3 org, dir; };

template <int FEAT = kFeatAll>
MI_DEV f3 connect_prepare(const float4* __restrict__ sb, const SceneView& sv, const float4* __restrict__ light0, Rng& g, const Material& mat, const Surf& x,
                          f3 x_omega, f3 x_throughput, float beta, bool& has_shadow, ShadowRay& ray) {
  const float u = rng_f(g);
  const float4* L = light0;
  if (FEAT & kFeatLights) {
    const float* cdf = reinterpret_cast<const float*>(sb + sv.off_cdf);
    uint32_t id = sv.n_lights - 1;
    for (uint32_t i = 0; i + 1 < sv.n_lights; ++i) {
      if (u < cdf[i + 1]) { id = i; break; }
    }
    L = light_rec(sb, sv, id);
  }
  const float4 l0 = L[0], l1 = L[1], l2 = L[2], l3 = L[3], l4 = L[4], l5 = L[5];
  const float sx = rng_f(g), sy = rng_f(g);
  const float ux = (sx - 0.5f) * l2.w, uy = (sy - 0.5f) * l3.w;
  const f3 lpos = (xyz(l0) + xyz(l1) * ux) + xyz(l3) * uy;
  const f3 lnormal = xyz(l2);
  const f3 xl = x.position - lpos;
  const float len2 = dot(xl, xl);  // == dot(lpos - x, lpos - x) bit for bit
  const f3 omega = xl * mi_rsqrt(len2);
  // light-side "BSDF": front side only; sun lights contribute nothing through NEE
  const float front = (__float_as_uint(l5.z) != 0u && dot(lnormal, omega) > 0.0f) ? 1.0f : 0.0f;
  has_shadow = !(front * 3.0f < MI_FLT_EPSILON);
  if (!has_shadow) return F3(0, 0, 0);
  const BQuery eb = bsdf_query<FEAT>(mat, x, -omega, x_omega);
  // Edge(light.surface, eye.surface, omega)
  const float distSqInv = mi_rcp(len2);
  const float fCos = fabsf(dot(omega, x.tangent.c1));
  const float bCos = fabsf(dot(omega, lnormal));
  const float fG = distSqInv * fCos;
  const float bG = distSqInv * bCos;
  const float cd = l5.y * l0.w;  // area_density * light_density (AreaLights.cpp:135-139)
  const float wInv = mi_div(powb<FEAT>(eb.densityRev * bG, beta), powb<FEAT>(cd, beta)) + 1.0f;
  // Scene::occluded's end points (Scene.cpp:153-167); signs from the unnormalised direction
  const f3 direction = lpos - x.position;
  const f3 ao = x.position + (x.gnormal * (dot(x.gnormal, direction) > 0.0f ? 1.0f : -1.0f)) * 0.0001f;
  const f3 at = lpos + (lnormal * (dot(lnormal, direction) < 0.0f ? 1.0f : -1.0f)) * 0.0001f;
  ray.org = ao;
  ray.dir = at - ao;
  // radiance / cd = radiance * (1 / cd) by the contract.  One light (its record arrives through scalar loads): 1 / cd is a constant of the light, divided once on
  // the host (DevLight::inv_cd: the same IEEE division).  Several lights (the record is read per lane): the division stays — the fourth word of l1 cost the
  // 80-register kernels four more spilled dwords per lane (profiles/r04/ab_c2_instruction_cuts.txt).  Same bits either way.
  f3 r = (FEAT & kFeatLights) ? xyz(l4) / cd : xyz(l4) * l1.w;
  r = r * x_throughput;
  r = r * eb.throughput;
  r = r * bCos;
  r = r * fG;
  return r / wInv;
}

}  // namespace mi
