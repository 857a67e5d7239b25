// bpt_kernels.hip — bidirectional path tracing (BPTBase<Beta>, BPT.cpp:13-337) on the PT device functions (gfx950).
//
// First device version of SURVEY.md 8(f) rank 4: one lane = one camera sample = one light sub-path + one eye sub-path.
//   bpt_frame    per lane: roulette, _traceLight into the lane's vertex slab in HBM (SoA over lanes), then _traceEye with the
//                connections _connect / _connect_light / _connect_directional / _connect_eye; the eye-image value goes to
//                eye[pixel], light-image splats are FP64 atomics into light[pixel'] (Technique::_accumulate,
//                Technique.cpp:276-306, adds under a mutex into a double image)
//   bpt_commit   per pixel and frame: light + eye passes the finite filter together (Technique.cpp:194-244)
// Arithmetic follows oracle/bpt_oracle.inc operation by operation (same own asin / atan2 / sincos definitions), so eye
// radiance, ray counts and splat sums of a path are bit-identical to the oracle's; only the FP64 order of a pixel's splats is free.
// The scene is read from HBM through the quantised nodes; no LDS-resident variant yet.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cmath>
#include <algorithm>

#include "pt_device.h"
#include "bpt.h"

// This file is compiled three times (master_amd/build.py): with every BSDF (bpt_all), without pow(x, beta) for the reference's
// FixedBeta<0|1|2> techniques (bpt_fixed), and additionally without Phong lobes, mirrors and glass (bpt_plain).  The host picks
// the set that covers the scene and beta; kernels that do not contain the unused code spill less (Cornell box: 55.3 -> 49.6 ms).
#ifndef MI_BPT_FEAT
#define MI_BPT_FEAT 7  // kFeat* bits (pt_device.h) without kFeatLights: the templates' default (all) is what the light code uses here
#define MI_BPT_NS bpt_all
#endif

namespace mi {
namespace MI_BPT_NS {


namespace {

struct LVert { Surf surface; f3 omega, throughput; float a, A; int finite; };  // BPT.hpp:14-20
struct EVert { Surf surface; f3 omega, throughput; float c, C; int finite; };  // BPT.hpp:22-28

struct Ctx {
  const float4* sb; const SceneView* sv;
  void* stack;  // TravStackT<QN != 0>: the LDS-resident kernels (QN == 0) run only on trees whose whole stack fits its LDS rows (host), without the spill path
  float beta, roulette, rinv;
  f3 sphere_c; float sphere_r;
  f3 sky_horizon, sky_zenith;
  uint32_t n_basic, n_shadow;
  float pre_occ;  // bpt_connect<QN, true>: the visibility stage has walked this connection's shadow ray already (1 = visible)
  // LDS-resident scenes (QN == 0) with a leaf table: the flat leaf list instead of the tree walk (traverse_flat, pt_device.h; r03).  flat_k = 0: tree.
  const float* flat_table; uint32_t flat_k, flat_k_mesh;
};

MI_DEV float betaf(const Ctx& c, float x) {  // Beta.hpp:24-41
  if (c.beta == 0.0f) return x == 0.0f ? 0.0f : 1.0f;
  if (c.beta == 1.0f) return x;
  if (c.beta == 2.0f) return x * x;
  return (MI_BPT_FEAT & kFeatPow) ? mi_powf(x, c.beta) : x * x;
}

// ---- Sample.inl:5-37 angular_bound, :121-133 lambert_adjust ----
struct LRange { float theta_range, phi_range, uniform_theta_inf, uniform_phi_inf; };
MI_DEV LRange lambert_ranges(f3 center, float radius) {
  const float half_pi = 1.57079632679489661923132169163975144f, two_pi = 6.28318530717958647692528676655900576f;
  float theta_inf = 0.0f, theta_sup = half_pi, phi_inf = 0.0f, phi_sup = two_pi;
  const float lateral_distance_sq = center.x * center.x + center.z * center.z;
  const float distance_sq = lateral_distance_sq + center.y * center.y;
  const float radius_sq = radius * radius;
  if (radius_sq < distance_sq) {
    const float lateral_distance = sqrtf(lateral_distance_sq);
    const float distance = sqrtf(distance_sq);
    const float theta_center = mi_asinf(lateral_distance / distance);
    const float theta_radius = mi_asinf(radius / distance);
    if (lateral_distance_sq < radius_sq) {
      theta_sup = fminf(half_pi, theta_center + theta_radius);
    } else if (radius_sq < distance_sq) {
      theta_inf = theta_center - theta_radius;
      theta_sup = fminf(half_pi, theta_center + theta_radius);
      const float phi_center = mi_atan2f(center.z, center.x);
      const float phi_radius = mi_asinf(radius / lateral_distance);
      phi_inf = phi_center - phi_radius;
      phi_sup = phi_center + phi_radius;
    }
  }
  LRange r;
  float sn, c_sup, c_inf;
  mi_sincosf(theta_sup, &sn, &c_sup); mi_sincosf(theta_inf, &sn, &c_inf);
  r.uniform_theta_inf = c_sup * c_sup;
  const float uniform_theta_sup = c_inf * c_inf;
  r.uniform_phi_inf = phi_inf * MI_ONE_OVER_PI * 0.5f;
  const float uniform_phi_sup = phi_sup * MI_ONE_OVER_PI * 0.5f;
  r.theta_range = uniform_theta_sup - r.uniform_theta_inf;
  r.phi_range = uniform_phi_sup - r.uniform_phi_inf;
  return r;
}
MI_DEV float lambert_adjust(f3 center, float radius) { const LRange r = lambert_ranges(center, radius); return r.theta_range * r.phi_range; }
MI_DEV f3 local_sphere_center(const Ctx& c, const Surf& sf) { return to_surface(sf, c.sphere_c - sf.position); }

// ---- Scene::queryBSDF / sampleBSDF with the emitter and camera "BSDFs" (BSDF.cpp:75-232) ----
MI_DEV BQuery bpt_bsdf_query(const Ctx& c, const Surf& sf, f3 incident, f3 outgoing) {
  const Material m = load_material(c.sb, *c.sv, sf.material_id);
  if (m.type == MI_BSDF_LIGHT) {
    BQuery q = bq_zero();
    const f3 lo = to_surface(sf, outgoing);
    q.throughput = lo.y > 0.0f ? F3(1, 1, 1) : F3(0, 0, 0);
    q.density = (lo.y > 0.0f ? 1.0f : 0.0f) * lo.y * MI_ONE_OVER_PI / lambert_adjust(local_sphere_center(c, sf), c.sphere_r);
    q.densityRev = 0.0f;
    return q;
  }
  if (m.type == MI_BSDF_SUN) { BQuery q = bq_zero(); q.density = 1.0f; q.densityRev = 1.0f; return q; }
  if (m.type == MI_BSDF_CAMERA) {
    BQuery q = bq_zero();
    const f3 li = to_surface(sf, incident);
    const float v = (li.y > 0.0f ? 1.0f : 0.0f) / fabsf(li.y);
    q.throughput = F3(v, v, v); q.density = 0.0f; q.densityRev = 1.0f;
    return q;
  }
  return bsdf_query<MI_BPT_FEAT>(m, sf, incident, outgoing);
}
MI_DEV BSample bpt_bsdf_sample(const Ctx& c, Rng& g, const Surf& sf, f3 omega) {
  const Material m = load_material(c.sb, *c.sv, sf.material_id);
  BSample r; r.q = bq_zero(); r.omega = F3(0, 0, 0);
  if (m.type == MI_BSDF_LIGHT) {
    f3 ctr = local_sphere_center(c, sf);
    const f3 lo = to_surface(sf, omega);
    ctr.y *= gsign(lo.y);
    const LRange lr = lambert_ranges(ctr, c.sphere_r);
    const float adjust = lr.theta_range * lr.phi_range;
    const float y = sqrtf(rng_f(g) * lr.theta_range + lr.uniform_theta_inf) * gsign(lo.y);
    const float turns = rng_f(g) * lr.phi_range + lr.uniform_phi_inf;
    const float rr = sqrtf(1 - y * y);
    float sphi, cphi; sincos_2pi(turns - floorf(turns), &sphi, &cphi);
    const f3 d = F3(rr * cphi, y, rr * sphi);
    r.q.throughput = F3(1, 1, 1);
    r.omega = to_world(sf, d);
    r.q.density = fabsf(d.y) * MI_ONE_OVER_PI / adjust;
    r.q.densityRev = 0.0f; r.q.finite = 1;
    return r;
  }
  if (m.type == MI_BSDF_SUN) { r.q.throughput = F3(1, 1, 1); r.omega = omega; r.q.density = 1.0f; r.q.densityRev = 0.0f; r.q.finite = 1; return r; }
  if (m.type == MI_BSDF_CAMERA) {
    const float v = 1.0f / fabsf(dot(sf.tangent.c1, omega));
    r.omega = -omega; r.q.throughput = F3(v, v, v); r.q.density = 1.0f; r.q.densityRev = 0.0f; r.q.finite = 1;
    return r;
  }
  return bsdf_sample<MI_BPT_FEAT>(m, g, sf, omega);
}

// Scene::intersect / intersectMesh (Scene.cpp:182-227): closest hit with a geometry mask
template <int QN>
MI_DEV Surf scene_intersect(Ctx& c, const Surf& from, f3 dir, uint32_t mask) {
  const f3 org = nudge(from.position, from.gnormal, dir);
  Hit h; h.t = __builtin_inff(); h.u = h.v = 0.0f; h.id = 0xFFFFFFFFu; h.pos = 0;
  if (QN == 0 && c.flat_k != 0u) { traverse_flat<false, false, true>(c.sb, (cfloat*)c.flat_table, (c.flat_k + 3u) >> 2, 0xFFFFFFFFu, org, dir, h, nullptr, mask); finish_hit(h); }
  else
  traverse<false, false, QN, QN == 0 ? 5 : 4>(c.sb, *c.sv, *static_cast<TravStackT<(QN != 0)>*>(c.stack), org, dir, mask, h);  // QN == 0: padded LDS copy of the scene
  ++c.n_basic;
  if (h.id == 0xFFFFFFFFu) { Surf s; s.position = F3(0, 0, 0); s.gnormal = F3(0, 0, 0); s.tangent.c0 = s.tangent.c1 = s.tangent.c2 = F3(0, 0, 0); s.material_id = 0xFFFFFFFFu; return s; }
  return query_surface<QN == 0 ? 9 : 8>(c.sb, *c.sv, org, dir, h);
}
template <int QN>
MI_DEV float scene_occluded(Ctx& c, const Surf& origin, const Surf& target) {
  ++c.n_shadow;
  if (QN == 0 && c.flat_k != 0u) {  // Scene::occluded (Scene.cpp:151-180) through the flat leaf list: occluded() of pt_device.h with traverse_flat
    const f3 direction = target.position - origin.position;
    const f3 ao = origin.position + (origin.gnormal * (dot(origin.gnormal, direction) > 0.0f ? 1.0f : -1.0f)) * 0.0001f;
    const f3 at = target.position + (target.gnormal * (dot(target.gnormal, direction) < 0.0f ? 1.0f : -1.0f)) * 0.0001f;
    Hit h; h.t = 1.0f; h.u = h.v = 0.0f; h.id = 0xFFFFFFFFu; h.pos = 0;
    traverse_flat<true>(c.sb, (cfloat*)c.flat_table, (c.flat_k_mesh + 3u) >> 2, c.flat_k_mesh >= 32u ? 0xFFFFFFFFu : (1u << c.flat_k_mesh) - 1u, ao, at - ao, h, nullptr);
    return h.id != 0xFFFFFFFFu ? 0.f : 1.f;
  }
  return occluded<false, QN, QN == 0 ? 5 : 4>(c.sb, *c.sv, *static_cast<TravStackT<(QN != 0)>*>(c.stack), origin.position, origin.gnormal, target.position, target.gnormal);
}

// AreaLights::sample (AreaLights.cpp:121-140)
struct LSample { Surf surface; f3 radiance; float area_density, light_density; bool directional; };
MI_DEV LSample light_sample(const Ctx& c, Rng& g) {
  const SceneView& sv = *c.sv;
  const float u = rng_f(g);
  const float* cdf = reinterpret_cast<const float*>(c.sb + sv.off_cdf);
  uint32_t id = sv.n_lights - 1;
  for (uint32_t i = 0; i + 1 < sv.n_lights; ++i) {
    if (u < cdf[i + 1]) { id = i; break; }
  }
  const float4* L = light_rec(c.sb, sv, id);
  const float4 l0 = L[0], l1 = L[1], l2 = L[2], l3 = L[3], l4 = L[4], l5 = L[5];
  const float sx = rng_f(g), sy = rng_f(g);
  const float ux = (sx - 0.5f) * l2.w, uy = (sy - 0.5f) * l3.w;
  LSample r;
  r.surface.position = (xyz(l0) + xyz(l1) * ux) + xyz(l3) * uy;
  r.surface.tangent.c0 = xyz(l1); r.surface.tangent.c1 = xyz(l2); r.surface.tangent.c2 = xyz(l3);
  r.surface.gnormal = xyz(l2);
  r.surface.material_id = __float_as_uint(l4.w);
  r.radiance = xyz(l4);
  r.area_density = l5.y;
  r.light_density = l0.w;
  r.directional = __float_as_uint(l5.z) == 0u;
  return r;
}
MI_DEV bool bpt_roulette(const Ctx& c, Rng& g) { return c.roulette < rng_f(g); }

struct Edge { float distSqInv, fCos, bCos, fG, bG; };
MI_DEV Edge make_edge(const Surf& fst, const Surf& snd, f3 omega) {  // SurfacePoint.hpp:65-83
  Edge e; const f3 d = fst.position - snd.position;
  e.distSqInv = 1.0f / dot(d, d);
  e.fCos = fabsf(dot(omega, snd.tangent.c1));
  e.bCos = fabsf(dot(omega, fst.tangent.c1));
  e.fG = e.distSqInv * e.fCos;
  e.bG = e.distSqInv * e.bCos;
  return e;
}

MI_DEV LVert sample_to_vertex(const Ctx& c, const LSample& b) {  // BPT.cpp:103-114
  LVert v; const float cd = b.area_density * b.light_density;
  v.surface = b.surface;
  v.omega = v.surface.tangent.c1;
  v.throughput = (b.radiance / cd) * c.rinv;
  v.a = b.directional ? 0.0f : 1.0f / betaf(c, cd);
  v.A = 0.0f; v.finite = 1;
  return v;
}

// the lane's light sub-path in HBM: 7 float4 per vertex, SoA over lanes
struct Slab { float4* base; uint32_t lanes, lane, cap; };
MI_DEV void slab_store(const Slab& s, uint32_t v, const LVert& x) {
  float4* p = s.base + (size_t(v) * 7u) * s.lanes + s.lane;
  p[0] = make_float4(x.surface.position.x, x.surface.position.y, x.surface.position.z, x.a);
  p[size_t(1) * s.lanes] = make_float4(x.surface.gnormal.x, x.surface.gnormal.y, x.surface.gnormal.z, x.A);
  p[size_t(2) * s.lanes] = make_float4(x.surface.tangent.c0.x, x.surface.tangent.c0.y, x.surface.tangent.c0.z, __int_as_float(x.finite));
  p[size_t(3) * s.lanes] = make_float4(x.surface.tangent.c1.x, x.surface.tangent.c1.y, x.surface.tangent.c1.z, __uint_as_float(x.surface.material_id));
  p[size_t(4) * s.lanes] = make_float4(x.surface.tangent.c2.x, x.surface.tangent.c2.y, x.surface.tangent.c2.z, 0.f);
  p[size_t(5) * s.lanes] = make_float4(x.omega.x, x.omega.y, x.omega.z, 0.f);
  p[size_t(6) * s.lanes] = make_float4(x.throughput.x, x.throughput.y, x.throughput.z, 0.f);
}
MI_DEV LVert slab_load(const Slab& s, uint32_t v) {
  const float4* p = s.base + (size_t(v) * 7u) * s.lanes + s.lane;
  const float4 q0 = p[0], q1 = p[size_t(1) * s.lanes], q2 = p[size_t(2) * s.lanes], q3 = p[size_t(3) * s.lanes], q4 = p[size_t(4) * s.lanes],
               q5 = p[size_t(5) * s.lanes], q6 = p[size_t(6) * s.lanes];
  LVert x;
  x.surface.position = xyz(q0); x.a = q0.w;
  x.surface.gnormal = xyz(q1); x.A = q1.w;
  x.surface.tangent.c0 = xyz(q2); x.finite = __float_as_int(q2.w);
  x.surface.tangent.c1 = xyz(q3); x.surface.material_id = __float_as_uint(q3.w);
  x.surface.tangent.c2 = xyz(q4);
  x.omega = xyz(q5); x.throughput = xyz(q6);
  return x;
}

// BPTBase::_traceLight (BPT.cpp:121-190); returns the number of vertices kept
template <int QN>
MI_DEV uint32_t trace_light(Ctx& c, Rng& g, const Slab& slab, bool& overflow) {
  if (bpt_roulette(c, g)) return 0;
  const LSample ls = light_sample(c, g);
  LVert prev = sample_to_vertex(c, ls);  // path[prv], kept in registers and written back when it is final
  uint32_t size = 1, prv = 0;
  while (!bpt_roulette(c, g)) {
    const BSample b = bpt_bsdf_sample(c, g, prev.surface, prev.omega);
    const Surf surface = scene_intersect<QN>(c, prev.surface, b.omega, 1u << MI_ENTITY_MESH);
    if (surface.material_id == 0xFFFFFFFFu) break;
    if (size >= slab.cap) { overflow = true; break; }
    LVert cur;
    cur.surface = surface;
    cur.omega = -b.omega;
    const Edge e = make_edge(prev.surface, cur.surface, cur.omega);
    cur.throughput = ((prev.throughput * b.q.throughput) * e.bCos) * c.rinv;
    if (l1norm(cur.throughput) < MI_FLT_EPSILON) break;
    cur.throughput = cur.throughput / b.q.density;
    prev.finite = prev.finite < b.q.finite ? prev.finite : b.q.finite;
    cur.finite = b.q.finite;
    cur.a = 1.0f / betaf(c, e.fG * b.q.density);
    cur.A = (prev.A * betaf(c, b.q.densityRev) + prev.a * float(prev.finite)) * betaf(c, e.bG) * cur.a;
    if (b.q.finite == 0) {
      prev = cur;  // path[prv] = path[itr]; pop_back
    } else {
      slab_store(slab, prv, prev);
      prev = cur; prv = size; ++size;
    }
  }
  const BSample last = bpt_bsdf_sample(c, g, prev.surface, prev.omega);
  if (last.q.finite == 0) --size; else slab_store(slab, prv, prev);
  return size;
}

// BPTBase::_connect(light, eye) (BPT.cpp:192-224)
template <int QN, bool PRE = false>
MI_DEV f3 bpt_connect(Ctx& c, const LVert& light, const EVert& eye) {
  const f3 omega = normalize(eye.surface.position - light.surface.position);
  const BQuery lb = bpt_bsdf_query(c, light.surface, light.omega, omega);
  const BQuery eb = bpt_bsdf_query(c, eye.surface, -omega, eye.omega);
  const Edge e = make_edge(light.surface, eye.surface, omega);
  const float Ap = (light.A * betaf(c, lb.densityRev) + light.a * float(light.finite)) * betaf(c, e.bG * eb.densityRev);
  const float Cp = (eye.C * betaf(c, eb.density) + eye.c * float(eye.finite)) * betaf(c, e.fG * lb.density);
  const float weightInv = Ap + Cp + 1.0f;
  const float occ = PRE ? c.pre_occ : scene_occluded<QN>(c, eye.surface, light.surface);
  f3 r = light.throughput * occ;
  r = r * lb.throughput; r = r * eye.throughput; r = r * eb.throughput;
  r = r * e.bCos; r = r * e.fG;
  return l1norm(r) < MI_FLT_EPSILON ? F3(0, 0, 0) : r / weightInv;
}
// BPTBase::_connect_light (BPT.cpp:226-245)
MI_DEV f3 bpt_connect_light(const Ctx& c, const EVert& eye) {
  const BQuery b = bpt_bsdf_query(c, eye.surface, F3(0, 0, 0), eye.omega);
  if (l1norm(b.throughput) < 1.17549435e-38f) return F3(0, 0, 0);  // FLT_MIN
  const Material lm = load_material(c.sb, *c.sv, eye.surface.material_id);
  f3 le; float dens;
  query_lsdf(c.sb, *c.sv, nullptr, lm.light_id, eye.omega, le, dens);
  const float Cp = (eye.C * betaf(c, b.density) + eye.c * float(eye.finite)) * betaf(c, dens);
  return (le * eye.throughput) / (Cp + 1.0f);
}
// BPTBase::_connect_directional (BPT.cpp:247-273)
template <int QN>
MI_DEV f3 bpt_connect_directional(Ctx& c, const EVert& eye, const LSample& b) {
  const f3 ln = b.surface.tangent.c1;
  const Surf isect = scene_intersect<QN>(c, eye.surface, -ln, 0xFFFFFFFFu);
  if (isect.material_id != b.surface.material_id) return F3(0, 0, 0);
  const BQuery eb = bpt_bsdf_query(c, eye.surface, -ln, eye.omega);
  const f3 d = isect.position - eye.surface.position;
  const float cosn = fabsf(dot(ln, eye.surface.tangent.c1));
  const float Cp = (eye.C * betaf(c, eb.density) + eye.c * float(eye.finite)) * betaf(c, cosn / dot(d, d));
  f3 r = (b.radiance / b.light_density) * (1.0f / c.roulette);
  r = r * eye.throughput; r = r * eb.throughput; r = r * cosn;
  return l1norm(r) < MI_FLT_EPSILON ? F3(0, 0, 0) : r / (Cp + 1.0f);
}

struct Cam { m33 w2v; float rx, ry, ry_inv, fl; };
struct SplatOut { double* light; uint32_t n; f3 sum; };  // light == nullptr: list mode (sum only)

// BPTBase::_connect_eye (BPT.cpp:295-321) with Technique::_camera_coefficient / _accumulate (Technique.cpp:246-306)
template <int QN>
MI_DEV void bpt_connect_eye(Ctx& c, const Cam& cam, const EVert& eye, const Slab& slab, uint32_t size, SplatOut& out) {
  const float focal_factor_y = cam.fl * cam.fl * 0.25f;
  for (uint32_t i = 0; i < size; ++i) {
    const LVert lv = slab_load(slab, i);
    const f3 omega = normalize(lv.surface.position - eye.surface.position);
    const f3 vd = mulmv(cam.w2v, omega);
    const float factor = cam.fl / -vd.z;  // pixel_position (Cameras.cpp:134-144)
    const float x = vd.x * factor, y = vd.y * factor;
    const float py = (y + 1.0f) * cam.ry * 0.5f;
    const float px = (x + cam.rx * cam.ry_inv) * cam.ry * 0.5f;
    if (!(0 <= px && px < cam.rx && 0 <= py && py < cam.ry)) continue;
    const int ix = int(px), iy = int(py);
    const f3 ln = lv.surface.tangent.c1, en = eye.surface.tangent.c1;
    const float normal_coefficient = fabsf(dot(omega, lv.surface.gnormal) * dot(lv.omega, ln) / (dot(omega, ln) * dot(lv.omega, lv.surface.gnormal)));
    const float ce = fabsf(dot(en, omega));
    const float focal_coefficient = 1.0f / (ce * ce * ce);
    const f3 r = (bpt_connect<QN>(c, lv, eye) * focal_factor_y) * (normal_coefficient * focal_coefficient);
    out.sum = out.sum + r; ++out.n;
    if (out.light) {
      double* l = out.light + 3 * (size_t(iy) * size_t(cam.rx) + size_t(ix));
      atomicAdd(&l[0], double(r.x)); atomicAdd(&l[1], double(r.y)); atomicAdd(&l[2], double(r.z));
    }
  }
}
// BPTBase::_connect(context, eye, path) (BPT.cpp:275-293)
template <int QN>
MI_DEV f3 bpt_connect_all(Ctx& c, Rng& g, const EVert& eye, const Slab& slab, uint32_t size) {
  f3 radiance = F3(0, 0, 0);
  if (!bpt_roulette(c, g)) {
    const LSample b = light_sample(c, g);
    if (!b.directional) { const LVert lv = sample_to_vertex(c, b); radiance = radiance + bpt_connect<QN>(c, lv, eye); }
    else if ((eye.surface.material_id & 3u) != MI_ENTITY_CAMERA) radiance = radiance + bpt_connect_directional<QN>(c, eye, b);
  }
  for (uint32_t i = 1; i < size; ++i) { const LVert lv = slab_load(slab, i); radiance = radiance + bpt_connect<QN>(c, lv, eye); }
  return radiance;
}

// BPTBase::_traceEye (BPT.cpp:13-101)
template <int QN>
MI_DEV f3 bpt_trace_eye(Ctx& c, Rng& g, const Cam& cam, const Surf& camera_surface, f3 dir, const Slab& slab, SplatOut& splats, bool& overflow) {
  f3 radiance = F3(0, 0, 0);
  if (bpt_roulette(c, g)) return radiance;
  const uint32_t lsize = trace_light<QN>(c, g, slab, overflow);
  EVert prev, cur;
  Surf surface = camera_surface;
  prev.surface = surface; prev.omega = -dir; prev.throughput = F3(1, 1, 1) * c.rinv;
  prev.finite = 1; prev.c = 0.0f; prev.C = 0.0f;
  for (;;) {
    const bool at_camera = (prev.surface.material_id & 3u) == MI_ENTITY_CAMERA;
    if (at_camera) bpt_connect_eye<QN>(c, cam, prev, slab, lsize, splats);
    else radiance = radiance + bpt_connect_all<QN>(c, g, prev, slab, lsize);
    const BSample b = bpt_bsdf_sample(c, g, prev.surface, prev.omega);
    for (;;) {
      surface = scene_intersect<QN>(c, surface, b.omega, 0xFFFFFFFFu);
      if (surface.material_id == 0xFFFFFFFFu)  // BPT.cpp:49-51: a camera ray that leaves the scene returns the sky gradient (Technique.cpp:86-88)
        return at_camera ? (c.sky_horizon * (1 - b.omega.z) + c.sky_zenith * b.omega.z) * c.rinv : radiance;
      cur.surface = surface; cur.omega = -b.omega;
      const Edge e = make_edge(prev.surface, cur.surface, cur.omega);
      cur.throughput = (prev.throughput * b.q.throughput) * e.bCos;
      if (l1norm(cur.throughput) < MI_FLT_EPSILON) return radiance;
      cur.throughput = cur.throughput / b.q.density;
      prev.finite = prev.finite < b.q.finite ? prev.finite : b.q.finite;
      cur.finite = b.q.finite;
      cur.c = 1.0f / betaf(c, e.fG * b.q.density);
      cur.C = (prev.C * betaf(c, b.q.densityRev) + prev.c * float(prev.finite)) * betaf(c, e.bG) * cur.c;
      if (surf_is_light(surface)) radiance = radiance + bpt_connect_light(c, cur);
      else break;
    }
    prev = cur;
    if (bpt_roulette(c, g)) return radiance;
    prev.throughput = prev.throughput * c.rinv;
  }
}


// =====================================================================================================================
// Staged form (default): the same arithmetic cut into three kernels so that the connections — the bulk of BPT's rays, one
// shadow ray each — run as a flat list of independent work items instead of inside the lane that traced the path.
//   bpt_trace     per lane: roulette, light sub-path, eye sub-path (every RNG draw in the reference's order, emission terms of
//                 _connect_light inline), vertices to path-major slabs in HBM; counts the path's connection items
//   (scan)        item offsets per path
//   bpt_items     one lane per item: (camera vertex, light vertex i) -> light-image splat; (eye vertex k, NEE sample) and
//                 (eye vertex k, light vertex i >= 1) -> value slot
//   bpt_gather    per path: the float sums in the reference's order (BPT.cpp:275-293 sums a vertex's connections locally,
//                 then adds them to the path's radiance; emission terms follow), eye image / list outputs, ray counts
// =====================================================================================================================

// path-major vertex records: 7 float4 per vertex, a path's vertices contiguous (an item's two vertices are two 112-byte reads)
MI_DEV void rec_store_l(float4* base, const LVert& x) {
  base[0] = make_float4(x.surface.position.x, x.surface.position.y, x.surface.position.z, x.a);
  base[1] = make_float4(x.surface.gnormal.x, x.surface.gnormal.y, x.surface.gnormal.z, x.A);
  base[2] = make_float4(x.surface.tangent.c0.x, x.surface.tangent.c0.y, x.surface.tangent.c0.z, __int_as_float(x.finite));
  base[3] = make_float4(x.surface.tangent.c1.x, x.surface.tangent.c1.y, x.surface.tangent.c1.z, __uint_as_float(x.surface.material_id));
  base[4] = make_float4(x.surface.tangent.c2.x, x.surface.tangent.c2.y, x.surface.tangent.c2.z, 0.f);
  base[5] = make_float4(x.omega.x, x.omega.y, x.omega.z, 0.f);
  base[6] = make_float4(x.throughput.x, x.throughput.y, x.throughput.z, 0.f);
}
MI_DEV LVert rec_load_l(const float4* base) {
  const float4 q0 = base[0], q1 = base[1], q2 = base[2], q3 = base[3], q4 = base[4], q5 = base[5], q6 = base[6];
  LVert x;
  x.surface.position = xyz(q0); x.a = q0.w;
  x.surface.gnormal = xyz(q1); x.A = q1.w;
  x.surface.tangent.c0 = xyz(q2); x.finite = __float_as_int(q2.w);
  x.surface.tangent.c1 = xyz(q3); x.surface.material_id = __float_as_uint(q3.w);
  x.surface.tangent.c2 = xyz(q4);
  x.omega = xyz(q5); x.throughput = xyz(q6);
  return x;
}
// eye vertex: like a light vertex with (c, C); .w of records 4 / 5 carry the NEE kind (0 none, 1 area, 2 directional) and the
// vertex's first item (offset inside the path's item range)
MI_DEV void rec_store_e(float4* base, const EVert& x, uint32_t nee_kind, uint32_t item0) {
  LVert t; t.surface = x.surface; t.omega = x.omega; t.throughput = x.throughput; t.a = x.c; t.A = x.C; t.finite = x.finite;
  rec_store_l(base, t);
  base[4].w = __uint_as_float(nee_kind); base[5].w = __uint_as_float(item0);
}
MI_DEV EVert rec_load_e(const float4* base, uint32_t& nee_kind, uint32_t& item0) {
  const LVert t = rec_load_l(base);
  nee_kind = __float_as_uint(base[4].w); item0 = __float_as_uint(base[5].w);
  EVert x; x.surface = t.surface; x.omega = t.omega; x.throughput = t.throughput; x.c = t.a; x.C = t.A; x.finite = t.finite;
  return x;
}
// directional NEE sample (BPT.cpp:247-273 needs the light's surface, radiance and light_density)
MI_DEV void rec_store_dir(float4* base, const LSample& b) {
  LVert t; t.surface = b.surface; t.omega = F3(0, 0, 0); t.throughput = b.radiance; t.a = b.light_density; t.A = b.area_density; t.finite = 1;
  rec_store_l(base, t);
}
MI_DEV LSample rec_load_dir(const float4* base) {
  const LVert t = rec_load_l(base);
  LSample b; b.surface = t.surface; b.radiance = t.throughput; b.light_density = t.a; b.area_density = t.A; b.directional = true;
  return b;
}

struct Lane { uint32_t px, py, fl; uint64_t sample; bool ok; };
template <bool LIST>
MI_DEV Lane lane_decode(const RenderParams& p, const BptState& w, uint32_t i) {
  Lane l; l.px = l.py = l.fl = 0; l.sample = 0; l.ok = false;
  if (LIST) {
    const uint64_t item = w.first + i;
    l.ok = i < w.lanes && item < p.list_n;
    if (l.ok) { l.px = p.list_xy[2 * item]; l.py = p.list_xy[2 * item + 1]; l.sample = p.list_sample[item]; }
  } else {
    const uint32_t per_frame = p.tiles_x * p.tiles_y * 64u;
    const uint32_t gi = w.path_ids ? (i < w.lanes ? w.path_ids[i] : 0u) : w.first + i;
    l.fl = gi / per_frame;
    const uint32_t rem = gi - l.fl * per_frame, tile = rem >> 6, pix = rem & 63u;
    const uint32_t ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    l.px = p.win_x0 + tx * 8u + (pix & 7u); l.py = p.win_y0 + ty * 8u + (pix >> 3);
    l.sample = p.sample_offset + w.frame + l.fl;
    l.ok = i < w.lanes && l.fl < w.frames && l.px < p.win_x0 + p.win_w && l.py < p.win_y0 + p.win_h;
  }
  return l;
}
MI_DEV void ctx_init(Ctx& c, const RenderParams& p, const BptState& w, void* stack, const float4* sb, const SceneView* sv) {
  c.sb = sb; c.sv = sv; c.stack = stack;
  c.flat_table = p.flat_table; c.flat_k = p.flat_k; c.flat_k_mesh = p.flat_k_mesh;
  c.beta = p.beta; c.roulette = p.roulette; c.rinv = 1.0f / p.roulette;
  c.sphere_c = F3(w.sphere[0], w.sphere[1], w.sphere[2]); c.sphere_r = w.sphere[3];
  c.sky_horizon = F3(w.sky_horizon[0], w.sky_horizon[1], w.sky_horizon[2]); c.sky_zenith = F3(w.sky_zenith[0], w.sky_zenith[1], w.sky_zenith[2]);
  c.n_basic = 0; c.n_shadow = 0;
}

// a sub-path outgrew the slab share: set the path aside for the batch's launch at the reference's capacity (BptState::over_ids), or count it for the host
MI_DEV bool set_aside(const BptState& w, uint32_t i, bool& overflow) {
  if (!w.over_ids) { overflow = true; return false; }
  const uint32_t slot = atomicAdd(w.over_count, 1u);
  w.over_ids[slot] = w.path_ids ? w.path_ids[i] : w.first + i;  // the list holds every path of the batch (host): no bound to test
  return true;
}

}  // namespace

// ---- stage A ----
#ifndef MI_BPT_TRACE_WAVES
#define MI_BPT_TRACE_WAVES 4
#endif
#ifndef MI_BPT_ITEMS_WAVES
#define MI_BPT_ITEMS_WAVES 6  // tools/sessions/ab_bpt_stage_waves.sh, 512^2 x 32: items 3/4/5/6 waves = 53.9/56.6/54.4/55.4 ms (Cornell), 280/281/282/272 ms (LivingRoomLit);
#endif                        // trace 3/4/5 waves = 61.4/56.6/62.4 and 303/281/286 ms
// WAVES: the register budget (waves per SIMD).  r04, 1 M paths per launch: six waves (80 VGPRs) beat four (128) on the models whose walk is a chain of dependent
// fetches from L2 / HBM (LivingRoomLit 219 -> 207 ms, MetalRings 84.6 -> 82 ms per 64 frames) and lose on small trees (CornellBoxSpecular 129 -> 136 ms): chosen by
// the size of the scene (bpt_stage_trace)
template <bool LIST, int QN, int WAVES = MI_BPT_TRACE_WAVES>
__global__ __launch_bounds__(kBlock, WAVES) void bpt_trace(const RenderParams p, const BptState w) {
  extern __shared__ float4 smem[];
  SceneView sv = p.sv;
  const float4* sb = sv.blob;
  uint32_t scene_f4 = 0;
  if (QN == 0) {
    if (p.flat_k) { scene_f4 = flat_scene_f4(sv, p.flat_k); stage_scene_flat(smem, sv, p.flat_table, p.flat_k, threadIdx.x); }
    else { scene_f4 = lds_scene_f4(sv); stage_scene_to_lds(smem, sv, threadIdx.x); }
    sb = smem; __syncthreads();
  }
  TravStackT<(QN != 0)> stack;
  stack.lds = (lds_u32*)(reinterpret_cast<uint32_t*>(smem + scene_f4) + threadIdx.x);
  stack.cap = p.stack_entries;
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  const Lane ln = lane_decode<LIST>(p, w, i);
  bool overflow = false;
  if (i < w.lanes) {
    uint32_t L = 0, E = 0, n_items = 0, n_em = 0, n_dir = 0, basic = 0;
    bool aside = false;
    if (ln.ok) {
      Ctx c; ctx_init(c, p, w, &stack, sb, &sv);
      const m33 v2w = {F3(p.v2w[0], p.v2w[1], p.v2w[2]), F3(p.v2w[3], p.v2w[4], p.v2w[5]), F3(p.v2w[6], p.v2w[7], p.v2w[8])};
      Surf cs;  // Technique::_camera_surface (Technique.cpp:107-116)
      cs.position = F3(p.cam_pos[0], p.cam_pos[1], p.cam_pos[2]);
      cs.tangent.c0 = v2w.c1; cs.tangent.c1 = -v2w.c2; cs.tangent.c2 = v2w.c0;
      cs.material_id = (0u << 2) | MI_ENTITY_CAMERA;
      cs.gnormal = -v2w.c2;
      Rng g = rng_seed(p.seed, ln.py * p.width + ln.px, ln.sample);
      const float u0 = rng_f(g), u1 = rng_f(g);
      const float fx = float(ln.px) + u0, fy = float(ln.py) + u1;
      const float vx = fx * p.res_y_inv * 2.0f - p.res_x * p.res_y_inv;
      const float vy = fy * p.res_y_inv * 2.0f - 1.0f;
      const f3 dir = mulmv(v2w, normalize(F3(vx, vy, -p.focal_length_y)));
      float4* lrec = w.lslab + size_t(i) * w.max_vertices * 7u;
      float4* erec = w.eslab + size_t(i) * w.max_vertices * 7u;
      float4* nrec = w.nslab + size_t(i) * w.max_vertices * 7u;
      float4* em = w.emission + size_t(i) * w.max_vertices;  // emission terms (eye sub-path hits on emitters): as many as vertices
      uint2* evi = w.evinfo + size_t(i) * w.max_vertices;
      if (!bpt_roulette(c, g)) {  // BPT.cpp:17-19
        // ---- _traceLight (BPT.cpp:121-190) ----
        if (!bpt_roulette(c, g)) {
          const LSample ls = light_sample(c, g);
          LVert prev = sample_to_vertex(c, ls);
          uint32_t size = 1, prv = 0;
          while (!bpt_roulette(c, g)) {
            const BSample b = bpt_bsdf_sample(c, g, prev.surface, prev.omega);
            const Surf surface = scene_intersect<QN>(c, prev.surface, b.omega, 1u << MI_ENTITY_MESH);
            if (surface.material_id == 0xFFFFFFFFu) break;
            if (size >= w.max_vertices) { aside = set_aside(w, i, overflow); break; }
            LVert cur;
            cur.surface = surface;
            cur.omega = -b.omega;
            const Edge e = make_edge(prev.surface, cur.surface, cur.omega);
            cur.throughput = ((prev.throughput * b.q.throughput) * e.bCos) * c.rinv;
            if (l1norm(cur.throughput) < MI_FLT_EPSILON) break;
            cur.throughput = cur.throughput / b.q.density;
            prev.finite = prev.finite < b.q.finite ? prev.finite : b.q.finite;
            cur.finite = b.q.finite;
            cur.a = 1.0f / betaf(c, e.fG * b.q.density);
            cur.A = (prev.A * betaf(c, b.q.densityRev) + prev.a * float(prev.finite)) * betaf(c, e.bG) * cur.a;
            if (b.q.finite == 0) { prev = cur; }
            else { rec_store_l(lrec + size_t(prv) * 7u, prev); prev = cur; prv = size; ++size; }
          }
          const BSample last = bpt_bsdf_sample(c, g, prev.surface, prev.omega);
          if (last.q.finite == 0) --size; else rec_store_l(lrec + size_t(prv) * 7u, prev);
          L = size;
        }
        // ---- _traceEye without the connections (BPT.cpp:24-98) ----
        EVert prev, cur;
        Surf surface = cs;
        prev.surface = surface; prev.omega = -dir; prev.throughput = F3(1, 1, 1) * c.rinv;
        prev.finite = 1; prev.c = 0.0f; prev.C = 0.0f;
        if (!aside) for (;;) {
          const bool at_camera = (prev.surface.material_id & 3u) == MI_ENTITY_CAMERA;
          if (E >= w.max_vertices) { aside = set_aside(w, i, overflow); break; }
          if (at_camera) {
            rec_store_e(erec + size_t(E) * 7u, prev, 0u, 0u);  // items [0, L): the splats of _connect_eye
            evi[E] = make_uint2(0u, 0u);
            n_items = L;
          } else {
            uint32_t kind = 0;
            if (!bpt_roulette(c, g)) {  // BPT.cpp:278-288
              const LSample b = light_sample(c, g);
              if (!b.directional) { kind = 1; rec_store_l(nrec + size_t(E) * 7u, sample_to_vertex(c, b)); }
              else { kind = 2; rec_store_dir(nrec + size_t(E) * 7u, b); ++n_dir; }
            }
            rec_store_e(erec + size_t(E) * 7u, prev, kind, n_items);
            evi[E] = make_uint2(kind, n_items);
            n_items += (kind ? 1u : 0u) + (L > 1u ? L - 1u : 0u);
          }
          const uint32_t k = E;
          ++E;
          const BSample b = bpt_bsdf_sample(c, g, prev.surface, prev.omega);
          bool ended = false;
          for (;;) {
            surface = scene_intersect<QN>(c, surface, b.omega, 0xFFFFFFFFu);
            if (surface.material_id == 0xFFFFFFFFu) {
              if (at_camera) {  // BPT.cpp:49-51: a miss from the camera vertex returns the sky gradient instead of what was gathered through emitters
                const f3 sky = (c.sky_horizon * (1 - b.omega.z) + c.sky_zenith * b.omega.z) * c.rinv;
                em[0] = make_float4(sky.x, sky.y, sky.z, __uint_as_float(0u)); n_em = 1;
              }
              ended = true; break;
            }
            cur.surface = surface; cur.omega = -b.omega;
            const Edge e = make_edge(prev.surface, cur.surface, cur.omega);
            cur.throughput = (prev.throughput * b.q.throughput) * e.bCos;
            if (l1norm(cur.throughput) < MI_FLT_EPSILON) { ended = true; break; }
            cur.throughput = cur.throughput / b.q.density;
            prev.finite = prev.finite < b.q.finite ? prev.finite : b.q.finite;
            cur.finite = b.q.finite;
            cur.c = 1.0f / betaf(c, e.fG * b.q.density);
            cur.C = (prev.C * betaf(c, b.q.densityRev) + prev.c * float(prev.finite)) * betaf(c, e.bG) * cur.c;
            if (surf_is_light(surface)) {
              const f3 t = bpt_connect_light(c, cur);
              if (n_em >= w.max_vertices) { aside = set_aside(w, i, overflow); ended = true; break; }
              em[n_em++] = make_float4(t.x, t.y, t.z, __uint_as_float(k));
            } else break;
          }
          if (ended) break;
          prev = cur;
          if (bpt_roulette(c, g)) break;
          prev.throughput = prev.throughput * c.rinv;
        }
      }
      basic = c.n_basic;
    }
    if (aside) { L = E = n_items = n_em = n_dir = basic = 0u; }  // set aside: nothing of it counts in this launch
    w.info[2 * size_t(i)] = make_uint4(L, E, n_items, n_em);
    w.info[2 * size_t(i) + 1] = make_uint4(basic, n_dir, (ln.py << 16) | ln.px, (ln.fl << 1) | (ln.ok && !aside ? 1u : 0u));
    w.item_offset[i] = n_items;
  }
  uint32_t o = overflow ? 1u : 0u;
  for (int k = 32; k > 0; k >>= 1) o += __shfl_xor(o, k, 64);
  if (o && (threadIdx.x & 63u) == 0 && p.counters) atomicAdd(&p.counters[15], (unsigned long long)o);
}

// =====================================================================================================================
// Stage A with PATH REGENERATION (r04, the default): bpt_trace above gives a lane one path — a wave runs as long as its longest light sub-path plus its
// longest eye sub-path (3.5 vertices on average, dozens at the tail: 9 % of the lanes of an issued instruction were active, profiles/r04/pmc_bpt_livingroom.txt).
// Here the two sub-paths are two kernels of resident waves, and a trip of a wave's loop extends every live sub-path by ONE vertex; a lane whose sub-path has
// ended takes the next path of the launch from a cursor (one atomic per wave and trip), as pt_megakernel regenerates camera paths.  The loop bodies are
// bpt_trace's, draw for draw: a path's records, counts and generator states are the same whatever lane walks it, everything downstream is unchanged.
// Between the kernels a path's info[2 i] holds (L, generator state after the light sub-path, closest-hit rays so far, 1 = the path passed BPT.cpp:17-19).
template <bool LIST, int QN, int WAVES = MI_BPT_TRACE_WAVES>
__global__ __launch_bounds__(kBlock, WAVES) void bpt_trace_light(const RenderParams p, const BptState w) {
  extern __shared__ float4 smem[];
  SceneView sv = p.sv;
  const float4* sb = sv.blob;
  uint32_t scene_f4 = 0;
  if (QN == 0) {
    if (p.flat_k) { scene_f4 = flat_scene_f4(sv, p.flat_k); stage_scene_flat(smem, sv, p.flat_table, p.flat_k, threadIdx.x); }
    else { scene_f4 = lds_scene_f4(sv); stage_scene_to_lds(smem, sv, threadIdx.x); }
    sb = smem; __syncthreads();
  }
  TravStackT<(QN != 0)> stack;
  stack.lds = (lds_u32*)(reinterpret_cast<uint32_t*>(smem + scene_f4) + threadIdx.x);
  stack.cap = p.stack_entries;
  Ctx c; ctx_init(c, p, w, &stack, sb, &sv);
  uint32_t* cursor = w.step_count + 4;
  bool alive = false, ending = false, more = true, overflow = false;
  uint32_t i = 0, size = 0, prv = 0;
  Rng g; g.state = 0u;
  LVert prev;
  prev.surface.position = prev.surface.gnormal = prev.surface.tangent.c0 = prev.surface.tangent.c1 = prev.surface.tangent.c2 = F3(0, 0, 0);
  prev.surface.material_id = 0u; prev.omega = prev.throughput = F3(0, 0, 0); prev.a = prev.A = 0.0f; prev.finite = 1;
  for (;;) {
    if (alive && ending) {  // BPT.cpp:178-189: the last vertex stays only if a path can be connected to it
      const BSample last = bpt_bsdf_sample(c, g, prev.surface, prev.omega);
      float4* lrec = w.lslab + size_t(i) * w.max_vertices * 7u;
      if (last.q.finite == 0) --size; else rec_store_l(lrec + size_t(prv) * 7u, prev);
      w.info[2 * size_t(i)] = make_uint4(size, g.state, c.n_basic, 1u);
      alive = false;
    }
    for (;;) {  // idle lanes take the next paths of the launch; a path that ends before its first vertex (19 % of them) leaves the lane idle for another round
      const bool want = !alive && more;
      const uint64_t m = __ballot(want);
      if (m == 0ull) break;
      uint32_t base = 0u;
      const uint32_t leader = uint32_t(__builtin_ctzll(m));
      if ((threadIdx.x & 63u) == leader) base = atomicAdd(cursor, uint32_t(__popcll(m)));
      base = uint32_t(__shfl(int(base), int(leader), 64));
      if (want) {
        i = base + __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
        if (i >= w.lanes) more = false;
        else {
          const Lane ln = lane_decode<LIST>(p, w, i);
          uint4 res = make_uint4(0u, 0u, 0u, 0u);
          if (ln.ok) {
            g = rng_seed(p.seed, ln.py * p.width + ln.px, ln.sample);
            (void)rng_f(g); (void)rng_f(g);  // the camera sample's two draws come first (bpt_trace_eye draws them again)
            if (!bpt_roulette(c, g)) {  // BPT.cpp:17-19
              res = make_uint4(0u, 0u, 0u, 1u);
              if (!bpt_roulette(c, g)) {  // _traceLight (BPT.cpp:121-190)
                const LSample ls = light_sample(c, g);
                prev = sample_to_vertex(c, ls);
                size = 1u; prv = 0u; c.n_basic = 0u; alive = true; ending = false;
              }
              res.y = g.state;
            }
          }
          if (!alive) w.info[2 * size_t(i)] = res;
        }
      }
    }
    if (!__any(alive)) break;  // no lane holds a sub-path and none could take a path: the launch is dealt
    if (alive && !ending) {
      if (bpt_roulette(c, g)) ending = true;
      else {
        const BSample b = bpt_bsdf_sample(c, g, prev.surface, prev.omega);
        const Surf surface = scene_intersect<QN>(c, prev.surface, b.omega, 1u << MI_ENTITY_MESH);
        if (surface.material_id == 0xFFFFFFFFu) ending = true;
        else if (size >= w.max_vertices) {
          if (set_aside(w, i, overflow)) { w.info[2 * size_t(i)] = make_uint4(0u, 0u, 0u, 2u); alive = false; }  // 2: bpt_trace_eye leaves the path alone
          else ending = true;
        }
        else {
          LVert cur;
          cur.surface = surface;
          cur.omega = -b.omega;
          const Edge e = make_edge(prev.surface, cur.surface, cur.omega);
          cur.throughput = ((prev.throughput * b.q.throughput) * e.bCos) * c.rinv;
          if (l1norm(cur.throughput) < MI_FLT_EPSILON) ending = true;
          else {
            cur.throughput = cur.throughput / b.q.density;
            prev.finite = prev.finite < b.q.finite ? prev.finite : b.q.finite;
            cur.finite = b.q.finite;
            cur.a = 1.0f / betaf(c, e.fG * b.q.density);
            cur.A = (prev.A * betaf(c, b.q.densityRev) + prev.a * float(prev.finite)) * betaf(c, e.bG) * cur.a;
            if (b.q.finite == 0) { prev = cur; }
            else { rec_store_l(w.lslab + (size_t(i) * w.max_vertices + prv) * 7u, prev); prev = cur; prv = size; ++size; }
          }
        }
      }
    }
  }
  uint32_t o = overflow ? 1u : 0u;
  for (int k = 32; k > 0; k >>= 1) o += __shfl_xor(o, k, 64);
  if (o && (threadIdx.x & 63u) == 0 && p.counters) atomicAdd(&p.counters[15], (unsigned long long)o);
}

template <bool LIST, int QN, int WAVES = MI_BPT_TRACE_WAVES>
__global__ __launch_bounds__(kBlock, WAVES) void bpt_trace_eye(const RenderParams p, const BptState w) {
  extern __shared__ float4 smem[];
  SceneView sv = p.sv;
  const float4* sb = sv.blob;
  uint32_t scene_f4 = 0;
  if (QN == 0) {
    if (p.flat_k) { scene_f4 = flat_scene_f4(sv, p.flat_k); stage_scene_flat(smem, sv, p.flat_table, p.flat_k, threadIdx.x); }
    else { scene_f4 = lds_scene_f4(sv); stage_scene_to_lds(smem, sv, threadIdx.x); }
    sb = smem; __syncthreads();
  }
  TravStackT<(QN != 0)> stack;
  stack.lds = (lds_u32*)(reinterpret_cast<uint32_t*>(smem + scene_f4) + threadIdx.x);
  stack.cap = p.stack_entries;
  Ctx c; ctx_init(c, p, w, &stack, sb, &sv);
  uint32_t* cursor = w.step_count + 5;
  const m33 v2w = {F3(p.v2w[0], p.v2w[1], p.v2w[2]), F3(p.v2w[3], p.v2w[4], p.v2w[5]), F3(p.v2w[6], p.v2w[7], p.v2w[8])};
  Surf cs;  // Technique::_camera_surface (Technique.cpp:107-116)
  cs.position = F3(p.cam_pos[0], p.cam_pos[1], p.cam_pos[2]);
  cs.tangent.c0 = v2w.c1; cs.tangent.c1 = -v2w.c2; cs.tangent.c2 = v2w.c0;
  cs.material_id = (0u << 2) | MI_ENTITY_CAMERA;
  cs.gnormal = -v2w.c2;
  bool alive = false, more = true, overflow = false;
  uint32_t i = 0, L = 0, E = 0, n_items = 0, n_em = 0, n_dir = 0, pxy = 0, flk = 0;
  Rng g; g.state = 0u;
  EVert prev;
  prev.surface = cs; prev.omega = prev.throughput = F3(0, 0, 0); prev.c = prev.C = 0.0f; prev.finite = 1;
  for (;;) {
    for (;;) {
      const bool want = !alive && more;
      const uint64_t m = __ballot(want);
      if (m == 0ull) break;
      uint32_t base = 0u;
      const uint32_t leader = uint32_t(__builtin_ctzll(m));
      if ((threadIdx.x & 63u) == leader) base = atomicAdd(cursor, uint32_t(__popcll(m)));
      base = uint32_t(__shfl(int(base), int(leader), 64));
      if (want) {
        i = base + __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
        if (i >= w.lanes) more = false;
        else {
          const Lane ln = lane_decode<LIST>(p, w, i);
          pxy = (ln.py << 16) | ln.px; flk = (ln.fl << 1) | (ln.ok ? 1u : 0u);
          uint4 res = make_uint4(0u, 0u, 0u, 0u);
          if (ln.ok) res = w.info[2 * size_t(i)];
          if (res.w == 2u) { res.w = 0u; flk &= ~1u; }  // set aside by bpt_trace_light
          if (res.w != 0u) {
            g = rng_seed(p.seed, ln.py * p.width + ln.px, ln.sample);
            const float u0 = rng_f(g), u1 = rng_f(g);
            const float fx = float(ln.px) + u0, fy = float(ln.py) + u1;
            const float vx = fx * p.res_y_inv * 2.0f - p.res_x * p.res_y_inv;
            const float vy = fy * p.res_y_inv * 2.0f - 1.0f;
            const f3 dir = mulmv(v2w, normalize(F3(vx, vy, -p.focal_length_y)));
            g.state = res.y; L = res.x; c.n_basic = res.z;
            E = 0u; n_items = 0u; n_em = 0u; n_dir = 0u;
            prev.surface = cs; prev.omega = -dir; prev.throughput = F3(1, 1, 1) * c.rinv;
            prev.finite = 1; prev.c = 0.0f; prev.C = 0.0f;
            alive = true;
          } else {
            w.info[2 * size_t(i)] = make_uint4(0u, 0u, 0u, 0u);
            w.info[2 * size_t(i) + 1] = make_uint4(0u, 0u, pxy, flk);
            w.item_offset[i] = 0u;
          }
        }
      }
    }
    if (!__any(alive)) break;
    if (alive) {  // one vertex of _traceEye without the connections (BPT.cpp:24-98)
      float4* erec = w.eslab + size_t(i) * w.max_vertices * 7u;
      float4* nrec = w.nslab + size_t(i) * w.max_vertices * 7u;
      float4* em = w.emission + size_t(i) * w.max_vertices;
      uint2* evi = w.evinfo + size_t(i) * w.max_vertices;
      bool ended = false, aside = false;
      if (E >= w.max_vertices) { aside = set_aside(w, i, overflow); ended = true; }
      else {
        const bool at_camera = (prev.surface.material_id & 3u) == MI_ENTITY_CAMERA;
        if (at_camera) {
          rec_store_e(erec + size_t(E) * 7u, prev, 0u, 0u);  // items [0, L): the splats of _connect_eye
          evi[E] = make_uint2(0u, 0u);
          n_items = L;
        } else {
          uint32_t kind = 0;
          if (!bpt_roulette(c, g)) {  // BPT.cpp:278-288
            const LSample b = light_sample(c, g);
            if (!b.directional) { kind = 1; rec_store_l(nrec + size_t(E) * 7u, sample_to_vertex(c, b)); }
            else { kind = 2; rec_store_dir(nrec + size_t(E) * 7u, b); ++n_dir; }
          }
          rec_store_e(erec + size_t(E) * 7u, prev, kind, n_items);
          evi[E] = make_uint2(kind, n_items);
          n_items += (kind ? 1u : 0u) + (L > 1u ? L - 1u : 0u);
        }
        const uint32_t k = E;
        ++E;
        const BSample b = bpt_bsdf_sample(c, g, prev.surface, prev.omega);
        Surf surface = prev.surface;
        EVert cur = prev;
        for (;;) {
          surface = scene_intersect<QN>(c, surface, b.omega, 0xFFFFFFFFu);
          if (surface.material_id == 0xFFFFFFFFu) {
            if (at_camera) {  // BPT.cpp:49-51
              const f3 sky = (c.sky_horizon * (1 - b.omega.z) + c.sky_zenith * b.omega.z) * c.rinv;
              em[0] = make_float4(sky.x, sky.y, sky.z, __uint_as_float(0u)); n_em = 1;
            }
            ended = true; break;
          }
          cur.surface = surface; cur.omega = -b.omega;
          const Edge e = make_edge(prev.surface, cur.surface, cur.omega);
          cur.throughput = (prev.throughput * b.q.throughput) * e.bCos;
          if (l1norm(cur.throughput) < MI_FLT_EPSILON) { ended = true; break; }
          cur.throughput = cur.throughput / b.q.density;
          prev.finite = prev.finite < b.q.finite ? prev.finite : b.q.finite;
          cur.finite = b.q.finite;
          cur.c = 1.0f / betaf(c, e.fG * b.q.density);
          cur.C = (prev.C * betaf(c, b.q.densityRev) + prev.c * float(prev.finite)) * betaf(c, e.bG) * cur.c;
          if (surf_is_light(surface)) {
            const f3 t = bpt_connect_light(c, cur);
            if (n_em >= w.max_vertices) { aside = set_aside(w, i, overflow); ended = true; break; }
            em[n_em++] = make_float4(t.x, t.y, t.z, __uint_as_float(k));
          } else break;
        }
        if (!ended) {
          prev = cur;
          if (bpt_roulette(c, g)) ended = true;
          else prev.throughput = prev.throughput * c.rinv;
        }
      }
      if (ended) {
        if (aside) { L = E = n_items = n_em = n_dir = 0u; c.n_basic = 0u; flk &= ~1u; }  // set aside: nothing of it counts in this launch
        w.info[2 * size_t(i)] = make_uint4(L, E, n_items, n_em);
        w.info[2 * size_t(i) + 1] = make_uint4(c.n_basic, n_dir, pxy, flk);
        w.item_offset[i] = n_items;
        alive = false;
      }
    }
  }
  uint32_t o = overflow ? 1u : 0u;
  for (int k = 32; k > 0; k >>= 1) o += __shfl_xor(o, k, 64);
  if (o && (threadIdx.x & 63u) == 0 && p.counters) atomicAdd(&p.counters[15], (unsigned long long)o);
}

// =====================================================================================================================
// Stage A as UNIFORM STEPS (r04; scenes read from HBM).  bpt_trace above walks a whole light sub-path and a whole eye sub-path per lane: a wave runs as
// long as its longest sub-path (3.5 vertices on average, dozens at the tail) and every scene_intersect is a per-lane tree walk inside that loop —
// 9 % of the lanes of an issued instruction were active (profiles/r03/pmc_bpt_livingroom.txt).  Here a path is a coroutine that is suspended at each
// closest-hit ray:  bpt_step  resumes every path that has a hit waiting, runs it to its next ray (roulette, BSDF sample, vertex records, emission
// terms — bounded, traversal-free work) and appends it to the list of paths in flight;  bpt_closest  walks the rays of that list.  Rounds repeat until
// the list is empty.  Same draws in the same order, same arithmetic, same records as bpt_trace: everything downstream (scan, items, gather) is unchanged.
//
// Path state between two rays (kBptStepF4 = 16 float4): [0] phase, rng, L, E  [1] items, emission terms, directional NEE, rays  [2] size, prv, k, flags
// (1 at_camera, 2 b.finite)  [3..9] prev (record format of rec_store_l: LVert in the light phase, EVert in the eye phase)  [10..13] unused  [14] b.omega | b.density
// [15] b.throughput | b.densityRev.  The pending ray (nudged origin | geometry mask, direction) and its hit live in step_rays / step_hits.
constexpr uint32_t kStepStart = 0, kStepLight = 1, kStepEye = 2;
#ifndef MI_BPT_STEP_WAVES
#define MI_BPT_STEP_WAVES 4
#endif
#ifndef MI_BPT_STEP_ROUNDS
#define MI_BPT_STEP_ROUNDS 24  // rounds of (walk, step) before the tail kernel takes the paths still in flight (0.9^24 = 8 % of the rays of a sub-path lie beyond)
#endif
MI_DEV void step_store_surf(float4* q, const Surf& s) {
  q[0] = make_float4(s.position.x, s.position.y, s.position.z, __uint_as_float(s.material_id));
  q[1] = make_float4(s.gnormal.x, s.gnormal.y, s.gnormal.z, s.tangent.c0.x);
  q[2] = make_float4(s.tangent.c0.y, s.tangent.c0.z, s.tangent.c1.x, s.tangent.c1.y);
  q[3] = make_float4(s.tangent.c1.z, s.tangent.c2.x, s.tangent.c2.y, s.tangent.c2.z);
}
MI_DEV Surf step_load_surf(const float4* q) {
  const float4 a = q[0], b = q[1], c = q[2], d = q[3];
  Surf s;
  s.position = xyz(a); s.material_id = __float_as_uint(a.w);
  s.gnormal = xyz(b);
  s.tangent.c0 = F3(b.w, c.x, c.y); s.tangent.c1 = F3(c.z, c.w, d.x); s.tangent.c2 = F3(d.y, d.z, d.w);
  return s;
}
template <bool ON> struct StepStack { TravStackT<true> s; };
template <> struct StepStack<false> {};
MI_DEV EVert lvert_as_evert(const LVert& t) { EVert x; x.surface = t.surface; x.omega = t.omega; x.throughput = t.throughput; x.c = t.a; x.C = t.A; x.finite = t.finite; return x; }
MI_DEV LVert evert_as_lvert(const EVert& x) { LVert t; t.surface = x.surface; t.omega = x.omega; t.throughput = x.throughput; t.a = x.c; t.A = x.C; t.finite = x.finite; return t; }

// in_slot < 0: the first round (every path of the launch starts); else the paths of step_active[in_slot] (step_count[in_slot] of them, read on the device: no
// round waits for the host) resume.  INLINE (QN = node format): the tail — a resumed path walks its own rays here and runs to its end, nothing is appended.
template <bool LIST, bool INLINE, int QN>
// budget (INLINE only; 0 = none): a path that has walked this many rays in this launch is suspended like in the rounds' kernel — its ray saved UNWALKED — and
// continues in the next pass (pending_unwalked = 1 there: the resumed path walks that ray first).  Passes with growing budgets bound the time a wave waits for
// its longest path while the survivors of a pass are packed into full waves again (bpt_stage_trace_passes).
__global__ __launch_bounds__(kBlock, INLINE ? 4 : MI_BPT_STEP_WAVES) void bpt_step(const RenderParams p, const BptState w, int in_slot, uint32_t out_slot, uint32_t budget,
                                                                                   uint32_t pending_unwalked) {
  extern __shared__ float4 smem[];
  StepStack<INLINE> stack_holder;  // only the tail walks rays itself: the rounds' kernel carries no traversal stack (its 512-byte spill array would be scratch)
  void* stack_ptr = nullptr;
  if constexpr (INLINE) {
    stack_holder.s.lds = (lds_u32*)(reinterpret_cast<uint32_t*>(smem) + threadIdx.x);
    stack_holder.s.cap = p.stack_entries;
    stack_ptr = &stack_holder.s;
  }
  const uint32_t* __restrict__ active_in = in_slot < 0 ? nullptr : w.step_active[in_slot];
  const uint32_t n_in = in_slot < 0 ? w.lanes : w.step_count[in_slot];
  uint32_t n_overflow = 0;
  for (uint32_t base = blockIdx.x * kBlock; base < n_in; base += gridDim.x * kBlock) {  // wave-uniform trip count: the ballots below see whole waves
  const uint32_t j = base + threadIdx.x;
  const bool have = j < n_in;
  const uint32_t i = have ? (active_in ? active_in[j] : j) : 0u;
  bool emit = false, overflow = false;
  if (have) {
    SceneView sv = p.sv;
    Ctx c; ctx_init(c, p, w, stack_ptr, sv.blob, &sv);
    c.flat_k = 0u; c.flat_k_mesh = 0u;
    const Lane ln = lane_decode<LIST>(p, w, i);
    float4* st = w.step_state + size_t(i) * kBptStepF4;
    float4* lrec = w.lslab + size_t(i) * w.max_vertices * 7u;
    float4* erec = w.eslab + size_t(i) * w.max_vertices * 7u;
    float4* nrec = w.nslab + size_t(i) * w.max_vertices * 7u;
    float4* em = w.emission + size_t(i) * w.max_vertices;
    uint2* evi = w.evinfo + size_t(i) * w.max_vertices;
    uint32_t phase = kStepStart, L = 0, E = 0, n_items = 0, n_em = 0, n_dir = 0, basic = 0, size = 0, prv = 0, k = 0;
    bool at_camera = false;
    Rng g; g.state = 0u;
    LVert pv; Surf surface; BSample b;  // pv: the previous vertex of the sub-path being traced (an EVert's c / C ride in a / A: same record, one set of registers)
    pv.surface.position = F3(0, 0, 0); pv.surface.gnormal = F3(0, 0, 0); pv.surface.tangent.c0 = pv.surface.tangent.c1 = pv.surface.tangent.c2 = F3(0, 0, 0);
    pv.surface.material_id = 0u; pv.omega = F3(0, 0, 0); pv.throughput = F3(0, 0, 0); pv.a = pv.A = 0.0f; pv.finite = 1;
    surface = pv.surface;
    b.q = bq_zero(); b.omega = F3(0, 0, 0);
    if (active_in) {
      const uint4 s0 = reinterpret_cast<const uint4*>(st)[0], s1 = reinterpret_cast<const uint4*>(st)[1], s2 = reinterpret_cast<const uint4*>(st)[2];
      phase = s0.x; g.state = s0.y; L = s0.z; E = s0.w;
      n_items = s1.x; n_em = s1.y; n_dir = s1.z; basic = s1.w;
      size = s2.x; prv = s2.y; k = s2.z; at_camera = (s2.w & 1u) != 0u;
      pv = rec_load_l(st + 3);
      const float4 b0 = st[14], b1 = st[15];
      b.omega = xyz(b0); b.q.density = b0.w; b.q.throughput = xyz(b1); b.q.densityRev = b1.w; b.q.finite = (s2.w & 2u) ? 1 : 0;
      // Scene::intersect's second half (Scene.cpp:198-202): the surface point of the hit the walk found for this path's ray
      const float4 ro = w.step_rays[2 * size_t(i)], rd = w.step_rays[2 * size_t(i) + 1];
      Hit h; h.t = __builtin_inff(); h.u = h.v = 0.0f; h.id = 0xFFFFFFFFu; h.pos = 0; h.den = 1.0f;
      bool walked = false;
      if constexpr (INLINE) {
        if (pending_unwalked) {  // the ray a budgeted pass left behind: rtcIntersect of Scene::intersect (Scene.cpp:198) with its mask, as scene_intersect does
          traverse<false, false, QN, 4>(sv.blob, sv, stack_holder.s, xyz(ro), xyz(rd), __float_as_uint(ro.w), h);
          walked = true;
        }
      }
      if (!walked) { const float4 hv = w.step_hits[i]; h.t = hv.x; h.u = hv.y; h.v = hv.z; h.pos = __float_as_uint(hv.w); h.id = h.pos == 0xFFFFFFFFu ? 0xFFFFFFFFu : 0u; }
      // `surface` = what the path's ray found (material_id 0xFFFFFFFF: nothing); the surface the ray left from is not needed again
      if (h.id == 0xFFFFFFFFu) { surface.material_id = 0xFFFFFFFFu; }
      else surface = query_surface<8>(sv.blob, sv, xyz(ro), xyz(rd), h);
    }
    // the coroutine: pc names the point of bpt_trace the path continues at
    enum { PC_START, PC_LIGHT_HEAD, PC_LIGHT_RESUME, PC_LIGHT_END, PC_EYE_START, PC_EYE_HEAD, PC_EYE_RESUME, PC_END, PC_SUSPEND };
    int pc = phase == kStepStart ? PC_START : (phase == kStepLight ? PC_LIGHT_RESUME : PC_EYE_RESUME);
    f3 ray_dir = F3(0, 0, 1), ray_pos = F3(0, 0, 0), ray_gn = F3(0, 0, 1); uint32_t ray_mask = 0xFFFFFFFFu;  // the ray about to leave: Scene::intersect(from, dir) with its mask
    bool done = false;
    uint32_t rays_here = 0;  // rays this path has walked in this launch (INLINE with a budget)
    while (!done) {
      switch (pc) {
        case PC_START: {
          if (!ln.ok) { pc = PC_END; break; }
          g = rng_seed(p.seed, ln.py * p.width + ln.px, ln.sample);
          (void)rng_f(g); (void)rng_f(g);  // the camera sample's two uniforms (re-drawn at PC_EYE_START)
          if (bpt_roulette(c, g)) { pc = PC_END; break; }  // BPT.cpp:17-19
          if (bpt_roulette(c, g)) { pc = PC_EYE_START; break; }  // _traceLight (BPT.cpp:121-190) returns an empty sub-path
          const LSample ls = light_sample(c, g);
          pv = sample_to_vertex(c, ls);
          size = 1; prv = 0;
          pc = PC_LIGHT_HEAD;
        } break;
        case PC_LIGHT_HEAD: {
          if (bpt_roulette(c, g)) { pc = PC_LIGHT_END; break; }
          b = bpt_bsdf_sample(c, g, pv.surface, pv.omega);
          ray_pos = pv.surface.position; ray_gn = pv.surface.gnormal; ray_dir = b.omega; ray_mask = 1u << MI_ENTITY_MESH;
          phase = kStepLight; pc = PC_SUSPEND;
        } break;
        case PC_LIGHT_RESUME: {
          if (surface.material_id == 0xFFFFFFFFu) { pc = PC_LIGHT_END; break; }
          if (size >= w.max_vertices) { overflow = true; pc = PC_LIGHT_END; break; }
          LVert cur;
          cur.surface = surface;
          cur.omega = -b.omega;
          const Edge e = make_edge(pv.surface, cur.surface, cur.omega);
          cur.throughput = ((pv.throughput * b.q.throughput) * e.bCos) * c.rinv;
          if (l1norm(cur.throughput) < MI_FLT_EPSILON) { pc = PC_LIGHT_END; break; }
          cur.throughput = cur.throughput / b.q.density;
          pv.finite = pv.finite < b.q.finite ? pv.finite : b.q.finite;
          cur.finite = b.q.finite;
          cur.a = 1.0f / betaf(c, e.fG * b.q.density);
          cur.A = (pv.A * betaf(c, b.q.densityRev) + pv.a * float(pv.finite)) * betaf(c, e.bG) * cur.a;
          if (b.q.finite == 0) { pv = cur; }
          else { rec_store_l(lrec + size_t(prv) * 7u, pv); pv = cur; prv = size; ++size; }
          pc = PC_LIGHT_HEAD;
        } break;
        case PC_LIGHT_END: {
          const BSample last = bpt_bsdf_sample(c, g, pv.surface, pv.omega);
          if (last.q.finite == 0) --size; else rec_store_l(lrec + size_t(prv) * 7u, pv);
          L = size;
          pc = PC_EYE_START;
        } break;
        case PC_EYE_START: {
          const m33 v2w = {F3(p.v2w[0], p.v2w[1], p.v2w[2]), F3(p.v2w[3], p.v2w[4], p.v2w[5]), F3(p.v2w[6], p.v2w[7], p.v2w[8])};
          Surf cs;  // Technique::_camera_surface (Technique.cpp:107-116)
          cs.position = F3(p.cam_pos[0], p.cam_pos[1], p.cam_pos[2]);
          cs.tangent.c0 = v2w.c1; cs.tangent.c1 = -v2w.c2; cs.tangent.c2 = v2w.c0;
          cs.material_id = (0u << 2) | MI_ENTITY_CAMERA;
          cs.gnormal = -v2w.c2;
          Rng g0 = rng_seed(p.seed, ln.py * p.width + ln.px, ln.sample);  // the camera ray of this sample (bpt_trace forms it before the light sub-path)
          const float u0 = rng_f(g0), u1 = rng_f(g0);
          const float fx = float(ln.px) + u0, fy = float(ln.py) + u1;
          const float vx = fx * p.res_y_inv * 2.0f - p.res_x * p.res_y_inv;
          const float vy = fy * p.res_y_inv * 2.0f - 1.0f;
          const f3 dir = mulmv(v2w, normalize(F3(vx, vy, -p.focal_length_y)));
          surface = cs;
          pv.surface = surface; pv.omega = -dir; pv.throughput = F3(1, 1, 1) * c.rinv;
          pv.finite = 1; pv.a = 0.0f; pv.A = 0.0f;
          pc = PC_EYE_HEAD;
        } break;
        case PC_EYE_HEAD: {
          at_camera = (pv.surface.material_id & 3u) == MI_ENTITY_CAMERA;
          if (E >= w.max_vertices) { overflow = true; pc = PC_END; break; }
          if (at_camera) {
            rec_store_e(erec + size_t(E) * 7u, lvert_as_evert(pv), 0u, 0u);  // items [0, L): the splats of _connect_eye
            evi[E] = make_uint2(0u, 0u);
            n_items = L;
          } else {
            uint32_t kind = 0;
            if (!bpt_roulette(c, g)) {  // BPT.cpp:278-288
              const LSample nb = light_sample(c, g);
              if (!nb.directional) { kind = 1; rec_store_l(nrec + size_t(E) * 7u, sample_to_vertex(c, nb)); }
              else { kind = 2; rec_store_dir(nrec + size_t(E) * 7u, nb); ++n_dir; }
            }
            rec_store_e(erec + size_t(E) * 7u, lvert_as_evert(pv), kind, n_items);
            evi[E] = make_uint2(kind, n_items);
            n_items += (kind ? 1u : 0u) + (L > 1u ? L - 1u : 0u);
          }
          k = E;
          ++E;
          b = bpt_bsdf_sample(c, g, pv.surface, pv.omega);
          ray_pos = surface.position; ray_gn = surface.gnormal; ray_dir = b.omega; ray_mask = 0xFFFFFFFFu;
          phase = kStepEye; pc = PC_SUSPEND;
        } break;
        case PC_EYE_RESUME: {
          if (surface.material_id == 0xFFFFFFFFu) {
            if (at_camera) {  // BPT.cpp:49-51: a miss from the camera vertex returns the sky gradient instead of what was gathered through emitters
              const f3 sky = (c.sky_horizon * (1 - b.omega.z) + c.sky_zenith * b.omega.z) * c.rinv;
              em[0] = make_float4(sky.x, sky.y, sky.z, __uint_as_float(0u)); n_em = 1;
            }
            pc = PC_END; break;
          }
          EVert cur;
          cur.surface = surface; cur.omega = -b.omega;
          const Edge e = make_edge(pv.surface, cur.surface, cur.omega);
          cur.throughput = (pv.throughput * b.q.throughput) * e.bCos;
          if (l1norm(cur.throughput) < MI_FLT_EPSILON) { pc = PC_END; break; }
          cur.throughput = cur.throughput / b.q.density;
          pv.finite = pv.finite < b.q.finite ? pv.finite : b.q.finite;
          cur.finite = b.q.finite;
          cur.c = 1.0f / betaf(c, e.fG * b.q.density);
          cur.C = (pv.A * betaf(c, b.q.densityRev) + pv.a * float(pv.finite)) * betaf(c, e.bG) * cur.c;
          if (surf_is_light(surface)) {
            const f3 t = bpt_connect_light(c, cur);
            if (n_em >= w.max_vertices) { overflow = true; pc = PC_END; break; }
            em[n_em++] = make_float4(t.x, t.y, t.z, __uint_as_float(k));
            ray_pos = surface.position; ray_gn = surface.gnormal; ray_dir = b.omega; ray_mask = 0xFFFFFFFFu;  // through the emitter, same direction
            phase = kStepEye; pc = PC_SUSPEND;
          } else {
            pv = evert_as_lvert(cur);
            if (bpt_roulette(c, g)) { pc = PC_END; break; }
            pv.throughput = pv.throughput * c.rinv;
            pc = PC_EYE_HEAD;
          }
        } break;
        case PC_END: {
          w.info[2 * size_t(i)] = make_uint4(L, E, n_items, n_em);
          w.info[2 * size_t(i) + 1] = make_uint4(basic, n_dir, (ln.py << 16) | ln.px, (ln.fl << 1) | (ln.ok ? 1u : 0u));
          w.item_offset[i] = n_items;
          done = true;
        } break;
        default: {  // PC_SUSPEND: Scene::intersect's first half (Scene.cpp:185-197) — the nudged origin; the walk and the surface query follow in the next round
          bool walk_here = INLINE;
          if constexpr (INLINE) { if (budget != 0u && rays_here >= budget) walk_here = false; }
          if (walk_here) {  // the ray is walked here (scene_intersect as in bpt_trace) and the path goes on
            ++rays_here;
            Surf from; from.position = ray_pos; from.gnormal = ray_gn; from.tangent = surface.tangent; from.material_id = 0u;  // scene_intersect reads position and gnormal
            if constexpr (INLINE) surface = scene_intersect<QN>(c, from, ray_dir, ray_mask);
            ++basic;
            pc = phase == kStepLight ? PC_LIGHT_RESUME : PC_EYE_RESUME;
            break;
          }
          const f3 org = nudge(ray_pos, ray_gn, ray_dir);
          w.step_rays[2 * size_t(i)] = make_float4(org.x, org.y, org.z, __uint_as_float(ray_mask));
          w.step_rays[2 * size_t(i) + 1] = make_float4(ray_dir.x, ray_dir.y, ray_dir.z, 0.0f);
          ++basic;
          reinterpret_cast<uint4*>(st)[0] = make_uint4(phase, g.state, L, E);
          reinterpret_cast<uint4*>(st)[1] = make_uint4(n_items, n_em, n_dir, basic);
          reinterpret_cast<uint4*>(st)[2] = make_uint4(size, prv, k, (at_camera ? 1u : 0u) | (b.q.finite ? 2u : 0u));
          rec_store_l(st + 3, pv);
          st[14] = make_float4(b.omega.x, b.omega.y, b.omega.z, b.q.density);
          st[15] = make_float4(b.q.throughput.x, b.q.throughput.y, b.q.throughput.z, b.q.densityRev);
          emit = true; done = true;
        } break;
      }
    }
  }
  // the paths with a ray in flight, appended wave by wave
  const uint64_t m = __ballot(emit);
  if (m != 0ull) {
    const uint32_t lane = threadIdx.x & 63u, leader = uint32_t(__ffsll((unsigned long long)m)) - 1u;
    uint32_t at = 0;
    if (lane == leader) at = atomicAdd(&w.step_count[out_slot], uint32_t(__popcll(m)));
    at = __shfl(at, int(leader), 64);
    if (emit) w.step_active[out_slot][at + __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u))] = i;
  }
  n_overflow += uint32_t(__popcll(__ballot(overflow)));
  }
  if (n_overflow != 0u && (threadIdx.x & 63u) == 0u && p.counters) atomicAdd(&p.counters[15], (unsigned long long)n_overflow);
}

// The closest-hit walk of the rays in flight — Scene::intersect's rtcIntersect (Scene.cpp:198) with the ray's geometry mask — by PERSISTENT waves: a lane that
// has finished its ray takes the next unstarted one of the list (chunks of kStepChunk rays from interleaved cursors, refill whenever `th` lanes idle), like the
// any-hit walk of bpt_visibility.  Of a ray only (t, position) of the best hit live in the loop; U, V, |den| are formed again for the winner (the leaf's own
// expressions on the same operands: the same bits as tri_test + finish_hit, as in traverse_dyn).
constexpr uint32_t kStepChunk = 128u, kStepCursors = 64u;
template <int QN>
__global__ __launch_bounds__(kBlock, 6) void bpt_closest(const RenderParams p, const BptState w, uint32_t slot, uint32_t n_cursors, uint32_t th) {
  extern __shared__ float4 smem[];
  const SceneView& sv = p.sv;
  TravStackT<true> stack;
  stack.lds = (lds_u32*)(reinterpret_cast<uint32_t*>(smem) + threadIdx.x);
  stack.cap = p.stack_entries;
  const uint32_t* __restrict__ active = w.step_active[slot];
  const uint32_t n_rays = w.step_count[slot];
  const float4* tris = sv.blob + sv.off_tris;
  const uint4* __restrict__ qn = sv.qnodes;
  const uint4* __restrict__ q4 = sv.qnodes4;
  const f3 glo = F3(sv.grid_lo[0], sv.grid_lo[1], sv.grid_lo[2]), gis = F3(sv.grid_inv_step[0], sv.grid_inv_step[1], sv.grid_inv_step[2]);
  const uint32_t lane = threadIdx.x & 63u;
  const int root = sv.n_nodes == 0 ? ~0 : 0;
  uint32_t* cursors = w.step_count + 8u;  // [kStepCursors], zeroed by the host before the launch
  const uint32_t group = blockIdx.x % n_cursors;
  uint32_t next = 0, end = 0;
  bool exhausted = false, walking = false;
  uint32_t item = 0, ray_mask = 0, best_pos = 0xFFFFFFFFu;
  float best_t = __builtin_inff();
  f3 co = F3(0, 0, 0), cd = F3(0, 0, 1);
  RayBox rb = make_raybox(co, F3(1, 1, 1));
  int sp = 0, node = 0;
  for (;;) {
    const uint64_t busy = __ballot(walking);
    const uint32_t n_idle = 64u - uint32_t(__popcll(busy));
    if (!exhausted && (busy == 0ull || n_idle >= th)) {
      if (next == end) {
        uint32_t k = 0;
        if (lane == 0u) k = atomicAdd(&cursors[group], 1u);
        k = __shfl(k, 0, 64);
        const uint64_t v = (uint64_t(k) * n_cursors + group) * kStepChunk;
        if (v >= uint64_t(n_rays)) { exhausted = true; continue; }
        next = uint32_t(v); end = n_rays - next < kStepChunk ? n_rays : next + kStepChunk;
      }
      const uint64_t idle = ~busy;
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(idle >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(idle), 0u));
      const uint32_t avail = end - next, take = n_idle < avail ? n_idle : avail;
      if (!walking && rank < take) {
        item = active[next + rank];
        const float4 a = w.step_rays[2 * size_t(item)], b = w.step_rays[2 * size_t(item) + 1];
        co = xyz(a); cd = xyz(b); ray_mask = __float_as_uint(a.w);
        rb = QN ? make_raybox((co - glo) * gis, cd * gis) : make_raybox(co, cd);
        best_t = __builtin_inff(); best_pos = 0xFFFFFFFFu;
        sp = 0; node = root; walking = true;
      }
      next += take;
      continue;
    }
    if (busy == 0ull) break;
    if (walking) {
      bool pop = false;
      if (node >= 0) {
        if (QN == 2) {
          float t[4]; int l[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint4 a = q4[4 * node + k];
            float tn;
            const bool hk = wide_child_test(a, rb, best_t, tn);
            t[k] = hk ? tn : __builtin_inff();
            l[k] = int(a.w);
          }
#define MI_CSWAP(a, b) do { const bool s_ = t[b] < t[a]; const float ta_ = s_ ? t[b] : t[a], tb_ = s_ ? t[a] : t[b]; \
                            const int la_ = s_ ? l[b] : l[a], lb_ = s_ ? l[a] : l[b]; t[a] = ta_; t[b] = tb_; l[a] = la_; l[b] = lb_; } while (0)
          MI_CSWAP(0, 1); MI_CSWAP(2, 3); MI_CSWAP(0, 2); if (MI_WIDE_SORT_FULL) { MI_CSWAP(1, 3); MI_CSWAP(1, 2); }
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
          const uint4 a = qn[2 * node], b = qn[2 * node + 1];
          const int l0 = int(a.w), l1 = int(b.w);
          float tn0, tn1;
          const bool h0 = wide_child_test(a, rb, best_t, tn0), h1 = wide_child_test(b, rb, best_t, tn1);
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
        // Embree single-ray Moeller-Trumbore (tri_test in pt_device.h), closest hit by (t, id)
        const uint32_t pos = uint32_t(~node) & kLeafPosMask;  // bit 30 of ~node: pair leaf (layout.h), the triangle at pos + 1 is this lane's next iteration
        const float4 a = tris[3 * pos], b = tris[3 * pos + 1], c = tris[3 * pos + 2];
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
        pop = true;
        if ((gmask & ray_mask) != 0u && den != 0.0f && U >= 0.0f && V >= 0.0f && U + V <= absden && absden * 0.0f < T) {
          const float t = T / absden;
          if (t < best_t || (t == best_t && (best_pos == 0xFFFFFFFFu || id < __float_as_uint(tris[3 * best_pos + 2].y)))) { best_t = t; best_pos = pos; }
        }
        if ((uint32_t(node) & kLeafPairBit) == 0u) { node = int((uint32_t(node) | kLeafPairBit) - 1u); pop = false; }  // ~(pos + 1), a single leaf
      }
      if (pop) {
        if (sp == 0) {
          // the ray is done: U, V, |den| of the winner (same expressions, same operands: same bits), then finish_hit's two divisions
          float hu = 0.0f, hv = 0.0f;
          if (best_pos != 0xFFFFFFFFu) {
            const float4 a = tris[3 * best_pos], b = tris[3 * best_pos + 1], c = tris[3 * best_pos + 2];
            const f3 v0 = F3(a.x, a.y, a.z), e1 = F3(a.w, b.x, b.y), e2 = F3(b.z, b.w, c.x);
            const f3 ng = cross(e2, e1);
            const f3 R = cross(v0 - co, cd);
            const float den = dot(ng, cd);
            const float sgn = den < 0.0f ? -1.0f : 1.0f;
            const float ad = fabsf(den);
            hu = (dot(R, e2) * sgn) / ad; hv = (dot(R, e1) * sgn) / ad;
          }
          w.step_hits[item] = make_float4(best_t, hu, hv, __uint_as_float(best_pos));
          walking = false;
        } else { --sp; node = int(stack.pop(sp)); }
      }
    }
  }
}

// exclusive scan of the per-path item counts (w.lanes + 1 entries, the last receives the total): tiles of 2048 entries are
// scanned by one workgroup each, their totals by a single workgroup, then added back
constexpr uint32_t kScanTile = 2048;
__global__ __launch_bounds__(256) void bpt_scan_tiles(uint32_t* __restrict__ data, uint32_t total, uint32_t* __restrict__ tile_sums) {
  __shared__ uint32_t wsum[4];
  const uint32_t tid = threadIdx.x, base = blockIdx.x * kScanTile + tid * 8u;
  uint32_t v[8], s = 0;
  for (int k = 0; k < 8; ++k) { v[k] = base + k < total ? data[base + k] : 0u; s += v[k]; }
  uint32_t incl = s;  // inclusive scan of the per-thread sums over the wave, then over the four waves
  for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o, 64); if ((tid & 63u) >= uint32_t(o)) incl += t; }
  if ((tid & 63u) == 63u) wsum[tid >> 6] = incl;
  __syncthreads();
  uint32_t run = incl - s;
  for (uint32_t w2 = 0; w2 < (tid >> 6); ++w2) run += wsum[w2];
  for (int k = 0; k < 8; ++k) { if (base + k < total) data[base + k] = run; run += v[k]; }
  if (tid == 255u) tile_sums[blockIdx.x] = run;
}
__global__ __launch_bounds__(1024) void bpt_scan_sums(uint32_t* __restrict__ sums, uint32_t n) {  // n <= 1024 * chunk
  __shared__ uint32_t part[1024];
  const uint32_t tid = threadIdx.x, chunk = (n + 1023u) / 1024u, b = tid * chunk, e = b + chunk < n ? b + chunk : n;
  uint32_t s = 0;
  for (uint32_t i = b; i < e; ++i) s += sums[i];
  part[tid] = s;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) { const uint32_t v = tid >= off ? part[tid - off] : 0u; __syncthreads(); part[tid] += v; __syncthreads(); }
  uint32_t run = tid ? part[tid - 1] : 0u;
  for (uint32_t i = b; i < e; ++i) { const uint32_t v = sums[i]; sums[i] = run; run += v; }
}
__global__ __launch_bounds__(256) void bpt_scan_add(uint32_t* __restrict__ data, uint32_t total, const uint32_t* __restrict__ tile_sums) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < total) data[i] += tile_sums[i / kScanTile];
}

// ---- stage B: one lane per connection item ----
// Which connection is item `item` of the launch?  Shared by bpt_rays and bpt_items.
struct ItemRef {
  uint32_t path, eye_k, lv_i;  // eye vertex (erec index), light-side record index
  uint32_t type;               // 0 light vertex lrec[lv_i] -> camera erec[0] (a splat); 1 area NEE sample nrec[eye_k]; 2 directional NEE nrec[eye_k]; 3 light vertex lrec[lv_i]
};
MI_DEV ItemRef item_decode(const BptState& w, uint32_t item) {
  ItemRef r;
  // which path?  last path whose first item is <= item (paths without items share an offset with their successor)
  uint32_t lo = 0, hi = w.lanes;
  while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (w.item_offset[mid] <= item) lo = mid; else hi = mid; }
  r.path = lo;
  const uint32_t local = item - w.item_offset[lo];
  const uint4 inf = w.info[2 * size_t(lo)];
  const uint32_t L = inf.x, E = inf.y;
  if (local < L) { r.type = 0u; r.eye_k = 0u; r.lv_i = local; return r; }
  const uint2* evi = w.evinfo + size_t(lo) * w.max_vertices;
  // which eye vertex?  last one (k >= 1) whose first item is <= local
  uint32_t klo = 1, khi = E;
  while (khi - klo > 1u) { const uint32_t mid = (klo + khi) >> 1; if (evi[mid].y <= local) klo = mid; else khi = mid; }
  const uint32_t kind = evi[klo].x, item0 = evi[klo].y;
  const uint32_t idx = local - item0;
  r.eye_k = klo;
  if (kind != 0u && idx == 0u) { r.type = kind == 1u ? 1u : 2u; r.lv_i = klo; }
  else { r.type = 3u; r.lv_i = idx - (kind ? 1u : 0u) + 1u; }
  return r;
}
// pixel_position (Cameras.cpp:134-144) of a light vertex seen from the camera: false = outside the image (no splat, no shadow ray)
MI_DEV bool splat_pixel(const RenderParams& p, const BptState& w, f3 eye_pos, f3 lv_pos, f3& omega, int& ix, int& iy) {
  const m33 w2v = {F3(w.w2v[0], w.w2v[1], w.w2v[2]), F3(w.w2v[3], w.w2v[4], w.w2v[5]), F3(w.w2v[6], w.w2v[7], w.w2v[8])};
  omega = normalize(lv_pos - eye_pos);
  const f3 vd = mulmv(w2v, omega);
  const float factor = p.focal_length_y / -vd.z;
  const float x = vd.x * factor, y = vd.y * factor;
  const float py = (y + 1.0f) * p.res_y * 0.5f;
  const float px = (x + p.res_x * p.res_y_inv) * p.res_y * 0.5f;
  if (!(0 <= px && px < p.res_x && 0 <= py && py < p.res_y)) return false;
  ix = int(px); iy = int(py);
  return true;
}

// ---- stage B1 (visibility, r02): the items' shadow rays as a list ----
// Scene::occluded (Scene.cpp:151-180) of a connection needs only the two positions and geometric normals; everything else of _connect
// waits for stage B3.  One lane per item, uniform work, no atomics (a compacted list needs one same-address atomic per wave: 55 000 per
// launch serialise to 0.7 ms, as long as the tracing stage).
__global__ __launch_bounds__(256) void bpt_rays(const RenderParams p, const BptState w, uint32_t item_count) {
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  bool has = false;
  f3 ao = F3(0, 0, 0), ad = F3(0, 0, 0);
  if (j < item_count) {
    const ItemRef r = item_decode(w, j);
    if (r.type != 2u) {
      const float4* e = w.eslab + (size_t(r.path) * w.max_vertices + r.eye_k) * 7u;
      const float4* l = (r.type == 1u ? w.nslab : w.lslab) + (size_t(r.path) * w.max_vertices + r.lv_i) * 7u;
      const f3 opos = xyz(e[0]), ogn = xyz(e[1]), tpos = xyz(l[0]), tgn = xyz(l[1]);
      has = true;
      if (r.type == 0u) { f3 omega; int ix, iy; has = splat_pixel(p, w, opos, tpos, omega, ix, iy); }
      // the ray of occluded() (pt_device.h): both ends nudged off their surfaces, t in (0, 1]
      const f3 direction = tpos - opos;
      ao = opos + (ogn * (dot(ogn, direction) > 0.0f ? 1.0f : -1.0f)) * 0.0001f;
      const f3 at = tpos + (tgn * (dot(tgn, direction) < 0.0f ? 1.0f : -1.0f)) * 0.0001f;
      ad = at - ao;
    }
  }
  if (j < item_count) {  // ray j belongs to item j; w = 1 marks an item that casts a shadow ray (directional NEE and splats outside the image do not)
    float4* o = w.rays + 2 * size_t(j);
    o[0] = make_float4(ao.x, ao.y, ao.z, has ? 1.0f : 0.0f);
    o[1] = make_float4(ad.x, ad.y, ad.z, 0.0f);
  }
}

// ---- stage B2 (visibility, r02): persistent waves walk the ray list, idle lanes refill from it ----
// One lane per item (round 1) pays the longest shadow ray of every wave: any-hit rays end after anything between one and a hundred
// node visits, lane efficiency of the walk was 0.2-0.3.  Here a wave keeps walking: whenever BptState::vis_th lanes are idle they take
// the next rays of the wave's chunk (chunks of kVisChunk rays come from one global cursor), so the wave is full until the list
// is empty.  A ray's result goes to occl[item]; nothing else of the connection lives in this kernel (32 VGPRs of ray state).
// Exit: the cursor passes the ray count -> `exhausted` is wave-uniform -> the wave leaves when its last ray ends.
#ifndef MI_BPT_VIS_WAVES
#define MI_BPT_VIS_WAVES 8
#endif
constexpr uint32_t kVisChunk = 128u, kVisCursors = 64u;
template <int QN>
__global__ __launch_bounds__(kBlock, MI_BPT_VIS_WAVES) void bpt_visibility(const RenderParams p, const BptState w, uint32_t n_rays, uint32_t n_cursors) {
  extern __shared__ float4 smem[];
  SceneView sv = p.sv;
  const float4* sb = sv.blob;
  uint32_t scene_f4 = 0;
  if (QN == 0) { scene_f4 = lds_scene_f4(sv); stage_scene_to_lds(smem, sv, threadIdx.x); sb = smem; __syncthreads(); }  // this stage walks the tree itself: always the tree copy
  TravStackT<(QN != 0)> stack;
  stack.lds = (lds_u32*)(reinterpret_cast<uint32_t*>(smem + scene_f4) + threadIdx.x);
  stack.cap = p.stack_entries;
  constexpr int NS = QN == 0 ? 5 : 4;
  const float4* nodes = sb + sv.off_nodes;
  const float4* tris = sb + sv.off_tris;
  const uint4* __restrict__ qn = sv.qnodes;
  const uint4* __restrict__ q4 = sv.qnodes4;
  const f3 glo = F3(sv.grid_lo[0], sv.grid_lo[1], sv.grid_lo[2]), gis = F3(sv.grid_inv_step[0], sv.grid_inv_step[1], sv.grid_inv_step[2]);
  const uint32_t lane = threadIdx.x & 63u;
  const int root = sv.n_nodes == 0 ? ~0 : 0;  // a one-triangle scene has no nodes: its "tree" is the leaf of triangle 0
  // kVisCursors cursors instead of one: cursor r hands out the chunks r, r + K, r + 2K, ... (interleaved, so every cursor sees the whole
  // list) to the workgroups with blockIdx % K == r; same-address atomics serialise at ~13 ns each
  const uint32_t group = blockIdx.x % n_cursors;  // n_cursors = min(kVisCursors, workgroups): every cursor has a workgroup
  uint32_t next = 0, end = 0;
  bool exhausted = false, walking = false;
  uint32_t item = 0;
  f3 co = F3(0, 0, 0), cd = F3(0, 0, 0);
  RayBox rb = make_raybox(co, F3(1, 1, 1));
  int sp = 0, node = 0;
  for (;;) {
    const uint64_t busy = __ballot(walking);
    const uint32_t n_idle = 64u - uint32_t(__popcll(busy));
    if (!exhausted && (busy == 0ull || n_idle >= w.vis_th)) {
      if (next == end) {
        uint32_t k = 0;
        if (lane == 0u) k = atomicAdd(&w.pool[group], 1u);
        k = __shfl(k, 0, 64);
        const uint64_t v = (uint64_t(k) * n_cursors + group) * kVisChunk;
        if (v >= uint64_t(n_rays)) { exhausted = true; continue; }
        next = uint32_t(v); end = n_rays - next < kVisChunk ? n_rays : next + kVisChunk;
      }
      const uint64_t idle = ~busy;
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(idle >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(idle), 0u));
      const uint32_t avail = end - next, take = n_idle < avail ? n_idle : avail;
      if (!walking && rank < take) {
        const float4 a = w.rays[2 * size_t(next + rank)], b = w.rays[2 * size_t(next + rank) + 1];
        if (a.w != 0.0f) {
          co = xyz(a); cd = xyz(b); item = next + rank;
          rb = QN ? make_raybox((co - glo) * gis, cd * gis) : make_raybox(co, cd);
          sp = 0; node = root; walking = true;
        }
      }
      next += take;
      continue;
    }
    if (busy == 0ull) break;
    if (walking) {
      bool pop = false;
      if (node >= 0) {
        if (QN == 2) {
          float t[4]; int l[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint4 a = q4[4 * node + k];
            float tn;
            const bool hk = wide_child_test(a, rb, 1.0f, tn);
            t[k] = hk ? tn : __builtin_inff();
            l[k] = int(a.w);
          }
#define MI_CSWAP(a, b) do { const bool s_ = t[b] < t[a]; const float ta_ = s_ ? t[b] : t[a], tb_ = s_ ? t[a] : t[b]; \
                            const int la_ = s_ ? l[b] : l[a], lb_ = s_ ? l[a] : l[b]; t[a] = ta_; t[b] = tb_; l[a] = la_; l[b] = lb_; } while (0)
          MI_CSWAP(0, 1); MI_CSWAP(2, 3); MI_CSWAP(0, 2); if (MI_WIDE_SORT_FULL) { MI_CSWAP(1, 3); MI_CSWAP(1, 2); }
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
          if (QN) {
            const uint4 a = qn[2 * node], b = qn[2 * node + 1];
            l0 = int(a.w); l1 = int(b.w);
            h0 = wide_child_test(a, rb, 1.0f, tn0);
            h1 = wide_child_test(b, rb, 1.0f, tn1);
          } else {
            const float4 n0 = nodes[NS * node], n1 = nodes[NS * node + 1], n2 = nodes[NS * node + 2], n3 = nodes[NS * node + 3];
            l0 = __float_as_int(n0.w); l1 = __float_as_int(n1.w);
            h0 = ce_box_test(xyz(n0), xyz(n1), rb, 1.0f, tn0); h1 = ce_box_test(xyz(n2), xyz(n3), rb, 1.0f, tn1);  // QN == 0: the LDS copy (centre + half extent)
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
        // rtcOccluded's single-ray Moeller-Trumbore (tri_test<true> in pt_device.h): mesh geometry only, first hit with t <= 1 ends the ray
        const uint32_t pos = uint32_t(~node) & kLeafPosMask;  // bit 30 of ~node: pair leaf (layout.h), the triangle at pos + 1 is this lane's next iteration
        const float4 a = tris[3 * pos], b = tris[3 * pos + 1], c = tris[3 * pos + 2];
        const f3 v0 = F3(a.x, a.y, a.z), e1 = F3(a.w, b.x, b.y), e2 = F3(b.z, b.w, c.x);
        const uint32_t gmask = __float_as_uint(c.z);
        const f3 ng = cross(e2, e1);
        const f3 C = v0 - co;
        const f3 R = cross(C, cd);
        const float den = dot(ng, cd);
        const float absden = fabsf(den);
        const float sgn = den < 0.0f ? -1.0f : 1.0f;
        const float U = dot(R, e2) * sgn;
        const float V = dot(R, e1) * sgn;
        const float T = dot(ng, C) * sgn;
        pop = true;
        if ((gmask & (1u << MI_ENTITY_MESH)) != 0u && den != 0.0f && U >= 0.0f && V >= 0.0f && U + V <= absden && absden * 0.0f < T) {
          if (T <= absden) { w.occl[item] = 1u; walking = false; pop = false; }  // t = T / |den| <= 1 <=> T <= |den| (pt_device.h tri_test): no division
        }
        if ((uint32_t(node) & kLeafPairBit) == 0u && walking) { node = int((uint32_t(node) | kLeafPairBit) - 1u); pop = false; }
      }
      if (pop) {
        if (sp == 0) { w.occl[item] = 0u; walking = false; }
        else { --sp; node = int(stack.pop(sp)); }
      }
    }
  }
}

// ---- stage B3: one lane per connection item: both BSDF queries, MIS weight, value; PRE = visibility comes from stage B2 ----
template <bool LIST, int QN, bool PRE>
#ifndef MI_BPT_ITEMS_PRE_WAVES
#define MI_BPT_ITEMS_PRE_WAVES 4  // with the traversal gone (PRE) the kernel is uniform BSDF arithmetic over 224 B of vertices: registers before occupancy
#endif
__global__ __launch_bounds__(kBlock, PRE ? MI_BPT_ITEMS_PRE_WAVES : MI_BPT_ITEMS_WAVES) void bpt_items(const RenderParams p, const BptState w, uint32_t item_first, uint32_t item_count) {
  extern __shared__ float4 smem[];
  SceneView sv = p.sv;
  const float4* sb = sv.blob;
  uint32_t scene_f4 = 0;
  if (QN == 0) {
    if (p.flat_k) { scene_f4 = flat_scene_f4(sv, p.flat_k); stage_scene_flat(smem, sv, p.flat_table, p.flat_k, threadIdx.x); }
    else { scene_f4 = lds_scene_f4(sv); stage_scene_to_lds(smem, sv, threadIdx.x); }
    sb = smem; __syncthreads();
  }
  TravStackT<(QN != 0)> stack;
  stack.lds = (lds_u32*)(reinterpret_cast<uint32_t*>(smem + scene_f4) + threadIdx.x);
  stack.cap = p.stack_entries;
  const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= item_count) return;
  const uint32_t item = item_first + j;
  const ItemRef ref = item_decode(w, item);
  const uint32_t path = ref.path;
  const float4* lrec = w.lslab + size_t(path) * w.max_vertices * 7u;
  const float4* erec = w.eslab + size_t(path) * w.max_vertices * 7u;
  const float4* nrec = w.nslab + size_t(path) * w.max_vertices * 7u;
  Ctx c; ctx_init(c, p, w, &stack, sb, &sv);
  if (PRE) c.pre_occ = w.occl[j] ? 0.0f : 1.0f;
  f3 value = F3(0, 0, 0); uint32_t flags = 0;  // bit 0: a shadow ray was cast, bit 1: a closest-hit ray was cast, bit 2: splat inside the image
  uint32_t kind, item0;
  const EVert eye = rec_load_e(erec + size_t(ref.eye_k) * 7u, kind, item0);
  if (ref.type == 0u) {
    // ---- _connect_eye (BPT.cpp:295-321): light vertex `lv_i` to the camera ----
    const LVert lv = rec_load_l(lrec + size_t(ref.lv_i) * 7u);
    const float focal_factor_y = p.focal_length_y * p.focal_length_y * 0.25f;
    f3 omega; int ix = 0, iy = 0;
    if (splat_pixel(p, w, eye.surface.position, lv.surface.position, omega, ix, iy)) {
      const f3 ln = lv.surface.tangent.c1, en = eye.surface.tangent.c1;
      const float normal_coefficient = fabsf(dot(omega, lv.surface.gnormal) * dot(lv.omega, ln) / (dot(omega, ln) * dot(lv.omega, lv.surface.gnormal)));
      const float ce = fabsf(dot(en, omega));
      const float focal_coefficient = 1.0f / (ce * ce * ce);
      value = (bpt_connect<QN, PRE>(c, lv, eye) * focal_factor_y) * (normal_coefficient * focal_coefficient);
      flags = 1u | 4u;
      if (!LIST) {
        const uint32_t fl = w.info[2 * size_t(path) + 1].w >> 1;
        double* l = w.light + 3 * (size_t(fl) * p.width * p.height + size_t(iy) * size_t(p.res_x) + size_t(ix));
        atomicAdd(&l[0], double(value.x)); atomicAdd(&l[1], double(value.y)); atomicAdd(&l[2], double(value.z));
      }
    }
  } else if (ref.type == 1u) {
    const LVert lv = rec_load_l(nrec + size_t(ref.lv_i) * 7u); value = bpt_connect<QN, PRE>(c, lv, eye); flags = 1u;
  } else if (ref.type == 2u) {
    const LSample b = rec_load_dir(nrec + size_t(ref.lv_i) * 7u); value = bpt_connect_directional<QN>(c, eye, b); flags = 2u;
  } else {
    const LVert lv = rec_load_l(lrec + size_t(ref.lv_i) * 7u);
    value = bpt_connect<QN, PRE>(c, lv, eye); flags = 1u;
  }
  w.values[j] = make_float4(value.x, value.y, value.z, __uint_as_float(flags));
}

// ---- stage C: per path, the sums in the reference's order ----
template <bool LIST>
__global__ __launch_bounds__(256) void bpt_gather(const RenderParams p, const BptState w, uint32_t item_first) {
  // grid-stride over the paths: the three ray / path counters are same-line atomics, which serialise at ~13 ns each — one set per WAVE of a
  // one-path-per-lane grid (49 000 per launch) cost more than the sums themselves; one set per workgroup of a capped grid is 3 x <= 1024
  __shared__ uint32_t red[3][4];
  uint32_t nb = 0, ns = 0, np = 0;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < w.lanes; i += gridDim.x * 256u) {
    const uint4 inf = w.info[2 * size_t(i)], inf2 = w.info[2 * size_t(i) + 1];
    if (inf2.w & 1u) {
      const uint32_t L = inf.x, E = inf.y, n_em = inf.w;
      const float4* vals = w.values + (size_t(w.item_offset[i]) - item_first);
      const uint2* evi = w.evinfo + size_t(i) * w.max_vertices;
      const float4* em = w.emission + size_t(i) * w.max_vertices;  // emission terms (eye sub-path hits on emitters): as many as vertices
      f3 radiance = F3(0, 0, 0), splat_sum = F3(0, 0, 0);
      uint32_t n_splat = 0, shadow = 0, basic = inf2.x, m = 0;
      if (E > 0u) for (uint32_t s = 0; s < L; ++s) {
        const float4 v = vals[s]; const uint32_t f = __float_as_uint(v.w);
        if (f & 4u) { splat_sum = splat_sum + F3(v.x, v.y, v.z); ++n_splat; ++shadow; }
      }
      for (uint32_t k = 0; k < E; ++k) {
        if (k >= 1u) {
          const uint2 ev = evi[k];
          const uint32_t kind = ev.x, item0 = ev.y;
          const uint32_t n_k = (kind ? 1u : 0u) + (L > 1u ? L - 1u : 0u);
          f3 local = F3(0, 0, 0);  // BPT.cpp:276: a vertex's connections are summed on their own, then added to the path's radiance
          for (uint32_t t = 0; t < n_k; ++t) {
            const float4 v = vals[item0 + t]; const uint32_t f = __float_as_uint(v.w);
            local = local + F3(v.x, v.y, v.z);
            shadow += f & 1u; basic += (f >> 1) & 1u;
          }
          radiance = radiance + local;
        }
        while (m < n_em && __float_as_uint(em[m].w) == k) { radiance = radiance + F3(em[m].x, em[m].y, em[m].z); ++m; }
      }
      nb += basic; ns += shadow; np += 1;
      if (LIST) {
        const size_t item = size_t(w.first) + i;
        p.list_radiance[3 * item] = radiance.x; p.list_radiance[3 * item + 1] = radiance.y; p.list_radiance[3 * item + 2] = radiance.z;
        w.list_splat_sum[3 * item] = splat_sum.x; w.list_splat_sum[3 * item + 1] = splat_sum.y; w.list_splat_sum[3 * item + 2] = splat_sum.z;
        w.list_counts3[3 * item] = basic; w.list_counts3[3 * item + 1] = shadow; w.list_counts3[3 * item + 2] = n_splat;
      } else {
        const uint32_t px = inf2.z & 0xFFFFu, py = inf2.z >> 16, fl = inf2.w >> 1;
        float* e = w.eye + 3 * (size_t(fl) * p.width * p.height + size_t(py) * p.width + px);
        e[0] = radiance.x; e[1] = radiance.y; e[2] = radiance.z;
      }
    }
  }
  for (int k = 32; k > 0; k >>= 1) { nb += __shfl_xor(nb, k, 64); ns += __shfl_xor(ns, k, 64); np += __shfl_xor(np, k, 64); }
  if ((threadIdx.x & 63u) == 0) { red[0][threadIdx.x >> 6] = nb; red[1][threadIdx.x >> 6] = ns; red[2][threadIdx.x >> 6] = np; }
  __syncthreads();
  if (threadIdx.x == 0 && p.counters) {
    const unsigned long long b = (unsigned long long)red[0][0] + red[0][1] + red[0][2] + red[0][3], sh = (unsigned long long)red[1][0] + red[1][1] + red[1][2] + red[1][3],
                             pa = (unsigned long long)red[2][0] + red[2][1] + red[2][2] + red[2][3];
    if (b) atomicAdd(&p.counters[0], b);
    if (sh) atomicAdd(&p.counters[1], sh);
    if (pa) atomicAdd(&p.counters[3], pa);
  }
}

namespace {
}  // namespace

// one lane = one (pixel, sample): shoot() + _traceEye (Technique.cpp:321-338)
#ifndef MI_BPT_WAVES
#define MI_BPT_WAVES 3  // 201 VGPRs without a bound (2 waves); measured on C2-sized BPT frames: 2 waves 804, 3 waves 894, 4 waves 872, 5 waves 754 Mrays/s
#endif
template <bool LIST, int QN>
__global__ __launch_bounds__(kBlock, MI_BPT_WAVES) void bpt_frame(const RenderParams p, const BptState w) {
  extern __shared__ float4 smem[];
  TravStackT<(QN != 0)> stack;
  stack.lds = (lds_u32*)(reinterpret_cast<uint32_t*>(smem) + threadIdx.x);
  stack.cap = p.stack_entries;
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  uint32_t px = 0, py = 0, fl = 0; uint64_t sample = 0; bool ok = false;
  if (LIST) {
    const uint64_t item = w.first + i;
    ok = i < w.lanes && item < p.list_n;
    if (ok) { px = p.list_xy[2 * item]; py = p.list_xy[2 * item + 1]; sample = p.list_sample[item]; }
  } else {
    // lane -> (frame of the batch, pixel in 8x8 tiles of the window): coherent camera rays in a wave
    const uint32_t per_frame = p.tiles_x * p.tiles_y * 64u;
    const uint32_t gi = w.first + i;
    fl = gi / per_frame;
    const uint32_t rem = gi - fl * per_frame, tile = rem >> 6, pix = rem & 63u;
    const uint32_t ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    px = p.win_x0 + tx * 8u + (pix & 7u); py = p.win_y0 + ty * 8u + (pix >> 3);
    sample = p.sample_offset + w.frame + fl;
    ok = i < w.lanes && fl < w.frames && px < p.win_x0 + p.win_w && py < p.win_y0 + p.win_h;
  }
  bool overflow = false;
  uint32_t nb = 0, ns = 0;
  if (ok) {
    Ctx c;
    c.sb = p.sv.blob; c.sv = &p.sv; c.stack = &stack;
    c.flat_table = nullptr; c.flat_k = 0u; c.flat_k_mesh = 0u;  // the one-kernel form reads the scene from HBM
    c.beta = p.beta; c.roulette = p.roulette; c.rinv = 1.0f / p.roulette;
    c.sphere_c = F3(w.sphere[0], w.sphere[1], w.sphere[2]); c.sphere_r = w.sphere[3];
    c.sky_horizon = F3(w.sky_horizon[0], w.sky_horizon[1], w.sky_horizon[2]); c.sky_zenith = F3(w.sky_zenith[0], w.sky_zenith[1], w.sky_zenith[2]);
    c.n_basic = 0; c.n_shadow = 0;
    const m33 v2w = {F3(p.v2w[0], p.v2w[1], p.v2w[2]), F3(p.v2w[3], p.v2w[4], p.v2w[5]), F3(p.v2w[6], p.v2w[7], p.v2w[8])};
    Cam cam;
    cam.w2v = {F3(w.w2v[0], w.w2v[1], w.w2v[2]), F3(w.w2v[3], w.w2v[4], w.w2v[5]), F3(w.w2v[6], w.w2v[7], w.w2v[8])};
    cam.rx = p.res_x; cam.ry = p.res_y; cam.ry_inv = p.res_y_inv; cam.fl = p.focal_length_y;
    Surf cs;  // Technique::_camera_surface (Technique.cpp:107-116)
    cs.position = F3(p.cam_pos[0], p.cam_pos[1], p.cam_pos[2]);
    cs.tangent.c0 = v2w.c1; cs.tangent.c1 = -v2w.c2; cs.tangent.c2 = v2w.c0;
    cs.material_id = (0u << 2) | MI_ENTITY_CAMERA;
    cs.gnormal = -v2w.c2;
    Rng rng = rng_seed(p.seed, py * p.width + px, sample);
    const float u0 = rng_f(rng), u1 = rng_f(rng);
    const float fx = float(px) + u0, fy = float(py) + u1;
    const float vx = fx * p.res_y_inv * 2.0f - p.res_x * p.res_y_inv;
    const float vy = fy * p.res_y_inv * 2.0f - 1.0f;
    const f3 dir = mulmv(v2w, normalize(F3(vx, vy, -p.focal_length_y)));
    Slab slab; slab.base = w.slab; slab.lanes = w.lanes; slab.lane = i; slab.cap = w.max_vertices;
    SplatOut sp; sp.light = LIST ? nullptr : w.light + size_t(fl) * 3 * p.width * p.height; sp.n = 0; sp.sum = F3(0, 0, 0);
    const f3 r = bpt_trace_eye<QN>(c, rng, cam, cs, dir, slab, sp, overflow);
    nb = c.n_basic; ns = c.n_shadow;
    if (LIST) {
      const size_t item = size_t(w.first) + i;
      p.list_radiance[3 * item] = r.x; p.list_radiance[3 * item + 1] = r.y; p.list_radiance[3 * item + 2] = r.z;
      w.list_splat_sum[3 * item] = sp.sum.x; w.list_splat_sum[3 * item + 1] = sp.sum.y; w.list_splat_sum[3 * item + 2] = sp.sum.z;
      w.list_counts3[3 * item] = nb; w.list_counts3[3 * item + 1] = ns; w.list_counts3[3 * item + 2] = sp.n;
    } else {
      float* e = w.eye + 3 * (size_t(fl) * p.width * p.height + size_t(py) * p.width + px);
      e[0] = r.x; e[1] = r.y; e[2] = r.z;
    }
  }
  // counters: rays, paths, light sub-paths cut by the slab capacity
  uint32_t a = nb, b = ns, o = overflow ? 1u : 0u, n = ok ? 1u : 0u;
  for (int k = 32; k > 0; k >>= 1) { a += __shfl_xor(a, k, 64); b += __shfl_xor(b, k, 64); o += __shfl_xor(o, k, 64); n += __shfl_xor(n, k, 64); }
  if ((threadIdx.x & 63u) == 0 && p.counters) {
    if (a) atomicAdd(&p.counters[0], (unsigned long long)a);
    if (b) atomicAdd(&p.counters[1], (unsigned long long)b);
    if (n) atomicAdd(&p.counters[3], (unsigned long long)n);
    if (o) atomicAdd(&p.counters[15], (unsigned long long)o);
  }
}

// Technique::_commit_images (Technique.cpp:194-244) for the frames of a batch, in frame order
__global__ __launch_bounds__(256) void bpt_commit(const RenderParams p, const BptState w) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  uint32_t errors = 0;
  if (i < p.width * p.height) {
    const uint32_t y = i / p.width, x = i - y * p.width;
    const bool inside = x >= p.win_x0 && x < p.win_x0 + p.win_w && y >= p.win_y0 && y < p.win_y0 + p.win_h;
    double* o = p.partial + 4 * size_t(i);
    double a0 = o[0], a1 = o[1], a2 = o[2], a3 = o[3];
    for (uint32_t f = 0; f < w.frames; ++f) {
      double* l = w.light + 3 * (size_t(f) * p.width * p.height + i); float* e = w.eye + 3 * (size_t(f) * p.width * p.height + i);
      if (inside) {
        const double v0 = l[0] + double(e[0]), v1 = l[1] + double(e[1]), v2 = l[2] + double(e[2]);
        if (isfinite(fabs(v0) + fabs(v1) + fabs(v2))) { a0 += v0; a1 += v1; a2 += v2; a3 += 1.0; } else ++errors;
      }
      l[0] = l[1] = l[2] = 0.0; e[0] = e[1] = e[2] = 0.0f;
    }
    o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
  }
  for (int k = 32; k > 0; k >>= 1) errors += __shfl_xor(errors, k, 64);
  if (errors && p.counters && (threadIdx.x & 63u) == 0) atomicAdd(&p.counters[2], (unsigned long long)errors);
}

hipError_t bpt_launch_frame(const RenderParams& p, const BptState& w, bool list, hipStream_t stream) {
  const size_t lds = size_t(p.stack_entries) * kBlock * 4;
  const dim3 grid((w.lanes + kBlock - 1) / kBlock), block(kBlock);
  if (list) { if (p.wide_nodes == 1u) hipLaunchKernelGGL((bpt_frame<true, 2>), grid, block, lds, stream, p, w); else hipLaunchKernelGGL((bpt_frame<true, 1>), grid, block, lds, stream, p, w); }
  else { if (p.wide_nodes == 1u) hipLaunchKernelGGL((bpt_frame<false, 2>), grid, block, lds, stream, p, w); else hipLaunchKernelGGL((bpt_frame<false, 1>), grid, block, lds, stream, p, w); }
  return hipGetLastError();
}
// Dynamic LDS of the staged kernels: [scene copy][traversal stack rows].  bpt_trace / bpt_items stage the flat leaf list when p.flat_k is set
// (flat_scene_f4: 25 K + tables float4), bpt_visibility always the tree copy (blob + one float4 of padding per node and triangle) — with few pair leaves
// the flat copy is the LARGER one, so the launch reserves the larger of the two in front of the stack rows.
static size_t bpt_stage_lds_bytes(const RenderParams& p, bool lds_scene) {
  size_t scene = 0;
  if (lds_scene) {
    const size_t tree = size_t(p.sv.blob_f4 + p.sv.n_nodes + p.sv.n_tris) * 16;
    const size_t flat = p.flat_k ? size_t(kFlatLeafF4 * p.flat_k + 18u * p.flat_k + (p.sv.blob_f4 - p.sv.off_mats)) * 16 : 0;
    scene = tree > flat ? tree : flat;
  }
  return scene + size_t(p.stack_entries) * kBlock * 4;
}
// staged form: trace + scan (returns the number of connection items of the launch's paths), then items + gather
hipError_t bpt_stage_trace(const RenderParams& p, const BptState& w, bool list, bool lds_scene, hipStream_t stream, uint32_t* total_items) {
  const size_t lds = bpt_stage_lds_bytes(p, lds_scene);
  const dim3 grid((w.lanes + kBlock - 1) / kBlock), block(kBlock);
  hipError_t e = hipMemsetAsync(w.item_offset + w.lanes, 0, sizeof(uint32_t), stream);
  if (e != hipSuccess) return e;
  void (*fn)(const RenderParams, const BptState) = nullptr;
  if (lds_scene) fn = list ? bpt_trace<true, 0> : bpt_trace<false, 0>;
  else if (p.sv.n_tris >= 16384u) {  // walks bound by dependent fetches: occupancy before registers
    if (p.wide_nodes == 1u) fn = list ? bpt_trace<true, 2, 6> : bpt_trace<false, 2, 6>;
    else fn = list ? bpt_trace<true, 1, 6> : bpt_trace<false, 1, 6>;
  }
  else if (p.wide_nodes == 1u) fn = list ? bpt_trace<true, 2> : bpt_trace<false, 2>;
  else fn = list ? bpt_trace<true, 1> : bpt_trace<false, 1>;
  // r04: the two sub-paths as two kernels of resident waves with path regeneration (bpt_trace_light / bpt_trace_eye) where they pay (profiles/r04/ab_bpt_steps.txt #4,
  // #7, #9): the models walked at six waves per SIMD (LivingRoomLit -23 %, MetalRings -13 %) and, of the smaller scenes read from HBM, those whose paths are long —
  // every one with >= 6.7 closest-hit rays per path gains (-8 .. -37 %), every one with <= 6 loses (+5 .. +85 %: two kernels and a cursor for paths of two vertices);
  // the host measures that on the launches it has finished (BptState::persist_hint).  Scenes in LDS keep one lane per path (67 -> 99 ms with regeneration).
  // MI_BPT_PERSIST=0 / 1 forces one lane per path / this form.
  bool persist = !lds_scene && (p.sv.n_tris >= 16384u || w.persist_hint == 2u);
  if (const char* v = std::getenv("MI_BPT_PERSIST")) persist = std::atoi(v) != 0 && !lds_scene;
  if (persist && w.step_count) {
    void (*fl)(const RenderParams, const BptState) = nullptr;
    void (*fe)(const RenderParams, const BptState) = nullptr;
    const uint32_t waves = 6u;
    // register budget as for bpt_trace: six waves per SIMD (80 VGPRs) on the models whose walk is a chain of dependent fetches from L2 / HBM, four (128) on small
    // trees — CornellBoxSpecular 103.0 -> 93.4 ms with regeneration at four, LivingRoomLit 134.9 (six) against 139.6 (four); MI_BPT_PERSIST_WAVES=4/6 forces one
    bool four = p.sv.n_tris < 16384u;
    if (const char* v = std::getenv("MI_BPT_PERSIST_WAVES")) four = std::atoi(v) == 4;
    if (four) {
      if (p.wide_nodes == 1u) { fl = list ? bpt_trace_light<true, 2, 4> : bpt_trace_light<false, 2, 4>; fe = list ? bpt_trace_eye<true, 2, 4> : bpt_trace_eye<false, 2, 4>; }
      else { fl = list ? bpt_trace_light<true, 1, 4> : bpt_trace_light<false, 1, 4>; fe = list ? bpt_trace_eye<true, 1, 4> : bpt_trace_eye<false, 1, 4>; }
    } else
    if (p.wide_nodes == 1u) { fl = list ? bpt_trace_light<true, 2, 6> : bpt_trace_light<false, 2, 6>; fe = list ? bpt_trace_eye<true, 2, 6> : bpt_trace_eye<false, 2, 6>; }
    else { fl = list ? bpt_trace_light<true, 1, 6> : bpt_trace_light<false, 1, 6>; fe = list ? bpt_trace_eye<true, 1, 6> : bpt_trace_eye<false, 1, 6>; }
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(fl), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(fe), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (e != hipSuccess) return e;
    static const uint32_t n_cu = [] { int dev = 0, v = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256; return uint32_t(v); }();
    // resident waves only, and not all of them: four workgroups per CU of the six the register budget allows (512 / 768 / 1024 / 1536 / 3072 workgroups: 139.5 / 134.6 /
    // 133.2 / 138.0 / 140.1 ms LivingRoomLit, 48.3 / 47.3 / 46.6 / 49.9 / 50.7 ms MetalRings) — more paths per lane, and room for the kernels of the other launch in flight
    (void)waves;
    uint32_t blocks = n_cu * 4u;
    if (const char* v = std::getenv("MI_BPT_PERSIST_BLOCKS")) { const long b = std::atol(v); if (b > 0) blocks = uint32_t(b); }  // measurement: fewer resident waves, more paths per lane
    if (blocks > grid.x) blocks = grid.x;
    e = hipMemsetAsync(w.step_count + 4, 0, 2 * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fl, dim3(blocks), block, lds, stream, p, w);
    hipLaunchKernelGGL(fe, dim3(blocks), block, lds, stream, p, w);
  } else {
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fn, grid, block, lds, stream, p, w);
  }
  {
    const uint32_t total = w.lanes + 1u, tiles = (total + kScanTile - 1u) / kScanTile;
    hipLaunchKernelGGL(bpt_scan_tiles, dim3(tiles), dim3(256), 0, stream, w.item_offset, total, w.scan_tmp);
    hipLaunchKernelGGL(bpt_scan_sums, dim3(1), dim3(1024), 0, stream, w.scan_tmp, tiles);
    hipLaunchKernelGGL(bpt_scan_add, dim3((total + 255u) / 256u), dim3(256), 0, stream, w.item_offset, total, w.scan_tmp);
  }
  e = hipMemcpyAsync(total_items, w.item_offset + w.lanes, sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
  if (e != hipSuccess) return e;
  if (w.async_total) return hipGetLastError();  // the caller waits for this stream when it needs the count (mi_bpt_render: launches in flight on several streams)
  e = hipStreamSynchronize(stream);
  if (e != hipSuccess) return e;
  return hipGetLastError();
}
// stage A as uniform steps: MI_BPT_STEP_ROUNDS rounds of (bpt_closest, bpt_step) — the list of paths in flight and its count stay on the device, no round waits
// for the host, the grids shrink with the expected survivors and stride over whatever the count turns out to be — then the tail kernel (paths still in flight
// walk their remaining rays themselves), then the scan of the item counts
hipError_t bpt_stage_trace_steps(const RenderParams& p, const BptState& w, bool list, hipStream_t stream, uint32_t* total_items, uint32_t* rounds) {
  hipError_t e = hipMemsetAsync(w.item_offset + w.lanes, 0, sizeof(uint32_t), stream);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(w.step_count, 0, 2 * sizeof(uint32_t), stream);
  if (e != hipSuccess) return e;
  const bool wide = p.wide_nodes == 1u;
  void (*step)(const RenderParams, const BptState, int, uint32_t, uint32_t, uint32_t) = list ? bpt_step<true, false, 1> : bpt_step<false, false, 1>;
  void (*tail)(const RenderParams, const BptState, int, uint32_t, uint32_t, uint32_t) = list ? (wide ? bpt_step<true, true, 2> : bpt_step<true, true, 1>) : (wide ? bpt_step<false, true, 2> : bpt_step<false, true, 1>);
  const size_t lds = size_t(p.stack_entries) * kBlock * 4;
  void (*walk)(const RenderParams, const BptState, uint32_t, uint32_t, uint32_t) = wide ? bpt_closest<2> : bpt_closest<1>;
  uint32_t walk_th = 16u;
  if (const char* t = std::getenv("MI_BPT_STEP_TH")) { const int v = std::atoi(t); if (v >= 1 && v <= 64) walk_th = uint32_t(v); }
  // persistent walkers: at most 8 workgroups per CU are resident; a launch for fewer rays than that brings fewer workgroups
  auto launch_walk = [&](uint32_t slot_, double share) -> hipError_t {
    hipError_t e2 = hipMemsetAsync(w.step_count + 8, 0, kStepCursors * sizeof(uint32_t), stream);
    if (e2 != hipSuccess) return e2;
    uint32_t want = uint32_t(double(w.lanes) * share / double(kStepChunk * kWavesPerBlock)) + 64u;
    const uint32_t cap = 256u * 6u;
    if (want > cap) want = cap;
    hipLaunchKernelGGL(walk, dim3(want), dim3(kBlock), lds, stream, p, w, slot_, want < kStepCursors ? want : kStepCursors, walk_th);
    return hipSuccess;
  };
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(walk), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(tail), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
  if (e != hipSuccess) return e;
  const uint32_t all_blocks = (w.lanes + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(step, dim3(all_blocks), dim3(kBlock), 0, stream, p, w, -1, 0u, 0u, 0u);
  uint32_t slot = 0;
  int n_rounds = MI_BPT_STEP_ROUNDS;
  if (const char* r = std::getenv("MI_BPT_STEP_ROUNDS")) { const int v = std::atoi(r); if (v >= 0 && v <= 4096) n_rounds = v; }
  double expect = 1.0;  // upper estimate of the share of the launch's paths still in flight: a sub-path survives a round with probability < roulette
  for (int r = 0; r < n_rounds; ++r) {
    uint32_t blocks = uint32_t(double(all_blocks) * expect * 1.25) + 64u;
    if (blocks > all_blocks) blocks = all_blocks;
    e = hipMemsetAsync(w.step_count + (slot ^ 1u), 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    e = launch_walk(slot, expect);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(step, dim3(blocks), dim3(kBlock), 0, stream, p, w, int(slot), slot ^ 1u, 0u, 0u);
    slot ^= 1u;
    const double q = p.roulette < 0.98f ? double(p.roulette) * 1.02 : 1.0;  // a path is in flight while either of its two sub-paths is: P(sum of two lengths > r)
    expect = q >= 1.0 ? 1.0 : std::min(1.0, (1.0 + (1.0 - q) * double(r + 1)) * std::pow(q, double(r + 1)) * 1.5);
  }
  {  // the paths still in flight have a ray waiting: one more walk, then they run to their ends in the tail kernel (which walks its further rays itself)
    uint32_t blocks = uint32_t(double(all_blocks) * expect * 1.25) + 64u;
    if (blocks > all_blocks) blocks = all_blocks;
    e = launch_walk(slot, expect);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(tail, dim3(blocks), dim3(kBlock), lds, stream, p, w, int(slot), slot ^ 1u, 0u, 0u);
  }
  if (rounds) *rounds = uint32_t(n_rounds);
  {
    const uint32_t total = w.lanes + 1u, tiles = (total + kScanTile - 1u) / kScanTile;
    hipLaunchKernelGGL(bpt_scan_tiles, dim3(tiles), dim3(256), 0, stream, w.item_offset, total, w.scan_tmp);
    hipLaunchKernelGGL(bpt_scan_sums, dim3(1), dim3(1024), 0, stream, w.scan_tmp, tiles);
    hipLaunchKernelGGL(bpt_scan_add, dim3((total + 255u) / 256u), dim3(256), 0, stream, w.item_offset, total, w.scan_tmp);
  }
  e = hipMemcpyAsync(total_items, w.item_offset + w.lanes, sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
  if (e != hipSuccess) return e;
  e = hipStreamSynchronize(stream);
  if (e != hipSuccess) return e;
  return hipGetLastError();
}
// stage A as PASSES with growing ray budgets (MI_BPT_STEPS=2): every path walks its own rays (the tail kernel's form) but is suspended after `budget` rays of a
// pass; the survivors of a pass are packed into full waves for the next one, so a wave waits for at most `budget` rays instead of its longest path.
hipError_t bpt_stage_trace_passes(const RenderParams& p, const BptState& w, bool list, hipStream_t stream, uint32_t* total_items, uint32_t* rounds) {
  hipError_t e = hipMemsetAsync(w.item_offset + w.lanes, 0, sizeof(uint32_t), stream);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(w.step_count, 0, 2 * sizeof(uint32_t), stream);
  if (e != hipSuccess) return e;
  const bool wide = p.wide_nodes == 1u;
  void (*pass)(const RenderParams, const BptState, int, uint32_t, uint32_t, uint32_t) = list ? (wide ? bpt_step<true, true, 2> : bpt_step<true, true, 1>) : (wide ? bpt_step<false, true, 2> : bpt_step<false, true, 1>);
  const size_t lds = size_t(p.stack_entries) * kBlock * 4;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(pass), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
  if (e != hipSuccess) return e;
  uint32_t caps[8] = {10u, 20u, 40u, 80u, 0u, 0u, 0u, 0u}; int n_caps = 5;
  if (const char* c = std::getenv("MI_BPT_PASS_CAPS")) {  // "12,24,48": budgets of the passes; a last pass without a budget is always added
    n_caps = 0;
    for (const char* q = c; *q && n_caps < 7;) { caps[n_caps++] = uint32_t(std::strtoul(q, nullptr, 10)); while (*q && *q != ',') ++q; if (*q == ',') ++q; }
    caps[n_caps++] = 0u;
  }
  const uint32_t all_blocks = (w.lanes + kBlock - 1) / kBlock;
  uint32_t slot = 0;
  double expect = 1.0;
  for (int k = 0; k < n_caps; ++k) {
    uint32_t blocks = uint32_t(double(all_blocks) * expect * 1.5) + 64u;
    if (blocks < all_blocks / 8u) blocks = all_blocks / 8u;  // the estimate compounds optimistically over several passes; a small grid strides serially over what is left
    if (blocks > all_blocks) blocks = all_blocks;
    if (k == 0) {
      hipLaunchKernelGGL(pass, dim3(all_blocks), dim3(kBlock), lds, stream, p, w, -1, 0u, caps[0], 0u);
    } else {
      e = hipMemsetAsync(w.step_count + (slot ^ 1u), 0, sizeof(uint32_t), stream);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(pass, dim3(blocks), dim3(kBlock), lds, stream, p, w, int(slot), slot ^ 1u, caps[k], 1u);
      slot ^= 1u;
    }
    if (caps[k] == 0u) break;
    const double q = p.roulette < 0.98f ? double(p.roulette) * 1.02 : 1.0;  // P(a path has more than `cap` rays left) <= (1 + (1 - q) cap) q^cap
    expect = q >= 1.0 ? 1.0 : std::min(1.0, expect * (1.0 + (1.0 - q) * double(caps[k])) * std::pow(q, double(caps[k])) * 1.5);
  }
  if (rounds) *rounds = uint32_t(n_caps);
  {
    const uint32_t total = w.lanes + 1u, tiles = (total + kScanTile - 1u) / kScanTile;
    hipLaunchKernelGGL(bpt_scan_tiles, dim3(tiles), dim3(256), 0, stream, w.item_offset, total, w.scan_tmp);
    hipLaunchKernelGGL(bpt_scan_sums, dim3(1), dim3(1024), 0, stream, w.scan_tmp, tiles);
    hipLaunchKernelGGL(bpt_scan_add, dim3((total + 255u) / 256u), dim3(256), 0, stream, w.item_offset, total, w.scan_tmp);
  }
  e = hipMemcpyAsync(total_items, w.item_offset + w.lanes, sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
  if (e != hipSuccess) return e;
  e = hipStreamSynchronize(stream);
  if (e != hipSuccess) return e;
  return hipGetLastError();
}
hipError_t bpt_stage_connect(const RenderParams& p, const BptState& w, bool list, bool lds_scene, uint32_t total_items, hipStream_t stream) {
  const size_t lds = bpt_stage_lds_bytes(p, lds_scene);
  if (total_items) {
    hipError_t e;
    if (w.dyn_vis) {
      // visibility first: ray list (uniform), then persistent waves over it; 8 workgroups per CU at most are resident, later ones find the cursor at the end
      e = hipMemsetAsync(w.pool, 0, kVisCursors * sizeof(uint32_t), stream);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(bpt_rays, dim3((total_items + 255u) / 256u), dim3(256), 0, stream, p, w, total_items);
      void (*vis)(const RenderParams, const BptState, uint32_t, uint32_t) = lds_scene ? bpt_visibility<0> : (w.vis_wide ? bpt_visibility<2> : bpt_visibility<1>);
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(vis), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
      if (e != hipSuccess) return e;
      const uint32_t want = (total_items + kVisChunk * kWavesPerBlock - 1u) / (kVisChunk * kWavesPerBlock), cap = 256u * 8u;
      const uint32_t n_wg = want < cap ? want : cap;
      hipLaunchKernelGGL(vis, dim3(n_wg), dim3(kBlock), lds, stream, p, w, total_items, n_wg < kVisCursors ? n_wg : kVisCursors);
    }
    const dim3 grid((total_items + kBlock - 1) / kBlock), block(kBlock);
    void (*fn)(const RenderParams, const BptState, uint32_t, uint32_t) = nullptr;
    if (w.dyn_vis) {
      if (lds_scene) fn = list ? bpt_items<true, 0, true> : bpt_items<false, 0, true>;
      else if (p.wide_nodes == 1u) fn = list ? bpt_items<true, 2, true> : bpt_items<false, 2, true>;
      else fn = list ? bpt_items<true, 1, true> : bpt_items<false, 1, true>;
    } else {
      if (lds_scene) fn = list ? bpt_items<true, 0, false> : bpt_items<false, 0, false>;
      else if (p.wide_nodes == 1u) fn = list ? bpt_items<true, 2, false> : bpt_items<false, 2, false>;
      else fn = list ? bpt_items<true, 1, false> : bpt_items<false, 1, false>;
    }
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fn, grid, block, lds, stream, p, w, 0u, total_items);
  }
  const uint32_t gb = (w.lanes + 255u) / 256u;
  const dim3 g2(gb < 1024u ? gb : 1024u);
  if (list) hipLaunchKernelGGL(bpt_gather<true>, g2, dim3(256), 0, stream, p, w, 0u);
  else hipLaunchKernelGGL(bpt_gather<false>, g2, dim3(256), 0, stream, p, w, 0u);
  return hipGetLastError();
}
hipError_t bpt_launch_commit(const RenderParams& p, const BptState& w, hipStream_t stream) {
  hipLaunchKernelGGL(bpt_commit, dim3((p.width * p.height + 255u) / 256u), dim3(256), 0, stream, p, w);
  return hipGetLastError();
}

}  // namespace MI_BPT_NS
}  // namespace mi
