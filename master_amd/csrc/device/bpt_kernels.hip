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

#include "pt_device.h"
#include "bpt.h"

namespace mi {

namespace {

struct LVert { Surf surface; f3 omega, throughput; float a, A; int finite; };  // BPT.hpp:14-20
struct EVert { Surf surface; f3 omega, throughput; float c, C; int finite; };  // BPT.hpp:22-28

struct Ctx {
  const float4* sb; const SceneView* sv; TravStack* stack;
  float beta, roulette, rinv;
  f3 sphere_c; float sphere_r;
  uint32_t n_basic, n_shadow;
};

MI_DEV float betaf(const Ctx& c, float x) {  // Beta.hpp:24-41
  if (c.beta == 0.0f) return x == 0.0f ? 0.0f : 1.0f;
  if (c.beta == 1.0f) return x;
  if (c.beta == 2.0f) return x * x;
  return powf(x, c.beta);
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
  return bsdf_query(m, sf, incident, outgoing);
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
  return bsdf_sample(m, g, sf, omega);
}

// Scene::intersect / intersectMesh (Scene.cpp:182-227): closest hit with a geometry mask
template <int QN>
MI_DEV Surf scene_intersect(Ctx& c, const Surf& from, f3 dir, uint32_t mask) {
  const f3 org = nudge(from.position, from.gnormal, dir);
  Hit h; h.t = __builtin_inff(); h.u = h.v = 0.0f; h.id = 0xFFFFFFFFu; h.pos = 0;
  traverse<false, false, QN>(c.sb, *c.sv, *c.stack, org, dir, mask, h);
  ++c.n_basic;
  if (h.id == 0xFFFFFFFFu) { Surf s; s.position = F3(0, 0, 0); s.gnormal = F3(0, 0, 0); s.tangent.c0 = s.tangent.c1 = s.tangent.c2 = F3(0, 0, 0); s.material_id = 0xFFFFFFFFu; return s; }
  return query_surface(c.sb, *c.sv, org, dir, h);
}
template <int QN>
MI_DEV float scene_occluded(Ctx& c, const Surf& origin, const Surf& target) {
  ++c.n_shadow;
  return occluded<false, QN>(c.sb, *c.sv, *c.stack, origin.position, origin.gnormal, target.position, target.gnormal);
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
template <int QN>
MI_DEV f3 bpt_connect(Ctx& c, const LVert& light, const EVert& eye) {
  const f3 omega = normalize(eye.surface.position - light.surface.position);
  const BQuery lb = bpt_bsdf_query(c, light.surface, light.omega, omega);
  const BQuery eb = bpt_bsdf_query(c, eye.surface, -omega, eye.omega);
  const Edge e = make_edge(light.surface, eye.surface, omega);
  const float Ap = (light.A * betaf(c, lb.densityRev) + light.a * float(light.finite)) * betaf(c, e.bG * eb.densityRev);
  const float Cp = (eye.C * betaf(c, eb.density) + eye.c * float(eye.finite)) * betaf(c, e.fG * lb.density);
  const float weightInv = Ap + Cp + 1.0f;
  const float occ = scene_occluded<QN>(c, eye.surface, light.surface);
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
  query_lsdf(c.sb, *c.sv, lm.light_id, eye.omega, le, dens);
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
      if (surface.material_id == 0xFFFFFFFFu) return at_camera ? F3(0, 0, 0) : radiance;  // sky_gradient is zero (Technique.hpp:46-47)
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

}  // namespace

// one lane = one (pixel, sample): shoot() + _traceEye (Technique.cpp:321-338)
#ifndef MI_BPT_WAVES
#define MI_BPT_WAVES 3  // 201 VGPRs without a bound (2 waves); measured on C2-sized BPT frames: 2 waves 804, 3 waves 894, 4 waves 872, 5 waves 754 Mrays/s
#endif
template <bool LIST, int QN>
__global__ __launch_bounds__(kBlock, MI_BPT_WAVES) void bpt_frame(const RenderParams p, const BptState w) {
  extern __shared__ float4 smem[];
  TravStack stack;
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
    c.beta = p.beta; c.roulette = p.roulette; c.rinv = 1.0f / p.roulette;
    c.sphere_c = F3(w.sphere[0], w.sphere[1], w.sphere[2]); c.sphere_r = w.sphere[3];
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
  if (list) { if (p.wide_nodes) hipLaunchKernelGGL((bpt_frame<true, 2>), grid, block, lds, stream, p, w); else hipLaunchKernelGGL((bpt_frame<true, 1>), grid, block, lds, stream, p, w); }
  else { if (p.wide_nodes) hipLaunchKernelGGL((bpt_frame<false, 2>), grid, block, lds, stream, p, w); else hipLaunchKernelGGL((bpt_frame<false, 1>), grid, block, lds, stream, p, w); }
  return hipGetLastError();
}
hipError_t bpt_launch_commit(const RenderParams& p, const BptState& w, hipStream_t stream) {
  hipLaunchKernelGGL(bpt_commit, dim3((p.width * p.height + 255u) / 256u), dim3(256), 0, stream, p, w);
  return hipGetLastError();
}

}  // namespace mi
