/*
 * pt_oracle.c — CPU restatement of the reference's unidirectional path tracer (PT).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (master_amd/, libmi_pt.so) may
 * include, link or call this file.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the reported CPU baseline.
 *
 * What it restates (reference file:line, tree = ciechowoj/master):
 *   Technique::render / _trace_paths / _for_each_ray / _commit_images   Technique.cpp:15-77,163-244,307-352
 *   PathTracing::_traceEye / _connect                                   PT.cpp:15-120
 *   Scene::intersect / occluded / querySurface                          Scene.cpp:80-126,151-203
 *   SurfacePoint / Edge / material-id encoding                          SurfacePoint.hpp:8-83
 *   Diffuse / Phong / Reflection / Transmission / Light / sun BSDFs     BSDF.cpp:95-114,181-191,239-262,291-391,438-504
 *   sample_lambert / sample_phong / reflection_to_surface               Sample.inl:43-60,139-151
 *   AreaLights::sample / queryLSDF / _updateSampler                     AreaLights.cpp:121-155,199-231
 *   Cameras: lookAt frame, fovy, focal_length_y, ray_direction          Cameras.cpp:7-28,81-127
 *   rms_abs_errors                                                      ImageView.cpp:60-85
 *
 * Third-party arithmetic on the path that is NOT in the reference tree:
 *   Embree 2.x (github.com/embree/embree, submodule pin unknown — the submodule directory is
 *   empty): rtcIntersect / rtcOccluded, single-ray API, called at Scene.cpp:175,198.  Restated
 *   here from its published single-ray Moeller–Trumbore triangle test ("TriangleM" layout:
 *   v0, e1 = v0-v1, e2 = v2-v0, Ng = cross(e2, e1); hit if den != 0, U >= 0, V >= 0,
 *   U+V <= |den|, |den|*tnear < T <= |den|*tfar; u = U/|den|, v = V/|den|, t = T/|den|;
 *   P = (1-u-v) v0 + u v1 + v v2; Ng unnormalised; geometry/ray mask test).
 *   Tie-breaking between equal-t hits is not pinned by Embree; this restatement (and the
 *   GPU path) define it as: smaller t wins, equal t -> smaller global triangle index wins.
 *
 * Parity status: PARITY UNPINNED for the hot path.  The reference executable cannot be built in
 * this image (glm, Embree 2, OpenEXR, the assimp fork are absent and may not be stood in for) and
 * the reference holds no executable test, golden vector or image for PT.  What the reference DOES
 * hold pins only the cameras: unit_tests/Cameras.test.cpp:22-44, the inline unittest blocks of
 * Cameras.cpp:164-189, Technique.cpp:118-152, main.cpp:24-56 and unittest.cpp:177-182
 * (tests/test_camera.py, tests/test_oracle_kat.py, tests/golden/reference_constants.json).
 * Everything else — _traceEye / _connect, the BSDFs, light sampling, intersect / occluded — is
 * checked against closed-form results of the estimator it restates (white furnace, analytic
 * rectangle-light irradiance, sampling densities) and by line-by-line citation, not against
 * reference outputs.  unit_test.py's constant 0.01 is a scheduling heuristic of a stale script
 * (see reference_constants.json), not a value to meet: the absolute radiometric scale of the
 * importer, Embree's tie-breaks and the assimp fork's conventions are unpinned.
 * What IS exact: the GPU path equals this restatement bit for bit.
 *
 * Random numbers: the reference PT cannot be seeded (Options.cpp:821-833; every tile reseeds
 * from std::random_device, Technique.cpp:170-174), so streams are defined here, not copied:
 * SplitMix32 (a Weyl sequence through a 32-bit finaliser), seeded per path by the same finaliser over (seed, pixel, sample).
 * The GPU path uses the same definition, so per-path results are comparable 1:1.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fPIC -shared -pthread (see Makefile).
 */
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mi_pt.h"

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ vec3 (glm semantics) */
typedef struct { float x, y, z; } v3;
typedef struct { v3 c[3]; } m3; /* column-major like glm::mat3 */

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
/* vec3 / float is DEFINED as one reciprocal and three multiplies (glm divides per component; the
 * difference is <= 1 ulp and the definition is this build's, stated identically on the device) */
static inline v3 vdivs(v3 a, float s) { float r = 1.0f / s; return V(a.x * r, a.y * r, a.z * r); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
/* Arithmetic contract: the file is built with -ffp-contract=off and fuses ONLY where
 * fmaf() is spelled.  glm's dot/cross/mat*vec compiled with the reference's flags
 * (g++ -O2 -march=native, Makefile:28) contract to FMAs as well; the exact placement is a
 * definition of this build, stated identically (and independently) on the device side. */
static inline float vdot(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline v3 vcross(v3 a, v3 b) {
  return V(fmaf(a.y, b.z, -(b.y * a.z)), fmaf(a.z, b.x, -(b.z * a.x)), fmaf(a.x, b.y, -(b.x * a.y)));
}
static inline v3 vmadd(v3 a, v3 b, float s) { return V(fmaf(b.x, s, a.x), fmaf(b.y, s, a.y), fmaf(b.z, s, a.z)); }
/* glm::normalize = v * inversesqrt(dot(v,v)), inversesqrt(x) = 1/sqrt(x) */
static inline v3 vnormalize(v3 a) { return vscale(a, 1.0f / sqrtf(vdot(a, a))); }
static inline float l1norm(v3 a) { return fabsf(a.x) + fabsf(a.y) + fabsf(a.z); }
static inline float gsign(float x) { return (float)((0.0f < x) - (x < 0.0f)); }
static inline v3 ld3(const float* p) { return V(p[0], p[1], p[2]); }
static inline void st3(float* p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
static inline m3 ldm3(const float* p) { m3 m; m.c[0] = ld3(p); m.c[1] = ld3(p + 3); m.c[2] = ld3(p + 6); return m; }
static inline void stm3(float* p, m3 m) { st3(p, m.c[0]); st3(p + 3, m.c[1]); st3(p + 6, m.c[2]); }
/* glm mat3 * vec3 */
static inline v3 m3mulv(m3 m, v3 v) {
  return V(fmaf(m.c[2].x, v.z, fmaf(m.c[1].x, v.y, m.c[0].x * v.x)),
           fmaf(m.c[2].y, v.z, fmaf(m.c[1].y, v.y, m.c[0].y * v.x)),
           fmaf(m.c[2].z, v.z, fmaf(m.c[1].z, v.y, m.c[0].z * v.x)));
}
/* glm vec3 * mat3  (SurfacePoint::toSurface, SurfacePoint.hpp:50) */
static inline v3 vmulm3(v3 v, m3 m) { return V(vdot(m.c[0], v), vdot(m.c[1], v), vdot(m.c[2], v)); }
static m3 m3inverse(m3 m) { /* glm::inverse(mat3): cofactors * (1/det) */
  float a = m.c[0].x, b = m.c[0].y, c = m.c[0].z;
  float d = m.c[1].x, e = m.c[1].y, f = m.c[1].z;
  float g = m.c[2].x, h = m.c[2].y, i = m.c[2].z;
  float inv = 1.0f / (a * (e * i - h * f) - d * (b * i - h * c) + g * (b * f - e * c));
  m3 r;
  r.c[0] = V((e * i - h * f) * inv, -(b * i - h * c) * inv, (b * f - e * c) * inv);
  r.c[1] = V(-(d * i - g * f) * inv, (a * i - g * c) * inv, -(a * f - d * c) * inv);
  r.c[2] = V((d * h - g * e) * inv, -(a * h - g * b) * inv, (a * e - d * b) * inv);
  return r;
}
static m3 m3transpose(m3 m) {
  m3 r;
  r.c[0] = V(m.c[0].x, m.c[1].x, m.c[2].x);
  r.c[1] = V(m.c[0].y, m.c[1].y, m.c[2].y);
  r.c[2] = V(m.c[0].z, m.c[1].z, m.c[2].z);
  return r;
}

#define ONE_OVER_PI 0.318309886183790671537767526745028724f
#define PI_F 3.14159265358979323846264338327950288f

/* sin/cos of phi = (u*2)*pi, u in [0,1) (Sample.inl:55,146).  libm and the device math library
 * differ in the last bits, so the function is DEFINED here (quadrant reduction on u, Cephes
 * sinf/cosf kernels on |theta| <= pi/4; abs error ~1.5e-7) and stated identically on the device. */
static inline void sincos_2pi(float u, float* s, float* c) {
  float k = floorf(fmaf(u, 4.0f, 0.5f));
  float r = fmaf(k, -0.25f, u);
  float t = r * 6.28318530717958647692f;
  float t2 = t * t;
  float ps = fmaf(fmaf(-1.9515295891e-4f, t2, 8.3321608736e-3f), t2, -1.6666654611e-1f);
  float sn = fmaf(t * t2, ps, t);
  float pc = fmaf(fmaf(2.443315711809948e-5f, t2, -1.388731625493765e-3f), t2, 4.166664568298827e-2f);
  float cs = fmaf(t2 * t2, pc, fmaf(t2, -0.5f, 1.0f));
  int q = (int)k & 3;
  float so = (q & 1) ? cs : sn, co = (q & 1) ? sn : cs;
  *s = (q == 2 || q == 3) ? -so : so;
  *c = (q == 1 || q == 2) ? -co : co;
}


/* asin, atan2 and sin/cos of an angle in radians for the bounded cosine sampling of the emitters (Sample.inl:5-37,62-137).
 * DEFINED here (Cephes asinf / atanf kernels, every step an explicit fma or a single operation) and stated identically on the
 * device, like sincos_2pi: libm and the device library differ in the last bits.  |error| ~ 1e-7. */
static inline float mi_asinf(float x) {
  float a = fabsf(x), z, xs; int big = a > 0.5f;
  if (big) { z = 0.5f * (1.0f - a); xs = sqrtf(z); } else { xs = a; z = a * a; }
  float p = fmaf(fmaf(fmaf(fmaf(4.2163199048e-2f, z, 2.4181311049e-2f), z, 4.5470025998e-2f), z, 7.4953002686e-2f), z, 1.6666752422e-1f);
  float r = fmaf(p * z, xs, xs);
  if (big) r = 1.57079632679489661923f - (r + r);
  return x < 0.0f ? -r : r;
}
static inline float mi_atanf(float t) {
  float x = fabsf(t), y;
  if (x > 2.414213562373095f) { y = 1.57079632679489661923f; x = -(1.0f / x); }
  else if (x > 0.4142135623730950f) { y = 0.785398163397448309616f; x = (x - 1.0f) / (x + 1.0f); }
  else y = 0.0f;
  float z = x * x;
  float p = fmaf(fmaf(fmaf(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z, -3.33329491539e-1f);
  y = y + fmaf(p * z, x, x);
  return t < 0.0f ? -y : y;
}
static inline float mi_atan2f(float y, float x) {
  if (x > 0.0f) return mi_atanf(y / x);
  if (x < 0.0f) return y < 0.0f ? mi_atanf(y / x) - 3.14159265358979323846f : mi_atanf(y / x) + 3.14159265358979323846f;
  return y > 0.0f ? 1.57079632679489661923f : (y < 0.0f ? -1.57079632679489661923f : 0.0f);
}
static inline void mi_sincosf(float rad, float* s, float* c) {
  float t = rad * 0.159154943091895335769f; /* turns */
  sincos_2pi(t - floorf(t), s, c);
}

/* pow(x, y) of the Phong lobe (BSDF.cpp:306-391, Sample.inl:139-151) and of the MIS weights with a variable beta (Beta.hpp:24-41).
 * DEFINED here like the functions above (glibc's powf and the device library's differ in the last bit): 2^(y log2 x) in FP64, each
 * step a single IEEE operation or an explicit fma, rounded once to FP32 — within 1 ulp of the exact power, so it is the reference's
 * std::pow up to a rare last-bit tie.  Special cases follow C99 pow; results below FLT_MIN are zero (FTZ, main.cpp:70-71). */
static inline uint64_t mi_d2u(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static inline double mi_u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
#define MI_D2U(d) mi_d2u(d)
#define MI_U2D(u) mi_u2d(u)
static inline float mi_powf(float x, float y) {
  if (y == 0.0f || x == 1.0f) return 1.0f;
  if (x != x || y != y) return x + y;
  const float ax = fabsf(x), ay = fabsf(y);
  const int y_int = floorf(y) == y;
  const int y_odd = y_int && ay < 16777216.0f && (((int)y) & 1);
  const int neg = x < 0.0f;
  if (neg && !y_int) return __builtin_nanf("");
  float r;
  if (ax == 1.0f) r = 1.0f;
  else if (ax == 0.0f) r = y > 0.0f ? 0.0f : __builtin_inff();
  else if (ax == __builtin_inff()) r = y > 0.0f ? __builtin_inff() : 0.0f;
  else if (ay == __builtin_inff()) r = ((ax < 1.0f) == (y > 0.0f)) ? 0.0f : __builtin_inff();
  else {
    /* log2(ax): ax = m * 2^e, m in (sqrt(1/2), sqrt(2)]; ln m = 2 atanh(s), s = (m - 1) / (m + 1), |s| <= 0.1716 */
    const uint64_t u = MI_D2U((double)ax);
    int e = (int)(u >> 52) - 1023;
    double m = MI_U2D((u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    double p = fma(0.11764705882352941, z, 0.13333333333333333);
    p = fma(p, z, 0.15384615384615385);
    p = fma(p, z, 0.18181818181818182);
    p = fma(p, z, 0.22222222222222221);
    p = fma(p, z, 0.2857142857142857);
    p = fma(p, z, 0.4);
    p = fma(p, z, 0.66666666666666663);
    const double ln_m = fma(s * z, p, s + s);
    const double t = (double)y * fma(ln_m, 1.4426950408889634, (double)e);
    /* 2^t: t = n + q, |q| <= 1/2, e^(q ln 2) by its Taylor polynomial of degree 12; float results below FLT_MIN are flushed */
    if (t >= 128.0) r = __builtin_inff();
    else if (t < -126.0) r = 0.0f;
    else {
      const double n = floor(t + 0.5);
      const double w = (t - n) * 0.69314718055994531;
      double c = fma(2.08767569878681e-9, w, 2.505210838544172e-8);
      c = fma(c, w, 2.7557319223985888e-7);
      c = fma(c, w, 2.7557319223985893e-6);
      c = fma(c, w, 2.4801587301587302e-5);
      c = fma(c, w, 1.9841269841269841e-4);
      c = fma(c, w, 1.3888888888888889e-3);
      c = fma(c, w, 8.3333333333333332e-3);
      c = fma(c, w, 4.1666666666666664e-2);
      c = fma(c, w, 0.16666666666666666);
      c = fma(c, w, 0.5);
      c = fma(c, w, 1.0);
      c = fma(c, w, 1.0);
      r = (float)(c * MI_U2D((uint64_t)(1023 + (int)n) << 52));
    }
  }
  return neg && y_odd ? -r : r;
}

ORC_API void orc_powf(uint32_t n, const float* x, const float* y, float* out) { for (uint32_t i = 0; i < n; ++i) out[i] = mi_powf(x[i], y[i]); }
/* the build's own sin / cos (argument in turns) and asin, exposed so tests can hold them against the reference's start-up assertions (main.cpp:24-39) */
ORC_API void orc_sincos_2pi(float u, float* s, float* c) { sincos_2pi(u, s, c); }
ORC_API float orc_asinf(float x) { return mi_asinf(x); }

/* ------------------------------------------------------------------ RNG (defined here) */
/* The reference draws from mt19937 and cannot be seeded (Sample.hpp:9-31, Options.cpp:821-833): the stream is this build's definition, keyed on
 * (seed, pixel, sample).  Round 4: one 32-bit word — SplitMix32 draws (Weyl step 0x9E3779B9, "lowbias32" finaliser: two 32-bit multiplies), seeded by
 * two rounds of the same finaliser.  master_amd/csrc/device/rng.h states the same; tests/golden/rng_kat.json pins both. */
typedef struct { uint32_t state; } rng_t;
static inline uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x21F0AAADu;
  x ^= x >> 15; x *= 0x735A2D97u;
  x ^= x >> 15; return x;
}
static inline rng_t rng_seed(uint64_t seed, uint32_t pixel_index, uint64_t sample_index) {
  const uint32_t a = mix32((uint32_t)seed ^ mix32((uint32_t)(seed >> 32) + 0x9E3779B9u));
  rng_t r;
  r.state = mix32(mix32(a ^ pixel_index) ^ (uint32_t)sample_index ^ ((uint32_t)(sample_index >> 32) * 0x9E3779B1u));
  return r;
}
static inline uint32_t rng_u32(rng_t* r) {
  r->state += 0x9E3779B9u;
  return mix32(r->state);
}
/* uniform in [0,1): top 24 bits (random_generator_t::sample<float>, Sample.inl:259-262) */
static inline float rng_f(rng_t* r) { return (float)(rng_u32(r) >> 8) * 0x1p-24f; }

/* ------------------------------------------------------------------ scene */
typedef struct {
  v3 v0, e1, e2, ng; /* Embree TriangleM layout: e1 = v0-v1, e2 = v2-v0, ng = cross(e2,e1) */
  uint32_t id;       /* global triangle index */
  uint32_t mask;     /* 1u << (material_id & 3)  (Scene.cpp:42) */
} tri_t;

typedef struct {
  float lo[2][3], hi[2][3];
  int32_t link[2];
  uint32_t parent;
} bnode_t;

#ifndef ORC_COUNT_NODE /* hooks for tests/lab (tree-quality experiments); empty in the oracle proper */
#define ORC_COUNT_NODE(closest)
#define ORC_COUNT_TRI(closest)
#define ORC_RAY_BEGIN(closest)
#endif
#define ORC_STACK 256 /* traversal stack entries; the build aborts on a deeper tree */

typedef struct orc_scene {
  mi_scene_desc d; /* deep copy */
  float* positions; float* tangents; uint32_t* indices; uint32_t* mesh_tri_offset;
  uint32_t* mesh_material_id; mi_material* materials; mi_light* lights; mi_camera* cameras;
  uint32_t* tri_material; /* [n_triangles] encoded material id of the owning mesh */
  tri_t* tris;            /* [n_triangles] in ORIGINAL order (brute force)        */
  /* light sampler (AreaLights::_updateSampler, AreaLights.cpp:199-214) */
  float* light_weight; float* light_cdf; /* cdf[n_lights+1] */
  /* BVH restatement (same algorithm as the device builder; bit-exact comparable) */
  uint32_t n_nodes; bnode_t* nodes; uint32_t* sorted_tri; uint64_t* morton; tri_t* tris_sorted;
  float scene_lo[3], scene_hi[3]; uint32_t max_depth, builder, build_rounds;
  /* PT params */
  mi_pt_params p;
  int use_bvh;
  float sky_horizon[3], sky_zenith[3]; /* Technique::_sky_horizon / _sky_zenith (Technique.hpp:46-47), used by BPT only */
} orc_scene;

typedef struct {
  v3 position, gnormal; m3 tangent; uint32_t material_id;
} surf_t;

static inline int s_is_light(const surf_t* s) { return (s->material_id & 3u) == MI_ENTITY_LIGHT; }
static inline int s_is_present(const surf_t* s) { return s->material_id != UINT32_MAX; }
static inline v3 s_normal(const surf_t* s) { return s->tangent.c[1]; }
static inline v3 s_to_world(const surf_t* s, v3 v) { return m3mulv(s->tangent, v); }
static inline v3 s_to_surface(const surf_t* s, v3 v) { return vmulm3(v, s->tangent); }

/* ------------------------------------------------------------------ BVH2 build (sequential restatement of
 * master_amd/csrc/device/bvh_build.hip: 63-bit Morton order, then PLOC clustering or the Karras 2012 hierarchy;
 * the tree is bit-comparable with the device's) */
static inline uint64_t expand_bits21(uint32_t v) {
  uint64_t x = v & 0x1FFFFFu;
  x = (x | (x << 32)) & 0x001F00000000FFFFull;
  x = (x | (x << 16)) & 0x001F0000FF0000FFull;
  x = (x | (x << 8)) & 0x100F00F00F00F00Full;
  x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
  x = (x | (x << 2)) & 0x1249249249249249ull;
  return x;
}
static inline uint32_t quant21(float c, float lo, float hi) {
  float ext = hi - lo;
  float n = ext > 0.0f ? (c - lo) / ext : 0.0f;
  float q = n * 2097152.0f;
  if (!(q > 0.0f)) q = 0.0f;
  if (q > 2097151.0f) q = 2097151.0f;
  return (uint32_t)q;
}
static inline void tri_bounds(const orc_scene* s, uint32_t t, float lo[3], float hi[3]) {
  for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; }
  for (int k = 0; k < 3; ++k) {
    const float* p = s->positions + 3 * (size_t)s->indices[3 * (size_t)t + k];
    for (int a = 0; a < 3; ++a) { if (p[a] < lo[a]) lo[a] = p[a]; if (p[a] > hi[a]) hi[a] = p[a]; }
  }
}
/* leaf boxes are padded so that a hit the FP triangle test accepts marginally outside the exact
 * triangle is never culled by the box test (same formula on the device). */
static inline void pad_box(float lo[3], float hi[3]) {
  for (int a = 0; a < 3; ++a) {
    float m = fmaxf(fabsf(lo[a]), fabsf(hi[a]));
    float pad = m * 0x1p-20f + 0x1p-40f;
    lo[a] = lo[a] - pad; hi[a] = hi[a] + pad;
  }
}
typedef struct { uint64_t code; uint32_t id; } mkey_t;
static int cmp_key(const void* a, const void* b) {
  const mkey_t *x = (const mkey_t*)a, *y = (const mkey_t*)b;
  if (x->code != y->code) return x->code < y->code ? -1 : 1;
  return x->id < y->id ? -1 : x->id > y->id ? 1 : 0;
}
static inline int delta_fn(const mkey_t* keys, int n, int i, int j) { /* common prefix of the 96-bit key (code, id) */
  if (j < 0 || j >= n) return -1;
  uint64_t x = keys[i].code ^ keys[j].code;
  return x ? __builtin_clzll(x) : 64 + __builtin_clz(keys[i].id ^ keys[j].id);
}
static void node_child_box(const orc_scene* s, int32_t link, float lo[3], float hi[3]) {
  if (link < 0) {
    tri_bounds(s, s->sorted_tri[~link], lo, hi);
    pad_box(lo, hi);
  } else {
    const bnode_t* n = &s->nodes[link];
    for (int a = 0; a < 3; ++a) {
      lo[a] = fminf(n->lo[0][a], n->lo[1][a]);
      hi[a] = fmaxf(n->hi[0][a], n->hi[1][a]);
    }
  }
}
static void refit(orc_scene* s, int32_t node) { /* post-order */
  bnode_t* n = &s->nodes[node];
  for (int c = 0; c < 2; ++c) {
    if (n->link[c] >= 0) refit(s, n->link[c]);
    node_child_box(s, n->link[c], n->lo[c], n->hi[c]);
  }
}
static uint32_t depth_of(const orc_scene* s, int32_t link) {
  if (link < 0) return 1;
  uint32_t a = depth_of(s, s->nodes[link].link[0]), b = depth_of(s, s->nodes[link].link[1]);
  return 1 + (a > b ? a : b);
}
/* PLOC (Meister & Bittner 2018) as the device runs it: clusters in Morton order, every cluster picks the partner
 * within 16 places minimising (union area, pair hash, min index, max index), mutual pairs merge, survivors keep
 * their order.  Nodes are numbered downwards so that the root is node 0. */
#define PLOC_RADIUS 16
typedef struct { float lo[3], hi[3]; int32_t link; } cluster_t;
static inline float union_area(const cluster_t* a, const cluster_t* b) {
  float dx = fmaxf(a->hi[0], b->hi[0]) - fminf(a->lo[0], b->lo[0]);
  float dy = fmaxf(a->hi[1], b->hi[1]) - fminf(a->lo[1], b->lo[1]);
  float dz = fmaxf(a->hi[2], b->hi[2]) - fminf(a->lo[2], b->lo[2]);
  return fmaf(dx, dy, fmaf(dy, dz, dz * dx));
}
static inline uint32_t pair_hash(uint32_t a, uint32_t b) {
  uint32_t h = (a * 0x9E3779B1u) ^ (b * 0x85EBCA77u + 0x165667B1u);
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}
static int build_ploc(orc_scene* s, int n, uint32_t* rounds_out) {
  cluster_t* a = (cluster_t*)malloc(sizeof(cluster_t) * (size_t)n);
  cluster_t* b = (cluster_t*)malloc(sizeof(cluster_t) * (size_t)n);
  int* nn = (int*)malloc(sizeof(int) * (size_t)n);
  for (int i = 0; i < n; ++i) {
    tri_bounds(s, s->sorted_tri[i], a[i].lo, a[i].hi);
    pad_box(a[i].lo, a[i].hi);
    a[i].link = ~i;
  }
  int cur = n; uint32_t next = (uint32_t)(n - 1), rounds = 0;
  while (cur > 1) {
    for (int i = 0; i < cur; ++i) {
      int best = -1; float best_area = 0.0f;
      int j0 = i - PLOC_RADIUS < 0 ? 0 : i - PLOC_RADIUS, j1 = i + PLOC_RADIUS > cur - 1 ? cur - 1 : i + PLOC_RADIUS;
      for (int j = j0; j <= j1; ++j) {
        if (j == i) continue;
        float area = union_area(&a[i], &a[j]);
        int take;
        if (best < 0) take = 1;
        else if (area != best_area) take = area < best_area;
        else {
          uint32_t a0 = (uint32_t)(i < j ? i : j), a1 = (uint32_t)(i < j ? j : i);
          uint32_t b0 = (uint32_t)(i < best ? i : best), b1 = (uint32_t)(i < best ? best : i);
          uint32_t ha = pair_hash(a0, a1), hb = pair_hash(b0, b1);
          take = ha != hb ? ha < hb : (a0 != b0 ? a0 < b0 : a1 < b1);
        }
        if (take) { best = j; best_area = area; }
      }
      nn[i] = best;
    }
    uint32_t leaders = 0; int out = 0;
    for (int i = 0; i < cur; ++i) {
      int j = nn[i];
      int mutual = nn[j] == i;
      if (mutual && i > j) continue; /* absorbed by its partner */
      if (!mutual) { b[out++] = a[i]; continue; }
      uint32_t node = next - 1u - leaders++;
      bnode_t* nd = &s->nodes[node];
      for (int k = 0; k < 3; ++k) {
        nd->lo[0][k] = a[i].lo[k]; nd->hi[0][k] = a[i].hi[k];
        nd->lo[1][k] = a[j].lo[k]; nd->hi[1][k] = a[j].hi[k];
        b[out].lo[k] = fminf(a[i].lo[k], a[j].lo[k]); b[out].hi[k] = fmaxf(a[i].hi[k], a[j].hi[k]);
      }
      nd->link[0] = a[i].link; nd->link[1] = a[j].link; nd->parent = UINT32_MAX;
      if (a[i].link >= 0) s->nodes[a[i].link].parent = node;
      if (a[j].link >= 0) s->nodes[a[j].link].parent = node;
      b[out++].link = (int32_t)node;
    }
    if (leaders == 0) { free(a); free(b); free(nn); return -1; }
    cur = out; next -= leaders; ++rounds;
    cluster_t* t = a; a = b; b = t;
  }
  free(a); free(b); free(nn);
  *rounds_out = rounds;
  if (next != 0) return -1;
  /* depth-first (pre-order) renumbering: subtrees become contiguous, the first child follows its parent */
  uint32_t nn_nodes = (uint32_t)(n - 1);
  bnode_t* src = (bnode_t*)malloc(sizeof(bnode_t) * nn_nodes);
  uint32_t* new_index = (uint32_t*)malloc(sizeof(uint32_t) * nn_nodes);
  int32_t* stack = (int32_t*)malloc(sizeof(int32_t) * (nn_nodes + 1));
  memcpy(src, s->nodes, sizeof(bnode_t) * nn_nodes);
  uint32_t counter = 0; int sp = 0; stack[sp++] = 0;
  while (sp) {
    int32_t x = stack[--sp];
    new_index[x] = counter++;
    if (src[x].link[1] >= 0) stack[sp++] = src[x].link[1];
    if (src[x].link[0] >= 0) stack[sp++] = src[x].link[0];
  }
  for (uint32_t x = 0; x < nn_nodes; ++x) {
    bnode_t nd = src[x];
    for (int c = 0; c < 2; ++c) if (nd.link[c] >= 0) nd.link[c] = (int32_t)new_index[nd.link[c]];
    if (nd.parent != UINT32_MAX) nd.parent = new_index[nd.parent];
    s->nodes[new_index[x]] = nd;
  }
  free(src); free(new_index); free(stack);
  return 0;
}

static void build_bvh(orc_scene* s) {
  int n = (int)s->d.n_triangles;
  const char* bsel = getenv("MI_PT_BVH"); /* same switch as the product: "lbvh" = Karras hierarchy */
  s->builder = (bsel && strcmp(bsel, "lbvh") == 0) ? 0 : 1;
  s->build_rounds = 0;
  for (int a = 0; a < 3; ++a) { s->scene_lo[a] = INFINITY; s->scene_hi[a] = -INFINITY; }
  for (int t = 0; t < n; ++t) {
    float lo[3], hi[3]; tri_bounds(s, t, lo, hi);
    for (int a = 0; a < 3; ++a) {
      if (lo[a] < s->scene_lo[a]) s->scene_lo[a] = lo[a];
      if (hi[a] > s->scene_hi[a]) s->scene_hi[a] = hi[a];
    }
  }
  mkey_t* keys = (mkey_t*)malloc(sizeof(mkey_t) * (size_t)n);
  for (int t = 0; t < n; ++t) {
    float lo[3], hi[3]; tri_bounds(s, t, lo, hi);
    uint32_t q[3];
    for (int a = 0; a < 3; ++a) q[a] = quant21((lo[a] + hi[a]) * 0.5f, s->scene_lo[a], s->scene_hi[a]);
    keys[t].code = (expand_bits21(q[0]) << 2) | (expand_bits21(q[1]) << 1) | expand_bits21(q[2]);
    keys[t].id = (uint32_t)t;
  }
  qsort(keys, (size_t)n, sizeof(mkey_t), cmp_key);
  s->sorted_tri = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n);
  s->morton = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)n);
  s->tris_sorted = (tri_t*)malloc(sizeof(tri_t) * (size_t)n);
  for (int i = 0; i < n; ++i) {
    s->sorted_tri[i] = keys[i].id;
    s->morton[i] = keys[i].code;
    s->tris_sorted[i] = s->tris[s->sorted_tri[i]];
  }
  s->n_nodes = n > 1 ? (uint32_t)(n - 1) : 0;
  s->nodes = (bnode_t*)calloc(s->n_nodes ? s->n_nodes : 1, sizeof(bnode_t));
  if (s->n_nodes && s->builder == 1) {
    if (build_ploc(s, n, &s->build_rounds) != 0) { fprintf(stderr, "oracle: PLOC build stalled\n"); abort(); }
  }
  for (int i = 0; s->builder == 0 && i < (int)s->n_nodes; ++i) {
    int d = (delta_fn(keys, n, i, i + 1) - delta_fn(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = delta_fn(keys, n, i, i - d);
    int lmax = 2;
    while (delta_fn(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
      if (delta_fn(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = delta_fn(keys, n, i, j);
    int sp = 0;
    int t = l;
    do {
      t = (t + 1) / 2;
      if (delta_fn(keys, n, i, i + (sp + t) * d) > dnode) sp += t;
    } while (t > 1);
    int gamma = i + sp * d + (d < 0 ? d : 0);
    int lo_i = i < j ? i : j, hi_i = i < j ? j : i;
    int32_t left = (lo_i == gamma) ? ~gamma : gamma;
    int32_t right = (hi_i == gamma + 1) ? ~(gamma + 1) : gamma + 1;
    s->nodes[i].link[0] = left; s->nodes[i].link[1] = right;
    if (left >= 0) s->nodes[left].parent = (uint32_t)i;
    if (right >= 0) s->nodes[right].parent = (uint32_t)i;
  }
  if (s->n_nodes) {
    s->nodes[0].parent = UINT32_MAX;
    if (s->builder == 0) refit(s, 0);
    s->max_depth = depth_of(s, 0);
    if (s->max_depth > ORC_STACK) { fprintf(stderr, "oracle: BVH depth %u exceeds the traversal stack\n", s->max_depth); abort(); }
  } else s->max_depth = 1;
  free(keys);
}

/* ------------------------------------------------------------------ ray / triangle (Embree 2 MT) */
typedef struct { float t, u, v; uint32_t id; const tri_t* tri; } hit_t;

/* closest == 1: rtcIntersect semantics, `h` carries the best hit so far (h->t = tfar).
 * returns 1 if this triangle becomes the new best hit. */
static inline int tri_test(const tri_t* tr, v3 org, v3 dir, uint32_t ray_mask, float tnear, hit_t* h,
                           int closest) {
  if (!(tr->mask & ray_mask)) return 0;
  v3 C = vsub(tr->v0, org);
  v3 R = vcross(C, dir);
  float den = vdot(tr->ng, dir);
  float absden = fabsf(den);
  float sgn = den < 0.0f ? -1.0f : 1.0f; /* Embree xors the sign bit of den */
  float U = vdot(R, tr->e2) * sgn;
  float Vv = vdot(R, tr->e1) * sgn;
  if (den == 0.0f) return 0;
  if (!(U >= 0.0f) || !(Vv >= 0.0f) || !(U + Vv <= absden)) return 0;
  float T = vdot(tr->ng, C) * sgn;
  if (!(absden * tnear < T)) return 0;
  float t = T / absden;
  if (closest) {
    if (t < h->t || (t == h->t && tr->id < h->id)) {
      h->t = t; h->u = U / absden; h->v = Vv / absden; h->id = tr->id; h->tri = tr;
      return 1;
    }
    return 0;
  }
  if (t <= h->t) { h->id = tr->id; return 1; } /* occluded: T <= |den| * tfar */
  return 0;
}

static inline int box_test(const float lo[3], const float hi[3], v3 org, v3 inv, float tmax, float* tnear_out) {
  float t0x = (lo[0] - org.x) * inv.x, t1x = (hi[0] - org.x) * inv.x;
  float t0y = (lo[1] - org.y) * inv.y, t1y = (hi[1] - org.y) * inv.y;
  float t0z = (lo[2] - org.z) * inv.z, t1z = (hi[2] - org.z) * inv.z;
  float tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), 0.0f));
  float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tmax));
  *tnear_out = tn;
  /* conservative like the device's slab test (pt_device.h box_test): a relative widening plus an absolute slack for the rounding of
   * (plane - org) * inv at large |org * inv| — a ray that runs along a box edge must not lose the box (r03: three edge-on shadow rays of
   * CornellBoxPhong were lost with the bare 4e-7 widening; brute force and the device found their hits).  Which boxes are opened never
   * changes a result: the hit is the (t, id) minimum over the triangles tested. */
  float sx = fabsf(org.x * inv.x), sy = fabsf(org.y * inv.y), sz = fabsf(org.z * inv.z);
  float slack = ((sx < INFINITY ? sx : 0.0f) + (sy < INFINITY ? sy : 0.0f) + (sz < INFINITY ? sz : 0.0f)) * 2.5e-7f;
  return tn <= tf * 1.000002f + slack;
}

static void traverse(const orc_scene* s, v3 org, v3 dir, uint32_t ray_mask, hit_t* h, int closest) {
  uint32_t n = s->d.n_triangles;
  if (!s->use_bvh || n < 2) {
    for (uint32_t i = 0; i < n; ++i)
      if (tri_test(&s->tris[i], org, dir, ray_mask, 0.0f, h, closest) && !closest) return;
    return;
  }
  /* a NaN direction (TransmissionBSDF past the critical angle, BSDF.cpp:467-504) can hit nothing: every dot product in
   * tri_test is NaN.  Leave before the slab test, whose NaN-ignoring min/max would open every box of the tree. */
  if (dir.x != dir.x || dir.y != dir.y || dir.z != dir.z) return;
  ORC_RAY_BEGIN(closest);
  v3 inv = V(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
  int32_t stack[ORC_STACK]; int sp = 0;
  int32_t node = 0;
  for (;;) {
    if (node >= 0) {
      const bnode_t* nd = &s->nodes[node];
      float tn0, tn1;
      ORC_COUNT_NODE(closest);
      int h0 = box_test(nd->lo[0], nd->hi[0], org, inv, h->t, &tn0);
      int h1 = box_test(nd->lo[1], nd->hi[1], org, inv, h->t, &tn1);
      if (h0 && h1) {
        int first = tn1 < tn0;
        stack[sp++] = nd->link[first ? 0 : 1];
        node = nd->link[first ? 1 : 0];
        continue;
      } else if (h0) { node = nd->link[0]; continue; }
      else if (h1) { node = nd->link[1]; continue; }
    } else {
      ORC_COUNT_TRI(closest);
      if (tri_test(&s->tris_sorted[~node], org, dir, ray_mask, 0.0f, h, closest) && !closest) return;
    }
    if (sp == 0) return;
    node = stack[--sp];
  }
}

/* Scene::querySurface (Scene.cpp:80-126) */
static surf_t query_surface(const orc_scene* s, v3 org, v3 dir, const hit_t* h) {
  surf_t p;
  if (h->id == UINT32_MAX) { memset(&p, 0, sizeof p); p.material_id = UINT32_MAX; return p; }
  const float w = 1.f - h->u - h->v;
  const uint32_t* idx = s->indices + 3 * (size_t)h->id;
  m3 t0 = ldm3(s->tangents + 9 * (size_t)idx[0]);
  m3 t1 = ldm3(s->tangents + 9 * (size_t)idx[1]);
  m3 t2 = ldm3(s->tangents + 9 * (size_t)idx[2]);
  p.position = vmadd(org, dir, h->t);
  for (int c = 0; c < 3; ++c)
    p.tangent.c[c] = vadd(vadd(vscale(t0.c[c], w), vscale(t1.c[c], h->u)), vscale(t2.c[c], h->v));
  p.tangent.c[1] = vnormalize(p.tangent.c[1]);
  p.tangent.c[0] = vsub(p.tangent.c[0], vscale(p.tangent.c[1], vdot(p.tangent.c[0], p.tangent.c[1])));
  p.tangent.c[0] = vnormalize(p.tangent.c[0]);
  p.tangent.c[2] = vsub(vsub(p.tangent.c[2], vscale(p.tangent.c[1], vdot(p.tangent.c[2], p.tangent.c[1]))),
                        vscale(p.tangent.c[0], vdot(p.tangent.c[2], p.tangent.c[0])));
  p.tangent.c[2] = vnormalize(p.tangent.c[2]);
  /* RayIsect::gnormal = normalize(-Ng), omega = normalize(-dir) (RayIsect.hpp:24-25) */
  v3 g = vnormalize(vneg(h->tri->ng));
  /* the reference normalises -dir first (RayIsect::omega); only the sign of the dot product is used */
  p.gnormal = vscale(g, vdot(vneg(dir), g) < 0.0f ? -1.0f : 1.0f);
  p.material_id = s->tri_material[h->id];
  return p;
}

typedef struct { uint64_t basic, shadow; } counters_t;

/* Scene::intersect (Scene.cpp:182-203) */
static surf_t scene_intersect(const orc_scene* s, const surf_t* from, v3 dir, counters_t* cnt, hit_t* out_hit) {
  v3 org = vadd(from->position,
                vscale(vscale(from->gnormal, vdot(from->gnormal, dir) > 0.0f ? 1.0f : -1.0f), 0.0001f));
  hit_t h; h.t = INFINITY; h.u = h.v = 0; h.id = UINT32_MAX; h.tri = NULL;
  traverse(s, org, dir, 0xFFFFFFFFu, &h, 1);
  cnt->basic++;
  if (out_hit) *out_hit = h;
  return query_surface(s, org, dir, &h);
}

/* Scene::occluded (Scene.cpp:151-180): 1 = visible */
static float scene_occluded(const orc_scene* s, const surf_t* origin, const surf_t* target, counters_t* cnt) {
  v3 direction = vsub(target->position, origin->position); /* Scene.cpp:153 normalises; only signs are used */
  v3 ao = vadd(origin->position,
               vscale(vscale(origin->gnormal, vdot(origin->gnormal, direction) > 0.0f ? 1.0f : -1.0f), 0.0001f));
  v3 at = vadd(target->position,
               vscale(vscale(target->gnormal, vdot(target->gnormal, direction) < 0.0f ? 1.0f : -1.0f), 0.0001f));
  hit_t h; h.t = 1.0f; h.id = UINT32_MAX; h.tri = NULL; h.u = h.v = 0;
  traverse(s, ao, vsub(at, ao), 1u << MI_ENTITY_MESH, &h, 0);
  cnt->shadow++;
  return h.id != UINT32_MAX ? 0.f : 1.f;
}

/* ------------------------------------------------------------------ BSDFs */
typedef struct { v3 throughput; float density, densityRev; int finite; } bq_t;
typedef struct { bq_t q; v3 omega; } bs_t;

static inline bq_t bq_zero(void) { bq_t q; q.throughput = V(0, 0, 0); q.density = 0; q.densityRev = 0; q.finite = 1; return q; }

/* DiffuseBSDF::_query (BSDF.cpp:291-304) — all vectors in the local frame */
static bq_t diffuse_query_local(const mi_material* m, v3 gn, v3 incident, v3 outgoing) {
  float same_side = vdot(incident, gn) * vdot(outgoing, gn) > 0.0f ? 1.0f : 0.0f;
  bq_t q;
  q.throughput = vscale(vscale(ld3(m->diffuse), ONE_OVER_PI), same_side);
  q.density = fabsf(outgoing.y * ONE_OVER_PI) * same_side;
  q.densityRev = fabsf(incident.y * ONE_OVER_PI) * same_side;
  q.finite = 1;
  return q;
}
static float phong_diffuse_probability(const mi_material* m) { /* PhongBSDF ctor, BSDF.cpp:306-315 */
  float dr = l1norm(ld3(m->diffuse)) * ONE_OVER_PI;
  float sr = l1norm(ld3(m->specular)) * 2.0f * PI_F / (m->power + 1.0f);
  return dr / (dr + sr);
}
/* PhongBSDF::_query (BSDF.cpp:354-391) */
static bq_t phong_query_local(const mi_material* m, v3 incident, v3 outgoing, float same_side) {
  float pd = phong_diffuse_probability(m), ps = 1.0f - pd;
  float dd = fabsf(outgoing.y * ONE_OVER_PI), ddr = fabsf(incident.y * ONE_OVER_PI);
  v3 diffuse = vscale(ld3(m->diffuse), ONE_OVER_PI);
  const float half_over_pi = 0.5f * ONE_OVER_PI;
  v3 reflected = V(-incident.x, incident.y, -incident.z);
  float ca = vdot(outgoing, reflected); ca = ca < 0.0f ? 0.0f : (ca > 1.0f ? 1.0f : ca);
  float cap = mi_powf(ca, m->power);
  float sd = (m->power + 1.0f) * half_over_pi * cap;
  v3 specular = vscale(vscale(vscale(ld3(m->specular), m->power + 2.0f), half_over_pi), cap);
  bq_t q;
  q.density = same_side * (sd * ps + dd * pd);
  q.densityRev = same_side * (sd * ps + ddr * pd);
  q.throughput = vscale(vadd(diffuse, specular), same_side);
  q.finite = 1;
  return q;
}
/* Scene::queryBSDF(surface, incident, outgoing) (Scene.cpp:142-149) */
static bq_t bsdf_query(const orc_scene* s, const surf_t* sf, v3 incident, v3 outgoing) {
  const mi_material* m = &s->materials[sf->material_id >> 2];
  switch (m->type) {
    case MI_BSDF_DIFFUSE: /* BSDF.cpp:239-243 */
      return diffuse_query_local(m, s_to_surface(sf, sf->gnormal), s_to_surface(sf, incident), s_to_surface(sf, outgoing));
    case MI_BSDF_PHONG: { /* BSDF.cpp:317-326 */
      float same_side = vdot(incident, sf->gnormal) * vdot(outgoing, sf->gnormal) > 0.0f ? 1.0f : 0.0f;
      return phong_query_local(m, s_to_surface(sf, incident), s_to_surface(sf, outgoing), same_side);
    }
    case MI_BSDF_LIGHT: { /* LightBSDF::query BSDF.cpp:95-114; PT reads only .throughput (PT.cpp:103-107) */
      bq_t q = bq_zero();
      v3 lo = s_to_surface(sf, outgoing);
      q.throughput = lo.y > 0.0f ? V(1, 1, 1) : V(0, 0, 0);
      return q;
    }
    case MI_BSDF_SUN: { bq_t q = bq_zero(); q.density = 1; q.densityRev = 1; return q; } /* BSDF.cpp:181-191 */
    case MI_BSDF_REFLECTION:
    case MI_BSDF_TRANSMISSION: { bq_t q = bq_zero(); q.finite = 0; return q; } /* DeltaBSDF::query BSDF.cpp:438-448 */
    default: { /* CameraBSDF::query BSDF.cpp:210-222 (never reached by PT) */
      bq_t q = bq_zero(); v3 li = s_to_surface(sf, incident);
      float v = (li.y > 0.0f ? 1.0f : 0.0f) / fabsf(li.y);
      q.throughput = V(v, v, v); q.densityRev = 1.0f; return q;
    }
  }
}
/* sample_lambert (Sample.inl:52-60) */
static v3 sample_lambert(rng_t* g, v3 omega) {
  float y = sqrtf(rng_f(g)) * gsign(omega.y);
  float r = sqrtf(1.0f - y * y);
  float sn, cs; sincos_2pi(rng_f(g), &sn, &cs); /* phi = u * 2 * pi */
  return V(r * cs, y, r * sn);
}
/* reflection_to_surface (Sample.inl:43-50) + sample_phong (Sample.inl:139-151) */
static v3 sample_phong(rng_t* g, v3 omega, float power) {
  m3 m;
  m.c[1] = V(-omega.x, omega.y, -omega.z);
  m.c[2] = vnormalize(vsub(V(0.0f, 1.0f, 0.0f), vscale(m.c[1], m.c[1].y)));
  m.c[0] = vnormalize(vcross(m.c[1], m.c[2]));
  float y = mi_powf(rng_f(g), 1.0f / (power + 1.0f));
  float r = sqrtf(1.0f - y * y);
  float sn, cs; sincos_2pi(rng_f(g), &sn, &cs);
  return m3mulv(m, V(r * cs, y, r * sn));
}
/* Scene::sampleBSDF (Scene.cpp:133-140) */
static bs_t bsdf_sample(const orc_scene* s, rng_t* g, const surf_t* sf, v3 omega) {
  const mi_material* m = &s->materials[sf->material_id >> 2];
  bs_t r; r.q = bq_zero(); r.omega = V(0, 0, 0);
  v3 lo = s_to_surface(sf, omega);
  switch (m->type) {
    case MI_BSDF_DIFFUSE: { /* BSDF.cpp:245-262 */
      v3 d = sample_lambert(g, lo);
      r.q = diffuse_query_local(m, s_to_surface(sf, sf->gnormal), lo, d);
      r.omega = s_to_world(sf, d);
      return r;
    }
    case MI_BSDF_PHONG: { /* BSDF.cpp:328-352 */
      v3 d = rng_f(g) < phong_diffuse_probability(m) ? sample_lambert(g, lo) : sample_phong(g, lo, m->power);
      r.omega = s_to_world(sf, d);
      float same_side = vdot(omega, sf->gnormal) * vdot(r.omega, sf->gnormal) > 0.0f ? 1.0f : 0.0f;
      r.q = phong_query_local(m, lo, d, same_side);
      return r;
    }
    case MI_BSDF_REFLECTION: { /* BSDF.cpp:450-465 */
      float v = 1.0f / lo.y;
      r.q.throughput = V(v, v, v);
      r.omega = s_to_world(sf, V(-lo.x, lo.y, -lo.z));
      r.q.density = 1.0f; r.q.densityRev = 1.0f; r.q.finite = 0;
      return r;
    }
    case MI_BSDF_TRANSMISSION: { /* BSDF.cpp:467-504 */
      float ext_over_int = m->ior_external / m->ior_internal;
      v3 o;
      if (lo.y > 0.f) {
        const float eta = ext_over_int;
        float yy = sqrtf(1 - eta * eta * (1 - lo.y * lo.y));
        o = vsub(vscale(vsub(lo, V(0.0f, lo.y, 0.0f)), -eta), V(0.0f, yy, 0.0f));
      } else {
        const float eta = 1.0f / ext_over_int;
        float yy = sqrtf(1 - eta * eta * (1 - lo.y * lo.y));
        o = vadd(vscale(vsub(lo, V(0.0f, lo.y, 0.0f)), -eta), V(0.0f, yy, 0.0f));
      }
      float v = 1.0f / fabsf(o.y);
      r.q.throughput = V(v, v, v);
      r.omega = s_to_world(sf, o);
      r.q.density = 1.0f; r.q.densityRev = 1.0f; r.q.finite = 0;
      return r;
    }
    default: /* lights are passed through, cameras are never hit: PT never samples these */
      r.omega = vneg(omega); r.q.throughput = V(0, 0, 0); r.q.density = 1.0f; return r;
  }
}

/* ------------------------------------------------------------------ lights */
typedef struct { surf_t surface; v3 radiance; float area_density, light_density; } lsample_t;

static inline float light_area(const mi_light* l) { return l->size[0] * l->size[1]; }
static inline v3 light_radiance(const mi_light* l) { return vscale(ld3(l->exitance), ONE_OVER_PI); }

/* AreaLights::sample (AreaLights.cpp:121-140), _sampleLight (:216-221), _samplePosition (:223-231) */
static lsample_t light_sample(const orc_scene* s, rng_t* g) {
  uint32_t n = s->d.n_lights;
  float u = rng_f(g);
  uint32_t id = n - 1;
  for (uint32_t i = 0; i < n; ++i) if (u < s->light_cdf[i + 1]) { id = i; break; }
  const mi_light* l = &s->lights[id];
  float sx = rng_f(g), sy = rng_f(g);
  float ux = (sx - 0.5f) * l->size[0], uy = (sy - 0.5f) * l->size[1];
  m3 T = ldm3(l->tangent);
  lsample_t r;
  r.surface.position = vadd(vadd(ld3(l->position), vscale(T.c[0], ux)), vscale(T.c[2], uy));
  r.surface.tangent = T;
  r.surface.gnormal = T.c[1];
  r.surface.material_id = l->material_id;
  r.radiance = light_radiance(l);
  r.area_density = 1.0f / light_area(l);
  r.light_density = s->light_weight[id];
  return r;
}
/* Scene::queryLSDF (Scene.cpp:128-131) -> AreaLights::queryLSDF (AreaLights.cpp:142-155) */
static void query_lsdf(const orc_scene* s, const surf_t* sf, v3 omega, v3* radiance, float* density) {
  uint32_t lid = s->materials[sf->material_id >> 2].light_id;
  const mi_light* l = &s->lights[lid];
  float c = vdot(omega, ld3(l->tangent + 3));
  *radiance = vscale(light_radiance(l), c > 0.0f ? 1.0f : 0.0f);
  *density = s->light_weight[lid] / light_area(l);
}

/* Edge (SurfacePoint.hpp:65-83) */
typedef struct { float distSqInv, fCos, bCos, fG, bG; } edge_t;
static edge_t make_edge(const surf_t* fst, const surf_t* snd, v3 omega) {
  edge_t e; v3 d = vsub(fst->position, snd->position);
  e.distSqInv = 1.0f / vdot(d, d);
  e.fCos = fabsf(vdot(omega, s_normal(snd)));
  e.bCos = fabsf(vdot(omega, s_normal(fst)));
  e.fG = e.distSqInv * e.fCos;
  e.bG = e.distSqInv * e.bCos;
  return e;
}
/* pow(x, beta) of the MIS weights (PT.cpp:72-74,113-115).  pow(x,1) == x and pow(x,2) == x*x
 * hold exactly for a correctly rounded pow; spelled out so CPU and device agree. */
static inline float powb(float x, float beta) { return beta == 1.0f ? x : (beta == 2.0f ? x * x : mi_powf(x, beta)); }

/* ------------------------------------------------------------------ PT */
typedef struct { surf_t surface; v3 omega, throughput; float density; int finite; } eye_t;

/* PathTracing::_connect (PT.cpp:100-120) */
static v3 pt_connect(const orc_scene* s, rng_t* g, const eye_t* eye, counters_t* cnt) {
  lsample_t light = light_sample(s, g);
  v3 omega = vnormalize(vsub(eye->surface.position, light.surface.position));
  bq_t lb = bsdf_query(s, &light.surface, s_normal(&light.surface), omega);
  if (l1norm(lb.throughput) < FLT_EPSILON) return V(0, 0, 0);
  bq_t eb = bsdf_query(s, &eye->surface, vneg(omega), eye->omega);
  edge_t e = make_edge(&light.surface, &eye->surface, omega);
  float cd = light.area_density * light.light_density;
  float wInv = powb(eb.densityRev * e.bG, s->p.beta) / powb(cd, s->p.beta) + 1.0f;
  float occ = scene_occluded(s, &eye->surface, &light.surface, cnt);
  /* PT.cpp:117-119 multiplies the visibility in first; here it is applied last (the same value
   * unless a factor is inf/NaN or the product overflows) so the device can resolve the shadow ray later */
  v3 r = vdivs(light.radiance, cd);
  r = vmul(r, eye->throughput);
  r = vmul(r, eb.throughput);
  r = vscale(r, e.bCos);
  r = vscale(r, e.fG);
  return vscale(vdivs(r, wInv), occ);
}

/* PathTracing::_traceEye (PT.cpp:15-98) */
static v3 pt_trace_eye(const orc_scene* s, rng_t* g, const surf_t* camera_surface, v3 dir, counters_t* cnt) {
  const uint64_t max_path = s->p.max_path;
  v3 radiance = V(0, 0, 0);
  eye_t eye[2]; int itr = 0, prv = 1;
  surf_t surface = scene_intersect(s, camera_surface, dir, cnt, NULL);
  while (s_is_light(&surface) && max_path > 0) {
    v3 le; float dens; query_lsdf(s, &surface, vneg(dir), &le, &dens);
    radiance = vadd(radiance, vscale(le, s->p.lights));
    surface = scene_intersect(s, &surface, dir, cnt, NULL);
  }
  if (!s_is_present(&surface) || max_path < 2) return radiance;
  eye[prv].surface = surface; eye[prv].omega = vneg(dir); eye[prv].throughput = V(1, 1, 1);
  eye[prv].finite = 1; eye[prv].density = 1.0f;
  uint64_t path_size = 2;
  while (path_size <= max_path) {
    radiance = vadd(radiance, pt_connect(s, g, &eye[prv], cnt));
    bs_t b = bsdf_sample(s, g, &eye[prv].surface, eye[prv].omega);
    for (;;) {
      surface = scene_intersect(s, &surface, b.omega, cnt, NULL);
      if (!s_is_present(&surface)) return radiance;
      eye[itr].surface = surface; eye[itr].omega = vneg(b.omega);
      edge_t e = make_edge(&eye[prv].surface, &eye[itr].surface, eye[itr].omega);
      eye[itr].throughput = vscale(vmul(eye[prv].throughput, b.q.throughput), e.bCos);
      if (l1norm(eye[itr].throughput) < FLT_EPSILON) return radiance;
      eye[itr].throughput = vdivs(eye[itr].throughput, b.q.density);
      eye[prv].finite = b.q.finite;
      eye[itr].density = eye[prv].density * e.fG * b.q.density;
      if (s_is_light(&surface)) {
        v3 le; float dens; query_lsdf(s, &eye[itr].surface, eye[itr].omega, &le, &dens);
        float wInv = powb(dens, s->p.beta) / powb(e.fG * b.q.density, s->p.beta) + 1.0f;
        if (b.q.finite == 0) wInv = 1.0f;
        radiance = vadd(radiance, vdivs(vmul(le, eye[itr].throughput), wInv));
      } else break;
    }
    { int t = itr; itr = prv; prv = t; }
    float roulette = path_size < s->p.min_subpath ? 1.0f : s->p.roulette;
    float uniform = rng_f(g);
    if (roulette < uniform) return radiance;
    eye[prv].throughput = vdivs(eye[prv].throughput, roulette);
    ++path_size;
  }
  return radiance;
}

/* ------------------------------------------------------------------ camera (Cameras.cpp) */
static void camera_setup(const mi_camera* c, float aspect, mi_camera_frame* out) {
  /* glm::lookAt(eye, eye + dir, up), RH: f = normalize(center-eye); s = normalize(cross(f,up)); u = cross(s,f) */
  v3 f = vnormalize(vsub(vadd(ld3(c->position), ld3(c->direction)), ld3(c->position)));
  v3 sv = vnormalize(vcross(f, ld3(c->up)));
  v3 u = vcross(sv, f);
  m3 view3; /* upper-left 3x3 of the lookAt matrix, column-major */
  view3.c[0] = V(sv.x, u.x, -f.x); view3.c[1] = V(sv.y, u.y, -f.y); view3.c[2] = V(sv.z, u.z, -f.z);
  m3 w2v = m3transpose(m3inverse(view3));   /* Cameras.cpp:108-110 */
  m3 v2w = m3inverse(w2v);                  /* Cameras.cpp:104-106 */
  stm3(out->world_to_view, w2v); stm3(out->view_to_world, v2w);
  st3(out->position, ld3(c->position));
  float focal = 1.0f / tanf(c->fovx * 0.5f);                 /* Cameras.cpp:23-25 */
  out->fovy = 2.0f * atan2f(1.0f / aspect, focal);           /* Cameras.cpp:85 */
  out->focal_length_y = 1.0f / tanf(out->fovy * 0.5f);       /* Cameras.cpp:116 */
}
/* ray_direction (Cameras.cpp:120-127) */
static v3 ray_direction(float px, float py, float rx, float ry_inv, float fl) {
  float x = px * ry_inv * 2.0f - rx * ry_inv;
  float y = py * ry_inv * 2.0f - 1.0f;
  return vnormalize(V(x, y, -fl));
}
/* Technique::_camera_surface (Technique.cpp:107-116) */
static surf_t camera_surface(const mi_camera_frame* cf) {
  surf_t r; m3 v2w = ldm3(cf->view_to_world);
  r.position = ld3(cf->position);
  r.tangent.c[0] = v2w.c[1]; r.tangent.c[1] = vneg(v2w.c[2]); r.tangent.c[2] = v2w.c[0];
  r.material_id = (0u << 2) | MI_ENTITY_CAMERA;
  r.gnormal = vneg(v2w.c[2]);
  return r;
}

typedef struct { mi_camera_frame cf; surf_t cs; float rx, ry, ry_inv; } cam_ctx_t;
static void cam_ctx(const orc_scene* s, uint32_t camera_id, uint32_t w, uint32_t h, cam_ctx_t* c) {
  camera_setup(&s->cameras[camera_id], (float)w / (float)h, &c->cf); /* Technique.cpp:37-45 */
  c->cs = camera_surface(&c->cf);
  c->rx = (float)w; c->ry = (float)h; c->ry_inv = 1.0f / c->ry;
}
/* shoot() + _traceEye for one (pixel, sample) (Technique.cpp:321-338) */
static v3 trace_one(const orc_scene* s, const cam_ctx_t* c, uint32_t x, uint32_t y, uint32_t width,
                    uint64_t sample, uint64_t seed, counters_t* cnt) {
  rng_t g = rng_seed(seed, y * width + x, sample);
  float u0 = rng_f(&g), u1 = rng_f(&g);
  v3 d = ray_direction((float)x + u0, (float)y + u1, c->rx, c->ry_inv, c->cf.focal_length_y);
  v3 wd = m3mulv(ldm3(c->cf.view_to_world), d);
  return pt_trace_eye(s, &g, &c->cs, wd, cnt);
}

#include "bpt_oracle.inc" /* BPT restatement (next row of SURVEY 8(f)); shares every function above */

/* ------------------------------------------------------------------ exported API */
ORC_API orc_scene* orc_create(const mi_scene_desc* d, const mi_pt_params* p, int use_bvh) {
  orc_scene* s = (orc_scene*)calloc(1, sizeof *s);
  s->d = *d; s->p = *p; s->use_bvh = use_bvh;
#define DUP(field, type, count) do { size_t nb = sizeof(type) * (size_t)(count); s->field = (type*)malloc(nb ? nb : 1); memcpy(s->field, d->field, nb); } while (0)
  DUP(positions, float, 3 * (size_t)d->n_vertices); DUP(tangents, float, 9 * (size_t)d->n_vertices);
  DUP(indices, uint32_t, 3 * (size_t)d->n_triangles); DUP(mesh_tri_offset, uint32_t, d->n_meshes + 1);
  DUP(mesh_material_id, uint32_t, d->n_meshes); DUP(materials, mi_material, d->n_materials);
  DUP(lights, mi_light, d->n_lights); DUP(cameras, mi_camera, d->n_cameras);
#undef DUP
  s->tri_material = (uint32_t*)malloc(sizeof(uint32_t) * (d->n_triangles ? d->n_triangles : 1));
  s->tris = (tri_t*)malloc(sizeof(tri_t) * (d->n_triangles ? d->n_triangles : 1));
  for (uint32_t m = 0; m < d->n_meshes; ++m)
    for (uint32_t t = s->mesh_tri_offset[m]; t < s->mesh_tri_offset[m + 1]; ++t) s->tri_material[t] = s->mesh_material_id[m];
  for (uint32_t t = 0; t < d->n_triangles; ++t) {
    v3 v0 = ld3(s->positions + 3 * (size_t)s->indices[3 * t]);
    v3 v1 = ld3(s->positions + 3 * (size_t)s->indices[3 * t + 1]);
    v3 v2 = ld3(s->positions + 3 * (size_t)s->indices[3 * t + 2]);
    tri_t* tr = &s->tris[t];
    tr->v0 = v0; tr->e1 = vsub(v0, v1); tr->e2 = vsub(v2, v0); tr->ng = vcross(tr->e2, tr->e1);
    tr->id = t; tr->mask = 1u << (s->tri_material[t] & 3u);
  }
  /* AreaLights::_updateSampler (AreaLights.cpp:199-214): weight = power / totalPower */
  s->light_weight = (float*)calloc(d->n_lights + 1, sizeof(float));
  s->light_cdf = (float*)calloc(d->n_lights + 2, sizeof(float));
  float total = 0.0f;
  for (uint32_t i = 0; i < d->n_lights; ++i) total += light_area(&s->lights[i]) * l1norm(ld3(s->lights[i].exitance));
  float total_inv = 1.0f / total;
  for (uint32_t i = 0; i < d->n_lights; ++i) {
    s->light_weight[i] = light_area(&s->lights[i]) * l1norm(ld3(s->lights[i].exitance)) * total_inv;
    s->light_cdf[i + 1] = s->light_cdf[i] + s->light_weight[i];
  }
  if (!(s->d.bounding_sphere[3] > 0.0f)) { /* compute_bounding_sphere (loader.cpp:408-432): surface meshes only, before the light quads */
    double c[3] = {0, 0, 0}; size_t nv = 0;
    for (uint32_t m = 0; m < d->n_meshes; ++m) {
      if ((s->mesh_material_id[m] & 3u) != MI_ENTITY_MESH) continue;
      for (uint32_t t = s->mesh_tri_offset[m]; t < s->mesh_tri_offset[m + 1]; ++t)
        for (int k = 0; k < 3; ++k) { const float* p = s->positions + 3 * (size_t)s->indices[3 * t + k]; c[0] += p[0]; c[1] += p[1]; c[2] += p[2]; ++nv; }
    }
    if (nv) {
      float cx = (float)(c[0] / (double)nv), cy = (float)(c[1] / (double)nv), cz = (float)(c[2] / (double)nv), r2 = 0.0f;
      for (uint32_t m = 0; m < d->n_meshes; ++m) {
        if ((s->mesh_material_id[m] & 3u) != MI_ENTITY_MESH) continue;
        for (uint32_t t = s->mesh_tri_offset[m]; t < s->mesh_tri_offset[m + 1]; ++t)
          for (int k = 0; k < 3; ++k) { const float* p = s->positions + 3 * (size_t)s->indices[3 * t + k];
            float dx = p[0] - cx, dy = p[1] - cy, dz = p[2] - cz, q = dx * dx + dy * dy + dz * dz; if (q > r2) r2 = q; }
      }
      s->d.bounding_sphere[0] = cx; s->d.bounding_sphere[1] = cy; s->d.bounding_sphere[2] = cz; s->d.bounding_sphere[3] = sqrtf(r2);
    }
  }
  build_bvh(s);
  return s;
}
ORC_API void orc_destroy(orc_scene* s) {
  if (!s) return;
  free(s->positions); free(s->tangents); free(s->indices); free(s->mesh_tri_offset); free(s->mesh_material_id);
  free(s->materials); free(s->lights); free(s->cameras); free(s->tri_material); free(s->tris);
  free(s->light_weight); free(s->light_cdf); free(s->nodes); free(s->sorted_tri); free(s->morton); free(s->tris_sorted);
  free(s);
}
ORC_API void orc_set_use_bvh(orc_scene* s, int use_bvh) { s->use_bvh = use_bvh; }

ORC_API void orc_bvh_info(const orc_scene* s, mi_bvh_info* out) {
  memset(out, 0, sizeof *out);
  out->n_triangles = s->d.n_triangles; out->n_nodes = s->n_nodes; out->max_depth = s->max_depth;
  memcpy(out->scene_lo, s->scene_lo, 12); memcpy(out->scene_hi, s->scene_hi, 12);
  out->builder = s->builder; out->build_rounds = s->build_rounds;
}
ORC_API void orc_bvh_download(const orc_scene* s, mi_bvh_node* nodes, uint32_t* sorted_tri, uint64_t* morton) {
  if (nodes) for (uint32_t i = 0; i < s->n_nodes; ++i) {
    const bnode_t* n = &s->nodes[i]; mi_bvh_node* o = &nodes[i];
    memcpy(o->lo0, n->lo[0], 12); memcpy(o->hi0, n->hi[0], 12); memcpy(o->lo1, n->lo[1], 12); memcpy(o->hi1, n->hi[1], 12);
    o->link0 = n->link[0]; o->link1 = n->link[1]; o->parent = n->parent; o->reserved = 0;
  }
  if (sorted_tri) memcpy(sorted_tri, s->sorted_tri, sizeof(uint32_t) * s->d.n_triangles);
  if (morton) memcpy(morton, s->morton, sizeof(uint64_t) * s->d.n_triangles);
}

static surf_t surf_from_abi(const mi_surface_point* p) {
  surf_t s; s.position = ld3(p->position); s.gnormal = ld3(p->gnormal); s.tangent = ldm3(p->tangent); s.material_id = p->material_id; return s;
}
static void surf_to_abi(const surf_t* s, mi_surface_point* p) {
  st3(p->position, s->position); st3(p->gnormal, s->gnormal); stm3(p->tangent, s->tangent); p->material_id = s->material_id;
}
ORC_API void orc_intersect(const orc_scene* s, uint32_t n, const mi_surface_point* origins, const float* dirs,
                           mi_surface_point* out_hits, float* out_t, uint32_t* out_prim) {
  counters_t c = {0, 0};
  for (uint32_t i = 0; i < n; ++i) {
    surf_t o = surf_from_abi(&origins[i]); hit_t h;
    surf_t r = scene_intersect(s, &o, ld3(dirs + 3 * (size_t)i), &c, &h);
    if (out_hits) surf_to_abi(&r, &out_hits[i]);
    if (out_t) out_t[i] = h.t;
    if (out_prim) out_prim[i] = h.id;
  }
}
ORC_API void orc_occluded(const orc_scene* s, uint32_t n, const mi_surface_point* origins,
                          const mi_surface_point* targets, float* out) {
  counters_t c = {0, 0};
  for (uint32_t i = 0; i < n; ++i) {
    surf_t a = surf_from_abi(&origins[i]), b = surf_from_abi(&targets[i]);
    out[i] = scene_occluded(s, &a, &b, &c);
  }
}
ORC_API void orc_trace_paths(const orc_scene* s, uint32_t camera_id, uint32_t width, uint32_t height, uint32_t n,
                             const uint32_t* pixel_xy, const uint64_t* sample_index, uint64_t seed,
                             float* out_radiance, uint32_t* out_ray_counts) {
  cam_ctx_t c; cam_ctx(s, camera_id, width, height, &c);
  for (uint32_t i = 0; i < n; ++i) {
    counters_t cnt = {0, 0};
    v3 r = trace_one(s, &c, pixel_xy[2 * i], pixel_xy[2 * i + 1], width, sample_index[i], seed, &cnt);
    st3(out_radiance + 3 * (size_t)i, r);
    if (out_ray_counts) { out_ray_counts[2 * i] = (uint32_t)cnt.basic; out_ray_counts[2 * i + 1] = (uint32_t)cnt.shadow; }
  }
}

/* Technique::render for spp frames: 32x32 tiles over a thread pool (Technique.cpp:163-192),
 * FP64 accumulation (Technique.cpp:338), finite filter (Technique.cpp:222-230). */
typedef struct {
  const orc_scene* s; cam_ctx_t cam; uint32_t width, height; mi_window win; uint32_t spp; uint64_t seed, sample_offset;
  float* rgbn; uint32_t tiles_x, tiles_y; volatile uint32_t next_tile; pthread_mutex_t mu; mi_pt_stats stats;
} job_t;
static void* worker(void* arg) {
  job_t* j = (job_t*)arg; counters_t cnt = {0, 0}; uint64_t errors = 0, paths = 0;
  for (;;) {
    uint32_t t = __sync_fetch_and_add(&j->next_tile, 1);
    if (t >= j->tiles_x * j->tiles_y) break;
    uint32_t x0 = j->win.x0 + (t % j->tiles_x) * 32, y0 = j->win.y0 + (t / j->tiles_x) * 32;
    uint32_t x1 = x0 + 32 < j->win.x0 + j->win.w ? x0 + 32 : j->win.x0 + j->win.w;
    uint32_t y1 = y0 + 32 < j->win.y0 + j->win.h ? y0 + 32 : j->win.y0 + j->win.h;
    for (uint32_t y = y0; y < y1; ++y) for (uint32_t x = x0; x < x1; ++x) {
      double acc[3] = {0, 0, 0}; uint32_t denom = 0;
      for (uint32_t k = 0; k < j->spp; ++k) {
        v3 r = trace_one(j->s, &j->cam, x, y, j->width, j->sample_offset + k, j->seed, &cnt);
        ++paths;
        if (isfinite(l1norm(r))) { acc[0] += r.x; acc[1] += r.y; acc[2] += r.z; ++denom; } else ++errors;
      }
      float* o = j->rgbn + 4 * ((size_t)y * j->width + x);
      o[0] = (float)acc[0]; o[1] = (float)acc[1]; o[2] = (float)acc[2]; o[3] = (float)denom;
    }
  }
  pthread_mutex_lock(&j->mu);
  j->stats.num_basic_rays += cnt.basic; j->stats.num_shadow_rays += cnt.shadow;
  j->stats.numeric_errors += errors; j->stats.num_paths += paths;
  pthread_mutex_unlock(&j->mu);
  return NULL;
}
ORC_API int orc_render(const orc_scene* s, uint32_t camera_id, uint32_t width, uint32_t height, mi_window win,
                       uint32_t spp, uint64_t seed, uint64_t sample_offset, float* rgbn, mi_pt_stats* stats,
                       int num_threads) {
  if (camera_id >= s->d.n_cameras || !rgbn || width == 0 || height == 0) return MI_ERR_INVALID_ARGUMENT;
  if (win.w == 0 || win.h == 0) { win.x0 = 0; win.y0 = 0; win.w = width; win.h = height; }
  if (win.x0 + win.w > width || win.y0 + win.h > height) return MI_ERR_INVALID_ARGUMENT;
  job_t j; memset(&j, 0, sizeof j);
  j.s = s; cam_ctx(s, camera_id, width, height, &j.cam); j.width = width; j.height = height; j.win = win; j.spp = spp;
  j.seed = seed; j.sample_offset = sample_offset; j.rgbn = rgbn;
  j.tiles_x = (win.w + 31) / 32; j.tiles_y = (win.h + 31) / 32;
  pthread_mutex_init(&j.mu, NULL);
  memset(rgbn, 0, sizeof(float) * 4 * (size_t)width * height);
  if (num_threads < 1) num_threads = 1;
  if (num_threads > 256) num_threads = 256;
  pthread_t th[256];
  for (int i = 1; i < num_threads; ++i) pthread_create(&th[i], NULL, worker, &j);
  worker(&j);
  for (int i = 1; i < num_threads; ++i) pthread_join(th[i], NULL);
  pthread_mutex_destroy(&j.mu);
  if (stats) *stats = j.stats;
  return MI_OK;
}

ORC_API void orc_camera_setup(const mi_camera* cam, float aspect, mi_camera_frame* out) { camera_setup(cam, aspect, out); }
ORC_API void orc_ray_direction(float px, float py, float rx, float ry, float fl, float out[3]) {
  st3(out, ray_direction(px, py, rx, 1.0f / ry, fl));
}
/* pixel_position (Cameras.cpp:134-144) */
ORC_API void orc_pixel_position(const float dir[3], float rx, float ry, float fl, float out[2]) {
  float ry_inv = 1.0f / ry;
  float factor = fl / -dir[2];
  float x = dir[0] * factor, y = dir[1] * factor;
  y = (y + 1.0f) * ry * 0.5f;
  x = (x + rx * ry_inv) * ry * 0.5f;
  out[0] = x; out[1] = y;
}
/* individual building blocks, exported for function-level tests */
ORC_API void orc_rng_floats(uint64_t seed, uint32_t pixel, uint64_t sample, uint32_t n, float* out) {
  rng_t g = rng_seed(seed, pixel, sample);
  for (uint32_t i = 0; i < n; ++i) out[i] = rng_f(&g);
}
ORC_API void orc_bsdf_query(const orc_scene* s, const mi_surface_point* sp, const float inc[3], const float outg[3],
                            float throughput[3], float* density, float* density_rev, int* finite) {
  surf_t sf = surf_from_abi(sp); bq_t q = bsdf_query(s, &sf, ld3(inc), ld3(outg));
  st3(throughput, q.throughput); *density = q.density; *density_rev = q.densityRev; *finite = q.finite;
}
ORC_API void orc_bsdf_sample(const orc_scene* s, const mi_surface_point* sp, const float omega[3], uint64_t seed,
                             uint32_t pixel, uint64_t sample, float out_omega[3], float throughput[3],
                             float* density, float* density_rev, int* finite) {
  surf_t sf = surf_from_abi(sp); rng_t g = rng_seed(seed, pixel, sample);
  bs_t b = bsdf_sample(s, &g, &sf, ld3(omega));
  st3(out_omega, b.omega); st3(throughput, b.q.throughput); *density = b.q.density; *density_rev = b.q.densityRev; *finite = b.q.finite;
}
ORC_API void orc_light_sample(const orc_scene* s, uint64_t seed, uint32_t pixel, uint64_t sample,
                              mi_surface_point* out_surface, float radiance[3], float* area_density, float* light_density) {
  rng_t g = rng_seed(seed, pixel, sample); lsample_t l = light_sample(s, &g);
  surf_to_abi(&l.surface, out_surface); st3(radiance, l.radiance); *area_density = l.area_density; *light_density = l.light_density;
}
ORC_API void orc_light_table(const orc_scene* s, float* weight, float* cdf) {
  memcpy(weight, s->light_weight, sizeof(float) * s->d.n_lights);
  memcpy(cdf, s->light_cdf, sizeof(float) * (s->d.n_lights + 1));
}
/* rms_abs_errors (ImageView.cpp:60-85) */
ORC_API void orc_rms_abs_errors(const float* rgbn, const float* ref, uint32_t w, uint32_t h, float* rms, float* abs_err) {
  float r = 0, a = 0;
  for (size_t i = 0; i < (size_t)w * h; ++i) {
    double ww = rgbn[4 * i + 3];
    v3 d = V(fabsf((float)(rgbn[4 * i] / ww) - ref[3 * i]), fabsf((float)(rgbn[4 * i + 1] / ww) - ref[3 * i + 1]),
             fabsf((float)(rgbn[4 * i + 2] / ww) - ref[3 * i + 2]));
    a += d.x + d.y + d.z; r += vdot(d, d);
  }
  float n = (float)((size_t)w * h * 3);
  *rms = sqrtf(r / n); *abs_err = a / n;
}
