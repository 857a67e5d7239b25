// vecmath.h — float3 arithmetic of the device path (gfx950).
//
// The whole device side is compiled with -ffp-contract=off; fusion happens ONLY where this
// header spells fmaf().  DESIGN.md "Arithmetic contract" lists these definitions; the CPU
// oracle states the same ones independently, so results can be compared bit for bit:
//   dot(a,b)      = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
//   cross(a,b).x  = fma(a.y,b.z, -(b.y*a.z))   (cyclic)
//   M*v  (r)      = fma(M.c2[r],v.z, fma(M.c1[r],v.y, M.c0[r]*v.x))
//   v*M  (i)      = dot(M.c[i], v)
//   normalize(a)  = a * (1 / sqrt(dot(a,a)))
//   madd(a,b,s)   = fma(b,s,a) per component
//   v / s         = v * (1 / s)   (one correctly rounded reciprocal, three multiplies)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi {

struct f3 { float x, y, z; };
struct m33 { f3 c0, c1, c2; };  // column-major like glm::mat3

#define MI_DEV __device__ __forceinline__

// The four operations the arithmetic contract spells as IEEE: reciprocal, quotient, square root, reciprocal square root.  The product computes
// them correctly rounded (what the CPU oracle and the reference's glm / libm compute): ~10, ~10, ~16 and ~26 instructions on gfx950.  MI_PT_FAST
// (a second build of pt_kernels.hip, opt-in at run time through MI_PT_FAST=1, never the default and never what the parity tests run) replaces them
// by the 1-ulp hardware approximations v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 and sin / cos(2 pi u) by v_sin_f32 / v_cos_f32 (input in turns): the
// price of bit-exactness, measured (VERDICT r03 #2 iii; bench.py `value_fast`).
#ifdef MI_PT_FAST
MI_DEV float mi_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
MI_DEV float mi_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
MI_DEV float mi_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
MI_DEV float mi_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
#else
MI_DEV float mi_rcp(float x) { return 1.0f / x; }
MI_DEV float mi_div(float a, float b) { return a / b; }
MI_DEV float mi_sqrt(float x) { return sqrtf(x); }
MI_DEV float mi_rsqrt(float x) { return 1.0f / sqrtf(x); }
#endif

MI_DEV f3 F3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
MI_DEV f3 operator+(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
MI_DEV f3 operator-(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
MI_DEV f3 operator*(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
MI_DEV f3 operator*(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
MI_DEV f3 operator/(f3 a, float s) { const float r = mi_rcp(s); return F3(a.x * r, a.y * r, a.z * r); }  // contract: v / s = v * (1 / s)
MI_DEV f3 operator-(f3 a) { return F3(-a.x, -a.y, -a.z); }
MI_DEV float dot(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
MI_DEV f3 cross(f3 a, f3 b) {
  return F3(fmaf(a.y, b.z, -(b.y * a.z)), fmaf(a.z, b.x, -(b.z * a.x)), fmaf(a.x, b.y, -(b.x * a.y)));
}
MI_DEV f3 normalize(f3 a) { return a * mi_rsqrt(dot(a, a)); }
MI_DEV f3 madd(f3 a, f3 b, float s) { return F3(fmaf(b.x, s, a.x), fmaf(b.y, s, a.y), fmaf(b.z, s, a.z)); }
MI_DEV float l1norm(f3 a) { return fabsf(a.x) + fabsf(a.y) + fabsf(a.z); }
MI_DEV float gsign(float x) { return float((0.0f < x) - (x < 0.0f)); }
MI_DEV f3 mulmv(const m33& m, f3 v) {  // glm mat3 * vec3
  return F3(fmaf(m.c2.x, v.z, fmaf(m.c1.x, v.y, m.c0.x * v.x)), fmaf(m.c2.y, v.z, fmaf(m.c1.y, v.y, m.c0.y * v.x)),
            fmaf(m.c2.z, v.z, fmaf(m.c1.z, v.y, m.c0.z * v.x)));
}
MI_DEV f3 mulvm(f3 v, const m33& m) { return F3(dot(m.c0, v), dot(m.c1, v), dot(m.c2, v)); }  // glm vec3 * mat3

#define MI_ONE_OVER_PI 0.318309886183790671537767526745028724f
#define MI_PI 3.14159265358979323846264338327950288f
#define MI_FLT_EPSILON 1.1920928955078125e-7f

// sin / cos of phi = (u * 2) * pi for u in [0,1) (sample_lambert / sample_phong, Sample.inl:55,146).
// Own definition (libm and the device math library differ in the last bits): quadrant
// reduction on u, which is exact in binary floating point, then odd/even polynomials on
// |theta| <= pi/4.  Max error ~1.5e-7 absolute.  The oracle states the same formula.
MI_DEV void sincos_2pi(float u, float* s, float* c) {
#ifdef MI_PT_FAST
  *s = __builtin_amdgcn_sinf(u); *c = __builtin_amdgcn_cosf(u);  // v_sin_f32 / v_cos_f32 take their argument in turns
  return;
#endif
  float k = floorf(fmaf(u, 4.0f, 0.5f));  // nearest quadrant 0..4
  float r = fmaf(k, -0.25f, u);           // exact: u - k/4 in [-1/8, 1/8]
  float t = r * 6.28318530717958647692f;  // theta
  float t2 = t * t;
  // sin(t) ~ t + t^3 * P(t^2), cos(t) ~ 1 + t^2 * Q(t^2)   (Cephes sinf/cosf kernels)
  float ps = fmaf(fmaf(-1.9515295891e-4f, t2, 8.3321608736e-3f), t2, -1.6666654611e-1f);
  float sn = fmaf(t * t2, ps, t);
  float pc = fmaf(fmaf(2.443315711809948e-5f, t2, -1.388731625493765e-3f), t2, 4.166664568298827e-2f);
  float cs = fmaf(t2 * t2, pc, fmaf(t2, -0.5f, 1.0f));
  int q = int(k) & 3;
  float so = (q & 1) ? cs : sn, co = (q & 1) ? sn : cs;
  *s = (q == 2 || q == 3) ? -so : so;
  *c = (q == 1 || q == 2) ? -co : co;
}


// asin, atan2 and sin/cos of an angle in radians for the bounded cosine sampling of the emitters (Sample.inl:5-37,62-137).
// DEFINED in the oracle (Cephes asinf / atanf kernels, every step an explicit fma or a single operation) and stated identically on the
// device and oracle agree bit for bit.  |error| ~ 1e-7.
MI_DEV float mi_asinf(float x) {
  float a = fabsf(x), z, xs; int big = a > 0.5f;
  if (big) { z = 0.5f * (1.0f - a); xs = sqrtf(z); } else { xs = a; z = a * a; }
  float p = fmaf(fmaf(fmaf(fmaf(4.2163199048e-2f, z, 2.4181311049e-2f), z, 4.5470025998e-2f), z, 7.4953002686e-2f), z, 1.6666752422e-1f);
  float r = fmaf(p * z, xs, xs);
  if (big) r = 1.57079632679489661923f - (r + r);
  return x < 0.0f ? -r : r;
}
MI_DEV float mi_atanf(float t) {
  float x = fabsf(t), y;
  if (x > 2.414213562373095f) { y = 1.57079632679489661923f; x = -(1.0f / x); }
  else if (x > 0.4142135623730950f) { y = 0.785398163397448309616f; x = (x - 1.0f) / (x + 1.0f); }
  else y = 0.0f;
  float z = x * x;
  float p = fmaf(fmaf(fmaf(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z, -3.33329491539e-1f);
  y = y + fmaf(p * z, x, x);
  return t < 0.0f ? -y : y;
}
MI_DEV float mi_atan2f(float y, float x) {
  if (x > 0.0f) return mi_atanf(y / x);
  if (x < 0.0f) return y < 0.0f ? mi_atanf(y / x) - 3.14159265358979323846f : mi_atanf(y / x) + 3.14159265358979323846f;
  return y > 0.0f ? 1.57079632679489661923f : (y < 0.0f ? -1.57079632679489661923f : 0.0f);
}
MI_DEV void mi_sincosf(float rad, float* s, float* c) {
  float t = rad * 0.159154943091895335769f; // turns
  sincos_2pi(t - floorf(t), s, c);
}

// pow(x, y) of the Phong lobe and of the MIS weights with a variable beta: DEFINED in the oracle (2^(y log2 x) in FP64, every step a
// single IEEE operation or an explicit fma, one rounding to FP32) and stated identically here, so Phong scenes are bit-exact too.
// A real function (not inlined): six inlined copies cost every megakernel variant 100-140 B/lane more scratch and the LDS-resident kernel 2 %
// (10 932 -> 10 687 Msamples/s on C2, which never calls it); as a call it needs 14 VGPRs and the callers spill less than with the library powf.
#define MI_D2U(d) ((uint64_t)__double_as_longlong(d))
#define MI_U2D(u) __longlong_as_double((long long)(u))
__device__ __attribute__((noinline)) float mi_powf(float x, float y) {
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

}  // namespace mi
