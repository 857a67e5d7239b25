// boxtest_rate.hip — two formulations of the ray/AABB slab test, issue-bound comparison on gfx950.
//   A: t0 = lo*inv - oi, t1 = hi*inv - oi, per-axis min/max           (6 fma + 6 min/max + 2 min3/max3 + 2 clamps)
//   B: m = c*inv - oi, tn = m - e*|inv|, tf = m + e*|inv|              (9 fma + 2 min3/max3 + 2 clamps)
// min/max/cmp/cndmask issue at about half the rate of fma/mul/add on this chip but overlap with them (valu_rate.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int kIters = 4096;

template <int MODE> __global__ void __launch_bounds__(256) k(float* out, const float* __restrict__ in) {
  float lox = in[0] + threadIdx.x * 1e-4f, loy = in[1], loz = in[2], hix = in[3], hiy = in[4], hiz = in[5];
  const float ix = in[6], iy = in[7], iz = in[8], ox = in[9], oy = in[10], oz = in[11];
  const float ax = fabsf(ix), ay = fabsf(iy), az = fabsf(iz);
  float tmax = in[12], acc = 0.f; int hits = 0;
  for (int i = 0; i < kIters; ++i) {
    float tn, tf;
    if (MODE == 0) {
      const float t0x = fmaf(lox, ix, -ox), t1x = fmaf(hix, ix, -ox);
      const float t0y = fmaf(loy, iy, -oy), t1y = fmaf(hiy, iy, -oy);
      const float t0z = fmaf(loz, iz, -oz), t1z = fmaf(hiz, iz, -oz);
      tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), 0.0f));
      tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tmax));
    } else {
      const float mx = fmaf(lox, ix, -ox), my = fmaf(loy, iy, -oy), mz = fmaf(loz, iz, -oz);
      const float nx = fmaf(-hix, ax, mx), ny = fmaf(-hiy, ay, my), nz = fmaf(-hiz, az, mz);
      const float fx = fmaf(hix, ax, mx), fy = fmaf(hiy, ay, my), fz = fmaf(hiz, az, mz);
      tn = fmaxf(fmaxf(nx, ny), fmaxf(nz, 0.0f));
      tf = fminf(fminf(fx, fy), fminf(fz, tmax));
    }
    const bool h = tn <= fmaf(tf, 1.000002f, 1e-6f);
    hits += h;
    // perturb the inputs so nothing is hoisted (cheap full-rate adds)
    lox += 1e-3f; loy -= 1e-3f; loz += 2e-3f; hix += 1e-3f; hiy += 3e-3f; hiz -= 1e-3f; tmax += 1e-3f;
    acc += tn;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + hits;
}
template <int MODE> void run(const char* name) {
  const int blocks = 256 * 6;
  float *out, *in; hipMalloc(&out, sizeof(float) * blocks * 256); hipMalloc(&in, 64);
  const float h[13] = {0.1f, 0.2f, 0.3f, 1.1f, 1.2f, 1.3f, 0.7f, -1.3f, 2.1f, 0.2f, 0.3f, 0.1f, 100.f};
  hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, in);
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, in);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %.3f ms  %.2f ns per box test per SIMD\n", name, ms, ms * 1e6 / (double(blocks) * 4 * kIters / 1024));
}
int main() { run<0>("min/max slabs"); run<1>("centre / half-extent slabs"); return 0; }
