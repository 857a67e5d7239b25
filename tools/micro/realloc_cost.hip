// What does growing a buffer cost while a large arena is held?  (r04: a BPT launch spent 3.5 s in hipFree + hipMalloc of its 0.9 GB values buffer.)
//   hipcc --offload-arch=gfx950 -O2 tools/micro/realloc_cost.hip -o /tmp/realloc_cost && /tmp/realloc_cost [arena GB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void touch(float* p, size_t n) { size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; if (i < n) p[i] = 1.0f; }
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const size_t arena_gb = argc > 1 ? std::atoll(argv[1]) : 116;
  void* arena = nullptr;
  double t0 = now();
  if (hipMalloc(&arena, arena_gb << 30) != hipSuccess) { std::printf("arena alloc failed\n"); return 1; }
  std::printf("hipMalloc %zu GB: %.1f ms\n", arena_gb, now() - t0);
  t0 = now(); touch<<<dim3(1u << 20), dim3(256)>>>((float*)arena, size_t(1) << 28); hipDeviceSynchronize(); std::printf("touch 1 GB of it: %.1f ms\n", now() - t0);
  for (int rep = 0; rep < 3; ++rep) {
    float* a = nullptr;
    const size_t n1 = size_t(750) << 20, n2 = size_t(925) << 20;
    t0 = now(); hipMalloc((void**)&a, n1); std::printf("hipMalloc 750 MB: %.1f ms\n", now() - t0);
    t0 = now(); touch<<<dim3(unsigned(n1 / 4 / 256)), dim3(256)>>>(a, n1 / 4); hipDeviceSynchronize(); std::printf("  kernel over it: %.1f ms\n", now() - t0);
    t0 = now(); hipFree(a); std::printf("  hipFree: %.1f ms\n", now() - t0);
    t0 = now(); hipMalloc((void**)&a, n2); std::printf("  hipMalloc 925 MB: %.1f ms\n", now() - t0);
    t0 = now(); touch<<<dim3(unsigned(n2 / 4 / 256)), dim3(256)>>>(a, n2 / 4); hipDeviceSynchronize(); std::printf("  kernel over it: %.1f ms\n", now() - t0);
    t0 = now(); hipFree(a); std::printf("  hipFree: %.1f ms\n", now() - t0);
  }
  t0 = now(); hipFree(arena); std::printf("hipFree arena: %.1f ms\n", now() - t0);
  return 0;
}
