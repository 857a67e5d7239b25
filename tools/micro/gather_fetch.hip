// gather_fetch.hip — what does FETCH_SIZE report for the access shapes of the BVH walk?  
//   hipcc -O3 --offload-arch=gfx950 tools/micro/gather_fetch.hip -o tools/micro/gather_fetch
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/gather_fetch/fetch -- tools/micro/gather_fetch
//   rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_HIT_sum --output-format csv -d ... -- tools/micro/gather_fetch
//
// MI355X_MICROARCH.md calibrates FETCH_SIZE only for wide coalesced streaming reads (it reports exactly half the bytes
// there).  The traversal kernels read something else: one lane = one random RECORD of 32 B (quantised binary node),
// 64 B (wide node / full-precision node) or 48 B (triangle, not line-aligned), as 16-byte loads.  Every kernel below
// gathers a KNOWN number of such records, each exactly once per launch (a random permutation, so no record is served
// twice from a cache), from a table far larger than the 256 MB Infinity Cache; the printed byte counts are what the
// counter has to be compared with.  k_stream is the guide's calibrated shape, as the control.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// one lane = one record of REC bytes at a random (permuted) index; REC / 16 dwordx4 loads per lane
template <int REC>
__global__ __launch_bounds__(256) void k_gather(const uint4* __restrict__ table, const uint32_t* __restrict__ perm, uint32_t n, uint32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint4* r = table + size_t(perm[i]) * (REC / 16);
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < REC / 16; ++k) { const uint4 v = r[k]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[i & 255u] = acc;  // never true for the fill pattern: keeps the loads alive, writes nothing
}

// control: wide coalesced streaming read, 16 B per lane
__global__ __launch_bounds__(256) void k_stream(const uint4* __restrict__ table, size_t n16, uint32_t* __restrict__ out) {
  uint32_t acc = 0;
  for (size_t i = size_t(blockIdx.x) * 256u + threadIdx.x; i < n16; i += size_t(gridDim.x) * 256u) { const uint4 v = table[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

__global__ void k_fill(uint4* t, size_t n16) {
  for (size_t i = size_t(blockIdx.x) * 256u + threadIdx.x; i < n16; i += size_t(gridDim.x) * 256u) t[i] = make_uint4(uint32_t(i) | 1u, 3u, 5u, 7u);
}

int main() {
  const size_t table_bytes = size_t(1536) << 20;  // 1.5 GiB >> 256 MiB Infinity Cache
  uint4* table; uint32_t* out; uint32_t* d_perm;
  CHECK(hipMalloc(&table, table_bytes)); CHECK(hipMalloc(&out, 4096));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, table, table_bytes / 16);
  CHECK(hipDeviceSynchronize());
  const uint32_t n = 16u << 20;  // 16 Mi records per launch
  CHECK(hipMalloc(&d_perm, size_t(n) * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto gather = [&](int rec) {
    // a random sample WITHOUT repetition of record indices in [0, table_bytes / rec): multiplicative permutation of the index space
    const uint64_t space = table_bytes / rec;
    std::vector<uint32_t> perm(n), order(n);
    uint64_t x = 0x9E3779B97F4A7C15ull;
    const uint64_t stride = space / n;  // one record out of every `stride` (at a random place inside its stride), visited in a shuffled order
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    for (uint32_t i = n - 1; i > 0; --i) { x = x * 6364136223846793005ull + 1442695040888963407ull; const uint32_t j = uint32_t((x >> 33) % (i + 1)); std::swap(order[i], order[j]); }
    for (uint32_t i = 0; i < n; ++i) { x = x * 6364136223846793005ull + 1442695040888963407ull; perm[i] = uint32_t(uint64_t(order[i]) * stride + (x >> 33) % stride); }
    CHECK(hipMemcpy(d_perm, perm.data(), size_t(n) * 4, hipMemcpyHostToDevice));
    CHECK(hipEventRecord(e0));
    if (rec == 32) hipLaunchKernelGGL(k_gather<32>, dim3(n / 256), dim3(256), 0, 0, table, d_perm, n, out);
    if (rec == 48) hipLaunchKernelGGL(k_gather<48>, dim3(n / 256), dim3(256), 0, 0, table, d_perm, n, out);
    if (rec == 64) hipLaunchKernelGGL(k_gather<64>, dim3(n / 256), dim3(256), 0, 0, table, d_perm, n, out);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double rec_bytes = double(n) * rec, idx_bytes = double(n) * 4;
    printf("k_gather<%d>: %u records, %.0f record bytes (+ %.0f index bytes, streamed) = %.1f KiB; %.3f ms, %.1f GB/s of records\n", rec, n, rec_bytes, idx_bytes,
           (rec_bytes + idx_bytes) / 1024.0, ms, rec_bytes / ms * 1e-6);
  };
  gather(32); gather(48); gather(64);
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_stream, dim3(256 * 16), dim3(256), 0, 0, table, size_t(1) << 26, out);  // 1 GiB
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("k_stream: %.0f bytes = %.1f KiB; %.3f ms, %.1f GB/s\n", double(size_t(1) << 30), double(size_t(1) << 30) / 1024.0, ms, double(size_t(1) << 30) / ms * 1e-6);
  return 0;
}
