// lbvh_build.hip — linear BVH built on the device (gfx950); replaces Embree's rtcCommit
// (Scene.cpp:47-66: RTC_SCENE_STATIC | RTC_SCENE_HIGH_QUALITY, one geometry per mesh).
//
//   1. k_tri_bounds   per-triangle AABB + scene AABB (ordered-uint atomic min/max)
//   2. k_morton       30-bit Morton code of the AABB centre, key = code, value = triangle id
//   3. radix sort     4 stable LSD passes x 8 bits (k_hist / k_scan / k_scatter); stability makes
//                     the order equal to sorting (code, triangle id) — keys are unique
//   4. k_karras       Karras 2012 hierarchy over the 64-bit keys (code << 32 | id)
//   5. k_refit        bottom-up box propagation, second arriver continues (agent-scope fences)
//   6. k_depth        longest root-to-leaf path -> traversal stack capacity
//   7. k_emit         Morton-ordered triangle records (intersection stream + shading stream)
// All arithmetic is integer or exact float min/max except the Morton quantisation and the leaf
// padding, which are written with -ffp-contract=off so the CPU oracle reproduces them bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../../include/mi_pt.h"
#include "layout.h"

namespace mi {

namespace {

__device__ __forceinline__ uint32_t f2ord(float f) {
  uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

__global__ void k_tri_bounds(uint32_t n, const float* __restrict__ pos, const uint32_t* __restrict__ idx, float* __restrict__ tri_lo,
                             float* __restrict__ tri_hi, uint32_t* __restrict__ scene_ord /*[6]: lo xyz (min), hi xyz (max)*/) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  for (int k = 0; k < 3; ++k) {
    const float* p = pos + 3 * size_t(idx[3 * size_t(t) + k]);
    for (int a = 0; a < 3; ++a) { if (p[a] < lo[a]) lo[a] = p[a]; if (p[a] > hi[a]) hi[a] = p[a]; }
  }
  for (int a = 0; a < 3; ++a) {
    tri_lo[3 * size_t(t) + a] = lo[a]; tri_hi[3 * size_t(t) + a] = hi[a];
    atomicMin(&scene_ord[a], f2ord(lo[a]));
    atomicMax(&scene_ord[3 + a], f2ord(hi[a]));
  }
}

__device__ __forceinline__ uint32_t expand_bits10(uint32_t v) {
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}
__device__ __forceinline__ uint32_t quant10(float c, float lo, float hi) {
  const float ext = hi - lo;
  const float nrm = ext > 0.0f ? (c - lo) / ext : 0.0f;
  float q = nrm * 1024.0f;
  if (!(q > 0.0f)) q = 0.0f;
  if (q > 1023.0f) q = 1023.0f;
  return uint32_t(q);
}

__global__ void k_morton(uint32_t n, const float* __restrict__ tri_lo, const float* __restrict__ tri_hi,
                         const uint32_t* __restrict__ scene_ord, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  uint32_t q[3];
  for (int a = 0; a < 3; ++a) {
    const float c = (tri_lo[3 * size_t(t) + a] + tri_hi[3 * size_t(t) + a]) * 0.5f;
    q[a] = quant10(c, ord2f(scene_ord[a]), ord2f(scene_ord[3 + a]));
  }
  keys[t] = (expand_bits10(q[0]) << 2) | (expand_bits10(q[1]) << 1) | expand_bits10(q[2]);
  vals[t] = t;
}

// ---- stable LSD radix sort, 8 bits per pass, 2048 keys per workgroup ----
constexpr uint32_t kSortTile = 2048;

__global__ __launch_bounds__(256) void k_hist(const uint32_t* __restrict__ keys, uint32_t n, uint32_t shift, uint32_t* __restrict__ hist,
                                             uint32_t nblk) {
  __shared__ uint32_t h[256];
  const uint32_t tid = threadIdx.x;
  h[tid] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * kSortTile;
  for (uint32_t r = 0; r < kSortTile / 256; ++r) {
    const uint32_t i = base + r * 256 + tid;
    if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[tid * nblk + blockIdx.x] = h[tid];
}

// exclusive scan of `total` counters in place, one workgroup of 1024 threads
__global__ __launch_bounds__(1024) void k_scan(uint32_t* __restrict__ data, uint32_t total) {
  __shared__ uint32_t sums[1024];
  const uint32_t tid = threadIdx.x;
  const uint32_t chunk = (total + 1023u) / 1024u;
  const uint32_t b = tid * chunk, e = b + chunk < total ? b + chunk : total;
  uint32_t s = 0;
  for (uint32_t i = b; i < e; ++i) s += data[i];
  sums[tid] = s;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {  // Hillis–Steele inclusive scan
    uint32_t v = tid >= off ? sums[tid - off] : 0u;
    __syncthreads();
    sums[tid] += v;
    __syncthreads();
  }
  uint32_t run = tid ? sums[tid - 1] : 0u;
  for (uint32_t i = b; i < e; ++i) { const uint32_t v = data[i]; data[i] = run; run += v; }
}

__global__ __launch_bounds__(256) void k_scatter(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint32_t n, uint32_t shift,
                                                const uint32_t* __restrict__ hist, uint32_t nblk) {
  __shared__ uint32_t base[256];
  __shared__ uint32_t wcount[4][256];
  const uint32_t tid = threadIdx.x, wave = tid >> 6;
  base[tid] = hist[tid * nblk + blockIdx.x];
  for (int w = 0; w < 4; ++w) wcount[w][tid] = 0;
  __syncthreads();
  const uint32_t tile = blockIdx.x * kSortTile;
  for (uint32_t r = 0; r < kSortTile / 256; ++r) {
    const uint32_t i = tile + r * 256 + tid;
    const bool valid = i < n;
    const uint32_t key = valid ? keys_in[i] : 0u;
    const uint32_t digit = (key >> shift) & 255u;
    // lanes of this wave holding the same digit (match-any by 8 ballots)
    uint64_t mask = __ballot(valid);
    for (uint32_t bit = 0; bit < 8; ++bit) {
      const bool set = (digit >> bit) & 1u;
      const uint64_t bal = __ballot(set);
      mask &= set ? bal : ~bal;
    }
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
    if (valid && rank == 0) wcount[wave][digit] = uint32_t(__popcll(mask));
    __syncthreads();
    if (valid) {
      uint32_t off = base[digit];
      for (uint32_t w = 0; w < wave; ++w) off += wcount[w][digit];
      keys_out[off + rank] = key;
      vals_out[off + rank] = vals_in[i];
    }
    __syncthreads();
    base[tid] += wcount[0][tid] + wcount[1][tid] + wcount[2][tid] + wcount[3][tid];
    for (int w = 0; w < 4; ++w) wcount[w][tid] = 0;
    __syncthreads();
  }
}

// ---- Karras 2012: one thread per internal node ----
__device__ __forceinline__ int delta_fn(const uint32_t* __restrict__ code, const uint32_t* __restrict__ id, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const uint64_t a = (uint64_t(code[i]) << 32) | id[i], b = (uint64_t(code[j]) << 32) | id[j];
  return __clzll((long long)(a ^ b));
}

__global__ void k_karras(int n, const uint32_t* __restrict__ code, const uint32_t* __restrict__ id, mi_bvh_node* __restrict__ nodes,
                         uint32_t* __restrict__ leaf_parent) {
  const int i = int(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n - 1) return;
  const int d = (delta_fn(code, id, n, i, i + 1) - delta_fn(code, id, n, i, i - 1)) >= 0 ? 1 : -1;
  const int dmin = delta_fn(code, id, n, i, i - d);
  int lmax = 2;
  while (delta_fn(code, id, n, i, i + lmax * d) > dmin) lmax *= 2;
  int l = 0;
  for (int t = lmax / 2; t >= 1; t /= 2)
    if (delta_fn(code, id, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = delta_fn(code, id, n, i, j);
  int s = 0, t = l;
  do {
    t = (t + 1) / 2;
    if (delta_fn(code, id, n, i, i + (s + t) * d) > dnode) s += t;
  } while (t > 1);
  const int gamma = i + s * d + (d < 0 ? d : 0);
  const int lo_i = i < j ? i : j, hi_i = i < j ? j : i;
  const int left = (lo_i == gamma) ? ~gamma : gamma;
  const int right = (hi_i == gamma + 1) ? ~(gamma + 1) : gamma + 1;
  nodes[i].link0 = left;
  nodes[i].link1 = right;
  nodes[i].reserved = 0;
  if (left >= 0) nodes[left].parent = uint32_t(i); else leaf_parent[~left] = uint32_t(i);
  if (right >= 0) nodes[right].parent = uint32_t(i); else leaf_parent[~right] = uint32_t(i);
  if (i == 0) nodes[0].parent = 0xFFFFFFFFu;
}

__device__ __forceinline__ float ld_agent(const float* p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// one thread per leaf; the second thread to reach a node carries the union upward
__global__ void k_refit(uint32_t n, const uint32_t* __restrict__ sorted_tri, const float* __restrict__ tri_lo, const float* __restrict__ tri_hi,
                        mi_bvh_node* nodes, const uint32_t* __restrict__ leaf_parent, uint32_t* visit) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t t = sorted_tri[i];
  float lo[3], hi[3];
  for (int a = 0; a < 3; ++a) {  // padded leaf box (same formula as the oracle's pad_box)
    const float l = tri_lo[3 * size_t(t) + a], h = tri_hi[3 * size_t(t) + a];
    const float m = fmaxf(fabsf(l), fabsf(h));
    const float pad = m * 0x1p-20f + 0x1p-40f;
    lo[a] = l - pad; hi[a] = h + pad;
  }
  int cur = ~int(i);
  uint32_t parent = leaf_parent[i];
  while (parent != 0xFFFFFFFFu) {
    mi_bvh_node* nd = &nodes[parent];
    const bool slot0 = nd->link0 == cur;
    float* dlo = slot0 ? nd->lo0 : nd->lo1;
    float* dhi = slot0 ? nd->hi0 : nd->hi1;
    for (int a = 0; a < 3; ++a) { dlo[a] = lo[a]; dhi[a] = hi[a]; }
    __threadfence();
    const uint32_t old = atomicAdd(&visit[parent], 1u);
    if (old == 0u) return;
    __threadfence();
    const float* slo = slot0 ? nd->lo1 : nd->lo0;
    const float* shi = slot0 ? nd->hi1 : nd->hi0;
    for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], ld_agent(&slo[a])); hi[a] = fmaxf(hi[a], ld_agent(&shi[a])); }
    cur = int(parent);
    parent = __hip_atomic_load(&nd->parent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ void k_depth(uint32_t n, const mi_bvh_node* __restrict__ nodes, const uint32_t* __restrict__ leaf_parent, uint32_t* max_depth) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t d = 1, parent = leaf_parent[i];
  while (parent != 0xFFFFFFFFu) { ++d; parent = nodes[parent].parent; }
  atomicMax(max_depth, d);
}

__global__ void k_emit(uint32_t n, const uint32_t* __restrict__ sorted_tri, const float* __restrict__ pos, const float* __restrict__ tan,
                       const uint32_t* __restrict__ idx, const uint32_t* __restrict__ tri_material, float4* __restrict__ tri_isect,
                       float4* __restrict__ tri_shade) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t t = sorted_tri[i];
  const uint32_t i0 = idx[3 * size_t(t)], i1 = idx[3 * size_t(t) + 1], i2 = idx[3 * size_t(t) + 2];
  const float* a = pos + 3 * size_t(i0); const float* b = pos + 3 * size_t(i1); const float* c = pos + 3 * size_t(i2);
  const float e1x = a[0] - b[0], e1y = a[1] - b[1], e1z = a[2] - b[2];  // e1 = v0 - v1
  const float e2x = c[0] - a[0], e2y = c[1] - a[1], e2z = c[2] - a[2];  // e2 = v2 - v0
  const uint32_t mat = tri_material[t];
  tri_isect[3 * size_t(i)] = make_float4(a[0], a[1], a[2], e1x);
  tri_isect[3 * size_t(i) + 1] = make_float4(e1y, e1z, e2x, e2y);
  tri_isect[3 * size_t(i) + 2] = make_float4(e2z, __uint_as_float(t), __uint_as_float(1u << (mat & 3u)), 0.0f);
  float f[32];
  const float* t0 = tan + 9 * size_t(i0); const float* t1 = tan + 9 * size_t(i1); const float* t2 = tan + 9 * size_t(i2);
  for (int k = 0; k < 9; ++k) { f[k] = t0[k]; f[9 + k] = t1[k]; f[18 + k] = t2[k]; }
  f[27] = __uint_as_float(mat);
  // unit geometric normal g = normalize(-cross(e2, e1)) (RayIsect.hpp:24), same fma placement as the contract's cross/dot
  const float nx = -fmaf(e2y, e1z, -(e1y * e2z)), ny = -fmaf(e2z, e1x, -(e1z * e2x)), nz = -fmaf(e2x, e1y, -(e1x * e2y));
  const float inv = 1.0f / sqrtf(fmaf(nz, nz, fmaf(ny, ny, nx * nx)));
  f[28] = nx * inv; f[29] = ny * inv; f[30] = nz * inv; f[31] = 0.0f;
  for (int k = 0; k < 8; ++k) tri_shade[8 * size_t(i) + k] = make_float4(f[4 * k], f[4 * k + 1], f[4 * k + 2], f[4 * k + 3]);
}

// 16-bit grid coordinates of the child boxes, rounded outward by one extra cell so that float rounding in the
// ray's grid transform can never cut into the true box.
__device__ __forceinline__ uint32_t q_lo(float v, float lo, float inv_step) {
  float q = floorf((v - lo) * inv_step) - 1.0f;
  q = q < 0.0f ? 0.0f : (q > 65535.0f ? 65535.0f : q);
  return uint32_t(q);
}
__device__ __forceinline__ uint32_t q_hi(float v, float lo, float inv_step) {
  float q = ceilf((v - lo) * inv_step) + 1.0f;
  q = q < 0.0f ? 0.0f : (q > 65535.0f ? 65535.0f : q);
  return uint32_t(q);
}
__global__ void k_quantize(uint32_t n_nodes, const mi_bvh_node* __restrict__ nodes, uint4* __restrict__ qnodes, float lx, float ly, float lz,
                           float ix, float iy, float iz) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  const mi_bvh_node n = nodes[i];
  uint4 a, b;
  a.x = q_lo(n.lo0[0], lx, ix) | (q_lo(n.lo0[1], ly, iy) << 16);
  a.y = q_lo(n.lo0[2], lz, iz) | (q_hi(n.hi0[0], lx, ix) << 16);
  a.z = q_hi(n.hi0[1], ly, iy) | (q_hi(n.hi0[2], lz, iz) << 16);
  a.w = uint32_t(n.link0);
  b.x = q_lo(n.lo1[0], lx, ix) | (q_lo(n.lo1[1], ly, iy) << 16);
  b.y = q_lo(n.lo1[2], lz, iz) | (q_hi(n.hi1[0], lx, ix) << 16);
  b.z = q_hi(n.hi1[1], ly, iy) | (q_hi(n.hi1[2], lz, iz) << 16);
  b.w = uint32_t(n.link1);
  qnodes[2 * size_t(i)] = a;
  qnodes[2 * size_t(i) + 1] = b;
}

#define BUILD_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

}  // namespace

// Device inputs: pos [nv][3], tan [nv][9], idx [nt][3], tri_material [nt].
// Outputs: nodes / tri_isect / tri_shade (sections of the scene blob), sorted_tri [nt],
// morton [nt], scene bounds, depth.  `scratch` allocations are made and freed here (one-off).
hipError_t build_lbvh(uint32_t nt, const float* pos, const float* tan, const uint32_t* idx, const uint32_t* tri_material,
                      mi_bvh_node* nodes, float4* tri_isect, float4* tri_shade, uint32_t* sorted_tri, uint32_t* morton,
                      float scene_lo[3], float scene_hi[3], uint32_t* max_depth_out, float* build_ms, hipStream_t stream) {
  const uint32_t nblk_sort = (nt + kSortTile - 1) / kSortTile;
  const uint32_t g256 = (nt + 255) / 256;
  float *tri_lo = nullptr, *tri_hi = nullptr;
  uint32_t *scene_ord = nullptr, *keys_b = nullptr, *vals_b = nullptr, *hist = nullptr, *leaf_parent = nullptr, *visit = nullptr, *depth = nullptr;
  BUILD_CHECK(hipMalloc(&tri_lo, sizeof(float) * 3 * size_t(nt)));
  BUILD_CHECK(hipMalloc(&tri_hi, sizeof(float) * 3 * size_t(nt)));
  BUILD_CHECK(hipMalloc(&scene_ord, sizeof(uint32_t) * 8));
  BUILD_CHECK(hipMalloc(&keys_b, sizeof(uint32_t) * size_t(nt)));
  BUILD_CHECK(hipMalloc(&vals_b, sizeof(uint32_t) * size_t(nt)));
  BUILD_CHECK(hipMalloc(&hist, sizeof(uint32_t) * 256 * size_t(nblk_sort)));
  BUILD_CHECK(hipMalloc(&leaf_parent, sizeof(uint32_t) * size_t(nt)));
  BUILD_CHECK(hipMalloc(&visit, sizeof(uint32_t) * size_t(nt)));
  BUILD_CHECK(hipMalloc(&depth, sizeof(uint32_t)));
  const uint32_t ord_init[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u};
  BUILD_CHECK(hipMemcpyAsync(scene_ord, ord_init, sizeof ord_init, hipMemcpyHostToDevice, stream));
  BUILD_CHECK(hipMemsetAsync(visit, 0, sizeof(uint32_t) * size_t(nt), stream));
  BUILD_CHECK(hipMemsetAsync(depth, 0, sizeof(uint32_t), stream));
  BUILD_CHECK(hipMemsetAsync(leaf_parent, 0xFF, sizeof(uint32_t) * size_t(nt), stream));
  hipEvent_t ev0, ev1;
  BUILD_CHECK(hipEventCreate(&ev0)); BUILD_CHECK(hipEventCreate(&ev1));
  BUILD_CHECK(hipEventRecord(ev0, stream));

  hipLaunchKernelGGL(k_tri_bounds, dim3(g256), dim3(256), 0, stream, nt, pos, idx, tri_lo, tri_hi, scene_ord);
  uint32_t* keys_a = morton;      // ping
  uint32_t* vals_a = sorted_tri;  // ping
  hipLaunchKernelGGL(k_morton, dim3(g256), dim3(256), 0, stream, nt, tri_lo, tri_hi, scene_ord, keys_a, vals_a);
  uint32_t *kin = keys_a, *vin = vals_a, *kout = keys_b, *vout = vals_b;
  for (uint32_t pass = 0; pass < 4; ++pass) {  // 30-bit codes: 4 passes of 8 bits; ends back in (morton, sorted_tri)
    hipLaunchKernelGGL(k_hist, dim3(nblk_sort), dim3(256), 0, stream, kin, nt, pass * 8, hist, nblk_sort);
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, stream, hist, 256u * nblk_sort);
    hipLaunchKernelGGL(k_scatter, dim3(nblk_sort), dim3(256), 0, stream, kin, vin, kout, vout, nt, pass * 8, hist, nblk_sort);
    uint32_t* t = kin; kin = kout; kout = t;
    t = vin; vin = vout; vout = t;
  }
  if (nt > 1) {
    hipLaunchKernelGGL(k_karras, dim3((nt - 1 + 255) / 256), dim3(256), 0, stream, int(nt), morton, sorted_tri, nodes, leaf_parent);
    hipLaunchKernelGGL(k_refit, dim3(g256), dim3(256), 0, stream, nt, sorted_tri, tri_lo, tri_hi, nodes, leaf_parent, visit);
  }
  hipLaunchKernelGGL(k_depth, dim3(g256), dim3(256), 0, stream, nt, nodes, leaf_parent, depth);
  hipLaunchKernelGGL(k_emit, dim3(g256), dim3(256), 0, stream, nt, sorted_tri, pos, tan, idx, tri_material, tri_isect, tri_shade);
  BUILD_CHECK(hipGetLastError());
  BUILD_CHECK(hipEventRecord(ev1, stream));
  uint32_t ord[8];
  BUILD_CHECK(hipMemcpyAsync(ord, scene_ord, sizeof ord, hipMemcpyDeviceToHost, stream));
  BUILD_CHECK(hipMemcpyAsync(max_depth_out, depth, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
  BUILD_CHECK(hipStreamSynchronize(stream));
  for (int a = 0; a < 3; ++a) {
    const uint32_t ul = ord[a], uh = ord[3 + a];
    uint32_t bl = (ul & 0x80000000u) ? (ul & 0x7FFFFFFFu) : ~ul, bh = (uh & 0x80000000u) ? (uh & 0x7FFFFFFFu) : ~uh;
    __builtin_memcpy(&scene_lo[a], &bl, 4); __builtin_memcpy(&scene_hi[a], &bh, 4);
  }
  BUILD_CHECK(hipEventElapsedTime(build_ms, ev0, ev1));
  hipEventDestroy(ev0); hipEventDestroy(ev1);
  hipFree(tri_lo); hipFree(tri_hi); hipFree(scene_ord); hipFree(keys_b); hipFree(vals_b); hipFree(hist); hipFree(leaf_parent);
  hipFree(visit); hipFree(depth);
  return hipSuccess;
}

hipError_t quantize_nodes(uint32_t n_nodes, const mi_bvh_node* nodes, uint4* qnodes, const float lo[3], const float inv_step[3],
                          hipStream_t stream) {
  if (n_nodes == 0) return hipSuccess;
  hipLaunchKernelGGL(k_quantize, dim3((n_nodes + 255) / 256), dim3(256), 0, stream, n_nodes, nodes, qnodes, lo[0], lo[1], lo[2], inv_step[0],
                     inv_step[1], inv_step[2]);
  return hipGetLastError();
}

}  // namespace mi
