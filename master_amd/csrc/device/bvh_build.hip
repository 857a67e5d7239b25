// bvh_build.hip — BVH2 built on the device (gfx950); replaces Embree's rtcCommit
// (Scene.cpp:47-66: RTC_SCENE_STATIC | RTC_SCENE_HIGH_QUALITY, one geometry per mesh).
//
//   1. k_tri_bounds   per-triangle AABB + scene AABB (ordered-uint atomic min/max)
//   2. k_morton       63-bit Morton code (21 bits per axis) of the AABB centre, value = triangle id.  21 bits keep
//                     small objects apart when far-away light quads stretch the scene box ("teapot in a stadium":
//                     MetalRings.blend spans 400 units around 0.1-unit triangles)
//   3. radix sort     8 stable LSD passes x 8 bits (k_hist / k_scan / k_scatter); stability makes
//                     the order equal to sorting (code, triangle id) — keys are unique
//   4. hierarchy      builder 1 (default): PLOC — parallel locally-ordered clustering (Meister & Bittner 2018):
//                     every cluster looks 16 places left and right in Morton order for the partner with the
//                     smallest union surface area, mutual pairs merge, survivors are compacted; repeated until
//                     one cluster is left.  Ties are broken by a symmetric pair hash so that regular meshes
//                     still merge a constant fraction per round.  The finished tree is renumbered in depth-first
//                     order (root = node 0, first child right behind its parent, subtrees contiguous).
//                     builder 0: Karras 2012 LBVH over (code, id) + bottom-up refit (agent-scope fences)
//   5. k_depth        longest root-to-leaf path -> traversal stack capacity
//   6. k_emit         Morton-ordered triangle records (intersection stream + shading stream)
// All arithmetic is integer or exact float min/max except the Morton quantisation, the leaf padding and
// the surface-area measure, which are written with -ffp-contract=off / explicit fmaf so the CPU oracle
// reproduces the tree bit for bit.
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

__device__ __forceinline__ uint64_t expand_bits21(uint32_t v) {
  uint64_t x = v & 0x1FFFFFu;
  x = (x | (x << 32)) & 0x001F00000000FFFFull;
  x = (x | (x << 16)) & 0x001F0000FF0000FFull;
  x = (x | (x << 8)) & 0x100F00F00F00F00Full;
  x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
  x = (x | (x << 2)) & 0x1249249249249249ull;
  return x;
}
__device__ __forceinline__ uint32_t quant21(float c, float lo, float hi) {
  const float ext = hi - lo;
  const float nrm = ext > 0.0f ? (c - lo) / ext : 0.0f;
  float q = nrm * 2097152.0f;
  if (!(q > 0.0f)) q = 0.0f;
  if (q > 2097151.0f) q = 2097151.0f;
  return uint32_t(q);
}

__global__ void k_morton(uint32_t n, const float* __restrict__ tri_lo, const float* __restrict__ tri_hi,
                         const uint32_t* __restrict__ scene_ord, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  uint32_t q[3];
  for (int a = 0; a < 3; ++a) {
    const float c = (tri_lo[3 * size_t(t) + a] + tri_hi[3 * size_t(t) + a]) * 0.5f;
    q[a] = quant21(c, ord2f(scene_ord[a]), ord2f(scene_ord[3 + a]));
  }
  keys[t] = (expand_bits21(q[0]) << 2) | (expand_bits21(q[1]) << 1) | expand_bits21(q[2]);
  vals[t] = t;
}

// ---- stable LSD radix sort, 8 bits per pass, 2048 keys per workgroup ----
constexpr uint32_t kSortTile = 2048;

__global__ __launch_bounds__(256) void k_hist(const uint64_t* __restrict__ keys, uint32_t n, uint32_t shift, uint32_t* __restrict__ hist,
                                             uint32_t nblk) {
  __shared__ uint32_t h[256];
  const uint32_t tid = threadIdx.x;
  h[tid] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * kSortTile;
  for (uint32_t r = 0; r < kSortTile / 256; ++r) {
    const uint32_t i = base + r * 256 + tid;
    if (i < n) atomicAdd(&h[uint32_t(keys[i] >> shift) & 255u], 1u);
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

// Tiled exclusive scan (r03): one workgroup walking a 2 M-entry array in per-thread chunks was serial and uncoalesced (3.9 ms of a 14 ms
// build).  Tiles of 4096 entries: (1) every tile is scanned in place, coalesced, and leaves its sum; (2) k_scan over the tile sums (one workgroup, at
// most a few thousand entries); (3) every tile adds its offset.
constexpr uint32_t kScanTile = 4096u;
__global__ __launch_bounds__(1024) void k_scan_tile(uint32_t* __restrict__ data, uint32_t total, uint32_t* __restrict__ tile_sums) {
  __shared__ uint32_t sums[1024];
  const uint32_t tid = threadIdx.x, base = blockIdx.x * kScanTile + tid * 4u;
  uint32_t v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = base + k < total ? data[base + k] : 0u;
  sums[tid] = v[0] + v[1] + v[2] + v[3];
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {
    const uint32_t u = tid >= off ? sums[tid - off] : 0u;
    __syncthreads();
    sums[tid] += u;
    __syncthreads();
  }
  uint32_t run = tid ? sums[tid - 1] : 0u;
#pragma unroll
  for (int k = 0; k < 4; ++k) { if (base + k < total) data[base + k] = run; run += v[k]; }
  if (tid == 1023u) tile_sums[blockIdx.x] = sums[1023];
}
__global__ __launch_bounds__(256) void k_scan_add(uint32_t* __restrict__ data, uint32_t total, const uint32_t* __restrict__ tile_offsets) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < total) data[i] += tile_offsets[i / kScanTile];
}
// exclusive scan of data[0, total) in place; tile_sums: scratch of at least total / 4096 + 2 entries (unused for short arrays)
void scan_exclusive(uint32_t* data, uint32_t total, uint32_t* tile_sums, hipStream_t stream) {
  if (total <= 4u * kScanTile || tile_sums == nullptr) { hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, stream, data, total); return; }
  const uint32_t tiles = (total + kScanTile - 1u) / kScanTile;
  hipLaunchKernelGGL(k_scan_tile, dim3(tiles), dim3(1024), 0, stream, data, total, tile_sums);
  hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, stream, tile_sums, tiles);
  hipLaunchKernelGGL(k_scan_add, dim3((total + 255u) / 256u), dim3(256), 0, stream, data, total, tile_sums);
}

__global__ __launch_bounds__(256) void k_scatter(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint32_t n, uint32_t shift,
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
    const uint64_t key = valid ? keys_in[i] : 0ull;
    const uint32_t digit = uint32_t(key >> shift) & 255u;
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
__device__ __forceinline__ int delta_fn(const uint64_t* __restrict__ code, const uint32_t* __restrict__ id, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const uint64_t x = code[i] ^ code[j];  // common prefix of the 96-bit key (code, id)
  return x ? __clzll((long long)x) : 64 + __clz(int(id[i] ^ id[j]));
}

__global__ void k_karras(int n, const uint64_t* __restrict__ code, const uint32_t* __restrict__ id, mi_bvh_node* __restrict__ nodes,
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

// ---- PLOC: parallel locally-ordered clustering ----
// A cluster is (box, link): float4 {lo.xyz, link bits}, float4 {hi.xyz, 0}.  link as in mi_bvh_node.
#ifndef MI_PLOC_RADIUS
#define MI_PLOC_RADIUS 16
#endif
constexpr int kPlocRadius = MI_PLOC_RADIUS;

__device__ __forceinline__ float union_area(const float4 alo, const float4 ahi, const float4 blo, const float4 bhi) {
  const float dx = fmaxf(ahi.x, bhi.x) - fminf(alo.x, blo.x);
  const float dy = fmaxf(ahi.y, bhi.y) - fminf(alo.y, blo.y);
  const float dz = fmaxf(ahi.z, bhi.z) - fminf(alo.z, blo.z);
  return fmaf(dx, dy, fmaf(dy, dz, dz * dx));  // half the surface area
}
__device__ __forceinline__ uint32_t pair_hash(uint32_t a, uint32_t b) {  // a < b
  uint32_t h = (a * 0x9E3779B1u) ^ (b * 0x85EBCA77u + 0x165667B1u);
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}

__global__ __launch_bounds__(256) void k_ploc_init(uint32_t n, const uint32_t* __restrict__ sorted_tri, const float* __restrict__ tri_lo,
                                                  const float* __restrict__ tri_hi, float4* __restrict__ cl) {
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
  cl[2 * size_t(i)] = make_float4(lo[0], lo[1], lo[2], __int_as_float(~int(i)));
  cl[2 * size_t(i) + 1] = make_float4(hi[0], hi[1], hi[2], __uint_as_float(0u));  // .w = internal nodes below: none
}

// nn[i] = the partner j in [i - R, i + R] minimising (union area, pair hash, min(i,j), max(i,j)): a strict total
// order on unordered pairs, so "i and j chose each other" is well defined and the globally smallest pair always merges.
__global__ __launch_bounds__(256) void k_ploc_nn(uint32_t n, const float4* __restrict__ cl, uint32_t* __restrict__ nn) {
  __shared__ float4 tlo[256 + 2 * kPlocRadius], thi[256 + 2 * kPlocRadius];
  const int base = int(blockIdx.x * 256u) - kPlocRadius;
  for (int k = int(threadIdx.x); k < 256 + 2 * kPlocRadius; k += 256) {
    const int g = base + k;
    if (g >= 0 && g < int(n)) { tlo[k] = cl[2 * size_t(g)]; thi[k] = cl[2 * size_t(g) + 1]; }
  }
  __syncthreads();
  const int i = int(blockIdx.x * 256u + threadIdx.x);
  if (i >= int(n)) return;
  const int li = int(threadIdx.x) + kPlocRadius;
  const float4 alo = tlo[li], ahi = thi[li];
  int best = -1; float best_area = 0.0f;
  const int j0 = i - kPlocRadius < 0 ? 0 : i - kPlocRadius, j1 = i + kPlocRadius > int(n) - 1 ? int(n) - 1 : i + kPlocRadius;
  for (int j = j0; j <= j1; ++j) {
    if (j == i) continue;
    const float area = union_area(alo, ahi, tlo[j - base], thi[j - base]);
    bool take;
    if (best < 0) take = true;
    else if (area != best_area) take = area < best_area;
    else {
      const uint32_t a0 = uint32_t(i < j ? i : j), a1 = uint32_t(i < j ? j : i);
      const uint32_t b0 = uint32_t(i < best ? i : best), b1 = uint32_t(i < best ? best : i);
      const uint32_t ha = pair_hash(a0, a1), hb = pair_hash(b0, b1);
      take = ha != hb ? ha < hb : (a0 != b0 ? a0 < b0 : a1 < b1);
    }
    if (take) { best = j; best_area = area; }
  }
  nn[i] = uint32_t(best);
}

// counts[b] = leaders (i < nn[i], mutual) in block b; counts[nblk + b] = removed (i > nn[i], mutual); counts[2 nblk] = 0
__global__ __launch_bounds__(256) void k_ploc_count(uint32_t n, const uint32_t* __restrict__ nn, uint32_t* __restrict__ counts, uint32_t nblk) {
  __shared__ uint32_t wl[4], wr[4];
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  bool leader = false, removed = false;
  if (i < n) {
    const uint32_t j = nn[i];
    const bool mutual = nn[j] == i;
    leader = mutual && i < j; removed = mutual && i > j;
  }
  const uint32_t cl_ = uint32_t(__popcll(__ballot(leader))), cr = uint32_t(__popcll(__ballot(removed)));
  if ((threadIdx.x & 63u) == 0) { wl[threadIdx.x >> 6] = cl_; wr[threadIdx.x >> 6] = cr; }
  __syncthreads();
  if (threadIdx.x == 0) {
    counts[blockIdx.x] = wl[0] + wl[1] + wl[2] + wl[3];
    counts[nblk + blockIdx.x] = wr[0] + wr[1] + wr[2] + wr[3];
    if (blockIdx.x == 0) counts[2 * nblk] = 0;
  }
}

// after the exclusive scan of counts: S[b] leaders before block b, S[nblk] all leaders, S[nblk + b] - S[nblk] removed before block b
__global__ __launch_bounds__(256) void k_ploc_apply(uint32_t n, const float4* __restrict__ cl_in, const uint32_t* __restrict__ nn,
                                                   const uint32_t* __restrict__ S, uint32_t nblk, uint32_t next_node,
                                                   float4* __restrict__ cl_out, mi_bvh_node* __restrict__ nodes, uint32_t* __restrict__ leaf_parent) {
  __shared__ uint32_t wl[4], wr[4];
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  bool leader = false, removed = false;
  uint32_t j = 0;
  if (i < n) {
    j = nn[i];
    const bool mutual = nn[j] == i;
    leader = mutual && i < j; removed = mutual && i > j;
  }
  const uint64_t bl = __ballot(leader), br = __ballot(removed);
  const uint32_t wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63u) == 0) { wl[wave] = uint32_t(__popcll(bl)); wr[wave] = uint32_t(__popcll(br)); }
  __syncthreads();
  if (i >= n || removed) return;
  uint32_t lrank = S[blockIdx.x] + __builtin_amdgcn_mbcnt_hi(uint32_t(bl >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(bl), 0u));
  uint32_t rrank = S[nblk + blockIdx.x] - S[nblk] + __builtin_amdgcn_mbcnt_hi(uint32_t(br >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(br), 0u));
  for (uint32_t w = 0; w < wave; ++w) { lrank += wl[w]; rrank += wr[w]; }
  const uint32_t pos = i - rrank;
  const float4 alo = cl_in[2 * size_t(i)], ahi = cl_in[2 * size_t(i) + 1];
  if (!leader) { cl_out[2 * size_t(pos)] = alo; cl_out[2 * size_t(pos) + 1] = ahi; return; }
  const float4 blo = cl_in[2 * size_t(j)], bhi = cl_in[2 * size_t(j) + 1];
  const uint32_t node = next_node - 1u - lrank;
  const int la = __float_as_int(alo.w), lb = __float_as_int(blo.w);
  float4* nd = reinterpret_cast<float4*>(&nodes[node]);
  nd[0] = make_float4(alo.x, alo.y, alo.z, alo.w);                          // lo0 | link0
  nd[1] = make_float4(ahi.x, ahi.y, ahi.z, blo.w);                          // hi0 | link1
  nd[2] = make_float4(blo.x, blo.y, blo.z, __uint_as_float(0xFFFFFFFFu));   // lo1 | parent (set when the parent is made)
  const uint32_t below = __float_as_uint(ahi.w) + __float_as_uint(bhi.w) + 1u;
  nd[3] = make_float4(bhi.x, bhi.y, bhi.z, __uint_as_float(below));         // hi1 | internal nodes in this subtree
  if (la >= 0) nodes[la].parent = node; else leaf_parent[~la] = node;
  if (lb >= 0) nodes[lb].parent = node; else leaf_parent[~lb] = node;
  cl_out[2 * size_t(pos)] = make_float4(fminf(alo.x, blo.x), fminf(alo.y, blo.y), fminf(alo.z, blo.z), __int_as_float(int(node)));
  cl_out[2 * size_t(pos) + 1] = make_float4(fmaxf(ahi.x, bhi.x), fmaxf(ahi.y, bhi.y), fmaxf(ahi.z, bhi.z), __uint_as_float(below));
}

// Depth-first (pre-order) renumbering of the PLOC tree: a subtree becomes a contiguous index range and the first
// child sits right behind its parent, as in the Karras layout.  new index = number of nodes visited before x in
// pre-order = sum over the ancestors a of x: 1, plus the size of a's first subtree when x hangs under the second.
__global__ void k_preorder_index(uint32_t n_nodes, const mi_bvh_node* __restrict__ nodes, uint32_t* __restrict__ new_index) {
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= n_nodes) return;
  uint32_t idx = 0, cur = x, parent = nodes[x].parent;
  while (parent != 0xFFFFFFFFu) {
    const int l0 = nodes[parent].link0;
    idx += 1u;
    if (l0 != int(cur)) idx += l0 >= 0 ? nodes[l0].reserved : 0u;
    cur = parent; parent = nodes[parent].parent;
  }
  new_index[x] = idx;
}
__global__ void k_preorder_move(uint32_t n_nodes, const mi_bvh_node* __restrict__ src, const uint32_t* __restrict__ new_index,
                                mi_bvh_node* __restrict__ dst, uint32_t* __restrict__ leaf_parent) {
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= n_nodes) return;
  mi_bvh_node n = src[x];
  const uint32_t me = new_index[x];
  if (n.link0 >= 0) n.link0 = int(new_index[n.link0]); else leaf_parent[~n.link0] = me;
  if (n.link1 >= 0) n.link1 = int(new_index[n.link1]); else leaf_parent[~n.link1] = me;
  if (n.parent != 0xFFFFFFFFu) n.parent = new_index[n.parent];
  n.reserved = 0;
  dst[me] = n;
}

__global__ void k_depth(uint32_t n, const mi_bvh_node* __restrict__ nodes, const uint32_t* __restrict__ leaf_parent, uint32_t* max_depth) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t d = 1, parent = leaf_parent[i];
  while (parent != 0xFFFFFFFFu) { ++d; parent = nodes[parent].parent; }
  atomicMax(max_depth, d);
}

// ---- pair leaves (r02) ----
// A child that is a node over two triangles A, B becomes one leaf link naming both, so that every walk tests the two triangles under their
// common box without opening (and, for node records read from HBM, fetching) a node for them: C2 -1.4 of 8.2 node visits per closest-hit
// ray.  To name two triangles with one link and no extra state in the walking lane, the intersection / shading streams are emitted in an
// order in which B follows A: position' = Morton position, except that a partner moves up behind its A (a scan over 2 / 0 / 1 records per
// position).  Leaf link = ~position' with bit 30 cleared for a pair (layout.h).  Hits do not depend on any of this ((t, id) minimum /
// boolean); `plain` keeps the builder's links in Morton positions for mi_pt_bvh_download (the oracle's tree), sorted_tri stays Morton order.
__global__ void k_plain_links(uint32_t n_nodes, const mi_bvh_node* __restrict__ nodes, int2* __restrict__ plain) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n_nodes) plain[i] = make_int2(nodes[i].link0, nodes[i].link1);
}
__global__ void k_pair_init(uint32_t nt, uint32_t* __restrict__ emits, uint32_t* __restrict__ a_of) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i <= nt) emits[i] = i < nt ? 1u : 0u;
  if (i < nt) a_of[i] = 0xFFFFFFFFu;
}
__global__ void k_pair_mark(uint32_t n_nodes, const int2* __restrict__ plain, uint32_t* __restrict__ emits, uint32_t* __restrict__ a_of) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n_nodes) return;
  const int l[2] = {plain[i].x, plain[i].y};
  for (int c = 0; c < 2; ++c) {
    if (l[c] < 0) continue;
    const int2 ch = plain[l[c]];
    if (ch.x >= 0 || ch.y >= 0) continue;
    const uint32_t pa = uint32_t(~ch.x), pb = uint32_t(~ch.y);  // a leaf has one parent: one writer per position
    emits[pa] = 2u; emits[pb] = 0u; a_of[pb] = pa;
  }
}
__global__ void k_pair_order(uint32_t nt, const uint32_t* __restrict__ sorted_tri, const uint32_t* __restrict__ prefix, const uint32_t* __restrict__ a_of,
                             uint32_t* __restrict__ perm, uint32_t* __restrict__ sorted2) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= nt) return;
  const uint32_t a = a_of[i];
  const uint32_t q = a != 0xFFFFFFFFu ? prefix[a] + 1u : prefix[i];
  perm[i] = q;
  sorted2[q] = sorted_tri[i];
}
__global__ void k_pair_relink(uint32_t n_nodes, mi_bvh_node* __restrict__ nodes, const int2* __restrict__ plain, const uint32_t* __restrict__ perm) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n_nodes) return;
  int l[2] = {plain[i].x, plain[i].y};
  for (int c = 0; c < 2; ++c) {
    if (l[c] < 0) { l[c] = int(~perm[uint32_t(~l[c])]); continue; }
    const int2 ch = plain[l[c]];
    if (ch.x < 0 && ch.y < 0) l[c] = int(~(perm[uint32_t(~ch.x)] | kLeafPairBit));
  }
  nodes[i].link0 = l[0]; nodes[i].link1 = l[1];
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

// 16-bit grid coordinates of the child boxes, rounded outward (plus 1/16 cell: the float grid transform of the box corners and of
// the ray is good to ~0.01 cell) so that rounding can never cut into the true box.  The slab test adds its own per-ray slack.
__device__ __forceinline__ uint32_t q_lo(float v, float lo, float inv_step) {
  float q = floorf((v - lo) * inv_step - 0.0625f);  // the float grid transform is good to ~0.01 cell at 65 535: 1/16 cell of slack
  q = q < 0.0f ? 0.0f : (q > 65535.0f ? 65535.0f : q);
  return uint32_t(q);
}
__device__ __forceinline__ uint32_t q_hi(float v, float lo, float inv_step) {
  float q = ceilf((v - lo) * inv_step + 0.0625f);
  q = q < 0.0f ? 0.0f : (q > 65535.0f ? 65535.0f : q);
  return uint32_t(q);
}
__global__ void k_quantize(uint32_t n_nodes, const mi_bvh_node* __restrict__ nodes, uint4* __restrict__ qnodes, float lx, float ly, float lz,
                           float ix, float iy, float iz, uint32_t pad_cells) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  const mi_bvh_node n = nodes[i];
  // each child: centre and half extent of its quantised box (+ 1 cell), the encoding of the wide records (k_collapse4; wide_child_test, pt_device.h)
  auto enc = [&](const float* lo, const float* hi, int link) {
    uint32_t c[3], e[3];
    const float glo[3] = {lx, ly, lz}, gis[3] = {ix, iy, iz};
    for (int k = 0; k < 3; ++k) {
      const uint32_t ql = q_lo(lo[k], glo[k], gis[k]), qh = q_hi(hi[k], glo[k], gis[k]);
      c[k] = (ql + qh) >> 1;
      const uint32_t ext = (qh - c[k] > c[k] - ql ? qh - c[k] : c[k] - ql) + pad_cells;
      e[k] = ext > 65535u ? 65535u : ext;
    }
    return make_uint4(c[0] | (c[1] << 16), c[2] | (e[0] << 16), e[1] | (e[2] << 16), uint32_t(link));
  };
  const uint4 a = enc(n.lo0, n.hi0, n.link0), b = enc(n.lo1, n.hi1, n.link1);
  qnodes[2 * size_t(i)] = a;
  qnodes[2 * size_t(i) + 1] = b;
}

// wide nodes: one record per even-depth BVH2 node = its children, internal ones replaced by their own children.
// k_even_depth flags those nodes; an exclusive scan of the flags numbers the records contiguously (pre-order of the BVH2
// numbering is kept), so that every fetched cache line holds two live records.
__global__ void k_even_depth(uint32_t n_nodes, const mi_bvh_node* __restrict__ nodes, uint32_t* __restrict__ flag) {
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= n_nodes) return;
  uint32_t depth = 0;
  for (uint32_t p = nodes[x].parent; p != 0xFFFFFFFFu; p = nodes[p].parent) ++depth;
  flag[x] = (depth & 1u) ^ 1u;
}
__global__ void k_collapse4(uint32_t n_nodes, const mi_bvh_node* __restrict__ nodes, const uint32_t* __restrict__ flag_scan, uint4* __restrict__ q4,
                            float lx, float ly, float lz, float ix, float iy, float iz, uint32_t pad_cells) {
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= n_nodes) return;
  const uint32_t me = flag_scan[x];
  const bool even = (x + 1 < n_nodes ? flag_scan[x + 1] : flag_scan[n_nodes]) != me;  // exclusive scan: the flag is the difference
  if (!even) return;
  uint4 out[4];
  // an unused slot (r04): a box nothing enters in practice — the single point at the far corner of the grid, which overhangs the scene box by two cells, half
  // extent 0 — and, should a ray pass exactly through it, a link that is harmless to follow: the leaf of the triangle at position 0 (a second test of a
  // triangle changes neither the (t, id) minimum nor an occlusion).  The walks no longer test every child's link against a sentinel (4 v_cmp + 4 s_and per visit).
  for (int k = 0; k < 4; ++k) out[k] = make_uint4(0xFFFFFFFFu, 0x0000FFFFu, 0u, 0xFFFFFFFFu);
  const mi_bvh_node n = nodes[x];
  int k = 0;
  auto put = [&](const float* lo, const float* hi, int link) {
    // centre and half extent of the quantised box (wide_child_test, pt_device.h): c = (lo + hi) / 2 rounded down, e = what covers both ends + 1 cell
    uint4 a;
    uint32_t c[3], e[3];
    const float glo[3] = {lx, ly, lz}, gis[3] = {ix, iy, iz};
    for (int k3 = 0; k3 < 3; ++k3) {
      const uint32_t ql = q_lo(lo[k3], glo[k3], gis[k3]), qh = q_hi(hi[k3], glo[k3], gis[k3]);
      c[k3] = (ql + qh) >> 1;
      const uint32_t ext = (qh - c[k3] > c[k3] - ql ? qh - c[k3] : c[k3] - ql) + pad_cells;  // 1 unless a camera is far outside the grid (mi_pt_create)
      e[k3] = ext > 65535u ? 65535u : ext;
    }
    a.x = c[0] | (c[1] << 16);
    a.y = c[2] | (e[0] << 16);
    a.z = e[1] | (e[2] << 16);
    a.w = link >= 0 ? flag_scan[link] : uint32_t(link);  // internal grandchildren are even-depth nodes: their record number
    out[k++] = a;
  };
  const int links[2] = {n.link0, n.link1};
  for (int c = 0; c < 2; ++c) {
    if (links[c] < 0) {
      if (c == 0) put(n.lo0, n.hi0, links[c]); else put(n.lo1, n.hi1, links[c]);
    } else {
      const mi_bvh_node m = nodes[links[c]];
      put(m.lo0, m.hi0, m.link0);
      put(m.lo1, m.hi1, m.link1);
    }
  }
  for (int k2 = 0; k2 < 4; ++k2) q4[4 * size_t(me) + k2] = out[k2];
}

// r04 — area-guided collapse (the usual way a BVH4 is made from a BVH2; the strict two-level collapse above stays selectable, MI_PT_COLLAPSE=levels): a record
// starts as the two children of its BVH2 node and, while it has fewer than four and one of them is internal, replaces the internal child of LARGEST surface
// area by that child's two children.  Large boxes are the ones rays enter most often: opening them inside the record saves the visit of a record of their
// own (lab on real segments, tests/lab: -6 .. -10 % wide-record visits per ray against the two-level collapse).  Which BVH2 nodes head a record is no longer a
// matter of depth parity, so they are marked top-down, one level of records per pass; wide_kids() is the one definition both passes use.
__device__ __forceinline__ float box_area3(const float* lo, const float* hi) {
  const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
  return dx * dy + dy * dz + dz * dx;
}
struct WideKids { float lo[4][3], hi[4][3]; int link[4]; int n; };
__device__ __forceinline__ void wide_kids(const mi_bvh_node* __restrict__ nodes, uint32_t x, WideKids& w) {
  const mi_bvh_node n = nodes[x];
  for (int a = 0; a < 3; ++a) { w.lo[0][a] = n.lo0[a]; w.hi[0][a] = n.hi0[a]; w.lo[1][a] = n.lo1[a]; w.hi[1][a] = n.hi1[a]; }
  w.link[0] = n.link0; w.link[1] = n.link1; w.n = 2;
  while (w.n < 4) {
    int best = -1; float best_area = -1.0f;
    for (int k = 0; k < w.n; ++k)
      if (w.link[k] >= 0) { const float a = box_area3(w.lo[k], w.hi[k]); if (a > best_area) { best_area = a; best = k; } }
    if (best < 0) break;
    const mi_bvh_node m = nodes[w.link[best]];
    for (int a = 0; a < 3; ++a) { w.lo[best][a] = m.lo0[a]; w.hi[best][a] = m.hi0[a]; w.lo[w.n][a] = m.lo1[a]; w.hi[w.n][a] = m.hi1[a]; }
    w.link[best] = m.link0; w.link[w.n] = m.link1; ++w.n;
  }
}
// state: 0 = not (yet) the head of a record, p + 1 = head to be expanded by pass p, kHeadDone = head, children marked.  pending[x] = entries a walk can have on its stack
// when it enters record x (every visit pushes at most its other children): the maximum over all records + 3 is the exact capacity the wide walk needs.
constexpr uint32_t kHeadDone = 0x80000000u;
__global__ void k_mark_heads(uint32_t n_nodes, const mi_bvh_node* __restrict__ nodes, uint32_t* __restrict__ state, uint32_t* __restrict__ pending, uint32_t* __restrict__ max_need,
                             uint32_t pass) {
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= n_nodes || state[x] != pass + 1u) return;  // heads found by pass p - 1 carry the tag p + 1: strictly one level of records per launch
  WideKids w; wide_kids(nodes, x, w);
  const uint32_t below = pending[x] + uint32_t(w.n - 1);
  for (int k = 0; k < w.n; ++k) if (w.link[k] >= 0) { pending[w.link[k]] = below; state[w.link[k]] = pass + 2u; }
  atomicMax(max_need, below);
  state[x] = kHeadDone;
}
__global__ void k_heads_to_flags(uint32_t n_nodes, const uint32_t* __restrict__ state, uint32_t* __restrict__ flag) {
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x < n_nodes) flag[x] = state[x] == kHeadDone ? 1u : 0u;
}
__global__ void k_collapse4_area(uint32_t n_nodes, const mi_bvh_node* __restrict__ nodes, const uint32_t* __restrict__ flag_scan, uint4* __restrict__ q4,
                                 float lx, float ly, float lz, float ix, float iy, float iz, uint32_t pad_cells) {
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= n_nodes) return;
  const uint32_t me = flag_scan[x];
  if ((x + 1 < n_nodes ? flag_scan[x + 1] : flag_scan[n_nodes]) == me) return;  // exclusive scan: the flag is the difference
  WideKids w; wide_kids(nodes, x, w);
  const float glo[3] = {lx, ly, lz}, gis[3] = {ix, iy, iz};
  for (int k = 0; k < 4; ++k) {
    uint4 a = make_uint4(0xFFFFFFFFu, 0x0000FFFFu, 0u, 0xFFFFFFFFu);  // unused slot: see k_collapse4
    if (k < w.n) {
      uint32_t c[3], e[3];
      for (int k3 = 0; k3 < 3; ++k3) {
        const uint32_t ql = q_lo(w.lo[k][k3], glo[k3], gis[k3]), qh = q_hi(w.hi[k][k3], glo[k3], gis[k3]);
        c[k3] = (ql + qh) >> 1;
        const uint32_t ext = (qh - c[k3] > c[k3] - ql ? qh - c[k3] : c[k3] - ql) + pad_cells;
        e[k3] = ext > 65535u ? 65535u : ext;
      }
      a.x = c[0] | (c[1] << 16); a.y = c[2] | (e[0] << 16); a.z = e[1] | (e[2] << 16);
      a.w = w.link[k] >= 0 ? flag_scan[w.link[k]] : uint32_t(w.link[k]);
    }
    q4[4 * size_t(me) + k] = a;
  }
}

#define BUILD_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

}  // namespace

// Device inputs: pos [nv][3], tan [nv][9], idx [nt][3], tri_material [nt].
// Outputs: nodes / tri_isect / tri_shade (sections of the scene blob), sorted_tri [nt],
// morton [nt], scene bounds, depth.  `scratch` allocations are made and freed here (one-off).
// builder: 1 = PLOC (default), 0 = Karras LBVH.
hipError_t build_bvh(int builder, uint32_t nt, const float* pos, const float* tan, const uint32_t* idx, const uint32_t* tri_material,
                     mi_bvh_node* nodes, float4* tri_isect, float4* tri_shade, uint32_t* sorted_tri, uint64_t* morton,
                     float scene_lo[3], float scene_hi[3], uint32_t* max_depth_out, float* build_ms, uint32_t* rounds_out, int2* plain_links,
                     bool pairs, hipStream_t stream) {
  const uint32_t nblk_sort = (nt + kSortTile - 1) / kSortTile;
  const uint32_t g256 = (nt + 255) / 256;
  struct Scratch {
    void* p[32]; int n = 0;
    hipError_t get(void** out, size_t bytes) { hipError_t e = hipMalloc(out, bytes ? bytes : 4); if (e == hipSuccess) p[n++] = *out; return e; }
    ~Scratch() { for (int i = 0; i < n; ++i) hipFree(p[i]); }
  } scratch;
  float *tri_lo = nullptr, *tri_hi = nullptr;
  uint64_t* keys_b = nullptr;
  uint32_t *scene_ord = nullptr, *vals_b = nullptr, *hist = nullptr, *leaf_parent = nullptr, *visit = nullptr, *depth = nullptr;
  BUILD_CHECK(scratch.get(reinterpret_cast<void**>(&tri_lo), sizeof(float) * 3 * size_t(nt)));
  BUILD_CHECK(scratch.get(reinterpret_cast<void**>(&tri_hi), sizeof(float) * 3 * size_t(nt)));
  BUILD_CHECK(scratch.get(reinterpret_cast<void**>(&scene_ord), sizeof(uint32_t) * 8));
  BUILD_CHECK(scratch.get(reinterpret_cast<void**>(&keys_b), sizeof(uint64_t) * size_t(nt)));
  BUILD_CHECK(scratch.get(reinterpret_cast<void**>(&vals_b), sizeof(uint32_t) * size_t(nt)));
  BUILD_CHECK(scratch.get(reinterpret_cast<void**>(&hist), sizeof(uint32_t) * 256 * size_t(nblk_sort)));
  BUILD_CHECK(scratch.get(reinterpret_cast<void**>(&leaf_parent), sizeof(uint32_t) * size_t(nt)));
  BUILD_CHECK(scratch.get(reinterpret_cast<void**>(&depth), sizeof(uint32_t)));
  uint32_t* scan_tmp = nullptr;  // tile sums of scan_exclusive: the longest scanned array is max(256 x sort blocks, nt + 1)
  BUILD_CHECK(scratch.get(reinterpret_cast<void**>(&scan_tmp), sizeof(uint32_t) * ((size_t(nt) + 256u * size_t(nblk_sort)) / kScanTile + 8)));
  const uint32_t ord_init[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u};
  BUILD_CHECK(hipMemcpyAsync(scene_ord, ord_init, sizeof ord_init, hipMemcpyHostToDevice, stream));
  BUILD_CHECK(hipMemsetAsync(depth, 0, sizeof(uint32_t), stream));
  BUILD_CHECK(hipMemsetAsync(leaf_parent, 0xFF, sizeof(uint32_t) * size_t(nt), stream));
  hipEvent_t ev0, ev1;
  BUILD_CHECK(hipEventCreate(&ev0)); BUILD_CHECK(hipEventCreate(&ev1));
  struct Ev { hipEvent_t a, b; ~Ev() { hipEventDestroy(a); hipEventDestroy(b); } } evguard{ev0, ev1};
  BUILD_CHECK(hipEventRecord(ev0, stream));

  hipLaunchKernelGGL(k_tri_bounds, dim3(g256), dim3(256), 0, stream, nt, pos, idx, tri_lo, tri_hi, scene_ord);
  uint64_t* keys_a = morton;      // ping
  uint32_t* vals_a = sorted_tri;  // ping
  hipLaunchKernelGGL(k_morton, dim3(g256), dim3(256), 0, stream, nt, tri_lo, tri_hi, scene_ord, keys_a, vals_a);
  uint64_t *kin = keys_a, *kout = keys_b;
  uint32_t *vin = vals_a, *vout = vals_b;
  for (uint32_t pass = 0; pass < 8; ++pass) {  // 63-bit codes: 8 passes of 8 bits; ends back in (morton, sorted_tri)
    hipLaunchKernelGGL(k_hist, dim3(nblk_sort), dim3(256), 0, stream, kin, nt, pass * 8, hist, nblk_sort);
    scan_exclusive(hist, 256u * nblk_sort, scan_tmp, stream);
    hipLaunchKernelGGL(k_scatter, dim3(nblk_sort), dim3(256), 0, stream, kin, vin, kout, vout, nt, pass * 8, hist, nblk_sort);
    uint64_t* t = kin; kin = kout; kout = t;
    uint32_t* u = vin; vin = vout; vout = u;
  }
  uint32_t rounds = 0;
  if (nt > 1 && builder == 0) {
    BUILD_CHECK(scratch.get(reinterpret_cast<void**>(&visit), sizeof(uint32_t) * size_t(nt)));
    BUILD_CHECK(hipMemsetAsync(visit, 0, sizeof(uint32_t) * size_t(nt), stream));
    hipLaunchKernelGGL(k_karras, dim3((nt - 1 + 255) / 256), dim3(256), 0, stream, int(nt), morton, sorted_tri, nodes, leaf_parent);
    hipLaunchKernelGGL(k_refit, dim3(g256), dim3(256), 0, stream, nt, sorted_tri, tri_lo, tri_hi, nodes, leaf_parent, visit);
  } else if (nt > 1) {
    float4 *cl_a = nullptr, *cl_b = nullptr; uint32_t *nn = nullptr, *counts = nullptr;
    BUILD_CHECK(scratch.get(reinterpret_cast<void**>(&cl_a), sizeof(float4) * 2 * size_t(nt)));
    BUILD_CHECK(scratch.get(reinterpret_cast<void**>(&cl_b), sizeof(float4) * 2 * size_t(nt)));
    BUILD_CHECK(scratch.get(reinterpret_cast<void**>(&nn), sizeof(uint32_t) * size_t(nt)));
    BUILD_CHECK(scratch.get(reinterpret_cast<void**>(&counts), sizeof(uint32_t) * (2 * size_t(g256) + 1)));
    hipLaunchKernelGGL(k_ploc_init, dim3(g256), dim3(256), 0, stream, nt, sorted_tri, tri_lo, tri_hi, cl_a);
    uint32_t n_cur = nt, next_node = nt - 1;
    while (n_cur > 1) {
      const uint32_t nblk = (n_cur + 255) / 256;
      hipLaunchKernelGGL(k_ploc_nn, dim3(nblk), dim3(256), 0, stream, n_cur, cl_a, nn);
      hipLaunchKernelGGL(k_ploc_count, dim3(nblk), dim3(256), 0, stream, n_cur, nn, counts, nblk);
      scan_exclusive(counts, 2 * nblk + 1, scan_tmp, stream);
      hipLaunchKernelGGL(k_ploc_apply, dim3(nblk), dim3(256), 0, stream, n_cur, cl_a, nn, counts, nblk, next_node, cl_b, nodes, leaf_parent);
      uint32_t merged = 0;
      BUILD_CHECK(hipMemcpyAsync(&merged, counts + nblk, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
      BUILD_CHECK(hipStreamSynchronize(stream));
      if (merged == 0 || merged > n_cur / 2) return hipErrorAssert;  // cannot happen for finite boxes: the smallest pair is always mutual
      n_cur -= merged; next_node -= merged;
      float4* t = cl_a; cl_a = cl_b; cl_b = t;
      ++rounds;
    }
    if (next_node != 0) return hipErrorAssert;
    {
      mi_bvh_node* tmp = nullptr; uint32_t* new_index = nullptr;
      BUILD_CHECK(scratch.get(reinterpret_cast<void**>(&tmp), sizeof(mi_bvh_node) * size_t(nt - 1)));
      BUILD_CHECK(scratch.get(reinterpret_cast<void**>(&new_index), sizeof(uint32_t) * size_t(nt - 1)));
      BUILD_CHECK(hipMemcpyAsync(tmp, nodes, sizeof(mi_bvh_node) * size_t(nt - 1), hipMemcpyDeviceToDevice, stream));
      hipLaunchKernelGGL(k_preorder_index, dim3((nt - 1 + 255) / 256), dim3(256), 0, stream, nt - 1, tmp, new_index);
      hipLaunchKernelGGL(k_preorder_move, dim3((nt - 1 + 255) / 256), dim3(256), 0, stream, nt - 1, tmp, new_index, nodes, leaf_parent);
    }
  }
  hipLaunchKernelGGL(k_depth, dim3(g256), dim3(256), 0, stream, nt, nodes, leaf_parent, depth);
  const uint32_t* emit_order = sorted_tri;
  if (nt > 1) {
    const uint32_t n_nodes = nt - 1, gn = (n_nodes + 255) / 256;
    hipLaunchKernelGGL(k_plain_links, dim3(gn), dim3(256), 0, stream, n_nodes, nodes, plain_links);
    if (pairs && nt < kLeafPairBit) {
      uint32_t *emits = nullptr, *a_of = nullptr, *perm = nullptr, *sorted2 = nullptr;
      BUILD_CHECK(scratch.get(reinterpret_cast<void**>(&emits), sizeof(uint32_t) * (size_t(nt) + 1)));
      BUILD_CHECK(scratch.get(reinterpret_cast<void**>(&a_of), sizeof(uint32_t) * size_t(nt)));
      BUILD_CHECK(scratch.get(reinterpret_cast<void**>(&perm), sizeof(uint32_t) * size_t(nt)));
      BUILD_CHECK(scratch.get(reinterpret_cast<void**>(&sorted2), sizeof(uint32_t) * size_t(nt)));
      hipLaunchKernelGGL(k_pair_init, dim3((nt + 256) / 256), dim3(256), 0, stream, nt, emits, a_of);
      hipLaunchKernelGGL(k_pair_mark, dim3(gn), dim3(256), 0, stream, n_nodes, plain_links, emits, a_of);
      scan_exclusive(emits, nt + 1, scan_tmp, stream);
      hipLaunchKernelGGL(k_pair_order, dim3(g256), dim3(256), 0, stream, nt, sorted_tri, emits, a_of, perm, sorted2);
      hipLaunchKernelGGL(k_pair_relink, dim3(gn), dim3(256), 0, stream, n_nodes, nodes, plain_links, perm);
      emit_order = sorted2;
    }
  }
  hipLaunchKernelGGL(k_emit, dim3(g256), dim3(256), 0, stream, nt, emit_order, pos, tan, idx, tri_material, tri_isect, tri_shade);
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
  if (rounds_out) *rounds_out = rounds;
  return hipSuccess;
}

// centre / half-extent copy of the full-precision nodes (ce_box_test<SLACK>, pt_device.h): half extent from the rounded centre, + 2^-20 relative
// + 2^-21 of the box's own largest coordinate (the roundings of c * inv and e * |inv|; the origin's share is the test's per-ray slack)
__global__ void k_ce_nodes(uint32_t n_nodes, const float4* __restrict__ nodes, float4* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes * 2u) return;
  const uint32_t n = i >> 1, c = i & 1u;
  const float4 lo = nodes[4u * n + 2u * c], hi = nodes[4u * n + 2u * c + 1u];
  const float cx = (lo.x + hi.x) * 0.5f, cy = (lo.y + hi.y) * 0.5f, cz = (lo.z + hi.z) * 0.5f;
  const float px = fmaxf(fabsf(lo.x), fabsf(hi.x)) * 0x1p-21f, py = fmaxf(fabsf(lo.y), fabsf(hi.y)) * 0x1p-21f, pz = fmaxf(fabsf(lo.z), fabsf(hi.z)) * 0x1p-21f;
  const float ex = fmaf(fmaxf(hi.x - cx, cx - lo.x), 1.000001f, px), ey = fmaf(fmaxf(hi.y - cy, cy - lo.y), 1.000001f, py), ez = fmaf(fmaxf(hi.z - cz, cz - lo.z), 1.000001f, pz);
  out[4u * n + 2u * c] = make_float4(cx, cy, cz, lo.w);
  out[4u * n + 2u * c + 1u] = make_float4(ex, ey, ez, hi.w);
}
hipError_t ce_nodes(uint32_t n_nodes, const float4* nodes, float4* out, hipStream_t stream) {
  if (n_nodes == 0) return hipSuccess;
  hipLaunchKernelGGL(k_ce_nodes, dim3((2u * n_nodes + 255u) / 256u), dim3(256), 0, stream, n_nodes, nodes, out);
  return hipGetLastError();
}

hipError_t quantize_nodes(uint32_t n_nodes, const mi_bvh_node* nodes, uint4* qnodes, uint4* qnodes4, const float lo[3], const float inv_step[3],
                          uint32_t pad_cells, uint32_t depth, bool area_collapse, uint32_t* wide_stack_need, hipStream_t stream) {
  if (wide_stack_need) *wide_stack_need = 0;
  if (n_nodes == 0) return hipSuccess;
  uint32_t* flag = nullptr;
  // flags + scan scratch | head state | pending entries | maximum
  const size_t n_flag = size_t(n_nodes) + 1 + size_t(n_nodes) / kScanTile + 8;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&flag), sizeof(uint32_t) * (n_flag + 2 * size_t(n_nodes) + 8));
  if (e != hipSuccess) return e;
  uint32_t* state = flag + n_flag, *pending = state + n_nodes, *max_need = pending + n_nodes;
  hipMemsetAsync(flag + n_nodes, 0, sizeof(uint32_t), stream);
  const dim3 grid((n_nodes + 255) / 256), block(256);
  if (area_collapse) {
    hipMemsetAsync(state, 0, sizeof(uint32_t) * (2 * size_t(n_nodes) + 8), stream);
    const uint32_t one = 1u;
    hipMemcpyAsync(state, &one, sizeof one, hipMemcpyHostToDevice, stream);  // the root heads the first record
    for (uint32_t pass = 0; pass < depth + 1u; ++pass)  // a record covers at least one BVH2 level: `depth` passes reach every head
      hipLaunchKernelGGL(k_mark_heads, grid, block, 0, stream, n_nodes, nodes, state, pending, max_need, pass);
    hipLaunchKernelGGL(k_heads_to_flags, grid, block, 0, stream, n_nodes, state, flag);
    scan_exclusive(flag, n_nodes + 1, flag + n_nodes + 1, stream);
    hipLaunchKernelGGL(k_collapse4_area, grid, block, 0, stream, n_nodes, nodes, flag, qnodes4, lo[0], lo[1], lo[2], inv_step[0], inv_step[1], inv_step[2], pad_cells);
    uint32_t need = 0;
    e = hipMemcpyAsync(&need, max_need, sizeof need, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (wide_stack_need) *wide_stack_need = need;
  } else {
    hipLaunchKernelGGL(k_even_depth, grid, block, 0, stream, n_nodes, nodes, flag);
    scan_exclusive(flag, n_nodes + 1, flag + n_nodes + 1, stream);
    hipLaunchKernelGGL(k_collapse4, grid, block, 0, stream, n_nodes, nodes, flag, qnodes4, lo[0], lo[1], lo[2], inv_step[0],
                       inv_step[1], inv_step[2], pad_cells);
    e = hipStreamSynchronize(stream);
    if (wide_stack_need) *wide_stack_need = 3u * ((depth > 1 ? depth - 1u : 1u) + 1u) / 2u;  // up to three children pending per two binary levels
  }
  hipFree(flag);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_quantize, grid, block, 0, stream, n_nodes, nodes, qnodes, lo[0], lo[1], lo[2], inv_step[0],
                     inv_step[1], inv_step[2], pad_cells);
  return hipGetLastError();
}

}  // namespace mi
