// mi_pt_api.hip — host side of the C ABI (include/mi_pt.h): handle life cycle, scene upload,
// LBVH build, render calls, parity hooks.  Everything that computes runs on the GPU; there is no
// CPU fallback — without a HIP device every compute entry point fails with MI_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <chrono>
#include <functional>
#include <thread>
#include <vector>

#include "device/launch.h"
#include "device/wavefront.h"
#include "device/bpt.h"
#include "scene_host.hpp"

using mi::fail;

#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess)                                                                               \
      return fail(e_ == hipErrorOutOfMemory ? MI_ERR_OUT_OF_MEMORY : MI_ERR_NO_DEVICE,                  \
                  std::string(__FILE__) + ":" + std::to_string(__LINE__) + ": " #expr " failed: " + hipGetErrorString(e_)); \
  } while (0)

struct mi_pt_handle {
  int device = 0;
  mi::SceneData scene;
  mi_pt_params params{};
  float4* blob = nullptr;
  uint4* qnodes = nullptr;
  uint4* qnodes4 = nullptr;
  float4* ce_nodes = nullptr;       // centre / half-extent copy of the full-precision nodes (kernels that read them from HBM: scenes the 16-bit grid is too coarse for)
  float* flat_table = nullptr; uint32_t flat_k = 0, flat_k_mesh = 0; float flat_amax = 0.0f;  // flat_amax: largest |coordinate| the padding of the table's boxes covers  // flat leaf list of small scenes (traverse_flat, pt_device.h): leaf boxes + links, mesh leaves first
  int2* plain_links = nullptr;    // the builder's links of every node in Morton positions (mi_pt_bvh_download); the blob's nodes carry pair leaves
  void* rccl_comm = nullptr; uint32_t reduce_rank = 0, reduce_world = 0;  // mi_pt_reduce_init: this handle's RCCL communicator (one process per GPU)
  uint32_t quant_pad_cells = 1;    // padding of the quantised child boxes in grid cells (1 unless a camera is > 2^20 cells away: mi_pt_create)
  bool float_nodes = false;        // HBM-resident kernels read the full-precision nodes: the 16-bit grid is too coarse for this scene
  bool wide_nodes = false;         // the PT megakernel walks the wide nodes (every HBM-resident scene the 16-bit grid is fine enough for; MI_PT_WIDE_NODES=0/1 overrides)
  bool wide_large = false;         // scenes of >= 100 000 triangles: the BPT kernels (two per-lane loops) walk the wide nodes only there
  bool stack_fits_lds = false;     // depth - 1 <= info.stack_entries: the binary walk needs no spill entries
  uint32_t stack_entries_hbm = 0;  // LDS rows of the traversal stack for kernels that read the scene from HBM (wide walk)
  mi::SceneView sv{};
  uint32_t* d_sorted_tri = nullptr;
  uint64_t* d_morton = nullptr;
  mi_bvh_info info{};
  int kernel_choice = MI_PT_KERNEL_AUTO;
  uint32_t shard_rank = 0, shard_world = 1;  // mi_pt_set_tile_shard
  bool instrumented = false;
  bool lds_fits = false;
  double* partial = nullptr; size_t partial_bytes = 0;
  float* d_rgbn = nullptr; size_t rgbn_bytes = 0;
  float* h_stage = nullptr; size_t h_stage_bytes = 0;  // pinned host copy of d_rgbn (mi_pt_render_multi, host-side merge)
  float* d_merge = nullptr; size_t merge_bytes = 0;    // the merged frame on the first handle's device (mi_pt_render_multi, device-side merge)
  hipEvent_t ev_multi = nullptr;                        // this handle's render of a mi_pt_render_multi call is complete
  unsigned long long* d_counters = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
  // wavefront pipeline: one arena for the per-slot arrays
  char* wf_arena = nullptr; size_t wf_arena_bytes = 0;
  uint32_t wf_iterations = 0;
  // BPT: light sub-path slab, eye / light images of a frame
  float4* bpt_slab = nullptr; size_t bpt_slab_bytes = 0;
  char* bpt_arena = nullptr; size_t bpt_arena_bytes = 0;      // staged form: slabs, emission terms, path info, item offsets
  float4* bpt_values = nullptr; size_t bpt_values_bytes = 0;  // staged form: connection item values
  // launches in flight (mi_bpt_render, r04): each flight has a stream, the event of its last launch, its connection values and two pinned host words
  // (item count, overflow count) of its launch in flight
  static constexpr uint32_t kBptFlights = 4;
  struct BptFlight { hipStream_t stream = nullptr; hipEvent_t done = nullptr; float4* values = nullptr; size_t values_bytes = 0; };
  BptFlight bpt_flight[kBptFlights];
  unsigned long long* bpt_pinned = nullptr;  // [kBptFlights][8]: item count, overflow count, counters[0 .. 3] as the launch's trace saw them
  uint32_t* bpt_aside = nullptr; size_t bpt_aside_bytes = 0;  // [0] count, [16 ..] indices of the batch's paths set aside for the launch at 1024 vertices
  hipEvent_t bpt_fork = nullptr;             // h->stream -> the flights' streams
  float* bpt_eye = nullptr; size_t bpt_eye_bytes = 0;
  double* bpt_light = nullptr; size_t bpt_light_bytes = 0;
  float sphere[4] = {0, 0, 0, 0};
  float bpt_rays_per_path = -1.0f;  // closest-hit rays per BPT path measured on this handle's finished launches (< 0: none yet): picks the tracing stage's form for small scenes
  uint32_t bpt_step_rounds = 0;  // rounds of the last BPT launch's tracing stage as uniform steps (0: the per-lane form ran)
  float sky_horizon[3] = {0, 0, 0}, sky_zenith[3] = {0, 0, 0};
  // frames in flight (mi_pt_render_frames_async / mi_pt_render_async / mi_pt_wait): a batch = the frames of ONE launch; per batch slot the
  // frames' device framebuffers (contiguous), their pinned host copies, counters, partial sums and a stream of its own
  struct BatchSlot {
    float* d_rgbn = nullptr; size_t d_bytes = 0;
    float* h_rgbn = nullptr; size_t h_bytes = 0;
    unsigned long long* d_counters = nullptr; unsigned long long* h_counters = nullptr;
    hipStream_t stream = nullptr;
    double* partial = nullptr; size_t partial_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev_copied = nullptr;
    uint64_t first_ticket = 0; uint32_t n_frames = 0, pending_mask = 0;
    uint32_t width = 0, height = 0; mi_window win{};
    bool launched = false, per_frame_counts = false, synced = false;
  } batches[MI_PT_BATCHES_IN_FLIGHT];
  uint32_t next_batch = 0;
  uint64_t next_ticket = 1;
  mi_pt_launch_info last{};
};

// mi_pt_render_multi, device-side merge: device 0 assembles the frame from the owners' framebuffers — every 32x32 tile of the window (Technique.cpp:167)
// read from the device that rendered it, over xGMI when the owners are peers — and only the merged frame crosses PCIe.  A copy, not a sum: the
// merged frame is bit-identical to a one-device render.  `master merge` (Options.cpp:1340-1409) adds whole images on the host; here nothing overlaps.
constexpr uint32_t kMaxMergeSources = 16;
struct MergeSources { const float4* fb[kMaxMergeSources]; };
__global__ __launch_bounds__(256) void k_merge_tiles(MergeSources src, uint32_t n, float4* __restrict__ out, uint32_t width, uint32_t height,
                                                    uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint32_t mtx) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= width * height) return;
  const uint32_t y = i / width, x = i - y * width;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (x >= x0 && x < x0 + w && y >= y0 && y < y0 + h) {
    const uint32_t owner = (((y - y0) >> 5) * mtx + ((x - x0) >> 5)) % n;
    const float4* p = src.fb[0];
#pragma unroll
    for (uint32_t k = 1; k < kMaxMergeSources; ++k) if (owner == k) p = src.fb[k];
    v = p[i];
  }
  out[i] = v;
}

namespace {

thread_local int g_last_multi_merge = -1;  // mi_pt_last_multi_merge(): 1 = the last mi_pt_render_multi of this thread merged on the device, 0 = on the host

constexpr uint32_t kFlatLeavesDefault = 24;  // flat leaf list by default up to this many leaf links (the uniform box loop costs ~14 VALU per leaf and ray)
constexpr uint32_t kFlatLeavesMin = 8;       // ... and from this many on: a tree over fewer leaves hardly diverges (profiles/r03/flat_vs_tree_small_scenes.txt: -3..-9 % at 2-7 leaves)
constexpr size_t kLdsSceneLimit = 52 * 1024;  // LDS bytes per workgroup of the LDS-resident kernels (scene copy + stack + sums): 3 workgroups per CU (160 KB / 3, allocation granules);
                                              // r02, tests/tools/lds_limit.py: 114 triangles (51 KB) LDS 11 107 vs HBM 9 566 Msamples/s, 144 triangles (63 KB, 2 per CU) 7 763 vs 8 763

template <class T> int upload(T** dst, const void* src, size_t bytes) {
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(dst), bytes ? bytes : 16));
  if (bytes) HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
  return MI_OK;
}

float l1(const float* v) { return std::fabs(v[0]) + std::fabs(v[1]) + std::fabs(v[2]); }

int ensure(void** p, size_t* have, size_t need) {
  if (*have >= need) return MI_OK;
  const bool debug = std::getenv("MI_BPT_DEBUG") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  if (*p) hipFree(*p);
  const auto t1 = std::chrono::steady_clock::now();
  *p = nullptr; *have = 0;
  HIP_TRY(hipMalloc(p, need));
  if (debug) {
    const auto t2 = std::chrono::steady_clock::now();
    const double f = std::chrono::duration<double, std::milli>(t1 - t0).count(), m = std::chrono::duration<double, std::milli>(t2 - t1).count();
    if (f + m > 20.0) std::fprintf(stderr, "[mi_pt] ensure(%zu bytes): hipFree %.1f ms, hipMalloc %.1f ms\n", need, f, m);
  }
  *have = need;
  return MI_OK;
}

bool use_lds_scene(const mi_pt_handle* h) {
  if (h->kernel_choice == MI_PT_KERNEL_MEGA_LDS) return h->lds_fits;
  if (h->kernel_choice == MI_PT_KERNEL_MEGA_GLOBAL) return false;
  return h->lds_fits;
}

// Carves the wavefront state out of the handle's arena (grown on demand).
int wf_prepare(mi_pt_handle* h, uint32_t P, mi::WfState& w) {
  const size_t need = size_t(P) * mi::kWfBytesPerSlot + 16 * 256 + 4096;
  int rc = ensure(reinterpret_cast<void**>(&h->wf_arena), &h->wf_arena_bytes, need);
  if (rc) return rc;
  char* a = h->wf_arena;
  auto take = [&](size_t bytes) { char* r = a; a += (bytes + 255) / 256 * 256; return r; };
  std::memset(&w, 0, sizeof w);
  w.P = P;
  w.n_active = reinterpret_cast<unsigned long long*>(take(8));
  w.acc = reinterpret_cast<double4*>(take(size_t(P) * 32));
  w.ray_o = reinterpret_cast<float4*>(take(size_t(P) * 16)); w.ray_d = reinterpret_cast<float4*>(take(size_t(P) * 16));
  w.st_a = reinterpret_cast<float4*>(take(size_t(P) * 16)); w.st_b = reinterpret_cast<float4*>(take(size_t(P) * 16));
  w.st_c = reinterpret_cast<float4*>(take(size_t(P) * 16)); w.hit = reinterpret_cast<float4*>(take(size_t(P) * 16));
  w.sh_o = reinterpret_cast<float4*>(take(size_t(P) * 16)); w.sh_d = reinterpret_cast<float4*>(take(size_t(P) * 16));
  w.rng = reinterpret_cast<uint64_t*>(take(size_t(P) * 8)); w.cnt = reinterpret_cast<uint2*>(take(size_t(P) * 8));
  w.sh_z = reinterpret_cast<float*>(take(size_t(P) * 4));
  if (size_t(a - h->wf_arena) > h->wf_arena_bytes) return fail(MI_ERR_INTERNAL, "wavefront arena too small");
  return MI_OK;
}

uint64_t wf_target_slots() {
  uint64_t P = 1ull << 21;  // ~2 M paths in flight (~400 MB of state): 8 k waves per kernel launch
  if (const char* e = std::getenv("MI_PT_WF_SLOTS")) { const long long v = std::atoll(e); if (v >= 256) P = uint64_t(v); }
  return P;
}

int fill_camera(const mi_pt_handle* h, uint32_t camera_id, uint32_t width, uint32_t height, mi::RenderParams& p) {
  if (camera_id >= h->scene.cameras.size()) return fail(MI_ERR_INVALID_ARGUMENT, "camera_id out of range");  // Cameras.cpp:48 runtime_assert
  if (width == 0 || height == 0 || uint64_t(width) * height > (1ull << 31)) return fail(MI_ERR_INVALID_ARGUMENT, "bad resolution");
  mi_camera_frame cf;
  mi::camera_setup(h->scene.cameras[camera_id], float(width) / float(height), cf);  // Technique.cpp:37-45
  std::memcpy(p.v2w, cf.view_to_world, sizeof p.v2w);
  std::memcpy(p.cam_pos, cf.position, sizeof p.cam_pos);
  p.focal_length_y = cf.focal_length_y;
  p.res_x = float(width); p.res_y = float(height); p.res_y_inv = 1.0f / p.res_y;
  { volatile float prod = p.res_x * p.res_y_inv; p.res_x_res_y_inv = prod; }  // one rounding, like the device's v_mul_f32
  p.width = width; p.height = height;
  return MI_OK;
}

void fill_pt(const mi_pt_handle* h, mi::RenderParams& p) {
  p.sv = h->sv;
  p.wide_nodes = h->float_nodes ? 2u : (h->wide_nodes ? 1u : 0u);
  p.stack_entries = (use_lds_scene(h) && h->kernel_choice != MI_PT_KERNEL_WAVEFRONT) ? h->info.stack_entries : h->stack_entries_hbm;
  {  // flat leaf list (r03): LDS-resident scenes of at most kFlatLeavesDefault leaf links; MI_PT_FLAT=0/1 overrides (1: up to kFlatMaxLeaves)
    const char* f = std::getenv("MI_PT_FLAT");
    const char* d = std::getenv("MI_PT_DYN");  // an explicit MI_PT_DYN=1 asks for the dynamic-fetch tree walk (A/B, parity tests)
    const bool want = f ? std::atoi(f) != 0 : (h->flat_k >= kFlatLeavesMin && h->flat_k <= kFlatLeavesDefault && !(d && std::atoi(d) != 0));
    p.flat_table = h->flat_table; p.flat_k = 0; p.flat_k_mesh = 0;
    if (want && h->flat_k && use_lds_scene(h) && h->kernel_choice != MI_PT_KERNEL_WAVEFRONT) { p.flat_k = h->flat_k; p.flat_k_mesh = h->flat_k_mesh; p.stack_entries = 0; }
  }
  p.stack_in_lds = (h->stack_fits_lds && p.stack_entries == h->info.stack_entries) ? 1u : 0u;
  const uint64_t mp = h->params.max_path;
  // "Unlimited" (Options.hpp:34, PTRDIFF_MAX) is 2^20 edges on the device: with roulette < 1 a path that long has probability < e^-100, so
  // nothing changes — but roulette = 1 in a lossless closed scene never terminates in the reference, and a kernel that never ends takes the GPU
  // with it.  Such a path is cut after 2^20 edges (its radiance so far is kept), which bounds a launch instead of hanging it.
  p.max_path = mp >= (1ull << 20) ? (1u << 20) : uint32_t(mp);
  p.min_subpath = h->params.min_subpath;
  p.beta = h->params.beta; p.roulette = h->params.roulette; p.lights = h->params.lights;
  { volatile float inv = 1.0f / p.roulette; p.inv_roulette = inv; }
  // what the scene needs of the BSDF code (kFeat* in pt_device.h); MI_PT_PLAIN_KERNEL=0 keeps the general variant (A/B)
  uint32_t f = 0;
  for (const mi_material& m : h->scene.materials) {
    if (m.type == MI_BSDF_PHONG) f |= 1u;
    if (m.type == MI_BSDF_REFLECTION || m.type == MI_BSDF_TRANSMISSION) f |= 2u;
  }
  if (p.beta != 1.0f && p.beta != 2.0f) f |= 4u;
  if (h->scene.lights.size() != 1) f |= 8u;
  const char* e = std::getenv("MI_PT_PLAIN_KERNEL");
  p.features = (e && std::atoi(e) == 0) ? 15u : f;
  // kernels that read the scene from HBM stage materials + lights + light CDF into LDS while that keeps the workgroup within the LDS share of
  // its occupancy target (160 KB / 7 workgroups for the wide walk, / 6 otherwise); MI_PT_LDS_TABLES=0/1 overrides (A/B)
  const size_t table_bytes = size_t(h->sv.blob_f4 - h->sv.off_mats) * 16;
  const size_t share = (160u * 1024u) / (h->wide_nodes && !h->float_nodes ? 7u : 6u);
  p.lds_tables = (!use_lds_scene(h) && table_bytes + size_t(h->stack_entries_hbm) * 1024 + 7424 <= share) ? 1u : 0u;
  {  // with the dynamic-fetch traversal (below) the parked rays take 6.5 KB more and five or six workgroups share a CU: the tables may use that room
    const char* d = std::getenv("MI_PT_DYN");
    if (!use_lds_scene(h) && !(d && std::atoi(d) == 0) && table_bytes + size_t(h->stack_entries_hbm) * 1024 + 7424 + 6464 <= (160u * 1024u) / 5u) p.lds_tables = 1u;
  }
  if (const char* t = std::getenv("MI_PT_LDS_TABLES")) p.lds_tables = (std::atoi(t) != 0 && !use_lds_scene(h) && table_bytes <= 48u * 1024u) ? 1u : 0u;
  {  // unified traversal with dynamic fetch (traverse_dyn): on for scenes read from HBM (+24..53 %: the loop trip is a dependent fetch, and one
     // loop over closest-hit + shadow rays with refill needs less than half the trips), off for LDS-resident scenes (-9 % at equal occupancy: the
     // loop is instruction-bound there and leaf tests run at 14 % lane utilisation either way); MI_PT_DYN=0/1 overrides (profiles/r02/ab_dynamic_fetch.txt)
    const char* d = std::getenv("MI_PT_DYN");
    const bool lds = use_lds_scene(h);
    const bool want = d ? std::atoi(d) != 0 : !lds;
    if (lds) p.dyn_traverse = (want && h->sv.n_nodes >= 1u && h->stack_fits_lds) ? 1u : 0u;
    else p.dyn_traverse = (want && h->sv.n_nodes >= 1u && p.lds_tables) ? 1u : 0u;
    if (p.flat_k) p.dyn_traverse = 0u;
  }
}

// MI_PT_INTERSECT_FLAT=1: the parity hooks run the flat leaf list (scenes that have a leaf table).  Its boxes are padded for rays that start within the
// scene's bounds (cameras included), which is where a path's rays start; a hook call with a point beyond them takes the tree walk.
bool hooks_use_flat(const mi_pt_handle* h, const mi_surface_point* a, const mi_surface_point* b, uint32_t n) {
  const char* e = std::getenv("MI_PT_INTERSECT_FLAT");
  if (!(e && std::atoi(e) != 0) || !h->flat_k) return false;
  for (uint32_t i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) {
      if (!(std::fabs(a[i].position[k]) <= h->flat_amax)) return false;
      if (b && !(std::fabs(b[i].position[k]) <= h->flat_amax)) return false;
    }
  return true;
}

}  // namespace

extern "C" {

int mi_pt_create(const mi_scene_desc* desc, const mi_pt_params* params, int device, mi_pt_handle** out) {
  if (!desc || !params || !out) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_create: null argument");
  *out = nullptr;
  mi_scene* tmp = nullptr;
  int rc = mi_scene_from_desc(desc, &tmp);
  if (rc != MI_OK) return rc;
  if (!(params->roulette > 0.0f) || !(params->roulette <= 1.0f)) {  // Options.cpp: --roulette in (0,1]
    mi_scene_free(tmp);
    return fail(MI_ERR_INVALID_ARGUMENT, "roulette must be in (0, 1]");
  }
  if (tmp->data.indices.size() / 3 > (size_t(1) << 27) || tmp->data.materials.size() >= (size_t(1) << 30)) {
    mi_scene_free(tmp);  // the scene blob is addressed in 32-bit float4 units (15 per triangle) and material ids carry 2 tag bits
    return fail(MI_ERR_UNSUPPORTED, "scene too large: at most 2^27 triangles and 2^30 materials");
  }
  if (tmp->data.lights.empty()) {  // AreaLights.cpp:217 runtime_assert(num_lights() != 0)
    mi_scene_free(tmp);
    return fail(MI_ERR_INVALID_ARGUMENT, "scene has no area lights");
  }
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
    mi_scene_free(tmp);
    return fail(MI_ERR_NO_DEVICE, "no HIP device available (libmi_pt has no CPU path)");
  }
  if (device < 0 || device >= n_dev) { mi_scene_free(tmp); return fail(MI_ERR_INVALID_ARGUMENT, "device index out of range"); }
  mi_pt_handle* h = new mi_pt_handle();
  h->device = device;
  h->scene = std::move(tmp->data);
  mi_scene_free(tmp);
  h->params = *params;
  if (h->params.min_subpath == 0) h->params.min_subpath = 3;  // PT.hpp:24
  struct Guard { mi_pt_handle* h; ~Guard() { if (h) mi_pt_destroy(h); } } guard{h};

  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipStreamCreate(&h->stream));
  HIP_TRY(hipEventCreate(&h->ev0)); HIP_TRY(hipEventCreate(&h->ev1)); HIP_TRY(hipEventCreate(&h->ev2));
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_counters), mi::kCounterWords * sizeof(unsigned long long)));

  const mi::SceneData& s = h->scene;
  const uint32_t nt = uint32_t(s.indices.size() / 3), nmat = uint32_t(s.materials.size()), nl = uint32_t(s.lights.size());
  const uint32_t n_nodes = nt > 1 ? nt - 1 : 0;
  // leaf links are ~position with bit 30 as the pair flag (layout.h), blob offsets are 32-bit float4 indices (11 float4 per triangle + nodes)
  if (s.indices.size() / 3 >= (1ull << 28)) return fail(MI_ERR_UNSUPPORTED, "scenes of 2^28 triangles or more are not supported (32-bit offsets into the scene blob)");

  // blob layout (float4 units)
  mi::SceneView sv{};
  sv.n_tris = nt; sv.n_nodes = n_nodes; sv.n_mats = nmat; sv.n_lights = nl;
  sv.off_nodes = 0;
  sv.off_tris = sv.off_nodes + 4 * n_nodes;
  sv.off_shade = sv.off_tris + 3 * nt;
  sv.off_mats = sv.off_shade + 8 * nt;
  sv.off_lights = sv.off_mats + 3 * nmat;
  sv.off_cdf = sv.off_lights + 6 * nl;
  sv.blob_f4 = sv.off_cdf + (nl + 1 + 3) / 4;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->blob), size_t(sv.blob_f4) * 16));
  HIP_TRY(hipMemset(h->blob, 0, size_t(sv.blob_f4) * 16));
  sv.blob = h->blob;
  h->sv = sv;

  // materials: mi_material with reserved := Phong diffuse probability (PhongBSDF ctor, BSDF.cpp:306-315)
  {
    std::vector<mi_material> mats = s.materials;
    for (mi_material& m : mats) {
      float pd = 0.0f;
      if (m.type == MI_BSDF_PHONG) {
        const float dr = l1(m.diffuse) * 0.318309886183790671537767526745028724f;
        const float sr = l1(m.specular) * 2.0f * 3.14159265358979323846264338327950288f / (m.power + 1.0f);
        pd = dr / (dr + sr);
      }
      std::memcpy(&m.reserved, &pd, 4);
    }
    static_assert(sizeof(mi_material) == 48, "mi_material must be 3 float4");
    HIP_TRY(hipMemcpy(h->blob + sv.off_mats, mats.data(), mats.size() * sizeof(mi_material), hipMemcpyHostToDevice));
  }
  // lights + piecewise sampler (AreaLights::_updateSampler, AreaLights.cpp:199-214)
  {
    static_assert(sizeof(mi::DevLight) == 96, "DevLight must be 6 float4");
    std::vector<mi::DevLight> dl(nl);
    std::vector<float> cdf(size_t(nl + 1 + 3) / 4 * 4, 0.0f);
    float total = 0.0f;
    for (uint32_t i = 0; i < nl; ++i) total += (s.lights[i].size[0] * s.lights[i].size[1]) * l1(s.lights[i].exitance);
    const float total_inv = 1.0f / total;
    for (uint32_t i = 0; i < nl; ++i) {
      const mi_light& l = s.lights[i];
      mi::DevLight& d = dl[i];
      std::memset(&d, 0, sizeof d);
      const float area = l.size[0] * l.size[1];
      const float weight = area * l1(l.exitance) * total_inv;
      std::memcpy(d.position, l.position, 12); d.weight = weight;
      std::memcpy(d.t0, l.tangent, 12);
      std::memcpy(d.t1, l.tangent + 3, 12); d.size_x = l.size[0];
      std::memcpy(d.t2, l.tangent + 6, 12); d.size_y = l.size[1];
      for (int k = 0; k < 3; ++k) d.radiance[k] = l.exitance[k] * 0.318309886183790671537767526745028724f;  // AreaLights.hpp:54
      d.material_id = l.material_id;
      d.lsdf_density = weight / area;
      d.area_density = 1.0f / area;
      d.diffuse = l.diffuse;
      { volatile float cd = d.area_density * weight; volatile float inv = 1.0f / cd; d.inv_cd = inv; }  // connect_prepare: cd = l5.y * l0.w, one rounding each
      cdf[i + 1] = cdf[i] + weight;
    }
    HIP_TRY(hipMemcpy(h->blob + sv.off_lights, dl.data(), dl.size() * sizeof(mi::DevLight), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->blob + sv.off_cdf, cdf.data(), cdf.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  // geometry upload + on-device LBVH
  {
    std::vector<uint32_t> tri_material(nt);
    for (size_t m = 0; m + 1 < s.mesh_tri_offset.size(); ++m)
      for (uint32_t t = s.mesh_tri_offset[m]; t < s.mesh_tri_offset[m + 1]; ++t) tri_material[t] = s.mesh_material_id[m];
    float *d_pos = nullptr, *d_tan = nullptr; uint32_t *d_idx = nullptr, *d_tm = nullptr;
    struct Tmp { void* p[4]; ~Tmp() { for (void* q : p) if (q) hipFree(q); } } tmpbuf{{nullptr, nullptr, nullptr, nullptr}};
    rc = upload(&d_pos, s.positions.data(), s.positions.size() * 4); tmpbuf.p[0] = d_pos; if (rc) return rc;
    rc = upload(&d_tan, s.tangents.data(), s.tangents.size() * 4); tmpbuf.p[1] = d_tan; if (rc) return rc;
    rc = upload(&d_idx, s.indices.data(), s.indices.size() * 4); tmpbuf.p[2] = d_idx; if (rc) return rc;
    rc = upload(&d_tm, tri_material.data(), tri_material.size() * 4); tmpbuf.p[3] = d_tm; if (rc) return rc;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_sorted_tri), size_t(nt) * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->d_morton), size_t(nt) * 8));
    float build_ms = 0.0f; uint32_t depth = 1, rounds = 0;
    const char* bsel = std::getenv("MI_PT_BVH");  // "lbvh" selects the Karras builder (A/B and regression runs)
    const int builder = (bsel && std::strcmp(bsel, "lbvh") == 0) ? 0 : 1;
    const char* pe = std::getenv("MI_PT_PAIRS");  // MI_PT_PAIRS=0: no pair leaves (A/B)
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->plain_links), size_t(n_nodes ? n_nodes : 1) * sizeof(int2)));
    HIP_TRY(mi::build_bvh(builder, nt, d_pos, d_tan, d_idx, d_tm, reinterpret_cast<mi_bvh_node*>(h->blob + sv.off_nodes), h->blob + sv.off_tris,
                          h->blob + sv.off_shade, h->d_sorted_tri, h->d_morton, h->info.scene_lo, h->info.scene_hi, &depth, &build_ms,
                          &rounds, h->plain_links, !(pe && std::atoi(pe) == 0), h->stream));
    h->info.n_triangles = nt; h->info.n_nodes = n_nodes; h->info.max_depth = depth; h->info.build_ms = build_ms;
    h->info.builder = uint32_t(builder); h->info.build_rounds = rounds;
    // quantised node copy on a 65536^3 grid over the scene box (used by the kernels that read the scene from HBM)
    for (int a = 0; a < 3; ++a) {
      // the grid overhangs the scene box by two cells on every side so padded leaf boxes never meet the clamp
      const float ext = h->info.scene_hi[a] - h->info.scene_lo[a];
      const float inv_step = ext > 0.0f ? 65531.0f / ext : 1.0f;
      h->sv.grid_inv_step[a] = inv_step;
      h->sv.grid_lo[a] = h->info.scene_lo[a] - 2.0f / inv_step;
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->qnodes), size_t(n_nodes ? n_nodes : 1) * 32));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->qnodes4), size_t(n_nodes ? n_nodes : 1) * 64));
    // the quantised walks have no per-ray slack; the padding of a child's half extent (whole cells) must cover the roundings of
    // (org - grid_lo) * cells_per_unit, org_g * inv and the fmas of wide_child_test, each ~2^-24 |org_g|: together below 2^-22 |org_g| cells.  The grid spans
    // the scene box, not the cameras, so a camera far outside a thin scene (tiny extent on one axis = tiny cells) can be millions of cells away: the pad is
    // 1 cell up to 2^20 cells (error <= 1/4 cell) and grows by one cell per further 2^20 (a margin of 4 throughout).  Paths only ever start at a camera or
    // on a surface inside the box; the parity hooks, whose callers may pass any point, use the FAR variants with the slack term.
    double far_cells = 65536.0;
    for (const mi_camera& cam : s.cameras)
      for (int a = 0; a < 3; ++a) far_cells = std::max(far_cells, std::fabs((double(cam.position[a]) - double(h->sv.grid_lo[a])) * double(h->sv.grid_inv_step[a])));
    uint32_t pad_cells = 1u + uint32_t(std::min(far_cells * 0x1p-20, 60000.0));
    if (const char* e = std::getenv("MI_PT_QUANT_PAD")) { const int v = std::atoi(e); if (v >= 1 && v <= 60000) pad_cells = uint32_t(v); }  // tests: shows what the guard prevents
    h->quant_pad_cells = pad_cells;
    bool area_collapse = true;  // MI_PT_COLLAPSE=levels: two BVH2 levels per wide record (rounds 1-3; A/B)
    if (const char* e = std::getenv("MI_PT_COLLAPSE")) area_collapse = std::strcmp(e, "levels") != 0;
    uint32_t wide_need = 0;
    HIP_TRY(mi::quantize_nodes(n_nodes, reinterpret_cast<const mi_bvh_node*>(h->blob + sv.off_nodes), h->qnodes, h->qnodes4, h->sv.grid_lo, h->sv.grid_inv_step, pad_cells, depth,
                               area_collapse, &wide_need, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->sv.qnodes = h->qnodes; h->sv.qnodes4 = h->qnodes4;
    // one fetch per iteration of the dynamic-fetch loop (pt_device.h traverse_dyn) where the walk is latency-bound: the wide records (about half as many as BVH2 nodes,
    // 64 B each) fit the 4 MB L2 of an XCD.  Beyond that the same loop measured slower (atrium 269 k -2.8 %, 2 M -10 %: twice the L2 requests for equal fabric traffic).
    // (clutter, 158 k triangles = 5 MB of records: +3.6 %; atrium 269 k = 8.6 MB: -2.8 %: the switch sits between them)
    h->sv.dyn_uni = (uint64_t(n_nodes) * 32ull <= (13ull << 19)) ? 1u : 0u;
    if (const char* e = std::getenv("MI_PT_DYN_UNI")) h->sv.dyn_uni = std::atoi(e) != 0 ? 1u : 0u;
    {  // absolute padding of centre / half-extent boxes (pt_device.h ce_box_test, traverse_flat): rays start within the scene box or at a camera
      double amax = 0.0;
      for (int a = 0; a < 3; ++a) amax = std::max(amax, std::max(std::fabs(double(h->info.scene_lo[a])), std::fabs(double(h->info.scene_hi[a]))));
      for (const mi_camera& cam : s.cameras) for (int a = 0; a < 3; ++a) amax = std::max(amax, std::fabs(double(cam.position[a])));
      h->sv.box_pad = float(amax * 0x1p-20 + 1e-30);
      h->flat_amax = float(amax);
    }
    // every scene gets the copy (64 B per node): the instrumented / list variants of small scenes read full-precision nodes from HBM too
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->ce_nodes), size_t(n_nodes ? n_nodes : 1) * 64));
    HIP_TRY(mi::ce_nodes(n_nodes, h->blob + sv.off_nodes, h->ce_nodes, h->stream));  // padded by 2^-21 of each box's own coordinates; the walk keeps the per-ray slack (ce_box_test<SLACK>)
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->sv.ce_nodes = h->ce_nodes;
    // flat leaf list (small scenes): the box and the link of every leaf link of the tree, pair leaves counting once; the leaves that hold a
    // mesh triangle come first (shadow rays test only those, Scene.cpp:42,173)
    if (n_nodes >= 1u && n_nodes <= 2u * mi::kFlatMaxLeaves) {
      std::vector<mi_bvh_node> nd(n_nodes);
      std::vector<float4> ti(size_t(nt) * 3);
      HIP_TRY(hipMemcpy(nd.data(), h->blob + sv.off_nodes, size_t(n_nodes) * sizeof(mi_bvh_node), hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(ti.data(), h->blob + sv.off_tris, ti.size() * sizeof(float4), hipMemcpyDeviceToHost));
      struct Leaf { float lo[3], hi[3]; uint32_t link; bool mesh; };
      std::vector<Leaf> leaves;
      auto entity_mask = [&](uint32_t pos) { uint32_t m; std::memcpy(&m, &ti[3 * size_t(pos) + 2].z, 4); return m; };
      std::vector<int32_t> todo{0};  // walk from the root: the nodes a pair leaf replaced are still in the array, unreferenced
      while (!todo.empty() && leaves.size() <= mi::kFlatMaxLeaves) {
        const mi_bvh_node& n = nd[size_t(todo.back())]; todo.pop_back();
        for (int c = 0; c < 2; ++c) {
          const int32_t link = c ? n.link1 : n.link0;
          if (link >= 0) { if (uint32_t(link) < n_nodes) todo.push_back(link); continue; }
          Leaf l;
          std::memcpy(l.lo, c ? n.lo1 : n.lo0, 12); std::memcpy(l.hi, c ? n.hi1 : n.hi0, 12);
          l.link = uint32_t(~link);
          const uint32_t pos = l.link & mi::kLeafPosMask;
          uint32_t m = entity_mask(pos);
          if (l.link & mi::kLeafPairBit) m |= entity_mask(pos + 1u);
          l.mesh = (m & (1u << MI_ENTITY_MESH)) != 0u;
          leaves.push_back(l);
        }
      }
      if (leaves.size() <= mi::kFlatMaxLeaves) {
        std::stable_partition(leaves.begin(), leaves.end(), [](const Leaf& l) { return l.mesh; });
        // entry = centre + half extent of the box, the half extent rounded up and padded by 2^-20 of the largest coordinate a ray can start from
        // or a box can have (scene box and cameras): covers the roundings of traverse_flat's three fmas per axis and of v_rcp_f32 (pt_device.h)
        const double pad = double(h->sv.box_pad);  // = flat_amax * 2^-20, computed once above: the table's padding and hooks_use_flat must agree
        const size_t k_pad = (leaves.size() + 3) / 4 * 4;
        std::vector<float> table(k_pad * 8, 0.0f);
        for (size_t k = leaves.size(); k < k_pad; ++k) table[8 * k + 3] = table[8 * k + 4] = table[8 * k + 5] = -1e30f;  // nothing enters a padding entry
        uint32_t k_mesh = 0;
        for (size_t k = 0; k < leaves.size(); ++k) {
          for (int a = 0; a < 3; ++a) {
            const float c = float(0.5 * (double(leaves[k].lo[a]) + double(leaves[k].hi[a])));
            const double e = std::max(double(leaves[k].hi[a]) - double(c), double(c) - double(leaves[k].lo[a])) + pad;
            float ef = float(e);
            if (double(ef) < e) ef = std::nextafter(ef, std::numeric_limits<float>::infinity());
            table[8 * k + a] = c; table[8 * k + 3 + a] = ef;
          }
          std::memcpy(&table[8 * k + 6], &leaves[k].link, 4);
          if (leaves[k].mesh) ++k_mesh;
        }
        rc = upload(&h->flat_table, table.data(), table.size() * 4); if (rc) return rc;
        h->flat_k = uint32_t(leaves.size()); h->flat_k_mesh = k_mesh;
      }
    }
    // a root-to-leaf path of `depth` nodes has depth - 1 internal nodes, each of which can leave at most one
    // far child pending: that is the stack's capacity (rounded up to 4; LDS per workgroup = 1 KB per entry)
    uint32_t need = (depth > 1 ? depth - 1u : 1u);
    need = (need + 3u) / 4u * 4u;
    // LDS part of the stack: at most 12 entries (12 KB per workgroup); deeper levels spill to private memory
    uint32_t se = need < 12u ? need : 12u;
    // the wide walk of the HBM-resident kernels leaves up to three children pending per two binary levels
    uint32_t need4 = wide_need;  // exact for the area-guided collapse (k_mark_heads), the two-levels bound otherwise
    need4 = (need4 + 3u) / 4u * 4u;
    if (need4 < need) need4 = need;
    uint32_t lds_rows = 12u;  // LDS rows of the traversal stack (1 KB each per workgroup); deeper levels live in private memory
    {  // r04: a scene whose tables (materials, lights) push a workgroup of the dynamic-fetch kernels past a sixth of a CU's LDS with twelve rows keeps eight: six
       // workgroups per CU instead of five (LivingRoomLit, 65 materials: 3 100 -> 3 177 Msamples/s; deeper stack levels live in private memory either way)
      const size_t tables = size_t(sv.blob_f4 - sv.off_mats) * 16, rest = 7424 + 6464 + tables, share = (160u * 1024u) / 6u;
      if (12u * 1024u + rest > share && 8u * 1024u + rest <= share) lds_rows = 8u;
    }
    if (const char* e = std::getenv("MI_PT_STACK_ROWS")) { const int v = std::atoi(e); if (v >= 4 && v <= 12) lds_rows = uint32_t(v) / 4u * 4u; }
    // a tree so lopsided that the wide walk could overrun the stack (LDS rows + 128 private entries) is walked through its binary records, whose need is the depth
    bool wide_fits = true;
    if (need4 > (need4 < lds_rows ? need4 : lds_rows) + 128u) { wide_fits = false; need4 = need; }
    const uint32_t se4 = need4 < lds_rows ? need4 : lds_rows;
    if (need4 > se4 + 128u) return fail(MI_ERR_UNSUPPORTED, "BVH depth " + std::to_string(depth) + " exceeds the traversal stack (12 LDS + 128 spill entries)");
    h->info.stack_entries = se;
    h->stack_fits_lds = need <= se;
    h->stack_entries_hbm = se4;
    h->wide_nodes = nt >= 100000u;  // refined below: with the dynamic-fetch traversal the wide walk wins on every scene the 16-bit grid is fine enough for
    // Is the 16-bit grid fine enough for this scene?  Median triangle box, longest side in grid cells: far-away light quads
    // stretch the scene box of some of the reference's models (MetalRings: 400 units around 0.1-unit triangles, 16 cells per
    // triangle), and boxes rounded outward to whole cells then overlap their neighbours (16 instead of 9 triangle tests per ray).
    // Below 24 cells the kernels read the full-precision 64-byte nodes instead (+9 % on MetalRings; -10..-25 % on well-scaled scenes).
    {
      std::vector<float> cells(nt);
      for (uint32_t t = 0; t < nt; ++t) {
        float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
        for (int k = 0; k < 3; ++k) {
          const float* q = &s.positions[3 * size_t(s.indices[3 * size_t(t) + k])];
          for (int a = 0; a < 3; ++a) { if (q[a] < lo[a]) lo[a] = q[a]; if (q[a] > hi[a]) hi[a] = q[a]; }
        }
        float m = 0.0f, coarsest = h->sv.grid_inv_step[0];  // longest side of the triangle's box, in cells of the coarsest axis of the grid
        for (int a = 0; a < 3; ++a) { if (hi[a] - lo[a] > m) m = hi[a] - lo[a]; if (h->sv.grid_inv_step[a] < coarsest) coarsest = h->sv.grid_inv_step[a]; }
        cells[t] = m * coarsest;
      }
      std::nth_element(cells.begin(), cells.begin() + nt / 2, cells.end());
      // r02: a loop iteration of the dynamic-fetch traversal is a dependent fetch, and the wide walk needs half as many: LivingRoomLit +4 %, CornellBoxSpecular
      // +3 %, TestCase8 (126 triangles) +14 % over the binary quantised walk (round 1, two loops: -3..-12 % below 100 000 triangles).  Scenes whose triangles
      // are small against the grid keep the full-precision binary nodes (MetalRings: wide quantised -12 %).
      const bool coarse = cells[nt / 2] < 24.0f;
      h->wide_large = h->wide_nodes;
      h->float_nodes = coarse && !h->wide_nodes;
      if (!coarse) h->wide_nodes = true;
      if (const char* e = std::getenv("MI_PT_WIDE_NODES")) { h->wide_nodes = h->wide_large = std::atoi(e) != 0; if (h->wide_nodes) h->float_nodes = false; else h->float_nodes = coarse; }
      if (const char* e = std::getenv("MI_PT_FLOAT_NODES")) h->float_nodes = std::atoi(e) != 0;
      if (!wide_fits) { h->wide_nodes = false; h->wide_large = false; if (!h->float_nodes) h->float_nodes = coarse; }  // no override re-enables a walk that can overrun its stack
    }
  }
  // scene bounding sphere for the emitters' bounded cosine sampling (BPT): the loader's value, or compute_bounding_sphere
  // (loader.cpp:408-432) over the surface meshes when the description carries none
  std::memcpy(h->sphere, s.bounding_sphere, sizeof h->sphere);
  if (!(h->sphere[3] > 0.0f)) {
    double c[3] = {0, 0, 0}; size_t nv = 0;
    for (size_t m = 0; m + 1 < s.mesh_tri_offset.size(); ++m) {
      if ((s.mesh_material_id[m] & 3u) != MI_ENTITY_MESH) continue;
      for (uint32_t t = s.mesh_tri_offset[m]; t < s.mesh_tri_offset[m + 1]; ++t)
        for (int k = 0; k < 3; ++k) { const float* q = &s.positions[3 * size_t(s.indices[3 * size_t(t) + k])]; c[0] += q[0]; c[1] += q[1]; c[2] += q[2]; ++nv; }
    }
    if (nv) {
      const float cx = float(c[0] / double(nv)), cy = float(c[1] / double(nv)), cz = float(c[2] / double(nv));
      float r2 = 0.0f;
      for (size_t m = 0; m + 1 < s.mesh_tri_offset.size(); ++m) {
        if ((s.mesh_material_id[m] & 3u) != MI_ENTITY_MESH) continue;
        for (uint32_t t = s.mesh_tri_offset[m]; t < s.mesh_tri_offset[m + 1]; ++t)
          for (int k = 0; k < 3; ++k) {
            const float* q = &s.positions[3 * size_t(s.indices[3 * size_t(t) + k])];
            const float dx = q[0] - cx, dy = q[1] - cy, dz = q[2] - cz, d2 = dx * dx + dy * dy + dz * dz;
            if (d2 > r2) r2 = d2;
          }
      }
      h->sphere[0] = cx; h->sphere[1] = cy; h->sphere[2] = cz; h->sphere[3] = std::sqrt(r2);
    }
  }
  // the LDS copy pads nodes and shading records by one float4; with the traversal stack (1 KB per entry) and the FP64 sums (~8 KB) a workgroup
  // must stay within 52 KB so that three of them fit a CU: measured on seeded soups (tests/tools/lds_limit.py; r02 with pair leaves), the LDS-resident
  // kernel wins up to 114 triangles (+35 % at 94, +16 % at 114 = 51 KB) and loses 11 % at 144 (two workgroups per CU); TestCase8, 126 triangles: +56 %
  // with the HBM-resident kernel
  size_t lds_limit = kLdsSceneLimit;
  if (const char* e = std::getenv("MI_PT_LDS_LIMIT_KB")) { const int v = std::atoi(e); if (v >= 16 && v <= 156) lds_limit = size_t(v) * 1024; }  // measurement: tests/tools/lds_limit.py
  h->lds_fits = size_t(sv.blob_f4 + sv.n_nodes + sv.n_tris) * 16 + size_t(h->info.stack_entries) * 1024 + 8192 <= lds_limit;
  guard.h = nullptr;
  *out = h;
  return MI_OK;
}

void mi_pt_destroy(mi_pt_handle* h) {
  if (!h) return;
  // a failing release must not stay behind as the thread's sticky HIP error (the next mi_pt_create would report it from its own hipGetLastError);
  // the first one is kept in mi_pt_last_error() for diagnosis
  bool failed = false;
#define D(expr) do { const hipError_t e_ = (expr); if (e_ != hipSuccess && !failed) { failed = true; fail(MI_ERR_INTERNAL, std::string("mi_pt_destroy: " #expr ": ") + hipGetErrorString(e_)); } } while (0)
  (void)hipSetDevice(h->device);
  if (h->rccl_comm) { (void)mi_pt_reduce_finalize(h); }
  if (h->blob) D(hipFree(h->blob));
  if (h->qnodes) D(hipFree(h->qnodes));
  if (h->qnodes4) D(hipFree(h->qnodes4));
  if (h->ce_nodes) D(hipFree(h->ce_nodes));
  if (h->plain_links) D(hipFree(h->plain_links));
  if (h->flat_table) D(hipFree(h->flat_table));
  if (h->d_sorted_tri) D(hipFree(h->d_sorted_tri));
  if (h->d_morton) D(hipFree(h->d_morton));
  if (h->partial) D(hipFree(h->partial));
  if (h->wf_arena) D(hipFree(h->wf_arena));
  if (h->bpt_slab) D(hipFree(h->bpt_slab));
  if (h->bpt_arena) D(hipFree(h->bpt_arena));
  if (h->bpt_values) D(hipFree(h->bpt_values));
  for (auto& fl : h->bpt_flight) {
    if (fl.values) D(hipFree(fl.values));
    if (fl.done) D(hipEventDestroy(fl.done));
    if (fl.stream) D(hipStreamDestroy(fl.stream));
  }
  if (h->bpt_pinned) D(hipHostFree(h->bpt_pinned));
  if (h->bpt_aside) D(hipFree(h->bpt_aside));
  if (h->bpt_fork) D(hipEventDestroy(h->bpt_fork));
  if (h->bpt_eye) D(hipFree(h->bpt_eye));
  if (h->bpt_light) D(hipFree(h->bpt_light));
  if (h->d_rgbn) D(hipFree(h->d_rgbn));
  if (h->d_counters) D(hipFree(h->d_counters));
  if (h->ev0) D(hipEventDestroy(h->ev0));
  if (h->ev1) D(hipEventDestroy(h->ev1));
  if (h->d_merge) D(hipFree(h->d_merge));
  if (h->h_stage) D(hipHostFree(h->h_stage));
  if (h->ev_multi) D(hipEventDestroy(h->ev_multi));
  if (h->ev2) D(hipEventDestroy(h->ev2));
  for (auto& bs : h->batches) {
    if (bs.pending_mask && bs.ev_copied) (void)hipEventSynchronize(bs.ev_copied);
    if (bs.d_rgbn) D(hipFree(bs.d_rgbn));
    if (bs.h_rgbn) D(hipHostFree(bs.h_rgbn));
    if (bs.h_counters) D(hipHostFree(bs.h_counters));
    if (bs.partial) D(hipFree(bs.partial));
    if (bs.d_counters) D(hipFree(bs.d_counters));
    for (hipEvent_t e : {bs.ev0, bs.ev1, bs.ev2, bs.ev_copied}) if (e) D(hipEventDestroy(e));
    if (bs.stream) D(hipStreamDestroy(bs.stream));
  }
  if (h->stream) D(hipStreamDestroy(h->stream));
  delete h;
#undef D
  (void)hipGetLastError();
}

int mi_pt_set_kernel(mi_pt_handle* h, int kernel) {
  if (!h || kernel < MI_PT_KERNEL_AUTO || kernel > MI_PT_KERNEL_WAVEFRONT) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_set_kernel: bad argument");
  if (kernel == MI_PT_KERNEL_MEGA_LDS && !h->lds_fits) return fail(MI_ERR_UNSUPPORTED, "scene does not fit the LDS-resident kernel");
  h->kernel_choice = kernel;
  return MI_OK;
}
int mi_pt_set_tile_shard(mi_pt_handle* h, uint32_t rank, uint32_t world) {
  if (!h) return fail(MI_ERR_INVALID_ARGUMENT, "null handle");
  if (world <= 1) { h->shard_rank = 0; h->shard_world = 1; return MI_OK; }
  if (rank >= world) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_set_tile_shard: rank must be < world");
  h->shard_rank = rank; h->shard_world = world;
  return MI_OK;
}
int mi_pt_set_instrumented(mi_pt_handle* h, int on) {
  if (!h) return fail(MI_ERR_INVALID_ARGUMENT, "null handle");
  h->instrumented = on != 0;
  return MI_OK;
}
int mi_pt_get_kernel(mi_pt_handle* h) {
  if (!h) return fail(MI_ERR_INVALID_ARGUMENT, "null handle");
  if (h->kernel_choice == MI_PT_KERNEL_WAVEFRONT) return MI_PT_KERNEL_WAVEFRONT;
  return use_lds_scene(h) ? MI_PT_KERNEL_MEGA_LDS : MI_PT_KERNEL_MEGA_GLOBAL;
}

namespace {
// counters and event times of the megakernel launch last recorded on `stream` (waits for it)
// what one render call launches into: timing events, the FP64 partial-sum buffer and the counters (the handle's own, or a frame slot's)
struct EventSet { hipEvent_t ev0, ev1, ev2; double** partial; size_t* partial_bytes; unsigned long long* counters; };
int collect_stats(mi_pt_handle* h, hipStream_t stream, mi_pt_stats* stats, const EventSet& ev) {
  unsigned long long c[24];
  HIP_TRY(hipMemcpyAsync(c, ev.counters, sizeof c, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  float t01 = 0.0f, t02 = 0.0f;
  HIP_TRY(hipEventElapsedTime(&t01, ev.ev0, ev.ev1));
  HIP_TRY(hipEventElapsedTime(&t02, ev.ev0, ev.ev2));
  stats->num_basic_rays = c[0]; stats->num_shadow_rays = c[1]; stats->numeric_errors = c[2]; stats->num_paths = c[3];
  stats->trace_ms = t01; stats->gpu_ms = t02;
  stats->nodes_closest = c[4]; stats->tris_closest = c[5]; stats->nodes_shadow = c[6]; stats->tris_shadow = c[7]; stats->num_hits = c[8]; stats->wave_steps_closest = c[9]; stats->wave_steps_shadow = c[10];
  for (int k = 0; k < 8; ++k) stats->phase_cycles[k] = c[16 + k];
  for (int k = 0; k < 4; ++k) stats->wave_loop_bodies[k] = c[11 + k];
  return MI_OK;
}

// Technique::render for `spp` frames: the launches of one call on `stream`, timed by the events of `ev`
// n_frames > 1 (frame mode only, spp == 1): samples sample_offset .. sample_offset + n_frames - 1 as separate frames into n_frames
// consecutive framebuffers, ONE launch (frame_mode_available() must hold)
bool frame_mode_available(const mi_pt_handle* h) {
  const char* e = std::getenv("MI_PT_FRAME_MODE");
  return h->kernel_choice != MI_PT_KERNEL_WAVEFRONT && !h->instrumented && !(e && std::atoi(e) == 0);
}
int render_impl(mi_pt_handle* h, uint32_t camera_id, uint32_t width, uint32_t height, mi_window win, uint32_t spp,
                uint64_t seed, uint64_t sample_offset, float* rgbn_sum_device, void* stream_v, mi_pt_stats* stats, const EventSet& ev, uint32_t n_frames = 1) {
  if (n_frames == 0 || n_frames > mi::kMaxFramesPerLaunch || (n_frames > 1 && (spp != 1 || !frame_mode_available(h))))
    return fail(MI_ERR_INTERNAL, "render_impl: bad frame batch");
  if (!h || !rgbn_sum_device) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_render: null argument");
  if (spp == 0) return fail(MI_ERR_INVALID_ARGUMENT, "spp must be > 0");
  mi::RenderParams p;
  std::memset(&p, 0, sizeof p);
  std::memset(&h->last, 0, sizeof h->last);
  int rc = fill_camera(h, camera_id, width, height, p);
  if (rc) return rc;
  if (win.w == 0 || win.h == 0) { win.x0 = 0; win.y0 = 0; win.w = width; win.h = height; }
  if (uint64_t(win.x0) + win.w > width || uint64_t(win.y0) + win.h > height)
    return fail(MI_ERR_INVALID_ARGUMENT, "window exceeds the image");  // Technique.cpp:318-319 runtime_assert
  fill_pt(h, p);
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t stream = stream_v ? static_cast<hipStream_t>(stream_v) : h->stream;

  p.win_x0 = win.x0; p.win_y0 = win.y0; p.win_w = win.w; p.win_h = win.h;
  p.tiles_x = (win.w + 7) / 8; p.tiles_y = (win.h + 7) / 8;
  const bool sharded = h->shard_world > 1;
  if (sharded) {  // this rank's 32x32 tiles (Technique.cpp:167) of the window, 16 wave tiles each; a rank may own none
    const uint64_t mtx = (uint64_t(win.w) + 31) / 32, mty = (uint64_t(win.h) + 31) / 32, mt = mtx * mty;
    const uint64_t owned = mt > h->shard_rank ? (mt - h->shard_rank + h->shard_world - 1) / h->shard_world : 0;
    p.shard_rank = h->shard_rank; p.shard_world = h->shard_world; p.shard_mtx = uint32_t(mtx);
    p.tiles_x = uint32_t(owned * 16); p.tiles_y = 1;
  }
  const uint64_t n_tiles = uint64_t(p.tiles_x) * p.tiles_y;
  if (n_tiles == 0) {  // nothing owned: the framebuffer is all zeros
    HIP_TRY(hipMemsetAsync(rgbn_sum_device, 0, size_t(width) * height * 16, stream));
    if (stats) { HIP_TRY(hipStreamSynchronize(stream)); std::memset(stats, 0, sizeof *stats); }
    return MI_OK;
  }
  if (h->kernel_choice == MI_PT_KERNEL_WAVEFRONT) {
    // slot i = pixel slot i % per_sample, replica i / per_sample: R replicas of every pixel are in flight
    const uint64_t per_sample = n_tiles * 64ull;
    uint64_t R = (wf_target_slots() + per_sample / 2) / per_sample;
    if (R < 1) R = 1;
    if (R > spp) R = spp;
    if (per_sample * R > 0xFFFFFF00ull) return fail(MI_ERR_UNSUPPORTED, "render too large for the wavefront pipeline");
    mi::WfState w;
    rc = wf_prepare(h, uint32_t(per_sample * R), w);
    if (rc) return rc;
    w.per_sample = uint32_t(per_sample); w.R = uint32_t(R); w.list = 0; w.n_items = 0;
    rc = ensure(reinterpret_cast<void**>(ev.partial), ev.partial_bytes, size_t(width) * height * 32);
    if (rc) return rc;
    p.partial = (*ev.partial); p.counters = ev.counters; p.n_chunks = 1; p.chunk_spp = spp;
    p.spp = spp; p.seed = seed; p.sample_offset = sample_offset;
    HIP_TRY(hipMemsetAsync(ev.counters, 0, mi::kCounterWords * sizeof(unsigned long long), stream));
    HIP_TRY(hipEventRecord(ev.ev0, stream));
    HIP_TRY(hipMemsetAsync((*ev.partial), 0, size_t(width) * height * 32, stream));
    h->wf_iterations = 0;
    h->last.kernel = MI_PT_KERNEL_WAVEFRONT; h->last.n_chunks = 1; h->last.chunk_spp = spp; h->last.partial_bytes = uint64_t(width) * height * 32ull;
    HIP_TRY(mi::wf_run(p, w, h->instrumented, stream, &h->wf_iterations));
    HIP_TRY(hipEventRecord(ev.ev1, stream));
    HIP_TRY(mi::launch_finalize((*ev.partial), rgbn_sum_device, width, height, win.x0, win.y0, win.w, win.h, 1, stream));
    HIP_TRY(hipEventRecord(ev.ev2, stream));
    if (stats) {
      unsigned long long c[24];
      HIP_TRY(hipMemcpyAsync(c, ev.counters, sizeof c, hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipStreamSynchronize(stream));
      float t01 = 0.0f, t02 = 0.0f;
      HIP_TRY(hipEventElapsedTime(&t01, ev.ev0, ev.ev1));
      HIP_TRY(hipEventElapsedTime(&t02, ev.ev0, ev.ev2));
      std::memset(stats, 0, sizeof *stats);
      stats->num_basic_rays = c[0]; stats->num_shadow_rays = c[1]; stats->numeric_errors = c[2]; stats->num_paths = c[3];
      stats->trace_ms = t01; stats->gpu_ms = t02;
      stats->nodes_closest = c[4]; stats->tris_closest = c[5]; stats->nodes_shadow = c[6]; stats->tris_shadow = c[7]; stats->num_hits = c[8];
      stats->wave_loop_bodies[0] = h->wf_iterations;  // pipeline rounds (extend / shade / shadow / regen)
    }
    return MI_OK;
  }
  // frame mode — the reference's cadence, one Technique::render per sample (Application.cpp:66): nothing to accumulate, so the paths write
  // (r, g, b, 1) straight into the framebuffer (no partial sums, no pt_finalize); a wave owns its tiles in every frame of the launch and
  // regenerates dead lanes onto the same pixels of the next frame
  if (spp == 1 && frame_mode_available(h)) {
    uint32_t tiles_per_wave = 1;  // measured on C2 (profiles/r02/cadence.txt): with several frames per launch one tile per wave keeps the chip full
    if (const char* t = std::getenv("MI_PT_FRAME_TILES")) { const int v = std::atoi(t); if (v >= 1 && v <= 256) tiles_per_wave = uint32_t(v); }
    // frames per wave: a wave that owns its tile in 2-4 frames runs 2-4 paths per lane (steady state: 0.17 ms per 512 x 512 frame with one
    // path per lane, 0.119 with two, 0.104 with four, 0.095 with sixteen; tools/sessions/spp_scaling.py) as long as the launch keeps several
    // rounds of waves (6 144 are resident): measured best at 4 for batches of 4-8 frames (profiles/r02/cadence.txt)
    uint32_t frame_chunk = n_frames >= 4 ? 4u : n_frames;
    if (const char* t = std::getenv("MI_PT_FRAME_CHUNK")) { const int v = std::atoi(t); if (v >= 1 && v <= int(mi::kMaxFramesPerLaunch)) frame_chunk = uint32_t(v); }
    if (frame_chunk > n_frames) frame_chunk = n_frames;
    p.frame_chunk = frame_chunk;
    const uint64_t n_waves = ((n_tiles + tiles_per_wave - 1) / tiles_per_wave) * ((n_frames + frame_chunk - 1) / frame_chunk);
    const uint64_t n_blocks = (n_waves + mi::kWavesPerBlock - 1) / mi::kWavesPerBlock;
    if (n_blocks > 0x7FFFFFFFull || n_tiles * 64ull * n_frames > 0xFFFFFFFFull || uint64_t(width) * height * n_frames > 0xFFFFFFFFull)
      return fail(MI_ERR_UNSUPPORTED, "render too large for one launch");
    p.frame_rgbn = reinterpret_cast<float4*>(rgbn_sum_device); p.frame_tiles_per_wave = tiles_per_wave;
    p.frame_count = n_frames; p.frame_stride = width * height;
    p.spp = 1; p.n_chunks = 1; p.chunk_spp = 1; p.seed = seed; p.sample_offset = sample_offset;
    p.partial = nullptr; p.counters = ev.counters;
    HIP_TRY(hipMemsetAsync(ev.counters, 0, mi::kCounterWords * sizeof(unsigned long long), stream));
    HIP_TRY(hipEventRecord(ev.ev0, stream));
    if (sharded || win.x0 != 0 || win.y0 != 0 || win.w != width || win.h != height)  // pixels no path of this call writes
      HIP_TRY(hipMemsetAsync(rgbn_sum_device, 0, size_t(width) * height * 16 * n_frames, stream));
    HIP_TRY(mi::launch_megakernel(p, use_lds_scene(h), 2, false, uint32_t(n_blocks), stream));
    HIP_TRY(hipEventRecord(ev.ev1, stream));
    HIP_TRY(hipEventRecord(ev.ev2, stream));
    mi_pt_launch_info& li = h->last;
    li.kernel = use_lds_scene(h) ? MI_PT_KERNEL_MEGA_LDS : MI_PT_KERNEL_MEGA_GLOBAL;
    li.n_blocks = uint32_t(n_blocks); li.n_chunks = 1; li.chunk_spp = 1; li.frame_tiles_per_wave = tiles_per_wave; li.frames = n_frames;
    li.lds_bytes = uint32_t(mi::pt_lds_bytes(p, use_lds_scene(h)));
    li.wide_nodes = p.wide_nodes; li.features = p.features; li.lds_tables = (!use_lds_scene(h) && p.lds_tables) ? 1u : 0u; li.dynamic_fetch = p.dyn_traverse ? 1u : 0u; li.flat_leaves = p.flat_k;
    li.partial_bytes = 0;
    li.scene_bytes = uint64_t(h->sv.blob_f4) * 16ull + uint64_t(h->sv.n_nodes) * 160ull;
    if (stats) return collect_stats(h, stream, stats, ev);
    return MI_OK;
  }
  // sample chunks: one wave owns a tile x chunk.  Measured on C2 (profiles/r01/ab_chunk.txt): 16-64 samples per
  // chunk are best (7.8 Gsamples/s), 128 loses 4 %, 256 14 %: the chip holds ~6 k waves, so aim for >= 16 rounds of
  // waves, keep >= 16 samples per wave, and bound the FP64 partial buffer (32 B per pixel and chunk) by 2 GiB.
  uint64_t n_chunks = (100000 + n_tiles - 1) / n_tiles;
  const uint64_t max_chunks = spp >= 32 ? spp / 16 : 1;
  if (n_chunks > max_chunks) n_chunks = max_chunks;
  // r02: a short render (Technique::render with a few samples per call) would leave the chip part empty — 16 spp at 512^2 is 4 096 waves for 6 144 slots,
  // 2.6 ms = 0.16 ms per sample against 0.095 in a long launch.  Four samples per wave cost 9 % per sample (tools/sessions/spp_scaling.py) but fill it:
  // below four rounds of waves the chunks go down to 4 samples.
  if (n_chunks * n_tiles < 24576 && spp >= 8) {
    const uint64_t want = (24576 + n_tiles - 1) / n_tiles, cap = spp / 4;
    const uint64_t n2 = want < cap ? want : cap;
    if (n2 > n_chunks) n_chunks = n2;
  }
  const uint64_t mem_chunks = (2ull << 30) / (uint64_t(width) * height * 32ull);
  if (n_chunks > mem_chunks) n_chunks = mem_chunks;
  if (n_chunks < 1) n_chunks = 1;
  if (const char* e = std::getenv("MI_PT_CHUNK_SPP")) {  // tuning override: samples per pixel one wave owns
    const long v = std::atol(e);
    if (v > 0) n_chunks = (spp + uint64_t(v) - 1) / uint64_t(v);
  }
  p.chunk_spp = uint32_t((spp + n_chunks - 1) / n_chunks);
  p.n_chunks = (spp + p.chunk_spp - 1) / p.chunk_spp;
  p.spp = spp; p.seed = seed; p.sample_offset = sample_offset;
  const uint64_t n_waves = n_tiles * p.n_chunks;
  const uint64_t n_blocks = (n_waves + mi::kWavesPerBlock - 1) / mi::kWavesPerBlock;
  if (n_blocks > 0x7FFFFFFFull) return fail(MI_ERR_UNSUPPORTED, "render too large for one launch");

  rc = ensure(reinterpret_cast<void**>(ev.partial), ev.partial_bytes, size_t(p.n_chunks) * width * height * 32);
  if (rc) return rc;
  p.partial = (*ev.partial);
  p.counters = ev.counters;
  HIP_TRY(hipMemsetAsync(ev.counters, 0, mi::kCounterWords * sizeof(unsigned long long), stream));
  HIP_TRY(hipEventRecord(ev.ev0, stream));
  if (sharded) HIP_TRY(hipMemsetAsync((*ev.partial), 0, size_t(p.n_chunks) * width * height * 32, stream));  // pixels of other ranks' tiles
  // MI_PT_FAST=1 (opt-in, measurement): the megakernel built with the hardware's approximate reciprocal / square root / sin / cos — NOT bit-exact against
  // the oracle, checked statistically only (tests/test_gpu_fast_math.py); the instrumented variant and every other mode stay exact
  const char* fast_env = std::getenv("MI_PT_FAST");
  const bool fast = fast_env && std::atoi(fast_env) != 0 && !h->instrumented;
  if (fast) HIP_TRY(mi::fastmath::launch_megakernel(p, use_lds_scene(h), 0, false, uint32_t(n_blocks), stream));
  else HIP_TRY(mi::launch_megakernel(p, use_lds_scene(h), 0, h->instrumented, uint32_t(n_blocks), stream));
  {
    mi_pt_launch_info& li = h->last;
    li.kernel = use_lds_scene(h) ? MI_PT_KERNEL_MEGA_LDS : MI_PT_KERNEL_MEGA_GLOBAL;
    li.n_blocks = uint32_t(n_blocks); li.n_chunks = p.n_chunks; li.chunk_spp = p.chunk_spp;
    li.lds_bytes = uint32_t(mi::pt_lds_bytes(p, use_lds_scene(h)));
    li.wide_nodes = p.wide_nodes; li.features = p.features; li.lds_tables = (!use_lds_scene(h) && p.lds_tables) ? 1u : 0u; li.dynamic_fetch = p.dyn_traverse ? 1u : 0u; li.flat_leaves = p.flat_k;
    li.partial_bytes = uint64_t(p.n_chunks) * win.w * win.h * 32ull;
    li.scene_bytes = uint64_t(h->sv.blob_f4) * 16ull + uint64_t(h->sv.n_nodes) * 160ull;
  }
  if (fast) h->last.features |= 0x80000000u;
  HIP_TRY(hipEventRecord(ev.ev1, stream));
  HIP_TRY(mi::launch_finalize((*ev.partial), rgbn_sum_device, width, height, win.x0, win.y0, win.w, win.h, p.n_chunks, stream));
  HIP_TRY(hipEventRecord(ev.ev2, stream));
  if (stats) return collect_stats(h, stream, stats, ev);
  return MI_OK;
}

}  // namespace

int mi_pt_render_device(mi_pt_handle* h, uint32_t camera_id, uint32_t width, uint32_t height, mi_window win, uint32_t spp,
                        uint64_t seed, uint64_t sample_offset, float* rgbn_sum_device, void* stream_v, mi_pt_stats* stats) {
  if (!h) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_render: null argument");
  return render_impl(h, camera_id, width, height, win, spp, seed, sample_offset, rgbn_sum_device, stream_v, stats, EventSet{h->ev0, h->ev1, h->ev2, &h->partial, &h->partial_bytes, h->d_counters});
}

int mi_pt_last_launch(mi_pt_handle* h, mi_pt_launch_info* out) {
  if (!h || !out) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_last_launch: null argument");
  *out = h->last;
  return MI_OK;
}

namespace {
// one batch = `n_frames` frames of `spp` samples each (n_frames > 1 needs spp == 1), all on the slot's own stream, nothing synchronises with the host
int enqueue_batch(mi_pt_handle* h, uint32_t camera_id, uint32_t width, uint32_t height, mi_window win, uint32_t spp, uint32_t n_frames, uint64_t seed,
                  uint64_t first_sample, uint64_t* tickets) {
  if (width == 0 || height == 0 || uint64_t(width) * height > (1ull << 31)) return fail(MI_ERR_INVALID_ARGUMENT, "bad resolution");
  HIP_TRY(hipSetDevice(h->device));
  auto& bs = h->batches[h->next_batch];
  if (bs.pending_mask) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_render_async: " + std::to_string(MI_PT_BATCHES_IN_FLIGHT) + " batches are pending; call mi_pt_wait for tickets " +
                                   std::to_string(bs.first_ticket) + " .. " + std::to_string(bs.first_ticket + bs.n_frames - 1) + " first");
  const size_t frame_bytes = size_t(width) * height * 16, bytes = frame_bytes * n_frames;
  if (!bs.ev0) {
    // HIP multiplexes streams onto 4 hardware queues (the handle's stream, the null stream, ...): a third stream of our own would share a
    // queue with the second and its kernel would wait behind that one's 16 MiB copy (measured: 340 us gaps, profiles/r02/cadence.txt).
    // Slot 0 therefore runs on the handle's stream, which is idle while frames are in flight.
    if (&bs != &h->batches[0]) HIP_TRY(hipStreamCreateWithFlags(&bs.stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&bs.ev0)); HIP_TRY(hipEventCreate(&bs.ev1)); HIP_TRY(hipEventCreate(&bs.ev2));
    HIP_TRY(hipEventCreateWithFlags(&bs.ev_copied, hipEventDisableTiming));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&bs.h_counters), mi::kCounterWords * sizeof(unsigned long long), hipHostMallocDefault));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&bs.d_counters), mi::kCounterWords * sizeof(unsigned long long)));
  }
  int rc = ensure(reinterpret_cast<void**>(&bs.d_rgbn), &bs.d_bytes, bytes);
  if (rc) return rc;
  if (bs.h_bytes < bytes) {
    if (bs.h_rgbn) hipHostFree(bs.h_rgbn);
    bs.h_rgbn = nullptr; bs.h_bytes = 0;
    if (hipHostMalloc(reinterpret_cast<void**>(&bs.h_rgbn), bytes, hipHostMallocDefault) != hipSuccess) return fail(MI_ERR_OUT_OF_MEMORY, "pinned host framebuffers");
    bs.h_bytes = bytes;
  }
  // (The wavefront pipeline keeps its path state in one arena per handle: its frames share the handle's stream and run in order.)
  hipStream_t stream = (h->kernel_choice == MI_PT_KERNEL_WAVEFRONT || !bs.stream) ? h->stream : bs.stream;
  const EventSet ev{bs.ev0, bs.ev1, bs.ev2, &bs.partial, &bs.partial_bytes, bs.d_counters};
  bs.per_frame_counts = false;
  if (n_frames > 1 && frame_mode_available(h)) {
    // ONE launch for all frames: a wave regenerates from frame to frame on its own pixels, so the chip stays full although a frame alone
    // is only 1.3 rounds of waves; per-frame counts come back in counters[32 + 4 f ..]
    rc = render_impl(h, camera_id, width, height, win, 1, seed, first_sample, bs.d_rgbn, stream, nullptr, ev, n_frames);
    if (rc) return rc;
    bs.launched = h->last.n_blocks != 0;
    bs.per_frame_counts = true;
    if (bs.launched) HIP_TRY(hipMemcpyAsync(bs.h_counters, bs.d_counters, mi::kCounterWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
  } else {
    for (uint32_t f = 0; f < n_frames; ++f) {  // one launch per frame; each frame's totals are parked in the per-frame section of the host copy
      rc = render_impl(h, camera_id, width, height, win, spp, seed, first_sample + f, bs.d_rgbn + size_t(f) * width * height * 4, stream, nullptr, ev, 1);
      if (rc) return rc;
      bs.launched = h->last.n_blocks != 0 || h->kernel_choice == MI_PT_KERNEL_WAVEFRONT;
      if (bs.launched) HIP_TRY(hipMemcpyAsync(bs.h_counters + 32 + 4 * f, bs.d_counters, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    }
  }
  HIP_TRY(hipMemcpyAsync(bs.h_rgbn, bs.d_rgbn, bytes, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipEventRecord(bs.ev_copied, stream));
  if (win.w == 0 || win.h == 0) { win.x0 = 0; win.y0 = 0; win.w = width; win.h = height; }
  bs.width = width; bs.height = height; bs.win = win; bs.n_frames = n_frames; bs.synced = false;
  bs.first_ticket = h->next_ticket; h->next_ticket += n_frames;
  bs.pending_mask = n_frames >= 32 ? 0xFFFFFFFFu : ((1u << n_frames) - 1u);
  for (uint32_t f = 0; f < n_frames; ++f) tickets[f] = bs.first_ticket + f;
  h->next_batch = (h->next_batch + 1) % MI_PT_BATCHES_IN_FLIGHT;
  return MI_OK;
}

int wait_frame(mi_pt_handle* h, uint64_t ticket, mi_pt_handle::BatchSlot** out, uint32_t* frame, mi_pt_stats* stats) {
  mi_pt_handle::BatchSlot* bs = nullptr;
  for (auto& b : h->batches)
    if (b.pending_mask && ticket >= b.first_ticket && ticket < b.first_ticket + b.n_frames && (b.pending_mask >> (ticket - b.first_ticket) & 1u)) bs = &b;
  if (!bs) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_wait: ticket " + std::to_string(ticket) + " is not pending");
  const uint32_t f = uint32_t(ticket - bs->first_ticket);
  if (!bs->synced) {
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipEventSynchronize(bs->ev_copied));
    bs->synced = true;
  }
  bs->pending_mask &= ~(1u << f);
  *out = bs; *frame = f;
  if (stats) {
    std::memset(stats, 0, sizeof *stats);
    if (bs->launched) {
      // per-frame counts: [closest-hit rays, shadow rays, numeric errors, paths] (frame launches) or the launch totals [basic, shadow, errors, paths]
      const unsigned long long* c = bs->h_counters + 32 + 4 * f;
      stats->num_basic_rays = c[0]; stats->num_shadow_rays = c[1]; stats->numeric_errors = c[2]; stats->num_paths = c[3];
      float t01 = 0.0f, t02 = 0.0f;  // device time of the batch's (last) launch, shared evenly by its frames
      HIP_TRY(hipEventElapsedTime(&t01, bs->ev0, bs->ev1));
      HIP_TRY(hipEventElapsedTime(&t02, bs->ev0, bs->ev2));
      const double share = bs->per_frame_counts ? 1.0 / double(bs->n_frames) : 1.0;
      stats->trace_ms = t01 * share; stats->gpu_ms = t02 * share;
    }
  }
  return MI_OK;
}
}  // namespace

int mi_pt_render_async(mi_pt_handle* h, uint32_t camera_id, uint32_t width, uint32_t height, mi_window win, uint32_t spp, uint64_t seed,
                       uint64_t sample_offset, uint64_t* ticket) {
  if (!h || !ticket) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_render_async: null argument");
  if (spp == 0) return fail(MI_ERR_INVALID_ARGUMENT, "spp must be > 0");
  return enqueue_batch(h, camera_id, width, height, win, spp, 1, seed, sample_offset, ticket);
}

int mi_pt_render_frames_async(mi_pt_handle* h, uint32_t camera_id, uint32_t width, uint32_t height, mi_window win, uint32_t n_frames, uint64_t seed,
                              uint64_t first_sample, uint64_t* tickets) {
  if (!h || !tickets) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_render_frames_async: null argument");
  if (n_frames == 0 || n_frames > MI_PT_MAX_FRAMES_PER_BATCH) return fail(MI_ERR_INVALID_ARGUMENT, "n_frames must be in [1, " + std::to_string(MI_PT_MAX_FRAMES_PER_BATCH) + "]");
  return enqueue_batch(h, camera_id, width, height, win, 1, n_frames, seed, first_sample, tickets);
}

int mi_pt_wait(mi_pt_handle* h, uint64_t ticket, const float** rgbn_sum, mi_pt_stats* stats) {
  if (!h || !rgbn_sum) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_wait: null argument");
  mi_pt_handle::BatchSlot* bs = nullptr; uint32_t f = 0;
  const int rc = wait_frame(h, ticket, &bs, &f, stats);
  if (rc) return rc;
  *rgbn_sum = bs->h_rgbn + size_t(f) * bs->width * bs->height * 4;
  return MI_OK;
}

int mi_pt_wait_add(mi_pt_handle* h, uint64_t ticket, double* view, mi_pt_stats* stats) {
  if (!h || !view) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_wait_add: null argument");
  mi_pt_handle::BatchSlot* bs = nullptr; uint32_t f = 0;
  const int rc = wait_frame(h, ticket, &bs, &f, stats);
  if (rc) return rc;
  mi::add_frame_to_view(bs->h_rgbn + size_t(f) * bs->width * bs->height * 4, view, bs->width, bs->win.x0, bs->win.y0, bs->win.w, bs->win.h);
  return MI_OK;
}

int mi_pt_render(mi_pt_handle* h, uint32_t camera_id, uint32_t width, uint32_t height, mi_window win, uint32_t spp, uint64_t seed,
                 uint64_t sample_offset, float* rgbn_sum, mi_pt_stats* stats) {
  if (!h || !rgbn_sum) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_render: null argument");
  if (width == 0 || height == 0 || uint64_t(width) * height > (1ull << 31)) return fail(MI_ERR_INVALID_ARGUMENT, "bad resolution");
  HIP_TRY(hipSetDevice(h->device));
  const size_t bytes = size_t(width) * height * 16;
  int rc = ensure(reinterpret_cast<void**>(&h->d_rgbn), &h->rgbn_bytes, bytes);
  if (rc) return rc;
  mi_pt_stats local;
  rc = mi_pt_render_device(h, camera_id, width, height, win, spp, seed, sample_offset, h->d_rgbn, nullptr, stats ? stats : &local);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(rgbn_sum, h->d_rgbn, bytes, hipMemcpyDeviceToHost));
  return MI_OK;
}

int mi_pt_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int mi_pt_render_multi(mi_pt_handle* const* handles, uint32_t n_handles, uint32_t camera_id, uint32_t width, uint32_t height, mi_window win,
                       uint32_t spp, uint64_t seed, uint64_t sample_offset, float* rgbn_sum, mi_pt_stats* stats) {
  if (!handles || n_handles == 0 || !rgbn_sum) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_render_multi: null argument");
  for (uint32_t k = 0; k < n_handles; ++k) {
    if (!handles[k]) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_render_multi: null handle");
    for (uint32_t j = 0; j < k; ++j) if (handles[j] == handles[k]) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_render_multi: a handle is listed twice");
  }
  if (width == 0 || height == 0 || uint64_t(width) * height > (1ull << 31)) return fail(MI_ERR_INVALID_ARGUMENT, "bad resolution");
  if (win.w == 0 || win.h == 0) { win.x0 = 0; win.y0 = 0; win.w = width; win.h = height; }
  if (uint64_t(win.x0) + win.w > width || uint64_t(win.y0) + win.h > height)
    return fail(MI_ERR_INVALID_ARGUMENT, "window exceeds the image");  // Technique.cpp:318-319 runtime_assert
  const size_t bytes = size_t(width) * height * 16;
  // Where is the frame put together?  On the first handle's device when it can read every owner's framebuffer (the same device, or a peer: xGMI
  // between the GPUs of one node), one copy of the merged frame to the host; otherwise — or with MI_PT_MULTI_HOST_MERGE=1 — every framebuffer
  // goes to pinned host memory and the tiles are assembled there (n x 16 B per pixel over PCIe; the round-2 path).
  bool device_merge = n_handles <= kMaxMergeSources;
  if (const char* e = std::getenv("MI_PT_MULTI_HOST_MERGE")) if (std::atoi(e) != 0) device_merge = false;
  for (uint32_t k = 1; k < n_handles && device_merge; ++k) {
    if (handles[k]->device == handles[0]->device) continue;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, handles[0]->device, handles[k]->device) != hipSuccess || !can) { device_merge = false; break; }
    if (hipSetDevice(handles[0]->device) != hipSuccess) { device_merge = false; break; }
    const hipError_t pe = hipDeviceEnablePeerAccess(handles[k]->device, 0);
    if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) device_merge = false;
    (void)hipGetLastError();
  }
  // 1. every device renders its tiles on its own stream (host-side merge: and copies its framebuffer to pinned host memory)
  int rc = MI_OK;
  uint32_t launched = 0;
  for (; launched < n_handles && rc == MI_OK; ++launched) {
    mi_pt_handle* h = handles[launched];
    rc = hipSetDevice(h->device) == hipSuccess ? MI_OK : fail(MI_ERR_NO_DEVICE, "hipSetDevice failed");
    if (rc == MI_OK) rc = ensure(reinterpret_cast<void**>(&h->d_rgbn), &h->rgbn_bytes, bytes);
    if (rc == MI_OK && !device_merge && h->h_stage_bytes < bytes) {
      if (h->h_stage) hipHostFree(h->h_stage);
      h->h_stage = nullptr; h->h_stage_bytes = 0;
      if (hipHostMalloc(reinterpret_cast<void**>(&h->h_stage), bytes, hipHostMallocDefault) != hipSuccess) rc = fail(MI_ERR_OUT_OF_MEMORY, "pinned host staging buffer");
      else h->h_stage_bytes = bytes;
    }
    if (rc != MI_OK) break;
    const uint32_t saved_rank = h->shard_rank, saved_world = h->shard_world;
    h->shard_rank = launched; h->shard_world = n_handles;
    rc = mi_pt_render_device(h, camera_id, width, height, win, spp, seed, sample_offset, h->d_rgbn, nullptr, nullptr);  // asynchronous
    h->shard_rank = saved_rank; h->shard_world = saved_world;
    if (rc == MI_OK && !device_merge && hipMemcpyAsync(h->h_stage, h->d_rgbn, bytes, hipMemcpyDeviceToHost, h->stream) != hipSuccess) rc = fail(MI_ERR_NO_DEVICE, "framebuffer copy failed");
    if (rc == MI_OK && device_merge) {
      if (!h->ev_multi && hipEventCreateWithFlags(&h->ev_multi, hipEventDisableTiming) != hipSuccess) rc = fail(MI_ERR_NO_DEVICE, "event creation failed");
      if (rc == MI_OK && hipEventRecord(h->ev_multi, h->stream) != hipSuccess) rc = fail(MI_ERR_NO_DEVICE, "event record failed");
    }
  }
  // 1b. device-side merge on the first handle's stream, behind every owner's render; the merged frame goes straight to the caller's buffer
  if (rc == MI_OK && device_merge) {
    mi_pt_handle* h0 = handles[0];
    rc = hipSetDevice(h0->device) == hipSuccess ? MI_OK : fail(MI_ERR_NO_DEVICE, "hipSetDevice failed");
    if (rc == MI_OK) rc = ensure(reinterpret_cast<void**>(&h0->d_merge), &h0->merge_bytes, bytes);
    if (rc == MI_OK) {
      MergeSources src;
      for (uint32_t k = 0; k < kMaxMergeSources; ++k) src.fb[k] = reinterpret_cast<const float4*>(handles[k < n_handles ? k : 0]->d_rgbn);
      for (uint32_t k = 1; k < n_handles && rc == MI_OK; ++k)
        if (hipStreamWaitEvent(h0->stream, handles[k]->ev_multi, 0) != hipSuccess) rc = fail(MI_ERR_NO_DEVICE, "cross-device wait failed");
      if (rc == MI_OK) {
        const uint32_t n_px = width * height;
        hipLaunchKernelGGL(k_merge_tiles, dim3((n_px + 255u) / 256u), dim3(256), 0, h0->stream, src, n_handles, reinterpret_cast<float4*>(h0->d_merge), width, height,
                           win.x0, win.y0, win.w, win.h, (win.w + 31u) / 32u);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(rgbn_sum, h0->d_merge, bytes, hipMemcpyDeviceToHost, h0->stream) != hipSuccess)
          rc = fail(MI_ERR_NO_DEVICE, "device-side merge failed");
      }
    }
  }
  // 2. wait for every device that was started, also after an error
  mi_pt_stats total; std::memset(&total, 0, sizeof total);
  for (uint32_t k = 0; k < launched; ++k) {
    mi_pt_handle* h = handles[k];
    if (hipSetDevice(h->device) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) { if (rc == MI_OK) rc = fail(MI_ERR_NO_DEVICE, "device synchronisation failed"); continue; }
    if (rc != MI_OK || !stats) continue;
    const uint64_t mtx = (uint64_t(win.w) + 31) / 32, mt = mtx * ((uint64_t(win.h) + 31) / 32);
    if (mt <= k) continue;  // this handle owned no tile: nothing was launched on it
    mi_pt_stats one; std::memset(&one, 0, sizeof one);
    if (h->kernel_choice == MI_PT_KERNEL_WAVEFRONT) {  // the pipeline has no per-launch record to read back: counts only
      unsigned long long c[24];
      if (hipMemcpy(c, h->d_counters, sizeof c, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(MI_ERR_NO_DEVICE, "counter copy failed"); continue; }
      one.num_basic_rays = c[0]; one.num_shadow_rays = c[1]; one.numeric_errors = c[2]; one.num_paths = c[3];
    } else {
      const int r2 = collect_stats(h, h->stream, &one, EventSet{h->ev0, h->ev1, h->ev2, &h->partial, &h->partial_bytes, h->d_counters});
      if (r2 != MI_OK) { rc = r2; continue; }
    }
    total.num_basic_rays += one.num_basic_rays; total.num_shadow_rays += one.num_shadow_rays; total.numeric_errors += one.numeric_errors;
    total.num_paths += one.num_paths; total.nodes_closest += one.nodes_closest; total.tris_closest += one.tris_closest;
    total.nodes_shadow += one.nodes_shadow; total.tris_shadow += one.tris_shadow; total.num_hits += one.num_hits;
    total.wave_steps_closest += one.wave_steps_closest; total.wave_steps_shadow += one.wave_steps_shadow;
    for (int j = 0; j < 8; ++j) total.phase_cycles[j] += one.phase_cycles[j];
    for (int j = 0; j < 4; ++j) total.wave_loop_bodies[j] += one.wave_loop_bodies[j];
    if (one.trace_ms > total.trace_ms) total.trace_ms = one.trace_ms;
    if (one.gpu_ms > total.gpu_ms) total.gpu_ms = one.gpu_ms;
  }
  if (rc != MI_OK) return rc;
  if (device_merge) {  // step 2 waited for handles[0]'s stream too: the merged frame is in rgbn_sum
    g_last_multi_merge = 1;
    if (stats) *stats = total;
    return MI_OK;
  }
  g_last_multi_merge = 0;
  // 3. (host-side merge) every 32x32 tile of the window from its owner; everything else is zero
  std::memset(rgbn_sum, 0, bytes);
  const uint32_t mtx = (win.w + 31) / 32, mty = (win.h + 31) / 32;
  for (uint32_t ty = 0; ty < mty; ++ty)
    for (uint32_t tx = 0; tx < mtx; ++tx) {
      const float* src = handles[(uint64_t(ty) * mtx + tx) % n_handles]->h_stage;
      const uint32_t x0 = win.x0 + tx * 32, x1 = x0 + 32 < win.x0 + win.w ? x0 + 32 : win.x0 + win.w;
      const uint32_t y0 = win.y0 + ty * 32, y1 = y0 + 32 < win.y0 + win.h ? y0 + 32 : win.y0 + win.h;
      for (uint32_t y = y0; y < y1; ++y)
        std::memcpy(rgbn_sum + (size_t(y) * width + x0) * 4, src + (size_t(y) * width + x0) * 4, size_t(x1 - x0) * 16);
    }
  if (stats) *stats = total;
  return MI_OK;
}

int mi_pt_last_multi_merge(void) { return g_last_multi_merge; }

// ---- RCCL sum-reduce of the per-GPU framebuffers (one process per GPU; north_star: "RCCL reduce over xGMI of the per-GPU float3 framebuffer") ----
// The in-memory form of `master merge` (merge_exr, Options.cpp:1340-1409: dst = fst + snd on R, G, B, denom).  librccl.so is looked up at the first
// call (dlopen), never linked: a host without it gets MI_ERR_UNSUPPORTED and everything else keeps working.  The types below restate the few ABI
// facts of rccl.h that are used (ncclUniqueId = 128 opaque bytes, ncclFloat32 = 7, ncclSum = 0, ncclSuccess = 0).
namespace {
struct RcclId { char internal[MI_PT_REDUCE_ID_BYTES]; };
struct RcclApi {
  void* so = nullptr;
  int (*GetUniqueId)(RcclId*) = nullptr;
  int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Reduce)(const void*, void*, size_t, int, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string why;
};
RcclApi* rccl() {
  static RcclApi api;
  static bool tried = false;
  if (tried) return &api;
  tried = true;
  // The RCCL that belongs to the HIP runtime THIS library runs on comes first: the librccl next to the loaded libamdhip64 (dladdr).  A process may hold a
  // second ROCm stack (a Python host with a framework that bundles its own librccl / libhsa-runtime64): an RCCL bound to the other stack's HSA instance
  // finds that one uninitialised ("no ROCm-capable device is detected").
  std::string beside_hip[2];
  Dl_info info;
  if (dladdr(reinterpret_cast<const void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
    std::string dir(info.dli_fname);
    const size_t slash = dir.rfind('/');
    if (slash != std::string::npos) { dir.resize(slash); beside_hip[0] = dir + "/librccl.so"; beside_hip[1] = dir + "/librccl.so.1"; }
  }
  const char* names[] = {std::getenv("MI_PT_RCCL_LIB"), beside_hip[0].c_str(), beside_hip[1].c_str(), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char* n : names) {
    if (!n || !*n) continue;
    api.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (api.so) break;
    api.why = dlerror();
  }
  if (!api.so) return &api;
#define MI_SYM(field, name) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.so, name)); if (!api.field) { api.why = std::string("librccl has no ") + name; dlclose(api.so); api.so = nullptr; return &api; }
  MI_SYM(GetUniqueId, "ncclGetUniqueId") MI_SYM(CommInitRank, "ncclCommInitRank") MI_SYM(AllReduce, "ncclAllReduce") MI_SYM(Reduce, "ncclReduce")
  MI_SYM(CommDestroy, "ncclCommDestroy") MI_SYM(GetErrorString, "ncclGetErrorString")
#undef MI_SYM
  return &api;
}
int rccl_fail(const char* what, int rc) { return fail(MI_ERR_INTERNAL, std::string(what) + ": " + (rccl()->GetErrorString ? rccl()->GetErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")"); }
}  // namespace

int mi_pt_reduce_available(void) { return rccl()->so ? 1 : 0; }

int mi_pt_reduce_unique_id(unsigned char id[MI_PT_REDUCE_ID_BYTES]) {
  if (!id) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_reduce_unique_id: null argument");
  RcclApi* r = rccl();
  if (!r->so) return fail(MI_ERR_UNSUPPORTED, "RCCL is not available on this host (librccl.so: " + r->why + ")");
  RcclId u;
  const int rc = r->GetUniqueId(&u);
  if (rc) return rccl_fail("ncclGetUniqueId", rc);
  std::memcpy(id, u.internal, MI_PT_REDUCE_ID_BYTES);
  return MI_OK;
}

int mi_pt_reduce_init(mi_pt_handle* h, const unsigned char id[MI_PT_REDUCE_ID_BYTES], uint32_t rank, uint32_t world) {
  if (!h || !id) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_reduce_init: null argument");
  if (world == 0 || rank >= world) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_reduce_init: rank must be < world");
  if (h->rccl_comm) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_reduce_init: the handle has a communicator already (mi_pt_reduce_finalize first)");
  RcclApi* r = rccl();
  if (!r->so) return fail(MI_ERR_UNSUPPORTED, "RCCL is not available on this host (librccl.so: " + r->why + ")");
  HIP_TRY(hipSetDevice(h->device));  // the communicator binds to the current device: one rank per GPU
  RcclId u;
  std::memcpy(u.internal, id, MI_PT_REDUCE_ID_BYTES);
  void* comm = nullptr;
  const int rc = r->CommInitRank(&comm, int(world), u, int(rank));
  if (rc) return rccl_fail("ncclCommInitRank", rc);
  h->rccl_comm = comm; h->reduce_rank = rank; h->reduce_world = world;
  return MI_OK;
}

int mi_pt_reduce_rgbn(mi_pt_handle* h, float* rgbn_sum_device, uint32_t width, uint32_t height, int root, void* stream_v) {
  if (!h || !rgbn_sum_device) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_reduce_rgbn: null argument");
  if (!h->rccl_comm) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_reduce_rgbn: call mi_pt_reduce_init first");
  if (width == 0 || height == 0 || uint64_t(width) * height > (1ull << 31)) return fail(MI_ERR_INVALID_ARGUMENT, "bad resolution");
  if (root >= int(h->reduce_world)) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_reduce_rgbn: root must be < world (or negative: all-reduce)");
  RcclApi* r = rccl();
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t stream = stream_v ? static_cast<hipStream_t>(stream_v) : h->stream;
  const size_t count = size_t(width) * height * 4;  // (R, G, B, denom) per pixel, FP32: 126.6 MiB at 3840x2160
  const int rc = root < 0 ? r->AllReduce(rgbn_sum_device, rgbn_sum_device, count, 7 /* ncclFloat32 */, 0 /* ncclSum */, h->rccl_comm, stream)
                          : r->Reduce(rgbn_sum_device, rgbn_sum_device, count, 7, 0, root, h->rccl_comm, stream);
  if (rc) return rccl_fail(root < 0 ? "ncclAllReduce" : "ncclReduce", rc);
  if (!stream_v) HIP_TRY(hipStreamSynchronize(stream));  // the handle's own stream: the caller has nothing to wait on
  return MI_OK;
}

int mi_pt_reduce_finalize(mi_pt_handle* h) {
  if (!h) return fail(MI_ERR_INVALID_ARGUMENT, "null handle");
  if (!h->rccl_comm) return MI_OK;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  const int rc = rccl()->CommDestroy(h->rccl_comm);
  h->rccl_comm = nullptr; h->reduce_world = 0; h->reduce_rank = 0;
  if (rc) return rccl_fail("ncclCommDestroy", rc);
  return MI_OK;
}

int mi_pt_intersect(mi_pt_handle* h, uint32_t n, const mi_surface_point* origins, const float* directions, mi_surface_point* out_hits,
                    float* out_t, uint32_t* out_prim) {
  if (!h || (n && (!origins || !directions))) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_intersect: null argument");
  if (n == 0) return MI_OK;
  HIP_TRY(hipSetDevice(h->device));
  mi_surface_point *d_o = nullptr, *d_h = nullptr; float *d_d = nullptr, *d_t = nullptr; uint32_t* d_p = nullptr;
  struct Tmp { void* p[5]; ~Tmp() { for (void* q : p) if (q) hipFree(q); } } tmp{{nullptr, nullptr, nullptr, nullptr, nullptr}};
  int rc = upload(&d_o, origins, size_t(n) * sizeof(mi_surface_point)); tmp.p[0] = d_o; if (rc) return rc;
  rc = upload(&d_d, directions, size_t(n) * 12); tmp.p[1] = d_d; if (rc) return rc;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_h), size_t(n) * sizeof(mi_surface_point))); tmp.p[2] = d_h;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_t), size_t(n) * 4)); tmp.p[3] = d_t;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_p), size_t(n) * 4)); tmp.p[4] = d_p;
  if (hooks_use_flat(h, origins, nullptr, n)) HIP_TRY(mi::launch_intersect_flat(h->sv, h->flat_table, h->flat_k, n, d_o, d_d, d_h, d_t, d_p, h->stream));
  else
  HIP_TRY(mi::launch_intersect(h->sv, h->wide_nodes, h->stack_entries_hbm, n, d_o, d_d, d_h, d_t, d_p, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (out_hits) HIP_TRY(hipMemcpy(out_hits, d_h, size_t(n) * sizeof(mi_surface_point), hipMemcpyDeviceToHost));
  if (out_t) HIP_TRY(hipMemcpy(out_t, d_t, size_t(n) * 4, hipMemcpyDeviceToHost));
  if (out_prim) HIP_TRY(hipMemcpy(out_prim, d_p, size_t(n) * 4, hipMemcpyDeviceToHost));
  return MI_OK;
}

int mi_pt_occluded(mi_pt_handle* h, uint32_t n, const mi_surface_point* origins, const mi_surface_point* targets, float* out_visibility) {
  if (!h || (n && (!origins || !targets || !out_visibility))) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_occluded: null argument");
  if (n == 0) return MI_OK;
  HIP_TRY(hipSetDevice(h->device));
  mi_surface_point *d_a = nullptr, *d_b = nullptr; float* d_o = nullptr;
  struct Tmp { void* p[3]; ~Tmp() { for (void* q : p) if (q) hipFree(q); } } tmp{{nullptr, nullptr, nullptr}};
  int rc = upload(&d_a, origins, size_t(n) * sizeof(mi_surface_point)); tmp.p[0] = d_a; if (rc) return rc;
  rc = upload(&d_b, targets, size_t(n) * sizeof(mi_surface_point)); tmp.p[1] = d_b; if (rc) return rc;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_o), size_t(n) * 4)); tmp.p[2] = d_o;
  if (hooks_use_flat(h, origins, targets, n)) HIP_TRY(mi::launch_occluded_flat(h->sv, h->flat_table, h->flat_k, h->flat_k_mesh, n, d_a, d_b, d_o, h->stream));
  else
  HIP_TRY(mi::launch_occluded(h->sv, h->wide_nodes, h->stack_entries_hbm, n, d_a, d_b, d_o, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(out_visibility, d_o, size_t(n) * 4, hipMemcpyDeviceToHost));
  return MI_OK;
}

int mi_pt_trace_paths(mi_pt_handle* h, uint32_t camera_id, uint32_t width, uint32_t height, uint32_t n, const uint32_t* pixel_xy,
                      const uint64_t* sample_index, uint64_t seed, float* out_radiance, uint32_t* out_ray_counts) {
  if (!h || (n && (!pixel_xy || !sample_index || !out_radiance))) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_trace_paths: null argument");
  if (n == 0) return MI_OK;
  mi::RenderParams p;
  std::memset(&p, 0, sizeof p);
  int rc = fill_camera(h, camera_id, width, height, p);
  if (rc) return rc;
  for (uint32_t i = 0; i < n; ++i)
    if (pixel_xy[2 * i] >= width || pixel_xy[2 * i + 1] >= height) return fail(MI_ERR_INVALID_ARGUMENT, "pixel outside the image");
  fill_pt(h, p);
  HIP_TRY(hipSetDevice(h->device));
  uint32_t *d_xy = nullptr, *d_c = nullptr; uint64_t* d_s = nullptr; float* d_r = nullptr;
  struct Tmp { void* p[4]; ~Tmp() { for (void* q : p) if (q) hipFree(q); } } tmp{{nullptr, nullptr, nullptr, nullptr}};
  rc = upload(&d_xy, pixel_xy, size_t(n) * 8); tmp.p[0] = d_xy; if (rc) return rc;
  rc = upload(&d_s, sample_index, size_t(n) * 8); tmp.p[1] = d_s; if (rc) return rc;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_r), size_t(n) * 12)); tmp.p[2] = d_r;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_c), size_t(n) * 8)); tmp.p[3] = d_c;
  p.win_w = width; p.win_h = height;
  p.seed = seed; p.list_xy = d_xy; p.list_sample = d_s; p.list_n = n; p.list_radiance = d_r; p.list_counts = d_c;
  p.counters = nullptr;
  if (h->kernel_choice == MI_PT_KERNEL_WAVEFRONT) {
    mi::WfState w;
    uint64_t P = wf_target_slots();
    if (P > n) P = (uint64_t(n) + 255) / 256 * 256;
    rc = wf_prepare(h, uint32_t(P), w);
    if (rc) return rc;
    w.list = 1; w.n_items = n; w.per_sample = w.P; w.R = 1;
    p.counters = h->d_counters;
    HIP_TRY(hipMemsetAsync(h->d_counters, 0, mi::kCounterWords * sizeof(unsigned long long), h->stream));
    HIP_TRY(mi::wf_run(p, w, false, h->stream, nullptr));
  } else {
    const uint32_t per_block = uint32_t(mi::kWavesPerBlock) * 64u * 16u;
    HIP_TRY(mi::launch_megakernel(p, use_lds_scene(h), 1, false, (n + per_block - 1) / per_block, h->stream));
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(out_radiance, d_r, size_t(n) * 12, hipMemcpyDeviceToHost));
  if (out_ray_counts) HIP_TRY(hipMemcpy(out_ray_counts, d_c, size_t(n) * 8, hipMemcpyDeviceToHost));
  return MI_OK;
}

namespace {
bool bpt_staged() {  // MI_BPT_STAGED=0 selects the one-kernel form (kept for A/B)
  const char* e = std::getenv("MI_BPT_STAGED");
  return !(e && std::atoi(e) == 0);
}
// one launch of `w.lanes` paths: the one-kernel form, or trace -> (item count) -> items -> gather
int bpt_launch(mi_pt_handle* h, const mi::RenderParams& p, mi::BptState& w, bool list, hipStream_t stream) {
  const mi::BptLaunchers bl = mi::bpt_launchers(p.features);
  if (!bpt_staged()) { HIP_TRY(bl.frame(p, w, list, stream)); return MI_OK; }
  const bool lds = use_lds_scene(h) && h->stack_fits_lds;  // small scenes with shallow trees: padded copy of the blob in LDS, binary walk, stack without a spill path
  auto run = [&](mi::BptState& ws, bool* overflow) -> int {
    uint32_t total = 0;
    const bool debug = std::getenv("MI_BPT_DEBUG") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    // r04: scenes read from HBM trace their sub-paths as uniform steps (bpt_step / bpt_closest rounds, bpt_kernels.hip) instead of one lane walking both
    // sub-paths of its path; MI_BPT_STEPS=0/1 forces the per-lane form / the steps (the kernels of LDS-resident scenes always walk per lane)
    bool steps = false;  // measured (profiles/r04/ab_bpt_steps.txt): not yet ahead of the per-lane form — opt-in
    if (const char* e = std::getenv("MI_BPT_STEPS")) steps = std::atoi(e) != 0 && !lds && ws.step_state != nullptr;
    int steps_mode = 1;  // MI_BPT_STEPS=1: rounds of (persistent walk, step) + tail; 2: passes with growing ray budgets, every path walking its own rays
    if (const char* e = std::getenv("MI_BPT_STEPS")) steps_mode = std::atoi(e);
    if (steps && steps_mode == 2) { uint32_t rounds = 0; HIP_TRY(bl.trace_passes(p, ws, list, stream, &total, &rounds)); h->bpt_step_rounds = rounds; }
    else if (steps) { uint32_t rounds = 0; HIP_TRY(bl.trace_steps(p, ws, list, stream, &total, &rounds)); h->bpt_step_rounds = rounds; }
    else { HIP_TRY(bl.trace(p, ws, list, lds, stream, &total)); h->bpt_step_rounds = 0; }
    const auto t_trace = std::chrono::steady_clock::now();
    unsigned long long over = 0;
    HIP_TRY(hipMemcpy(&over, h->d_counters + 15, sizeof over, hipMemcpyDeviceToHost));
    if (over) {
      if (std::getenv("MI_BPT_DEBUG")) std::fprintf(stderr, "[mi_bpt] %llu sub-paths outgrew %u vertices in a launch of %u paths: redone in slices\n", over, ws.max_vertices, ws.lanes);
      *overflow = true; HIP_TRY(hipMemsetAsync(h->d_counters + 15, 0, sizeof over, stream)); return MI_OK;
    }
    // The visibility stage pays where a launch holds many more shadow rays than the chip has lanes (524 288 resident) and a node is a dependent
    // fetch from L2 / HBM: LivingRoomLit (20 M items per launch) +13 %, CornellBoxSpecular (5 M) +8 %; MetalRings (0.6 M) -4 %, scenes walked
    // in LDS -8 % (profiles/r02/ab_bpt_visibility.txt).  MI_BPT_DYN_VIS=0/1 forces it off / on.
    ws.dyn_vis = (!lds && total >= (2u << 20)) ? 1u : 0u;
    if (const char* e = std::getenv("MI_BPT_DYN_VIS")) ws.dyn_vis = std::atoi(e) != 0 ? 1u : 0u;
    // values [items][16 B]; with the visibility stage also: shadow rays [items][32 B] | occlusion bytes [items] | ray count + chunk cursor
    // (launches without it — LDS-resident scenes, fewer than 2 M items — hold the values alone: a third of the bytes)
    const size_t n_it = total ? total : 1, occl_bytes = ws.dyn_vis ? (n_it + 255) / 256 * 256 : 0;
    // grown with a quarter of headroom: the item count moves a few per cent from launch to launch, and an allocation made while the driver is still clearing
    // the memory of a handle destroyed a moment ago waits for it (seen: 3.5 s for 0.9 GB after a 116 GB arena was freed; MI_BPT_DEBUG=1 prints such waits)
    const size_t values_need = n_it * (ws.dyn_vis ? 48 : 16) + occl_bytes + 256;
    int rc = h->bpt_values_bytes >= values_need ? MI_OK : ensure(reinterpret_cast<void**>(&h->bpt_values), &h->bpt_values_bytes, values_need + values_need / 4);
    if (rc) return rc;
    const auto t_ensure = std::chrono::steady_clock::now();
    ws.values = h->bpt_values;
    ws.rays = ws.dyn_vis ? h->bpt_values + n_it : nullptr;
    ws.occl = ws.dyn_vis ? reinterpret_cast<uint8_t*>(h->bpt_values + 3 * n_it) : nullptr;
    ws.pool = ws.dyn_vis ? reinterpret_cast<uint32_t*>(ws.occl + occl_bytes) : nullptr;
    ws.vis_th = 16u; ws.vis_wide = p.wide_nodes == 1u ? 1u : 0u;
    if (const char* e = std::getenv("MI_BPT_VIS_TH")) { const int v = std::atoi(e); if (v >= 1 && v <= 64) ws.vis_th = uint32_t(v); }
    if (const char* e = std::getenv("MI_BPT_VIS_WIDE")) ws.vis_wide = std::atoi(e) != 0 ? 1u : 0u;
    HIP_TRY(bl.connect(p, ws, list, lds, total, stream));
    if (debug) {
      HIP_TRY(hipStreamSynchronize(stream));
      const auto t_end = std::chrono::steady_clock::now();
      auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
      if (ms(t_begin, t_end) > 50.0)
        std::fprintf(stderr, "[mi_bpt] slow launch: trace %.1f ms, overflow read + buffers %.1f ms, connect %.1f ms (%u items, %zu values bytes)\n", ms(t_begin, t_trace), ms(t_trace, t_ensure),
                     ms(t_ensure, t_end), total, h->bpt_values_bytes);
    }
    return MI_OK;
  };
  bool overflow = false;
  int rc = run(w, &overflow);
  if (rc || !overflow) return rc;
  // A sub-path outgrew the slab share of this launch (long paths: roulette close to 1, or little free memory).  The same memory holds fewer paths at a
  // larger capacity: redo the launch in slices at four times the capacity (r04: not at once at the reference's 1024 vertices, BPT.hpp:30 — with a share of 32
  // vertices that was 32 slices per launch, 5 s instead of 0.12 s for 64 frames of CornellBoxSpecular when another handle held most of the device), and a
  // slice that still overflows again at four times its own, up to 1024.
  std::function<int(const mi::BptState&)> redo = [&](const mi::BptState& wf) -> int {
    if (wf.max_vertices >= 1024u) return fail(MI_ERR_UNSUPPORTED, "BPT: a sub-path exceeds 1024 vertices (the reference's fixed_vector capacity)");
    const uint32_t cap = wf.max_vertices * 4u > 1024u ? 1024u : wf.max_vertices * 4u;
    const uint64_t slice = uint64_t(wf.lanes) * wf.max_vertices / cap;
    if (slice == 0) return fail(MI_ERR_UNSUPPORTED, "BPT: a sub-path exceeds the vertex slab of a single path");
    for (uint32_t done = 0; done < wf.lanes; done += uint32_t(slice)) {
      mi::BptState ws = wf;
      ws.first = wf.first + done; ws.lanes = wf.lanes - done < slice ? wf.lanes - done : uint32_t(slice); ws.max_vertices = cap;
      bool again = false;
      int r = run(ws, &again);
      if (r) return r;
      if (again) { r = redo(ws); if (r) return r; }
    }
    return MI_OK;
  };
  return redo(w);
}
// buffers and per-launch constants shared by the two BPT entry points
// flights > 1 (mi_bpt_render): the launch's paths and the arena are dealt to `flights` launches in flight; w is the first one's state, more[0 .. flights - 2] the others'
int bpt_prepare(mi_pt_handle* h, uint32_t camera_id, uint32_t width, uint32_t height, uint64_t total_lanes, mi::RenderParams& p, mi::BptState& w,
                uint32_t* lanes_per_launch, uint32_t flights = 1, mi::BptState* more = nullptr, bool set_aside = false) {
  std::memset(&p, 0, sizeof p); std::memset(&w, 0, sizeof w);
  int rc = fill_camera(h, camera_id, width, height, p);
  if (rc) return rc;
  fill_pt(h, p);
  if (p.beta == 0.0f && p.features != 15u) p.features &= ~4u;  // FixedBeta<0> needs no pow either (Beta.hpp:24-41)
  p.wide_nodes = h->float_nodes ? 2u : (h->wide_nodes ? 1u : 0u);  // r03: with centre / half-extent children the wide walk is no worse on small scenes and +4 % on LivingRoomLit (r01: from 100 000 triangles on)
  p.stack_entries = (bpt_staged() && use_lds_scene(h) && h->stack_fits_lds) ? h->info.stack_entries : h->stack_entries_hbm;  // staged kernels stage small scenes into LDS
  {  // staged kernels of LDS-resident scenes: the flat leaf list under the PT rule (8..24 leaf links; MI_PT_FLAT / MI_BPT_FLAT = 0/1 override)
    const bool lds = bpt_staged() && use_lds_scene(h) && h->stack_fits_lds;
    bool want = h->flat_k >= kFlatLeavesMin && h->flat_k <= kFlatLeavesDefault;
    if (const char* e = std::getenv("MI_PT_FLAT")) want = std::atoi(e) != 0;
    if (const char* e = std::getenv("MI_BPT_FLAT")) want = std::atoi(e) != 0;
    p.flat_table = h->flat_table;
    if (lds && want && h->flat_k) { p.flat_k = h->flat_k; p.flat_k_mesh = h->flat_k_mesh; } else { p.flat_k = 0; p.flat_k_mesh = 0; }
  }
  mi_camera_frame fr;
  rc = mi_camera_setup(&h->scene.cameras[camera_id], float(width) / float(height), &fr);
  if (rc) return rc;
  std::memcpy(w.w2v, fr.world_to_view, sizeof w.w2v);
  std::memcpy(w.sphere, h->sphere, sizeof w.sphere);
  std::memcpy(w.sky_horizon, h->sky_horizon, sizeof w.sky_horizon); std::memcpy(w.sky_zenith, h->sky_zenith, sizeof w.sky_zenith);
  if (width > 65535u || height > 65535u)  // the staged form packs a path's pixel as (y << 16) | x in its info record (bpt_kernels.hip)
    return fail(MI_ERR_UNSUPPORTED, "BPT: width and height must not exceed 65535");
  // up to 1 M paths per launch (MI_BPT_LANES_LOG2 = 16 .. 22: measurement); 4 M where a path that outgrows its slab share is set aside for a launch of its own
  // (mi_bpt_render, r04): the share may then be short — 80 vertices hold all but 0.02 % of the sub-paths at roulette 0.9 — and the same slabs hold four times the paths
  uint64_t max_lanes = set_aside ? (1ull << 22) : (1ull << 20);
  if (const char* e = std::getenv("MI_BPT_LANES_LOG2")) { const int v = std::atoi(e); if (v >= 16 && v <= 22) max_lanes = 1ull << v; }
  uint64_t lanes = total_lanes < max_lanes ? total_lanes : max_lanes;
  lanes = (lanes + 255) / 256 * 256;
  const bool staged = bpt_staged();
  // vertex slabs: at most 3 x 36 GB (staged) or 24 GB, and at most 40 % of what the device has free right now (another process may share
  // the GPU; a smaller GPU has less): the capacity per sub-path shrinks, long paths then go through the slice path of bpt_launch
  // r04: 3 x 36 GB — a launch of 1 M paths (several frames) keeps 320 vertices per sub-path: with fewer, a sub-path outgrows its share every few launches at
  // roulette 0.9 (P(length > 146) = 2e-7 per sub-path, two million sub-paths per launch) and the launch is redone in slices; the device has 288 GB
  uint64_t budget = staged ? (36ull << 30) : (24ull << 30);
  {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 0) {
      const uint64_t have = staged ? h->bpt_arena_bytes : h->bpt_slab_bytes;  // what this handle already holds counts as available to it
      // the memory of a process that has just exited (or of a handle destroyed a moment ago) comes back asynchronously: seen as 64 frames in 5 s instead of 0.12 s
      // because the share was 16 vertices per sub-path.  While the free figure is small AND still growing, wait for it (at most 2 s).
      for (int k = 0; k < 20 && (uint64_t(free_b) + have) * 2 / 5 / (staged ? 3 : 1) < (6ull << 30); ++k) {  // < 192 vertices for 2^18 paths
        size_t f2 = 0, t2 = 0;
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
        if (hipMemGetInfo(&f2, &t2) != hipSuccess) break;
        const bool growing = f2 > free_b + (64u << 20);
        free_b = f2;
        if (!growing && k >= 2) break;
      }
      const uint64_t share = (uint64_t(free_b) + have) * 2 / 5 / (staged ? 3 : 1);
      if (share < budget) budget = share;
    }
  }
  if (const char* e = std::getenv("MI_BPT_SLAB_MB")) { const long long v = std::atoll(e); if (v > 0) budget = uint64_t(v) << 20; }
  uint64_t cap = budget / (lanes * 112ull);
  // little memory (another handle or process holds the device): fewer paths per launch before a short slab — a sub-path of more than 192 vertices is a 2e-9
  // event at roulette 0.9, one of more than 80 happens in every launch, and an overflowing launch is redone in slices (bpt_launch)
  while (cap < (set_aside ? 64u : 192u) && lanes > (1ull << 18)) { lanes = (lanes / 2 + 255) / 256 * 256; cap = budget / (lanes * 112ull); }
  if (cap > 1024) cap = 1024;
  if (cap < 16) cap = 16;
  uint64_t lanes_flight = lanes;
  for (;;) {  // an allocation that fails is retried at half the capacity, then at half the paths per launch, down to 16 vertices x 256 paths
    w.max_vertices = uint32_t(cap);
    if (staged) {
      const uint64_t lf = flights > 1 ? (lanes / flights / 256 * 256 > 256 ? lanes / flights / 256 * 256 : 256) : lanes;  // paths of one launch in flight
      lanes_flight = lf;
      const size_t slab = size_t(lf) * cap * 112;
      const size_t step_bytes = size_t(lf) * (mi::kBptStepF4 * 16 + 32 + 16 + 8) + 4096;  // tracing stage as uniform steps: state, ray, hit, two index lists per path
      const size_t need = (3 * slab + size_t(lf) * (cap * 24 + 32 + 4 + 1) + 16384 + step_bytes) * flights;
      // a little more than asked: the per-path arrays make the need move by a per cent with the split into paths x vertices, and growing the arena means waiting for
      // the driver to take back ~100 GB first (seconds)
      rc = h->bpt_arena_bytes >= need ? MI_OK : ensure(reinterpret_cast<void**>(&h->bpt_arena), &h->bpt_arena_bytes, need + need / 32);
      if (rc == MI_OK) {
        char* a = h->bpt_arena;
        auto take = [&](size_t bytes) { char* r = a; a += (bytes + 255) / 256 * 256; return r; };
        for (uint32_t f = 0; f < flights; ++f) {
          mi::BptState& ws = f == 0 ? w : more[f - 1];
          if (f) ws = w;
          ws.lslab = reinterpret_cast<float4*>(take(slab)); ws.eslab = reinterpret_cast<float4*>(take(slab)); ws.nslab = reinterpret_cast<float4*>(take(slab));
          ws.emission = reinterpret_cast<float4*>(take(size_t(lf) * cap * 16));
          ws.evinfo = reinterpret_cast<uint2*>(take(size_t(lf) * cap * 8));
          ws.info = reinterpret_cast<uint4*>(take(size_t(lf) * 32));
          ws.item_offset = reinterpret_cast<uint32_t*>(take((size_t(lf) + 1) * 4));
          ws.scan_tmp = reinterpret_cast<uint32_t*>(take((size_t(lf) / 2048 + 2) * 4));
          ws.step_state = reinterpret_cast<float4*>(take(size_t(lf) * mi::kBptStepF4 * 16));
          ws.step_rays = reinterpret_cast<float4*>(take(size_t(lf) * 32));
          ws.step_hits = reinterpret_cast<float4*>(take(size_t(lf) * 16));
          ws.step_active[0] = reinterpret_cast<uint32_t*>(take(size_t(lf) * 4));
          ws.step_active[1] = reinterpret_cast<uint32_t*>(take(size_t(lf) * 4));
          ws.step_count = reinterpret_cast<uint32_t*>(take(1024));  // [0..1] list counts, [4..5] path cursors of the regenerating kernels, [8..71] chunk cursors of the walking waves
        }
      }
    } else {
      rc = ensure(reinterpret_cast<void**>(&h->bpt_slab), &h->bpt_slab_bytes, size_t(lanes) * cap * 112);
      if (rc == MI_OK) w.slab = h->bpt_slab;
    }
    if (rc == MI_OK) break;
    if (rc != MI_ERR_OUT_OF_MEMORY) return rc;
    (void)hipGetLastError();  // clear the sticky out-of-memory error before the retry
    if (cap > 16) cap = cap / 2 < 16 ? 16 : cap / 2;
    else if (lanes > 256) lanes = (lanes / 2 + 255) / 256 * 256;
    else return rc;
  }
  if (!staged) lanes_flight = lanes;
  *lanes_per_launch = uint32_t(lanes_flight);
  if (std::getenv("MI_BPT_DEBUG")) {
    size_t f = 0, t = 0; (void)hipMemGetInfo(&f, &t);
    std::fprintf(stderr, "[mi_bpt] %u launch(es) in flight, paths per launch %llu, vertices per sub-path %u, slab budget %.1f GB, device free %.1f of %.1f GB, arena %.1f GB\n", flights, (unsigned long long)lanes_flight,
                 w.max_vertices, double(budget) / 1073741824.0, double(f) / 1073741824.0, double(t) / 1073741824.0, double(h->bpt_arena_bytes) / 1073741824.0);
  }
  p.counters = h->d_counters;
  return MI_OK;
}
}  // namespace

int mi_bpt_set_sky(mi_pt_handle* h, const float horizon[3], const float zenith[3]) {
  if (!h || !horizon || !zenith) return fail(MI_ERR_INVALID_ARGUMENT, "mi_bpt_set_sky: null argument");
  std::memcpy(h->sky_horizon, horizon, 12); std::memcpy(h->sky_zenith, zenith, 12);
  return MI_OK;
}

int mi_bpt_render(mi_pt_handle* h, uint32_t camera_id, uint32_t width, uint32_t height, mi_window win, uint32_t spp, uint64_t seed,
                  uint64_t sample_offset, float* rgbn_sum, mi_pt_stats* stats) {
  if (!h || !rgbn_sum) return fail(MI_ERR_INVALID_ARGUMENT, "mi_bpt_render: null argument");
  if (h->shard_world > 1) return fail(MI_ERR_UNSUPPORTED, "mi_bpt_render: light-image splats cross tiles; shard BPT by sample ranges (sample_offset)");
  if (spp == 0) return fail(MI_ERR_INVALID_ARGUMENT, "spp must be > 0");
  if (win.w == 0 || win.h == 0) { win.x0 = 0; win.y0 = 0; win.w = width; win.h = height; }
  if (uint64_t(win.x0) + win.w > width || uint64_t(win.y0) + win.h > height)
    return fail(MI_ERR_INVALID_ARGUMENT, "window exceeds the image");  // Technique.cpp:318-319 runtime_assert
  HIP_TRY(hipSetDevice(h->device));
  mi::RenderParams p; mi::BptState w; uint32_t per_launch = 0;
  const uint64_t tiles_x = (uint64_t(win.w) + 7) / 8, tiles_y = (uint64_t(win.h) + 7) / 8, total = tiles_x * tiles_y * 64;
  // r04 — LAUNCHES IN FLIGHT.  The tracing stage of a launch lasts as long as its longest sub-path (~140 vertices of dependent tree walks at roulette 0.9: 1.3 ms of
  // a sub-path kernel however few lanes are left in it), and one launch is only a few paths per resident lane.  The paths of a "launch" are therefore dealt to
  // launches on streams of their own (same arena, a share each); while one of them is in its tail another traces or connects.  Each keeps the order trace ->
  // item count on the host -> connect; the frames of a batch are committed when all of its launches are done.  Measured (profiles/r04/ab_bpt_steps.txt #6): two
  // in flight -5 .. -11 % (LivingRoomLit 188 -> 167 ms), MetalRings +6 %; four need more hardware queues than the runtime gives a process by default (two streams then
  // share one: +15 %) and reach no more with GPU_MAX_HW_QUEUES=8 — overlapping kernels slow each other down, the chip was not idle through those tails, it was
  // waiting on memory.  MI_BPT_FLIGHTS=1..4 (1: one launch at a time).
  uint32_t flights = bpt_staged() ? 2u : 1u;
  if (const char* e = std::getenv("MI_BPT_FLIGHTS")) { const int v = std::atoi(e); if (v >= 1 && v <= int(mi_pt_handle::kBptFlights)) flights = bpt_staged() ? uint32_t(v) : 1u; }
  if (total * uint64_t(spp) < (1ull << 18)) flights = 1;  // nothing to overlap
  mi::BptState wmore[mi_pt_handle::kBptFlights];
  // lanes of a launch: up to 1 M paths over SEVERAL frames in flight (r04: a launch of one 512 x 512 frame is 4 096 waves — one round of the chip, which then waits for
  // its longest sub-path)
  const bool aside = bpt_staged() && !(std::getenv("MI_BPT_SET_ASIDE") && std::atoi(std::getenv("MI_BPT_SET_ASIDE")) == 0);
  int rc = bpt_prepare(h, camera_id, width, height, total * uint64_t(spp), p, w, &per_launch, flights, wmore, aside);
  if (rc) return rc;
  p.win_x0 = win.x0; p.win_y0 = win.y0; p.win_w = win.w; p.win_h = win.h; p.tiles_x = uint32_t(tiles_x); p.tiles_y = uint32_t(tiles_y);
  p.spp = spp; p.seed = seed; p.sample_offset = sample_offset; p.n_chunks = 1;
  const size_t np = size_t(width) * height;
  rc = ensure(reinterpret_cast<void**>(&h->partial), &h->partial_bytes, np * 32); if (rc) return rc;
  // frames are rendered in batches: each frame of a batch has its own eye / light image; the commit walks them in frame order.  A batch holds at least four
  // rounds of the launches in flight (the pipeline drains at a commit) within 4 GB of images.
  uint64_t batch = uint64_t(per_launch) * flights * (flights > 1 ? 16u : 1u) / total;
  if (batch < 1) batch = 1;
  { const uint64_t by_mem = (4ull << 30) / (np * 36); if (batch > by_mem) batch = by_mem < 1 ? 1 : by_mem; }
  if (aside) { const uint64_t by_ids = (1ull << 28) / total; if (batch > by_ids) batch = by_ids < 1 ? 1 : by_ids; }  // the list of paths set aside holds every path of a batch: <= 1 GB
  if (batch > spp) batch = spp;
  if (batch > 64) batch = 64;
  rc = ensure(reinterpret_cast<void**>(&h->bpt_eye), &h->bpt_eye_bytes, np * 12 * batch); if (rc) return rc;
  rc = ensure(reinterpret_cast<void**>(&h->bpt_light), &h->bpt_light_bytes, np * 24 * batch); if (rc) return rc;
  rc = ensure(reinterpret_cast<void**>(&h->d_rgbn), &h->rgbn_bytes, np * 16); if (rc) return rc;
  p.partial = h->partial; w.eye = h->bpt_eye; w.light = h->bpt_light;
  if (aside) {
    const uint64_t ids = total * (batch < spp ? batch : uint64_t(spp));
    rc = ensure(reinterpret_cast<void**>(&h->bpt_aside), &h->bpt_aside_bytes, (ids + 16) * 4); if (rc) return rc;
    w.over_count = h->bpt_aside; w.over_ids = h->bpt_aside + 16;
  }
  for (uint32_t f = 1; f < flights; ++f) { wmore[f - 1].eye = w.eye; wmore[f - 1].light = w.light; wmore[f - 1].over_count = w.over_count; wmore[f - 1].over_ids = w.over_ids; }
  // the paths of a batch that outgrew their slab share, traced again at the reference's capacity of 1024 vertices (BPT.hpp:30) through the first launch's arena,
  // which holds share / 1024 of its paths at that capacity; `on` = the stream that arena's work is ordered on.  Called when every trace of the batch is done.
  auto flush_aside = [&](uint32_t f0, uint32_t frames, hipStream_t on) -> int {
    if (!aside) return MI_OK;
    uint32_t n = 0;
    HIP_TRY(hipMemcpy(&n, h->bpt_aside, sizeof n, hipMemcpyDeviceToHost));
    if (n == 0) return MI_OK;
    if (std::getenv("MI_BPT_DEBUG")) std::fprintf(stderr, "[mi_bpt] %u paths of the batch outgrew %u vertices: traced again at 1024\n", n, w.max_vertices);
    if (w.max_vertices >= 1024u) return fail(MI_ERR_UNSUPPORTED, "BPT: a sub-path exceeds 1024 vertices (the reference's fixed_vector capacity)");
    const uint64_t chunk = uint64_t(per_launch) * w.max_vertices / 1024u;
    if (chunk == 0) return fail(MI_ERR_UNSUPPORTED, "BPT: a sub-path exceeds the vertex slab of a single path");
    for (uint64_t done = 0; done < n; done += chunk) {
      mi::BptState wm = w;
      wm.frame = f0; wm.frames = frames; wm.first = 0; wm.lanes = uint32_t(n - done < chunk ? n - done : chunk); wm.max_vertices = 1024u;
      wm.path_ids = h->bpt_aside + 16 + done; wm.over_ids = nullptr; wm.over_count = nullptr; wm.async_total = 0u;
      const int r = bpt_launch(h, p, wm, false, on);
      if (r) return r;
    }
    return MI_OK;
  };
  hipStream_t stream = h->stream;
  HIP_TRY(hipMemsetAsync(h->d_counters, 0, mi::kCounterWords * sizeof(unsigned long long), stream));
  HIP_TRY(hipMemsetAsync(h->partial, 0, np * 32, stream));
  HIP_TRY(hipMemsetAsync(h->bpt_eye, 0, np * 12 * batch, stream));
  HIP_TRY(hipMemsetAsync(h->bpt_light, 0, np * 24 * batch, stream));
  HIP_TRY(hipEventRecord(h->ev0, stream));
  if (flights <= 1) {
    for (uint32_t f = 0; f < spp; f += uint32_t(batch)) {
      w.frame = f; w.frames = uint32_t(spp - f < batch ? spp - f : batch);
      const uint64_t lanes_total = total * w.frames;
      if (aside) HIP_TRY(hipMemsetAsync(h->bpt_aside, 0, 64, stream));
      for (uint64_t first = 0; first < lanes_total; first += per_launch) {
        w.first = uint32_t(first); w.lanes = uint32_t(lanes_total - first < per_launch ? lanes_total - first : per_launch);
        w.persist_hint = h->bpt_rays_per_path < 0.0f ? 0u : (h->bpt_rays_per_path >= 6.5f ? 2u : 1u);
        rc = bpt_launch(h, p, w, false, stream);
        if (rc) return rc;
        unsigned long long c4[4] = {0, 0, 0, 0};
        HIP_TRY(hipMemcpyAsync(c4, h->d_counters, sizeof c4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (c4[3] >= 4096ull) h->bpt_rays_per_path = float(double(c4[0]) / double(c4[3]));
      }
      rc = flush_aside(f, w.frames, stream);
      if (rc) return rc;
      HIP_TRY(mi::bpt_launchers(p.features).commit(p, w, stream));
    }
  } else {
    const mi::BptLaunchers bl = mi::bpt_launchers(p.features);
    const bool lds = use_lds_scene(h) && h->stack_fits_lds;
    if (!h->bpt_pinned) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->bpt_pinned), sizeof(unsigned long long) * 8 * mi_pt_handle::kBptFlights, hipHostMallocDefault));
    if (!h->bpt_fork) HIP_TRY(hipEventCreateWithFlags(&h->bpt_fork, hipEventDisableTiming));
    for (uint32_t f = 0; f < flights; ++f) {
      auto& fl = h->bpt_flight[f];
      if (!fl.stream) HIP_TRY(hipStreamCreateWithFlags(&fl.stream, hipStreamNonBlocking));
      if (!fl.done) HIP_TRY(hipEventCreateWithFlags(&fl.done, hipEventDisableTiming));
    }
    const bool debug = std::getenv("MI_BPT_DEBUG") != nullptr;
    mi::BptState cur[mi_pt_handle::kBptFlights];
    bool pending[mi_pt_handle::kBptFlights] = {false, false, false, false};
    auto state_of = [&](uint32_t f) -> mi::BptState& { return f == 0 ? w : wmore[f - 1]; };
    auto begin = [&](uint32_t f, uint32_t frame, uint32_t frames, uint64_t first, uint32_t lanes) -> int {
      auto& fl = h->bpt_flight[f];
      mi::BptState ws = state_of(f);
      ws.frame = frame; ws.frames = frames; ws.first = uint32_t(first); ws.lanes = lanes; ws.async_total = 1u;
      ws.persist_hint = h->bpt_rays_per_path < 0.0f ? 0u : (h->bpt_rays_per_path >= 6.5f ? 2u : 1u);
      uint32_t* total_word = reinterpret_cast<uint32_t*>(h->bpt_pinned + 8 * f);
      HIP_TRY(bl.trace(p, ws, false, lds, fl.stream, total_word));
      HIP_TRY(hipMemcpyAsync(h->bpt_pinned + 8 * f + 1, h->d_counters + 15, sizeof(unsigned long long), hipMemcpyDeviceToHost, fl.stream));
      HIP_TRY(hipMemcpyAsync(h->bpt_pinned + 8 * f + 2, h->d_counters, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, fl.stream));  // rays and paths of the launches gathered so far
      cur[f] = ws; pending[f] = true;
      return MI_OK;
    };
    // a sub-path outgrew its slab share in one of the traces in flight (the counter does not say which): none of them has been connected yet — wait for all,
    // and redo each through the one-at-a-time path, which slices it at a larger share (bpt_launch)
    auto redo_pending = [&]() -> int {
      for (uint32_t g = 0; g < flights; ++g) if (pending[g]) HIP_TRY(hipStreamSynchronize(h->bpt_flight[g].stream));
      HIP_TRY(hipMemset(h->d_counters + 15, 0, sizeof(unsigned long long)));
      for (uint32_t g = 0; g < flights; ++g) {
        if (!pending[g]) continue;
        mi::BptState ws = cur[g]; ws.async_total = 0u;
        if (debug) std::fprintf(stderr, "[mi_bpt] launch in flight %u (%u paths) redone one at a time after an overflow\n", g, ws.lanes);
        const int r = bpt_launch(h, p, ws, false, h->bpt_flight[g].stream);
        if (r) return r;
        pending[g] = false;
      }
      return MI_OK;
    };
    auto finish = [&](uint32_t f) -> int {
      auto& fl = h->bpt_flight[f];
      HIP_TRY(hipStreamSynchronize(fl.stream));
      if (h->bpt_pinned[8 * f + 1] != 0ull) return redo_pending();
      const uint32_t items = *reinterpret_cast<const uint32_t*>(h->bpt_pinned + 8 * f);
      if (h->bpt_pinned[8 * f + 5] >= 4096ull) h->bpt_rays_per_path = float(double(h->bpt_pinned[8 * f + 2]) / double(h->bpt_pinned[8 * f + 5]));
      mi::BptState& ws = cur[f];
      ws.dyn_vis = (!lds && items >= (1u << 19) && uint64_t(items) >= 2ull * ws.lanes) ? 1u : 0u;  // the rule of bpt_launch (2 M items for 1 M paths) per path
      if (const char* e = std::getenv("MI_BPT_DYN_VIS")) ws.dyn_vis = std::atoi(e) != 0 ? 1u : 0u;
      const size_t n_it = items ? items : 1, occl_bytes = ws.dyn_vis ? (n_it + 255) / 256 * 256 : 0;
      const size_t values_need = n_it * (ws.dyn_vis ? 48 : 16) + occl_bytes + 256;
      if (fl.values_bytes < values_need) { const int r = ensure(reinterpret_cast<void**>(&fl.values), &fl.values_bytes, values_need + values_need / 4); if (r) return r; }
      ws.values = fl.values;
      ws.rays = ws.dyn_vis ? fl.values + n_it : nullptr;
      ws.occl = ws.dyn_vis ? reinterpret_cast<uint8_t*>(fl.values + 3 * n_it) : nullptr;
      ws.pool = ws.dyn_vis ? reinterpret_cast<uint32_t*>(ws.occl + occl_bytes) : nullptr;
      ws.vis_th = 16u; ws.vis_wide = p.wide_nodes == 1u ? 1u : 0u;
      if (const char* e = std::getenv("MI_BPT_VIS_TH")) { const int v = std::atoi(e); if (v >= 1 && v <= 64) ws.vis_th = uint32_t(v); }
      if (const char* e = std::getenv("MI_BPT_VIS_WIDE")) ws.vis_wide = std::atoi(e) != 0 ? 1u : 0u;
      HIP_TRY(bl.connect(p, ws, false, lds, items, fl.stream));
      pending[f] = false;
      return MI_OK;
    };
    h->bpt_step_rounds = 0;
    uint64_t k = 0;  // launches begun so far: launch k runs on flight k mod flights
    for (uint32_t f0 = 0; f0 < spp; f0 += uint32_t(batch)) {
      const uint32_t frames = uint32_t(spp - f0 < batch ? spp - f0 : batch);
      const uint64_t lanes_total = total * frames;
      if (aside) HIP_TRY(hipMemsetAsync(h->bpt_aside, 0, 64, stream));
      HIP_TRY(hipEventRecord(h->bpt_fork, stream));  // the images are cleared (first batch) or committed (later ones)
      for (uint32_t f = 0; f < flights; ++f) HIP_TRY(hipStreamWaitEvent(h->bpt_flight[f].stream, h->bpt_fork, 0));
      for (uint64_t first = 0; first < lanes_total; ++k) {
        const uint32_t f = uint32_t(k % flights);
        const uint32_t lanes = uint32_t(lanes_total - first < per_launch ? lanes_total - first : per_launch);
        if (pending[f]) { rc = finish(f); if (rc) return rc; }
        rc = begin(f, f0, frames, first, lanes);
        if (rc) return rc;
        first += lanes;
      }
      for (uint32_t j = 0; j < flights; ++j) {  // drain in the order the launches were begun
        const uint32_t f = uint32_t((k + j) % flights);
        if (pending[f]) { rc = finish(f); if (rc) return rc; }
      }
      rc = flush_aside(f0, frames, h->bpt_flight[0].stream);
      if (rc) return rc;
      for (uint32_t f = 0; f < flights; ++f) {
        HIP_TRY(hipEventRecord(h->bpt_flight[f].done, h->bpt_flight[f].stream));
        HIP_TRY(hipStreamWaitEvent(stream, h->bpt_flight[f].done, 0));
      }
      w.frame = f0; w.frames = frames;
      HIP_TRY(bl.commit(p, w, stream));
    }
  }
  HIP_TRY(hipEventRecord(h->ev1, stream));
  HIP_TRY(mi::launch_finalize(h->partial, h->d_rgbn, width, height, win.x0, win.y0, win.w, win.h, 1, stream));
  HIP_TRY(hipEventRecord(h->ev2, stream));
  unsigned long long c[24];
  HIP_TRY(hipMemcpyAsync(c, h->d_counters, sizeof c, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipMemcpyAsync(rgbn_sum, h->d_rgbn, np * 16, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  if (c[15]) return fail(MI_ERR_UNSUPPORTED, "BPT: " + std::to_string(c[15]) + " light sub-paths exceeded the vertex slab");
  if (c[3] >= 4096ull) h->bpt_rays_per_path = float(double(c[0]) / double(c[3]));
  if (stats) {
    float t01 = 0.0f, t02 = 0.0f;
    HIP_TRY(hipEventElapsedTime(&t01, h->ev0, h->ev1));
    HIP_TRY(hipEventElapsedTime(&t02, h->ev0, h->ev2));
    std::memset(stats, 0, sizeof *stats);
    stats->num_basic_rays = c[0]; stats->num_shadow_rays = c[1]; stats->numeric_errors = c[2]; stats->num_paths = c[3];
    stats->trace_ms = t01; stats->gpu_ms = t02;
  }
  return MI_OK;
}

int mi_bpt_trace_paths(mi_pt_handle* h, uint32_t camera_id, uint32_t width, uint32_t height, uint32_t n, const uint32_t* pixel_xy,
                       const uint64_t* sample_index, uint64_t seed, float* out_radiance, float* out_splat_sum, uint32_t* out_counts3) {
  if (!h || (n && (!pixel_xy || !sample_index || !out_radiance))) return fail(MI_ERR_INVALID_ARGUMENT, "mi_bpt_trace_paths: null argument");
  if (n == 0) return MI_OK;
  for (uint32_t i = 0; i < n; ++i)
    if (pixel_xy[2 * i] >= width || pixel_xy[2 * i + 1] >= height) return fail(MI_ERR_INVALID_ARGUMENT, "pixel outside the image");
  HIP_TRY(hipSetDevice(h->device));
  mi::RenderParams p; mi::BptState w; uint32_t per_launch = 0;
  int rc = bpt_prepare(h, camera_id, width, height, n, p, w, &per_launch);
  if (rc) return rc;
  uint32_t *d_xy = nullptr, *d_c = nullptr; uint64_t* d_s = nullptr; float *d_r = nullptr, *d_sp = nullptr;
  struct Tmp { void* p[5]; ~Tmp() { for (void* q : p) if (q) hipFree(q); } } tmp{{nullptr, nullptr, nullptr, nullptr, nullptr}};
  rc = upload(&d_xy, pixel_xy, size_t(n) * 8); tmp.p[0] = d_xy; if (rc) return rc;
  rc = upload(&d_s, sample_index, size_t(n) * 8); tmp.p[1] = d_s; if (rc) return rc;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_r), size_t(n) * 12)); tmp.p[2] = d_r;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_sp), size_t(n) * 12)); tmp.p[3] = d_sp;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_c), size_t(n) * 12)); tmp.p[4] = d_c;
  p.win_w = width; p.win_h = height; p.seed = seed;
  p.list_xy = d_xy; p.list_sample = d_s; p.list_n = n; p.list_radiance = d_r;
  w.list_splat_sum = d_sp; w.list_counts3 = d_c;
  HIP_TRY(hipMemsetAsync(h->d_counters, 0, mi::kCounterWords * sizeof(unsigned long long), h->stream));
  for (uint64_t first = 0; first < n; first += per_launch) {
    w.first = uint32_t(first); w.lanes = uint32_t(n - first < per_launch ? n - first : per_launch);
    rc = bpt_launch(h, p, w, true, h->stream);
    if (rc) return rc;
  }
  unsigned long long c[24];
  HIP_TRY(hipMemcpyAsync(c, h->d_counters, sizeof c, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (c[15]) return fail(MI_ERR_UNSUPPORTED, "BPT: " + std::to_string(c[15]) + " light sub-paths exceeded the vertex slab");
  HIP_TRY(hipMemcpy(out_radiance, d_r, size_t(n) * 12, hipMemcpyDeviceToHost));
  if (out_splat_sum) HIP_TRY(hipMemcpy(out_splat_sum, d_sp, size_t(n) * 12, hipMemcpyDeviceToHost));
  if (out_counts3) HIP_TRY(hipMemcpy(out_counts3, d_c, size_t(n) * 12, hipMemcpyDeviceToHost));
  return MI_OK;
}

int mi_pt_bvh_info(mi_pt_handle* h, mi_bvh_info* out) {
  if (!h || !out) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_bvh_info: null argument");
  *out = h->info;
  return MI_OK;
}

int mi_pt_blob_download(mi_pt_handle* h, uint32_t offsets_f4[7], float* blob, size_t capacity_f4) {
  if (!h || !offsets_f4) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_blob_download: null argument");
  const mi::SceneView& v = h->sv;
  const uint32_t o[7] = {v.off_nodes, v.off_tris, v.off_shade, v.off_mats, v.off_lights, v.off_cdf, v.blob_f4};
  std::memcpy(offsets_f4, o, sizeof o);
  if (blob) {
    if (capacity_f4 < v.blob_f4) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_blob_download: buffer too small");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpy(blob, h->blob, size_t(v.blob_f4) * 16, hipMemcpyDeviceToHost));
  }
  return MI_OK;
}

int mi_pt_bvh_download(mi_pt_handle* h, mi_bvh_node* nodes, uint32_t* sorted_tri, uint64_t* morton) {
  if (!h) return fail(MI_ERR_INVALID_ARGUMENT, "mi_pt_bvh_download: null handle");
  HIP_TRY(hipSetDevice(h->device));
  static_assert(sizeof(mi_bvh_node) == 64, "mi_bvh_node must be 4 float4");
  if (nodes && h->sv.n_nodes) {
    HIP_TRY(hipMemcpy(nodes, h->blob + h->sv.off_nodes, size_t(h->sv.n_nodes) * 64, hipMemcpyDeviceToHost));
    std::vector<int32_t> plain(size_t(h->sv.n_nodes) * 2);  // the builder's links, Morton positions (the device copy carries pair leaves over re-ordered streams)
    HIP_TRY(hipMemcpy(plain.data(), h->plain_links, plain.size() * 4, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < h->sv.n_nodes; ++i) { nodes[i].link0 = plain[2 * size_t(i)]; nodes[i].link1 = plain[2 * size_t(i) + 1]; }
  }
  if (sorted_tri) HIP_TRY(hipMemcpy(sorted_tri, h->d_sorted_tri, size_t(h->sv.n_tris) * 4, hipMemcpyDeviceToHost));
  if (morton) HIP_TRY(hipMemcpy(morton, h->d_morton, size_t(h->sv.n_tris) * 8, hipMemcpyDeviceToHost));
  return MI_OK;
}

}  // extern "C"
