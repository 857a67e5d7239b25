/*
 * mi_pt.h — C ABI of the MI355X path-tracing integrator (libmi_pt.so).
 *
 * This is the drop-in boundary behind the reference renderer's `haste::Technique`
 * interface for the PT path.  Every entry point names the reference interface it
 * replaces (file:line relative to the reference tree).  Plain C: pointers, sizes,
 * POD structs.  No C++ types, no torch types, never throws.
 *
 * Conventions shared with the reference:
 *   - mat3 values are column-major, 9 floats: {col0.xyz, col1.xyz, col2.xyz}.
 *   - a tangent frame is [col0 = bitangent | col1 = shading normal | col2 = tangent]
 *     (SurfacePoint.hpp:46-50); local shading space has y = normal.
 *   - material_id = (material_index << 2) | entity_type (SurfacePoint.hpp:8-21),
 *     entity_type: camera 0, mesh 1, light 2, empty 3; "no hit" = UINT32_MAX.
 *   - images are row-major [H][W][4] float, row 0 = BOTTOM of the image
 *     (Cameras.cpp:124 maps y = 0 to view-space -1), channels = (R sum, G sum, B sum, denom).
 */
#ifndef MI_PT_H
#define MI_PT_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden */
#endif

#define MI_PT_ABI_VERSION 2 /* 2: + mi_pt_render_frames_async / mi_pt_render_async / mi_pt_wait / mi_pt_wait_add / mi_view_add_frame / mi_pt_last_launch (additive) */

/* ---- error codes (reference: C++ exceptions, runtime_assert.cpp:7-11, Scene.cpp:55-57) ---- */
enum {
  MI_OK = 0,
  MI_ERR_INVALID_ARGUMENT = -1, /* runtime_assert analogue: bad index / null / size mismatch        */
  MI_ERR_NO_DEVICE = -2,        /* HIP device missing or HIP call failed (message has the detail)  */
  MI_ERR_OUT_OF_MEMORY = -3,
  MI_ERR_IO = -4,               /* loader.cpp:469 "Cannot load scene" analogue                      */
  MI_ERR_UNSUPPORTED = -5,
  MI_ERR_INTERNAL = -6
};

/* ---- entity / bsdf tags ---- */
enum { MI_ENTITY_CAMERA = 0, MI_ENTITY_MESH = 1, MI_ENTITY_LIGHT = 2, MI_ENTITY_EMPTY = 3 };

enum {
  MI_BSDF_CAMERA = 0,       /* CameraBSDF        BSDF.cpp:195-235  (never sampled by PT)            */
  MI_BSDF_DIFFUSE = 1,      /* DiffuseBSDF       BSDF.cpp:237-304                                  */
  MI_BSDF_PHONG = 2,        /* PhongBSDF         BSDF.cpp:306-391                                  */
  MI_BSDF_REFLECTION = 3,   /* ReflectionBSDF    BSDF.cpp:450-465                                  */
  MI_BSDF_TRANSMISSION = 4, /* TransmissionBSDF  BSDF.cpp:467-504                                  */
  MI_BSDF_LIGHT = 5,        /* LightBSDF         BSDF.cpp:73-114                                   */
  MI_BSDF_SUN = 6           /* sun_light_bsdf    BSDF.cpp:164-193                                  */
};

/* One entry of Materials::bsdfs (Materials.hpp:14-27), flattened.  48 bytes. */
typedef struct mi_material {
  uint32_t type;        /* MI_BSDF_*                                                   */
  float diffuse[3];     /* DiffuseBSDF::_diffuse / PhongBSDF::_diffuse                  */
  float specular[3];    /* PhongBSDF::_specular                                        */
  float power;          /* PhongBSDF::_power                                           */
  float ior_internal;   /* TransmissionBSDF ctor arg 0 (loader.cpp:381-382)            */
  float ior_external;   /* TransmissionBSDF ctor arg 1 (always 1.0 in the loader)      */
  uint32_t light_id;    /* LightBSDF::_light_id / sun_light_bsdf::_light_id            */
  uint32_t reserved;
} mi_material;

/* AreaLight (AreaLights.hpp:41-58).  80 bytes. */
typedef struct mi_light {
  float position[3];
  float tangent[9];     /* column-major; col1 = emission normal (AreaLights.cpp:78-85) */
  float size[2];
  float exitance[3];
  uint32_t diffuse;     /* 1 = LightBSDF (area), 0 = sun_light_bsdf (AreaLights.cpp:26-36) */
  uint32_t material_id; /* encoded, entity_type light                                   */
  uint32_t reserved;
} mi_light;

/* Cameras::Desc as filled by Cameras::addCameraFovX (Cameras.cpp:7-28).  40 bytes. */
typedef struct mi_camera {
  float position[3];
  float direction[3];
  float up[3];
  float fovx;           /* radians, full horizontal angle */
} mi_camera;

/*
 * Flattened haste::Scene (Scene.hpp:27-76): meshes (incl. the light quads the loader
 * appends, loader.cpp:448), materials, lights, cameras.  All arrays are copied by
 * mi_pt_create; the caller may free them afterwards.
 *   mesh m owns triangles [mesh_tri_offset[m], mesh_tri_offset[m+1]); Embree geomID == m
 *   (Scene.cpp:59-63); indices are GLOBAL vertex indices.
 */
typedef struct mi_scene_desc {
  uint32_t n_vertices;
  uint32_t n_triangles;
  uint32_t n_meshes;
  uint32_t n_materials;
  uint32_t n_lights;
  uint32_t n_cameras;
  const float* positions;            /* [n_vertices][3]    Mesh::vertices (AreaLights.hpp:11-17) */
  const float* tangents;             /* [n_vertices][9]    Mesh::tangents                        */
  const uint32_t* indices;           /* [n_triangles][3]   Mesh::indices (+ vertex base)         */
  const uint32_t* mesh_tri_offset;   /* [n_meshes + 1]                                           */
  const uint32_t* mesh_material_id;  /* [n_meshes]         Mesh::material_id (encoded)           */
  const mi_material* materials;      /* [n_materials]                                            */
  const mi_light* lights;            /* [n_lights]                                               */
  const mi_camera* cameras;          /* [n_cameras]                                              */
  float bounding_sphere[4];          /* center xyz, radius (loader.cpp:408-432)                  */
} mi_scene_desc;

/* PathTracing ctor arguments (PT.cpp:5-13, make_technique.cpp:132-141, Options.hpp:30-37). */
typedef struct mi_pt_params {
  uint64_t max_path;   /* --max-path; UINT64_MAX / PTRDIFF_MAX = unlimited (roulette-terminated); the device cuts a
                          path after 2^20 edges so that roulette = 1 in a lossless closed scene cannot hang the GPU */
  float beta;          /* --beta      default 1.0                                              */
  float roulette;      /* --roulette  default 0.9                                              */
  float lights;        /* --no-lights => 0, default 1.0                                        */
  uint32_t min_subpath;/* PT.hpp:24, always 3 in the reference                                 */
} mi_pt_params;

/* subimage_view_t window (ImageView.hpp:10-67): pixels [x0, x0+w) x [y0, y0+h). w == 0 => full image. */
typedef struct mi_window {
  uint32_t x0, y0, w, h;
} mi_window;

/* What Technique::render needs to fill statistics_t (Technique.cpp:55-67). */
typedef struct mi_pt_stats {
  uint64_t num_paths;        /* camera samples started (pixels in window * spp)                 */
  uint64_t num_basic_rays;   /* closest-hit rays  == Scene::_numIntersectRays delta (Scene.cpp:200) */
  uint64_t num_shadow_rays;  /* any-hit rays      == Scene::_numOccludedRays delta  (Scene.cpp:177) */
  uint64_t numeric_errors;   /* samples dropped as non-finite (Technique.cpp:224-230)           */
  double gpu_ms;             /* device time of the render kernels of this call (HIP events)     */
  double trace_ms;           /* device time of the dominant (path tracing) kernel alone         */
  /* traversal work counters, filled only after mi_pt_set_instrumented(h, 1) (zero otherwise);
   * they feed the roofline's algorithmic bytes (64 B per node, 48 B per triangle) */
  uint64_t nodes_closest, tris_closest;  /* BVH nodes fetched / triangles tested by closest-hit rays */
  uint64_t nodes_shadow, tris_shadow;    /* the same for shadow rays                                 */
  uint64_t num_hits;                     /* closest-hit rays that hit a surface                      */
  uint64_t wave_steps_closest;           /* sum over waves and loop trips of the SLOWEST lane's traversal steps */
  uint64_t wave_steps_shadow;            /* (nodes + triangles); 64 x this vs the per-lane totals = SIMD efficiency.  With the dynamic-fetch
                                            traversal (mi_pt_launch_info::dynamic_fetch) closest-hit and shadow rays share one loop:
                                            wave_steps_closest counts its trips, wave_steps_shadow is 0 */
  /* diagnostic builds only (-DMI_PHASE_TIMING): shader cycles summed over waves per phase of the loop trip:
   * 0 regeneration, 1 closest-hit traversal, 2 querySurface + path logic, 3 NEE set-up, 4 shadow traversal,
   * 5 BSDF sample, 6 commit, 7 loop overhead.  Zero in the product build. */
  uint64_t phase_cycles[8];
  /* instrumented: loop bodies executed by waves (whatever the number of active lanes): closest-hit node steps,
   * closest-hit leaf phases, shadow node steps, shadow leaf phases */
  uint64_t wave_loop_bodies[4];
} mi_pt_stats;

typedef struct mi_pt_handle mi_pt_handle;

/* ------------------------------------------------------------------------------------------
 * Technique / PathTracing life cycle
 * ---------------------------------------------------------------------------------------- */

/* Replaces: PathTracing::PathTracing (PT.cpp:5-13) + Scene::buildAccelStructs (Scene.cpp:68-73,
 * Embree rtcCommit Scene.cpp:47-66).  Copies the scene to `device`, builds the LBVH there. */
int mi_pt_create(const mi_scene_desc* scene, const mi_pt_params* params, int device,
                 mi_pt_handle** out);

/* Replaces: Technique::~Technique (Technique.cpp:13) + rtcDeleteScene. */
void mi_pt_destroy(mi_pt_handle* h);

/* Replaces: Technique::render (Technique.cpp:15-77) for `spp` consecutive frames:
 * _trace_paths + _for_each_ray + PathTracing::_traceEye + _commit_images.
 * Writes (not adds) per-pixel (R,G,B sums, denom) of samples
 * [sample_offset, sample_offset + spp) into rgbn_sum ([height][width][4] floats, HOST memory,
 * row 0 = bottom); pixels outside `win` are written as zeros.  The caller adds the result into
 * its dvec4 view exactly like Technique.cpp:222-226.  Random streams are keyed on
 * (seed, pixel index, sample index), so disjoint sample ranges rendered anywhere sum to the
 * same image as one call. */
int mi_pt_render(mi_pt_handle* h, uint32_t camera_id, uint32_t width, uint32_t height,
                 mi_window win, uint32_t spp, uint64_t seed, uint64_t sample_offset,
                 float* rgbn_sum, mi_pt_stats* stats);

/* Same, result left in DEVICE memory (hipMalloc'ed / torch CUDA tensor storage of
 * height*width*4 floats) so it can go straight into an RCCL reduce — the collective
 * counterpart of merge_exr (Options.cpp:1340-1409).  `stream` is a hipStream_t (NULL = default);
 * the call is synchronous with respect to the host when stats != NULL. */
int mi_pt_render_device(mi_pt_handle* h, uint32_t camera_id, uint32_t width, uint32_t height,
                        mi_window win, uint32_t spp, uint64_t seed, uint64_t sample_offset,
                        float* rgbn_sum_device, void* stream, mi_pt_stats* stats);

/* Pixel-tile sharding of mi_pt_render / mi_pt_render_device across `world` processes (one per GPU):
 * the window is cut into the 32x32 tiles of Technique::_trace_paths (Technique.cpp:167, exec2d
 * threadpool.cpp:190-233), numbered row-major from the window's origin, and this handle renders
 * the tiles {t : t mod world == rank}; every other pixel is written as zeros (R, G, B and denom),
 * so the sum of the ranks' framebuffers — merge_exr (Options.cpp:1356-1358), one RCCL reduce — is
 * bit-identical to the unsharded render.  world <= 1 switches sharding off (the default).
 * The other way to shard, by sample ranges, needs no call: see sample_offset above. */
int mi_pt_set_tile_shard(mi_pt_handle* h, uint32_t rank, uint32_t world);

/* Several GPUs from ONE host process — the reference is a single process whose Technique::_trace_paths
 * (Technique.cpp:163-192) deals 32x32 tiles to the threads of its pool (exec2d, threadpool.cpp:190-233); here the
 * same tiles are dealt to `n_handles` handles created from the same scene on different devices
 * (tile t goes to handle t mod n_handles, like mi_pt_set_tile_shard).  All devices render concurrently on
 * their own streams; the caller's framebuffer receives every tile from its owner, so the result is
 * bit-identical to mi_pt_render on one handle, also at spp = 1 per call (the cadence of Application::render,
 * Application.cpp:66).  `stats` receives the sums of the ray / path / error counts and the slowest device's
 * times.  Any tile shard set on the handles is ignored for the call and restored afterwards. */
int mi_pt_render_multi(mi_pt_handle* const* handles, uint32_t n_handles, uint32_t camera_id, uint32_t width,
                       uint32_t height, mi_window win, uint32_t spp, uint64_t seed, uint64_t sample_offset,
                       float* rgbn_sum, mi_pt_stats* stats);
/* Where the last mi_pt_render_multi call of this thread assembled the frame: 1 = on the first handle's device (peer reads of the owners' tiles,
 * xGMI between GPUs of one node; one copy of the merged frame to the host), 0 = on the host from every device's framebuffer (no peer access, or
 * MI_PT_MULTI_HOST_MERGE=1), -1 = no call yet.  The in-process form of `master merge` (Options.cpp:1340-1409). */
int mi_pt_last_multi_merge(void);

/* One PROCESS per GPU: the sum of the per-GPU framebuffers as ONE RCCL collective over xGMI — the in-memory form of `master merge`
 * (merge_exr, Options.cpp:1340-1409: dst = fst + snd on R, G, B, denom), for hosts that run one rank per device (sample ranges through
 * sample_offset, or pixel tiles through mi_pt_set_tile_shard; both sum to the one-GPU image).  librccl.so is opened at the first call
 * (dlopen; MI_PT_RCCL_LIB overrides the name) and is NOT a link-time dependency: without it these calls return MI_ERR_UNSUPPORTED.
 *   rank 0:      mi_pt_reduce_unique_id(id)  and hands the 128 bytes to the other ranks (file, socket, MPI ... — the host's business)
 *   every rank:  mi_pt_reduce_init(h, id, rank, world)                   collective: ncclCommInitRank on the handle's device
 *                mi_pt_render_device(h, ..., rgbn_device, stream, ...)    this rank's share
 *                mi_pt_reduce_rgbn(h, rgbn_device, W, H, root, stream)     in place, FP32 sum of W*H*4 values; root < 0: all-reduce, else
 *                                                                          reduce to `root`; stream NULL = the handle's own, synchronised
 *                mi_pt_reduce_finalize(h)                                 (mi_pt_destroy does it too) */
#define MI_PT_REDUCE_ID_BYTES 128
int mi_pt_reduce_available(void); /* 1 if librccl.so could be opened */
int mi_pt_reduce_unique_id(unsigned char id[MI_PT_REDUCE_ID_BYTES]);
int mi_pt_reduce_init(mi_pt_handle* h, const unsigned char id[MI_PT_REDUCE_ID_BYTES], uint32_t rank, uint32_t world);
int mi_pt_reduce_rgbn(mi_pt_handle* h, float* rgbn_sum_device, uint32_t width, uint32_t height, int root, void* stream);
int mi_pt_reduce_finalize(mi_pt_handle* h);

/* ------------------------------------------------------------------------------------------
 * Frames in flight — the reference's cadence.  Application::render calls Technique::render ONCE PER SAMPLE
 * (Application.cpp:41-79; the batch loop is Framework::runBatch, framework.cpp:426-437) and reads the dvec4 view only
 * when it saves (Application::_save) or compares with a reference image.  A frame of 512 x 512 is 4 096 waves: called
 * synchronously (mi_pt_render, spp = 1) the GPU idles while the host copies 16 B per pixel and adds it to the view.
 *
 * mi_pt_render_frames_async enqueues `n_frames` consecutive frames (samples first_sample .. first_sample + n_frames - 1 of the
 * window, ONE sample per pixel each, exactly what n_frames calls of mi_pt_render(spp = 1) return) as ONE launch and returns at
 * once with one ticket per frame; inside the launch a wave that has finished a path of frame f starts the same pixel's path of
 * frame f + 1, so the chip stays full.  The frames land in pinned host buffers owned by the handle.  mi_pt_wait blocks until
 * the ticket's frame is complete and hands out its buffer ([height][width][4] floats, row 0 = bottom) and its statistics (ray and
 * error counts are exact per frame; the device time of the launch is shared evenly by its frames).  mi_pt_render_async is the
 * same for ONE frame of `spp` samples.  Up to MI_PT_BATCHES_IN_FLIGHT calls may be pending; every ticket of a call must be
 * waited for before its slot is reused (the enqueue fails with MI_ERR_INVALID_ARGUMENT otherwise), and a buffer stays valid
 * until then.  The adapter's loop with batches of B frames: render(k) = [k mod B == 0: enqueue batch k / B + 2] ->
 * wait_add(frame k): the next batches render and cross PCIe while the host adds the frames of this one, and the view after call
 * k holds exactly frames 0..k, so `--num-samples`, snapshots and `continue` (Application.cpp:226-229,245) see what they saw
 * before.  Random streams depend only on (seed, pixel, sample index): rendering ahead changes nothing in the image.
 * ---------------------------------------------------------------------------------------- */
#define MI_PT_MAX_FRAMES_PER_BATCH 16
#define MI_PT_BATCHES_IN_FLIGHT 3
int mi_pt_render_frames_async(mi_pt_handle* h, uint32_t camera_id, uint32_t width, uint32_t height, mi_window win,
                              uint32_t n_frames, uint64_t seed, uint64_t first_sample, uint64_t* tickets /*[n_frames]*/);
int mi_pt_render_async(mi_pt_handle* h, uint32_t camera_id, uint32_t width, uint32_t height, mi_window win,
                       uint32_t spp, uint64_t seed, uint64_t sample_offset, uint64_t* ticket);
int mi_pt_wait(mi_pt_handle* h, uint64_t ticket, const float** rgbn_sum, mi_pt_stats* stats);
/* mi_pt_wait + Technique::_commit_images for the PT path (Technique.cpp:215-236) in one call: view[p] += dvec4(rgbn[p]) for the
 * pixels of the frame's window, `view` being subimage_view_t's dvec4 data ([height][width][4] doubles, ImageView.hpp:10-67).
 * The add is dealt to host threads of the library the way exec2d deals tiles (threadpool.cpp:190-233): on one core it costs as
 * much as the frame's kernel (16 B read + 64 B read-modify-write per pixel).  MI_PT_HOST_THREADS overrides the thread count (<= 8). */
int mi_pt_wait_add(mi_pt_handle* h, uint64_t ticket, double* view, mi_pt_stats* stats);
/* The add alone, for a frame the caller already holds (mi_pt_render / mi_pt_wait): view[p] += dvec4(rgbn[p]) over `win`
 * (w == 0: the whole image), on the same host threads. */
int mi_view_add_frame(const float* rgbn, double* view, uint32_t width, uint32_t height, mi_window win);

/* What the last mi_pt_render / mi_pt_render_device / mi_pt_render_async call launched (for measurement: bench.py
 * prices the launch's compulsory HBM bytes with it). */
typedef struct mi_pt_launch_info {
  uint32_t kernel;          /* MI_PT_KERNEL_* that ran                                                        */
  uint32_t n_blocks;        /* workgroups of 256 threads                                                      */
  uint32_t n_chunks;        /* sample chunks per pixel tile (one wave owns a tile x chunk)                    */
  uint32_t chunk_spp;       /* samples per pixel in a chunk                                                   */
  uint32_t lds_bytes;       /* dynamic LDS per workgroup                                                      */
  uint32_t wide_nodes;      /* 0 = 32-byte quantised binary nodes, 1 = 64-byte wide nodes, 2 = 64-byte float nodes */
  uint32_t features;        /* kFeat* bits of the kernel variant; bit 31: the approximate-arithmetic build ran (MI_PT_FAST=1, opt-in) */
  uint32_t lds_tables;      /* 1: materials, lights and the light CDF were staged into LDS by every workgroup */
  uint32_t frame_tiles_per_wave; /* > 0: frame mode (spp == 1): paths write the framebuffer directly, a wave owns this many 8x8 tiles */
  uint32_t frames;               /* frame mode: frames of the launch */
  uint32_t dynamic_fetch;        /* 1: closest-hit and shadow rays shared one traversal loop with dynamic fetch (scenes read from HBM) */
  uint32_t flat_leaves;          /* > 0: the flat leaf list ran instead of the tree walk (LDS-resident scenes): leaf boxes per ray */
  uint64_t partial_bytes;   /* FP64 partial sums written by the path kernel and read by pt_finalize           */
  uint64_t scene_bytes;     /* scene blob (+ quantised node copies) resident in HBM                           */
} mi_pt_launch_info;
int mi_pt_last_launch(mi_pt_handle* h, mi_pt_launch_info* out);

/* Number of HIP devices visible to the process (0 without a GPU; never fails). */
int mi_pt_device_count(void);

/* Replaces: reading the message of std::logic_error / std::runtime_error thrown by the
 * reference (runtime_assert.cpp:7-11).  Thread-local, never NULL. */
const char* mi_pt_last_error(void);

int mi_pt_abi_version(void);
/* Hash of the sources, headers and compiler flags this library was built from (master_amd/build.py: source_hash).  __graft_entry__.build()
 * rebuilds unless it equals the hash of the tree next to the library — a stale prebuilt library is never reused silently. */
const char* mi_pt_build_id(void);

/* Kernel variant selection (for measurement; default MI_PT_KERNEL_AUTO). */
enum {
  MI_PT_KERNEL_AUTO = 0,
  MI_PT_KERNEL_MEGA_LDS = 1,     /* megakernel, scene blob staged into LDS (scenes up to 48 KB)            */
  MI_PT_KERNEL_MEGA_GLOBAL = 2,  /* megakernel, scene read from HBM                                        */
  MI_PT_KERNEL_WAVEFRONT = 3     /* wavefront pipeline: path state in HBM, one kernel per ray cast / vertex */
};
int mi_pt_set_kernel(mi_pt_handle* h, int kernel);
int mi_pt_get_kernel(mi_pt_handle* h); /* variant AUTO resolves to for this scene */
/* 1: render calls run the instrumented kernel variant (same results, plus visit counters). */
int mi_pt_set_instrumented(mi_pt_handle* h, int on);

/* ------------------------------------------------------------------------------------------
 * Scene services exposed for parity tests (each is a batched form of one Scene method)
 * ---------------------------------------------------------------------------------------- */

/* SurfacePoint (SurfacePoint.hpp:37-63), 64 bytes. */
typedef struct mi_surface_point {
  float position[3];
  float gnormal[3];
  float tangent[9];
  uint32_t material_id;
} mi_surface_point;

/* Replaces: Scene::intersect (Scene.cpp:182-203) + Scene::querySurface (Scene.cpp:80-126),
 * n rays at once.  origins[i] supplies position + gnormal (+ material_id, ignored);
 * out_t[i] (optional) receives Embree's tfar, out_prim[i] (optional) the global triangle id
 * (UINT32_MAX on a miss).  All pointers are HOST memory. */
int mi_pt_intersect(mi_pt_handle* h, uint32_t n, const mi_surface_point* origins,
                    const float* directions /*[n][3]*/, mi_surface_point* out_hits,
                    float* out_t, uint32_t* out_prim);

/* Replaces: Scene::occluded (Scene.cpp:151-180): 1.0 = visible, 0.0 = blocked. */
int mi_pt_occluded(mi_pt_handle* h, uint32_t n, const mi_surface_point* origins,
                   const mi_surface_point* targets, float* out_visibility);

/* Replaces: Technique::_for_each_ray's shoot() + PathTracing::_traceEye (Technique.cpp:321-338,
 * PT.cpp:15-98) for an explicit list of (pixel x, pixel y, sample index): per-path radiance
 * (float[n][3]) and per-path ray counts (uint32[n][2] = basic, shadow; optional). */
int mi_pt_trace_paths(mi_pt_handle* h, uint32_t camera_id, uint32_t width, uint32_t height,
                      uint32_t n, const uint32_t* pixel_xy /*[n][2]*/,
                      const uint64_t* sample_index /*[n]*/, uint64_t seed,
                      float* out_radiance, uint32_t* out_ray_counts);

/* ------------------------------------------------------------------------------------------
 * LBVH (replaces Embree's rtcCommit, Scene.cpp:52-65) — download for parity tests
 * ---------------------------------------------------------------------------------------- */

/* 64-byte BVH2 node: child boxes + links.  link >= 0: internal node index; link < 0: leaf,
 * ~link = position in the Morton-sorted triangle order. */
typedef struct mi_bvh_node {
  float lo0[3]; int32_t link0;
  float hi0[3]; int32_t link1;
  float lo1[3]; uint32_t parent;
  float hi1[3]; uint32_t reserved;
} mi_bvh_node;

typedef struct mi_bvh_info {
  uint32_t n_triangles;
  uint32_t n_nodes;      /* n_triangles - 1 internal nodes (0 when n_triangles == 1)   */
  uint32_t max_depth;    /* longest root-to-leaf path in nodes                          */
  uint32_t stack_entries;/* traversal stack capacity the kernels were sized with        */
  float scene_lo[3], scene_hi[3];
  double build_ms;       /* device time of the build kernels                            */
  uint32_t builder;      /* 1 = PLOC (default), 0 = Karras LBVH (env MI_PT_BVH=lbvh)    */
  uint32_t build_rounds; /* PLOC merge rounds                                           */
} mi_bvh_info;

/* ------------------------------------------------------------------------------------------
 * Bidirectional path tracing (BPTBase<Beta>, BPT.cpp:13-337; created by make_bpt_technique, make_technique.cpp:112-130)
 * on the same handle: roulette and beta come from mi_pt_params (beta 0, 1, 2 = FixedBeta<0|1|2>, anything else VariableBeta,
 * Beta.hpp); max_path and lights are not used by BPT.  SURVEY.md 8(f) rank 4 — first device version.
 *
 * mi_bpt_render: `spp` frames of the view window (w == 0: the whole image; splats that land outside the window are dropped, as
 * _commit_images only commits the window).  Per frame every pixel traces one light sub-path and one eye sub-path
 * (BPT.cpp:13-101); connections to the camera are splatted into the frame's light image (Technique.cpp:276-306) and the sum
 * light + eye of a pixel passes the finite filter as one sample (Technique.cpp:194-244).  rgbn_sum as in mi_pt_render.
 * mi_bpt_trace_paths: parity hook — per listed (pixel, sample) the eye-image radiance, the sum of its light-image splats and
 * the counts (closest-hit rays, shadow rays, splats). */
/* Technique::set_sky_gradient (Technique.cpp:90-93; --sky-horizon / --sky-zenith / --blue-sky, Options.cpp:74-76): the colour a
 * camera ray that leaves the scene returns in BPT (BPT.cpp:49-51); zero by default.  PT ignores the sky (PT.cpp:28,49-51). */
int mi_bpt_set_sky(mi_pt_handle* h, const float horizon[3], const float zenith[3]);
int mi_bpt_render(mi_pt_handle* h, uint32_t camera_id, uint32_t width, uint32_t height, mi_window win, uint32_t spp, uint64_t seed,
                  uint64_t sample_offset, float* rgbn_sum, mi_pt_stats* stats);
int mi_bpt_trace_paths(mi_pt_handle* h, uint32_t camera_id, uint32_t width, uint32_t height, uint32_t n, const uint32_t* pixel_xy,
                       const uint64_t* sample_index, uint64_t seed, float* out_radiance, float* out_splat_sum, uint32_t* out_counts3);

int mi_pt_bvh_info(mi_pt_handle* h, mi_bvh_info* out);
/* The device scene blob as float4 records (layout: master_amd/csrc/device/layout.h): sections in
 * order nodes | triangles | shading | materials | lights | light cdf.  offsets_f4[7] receives the
 * six section offsets and the total size (float4 units); blob may be NULL to query sizes only. */
int mi_pt_blob_download(mi_pt_handle* h, uint32_t offsets_f4[7], float* blob, size_t capacity_f4);
/* nodes: [n_nodes]; sorted_tri: [n_triangles] global triangle id at each sorted position;
 * morton: [n_triangles] 63-bit codes in sorted order.  Any pointer may be NULL.
 * The links returned are the builder's (leaf = ~Morton position).  The device copy of the nodes (mi_pt_blob_download) carries pair leaves
 * instead: a node over two triangles is replaced in its parent by one leaf link, positions refer to streams in which the partner follows
 * its triangle, bit 30 of ~link marks the pair (master_amd/csrc/device/layout.h). */
int mi_pt_bvh_download(mi_pt_handle* h, mi_bvh_node* nodes, uint32_t* sorted_tri,
                       uint64_t* morton);

/* ------------------------------------------------------------------------------------------
 * Host-side camera helpers (Cameras.cpp) — exported so tests can pin them against the
 * reference's own camera tests (unit_tests/Cameras.test.cpp:22-44, Cameras.cpp:164-189).
 * ---------------------------------------------------------------------------------------- */
typedef struct mi_camera_frame {
  float view_to_world[9];  /* Cameras::view_to_world_mat3 (Cameras.cpp:104-106)  */
  float world_to_view[9];  /* Cameras::world_to_view_mat3 (Cameras.cpp:108-110)  */
  float position[3];
  float focal_length_y;    /* Cameras::focal_length_y (Cameras.cpp:112-116)      */
  float fovy;              /* Cameras::fovy (Cameras.cpp:81-88)                  */
} mi_camera_frame;

int mi_camera_setup(const mi_camera* cam, float aspect, mi_camera_frame* out);
/* ray_direction (Cameras.cpp:120-127), view space. */
void mi_camera_ray_direction(float px, float py, float res_x, float res_y, float focal_length_y,
                             float out_dir[3]);
/* pixel_position (Cameras.cpp:134-144). */
void mi_camera_pixel_position(const float dir[3], float res_x, float res_y, float focal_length_y,
                              float out_xy[2]);

/* ------------------------------------------------------------------------------------------
 * Scene files: own .blend reader (replaces loadScene, loader.cpp:458-487, whose assimp fork is
 * not available) and a flat binary scene container (.miscene) so scenes travel without the
 * reference tree.
 * ---------------------------------------------------------------------------------------- */
typedef struct mi_blend_options {
  float diffuse_scale_by_ref;  /* 1 => diffuse colour *= Material.ref   (fork behaviour unpinned) */
  float specular_scale_by_spec;/* 1 => specular colour *= Material.spec                            */
  float lamp_energy_scale;     /* exitance = rgb * energy * scale (loader.cpp:434-456).  DEFAULT 1 = stock assimp's lamp units: the
                                  normalised models/TestCase*.blend then average 1.000.  0.01 reproduces the constant of the reference's own
                                  protocol, `expected = [0.01]*3` (unit_test.py:77-83): the same models average 0.0100 +- 0.0002
                                  (tests/test_oracle_bpt.py).  Which one the assimp fork implements is unpinned (it cannot be run here); the
                                  drop-in path is unaffected (there haste::Scene comes from the reference's own loader).  <= 0 means 1. */
  uint32_t reserved;
} mi_blend_options;

typedef struct mi_scene mi_scene; /* owning container around a mi_scene_desc */

int mi_scene_load_blend(const char* path, const mi_blend_options* opts, mi_scene** out);
int mi_scene_load(const char* path, mi_scene** out);           /* .miscene */
int mi_scene_save(const mi_scene* scene, const char* path);    /* .miscene */
int mi_scene_from_desc(const mi_scene_desc* desc, mi_scene** out); /* deep copy */
const mi_scene_desc* mi_scene_get_desc(const mi_scene* scene);
const char* mi_scene_material_name(const mi_scene* scene, uint32_t i);
const char* mi_scene_mesh_name(const mi_scene* scene, uint32_t i);
void mi_scene_free(mi_scene* scene);

/* ------------------------------------------------------------------------------------------
 * EXR output in the reference's layout (replaces save_exr / load_exr, exr.cpp:177-232,245-297):
 * float channels R,G,B,denom, rows flipped (exr.cpp:207-214), metadata as string attributes.
 * ---------------------------------------------------------------------------------------- */
int mi_exr_save_rgbn(const char* path, uint32_t width, uint32_t height, const float* rgbn,
                     uint32_t n_meta, const char* const* meta_keys, const char* const* meta_values);
int mi_exr_load_rgbn(const char* path, uint32_t* width, uint32_t* height, float** rgbn /* free with mi_free */);
void mi_free(void* p);

/* rms_abs_errors (ImageView.cpp:60-85): image = rgbn sums ([h][w][4]), reference = [h][w][3]. */
int mi_rms_abs_errors(const float* rgbn, const float* reference_rgb, uint32_t width,
                      uint32_t height, float* rms, float* abs_err);
/* The same over the caller's dvec4 view itself (what ImageView.cpp:60-85 takes: image_view_t<dvec4>): view = [h][w][4] double sums. */
int mi_rms_abs_errors_view(const double* view, const float* reference_rgb, uint32_t width,
                           uint32_t height, float* rms, float* abs_err);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* MI_PT_H */
