/* cadence.c — the reference's frame loop against libmi_pt.so, synchronous and with frames in flight.
 *
 * Application::render (Application.cpp:41-79) calls Technique::render once per sample; Framework::runBatch
 * (framework.cpp:426-437) does nothing else between two calls.  This program is that loop as the GpuPathTracing adapter runs
 * it (integration/GpuPathTracing.cpp): per frame `view[p] += dvec4(rgbn[p])` (Technique.cpp:222-226) on the host, with
 *   sync   one mi_pt_render(spp = 1) per frame: kernel, copy and host add run one after the other;
 *   async  mi_pt_render_frames_async / mi_pt_wait: batches of B frames per launch, two batches ahead: the next frames render and
 *          cross PCIe while the host adds frame k (one host thread, like the loop above);
 *   async+ the same with mi_pt_wait_add: the add runs on the library's host threads (what the adapter calls).
 * All loops produce the same dvec4 view bit for bit (checked here).  Prints one JSON line.
 *
 *   cc -O2 -std=c11 -I include examples/cadence.c -o cadence master_amd/libmi_pt.so -Wl,-rpath,$PWD/master_amd
 *   ./cadence scenes/CornellBoxDiffuse.miscene 512 512 400 [max_path] [frames per launch, default 8 up to 1 Mpixel, else 4]
 */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "mi_pt.h"

static double now_s(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

/* _commit_images for the PT path: the frame's sums into the dvec4 view */
static void add_frame(double* view, const float* rgbn, size_t n4) {
  for (size_t i = 0; i < n4; ++i) view[i] += (double)rgbn[i];
}

#define CHECK(x) do { if ((x) != MI_OK) { fprintf(stderr, "%s: %s\n", #x, mi_pt_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: %s scene.miscene width height frames [max_path (0 = unlimited)] [frames per launch]\n", argv[0]); return 2; }
  const unsigned width = (unsigned)atoi(argv[2]), height = (unsigned)atoi(argv[3]), frames = (unsigned)atoi(argv[4]);
  mi_pt_params params = {UINT64_MAX >> 1, 1.0f, 0.9f, 1.0f, 3};
  if (argc > 5 && atoll(argv[5]) > 0) params.max_path = (uint64_t)atoll(argv[5]);
  unsigned batch = (size_t)width * height <= (1u << 20) ? 8 : 4;  /* small frames: more of them per launch (a 512 x 512 frame is 1.3 rounds of waves) */
  if (argc > 6) batch = (unsigned)atoi(argv[6]);
  if (batch < 1 || batch > MI_PT_MAX_FRAMES_PER_BATCH) { fprintf(stderr, "batch must be in [1, %d]\n", MI_PT_MAX_FRAMES_PER_BATCH); return 2; }
  mi_scene* scene = NULL;
  CHECK(mi_scene_load(argv[1], &scene));
  mi_pt_handle* h = NULL;
  CHECK(mi_pt_create(mi_scene_get_desc(scene), &params, 0, &h));
  const size_t n4 = (size_t)width * height * 4;
  float* rgbn = (float*)malloc(n4 * sizeof(float));
  double* view_s = (double*)calloc(n4, sizeof(double));
  double* view_a = (double*)calloc(n4, sizeof(double));
  const mi_window whole = {0, 0, 0, 0};
  const uint64_t seed = 0x5EED;
  mi_pt_stats st;

  /* warm-up: first launch, buffers, pinned memory */
  CHECK(mi_pt_render(h, 0, width, height, whole, 1, seed, 1u << 30, rgbn, &st));
  for (int j = 0; j < MI_PT_BATCHES_IN_FLIGHT; ++j) {
    uint64_t t[MI_PT_MAX_FRAMES_PER_BATCH]; const float* p;
    CHECK(mi_pt_render_frames_async(h, 0, width, height, whole, batch, seed, 1u << 30, t));
    for (unsigned f = 0; f < batch; ++f) CHECK(mi_pt_wait(h, t[f], &p, &st));
  }

  /* ---- synchronous loop ---- */
  double t0 = now_s(), dev_ms_s = 0.0, add_s = 0.0;
  unsigned long long rays_s = 0;
  for (unsigned k = 0; k < frames; ++k) {
    CHECK(mi_pt_render(h, 0, width, height, whole, 1, seed, k, rgbn, &st));
    const double a0 = now_s();
    add_frame(view_s, rgbn, n4);
    add_s += now_s() - a0;
    dev_ms_s += st.gpu_ms; rays_s += st.num_basic_rays;
  }
  const double sync_s = now_s() - t0;

  /* ---- frames in flight: batches of B frames per launch, MI_PT_BATCHES_IN_FLIGHT - 1 batches ahead; the host adds with one thread ---- */
  const unsigned B = batch, n_batches = (frames + B - 1) / B;
  uint64_t (*tickets)[MI_PT_MAX_FRAMES_PER_BATCH] = calloc(n_batches + MI_PT_BATCHES_IN_FLIGHT, sizeof *tickets);
#define ENQUEUE(j) do { if ((j) < n_batches) { const unsigned k0_ = (j) * B, n_ = frames - k0_ < B ? frames - k0_ : B; \
                        CHECK(mi_pt_render_frames_async(h, 0, width, height, whole, n_, seed, k0_, tickets[j])); } } while (0)
  t0 = now_s();
  double dev_ms_a = 0.0;
  unsigned long long rays_a = 0;
  for (unsigned j = 0; j + 1 < MI_PT_BATCHES_IN_FLIGHT; ++j) ENQUEUE(j);
  for (unsigned k = 0; k < frames; ++k) {
    const float* p = NULL;
    if (k % B == 0) ENQUEUE(k / B + MI_PT_BATCHES_IN_FLIGHT - 1);
    CHECK(mi_pt_wait(h, tickets[k / B][k % B], &p, &st));
    add_frame(view_a, p, n4);
    dev_ms_a += st.gpu_ms; rays_a += st.num_basic_rays;
  }
  const double async_s = now_s() - t0;

  /* ---- the same with mi_pt_wait_add: the add runs on the library's host threads (what the adapter calls) ---- */
  double* view_b = (double*)calloc(n4, sizeof(double));
  t0 = now_s();
  unsigned long long rays_b = 0;
  for (unsigned j = 0; j + 1 < MI_PT_BATCHES_IN_FLIGHT; ++j) ENQUEUE(j);
  for (unsigned k = 0; k < frames; ++k) {
    if (k % B == 0) ENQUEUE(k / B + MI_PT_BATCHES_IN_FLIGHT - 1);
    CHECK(mi_pt_wait_add(h, tickets[k / B][k % B], view_b, &st));
    rays_b += st.num_basic_rays;
  }
  const double asyncb_s = now_s() - t0;

  /* ---- the batched call the standalone benchmark uses: all frames in one launch ---- */
  t0 = now_s();
  CHECK(mi_pt_render(h, 0, width, height, whole, frames, seed, 0, rgbn, &st));
  const double batch_s = now_s() - t0;

  const int same = memcmp(view_s, view_a, n4 * sizeof(double)) == 0 && memcmp(view_s, view_b, n4 * sizeof(double)) == 0 && rays_s == rays_a && rays_s == rays_b;
  printf("{\"scene\": \"%s\", \"width\": %u, \"height\": %u, \"frames\": %u, \"frames_per_launch\": %u, \"batches_in_flight\": %d, "
         "\"sync_ms_per_frame\": %.4f, \"async_ms_per_frame\": %.4f, \"async_wait_add_ms_per_frame\": %.4f, \"speedup\": %.2f, \"speedup_wait_add\": %.2f, "
         "\"sync_device_ms_per_frame\": %.4f, \"async_device_ms_per_frame\": %.4f, \"host_add_ms_per_frame\": %.4f, "
         "\"batched_call_ms_per_frame\": %.4f, \"Msamples_per_s\": {\"sync\": %.1f, \"async\": %.1f, \"async_wait_add\": %.1f, \"batched\": %.1f}, \"views_bit_identical\": %s}\n",
         argv[1], width, height, frames, batch, MI_PT_BATCHES_IN_FLIGHT, sync_s * 1e3 / frames, async_s * 1e3 / frames, asyncb_s * 1e3 / frames, sync_s / async_s, sync_s / asyncb_s,
         dev_ms_s / frames, dev_ms_a / frames, add_s * 1e3 / frames, batch_s * 1e3 / frames,
         (double)rays_s / sync_s * 1e-6, (double)rays_a / async_s * 1e-6, (double)rays_b / asyncb_s * 1e-6, (double)st.num_basic_rays / batch_s * 1e-6, same ? "true" : "false");
  free(rgbn); free(view_s); free(view_a); free(view_b); free(tickets);
  mi_pt_destroy(h);
  mi_scene_free(scene);
  return same ? 0 : 1;
}
