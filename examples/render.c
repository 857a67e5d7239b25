/* render.c — the C ABI of libmi_pt.so from plain C: what `master scene.blend --PT|--BPT --batch --num-samples=N --output=out.exr`
 * does in the reference (main.cpp -> Application -> Technique::render -> save_exr, exr.cpp:177-232), without its frame loop.
 *
 *   cc -std=c11 -I include examples/render.c -o render master_amd/libmi_pt.so -Wl,-rpath,$PWD/master_amd
 *   ./render scenes/CornellBoxDiffuse.miscene out.exr [--BPT] [--spp 64] [--size 512x512] [--max-path 8] [--beta 1] [--roulette 0.9] [--gpus N] [--lamp-scale 0.01]
 * --lamp-scale S: mi_blend_options.lamp_energy_scale of the .blend reader (1 = stock assimp units, 0.01 = the constant of unit_test.py:77-83).
 * --gpus N renders PT on N devices of this one process (mi_pt_render_multi); with fewer devices than N they are shared.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mi_pt.h"

static int ends_with(const char* s, const char* suffix) {
  size_t a = strlen(s), b = strlen(suffix);
  return a >= b && strcmp(s + a - b, suffix) == 0;
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s scene.(blend|miscene) out.exr [--BPT] [--spp N] [--size WxH] [--max-path N] [--beta B] [--roulette R] [--gpus N] [--lamp-scale S]\n", argv[0]); return 2; }
  unsigned width = 512, height = 512, spp = 16, gpus = 1; int bpt = 0;
  mi_pt_params params = {UINT64_MAX >> 1, 1.0f, 0.9f, 1.0f, 3}; /* Options.hpp:30-36 defaults: max_path unlimited, beta 1, roulette 0.9 */
  mi_blend_options blend_opts = {0.0f, 0.0f, 1.0f, 0u};         /* stock assimp units (include/mi_pt.h) */
  for (int i = 3; i < argc; ++i) {
    if (!strcmp(argv[i], "--BPT")) bpt = 1;
    else if (!strcmp(argv[i], "--PT")) bpt = 0;
    else if (!strcmp(argv[i], "--spp") && i + 1 < argc) spp = (unsigned)atoi(argv[++i]);
    else if (!strcmp(argv[i], "--size") && i + 1 < argc) { if (sscanf(argv[++i], "%ux%u", &width, &height) != 2) return 2; }
    else if (!strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = (unsigned)atoi(argv[++i]);
    else if (!strcmp(argv[i], "--max-path") && i + 1 < argc) params.max_path = (uint64_t)atoll(argv[++i]);
    else if (!strcmp(argv[i], "--beta") && i + 1 < argc) params.beta = (float)atof(argv[++i]);
    else if (!strcmp(argv[i], "--roulette") && i + 1 < argc) params.roulette = (float)atof(argv[++i]);
    else if (!strcmp(argv[i], "--lamp-scale") && i + 1 < argc) blend_opts.lamp_energy_scale = (float)atof(argv[++i]);
    else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
  }
  mi_scene* scene = NULL;
  int rc = ends_with(argv[1], ".blend") ? mi_scene_load_blend(argv[1], &blend_opts, &scene) : mi_scene_load(argv[1], &scene);
  if (rc) { fprintf(stderr, "%s\n", mi_pt_last_error()); return 1; }
  if (gpus < 1 || gpus > 64 || (bpt && gpus != 1)) { fprintf(stderr, "--gpus: 1..64, PT only\n"); return 2; }
  mi_pt_handle* hs[64] = {NULL};
  const int devices = mi_pt_device_count();
  for (unsigned k = 0; k < gpus; ++k) {
    rc = mi_pt_create(mi_scene_get_desc(scene), &params, devices > 0 ? (int)(k % (unsigned)devices) : 0, &hs[k]);
    if (rc) { fprintf(stderr, "%s\n", mi_pt_last_error()); mi_scene_free(scene); return 1; }
  }
  mi_pt_handle* h = hs[0];
  float* rgbn = (float*)malloc(sizeof(float) * 4 * (size_t)width * height);
  mi_window whole = {0, 0, 0, 0};
  mi_pt_stats st;
  rc = bpt ? mi_bpt_render(h, 0, width, height, whole, spp, 0x5EED, 0, rgbn, &st)
      : gpus > 1 ? mi_pt_render_multi(hs, gpus, 0, width, height, whole, spp, 0x5EED, 0, rgbn, &st)
                 : mi_pt_render(h, 0, width, height, whole, spp, 0x5EED, 0, rgbn, &st);
  if (rc) { fprintf(stderr, "%s\n", mi_pt_last_error()); return 1; }
  double mean = 0.0;
  for (size_t p = 0; p < (size_t)width * height; ++p)
    if (rgbn[4 * p + 3] > 0.0f) mean += (rgbn[4 * p] + rgbn[4 * p + 1] + rgbn[4 * p + 2]) / (3.0 * rgbn[4 * p + 3]);
  mean /= (double)width * height;
  char samples[32], technique[8];
  snprintf(samples, sizeof samples, "%u", spp); snprintf(technique, sizeof technique, "%s", bpt ? "BPT" : "PT");
  const char* keys[] = {"technique", "statistics.num_samples"};
  const char* values[] = {technique, samples};
  rc = mi_exr_save_rgbn(argv[2], width, height, rgbn, 2, keys, values);
  if (rc) { fprintf(stderr, "%s\n", mi_pt_last_error()); return 1; }
  printf("%s %ux%u %u spp: mean %.6f, %llu closest + %llu shadow rays, %llu numeric errors, %.2f ms on the device -> %s\n", technique, width, height, spp, mean,
         (unsigned long long)st.num_basic_rays, (unsigned long long)st.num_shadow_rays, (unsigned long long)st.numeric_errors, st.gpu_ms, argv[2]);
  free(rgbn);
  for (unsigned k = 0; k < gpus; ++k) mi_pt_destroy(hs[k]);
  mi_scene_free(scene);
  return 0;
}
