/* bvh_lab.c — tree-quality experiments on the CPU: builds the oracle's scene, optionally replaces / post-processes
 * its BVH, and counts node visits and triangle tests over real path segments (orc_trace_paths).
 * Test infrastructure only (includes the oracle source).  Build: see tests/lab/run_lab.py. */
#include <stdint.h>
static __thread uint64_t g_cnt[4]; /* nodes closest, tris closest, nodes shadow, tris shadow */
/* optional event log: 'C' / 'S' = a closest / shadow ray begins, 'n' = node visit, 'l' = leaf (triangle) test, 'P' = new path */
static unsigned char* g_log; static uint64_t g_log_n, g_log_cap;
#define LOG_EV(c) do { if (g_log && g_log_n < g_log_cap) g_log[g_log_n++] = (c); } while (0)
#define ORC_COUNT_NODE(closest) do { g_cnt[(closest) ? 0 : 2]++; LOG_EV('n'); } while (0)
#define ORC_COUNT_TRI(closest) do { g_cnt[(closest) ? 1 : 3]++; LOG_EV('l'); } while (0)
struct orc_scene; static void lab_flat_ray_fwd(const struct orc_scene* s, float ox, float oy, float oz, float dx, float dy, float dz, uint32_t ray_mask, float tmax, int closest);
#define ORC_RAY_BEGIN(closest) do { LOG_EV((closest) ? 'C' : 'S'); lab_flat_ray_fwd(s, org.x, org.y, org.z, dir.x, dir.y, dir.z, ray_mask, h->t, closest); } while (0)
#include "../../oracle/pt_oracle.c"

ORC_API void lab_log(unsigned char* buf, uint64_t cap) { g_log = buf; g_log_cap = cap; g_log_n = 0; }
ORC_API uint64_t lab_log_size(void) { return g_log_n; }
ORC_API void lab_log_mark(void) { LOG_EV('P'); }
ORC_API void lab_counters(uint64_t out[4], int reset) { for (int i = 0; i < 4; ++i) { out[i] = g_cnt[i]; if (reset) g_cnt[i] = 0; } }

/* ---- helpers over the node array ---- */
static float box_area(const float lo[3], const float hi[3]) {
  float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
  return dx * dy + dy * dz + dz * dx;
}
ORC_API double lab_sah(const orc_scene* s) {
  if (!s->n_nodes) return 0;
  float lo[3], hi[3];
  for (int a = 0; a < 3; ++a) { lo[a] = fminf(s->nodes[0].lo[0][a], s->nodes[0].lo[1][a]); hi[a] = fmaxf(s->nodes[0].hi[0][a], s->nodes[0].hi[1][a]); }
  double root = box_area(lo, hi), sum = root;
  for (uint32_t i = 0; i < s->n_nodes; ++i) for (int c = 0; c < 2; ++c) sum += box_area(s->nodes[i].lo[c], s->nodes[i].hi[c]);
  return sum / root;
}

/* ---- top-down full-sweep SAH over the Morton-sorted leaves (reference quality) ---- */
typedef struct { float lo[3], hi[3], c[3]; int32_t leaf; } prim_t;
static int g_axis;
static int cmp_prim(const void* a, const void* b) {
  float x = ((const prim_t*)a)->c[g_axis], y = ((const prim_t*)b)->c[g_axis];
  return x < y ? -1 : x > y ? 1 : 0;
}
static uint32_t g_next;
static int32_t sah_build(orc_scene* s, prim_t* p, int n, float* tmp, float out_lo[3], float out_hi[3]) {
  for (int a = 0; a < 3; ++a) { out_lo[a] = INFINITY; out_hi[a] = -INFINITY; }
  for (int i = 0; i < n; ++i) for (int a = 0; a < 3; ++a) { out_lo[a] = fminf(out_lo[a], p[i].lo[a]); out_hi[a] = fmaxf(out_hi[a], p[i].hi[a]); }
  if (n == 1) return p[0].leaf;
  int best_axis = 0, best_k = n / 2; float best = INFINITY;
  for (int ax = 0; ax < 3; ++ax) {
    g_axis = ax; qsort(p, (size_t)n, sizeof(prim_t), cmp_prim);
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = n - 1; i > 0; --i) { for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], p[i].lo[a]); hi[a] = fmaxf(hi[a], p[i].hi[a]); } tmp[i] = box_area(lo, hi) * (float)(n - i); }
    for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; }
    for (int i = 0; i < n - 1; ++i) {
      for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], p[i].lo[a]); hi[a] = fmaxf(hi[a], p[i].hi[a]); }
      float c = box_area(lo, hi) * (float)(i + 1) + tmp[i + 1];
      if (c < best) { best = c; best_axis = ax; best_k = i + 1; }
    }
  }
  g_axis = best_axis; qsort(p, (size_t)n, sizeof(prim_t), cmp_prim);
  uint32_t me = g_next++;
  float l0[3], h0[3], l1[3], h1[3];
  int32_t a = sah_build(s, p, best_k, tmp, l0, h0);
  int32_t b = sah_build(s, p + best_k, n - best_k, tmp, l1, h1);
  bnode_t* nd = &s->nodes[me];
  for (int k = 0; k < 3; ++k) { nd->lo[0][k] = l0[k]; nd->hi[0][k] = h0[k]; nd->lo[1][k] = l1[k]; nd->hi[1][k] = h1[k]; }
  nd->link[0] = a; nd->link[1] = b;
  if (a >= 0) s->nodes[a].parent = me;
  if (b >= 0) s->nodes[b].parent = me;
  return (int32_t)me;
}
/* ---- early split clipping: a triangle whose box is much larger than the triangle becomes 2..16 references with the boxes
 * of its pieces (bisect the piece's box along its longest axis, clip the polygon) ---- */
typedef struct { float v[12][3]; int n; } poly_t;
static void poly_bounds(const poly_t* q, float lo[3], float hi[3]) {
  for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; }
  for (int i = 0; i < q->n; ++i) for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], q->v[i][a]); hi[a] = fmaxf(hi[a], q->v[i][a]); }
}
static void poly_clip(const poly_t* in, int axis, float c, int keep_less, poly_t* out) {
  out->n = 0;
  for (int i = 0; i < in->n; ++i) {
    const float* a = in->v[i]; const float* b = in->v[(i + 1) % in->n];
    int ia = keep_less ? a[axis] <= c : a[axis] >= c, ib = keep_less ? b[axis] <= c : b[axis] >= c;
    if (ia) { memcpy(out->v[out->n++], a, 12); }
    if (ia != ib) {
      float t = (c - a[axis]) / (b[axis] - a[axis]);
      float* o = out->v[out->n++];
      for (int k = 0; k < 3; ++k) o[k] = a[k] + (b[k] - a[k]) * t;
      o[axis] = c;
    }
  }
}
static int split_rec(const poly_t* q, int levels, prim_t* out, int32_t leaf) {
  float lo[3], hi[3]; poly_bounds(q, lo, hi);
  if (levels == 0 || q->n < 3) {
    if (q->n < 1) return 0;
    for (int a = 0; a < 3; ++a) { out->lo[a] = lo[a]; out->hi[a] = hi[a]; }
    pad_box(out->lo, out->hi);
    for (int a = 0; a < 3; ++a) out->c[a] = 0.5f * (out->lo[a] + out->hi[a]);
    out->leaf = leaf;
    return 1;
  }
  int ax = 0; float e = hi[0] - lo[0];
  for (int a = 1; a < 3; ++a) if (hi[a] - lo[a] > e) { e = hi[a] - lo[a]; ax = a; }
  float c = 0.5f * (lo[ax] + hi[ax]);
  poly_t l, r; poly_clip(q, ax, c, 1, &l); poly_clip(q, ax, c, 0, &r);
  int n = split_rec(&l, levels - 1, out, leaf);
  return n + split_rec(&r, levels - 1, out + n, leaf);
}
static int g_split_levels_max = 0, g_split_big = 0;
ORC_API void lab_set_split(int max_levels) { g_split_levels_max = max_levels; }
ORC_API void lab_set_split_big(int factor) { g_split_big = factor; } /* also split boxes whose longest side exceeds factor x the median */
static int cmp_f(const void* a, const void* b) { float x = *(const float*)a, y = *(const float*)b; return x < y ? -1 : x > y ? 1 : 0; }

static prim_t* gen_prims(orc_scene* s, int* np_out) {
  int n = (int)s->d.n_triangles;
  prim_t* p = (prim_t*)malloc(sizeof(prim_t) * (size_t)n * 64);
  int np = 0;
  float median = 0.0f;
  if (g_split_big > 0) {
    float* ext = (float*)malloc(sizeof(float) * (size_t)n);
    for (int i = 0; i < n; ++i) { float lo[3], hi[3]; tri_bounds(s, s->sorted_tri[i], lo, hi); ext[i] = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]); }
    qsort(ext, (size_t)n, sizeof(float), cmp_f); median = ext[n / 2]; free(ext);
  }
  for (int i = 0; i < n; ++i) {
    float lo[3], hi[3]; tri_bounds(s, s->sorted_tri[i], lo, hi);
    int levels = 0;
    if (g_split_big > 0) {
      float e = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
      float ratio = e / (median * (float)g_split_big);
      while (levels < 6 && ratio > 1.0f) { ++levels; ratio *= 0.7071f; } /* each bisection of the longest axis shrinks it by ~sqrt(2) on average */
    }
    if (g_split_levels_max > 0 && levels == 0) {
      const tri_t* t = &s->tris_sorted[i];
      float at = sqrtf(vdot(t->ng, t->ng));                 /* 2 x triangle area */
      float ab = 2.0f * box_area(lo, hi);                   /* box surface area */
      float r = ab / (2.0f * at + 1e-30f);
      while (levels < g_split_levels_max && r > 4.0f) { ++levels; r *= 0.5f; }
    }
    if (levels == 0) {
      memcpy(p[np].lo, lo, 12); memcpy(p[np].hi, hi, 12); pad_box(p[np].lo, p[np].hi);
      for (int a = 0; a < 3; ++a) p[np].c[a] = 0.5f * (p[np].lo[a] + p[np].hi[a]);
      p[np].leaf = ~i; ++np;
    } else {
      poly_t q; q.n = 3;
      const uint32_t* idx = s->indices + 3 * (size_t)s->sorted_tri[i];
      for (int k = 0; k < 3; ++k) memcpy(q.v[k], s->positions + 3 * (size_t)idx[k], 12);
      np += split_rec(&q, levels, p + np, ~i);
    }
  }
  fprintf(stderr, "lab: %d triangles -> %d references\n", n, np);
  *np_out = np;
  return p;
}

ORC_API void lab_build_sah(orc_scene* s) {
  int n = (int)s->d.n_triangles;
  if (n < 2) return;
  int np = 0;
  prim_t* p = gen_prims(s, &np);
  free(s->nodes); s->n_nodes = (uint32_t)(np - 1); s->nodes = (bnode_t*)calloc(s->n_nodes, sizeof(bnode_t));
  n = np;
  float* tmp = (float*)malloc(sizeof(float) * (size_t)n);
  g_next = 0; float lo[3], hi[3];
  sah_build(s, p, n, tmp, lo, hi);
  s->nodes[0].parent = UINT32_MAX;
  s->max_depth = depth_of(s, 0);
  free(p); free(tmp);
}

/* ---- tree rotations (Kensler 2008): swap a child with a grandchild under the other child when it lowers the
 * surface area of that other child; bottom-up sweeps until no change or `passes` ---- */
static void child_box(const orc_scene* s, int32_t node, int c, float lo[3], float hi[3]) {
  for (int a = 0; a < 3; ++a) { lo[a] = s->nodes[node].lo[c][a]; hi[a] = s->nodes[node].hi[c][a]; }
}
static float union_area2(const float alo[3], const float ahi[3], const float blo[3], const float bhi[3]) {
  float lo[3], hi[3];
  for (int a = 0; a < 3; ++a) { lo[a] = fminf(alo[a], blo[a]); hi[a] = fmaxf(ahi[a], bhi[a]); }
  return box_area(lo, hi);
}
/* ---- PLOC (radius 16, union-area metric) over the same references: tree quality of the device's builder with pre-split triangles ---- */
static uint64_t lab_expand21(uint64_t v) {
  v &= 0x1FFFFF; v = (v | v << 32) & 0x1F00000000FFFFull; v = (v | v << 16) & 0x1F0000FF0000FFull;
  v = (v | v << 8) & 0x100F00F00F00F00Full; v = (v | v << 4) & 0x10C30C30C30C30C3ull; v = (v | v << 2) & 0x1249249249249249ull; return v;
}
typedef struct { uint64_t key; prim_t p; } kprim_t;
static int cmp_kprim(const void* a, const void* b) { uint64_t x = ((const kprim_t*)a)->key, y = ((const kprim_t*)b)->key; return x < y ? -1 : x > y ? 1 : 0; }
typedef struct { float lo[3], hi[3]; int32_t link; } clus_t;
static uint32_t g_renum;
static int32_t renumber(const bnode_t* src, int32_t node, bnode_t* dst) {
  if (node < 0) return node;
  uint32_t me = g_renum++;
  dst[me] = src[node];
  int32_t a = renumber(src, src[node].link[0], dst), b = renumber(src, src[node].link[1], dst);
  dst[me].link[0] = a; dst[me].link[1] = b;
  if (a >= 0) dst[a].parent = me;
  if (b >= 0) dst[b].parent = me;
  return (int32_t)me;
}
ORC_API void lab_build_ploc_refs(orc_scene* s) {
  if (s->d.n_triangles < 2) return;
  int np = 0;
  prim_t* p = gen_prims(s, &np);
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = 0; i < np; ++i) for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], p[i].c[a]); hi[a] = fmaxf(hi[a], p[i].c[a]); }
  kprim_t* kp = (kprim_t*)malloc(sizeof(kprim_t) * (size_t)np);
  for (int i = 0; i < np; ++i) {
    uint64_t q[3];
    for (int a = 0; a < 3; ++a) { float e = hi[a] - lo[a]; float t = e > 0 ? (p[i].c[a] - lo[a]) / e : 0.0f; q[a] = (uint64_t)fminf(fmaxf(t * 2097152.0f, 0.0f), 2097151.0f); }
    kp[i].key = lab_expand21(q[0]) << 2 | lab_expand21(q[1]) << 1 | lab_expand21(q[2]); kp[i].p = p[i];
  }
  qsort(kp, (size_t)np, sizeof(kprim_t), cmp_kprim);
  clus_t* c = (clus_t*)malloc(sizeof(clus_t) * (size_t)np); clus_t* c2 = (clus_t*)malloc(sizeof(clus_t) * (size_t)np);
  for (int i = 0; i < np; ++i) { memcpy(c[i].lo, kp[i].p.lo, 12); memcpy(c[i].hi, kp[i].p.hi, 12); c[i].link = kp[i].p.leaf; }
  bnode_t* tmp = (bnode_t*)calloc((size_t)np, sizeof(bnode_t)); uint32_t nn = 0;
  int* nb = (int*)malloc(sizeof(int) * (size_t)np);
  int n = np;
  while (n > 1) {
    for (int i = 0; i < n; ++i) {
      float best = INFINITY; int bj = -1;
      for (int j = (i > 16 ? i - 16 : 0); j <= i + 16 && j < n; ++j) {
        if (j == i) continue;
        float a = union_area2(c[i].lo, c[i].hi, c[j].lo, c[j].hi);
        if (a < best) { best = a; bj = j; }
      }
      nb[i] = bj;
    }
    int m = 0;
    for (int i = 0; i < n; ++i) {
      int j = nb[i];
      if (nb[j] == i) {
        if (i < j) {
          bnode_t* nd = &tmp[nn];
          memcpy(nd->lo[0], c[i].lo, 12); memcpy(nd->hi[0], c[i].hi, 12); memcpy(nd->lo[1], c[j].lo, 12); memcpy(nd->hi[1], c[j].hi, 12);
          nd->link[0] = c[i].link; nd->link[1] = c[j].link;
          for (int a = 0; a < 3; ++a) { c2[m].lo[a] = fminf(c[i].lo[a], c[j].lo[a]); c2[m].hi[a] = fmaxf(c[i].hi[a], c[j].hi[a]); }
          c2[m].link = (int32_t)nn++; ++m;
        }
      } else c2[m++] = c[i];
    }
    clus_t* t = c; c = c2; c2 = t; n = m;
  }
  free(s->nodes); s->n_nodes = nn; s->nodes = (bnode_t*)calloc(nn, sizeof(bnode_t));
  g_renum = 0; renumber(tmp, c[0].link, s->nodes);
  s->nodes[0].parent = UINT32_MAX;
  s->max_depth = depth_of(s, 0);
  free(p); free(kp); free(c); free(c2); free(tmp); free(nb);
}

static int rotate_node(orc_scene* s, int32_t x) {
  bnode_t* nd = &s->nodes[x];
  int changed = 0;
  for (int c = 0; c < 2; ++c) {          /* child c stays a child; the other child o must be internal */
    int o = 1 - c; int32_t y = nd->link[o];
    if (y < 0) continue;
    bnode_t* yn = &s->nodes[y];
    float clo[3], chi[3]; child_box(s, x, c, clo, chi);
    float cur = box_area(nd->lo[o], nd->hi[o]);
    /* option g: swap child c of x with grandchild g of y -> y becomes (c, other grandchild) */
    int bestg = -1; float best = cur;
    for (int g = 0; g < 2; ++g) {
      float a = union_area2(clo, chi, yn->lo[1 - g], yn->hi[1 - g]);
      if (a < best) { best = a; bestg = g; }
    }
    if (bestg < 0) continue;
    int g = bestg;
    int32_t cl = nd->link[c], gl = yn->link[g];
    float glo[3], ghi[3]; child_box(s, y, g, glo, ghi);
    /* x.child[c] <- grandchild g ; y.child[g] <- old child c */
    for (int a = 0; a < 3; ++a) { nd->lo[c][a] = glo[a]; nd->hi[c][a] = ghi[a]; yn->lo[g][a] = clo[a]; yn->hi[g][a] = chi[a]; }
    nd->link[c] = gl; yn->link[g] = cl;
    if (gl >= 0) s->nodes[gl].parent = (uint32_t)x;
    if (cl >= 0) s->nodes[cl].parent = (uint32_t)y;
    for (int a = 0; a < 3; ++a) { nd->lo[o][a] = fminf(yn->lo[0][a], yn->lo[1][a]); nd->hi[o][a] = fmaxf(yn->hi[0][a], yn->hi[1][a]); }
    changed = 1;
  }
  return changed;
}
static int rotate_post(orc_scene* s, int32_t x) {
  int ch = 0;
  for (int c = 0; c < 2; ++c) if (s->nodes[x].link[c] >= 0) ch += rotate_post(s, s->nodes[x].link[c]);
  /* children may have changed boxes: refresh this node's child boxes */
  for (int c = 0; c < 2; ++c) { int32_t l = s->nodes[x].link[c]; if (l >= 0) for (int a = 0; a < 3; ++a) {
    s->nodes[x].lo[c][a] = fminf(s->nodes[l].lo[0][a], s->nodes[l].lo[1][a]); s->nodes[x].hi[c][a] = fmaxf(s->nodes[l].hi[0][a], s->nodes[l].hi[1][a]); } }
  return ch + rotate_node(s, x);
}
ORC_API int lab_rotate(orc_scene* s, int passes) {
  int total = 0;
  for (int p = 0; p < passes; ++p) { int c = rotate_post(s, 0); total += c; if (!c) break; }
  s->max_depth = depth_of(s, 0);
  return total;
}

/* ---- flat leaf-list experiment (round 3): every ray is tested against the boxes of ALL leaf links (pair leaves = a node over two
 * triangles count as one), then only the leaves whose box it enters are tested.  Records, per ray, how many leaf boxes it enters
 * (closest: tmax = inf; shadow: tmax = 1) and how many of them an any-hit walk tests before the first hit. ---- */
typedef struct { float lo[3], hi[3]; int32_t a, b; } flat_leaf_t;
static flat_leaf_t g_flat[4096]; static int g_nflat; static const orc_scene* g_flat_scene;
static uint32_t* g_flat_log; static uint64_t g_flat_n, g_flat_cap;
static void flat_collect(const orc_scene* s) {
  g_nflat = 0; g_flat_scene = s;
  for (uint32_t i = 0; i < s->n_nodes; ++i) {
    const bnode_t* nd = &s->nodes[i];
    for (int c = 0; c < 2; ++c) {
      int32_t l = nd->link[c];
      flat_leaf_t f;
      if (l < 0) { memcpy(f.lo, nd->lo[c], 12); memcpy(f.hi, nd->hi[c], 12); f.a = ~l; f.b = -1; g_flat[g_nflat++] = f; }
      else if (s->nodes[l].link[0] < 0 && s->nodes[l].link[1] < 0) {
        memcpy(f.lo, nd->lo[c], 12); memcpy(f.hi, nd->hi[c], 12); f.a = ~s->nodes[l].link[0]; f.b = ~s->nodes[l].link[1]; g_flat[g_nflat++] = f; }
    }
  }
  /* singles that are children of a pair node were added twice: drop singles whose triangle is in a pair */
  int m = 0;
  for (int i = 0; i < g_nflat; ++i) {
    int dup = 0;
    if (g_flat[i].b < 0) for (int j = 0; j < g_nflat; ++j) if (g_flat[j].b >= 0 && (g_flat[j].a == g_flat[i].a || g_flat[j].b == g_flat[i].a)) dup = 1;
    if (!dup) g_flat[m++] = g_flat[i];
  }
  g_nflat = m;
}
ORC_API int lab_flat_begin(const orc_scene* s, uint32_t* log, uint64_t cap) { flat_collect(s); g_flat_log = log; g_flat_cap = cap; g_flat_n = 0; return g_nflat; }
ORC_API uint64_t lab_flat_size(void) { return g_flat_n; }
static void lab_flat_ray(const orc_scene* s, v3 org, v3 dir, uint32_t ray_mask, float tmax, int closest) {
  if (!g_flat_log || s != g_flat_scene || g_flat_n >= g_flat_cap) return;
  v3 inv = V(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
  int entered = 0, tested = 0, done = 0, entered2 = 0;
  hit_t h; h.t = tmax; h.u = h.v = 0; h.id = UINT32_MAX; h.tri = 0;
  /* pass 1: nearest box first */
  float best_tn = INFINITY; int best = -1;
  for (int i = 0; i < g_nflat; ++i) { float tn; if (box_test(g_flat[i].lo, g_flat[i].hi, org, inv, tmax, &tn)) { ++entered; if (tn < best_tn) { best_tn = tn; best = i; } } }
  if (closest && best >= 0) {
    tri_test(&s->tris_sorted[g_flat[best].a], org, dir, ray_mask, 0.0f, &h, 1);
    if (g_flat[best].b >= 0) tri_test(&s->tris_sorted[g_flat[best].b], org, dir, ray_mask, 0.0f, &h, 1);
    for (int i = 0; i < g_nflat; ++i) { float tn; if (i != best && box_test(g_flat[i].lo, g_flat[i].hi, org, inv, h.t, &tn)) ++entered2; }
  }
  if (!closest) {
    for (int i = 0; i < g_nflat && !done; ++i) { float tn; if (box_test(g_flat[i].lo, g_flat[i].hi, org, inv, tmax, &tn)) {
      ++tested;
      if (tri_test(&s->tris_sorted[g_flat[i].a], org, dir, ray_mask, 0.0f, &h, 0)) done = 1;
      else if (g_flat[i].b >= 0 && tri_test(&s->tris_sorted[g_flat[i].b], org, dir, ray_mask, 0.0f, &h, 0)) done = 1;
    } }
  }
  g_flat_log[g_flat_n++] = (uint32_t)closest | ((uint32_t)entered << 4) | ((uint32_t)tested << 12) | ((uint32_t)entered2 << 20) | ((uint32_t)done << 28);
}

static void lab_flat_ray_fwd(const struct orc_scene* s, float ox, float oy, float oz, float dx, float dy, float dz, uint32_t ray_mask, float tmax, int closest) { lab_flat_ray(s, V(ox, oy, oz), V(dx, dy, dz), ray_mask, tmax, closest); }
