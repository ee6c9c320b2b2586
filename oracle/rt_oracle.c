/*
 * rt_oracle.c — CPU restatement (IEEE float64, scalar per ray) of the reference's vectorized
 * renderer.  TEST INFRASTRUCTURE ONLY: it is the checker for the HIP path (tests/,
 * __graft_entry__.smoke(), bench.py's cpu_baseline leg).  Nothing under raytracer_amd/ may
 * import, link or call it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against vectors
 * produced by importing the reference itself (tests/gen_golden.py, fixtures in tests/golden/).
 * The reference has no tests or golden vectors of its own.
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference).
 * Operation order mirrors NumPy's evaluation order so that results are bit-identical wherever the
 * reference itself is deterministic:
 *   np.sum(x*y, axis=1) on (M,3)  ==  (x0*y0 + x1*y1) + x2*y2            (no FMA)
 *   np.linalg.norm(v, axis=1)     ==  sqrt((x*x + y*y) + z*z)
 * Known non-bit-exact spots: the plane test's np.dot goes through BLAS in the reference
 * (infinite_plane.py:30,39; last-bit differences for normals that are not axis aligned) and
 * np.power vs libm pow (<= 1-2 ulp).  Build with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define EPS 1e-6          /* ray_tracer.py:14 */
#define HIT_EPS 1e-6      /* sphere.py:70-71, infinite_plane.py:42, cube.py:112-120 */
#define PAR_EPS 1e-10     /* infinite_plane.py:35, cube.py:80,128 */
#define WALK_CAP 20       /* ray_tracer.py:257 */

enum { SURF_SPHERE = 0, SURF_PLANE = 1, SURF_BOX = 2 };
enum { JIT_PIXEL_KEYED = 0, JIT_NUMPY_STREAM = 1 };

typedef struct {
  int n_surfaces;
  const int* surf_type;       /* [n] list order == scene-file order (tie-breaking) */
  const double* surf_data;    /* [n][8] sphere: cx cy cz r | plane: nx ny nz off | box: min xyz, max xyz */
  const int* surf_material;   /* [n] 0-based */
  int n_materials;
  const double* materials;    /* [n][11] diffuse rgb, specular rgb, reflection rgb, shininess, transparency */
  int n_lights;
  const double* lights;       /* [n][9] pos xyz, color rgb, specular_intensity, shadow_intensity, radius */
  double background[3];
} orc_scene;

typedef struct {
  double pos[3], forward[3], right[3], up[3];
  double screen_distance, screen_width, screen_height;
} orc_camera;

typedef struct {
  long long rays;         /* R: sum of batch sizes (what the reference prints as "total rays") */
  long long hits;
  long long shadow_rays;  /* Sh: hits x lights x max(n,1)^2 */
  long long segments;     /* extra shadow-walk queries beyond the first one of each shadow ray */
  int batches;
} orc_stats;

static int g_threads = 1;
void orc_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int orc_get_threads(void) { return g_threads; }

/* ---------------------------------------------------------------- MT19937 (mode A jitter) --- */
/* np.random.seed(int) -> init_genrand; np.random.random -> genrand_res53 (legacy RandomState). */
typedef struct { uint32_t mt[624]; int idx; } mt_state;

static void mt_seed(mt_state* s, uint32_t seed) {
  s->mt[0] = seed;
  for (int i = 1; i < 624; ++i)
    s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
  s->idx = 624;
}
static uint32_t mt_next(mt_state* s) {
  if (s->idx >= 624) {
    uint32_t* mt = s->mt;
    for (int k = 0; k < 624; ++k) {
      uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
      mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    s->idx = 0;
  }
  uint32_t y = s->mt[s->idx++];
  y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
  return y;
}
static double mt_double(mt_state* s) {
  uint32_t a = mt_next(s) >> 5, b = mt_next(s) >> 6;
  return (a * 67108864.0 + b) / 9007199254740992.0;
}
void orc_mt_random(uint32_t seed, long long n, double* out) {
  mt_state s; mt_seed(&s, seed);
  for (long long i = 0; i < n; ++i) out[i] = mt_double(&s);
}

/* ------------------------------------------------------- pixel-keyed jitter (mode B) --------- */
/* Same integer mixer as raytracer_amd/jitter.py and csrc/rt_hip.hip. */
static inline uint32_t lowbias32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}
static inline uint32_t jit_light_key(uint32_t seed, uint32_t pixel, uint32_t depth, uint32_t light) {
  uint32_t h = lowbias32(pixel ^ seed);
  h = lowbias32(h + depth);
  return lowbias32(h + light);
}
/* per-pixel rotated lattice: u = f32(k + (cell+1)*PHI) * 2^-32 (uint32 wrap, RNE conversion) */
static inline void jit_cell(uint32_t key, uint32_t cell, double* ui, double* uj) {
  uint32_t kb = lowbias32(key ^ 0x68E31DA4u);
  uint32_t xa = key + (cell + 1u) * 0x9E3779B9u, xb = kb + (cell + 1u) * 0x7F4A7C15u;
  *ui = (double)(float)xa * (1.0 / 4294967296.0);
  *uj = (double)(float)xb * (1.0 / 4294967296.0);
}
void orc_jitter_cell(uint32_t seed, const uint32_t* pixels, long long n, int depth, int light, int cell,
                     double* ui, double* uj) {
  for (long long m = 0; m < n; ++m)
    jit_cell(jit_light_key(seed, pixels[m], (uint32_t)depth, (uint32_t)light), (uint32_t)cell, &ui[m], &uj[m]);
}

/* ---------------------------------------------------------------------- small helpers -------- */
static inline double dot3(const double* a, const double* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static inline double norm3(const double* a) { return sqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]); }
static inline void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

/* ------------------------------------------------------------------ primitives (A.3-A.5) ----- */
/* Sphere.intersect_batch, surfaces/sphere.py:45-92.  Returns t (INFINITY = miss), normal out. */
static double hit_sphere(const double* s, const double* o, const double* d, double* n) {
  double oc[3] = {o[0] - s[0], o[1] - s[1], o[2] - s[2]};
  double r = s[3];
  double a = dot3(d, d);
  double b = 2.0 * dot3(oc, d);
  double c = dot3(oc, oc) - r * r;
  double disc = b * b - 4 * a * c;
  n[0] = n[1] = n[2] = 0.0;
  if (!(disc >= 0)) return INFINITY;
  double sq = sqrt(disc);
  double t1 = (-b - sq) / (2 * a);
  double t2 = (-b + sq) / (2 * a);
  double t = (t1 > HIT_EPS) ? t1 : t2;
  if (!(t > HIT_EPS)) return INFINITY;
  double hp[3] = {o[0] + t * d[0], o[1] + t * d[1], o[2] + t * d[2]};
  n[0] = (hp[0] - s[0]) / r; n[1] = (hp[1] - s[1]) / r; n[2] = (hp[2] - s[2]) / r;
  if (dot3(d, n) > 0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
  return t;
}

/* InfinitePlane.intersect_batch, surfaces/infinite_plane.py:26-58. */
static double hit_plane(const double* p, const double* o, const double* d, double* n) {
  double den = dot3(d, p);
  n[0] = n[1] = n[2] = 0.0;
  if (!(fabs(den) >= PAR_EPS)) return INFINITY;
  double t = (p[3] - dot3(o, p)) / den;
  if (!(t > HIT_EPS)) return INFINITY;
  if (den > 0) { n[0] = -p[0]; n[1] = -p[1]; n[2] = -p[2]; }
  else { n[0] = p[0]; n[1] = p[1]; n[2] = p[2]; }
  return t;
}

/* Cube.intersect_batch, surfaces/cube.py:64-136 (slab method; exit-face normal is outward). */
static double hit_box(const double* b, const double* o, const double* d, double* n) {
  const double* bmin = b; const double* bmax = b + 3;
  double tmin = -INFINITY, tmax = INFINITY, sign = 1.0;
  int axis = 0, valid = 1;
  n[0] = n[1] = n[2] = 0.0;
  for (int ax = 0; ax < 3; ++ax) {
    int parallel = fabs(d[ax]) < PAR_EPS;
    int outside = (o[ax] < bmin[ax]) || (o[ax] > bmax[ax]);
    if (parallel && outside) valid = 0;                       /* cube.py:80-82 */
    if (parallel || !valid) continue;                         /* cube.py:84 */
    double t1 = (bmin[ax] - o[ax]) / d[ax];
    double t2 = (bmax[ax] - o[ax]) / d[ax];
    double sg = (t1 > t2) ? 1.0 : -1.0;
    double tn = fmin(t1, t2), tf = fmax(t1, t2);
    if (tn > tmin) { tmin = tn; axis = ax; sign = sg; }       /* strict: earlier axis wins ties */
    tmax = fmin(tmax, tf);
    valid = valid && (tmin <= tmax);
  }
  if (!valid) return INFINITY;
  if (tmin >= HIT_EPS) { n[axis] = sign; return tmin; }       /* cube.py:113-118 */
  if (tmax >= HIT_EPS) {                                      /* origin inside: cube.py:120-134 */
    for (int ax = 0; ax < 3; ++ax) {
      if (fabs(d[ax]) < PAR_EPS) continue;
      double t1 = (bmin[ax] - o[ax]) / d[ax];
      double t2 = (bmax[ax] - o[ax]) / d[ax];
      double far = (t1 > t2) ? t1 : t2;                       /* Python max(t1, t2) */
      if (fabs(far - tmax) < HIT_EPS) { n[ax] = (t2 > t1) ? 1.0 : -1.0; break; }
    }
    return tmax;
  }
  return INFINITY;
}

static inline double hit_surface(int type, const double* data, const double* o, const double* d, double* n) {
  switch (type) {
    case SURF_SPHERE: return hit_sphere(data, o, d, n);
    case SURF_PLANE: return hit_plane(data, o, d, n);
    default: return hit_box(data, o, d, n);
  }
}

void orc_intersect(int type, const double* data8, const double* o, const double* d, long long M,
                   double* t_out, double* n_out) {
  for (long long m = 0; m < M; ++m) t_out[m] = hit_surface(type, data8, o + 3 * m, d + 3 * m, n_out + 3 * m);
}

/* find_nearest_intersection_batch, ray_tracer.py:112-127: list order, strict '<'. */
static inline double nearest(const orc_scene* sc, const double* o, const double* d, int* idx, double* n) {
  double best = INFINITY; *idx = -1;
  if (n) n[0] = n[1] = n[2] = 0.0;
  for (int s = 0; s < sc->n_surfaces; ++s) {
    double ns[3];
    double t = hit_surface(sc->surf_type[s], sc->surf_data + 8 * s, o, d, ns);
    if (t < best) { best = t; *idx = s; if (n) { n[0] = ns[0]; n[1] = ns[1]; n[2] = ns[2]; } }
  }
  return best;
}

void orc_nearest_hit(const orc_scene* sc, const double* o, const double* d, long long M,
                     double* t_out, int* idx_out, double* n_out) {
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (long long m = 0; m < M; ++m) t_out[m] = nearest(sc, o + 3 * m, d + 3 * m, &idx_out[m], n_out + 3 * m);
}

/* compute_shadow_transmission_batch, ray_tracer.py:247-293, one ray. */
static inline double shadow_walk(const orc_scene* sc, const double* origin, const double* d, double maxdist,
                                 long long* queries) {
  double o[3] = {origin[0], origin[1], origin[2]};
  double remaining = maxdist, trans = 1.0;
  for (int step = 0; step < WALK_CAP; ++step) {
    int idx; double t = nearest(sc, o, d, &idx, 0);
    ++*queries;
    if (!((t < remaining) && idx >= 0)) break;                               /* :269-272, :290-291 */
    trans *= sc->materials[11 * sc->surf_material[idx] + 10];                /* :276-279 */
    if (trans < EPS) break;                                                  /* :281 (value kept) */
    double adv = t + EPS;                                                    /* :287-288 */
    o[0] += d[0] * adv; o[1] += d[1] * adv; o[2] += d[2] * adv;
    remaining -= adv;
  }
  return trans;
}

void orc_shadow_transmission(const orc_scene* sc, const double* o, const double* d, const double* maxdist,
                             long long M, double* out) {
  long long q = 0;
  for (long long m = 0; m < M; ++m) out[m] = shadow_walk(sc, o + 3 * m, d + 3 * m, maxdist[m], &q);
}

/* ----------------------------------------------------------------------- camera (A.2) -------- */
/* Camera.generate_all_rays / generate_rays_for_rows, camera.py:46-104 (basis from Camera.setup). */
static inline void camera_dir(const orc_camera* c, int W, int H, int x, int y, double* dir) {
  double px = 0.5 - ((double)x + 0.5) / W;
  double py = 0.5 - ((double)y + 0.5) / H;
  px *= c->screen_width;
  py *= c->screen_height;
  double v[3];
  for (int k = 0; k < 3; ++k) {
    double center = c->pos[k] + c->forward[k] * c->screen_distance;
    double sp = (center + px * c->right[k]) + py * c->up[k];
    v[k] = sp - c->pos[k];
  }
  double nrm = norm3(v);
  dir[0] = v[0] / nrm; dir[1] = v[1] / nrm; dir[2] = v[2] / nrm;
}

void orc_camera_rays(const orc_camera* c, int W, int H, int y0, int y1, double* o_out, double* d_out) {
  long long m = 0;
  for (int y = y0; y < y1; ++y)
    for (int x = 0; x < W; ++x, ++m) {
      o_out[3 * m] = c->pos[0]; o_out[3 * m + 1] = c->pos[1]; o_out[3 * m + 2] = c->pos[2];
      camera_dir(c, W, H, x, y, d_out + 3 * m);
    }
}

/* ------------------------------------------------------------------ the renderer (A.7) ------- */
typedef struct { double *o, *d, *w; uint32_t* pix; long long n; int depth; } batch_t;

static batch_t batch_alloc(long long n, int depth) {
  batch_t b; b.n = n; b.depth = depth;
  size_t m = (size_t)(n > 0 ? n : 1);
  b.o = (double*)malloc(m * 3 * sizeof(double));
  b.d = (double*)malloc(m * 3 * sizeof(double));
  b.w = (double*)malloc(m * 3 * sizeof(double));
  b.pix = (uint32_t*)malloc(m * sizeof(uint32_t));
  return b;
}
static void batch_free(batch_t* b) { free(b->o); free(b->d); free(b->w); free(b->pix); }

/* Row set of a render: rows y in [0,H) with ((y / block_rows) % n_parts) == part. */
static inline int row_in_set(int y, int block_rows, int part, int n_parts) {
  return n_parts <= 1 || ((y / block_rows) % n_parts) == part;
}

/*
 * render_vectorized, ray_tracer.py:437-604, restricted to a row set.  `out` is compact:
 * rows of the set in ascending y, each W x 3 doubles, unclipped linear radiance.
 * Returns 0, or -1 on bad arguments / allocation failure.
 */
int orc_render(const orc_scene* sc, const orc_camera* cam, int W, int H, int n_shadow, int max_depth,
               int jitter_mode, uint32_t seed, int block_rows, int part, int n_parts,
               double* out, orc_stats* stats) {
  if (W <= 0 || H <= 0 || block_rows <= 0) return -1;
  int* row_local = (int*)malloc((size_t)H * sizeof(int));
  long long nrows = 0;
  for (int y = 0; y < H; ++y) row_local[y] = row_in_set(y, block_rows, part, n_parts) ? (int)nrows++ : -1;
  long long N = nrows * W;
  memset(out, 0, (size_t)N * 3 * sizeof(double));
  orc_stats st = {0, 0, 0, 0, 0};
  mt_state mt; mt_seed(&mt, seed);
  const int L = sc->n_lights;
  const int ncell = n_shadow > 1 ? n_shadow * n_shadow : 1;

  /* LIFO stack of ray batches, ray_tracer.py:469-480 */
  int cap = 64, top = 0;
  batch_t* stack = (batch_t*)malloc(cap * sizeof(batch_t));
  {
    batch_t root = batch_alloc(N, 0);
    long long m = 0;
    for (int y = 0; y < H; ++y) {
      if (row_local[y] < 0) continue;
      for (int x = 0; x < W; ++x, ++m) {
        root.o[3 * m] = cam->pos[0]; root.o[3 * m + 1] = cam->pos[1]; root.o[3 * m + 2] = cam->pos[2];
        camera_dir(cam, W, H, x, y, root.d + 3 * m);
        root.w[3 * m] = root.w[3 * m + 1] = root.w[3 * m + 2] = 1.0;
        root.pix[m] = (uint32_t)((long long)y * W + x);
      }
    }
    stack[top++] = root;
  }

  while (top > 0) {
    batch_t b = stack[--top];
    if (b.depth > max_depth || b.n == 0) { batch_free(&b); continue; }   /* :482-483 */
    const long long M = b.n;
    st.rays += M; st.batches += 1;

    double* t = (double*)malloc((size_t)M * sizeof(double));
    int* sidx = (int*)malloc((size_t)M * sizeof(int));
    double* nrm = (double*)malloc((size_t)M * 3 * sizeof(double));
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (long long m = 0; m < M; ++m) t[m] = nearest(sc, b.o + 3 * m, b.d + 3 * m, &sidx[m], nrm + 3 * m);

    /* misses -> background (:500-502); hits compacted in order (:508-515) */
    long long Hn = 0;
    long long* hit = (long long*)malloc((size_t)M * sizeof(long long));
    for (long long m = 0; m < M; ++m) {
      if (sidx[m] >= 0) { hit[Hn++] = m; continue; }
      uint32_t p = b.pix[m];
      double* dst = out + 3 * ((long long)row_local[p / W] * W + (p % W));
      for (int k = 0; k < 3; ++k) dst[k] += b.w[3 * m + k] * sc->background[k];
    }
    if (Hn == 0) { free(t); free(sidx); free(nrm); free(hit); batch_free(&b); continue; }
    st.hits += Hn;
    st.shadow_rays += Hn * L * ncell;

    double* hp = (double*)malloc((size_t)Hn * 3 * sizeof(double));       /* hit points (:517) */
    double* view = (double*)malloc((size_t)Hn * 3 * sizeof(double));     /* view dirs (:519) */
    double* diff = (double*)calloc((size_t)Hn * 3, sizeof(double));
    double* spec = (double*)calloc((size_t)Hn * 3, sizeof(double));
    double* ttot = (double*)malloc((size_t)Hn * sizeof(double));
    double* ui = (double*)malloc((size_t)Hn * sizeof(double));
    double* uj = (double*)malloc((size_t)Hn * sizeof(double));
    uint32_t* lkey = (uint32_t*)malloc((size_t)Hn * sizeof(uint32_t));
    double* sp_p = (double*)malloc((size_t)Hn * 3 * sizeof(double));
    double* sp_r = (double*)malloc((size_t)Hn * 3 * sizeof(double));
    double* sp_u = (double*)malloc((size_t)Hn * 3 * sizeof(double));
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (long long h = 0; h < Hn; ++h) {
      long long m = hit[h];
      const double *o = b.o + 3 * m, *d = b.d + 3 * m;
      double v[3];
      for (int k = 0; k < 3; ++k) { hp[3 * h + k] = o[k] + t[m] * d[k]; v[k] = o[k] - hp[3 * h + k]; }
      double vn = norm3(v); if (vn < EPS) vn = EPS;                       /* normalize_batch :24-27 */
      for (int k = 0; k < 3; ++k) view[3 * h + k] = v[k] / vn;
    }

    for (int li = 0; li < L; ++li) {                                      /* :526 */
      const double* lg = sc->lights + 9 * li;
      const double radius = lg[8];
      memset(ttot, 0, (size_t)Hn * sizeof(double));
      if (jitter_mode == JIT_PIXEL_KEYED)
        for (long long h = 0; h < Hn; ++h) lkey[h] = jit_light_key(seed, b.pix[hit[h]], (uint32_t)b.depth, (uint32_t)li);
      long long queries = 0;
      if (n_shadow <= 1) {                                                /* :184-196 */
#pragma omp parallel for num_threads(g_threads) schedule(static) reduction(+ : queries)
        for (long long h = 0; h < Hn; ++h) {
          const double* n = nrm + 3 * hit[h];
          double p[3], tl[3], dir[3], so[3];
          for (int k = 0; k < 3; ++k) { p[k] = hp[3 * h + k] + n[k] * EPS; tl[k] = lg[k] - p[k]; }
          double dist = norm3(tl);
          for (int k = 0; k < 3; ++k) { dir[k] = tl[k] / dist; so[k] = p[k] + dir[k] * EPS; }
          ttot[h] = shadow_walk(sc, so, dir, dist - EPS, &queries);
        }
      } else {
        /* per-hit offset point and light-facing basis, once per light (:198-213) */
#pragma omp parallel for num_threads(g_threads) schedule(static)
        for (long long h = 0; h < Hn; ++h) {
          const double* n = nrm + 3 * hit[h];
          double *p = sp_p + 3 * h, *right = sp_r + 3 * h, *up = sp_u + 3 * h;
          double tl[3], ld[3], ref[3];
          for (int k = 0; k < 3; ++k) { p[k] = hp[3 * h + k] + n[k] * EPS; tl[k] = lg[k] - p[k]; }     /* :538, :198 */
          double ldist = norm3(tl);
          for (int k = 0; k < 3; ++k) ld[k] = tl[k] / ldist;                                            /* :200 */
          if (fabs(ld[0]) >= 0.9) { ref[0] = 0; ref[1] = 1; ref[2] = 0; } else { ref[0] = 1; ref[1] = 0; ref[2] = 0; }
          cross3(ld, ref, right);                                                                       /* :207-209 */
          double rn = norm3(right); if (rn < EPS) rn = EPS;
          for (int k = 0; k < 3; ++k) right[k] /= rn;
          cross3(right, ld, up);                                                                        /* :211-213 */
          double un = norm3(up); if (un < EPS) un = EPS;
          for (int k = 0; k < 3; ++k) up[k] /= un;
        }
        const double cell = radius / n_shadow;                            /* :215 */
        for (int i = 0; i < n_shadow; ++i)
          for (int j = 0; j < n_shadow; ++j) {                            /* :219-222 */
            if (jitter_mode == JIT_NUMPY_STREAM) {
              for (long long h = 0; h < Hn; ++h) ui[h] = mt_double(&mt);
              for (long long h = 0; h < Hn; ++h) uj[h] = mt_double(&mt);
            } else {
              for (long long h = 0; h < Hn; ++h) jit_cell(lkey[h], (uint32_t)(i * n_shadow + j), &ui[h], &uj[h]);
            }
#pragma omp parallel for num_threads(g_threads) schedule(static) reduction(+ : queries)
            for (long long h = 0; h < Hn; ++h) {
              const double *p = sp_p + 3 * h, *right = sp_r + 3 * h, *up = sp_u + 3 * h;
              double oi = -radius / 2 + cell * (i + ui[h]);                                             /* :224-225 */
              double oj = -radius / 2 + cell * (j + uj[h]);
              double ts[3], dir[3], so[3];
              for (int k = 0; k < 3; ++k) {
                double sp = (lg[k] + oi * right[k]) + oj * up[k];                                       /* :227-229 */
                ts[k] = sp - p[k];
              }
              double dist = norm3(ts);
              for (int k = 0; k < 3; ++k) { dir[k] = ts[k] / dist; so[k] = p[k] + dir[k] * EPS; }       /* :233-235 */
              ttot[h] += shadow_walk(sc, so, dir, dist - EPS, &queries);
            }
          }
      }
      st.segments += queries - Hn * ncell;

      /* Phong accumulate, ray_tracer.py:530-557 */
#pragma omp parallel for num_threads(g_threads) schedule(static)
      for (long long h = 0; h < Hn; ++h) {
        long long m = hit[h];
        const double* n = nrm + 3 * m;
        const double* mat = sc->materials + 11 * sc->surf_material[sidx[m]];
        double T = (n_shadow <= 1) ? ttot[h] : ttot[h] / (double)ncell;                                   /* :243 */
        double I = (1 - lg[7]) + lg[7] * T;                                                               /* :543 */
        double tl[3], l[3], ml[3];
        for (int k = 0; k < 3; ++k) tl[k] = lg[k] - hp[3 * h + k];                                        /* un-offset hp */
        double ldist = norm3(tl);
        for (int k = 0; k < 3; ++k) { l[k] = tl[k] / ldist; ml[k] = -l[k]; }
        double ndl = dot3(n, l); if (!(ndl > 0)) ndl = 0;                                                 /* np.maximum(0, .) */
        double s1 = I * ndl;
        for (int k = 0; k < 3; ++k) diff[3 * h + k] += (mat[k] * lg[3 + k]) * s1;                         /* :547 */
        double dn = dot3(ml, n);                                                                          /* reflect_batch :34-36 */
        double r[3];
        for (int k = 0; k < 3; ++k) r[k] = ml[k] - (2 * dn) * n[k];
        double rdv = dot3(r, view + 3 * h); if (!(rdv > 0)) rdv = 0;
        double s2 = I * pow(rdv, mat[9]);                                                                 /* :555-556 */
        for (int k = 0; k < 3; ++k) spec[3 * h + k] += ((mat[3 + k] * lg[3 + k]) * lg[6]) * s2;
      }
    }

    /* surface colour and children, ray_tracer.py:559-595 */
    long long nT = 0, nR = 0;
    for (long long h = 0; h < Hn; ++h) {
      long long m = hit[h];
      const double* mat = sc->materials + 11 * sc->surf_material[sidx[m]];
      uint32_t p = b.pix[m];
      double* dst = out + 3 * ((long long)row_local[p / W] * W + (p % W));
      double one_m = 1 - mat[10];
      for (int k = 0; k < 3; ++k) dst[k] += b.w[3 * m + k] * ((diff[3 * h + k] + spec[3 * h + k]) * one_m);
      if (mat[10] > EPS) ++nT;
      if (mat[6] > EPS || mat[7] > EPS || mat[8] > EPS) ++nR;
    }
    batch_t bt, br; int haveT = 0, haveR = 0;
    if (b.depth < max_depth && nT > 0) {                                  /* :565-579 */
      bt = batch_alloc(nT, b.depth + 1); haveT = 1; long long q = 0;
      for (long long h = 0; h < Hn; ++h) {
        long long m = hit[h];
        const double* mat = sc->materials + 11 * sc->surf_material[sidx[m]];
        if (!(mat[10] > EPS)) continue;
        for (int k = 0; k < 3; ++k) {
          bt.o[3 * q + k] = hp[3 * h + k] - nrm[3 * m + k] * EPS;
          bt.d[3 * q + k] = b.d[3 * m + k];
          bt.w[3 * q + k] = b.w[3 * m + k] * mat[10];
        }
        bt.pix[q++] = b.pix[m];
      }
    }
    if (b.depth < max_depth && nR > 0) {                                  /* :581-595 */
      br = batch_alloc(nR, b.depth + 1); haveR = 1; long long q = 0;
      for (long long h = 0; h < Hn; ++h) {
        long long m = hit[h];
        const double* mat = sc->materials + 11 * sc->surf_material[sidx[m]];
        if (!(mat[6] > EPS || mat[7] > EPS || mat[8] > EPS)) continue;
        const double *d = b.d + 3 * m, *n = nrm + 3 * m;
        double dn = dot3(d, n);
        for (int k = 0; k < 3; ++k) {
          br.d[3 * q + k] = d[k] - (2 * dn) * n[k];
          br.o[3 * q + k] = hp[3 * h + k] + n[k] * EPS;
          br.w[3 * q + k] = b.w[3 * m + k] * mat[6 + k];
        }
        br.pix[q++] = b.pix[m];
      }
    }
    if (top + 2 > cap) { cap *= 2; stack = (batch_t*)realloc(stack, cap * sizeof(batch_t)); }
    if (haveT) stack[top++] = bt;
    if (haveR) stack[top++] = br;      /* pushed last => processed first */

    free(t); free(sidx); free(nrm); free(hit); free(hp); free(view); free(diff); free(spec);
    free(ttot); free(ui); free(uj); free(lkey); free(sp_p); free(sp_r); free(sp_u);
    batch_free(&b);
  }
  free(stack); free(row_local);
  if (stats) *stats = st;
  return 0;
}

/* ------------------------------------------------- the sequential renderer (App. A.9) -------- */
/*
 * render(), ray_tracer.py:398-434, with trace_ray (:383-395), compute_color (:323-380),
 * compute_soft_shadow (:130-168), compute_shadow_ray_transmission (:296-320) and the scalar
 * intersect() methods (sphere.py:11-43, infinite_plane.py:11-24, cube.py:14-62) — the reference's
 * per-pixel recursive renderer.  It differs from render_vectorized in (A.9): the plane accepts
 * t == 1e-6; the shadow walk is unbounded and returns exactly 0.0 below EPS; a light whose
 * intensity factor is < EPS is skipped; specular needs r.v > 0; `normalize` leaves vectors shorter
 * than EPS unchanged; a transparent surface whose transmitted ray is not traced (depth cap, or
 * transparency <= EPS) adds background*transparency; products are grouped as Python evaluates them.
 * The scalar sphere and cube tests equal the batch ones (the cube's exit-face scan uses `> 1e-10`
 * where the batch code uses `>=`; a measure-zero difference that is not mirrored).  The reference's
 * scalar np.dot goes through BLAS, so results are pinned to <= 1e-9, not bit-exact.
 * Jitter: mode A = scalar draws from the MT19937 stream in the reference's per-pixel depth-first
 * order (u_i then u_j per cell; reflection subtree before transmission subtree; single-threaded);
 * mode B = the pixel-keyed lattice with level = max_depth - depth.
 */
typedef struct {
  const orc_scene* sc; int n_shadow, max_depth, jitter_mode; uint32_t seed; mt_state* mt; orc_stats st;
} seq_ctx;

static double seq_hit_surface(int type, const double* data, const double* o, const double* d, double* n) {
  if (type != SURF_PLANE) return hit_surface(type, data, o, d, n);
  double den = dot3(d, data);                                   /* infinite_plane.py:13-24 */
  n[0] = n[1] = n[2] = 0.0;
  if (fabs(den) < PAR_EPS) return INFINITY;
  double t = (data[3] - dot3(o, data)) / den;
  if (t < HIT_EPS || !(t == t)) return INFINITY;                /* `t < 1e-6` rejects: t == 1e-6 is a hit */
  for (int k = 0; k < 3; ++k) n[k] = den > 0 ? -data[k] : data[k];
  return t;
}

/* find_nearest_intersection, ray_tracer.py:94-109 */
static double seq_nearest(const orc_scene* sc, const double* o, const double* d, double max_t, int* idx, double* n) {
  double best = max_t; *idx = -1;
  for (int s = 0; s < sc->n_surfaces; ++s) {
    double ns[3];
    double t = seq_hit_surface(sc->surf_type[s], sc->surf_data + 8 * s, o, d, ns);
    if (t < best) { best = t; *idx = s; if (n) { n[0] = ns[0]; n[1] = ns[1]; n[2] = ns[2]; } }
  }
  return best;
}

/* compute_shadow_ray_transmission, ray_tracer.py:296-320 */
static double seq_shadow(seq_ctx* c, const double* origin, const double* target, double max_distance) {
  double dir[3], cur[3];
  for (int k = 0; k < 3; ++k) { dir[k] = (target[k] - origin[k]) / max_distance; cur[k] = origin[k] + dir[k] * EPS; }
  double remaining = max_distance - EPS, trans = 1.0;
  int first = 1;
  while (remaining > EPS) {
    int idx; double t = seq_nearest(c->sc, cur, dir, remaining, &idx, 0);
    if (!first) c->st.segments += 1;
    first = 0;
    if (idx < 0) break;
    trans *= c->sc->materials[11 * c->sc->surf_material[idx] + 10];
    if (trans < EPS) return 0.0;
    for (int k = 0; k < 3; ++k) cur[k] = cur[k] + dir[k] * (t + EPS);
    remaining -= (t + EPS);
  }
  return trans;
}

static void seq_normalize(double* v) {                          /* normalize, ray_tracer.py:17-21 */
  double n = norm3(v);
  if (n < EPS) return;
  v[0] /= n; v[1] /= n; v[2] /= n;
}

/* compute_soft_shadow, ray_tracer.py:130-168 */
static double seq_soft_shadow(seq_ctx* c, const double* p, const double* lg, uint32_t key) {
  double tl[3] = {lg[0] - p[0], lg[1] - p[1], lg[2] - p[2]};
  double dist = norm3(tl);
  const int n = c->n_shadow;
  c->st.shadow_rays += n <= 1 ? 1 : n * n;
  if (n <= 1) return seq_shadow(c, p, lg, dist);
  double ld[3] = {tl[0] / dist, tl[1] / dist, tl[2] / dist};
  double ref[3] = {0, 0, 0}; if (fabs(ld[0]) < 0.9) ref[0] = 1.0; else ref[1] = 1.0;
  double right[3], up[3];
  cross3(ld, ref, right); seq_normalize(right);
  cross3(right, ld, up); seq_normalize(up);
  const double radius = lg[8], cell = radius / n;
  double total = 0.0;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double ui, uj;
      if (c->jitter_mode == JIT_NUMPY_STREAM) { ui = mt_double(c->mt); uj = mt_double(c->mt); }
      else jit_cell(key, (uint32_t)(i * n + j), &ui, &uj);
      double oi = -radius / 2 + cell * (i + ui), oj = -radius / 2 + cell * (j + uj);
      double sp[3], ts[3];
      for (int k = 0; k < 3; ++k) { sp[k] = (lg[k] + right[k] * oi) + up[k] * oj; ts[k] = sp[k] - p[k]; }
      total += seq_shadow(c, p, sp, norm3(ts));
    }
  return total / (double)(n * n);
}

/* trace_ray + compute_color, ray_tracer.py:323-395; depth counts down from max_depth */
static void seq_trace(seq_ctx* c, const double* o, const double* d, int depth, uint32_t pixel, double* out) {
  const orc_scene* sc = c->sc;
  if (depth < 0) { for (int k = 0; k < 3; ++k) out[k] = sc->background[k]; return; }
  int idx; double n[3];
  c->st.rays += 1;
  double t = seq_nearest(sc, o, d, INFINITY, &idx, n);
  if (idx < 0) { for (int k = 0; k < 3; ++k) out[k] = sc->background[k]; return; }
  c->st.hits += 1;
  const double* mat = sc->materials + 11 * sc->surf_material[idx];
  double hp[3], v[3], p[3];
  for (int k = 0; k < 3; ++k) { hp[k] = o[k] + t * d[k]; v[k] = o[k] - hp[k]; p[k] = hp[k] + n[k] * EPS; }
  seq_normalize(v);
  double dif[3] = {0, 0, 0}, spc[3] = {0, 0, 0};
  for (int li = 0; li < sc->n_lights; ++li) {
    const double* lg = sc->lights + 9 * li;
    double l[3] = {lg[0] - hp[0], lg[1] - hp[1], lg[2] - hp[2]};
    seq_normalize(l);
    uint32_t key = jit_light_key(c->seed, pixel, (uint32_t)(c->max_depth - depth), (uint32_t)li);
    double T = seq_soft_shadow(c, p, lg, key);
    double I = (1 - lg[7]) + lg[7] * T;
    if (I < EPS) continue;                                                        /* :347-348 */
    double ndl = dot3(n, l); if (!(ndl > 0)) ndl = 0;
    for (int k = 0; k < 3; ++k) dif[k] += ((mat[k] * lg[3 + k]) * I) * ndl;       /* :351 */
    double ml[3] = {-l[0], -l[1], -l[2]};
    double dn = dot3(ml, n), r[3];
    for (int k = 0; k < 3; ++k) r[k] = ml[k] - (2 * dn) * n[k];                   /* reflect :30-31 */
    double rdv = dot3(r, v);
    if (rdv > 0) {                                                                /* :356-359 */
      double pw = pow(rdv, mat[9]);
      for (int k = 0; k < 3; ++k) spc[k] += (((mat[3 + k] * lg[3 + k]) * lg[6]) * I) * pw;
    }
  }
  double refl[3] = {0, 0, 0};
  if (depth > 0 && (mat[6] > EPS || mat[7] > EPS || mat[8] > EPS)) {              /* :362-368 */
    double dn = dot3(d, n), rd[3], ro[3], rc[3];
    for (int k = 0; k < 3; ++k) { rd[k] = d[k] - (2 * dn) * n[k]; ro[k] = hp[k] + n[k] * EPS; }
    seq_trace(c, ro, rd, depth - 1, pixel, rc);
    for (int k = 0; k < 3; ++k) refl[k] = mat[6 + k] * rc[k];
  }
  double bg[3] = {sc->background[0], sc->background[1], sc->background[2]};
  if (mat[10] > EPS && depth > 0) {                                               /* :371-374 */
    double co[3];
    for (int k = 0; k < 3; ++k) co[k] = hp[k] - n[k] * EPS;
    seq_trace(c, co, d, depth - 1, pixel, bg);
  }
  for (int k = 0; k < 3; ++k) out[k] = (bg[k] * mat[10] + (dif[k] + spc[k]) * (1 - mat[10])) + refl[k];   /* :376-378 */
}

int orc_render_sequential(const orc_scene* sc, const orc_camera* cam, int W, int H, int n_shadow, int max_depth,
                          int jitter_mode, uint32_t seed, int block_rows, int part, int n_parts,
                          double* out, orc_stats* stats) {
  if (W <= 0 || H <= 0 || block_rows <= 0) return -1;
  int* rows = (int*)malloc((size_t)H * sizeof(int));
  int nrows = 0;
  for (int y = 0; y < H; ++y) if (row_in_set(y, block_rows, part, n_parts)) rows[nrows++] = y;
  mt_state mt; mt_seed(&mt, seed);
  orc_stats total = {0, 0, 0, 0, 0};
  const int threads = jitter_mode == JIT_NUMPY_STREAM ? 1 : g_threads;           /* the stream is serial */
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
  for (int lr = 0; lr < nrows; ++lr) {
    seq_ctx c; c.sc = sc; c.n_shadow = n_shadow; c.max_depth = max_depth; c.jitter_mode = jitter_mode;
    c.seed = seed; c.mt = &mt; memset(&c.st, 0, sizeof c.st);
    const int y = rows[lr];
    for (int x = 0; x < W; ++x) {
      double d[3];
      camera_dir(cam, W, H, x, y, d);                                             /* camera.py:30-44 */
      seq_trace(&c, cam->pos, d, max_depth, (uint32_t)((long long)y * W + x), out + 3 * ((size_t)lr * W + x));
    }
#pragma omp critical
    { total.rays += c.st.rays; total.hits += c.st.hits; total.shadow_rays += c.st.shadow_rays; total.segments += c.st.segments; }
  }
  total.batches = 0;
  free(rows);
  if (stats) *stats = total;
  return 0;
}
