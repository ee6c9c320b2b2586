// rt_device.hpp — device-side scene records and the exact (IEEE double) geometry of the hot path.
//
// Every function cites the reference lines it implements (paths under the reference checkout).
// The fp64 functions keep the reference's evaluation order — np.sum(x*y,axis=1) is
// (x0*y0 + x1*y1) + x2*y2, np.linalg.norm is sqrt((x*x+y*y)+z*z) — and this file is compiled
// with -ffp-contract=off, so their results are bit-identical to NumPy's wherever NumPy itself is
// deterministic.  Code that wants fused multiply-adds calls fma() explicitly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

constexpr double EPS = 1e-6;       // ray_tracer.py:14
constexpr double HIT_EPS = 1e-6;   // sphere.py:70-71, infinite_plane.py:42, cube.py:112-120
constexpr double PAR_EPS = 1e-10;  // infinite_plane.py:35, cube.py:80,128
constexpr int WALK_CAP = 20;       // ray_tracer.py:257
#define RT_INF __builtin_huge_val()

// One primitive, 64 bytes.  Records are sorted by type on the device (spheres, planes, boxes), each
// group in scene-file order; `orig` is the index in the caller's surface list and breaks ties the
// way the reference's list-order loop with a strict '<' does (ray_tracer.py:119-125).
struct __attribute__((aligned(16))) Prim {
  double a[6];   // sphere: cx cy cz r r*r - | plane: nx ny nz offset - - | box: min xyz, max xyz
  int32_t mat;   // 0-based material index
  int32_t orig;  // index in the original surface list
  int32_t type;
  int32_t pad;   // bit 0: a surface of this primitive coincides with another primitive's, bit 1: a box face lies in a plane (rt_api.hip, mark_coincident),
                 // bit 2: its surface meets / touches another primitive's and one of the two is not opaque (mark_near_pairs)
};
static_assert(sizeof(Prim) == 64, "Prim must be 64 bytes");

// Conservative culling hierarchy for scenes too large for the brute-force loops (host-built in
// rt_api.hip).  It never changes a result: a node / primitive is skipped only when
// its bound (a box for nodes, a sphere per record; inflated for fp32 rounding) cannot be touched by the query, and every
// surviving primitive goes through the same exact test as in the brute-force loop; ties are still
// broken by (t, orig).  Nodes are stored in depth-first order with skip links, so a lane walks the
// tree without a stack: hit -> next node, miss -> node.skip.  Leaves own <= 8 consecutive records
// of ONE type (spheres and boxes are median-split ordered inside their type range); planes are unbounded
// and stay outside the tree (always tested).
struct __attribute__((aligned(16))) BvhNode {
  float lo[3], hi[3];      // axis-aligned bounds (rounded outward on the host)
  int32_t first;           // leaf: records [first, first+count); inner node: count == 0
  uint32_t skip_count;     // low 28 bits: index of the next node when this one is missed; high 4: count
  __host__ __device__ int count() const { return (int)(skip_count >> 28); }
  __host__ __device__ int skip() const { return (int)(skip_count & 0x0FFFFFFFu); }
};
static_assert(sizeof(BvhNode) == 32, "BvhNode must be 32 bytes");
struct Accel {
  const BvhNode* nodes;    // nullptr / n_nodes == 0: no hierarchy (small scenes)
  const float4* pbound;    // [n] bounding sphere of each record (planes: unused)
  int32_t n_nodes;
  int32_t lds_nodes;       // device side: LDS byte address of a staged copy of `nodes`, -1: none (host: -1)
  const int32_t* cut;      // <= 64 nodes that cover the tree (every leaf is one of them or below one): entry
  int32_t n_cut, pad;      // points of the wave-level scan (bvh_wave_leaves)
  float lo[3], hi[3];      // union of the root boxes
  // Second hierarchy over the same records for NEAREST-hit queries: ONE complete binary tree (heap layout: children
  // of i are 2i+1, 2i+2; n_groups leaves = the last n_groups indices) over the groups of <= 8 consecutive records of
  // one type (sphere groups first, then box groups).  A lane walks it front to back without a stack (PoolRay).
  // Inner node i stores the boxes of BOTH children (one step tests the pair), interleaved for packed math:
  // hpair[3i..3i+2] = (L.lox, R.lox, L.loy, R.loy) (L.loz, R.loz, L.hix, R.hix) (L.hiy, R.hiy, L.hiz, R.hiz), rounded outward.
  const float4* hpair;     // [max(n_groups - 1, 1)][3]; nullptr: no such tree
  const uint8_t* haxis;    // [n_groups - 1] bits 0-1: axis along which the children are ordered, bit 2: left child first for d[axis] > 0
  const uint32_t* hleaf;   // [n_groups] heap leaf -> group id
  int32_t n_groups, n_sphere_groups, box_first, h_nS, h_nB;   // group g: records 8g.. (spheres) or box_first + 8(g - n_sphere_groups).. (boxes)
  int32_t lds_hpair, lds_haxis, lds_hleaf, n_lds_inner;        // device side: LDS byte addresses of staged copies (-1: none), inner nodes staged
};

#ifdef __HIPCC__
// One node.  A staged hierarchy is read with LDS instructions (ds_read_b128): through the generic
// pointer the loads would be flat ones, which queue in the vector-memory address pipeline behind the
// scattered global loads of the walks — measured as the bottleneck of the large-scene kernels.
typedef float rt_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ BvhNode load_node(const Accel& ac, int i) {
  union { rt_v4f v[2]; BvhNode n; } u;
  if (ac.lds_nodes >= 0) {
    const __attribute__((address_space(3))) rt_v4f* q =
        (const __attribute__((address_space(3))) rt_v4f*)(uintptr_t)(uint32_t)ac.lds_nodes + 2 * i;
    u.v[0] = q[0]; u.v[1] = q[1];
  } else {
    const rt_v4f* q = reinterpret_cast<const rt_v4f*>(ac.nodes) + 2 * i;
    u.v[0] = q[0]; u.v[1] = q[1];
  }
  return u.n;
}
#endif

struct DevScene {
  const Prim* prims;      // device memory, type-sorted
  const double* mats;     // [nM][12]: diffuse rgb, specular rgb, reflection rgb, shininess, transparency, pad
  const double* lights;   // [nL][12]: pos xyz, color rgb, specular, shadow, radius, pad
  int32_t nS, nP, nB, n;  // counts per type and total
  int32_t nM, nL;
  double bg[3];
  Accel ac;
  int32_t all_opaque;        // every surface material has transparency == 0
  float scene_m;             // sum over the axes of the largest |coordinate| of the culling hierarchy's bounds
};
constexpr int MAT_STRIDE = 12;
constexpr int LIGHT_STRIDE = 12;

struct DevCamera {
  double pos[3], fwd[3], right[3], up[3];
  double sdist, swidth, sheight;
};

struct d3 { double x, y, z; };
__device__ __forceinline__ d3 mk3(double x, double y, double z) { d3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ double dot3(d3 a, d3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ double norm3(d3 a) { return sqrt((a.x * a.x + a.y * a.y) + a.z * a.z); }
__device__ __forceinline__ d3 cross3(d3 a, d3 b) {  // np.cross
  return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ double comp(d3 v, int ax) { return ax == 0 ? v.x : (ax == 1 ? v.y : v.z); }

// ----------------------------------------------------------------- pixel-keyed jitter (mode B) --
// Same definition as raytracer_amd/jitter.py and oracle/rt_oracle.c.  A strong integer hash gives
// two 32-bit words per (seed, pixel, depth, light); the uniforms of cell c are the per-pixel rotated
// lattice  u_i = f32(ka + (c+1)*PHI_A) * 2^-32,  u_j = f32(kb + (c+1)*PHI_B) * 2^-32  (uint32 wrap,
// f32() = round-to-nearest conversion, so u is an fp32 value in [0,1] on every platform): two
// integer adds and two conversions per cell in the inner loop instead of a full hash.
__host__ __device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}
constexpr uint32_t JIT_PHI_A = 0x9E3779B9u, JIT_PHI_B = 0x7F4A7C15u, JIT_KB_XOR = 0x68E31DA4u;
constexpr float JIT_SCALE = 2.3283064365386963e-10f;   // 2^-32
struct JitKey { uint32_t a, b; };
__host__ __device__ __forceinline__ JitKey jit_light_key(uint32_t seed, uint32_t pixel, uint32_t depth, uint32_t light) {
  uint32_t h = lowbias32(pixel ^ seed);
  h = lowbias32(h + depth);
  h = lowbias32(h + light);
  JitKey k; k.a = h; k.b = lowbias32(h ^ JIT_KB_XOR);
  return k;
}
__device__ __forceinline__ float jit_ui(JitKey k, uint32_t cell) { return (float)(k.a + (cell + 1u) * JIT_PHI_A) * JIT_SCALE; }
__device__ __forceinline__ float jit_uj(JitKey k, uint32_t cell) { return (float)(k.b + (cell + 1u) * JIT_PHI_B) * JIT_SCALE; }

// ------------------------------------------------------------------------ primitives, t only ---
// Sphere.intersect_batch, surfaces/sphere.py:49-71.
__device__ __forceinline__ double sphere_t(const Prim& s, d3 o, d3 d) {
  d3 oc = mk3(o.x - s.a[0], o.y - s.a[1], o.z - s.a[2]);
  double a = dot3(d, d);
  double b = 2.0 * dot3(oc, d);
  double c = dot3(oc, oc) - s.a[4];            // a[4] = r*r (same double the reference computes)
  double disc = b * b - (4 * a) * c;
  if (!(disc >= 0)) return RT_INF;
  double sq = sqrt(disc);
  double t1 = (-b - sq) / (2 * a);
  double t2 = (-b + sq) / (2 * a);
  double t = (t1 > HIT_EPS) ? t1 : t2;
  return (t > HIT_EPS) ? t : RT_INF;
}
// sphere.py:80-90: outward normal at the hit, flipped to face the ray.
__device__ __forceinline__ d3 sphere_normal(const Prim& s, d3 o, d3 d, double t) {
  double r = s.a[3];
  d3 hp = mk3(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z);
  d3 n = mk3((hp.x - s.a[0]) / r, (hp.y - s.a[1]) / r, (hp.z - s.a[2]) / r);
  if (dot3(d, n) > 0) n = mk3(-n.x, -n.y, -n.z);
  return n;
}

// InfinitePlane.intersect_batch, surfaces/infinite_plane.py:30-49.
// `thr`: hit threshold.  The batch code accepts t > 1e-6 (infinite_plane.py:42); the scalar code of
// the sequential renderer rejects t < 1e-6 (infinite_plane.py:20), i.e. accepts t > pred(1e-6).
__device__ __forceinline__ double plane_t(const Prim& p, d3 o, d3 d, double thr = HIT_EPS) {
  d3 N = mk3(p.a[0], p.a[1], p.a[2]);
  double den = dot3(d, N);
  if (!(fabs(den) >= PAR_EPS)) return RT_INF;
  double t = (p.a[3] - dot3(o, N)) / den;
  return (t > thr) ? t : RT_INF;
}
__device__ __forceinline__ d3 plane_normal(const Prim& p, d3 d) {   // infinite_plane.py:51-56
  d3 N = mk3(p.a[0], p.a[1], p.a[2]);
  return dot3(d, N) > 0 ? mk3(-N.x, -N.y, -N.z) : N;
}

// Cube.intersect_batch, surfaces/cube.py:64-136.  WANT_N: also produce the face normal.
template <bool WANT_N>
__device__ __forceinline__ double box_hit(const Prim& b, d3 o, d3 d, d3* n_out) {
  double tmin = -RT_INF, tmax = RT_INF, sign = 1.0;
  int axis = 0;
  bool valid = true;
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    double dv = comp(d, ax), ov = comp(o, ax);
    bool parallel = fabs(dv) < PAR_EPS;
    bool outside = (ov < b.a[ax]) || (ov > b.a[3 + ax]);
    if (parallel && outside) valid = false;                    // cube.py:80-82
    if (!parallel && valid) {                                  // cube.py:84
      double t1 = (b.a[ax] - ov) / dv;
      double t2 = (b.a[3 + ax] - ov) / dv;
      double sg = (t1 > t2) ? 1.0 : -1.0;
      double tn = fmin(t1, t2), tf = fmax(t1, t2);
      if (tn > tmin) { tmin = tn; axis = ax; sign = sg; }      // strict: earlier axis wins ties
      tmax = fmin(tmax, tf);
      valid = valid && (tmin <= tmax);
    }
  }
  if (WANT_N) *n_out = mk3(0, 0, 0);
  if (!valid) return RT_INF;
  if (tmin >= HIT_EPS) {                                       // cube.py:113-118
    if (WANT_N) *n_out = mk3(axis == 0 ? sign : 0.0, axis == 1 ? sign : 0.0, axis == 2 ? sign : 0.0);
    return tmin;
  }
  if (tmax >= HIT_EPS) {                                       // origin inside: cube.py:120-134
    if (WANT_N) {
      bool done = false;
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        double dv = comp(d, ax), ov = comp(o, ax);
        if (done || fabs(dv) < PAR_EPS) continue;
        double t1 = (b.a[ax] - ov) / dv;
        double t2 = (b.a[3 + ax] - ov) / dv;
        double far = (t1 > t2) ? t1 : t2;
        if (fabs(far - tmax) < HIT_EPS) {
          double s = (t2 > t1) ? 1.0 : -1.0;
          *n_out = mk3(ax == 0 ? s : 0.0, ax == 1 ? s : 0.0, ax == 2 ? s : 0.0);
          done = true;
        }
      }
    }
    return tmax;
  }
  return RT_INF;
}

// ------------------------------------------------------------------------------ nearest hit ---
// find_nearest_intersection_batch, ray_tracer.py:112-127.  The reference walks the surface list in
// file order and replaces on a strict '<'; on the type-sorted records that is the lexicographic
// minimum of (t, orig).  Returns t (INF = miss) and the *sorted* record index (-1 = miss).
// Wave-uniform candidate set over the first 128 records (fan_cull); records >= 128 always count as set.
struct PrimMask {
  unsigned long long lo, hi;
  __device__ __forceinline__ bool has(int k) const { return k >= 128 || (((k < 64) ? (lo >> k) : (hi >> (k - 64))) & 1ull); }
  __device__ __forceinline__ void set(int k) { if (k < 64) lo |= 1ull << k; else if (k < 128) hi |= 1ull << (k - 64); }
  __device__ __forceinline__ static PrimMask all() { PrimMask m; m.lo = m.hi = ~0ull; return m; }
};

template <bool WAVE = false>
__device__ __forceinline__ double nearest_accel(const Prim* __restrict__ P, int nS, int nP, const Accel& ac, d3 o, d3 d,
                                                int* k_out, double plane_thr, int* wave_list = nullptr, bool try_wave = true);
__device__ __forceinline__ double nearest(const Prim* __restrict__ P, int nS, int nP, int nB, d3 o, d3 d, int* k_out,
                                          double plane_thr = HIT_EPS, const Accel* ac = nullptr,
                                          PrimMask mask = PrimMask::all()) {
  // mask: wave-uniform candidate set (fan_cull), by value so that it stays in scalar registers
  if (ac != nullptr && ac->n_nodes > 0) return nearest_accel(P, nS, nP, *ac, o, d, k_out, plane_thr);
  double best = RT_INF; int bk = -1; int borig = 0x7fffffff;
  int k = 0;
  for (int e = nS; k < e; ++k) {
    if (!mask.has(k)) continue;
    double t = sphere_t(P[k], o, d);
    int og = P[k].orig;
    if (t < best || (t == best && og < borig && bk >= 0)) { best = t; bk = k; borig = og; }
  }
  for (int e = nS + nP; k < e; ++k) {
    if (!mask.has(k)) continue;
    double t = plane_t(P[k], o, d, plane_thr);
    int og = P[k].orig;
    if (t < best || (t == best && og < borig && bk >= 0)) { best = t; bk = k; borig = og; }
  }
  for (int e = nS + nP + nB; k < e; ++k) {
    if (!mask.has(k)) continue;
    double t = box_hit<false>(P[k], o, d, nullptr);
    int og = P[k].orig;
    if (t < best || (t == best && og < borig && bk >= 0)) { best = t; bk = k; borig = og; }
  }
  *k_out = bk;
  return best;
}

// nearest() for a wave of COHERENT unit-direction rays over a small scene (the fused level 0 of k_shade_ls: an 8x8 pixel
// tile of camera rays): same result, but a sphere's exact fp64 test (~70 instruction slots with its square root and two
// divisions) runs only when some lane of the wave can hit it by a conservative fp32 test — the discriminant
// (oc.d)^2 - (|oc|^2 - r^2) with a margin of 1e-5 (|oc|^2 + r^2), two orders of magnitude above its fp32 rounding.  A
// skipped sphere has an exactly negative discriminant for every lane (t = INF), which changes neither the minimum nor
// the (t, orig) tie rule.  Must be called convergently (wave-level vote).
__device__ __forceinline__ double nearest_coherent(const Prim* __restrict__ P, int nS, int nP, int nB, d3 o, d3 d, int* k_out, double plane_thr,
                                                   PrimMask mask = PrimMask::all()) {
  double best = RT_INF; int bk = -1; int borig = 0x7fffffff;
  const float dx = (float)d.x, dy = (float)d.y, dz = (float)d.z;
  const float a32 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));                        // (directions need not be unit)
  int k = 0;
  for (int e = nS; k < e; ++k) {
    if (!mask.has(k)) continue;
    const double ox = o.x - P[k].a[0], oy = o.y - P[k].a[1], oz = o.z - P[k].a[2];
    const float fx = (float)ox, fy = (float)oy, fz = (float)oz, r2 = (float)P[k].a[4];
    const float oc2 = fmaf(fz, fz, fmaf(fy, fy, fx * fx));
    const float b = fmaf(fz, dz, fmaf(fy, dy, fx * dx));
    const bool maybe = !(fmaf(b, b, a32 * (r2 - oc2)) < -1e-5f * a32 * (oc2 + r2));      // (NaN / inf: tested exactly)
    if (!__any(maybe)) continue;
    double t = RT_INF;
    if (maybe) t = sphere_t(P[k], o, d);
    const int og = P[k].orig;
    if (t < best || (t == best && og < borig && bk >= 0)) { best = t; bk = k; borig = og; }
  }
  for (int e = nS + nP; k < e; ++k) {
    if (!mask.has(k)) continue;
    double t = plane_t(P[k], o, d, plane_thr);
    int og = P[k].orig;
    if (t < best || (t == best && og < borig && bk >= 0)) { best = t; bk = k; borig = og; }
  }
  for (int e = nS + nP + nB; k < e; ++k) {
    if (!mask.has(k)) continue;
    double t = box_hit<false>(P[k], o, d, nullptr);
    int og = P[k].orig;
    if (t < best || (t == best && og < borig && bk >= 0)) { best = t; bk = k; borig = og; }
  }
  *k_out = bk;
  return best;
}

// Stackless walk of the hierarchy by one lane.  node_hit(node) / prim_may(bound) are the
// conservative tests of the query, on_prim(k) runs the exact code for record k.  The walk is
// latency-bound (every step depends on a load), so the successor of a hit node is fetched
// speculatively while the node is being tested and the <= 8 record bounds of a leaf are fetched
// together before any of them is tested.
// One leaf: its <= 8 record bounds are fetched together, tested, and the survivors handed to on_prim.
template <int HALF = 8, class PrimMay, class OnPrim>
__device__ __forceinline__ void bvh_leaf(const Accel& ac, const BvhNode& nd, PrimMay prim_may, OnPrim on_prim,
                                         uint32_t dbg_skip = 0) {
  const int nj = (dbg_skip & 16u) ? 0 : nd.count();
  if (nj <= 0) return;
  uint32_t cand = 0;
#pragma unroll
  for (int h = 0; h < 8; h += HALF) {        // HALF = 4: two rounds of 4 bounds (16 registers less in flight)
    if (h > 0 && nj <= h) break;
    float4 b[HALF];
#pragma unroll
    for (int j = 0; j < HALF; ++j) b[j] = ac.pbound[nd.first + (h + j < nj ? h + j : 0)];
#pragma unroll
    for (int j = 0; j < HALF; ++j) cand |= (h + j < nj && prim_may(b[j])) ? (1u << (h + j)) : 0u;
  }
  if (dbg_skip & 32u) cand = 0;
  while (cand) {
    const int j = __ffs((int)cand) - 1;
    cand &= cand - 1;
    on_prim(nd.first + j);
  }
}

template <int HALF = 8, class NodeHit, class PrimMay, class OnPrim>
__device__ __forceinline__ void bvh_walk(const Accel& ac, NodeHit node_hit, PrimMay prim_may, OnPrim on_prim,
                                         uint32_t dbg_skip = 0) {
  const int n = ac.n_nodes;
  if (n <= 0) return;
  int i = 0;
  BvhNode nd = load_node(ac, 0);
  while (i < n) {
    const int nxt = i + 1;
    const BvhNode nx = load_node(ac, nxt < n ? nxt : i);
    if (node_hit(nd)) {
      bvh_leaf<HALF>(ac, nd, prim_may, on_prim, dbg_skip);
      i = nxt; nd = nx;
    } else {
      i = nd.skip();
      if (i < n) nd = load_node(ac, i);
    }
  }
}

// Wave-cooperative scan of the hierarchy against ONE wave-uniform query (wave_node_hit): the leaves that
// pass are appended to a per-wave list in LDS.  First the <= 64 nodes of the host-made cut are tested, one
// per lane; then, for every inner cut node that passed, the 64 lanes test 64 consecutive nodes of its
// subtree (depth-first order) per round; a missed inner node lets the scan jump to its skip link once the
// round is over (descendants inside the same round are tested needlessly, which is harmless).  Returns the
// number of leaves, or -1 when more than `cap` passed (the caller then walks per lane).  All 64 lanes must
// call this together.
template <class WaveNodeHit>
__device__ __forceinline__ int bvh_wave_leaves(const Accel& ac, WaveNodeHit wave_node_hit, int* __restrict__ list, int cap) {
  const int lane = (int)__lane_id();
  int n_list = 0;
  auto append = [&](bool leaf_hit, int i) __attribute__((always_inline)) {
    const unsigned long long m = __ballot(leaf_hit);
    const int cnt = __popcll(m);
    if (n_list + cnt > cap) return false;
    if (leaf_hit) list[n_list + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = i;
    n_list += cnt;
    return true;
  };
  int ci = -1, cskip = 0; bool chit = false, cleaf = false;
  if (lane < ac.n_cut) {
    ci = ac.cut[lane];
    const BvhNode nd = load_node(ac, ci);
    chit = wave_node_hit(nd);
    cleaf = nd.count() > 0;
    cskip = nd.skip();
  }
  if (!append(chit & cleaf, ci)) return -1;
  unsigned long long inner = __ballot(chit & !cleaf);
  while (inner) {
    const int b = __ffsll((long long)inner) - 1;
    inner &= inner - 1;
    const int end = __builtin_amdgcn_readlane(cskip, b);
    for (int r0 = __builtin_amdgcn_readlane(ci, b) + 1; r0 < end;) {
      const int i = r0 + lane;
      bool leaf_hit = false, far = false; int skip = 0;
      if (i < end) {
        const BvhNode nd = load_node(ac, i);
        const bool hit = wave_node_hit(nd);
        leaf_hit = hit & (nd.count() > 0);
        skip = nd.skip();
        far = !hit & (skip > r0 + 64);           // a missed subtree that reaches beyond this round
      }
      if (!append(leaf_hit, i)) return -1;
      // nested missed subtrees: the outermost one belongs to the lowest lane
      const unsigned long long fm = __ballot(far);
      r0 = fm ? __builtin_amdgcn_readlane(skip, __ffsll((long long)fm) - 1) : r0 + 64;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  return n_list;
}

// Does the ray (origin of, UNIT direction uf, fp32 copies) possibly touch the sphere (c, r) at a
// parameter <= tmax_u (unit-length parameter, already rounded up)?  Conservative: the slack covers
// the fp32 rounding of the ray, of the bound and of this arithmetic by two orders of magnitude.
__device__ __forceinline__ bool ray_may_touch(float cx, float cy, float cz, float r, float ofx, float ofy, float ofz,
                                              float ux, float uy, float uz, float tmax_u) {
  const float vx = cx - ofx, vy = cy - ofy, vz = cz - ofz;
  const float rs = r + 2e-5f * (1.0f + fabsf(vx) + fabsf(vy) + fabsf(vz));
  const float tca = fmaf(vz, uz, fmaf(vy, uy, vx * ux));
  const float wx = fmaf(-tca, ux, vx), wy = fmaf(-tca, uy, vy), wz = fmaf(-tca, uz, vz);   // v minus its part along the ray
  const float d2 = fmaf(wz, wz, fmaf(wy, wy, wx * wx));
  const float v2 = fmaf(vz, vz, fmaf(vy, vy, vx * vx));
  const float r2 = rs * rs;
  return (v2 <= r2) | ((tca >= 0.0f) & (d2 <= r2) & (tca - rs <= tmax_u));
}

typedef float rt_v2f __attribute__((ext_vector_type(2)));
typedef float rt_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rt_v2f pk2_fma(rt_v2f a, rt_v2f b, rt_v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ rt_v2f bc_lo(rt_v2f a) { return __builtin_shufflevector(a, a, 0, 0); }   // -> op_sel broadcast of v_pk_fma_f32
__device__ __forceinline__ rt_v2f bc_hi(rt_v2f a) { return __builtin_shufflevector(a, a, 1, 1); }

#ifdef RT_DEBUG_KERNEL
// audit build: k_trace walk statistics (waves, wave-level / per-lane node steps, leaves, exact tests, cycles)
__device__ unsigned long long g_walk_dbg[8];
__device__ unsigned long long g_forced_cause[8]; // ... and which special case forced it (RT_UNC_FORCED tags)
__device__ unsigned long long g_walk_audit[8];   // audit build, bit 25: [0] pairs k_shade_ls / k_shade_fast decided themselves and checked, [1] of them wrong
__device__ unsigned long long g_fix_cause[8];    // k_shade_ls: why (hit, light) pairs went to the re-test queue (RT_DEBUG_STATS bit 23)
__device__ unsigned long long g_fast_cyc[12];    // k_shade_fast wave cycles by section (RT_DEBUG_STATS bit 26), see FCK in rt_shade_fast.hpp
__device__ unsigned long long g_pool_cyc[8];     // k_trace_pool wave cycles: refill, node steps, leaf bounds, exact tests, results, all, rounds
__device__ __forceinline__ bool wave_first_active() { return (int)__lane_id() == __ffsll((long long)__ballot(1)) - 1; }
#ifdef RT_NO_WALK_COUNT     // (cycle profiles: the counting atomics would dominate)
#define RT_WALK_COUNT(i) do { } while (0)
#else
#define RT_WALK_COUNT(i) do { atomicAdd(&g_walk_dbg[(i) + 1], 1ull); if (wave_first_active()) atomicAdd(&g_walk_dbg[i], 1ull); } while (0)
#endif
#else
#define RT_WALK_COUNT(i) do { } while (0)
#endif

// One ray's nearest-hit query over the heap-layout tree (Accel::hpair) as a state that can be suspended and resumed
// (k_trace_pool: lanes whose walk has ended take the next ray while the others carry on).  begin() is nearest_accel's
// set-up (planes, walk origin, conservative fp32 tests) and exact() its on_prim: the result is the same (t, record).
//
// The walk is stackless and front to back.  `idx` is the current node, `trail` holds one bit per level of the path
// (LSB = current node), set while the node's sibling is still to be visited.  One step at an inner node tests the boxes
// of both children (they see the current best distance) and enters the nearer one that passed — by the sign of the ray
// direction along the axis the children are ordered on —, leaving the other pending if it passed too; with none, or
// at a leaf that has been processed, it climbs to the nearest ancestor with a pending sibling and goes there.  A node
// reached by climbing is not tested again: its children are (their boxes lie inside it).
struct PoolRay {
  d3 o, d;
  double best, t0;
  int bk, borig;
  float ofx, ofy, ofz, ux, uy, uz, len, tmax_u;
  rt_v2f p0, p1, p2, p3, p4;   // (ix, clx) (iy, cly) (iz, clz) (chx, chy) (chz, -): slab test t(lo) = lo*i + cl, t(hi) = hi*i + ch
  int idx;
  uint32_t trail;
  bool fresh, done;

  __device__ __forceinline__ void begin(const Prim* __restrict__ P, int nS, int nP, const Accel& ac, d3 o_, d3 d_, double plane_thr) {
    o = o_; d = d_;
    best = RT_INF; bk = -1; borig = 0x7fffffff;
    for (int k = nS, e = nS + nP; k < e; ++k) {
      const double t = plane_t(P[k], o, d, plane_thr);
      const int og = P[k].orig;
      if (t < best || (t == best && og < borig && bk >= 0)) { best = t; bk = k; borig = og; }
    }
    t0 = 0.0;
    bool skip_tree = false;
    {   // far origins: walk from where the ray enters the hierarchy's bounds (see nearest_accel)
      const double far_ = fabs(o.x) + fabs(o.y) + fabs(o.z);
      const double ext = (double)(ac.hi[0] - ac.lo[0]) + (double)(ac.hi[1] - ac.lo[1]) + (double)(ac.hi[2] - ac.lo[2]);
      const bool outside = (o.x < ac.lo[0]) | (o.x > ac.hi[0]) | (o.y < ac.lo[1]) | (o.y > ac.hi[1]) | (o.z < ac.lo[2]) | (o.z > ac.hi[2]);
      if (outside & (far_ > 4.0 * ext + 16.0)) {
        double tn = 0.0, tf = RT_INF;
        const double oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
        const double mg = 1e-3 * (ext + 1.0);
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
          if (fabs(dd[ax]) > 1e-300) {
            const double a_ = ((double)ac.lo[ax] - mg - oo[ax]) / dd[ax], b_ = ((double)ac.hi[ax] + mg - oo[ax]) / dd[ax];
            tn = fmax(tn, fmin(a_, b_)); tf = fmin(tf, fmax(a_, b_));
          } else if (oo[ax] < (double)ac.lo[ax] - mg || oo[ax] > (double)ac.hi[ax] + mg) tf = -1.0;
        }
        if (!(tn <= tf)) skip_tree = true;
        else t0 = tn;
      }
    }
    ofx = (float)(o.x + t0 * d.x); ofy = (float)(o.y + t0 * d.y); ofz = (float)(o.z + t0 * d.z);
    ux = (float)d.x; uy = (float)d.y; uz = (float)d.z;
    const float len2 = fmaf(uz, uz, fmaf(uy, uy, ux * ux));
    const float inv = 1.0f / sqrtf(len2);
    len = len2 * inv;
    ux *= inv; uy *= inv; uz *= inv;
    tmax_u = best < 1e30 ? (float)((best - t0) * len) * 1.00001f + 1e-5f : 3.0e38f;
    if (skip_tree) tmax_u = -1.0f;       // every box / record test fails
    const float ix = fabsf(ux) > 1e-20f ? 1.0f / ux : 1e30f, iy = fabsf(uy) > 1e-20f ? 1.0f / uy : 1e30f,
                iz = fabsf(uz) > 1e-20f ? 1.0f / uz : 1e30f;
    // nearest_accel's slab test, (lo - dl - of) * i, as one fma per plane: the rounding of the constants is
    // 1.2e-7 |of + dl| |i|, far inside the dl |i| the inflation gives; the fma's own rounding is relative to t and
    // covered by the 1.00001 of the comparison.  (An overflow to NaN drops that axis: v_min/v_max return the other operand.)
    const float dl = 4e-5f * (1.0f + fabsf(ofx) + fabsf(ofy) + fabsf(ofz));
    p0.x = ix; p0.y = (-ofx - dl) * ix; p3.x = (-ofx + dl) * ix;
    p1.x = iy; p1.y = (-ofy - dl) * iy; p3.y = (-ofy + dl) * iy;
    p2.x = iz; p2.y = (-ofz - dl) * iz; p4.x = (-ofz + dl) * iz; p4.y = 0.0f;
    idx = 0; trail = 0u; fresh = true; done = false;
  }
  __device__ __forceinline__ bool may(const float4& b) const { return ray_may_touch(b.x, b.y, b.z, b.w, ofx, ofy, ofz, ux, uy, uz, tmax_u); }
  __device__ __forceinline__ void exact(const Prim* __restrict__ P, int nS, int k) {
    const double t = (k < nS) ? sphere_t(P[k], o, d) : box_hit<false>(P[k], o, d, nullptr);
    const int og = P[k].orig;
    if (t < best || (t == best && og < borig && bk >= 0)) {
      best = t; bk = k; borig = og;
      tmax_u = (float)((best - t0) * len) * 1.00001f + 1e-5f;
    }
  }
  // Steps until this lane has arrived at a leaf (returns the heap index of the leaf), has finished (returns -1,
  // done), or — the walk can be resumed — fewer than `min_lanes` lanes of the wave are still stepping (returns -1).
  __device__ __forceinline__ int to_next_leaf(const Accel& ac, int min_lanes) {
    const int n_inner = ac.n_groups - 1;
    if (fresh) { fresh = false; if (n_inner == 0) return 0; }      // a single group: the root is the leaf
    int leaf = -1;
    const uint32_t sgn = (ux > 0.0f ? 1u : 0u) | (uy > 0.0f ? 2u : 0u) | (uz > 0.0f ? 4u : 0u);
    const int min_eff = min(min_lanes, __popcll(__ballot(1)) >> 2);        // (a quarter of the lanes that entered at most)
    while (!done && leaf < 0) {
      if (__popcll(__ballot(1)) < min_eff) break;
      RT_WALK_COUNT(0);
      asm("" : "+v"(p0)); asm("" : "+v"(p1)); asm("" : "+v"(p2)); asm("" : "+v"(p3)); asm("" : "+v"(p4));   // keeps the pairs pairs: op_sel broadcasts
      const bool inner = idx < n_inner;
      const int ii = inner ? idx : 0;
      rt_v4f A, B, C;
      uint32_t ax;
      if (ii < ac.n_lds_inner) {
        const __attribute__((address_space(3))) rt_v4f* q = (const __attribute__((address_space(3))) rt_v4f*)(uintptr_t)(uint32_t)ac.lds_hpair + 3 * ii;
        A = q[0]; B = q[1]; C = q[2];
        ax = ((const __attribute__((address_space(3))) uint8_t*)(uintptr_t)(uint32_t)ac.lds_haxis)[ii];
      } else {
        const rt_v4f* q = reinterpret_cast<const rt_v4f*>(ac.hpair) + 3 * (size_t)ii;
        A = q[0]; B = q[1]; C = q[2];
        ax = ac.haxis[n_inner > 0 ? ii : 0];
      }
      rt_v2f v;
      v.x = A.x; v.y = A.y; const rt_v2f tlx = pk2_fma(v, bc_lo(p0), bc_hi(p0));
      v.x = A.z; v.y = A.w; const rt_v2f tly = pk2_fma(v, bc_lo(p1), bc_hi(p1));
      v.x = B.x; v.y = B.y; const rt_v2f tlz = pk2_fma(v, bc_lo(p2), bc_hi(p2));
      v.x = B.z; v.y = B.w; const rt_v2f thx = pk2_fma(v, bc_lo(p0), bc_lo(p3));
      v.x = C.x; v.y = C.y; const rt_v2f thy = pk2_fma(v, bc_lo(p1), bc_hi(p3));
      v.x = C.z; v.y = C.w; const rt_v2f thz = pk2_fma(v, bc_lo(p2), bc_lo(p4));
      const float tnL = fmaxf(fmaxf(fminf(tlx.x, thx.x), fminf(tly.x, thy.x)), fmaxf(fminf(tlz.x, thz.x), 0.0f));
      const float tfL = fminf(fminf(fmaxf(tlx.x, thx.x), fmaxf(tly.x, thy.x)), fminf(fmaxf(tlz.x, thz.x), tmax_u));
      const float tnR = fmaxf(fmaxf(fminf(tlx.y, thx.y), fminf(tly.y, thy.y)), fmaxf(fminf(tlz.y, thz.y), 0.0f));
      const float tfR = fminf(fminf(fmaxf(tlx.y, thx.y), fmaxf(tly.y, thy.y)), fminf(fmaxf(tlz.y, thz.y), tmax_u));
      const bool hL = inner & (tnL <= fmaf(tfL, 1.00001f, 1e-5f));
      const bool hR = inner & (tnR <= fmaf(tfR, 1.00001f, 1e-5f));
      const uint32_t left_first = ((sgn >> (ax & 3u)) ^ (ax >> 2) ^ 1u) & 1u;      // bit 2 of ax: left child first for d[axis] > 0
      const bool hn = left_first ? hL : hR, hf = left_first ? hR : hL;             // nearer / farther child
      const int near = 2 * idx + 2 - (int)left_first, far = 2 * idx + 1 + (int)left_first;
      const bool any = hn | hf;
      const int idx_d = hn ? near : far;
      const uint32_t trail_d = (trail << 1) | ((hn & hf) ? 1u : 0u);
      const int k = __ffs((int)trail) - 1;                                        // levels to climb to the nearest node with a pending sibling
      int idx_p = ((idx + 1) >> (k & 31)) - 1;
      idx_p = ((idx_p - 1) ^ 1) + 1;                                              // its sibling
      const uint32_t trail_p = (trail >> (k & 31)) ^ 1u;
      done = !any & (trail == 0u);
      idx = any ? idx_d : idx_p;
      trail = any ? trail_d : trail_p;
      leaf = (!done && idx >= n_inner) ? idx : -1;
    }
    return leaf;
  }
  // Records of a heap leaf: first record, *nj = their number (<= 8).
  __device__ __forceinline__ static int leaf_records(const Accel& ac, int leaf, int* nj) {
    const int li = leaf - (ac.n_groups - 1);
    const uint32_t g = ac.lds_hleaf >= 0 ? ((const __attribute__((address_space(3))) uint32_t*)(uintptr_t)(uint32_t)ac.lds_hleaf)[li] : ac.hleaf[li];
    const bool sph = (int)g < ac.n_sphere_groups;
    const int j = sph ? (int)g : (int)g - ac.n_sphere_groups;
    *nj = min(8, (sph ? ac.h_nS : ac.h_nB) - 8 * j);
    return (sph ? 0 : ac.box_first) + 8 * j;
  }
};

// nearest() for one lane over the heap-layout tree.
__device__ __forceinline__ double nearest_heap(const Prim* __restrict__ P, int nS, int nP, const Accel& ac, d3 o, d3 d, int* k_out, double plane_thr) {
  PoolRay pr;
  pr.begin(P, nS, nP, ac, o, d, plane_thr);
  for (;;) {
    const int leaf = pr.to_next_leaf(ac, 1);
    if (leaf < 0) break;
    int nj;
    const int rec = PoolRay::leaf_records(ac, leaf, &nj);
    BvhNode nd; nd.first = rec; nd.skip_count = (uint32_t)nj << 28;
    bvh_leaf(ac, nd, [&](const float4& b) __attribute__((always_inline)) { return pr.may(b); },
             [&](int k) __attribute__((always_inline)) { pr.exact(P, nS, k); });
  }
  *k_out = pr.bk;
  return pr.best;
}

// nearest() through the hierarchy: identical result (same exact tests, same (t, orig) rule), the
// spheres/boxes that cannot be hit or cannot beat the current best are skipped.
//
// WAVE = true (k_trace; all 64 lanes must call together, wave_list = WAVE_LEAF_CAP ints of LDS per wave):
// when the wave's rays are coherent — origins in a ball (cw, rho), unit directions within a cone around
// their mean a, |u_i - a| <= kk — every point o_i + t u_i lies within rho + kk t of the axis point
// cw + t a.  The hierarchy is then scanned ONCE per wave against that hull (bvh_wave_leaves) and each lane
// tests only the listed leaves with its own ray: no dependent chain of node loads, and the leaf code runs
// for all lanes together.  Scattered waves (cos < 0.9, or more than WAVE_LEAF_CAP leaves) walk per lane.
constexpr int WAVE_LEAF_CAP = 64;     // leaves a per-wave list holds
template <bool WAVE>
__device__ __forceinline__ double nearest_accel(const Prim* __restrict__ P, int nS, int nP, const Accel& ac, d3 o, d3 d,
                                                int* k_out, double plane_thr, int* wave_list, bool try_wave) {
  if (!WAVE && ac.hpair != nullptr) return nearest_heap(P, nS, nP, ac, o, d, k_out, plane_thr);
  double best = RT_INF; int bk = -1; int borig = 0x7fffffff;
  for (int k = nS, e = nS + nP; k < e; ++k) {
    double t = plane_t(P[k], o, d, plane_thr);
    int og = P[k].orig;
    if (t < best || (t == best && og < borig && bk >= 0)) { best = t; bk = k; borig = og; }
  }
  // The conservative fp32 tests below work relative to a walk origin ow = o + t0*d.  For an origin far outside the
  // hierarchy's bounds (a bounce ray leaving a floor hit 1e5 units away) t0 moves it to where the ray reaches the
  // bounds (computed in fp64): the fp32 slack, which scales with the origin's coordinates, would otherwise swallow
  // every node, and the walk would run the exact test on every record.  Results are unchanged (the exact tests use
  // o and d themselves); a ray that misses the bounds altogether skips the hierarchy.
  double t0 = 0.0;
  bool skip_tree = false;        // (no early return: with WAVE all 64 lanes take part in the wave-level steps below)
  {
    const double far_ = fabs(o.x) + fabs(o.y) + fabs(o.z);
    const double ext = (double)(ac.hi[0] - ac.lo[0]) + (double)(ac.hi[1] - ac.lo[1]) + (double)(ac.hi[2] - ac.lo[2]);
    const bool outside = (o.x < ac.lo[0]) | (o.x > ac.hi[0]) | (o.y < ac.lo[1]) | (o.y > ac.hi[1]) | (o.z < ac.lo[2]) | (o.z > ac.hi[2]);
    if (outside & (far_ > 4.0 * ext + 16.0)) {
      double tn = 0.0, tf = RT_INF;
      const double oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
      const double mg = 1e-3 * (ext + 1.0);                   // margin around the bounds
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        if (fabs(dd[ax]) > 1e-300) {
          const double a_ = ((double)ac.lo[ax] - mg - oo[ax]) / dd[ax], b_ = ((double)ac.hi[ax] + mg - oo[ax]) / dd[ax];
          tn = fmax(tn, fmin(a_, b_)); tf = fmin(tf, fmax(a_, b_));
        } else if (oo[ax] < (double)ac.lo[ax] - mg || oo[ax] > (double)ac.hi[ax] + mg) tf = -1.0;
      }
      if (!(tn <= tf)) skip_tree = true;                     // misses the hierarchy's bounds: planes only
      else t0 = tn;
    }
  }
  const float ofx = (float)(o.x + t0 * d.x), ofy = (float)(o.y + t0 * d.y), ofz = (float)(o.z + t0 * d.z);
  float ux = (float)d.x, uy = (float)d.y, uz = (float)d.z;
  const float len2 = fmaf(uz, uz, fmaf(uy, uy, ux * ux));
  const float inv = 1.0f / sqrtf(len2), len = len2 * inv;        // directions need not be unit (rt_nearest_hit)
  ux *= inv; uy *= inv; uz *= inv;
  float tmax_u = best < 1e30 ? (float)((best - t0) * len) * 1.00001f + 1e-5f : 3.0e38f;
  if (skip_tree) tmax_u = -1.0f;       // every node / record test below fails; also when a plane is hit before the bounds (tmax_u < 0)
  // slab test of the node boxes: inflating a box by dl in space covers an error of dl*|1/u| in t
  const float ix = fabsf(ux) > 1e-20f ? 1.0f / ux : 1e30f, iy = fabsf(uy) > 1e-20f ? 1.0f / uy : 1e30f,
              iz = fabsf(uz) > 1e-20f ? 1.0f / uz : 1e30f;
  const float dl = 4e-5f * (1.0f + fabsf(ofx) + fabsf(ofy) + fabsf(ofz));
  auto node_hit = [&](const BvhNode& nd) __attribute__((always_inline)) {
    const float ax = (nd.lo[0] - dl - ofx) * ix, bx = (nd.hi[0] + dl - ofx) * ix;
    const float ay = (nd.lo[1] - dl - ofy) * iy, by = (nd.hi[1] + dl - ofy) * iy;
    const float az = (nd.lo[2] - dl - ofz) * iz, bz = (nd.hi[2] + dl - ofz) * iz;
    const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax_u));
    return tn <= tf * 1.00001f + 1e-5f;
  };
  auto prim_may = [&](const float4& b) __attribute__((always_inline)) { return ray_may_touch(b.x, b.y, b.z, b.w, ofx, ofy, ofz, ux, uy, uz, tmax_u); };
  auto on_prim = [&](int k) __attribute__((always_inline)) {
    RT_WALK_COUNT(4);
    const double t = (k < nS) ? sphere_t(P[k], o, d) : box_hit<false>(P[k], o, d, nullptr);
    const int og = P[k].orig;
    if (t < best || (t == best && og < borig && bk >= 0)) {
      best = t; bk = k; borig = og;
      tmax_u = (float)((best - t0) * len) * 1.00001f + 1e-5f;
    }
  };
  int n_wl = -1;
  if (WAVE && try_wave) {      // (try_wave is wave-uniform)
    // parameter range of this ray inside the hierarchy's bounds (empty -> it can hit nothing in the tree)
    float T_lane;
    {
      const float ax = (ac.lo[0] - dl - ofx) * ix, bx = (ac.hi[0] + dl - ofx) * ix;
      const float ay = (ac.lo[1] - dl - ofy) * iy, by = (ac.hi[1] + dl - ofy) * iy;
      const float az = (ac.lo[2] - dl - ofz) * iz, bz = (ac.hi[2] + dl - ofz) * iz;
      const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
      const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax_u));
      T_lane = (tn <= tf * 1.00001f + 1e-5f) ? tf * 1.00001f + 1e-5f : 0.0f;
    }
    float lox = ofx, loy = ofy, loz = ofz, hix = ofx, hiy = ofy, hiz = ofz, sx = ux, sy = uy, sz = uz, T = T_lane;
#pragma unroll
    for (int o_ = 32; o_ > 0; o_ >>= 1) {
      lox = fminf(lox, __shfl_xor(lox, o_)); loy = fminf(loy, __shfl_xor(loy, o_)); loz = fminf(loz, __shfl_xor(loz, o_));
      hix = fmaxf(hix, __shfl_xor(hix, o_)); hiy = fmaxf(hiy, __shfl_xor(hiy, o_)); hiz = fmaxf(hiz, __shfl_xor(hiz, o_));
      sx += __shfl_xor(sx, o_); sy += __shfl_xor(sy, o_); sz += __shfl_xor(sz, o_);
      T = fmaxf(T, __shfl_xor(T, o_));
    }
    const float cwx = 0.5f * (lox + hix), cwy = 0.5f * (loy + hiy), cwz = 0.5f * (loz + hiz);
    const float s2 = fmaf(sz, sz, fmaf(sy, sy, sx * sx));
    const float sinv = s2 > 1e-12f ? 1.0f / sqrtf(s2) : 0.0f;
    const float axx = sx * sinv, axy = sy * sinv, axz = sz * sinv;                       // mean direction
    float r2 = (ofx - cwx) * (ofx - cwx) + (ofy - cwy) * (ofy - cwy) + (ofz - cwz) * (ofz - cwz);
    float cosm = fmaf(axz, uz, fmaf(axy, uy, axx * ux));
#pragma unroll
    for (int o_ = 32; o_ > 0; o_ >>= 1) { r2 = fmaxf(r2, __shfl_xor(r2, o_)); cosm = fminf(cosm, __shfl_xor(cosm, o_)); }
    if (cosm > 0.9f && T < 1e30f) {
      const float rho = sqrtf(r2) * 1.0001f + 4e-5f * (1.0f + fabsf(cwx) + fabsf(cwy) + fabsf(cwz));
      const float kk = sqrtf(fmaxf(2.0f - 2.0f * cosm, 0.0f)) * 1.001f + 2e-5f;          // >= |u_i - a| for every lane
      const float jx = fabsf(axx) > 1e-20f ? 1.0f / axx : 1e30f, jy = fabsf(axy) > 1e-20f ? 1.0f / axy : 1e30f,
                  jz = fabsf(axz) > 1e-20f ? 1.0f / axz : 1e30f;
      const float gT = fmaf(kk, T, rho);
      n_wl = bvh_wave_leaves(ac, [&](const BvhNode& nd) __attribute__((always_inline)) {
        const float ax = (nd.lo[0] - gT - cwx) * jx, bx = (nd.hi[0] + gT - cwx) * jx;
        const float ay = (nd.lo[1] - gT - cwy) * jy, by = (nd.hi[1] + gT - cwy) * jy;
        const float az = (nd.lo[2] - gT - cwz) * jz, bz = (nd.hi[2] + gT - cwz) * jz;
        const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
        const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), T));
        if (!(tn <= tf * 1.00001f + 1e-5f)) return false;
        // the hull's radius grows with t and t <= tf inside this box: second test with the radius reached there
        const float g2 = fmaf(kk, fminf(tf * 1.00001f + 1e-5f, T), rho);
        const float ax2 = (nd.lo[0] - g2 - cwx) * jx, bx2 = (nd.hi[0] + g2 - cwx) * jx;
        const float ay2 = (nd.lo[1] - g2 - cwy) * jy, by2 = (nd.hi[1] + g2 - cwy) * jy;
        const float az2 = (nd.lo[2] - g2 - cwz) * jz, bz2 = (nd.hi[2] + g2 - cwz) * jz;
        const float tn2 = fmaxf(fmaxf(fminf(ax2, bx2), fminf(ay2, by2)), fmaxf(fminf(az2, bz2), 0.0f));
        const float tf2 = fminf(fminf(fmaxf(ax2, bx2), fmaxf(ay2, by2)), fminf(fmaxf(az2, bz2), T));
        return tn2 <= tf2 * 1.00001f + 1e-5f;
      }, wave_list, WAVE_LEAF_CAP);
    }
  }
  if (n_wl >= 0) {
    for (int e = 0; e < n_wl; ++e) {
      const BvhNode nd = load_node(ac, wave_list[e]);
      if (node_hit(nd)) bvh_leaf(ac, nd, prim_may, on_prim);
    }
  } else if (ac.hpair != nullptr && ac.lds_nodes < 0) {       // (k_trace level 0 has the skip-link nodes in LDS: its scattered waves walk those)
    return nearest_heap(P, nS, nP, ac, o, d, k_out, plane_thr);
  } else {
    bvh_walk(ac, node_hit, prim_may, on_prim);
  }
  *k_out = bk;
  return best;
}

__device__ __forceinline__ d3 prim_normal(const Prim& p, d3 o, d3 d, double t) {
  if (p.type == 0) return sphere_normal(p, o, d, t);
  if (p.type == 1) return plane_normal(p, d);
  d3 n; box_hit<true>(p, o, d, &n); return n;
}

// compute_shadow_transmission_batch, ray_tracer.py:247-293, for one ray.  `seq`: the sequential
// renderer's compute_shadow_ray_transmission instead (ray_tracer.py:296-320): walks while
// remaining > EPS without the 20-step cap (SEQ_WALK_CAP only guards the GPU against a runaway
// loop) and returns exactly 0 once the product drops below EPS.
constexpr int SEQ_WALK_CAP = 1 << 16;
// COHERENT: the wave's rays are neighbours (the items of k_shade_exact_items: samples of a few adjacent hits towards one
// light) over a small scene — the queries go through nearest_coherent (same results, exact sphere tests skipped per wave).
template <bool COHERENT = false>
__device__ __forceinline__ double shadow_walk(const Prim* __restrict__ P, int nS, int nP, int nB,
                                              const double* __restrict__ mats, d3 o, d3 d, double maxdist,
                                              int* queries, bool seq = false, double plane_thr = HIT_EPS,
                                              const Accel* ac = nullptr, PrimMask mask = PrimMask::all()) {
  double remaining = maxdist, trans = 1.0;
  const int cap = seq ? SEQ_WALK_CAP : WALK_CAP;
  for (int step = 0; step < cap; ++step) {
    if (seq && !(remaining > EPS)) break;                        // :303
    int k; double t = COHERENT ? nearest_coherent(P, nS, nP, nB, o, d, &k, plane_thr, mask) : nearest(P, nS, nP, nB, o, d, &k, plane_thr, ac, mask);
    ++*queries;
    if (!((t < remaining) && k >= 0)) break;                     // :269-272, :290-291
    trans *= mats[MAT_STRIDE * P[k].mat + 10];                   // :276-279
    if (trans < EPS) { if (seq) trans = 0.0; break; }            // :281 (value kept) / :314-315 (0.0)
    double adv = t + EPS;                                        // :287-288
    o = mk3(o.x + d.x * adv, o.y + d.y * adv, o.z + d.z * adv);
    remaining -= adv;
  }
  return trans;
}

// Conservative fp32 pre-pass of the shading code for one (hit, light) of a scene without hierarchy
// (<= 128 records, the LDS-staged scenes).  All shadow segments of the light start at p and end within RL of Lp, so they lie
// in the cone-like hull  { p + tau*tl + e : |e| <= tau*RL, 0 <= tau <= 1 }.
//   act        wave-uniform set of records that can cross the hull for SOME lane: the walks test only
//              these.  A record outside the set can only be hit beyond the light, which ends a walk
//              exactly like "no hit" does (ray_tracer.py:269-272) -> results and query counts unchanged.
//   lane_can   some record can cross this lane's hull.  If not, every walk of the lane does one query
//              that finds nothing before the light and returns exactly 1.
//   lane_full  all-opaque scenes: at the central segment's closest approach the whole hull cross-section
//              lies inside one sphere, p and the light's samples lie outside it -> every walk's first
//              query hits an opaque surface before the light and returns exactly 0.
// The fp32 slack (1e-5 relative to the magnitudes involved) exceeds the rounding of this arithmetic by
// two orders of magnitude; far hits, where |oc|^2 ~ |tl|^2 >> r^2, never get a lane_full claim.
struct FanCull { PrimMask act; bool lane_can, lane_full; };
struct FanTerms { float tx, ty, tz, T0, qa, rqa, rT0, RL; };       // per (hit, light)
__device__ __forceinline__ FanTerms fan_terms(d3 p, d3 Lp, float RLin) {
  FanTerms f;
  f.RL = RLin * 1.001f;
  f.tx = (float)(Lp.x - p.x); f.ty = (float)(Lp.y - p.y); f.tz = (float)(Lp.z - p.z);
  f.T0 = fmaf(f.tz, f.tz, fmaf(f.ty, f.ty, f.tx * f.tx));
  f.qa = f.T0 - f.RL * f.RL; f.rqa = 1.0f / f.qa; f.rT0 = 1.0f / f.T0;
  return f;
}
// One primitive against the fan of one (hit, light): can = it may cross some shadow segment, full = (spheres of all-opaque
// scenes) it swallows the whole fan.
__device__ __forceinline__ void fan_test_prim(const DevScene& sc, const Prim* __restrict__ P, d3 p, const FanTerms& f, int k, bool& can, bool& full) {
  const float tx = f.tx, ty = f.ty, tz = f.tz, T0 = f.T0, qa = f.qa, rqa = f.rqa, rT0 = f.rT0, RL = f.RL;
  full = false;
  if (k < sc.nS) {
    const Prim& s = P[k];
    const double ox = p.x - s.a[0], oy = p.y - s.a[1], oz = p.z - s.a[2];
    const float cc = (float)(((ox * ox + oy * oy) + oz * oz) - s.a[4]);    // epsilon-carrying term from fp64
    const float r = (float)s.a[3], r2 = (float)s.a[4];
    const float A = fmaf((float)oz, tz, fmaf((float)oy, ty, (float)ox * tx));                 // oc.tl
    const float qb = A - r * RL;
    const float tau = fminf(fmaxf(-qb * rqa, 0.0f), 1.0f);
    const float g = fmaf(tau, fmaf(tau, qa, 2.0f * qb), cc);               // min over tau of |oc + tau tl|^2 - (r + tau RL)^2
    // every segment direction leaves the sphere: oc.ts >= A - RL*|oc| > 0 with the origin outside
    const float ocn = sqrtf(fmaxf(cc + r2, 0.0f));
    const bool away = (cc >= 0.0f) & (A - RL * ocn > 1e-5f * (fabsf(A) + ocn * sqrtf(T0) + 1e-3f));
    can = ((qa <= 0.0f) | (g <= 1e-5f * (T0 + fabsf(cc) + r2))) & !away;
    if (sc.all_opaque) {
      const float ts = fminf(fmaxf(-A * rT0, 0.0f), 1.0f);
      const float dist2 = fmaf(ts, fmaf(ts, T0, 2.0f * A), cc + r2);       // |oc + ts*tl|^2
      const float room = r * 0.999f - ts * RL;
      const float lend2 = (cc + r2) + 2.0f * A + T0;                       // |oc + tl|^2: light centre to sphere centre
      const bool trust = (fabsf(cc) < 1000.0f * r2) & (T0 < 1000.0f * r2) & (r > 0.1f);
      full = trust & (cc > 0.0f) & (room > 0.0f) & (dist2 < room * room) & (lend2 > (r + RL) * (r + RL) * 1.01f);
    }
  } else if (k < sc.nS + sc.nP) {
    const Prim& s = P[k];
    const float num = (float)(s.a[3] - ((p.x * s.a[0] + p.y * s.a[1]) + p.z * s.a[2]));
    const float D0 = fmaf((float)s.a[2], tz, fmaf((float)s.a[1], ty, (float)s.a[0] * tx));   // N.tl, |N| = 1
    const float spread = RL * 1.001f + 1e-5f * (fabsf(D0) + sqrtf(T0));
    // tau = num / (N.ts) in (0, 1) needs num and N.ts of the same sign; N.ts is within `spread` of D0
    can = num < 0.0f ? (D0 - spread <= 0.0f) : (num > 0.0f ? (D0 + spread >= 0.0f) : true);
  } else {                                       // boxes: hull against the bounding sphere (centre, half diagonal)
    const Prim& b = P[k];
    const float cx = (float)(0.5 * (b.a[0] + b.a[3]) - p.x), cy = (float)(0.5 * (b.a[1] + b.a[4]) - p.y),
                cz = (float)(0.5 * (b.a[2] + b.a[5]) - p.z);                                  // box centre - p = -oc
    const float hx = (float)(b.a[3] - b.a[0]), hy = (float)(b.a[4] - b.a[1]), hz = (float)(b.a[5] - b.a[2]);
    const float r = 0.5f * sqrtf(fmaf(hz, hz, fmaf(hy, hy, hx * hx))) * 1.0001f + 1e-5f;
    const float oc2 = fmaf(cz, cz, fmaf(cy, cy, cx * cx));
    const float A = -fmaf(cz, tz, fmaf(cy, ty, cx * tx));                                     // oc.tl
    const float qb = A - r * RL;
    const float tau = fminf(fmaxf(-qb * rqa, 0.0f), 1.0f);
    const float g = fmaf(tau, fmaf(tau, qa, 2.0f * qb), oc2 - r * r);
    can = (qa <= 0.0f) | (g <= 1e-5f * (T0 + oc2 + r * r));
  }
}
__device__ __forceinline__ FanCull fan_cull(const DevScene& sc, const Prim* __restrict__ P, d3 p, d3 Lp, float RLin) {
  FanCull fc; fc.act.lo = fc.act.hi = 0ull; fc.lane_can = false; fc.lane_full = false;
  const FanTerms f = fan_terms(p, Lp, RLin);
  for (int k = 0; k < sc.n; ++k) {               // (k is wave-uniform: the type branches of fan_test_prim are scalar)
    bool can, full;
    fan_test_prim(sc, P, p, f, k, can, full);
    fc.lane_can |= can; fc.lane_full |= full;
    if (__any(can)) fc.act.set(k);
  }
  if (!(f.T0 > 1e-6f)) { fc.lane_can = true; fc.lane_full = false; fc.act.lo = fc.act.hi = ~0ull; }   // light at the hit point
  return fc;
}
// The same for a wave whose lanes form GROUPS of g lanes (a power of two) that hold copies of one (hit, light) each
// (k_fix_exact: one shadow sample per lane): the group's lanes share the primitives out — lane `member` tests primitives
// member, member + g, ... — instead of every lane testing all of them.  Same tests, same results: act is the union over
// the wave's records, the claims are per record.  All 64 lanes must call it.
__device__ __forceinline__ FanCull fan_cull_group(const DevScene& sc, const Prim* __restrict__ P, d3 p, d3 Lp, float RLin, uint32_t member, int g) {
  FanCull fc; fc.act.lo = fc.act.hi = 0ull; fc.lane_can = false; fc.lane_full = false;
  const FanTerms f = fan_terms(p, Lp, RLin);
  const uint32_t gbase = __lane_id() - member;                                  // first lane of this lane's group
  const unsigned long long ones = g >= 64 ? ~0ull : ((1ull << g) - 1ull);
  const unsigned long long gmask = ones << gbase;
  const unsigned long long every_g = g >= 64 ? 1ull : ~0ull / ones;             // bit 0 of every group
  for (int k0 = 0; k0 < sc.n; k0 += g) {
    const int k = k0 + (int)member;
    bool can = false, full = false;
    if (k < sc.n) fan_test_prim(sc, P, p, f, k, can, full);
    const unsigned long long mc = __ballot(can), mf = __ballot(full);
    fc.lane_can |= (mc & gmask) != 0ull; fc.lane_full |= (mf & gmask) != 0ull;
    for (int j = 0; j < g && k0 + j < sc.n; ++j)                                // primitive k0 + j: the lanes with member == j
      if (mc & (every_g << j)) fc.act.set(k0 + j);
  }
  if (!(f.T0 > 1e-6f)) { fc.lane_can = true; fc.lane_full = false; }           // light at the hit point
  if (__any(!(f.T0 > 1e-6f))) fc.act.lo = fc.act.hi = ~0ull;
  return fc;
}

// Camera.generate_all_rays / generate_rays_for_rows, camera.py:46-104: unit direction of pixel (x, y).
__device__ __forceinline__ d3 camera_dir(const DevCamera& c, int W, int H, int x, int y) {
  double px = 0.5 - ((double)x + 0.5) / W;
  double py = 0.5 - ((double)y + 0.5) / H;
  px *= c.swidth;
  py *= c.sheight;
  double v[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    double center = c.pos[k] + c.fwd[k] * c.sdist;
    double sp = (center + px * c.right[k]) + py * c.up[k];
    v[k] = sp - c.pos[k];
  }
  d3 vv = mk3(v[0], v[1], v[2]);
  double nrm = norm3(vv);
  return mk3(vv.x / nrm, vv.y / nrm, vv.z / nrm);
}

// Row sets (include/rt_hip.h, rt_frame): local row -> global row.
struct RowSet { int32_t W, H, block_rows, part, n_parts, rows; int32_t tiled, pad; };   // tiled: level-0 slots walk 8x8 pixel tiles
__host__ __device__ __forceinline__ int rowset_global_y(const RowSet& rs, int local_row) {
  if (rs.n_parts <= 1) return local_row;
  int b = local_row / rs.block_rows, r = local_row - b * rs.block_rows;
  return (b * rs.n_parts + rs.part) * rs.block_rows + r;
}

// Wave64-aggregated queue append: one atomic per wave (ballot + mbcnt prefix).  Every lane of the
// wave must call it (pred false for lanes with nothing to append).  Returns the slot for this lane.
__device__ __forceinline__ uint32_t wave_append(uint32_t* counter, bool pred) {
  unsigned long long mask = __ballot(pred);
  if (mask == 0) return 0;
  uint32_t cnt = (uint32_t)__popcll(mask);
  uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
  int leader = __ffsll((long long)mask) - 1;
  uint32_t base = 0;
  if ((int)__lane_id() == leader) base = atomicAdd(counter, cnt);
  base = __shfl(base, leader);
  return base + prefix;
}

}  // namespace rt
