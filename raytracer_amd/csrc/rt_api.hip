// rt_api.hip — host side of librt_hip.so: the C-ABI of include/rt_hip.h over the kernels in
// rt_kernels.hpp.  One rt_scene owns the device copy of a scene, a HIP stream and a reusable
// workspace (framebuffer, hit queue, two ping-pong ray queues, counters), so a render call does
// no allocation once the workspace has grown to the frame size.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_kernels.hpp"
#include "rt_shade_fast.hpp"
#include "rt_shade_ls.hpp"
#include "rt_stream.hpp"
#include "rt_sort.hpp"
#include <cmath>
#include <hipcub/hipcub.hpp>

using namespace rt;

static_assert(RT_MAX_LEVELS == MAX_LEVELS, "level caps differ");

// ------------------------------------------------------------------------------- errors --------
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  g_err = buf;
  return code;
}
#define HIPCHK(call)                                                                           \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? RT_ERR_NOMEM : RT_ERR_HIP,   \
                                      "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// The ABI never throws: every extern "C" entry point is a function-try-block (the bodies use std::vector / deque /
// string; a std::bad_alloc crossing into ctypes would terminate the caller's process).
#define RT_CATCH_ALL                                                                               \
  catch (const std::bad_alloc&) { return fail(RT_ERR_NOMEM, "out of host memory"); }               \
  catch (const std::exception& e_) { return fail(RT_ERR_HIP, "internal error: %s", e_.what()); }   \
  catch (...) { return fail(RT_ERR_HIP, "internal error"); }

extern "C" const char* rt_last_error(void) { return g_err.c_str(); }

extern "C" int rt_device_count(int* count) try {
  if (!count) return fail(RT_ERR_INVALID, "rt_device_count: null pointer");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *count = 0; return fail(RT_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
  *count = n;
  return RT_OK;
} RT_CATCH_ALL

static int static_scratch_for(int device, unsigned char** out);   // rt_render_static, below

// --------------------------------------------------------------- culling hierarchy (host) ------
// Built once per scene for scenes that do not fit the LDS-staged brute-force kernels (n > LDS_PRIMS):
// spheres and boxes are ordered by recursive median splits of their centres inside their type range
// (median_order; tie-breaking uses Prim::orig, so the record order inside a type is free), then a 4-ary
// tree over that order (leaves of 8 records) is emitted in depth-first order with skip links
// (rt_device.hpp, BvhNode).  Bounds are spheres, inflated so that
// rounding the double-precision geometry to fp32 can only make them larger.
namespace {
struct BoundD { double c[3], r; };

BoundD prim_bound(const Prim& p) {
  BoundD b;
  if (p.type == RT_SPHERE) { for (int k = 0; k < 3; ++k) b.c[k] = p.a[k]; b.r = std::fabs(p.a[3]); }
  else {                      // box: centre + half diagonal
    double h2 = 0;
    for (int k = 0; k < 3; ++k) { b.c[k] = 0.5 * (p.a[k] + p.a[3 + k]); double h = 0.5 * std::fabs(p.a[3 + k] - p.a[k]); h2 += h * h; }
    b.r = std::sqrt(h2);
  }
  return b;
}

float4 to_float_bound(const BoundD& b) {
  const double cmax = std::max(std::fabs(b.c[0]), std::max(std::fabs(b.c[1]), std::fabs(b.c[2])));
  const double r = (b.r + 4.0 * 1.2e-7 * cmax) * (1.0 + 1e-6) + 1e-30;     // covers the centre's fp32 rounding
  float rf = (float)r;
  if ((double)rf < r) rf = std::nextafter(rf, INFINITY);
  if (!(rf == rf) || std::isinf(rf)) rf = 3.0e38f;
  return make_float4((float)b.c[0], (float)b.c[1], (float)b.c[2], rf);
}

uint32_t spread10(uint32_t v) {   // 10 bits -> every third bit
  v &= 1023u; v = (v | (v << 16)) & 0x030000FFu; v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u; v = (v | (v << 2)) & 0x09249249u; return v;
}

struct BoxD { double lo[3], hi[3]; };

BoxD prim_box(const Prim& p) {
  BoxD b;
  if (p.type == RT_SPHERE) { const double r = std::fabs(p.a[3]); for (int k = 0; k < 3; ++k) { b.lo[k] = p.a[k] - r; b.hi[k] = p.a[k] + r; } }
  else for (int k = 0; k < 3; ++k) { b.lo[k] = std::min(p.a[k], p.a[3 + k]); b.hi[k] = std::max(p.a[k], p.a[3 + k]); }
  return b;
}

struct BvhBuilder {
  const std::vector<Prim>& prims;
  std::vector<BvhNode> nodes;
  int leaf = 8, fanout = 4;      // leaf <= 8 records (BvhNode::count has 4 bits); fanout 4 measured best (8: +5 %, 2: +16 %)
  explicit BvhBuilder(const std::vector<Prim>& p) : prims(p) {
    if (const char* e = getenv("RT_BVH_LEAF")) leaf = std::min(8, std::max(1, atoi(e)));
    if (const char* e = getenv("RT_BVH_FANOUT")) fanout = std::min(64, std::max(2, atoi(e)));
  }
  static void grow(BoxD* a, const BoxD& b) { for (int k = 0; k < 3; ++k) { a->lo[k] = std::min(a->lo[k], b.lo[k]); a->hi[k] = std::max(a->hi[k], b.hi[k]); } }
  BoxD build(int first, int count) {
    const int idx = (int)nodes.size();
    nodes.push_back(BvhNode{});
    BoxD box; for (int k = 0; k < 3; ++k) { box.lo[k] = 1e300; box.hi[k] = -1e300; }
    int leaf_count = 0;
    if (count <= leaf) {
      for (int i = 0; i < count; ++i) grow(&box, prim_box(prims[first + i]));
      leaf_count = count;
    } else {
      long long child = leaf; while (child * fanout < count) child *= fanout;   // leaf * fanout^k, the largest below count
      for (long long off = 0; off < count; off += child) grow(&box, build(first + (int)off, (int)std::min<long long>(child, count - off)));
    }
    for (int k = 0; k < 3; ++k) {                                      // round outward
      float lo = (float)box.lo[k], hi = (float)box.hi[k];
      if ((double)lo > box.lo[k]) lo = std::nextafter(lo, -INFINITY);
      if ((double)hi < box.hi[k]) hi = std::nextafter(hi, INFINITY);
      if (!(lo == lo)) lo = -3.0e38f;
      if (!(hi == hi)) hi = 3.0e38f;
      nodes[idx].lo[k] = std::max(lo, -3.0e38f); nodes[idx].hi[k] = std::min(hi, 3.0e38f);
    }
    nodes[idx].first = first;
    nodes[idx].skip_count = ((uint32_t)leaf_count << 28) | (uint32_t)nodes.size();
    return box;
  }
};

// The nearest-hit tree (Accel::hpair): a complete binary tree in heap layout over the groups of <= 8 consecutive records
// of one type.  Built top down: the subtree sizes are fixed by the heap shape, the groups of a node are split by an
// nth_element of their centres along the axis that gives the smallest (area x count) sum of the two child boxes.
struct HeapTree {
  std::vector<float> pair;        // [max(n_groups - 1, 1)][12]
  std::vector<uint8_t> axis;      // [max(n_groups - 1, 1)]
  std::vector<uint32_t> leaf;     // [n_groups]
  int n_groups = 0, n_sphere_groups = 0;
};
static double box_area(const BoxD& b) {
  if (b.lo[0] > b.hi[0]) return 0.0;
  const double x = b.hi[0] - b.lo[0], y = b.hi[1] - b.lo[1], z = b.hi[2] - b.lo[2];
  return x * y + y * z + z * x;
}
static void heap_build_node(int i, int n_g, uint32_t* gl, int m, const std::vector<int>& leaves_below, const std::vector<BoxD>& gbox,
                            std::vector<BoxD>& nbox, HeapTree& t) {
  BoxD& b = nbox[i];
  for (int k = 0; k < 3; ++k) { b.lo[k] = 1e300; b.hi[k] = -1e300; }
  if (i >= n_g - 1) { t.leaf[i - (n_g - 1)] = gl[0]; b = gbox[gl[0]]; return; }
  const int nl = leaves_below[2 * i + 1];
  int best_axis = 0; double best_cost = 1e300;
  std::vector<uint32_t> tmp(gl, gl + m), best_order;
  for (int axis = 0; axis < 3; ++axis) {
    std::copy(gl, gl + m, tmp.begin());
    std::nth_element(tmp.begin(), tmp.begin() + nl, tmp.end(), [&](uint32_t a, uint32_t c) {
      const double ca = gbox[a].lo[axis] + gbox[a].hi[axis], cc = gbox[c].lo[axis] + gbox[c].hi[axis];
      return ca < cc || (ca == cc && a < c); });
    BoxD l, r;
    for (int k = 0; k < 3; ++k) { l.lo[k] = r.lo[k] = 1e300; l.hi[k] = r.hi[k] = -1e300; }
    for (int q = 0; q < nl; ++q) BvhBuilder::grow(&l, gbox[tmp[q]]);
    for (int q = nl; q < m; ++q) BvhBuilder::grow(&r, gbox[tmp[q]]);
    const double cost = box_area(l) * nl + box_area(r) * (m - nl);
    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_order = tmp; }
  }
  std::copy(best_order.begin(), best_order.end(), gl);
  heap_build_node(2 * i + 1, n_g, gl, nl, leaves_below, gbox, nbox, t);
  heap_build_node(2 * i + 2, n_g, gl + nl, m - nl, leaves_below, gbox, nbox, t);
  BvhBuilder::grow(&b, nbox[2 * i + 1]); BvhBuilder::grow(&b, nbox[2 * i + 2]);
  t.axis[i] = (uint8_t)(best_axis | 4);          // the left child holds the lower centres: first for d[axis] > 0
}
HeapTree build_heap(const std::vector<Prim>& prims, int nS, int box_first, int nB) {
  HeapTree t;
  t.n_sphere_groups = (nS + 7) / 8;
  const int n_g = t.n_sphere_groups + (nB + 7) / 8;
  t.n_groups = n_g;
  if (n_g <= 0) return t;
  std::vector<BoxD> gbox(n_g);
  for (int g = 0; g < n_g; ++g) {
    BoxD& b = gbox[g];
    for (int k = 0; k < 3; ++k) { b.lo[k] = 1e300; b.hi[k] = -1e300; }
    const bool sph = g < t.n_sphere_groups;
    const int j = sph ? g : g - t.n_sphere_groups, first = (sph ? 0 : box_first) + 8 * j, n = std::min(8, (sph ? nS : nB) - 8 * j);
    for (int r = 0; r < n; ++r) BvhBuilder::grow(&b, prim_box(prims[first + r]));
  }
  const int n_nodes = 2 * n_g - 1, n_inner = n_g - 1;
  std::vector<int> leaves_below(n_nodes);
  for (int i = n_nodes - 1; i >= 0; --i) leaves_below[i] = i >= n_inner ? 1 : leaves_below[2 * i + 1] + leaves_below[2 * i + 2];
  std::vector<BoxD> nbox(n_nodes);
  std::vector<uint32_t> gl(n_g);
  for (int g = 0; g < n_g; ++g) gl[g] = (uint32_t)g;
  t.leaf.resize(n_g); t.axis.assign(std::max(n_inner, 1), 0); t.pair.assign((size_t)std::max(n_inner, 1) * 12, 0.0f);
  heap_build_node(0, n_g, gl.data(), n_g, leaves_below, gbox, nbox, t);
  auto down = [](double v) { float f = (float)v; if ((double)f > v) f = std::nextafter(f, -INFINITY); if (!(f == f)) f = -3.0e38f; return std::max(f, -3.0e38f); };
  auto up = [](double v) { float f = (float)v; if ((double)f < v) f = std::nextafter(f, INFINITY); if (!(f == f)) f = 3.0e38f; return std::min(f, 3.0e38f); };
  for (int i = 0; i < std::max(n_inner, 1); ++i) {
    float* q = &t.pair[(size_t)i * 12];
    for (int c = 0; c < 2; ++c) {
      const bool have = i < n_inner;
      for (int k = 0; k < 3; ++k) {
        q[2 * k + c] = have ? down(nbox[2 * i + 1 + c].lo[k]) : 3.0e38f;            // (no inner node: an empty pair, never loaded for a test)
        q[6 + 2 * k + c] = have ? up(nbox[2 * i + 1 + c].hi[k]) : -3.0e38f;
      }
    }
  }
  return t;
}

// Morton order of the records [first, first+count) by bound centre (stable for equal codes).
void morton_sort(std::vector<Prim>& prims, int first, int count) {
  if (count <= 8) return;
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int i = 0; i < count; ++i) { BoundD b = prim_bound(prims[first + i]); for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.c[k]); hi[k] = std::max(hi[k], b.c[k]); } }
  std::vector<std::pair<uint32_t, int>> key(count);
  for (int i = 0; i < count; ++i) {
    BoundD b = prim_bound(prims[first + i]);
    uint32_t code = 0;
    for (int k = 0; k < 3; ++k) {
      // one scale for all axes: cells stay cubical in flat scenes (a segment then crosses few of them)
      double ext = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
      double u = ext > 0 ? (b.c[k] - lo[k]) / ext : 0.0;
      if (!(u >= 0)) u = 0;
      if (u > 1) u = 1;
      code |= spread10((uint32_t)(u * 1023.0)) << k;
    }
    key[i] = {code, i};
  }
  std::stable_sort(key.begin(), key.end(), [](const std::pair<uint32_t, int>& a, const std::pair<uint32_t, int>& b) { return a.first < b.first; });
  std::vector<Prim> tmp(count);
  for (int i = 0; i < count; ++i) tmp[i] = prims[first + key[i].second];
  std::copy(tmp.begin(), tmp.end(), prims.begin() + first);
}

// Order of the records [first, first+count) that makes every group the builder forms (leaf * fanout^k
// consecutive records, BvhBuilder::build) spatially compact: the range is bisected at group
// boundaries along the longest axis of its centres (nth_element), recursively down to the leaves.
// Unlike a Morton order it has no curve jumps inside a group, so the node boxes are tighter.
void median_order(std::vector<Prim>& prims, std::vector<BoundD>& ctr, int first, int count, int leaf, int fanout);
static void median_split(std::vector<Prim>& prims, std::vector<BoundD>& ctr, int first, int count, int group) {
  const int ngroups = (count + group - 1) / group;
  if (ngroups <= 1) return;
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int i = first; i < first + count; ++i)
    for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], ctr[i].c[k]); hi[k] = std::max(hi[k], ctr[i].c[k]); }
  int axis = 0;
  for (int k = 1; k < 3; ++k) if (hi[k] - lo[k] > hi[axis] - lo[axis]) axis = k;
  const int left = (ngroups / 2) * group;
  std::vector<int> idx(count);
  for (int i = 0; i < count; ++i) idx[i] = first + i;
  std::nth_element(idx.begin(), idx.begin() + left, idx.end(), [&](int a, int b) {
    return ctr[a].c[axis] < ctr[b].c[axis] || (ctr[a].c[axis] == ctr[b].c[axis] && a < b); });
  std::vector<Prim> tp(count); std::vector<BoundD> tc(count);
  for (int i = 0; i < count; ++i) { tp[i] = prims[idx[i]]; tc[i] = ctr[idx[i]]; }
  std::copy(tp.begin(), tp.end(), prims.begin() + first);
  std::copy(tc.begin(), tc.end(), ctr.begin() + first);
  median_split(prims, ctr, first, left, group);
  median_split(prims, ctr, first + left, count - left, group);
}
void median_order(std::vector<Prim>& prims, std::vector<BoundD>& ctr, int first, int count, int leaf, int fanout) {
  if (count <= leaf) return;
  long long child = leaf; while (child * fanout < count) child *= fanout;       // the builder's group size at this level
  median_split(prims, ctr, first, count, (int)child);
  for (long long off = 0; off < count; off += child)
    median_order(prims, ctr, first + (int)off, (int)std::min<long long>(child, count - off), leaf, fanout);
}
}  // namespace

// ------------------------------------------------------------------------------- scene ---------
struct EventPair { hipEvent_t a, b; int kind; };   // kind 0 trace, 1 shade, 2 other

struct PendingFrame {
  rt_stats* user = nullptr;
  Counters* h_counters = nullptr;   // pinned slot
  std::vector<EventPair> events;
  int levels = 0, n_lights = 0, n_cells = 1, precision = 0;
  bool timing = false;
  // numpy-stream mode: counted on the host per batch (depth-first batches, not levels)
  bool stream = false;
  int64_t s_rays = 0, s_hits = 0, s_queries = 0; int s_batches = 0;
  int64_t s_level_rays[MAX_LEVELS] = {0}, s_level_hits[MAX_LEVELS] = {0};
};

// Device buffers one frame's kernels work in: the scene's own (grown on demand, ensure_workspace) or a
// carve-up of the static scratch array (rt_render_static).
struct WorkBufs {
  void* fb = nullptr;                 // pixels*3 doubles (also used as float)
  HitRec* hits = nullptr;
  HitRec* hits_sorted = nullptr; uint16_t* sort_keys = nullptr; uint32_t* sort_table = nullptr;   // sort == true only
  double* qbuf[2] = {nullptr, nullptr};   // 9*cap_queue doubles each
  uint32_t* qpix[2] = {nullptr, nullptr};
  Counters* counters = nullptr;
  size_t cap_queue = 0;
  bool sort = false;                  // regroup the hits of level >= 1 by grid cell (large scenes)
  FixRec* fix = nullptr;              // fp32 mode: (hit, light) pairs queued for the fp64 re-test (k_fix_exact)
  size_t cap_fix = 0;
};

struct rt_scene {
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  int num_cu = 256;
  DevScene ds{};
  Prim* d_prims = nullptr; double* d_mats = nullptr; double* d_lights = nullptr;
  int32_t* d_cut = nullptr;
  BvhNode* d_nodes = nullptr; float4* d_pbound = nullptr;   // culling hierarchy (large scenes only)
  float* d_hpair = nullptr; uint8_t* d_haxis = nullptr; uint32_t* d_hleaf = nullptr;    // heap-layout tree for nearest-hit walks
  bool pool_trace = false;                                   // bounce levels through k_trace_pool (needs the heap trees)
  bool all_opaque = true;
  int n_coincident = 0;    // primitives marked by mark_coincident (scenes with transparency)
  int shade_variant = 0;   // RT_SHADE=ls|general: first fp32 shade kernel tried (A/B testing)
  // A used material with shininess 0: in the sequential renderer its specular term is a STEP at r.v = 0 (added only for
  // r.v > 0, where it is 1: ray_tracer.py:356-359).  k_shade_ls forms the Phong light direction from the offset point p
  // (1e-6 from the hit point), an error of 1e-6 / |tl| in r.v that decides the step for |r.v| below it (scaled_survey.py
  // SEQ=1: 9 of 6,500 scenes at scale 0.002 off by up to 84 levels); k_shade_fast forms it from the hit point itself, so
  // sequential fp32 frames of such scenes go there.
  bool any_shin0 = false;
  // workspace
  size_t cap_pixels = 0, cap_queue = 0, cap_hits = 0, cap_out = 0;
  size_t min_queue = 0;          // queue capacity requested after an overflow (rays per level)
  int retry_attempt = 0;         // rt_render's retry loop: which attempt is running
  size_t overflow_need = 0;      // largest level size seen by the last overflowed frame
  FixRec* d_fix = nullptr; size_t cap_fix = 0, min_fix = 0;   // re-test queue of the fp32 mode
  int sticky_overflow = 0;       // an RT_FLAG_ASYNC frame overflowed since the last rt_sync / blocking rt_render
  hipStream_t copy_stream = nullptr;   // host-output frames: device -> host copies of finished row bands (render_banded)
  void* d_fb = nullptr;          // cap_fb pixels * 3 doubles (also used as float); only frames that do not render straight into the caller's device buffer need it
  size_t cap_fb = 0;
  void* d_out = nullptr;         // conversion / staging target
  void* d_tile = nullptr; size_t cap_tile = 0;   // rt_render_multi: this handle's rows before they are copied into the frame
  HitRec* d_hits = nullptr;
  // large scenes: hits of level >= 1 are regrouped by grid cell before shading (rt_sort.hpp)
  bool sort_hits = false; SortGrid sort_grid{};
  HitRec* d_hits_sorted = nullptr; uint16_t* d_sort_keys = nullptr; uint32_t* d_sort_table = nullptr;
  double* d_qbuf[2] = {nullptr, nullptr};   // 9*cap_queue doubles each
  uint32_t* d_qpix[2] = {nullptr, nullptr};
  Counters* d_counters = nullptr;
  std::vector<Counters*> h_slots;           // pinned
  std::deque<PendingFrame> pending;
  std::vector<hipEvent_t> event_pool;
  // rt_render_static: this device's copy of the static scratch array (hipGetSymbolAddress at scene creation)
  // and a pinned counters block, so that the call itself allocates nothing
  unsigned char* static_scratch = nullptr;
  Counters* h_static = nullptr;
  bool static_async = false;     // an RT_FLAG_ASYNC frame of rt_render_static has copied its counters to h_static since the last rt_sync
  std::mutex mu;
};

static void free_workspace(rt_scene* s) {
  if (s->d_fb) hipFree(s->d_fb);
  if (s->d_out) hipFree(s->d_out);
  if (s->d_hits) hipFree(s->d_hits);
  if (s->d_hits_sorted) hipFree(s->d_hits_sorted);
  if (s->d_sort_keys) hipFree(s->d_sort_keys);
  if (s->d_sort_table) hipFree(s->d_sort_table);
  if (s->d_fix) hipFree(s->d_fix);
  s->d_fix = nullptr; s->cap_fix = 0;
  s->d_hits_sorted = nullptr; s->d_sort_keys = nullptr; s->d_sort_table = nullptr;
  for (int i = 0; i < 2; ++i) { if (s->d_qbuf[i]) hipFree(s->d_qbuf[i]); if (s->d_qpix[i]) hipFree(s->d_qpix[i]); }
  s->d_fb = s->d_out = nullptr; s->d_hits = nullptr;
  s->d_qbuf[0] = s->d_qbuf[1] = nullptr; s->d_qpix[0] = s->d_qpix[1] = nullptr;
  s->cap_pixels = s->cap_queue = s->cap_hits = s->cap_out = s->cap_fb = 0;
}

// Scenes with transparency, fp32 mode.  The reference's shadow walk steps 1e-6 past every crossing and accepts the next
// one only beyond another 1e-6 (ray_tracer.py:287-288 with the surfaces' t > 1e-6 thresholds), so two crossings within
// 2e-6 of each other along a segment count ONCE; the fp32 kernels' product over all crossings counts both.  Between
// different primitives that is systematic only where surfaces coincide: box faces on box faces or on a plane, duplicated
// spheres / planes (tangent contacts are a set of measure zero).  One primitive of every such pair is marked here
// (Prim::pad bit 0); a (hit, light) pair whose samples cross a marked primitive is left to the exact walk.  A box with a
// face in a plane gets bit 1: it only matters for samples that also cross a plane (boxes standing on a floor are the
// normal case, shadow segments through the floor are not).
static int mark_coincident(std::vector<Prim>& prims) {
  const double TOL = 2.5e-6;
  int marked = 0;
  auto mark = [&](Prim& p) { if (!(p.pad & 1)) { p.pad |= 1; ++marked; } };
  struct Face { double c; int prim; };
  for (int ax = 0; ax < 3; ++ax) {
    const int u = (ax + 1) % 3, v = (ax + 2) % 3;
    std::vector<Face> f;
    for (size_t i = 0; i < prims.size(); ++i)
      if (prims[i].type == RT_BOX) { f.push_back({prims[i].a[ax], (int)i}); f.push_back({prims[i].a[3 + ax], (int)i}); }
    std::sort(f.begin(), f.end(), [](const Face& x, const Face& y) { return x.c < y.c; });
    // groups of faces whose coordinates chain within TOL; inside a group a sweep along u keeps the faces whose
    // u-interval reaches the current one (a floor of n tiles: one row of them), and those are compared in v
    for (size_t g0 = 0; g0 < f.size();) {
      size_t g1 = g0 + 1;
      while (g1 < f.size() && f[g1].c - f[g1 - 1].c <= TOL) ++g1;
      if (g1 - g0 >= 2) {
        std::vector<Face> grp(f.begin() + g0, f.begin() + g1);
        std::sort(grp.begin(), grp.end(), [&](const Face& x, const Face& y) { return prims[x.prim].a[u] < prims[y.prim].a[u]; });
        std::vector<size_t> active;
        long long budget = 64ll * (long long)grp.size() + 100000;      // pathological overlap: give up and mark
        for (size_t i = 0; i < grp.size(); ++i) {
          Prim& A = prims[grp[i].prim];
          size_t w = 0;
          for (size_t k = 0; k < active.size(); ++k)
            if (prims[grp[active[k]].prim].a[3 + u] + TOL >= A.a[u]) active[w++] = active[k];
          active.resize(w);
          for (size_t k = 0; k < active.size(); ++k) {
            const Face& fb = grp[active[k]];
            if (fb.prim == grp[i].prim || std::fabs(fb.c - grp[i].c) > TOL) continue;
            if (--budget < 0) { mark(A); break; }
            const Prim& B = prims[fb.prim];
            if (A.a[v] <= B.a[3 + v] + TOL && B.a[v] <= A.a[3 + v] + TOL) { mark(A); break; }
          }
          active.push_back(i);
        }
      }
      g0 = g1;
    }
    for (const Prim& pl : prims) {            // a box face lying in a plane
      if (pl.type != RT_PLANE || std::fabs(pl.a[ax]) < 1.0 - 1e-9) continue;
      const double c = pl.a[3] / pl.a[ax];
      for (Prim& b : prims)
        if (b.type == RT_BOX && (std::fabs(b.a[ax] - c) <= TOL || std::fabs(b.a[3 + ax] - c) <= TOL) && !(b.pad & 2)) { b.pad |= 2; ++marked; }
    }
  }
  std::vector<Face> sp;
  for (size_t i = 0; i < prims.size(); ++i) if (prims[i].type == RT_SPHERE) sp.push_back({prims[i].a[0], (int)i});
  std::sort(sp.begin(), sp.end(), [](const Face& x, const Face& y) { return x.c < y.c; });
  for (size_t i = 0; i < sp.size(); ++i) {
    Prim& A = prims[sp[i].prim];
    int budget = 4096;
    for (size_t j = i + 1; j < sp.size() && sp[j].c - sp[i].c <= TOL; ++j) {
      if (--budget < 0) { mark(A); break; }
      const Prim& B = prims[sp[j].prim];
      if (std::fabs(A.a[1] - B.a[1]) <= TOL && std::fabs(A.a[2] - B.a[2]) <= TOL && std::fabs(A.a[3] - B.a[3]) <= TOL) { mark(A); break; }
    }
  }
  std::vector<int> pl;
  for (size_t i = 0; i < prims.size(); ++i) if (prims[i].type == RT_PLANE) pl.push_back((int)i);
  for (size_t i = 0; i < pl.size() && i < 4096; ++i)
    for (size_t j = i + 1; j < pl.size() && j < 4096; ++j) {
      Prim& A = prims[pl[i]]; const Prim& B = prims[pl[j]];
      const double d = A.a[0] * B.a[0] + A.a[1] * B.a[1] + A.a[2] * B.a[2];
      if (std::fabs(d) >= 1.0 - 1e-12 && std::fabs(A.a[3] - (d > 0 ? B.a[3] : -B.a[3])) <= TOL) { mark(A); break; }
    }
  return marked;
}

// Scenes with transparency, fp32 mode, second kind of mark (Prim::pad bit 2).  Two crossings of DIFFERENT primitives
// less than 2e-6 apart along a shadow segment also count once in the reference's walk (ray_tracer.py:279-291) — not only
// on coincident faces (mark_coincident) but on every segment that passes the curve where two surfaces meet: the corner
// line of floor and wall, the circle where a sphere is cut by a plane or another sphere, a sphere resting on a surface
// (contact point).  Those are 1-dimensional sets of segments, so the fp32 kernels can test for them per SAMPLE: the
// members of every pair of primitives whose surfaces come within NEAR_TOL of each other — and of which at least one
// is not opaque (two opaque crossings: the first one already ends the walk with 0) — are marked here, and the kernels
// compare the roots of the marked primitives a sample crosses (k_shade_fast, pair_roots); a sample with two of them within
// 2e-6 + its fp32 error bound sends the (hit, light) pair to the exact walk.  Returns the number of primitives marked.
static int mark_near_pairs(std::vector<Prim>& prims, const std::vector<double>& mats) {
  const double TOL = 4e-6;
  const size_t n = prims.size();
  auto transparent = [&](const Prim& p) { return mats[(size_t)p.mat * MAT_STRIDE + 10] != 0.0; };
  auto tol_of = [&](const Prim& a, const Prim& b) {
    double m = 0.0;
    for (int k = 0; k < 6; ++k) m = std::max(m, std::max(a.type == RT_PLANE && k < 3 ? 0.0 : std::fabs(a.a[k]), b.type == RT_PLANE && k < 3 ? 0.0 : std::fabs(b.a[k])));
    return TOL + 1e-12 * m;
  };
  // do the SURFACES of a and b come within tol of each other?
  auto near = [&](const Prim& a, const Prim& b, double tol) -> bool {
    const Prim* x = &a; const Prim* y = &b;
    if (x->type > y->type) std::swap(x, y);                       // (sphere, plane, box) order
    if (x->type == RT_SPHERE && y->type == RT_SPHERE) {
      const double dx = x->a[0] - y->a[0], dy = x->a[1] - y->a[1], dz = x->a[2] - y->a[2];
      const double d = std::sqrt(dx * dx + dy * dy + dz * dz), r1 = std::fabs(x->a[3]), r2 = std::fabs(y->a[3]);
      return d <= r1 + r2 + tol && d >= std::fabs(r1 - r2) - tol;
    }
    if (x->type == RT_SPHERE && y->type == RT_PLANE)
      return std::fabs(x->a[0] * y->a[0] + x->a[1] * y->a[1] + x->a[2] * y->a[2] - y->a[3]) <= std::fabs(x->a[3]) + tol;
    if (x->type == RT_SPHERE && y->type == RT_BOX) {
      // |q - c| over the box surface ranges over [lo, hi]: hi = farthest corner; lo = distance to the solid box from
      // outside, distance to the nearest face from inside
      double out2 = 0.0, far2 = 0.0, inside_m = 1e300; bool inside = true;
      for (int k = 0; k < 3; ++k) {
        const double c = x->a[k], mn = std::min(y->a[k], y->a[3 + k]), mx = std::max(y->a[k], y->a[3 + k]);
        if (c < mn) { out2 += (mn - c) * (mn - c); inside = false; }
        else if (c > mx) { out2 += (c - mx) * (c - mx); inside = false; }
        else inside_m = std::min(inside_m, std::min(c - mn, mx - c));
        const double f = std::max(std::fabs(c - mn), std::fabs(c - mx)); far2 += f * f;
      }
      const double lo = inside ? inside_m : std::sqrt(out2), hi = std::sqrt(far2), r = std::fabs(x->a[3]);
      return r >= lo - tol && r <= hi + tol;
    }
    if (x->type == RT_PLANE && y->type == RT_PLANE) {
      const double d = x->a[0] * y->a[0] + x->a[1] * y->a[1] + x->a[2] * y->a[2];
      if (std::fabs(d) < 1.0 - 1e-14) return true;                                  // they meet in a line
      return std::fabs(x->a[3] - (d > 0 ? y->a[3] : -y->a[3])) <= tol;             // parallel: only when they coincide
    }
    if (x->type == RT_PLANE && y->type == RT_BOX) {
      double mn = 1e300, mx = -1e300;
      for (int c = 0; c < 8; ++c) {
        const double v = x->a[0] * y->a[(c & 1) ? 3 : 0] + x->a[1] * y->a[(c & 2) ? 4 : 1] + x->a[2] * y->a[(c & 4) ? 5 : 2] - x->a[3];
        mn = std::min(mn, v); mx = std::max(mx, v);
      }
      return mn <= tol && mx >= -tol;
    }
    // box, box: the solids overlap (within tol) and neither lies strictly inside the other
    bool overlap = true, x_in_y = true, y_in_x = true;
    for (int k = 0; k < 3; ++k) {
      const double xl = std::min(x->a[k], x->a[3 + k]), xh = std::max(x->a[k], x->a[3 + k]);
      const double yl = std::min(y->a[k], y->a[3 + k]), yh = std::max(y->a[k], y->a[3 + k]);
      overlap &= xl <= yh + tol && yl <= xh + tol;
      x_in_y &= xl > yl + tol && xh < yh - tol;
      y_in_x &= yl > xl + tol && yh < xh - tol;
    }
    return overlap && !x_in_y && !y_in_x;
  };
  int marked = 0;
  auto mark = [&](Prim& p) { if (!(p.pad & 4)) { p.pad |= 4; ++marked; } };
  auto consider = [&](Prim& a, Prim& b) {
    if ((a.pad & 4) && (b.pad & 4)) return;
    if ((a.pad | b.pad) & 1) return;        // any crossing of a primitive marked by mark_coincident goes to the exact walk already
    if (!transparent(a) && !transparent(b)) return;
    if (near(a, b, tol_of(a, b))) { mark(a); mark(b); }
  };
  std::vector<int> planes, solids;
  for (size_t i = 0; i < n; ++i) (prims[i].type == RT_PLANE ? planes : solids).push_back((int)i);
  for (size_t i = 0; i < planes.size(); ++i) {
    for (size_t j = i + 1; j < planes.size() && j < i + 4096; ++j) consider(prims[planes[i]], prims[planes[j]]);
    for (int k : solids) consider(prims[planes[i]], prims[k]);
  }
  // spheres and boxes: sweep over the x-extent of their bounds
  struct Ext { double lo, hi; int prim; };
  std::vector<Ext> ext; ext.reserve(solids.size());
  std::vector<BoxD> bx(n);
  for (int k : solids) { bx[k] = prim_box(prims[k]); ext.push_back({bx[k].lo[0] - 2 * TOL, bx[k].hi[0] + 2 * TOL, k}); }
  std::sort(ext.begin(), ext.end(), [](const Ext& a, const Ext& b) { return a.lo < b.lo; });
  long long budget = 200000000ll;       // pathological overlap (everything on top of everything): mark the rest
  for (size_t i = 0; i < ext.size(); ++i)
    for (size_t j = i + 1; j < ext.size() && ext[j].lo <= ext[i].hi; ++j) {
      if (--budget < 0) { mark(prims[ext[i].prim]); mark(prims[ext[j].prim]); continue; }
      const BoxD &A = bx[ext[i].prim], &B = bx[ext[j].prim];
      if (A.lo[1] > B.hi[1] + 2 * TOL || B.lo[1] > A.hi[1] + 2 * TOL || A.lo[2] > B.hi[2] + 2 * TOL || B.lo[2] > A.hi[2] + 2 * TOL) continue;
      consider(prims[ext[i].prim], prims[ext[j].prim]);
    }
  return marked;
}

extern "C" int rt_scene_create(const rt_scene_desc* desc, int device, rt_scene** out) try {
  if (!desc || !out) return fail(RT_ERR_INVALID, "rt_scene_create: null pointer");
  *out = nullptr;
  if (desc->n_surfaces < 0 || desc->n_materials < 0 || desc->n_lights < 0)
    return fail(RT_ERR_INVALID, "rt_scene_create: negative count");
  if (desc->n_surfaces > 0 && (!desc->surf_type || !desc->surf_data || !desc->surf_material))
    return fail(RT_ERR_INVALID, "rt_scene_create: null surface arrays");
  if ((desc->n_materials > 0 && !desc->materials) || (desc->n_lights > 0 && !desc->lights))
    return fail(RT_ERR_INVALID, "rt_scene_create: null material/light arrays");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(RT_ERR_NO_DEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(RT_ERR_INVALID, "device %d out of range (have %d)", device, ndev);

  // type-sort, keeping file order inside each type and the original list index for tie-breaks
  std::vector<Prim> prims;
  prims.reserve(desc->n_surfaces);
  int cnt[3] = {0, 0, 0};
  for (int type = 0; type < 3; ++type)
    for (int i = 0; i < desc->n_surfaces; ++i) {
      int ty = desc->surf_type[i];
      if (ty < 0 || ty > 2) return fail(RT_ERR_INVALID, "surface %d: unknown type %d", i, ty);
      if (ty != type) continue;
      int m = desc->surf_material[i];
      if (m < 0 || m >= desc->n_materials) return fail(RT_ERR_INVALID, "surface %d: material %d out of range", i, m);
      Prim p; std::memset(&p, 0, sizeof p);
      const double* a = desc->surf_data + 8 * (size_t)i;
      for (int k = 0; k < 6; ++k) p.a[k] = a[k];
      if (ty == RT_SPHERE) { p.a[4] = a[3] * a[3]; p.a[5] = 0; }
      if (ty == RT_PLANE) { p.a[4] = p.a[5] = 0; }
      p.mat = m; p.orig = i; p.type = ty;
      prims.push_back(p);
      ++cnt[type];
    }
  // large scenes: median-split order inside the sphere / box ranges + culling hierarchy (RT_ACCEL=0: brute force)
  float accel_lo[3] = {0, 0, 0}, accel_hi[3] = {0, 0, 0}; int sphere_nodes = 0;   // union of the hierarchy's root boxes
  std::vector<int32_t> cut;        // <= 64 nodes covering both trees: entry points of the wave-level scan
  std::vector<BvhNode> nodes;
  std::vector<float4> pbound;
  HeapTree heap;
  {
    const char* e = getenv("RT_ACCEL");
    const bool want = !(e && e[0] == '0');
    if (want && (int)prims.size() > LDS_PRIMS && cnt[0] + cnt[2] > 0 && prims.size() < (1u << 27)) {
      BvhBuilder bb(prims);
      const char* ord = getenv("RT_BVH_ORDER");
      if (ord && ord[0] == 'm') {                       // A/B: Morton order of the centres
        morton_sort(prims, 0, cnt[0]);
        morton_sort(prims, cnt[0] + cnt[1], cnt[2]);
      } else {
        std::vector<BoundD> ctr(prims.size());
        for (size_t i = 0; i < prims.size(); ++i) if (prims[i].type != RT_PLANE) ctr[i] = prim_bound(prims[i]);
        median_order(prims, ctr, 0, cnt[0], bb.leaf, bb.fanout);
        median_order(prims, ctr, cnt[0] + cnt[1], cnt[2], bb.leaf, bb.fanout);
      }
      if (cnt[0] > 0) bb.build(0, cnt[0]);
      sphere_nodes = (int)bb.nodes.size();
      if (cnt[2] > 0) bb.build(cnt[0] + cnt[1], cnt[2]);
      nodes.swap(bb.nodes);
      for (int k = 0; k < 3; ++k) { accel_lo[k] = 3.0e38f; accel_hi[k] = -3.0e38f; }
      for (int root : {0, cnt[0] > 0 && cnt[2] > 0 ? sphere_nodes : 0})          // roots of the sphere and the box tree
        for (int k = 0; k < 3; ++k) { accel_lo[k] = std::min(accel_lo[k], nodes[root].lo[k]); accel_hi[k] = std::max(accel_hi[k], nodes[root].hi[k]); }
      // cut: start from the roots, replace inner nodes by their children level by level while <= 64 nodes
      cut.push_back(0);
      if (cnt[0] > 0 && cnt[2] > 0) cut.push_back(sphere_nodes);
      for (;;) {
        std::vector<int32_t> next;
        bool expanded = false;
        for (int32_t i : cut) {
          if (nodes[i].count() > 0) { next.push_back(i); continue; }
          for (int c = i + 1; c < nodes[i].skip(); c = nodes[c].skip()) next.push_back(c);
          expanded = true;
        }
        if (!expanded || next.size() > 64) break;
        cut.swap(next);
      }
      if (!(getenv("RT_BVH_HEAP") && atoi(getenv("RT_BVH_HEAP")) == 0)) {
        heap = build_heap(prims, cnt[0], cnt[0] + cnt[1], cnt[2]);
      }
      pbound.resize(prims.size());
      for (size_t i = 0; i < prims.size(); ++i)
        pbound[i] = prims[i].type == RT_PLANE ? make_float4(0, 0, 0, 3.0e38f) : to_float_bound(prim_bound(prims[i]));
    }
  }
  std::vector<double> mats((size_t)std::max(desc->n_materials, 1) * MAT_STRIDE, 0.0);
  for (int i = 0; i < desc->n_materials; ++i)
    for (int k = 0; k < 11; ++k) mats[(size_t)i * MAT_STRIDE + k] = desc->materials[(size_t)i * 11 + k];
  std::vector<double> lights((size_t)std::max(desc->n_lights, 1) * LIGHT_STRIDE, 0.0);
  for (int i = 0; i < desc->n_lights; ++i)
    for (int k = 0; k < 9; ++k) lights[(size_t)i * LIGHT_STRIDE + k] = desc->lights[(size_t)i * 9 + k];

  bool any_transparent = false;
  for (const Prim& p : prims) if (mats[(size_t)p.mat * MAT_STRIDE + 10] != 0.0) any_transparent = true;
  int n_coincident = any_transparent ? mark_coincident(prims) : 0;
  if (any_transparent && !(getenv("RT_NEAR_PAIRS") && atoi(getenv("RT_NEAR_PAIRS")) == 0)) {
    const int n_near = mark_near_pairs(prims, mats);
    if (getenv("RT_DEBUG_STATS")) fprintf(stderr, "[rt debug] scene of %zu primitives: %d marked as coincident, %d as meeting another surface\n", prims.size(), n_coincident, n_near);
    n_coincident += n_near;
  }

  rt_scene* s = new (std::nothrow) rt_scene();
  if (!s) return fail(RT_ERR_NOMEM, "out of host memory");
  s->device = device;
  s->n_coincident = n_coincident;
  auto bail = [&](int code) { rt_scene_destroy(s); return code; };
#define HIPCHK_S(call)                                                                         \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) return bail(fail(e_ == hipErrorOutOfMemory ? RT_ERR_NOMEM : RT_ERR_HIP, \
                                           "%s failed: %s", #call, hipGetErrorString(e_)));      \
  } while (0)
  HIPCHK_S(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK_S(hipGetDeviceProperties(&prop, device));
  s->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  HIPCHK_S(hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking));
  s->stream = s->own_stream;
  size_t np = std::max<size_t>(prims.size(), 1);
  HIPCHK_S(hipMalloc(&s->d_prims, np * sizeof(Prim)));
  HIPCHK_S(hipMalloc(&s->d_mats, mats.size() * sizeof(double)));
  HIPCHK_S(hipMalloc(&s->d_lights, lights.size() * sizeof(double)));
  HIPCHK_S(hipMalloc(&s->d_counters, sizeof(Counters)));
  HIPCHK_S(hipHostMalloc(&s->h_static, sizeof(Counters)));
  { int rc_ = static_scratch_for(device, &s->static_scratch); if (rc_ != RT_OK) return bail(rc_); }
  if (!prims.empty()) HIPCHK_S(hipMemcpy(s->d_prims, prims.data(), prims.size() * sizeof(Prim), hipMemcpyHostToDevice));
  HIPCHK_S(hipMemcpy(s->d_mats, mats.data(), mats.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK_S(hipMemcpy(s->d_lights, lights.data(), lights.size() * sizeof(double), hipMemcpyHostToDevice));
  if (!nodes.empty()) {
    HIPCHK_S(hipMalloc(&s->d_nodes, nodes.size() * sizeof(BvhNode)));
    HIPCHK_S(hipMalloc(&s->d_pbound, pbound.size() * sizeof(float4)));
    HIPCHK_S(hipMemcpy(s->d_nodes, nodes.data(), nodes.size() * sizeof(BvhNode), hipMemcpyHostToDevice));
    HIPCHK_S(hipMemcpy(s->d_pbound, pbound.data(), pbound.size() * sizeof(float4), hipMemcpyHostToDevice));
    HIPCHK_S(hipMalloc(&s->d_cut, 64 * sizeof(int32_t)));
    HIPCHK_S(hipMemcpy(s->d_cut, cut.data(), cut.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  if (heap.n_groups > 0) {
    HIPCHK_S(hipMalloc(&s->d_hpair, heap.pair.size() * sizeof(float)));
    HIPCHK_S(hipMalloc(&s->d_haxis, heap.axis.size()));
    HIPCHK_S(hipMalloc(&s->d_hleaf, heap.leaf.size() * sizeof(uint32_t)));
    HIPCHK_S(hipMemcpy(s->d_hpair, heap.pair.data(), heap.pair.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK_S(hipMemcpy(s->d_haxis, heap.axis.data(), heap.axis.size(), hipMemcpyHostToDevice));
    HIPCHK_S(hipMemcpy(s->d_hleaf, heap.leaf.data(), heap.leaf.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
#undef HIPCHK_S
  s->pool_trace = s->d_hpair != nullptr && !(getenv("RT_TRACE_POOL") && atoi(getenv("RT_TRACE_POOL")) == 0);
  s->ds.ac.hpair = reinterpret_cast<const float4*>(s->d_hpair); s->ds.ac.haxis = s->d_haxis; s->ds.ac.hleaf = s->d_hleaf;
  s->ds.ac.n_groups = heap.n_groups; s->ds.ac.n_sphere_groups = heap.n_sphere_groups; s->ds.ac.box_first = cnt[0] + cnt[1];
  s->ds.ac.h_nS = cnt[0]; s->ds.ac.h_nB = cnt[2];
  s->ds.ac.lds_hpair = s->ds.ac.lds_haxis = s->ds.ac.lds_hleaf = -1; s->ds.ac.n_lds_inner = 0;
  s->ds.ac.nodes = s->d_nodes; s->ds.ac.pbound = s->d_pbound; s->ds.ac.n_nodes = (int)nodes.size(); s->ds.ac.lds_nodes = -1;
  s->ds.ac.cut = s->d_cut; s->ds.ac.n_cut = (int)cut.size(); s->ds.ac.pad = 0;
  for (int k = 0; k < 3; ++k) { s->ds.ac.lo[k] = accel_lo[k]; s->ds.ac.hi[k] = accel_hi[k]; }
  s->ds.prims = s->d_prims; s->ds.mats = s->d_mats; s->ds.lights = s->d_lights;
  s->ds.nS = cnt[0]; s->ds.nP = cnt[1]; s->ds.nB = cnt[2]; s->ds.n = (int)prims.size();
  s->ds.nM = desc->n_materials; s->ds.nL = desc->n_lights;
  for (int k = 0; k < 3; ++k) s->ds.bg[k] = desc->background[k];
  s->all_opaque = true;
  for (const Prim& p : prims) if (mats[(size_t)p.mat * MAT_STRIDE + 10] != 0.0) s->all_opaque = false;
  s->ds.all_opaque = s->all_opaque ? 1 : 0;
  for (const Prim& p : prims) if (mats[(size_t)p.mat * MAT_STRIDE + 9] == 0.0) s->any_shin0 = true;
  s->ds.scene_m = 0.0f;       // L1 size of the hierarchy's bounds: scales the fp32 slack of k_shade_fast's node tests
  for (int k = 0; k < 3; ++k) s->ds.scene_m += std::max(std::fabs(accel_lo[k]), std::fabs(accel_hi[k]));
  if (const char* e = getenv("RT_SHADE")) s->shade_variant = e[0] == 'g' ? 2 : 0;
  if (!nodes.empty() && !(getenv("RT_SORT_HITS") && atoi(getenv("RT_SORT_HITS")) == 0)) {
    // grid over the hierarchy's bounds: 13 bits, each given to the axis whose cells are largest
    s->sort_hits = true;
    auto make_grid = [&](SortGrid& g, int total_bits) {
      float ext[3]; int bits[3] = {0, 0, 0};
      for (int k = 0; k < 3; ++k) { g.lo[k] = accel_lo[k]; ext[k] = std::max(accel_hi[k] - accel_lo[k], 1e-20f); }
      for (int b = 0; b < total_bits; ++b) {
        int best = 0;
        for (int k = 1; k < 3; ++k) if (std::ldexp(ext[k], -bits[k]) > std::ldexp(ext[best], -bits[best])) best = k;
        ++bits[best];
      }
      for (int k = 0; k < 3; ++k) { g.n[k] = 1 << bits[k]; g.inv[k] = (float)g.n[k] / ext[k]; }
      g.pad = 0;
    };
    make_grid(s->sort_grid, 13);
  }
  *out = s;
  return RT_OK;
} RT_CATCH_ALL

extern "C" void rt_scene_destroy(rt_scene* s) try {
  if (!s) return;
  hipSetDevice(s->device);
  if (s->stream) hipStreamSynchronize(s->stream);
  free_workspace(s);
  if (s->d_prims) hipFree(s->d_prims);
  if (s->d_mats) hipFree(s->d_mats);
  if (s->d_lights) hipFree(s->d_lights);
  if (s->d_nodes) hipFree(s->d_nodes);
  if (s->d_pbound) hipFree(s->d_pbound);
  if (s->d_tile) hipFree(s->d_tile);
  if (s->d_hpair) hipFree(s->d_hpair);
  if (s->d_haxis) hipFree(s->d_haxis);
  if (s->d_hleaf) hipFree(s->d_hleaf);
  if (s->d_cut) hipFree(s->d_cut);
  if (s->d_counters) hipFree(s->d_counters);
  for (Counters* h : s->h_slots) hipHostFree(h);
  if (s->h_static) hipHostFree(s->h_static);
  for (hipEvent_t e : s->event_pool) hipEventDestroy(e);
  for (PendingFrame& pf : s->pending) {   // frames never synced: their rt_stats are not written
    for (EventPair& ep : pf.events) { hipEventDestroy(ep.a); hipEventDestroy(ep.b); }
    if (pf.h_counters) hipHostFree(pf.h_counters);
  }
  if (s->copy_stream) hipStreamDestroy(s->copy_stream);
  if (s->own_stream) hipStreamDestroy(s->own_stream);
  delete s;
} catch (...) { }

extern "C" int rt_scene_set_stream(rt_scene* s, void* hip_stream) try {
  if (!s) return fail(RT_ERR_INVALID, "rt_scene_set_stream: null scene");
  std::lock_guard<std::mutex> lk(s->mu);
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipStreamSynchronize(s->stream));
  s->stream = hip_stream ? (hipStream_t)hip_stream : s->own_stream;
  return RT_OK;
} RT_CATCH_ALL

extern "C" int rt_rows_in_part(int32_t height, int32_t block_rows, int32_t part, int32_t n_parts, int32_t* rows) try {
  if (!rows || height < 0 || block_rows <= 0) return fail(RT_ERR_INVALID, "rt_rows_in_part: bad arguments");
  if (n_parts <= 1) { *rows = height; return RT_OK; }
  if (part < 0 || part >= n_parts) return fail(RT_ERR_INVALID, "rt_rows_in_part: part %d of %d", part, n_parts);
  int full_blocks = height / block_rows, tail = height % block_rows;
  int mine = full_blocks / n_parts + ((full_blocks % n_parts) > part ? 1 : 0);
  int r = mine * block_rows;
  if (tail && (full_blocks % n_parts) == part) r += tail;
  *rows = r;
  return RT_OK;
} RT_CATCH_ALL

// Fold a finished frame's device counters (and event timings) into the caller's rt_stats.
static void fill_stats(PendingFrame& pf, const Counters& c, rt_stats* out) {
  rt_stats st; std::memset(&st, 0, sizeof st);
  st.levels = pf.levels;
  if (pf.stream) {
    st.rays = pf.s_rays; st.hits = pf.s_hits; st.batches = pf.s_batches;
    for (int l = 0; l < pf.levels; ++l) { st.level_rays[l] = pf.s_level_rays[l]; st.level_hits[l] = pf.s_level_hits[l]; }
  }
  for (int l = 0; l < pf.levels && !pf.stream; ++l) {
    st.level_rays[l] = c.n_rays[l]; st.level_hits[l] = c.n_hits[l];
    st.rays += c.n_rays[l]; st.hits += c.n_hits[l];
    if (c.n_rays[l]) st.batches++;
  }
  st.shadow_rays = st.hits * (int64_t)pf.n_lights * pf.n_cells;
  if (pf.precision == RT_FP32) for (int q = 0; q < FIX_QUEUES; ++q) st.retested += c.fixq[q * 32];
  st.segments = pf.precision == RT_FP64 ? (pf.stream ? pf.s_queries : (int64_t)c.queries) - st.shadow_rays : -1;
  if (pf.timing && !pf.events.empty()) {
    for (EventPair& ep : pf.events) {
      float ms = 0; hipEventElapsedTime(&ms, ep.a, ep.b);
      if (ep.kind == 0) { st.trace_ms += ms; st.trace_launches++; }
      else if (ep.kind == 1) { st.shade_ms += ms; st.shade_launches++; }
      else st.other_ms += ms;
    }
    hipEventElapsedTime(&st.total_ms, pf.events.front().a, pf.events.back().b);
  }
  *out = st;
}

// Finish every pending frame of a scene: wait, fold counters/events into the callers' rt_stats.
static int drain(rt_scene* s) {
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipStreamSynchronize(s->stream));
  int rc = RT_OK;
  while (!s->pending.empty()) {
    PendingFrame pf = std::move(s->pending.front());
    s->pending.pop_front();
    const Counters& c = *pf.h_counters;
    if (c.overflow) {
      size_t need = 0;
      for (int l = 1; l <= MAX_LEVELS; ++l) need = std::max<size_t>(need, c.n_rays[l]);      // (level 0 lives in the hit list only)
      need = std::max<size_t>(need, (size_t)c.seg_need * NSEG);       // the fullest queue segment decides
      s->overflow_need = need;
      // the next frame (blocking retry, or the next async frame) gets larger queues; levels below the truncated
      // one were undercounted, so grow generously
      // (first retry of a call: what the overflowing level asked for plus 15 %; later retries double, because the
      // levels below a truncated one were undercounted)
      s->min_queue = std::max(s->min_queue, std::max(need + need * 3 / 20, s->retry_attempt >= 1 ? 2 * s->cap_queue : (size_t)0));
      rc = fail(RT_ERR_QUEUE_OVERFLOW, "bounce queue overflow (a level holds >= %zu rays, capacity %zu): a blocking "
                "rt_render call grows the queues and retries by itself", need, s->cap_queue);
    }
    if (c.overflow_fix) {
      size_t worst = 0;
      for (int q = 0; q < FIX_QUEUES; ++q) worst = std::max<size_t>(worst, c.fixq[q * 32]);
      s->min_fix = std::max(s->min_fix, (worst + worst / 4 + 1024) * FIX_QUEUES);
      rc = fail(RT_ERR_QUEUE_OVERFLOW, "re-test queue overflow (a sub-queue holds %zu (hit, light) pairs, capacity %zu): a blocking "
                "rt_render call grows the queue and retries by itself", worst, s->cap_fix / FIX_QUEUES);
    }
    if (getenv("RT_DEBUG_STATS") && (atoi(getenv("RT_DEBUG_STATS")) & (1 << 22))) {      // fill of the bounce-queue segments (last two levels)
      for (int par = 0; par < 2; ++par) {
        uint32_t mn = 0xFFFFFFFFu, mx = 0; unsigned long long sum = 0;
        for (int q = 0; q < NSEG; ++q) { const uint32_t v = c.segcnt[par][q * 32]; mn = std::min(mn, v); mx = std::max(mx, v); sum += v; }
        fprintf(stderr, "[rt debug] queue segments (level parity %d): min %u, mean %.0f, max %u (+%.1f %%)\n", par, mn, (double)sum / NSEG, mx,
                sum ? 100.0 * ((double)mx * NSEG / (double)sum - 1.0) : 0.0);
      }
    }
#ifdef RT_DEBUG_KERNEL
    if (getenv("RT_DEBUG_STATS") && (atoi(getenv("RT_DEBUG_STATS")) & (1 << 21))) {
      unsigned long long w[8] = {0}, z[8] = {0};
      hipMemcpyFromSymbol(w, HIP_SYMBOL(rt::g_walk_dbg), sizeof w);
      hipMemcpyToSymbol(HIP_SYMBOL(rt::g_walk_dbg), z, sizeof z);
      const double nw = (double)std::max(1ull, w[6]);
      fprintf(stderr, "[rt debug] per-lane heap walks since the last frame: %llu waves; per wave: %.1f node steps (lanes' mean %.1f), %.1f leaves (%.1f), "
              "%.1f exact tests (%.1f), %.0f cycles\n", w[6], w[0] / nw, w[1] / nw / 64, w[2] / nw, w[3] / nw / 64, w[4] / nw, w[5] / nw / 64, w[7] / nw);
      hipMemcpyFromSymbol(w, HIP_SYMBOL(rt::g_pool_cyc), sizeof w);
      hipMemcpyToSymbol(HIP_SYMBOL(rt::g_pool_cyc), z, sizeof z);
      if (w[5]) fprintf(stderr, "[rt debug] k_trace_pool wave cycles: refill %.3f, node steps %.3f, leaf bounds %.3f, exact tests %.3f, results %.3f of %llu; %llu rounds\n",
                        (double)w[0] / w[5], (double)w[1] / w[5], (double)w[2] / w[5], (double)w[3] / w[5], (double)w[4] / w[5], w[5], w[6]);
    }
#endif
#ifdef RT_DEBUG_KERNEL
    if (pf.precision == RT_FP32 && getenv("RT_DEBUG_STATS") && (atoi(getenv("RT_DEBUG_STATS")) & (1 << 26))) {
      unsigned long long w[12] = {0}, z[12] = {0};
      hipMemcpyFromSymbol(w, HIP_SYMBOL(rt::g_fast_cyc), sizeof w);
      hipMemcpyToSymbol(HIP_SYMBOL(rt::g_fast_cyc), z, sizeof z);
      if (w[10]) {
        const double a = (double)w[10];
        fprintf(stderr, "[rt debug] k_shade_fast wave cycles: per-hit set-up %.3f, light set-up %.3f, wave-level lists %.3f, sample directions %.3f, planes %.3f, "
                "hierarchy phases + sample blocks %.3f, deferred fp64 decisions %.3f, merged-crossing check %.3f, Phong + re-test queue %.3f, accumulate + spawn + grab %.3f of %llu\n",
                w[0] / a, w[1] / a, w[2] / a, w[3] / a, w[4] / a, w[5] / a, w[6] / a, w[7] / a, w[8] / a, w[9] / a, w[10]);
      }
    }
#endif
#ifdef RT_DEBUG_KERNEL
    if (pf.precision == RT_FP32 && getenv("RT_DEBUG_STATS") && (atoi(getenv("RT_DEBUG_STATS")) & (1 << 25))) {
      unsigned long long w[8] = {0}, z[8] = {0};
      hipMemcpyFromSymbol(w, HIP_SYMBOL(rt::g_walk_audit), sizeof w);
      hipMemcpyToSymbol(HIP_SYMBOL(rt::g_walk_audit), z, sizeof z);
      fprintf(stderr, "[rt debug] walk audit: %llu (hit, light) pairs decided in fp32 checked against the exact walk, %llu differ\n", w[0], w[1]);
    }
    if (pf.precision == RT_FP32 && getenv("RT_DEBUG_STATS") && (atoi(getenv("RT_DEBUG_STATS")) & (1 << 23))) {
      unsigned long long w[8] = {0}, z[8] = {0};
      hipMemcpyFromSymbol(w, HIP_SYMBOL(rt::g_fix_cause), sizeof w);
      hipMemcpyToSymbol(HIP_SYMBOL(rt::g_fix_cause), z, sizeof z);
      fprintf(stderr, "[rt debug] k_shade_ls re-test causes: %llu pairs; light's sample square touches a candidate %llu; of the rest (a pair can have several): "
              "grazing discriminant %llu, root at the shadow origin %llu, parallel to a plane %llu, light within 3 radii %llu, far hit %llu\n",
              w[0], w[1], w[2], w[3], w[4], w[5], w[6]);
      hipMemcpyFromSymbol(w, HIP_SYMBOL(rt::g_forced_cause), sizeof w);
      hipMemcpyToSymbol(HIP_SYMBOL(rt::g_forced_cause), z, sizeof z);
      fprintf(stderr, "[rt debug]   forced by special cases: own-sphere root near 2e-6 %llu, own-plane root %llu, direction parallel to a slab %llu, "
              "box edge / root at the sample end %llu, own-box root near 2e-6 %llu\n", w[0], w[1], w[2], w[3], w[4]);
    }
#endif
    if (pf.precision == RT_FP32 && getenv("RT_DEBUG_STATS"))
      fprintf(stderr, "[rt debug] k_shade_ls (wave, light) pairs: %llu total, %llu without a sample loop\n",
              (unsigned long long)(c.queries >> 32), (unsigned long long)(c.queries & 0xFFFFFFFFull));
    if (pf.precision == RT_FP32 && getenv("RT_DEBUG_STATS") && (atoi(getenv("RT_DEBUG_STATS")) & (1 << 20)))
      fprintf(stderr, "[rt debug] k_shade_ls wave cycles: hit setup %.3f, light setup + culling %.3f, sample loops (+ bounds) %.3f, chunk grab + spawn incl. barrier waits %.3f of %llu (rest: Phong, accumulate)\n",
              (double)c.dbg[0] / (double)c.dbg[3], (double)c.dbg[1] / (double)c.dbg[3], (double)c.dbg[2] / (double)c.dbg[3], (double)c.queries / (double)c.dbg[3], (unsigned long long)c.dbg[3]);
    else if (pf.precision == RT_FP32 && getenv("RT_DEBUG_STATS") && (atoi(getenv("RT_DEBUG_STATS")) & 2))
      fprintf(stderr, "[rt debug] shortcut audit: fully-lit claims wrong %llu, umbra claims wrong %llu, penumbra lanes %llu of %llu\n",
              (unsigned long long)c.dbg[0], (unsigned long long)c.dbg[1], (unsigned long long)c.dbg[2], (unsigned long long)c.dbg[3]);
    else if (pf.precision == RT_FP32 && getenv("RT_DEBUG_STATS") && (c.dbg[0] || c.dbg[1])) {
      const int bits = atoi(getenv("RT_DEBUG_STATS"));
      const double n0 = (double)c.dbg[0];
      if (bits & 65536)
        fprintf(stderr, "[rt debug] culling audit: %llu exact (fp64) crossings of records by shadow-segment bundles, %llu bundles with a crossed record that never reached the sample block\n",
                (unsigned long long)c.dbg[1], (unsigned long long)c.dbg[0]);
      else if (bits & 2048)
        fprintf(stderr, "[rt debug] k_shade_fast large-scene walk, share of the waves' cycles: find leaves %.3f, record bounds %.3f, sample blocks %.3f\n",
                c.dbg[1] / n0, c.dbg[2] / n0, c.dbg[3] / n0);
      else if (bits & 1024)
        fprintf(stderr, "[rt debug] k_shade_fast large-scene walks (wave, light): %llu; wave-level steps per walk: %.1f node tests, %.1f leaves, %.2f sample blocks\n",
                (unsigned long long)c.dbg[0], c.dbg[1] / n0, c.dbg[2] / n0, c.dbg[3] / n0);
      else if (bits & 128)
        fprintf(stderr, "[rt debug] k_shade_fast wave leaf lists (wave, light): %llu, %.3f too long, %.1f leaves and %.1f records on average\n",
                (unsigned long long)c.dbg[0], c.dbg[1] / n0, c.dbg[2] / n0, c.dbg[3] / n0);
      else
        fprintf(stderr, "[rt debug] k_shade_fast hierarchy walks (hit, light): %llu; per walk: %.1f nodes tested (holding %.1f record bounds), %.2f records processed\n",
                (unsigned long long)c.dbg[0], c.dbg[1] / n0, c.dbg[2] / n0, c.dbg[3] / n0);
    }
    if (pf.user) fill_stats(pf, c, pf.user);
    for (EventPair& ep : pf.events) { s->event_pool.push_back(ep.a); s->event_pool.push_back(ep.b); }
    s->h_slots.push_back(pf.h_counters);
  }
  return rc;
}

// A call that enqueued frames whose PendingFrame::user points into one of ITS OWN local arrays must not leave them
// pending when it fails: the next drain would write through the dangling pointers.  Called on every error exit of
// such a call (render_banded, rt_render_multi) with the scene's mutex held: the frames stay queued (their pinned
// counter slot and events go back to the pools at the next drain), only their stats are no longer reported.
static void forget_stats(rt_scene* s, const rt_stats* first, size_t n) {
  for (PendingFrame& pf : s->pending)
    if (pf.user >= first && pf.user < first + n) pf.user = nullptr;
}

extern "C" int rt_sync(rt_scene* s) try {
  if (!s) return fail(RT_ERR_INVALID, "rt_sync: null scene");
  std::lock_guard<std::mutex> lk(s->mu);
  int rc = drain(s);
  if (rc == RT_OK && s->sticky_overflow)      // an async frame's overflow that an internal drain (workspace growth) had to swallow
    rc = fail(RT_ERR_QUEUE_OVERFLOW, "an earlier RT_FLAG_ASYNC frame overflowed its bounce queues (they have been grown since): render it again");
  s->sticky_overflow = 0;
  if (s->static_async) {
    // RT_FLAG_ASYNC frames of rt_render_static copy their counters to h_static on the stream drain() has just waited
    // for: the LAST such frame's overflow flags are reported here (earlier ones were overwritten: sync after each
    // frame whose completeness matters).  That entry point cannot grow its queues, so there is nothing to retry with.
    s->static_async = false;
    if (rc == RT_OK && (s->h_static->overflow || s->h_static->overflow_fix))
      rc = fail(RT_ERR_QUEUE_OVERFLOW, "an RT_FLAG_ASYNC frame of rt_render_static overflowed the static %s queue: its image is incomplete "
                "(use rt_render, or render fewer rows per call)", s->h_static->overflow ? "bounce" : "re-test");
  }
  return rc;
} RT_CATCH_ALL

// Bounce-ray queues: a blocking frame starts with 2 slots per pixel and queue as long as that stays below ~3 GB, and
// with as little as 0.55 per pixel for larger row sets (the pool scene's largest bounce level holds 0.42 rays per
// pixel, its fullest queue segment 17 % more than the mean; at 7680x4320 the workspace is then 3.5 GB instead of
// 11 GB; RT_QUEUE_SLOTS_PER_PIXEL overrides), and relies on
// the overflow retry of rt_render, which re-renders with the size the overflowing level asked for; the sizes only
// ever grow, so later frames of the scene start where the last retry ended.  Frames that cannot be retried
// (RT_FLAG_ASYNC) get the old worst-case-ish 2 slots per pixel unless the workspace is larger already.
static int ensure_workspace(rt_scene* s, size_t pixels, size_t out_bytes, bool need_out, bool can_retry, bool need_fb = true) {
  // 2 slots per pixel while the two queues + hit list stay below ~3 GB (up to ~9 M pixels: a 4K frame never needs the
  // retry), shrinking to 0.55 per pixel for larger row sets (8K: 3.5 GB)
  double per_pixel = can_retry ? std::min(2.0, std::max(0.55, 3.0e9 / (168.0 * (double)std::max<size_t>(pixels, 1)))) : 2.0;
  if (const char* e = getenv("RT_QUEUE_SLOTS_PER_PIXEL")) per_pixel = std::min(64.0, std::max(0.01, atof(e)));
  const size_t base_q = (size_t)(per_pixel * (double)pixels) + 65536;      // (the constant: small frames fill their 64 segments unevenly)
  const size_t want_q = std::max<size_t>(base_q, s->min_queue);
  const size_t want_fix = std::max<size_t>(std::max<size_t>(pixels / 8, 65536), s->min_fix);
  if (pixels > s->cap_pixels || want_q > s->cap_queue || want_fix > s->cap_fix) {
    pixels = std::max(pixels, s->cap_pixels);
    int rc = drain(s); if (rc == RT_ERR_QUEUE_OVERFLOW) s->sticky_overflow = 1; else if (rc != RT_OK) return rc;
    void* keep_out = s->d_out; size_t keep_cap = s->cap_out; s->d_out = nullptr;
    free_workspace(s);
    s->d_out = keep_out; s->cap_out = keep_cap;
    size_t q = (std::max<size_t>(std::max(want_q, s->cap_queue), s->min_queue) + NSEG - 1) / NSEG * NSEG;   // NSEG equal segments
    const size_t nh = std::max(q, pixels);                                // hit list: level 0 has one ray per pixel
    if (nh > 0xFFFFFFF0ull) return fail(RT_ERR_INVALID, "row set too large (%zu pixels): render it in parts", pixels);
    HIPCHK(hipMalloc(&s->d_hits, nh * sizeof(HitRec)));
    if (s->sort_hits) {
      HIPCHK(hipMalloc(&s->d_hits_sorted, nh * sizeof(HitRec)));
      HIPCHK(hipMalloc(&s->d_sort_keys, nh * sizeof(uint16_t)));
      // table[bin][block] followed by the per-bin totals
      HIPCHK(hipMalloc(&s->d_sort_table, ((size_t)SORT_BINS * SORT_BLOCKS + SORT_BINS) * sizeof(uint32_t)));
    }
    for (int i = 0; i < 2; ++i) {
      HIPCHK(hipMalloc(&s->d_qbuf[i], q * 9 * sizeof(double)));
      HIPCHK(hipMalloc(&s->d_qpix[i], q * sizeof(uint32_t)));
    }
    const size_t nfix = std::max<size_t>(std::max<size_t>(pixels / 8, 65536), s->min_fix);
    HIPCHK(hipMalloc(&s->d_fix, nfix * sizeof(FixRec)));
    s->cap_pixels = pixels; s->cap_queue = q; s->cap_hits = nh; s->cap_fix = nfix;
  }
  if (need_fb && pixels > s->cap_fb) {
    int rc = drain(s); if (rc == RT_ERR_QUEUE_OVERFLOW) s->sticky_overflow = 1; else if (rc != RT_OK) return rc;
    if (s->d_fb) hipFree(s->d_fb);
    s->d_fb = nullptr; s->cap_fb = 0;
    HIPCHK(hipMalloc(&s->d_fb, pixels * 3 * sizeof(double)));
    s->cap_fb = pixels;
  }
  if (need_out && out_bytes > s->cap_out) {
    int rc = drain(s); if (rc == RT_ERR_QUEUE_OVERFLOW) s->sticky_overflow = 1; else if (rc != RT_OK) return rc;
    if (s->d_out) hipFree(s->d_out);
    s->d_out = nullptr; s->cap_out = 0;
    HIPCHK(hipMalloc(&s->d_out, out_bytes));
    s->cap_out = out_bytes;
  }
  return RT_OK;
}

static RayQueue make_queue(const WorkBufs& wb, int which) {
  RayQueue q; double* b = wb.qbuf[which]; size_t c = wb.cap_queue;
  q.ox = b; q.oy = b + c; q.oz = b + 2 * c; q.dx = b + 3 * c; q.dy = b + 4 * c; q.dz = b + 5 * c;
  q.wx = b + 6 * c; q.wy = b + 7 * c; q.wz = b + 8 * c; q.pix = wb.qpix[which];
  return q;
}

// The scene's own workspace as the buffers of a frame.
static WorkBufs own_bufs(rt_scene* s) {
  WorkBufs wb;
  wb.fb = s->d_fb; wb.hits = s->d_hits; wb.hits_sorted = s->d_hits_sorted; wb.sort_keys = s->d_sort_keys;
  wb.sort_table = s->d_sort_table; wb.counters = s->d_counters; wb.cap_queue = s->cap_queue; wb.sort = s->sort_hits;
  for (int i = 0; i < 2; ++i) { wb.qbuf[i] = s->d_qbuf[i]; wb.qpix[i] = s->d_qpix[i]; }
  wb.fix = s->d_fix; wb.cap_fix = s->cap_fix;
  return wb;
}

static DevCamera to_dev_camera(const rt_camera* c) {
  DevCamera d;
  for (int k = 0; k < 3; ++k) { d.pos[k] = c->pos[k]; d.fwd[k] = c->forward[k]; d.right[k] = c->right[k]; d.up[k] = c->up[k]; }
  d.sdist = c->screen_distance; d.swidth = c->screen_width; d.sheight = c->screen_height;
  return d;
}

static hipEvent_t get_event(rt_scene* s) {
  if (!s->event_pool.empty()) { hipEvent_t e = s->event_pool.back(); s->event_pool.pop_back(); return e; }
  hipEvent_t e = nullptr; hipEventCreate(&e); return e;
}

struct Timed {   // RAII-ish helper: bracket one launch with events when timing is on
  rt_scene* s; PendingFrame* pf; int kind; hipEvent_t a = nullptr, b = nullptr;
  Timed(rt_scene* s_, PendingFrame* pf_, int kind_) : s(s_), pf(pf_), kind(kind_) {
    if (pf->timing) { a = get_event(s); b = get_event(s); hipEventRecord(a, s->stream); }
  }
  ~Timed() { if (pf->timing) { hipEventRecord(b, s->stream); pf->events.push_back({a, b, kind}); } }
};

template <typename FB>
static int launch_frame(rt_scene* s, const WorkBufs& wb, const DevCamera& cam, const FrameParams& fp_in, const rt_frame* fr,
                        FB* fb, PendingFrame* pf) {
  const bool lds = s->ds.n <= LDS_PRIMS;
  // fp32 frames of scenes k_shade_ls serves: level 0 is ONE kernel (camera ray -> nearest hit -> shading per lane; see
  // k_shade_ls, `fused`), no k_trace launch and no hit list for the 75 % of a frame's hits that are camera-ray hits.
  // RT_FUSE0=0 keeps the two-kernel level 0 (A/B).
  FrameParams fp = fp_in;
  const bool ls_allowed = s->shade_variant == 0 && s->n_coincident == 0 && !(fp.sequential && s->any_shin0);
  const bool ls_kernel = fr->precision == RT_FP32 && lds && ls_allowed && shade_ls_serves(s->ds, fp);
  // (fp64 frames of LDS-resident scenes: k_shade_exact_items fuses its level 0 the same way)
  const bool items_kernel = fr->precision == RT_FP64 && lds && (fp.n_shadow > 1 ? fp.n_shadow * fp.n_shadow : 1) <= ITEM_CHUNK &&
                            !(getenv("RT_EXACT_ITEMS") && atoi(getenv("RT_EXACT_ITEMS")) == 0);
  // (measured neutral there — original.txt 1080p: trace -0.035 ms, shade +0.06 ms —, so off unless RT_FUSE0_FP64=1)
  const bool fuse_fp64 = items_kernel && getenv("RT_FUSE0_FP64") && atoi(getenv("RT_FUSE0_FP64")) == 1;
  fp.fuse0 = ((ls_kernel || fuse_fp64) && !(getenv("RT_FUSE0") && atoi(getenv("RT_FUSE0")) == 0)) ? 1u : 0u;
  // ... and EVERY level (fuse0 = 2): k_shade_ls also traces the children of its hits and hands the next level their hits
  // with the rays (HitRay queues in the ray-queue buffers): a frame is max_depth + 1 shade launches + k_fix_exact, no
  // k_trace.  Needs the pixel index in 25 bits; RT_FUSE_ALL=0 keeps k_trace for the bounce levels (A/B).
  if (ls_kernel && fp.fuse0 == 1u && (size_t)fp.rs.rows * (size_t)fp.rs.W <= (size_t)HITRAY_MAX_PIXELS &&
      !(getenv("RT_FUSE_ALL") && atoi(getenv("RT_FUSE_ALL")) == 0)) fp.fuse0 = 2u;
  const int grid = s->num_cu * 8;
  RayQueue q[2] = {make_queue(wb, 0), make_queue(wb, 1)};
  const uint32_t P = (uint32_t)fp.rs.rows * (uint32_t)fp.rs.W;
  { Timed t(s, pf, 2); hipLaunchKernelGGL(k_init_counters, dim3(1), dim3(256), 0, s->stream, wb.counters, P); }
  for (int level = 0; level <= fp.max_depth; ++level) {
    RayQueue cur = q[level & 1], nxt = q[(level + 1) & 1];
    int tgrid = grid;       // trace: one workgroup per TRACE_CHUNK rays, grid-stride beyond that
    int sgrid = grid;       // shade: an upper bound, the launchers cut it to the resident workgroups (shade_grid)
    if (level == 0) {
      tgrid = (int)std::min<size_t>((size_t)grid * 4, ((size_t)P + TRACE_CHUNK - 1) / TRACE_CHUNK);
      sgrid = (int)std::min<size_t>((size_t)grid, ((size_t)P + BLOCK - 1) / BLOCK);
    }
    if (tgrid < 1) tgrid = 1;
    if (sgrid < 1) sgrid = 1;
    if (!(level == 0 && fp.fuse0) && fp.fuse0 != 2u) {
      Timed t(s, pf, 0);
      if (lds) hipLaunchKernelGGL((k_trace<FB, true>), dim3(tgrid), dim3(BLOCK), 0, s->stream, s->ds, cam, fp, level, cur, wb.hits, wb.counters, fb);
      else if (level >= 1 && s->pool_trace) hipLaunchKernelGGL((k_trace_pool<FB>), dim3(s->num_cu * 4), dim3(BLOCK), 0, s->stream, s->ds, fp, level, cur, wb.hits, wb.counters, fb);
      else hipLaunchKernelGGL((k_trace<FB, false>), dim3(tgrid), dim3(BLOCK), 0, s->stream, s->ds, cam, fp, level, cur, wb.hits, wb.counters, fb);
    }
    const HitRec* level_hits = wb.hits;
    if (wb.sort && level >= 1) {
      Timed t(s, pf, 2);
      uint32_t* bin_total = wb.sort_table + (size_t)SORT_BINS * SORT_BLOCKS;
      hipMemsetAsync(bin_total, 0, SORT_BINS * sizeof(uint32_t), s->stream);
      hipLaunchKernelGGL(k_sort_count, dim3(SORT_BLOCKS), dim3(SORT_THREADS), 0, s->stream, cam, fp, level, cur, wb.hits,
                         wb.counters, s->sort_grid, wb.sort_keys, wb.sort_table, bin_total);
      hipLaunchKernelGGL(k_sort_scan, dim3(SORT_BINS / SCAN_BINS), dim3(SCAN_THREADS), 0, s->stream, wb.sort_table, bin_total);
      hipLaunchKernelGGL(k_sort_scatter, dim3(SORT_BLOCKS), dim3(SORT_THREADS), 0, s->stream, fp, level, wb.hits, wb.counters,
                         wb.sort_keys, wb.sort_table, wb.hits_sorted);
      level_hits = wb.hits_sorted;
    }
    {
      Timed t(s, pf, 1);
      if (fr->precision == RT_FP64) {
        // LDS-resident scenes: the shadow samples of the penumbra lanes spread over the wave (k_shade_exact_items); RT_EXACT_ITEMS=0: one lane per hit (A/B)
        if (items_kernel)
          hipLaunchKernelGGL((k_shade_exact_items<FB>), dim3(shade_grid<k_shade_exact_items<FB>>(s->num_cu, sgrid)), dim3(BLOCK), 0, s->stream, s->ds, cam, fp, level, cur, nxt, level_hits, wb.counters, fb);
        else if (lds) hipLaunchKernelGGL((k_shade_exact<FB, true>), dim3(shade_grid<k_shade_exact<FB, true>>(s->num_cu, sgrid)), dim3(BLOCK), 0, s->stream, s->ds, cam, fp, level, cur, nxt, level_hits, wb.counters, fb);
        else hipLaunchKernelGGL((k_shade_exact<FB, false>), dim3(shade_grid<k_shade_exact<FB, false>>(s->num_cu, sgrid)), dim3(BLOCK), 0, s->stream, s->ds, cam, fp, level, cur, nxt, level_hits, wb.counters, fb);
      } else {
        // small scenes: register-cached per-hit primitive terms; otherwise the general kernel
        bool done = false;
        if (ls_allowed) done = launch_shade_ls(s->ds, s->all_opaque, cam, fp, level, cur, nxt, level_hits, wb.counters, (float*)fb, wb.fix, sgrid, s->num_cu, s->stream);
        if (!done)
          launch_shade_fast(s->ds, s->all_opaque, cam, fp, level, cur, nxt, level_hits, wb.counters, (float*)fb, wb.fix, sgrid, s->num_cu, s->stream);
      }
    }
  }
  if (fr->precision == RT_FP32) {
    // the (hit, light) pairs the fp32 kernels could not decide within their error bounds: exact fp64 shadow terms
    Timed t(s, pf, 1);
    const int fgrid = FIX_QUEUES * std::max(1, s->num_cu * 4 / FIX_QUEUES);
    if (fp.sequential) {
      if (lds) hipLaunchKernelGGL((k_fix_exact<true, true>), dim3(fgrid), dim3(BLOCK), 0, s->stream, s->ds, fp, wb.fix, wb.counters, (float*)fb);
      else hipLaunchKernelGGL((k_fix_exact<false, true>), dim3(fgrid), dim3(BLOCK), 0, s->stream, s->ds, fp, wb.fix, wb.counters, (float*)fb);
    } else {
      if (lds) hipLaunchKernelGGL((k_fix_exact<true, false>), dim3(fgrid), dim3(BLOCK), 0, s->stream, s->ds, fp, wb.fix, wb.counters, (float*)fb);
      else hipLaunchKernelGGL((k_fix_exact<false, false>), dim3(fgrid), dim3(BLOCK), 0, s->stream, s->ds, fp, wb.fix, wb.counters, (float*)fb);
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(RT_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
  return RT_OK;
}

// Framebuffer -> requested output format (k_convert), on the scene's stream.
static void launch_convert(rt_scene* s, PendingFrame* pf, const void* fb, bool fb_is_double, int out_format, void* dst, size_t nval) {
  Timed t(s, pf, 2);
  const int cg = (int)std::min<size_t>((nval + BLOCK - 1) / BLOCK, (size_t)s->num_cu * 16);
  if (fb_is_double) {
    if (out_format == RT_OUT_F32) hipLaunchKernelGGL((k_convert<double, float>), dim3(cg), dim3(BLOCK), 0, s->stream, (const double*)fb, (float*)dst, nval);
    else hipLaunchKernelGGL((k_convert<double, unsigned char>), dim3(cg), dim3(BLOCK), 0, s->stream, (const double*)fb, (unsigned char*)dst, nval);
  } else {
    if (out_format == RT_OUT_F64) hipLaunchKernelGGL((k_convert<float, double>), dim3(cg), dim3(BLOCK), 0, s->stream, (const float*)fb, (double*)dst, nval);
    else hipLaunchKernelGGL((k_convert<float, unsigned char>), dim3(cg), dim3(BLOCK), 0, s->stream, (const float*)fb, (unsigned char*)dst, nval);
  }
}

// ------------------------------------------------- numpy-stream jitter (rt_stream.hpp) ---------
// Legacy NumPy generator: np.random.seed(int) = init_genrand, np.random.random = genrand_res53.
struct Mt19937 {
  uint32_t mt[624]; int idx;
  explicit Mt19937(uint32_t seed) {
    mt[0] = seed;
    for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    idx = 624;
  }
  uint32_t next() {
    if (idx >= 624) {
      for (int k = 0; k < 624; ++k) {
        uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
        mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      idx = 0;
    }
    uint32_t y = mt[idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
  }
  double uniform() { uint32_t a = next() >> 5, b = next() >> 6; return (a * 67108864.0 + b) / 9007199254740992.0; }
};

struct StreamBatch { double* buf = nullptr; uint32_t* pix = nullptr; uint32_t n = 0; int depth = 0; bool root = false; };

static RayQueue batch_queue(const StreamBatch& b) {
  RayQueue q; size_t c = b.n;
  q.ox = b.buf; q.oy = b.buf + c; q.oz = b.buf + 2 * c; q.dx = b.buf + 3 * c; q.dy = b.buf + 4 * c; q.dz = b.buf + 5 * c;
  q.wx = b.buf + 6 * c; q.wy = b.buf + 7 * c; q.wz = b.buf + 8 * c; q.pix = b.pix;
  return q;
}
static hipError_t batch_alloc(StreamBatch* b, uint32_t n, int depth) {
  b->n = n; b->depth = depth; b->root = false;
  hipError_t e = hipMalloc(&b->buf, (size_t)std::max<uint32_t>(n, 1) * 9 * sizeof(double));
  if (e != hipSuccess) return e;
  return hipMalloc(&b->pix, (size_t)std::max<uint32_t>(n, 1) * sizeof(uint32_t));
}
static void batch_free(StreamBatch* b) { if (b->buf) hipFree(b->buf); if (b->pix) hipFree(b->pix); b->buf = nullptr; b->pix = nullptr; }

static int render_stream(rt_scene* s, const DevCamera& cam, const FrameParams& fp, double* fb, PendingFrame* pf) {
  hipStream_t st = s->stream;
  const uint32_t P = (uint32_t)fp.rs.rows * (uint32_t)fp.rs.W;
  const int L = s->ds.nL, n_cells = fp.n_shadow > 1 ? fp.n_shadow * fp.n_shadow : 1;
  pf->stream = true;
  HIPCHK(hipMemsetAsync(fb, 0, (size_t)P * 3 * sizeof(double), st));     // final_colors = zeros, ray_tracer.py:467
  HIPCHK(hipMemsetAsync(s->d_counters, 0, sizeof(Counters), st));
  Mt19937 rng(fp.seed);
  std::vector<StreamBatch> stack;
  { StreamBatch root; root.n = P; root.depth = 0; root.root = true; stack.push_back(root); }
  std::vector<double> h_table;
  struct Tmp { std::vector<void*> ptrs; ~Tmp() { for (void* p : ptrs) hipFree(p); } };
  int rc = RT_OK;
  auto grid = [](uint32_t n) { return dim3((n + BLOCK - 1) / BLOCK); };
#define SCHK(call)                                                                                     \
  do {                                                                                                 \
    hipError_t e_ = (call);                                                                            \
    if (e_ != hipSuccess) { rc = fail(e_ == hipErrorOutOfMemory ? RT_ERR_NOMEM : RT_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); goto done; } \
  } while (0)
  while (!stack.empty()) {
    StreamBatch b = stack.back(); stack.pop_back();
    Tmp tmp;                                                            // per-batch device temporaries
    if (!b.root) { tmp.ptrs.push_back(b.buf); tmp.ptrs.push_back(b.pix); }
    if (b.depth > fp.max_depth || b.n == 0) continue;                   // ray_tracer.py:482-483
    const uint32_t M = b.n;
    pf->s_rays += M; pf->s_batches += 1; pf->s_level_rays[b.depth] += M;
    auto dalloc = [&](size_t bytes) -> void* { void* p = nullptr; if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr; tmp.ptrs.push_back(p); return p; };
    HitRec* rec = (HitRec*)dalloc((size_t)M * sizeof(HitRec));
    uint32_t* flag = (uint32_t*)dalloc((size_t)M * 4);
    uint32_t* off = (uint32_t*)dalloc((size_t)M * 4);
    if (!rec || !flag || !off) { rc = fail(RT_ERR_NOMEM, "out of device memory (stream batch of %u rays)", M); goto done; }
    RayQueue q = b.root ? RayQueue{} : batch_queue(b);
    if (b.root) hipLaunchKernelGGL(k_sa_trace<true>, grid(M), dim3(BLOCK), 0, st, s->ds, cam, fp, q, M, rec, flag, fb);
    else hipLaunchKernelGGL(k_sa_trace<false>, grid(M), dim3(BLOCK), 0, st, s->ds, cam, fp, q, M, rec, flag, fb);
    size_t scan_bytes = 0;
    SCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, flag, off, (int)M, st));
    void* scan_tmp = dalloc(scan_bytes);
    if (!scan_tmp) { rc = fail(RT_ERR_NOMEM, "out of device memory (scan)"); goto done; }
    SCHK(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, flag, off, (int)M, st));
    uint32_t last[2] = {0, 0};
    SCHK(hipMemcpyAsync(&last[0], off + (M - 1), 4, hipMemcpyDeviceToHost, st));
    SCHK(hipMemcpyAsync(&last[1], flag + (M - 1), 4, hipMemcpyDeviceToHost, st));
    SCHK(hipStreamSynchronize(st));
    const uint32_t H = last[0] + last[1];
    if (H == 0) continue;                                               // ray_tracer.py:504-506
    pf->s_hits += H; pf->s_level_hits[b.depth] += H;
    HitRec* hits = (HitRec*)dalloc((size_t)H * sizeof(HitRec));
    if (!hits) { rc = fail(RT_ERR_NOMEM, "out of device memory (hits)"); goto done; }
    hipLaunchKernelGGL(k_sa_scatter_hits, grid(M), dim3(BLOCK), 0, st, rec, flag, off, M, hits);
    // the next lights*cells*2*H uniforms of the stream, in draw order (ray_tracer.py:219-222)
    double* d_table = nullptr;
    if (fp.n_shadow > 1 && L > 0) {
      const size_t n_tab = (size_t)L * n_cells * 2 * H;
      if (n_tab * sizeof(double) > (size_t)8 << 30) { rc = fail(RT_ERR_INVALID, "numpy-stream jitter table of %.1f GB: use pixel-keyed jitter for frames this large", n_tab * 8.0 / (1 << 30)); goto done; }
      h_table.resize(n_tab);
      for (size_t i = 0; i < n_tab; ++i) h_table[i] = rng.uniform();
      d_table = (double*)dalloc(n_tab * sizeof(double));
      if (!d_table) { rc = fail(RT_ERR_NOMEM, "out of device memory (jitter table)"); goto done; }
      SCHK(hipMemcpyAsync(d_table, h_table.data(), n_tab * sizeof(double), hipMemcpyHostToDevice, st));
    }
    StreamBatch tT, tR;                                                 // dense child temporaries, slot = hit rank
    SCHK(batch_alloc(&tT, H, b.depth + 1)); tmp.ptrs.push_back(tT.buf); tmp.ptrs.push_back(tT.pix);
    SCHK(batch_alloc(&tR, H, b.depth + 1)); tmp.ptrs.push_back(tR.buf); tmp.ptrs.push_back(tR.pix);
    uint32_t* flagT = (uint32_t*)dalloc((size_t)H * 4); uint32_t* flagR = (uint32_t*)dalloc((size_t)H * 4);
    uint32_t* offT = (uint32_t*)dalloc((size_t)H * 4); uint32_t* offR = (uint32_t*)dalloc((size_t)H * 4);
    if (!flagT || !flagR || !offT || !offR) { rc = fail(RT_ERR_NOMEM, "out of device memory (child flags)"); goto done; }
    if (b.root) hipLaunchKernelGGL(k_sa_shade<true>, grid(H), dim3(BLOCK), 0, st, s->ds, cam, fp, b.depth, q, hits, H, d_table, n_cells, batch_queue(tT), batch_queue(tR), flagT, flagR, fb, &s->d_counters->queries);
    else hipLaunchKernelGGL(k_sa_shade<false>, grid(H), dim3(BLOCK), 0, st, s->ds, cam, fp, b.depth, q, hits, H, d_table, n_cells, batch_queue(tT), batch_queue(tR), flagT, flagR, fb, &s->d_counters->queries);
    if (b.depth < fp.max_depth) {
      size_t sb = 0;
      SCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, sb, flagT, offT, (int)H, st));
      void* stmp = dalloc(sb);
      if (!stmp) { rc = fail(RT_ERR_NOMEM, "out of device memory (scan)"); goto done; }
      SCHK(hipcub::DeviceScan::ExclusiveSum(stmp, sb, flagT, offT, (int)H, st));
      SCHK(hipcub::DeviceScan::ExclusiveSum(stmp, sb, flagR, offR, (int)H, st));
      uint32_t tail[4] = {0, 0, 0, 0};
      SCHK(hipMemcpyAsync(&tail[0], offT + (H - 1), 4, hipMemcpyDeviceToHost, st));
      SCHK(hipMemcpyAsync(&tail[1], flagT + (H - 1), 4, hipMemcpyDeviceToHost, st));
      SCHK(hipMemcpyAsync(&tail[2], offR + (H - 1), 4, hipMemcpyDeviceToHost, st));
      SCHK(hipMemcpyAsync(&tail[3], flagR + (H - 1), 4, hipMemcpyDeviceToHost, st));
      SCHK(hipStreamSynchronize(st));
      const uint32_t nT = tail[0] + tail[1], nR = tail[2] + tail[3];
      if (nT) {                                                         // pushed first (ray_tracer.py:573)
        StreamBatch c; SCHK(batch_alloc(&c, nT, b.depth + 1));
        hipLaunchKernelGGL(k_sa_scatter_rays, grid(H), dim3(BLOCK), 0, st, batch_queue(tT), flagT, offT, H, batch_queue(c));
        stack.push_back(c);
      }
      if (nR) {                                                         // pushed last => processed first (:589)
        StreamBatch c; SCHK(batch_alloc(&c, nR, b.depth + 1));
        hipLaunchKernelGGL(k_sa_scatter_rays, grid(H), dim3(BLOCK), 0, st, batch_queue(tR), flagR, offR, H, batch_queue(c));
        stack.push_back(c);
      }
    }
    SCHK(hipStreamSynchronize(st));                                     // temporaries are freed at scope exit
    SCHK(hipGetLastError());
  }
done:
#undef SCHK
  hipStreamSynchronize(st);
  for (StreamBatch& b : stack) if (!b.root) batch_free(&b);
  if (rc == RT_OK) {
    unsigned long long qv = 0;
    if (hipMemcpy(&qv, &s->d_counters->queries, sizeof qv, hipMemcpyDeviceToHost) == hipSuccess) pf->s_queries = (int64_t)qv;
    hipMemsetAsync(s->d_counters, 0, sizeof(Counters), st);             // the pending-frame copy must read zeros
  }
  return rc;
}

static int render_locked(rt_scene* s, const rt_camera* camera, const rt_frame* fr, void* out, int out_is_device,
                         rt_stats* stats);

extern "C" int rt_render(rt_scene* s, const rt_camera* camera, const rt_frame* fr, void* out, int out_is_device,
                         rt_stats* stats) try {
  if (!s || !camera || !fr || !out) return fail(RT_ERR_INVALID, "rt_render: null pointer");
  std::lock_guard<std::mutex> lk(s->mu);
  int rc = RT_OK;
  for (int attempt = 0; attempt < 10; ++attempt) {
    s->retry_attempt = attempt;
    rc = render_locked(s, camera, fr, out, out_is_device, stats);
    if (rc != RT_ERR_QUEUE_OVERFLOW || (fr->flags & RT_FLAG_ASYNC)) break;
    // A level outgrew the queues (more than ~2 live rays per pixel) or the re-test queue was too small: drain()
    // has raised min_queue / min_fix, render the frame again (rendering is deterministic).
  }
  return rc;
} RT_CATCH_ALL

// Argument checks shared by rt_render and rt_render_static; *rows = rows of the frame's row set.
static int validate_frame(const rt_frame* fr, int out_is_device, int32_t* rows) {
  if (fr->width <= 0 || fr->height <= 0) return fail(RT_ERR_INVALID, "rt_render: bad image size %dx%d", fr->width, fr->height);
  if (fr->block_rows <= 0) return fail(RT_ERR_INVALID, "rt_render: block_rows must be >= 1");
  if (fr->max_depth < 0 || fr->max_depth + 1 > MAX_LEVELS) return fail(RT_ERR_INVALID, "rt_render: max_depth %d outside [0,%d]", fr->max_depth, MAX_LEVELS - 1);
  if (fr->n_shadow < 0 || fr->n_shadow > 64) return fail(RT_ERR_INVALID, "rt_render: n_shadow %d outside [0,64]", fr->n_shadow);
  if (fr->precision != RT_FP64 && fr->precision != RT_FP32) return fail(RT_ERR_INVALID, "rt_render: unknown precision %d", fr->precision);
  if (fr->out_format < RT_OUT_F64 || fr->out_format > RT_OUT_U8) return fail(RT_ERR_INVALID, "rt_render: unknown out_format %d", fr->out_format);
  if (fr->jitter_mode != RT_JITTER_PIXEL_KEYED && fr->jitter_mode != RT_JITTER_NUMPY_STREAM)
    return fail(RT_ERR_INVALID, "rt_render: unknown jitter mode %d", fr->jitter_mode);
  if (fr->jitter_mode == RT_JITTER_NUMPY_STREAM && (fr->precision != RT_FP64 || (fr->flags & RT_FLAG_ASYNC)))
    return fail(RT_ERR_INVALID, "rt_render: numpy-stream jitter needs precision fp64 and a blocking call");
  if ((fr->flags & RT_FLAG_SEQUENTIAL) && fr->jitter_mode != RT_JITTER_PIXEL_KEYED)
    return fail(RT_ERR_INVALID, "rt_render: RT_FLAG_SEQUENTIAL needs pixel-keyed jitter (the scalar stream order is per-pixel depth-first)");
  if ((fr->flags & RT_FLAG_ASYNC) && !out_is_device) return fail(RT_ERR_INVALID, "rt_render: RT_FLAG_ASYNC needs a device output buffer");
  return rt_rows_in_part(fr->height, fr->block_rows, fr->part, fr->n_parts, rows);
}

static FrameParams make_params(const rt_frame* fr, int32_t rows, size_t cap_queue, size_t cap_hits, size_t cap_fix) {
  FrameParams fp;
  fp.rs.W = fr->width; fp.rs.H = fr->height; fp.rs.block_rows = fr->block_rows; fp.rs.part = fr->part;
  fp.rs.n_parts = fr->n_parts; fp.rs.rows = rows;
  // 8x8 pixel tiles per wave at level 0; the numpy-stream mode needs hits in ascending pixel order
  fp.rs.tiled = (fr->jitter_mode == RT_JITTER_PIXEL_KEYED && !getenv("RT_NO_TILES")) ? 1 : 0; fp.rs.pad = 0;
  fp.n_shadow = fr->n_shadow; fp.max_depth = fr->max_depth; fp.seed = fr->jitter_seed; fp.queue_cap = (uint32_t)cap_queue; fp.hit_cap = (uint32_t)cap_hits;
  fp.debug = getenv("RT_DEBUG_STATS") ? (uint32_t)std::max(1, atoi(getenv("RT_DEBUG_STATS"))) : 0u;
  fp.sequential = (fr->flags & RT_FLAG_SEQUENTIAL) ? 1u : 0u;
  fp.plane_thr = fp.sequential ? std::nextafter(1e-6, 0.0) : 1e-6;   // infinite_plane.py:20 (t < 1e-6 misses) vs :42 (t > 1e-6 hits)
  fp.fix_cap = (uint32_t)std::min<size_t>(cap_fix / FIX_QUEUES, 0x03FFFFFFu);
  fp.brute = (fr->flags & RT_FLAG_BRUTE_FORCE) ? 1u : 0u;
  fp.fuse0 = 0u;
  return fp;
}

// Host-output frames: the frame is rendered as BANDS of rows, and a finished band is copied to the caller's buffer
// on a second stream while the next band renders — at 4K the 99.5 MB copy (~2 ms over PCIe even from pinned
// memory; more into pageable memory, where the copy is staged) would otherwise double the time of the call.
// Bands are row sets of the frame (block_rows = band height, n_parts = bands), so pixel-keyed jitter and the camera
// maths are those of the whole frame and the result is identical.  Stats are summed over the bands.
static int render_banded(rt_scene* s, const rt_camera* camera, const rt_frame* fr, void* out, rt_stats* stats, int32_t rows, int n_bands) {
  const int32_t band_rows = ((rows + n_bands - 1) / n_bands + 7) / 8 * 8;
  const int nb = (rows + band_rows - 1) / band_rows;
  const size_t esz = fr->out_format == RT_OUT_F64 ? 8 : (fr->out_format == RT_OUT_F32 ? 4 : 1);
  const bool fb_is_double = fr->precision == RT_FP64;
  const size_t fsz = fb_is_double ? 8 : 4;
  const bool native = (fb_is_double && fr->out_format == RT_OUT_F64) || (!fb_is_double && fr->out_format == RT_OUT_F32);
  const size_t pixels = (size_t)rows * fr->width;
  int rc = ensure_workspace(s, pixels, pixels * 3 * esz, !native, true);
  if (rc != RT_OK) return rc;
  rc = drain(s); if (rc != RT_OK) return rc;
  if (!s->copy_stream) HIPCHK(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
  while ((int)s->h_slots.size() < nb) { Counters* h = nullptr; HIPCHK(hipHostMalloc(&h, sizeof(Counters))); s->h_slots.push_back(h); }
  std::vector<rt_stats> band_stats((size_t)nb);
  std::vector<hipEvent_t> done((size_t)nb, nullptr);
  const DevCamera cam = to_dev_camera(camera);
  const WorkBufs wb = own_bufs(s);
  // ONE failure path: bands already enqueued carry pointers into band_stats (a local) and wait on / are waited for by
  // the copy stream, so a mid-loop error first lets both streams finish, forgets those stats pointers (the frames stay
  // queued; the next drain returns their pinned slots and events), and returns this call's `done` events to the pool.
  auto abort_call = [&](int code, PendingFrame* unqueued) {
    const std::string msg = g_err;                       // (keep the first error's message)
    hipStreamSynchronize(s->stream);
    hipStreamSynchronize(s->copy_stream);
    forget_stats(s, band_stats.data(), band_stats.size());
    if (unqueued) {
      for (EventPair& ep : unqueued->events) { s->event_pool.push_back(ep.a); s->event_pool.push_back(ep.b); }
      if (unqueued->h_counters) s->h_slots.push_back(unqueued->h_counters);
    }
    for (hipEvent_t e : done) if (e) s->event_pool.push_back(e);
    g_err = msg;
    return code;
  };
#define BANDCHK(call, pfp)                                                                               \
  do {                                                                                                   \
    hipError_t e_ = (call);                                                                              \
    if (e_ != hipSuccess) return abort_call(fail(e_ == hipErrorOutOfMemory ? RT_ERR_NOMEM : RT_ERR_HIP,  \
                                                 "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__), pfp); \
  } while (0)
  for (int b = 0; b < nb; ++b) {
    rt_frame bf = *fr;
    bf.block_rows = band_rows; bf.n_parts = nb; bf.part = b;
    int32_t brows = 0;
    rc = rt_rows_in_part(bf.height, bf.block_rows, bf.part, bf.n_parts, &brows);
    if (rc != RT_OK) return abort_call(rc, nullptr);
    if (nb == 1) { bf.n_parts = 1; bf.part = 0; bf.block_rows = 1; }
    const size_t off_px = (size_t)b * band_rows * fr->width, bval = (size_t)brows * fr->width * 3;
    PendingFrame pf;
    pf.user = &band_stats[(size_t)b]; pf.levels = fr->max_depth + 1; pf.n_lights = s->ds.nL;
    pf.n_cells = fr->n_shadow > 1 ? fr->n_shadow * fr->n_shadow : 1;
    pf.precision = fr->precision; pf.timing = (fr->flags & RT_FLAG_TIMING) != 0;
    pf.h_counters = s->h_slots.back(); s->h_slots.pop_back();
    const FrameParams fp = make_params(&bf, brows, s->cap_queue, s->cap_hits, s->cap_fix);
    char* fb = (char*)s->d_fb + off_px * 3 * fsz;
    rc = fb_is_double ? launch_frame<double>(s, wb, cam, fp, &bf, (double*)fb, &pf) : launch_frame<float>(s, wb, cam, fp, &bf, (float*)fb, &pf);
    if (rc != RT_OK) return abort_call(rc, &pf);
    char* src = fb;
    if (!native) { src = (char*)s->d_out + off_px * 3 * esz; launch_convert(s, &pf, fb, fb_is_double, fr->out_format, src, bval); }
    BANDCHK(hipMemcpyAsync(pf.h_counters, s->d_counters, sizeof(Counters), hipMemcpyDeviceToHost, s->stream), &pf);
    done[(size_t)b] = get_event(s);
    BANDCHK(hipEventRecord(done[(size_t)b], s->stream), &pf);
    BANDCHK(hipStreamWaitEvent(s->copy_stream, done[(size_t)b], 0), &pf);
    BANDCHK(hipMemcpyAsync((char*)out + off_px * 3 * esz, src, bval * esz, hipMemcpyDeviceToHost, s->copy_stream), &pf);
    s->pending.push_back(std::move(pf));
  }
#undef BANDCHK
  rc = drain(s);                                   // waits for the render stream, folds the bands' counters
  if (rc != RT_OK) forget_stats(s, band_stats.data(), band_stats.size());     // (a drain that failed half-way)
  hipError_t ce = hipStreamSynchronize(s->copy_stream);
  for (hipEvent_t e : done) if (e) s->event_pool.push_back(e);
  if (ce != hipSuccess) return fail(RT_ERR_HIP, "copy of the rendered bands failed: %s", hipGetErrorString(ce));
  if (rc != RT_OK) return rc;
  if (stats) {
    rt_stats t; std::memset(&t, 0, sizeof t);
    t.levels = fr->max_depth + 1;
    for (const rt_stats& b : band_stats) {
      t.rays += b.rays; t.hits += b.hits; t.shadow_rays += b.shadow_rays; t.retested += b.retested;
      t.segments = b.segments < 0 ? -1 : t.segments + b.segments;
      for (int l = 0; l < t.levels; ++l) { t.level_rays[l] += b.level_rays[l]; t.level_hits[l] += b.level_hits[l]; }
      t.trace_ms += b.trace_ms; t.shade_ms += b.shade_ms; t.other_ms += b.other_ms; t.total_ms += b.total_ms;
      t.trace_launches += b.trace_launches; t.shade_launches += b.shade_launches;
    }
    for (int l = 0; l < t.levels; ++l) if (t.level_rays[l]) t.batches++;
    *stats = t;
  }
  return RT_OK;
}

static int render_locked(rt_scene* s, const rt_camera* camera, const rt_frame* fr, void* out, int out_is_device,
                         rt_stats* stats) {
  int32_t rows = 0;
  int rc = validate_frame(fr, out_is_device, &rows);
  if (rc != RT_OK) return rc;

  HIPCHK(hipSetDevice(s->device));
  const size_t pixels = (size_t)rows * fr->width, nval = pixels * 3;
  if (pixels == 0) { if (stats) std::memset(stats, 0, sizeof *stats); return RT_OK; }
  if (!out_is_device && fr->n_parts <= 1 && fr->jitter_mode == RT_JITTER_PIXEL_KEYED && pixels >= ((size_t)1 << 20) && rows >= 64) {
    const char* e = getenv("RT_BANDS");
    const int n_bands = e ? atoi(e) : 3;
    if (n_bands > 1) return render_banded(s, camera, fr, out, stats, rows, std::min(n_bands, 16));
  }
  const size_t esz = fr->out_format == RT_OUT_F64 ? 8 : (fr->out_format == RT_OUT_F32 ? 4 : 1);
  const bool fb_is_double = fr->precision == RT_FP64;
  const bool native = (fb_is_double && fr->out_format == RT_OUT_F64) || (!fb_is_double && fr->out_format == RT_OUT_F32);
  const bool direct = native && out_is_device;            // render straight into the caller's buffer
  const bool need_out = !native;                           // conversion target on the device
  rc = ensure_workspace(s, pixels, nval * esz, need_out && !out_is_device, !(fr->flags & RT_FLAG_ASYNC), !direct);
  if (rc != RT_OK) return rc;
  if (s->pending.size() >= 48) {      // (an old async frame's overflow is not this call's error: rt_sync reports it)
    rc = drain(s);
    if (rc == RT_ERR_QUEUE_OVERFLOW) s->sticky_overflow = 1; else if (rc != RT_OK) return rc;
  }

  PendingFrame pf;
  pf.user = stats; pf.levels = fr->max_depth + 1; pf.n_lights = s->ds.nL;
  pf.n_cells = fr->n_shadow > 1 ? fr->n_shadow * fr->n_shadow : 1;
  pf.precision = fr->precision; pf.timing = (fr->flags & RT_FLAG_TIMING) != 0;
  if (s->h_slots.empty()) { Counters* h = nullptr; HIPCHK(hipHostMalloc(&h, sizeof(Counters))); s->h_slots.push_back(h); }
  pf.h_counters = s->h_slots.back(); s->h_slots.pop_back();

  const FrameParams fp = make_params(fr, rows, s->cap_queue, s->cap_hits, s->cap_fix);
  DevCamera cam = to_dev_camera(camera);

  void* fb = direct ? out : s->d_fb;
  if (fr->jitter_mode == RT_JITTER_NUMPY_STREAM) rc = render_stream(s, cam, fp, (double*)fb, &pf);
  else {
    const WorkBufs wb = own_bufs(s);
    rc = fb_is_double ? launch_frame<double>(s, wb, cam, fp, fr, (double*)fb, &pf) : launch_frame<float>(s, wb, cam, fp, fr, (float*)fb, &pf);
  }
  if (rc != RT_OK) { s->h_slots.push_back(pf.h_counters); return rc; }

  // a failure after the launches: the pinned counter slot and the frame's events go back to their pools
  auto give_back = [&](hipError_t e_, const char* what) {
    hipStreamSynchronize(s->stream);
    for (EventPair& ep : pf.events) { s->event_pool.push_back(ep.a); s->event_pool.push_back(ep.b); }
    s->h_slots.push_back(pf.h_counters);
    return fail(RT_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e_));
  };
  if (!direct) {
    void* dst = out_is_device ? out : (native ? nullptr : s->d_out);
    if (!native) launch_convert(s, &pf, fb, fb_is_double, fr->out_format, dst, nval);
    if (!out_is_device) {
      const hipError_t e_ = hipMemcpyAsync(out, native ? fb : s->d_out, nval * esz, hipMemcpyDeviceToHost, s->stream);
      if (e_ != hipSuccess) return give_back(e_, "copy of the frame to the host");
    }
  }
  {
    const hipError_t e_ = hipMemcpyAsync(pf.h_counters, s->d_counters, sizeof(Counters), hipMemcpyDeviceToHost, s->stream);
    if (e_ != hipSuccess) return give_back(e_, "copy of the frame's counters");
  }
  s->pending.push_back(std::move(pf));
  if (fr->flags & RT_FLAG_ASYNC) return RT_OK;
  return drain(s);
}

// ------------------------------------------------------ rt_render_static (allocates nothing) ---
// The buffers a frame needs beyond its input and output (counters, framebuffer for converted output, hit queue,
// two ray queues) are carved out of ONE file-scope device array instead of the scene's grown-on-demand
// workspace, so the call makes no hipMalloc / hipFree / hipHostMalloc / event creation: usable under HIP graph
// capture, in serving loops and in harnesses that forbid allocation.  Each GPU holds its own copy of the array
// at its own address (the name below is only a host-side stand-in): the address is taken once per device with
// hipGetSymbolAddress, with that device current, when a scene is created there (static_scratch_for).
static __device__ __align__(256) unsigned char scratch[(size_t)RT_STATIC_SCRATCH_MIB << 20];
constexpr size_t STATIC_BYTES = (size_t)RT_STATIC_SCRATCH_MIB << 20;
constexpr int STATIC_MAX_DEVICES = 64;
static unsigned char* g_static_ptr[STATIC_MAX_DEVICES];     // per device; nullptr = not taken yet
static std::mutex g_static_mu;                              // guards g_static_ptr
static std::mutex g_static_call_mu[STATIC_MAX_DEVICES];     // calls on one GPU share its array: serialised per process

// Address of `device`'s copy of the scratch array.  The caller has made `device` current.
static int static_scratch_for(int device, unsigned char** out) {
  if (device < 0 || device >= STATIC_MAX_DEVICES) return fail(RT_ERR_INVALID, "device %d beyond the %d the static scratch table holds", device, STATIC_MAX_DEVICES);
  std::lock_guard<std::mutex> lk(g_static_mu);
  if (!g_static_ptr[device]) {
    unsigned char* ptr = nullptr;
    hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) return fail(RT_ERR_HIP, "hipGetSymbolAddress(scratch) on device %d: %s", device, hipGetErrorString(e));
    g_static_ptr[device] = ptr;
  }
  *out = g_static_ptr[device];
  return RT_OK;
}

extern "C" int rt_render_static(rt_scene* s, const rt_camera* camera, const rt_frame* fr, void* out_device, rt_stats* stats) try {
  if (!s || !camera || !fr || !out_device) return fail(RT_ERR_INVALID, "rt_render_static: null pointer");
  std::lock_guard<std::mutex> lk(s->mu);
  int32_t rows = 0;
  int rc = validate_frame(fr, 1, &rows);
  if (rc != RT_OK) return rc;
  if (fr->jitter_mode != RT_JITTER_PIXEL_KEYED) return fail(RT_ERR_INVALID, "rt_render_static: pixel-keyed jitter only (the numpy-stream mode allocates per batch)");
  if (fr->flags & RT_FLAG_TIMING) return fail(RT_ERR_INVALID, "rt_render_static: RT_FLAG_TIMING creates events; use rt_render for timing");
  if ((fr->flags & RT_FLAG_ASYNC) && stats) return fail(RT_ERR_INVALID, "rt_render_static: RT_FLAG_ASYNC needs stats == NULL (counters are read back by blocking calls only)");
  const size_t pixels = (size_t)rows * fr->width, nval = pixels * 3;
  if (pixels > (size_t)RT_STATIC_MAX_PIXELS)
    return fail(RT_ERR_INVALID, "rt_render_static: %zu pixels in the row set, the static scratch serves at most %d (render in parts or use rt_render)",
                pixels, RT_STATIC_MAX_PIXELS);
  if (pixels == 0) { if (stats) std::memset(stats, 0, sizeof *stats); return RT_OK; }
  if (!s->static_scratch || !s->h_static) return fail(RT_ERR_INVALID, "rt_render_static: scene has no static scratch (creation failed to take its address)");
  HIPCHK(hipSetDevice(s->device));
  std::lock_guard<std::mutex> dev_lk(g_static_call_mu[s->device]);

  // carve-up: counters | framebuffer | re-test queue | hit queue | ray queue 0 | ray queue 1 | pixel indices 0 | 1
  auto up256 = [](size_t v) { return (v + 255) & ~(size_t)255; };
  unsigned char* base = s->static_scratch;
  size_t off = 0;
  WorkBufs wb;
  wb.counters = (Counters*)(base + off); off += up256(sizeof(Counters));
  wb.fb = base + off; off += up256(pixels * 3 * sizeof(double));
  wb.cap_fix = (STATIC_BYTES / 16 / sizeof(FixRec)) / FIX_QUEUES * FIX_QUEUES;   // 2 MiB of re-test records (fp32 mode)
  wb.fix = (FixRec*)(base + off); off += up256(wb.cap_fix * sizeof(FixRec));
  const size_t per_slot = sizeof(HitRec) + 2 * (9 * sizeof(double) + sizeof(uint32_t));
  const size_t cap = ((STATIC_BYTES - off) / per_slot) & ~(size_t)63;
  wb.hits = (HitRec*)(base + off); off += cap * sizeof(HitRec);
  for (int i = 0; i < 2; ++i) { wb.qbuf[i] = (double*)(base + off); off += cap * 9 * sizeof(double); }
  for (int i = 0; i < 2; ++i) { wb.qpix[i] = (uint32_t*)(base + off); off += cap * sizeof(uint32_t); }
  wb.cap_queue = cap; wb.sort = false;      // (regrouping is a scheduling step only; its buffers are not carved)
  if (off > STATIC_BYTES || cap < pixels) return fail(RT_ERR_INVALID, "rt_render_static: internal carve-up error");

  PendingFrame pf;
  pf.user = stats; pf.levels = fr->max_depth + 1; pf.n_lights = s->ds.nL;
  pf.n_cells = fr->n_shadow > 1 ? fr->n_shadow * fr->n_shadow : 1;
  pf.precision = fr->precision; pf.timing = false;
  const FrameParams fp = make_params(fr, rows, cap, cap, wb.cap_fix);
  const DevCamera cam = to_dev_camera(camera);
  const bool fb_is_double = fr->precision == RT_FP64;
  const bool native = (fb_is_double && fr->out_format == RT_OUT_F64) || (!fb_is_double && fr->out_format == RT_OUT_F32);
  void* fb = native ? out_device : wb.fb;
  rc = fb_is_double ? launch_frame<double>(s, wb, cam, fp, fr, (double*)fb, &pf) : launch_frame<float>(s, wb, cam, fp, fr, (float*)fb, &pf);
  if (rc != RT_OK) return rc;
  if (!native) launch_convert(s, &pf, fb, fb_is_double, fr->out_format, out_device, nval);
  HIPCHK(hipMemcpyAsync(s->h_static, wb.counters, sizeof(Counters), hipMemcpyDeviceToHost, s->stream));
  if (fr->flags & RT_FLAG_ASYNC) { s->static_async = true; return RT_OK; }      // overflow is reported by rt_sync
  HIPCHK(hipStreamSynchronize(s->stream));
  const Counters& c = *s->h_static;
  if (c.overflow_fix)
    return fail(RT_ERR_QUEUE_OVERFLOW, "rt_render_static: more (hit, light) pairs need the fp64 re-test than the static queue holds (%zu) "
                "(this entry point cannot grow it: use rt_render, or render fewer rows per call)", wb.cap_fix);
  if (c.overflow) {
    size_t need = 0;
    for (int l = 0; l <= MAX_LEVELS; ++l) need = std::max<size_t>(need, std::max(c.n_rays[l], c.n_hits[l]));
    need = std::max<size_t>(need, (size_t)c.seg_need * NSEG);
    return fail(RT_ERR_QUEUE_OVERFLOW, "rt_render_static: a level holds >= %zu rays, the static queues hold %zu for a row set of %zu pixels "
                "(this entry point cannot grow them: use rt_render, or render fewer rows per call)", need, cap, pixels);
  }
  if (stats) fill_stats(pf, c, stats);
  return RT_OK;
} RT_CATCH_ALL

// Pinned host memory for output frames (hipHostMalloc): device -> host copies into it run at PCIe speed and
// asynchronously; into pageable memory they are staged by the runtime at a fraction of that.
// ------------------------------------------------------------------ rt_render_multi -----------
static int render_multi_body(rt_scene* const* scenes, int n_scenes, const rt_camera* camera, const rt_frame* frame, void* out,
                             int out_is_device, rt_stats* stats, std::vector<rt_stats>& st);

extern "C" int rt_render_multi(rt_scene* const* scenes, int n_scenes, const rt_camera* camera, const rt_frame* frame, void* out,
                               int out_is_device, rt_stats* stats) try {
  if (!scenes || n_scenes < 1 || !camera || !frame || !out) return fail(RT_ERR_INVALID, "rt_render_multi: null pointer or no scenes");
  if (n_scenes > 64) return fail(RT_ERR_INVALID, "rt_render_multi: at most 64 handles");
  for (int i = 0; i < n_scenes; ++i) if (!scenes[i]) return fail(RT_ERR_INVALID, "rt_render_multi: scenes[%d] is null", i);
  if (frame->jitter_mode != RT_JITTER_PIXEL_KEYED) return fail(RT_ERR_INVALID, "rt_render_multi: pixel-keyed jitter only (the numpy stream is consumed frame-wide)");
  if (frame->block_rows <= 0 || frame->width <= 0 || frame->height <= 0) return fail(RT_ERR_INVALID, "rt_render_multi: bad frame size / block_rows");
  // The handles' async frames report into `st`, a local of this call: on EVERY failure (a handle that rejects the
  // frame, runs out of memory growing its workspace, a failed copy) the frames the other handles still hold pending
  // must forget those pointers before `st` goes away — the next drain on such a handle would otherwise write into
  // freed memory (tests/test_gpu_multi.py: a failing handle, then a render on the surviving one).
  std::vector<rt_stats> st((size_t)n_scenes);
  int rc;
  try { rc = render_multi_body(scenes, n_scenes, camera, frame, out, out_is_device, stats, st); }
  catch (...) { rc = fail(RT_ERR_HIP, "rt_render_multi: internal error"); }
  if (rc != RT_OK) {
    const std::string msg = g_err;
    for (int i = 0; i < n_scenes; ++i) {
      rt_scene* s = scenes[i];
      std::lock_guard<std::mutex> lk(s->mu);
      hipSetDevice(s->device);
      hipStreamSynchronize(s->stream);               // nothing of this call is still in flight when it returns
      forget_stats(s, st.data(), st.size());
    }
    g_err = msg;
  }
  return rc;
} RT_CATCH_ALL

static int render_multi_body(rt_scene* const* scenes, int n_scenes, const rt_camera* camera, const rt_frame* frame, void* out,
                             int out_is_device, rt_stats* stats, std::vector<rt_stats>& st) {
  const size_t esz = frame->out_format == RT_OUT_F64 ? 8 : (frame->out_format == RT_OUT_F32 ? 4 : 1);
  const size_t row_bytes = (size_t)frame->width * 3 * esz, block_bytes = row_bytes * (size_t)frame->block_rows;
  std::vector<int32_t> rows((size_t)n_scenes, 0);
  std::vector<rt_frame> fr((size_t)n_scenes, *frame);
  // 1. every handle renders its rows into its own tile buffer, all at once
  for (int i = 0; i < n_scenes; ++i) {
    rt_scene* s = scenes[i];
    fr[i].part = i; fr[i].n_parts = n_scenes; fr[i].flags = (frame->flags | RT_FLAG_ASYNC) & ~RT_FLAG_TIMING;
    int rc = rt_rows_in_part(frame->height, frame->block_rows, i, n_scenes, &rows[i]);
    if (rc != RT_OK) return rc;
    if (rows[i] == 0) continue;
    {
      std::lock_guard<std::mutex> lk(s->mu);
      HIPCHK(hipSetDevice(s->device));
      const size_t need = (size_t)rows[i] * row_bytes;
      if (need > s->cap_tile) {
        rc = drain(s); if (rc == RT_ERR_QUEUE_OVERFLOW) s->sticky_overflow = 1; else if (rc != RT_OK) return rc;
        if (s->d_tile) hipFree(s->d_tile);
        s->d_tile = nullptr; s->cap_tile = 0;
        HIPCHK(hipMalloc(&s->d_tile, need));
        s->cap_tile = need;
      }
    }
    // fault injection for tests/test_gpu_multi.py: RT_TEST_FAIL_HANDLE=i makes handle i refuse its part of the frame
    // while the handles before it already hold their async frames pending
    if (const char* e = getenv("RT_TEST_FAIL_HANDLE")) if (atoi(e) == i) return fail(RT_ERR_INVALID, "rt_render_multi: failure of handle %d injected by RT_TEST_FAIL_HANDLE", i);
    rc = rt_render(s, camera, &fr[i], s->d_tile, 1, &st[i]);
    if (rc != RT_OK) return rc;
  }
  // 2. completion; a handle whose queues overflowed renders again, blocking (rt_render grows the queues and retries)
  for (int i = 0; i < n_scenes; ++i) {
    if (rows[i] == 0) continue;
    int rc = rt_sync(scenes[i]);
    if (rc == RT_ERR_QUEUE_OVERFLOW) { fr[i].flags &= ~RT_FLAG_ASYNC; rc = rt_render(scenes[i], camera, &fr[i], scenes[i]->d_tile, 1, &st[i]); }
    if (rc != RT_OK) return rc;
  }
  // 3. every device copies its row blocks into the frame: block b of handle i is block b*n + i of the frame
  const int dst_device = scenes[0]->device;
  for (int i = 0; i < n_scenes; ++i) {
    if (rows[i] == 0) continue;
    rt_scene* s = scenes[i];
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipSetDevice(s->device));
    if (out_is_device && s->device != dst_device) {
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, s->device, dst_device) == hipSuccess && can) {
        const hipError_t e = hipDeviceEnablePeerAccess(dst_device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(RT_ERR_HIP, "peer access %d -> %d: %s", s->device, dst_device, hipGetErrorString(e));
        (void)hipGetLastError();
      }
    }
    const int32_t full = rows[i] / frame->block_rows, tail = rows[i] - full * frame->block_rows;
    char* dst0 = (char*)out + (size_t)i * block_bytes;
    if (full > 0)
      HIPCHK(hipMemcpy2DAsync(dst0, (size_t)n_scenes * block_bytes, s->d_tile, block_bytes, block_bytes, (size_t)full,
                              out_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s->stream));
    if (tail > 0)     // the frame's last, short block
      HIPCHK(hipMemcpyAsync(dst0 + (size_t)full * n_scenes * block_bytes, (char*)s->d_tile + (size_t)full * block_bytes, (size_t)tail * row_bytes,
                            out_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s->stream));
  }
  for (int i = 0; i < n_scenes; ++i) {
    if (rows[i] == 0) continue;
    HIPCHK(hipSetDevice(scenes[i]->device));
    HIPCHK(hipStreamSynchronize(scenes[i]->stream));
  }
  HIPCHK(hipSetDevice(dst_device));
  if (stats) {
    rt_stats t; std::memset(&t, 0, sizeof t);
    t.levels = frame->max_depth + 1;
    for (int i = 0; i < n_scenes; ++i) {
      if (rows[i] == 0) continue;
      const rt_stats& b = st[i];
      t.rays += b.rays; t.hits += b.hits; t.shadow_rays += b.shadow_rays; t.retested += b.retested;
      t.segments = (b.segments < 0 || t.segments < 0) ? -1 : t.segments + b.segments;
      for (int l = 0; l < t.levels && l < RT_MAX_LEVELS; ++l) { t.level_rays[l] += b.level_rays[l]; t.level_hits[l] += b.level_hits[l]; }
      t.trace_launches += b.trace_launches; t.shade_launches += b.shade_launches;
    }
    for (int l = 0; l < t.levels && l < RT_MAX_LEVELS; ++l) if (t.level_rays[l]) t.batches++;
    *stats = t;
  }
  return RT_OK;
}

extern "C" int rt_host_alloc(size_t bytes, void** out) try {
  if (!out) return fail(RT_ERR_INVALID, "rt_host_alloc: null pointer");
  *out = nullptr;
  hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
  if (e != hipSuccess) { *out = nullptr; return fail(RT_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); }
  return RT_OK;
} RT_CATCH_ALL
extern "C" void rt_host_free(void* p) { if (p) hipHostFree(p); }

// ------------------------------------------------------------------- utility exports -----------
template <typename F>
static int with_device_arrays(rt_scene* s, F&& body) {
  std::lock_guard<std::mutex> lk(s->mu);
  HIPCHK(hipSetDevice(s->device));
  int rc = drain(s);
  if (rc != RT_OK) return rc;
  return body();
}

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
};

extern "C" int rt_nearest_hit(rt_scene* s, const double* origins, const double* dirs, int64_t M, double* t_out,
                              int32_t* idx_out, double* normals_out) try {
  if (!s || M < 0 || (M > 0 && (!origins || !dirs || !t_out || !idx_out || !normals_out)))
    return fail(RT_ERR_INVALID, "rt_nearest_hit: bad arguments");
  if (M == 0) return RT_OK;
  return with_device_arrays(s, [&]() -> int {
    DevBuf o, d, t, i, n;
    HIPCHK(o.alloc(M * 24)); HIPCHK(d.alloc(M * 24)); HIPCHK(t.alloc(M * 8)); HIPCHK(i.alloc(M * 4)); HIPCHK(n.alloc(M * 24));
    HIPCHK(hipMemcpyAsync(o.p, origins, M * 24, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemcpyAsync(d.p, dirs, M * 24, hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(k_nearest_rays, dim3((unsigned)((M + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s->stream, s->ds,
                       (const double*)o.p, (const double*)d.p, (long long)M, (double*)t.p, (int*)i.p, (double*)n.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(t_out, t.p, M * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipMemcpyAsync(idx_out, i.p, M * 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipMemcpyAsync(normals_out, n.p, M * 24, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return RT_OK;
  });
} RT_CATCH_ALL

extern "C" int rt_shadow_transmission(rt_scene* s, const double* origins, const double* dirs, const double* max_dist,
                                      int64_t M, double* out) try {
  if (!s || M < 0 || (M > 0 && (!origins || !dirs || !max_dist || !out)))
    return fail(RT_ERR_INVALID, "rt_shadow_transmission: bad arguments");
  if (M == 0) return RT_OK;
  return with_device_arrays(s, [&]() -> int {
    DevBuf o, d, m, r;
    HIPCHK(o.alloc(M * 24)); HIPCHK(d.alloc(M * 24)); HIPCHK(m.alloc(M * 8)); HIPCHK(r.alloc(M * 8));
    HIPCHK(hipMemcpyAsync(o.p, origins, M * 24, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemcpyAsync(d.p, dirs, M * 24, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemcpyAsync(m.p, max_dist, M * 8, hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(k_shadow_rays, dim3((unsigned)((M + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s->stream, s->ds,
                       (const double*)o.p, (const double*)d.p, (const double*)m.p, (long long)M, (double*)r.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, r.p, M * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return RT_OK;
  });
} RT_CATCH_ALL

extern "C" int rt_camera_rays(rt_scene* s, const rt_camera* camera, int32_t W, int32_t H, int32_t y0, int32_t y1,
                              double* origins_out, double* dirs_out) try {
  if (!s || !camera || W <= 0 || H <= 0 || y0 < 0 || y1 < y0 || !origins_out || !dirs_out)
    return fail(RT_ERR_INVALID, "rt_camera_rays: bad arguments");
  const int64_t M = (int64_t)(y1 - y0) * W;
  if (M == 0) return RT_OK;
  return with_device_arrays(s, [&]() -> int {
    DevBuf o, d;
    HIPCHK(o.alloc(M * 24)); HIPCHK(d.alloc(M * 24));
    hipLaunchKernelGGL(k_camera_rays, dim3((unsigned)((M + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s->stream,
                       to_dev_camera(camera), W, H, y0, (long long)M, (double*)o.p, (double*)d.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(origins_out, o.p, M * 24, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipMemcpyAsync(dirs_out, d.p, M * 24, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return RT_OK;
  });
} RT_CATCH_ALL
