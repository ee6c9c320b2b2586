// rt_kernels.hpp — the wavefront kernels of the renderer (gfx950, wave64).
//
// Frame schedule (pixel-keyed jitter): for level = 0 .. max_depth
//     k_trace  : one lane per radiance ray -> nearest hit; misses add weight*background to the
//                framebuffer; hits are compacted into a dense hit queue (ballot + one atomic/wave)
//     k_shade_*: one lane per hit -> soft shadows + Phong for every light, accumulate into the
//                framebuffer, spawn transparency / reflection children into the next level's ray
//                queue (same wave-aggregated append)
// which is the reference's loop over its LIFO stack of ray batches (ray_tracer.py:479-595) with the
// batches of one depth merged — legal because pixel-keyed jitter makes rays independent.
// Waves pull work in 64-item grabs from a device-side cursor, so no level size ever returns to the
// host inside a frame.
#pragma once
#include "rt_device.hpp"

namespace rt {

// Audit / statistics code in the kernels exists only in the debug build (make debug ->
// librt_hip_debug.so, selected with RT_HIP_LIBRARY); in the product library RT_DBG() is a constant 0.
// Bits of RT_DEBUG_STATS (FrameParams::debug), fp32 kernels:
//       1  event counts (k_shade_ls: (wave, light) pairs; k_shade_fast large scenes: node tests / records per walk)
//       2  shortcuts off, their claims audited against the sample loop (k_shade_ls, k_shade_fast LDS scenes)
//       4  k_shade_ls: force the exact own-surface (LO) path        8  fan_cull: every record is a candidate
//   16/32  large-scene walk: skip the record-bound / sample-block phases (timing experiments, wrong images)
//      64  no wave-level leaf list   128  statistics of the wave-level lists   256  no cone-tapered second node test
//     512  no wave-level record list   1024 / 2048  wave-level steps / cycles per phase
//    4096  k_trace: per-lane walks for the camera rays too (this bit is honoured by the product library as well)
//   16384 / 32768  the lane's fan test / node test always pass (bisecting a culling problem)
//   65536  exact audit of the large-scene culling (fp64: every crossed record must reach the sample block)
#ifdef RT_DEBUG_KERNEL
#define RT_DBG(fp) ((fp).debug)
#else
#define RT_DBG(fp) 0u
#endif

constexpr int MAX_LEVELS = 32;
constexpr int BLOCK = 256;
constexpr int LDS_PRIMS = 128;   // scenes up to this many primitives are staged in LDS (8 KiB)

struct Counters {
  uint32_t n_rays[MAX_LEVELS + 1];
  uint32_t n_hits[MAX_LEVELS + 1];
  uint32_t work_trace[MAX_LEVELS + 1];
  uint32_t work_shade[MAX_LEVELS + 1];
  unsigned long long queries;   // shadow-walk nearest-hit queries (fp64 kernels)
  uint32_t overflow;            // a ray / hit queue overflowed
  uint32_t overflow_fix;        // a re-test queue overflowed (its counter keeps counting: the size it needs)
  unsigned long long dbg[4];    // RT_DEBUG_STATS=1: kernel-specific event counts
  // fp32 mode: (hit, light) pairs queued for the exact re-test (FixRec).  FIX_QUEUES sub-queues, chosen by block
  // index, each with its counter in a cache line of its own: a single device counter sustains only ~88 atomics
  // per microsecond, and a 4K pool frame queues ~1e5 pairs from as many different waves.
  uint32_t fixq[64 * 32];       // count of sub-queue q at fixq[q * 32]
  // Bounce-ray queues are cut into NSEG segments of queue_cap / NSEG slots, each with its own append counter in a
  // cache line of its own (same reason): a wave appends its children with ONE atomic to the segment its index
  // selects, no block-level aggregation or barrier.  segcnt[parity of the level the rays belong to][segment * 32].
  uint32_t segcnt[2][64 * 32];
  uint32_t seg_need;            // largest segment fill any level reached (unclamped: what an overflowed frame needs)
  uint32_t stripes_done;        // k_shade_ls: bit s = stripe s has no chunk left
  uint32_t stripes[32 * 32];    // k_shade_ls: chunk cursors of the NSTRIPE stripes of a level's hit list (stripe * 32)
  // Frames in which k_shade_ls traces its own rays at every level (FrameParams::fuse0 == 2, no k_trace launch): level L
  // reads the hit-ray queue whose segment counters are segcnt3[L % 3], appends the hits of its children to the queue of
  // segcnt3[(L + 1) % 3] and clears segcnt3[(L + 2) % 3] (last read by level L - 1) for level L + 1 to append to;
  // likewise the chunk cursors: level L works through stripes3[L % 3] and clears stripes3[(L + 1) % 3].
  uint32_t segcnt3[3][64 * 32];
  uint32_t stripes3[3][32 * 32];
};

// One hit of a bounce ray together with the ray it belongs to, 64 bytes = one cache line per lane: what k_shade_ls's
// inline trace appends for the next level in fused frames (instead of a 76-byte SoA ray + a 16-byte HitRec that k_trace
// reads and writes in between).  The hit point is o + t*d as every kernel forms it; the origin is not needed again.
struct __attribute__((aligned(64))) HitRay {
  double hp[3];    // hit point
  double d[3];     // ray direction
  float w[3];      // weight of the ray's radiance in its pixel (fp32 mode)
  uint32_t pk;     // framebuffer index (25 bits) | record index << 25 (5 bits; scenes with boxes: 4 bits) | box face << 29
};
static_assert(sizeof(HitRay) == 64, "HitRay must be one 64-byte line");
constexpr uint32_t HITRAY_PIX_BITS = 25, HITRAY_MAX_PIXELS = 1u << HITRAY_PIX_BITS;
constexpr int FIX_QUEUES = 64;
constexpr int NSEG = 64;
static_assert(NSEG == 64, "k_trace scans the segment counters with one wave, one lane per segment");
constexpr int NSTRIPE = 32;

// Radiance-ray queue of one level, struct-of-arrays so that a wave's loads/stores coalesce.
struct RayQueue {
  double *ox, *oy, *oz, *dx, *dy, *dz, *wx, *wy, *wz;
  uint32_t* pix;   // local pixel index (row-set framebuffer index)
};

struct __attribute__((aligned(16))) HitRec {
  uint32_t slot;   // ray slot in the level's queue (level 0: local pixel index)
  int32_t prim;    // sorted primitive record index
  double t;
};

// fp32 mode: one (hit, light) pair whose shadow term the fp32 kernel could not decide within its error bounds
// (a sample grazes a silhouette, touches the light, starts on a second surface, ...).  The fp32 kernel
// contributes T = 0 for the pair and queues this record; k_fix_exact evaluates T with the literal fp64 walk
// of the fp64 mode and adds m * T to the pixel, so every hit / miss decision of the fp32 image is the fp64 one.
struct __attribute__((aligned(8))) FixRec {
  double px, py, pz;     // shadow origin p = hit point + n * 1e-6 (ray_tracer.py:538)
  float m[3];            // colour per unit of T: weight * (Kd Lc n.l + Ks Lc spec pow) * (1 - transparency) * shadow_intensity
  uint32_t lp;           // framebuffer index of the pixel (row-set local)
  uint32_t gpix;         // global pixel index (jitter key)
  uint16_t level, light;
};
static_assert(sizeof(FixRec) == 48, "FixRec must be 48 bytes");

struct FrameParams {
  RowSet rs;
  int32_t n_shadow, max_depth;
  uint32_t seed;
  uint32_t queue_cap;    // slots of each bounce-ray queue
  uint32_t hit_cap;      // slots of the hit list (>= pixels of the row set: level 0 has one ray per pixel, and >= queue_cap)
  uint32_t debug;        // RT_DEBUG_STATS=1: kernels count events into Counters::queries (fp32 mode)
  uint32_t sequential;   // RT_FLAG_SEQUENTIAL: semantics of the reference's per-pixel render()
  double plane_thr;      // plane hit threshold: HIT_EPS, or its predecessor in sequential mode
  uint32_t fix_cap;      // capacity of ONE re-test sub-queue (FixRec); sub-queue q starts at fix[q * fix_cap]
  uint32_t brute;        // RT_FLAG_BRUTE_FORCE: fp32 kernels test every surface for every shadow sample (no culling, no shortcuts)
  uint32_t fuse0;        // 1: level 0 of the shade kernel traces its own camera rays (no k_trace launch, no hit list at level 0);
                         // 2 (k_shade_ls): every level does, and traces its children too (HitRay queues, no k_trace at all); set by launch_frame
};

__device__ __forceinline__ uint32_t global_pixel(const RowSet& rs, uint32_t lp) {
  uint32_t lr = lp / (uint32_t)rs.W, x = lp - lr * (uint32_t)rs.W;
  return (uint32_t)rowset_global_y(rs, (int)lr) * (uint32_t)rs.W + x;
}

// Stage the primitive records in LDS when the scene is small; otherwise read them from global
// memory (wave-uniform addresses -> scalar loads).  Returns the pointer the kernel should use.
template <bool LDS_SCENE>
__device__ __forceinline__ const Prim* stage_scene(const DevScene& sc, Prim* lds) {
  if (!LDS_SCENE) return sc.prims;
  const int words = sc.n * (int)(sizeof(Prim) / 16);
  const uint4* src = reinterpret_cast<const uint4*>(sc.prims);
  uint4* dst = reinterpret_cast<uint4*>(lds);
  for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
  __syncthreads();
  return lds;
}

// Small hierarchies (<= LDS_NODES nodes, 32 KiB) are staged in LDS too: the walk is a chain of
// dependent loads, LDS latency is several times shorter than L2's.
constexpr int LDS_NODES = 980;      // (k_trace's nodes + hit staging + wave lists then fit four workgroups per CU)
template <bool LDS_SCENE>
__device__ __forceinline__ Accel stage_accel(const DevScene& sc, BvhNode* lds) {
  Accel a = sc.ac;
  a.lds_nodes = -1; a.lds_hpair = a.lds_haxis = a.lds_hleaf = -1; a.n_lds_inner = 0;
  if (LDS_SCENE) { a.n_nodes = 0; a.hpair = nullptr; return a; }
  if (a.n_nodes <= 0 || a.n_nodes > LDS_NODES) return a;
  const int words = a.n_nodes * (int)(sizeof(BvhNode) / 16);
  const uint4* src = reinterpret_cast<const uint4*>(a.nodes);
  uint4* dst = reinterpret_cast<uint4*>(lds);
  for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
  __syncthreads();
  a.nodes = lds;
  a.lds_nodes = (int32_t)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) BvhNode*)lds;
  return a;
}

// Wave-level dynamic work grab: returns the base index of a 64-item chunk (all lanes get it).
__device__ __forceinline__ uint32_t grab64(uint32_t* cursor) {
  uint32_t base = 0;
  if (__lane_id() == 0) base = atomicAdd(cursor, 64u);
  return __shfl(base, 0);
}

// 64-hit chunks of a level's hit list, handed out through NSTRIPE striped cursors (Counters::stripes; the level's trace
// kernel zeroes them): a single device counter sustains only ~88 atomics per microsecond and every wave waits ~1-2 us
// for its answer — a 1080p level has 27k chunks.  A wave works through its home stripe, then moves to the stripe with the
// most chunks left (one load and one atomic per move).  fetch() issues the atomic for the NEXT chunk — call it right
// after the current chunk's global loads —, take() consumes the answer at the chunk's end.  (k_shade_ls has the same
// scheme inline.)  All lanes of the wave must call both.
struct ChunkCursor {
  static constexpr uint32_t NONE = 0xFFFFFFFFu;
  uint32_t* stripes;
  uint32_t n_chunks, cps, stripe;
  __device__ __forceinline__ void init(uint32_t* s, uint32_t count, uint32_t wave_id) {
    stripes = s; n_chunks = (count + 63u) / 64u; cps = (n_chunks + NSTRIPE - 1) / NSTRIPE; stripe = wave_id % NSTRIPE;
  }
  __device__ __forceinline__ uint32_t fetch() const {
    uint32_t j = 0;
    if (__lane_id() == 0) j = atomicAdd(&stripes[stripe * 32], 1u);
    return j;
  }
  __device__ __forceinline__ uint32_t take(uint32_t fetched) {
    const uint32_t lane = __lane_id();
    uint32_t i_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)fetched);
    for (;;) {
      if (i_ < cps && stripe * cps + i_ < n_chunks) return stripe * cps + i_;
      int left = 0;
      if (lane < (uint32_t)NSTRIPE) {       // lane s looks at stripe s (device-scope load: not from this CU's vector cache)
        const uint32_t cur = __hip_atomic_load(&stripes[lane * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t first = lane * cps;
        const uint32_t n_s = first < n_chunks ? min(cps, n_chunks - first) : 0u;
        left = cur < n_s ? (int)(n_s - cur) : 0;
      }
      int key = (left << 6) | (int)(63u - lane);     // most chunks left, lowest stripe on ties
#pragma unroll
      for (int o_ = 32; o_ > 0; o_ >>= 1) key = max(key, __shfl_xor(key, o_));
      if ((key >> 6) <= 0) return NONE;
      stripe = 63u - (uint32_t)(key & 63);
      uint32_t j_ = 0;
      if (lane == 0) j_ = atomicAdd(&stripes[stripe * 32], 1u);
      i_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)j_);
    }
  }
};

template <typename FB>
__device__ __forceinline__ void fb_add(FB* fb, uint32_t lp, double r, double g, double b) {
  atomicAdd(&fb[3 * (size_t)lp + 0], (FB)r);
  atomicAdd(&fb[3 * (size_t)lp + 1], (FB)g);
  atomicAdd(&fb[3 * (size_t)lp + 2], (FB)b);
}

// Level 0: the hit's pixel has no other contribution yet (0 + w*c == w*c exactly) -> plain store.
template <typename FB>
__device__ __forceinline__ void fb_put(FB* fb, bool first, uint32_t lp, double r, double g, double b) {
  if (first) {
    fb[3 * (size_t)lp + 0] = (FB)r; fb[3 * (size_t)lp + 1] = (FB)g; fb[3 * (size_t)lp + 2] = (FB)b;
  } else {
    fb_add(fb, lp, r, g, b);
  }
}

// Level-0 slot -> local pixel index.  With rs.tiled the slots walk 8x8 pixel tiles (a wave = one tile,
// a 1024-ray trace chunk = a 128x8 strip) instead of 64x1 row segments: the hits of a wave are then
// spatially compact, so fewer waves straddle a shadow boundary and the per-wave candidate sets of the
// shade kernels are smaller.  Pixels outside the largest (multiple of 8) x (multiple of 8) region follow
// in row-major order: first the right strip, then the bottom rows.  A bijection on [0, rows*W).
__device__ __forceinline__ uint32_t level0_pixel(const RowSet& rs, uint32_t slot) {
  if (!rs.tiled) return slot;
  const uint32_t W = (uint32_t)rs.W, R = (uint32_t)rs.rows, W8 = W & ~7u, R8 = R & ~7u;
  const uint32_t n_tiled = W8 * R8;
  if (slot < n_tiled) {
    const uint32_t tile = slot >> 6, in = slot & 63u, tpr = W8 >> 3;
    const uint32_t ty = tile / tpr, tx = tile - ty * tpr;
    return (ty * 8u + (in >> 3)) * W + tx * 8u + (in & 7u);
  }
  uint32_t s = slot - n_tiled;
  const uint32_t wr = W - W8;                 // right strip: R8 rows x wr columns
  if (s < R8 * wr) { const uint32_t y = s / wr; return y * W + W8 + (s - y * wr); }
  s -= R8 * wr;                               // bottom rows: (R - R8) x W
  return R8 * W + s;
}

struct RayIn { d3 o, d, w; uint32_t lp; };

__device__ __forceinline__ RayIn load_ray(int level, uint32_t slot, const DevCamera& cam, const RowSet& rs, const RayQueue& q) {
  RayIn r;
  if (level == 0) {
    slot = level0_pixel(rs, slot);
    uint32_t lr = slot / (uint32_t)rs.W, x = slot - lr * (uint32_t)rs.W;
    r.o = mk3(cam.pos[0], cam.pos[1], cam.pos[2]);
    r.d = camera_dir(cam, rs.W, rs.H, (int)x, rowset_global_y(rs, (int)lr));
    r.w = mk3(1.0, 1.0, 1.0);
    r.lp = slot;          // (the mapped pixel)
  } else {
    r.o = mk3(q.ox[slot], q.oy[slot], q.oz[slot]);
    r.d = mk3(q.dx[slot], q.dy[slot], q.dz[slot]);
    r.w = mk3(q.wx[slot], q.wy[slot], q.wz[slot]);
    r.lp = q.pix[slot];
  }
  return r;
}

// ------------------------------------------------------------------------------- k_trace -------
// Primary-ray generation (level 0) / queue read, brute-force nearest hit, miss -> background,
// hit compaction.  ray_tracer.py:491-515 and camera.py:46-73.
//
// A workgroup owns chunks of TRACE_CHUNK consecutive rays.  Hits are compacted into an LDS staging
// buffer with wave-aggregated LDS atomics (ballot + mbcnt prefix), then the whole chunk is appended
// to the global hit queue with ONE global atomic and coalesced 16-byte stores.  (A single device
// counter sustains only ~88 atomics/us on MI355X, so per-wave appends would serialise a 4K frame's
// 130k waves into >1 ms.)
constexpr int TRACE_CHUNK = 1024;

template <typename FB, bool LDS_SCENE>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(LDS_SCENE ? 1 : 4))) void k_trace(DevScene sc, DevCamera cam, FrameParams fp, int level, RayQueue q,
                                                 HitRec* __restrict__ hits, Counters* C, FB* __restrict__ fb) {
  __shared__ Prim s_prims[LDS_SCENE ? LDS_PRIMS : 1];
  __shared__ BvhNode s_nodes[LDS_SCENE ? 1 : LDS_NODES];
  // large scenes: half-size chunks, so that nodes + staging leave room for 4 workgroups per CU
  constexpr int CHUNK = LDS_SCENE ? TRACE_CHUNK : TRACE_CHUNK / 2;
  __shared__ HitRec s_hits[CHUNK];
  __shared__ uint32_t s_count, s_base;
  __shared__ int s_wleaves[LDS_SCENE ? 1 : (BLOCK / 64) * WAVE_LEAF_CAP];      // per-wave leaf lists (nearest_accel<true>)
  const Prim* P = stage_scene<LDS_SCENE>(sc, s_prims);
  const Accel ac = stage_accel<LDS_SCENE>(sc, s_nodes);
  // wave-level scan for the camera rays only (8x8 pixel tiles: one origin, a thin cone; measured 1.13 -> 0.74 ms on
  // the synthetic 4K frame).  The deeper levels' waves mix transmitted and reflected rays of many surfaces: their
  // cones are too wide, and the attempt costs them ~3 %.  RT_DEBUG_STATS=4096: per-lane walks everywhere (A/B).
  const bool wave_mode = !LDS_SCENE && ac.n_nodes > 0 && ac.n_cut > 0;
  const bool try_wave = level == 0 && !(fp.debug & 4096u);
  // Level 0: slots are the pixels of the row set.  Deeper levels: the ray queue is cut into NSEG segments (see
  // Counters::segcnt); a chunk is CHUNK consecutive slots of ONE segment, chunk ids run through the segments in order.
  __shared__ uint32_t s_segfirst[NSEG + 1];     // first chunk id of each segment
  const uint32_t seg_cap = fp.queue_cap / NSEG;
  uint32_t n_chunks;
  if (level == 0) {
    uint32_t count0 = C->n_rays[0];
    if (count0 > fp.hit_cap) count0 = fp.hit_cap;
    n_chunks = (count0 + CHUNK - 1) / CHUNK;
    if (threadIdx.x == 0) { s_segfirst[0] = 0; s_segfirst[1] = n_chunks; }
  } else {
    if (threadIdx.x < 64) {                        // lane q: segment q; inclusive scans over the wave
      const uint32_t raw = C->segcnt[level & 1][threadIdx.x * 32];
      const uint32_t nq_ = min(raw, seg_cap);
      uint32_t ch = (nq_ + CHUNK - 1) / CHUNK, tot = nq_, worst = raw;
#pragma unroll
      for (int o_ = 1; o_ < 64; o_ <<= 1) {
        const uint32_t a = __shfl_up(ch, o_), b = __shfl_up(tot, o_);
        if ((int)threadIdx.x >= o_) { ch += a; tot += b; }
        worst = max(worst, (uint32_t)__shfl_xor((int)worst, o_));
      }
      s_segfirst[threadIdx.x + 1] = ch;
      if (threadIdx.x == 0) s_segfirst[0] = 0;
      if (blockIdx.x == 0 && threadIdx.x == 63) { C->n_rays[level] = tot; atomicMax(&C->seg_need, worst); }
    }
    // the segment counters the NEXT level's rays will be appended to (by this level's shade kernel)
    if (blockIdx.x == 0) for (int i = threadIdx.x; i < NSEG; i += BLOCK) C->segcnt[(level + 1) & 1][i * 32] = 0u;
  }
  // chunk cursors of this level's shade kernel
  if (blockIdx.x == 0) { if (threadIdx.x < NSTRIPE) C->stripes[threadIdx.x * 32] = 0u; if (threadIdx.x == 0) C->stripes_done = 0u; }
  __syncthreads();
  n_chunks = s_segfirst[level == 0 ? 1 : NSEG];
  for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    // chunk -> (first slot, number of rays in it)
    uint32_t slot0, count;       // slots [slot0, count) of this chunk are rays (count is an END, like the old level size)
    if (level == 0) {
      slot0 = chunk * CHUNK;
      count = C->n_rays[0]; if (count > fp.hit_cap) count = fp.hit_cap;
    } else {
      int lo_ = 0, hi_ = NSEG;                     // last segment whose first chunk id is <= chunk
      while (hi_ - lo_ > 1) { const int mid = (lo_ + hi_) >> 1; if (s_segfirst[mid] <= chunk) lo_ = mid; else hi_ = mid; }
      uint32_t nq_ = C->segcnt[level & 1][lo_ * 32]; if (nq_ > seg_cap) nq_ = seg_cap;
      slot0 = (uint32_t)lo_ * seg_cap + (chunk - s_segfirst[lo_]) * CHUNK;
      count = (uint32_t)lo_ * seg_cap + nq_;
    }
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < CHUNK / BLOCK; ++it) {
      const uint32_t slot = slot0 + it * BLOCK + threadIdx.x;
      const bool active = slot < count;
      bool hit = false;
      int k = -1; double t = RT_INF;
      if (active || wave_mode) {
        RayIn r;
        if (wave_mode) {
          // all 64 lanes take part in the wave-level scan: the idle lanes of the last wave copy its last ray
          r = load_ray(level, active ? slot : count - 1, cam, fp.rs, q);
          t = nearest_accel<true>(P, sc.nS, sc.nP, ac, r.o, r.d, &k, fp.plane_thr, s_wleaves + (threadIdx.x >> 6) * (LDS_SCENE ? 0 : WAVE_LEAF_CAP), try_wave);
        } else {
          r = load_ray(level, slot, cam, fp.rs, q);
          // (camera rays of a small scene: the wave's rays are coherent, most spheres are missed by all of them)
          if (LDS_SCENE && level == 0) t = nearest_coherent(P, sc.nS, sc.nP, sc.nB, r.o, r.d, &k, fp.plane_thr);
          else t = nearest(P, sc.nS, sc.nP, sc.nB, r.o, r.d, &k, fp.plane_thr, &ac);
        }
        hit = active && k >= 0;
        if (active) {
        if (level == 0) {
          // every pixel has exactly one level-0 ray: a plain store initialises the framebuffer
          // (0 + 1*bg == bg exactly); hit pixels are stored by the level-0 shade kernel
          if (!hit) {
            fb[3 * (size_t)r.lp + 0] = (FB)sc.bg[0];
            fb[3 * (size_t)r.lp + 1] = (FB)sc.bg[1];
            fb[3 * (size_t)r.lp + 2] = (FB)sc.bg[2];
          }
        } else if (!hit) {
          fb_add(fb, r.lp, r.w.x * sc.bg[0], r.w.y * sc.bg[1], r.w.z * sc.bg[2]);   // :500-502
        }
        }
      }
      const uint32_t ls = wave_append(&s_count, hit);   // LDS atomic, one per wave
      if (hit) { HitRec h; h.slot = slot; h.prim = k; h.t = t; s_hits[ls] = h; }
    }
    __syncthreads();
    const uint32_t n_local = s_count;
    if (threadIdx.x == 0) s_base = n_local ? atomicAdd(&C->n_hits[level], n_local) : 0u;
    __syncthreads();
    const uint32_t gbase = s_base;
    for (uint32_t i = threadIdx.x; i < n_local; i += BLOCK) {
      if (gbase + i < fp.hit_cap) hits[gbase + i] = s_hits[i];
      else C->overflow = 1u;
    }
    __syncthreads();
  }
}

// k_trace for the bounce levels of large scenes.  Bounce rays of one wave take very different paths through the
// hierarchy (measured on the synthetic 4608-object scene: 44 node steps per ray, but 118 per wave, which waits for
// its slowest lane), so here a wave keeps a POOL of rays instead of one ray per lane: a lane whose walk has ended
// writes its result and, once enough lanes are idle, takes the next ray of the pool while the others carry on.
// Pools are chunks of POOL_CHUNK consecutive queue slots, grabbed with one atomic on Counters::work_trace[level].
// Hits are gathered per wave in LDS and appended to the hit list 64+ at a time (one global atomic each), so nothing
// in the main loop needs a workgroup barrier.  Results are those of k_trace (same exact tests, same (t, orig) rule);
// only the order of the hit list differs, and no consumer depends on it.
constexpr int POOL_CHUNK = 512;
#ifndef RT_POOL_REFILL
#define RT_POOL_REFILL 24
#endif
#ifndef RT_POOL_MIN_STEP
#define RT_POOL_MIN_STEP 12
#endif
constexpr int POOL_REFILL = RT_POOL_REFILL;     // idle lanes that trigger a refill
constexpr int POOL_MIN_STEP = RT_POOL_MIN_STEP; // node steps stop when fewer lanes than this are still stepping (the rest wait at their leaves)
template <typename FB>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(4))) void k_trace_pool(DevScene sc, FrameParams fp, int level, RayQueue q,
                                                 HitRec* __restrict__ hits, Counters* C, FB* __restrict__ fb) {
  constexpr int TREE_BYTES = LDS_NODES * (int)sizeof(BvhNode);
  __shared__ __attribute__((aligned(16))) unsigned char s_tree[TREE_BYTES];     // child-box pairs | leaf table | axis bytes
  __shared__ HitRec s_whits[BLOCK / 64][128];
  __shared__ uint32_t s_segfirst[NSEG + 1];
  const Prim* P = sc.prims;
  Accel ac = sc.ac;
  ac.lds_nodes = -1; ac.lds_hleaf = -1;
  {
    // everything when it fits (53 bytes per group), else the top of the tree (heap order is level order) and the
    // leaf table stays in global memory
    const int n_inner = ac.n_groups - 1;
    const bool all = 48 * max(n_inner, 1) + 4 * ac.n_groups + n_inner <= TREE_BYTES;
    const int n_lp = all ? n_inner : min(n_inner, TREE_BYTES / 49);
    float4* dp = reinterpret_cast<float4*>(s_tree);
    for (int i = threadIdx.x; i < 3 * n_lp; i += BLOCK) dp[i] = ac.hpair[i];
    int off = 48 * max(n_lp, 1);
    if (all) {
      uint32_t* dl_ = reinterpret_cast<uint32_t*>(s_tree + off);
      for (int i = threadIdx.x; i < ac.n_groups; i += BLOCK) dl_[i] = ac.hleaf[i];
      ac.lds_hleaf = (int32_t)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(s_tree + off);
      off += 4 * ac.n_groups;
    }
    for (int i = threadIdx.x; i < n_lp; i += BLOCK) s_tree[off + i] = ac.haxis[i];
    ac.lds_hpair = (int32_t)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)s_tree;
    ac.lds_haxis = (int32_t)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(s_tree + off);
    ac.n_lds_inner = n_lp;
  }
  const uint32_t seg_cap = fp.queue_cap / NSEG;
  if (threadIdx.x < 64) {                          // lane q: segment q; inclusive scans over the wave (as in k_trace)
    const uint32_t raw = C->segcnt[level & 1][threadIdx.x * 32];
    const uint32_t nq_ = min(raw, seg_cap);
    uint32_t ch = (nq_ + POOL_CHUNK - 1) / POOL_CHUNK, tot = nq_, worst = raw;
#pragma unroll
    for (int o_ = 1; o_ < 64; o_ <<= 1) {
      const uint32_t a = __shfl_up(ch, o_), b = __shfl_up(tot, o_);
      if ((int)threadIdx.x >= o_) { ch += a; tot += b; }
      worst = max(worst, (uint32_t)__shfl_xor((int)worst, o_));
    }
    s_segfirst[threadIdx.x + 1] = ch;
    if (threadIdx.x == 0) s_segfirst[0] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 63) { C->n_rays[level] = tot; atomicMax(&C->seg_need, worst); }
  }
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < NSEG; i += BLOCK) C->segcnt[(level + 1) & 1][i * 32] = 0u;
    if (threadIdx.x < NSTRIPE) C->stripes[threadIdx.x * 32] = 0u;
    if (threadIdx.x == 0) C->stripes_done = 0u;
  }
  __syncthreads();
  const uint32_t n_chunks = s_segfirst[NSEG];
  const int lane = (int)__lane_id();
  HitRec* whits = s_whits[threadIdx.x >> 6];
  uint32_t n_wh = 0;                               // hits waiting in the wave's buffer (wave-uniform)
  uint32_t pool_next = 0, pool_end = 0;            // slots of the current pool not handed out yet (wave-uniform)
  bool exhausted = false;                          // no more chunks (wave-uniform)
  bool busy = false;                               // this lane holds a ray
  PoolRay pr; pr.done = true; pr.fresh = false; pr.idx = 0; pr.trail = 0u; pr.bk = -1; pr.best = RT_INF;
  uint32_t slot = 0;
#ifdef RT_DEBUG_KERNEL
  long long ck[7] = {0, 0, 0, 0, 0, 0, 0}; const long long ck_begin = clock64(); long long ck_t = ck_begin;
#define RT_POOL_TICK(i) do { const long long n_ = clock64(); ck[i] += n_ - ck_t; ck_t = n_; } while (0)
#else
#define RT_POOL_TICK(i) do { } while (0)
#endif
  for (;;) {
    // ---- refill: idle lanes take the next rays of the pool
    const unsigned long long idle = __ballot(!busy);
    const int n_idle = __popcll(idle);
    if (n_idle == 64 || (n_idle >= POOL_REFILL && !exhausted)) {
      int want = n_idle;                           // idle lanes still without a ray (the lowest ones are served first)
      while (want > 0 && !exhausted) {
        if (pool_next >= pool_end) {               // next chunk
          uint32_t chunk = 0;
          if (lane == 0) chunk = atomicAdd(&C->work_trace[level], 1u);
          chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)chunk);
          if (chunk >= n_chunks) { exhausted = true; break; }
          int lo_ = 0, hi_ = NSEG;                 // last segment whose first chunk id is <= chunk
          while (hi_ - lo_ > 1) { const int mid = (lo_ + hi_) >> 1; if (s_segfirst[mid] <= chunk) lo_ = mid; else hi_ = mid; }
          uint32_t nq_ = C->segcnt[level & 1][lo_ * 32]; if (nq_ > seg_cap) nq_ = seg_cap;
          nq_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)nq_);
          pool_next = (uint32_t)lo_ * seg_cap + (chunk - s_segfirst[lo_]) * POOL_CHUNK;
          pool_end = min(pool_next + (uint32_t)POOL_CHUNK, (uint32_t)lo_ * seg_cap + nq_);
        }
        const unsigned long long need = __ballot(!busy);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
        const uint32_t avail = pool_end - pool_next;
        if (!busy && rank < avail) {
          slot = pool_next + rank;
          const d3 o = mk3(q.ox[slot], q.oy[slot], q.oz[slot]);
          const d3 d = mk3(q.dx[slot], q.dy[slot], q.dz[slot]);
          pr.begin(P, sc.nS, sc.nP, ac, o, d, fp.plane_thr);
          busy = true;
        }
        const uint32_t took = min((uint32_t)want, avail);
        pool_next += took;
        want -= (int)took;
      }
    }
    if (__ballot(busy) == 0ull) break;             // every ray of every chunk this wave took is done
    RT_POOL_TICK(0);
    // ---- node steps until every busy lane holds a leaf or has finished, then the leaves together
    int nj = 0;
    int rec = -1;
    if (busy) {
      const int leaf = pr.to_next_leaf(ac, POOL_MIN_STEP);
      if (leaf >= 0) rec = PoolRay::leaf_records(ac, leaf, &nj);
    }
    RT_POOL_TICK(1);
    uint32_t cand = 0;
    if (rec >= 0) {
      RT_WALK_COUNT(2);
#pragma unroll
      for (int h = 0; h < 8; h += 4) {
        if (h > 0 && nj <= h) break;
        float4 b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = ac.pbound[rec + (h + j < nj ? h + j : 0)];
#pragma unroll
        for (int j = 0; j < 4; ++j) cand |= (h + j < nj && pr.may(b[j])) ? (1u << (h + j)) : 0u;
      }
    }
    RT_POOL_TICK(2);
    while (cand) {
      RT_WALK_COUNT(4);
      const int j = __ffs((int)cand) - 1;
      cand &= cand - 1;
      pr.exact(P, sc.nS, rec + j);
    }
    RT_POOL_TICK(3);
    // ---- finished walks: miss -> background, hit -> the wave's buffer
    const bool fin = busy && rec < 0 && pr.done;
    const bool hit = fin && pr.bk >= 0;
    if (fin && !hit) {
      const uint32_t lp = q.pix[slot];
      fb_add(fb, lp, q.wx[slot] * sc.bg[0], q.wy[slot] * sc.bg[1], q.wz[slot] * sc.bg[2]);   // :500-502
    }
    if (fin) busy = false;
    const unsigned long long mh = __ballot(hit);
    if (mh) {
      const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(mh >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mh, 0u));
      if (hit) { HitRec h; h.slot = slot; h.prim = pr.bk; h.t = pr.best; whits[n_wh + r] = h; }
      n_wh += (uint32_t)__popcll(mh);
      if (n_wh >= 64u) {
        uint32_t gbase = 0;
        if (lane == 0) gbase = atomicAdd(&C->n_hits[level], n_wh);
        gbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)gbase);
        for (uint32_t i = (uint32_t)lane; i < n_wh; i += 64u) {
          if (gbase + i < fp.hit_cap) hits[gbase + i] = whits[i];
          else C->overflow = 1u;
        }
        n_wh = 0;
      }
    }
    RT_POOL_TICK(4);
#ifdef RT_DEBUG_KERNEL
    ++ck[6];
#endif
  }
#ifdef RT_DEBUG_KERNEL
  if (lane == 0) {
    for (int i = 0; i < 5; ++i) atomicAdd(&g_pool_cyc[i], (unsigned long long)ck[i]);
    atomicAdd(&g_pool_cyc[5], (unsigned long long)(clock64() - ck_begin)); atomicAdd(&g_pool_cyc[6], (unsigned long long)ck[6]);
  }
#endif
  if (n_wh > 0u) {
    uint32_t gbase = 0;
    if (lane == 0) gbase = atomicAdd(&C->n_hits[level], n_wh);
    gbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)gbase);
    for (uint32_t i = (uint32_t)lane; i < n_wh; i += 64u) {
      if (gbase + i < fp.hit_cap) hits[gbase + i] = whits[i];
      else C->overflow = 1u;
    }
  }
}

// Spawn the children of one shaded hit (ray_tracer.py:565-595) into the next level's queue.
// Must be called convergently by every lane of the wave.
// Sequential semantics (ray_tracer.py:370-377): when the transmitted ray of a hit is not traced (depth cap, or
// transparency <= EPS) the surface still adds background * transparency.
__device__ __forceinline__ bool seq_background_term(const FrameParams& fp, int level, const double* mat) {
  return fp.sequential && mat[10] != 0.0 && !(level < fp.max_depth && mat[10] > EPS);
}

// Segment for the children of 64-hit chunk `chunk` of a level with `count` hits: groups of G consecutive chunks
// share a segment (their children stay together in the queue: neighbouring hits -> neighbouring rays for the next
// k_trace), the groups are dealt round-robin, and G is chosen so that every segment receives ~128 groups spread over
// the whole image whenever the level has that many chunks — the segments then fill evenly (within a few percent)
// whatever the image looks like and whatever wave happens to shade which chunk, so the queues need little headroom.
__device__ __forceinline__ uint32_t spawn_segment(uint32_t chunk, uint32_t count) {
  const uint32_t n_chunks = (count + 63u) >> 6;
#ifndef RT_SPAWN_GROUPS
#define RT_SPAWN_GROUPS 128
#endif
  const uint32_t G = max(1u, n_chunks / (uint32_t)(NSEG * RT_SPAWN_GROUPS));
  // round-robin with a skew per round: a plain g % NSEG hands a segment the same image columns in every round
  // (a 3840-wide frame has 32 groups per tile row), and its fill then follows the columns' content (measured +16 %)
  const uint32_t g = chunk / G;
  return (g + (g / (uint32_t)NSEG) * 37u) & (uint32_t)(NSEG - 1);
}

// Reserve queue slots for the children of one wave's hits: ONE atomic per wave on the counter of segment `seg`.  Returns physical slots (segment * seg_cap + offset), 0xFFFFFFFF = no slot
// (not needed, or the segment is full: Counters::overflow is set, the frame is rendered again with larger queues).
// Must be called convergently by every lane of the wave.
__device__ __forceinline__ void spawn_reserve(Counters* C, const FrameParams& fp, int level, uint32_t seg, bool needT, bool needR,
                                              uint32_t* sT, uint32_t* sR) {
  *sT = *sR = 0xFFFFFFFFu;
  const unsigned long long mT = __ballot(needT), mR = __ballot(needR);
  const uint32_t nT = (uint32_t)__popcll(mT), nR = (uint32_t)__popcll(mR);
  if (nT + nR == 0u) return;
  seg &= (uint32_t)(NSEG - 1);
  uint32_t base = 0;
  if (__lane_id() == 0) base = atomicAdd(&C->segcnt[(level + 1) & 1][seg * 32], nT + nR);
  base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
  const uint32_t seg_cap = fp.queue_cap / NSEG;
  const uint32_t oT = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mT >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mT, 0u));
  const uint32_t oR = base + nT + __builtin_amdgcn_mbcnt_hi((uint32_t)(mR >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mR, 0u));
  if (needT) { if (oT < seg_cap) *sT = seg * seg_cap + oT; else C->overflow = 1u; }
  if (needR) { if (oR < seg_cap) *sR = seg * seg_cap + oR; else C->overflow = 1u; }
}

__device__ __forceinline__ void spawn_children(bool active, int level, const FrameParams& fp, const double* mat,
                                               const RayIn& r, d3 hp, d3 n, RayQueue& nq, Counters* C, uint32_t seg) {
  bool deeper = active && level < fp.max_depth;
  bool needT = deeper && (mat[10] > EPS);
  bool needR = deeper && (mat[6] > EPS || mat[7] > EPS || mat[8] > EPS);
  uint32_t sT, sR;
  spawn_reserve(C, fp, level, seg, needT, needR, &sT, &sR);
  if (sT != 0xFFFFFFFFu) {
    nq.ox[sT] = hp.x - n.x * EPS; nq.oy[sT] = hp.y - n.y * EPS; nq.oz[sT] = hp.z - n.z * EPS;
    nq.dx[sT] = r.d.x; nq.dy[sT] = r.d.y; nq.dz[sT] = r.d.z;
    nq.wx[sT] = r.w.x * mat[10]; nq.wy[sT] = r.w.y * mat[10]; nq.wz[sT] = r.w.z * mat[10];
    nq.pix[sT] = r.lp;
  }
  if (sR != 0xFFFFFFFFu) {
    double dn = dot3(r.d, n);                                    // reflect_batch, :34-36
    nq.dx[sR] = r.d.x - (2 * dn) * n.x; nq.dy[sR] = r.d.y - (2 * dn) * n.y; nq.dz[sR] = r.d.z - (2 * dn) * n.z;
    nq.ox[sR] = hp.x + n.x * EPS; nq.oy[sR] = hp.y + n.y * EPS; nq.oz[sR] = hp.z + n.z * EPS;
    nq.wx[sR] = r.w.x * mat[6]; nq.wy[sR] = r.w.y * mat[7]; nq.wz[sR] = r.w.z * mat[8];
    nq.pix[sR] = r.lp;
  }
}

// ------------------------------------------------------------------------- k_shade_exact -------
// Jitter sources of the fp64 shading code.  get(light, cell, &ui, &uj) returns the two uniforms of
// one shadow-grid cell for the hit being shaded.
struct JitPixelKeyed {          // mode B: pure function of (seed, pixel, depth, light, cell)
  uint32_t seed, gpix, depth;
  JitKey key;
  __device__ __forceinline__ void begin_light(int li) { key = jit_light_key(seed, gpix, depth, (uint32_t)li); }
  __device__ __forceinline__ void get(int, int cell, double* ui, double* uj) const {
    *ui = (double)jit_ui(key, (uint32_t)cell); *uj = (double)jit_uj(key, (uint32_t)cell);
  }
};
struct JitTable {               // mode A: table[((light*cells + cell)*2 + axis)*H + h] drawn from the NumPy stream
  const double* tab; size_t H, h; int n_cells;
  __device__ __forceinline__ void begin_light(int) {}
  __device__ __forceinline__ void get(int li, int cell, double* ui, double* uj) const {
    const size_t base = ((size_t)li * n_cells + cell) * 2;
    *ui = tab[base * H + h]; *uj = tab[(base + 1) * H + h];
  }
};

// compute_soft_shadow_batch (ray_tracer.py:171-244) with compute_shadow_transmission_batch (:247-293) for ONE
// (shadow origin p, light li): mean transmission over the n x n jittered samples, literal fp64.
template <typename JIT>
__device__ __forceinline__ double soft_shadow_exact(const DevScene& sc, const Prim* __restrict__ P, d3 p, int li, int n_sh,
                                                    JIT& jit, int* queries, bool seq, double pthr, const Accel* ac) {
  const double* lg = sc.lights + LIGHT_STRIDE * li;
  const d3 Lp = mk3(lg[0], lg[1], lg[2]);
  double T;
  // small scenes: conservative candidate set + fully-lit / umbra shortcut (rt_device.hpp, fan_cull)
  PrimMask cmask = PrimMask::all();
  bool shortcut = false;
  if (sc.n <= 128 && (ac == nullptr || ac->n_nodes == 0)) {
    const FanCull fc = fan_cull(sc, P, p, Lp, n_sh > 1 ? 0.7072f * (float)lg[8] : 0.0f);
    cmask = fc.act;
    if (!__any(fc.lane_can & !fc.lane_full)) {
      shortcut = true;
      T = fc.lane_full ? 0.0 : 1.0;                                              // n^2 terms of exactly 0 or 1
      *queries += n_sh > 1 ? n_sh * n_sh : 1;                                    // one query per walk
    }
  }
  if (shortcut) {
  } else if (n_sh <= 1) {                                                               // :184-196
    d3 tl = mk3(Lp.x - p.x, Lp.y - p.y, Lp.z - p.z);
    double dist = norm3(tl);
    d3 dir = mk3(tl.x / dist, tl.y / dist, tl.z / dist);
    d3 so = mk3(p.x + dir.x * EPS, p.y + dir.y * EPS, p.z + dir.z * EPS);
    T = shadow_walk(P, sc.nS, sc.nP, sc.nB, sc.mats, so, dir, dist - EPS, queries, seq, pthr, ac, cmask);
  } else {
    d3 tl = mk3(Lp.x - p.x, Lp.y - p.y, Lp.z - p.z);                             // :198-200
    double ldist = norm3(tl);
    d3 ld = mk3(tl.x / ldist, tl.y / ldist, tl.z / ldist);
    d3 ref = (fabs(ld.x) >= 0.9) ? mk3(0, 1, 0) : mk3(1, 0, 0);                  // :202-205
    d3 right = cross3(ld, ref);                                                  // :207-209
    double rn = norm3(right); if (rn < EPS) rn = EPS;
    right = mk3(right.x / rn, right.y / rn, right.z / rn);
    d3 up = cross3(right, ld);                                                   // :211-213
    double un = norm3(up); if (un < EPS) un = EPS;
    up = mk3(up.x / un, up.y / un, up.z / un);
    const double radius = lg[8];
    const double cell = radius / n_sh;                                           // :215
    jit.begin_light(li);
    double total = 0.0;
    for (int i = 0; i < n_sh; ++i)
      for (int j = 0; j < n_sh; ++j) {                                           // :219-241
        double ui, uj;
        jit.get(li, i * n_sh + j, &ui, &uj);
        double oi = -radius / 2 + cell * (i + ui);
        double oj = -radius / 2 + cell * (j + uj);
        d3 sp = mk3((Lp.x + oi * right.x) + oj * up.x, (Lp.y + oi * right.y) + oj * up.y,
                    (Lp.z + oi * right.z) + oj * up.z);
        d3 ts = mk3(sp.x - p.x, sp.y - p.y, sp.z - p.z);
        double dist = norm3(ts);
        d3 dir = mk3(ts.x / dist, ts.y / dist, ts.z / dist);
        d3 so = mk3(p.x + dir.x * EPS, p.y + dir.y * EPS, p.z + dir.z * EPS);
        total += shadow_walk(P, sc.nS, sc.nP, sc.nB, sc.mats, so, dir, dist - EPS, queries, seq, pthr, ac, cmask);
      }
    T = total / (double)(n_sh * n_sh);                                           // :243
  }
  return T;
}

// The shadow ray of grid cell `cell` of light li from the shadow origin p, formed exactly as the reference forms it
// (ray_tracer.py:184-235, fp64, pixel-keyed jitter): origin p + dir*1e-6, unit direction, length |S - p| - 1e-6.
struct ExactRay { d3 so, dir; double maxdist; };
__device__ __forceinline__ ExactRay exact_shadow_ray_l(const double* __restrict__ lights, d3 p, int li, int n_sh, uint32_t seed, uint32_t gpix,
                                                       uint32_t depth, int cell) {
  const double* lg = lights + LIGHT_STRIDE * li;
  const d3 Lp = mk3(lg[0], lg[1], lg[2]);
  d3 sp = Lp;
  if (n_sh > 1) {
    const d3 tl = mk3(Lp.x - p.x, Lp.y - p.y, Lp.z - p.z);                       // :198-200
    const double ldist = norm3(tl);
    const d3 ld = mk3(tl.x / ldist, tl.y / ldist, tl.z / ldist);
    const d3 ref = (fabs(ld.x) >= 0.9) ? mk3(0, 1, 0) : mk3(1, 0, 0);            // :202-205
    d3 right = cross3(ld, ref);                                                  // :207-209
    double rn = norm3(right); if (rn < EPS) rn = EPS;
    right = mk3(right.x / rn, right.y / rn, right.z / rn);
    d3 up = cross3(right, ld);                                                   // :211-213
    double un = norm3(up); if (un < EPS) un = EPS;
    up = mk3(up.x / un, up.y / un, up.z / un);
    const double radius = lg[8], cw = radius / n_sh;                             // :215
    const JitKey key = jit_light_key(seed, gpix, depth, (uint32_t)li);
    const int i = cell / n_sh, j = cell - i * n_sh;
    const double oi = -radius / 2 + cw * (i + (double)jit_ui(key, (uint32_t)cell));
    const double oj = -radius / 2 + cw * (j + (double)jit_uj(key, (uint32_t)cell));
    sp = mk3((Lp.x + oi * right.x) + oj * up.x, (Lp.y + oi * right.y) + oj * up.y, (Lp.z + oi * right.z) + oj * up.z);
  }
  const d3 ts = mk3(sp.x - p.x, sp.y - p.y, sp.z - p.z);
  const double dist = norm3(ts);
  ExactRay r;
  r.dir = mk3(ts.x / dist, ts.y / dist, ts.z / dist);
  r.so = mk3(p.x + r.dir.x * EPS, p.y + r.dir.y * EPS, p.z + r.dir.z * EPS);
  r.maxdist = dist - EPS;
  return r;
}

// The switch of the light basis' helper vector, |ld.x| >= 0.9, decided as the reference decides it (fp64,
// ray_tracer.py:198-205): the fp32 kernels call this when their own |ld.x| is within rounding of 0.9.
__device__ __forceinline__ bool basis_switch_exact(const DevScene& sc, d3 p, int li) {
  const double* lg = sc.lights + LIGHT_STRIDE * li;
  const d3 tl = mk3(lg[0] - p.x, lg[1] - p.y, lg[2] - p.z);
  return fabs(tl.x / norm3(tl)) >= 0.9;
}
__device__ __forceinline__ ExactRay exact_shadow_ray(const DevScene& sc, d3 p, int li, int n_sh, uint32_t seed, uint32_t gpix,
                                                     uint32_t depth, int cell) {
  return exact_shadow_ray_l(sc.lights, p, li, n_sh, seed, gpix, depth, cell);
}

// The crossings of ONE primitive by a shadow ray, each decided in fp64 with the arithmetic of the reference's
// intersect_batch (sphere.py:49-71, infinite_plane.py:30-42, cube.py:76-120): in1 / in2 = the first / second crossing
// lies in (1e-6, maxdist) — the crossings the reference's walk (ray_tracer.py:257-291) multiplies transparencies for.
// The walk restarts 1e-6 behind every crossing and accepts hits beyond another 1e-6 (:287-288, sphere.py:70-71), so a
// second crossing of the same primitive closer than 2e-6 to the first (a ray through the very edge of a box) is not seen.
__device__ __forceinline__ void exact_crossings(const Prim& pr, d3 o, d3 d, double maxdist, bool* in1, bool* in2) {
  *in1 = *in2 = false;
  if (pr.type == 0) {
    const d3 oc = mk3(o.x - pr.a[0], o.y - pr.a[1], o.z - pr.a[2]);
    const double a = dot3(d, d), b = 2.0 * dot3(oc, d), c = dot3(oc, oc) - pr.a[4];
    const double disc = b * b - (4 * a) * c;
    if (!(disc >= 0)) return;
    const double sq = sqrt(disc);
    const double t1 = (-b - sq) / (2 * a), t2 = (-b + sq) / (2 * a);
    *in1 = (t1 > HIT_EPS) & (t1 < maxdist);
    *in2 = (t2 > HIT_EPS) & (t2 < maxdist) & !(*in1 & !(t2 - (t1 + EPS) > HIT_EPS));
  } else if (pr.type == 1) {
    const double t = plane_t(pr, o, d);
    *in1 = t < maxdist;
  } else {
    double tmin = -RT_INF, tmax = RT_INF;
    bool valid = true;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      const double dv = comp(d, ax), ov = comp(o, ax);
      const bool parallel = fabs(dv) < PAR_EPS;
      if (parallel && ((ov < pr.a[ax]) || (ov > pr.a[3 + ax]))) valid = false;
      if (!parallel && valid) {
        const double t1 = (pr.a[ax] - ov) / dv, t2 = (pr.a[3 + ax] - ov) / dv;
        tmin = fmax(tmin, fmin(t1, t2)); tmax = fmin(tmax, fmax(t1, t2));
        valid = valid && (tmin <= tmax);
      }
    }
    if (!valid) return;
    *in1 = (tmin >= HIT_EPS) & (tmin < maxdist);
    *in2 = (tmax >= HIT_EPS) & (tmax < maxdist) & !(*in1 & !(tmax - (tmin + EPS) >= HIT_EPS));
  }
}

// The exact decision for ONE (sample, primitive) pair, out of line: the fp32 shade kernels call it for the ~1e-5 of
// their tests that fall within the fp32 error bound.  A real function call on purpose — inlined at every test site it
// cost k_shade_fast 83 spilled VGPRs and 40 % more code in the loops that never need it.  Returns in1 | in2 << 1.
__device__ __attribute__((noinline)) uint32_t exact_retest(const double* __restrict__ lights, const Prim* prim, double px, double py, double pz,
                                                           int li, int n_sh, uint32_t seed, uint32_t gpix, uint32_t depth, int cell) {
  const ExactRay er = exact_shadow_ray_l(lights, mk3(px, py, pz), li, n_sh, seed, gpix, depth, cell);
  bool e1 = false, e2 = false;
  exact_crossings(*prim, er.so, er.dir, er.maxdist, &e1, &e2);
  return (e1 ? 1u : 0u) | (e2 ? 2u : 0u);
}


struct ExactHit { d3 hp, n; const double* mat; double c[3]; };   // shading result of one hit

// One light's Phong terms for one hit, given the light's shadow term T: ray_tracer.py:530-557, literal fp64
// (seq: the sequential renderer's grouping and rules, :343-359).
__device__ __forceinline__ void phong_light_exact(const double* __restrict__ lg, const double* __restrict__ mat, d3 hp, d3 n, d3 v, double T,
                                                  bool seq, double (&dif)[3], double (&spc)[3]) {
  const d3 Lp = mk3(lg[0], lg[1], lg[2]);
  double I = (1 - lg[7]) + lg[7] * T;                                            // :543
  if (seq && I < EPS) return;                                                    // sequential: :347-348
  d3 tl = mk3(Lp.x - hp.x, Lp.y - hp.y, Lp.z - hp.z);                            // :530-532 (un-offset hp)
  double ldist = norm3(tl);
  d3 l = mk3(tl.x / ldist, tl.y / ldist, tl.z / ldist);
  d3 ml = mk3(-l.x, -l.y, -l.z);
  double ndl = dot3(n, l); if (!(ndl > 0)) ndl = 0;                              // :545
  if (seq) {                                                                     // :351 ((Kd*Lc)*I)*ndl
    dif[0] += ((mat[0] * lg[3]) * I) * ndl; dif[1] += ((mat[1] * lg[4]) * I) * ndl; dif[2] += ((mat[2] * lg[5]) * I) * ndl;
  } else {
    double s1 = I * ndl;
    dif[0] += (mat[0] * lg[3]) * s1; dif[1] += (mat[1] * lg[4]) * s1; dif[2] += (mat[2] * lg[5]) * s1;   // :547
  }
  double dn = dot3(ml, n);                                                       // :549
  d3 rf = mk3(ml.x - (2 * dn) * n.x, ml.y - (2 * dn) * n.y, ml.z - (2 * dn) * n.z);
  double rdv = dot3(rf, v); if (!(rdv > 0)) rdv = 0;                             // :550
  if (seq) {                                                                     // :356-359, only when r.v > 0
    if (rdv > 0) {
      const double pw = pow(rdv, mat[9]);
      spc[0] += (((mat[3] * lg[3]) * lg[6]) * I) * pw; spc[1] += (((mat[4] * lg[4]) * lg[6]) * I) * pw;
      spc[2] += (((mat[5] * lg[5]) * lg[6]) * I) * pw;
    }
  } else {
    double s2 = I * pow(rdv, mat[9]);                                            // :555-556
    spc[0] += ((mat[3] * lg[3]) * lg[6]) * s2; spc[1] += ((mat[4] * lg[4]) * lg[6]) * s2;
    spc[2] += ((mat[5] * lg[5]) * lg[6]) * s2;
  }
}

// Literal fp64 restatement of ray_tracer.py:517-563 for one hit, with compute_soft_shadow_batch
// (:171-244) and compute_shadow_transmission_batch (:247-293) inlined.  c = (diffuse+specular)*(1-transparency).
template <typename JIT>
__device__ __forceinline__ ExactHit shade_hit_exact(const DevScene& sc, const Prim* __restrict__ P, const RayIn& r,
                                                    const HitRec& hr, int n_sh, JIT& jit, int* queries,
                                                    bool seq = false, double pthr = HIT_EPS, const Accel* ac = nullptr) {
  ExactHit out;
  const Prim& pr = P[hr.prim];
  const double t = hr.t;
  const d3 n = prim_normal(pr, r.o, r.d, t);
  const d3 hp = mk3(r.o.x + t * r.d.x, r.o.y + t * r.d.y, r.o.z + t * r.d.z);      // :517
  const double* mat = sc.mats + MAT_STRIDE * pr.mat;
  d3 v = mk3(r.o.x - hp.x, r.o.y - hp.y, r.o.z - hp.z);                            // :519
  double vn = norm3(v); if (vn < EPS) vn = seq ? 1.0 : EPS;                        // normalize_batch :24-27 / normalize :17-21
  v = mk3(v.x / vn, v.y / vn, v.z / vn);
  const d3 p = mk3(hp.x + n.x * EPS, hp.y + n.y * EPS, hp.z + n.z * EPS);          // :538
  double dif[3] = {0, 0, 0}, spc[3] = {0, 0, 0};
  for (int li = 0; li < sc.nL; ++li) {                                             // :526
    const double T = soft_shadow_exact(sc, P, p, li, n_sh, jit, queries, seq, pthr, ac);
    phong_light_exact(sc.lights + LIGHT_STRIDE * li, mat, hp, n, v, T, seq, dif, spc);
  }
  const double one_m = 1 - mat[10];                                                // :562
  out.hp = hp; out.n = n; out.mat = mat;
  out.c[0] = (dif[0] + spc[0]) * one_m; out.c[1] = (dif[1] + spc[1]) * one_m; out.c[2] = (dif[2] + spc[2]) * one_m;
  return out;
}

// fp64 shading of a hit queue (precision = RT_FP64, pixel-keyed jitter).
template <typename FB, bool LDS_SCENE>
__global__ __launch_bounds__(BLOCK) void k_shade_exact(DevScene sc, DevCamera cam, FrameParams fp, int level, RayQueue q,
                                                       RayQueue nq, const HitRec* __restrict__ hits, Counters* C,
                                                       FB* __restrict__ fb) {
  __shared__ Prim s_prims[LDS_SCENE ? LDS_PRIMS : 1];
  __shared__ BvhNode s_nodes[LDS_SCENE ? 1 : LDS_NODES];
  const Prim* P = stage_scene<LDS_SCENE>(sc, s_prims);
  const Accel ac = stage_accel<LDS_SCENE>(sc, s_nodes);
  uint32_t count = C->n_hits[level];
  if (count > fp.hit_cap) count = fp.hit_cap;
  const uint32_t lane = __lane_id();
  ChunkCursor cc; cc.init(C->stripes, count, blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6));
  for (uint32_t chunk = cc.take(cc.fetch()); chunk != ChunkCursor::NONE;) {
    const uint32_t base = chunk * 64u;
    uint32_t h = base + lane;
    bool active = h < count;
    RayIn r; d3 hp = mk3(0, 0, 0), n = mk3(0, 0, 0);
    const double* mat = sc.mats;
    r.d = r.o = r.w = mk3(0, 0, 0); r.lp = 0;
    HitRec hr; hr.slot = 0; hr.prim = 0; hr.t = 0.0;
    if (active) {
      hr = hits[h];
      r = load_ray(level, hr.slot, cam, fp.rs, q);
    }
    const uint32_t next_chunk = cc.fetch();      // (answered while this chunk is shaded)
    if (active) {
      JitPixelKeyed jit; jit.seed = fp.seed; jit.gpix = global_pixel(fp.rs, r.lp); jit.depth = (uint32_t)level;
      int queries = 0;
      ExactHit e = shade_hit_exact(sc, P, r, hr, fp.n_shadow, jit, &queries, fp.sequential != 0, fp.plane_thr, &ac);
      hp = e.hp; n = e.n; mat = e.mat;
      if (seq_background_term(fp, level, mat)) {                                          // :370-377
        e.c[0] += sc.bg[0] * mat[10]; e.c[1] += sc.bg[1] * mat[10]; e.c[2] += sc.bg[2] * mat[10];
      }
      fb_put(fb, level == 0, r.lp, r.w.x * e.c[0], r.w.y * e.c[1], r.w.z * e.c[2]);      // :563
      atomicAdd(&C->queries, (unsigned long long)queries);
    }
    spawn_children(active, level, fp, mat, r, hp, n, nq, C, spawn_segment(base >> 6, count));
    chunk = cc.take(next_chunk);
  }
}

// fp64 shading of a hit queue for LDS-resident scenes, shadow samples spread over the lanes (round 3).
// k_shade_exact gives a hit to a lane, and the lane walks its lights x n x n shadow samples one after the other; after
// fan_cull only the penumbra lanes of a wave are still in those loops (a quarter of them on original.txt), the others
// wait.  Here the samples of the penumbra lanes become ITEMS: per light, the lanes that need sampling are listed
// (ballot + mbcnt), each computes the light's basis for its hit once (ray_tracer.py:198-213) and parks it in LDS, and then
// the wave works through (listed hit, cell) items 64 at a time, so the walks run on full waves; the terms are parked in
// LDS and each listed hit's lane adds its own n^2 terms left to right.  Per sample AND per sum the arithmetic and its
// order are those of soft_shadow_exact: results are bit-identical to k_shade_exact's.
constexpr int ITEM_CHUNK = 512;
#ifndef EXACT_ITEMS_WAVES
#define EXACT_ITEMS_WAVES 3      // 168 VGPRs (35 spilled): the fp64 chains are latency bound, 3 waves per SIMD measured 10 % faster than 2
#endif
template <typename FB>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(EXACT_ITEMS_WAVES))) void k_shade_exact_items(DevScene sc, DevCamera cam, FrameParams fp, int level, RayQueue q,
                                                             RayQueue nq, const HitRec* __restrict__ hits, Counters* C,
                                                             FB* __restrict__ fb) {
  __shared__ Prim s_prims[LDS_PRIMS];
  __shared__ double s_p[3][BLOCK];              // shadow origin of each lane's hit
  __shared__ double s_basis[BLOCK / 64][64][6]; // per wave: (right, up) of the listed hits for the current light
  __shared__ double s_T[BLOCK / 64][64];        // per wave: shadow term of the listed hits
  __shared__ double s_cellT[BLOCK / 64][ITEM_CHUNK];   // per wave: the terms of a chunk of items
  __shared__ uint32_t s_gpix[BLOCK];
  __shared__ unsigned char s_list[BLOCK / 64][64];
  const Prim* P = stage_scene<true>(sc, s_prims);
  // level 0 fused with the nearest-hit query (fp.fuse0), as in k_shade_ls: slots are pixels, every lane traces its own
  // camera ray (nearest_coherent: the literal fp64 tests, skipped per wave for spheres no lane can hit)
  const bool fused = level == 0 && fp.fuse0 != 0u;
  uint32_t count = fused ? C->n_rays[0] : C->n_hits[level];
  if (count > fp.hit_cap) count = fp.hit_cap;
  uint32_t fused_hits = 0u;
  const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6, t_ = threadIdx.x;
  const int n_sh = fp.n_shadow;
  const int n_cells = n_sh > 1 ? n_sh * n_sh : 1;
  const int hits_per_chunk = max(1, ITEM_CHUNK / n_cells);       // (launch_frame uses this kernel for n^2 <= ITEM_CHUNK only)
  const bool seq = fp.sequential != 0;
  unsigned long long my_queries = 0ull;
  ChunkCursor cc; cc.init(C->stripes, count, blockIdx.x * (BLOCK / 64) + wave);
  for (uint32_t chunk = cc.take(cc.fetch()); chunk != ChunkCursor::NONE;) {
    const uint32_t base = chunk * 64u;
    const uint32_t h = base + lane;
    bool active = h < count;
    RayIn r; d3 hp = mk3(0, 0, 0), n = mk3(0, 0, 0), v = mk3(0, 0, 0), p = mk3(0, 0, 0);
    const double* mat = sc.mats;
    r.d = r.o = r.w = mk3(0, 0, 0); r.lp = 0;
    HitRec hr;
    if (fused) {
      r = load_ray(0, active ? h : count - 1, cam, fp.rs, q);
      int k_ = -1;
      hr.t = nearest_coherent(P, sc.nS, sc.nP, sc.nB, r.o, r.d, &k_, fp.plane_thr);
      hr.prim = k_; hr.slot = h;
      if (active && k_ < 0) {                  // ray_tracer.py:500-502 (0 + 1*bg == bg exactly)
        fb[3 * (size_t)r.lp + 0] = (FB)sc.bg[0]; fb[3 * (size_t)r.lp + 1] = (FB)sc.bg[1]; fb[3 * (size_t)r.lp + 2] = (FB)sc.bg[2];
      }
      active = active && k_ >= 0;
      const unsigned long long mh = __ballot(active);
      fused_hits += (uint32_t)__popcll(mh);
      if (mh == 0ull) { chunk = cc.take(cc.fetch()); continue; }       // a tile of background pixels
      if (k_ < 0) {                            // a lane without a hit of its own shades a copy of a neighbour's (results dropped)
        const int srcl = __ffsll((long long)mh) - 1;
        hr.prim = __shfl(hr.prim, srcl); hr.t = __shfl(hr.t, srcl);
        r.d = mk3(__shfl(r.d.x, srcl), __shfl(r.d.y, srcl), __shfl(r.d.z, srcl));
      }
    } else {
      // (the idle lanes of a level's last wave shade a copy of its last hit, results dropped: fan_cull votes per wave)
      hr = hits[active ? h : count - 1];
      r = load_ray(level, hr.slot, cam, fp.rs, q);
    }
    const uint32_t next_chunk = cc.fetch();      // (answered while this chunk is shaded)
    {
      const Prim& pr = P[hr.prim];
      n = prim_normal(pr, r.o, r.d, hr.t);
      hp = mk3(r.o.x + hr.t * r.d.x, r.o.y + hr.t * r.d.y, r.o.z + hr.t * r.d.z);      // :517
      mat = sc.mats + MAT_STRIDE * pr.mat;
      v = mk3(r.o.x - hp.x, r.o.y - hp.y, r.o.z - hp.z);                                // :519
      double vn = norm3(v); if (vn < EPS) vn = seq ? 1.0 : EPS;
      v = mk3(v.x / vn, v.y / vn, v.z / vn);
      p = mk3(hp.x + n.x * EPS, hp.y + n.y * EPS, hp.z + n.z * EPS);                    // :538
    }
    s_p[0][t_] = p.x; s_p[1][t_] = p.y; s_p[2][t_] = p.z;
    s_gpix[t_] = global_pixel(fp.rs, r.lp);
    double dif[3] = {0, 0, 0}, spc[3] = {0, 0, 0};
    for (int li = 0; li < sc.nL; ++li) {                                                // :526
      const double* lg = sc.lights + LIGHT_STRIDE * li;
      const d3 Lp = mk3(lg[0], lg[1], lg[2]);
      const FanCull fc = fan_cull(sc, P, p, Lp, n_sh > 1 ? 0.7072f * (float)lg[8] : 0.0f);
      const bool need = active & fc.lane_can & !fc.lane_full;    // (the copies in the idle lanes of a last wave are not listed)
      double T = fc.lane_full ? 0.0 : 1.0;                       // the conservative claims: n^2 terms of exactly 0 / 1
      if (active && !need) my_queries += (unsigned long long)n_cells;                   // (one query per walk, as the walks would count)
      const unsigned long long m = __ballot(need);
      const int n_need = __popcll(m);
      if (n_need > 0) {
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (need) {
          s_list[wave][rank] = (unsigned char)lane;
          if (n_sh > 1) {                                        // the light's basis for this hit, once (:198-213)
            const d3 tl = mk3(Lp.x - p.x, Lp.y - p.y, Lp.z - p.z);
            const double ldist = norm3(tl);
            const d3 ld = mk3(tl.x / ldist, tl.y / ldist, tl.z / ldist);
            const d3 ref = (fabs(ld.x) >= 0.9) ? mk3(0, 1, 0) : mk3(1, 0, 0);
            d3 right = cross3(ld, ref);
            double rn = norm3(right); if (rn < EPS) rn = EPS;
            right = mk3(right.x / rn, right.y / rn, right.z / rn);
            d3 up = cross3(right, ld);
            double un = norm3(up); if (un < EPS) un = EPS;
            up = mk3(up.x / un, up.y / un, up.z / un);
            double* b = s_basis[wave][rank];
            b[0] = right.x; b[1] = right.y; b[2] = right.z; b[3] = up.x; b[4] = up.y; b[5] = up.z;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const double radius = lg[8], cw = radius / (n_sh > 1 ? n_sh : 1);                // :215
        // items = (listed hit, cell), dense: item i is cell i % n^2 of listed hit i / n^2, 64 items per round, so a light
        // never takes more rounds than the n^2 sample iterations of the one-lane-per-hit kernel.  The terms go to LDS in
        // chunks of <= ITEM_CHUNK items (whole hits) and each listed hit's lane adds its n^2 terms left to right, the
        // reference's own order (ray_tracer.py:241).
        for (int e0 = 0; e0 < n_need; e0 += hits_per_chunk) {
          const int n_e = min(hits_per_chunk, n_need - e0), n_items = n_e * n_cells;
          for (int i0 = 0; i0 < n_items; i0 += 64) {
            const int it = i0 + (int)lane;
            if (it < n_items) {
              const int el = it / n_cells, cell = it - el * n_cells, e = e0 + el;
              const uint32_t src = wave * 64u + s_list[wave][e];
              const d3 pe = mk3(s_p[0][src], s_p[1][src], s_p[2][src]);
              d3 sp = Lp;
              if (n_sh > 1) {
                const double* b = s_basis[wave][e];
                const JitKey key = jit_light_key(fp.seed, s_gpix[src], (uint32_t)level, (uint32_t)li);
                const int i = cell / n_sh, j = cell - i * n_sh;
                const double oi = -radius / 2 + cw * (i + (double)jit_ui(key, (uint32_t)cell));    // :224-225
                const double oj = -radius / 2 + cw * (j + (double)jit_uj(key, (uint32_t)cell));
                sp = mk3((Lp.x + oi * b[0]) + oj * b[3], (Lp.y + oi * b[1]) + oj * b[4], (Lp.z + oi * b[2]) + oj * b[5]);
              }
              const d3 ts = mk3(sp.x - pe.x, sp.y - pe.y, sp.z - pe.z);
              const double dist = norm3(ts);
              const d3 dir = mk3(ts.x / dist, ts.y / dist, ts.z / dist);
              const d3 so = mk3(pe.x + dir.x * EPS, pe.y + dir.y * EPS, pe.z + dir.z * EPS);
              int qn = 0;
              s_cellT[wave][it] = shadow_walk<true>(P, sc.nS, sc.nP, sc.nB, sc.mats, so, dir, dist - EPS, &qn, seq, fp.plane_thr, nullptr, fc.act);
              my_queries += (unsigned long long)qn;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          if ((int)lane < n_e) {
            double total = 0.0;
            for (int c = 0; c < n_cells; ++c) total += s_cellT[wave][(int)lane * n_cells + c];
            s_T[wave][e0 + (int)lane] = total / (double)n_cells;                        // :243
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (need) T = s_T[wave][rank];
      }
      phong_light_exact(lg, mat, hp, n, v, T, seq, dif, spc);
    }
    if (active) {
      const double one_m = 1 - mat[10];                                                  // :562
      double c0 = (dif[0] + spc[0]) * one_m, c1 = (dif[1] + spc[1]) * one_m, c2 = (dif[2] + spc[2]) * one_m;
      if (seq_background_term(fp, level, mat)) {                                         // :370-377
        c0 += sc.bg[0] * mat[10]; c1 += sc.bg[1] * mat[10]; c2 += sc.bg[2] * mat[10];
      }
      fb_put(fb, level == 0, r.lp, r.w.x * c0, r.w.y * c1, r.w.z * c2);                  // :563
    }
    spawn_children(active, level, fp, mat, r, hp, n, nq, C, spawn_segment(base >> 6, count));
    chunk = cc.take(next_chunk);
  }
  for (int o_ = 32; o_ > 0; o_ >>= 1) my_queries += __shfl_xor(my_queries, o_);
  if (lane == 0 && my_queries) atomicAdd(&C->queries, my_queries);
  if (fused && lane == 0 && fused_hits) atomicAdd(&C->n_hits[0], fused_hits);
}

// --------------------------------------------------------------------------- k_fix_exact -------
// fp32 mode, after the last level: the queued (hit, light) pairs get their shadow term from the literal fp64
// code of the fp64 mode (soft_shadow_exact: same jitter, same walk), one lane per record.  The fp32 shade
// kernels queue a pair whenever one of its hit / miss decisions lies within the error bound of their fp32
// arithmetic (rt_shade_ls.hpp, rt_shade_fast.hpp), and shaded it with T = 0; the colour is linear in T.
template <bool LDS_SCENE, bool SEQ>
__global__ __launch_bounds__(BLOCK) void k_fix_exact(DevScene sc, FrameParams fp, const FixRec* __restrict__ fix,
                                                     Counters* C, float* __restrict__ fb) {
  __shared__ Prim s_prims[LDS_SCENE ? LDS_PRIMS : 1];
  __shared__ BvhNode s_nodes[LDS_SCENE ? 1 : LDS_NODES];
  // block -> (sub-queue, slice): grid = FIX_QUEUES * slices
  const uint32_t qi = blockIdx.x % FIX_QUEUES, slice = blockIdx.x / FIX_QUEUES, n_slices = gridDim.x / FIX_QUEUES;
  uint32_t count = C->fixq[qi * 32];
  if (count > fp.fix_cap) count = fp.fix_cap;
  // A record's n x n shadow walks are independent: they go to the lanes of a GROUP of g lanes (g = the power of two
  // >= n^2, at most 64), 64 / g records per wave — a record's latency is one walk, not n^2 of them.
  const int n_cells = fp.n_shadow > 1 ? fp.n_shadow * fp.n_shadow : 1;
  int g = 1; while (g < n_cells && g < 64) g <<= 1;
  const uint32_t per_block = (uint32_t)(BLOCK / g);
  if (slice * per_block >= count) return;       // (uniform over the block; most blocks of most frames)
  const Prim* P = stage_scene<LDS_SCENE>(sc, s_prims);
  const Accel ac = stage_accel<LDS_SCENE>(sc, s_nodes);
  const FixRec* __restrict__ mine = fix + (size_t)qi * fp.fix_cap;
  const uint32_t member = threadIdx.x & (uint32_t)(g - 1);
  for (uint32_t base = slice * per_block; base < count; base += n_slices * per_block) {
    const uint32_t i = base + threadIdx.x / (uint32_t)g;
    // the idle groups of a sub-queue's last wave re-test a copy of its last record (result dropped): the
    // conservative pre-pass (fan_cull) is wave-level
    const bool live = i < count;
    const FixRec rec = mine[live ? i : count - 1];
    const d3 p = mk3(rec.px, rec.py, rec.pz);
    const int li = (int)rec.light;
    const double* lg = sc.lights + LIGHT_STRIDE * li;
    // small scenes: conservative candidate set + fully-lit / umbra claim (rt_device.hpp, fan_cull); the lanes of a wave
    // may hold different records: the candidate set is the union, the claims are per lane
    PrimMask cmask = PrimMask::all();
    bool claim = false; double claimed = 0.0;
    if (LDS_SCENE) {
      // (the g lanes of a record share its primitives out: fan_cull_group)
      const FanCull fc = fan_cull_group(sc, P, p, mk3(lg[0], lg[1], lg[2]), fp.n_shadow > 1 ? 0.7072f * (float)lg[8] : 0.0f, member, g);
      cmask = fc.act;
      if (!fc.lane_can) { claim = true; claimed = 1.0; }
      else if (fc.lane_full) { claim = true; claimed = 0.0; }
    }
    double tsum = 0.0;
    if (!__all(claim)) {
      for (int cell = (int)member; cell < n_cells; cell += g) {
        const ExactRay er = exact_shadow_ray(sc, p, li, fp.n_shadow, fp.seed, rec.gpix, rec.level, cell);
        int queries = 0;
        tsum += shadow_walk(P, sc.nS, sc.nP, sc.nB, sc.mats, er.so, er.dir, er.maxdist, &queries, SEQ, fp.plane_thr, &ac, cmask);
      }
      for (int o_ = g >> 1; o_ > 0; o_ >>= 1) tsum += __shfl_xor(tsum, o_);
    }
    const double T = claim ? claimed : tsum / (double)n_cells;
    const float Tf = (float)T;
    if (live && member == 0u && Tf != 0.0f) {
      float* dst = fb + 3 * (size_t)rec.lp;
      atomicAdd(dst + 0, rec.m[0] * Tf); atomicAdd(dst + 1, rec.m[1] * Tf); atomicAdd(dst + 2, rec.m[2] * Tf);
    }
  }
}

// Queue one (hit, light) pair for the exact re-test.  Must be called convergently by every lane of the wave.
__device__ __forceinline__ void queue_fix(bool pred, FixRec* __restrict__ fix, Counters* C, const FrameParams& fp, d3 p,
                                          float m0, float m1, float m2, uint32_t lp, uint32_t gpix, int level, int li) {
  const uint32_t qi = blockIdx.x % FIX_QUEUES;
  const uint32_t slot = wave_append(&C->fixq[qi * 32], pred);
  if (pred) {
    if (slot < fp.fix_cap) {
      FixRec r; r.px = p.x; r.py = p.y; r.pz = p.z; r.m[0] = m0; r.m[1] = m1; r.m[2] = m2;
      r.lp = lp; r.gpix = gpix; r.level = (uint16_t)level; r.light = (uint16_t)li;
      fix[(size_t)qi * fp.fix_cap + slot] = r;
    } else C->overflow_fix = 1u;   // (the frame is rendered again with a larger queue, like a ray-queue overflow)
  }
}

// --------------------------------------------------------------------- utility kernels ---------
__global__ void k_init_counters(Counters* C, uint32_t n_level0) {
  uint32_t* w = reinterpret_cast<uint32_t*>(C);
  for (uint32_t i = threadIdx.x; i < sizeof(Counters) / 4; i += blockDim.x) w[i] = 0u;
  __syncthreads();
  if (threadIdx.x == 0) C->n_rays[0] = n_level0;
}

// Output conversion; U8 is the reference's save_image quantiser (ray_tracer.py:607-608):
// clip to [0,1], x255, truncate toward zero.
template <typename FB, typename OUT>
__global__ __launch_bounds__(BLOCK) void k_convert(const FB* __restrict__ fb, OUT* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    double v = (double)fb[i];
    if (sizeof(OUT) == 1) {
      v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
      out[i] = (OUT)(int)(v * 255.0);
    } else {
      out[i] = (OUT)v;
    }
  }
}

// rt_nearest_hit: find_nearest_intersection_batch on caller-supplied rays (AoS (M,3) doubles).
__global__ __launch_bounds__(BLOCK) void k_nearest_rays(DevScene sc, const double* __restrict__ o, const double* __restrict__ d,
                                                        long long M, double* t_out, int* idx_out, double* n_out) {
  long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  d3 ro = mk3(o[3 * m], o[3 * m + 1], o[3 * m + 2]), rd = mk3(d[3 * m], d[3 * m + 1], d[3 * m + 2]);
  int k; double t = nearest(sc.prims, sc.nS, sc.nP, sc.nB, ro, rd, &k, HIT_EPS, &sc.ac);
  d3 n = mk3(0, 0, 0);
  if (k >= 0) n = prim_normal(sc.prims[k], ro, rd, t);
  t_out[m] = t;
  idx_out[m] = k >= 0 ? sc.prims[k].orig : -1;
  n_out[3 * m] = n.x; n_out[3 * m + 1] = n.y; n_out[3 * m + 2] = n.z;
}

// rt_shadow_transmission: compute_shadow_transmission_batch on caller-supplied rays.
__global__ __launch_bounds__(BLOCK) void k_shadow_rays(DevScene sc, const double* __restrict__ o, const double* __restrict__ d,
                                                       const double* __restrict__ maxd, long long M, double* out) {
  long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  int q = 0;
  out[m] = shadow_walk(sc.prims, sc.nS, sc.nP, sc.nB, sc.mats, mk3(o[3 * m], o[3 * m + 1], o[3 * m + 2]),
                       mk3(d[3 * m], d[3 * m + 1], d[3 * m + 2]), maxd[m], &q, false, HIT_EPS, &sc.ac);
}

// rt_camera_rays: Camera.generate_rays_for_rows.
__global__ __launch_bounds__(BLOCK) void k_camera_rays(DevCamera cam, int W, int H, int y0, long long M, double* o, double* d) {
  long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  int y = y0 + (int)(m / W), x = (int)(m % W);
  d3 dir = camera_dir(cam, W, H, x, y);
  o[3 * m] = cam.pos[0]; o[3 * m + 1] = cam.pos[1]; o[3 * m + 2] = cam.pos[2];
  d[3 * m] = dir.x; d[3 * m + 1] = dir.y; d[3 * m + 2] = dir.z;
}

#ifndef RT_NO_LAUNCHERS
// Workgroups of a shade launch.  The shade kernels take 64-hit chunks dynamically until the level is used up, so the grid
// only has to fill the GPU: resident workgroups per CU (occupancy query, once per kernel instantiation) x CUs.  Every
// workgroup stages the scene and its per-light tables first (~5 us); with the former 8 workgroups per CU the ones beyond
// the 2-3 resident per CU started after the work was gone and only lengthened the launch by their prologues (pool 4K:
// 1.80 -> 1.69 ms per frame, original 1080p fp64 2.26 -> 2.14).  RT_SHADE_GRID_PER_CU=<n> overrides (A/B).
template <auto Kernel>
static int shade_grid(int num_cu, int cap) {
  static const int per_cu = [] {
    if (const char* e = getenv("RT_SHADE_GRID_PER_CU")) return std::max(1, atoi(e));
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, Kernel, BLOCK, 0) != hipSuccess || n < 1) { (void)hipGetLastError(); n = 4; }
    return n;
  }();
  return std::max(1, std::min(cap, num_cu * per_cu));
}
#endif

}  // namespace rt
