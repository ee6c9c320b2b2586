// Spatial regrouping of a bounce level's hit list (large scenes only).
//
// The hits of level >= 1 arrive in the order of their parent rays: one wave's 64 hits are spread
// over the scene, so its lanes walk different parts of the culling hierarchy (every wave-level
// step is paid for the slowest lane, node and bound loads do not coalesce) and the wave-level leaf
// list of k_shade_fast overflows.  A counting sort by the grid cell of the hit point makes waves
// compact again; the children they spawn inherit the order, which helps the next level's k_trace.
// Each hit's shaded value does not depend on the order (jitter is keyed by pixel/level/light); the frame is the
// same up to accumulation-order rounding (framebuffer updates are floating-point atomic adds, and the scatter's LDS
// cursors hand out slots in whatever order the lanes arrive), so this is a pure scheduling step — it has no counterpart in the
// reference, whose batches are NumPy arrays in parent order (ray_tracer.py:491-515).
//
// Three launches: per-block histograms in LDS -> table[bin][block] (+ per-bin totals, one global atomic
// per non-empty (block, bin)), exclusive scan of the table in bin-major order = final order (one block
// per 16 bins; its base is the sum of the totals of the bins before), scatter with LDS cursors.
#pragma once
#include "rt_kernels.hpp"

namespace rt {

constexpr int SORT_BINS = 8192;       // grid cells (13 bits split over the axes by the host; 4096: shade +2 %, 1024: +11 %)
constexpr int SORT_BLOCKS = 256;      // blocks of the count / scatter launches; block b owns one contiguous range
constexpr int SORT_THREADS = 1024;

struct SortGrid {                     // cell of p: clamp((p - lo) * inv, 0, n - 1) per axis, index (x * ny + y) * nz + z
  float lo[3], inv[3];
  int32_t n[3];
  int32_t pad;
};

__device__ __forceinline__ void sort_range(uint32_t count, const FrameParams& fp, uint32_t* begin, uint32_t* end) {
  if (count > fp.hit_cap) count = fp.hit_cap;
  const uint32_t chunk = (count + SORT_BLOCKS - 1) / SORT_BLOCKS;
  const uint32_t b = blockIdx.x * chunk;
  *begin = b < count ? b : count;
  *end = b + chunk < count ? b + chunk : count;
}

__device__ __forceinline__ uint32_t sort_cell(const SortGrid& g, float px, float py, float pz) {
  const int cx = (int)fminf(fmaxf((px - g.lo[0]) * g.inv[0], 0.0f), (float)(g.n[0] - 1));
  const int cy = (int)fminf(fmaxf((py - g.lo[1]) * g.inv[1], 0.0f), (float)(g.n[1] - 1));
  const int cz = (int)fminf(fmaxf((pz - g.lo[2]) * g.inv[2], 0.0f), (float)(g.n[2] - 1));
  return (uint32_t)((cx * g.n[1] + cy) * g.n[2] + cz);
}

__global__ __launch_bounds__(SORT_THREADS) void k_sort_count(DevCamera cam, FrameParams fp, int level, RayQueue q,
                                                             const HitRec* __restrict__ hits, const Counters* C, SortGrid g,
                                                             uint16_t* __restrict__ keys, uint32_t* __restrict__ table,
                                                             uint32_t* __restrict__ bin_total) {
  __shared__ uint32_t s_hist[SORT_BINS];
  for (int b = threadIdx.x; b < SORT_BINS; b += SORT_THREADS) s_hist[b] = 0u;
  __syncthreads();
  uint32_t begin, end;
  sort_range(C->n_hits[level], fp, &begin, &end);
  for (uint32_t i = begin + threadIdx.x; i < end; i += SORT_THREADS) {
    const HitRec hr = hits[i];
    const RayIn r = load_ray(level, hr.slot, cam, fp.rs, q);
    const float px = (float)(r.o.x + hr.t * r.d.x), py = (float)(r.o.y + hr.t * r.d.y), pz = (float)(r.o.z + hr.t * r.d.z);
    // (NaN / inf -> cell 0 or the last one: fminf/fmaxf drop NaNs, the conversion saturates)
    const uint32_t key = sort_cell(g, px, py, pz);
    keys[i] = (uint16_t)key;
    atomicAdd(&s_hist[key], 1u);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < SORT_BINS; b += SORT_THREADS) {
    const uint32_t c = s_hist[b];
    table[(size_t)b * SORT_BLOCKS + blockIdx.x] = c;
    if (c) atomicAdd(&bin_total[b], c);            // bin_total is zeroed by the host before this launch
  }
}

// Exclusive scan of table[SORT_BINS * SORT_BLOCKS] in place, bin-major.  Block g owns SCAN_BINS bins
// (SCAN_BINS * SORT_BLOCKS consecutive entries, one per thread-quad); its base is the sum of bin_total over
// the bins before its first one.
constexpr int SCAN_BINS = 16;
constexpr int SCAN_THREADS = 1024;
__global__ __launch_bounds__(SCAN_THREADS) void k_sort_scan(uint32_t* __restrict__ table, const uint32_t* __restrict__ bin_total) {
  constexpr int PER = SCAN_BINS * SORT_BLOCKS / SCAN_THREADS;        // consecutive entries per thread
  static_assert(PER == 4, "one uint4 per thread");
  __shared__ uint32_t s_sum[SCAN_THREADS];
  __shared__ uint32_t s_base;
  const int first_bin = blockIdx.x * SCAN_BINS;
  // base: totals of the bins before this block's (<= 4096 values, summed by the whole block)
  uint32_t part = 0;
  for (int b = threadIdx.x; b < first_bin; b += SCAN_THREADS) part += bin_total[b];
  s_sum[threadIdx.x] = part;
  __syncthreads();
  for (int o = SCAN_THREADS / 2; o > 0; o >>= 1) {
    if (threadIdx.x < (unsigned)o) s_sum[threadIdx.x] += s_sum[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) s_base = s_sum[0];
  __syncthreads();
  const uint32_t base = s_base;
  __syncthreads();
  uint4* mine = reinterpret_cast<uint4*>(table + (size_t)first_bin * SORT_BLOCKS) + threadIdx.x;
  const uint4 v = *mine;
  const uint32_t sum = (v.x + v.y) + (v.z + v.w);
  s_sum[threadIdx.x] = sum;
  __syncthreads();
  for (int o = 1; o < SCAN_THREADS; o <<= 1) {                       // Hillis-Steele inclusive scan of the sums
    const uint32_t add = threadIdx.x >= (unsigned)o ? s_sum[threadIdx.x - o] : 0u;
    __syncthreads();
    s_sum[threadIdx.x] += add;
    __syncthreads();
  }
  uint32_t run = base + s_sum[threadIdx.x] - sum;
  uint4 o;
  o.x = run; run += v.x; o.y = run; run += v.y; o.z = run; run += v.z; o.w = run;
  *mine = o;
}

__global__ __launch_bounds__(SORT_THREADS) void k_sort_scatter(FrameParams fp, int level, const HitRec* __restrict__ hits,
                                                               const Counters* C, const uint16_t* __restrict__ keys,
                                                               const uint32_t* __restrict__ table, HitRec* __restrict__ out) {
  __shared__ uint32_t s_cur[SORT_BINS];
  for (int b = threadIdx.x; b < SORT_BINS; b += SORT_THREADS) s_cur[b] = table[(size_t)b * SORT_BLOCKS + blockIdx.x];
  __syncthreads();
  uint32_t begin, end;
  sort_range(C->n_hits[level], fp, &begin, &end);
  for (uint32_t i = begin + threadIdx.x; i < end; i += SORT_THREADS) {
    const uint32_t pos = atomicAdd(&s_cur[keys[i]], 1u);
    out[pos] = hits[i];
  }
}

}  // namespace rt
