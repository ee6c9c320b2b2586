// rt_stream.hpp — rendering with the reference's own jitter stream (rt_jitter_mode NUMPY_STREAM).
//
// `np.random.seed(s); render_vectorized(...)` consumes the legacy MT19937 stream in an order that
// depends on the reference's batch structure: batches are processed depth-first
// from a LIFO stack (reflection child before transparency child, ray_tracer.py:573-595); in every
// batch with hits, for each light, for each grid cell (i outer, j inner), `random(H)` is drawn for
// u_i and then for u_j (ray_tracer.py:219-222), element m belonging to the m-th hit in ascending
// pixel order.  To reproduce the reference's image bit-for-bit in its jitter this path therefore
//   * keeps the reference's batches (no merging per depth) and processes them in the same order,
//   * compacts hits and children ORDER-PRESERVING (flag -> exclusive scan -> scatter),
//   * regenerates the stream on the host (same generator, rt_api.hip) and uploads, per batch, the
//     table  tab[((light*cells + cell)*2 + axis)*H + m]  — which is simply the next
//     lights*cells*2*H doubles of the stream in draw order.
// It costs a host round-trip per batch and is meant for parity, not speed; fp64 only.  Within a
// batch a pixel occurs at most once, so framebuffer updates are plain read-modify-writes and the
// accumulation order per pixel equals the reference's.
#pragma once
#include "rt_kernels.hpp"

namespace rt {

// Nearest hit for every ray of a batch (ROOT: camera rays of the row set).  rec[i] keeps ray i's
// result, flag[i] = 1 for a hit; misses add weight*background (ray_tracer.py:491-502).
template <bool ROOT>
__global__ __launch_bounds__(BLOCK) void k_sa_trace(DevScene sc, DevCamera cam, FrameParams fp, RayQueue q, uint32_t M,
                                                    HitRec* __restrict__ rec, uint32_t* __restrict__ flag,
                                                    double* __restrict__ fb) {
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= M) return;
  const RayIn r = load_ray(ROOT ? 0 : 1, i, cam, fp.rs, q);
  int k; const double t = nearest(sc.prims, sc.nS, sc.nP, sc.nB, r.o, r.d, &k, HIT_EPS, &sc.ac);
  HitRec h; h.slot = i; h.prim = k; h.t = t;
  rec[i] = h;
  flag[i] = k >= 0 ? 1u : 0u;
  if (k < 0) {
    double* dst = fb + 3 * (size_t)r.lp;
    dst[0] += r.w.x * sc.bg[0]; dst[1] += r.w.y * sc.bg[1]; dst[2] += r.w.z * sc.bg[2];
  }
}

__global__ __launch_bounds__(BLOCK) void k_sa_scatter_hits(const HitRec* __restrict__ rec, const uint32_t* __restrict__ flag,
                                                           const uint32_t* __restrict__ off, uint32_t M,
                                                           HitRec* __restrict__ hits) {
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i < M && flag[i]) hits[off[i]] = rec[i];
}

// Shade the ordered hits of one batch with table jitter; children go to slot h of dense temporaries
// (compacted afterwards in order).  ray_tracer.py:508-595.
template <bool ROOT>
__global__ __launch_bounds__(BLOCK) void k_sa_shade(DevScene sc, DevCamera cam, FrameParams fp, int depth, RayQueue q,
                                                    const HitRec* __restrict__ hits, uint32_t H,
                                                    const double* __restrict__ table, int n_cells, RayQueue tmpT,
                                                    RayQueue tmpR, uint32_t* __restrict__ flagT,
                                                    uint32_t* __restrict__ flagR, double* __restrict__ fb,
                                                    unsigned long long* queries_out) {
  const uint32_t h = blockIdx.x * BLOCK + threadIdx.x;
  if (h >= H) return;
  const HitRec hr = hits[h];
  const RayIn r = load_ray(ROOT ? 0 : 1, hr.slot, cam, fp.rs, q);
  JitTable jit; jit.tab = table; jit.H = H; jit.h = h; jit.n_cells = n_cells;
  int queries = 0;
  const ExactHit e = shade_hit_exact(sc, sc.prims, r, hr, fp.n_shadow, jit, &queries, false, HIT_EPS, &sc.ac);
  double* dst = fb + 3 * (size_t)r.lp;
  dst[0] += r.w.x * e.c[0]; dst[1] += r.w.y * e.c[1]; dst[2] += r.w.z * e.c[2];           // :563
  atomicAdd(queries_out, (unsigned long long)queries);
  const bool deeper = depth < fp.max_depth;
  const double* mat = e.mat;
  const bool needT = deeper && (mat[10] > EPS);                                            // :565-566
  const bool needR = deeper && (mat[6] > EPS || mat[7] > EPS || mat[8] > EPS);             // :581-582
  flagT[h] = needT ? 1u : 0u;
  flagR[h] = needR ? 1u : 0u;
  if (needT) {
    tmpT.ox[h] = e.hp.x - e.n.x * EPS; tmpT.oy[h] = e.hp.y - e.n.y * EPS; tmpT.oz[h] = e.hp.z - e.n.z * EPS;
    tmpT.dx[h] = r.d.x; tmpT.dy[h] = r.d.y; tmpT.dz[h] = r.d.z;
    tmpT.wx[h] = r.w.x * mat[10]; tmpT.wy[h] = r.w.y * mat[10]; tmpT.wz[h] = r.w.z * mat[10];
    tmpT.pix[h] = r.lp;
  }
  if (needR) {
    const double dn = dot3(r.d, e.n);                                                      // reflect_batch :34-36
    tmpR.dx[h] = r.d.x - (2 * dn) * e.n.x; tmpR.dy[h] = r.d.y - (2 * dn) * e.n.y; tmpR.dz[h] = r.d.z - (2 * dn) * e.n.z;
    tmpR.ox[h] = e.hp.x + e.n.x * EPS; tmpR.oy[h] = e.hp.y + e.n.y * EPS; tmpR.oz[h] = e.hp.z + e.n.z * EPS;
    tmpR.wx[h] = r.w.x * mat[6]; tmpR.wy[h] = r.w.y * mat[7]; tmpR.wz[h] = r.w.z * mat[8];
    tmpR.pix[h] = r.lp;
  }
}

__global__ __launch_bounds__(BLOCK) void k_sa_scatter_rays(RayQueue tmp, const uint32_t* __restrict__ flag,
                                                           const uint32_t* __restrict__ off, uint32_t H, RayQueue out) {
  const uint32_t h = blockIdx.x * BLOCK + threadIdx.x;
  if (h >= H || !flag[h]) return;
  const uint32_t o = off[h];
  out.ox[o] = tmp.ox[h]; out.oy[o] = tmp.oy[h]; out.oz[o] = tmp.oz[h];
  out.dx[o] = tmp.dx[h]; out.dy[o] = tmp.dy[h]; out.dz[o] = tmp.dz[h];
  out.wx[o] = tmp.wx[h]; out.wy[o] = tmp.wy[h]; out.wz[o] = tmp.wz[h];
  out.pix[o] = tmp.pix[h];
}

}  // namespace rt
