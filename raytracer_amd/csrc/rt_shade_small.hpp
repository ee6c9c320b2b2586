// rt_shade_small.hpp — fp32 shading kernel for small scenes (the pool.txt / original.txt class:
// up to MAXS spheres, MAXP planes, MAXB boxes), the dominant kernel of the headline benchmark.
//
// Arithmetic plan (see rt_shade_fast.hpp for the general version and the argument why the product
// over all surface crossings equals the reference's shadow walk):
//
//  * One lane per hit; a workgroup owns blocks of 256 consecutive hits of the level's hit queue
//    (static partition: every hit costs the same here, so no work cursor and no atomics).
//  * Phase 1 (fp64, once per hit): hit point, facing normal, offset point p = hp + n*1e-6 and, for
//    EVERY primitive, the origin-dependent terms of its intersection test — sphere: oc = p - c and
//    cc = |oc|^2 - r^2; plane: num = offset - p.N; box: slab offsets min - p, max - p — rounded to
//    fp32 only after the subtraction.  These live in registers for the whole hit (MAXS*4 + MAXP +
//    MAXB*6 VGPRs).  Because they come from the exact fp64 hit point, p sits on the correct side of
//    its own surface (cc = +2e-6*r for the sphere it lies on) and no self-intersection heuristic or
//    epsilon tuning is needed in fp32.
//  * Phase 2 (fp32): per light, the n x n jittered samples are generated one grid row (CH = n
//    samples) at a time and tested against all primitives: primitive-outer, sample-inner, fully
//    unrolled, 4 VGPRs of state per sample (direction + distance), blocked flags as wave masks.
//    Opaque scenes (every material transparency == 0) use an any-hit test with ONE root per
//    sphere: the reference's "t1 if t1 > 1e-6 else t2" picks the far root exactly when the origin
//    is inside the sphere (cc < 0) or within 1e-6 of its surface; the first case is known per
//    (hit, primitive), the second only happens on the hit's own surface with the light behind it,
//    which is handled by the `n.d < 0` rule below.
//  * Phase 3: framebuffer update (plain store at level 0, float atomics deeper) and child rays,
//    appended with one global atomic per workgroup (per-wave ballots combined through LDS).
#pragma once
#include "rt_kernels.hpp"
#include "rt_shade_fast.hpp"

namespace rt {

template <int MAXS, int MAXP, int MAXB, int CH, bool OPAQUE>
__global__ __launch_bounds__(BLOCK) void k_shade_small(DevScene sc, DevCamera cam, FrameParams fp, int level, RayQueue q,
                                                       RayQueue nq, const HitRec* __restrict__ hits, Counters* C,
                                                       float* __restrict__ fb) {
  constexpr int MAXN = MAXS + MAXP + MAXB;
  __shared__ Prim s_prims[MAXN];
  __shared__ uint32_t s_wcnt[BLOCK / 64][2];
  __shared__ uint32_t s_base;
  {
    const int words = sc.n * (int)(sizeof(Prim) / 16);
    const uint4* src = reinterpret_cast<const uint4*>(sc.prims);
    uint4* dst = reinterpret_cast<uint4*>(s_prims);
    for (int i = threadIdx.x; i < words; i += BLOCK) dst[i] = src[i];
    __syncthreads();
  }
  const Prim* P = s_prims;
  const int nS = sc.nS, nP = sc.nP, nB = sc.nB;
  uint32_t count = C->n_hits[level];
  if (count > fp.queue_cap) count = fp.queue_cap;
  const uint32_t n_iters = (count + BLOCK - 1) / BLOCK;
  const int n_sh = fp.n_shadow;
  const int n_rows = n_sh > 1 ? n_sh : 1;                    // CH == n_sh when n_sh > 1, else 1
  const float inv_cells = 1.0f / (float)(n_sh > 1 ? n_sh * n_sh : 1);
  const bool spawn_level = level < fp.max_depth;
  const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;

  for (uint32_t iter = blockIdx.x; iter < n_iters; iter += gridDim.x) {
    const uint32_t h = iter * BLOCK + threadIdx.x;
    const bool active = h < count;
    // ---------------------------------------------------------------- phase 1: per-hit setup ---
    float scx[MAXS], scy[MAXS], scz[MAXS], scc[MAXS];
    float pnum[MAXP > 0 ? MAXP : 1];
    float blx[MAXB > 0 ? MAXB : 1], bly[MAXB > 0 ? MAXB : 1], blz[MAXB > 0 ? MAXB : 1];
    float bhx[MAXB > 0 ? MAXB : 1], bhy[MAXB > 0 ? MAXB : 1], bhz[MAXB > 0 ? MAXB : 1];
    d3 p = mk3(0, 0, 0), hpd = mk3(0, 0, 0);
    f3 nf = mkf(0, 0, 1), v = mkf(0, 0, 1);
    float own_cthr = 0.0f;          // own-surface rule: blocked when n.d < own_cthr
    int mat_idx = 0;
    uint32_t lp = 0, gpix = 0;
    f3 wgt = mkf(0, 0, 0);
    if (active) {
      const HitRec hr = hits[h];
      const RayIn r = load_ray(level, hr.slot, cam, fp.rs, q);
      const Prim& pr = P[hr.prim];
      const d3 n = prim_normal(pr, r.o, r.d, hr.t);
      hpd = mk3(r.o.x + hr.t * r.d.x, r.o.y + hr.t * r.d.y, r.o.z + hr.t * r.d.z);
      p = mk3(hpd.x + n.x * EPS, hpd.y + n.y * EPS, hpd.z + n.z * EPS);
      v = normalize_clamped(mkf((float)(r.o.x - hpd.x), (float)(r.o.y - hpd.y), (float)(r.o.z - hpd.z)));
      nf = tof(n);
      mat_idx = pr.mat;
      lp = r.lp; gpix = global_pixel(fp.rs, lp);
      wgt = tof(r.w);
      // A sphere or box entered through its own surface always blocks (far side); the plane's own
      // surface is decided exactly by its num = -1e-6 below.
      own_cthr = (pr.type == 1) ? -RT_INF : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < MAXS; ++k) {
      scx[k] = scy[k] = scz[k] = 0.0f; scc[k] = 1.0f;          // disc = -1 -> never hit
      if (k < nS) {
        const Prim& s_ = P[k];
        const double ox = p.x - s_.a[0], oy = p.y - s_.a[1], oz = p.z - s_.a[2];
        scx[k] = (float)ox; scy[k] = (float)oy; scz[k] = (float)oz;
        scc[k] = (float)(((ox * ox + oy * oy) + oz * oz) - s_.a[4]);
      }
    }
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
      pnum[k] = 0.0f;
      if (k < nP) {
        const Prim& s_ = P[nS + k];
        pnum[k] = (float)(s_.a[3] - ((p.x * s_.a[0] + p.y * s_.a[1]) + p.z * s_.a[2]));
      }
    }
#pragma unroll
    for (int k = 0; k < MAXB; ++k) {
      blx[k] = bly[k] = blz[k] = 1.0f; bhx[k] = bhy[k] = bhz[k] = -1.0f;
      if (k < nB) {
        const Prim& s_ = P[nS + nP + k];
        blx[k] = (float)(s_.a[0] - p.x); bly[k] = (float)(s_.a[1] - p.y); blz[k] = (float)(s_.a[2] - p.z);
        bhx[k] = (float)(s_.a[3] - p.x); bhy[k] = (float)(s_.a[4] - p.y); bhz[k] = (float)(s_.a[5] - p.z);
      }
    }
    const double* mat = sc.mats + MAT_STRIDE * mat_idx;

    // ------------------------------------------------------------- phase 2: lights & shadows ---
    float dif0 = 0, dif1 = 0, dif2 = 0, spc0 = 0, spc1 = 0, spc2 = 0;
    for (int li = 0; li < sc.nL; ++li) {
      const double* lg = sc.lights + LIGHT_STRIDE * li;
      const f3 tl = mkf((float)(lg[0] - p.x), (float)(lg[1] - p.y), (float)(lg[2] - p.z));
      const f3 ld = scalef(tl, raw_rsq(dotf(tl, tl)));                     // ray_tracer.py:198-213
      const f3 ref = (fabsf(ld.x) >= 0.9f) ? mkf(0, 1, 0) : mkf(1, 0, 0);
      const f3 right = normalize_clamped(crossf(ld, ref));
      const f3 up = normalize_clamped(crossf(right, ld));
      const float radius = (float)lg[8];
      const float cell = radius / (float)n_rows;
      const float half = -0.5f * radius;
      const float cell16 = cell * (1.0f / 65536.0f);
      const uint32_t key = jit_light_key(fp.seed, gpix, (uint32_t)level, (uint32_t)li);
      float total = 0.0f;
      for (int row = 0; row < n_rows; ++row) {
        float dx[CH], dy[CH], dz[CH], md[CH];
        bool blk[CH];
        float tr[OPAQUE ? 1 : CH];
        const float oi_base = fmaf(cell, (float)row, half);
#pragma unroll
        for (int s = 0; s < CH; ++s) {
          f3 ts = tl;
          if (CH > 1) {                                                    // ray_tracer.py:219-235 (CH == n_sh)
            const uint32_t bits = jit_cell_bits(key, (uint32_t)(row * CH + s));
            const float oi = fmaf((float)(bits >> 16), cell16, oi_base);
            const float oj = fmaf((float)(bits & 0xFFFFu), cell16, fmaf(cell, (float)s, half));
            ts = mkf(fmaf(oj, up.x, fmaf(oi, right.x, tl.x)), fmaf(oj, up.y, fmaf(oi, right.y, tl.y)),
                     fmaf(oj, up.z, fmaf(oi, right.z, tl.z)));
          }
          const float d2 = dotf(ts, ts);
          const float rinv = raw_rsq(d2);
          dx[s] = ts.x * rinv; dy[s] = ts.y * rinv; dz[s] = ts.z * rinv;
          md[s] = d2 * rinv;
          if (OPAQUE) blk[s] = fmaf(nf.z, dz[s], fmaf(nf.y, dy[s], nf.x * dx[s])) < own_cthr;
          else { blk[s] = false; tr[s] = 1.0f; }
        }
#pragma unroll
        for (int k = 0; k < MAXS; ++k) {                                   // spheres: sphere.py:49-71
          if (k < nS) {
            const float tau = OPAQUE ? 0.0f : (float)sc.mats[MAT_STRIDE * P[k].mat + 10];
            const float sgn = scc[k] < 0.0f ? 1.0f : -1.0f;               // origin inside -> far root
#pragma unroll
            for (int s = 0; s < CH; ++s) {
              const float b = fmaf(scz[k], dz[s], fmaf(scy[k], dy[s], scx[k] * dx[s]));   // oc.d (a = 1)
              const float sq = raw_sqrt(fmaf(b, b, -scc[k]));                              // NaN on a miss
              if (OPAQUE) {
                const float t = fmaf(sgn, sq, -b);
                blk[s] = blk[s] | ((t > T_LO) & (t < md[s]));              // no short-circuit: keep it branch-free
              } else {
                const float t1 = -b - sq, t2 = sq - b;
                tr[s] *= ((t1 > T_LO) & (t1 < md[s])) ? tau : 1.0f;
                tr[s] *= ((t2 > T_LO) & (t2 < md[s])) ? tau : 1.0f;
              }
            }
          }
        }
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {                                   // planes: infinite_plane.py:30-42
          if (k < nP) {
            const Prim& s_ = P[nS + k];
            const f3 N = mkf((float)s_.a[0], (float)s_.a[1], (float)s_.a[2]);
            const float tau = OPAQUE ? 0.0f : (float)sc.mats[MAT_STRIDE * s_.mat + 10];
#pragma unroll
            for (int s = 0; s < CH; ++s) {
              const float den = fmaf(N.z, dz[s], fmaf(N.y, dy[s], N.x * dx[s]));
              const float t = pnum[k] * raw_rcp(den);                      // den == 0 -> inf/NaN -> no hit
              const bool in = (t > T_LO) & (t < md[s]);
              if (OPAQUE) blk[s] = blk[s] | in;
              else tr[s] *= in ? tau : 1.0f;
            }
          }
        }
#pragma unroll
        for (int k = 0; k < MAXB; ++k) {                                   // boxes: cube.py:76-107 (slab roots)
          if (k < nB) {
            const float tau = OPAQUE ? 0.0f : (float)sc.mats[MAT_STRIDE * P[nS + nP + k].mat + 10];
            const bool inside = (blx[k] < 0.0f) & (bly[k] < 0.0f) & (blz[k] < 0.0f) & (bhx[k] > 0.0f) & (bhy[k] > 0.0f) & (bhz[k] > 0.0f);
#pragma unroll
            for (int s = 0; s < CH; ++s) {
              const float ix = raw_rcp(dx[s]), iy = raw_rcp(dy[s]), iz = raw_rcp(dz[s]);
              const float ax = blx[k] * ix, bx = bhx[k] * ix;
              const float ay = bly[k] * iy, by = bhy[k] * iy;
              const float az = blz[k] * iz, bz = bhz[k] * iz;
              const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
              const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
              const bool ok = tn <= tf;
              if (OPAQUE) {
                const float t = inside ? tf : tn;
                blk[s] = blk[s] | (ok & (t > T_LO) & (t < md[s]));
              } else {
                tr[s] *= (ok & (tn > T_LO) & (tn < md[s])) ? tau : 1.0f;
                tr[s] *= (ok & (tf > T_LO) & (tf < md[s])) ? tau : 1.0f;
              }
            }
          }
        }
#pragma unroll
        for (int s = 0; s < CH; ++s) total += OPAQUE ? (blk[s] ? 0.0f : 1.0f) : tr[s];
      }
      const float T = total * inv_cells;
      const float I = fmaf((float)lg[7], T, 1.0f - (float)lg[7]);                         // ray_tracer.py:543
      const f3 l = normalize_clamped(mkf((float)(lg[0] - hpd.x), (float)(lg[1] - hpd.y), (float)(lg[2] - hpd.z)));
      const float ndl = fmaxf(dotf(nf, l), 0.0f);
      const float s1 = I * ndl;
      dif0 = fmaf((float)mat[0] * (float)lg[3], s1, dif0);
      dif1 = fmaf((float)mat[1] * (float)lg[4], s1, dif1);
      dif2 = fmaf((float)mat[2] * (float)lg[5], s1, dif2);
      const float dn = -dotf(l, nf);                                                      // (-l).n
      const f3 rf = mkf(fmaf(-2.0f * dn, nf.x, -l.x), fmaf(-2.0f * dn, nf.y, -l.y), fmaf(-2.0f * dn, nf.z, -l.z));
      const float rdv = fmaxf(dotf(rf, v), 0.0f);
      const float shin = (float)mat[9];
      const float pw = (shin == 0.0f) ? 1.0f : (rdv > 0.0f ? powf(rdv, shin) : 0.0f);     // np.power(0, 0) == 1
      const float s2 = I * pw * (float)lg[6];
      spc0 = fmaf((float)mat[3] * (float)lg[3], s2, spc0);
      spc1 = fmaf((float)mat[4] * (float)lg[4], s2, spc1);
      spc2 = fmaf((float)mat[5] * (float)lg[5], s2, spc2);
    }

    // ------------------------------------------------- phase 3: accumulate, spawn children -----
    const float tau_m = (float)mat[10];
    if (active) {
      const float one_m = 1.0f - tau_m;
      const float c0 = wgt.x * ((dif0 + spc0) * one_m), c1 = wgt.y * ((dif1 + spc1) * one_m), c2 = wgt.z * ((dif2 + spc2) * one_m);
      float* dst = fb + 3 * (size_t)lp;
      if (level == 0) { dst[0] = c0; dst[1] = c1; dst[2] = c2; }
      else { atomicAdd(dst + 0, c0); atomicAdd(dst + 1, c1); atomicAdd(dst + 2, c2); }
    }
    if (spawn_level) {                                                     // ray_tracer.py:565-595
      const bool needT = active && (mat[10] > EPS);
      const bool needR = active && (mat[6] > EPS || mat[7] > EPS || mat[8] > EPS);
      const unsigned long long mT = __ballot(needT), mR = __ballot(needR);
      if (lane == 0) { s_wcnt[wave][0] = (uint32_t)__popcll(mT); s_wcnt[wave][1] = (uint32_t)__popcll(mR); }
      __syncthreads();
      if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < BLOCK / 64; ++w) tot += s_wcnt[w][0] + s_wcnt[w][1];
        s_base = tot ? atomicAdd(&C->n_rays[level + 1], tot) : 0u;
      }
      __syncthreads();
      uint32_t base = s_base;
      for (uint32_t w = 0; w < wave; ++w) base += s_wcnt[w][0] + s_wcnt[w][1];
      const uint32_t preT = __builtin_amdgcn_mbcnt_hi((uint32_t)(mT >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mT, 0u));
      const uint32_t preR = __builtin_amdgcn_mbcnt_hi((uint32_t)(mR >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mR, 0u));
      const uint32_t sT = base + preT, sR = base + (uint32_t)__popcll(mT) + preR;
      if (needT || needR) {
        // recompute the fp64 hit geometry (cheaper than keeping ~30 VGPRs live through phase 2)
        const HitRec hr = hits[h];
        const RayIn r = load_ray(level, hr.slot, cam, fp.rs, q);
        const d3 n = prim_normal(P[hr.prim], r.o, r.d, hr.t);
        const d3 hp = mk3(r.o.x + hr.t * r.d.x, r.o.y + hr.t * r.d.y, r.o.z + hr.t * r.d.z);
        if (needT) {
          if (sT < fp.queue_cap) {
            nq.ox[sT] = hp.x - n.x * EPS; nq.oy[sT] = hp.y - n.y * EPS; nq.oz[sT] = hp.z - n.z * EPS;
            nq.dx[sT] = r.d.x; nq.dy[sT] = r.d.y; nq.dz[sT] = r.d.z;
            nq.wx[sT] = r.w.x * mat[10]; nq.wy[sT] = r.w.y * mat[10]; nq.wz[sT] = r.w.z * mat[10];
            nq.pix[sT] = r.lp;
          } else C->overflow = 1u;
        }
        if (needR) {
          if (sR < fp.queue_cap) {
            const double dn = dot3(r.d, n);                                // reflect_batch, :34-36
            nq.dx[sR] = r.d.x - (2 * dn) * n.x; nq.dy[sR] = r.d.y - (2 * dn) * n.y; nq.dz[sR] = r.d.z - (2 * dn) * n.z;
            nq.ox[sR] = hp.x + n.x * EPS; nq.oy[sR] = hp.y + n.y * EPS; nq.oz[sR] = hp.z + n.z * EPS;
            nq.wx[sR] = r.w.x * mat[6]; nq.wy[sR] = r.w.y * mat[7]; nq.wz[sR] = r.w.z * mat[8];
            nq.pix[sR] = r.lp;
          } else C->overflow = 1u;
        }
      }
      __syncthreads();   // s_wcnt / s_base are reused by the next iteration
    }
  }
}

#ifndef RT_NO_LAUNCHERS   // (single-kernel compile harness: tools/one_kernel.sh)
// Can this (scene, frame) be served by a k_shade_small instantiation?  Returns false -> use
// k_shade_fast.  Instantiated shapes: (8 spheres, 2 planes, 0 boxes) and (4, 2, 4); shadow grids
// n <= 1 and n in {2, 3, 4, 5, 8}.
template <int MAXS, int MAXP, int MAXB>
static bool launch_small_shape(const DevScene& ds, bool opaque, const DevCamera& cam, const FrameParams& fp, int level,
                               RayQueue cur, RayQueue nxt, const HitRec* hits, Counters* C, float* fb, int grid,
                               hipStream_t st) {
  const int n = fp.n_shadow;
#define RT_SMALL(CHV)                                                                                              \
  do {                                                                                                             \
    if (opaque) hipLaunchKernelGGL((k_shade_small<MAXS, MAXP, MAXB, CHV, true>), dim3(grid), dim3(BLOCK), 0, st, ds, \
                                   cam, fp, level, cur, nxt, hits, C, fb);                                           \
    else hipLaunchKernelGGL((k_shade_small<MAXS, MAXP, MAXB, CHV, false>), dim3(grid), dim3(BLOCK), 0, st, ds, cam,  \
                            fp, level, cur, nxt, hits, C, fb);                                                       \
    return true;                                                                                                   \
  } while (0)
  if (n <= 1) RT_SMALL(1);
  switch (n) {
    case 2: RT_SMALL(2);
    case 3: RT_SMALL(3);
    case 4: RT_SMALL(4);
    case 5: RT_SMALL(5);
    case 8: RT_SMALL(8);
    default: return false;
  }
#undef RT_SMALL
}

static bool launch_shade_small(const DevScene& ds, bool opaque, const DevCamera& cam, const FrameParams& fp, int level,
                               RayQueue cur, RayQueue nxt, const HitRec* hits, Counters* C, float* fb, int grid,
                               hipStream_t st) {
  if (ds.nB == 0 && ds.nS <= 8 && ds.nP <= 2)
    return launch_small_shape<8, 2, 0>(ds, opaque, cam, fp, level, cur, nxt, hits, C, fb, grid, st);
  if (ds.nS <= 4 && ds.nP <= 2 && ds.nB <= 4)
    return launch_small_shape<4, 2, 4>(ds, opaque, cam, fp, level, cur, nxt, hits, C, fb, grid, st);
  return false;
}

#endif  // RT_NO_LAUNCHERS

}  // namespace rt
