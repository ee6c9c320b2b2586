// rt_shade_fast.hpp — the fp32 shading kernel (precision = RT_FP32), the dominant kernel of a frame.
//
// Same inputs/outputs as k_shade_exact, different arithmetic plan (results within the fp32-mode
// tolerance of BASELINE.json: <= 1/255 per 8-bit channel):
//
//  * One lane per hit.  Per light the n x n shadow samples are processed in register-resident
//    chunks of CH samples; the loop nest is  light -> chunk -> primitive -> sample  so that
//    everything that depends on (hit, primitive) is computed once per chunk and reused CH times.
//  * Robustness without fp64 in the inner loop: the quantities in which the reference's 1e-6
//    offsets live — origin-to-centre vector and |oc|^2 - r^2 of a sphere, signed plane distance,
//    slab offsets of a box — are formed in fp64 from the fp64 hit point and only then rounded to
//    fp32.  The shadow origin p = hit + n*1e-6 therefore sits on the correct side of its own
//    surface exactly as in the reference, with no self-intersection heuristics.
//  * The shadow walk of the reference (compute_shadow_transmission_batch, ray_tracer.py:247-293:
//    nearest hit, multiply by the blocker's transparency, step past it, repeat) visits, in order,
//    every surface crossing of the segment; its result is the product of the transparencies of
//    all crossings.  The fast kernel forms that product directly from both roots of every
//    primitive: crossing parameter t' measured from p is in range iff 2e-6 < t' < dist (the
//    reference starts at p + dir*1e-6, accepts t > 1e-6 and stops at dist - 1e-6).
//    The walk stops after 20 crossings (ray_tracer.py:257): scenes with transparency count, per (hit, light, chunk),
//    the primitives a sample can cross (an upper bound of every sample's crossings) and leave pairs that may exceed 20
//    to the exact re-test.  Crossings within 2e-6 of each other count once in the walk: systematic only where surfaces of two
//    primitives coincide, which scene creation marks (Prim::pad) — pairs crossing a marked primitive go to the exact
//    re-test as well.  The early stop below 1e-6 is below the fp32 tolerance.
#pragma once
#include "rt_kernels.hpp"

namespace rt {

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mkf(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 tof(d3 v) { return mkf((float)v.x, (float)v.y, (float)v.z); }
__device__ __forceinline__ float dotf(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ f3 crossf(f3 a, f3 b) {
  return mkf(fmaf(a.y, b.z, -a.z * b.y), fmaf(a.z, b.x, -a.x * b.z), fmaf(a.x, b.y, -a.y * b.x));
}
__device__ __forceinline__ f3 scalef(f3 a, float s) { return mkf(a.x * s, a.y * s, a.z * s); }
// Raw hardware transcendentals (1 ulp, no denormal range fix-ups: ~1 instruction instead of ~7).
__device__ __forceinline__ float raw_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }   // v_sqrt_f32, NaN for x < 0
__device__ __forceinline__ float raw_rcp(float x) { return __builtin_amdgcn_rcpf(x); }     // v_rcp_f32
__device__ __forceinline__ float raw_rsq(float x) { return __builtin_amdgcn_rsqf(x); }     // v_rsq_f32
__device__ __forceinline__ f3 normalize_clamped(f3 a) {   // v / max(|v|, 1e-6)
  float n2 = dotf(a, a);
  float inv = n2 > 1e-12f ? raw_rsq(n2) : 1e6f;
  return scalef(a, inv);
}

constexpr float T_LO = 2e-6f;   // see header: 1e-6 origin shift + 1e-6 hit threshold

constexpr int LEAF_LIST = 20, CAND_LIST = 16;   // per-lane lists of the large-scene walk (entries, in LDS)
// (WAVE_LEAF_CAP, rt_device.hpp: leaves a (wave, light) list holds; more: the lanes walk the hierarchy themselves)
constexpr int WAVE_REC_CAP = 128;     // records a (wave, light) list holds; more: the lanes test their leaves' bounds themselves     // leaves a (wave, light) list holds; more: the lanes walk the hierarchy themselves

// Scenes with transparency: the crossings of ONE marked primitive (Prim::pad bit 2, rt_api.hip mark_near_pairs) by a
// shadow segment — direction (dx, dy, dz), unit to a few ulp, from the exact shadow origin p — with a bound on the fp32
// error of each root.  Used only to find two DIFFERENT primitives' crossings closer than the 2e-6 the reference's walk
// needs to tell them apart (ray_tracer.py:279-291); t < 0: no such root.  Same arithmetic as the sample blocks above.
struct PairRoots { float t0, t1, e0, e1; };
__device__ __forceinline__ PairRoots pair_roots(const Prim& s_, d3 p, float dx, float dy, float dz) {
  constexpr float U8 = 12.0f * 6.0e-8f, U1 = 6.0e-8f;
  PairRoots r; r.t0 = r.t1 = -1.0f; r.e0 = r.e1 = 0.0f;
  if (s_.type == 0) {
    const double ocx = p.x - s_.a[0], ocy = p.y - s_.a[1], ocz = p.z - s_.a[2];
    const float cc = (float)(((ocx * ocx + ocy * ocy) + ocz * ocz) - s_.a[4]);
    const float r_ = (float)s_.a[3];
    const float ocn = raw_sqrt(fmaxf(cc + r_ * r_, 0.0f));
    const float e_b = U8 * ocn, kap = 4.0f * e_b, kap0 = 4.0f * e_b * e_b + 4.0f * U1 * fabsf(cc);
    const float b = fmaf((float)ocz, dz, fmaf((float)ocy, dy, (float)ocx * dx));
    const float disc = fmaf(b, b, -cc);
    const float bound = fmaf(kap, fabsf(b), kap0);
    if (disc > bound) {
      const float sq = raw_sqrt(disc);
      const float t1 = b < 0.0f ? sq - b : -b - sq;       // the root that does not cancel
      const float e1 = e_b + bound * raw_rcp(sq) + 4.0f * U1 * fabsf(t1);
      const float t2 = cc * raw_rcp(t1);
      r.t0 = t1; r.e0 = e1;
      r.t1 = t2; r.e1 = fabsf(t2) * (e1 * raw_rcp(fabsf(t1)) + 4.0f * U1);
    } else if (disc >= -bound) {
      // a grazing sample (its own hit / miss decision is made in fp64 by the sample block): if the sphere is crossed at
      // all, both crossings lie within sqrt(|disc| + bound) of the point of closest approach
      r.t0 = -b; r.e0 = e_b + raw_sqrt(2.0f * bound) + 4.0f * U1 * fabsf(b);
    }
  } else if (s_.type == 1) {
    const float num = (float)(s_.a[3] - ((p.x * s_.a[0] + p.y * s_.a[1]) + p.z * s_.a[2]));
    const float den = fmaf((float)s_.a[2], dz, fmaf((float)s_.a[1], dy, (float)s_.a[0] * dx));
    if (fabsf(den) > 2.0f * U8) {
      const float t = num * raw_rcp(den);
      r.t0 = t; r.e0 = fabsf(t) * (1.5f * U8 * raw_rcp(fabsf(den)) + 4.0f * U1);
    } else { r.t0 = 1e-30f; r.e0 = 3.0e38f; }             // parallel within rounding: nothing is known about the root (t0: any value inside (0, segment length), see pair_close)
  } else {
    const float jx = raw_rcp(dx), jy = raw_rcp(dy), jz = raw_rcp(dz);
    const float ax = (float)(s_.a[0] - p.x) * jx, bx = (float)(s_.a[3] - p.x) * jx;
    const float ay = (float)(s_.a[1] - p.y) * jy, by = (float)(s_.a[4] - p.y) * jy;
    const float az = (float)(s_.a[2] - p.z) * jz, bz = (float)(s_.a[5] - p.z) * jz;
    const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    const float jm = fmaxf(fmaxf(fabsf(jx), fabsf(jy)), fabsf(jz));
    const float rel = 2.0f * fmaf(U8, jm, 4.0f * U1);
    // a component nearly parallel to a slab: nothing is known about the roots -> "close to everything".  (The stand-in root must
    // lie inside (0, segment length) for pair_close to look at it: it used to be 1.0, beyond any light closer than one unit —
    // scaled scene 1501416: a box's exit and a sphere's crossing 2e-7 apart on such a segment counted twice, 5 levels off)
    if (jm > 1e3f) { r.t0 = 1e-30f; r.e0 = 3.0e38f; }
    else if (tn <= tf + rel * (fabsf(tn) + fabsf(tf))) {
      r.t0 = tn; r.e0 = rel * fabsf(tn);
      r.t1 = tf; r.e1 = rel * fabsf(tf);
    }
  }
  return r;
}
// Do two primitives' crossings of one segment (length md) come closer than the walk's 2e-6 (+ 25 % and the fp32 bounds)?
__device__ __forceinline__ bool pair_close(const PairRoots& a, const PairRoots& b, float md) {
  const float top = fmaf(md, 1.0001f, 1e-5f);
  auto one = [&](float ta, float ea, float tb, float eb) __attribute__((always_inline)) {
    return (ta > 0.0f) & (tb > 0.0f) & (ta < top) & (tb < top) & (fabsf(ta - tb) <= 2.5e-6f + ea + eb);
  };
  return one(a.t0, a.e0, b.t0, b.e0) | one(a.t0, a.e0, b.t1, b.e1) | one(a.t1, a.e1, b.t0, b.e0) | one(a.t1, a.e1, b.t1, b.e1);
}

// r.v > 0 of the sequential renderer's specular term, formed like phong_light_exact does (rt_kernels.hpp) from the exact hit
// geometry.
__device__ __forceinline__ bool seq_specular_step(d3 o, d3 hp, d3 n, double lx, double ly, double lz) {
  d3 vd = mk3(o.x - hp.x, o.y - hp.y, o.z - hp.z);
  double vn = norm3(vd); if (vn < EPS) vn = 1.0;                  // normalize, ray_tracer.py:17-21: shorter than EPS stays as it is
  vd = mk3(vd.x / vn, vd.y / vn, vd.z / vn);
  const d3 tld = mk3(lx - hp.x, ly - hp.y, lz - hp.z);
  double ln = norm3(tld); if (ln < EPS) ln = 1.0;
  const d3 mld = mk3(-(tld.x / ln), -(tld.y / ln), -(tld.z / ln));
  const double dnd = dot3(mld, n);
  const d3 rfd = mk3(mld.x - (2 * dnd) * n.x, mld.y - (2 * dnd) * n.y, mld.z - (2 * dnd) * n.z);
  return dot3(rfd, vd) > 0;
}

template <int CH, bool OPAQUE, bool HAS_BOX, bool LDS_SCENE>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu((!LDS_SCENE && CH <= 9) ? 2 : 1))) void k_shade_fast(DevScene sc, DevCamera cam, FrameParams fp, int level, RayQueue q,
                                                      RayQueue nq, const HitRec* __restrict__ hits, Counters* C,
                                                      float* __restrict__ fb, int n_chunks, FixRec* __restrict__ fix) {
  __shared__ Prim s_prims[LDS_SCENE ? LDS_PRIMS : 1];
  __shared__ BvhNode s_nodes[LDS_SCENE ? 1 : LDS_NODES];
  __shared__ int s_wleaves[LDS_SCENE ? 1 : (BLOCK / 64) * WAVE_LEAF_CAP];
  __shared__ uint32_t s_leaf[LDS_SCENE ? 1 : LEAF_LIST * BLOCK], s_cand[LDS_SCENE ? 1 : CAND_LIST * BLOCK];
  int* const wl = s_wleaves + (LDS_SCENE ? 0 : (threadIdx.x >> 6) * WAVE_LEAF_CAP);
  __shared__ int s_wrecs[LDS_SCENE ? 1 : (BLOCK / 64) * WAVE_REC_CAP];
  int* const wr = s_wrecs + (LDS_SCENE ? 0 : (threadIdx.x >> 6) * WAVE_REC_CAP);
  // beside each listed record: (Lp - centre, R0) of its bound for this light — the hit-independent part of the lanes' fan
  // test (fan_may_touch), formed once by the lane that lists the record.  The lanes' filter loop then reads LDS only: with
  // the bound fetched from global memory per (lane, record) every step of that loop waited for a load whose address came
  // from an LDS read (~1 us per listed record and wave; 25 % of the kernel's wave cycles)
  __shared__ float4 s_wrb[LDS_SCENE ? 1 : (BLOCK / 64) * WAVE_REC_CAP];
  float4* const wrb = s_wrb + (LDS_SCENE ? 0 : (threadIdx.x >> 6) * WAVE_REC_CAP);
  const Prim* P = stage_scene<LDS_SCENE>(sc, s_prims);
  const Accel ac = stage_accel<LDS_SCENE>(sc, s_nodes);
  const float scene_m = sc.scene_m;   // scales the fp32 slack of the node tests
  uint32_t count = C->n_hits[level];
  if (count > fp.hit_cap) count = fp.hit_cap;
  const uint32_t lane = __lane_id();
  const int n_sh = fp.n_shadow;
  const int n_cells = n_sh > 1 ? n_sh * n_sh : 1;
  const float inv_cells = 1.0f / (float)n_cells;
  const long long ck_kernel = (RT_DBG(fp) & 2048u) ? clock64() : 0;
  // audit build, RT_DEBUG_STATS bit 26: wave cycles by section of the kernel -> g_fast_cyc (0 per-hit set-up, 1 light set-up,
  // 2 wave-level lists, 3 sample directions, 4 planes, 5 hierarchy phases + sample blocks, 6 deferred fp64 decisions,
  // 7 merged-crossing check, 8 Phong + re-test queue, 9 accumulate + spawn + chunk grab, 10 all)
#ifdef RT_DEBUG_KERNEL
  const bool fprof = (RT_DBG(fp) & (1u << 26)) != 0u;
  long long fck0 = 0, fck1 = 0, fck2 = 0, fck3 = 0, fck4 = 0, fck5 = 0, fck6 = 0, fck7 = 0, fck8 = 0, fck9 = 0;
  long long fck_t = fprof ? clock64() : 0;
  const long long fck_begin = fck_t;
#define FCK(v) do { if (fprof) { const long long t__ = clock64(); v += t__ - fck_t; fck_t = t__; } } while (0)
#else
#define FCK(v) do { } while (0)
#endif
  for (;;) {
    uint32_t base = grab64(&C->work_shade[level]);
    if (base >= count) {
      if ((RT_DBG(fp) & 2048u) && lane == 0) atomicAdd(&C->dbg[0], (unsigned long long)(clock64() - ck_kernel));
      break;
    }
    FCK(fck9);
    uint32_t h = base + lane;
    bool active = h < count;
    RayIn r; d3 hp = mk3(0, 0, 0), n = mk3(0, 0, 0);
    const double* mat = sc.mats;
    r.d = r.o = r.w = mk3(0, 0, 0); r.lp = 0;
    {   // the idle lanes of a level's last wave shade a copy of its last hit (results dropped): the
        // wave-level steps below need all 64 lanes
      HitRec hr = hits[active ? h : count - 1];
      r = load_ray(level, hr.slot, cam, fp.rs, q);
      const Prim& pr = P[hr.prim];
      const double t = hr.t;
      n = prim_normal(pr, r.o, r.d, t);
      hp = mk3(r.o.x + t * r.d.x, r.o.y + t * r.d.y, r.o.z + t * r.d.z);
      mat = sc.mats + MAT_STRIDE * pr.mat;
      const f3 v = normalize_clamped(mkf((float)(r.o.x - hp.x), (float)(r.o.y - hp.y), (float)(r.o.z - hp.z)));
      const f3 nf = tof(n);
      const uint32_t gpix = global_pixel(fp.rs, r.lp);
      const d3 p = mk3(hp.x + n.x * EPS, hp.y + n.y * EPS, hp.z + n.z * EPS);
      float dif[3] = {0, 0, 0}, spc[3] = {0, 0, 0};
      // Large scene: ball (cw, rho) around the wave's hits.  A wave's hits are compact (8x8 pixel tile, or
      // neighbours in the queue), so the hierarchy is first scanned ONCE per (wave, light) with the hull
      // of all its shadow fans (bvh_wave_leaves); the lanes then test only the leaves that survive.
      const bool coop = !LDS_SCENE && ac.n_nodes > 0 && !(RT_DBG(fp) & 64u) && !fp.brute;
      f3 cw = mkf(0, 0, 0); float rho = 0.0f;
      if (coop) {
        const f3 pf = tof(p);
        f3 lo = pf, hi = pf;
#pragma unroll
        for (int o_ = 32; o_ > 0; o_ >>= 1) {
          lo.x = fminf(lo.x, __shfl_xor(lo.x, o_)); lo.y = fminf(lo.y, __shfl_xor(lo.y, o_)); lo.z = fminf(lo.z, __shfl_xor(lo.z, o_));
          hi.x = fmaxf(hi.x, __shfl_xor(hi.x, o_)); hi.y = fmaxf(hi.y, __shfl_xor(hi.y, o_)); hi.z = fmaxf(hi.z, __shfl_xor(hi.z, o_));
        }
        cw = mkf(0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z));
        const f3 dw = mkf(pf.x - cw.x, pf.y - cw.y, pf.z - cw.z);
        float r2 = dotf(dw, dw);
#pragma unroll
        for (int o_ = 32; o_ > 0; o_ >>= 1) r2 = fmaxf(r2, __shfl_xor(r2, o_));
        rho = raw_sqrt(r2) * 1.0001f;
      }
      FCK(fck0);
      for (int li = 0; li < sc.nL; ++li) {
        const double* lg = sc.lights + LIGHT_STRIDE * li;
        const f3 tl = mkf((float)(lg[0] - p.x), (float)(lg[1] - p.y), (float)(lg[2] - p.z));
        // light-facing basis (ray_tracer.py:198-213)
        const f3 ld = scalef(tl, raw_rsq(dotf(tl, tl)));
        bool sw = fabsf(ld.x) >= 0.9f;
        if (__any(fabsf(fabsf(ld.x) - 0.9f) < 1e-5f)) {                      // within rounding of the switch: decide it in fp64
          if (fabsf(fabsf(ld.x) - 0.9f) < 1e-5f) sw = basis_switch_exact(sc, p, li);
        }
        const f3 ref = sw ? mkf(0, 1, 0) : mkf(1, 0, 0);
        const f3 right = normalize_clamped(crossf(ld, ref));
        const f3 up = normalize_clamped(crossf(right, ld));
        const float radius = (float)lg[8];
        const float cell = radius / (float)(n_sh > 1 ? n_sh : 1);
        const float half = -0.5f * radius;
        const JitKey key = jit_light_key(fp.seed, gpix, (uint32_t)level, (uint32_t)li);
        float total = 0.0f;
        // Exactness of the fp32 decisions (see rt_shade_ls.hpp): unc >= 0 after the sample blocks means that some
        // hit / miss decision of this (hit, light) lay within the error bound of its fp32 arithmetic; the pair is
        // then shaded with T = 0 and queued for the fp64 re-test (k_fix_exact).  Bounds: the unit directions carry
        // <= 8 ulp (absolute), oc / cc / num / slab offsets are rounded from fp64 (1 ulp, relative).
        float unc = -1.0f;
        constexpr float U8 = 12.0f * 6.0e-8f;
        const float RLx = n_sh > 1 ? 0.7072f * radius * 1.001f : 0.0f;
        const float T0x = dotf(tl, tl);
        // the light within 3 sample-square radii of the hit
        if ((T0x < 9.0f * RLx * RLx) | !(T0x > 1e-12f)) unc = 1.0f;
        if (RT_DBG(fp) & 16777216u) unc = 1.0f;
        // reach of the sample square + the error of a shadow segment's far end (grows with the segment's length)
        const float reach = RLx + 1e-5f * (1.0f + fabsf((float)lg[0]) + fabsf((float)lg[1]) + fabsf((float)lg[2])) + 2.0f * U8 * raw_sqrt(T0x);
        // LDS-staged scenes (<= 128 records): conservative candidate set of this (wave, light) and the
        // fully-lit / umbra shortcut (rt_device.hpp, fan_cull) — the sample blocks below run only for
        // records whose bound can touch some lane's shadow fan.
        PrimMask cand; cand.lo = cand.hi = ~0ull;
        bool lane_lit = false, lane_dark = false, skip_samples = false;
        if (LDS_SCENE) {
          const FanCull fc = fan_cull(sc, P, p, mk3(lg[0], lg[1], lg[2]), n_sh > 1 ? 0.7072f * radius : 0.0f);
          cand = fc.act; lane_lit = !fc.lane_can; lane_dark = fc.lane_full;
          skip_samples = !(RT_DBG(fp) & 2u) && !__any(fc.lane_can & !fc.lane_full);
          if (fp.brute) { cand.lo = cand.hi = ~0ull; lane_lit = lane_dark = false; skip_samples = false; }   // RT_FLAG_BRUTE_FORCE
          if (RT_DBG(fp) & 16777216u) { lane_lit = lane_dark = false; skip_samples = false; }   // audit build, bit 24: every pair through the exact walk
        }
        if (skip_samples) total = lane_dark ? 0.0f : (float)n_cells;        // n^2 terms of exactly 0 or 1
        int n_wl = -1, n_wr = -1;
        FCK(fck1);
        if (coop) {
          // segment cw -> Lp against the node box grown by max(rho, RL) (+ fp32 slack), 0 <= tau <= 1
          const f3 tw = mkf((float)lg[0] - cw.x, (float)lg[1] - cw.y, (float)lg[2] - cw.z);
          const float RLw = n_sh > 1 ? 0.7072f * radius * 1.001f : 0.0f;
          const float gs = 4e-5f * (1.0f + fabsf(tw.x) + fabsf(tw.y) + fabsf(tw.z) + fabsf(cw.x) + fabsf(cw.y) + fabsf(cw.z));
          const float gw = fmaxf(rho, RLw) + gs;
          const float hx = fabsf(tw.x) > 1e-20f ? raw_rcp(tw.x) : 1e30f, hy = fabsf(tw.y) > 1e-20f ? raw_rcp(tw.y) : 1e30f,
                      hz = fabsf(tw.z) > 1e-20f ? raw_rcp(tw.z) : 1e30f;
          n_wl = bvh_wave_leaves(ac, [&](const BvhNode& nd) __attribute__((always_inline)) {
            const float ax = (nd.lo[0] - gw - cw.x) * hx, bx = (nd.hi[0] + gw - cw.x) * hx;
            const float ay = (nd.lo[1] - gw - cw.y) * hy, by = (nd.hi[1] + gw - cw.y) * hy;
            const float az = (nd.lo[2] - gw - cw.z) * hz, bz = (nd.hi[2] + gw - cw.z) * hz;
            const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
            const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), 1.0f));
            if (!(tn <= tf + 1e-5f)) return false;
            // the hull's radius at parameter tau is rho + tau*(RLw - rho) and tau <= tf inside this box: when
            // the light end is the wide one, test again with the radius reached at tf
            const float g2 = fmaxf(rho, fmaf(tf + 1e-5f, RLw - rho, rho)) + gs;
            const float ax2 = (nd.lo[0] - g2 - cw.x) * hx, bx2 = (nd.hi[0] + g2 - cw.x) * hx;
            const float ay2 = (nd.lo[1] - g2 - cw.y) * hy, by2 = (nd.hi[1] + g2 - cw.y) * hy;
            const float az2 = (nd.lo[2] - g2 - cw.z) * hz, bz2 = (nd.hi[2] + g2 - cw.z) * hz;
            const float tn2 = fmaxf(fmaxf(fminf(ax2, bx2), fminf(ay2, by2)), fmaxf(fminf(az2, bz2), 0.0f));
            const float tf2 = fminf(fminf(fmaxf(ax2, bx2), fmaxf(ay2, by2)), fminf(fmaxf(az2, bz2), 1.0f));
            return tn2 <= tf2 + 1e-5f;
          }, wl, WAVE_LEAF_CAP);
          // Second wave-level step: the records of those leaves, 64 bounds per round (lane = leaf slot * 8 +
          // record), against the hull of the wave's fans: the cone from the ball (cw, rho) to the ball
          // (Lp, RLw).  The lanes then test the surviving records' bounds with their own fan and skip the
          // per-lane node and leaf steps altogether.
          if (n_wl >= 0 && !(RT_DBG(fp) & 512u)) {
            const float T0w = dotf(tw, tw), dlw = RLw - rho;
            const float qaw = T0w - dlw * dlw, rqaw = raw_rcp(qaw);
            n_wr = 0;
            for (int l0 = 0; l0 < n_wl && n_wr >= 0; l0 += 8) {
              const int slot = l0 + (int)(lane >> 3), j = (int)(lane & 7u);
              bool may = false; int rec = 0;
              float4 wv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
              if (slot < n_wl) {
                const BvhNode nd = load_node(ac, wl[slot]);
                if (j < nd.count()) {
                  rec = nd.first + j;
                  const float4 b = ac.pbound[rec];
                  {   // exactly the first lines of fan_may_touch (below)
                    const float wx_ = (float)lg[0] - b.x, wy_ = (float)lg[1] - b.y, wz_ = (float)lg[2] - b.z;
                    wv = make_float4(wx_, wy_, wz_, b.w + RLw + 4e-5f * (1.0f + fabsf(wx_) + fabsf(wy_) + fabsf(wz_)));
                  }
                  const f3 ow = mkf(cw.x - b.x, cw.y - b.y, cw.z - b.z);
                  const float w0 = b.w + rho + gs;
                  const float o2 = dotf(ow, ow), qb = dotf(ow, tw) - w0 * dlw;
                  const float tau = fminf(fmaxf(-qb * rqaw, 0.0f), 1.0f);
                  const float g = fmaf(tau, fmaf(tau, qaw, 2.0f * qb), o2 - w0 * w0);      // min of |ow + tau tw|^2 - (w0 + tau dlw)^2
                  may = (qaw <= 0.0f) | (g <= 1e-4f * (T0w + o2 + w0 * w0 + 1.0f));
                }
              }
              const unsigned long long m = __ballot(may);
              const int cnt = __popcll(m);
              if (n_wr + cnt > WAVE_REC_CAP) { n_wr = -1; break; }
              if (may) {
                const int at = n_wr + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                wr[at] = rec; wrb[at] = wv;
              }
              n_wr += cnt;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            n_wr = __builtin_amdgcn_readfirstlane(n_wr);      // (wave-uniform by construction: lets the lanes' filter loop run on scalar counters)
          }
          if ((RT_DBG(fp) & 128u) && lane == 0) {
            atomicAdd(&C->dbg[0], 1ull); atomicAdd(&C->dbg[1], n_wl < 0 ? 1ull : 0ull); atomicAdd(&C->dbg[2], (unsigned long long)(n_wl < 0 ? 0 : n_wl));
            atomicAdd(&C->dbg[3], (unsigned long long)(n_wr < 0 ? 0 : n_wr));
          }
        }
        FCK(fck2);
        for (int c = 0; c < (skip_samples ? 0 : n_chunks); ++c) {
          float dx[CH], dy[CH], dz[CH], md[CH], tr[CH];
          // inverse directions for the box slabs: kept per sample for LDS scenes; the large-scene kernel
          // recomputes them per (box, sample) instead — boxes are few after culling, registers are not
          constexpr bool KEEP_INV = HAS_BOX && LDS_SCENE;
          float ix[KEEP_INV ? CH : 1], iy[KEEP_INV ? CH : 1], iz[KEEP_INV ? CH : 1];
#pragma unroll
          for (int s = 0; s < CH; ++s) {
            const int cellidx = c * CH + s;
            f3 ts = tl;
            if (n_sh > 1) {                                   // ray_tracer.py:219-235
              const int i = cellidx / n_sh, j = cellidx - i * n_sh;
              const float oi = fmaf(jit_ui(key, (uint32_t)cellidx), cell, fmaf(cell, (float)i, half));
              const float oj = fmaf(jit_uj(key, (uint32_t)cellidx), cell, fmaf(cell, (float)j, half));
              ts = mkf(fmaf(oj, up.x, fmaf(oi, right.x, tl.x)), fmaf(oj, up.y, fmaf(oi, right.y, tl.y)),
                       fmaf(oj, up.z, fmaf(oi, right.z, tl.z)));
            }
            const float d2 = dotf(ts, ts);
            const float rinv = raw_rsq(d2);
            dx[s] = ts.x * rinv; dy[s] = ts.y * rinv; dz[s] = ts.z * rinv;
            md[s] = d2 * rinv;
            tr[s] = (cellidx < n_cells) ? 1.0f : 0.0f;
            if (KEEP_INV) { ix[s] = raw_rcp(dx[s]); iy[s] = raw_rcp(dy[s]); iz[s] = raw_rcp(dz[s]); }
          }
          FCK(fck3);
          // one primitive's crossings folded into a sample's transmission product
          auto apply = [&](float& t_, bool in1, bool in2, float tau) __attribute__((always_inline)) {
            if (OPAQUE) { t_ = (in1 | in2) ? 0.0f : t_; }
            else { t_ *= in1 ? tau : 1.0f; t_ *= in2 ? tau : 1.0f; }
          };
          // Samples whose decision for a primitive lay within the fp32 error bound are left out of the products above and
          // decided in fp64 from the exact shadow ray AFTER the chunk's primitives are through (rare: ~1e-5 of the tests): up
          // to two (primitive, sample mask) entries per lane and chunk are remembered in registers; a third one makes the
          // whole (hit, light) pair undecided (-> re-test queue).  ONE fp64 site per chunk instead of one per test site:
          // inlined behind every sample block the exact code cost 83 spilled VGPRs and 5 ms of a 22 ms frame.
          int dk0 = -1, dk1 = -1; uint32_t dm0 = 0u, dm1 = 0u;
          bool plane_any = false;   // scenes with transparency: some sample of the chunk crosses a plane
          int ncross = 0;   // scenes with transparency: roots in range, any sample (the walk multiplies at most 20 crossings)
          // scenes with transparency: the marked primitives (Prim::pad bit 2: their surface meets another primitive's) that some
          // sample of this chunk crosses; two of them in one sample may be closer than 2e-6 -> compared after the chunk's primitives
          int pk0 = -1, pk1 = -1, pk2 = -1, pk3 = -1, npk = 0;
          auto note_marked = [&](int k_, bool crossed, bool deferred) __attribute__((always_inline)) {
            // (a sample whose decision is left to fp64 may cross too: pair_roots gives such samples a wide error bound)
            if (crossed | deferred) {
              if (npk == 0) pk0 = k_; else if (npk == 1) pk1 = k_; else if (npk == 2) pk2 = k_; else if (npk == 3) pk3 = k_; else unc = 1.0f;
              ++npk;
            }
          };
          auto retest = [&](int k_, uint32_t fmask) __attribute__((always_inline)) {
            if (fmask != 0u) {
              if (dk0 < 0) { dk0 = k_; dm0 = fmask; }
              else if (dk1 < 0) { dk1 = k_; dm1 = fmask; }
              else unc = 1.0f;
            }
          };
          // per-primitive blocks: everything that depends on (hit, primitive) once, then CH samples
          auto do_sphere = [&](const Prim& s_, int k_) __attribute__((always_inline)) {   // sphere.py:49-71
            const double ocx = p.x - s_.a[0], ocy = p.y - s_.a[1], ocz = p.z - s_.a[2];
            const float cc = (float)(((ocx * ocx + ocy * ocy) + ocz * ocz) - s_.a[4]);
            const f3 oc = mkf((float)ocx, (float)ocy, (float)ocz);
            const float tau = OPAQUE ? 0.0f : (float)sc.mats[MAT_STRIDE * s_.mat + 10];
            // error bounds of this (hit, sphere): |disc error| <= kap |b| + kap0 with e_b = 8 ulp * |oc|
            const float r_ = (float)s_.a[3];
            const float ocn = raw_sqrt(fmaxf(cc + r_ * r_, 0.0f));
            const float e_b = U8 * ocn, kap = 4.0f * e_b, kap0 = 4.0f * e_b * e_b + 4.0f * 6.0e-8f * fabsf(cc);
            {   // a sample of the light may lie on the sphere (crossing at the far end of the segment)
              const float lx = (float)(lg[0] - s_.a[0]), ly = (float)(lg[1] - s_.a[1]), lz = (float)(lg[2] - s_.a[2]);
              const float dl = raw_sqrt(fmaf(lz, lz, fmaf(ly, ly, lx * lx)));
              // (and the fp32 error of a root near that end grows with the hit's distance from the sphere: the comparison
              // t < |S - p| has a margin of half the sample's distance from the surface against e_b + the square root's error,
              // <= sqrt(kap |b| + kap0) for a decided discriminant — see k_shade_ls; random scene 200792)
              if (fabsf(dl - r_) <= reach + 1e-5f * (dl + r_) + 2.0f * (e_b + raw_sqrt(fmaf(kap, ocn, kap0)))) unc = 1.0f;
            }
            // a root can only come near the 2e-6 bound when p lies within ~1e-5 of the sphere's surface
            // (t1 t2 = cc): the hit's own sphere, or one touching it at the hit point
            const bool nearsurf = fabsf(cc) <= 1.2e-5f * (ocn + r_);
            uint32_t fmask = 0u;
            bool any1 = false, any2 = false;
#pragma unroll
            for (int s = 0; s < CH; ++s) {
              const float b = fmaf(oc.z, dz[s], fmaf(oc.y, dy[s], oc.x * dx[s]));   // oc.d  (a = d.d = 1)
              const float disc = fmaf(b, b, -cc);
              bool f = fmaf(kap, fabsf(b), kap0) >= fabsf(disc);        // the ray grazes the silhouette within rounding
              const float sq = raw_sqrt(disc);                          // NaN on a miss: both tests fail
              // roots {q, cc/q} with q the root that does not cancel: the other one can be the crossing of
              // the hit's own sphere at t' ~ 1e-6/cos, which has to be compared with the 2e-6 bound
              const float t1 = b < 0.0f ? sq - b : -b - sq;
              const float t2 = cc * raw_rcp(t1);
              const bool in1 = (t1 > T_LO) & (t1 < md[s]);
              const bool in2 = (t2 > T_LO) & (t2 < md[s]);
              if (nearsurf) {
                // a root within 1e-3 (relative) of the 2e-6 bound is not decided in fp32 — unless the other root blocks
                // an opaque scene's sample anyway
                const bool n1 = fabsf(t1 - T_LO) <= 1e-3f * T_LO, n2 = fabsf(t2 - T_LO) <= 1e-3f * T_LO;
                const bool open_ = OPAQUE ? !((in1 & !n1) | (in2 & !n2)) : true;
                f |= (n1 | n2) & open_;
              }
              f &= (c * CH + s) < n_cells;
              if (!f) apply(tr[s], in1, in2, tau);
              fmask |= f ? (1u << s) : 0u;
              if (!OPAQUE) { any1 |= in1; any2 |= in2; }
            }
            if (!OPAQUE) {
              // (samples deferred to the fp64 decision may cross as well: counted as if they did)
              ncross += fmask != 0u ? 2 : (int)any1 + (int)any2;
              // coincident surfaces: the walk counts two crossings once (bit 1: only together with a plane crossing)
              if (((s_.pad & 1) | ((s_.pad & 2) && plane_any)) && (any1 | any2 | (fmask != 0u))) unc = 1.0f;
              if (s_.pad & 4) note_marked(k_, any1 | any2, fmask != 0u);
            }
            retest(k_, fmask);
          };
          auto do_box = [&](const Prim& s_, int k_) __attribute__((always_inline)) {      // cube.py:76-107 (slab roots)
            const f3 lo = mkf((float)(s_.a[0] - p.x), (float)(s_.a[1] - p.y), (float)(s_.a[2] - p.z));
            const f3 hi = mkf((float)(s_.a[3] - p.x), (float)(s_.a[4] - p.y), (float)(s_.a[5] - p.z));
            const float tau = OPAQUE ? 0.0f : (float)sc.mats[MAT_STRIDE * s_.mat + 10];
            {   // a sample of the light may lie on the box's surface: inside the box grown by `reach`, not inside it shrunk
              const float g_ = reach + 1e-5f * (fabsf(lo.x) + fabsf(lo.y) + fabsf(lo.z) + fabsf(hi.x) + fabsf(hi.y) + fabsf(hi.z));
              const bool big = (tl.x >= lo.x - g_) & (tl.x <= hi.x + g_) & (tl.y >= lo.y - g_) & (tl.y <= hi.y + g_) & (tl.z >= lo.z - g_) & (tl.z <= hi.z + g_);
              const bool small_ = (tl.x > lo.x + g_) & (tl.x < hi.x - g_) & (tl.y > lo.y + g_) & (tl.y < hi.y - g_) & (tl.z > lo.z + g_) & (tl.z < hi.z - g_);
              if (big & !small_) unc = 1.0f;
            }
            // p within ~1e-5 of one of the box's face planes: slab roots can come near the 2e-6 bound
            const float fmin_ = fminf(fminf(fminf(fabsf(lo.x), fabsf(hi.x)), fminf(fabsf(lo.y), fabsf(hi.y))), fminf(fabsf(lo.z), fabsf(hi.z)));
            const bool nearsurf = fmin_ <= 1.2e-5f;
            uint32_t fmask = 0u;
            bool any1 = false, any2 = false;
#pragma unroll
            for (int s = 0; s < CH; ++s) {
              const float jx = KEEP_INV ? ix[KEEP_INV ? s : 0] : raw_rcp(dx[s]), jy = KEEP_INV ? iy[KEEP_INV ? s : 0] : raw_rcp(dy[s]),
                          jz = KEEP_INV ? iz[KEEP_INV ? s : 0] : raw_rcp(dz[s]);
              const float ax = lo.x * jx, bx = hi.x * jx;
              const float ay = lo.y * jy, by = hi.y * jy;
              const float az = lo.z * jz, bz = hi.z * jz;
              const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
              const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
              const bool ok = tn <= tf;
              const bool in1 = ok & (tn > T_LO) & (tn < md[s]);
              const bool in2 = ok & (tf > T_LO) & (tf < md[s]);
              // the ray passes an edge of the box within the slabs' rounding: each slab value carries the relative error
              // of 1/d (12 ulp of a unit vector's component) plus a few ulp; a component within 1e-3 of parallel is never decided
              const float jm = fmaxf(fmaxf(fabsf(jx), fabsf(jy)), fabsf(jz));
              const float rel = fmaf(U8, jm, 4.0f * 6.0e-8f);
              // (or through it on a chord shorter than 2e-6, which the reference's walk sees as ONE crossing)
              bool f = (fabsf(tf - tn) <= 2.0f * rel * (fabsf(tn) + fabsf(tf)) + 4e-6f) | (jm > 1e3f);
              if (nearsurf) {
                const bool n1 = fabsf(tn - T_LO) <= 1e-3f * T_LO, n2 = fabsf(tf - T_LO) <= 1e-3f * T_LO;
                const bool open_ = OPAQUE ? !((in1 & !n1) | (in2 & !n2)) : true;
                f |= ok & (n1 | n2) & open_;
              }
              f &= (c * CH + s) < n_cells;
              if (!f) apply(tr[s], in1, in2, tau);
              fmask |= f ? (1u << s) : 0u;
              if (!OPAQUE) { any1 |= in1; any2 |= in2; }
            }
            if (!OPAQUE) {
              // (samples deferred to the fp64 decision may cross as well: counted as if they did)
              ncross += fmask != 0u ? 2 : (int)any1 + (int)any2;
              // coincident surfaces: the walk counts two crossings once (bit 1: only together with a plane crossing)
              if (((s_.pad & 1) | ((s_.pad & 2) && plane_any)) && (any1 | any2 | (fmask != 0u))) unc = 1.0f;
              if (s_.pad & 4) note_marked(k_, any1 | any2, fmask != 0u);
            }
            retest(k_, fmask);
          };
          for (int k = sc.nS, e = sc.nS + sc.nP; k < e; ++k) {   // planes: infinite_plane.py:30-42
            if (!cand.has(k)) continue;
            const Prim& s_ = P[k];
            const float num = (float)(s_.a[3] - ((p.x * s_.a[0] + p.y * s_.a[1]) + p.z * s_.a[2]));
            const f3 N = mkf((float)s_.a[0], (float)s_.a[1], (float)s_.a[2]);
            const float tau = OPAQUE ? 0.0f : (float)sc.mats[MAT_STRIDE * s_.mat + 10];
            // a sample of the light may lie on the plane (crossing at the far end of the segment)
            if (fabsf(fmaf(N.z, tl.z, fmaf(N.y, tl.y, N.x * tl.x)) - num) <= reach + 1e-5f * fabsf(num)) unc = 1.0f;
            const bool nearsurf = fabsf(num) <= 1.2e-5f;                 // p on the plane: the hit's own plane
            uint32_t fmask = 0u;
            bool any1 = false;
#pragma unroll
            for (int s = 0; s < CH; ++s) {
              const float den = fmaf(N.z, dz[s], fmaf(N.y, dy[s], N.x * dx[s]));
              const float t1 = num * raw_rcp(den);
              const bool in1 = (t1 > T_LO) & (t1 < md[s]);               // den == 0 -> inf/NaN -> no hit
              // ray parallel to the plane within rounding; own plane: the root within 1e-3 of the 2e-6 bound
              bool f = (fabsf(den) <= 2.0f * U8) | (nearsurf & (fabsf(t1 - T_LO) <= 1e-3f * T_LO));
              f &= (c * CH + s) < n_cells;
              if (!f) apply(tr[s], in1, false, tau);
              fmask |= f ? (1u << s) : 0u;
              if (!OPAQUE) any1 |= in1;
            }
            if (!OPAQUE) {
              ncross += (int)(any1 | (fmask != 0u));
              if ((s_.pad & 1) && (any1 | (fmask != 0u))) unc = 1.0f;
              if (s_.pad & 4) note_marked(k, any1, fmask != 0u);
              plane_any |= any1 | (fmask != 0u);
            }
            retest(k, fmask);
          }
          FCK(fck4);
          if (!LDS_SCENE && ac.n_nodes > 0 && !fp.brute) {
            // Large scene: each lane walks the culling hierarchy with the hull of ITS shadow segments
            // to this light (all samples lie within RL of the light centre) and runs the sample block
            // only for the spheres / boxes that can cross one of them.  Results are those of the
            // brute-force loops below (the skipped primitives contribute a factor of exactly 1).
            // All of it is parametrised from the LIGHT end (y(s) = Lp + s*(p - Lp), fan radius RL*(1 - s)):
            // a reflected ray can hit the floor 1e6 units away, and from that end every product and sum
            // near the scene would carry the rounding of 1e6-sized coordinates (bounds would have to be
            // inflated by tens of units: such a hit made every record a candidate).  From the light end
            // the far point only enters through s*d with s ~ 1e-4.
            const float RL = n_sh > 1 ? 0.7072f * radius * 1.001f : 0.0f;
            const float qa = dotf(tl, tl) - RL * RL, rqa = raw_rcp(qa);
            const float Lx = (float)lg[0], Ly = (float)lg[1], Lz = (float)lg[2];
            const float ddx = -tl.x, ddy = -tl.y, ddz = -tl.z;                      // p - Lp
            // (wx, wy, wz) = Lp - centre of the record's bound, R0 = its radius + RL + slack: hit independent (see s_wrb)
            auto fan_may_touch_rel = [&](float wx, float wy, float wz, float R0) __attribute__((always_inline)) {
              if (RT_DBG(fp) & 16384u) return true;
              const float qb = fmaf(wz, ddz, fmaf(wy, ddy, wx * ddx)) + RL * R0;
              const float sa = fminf(fmaxf(-qb * rqa, 0.0f), 1.0f);                // min of |w + s d|^2 - (R0 - s RL)^2
              const float xx = fmaf(sa, ddx, wx), xy = fmaf(sa, ddy, wy), xz = fmaf(sa, ddz, wz);
              const float wrad = fmaf(-sa, RL, R0);
              return bool((qa <= 0.0f) | (fmaf(xz, xz, fmaf(xy, xy, xx * xx)) <= wrad * wrad));
            };
            auto fan_may_touch = [&](float cx, float cy, float cz, float r) __attribute__((always_inline)) {
              const float wx = Lx - cx, wy = Ly - cy, wz = Lz - cz;
              return fan_may_touch_rel(wx, wy, wz, r + RL + 4e-5f * (1.0f + fabsf(wx) + fabsf(wy) + fabsf(wz)));
            };
            // node boxes: segment Lp -> p against the box grown by the fan radius (+ fp32 slack), 0 <= s <= 1
            const float gx = fabsf(ddx) > 1e-20f ? raw_rcp(ddx) : 1e30f, gy = fabsf(ddy) > 1e-20f ? raw_rcp(ddy) : 1e30f,
                        gz = fabsf(ddz) > 1e-20f ? raw_rcp(ddz) : 1e30f;
            const float nslack = 4e-5f * (1.0f + fabsf(Lx) + fabsf(Ly) + fabsf(Lz) + scene_m);
            const float grow = RL + nslack;
            unsigned dbg_nodes = 0, dbg_bounds = 0, dbg_cands = 0;
            auto node_hit = [&](const BvhNode& nd) __attribute__((always_inline)) {
              if (RT_DBG(fp)) { ++dbg_nodes; dbg_bounds += nd.count(); }
              if (RT_DBG(fp) & 32768u) return true;
              const float ax = (nd.lo[0] - grow - Lx) * gx, bx = (nd.hi[0] + grow - Lx) * gx;
              const float ay = (nd.lo[1] - grow - Ly) * gy, by = (nd.hi[1] + grow - Ly) * gy;
              const float az = (nd.lo[2] - grow - Lz) * gz, bz = (nd.hi[2] + grow - Lz) * gz;
              const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
              const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), 1.0f));
              if (!(tn <= tf + 2e-6f)) return false;
              if (RL == 0.0f || (RT_DBG(fp) & 256u)) return true;
              // the fan is a cone: its radius at parameter s is (1 - s)*RL, and s >= tn inside this box.
              // Second test with the box grown by that radius only.
              const float g2 = fmaf(1.0f - fmaxf(tn - 2e-6f, 0.0f), RL, nslack);
              const float ax2 = (nd.lo[0] - g2 - Lx) * gx, bx2 = (nd.hi[0] + g2 - Lx) * gx;
              const float ay2 = (nd.lo[1] - g2 - Ly) * gy, by2 = (nd.hi[1] + g2 - Ly) * gy;
              const float az2 = (nd.lo[2] - g2 - Lz) * gz, bz2 = (nd.hi[2] + g2 - Lz) * gz;
              const float tn2 = fmaxf(fmaxf(fminf(ax2, bx2), fminf(ay2, by2)), fmaxf(fminf(az2, bz2), 0.0f));
              const float tf2 = fminf(fminf(fmaxf(ax2, bx2), fmaxf(ay2, by2)), fminf(fmaxf(az2, bz2), 1.0f));
              return tn2 <= tf2 + 2e-6f;
            };
            // The walk is split into three phases so that the lanes of a wave stay together (a wave pays for
            // every branch any of its lanes takes):
            //   1. find leaves: node tests only; each lane lists the leaves its fan may touch (LDS, per lane),
            //      either by walking the tree or by filtering the wave's leaf list;
            //   2. every lane takes one listed leaf per step: its <= 8 record bounds are tested together, the
            //      records that pass go to the lane's candidate list;
            //   3. every lane takes one candidate per step: the exact sample block.
            // Lists are bounded: a full leaf list pauses phase 1, a nearly full candidate list pauses phase 2.
            // When the wave has a record list (compact wave: the usual case after the regrouping of
            // rt_sort.hpp), phases 1 and 2 shrink to one bound test per listed record.
            unsigned audit_crossed = 0;
            auto exact_crossed = [&](int k) {
              const Prim& s_ = P[k];
              bool crossed = false;
#pragma unroll
              for (int sidx = 0; sidx < CH; ++sidx) {
                if (c * CH + sidx >= n_cells) continue;
                const double ex = dx[sidx], ey = dy[sidx], ez = dz[sidx], mdd = md[sidx];
                if (k < sc.nS) {
                  const double ocx = p.x - s_.a[0], ocy = p.y - s_.a[1], ocz = p.z - s_.a[2];
                  double tt = -(ocx * ex + ocy * ey + ocz * ez);
                  tt = tt < 0.0 ? 0.0 : (tt > mdd ? mdd : tt);
                  const double qx = ocx + tt * ex, qy = ocy + tt * ey, qz = ocz + tt * ez;
                  crossed |= qx * qx + qy * qy + qz * qz < s_.a[4] * (1.0 - 1e-6);
                } else {
                  double tn = 0.0, tf = mdd;
                  const double o3[3] = {p.x, p.y, p.z}, e3[3] = {ex, ey, ez};
                  for (int ax = 0; ax < 3; ++ax) {
                    const double inv = 1.0 / e3[ax];
                    double t0 = (s_.a[ax] - o3[ax]) * inv, t1 = (s_.a[3 + ax] - o3[ax]) * inv;
                    if (t0 > t1) { const double tmp = t0; t0 = t1; t1 = tmp; }
                    tn = t0 > tn ? t0 : tn; tf = t1 < tf ? t1 : tf;
                  }
                  crossed |= tn < tf - 1e-6;
                }
              }
              return crossed;
            };
            uint32_t* const my_leaf = s_leaf + threadIdx.x;       // entry e at [e * BLOCK]
            uint32_t* const my_cand = s_cand + threadIdx.x;
            const int n_nodes = ac.n_nodes;
            const int w_end = n_wl >= 0 ? n_wl : n_nodes;
            int wi = 0;                     // phase-1 position: wave-list entry (uniform) or node index (per lane)
            int n_leaf = 0, le = 0, n_cand = 0;
            unsigned it1 = 0, it2 = 0, it3 = 0;
            long long ck1 = 0, ck2 = 0, ck3 = 0, ck0 = (RT_DBG(fp) & 2048u) ? clock64() : 0;
            int we = 0;                     // position in the wave's record list
            bool finished = false;
            while (!finished) {
              if (n_wr >= 0) {
                // the wave's record list: each lane keeps the records its own fan may touch
                if (we < n_wr) {
                  float4 nx = wrb[we];                                        // (broadcast reads, one entry ahead)
                  for (; we < n_wr && !__any(n_cand >= CAND_LIST); ++we) {
                    const float4 b = nx;
                    nx = wrb[we + 1 < n_wr ? we + 1 : we];
                    if (RT_DBG(fp)) ++it2;
                    if (fan_may_touch_rel(b.x, b.y, b.z, b.w)) { my_cand[n_cand * BLOCK] = (uint32_t)wr[we]; ++n_cand; }
                  }
                }
                finished = we >= n_wr;
              } else if (!__any(le < n_leaf)) {
                if (!__any(wi < w_end)) finished = true;       // (the candidate lists are empty here)
                else {
                n_leaf = 0; le = 0;
                if (n_wl >= 0) {
                  for (; wi < n_wl && !__any(n_leaf >= LEAF_LIST); ++wi) {
                    const BvhNode nd = load_node(ac, wl[wi]);
                    if (RT_DBG(fp)) ++it1;
                    if (node_hit(nd)) { my_leaf[n_leaf * BLOCK] = (uint32_t)nd.first | ((uint32_t)nd.count() << 28); ++n_leaf; }
                  }
                } else {
                  BvhNode nd = load_node(ac, wi < n_nodes ? wi : 0);
                  while (wi < n_nodes && n_leaf < LEAF_LIST) {
                    const int nxt = wi + 1;
                    if (RT_DBG(fp)) ++it1;
                    const BvhNode nx = load_node(ac, nxt < n_nodes ? nxt : wi);
                    if (node_hit(nd)) {
                      if (nd.count() > 0) { my_leaf[n_leaf * BLOCK] = (uint32_t)nd.first | ((uint32_t)nd.count() << 28); ++n_leaf; }
                      wi = nxt; nd = nx;
                    } else {
                      wi = nd.skip();
                      if (wi < n_nodes) nd = load_node(ac, wi);
                    }
                  }
                }
                }
              }
              if (RT_DBG(fp) & 2048u) { const long long t = clock64(); ck1 += t - ck0; ck0 = t; }
              if (RT_DBG(fp) & 16u) n_leaf = 0;
              while (n_wr < 0 && __any(le < n_leaf) && !__any(n_cand > CAND_LIST - 8)) {
                if (le < n_leaf) {
                  const uint32_t packed = my_leaf[le * BLOCK];
                  const int first = (int)(packed & 0x0FFFFFFFu), nj = (int)(packed >> 28);
                  float4 b[8];
#pragma unroll
                  for (int j = 0; j < 8; ++j) b[j] = ac.pbound[first + (j < nj ? j : 0)];
#pragma unroll
                  for (int j = 0; j < 8; ++j)
                    if (j < nj && fan_may_touch(b[j].x, b[j].y, b[j].z, b[j].w)) { my_cand[n_cand * BLOCK] = (uint32_t)(first + j); ++n_cand; }
                }
                ++le;
                if (RT_DBG(fp)) ++it2;
              }
              if (RT_DBG(fp) & 2048u) { const long long t = clock64(); ck2 += t - ck0; ck0 = t; }
              if (RT_DBG(fp) & 32u) n_cand = 0;
              for (int e = 0; __any(e < n_cand); ++e) {
                if (RT_DBG(fp)) ++it3;
                if (e < n_cand) {
                  const int k = (int)my_cand[e * BLOCK];
                  if (RT_DBG(fp)) ++dbg_cands;
                  if ((RT_DBG(fp) & 65536u) && exact_crossed(k)) ++audit_crossed;
                  if (k < sc.nS) do_sphere(P[k], k);
                  else if (HAS_BOX) do_box(P[k], k);
                }
              }
              n_cand = 0;
              if (RT_DBG(fp) & 2048u) { const long long t = clock64(); ck3 += t - ck0; ck0 = t; }
            }
            if ((RT_DBG(fp) & 2048u) && lane == 0) {       // cycles per phase, summed over waves
              atomicAdd(&C->dbg[1], (unsigned long long)ck1);
              atomicAdd(&C->dbg[2], (unsigned long long)ck2); atomicAdd(&C->dbg[3], (unsigned long long)ck3);
            }
            if (RT_DBG(fp) & 1024u) {       // wave-level step counts of the three phases
              for (int o_ = 32; o_ > 0; o_ >>= 1) it1 = max(it1, (unsigned)__shfl_xor((int)it1, o_));
              if (lane == 0) {
                atomicAdd(&C->dbg[0], 1ull); atomicAdd(&C->dbg[1], (unsigned long long)it1);
                atomicAdd(&C->dbg[2], (unsigned long long)it2); atomicAdd(&C->dbg[3], (unsigned long long)it3);
              }
            }
            if (RT_DBG(fp) & 65536u) {
              // audit: every record that one of the lane's shadow segments crosses (fp64, exact geometry) must
              // have reached the lane's sample block — whatever dropped it (wave-level lists, node tests, the
              // lane's fan test) would not be conservative.  (The fp32 sample block is no judge here: for hits
              // ~1e3 units away its sphere roots cancel and report spurious crossings.)
              unsigned n_all = 0; int first_k = -1;
              for (int k = 0; k < sc.n; ++k) {
                if (k >= sc.nS && k < sc.nS + sc.nP) continue;
                if (exact_crossed(k)) { ++n_all; if (first_k < 0) first_k = k; }
              }
              if (active) {
                atomicAdd(&C->dbg[1], (unsigned long long)n_all);
                if (n_all != audit_crossed && atomicAdd(&C->dbg[0], 1ull) < 6ull)
                  printf("[audit] %u records crossed, %u of them processed (first crossed: %d): p (%.9g %.9g %.9g) L (%.9g %.9g %.9g) RL %g level %d wave lists %d %d\n",
                         n_all, audit_crossed, first_k, p.x, p.y, p.z, lg[0], lg[1], lg[2], RL, level, n_wl, n_wr);
              }
            }
            if ((RT_DBG(fp) & 1u) && active) {
              atomicAdd(&C->dbg[0], 1ull); atomicAdd(&C->dbg[1], (unsigned long long)dbg_nodes);
              atomicAdd(&C->dbg[2], (unsigned long long)dbg_bounds); atomicAdd(&C->dbg[3], (unsigned long long)dbg_cands);
            }
          } else {
            for (int k = 0; k < sc.nS; ++k)
              if (cand.has(k)) do_sphere(P[k], k);
            if (HAS_BOX)
              for (int k = sc.nS + sc.nP, e = sc.n; k < e; ++k)
                if (cand.has(k)) do_box(P[k], k);
          }
          FCK(fck5);
          while (__any(dm0 != 0u)) {                         // the deferred exact decisions of this chunk, one sample per turn
            bool e1 = false, e2 = false;
            float tau = 0.0f;
            const int s_star = dm0 ? __ffs((int)dm0) - 1 : -1;
            if (dm0) {
              const Prim& s_ = P[dk0];
              if (!OPAQUE) tau = (float)sc.mats[MAT_STRIDE * s_.mat + 10];
              const ExactRay er = exact_shadow_ray(sc, p, li, n_sh, fp.seed, gpix, (uint32_t)level, c * CH + s_star);
              exact_crossings(s_, er.so, er.dir, er.maxdist, &e1, &e2);
            }
#pragma unroll
            for (int s = 0; s < CH; ++s)
              if (s == s_star) apply(tr[s], e1, e2, tau);
            dm0 &= dm0 - 1u;
            if (dm0 == 0u) { dk0 = dk1; dm0 = dm1; dk1 = -1; dm1 = 0u; }
          }
          FCK(fck6);
          if (!OPAQUE && ncross > WALK_CAP && !fp.sequential) unc = 1.0f;   // a sample may cross more surfaces than the walk visits
#pragma unroll
          for (int s = 0; s < CH; ++s) total += tr[s];
          // (after the sum: the chunk's per-sample register arrays are dead here, the check below has the registers to itself)
          if (!OPAQUE && __any((npk >= 2) & (npk <= 4) & !(unc >= 0.0f))) {
            // merged crossings (see pair_roots): per sample, the roots of every two marked primitives the chunk crosses.
            // Rare (segments through the neighbourhood of an intersection curve), so the sample's direction is formed
            // again here instead of indexing the register arrays above with a run-time sample number.
            if ((npk >= 2) & (npk <= 4) & !(unc >= 0.0f)) {
              bool close = false;
              for (int s = 0; s < CH && c * CH + s < n_cells; ++s) {
                const int cellidx = c * CH + s;
                f3 ts = tl;
                if (n_sh > 1) {
                  const int i = cellidx / n_sh, j = cellidx - i * n_sh;
                  const float oi = fmaf(jit_ui(key, (uint32_t)cellidx), cell, fmaf(cell, (float)i, half));
                  const float oj = fmaf(jit_uj(key, (uint32_t)cellidx), cell, fmaf(cell, (float)j, half));
                  ts = mkf(fmaf(oj, up.x, fmaf(oi, right.x, tl.x)), fmaf(oj, up.y, fmaf(oi, right.y, tl.y)),
                           fmaf(oj, up.z, fmaf(oi, right.z, tl.z)));
                }
                const float d2 = dotf(ts, ts), rinv = raw_rsq(d2), mds = d2 * rinv;
                const float sx = ts.x * rinv, sy = ts.y * rinv, sz = ts.z * rinv;
                for (int ia = 0; ia + 1 < npk; ++ia) {
                  const int ka = ia == 0 ? pk0 : (ia == 1 ? pk1 : pk2);
                  const PairRoots A = pair_roots(P[ka], p, sx, sy, sz);
                  for (int ib = ia + 1; ib < npk; ++ib) {
                    const int kb = ib == 1 ? pk1 : (ib == 2 ? pk2 : pk3);
                    close |= pair_close(A, pair_roots(P[kb], p, sx, sy, sz), mds);
                  }
                }
              }
              if (close) unc = 1.0f;
            }
          }
          FCK(fck7);
        }
        if ((RT_DBG(fp) & 2u) && LDS_SCENE && active) {      // audit build, shortcut disabled: check what it would have claimed
          if (lane_lit && total != (float)n_cells) atomicAdd(&C->dbg[0], 1ull);
          if (lane_dark && total != 0.0f) atomicAdd(&C->dbg[1], 1ull);
          if (!lane_lit && !lane_dark) atomicAdd(&C->dbg[2], 1ull);
          atomicAdd(&C->dbg[3], 1ull);
        }
#ifdef RT_DEBUG_KERNEL
        if ((RT_DBG(fp) & 33554432u) && active && !skip_samples && !(unc >= 0.0f) && !(LDS_SCENE && (lane_lit || (OPAQUE && lane_dark)))) {
          // audit build, bit 25: a pair decided in fp32 against the exact walk of every cell (tests/test_gpu_audit.py)
          double tsum = 0.0; int qd = 0;
          for (int cell_ = 0; cell_ < n_cells; ++cell_) {
            const ExactRay er = exact_shadow_ray(sc, p, li, n_sh, fp.seed, gpix, (uint32_t)level, cell_);
            tsum += shadow_walk(P, sc.nS, sc.nP, sc.nB, sc.mats, er.so, er.dir, er.maxdist, &qd, false, fp.plane_thr);
          }
          atomicAdd(&g_walk_audit[0], 1ull);
          if (fabs(tsum - (double)total) > 1e-3 && atomicAdd(&g_walk_audit[1], 1ull) < 12ull) {
            printf("[walk audit] level %d light %d prim %d: fp32 sum %.6f exact %.6f  p (%.17g %.17g %.17g) gpix %u\n", level, li, hits[h].prim, total, tsum, p.x, p.y, p.z, gpix);
            for (int cell_ = 0; cell_ < n_cells; ++cell_) {
              const ExactRay er = exact_shadow_ray(sc, p, li, n_sh, fp.seed, gpix, (uint32_t)level, cell_);
              const double T_ = shadow_walk(P, sc.nS, sc.nP, sc.nB, sc.mats, er.so, er.dir, er.maxdist, &qd, false, fp.plane_thr);
              printf("[walk audit]    cell %d exact T %.6f dir (%.17g %.17g %.17g) maxdist %.17g\n", cell_, T_, er.dir.x, er.dir.y, er.dir.z, er.maxdist);
            }
          }
        }
#endif
        bool undecided = !skip_samples & (unc >= 0.0f);
        // the conservative claims of fan_cull are exact whatever the sample blocks computed for the lane
        if (LDS_SCENE && lane_lit) { total = (float)n_cells; undecided = false; }
        if (LDS_SCENE && OPAQUE && lane_dark) { total = 0.0f; undecided = false; }
        undecided &= active;
        if (undecided) total = 0.0f;
        const float T = total * inv_cells;
        const float I = fmaf((float)lg[7], T, 1.0f - (float)lg[7]);                       // ray_tracer.py:543
        const f3 l = normalize_clamped(mkf((float)(lg[0] - hp.x), (float)(lg[1] - hp.y), (float)(lg[2] - hp.z)));
        const float ndl = fmaxf(dotf(nf, l), 0.0f);
        const float s1 = I * ndl;
        dif[0] = fmaf((float)mat[0] * (float)lg[3], s1, dif[0]);
        dif[1] = fmaf((float)mat[1] * (float)lg[4], s1, dif[1]);
        dif[2] = fmaf((float)mat[2] * (float)lg[5], s1, dif[2]);
        const float dn = -dotf(l, nf);                                                    // (-l).n
        const f3 rf = mkf(fmaf(-2.0f * dn, nf.x, -l.x), fmaf(-2.0f * dn, nf.y, -l.y), fmaf(-2.0f * dn, nf.z, -l.z));
        float rdv = fmaxf(dotf(rf, v), 0.0f);
        const float shin = (float)mat[9];
        // Sequential renderer, shininess 0: the specular term is a STEP at r.v = 0 (1 for r.v > 0, absent otherwise,
        // ray_tracer.py:356-359).  Grid-aligned scenes put r.v at zero up to the rounding of the reference's own fp64
        // arithmetic, which only the literal fp64 evaluation reproduces: within 1e-4 of the step it is formed like
        // phong_light_exact does (rare, wave-uniform).
        // (LDS-staged scenes only: their instantiations run one wave per SIMD and have the registers; in the large-scene
        // kernel the fp64 code cost 26 more spilled VGPRs — there the step stays an fp32 decision)
        if (LDS_SCENE && fp.sequential && __any((shin == 0.0f) & (fabsf(dotf(rf, v)) < 1e-4f))) {
          if ((shin == 0.0f) & (fabsf(dotf(rf, v)) < 1e-4f))
            rdv = seq_specular_step(r.o, hp, n, lg[0], lg[1], lg[2]) ? fmaxf(rdv, 1e-30f) : 0.0f;
        }
        // np.power(0, 0) == 1; the sequential renderer adds the specular term only for r.v > 0 (ray_tracer.py:356-359)
        // (rdv ** shininess as exp2(shininess * log2(rdv)) on the raw hardware transcendentals, as in k_shade_ls: 1 ulp each,
        // relative error <= shininess * 2^-22 — far inside the 8-bit tolerance; powf() is ~70 instructions per (hit, light))
        const float pw = rdv > 0.0f ? (shin == 0.0f ? 1.0f : __builtin_amdgcn_exp2f(shin * __builtin_amdgcn_logf(rdv))) : ((shin == 0.0f && !fp.sequential) ? 1.0f : 0.0f);
        const float s2 = I * pw * (float)lg[6];
        spc[0] = fmaf((float)mat[3] * (float)lg[3], s2, spc[0]);
        spc[1] = fmaf((float)mat[4] * (float)lg[4], s2, spc[1]);
        spc[2] = fmaf((float)mat[5] * (float)lg[5], s2, spc[2]);
        if (__any(undecided)) {
          // colour per unit of T of this (hit, light), times weight and (1 - transparency)
          const float cm = (float)lg[7] * (1.0f - (float)mat[10]), sp = pw * (float)lg[6];
          queue_fix(undecided, fix, C, fp, p, (float)r.w.x * cm * (float)lg[3] * fmaf((float)mat[0], ndl, (float)mat[3] * sp),
                    (float)r.w.y * cm * (float)lg[4] * fmaf((float)mat[1], ndl, (float)mat[4] * sp),
                    (float)r.w.z * cm * (float)lg[5] * fmaf((float)mat[2], ndl, (float)mat[5] * sp), r.lp, gpix, level, li);
        }
        FCK(fck8);
      }
      const float one_m = 1.0f - (float)mat[10];
      float bt[3] = {0.0f, 0.0f, 0.0f};
      if (seq_background_term(fp, level, mat)) {                                            // ray_tracer.py:370-377
        bt[0] = (float)(sc.bg[0] * mat[10]); bt[1] = (float)(sc.bg[1] * mat[10]); bt[2] = (float)(sc.bg[2] * mat[10]);
      }
      if (active)
        fb_put(fb, level == 0, r.lp, (float)r.w.x * ((dif[0] + spc[0]) * one_m + bt[0]),
               (float)r.w.y * ((dif[1] + spc[1]) * one_m + bt[1]), (float)r.w.z * ((dif[2] + spc[2]) * one_m + bt[2]));
    }
    spawn_children(active, level, fp, mat, r, hp, n, nq, C, spawn_segment(base >> 6, count));
  }
#ifdef RT_DEBUG_KERNEL
  if (fprof && lane == 0) {
    const long long v_[10] = {fck0, fck1, fck2, fck3, fck4, fck5, fck6, fck7, fck8, fck9};
    for (int i = 0; i < 10; ++i) atomicAdd(&g_fast_cyc[i], (unsigned long long)v_[i]);
    atomicAdd(&g_fast_cyc[10], (unsigned long long)(clock64() - fck_begin));
  }
#endif
#undef FCK
}

#ifndef RT_NO_LAUNCHERS   // (single-kernel compile harness: tools/one_kernel.sh)
// Chunk size for an n x n grid: 9 samples per pass unless the grid is smaller.  Larger chunks (16, 25) keep 5 arrays of
// CH floats in registers: with the flag masks of the exact re-test the 25-sample instantiation dropped to one wave per
// SIMD (pool + 2 spheres at 4K: 9.7 ms with chunks of 25, 7.1 with 16, 5.8 with 9; 8x8 grids: 19.0 / 12.2 / 10.2 ms).
static inline int pick_chunk(int n_cells) {
  if (const char* e = getenv("RT_SHADE_CH")) { const int c = atoi(e); if (c == 9 || c == 4 || c == 1) return c; }
  return n_cells <= 1 ? 1 : (n_cells <= 4 ? 4 : 9);
}

template <int CH>
static void launch_shade_fast_ch(const DevScene& ds, bool opaque, const DevCamera& cam, const FrameParams& fp, int level,
                                 RayQueue cur, RayQueue nxt, const HitRec* hits, Counters* C, float* fb, FixRec* fix, int grid,
                                 int num_cu, hipStream_t st, int n_chunks) {
  const bool lds = ds.n <= LDS_PRIMS, box = ds.nB > 0;
#define RT_LAUNCH(OP, BX, LD)                                                                                     \
  hipLaunchKernelGGL((k_shade_fast<CH, OP, BX, LD>), dim3(shade_grid<k_shade_fast<CH, OP, BX, LD>>(num_cu, grid)), dim3(BLOCK), 0, st, ds, cam, fp, level, cur, nxt, \
                     hits, C, fb, n_chunks, fix)
  if (opaque) {
    if (!box && lds) RT_LAUNCH(true, false, true);
    else if (lds) RT_LAUNCH(true, true, true);
    else RT_LAUNCH(true, true, false);
  } else {
    if (!box && lds) RT_LAUNCH(false, false, true);
    else if (lds) RT_LAUNCH(false, true, true);
    else RT_LAUNCH(false, true, false);
  }
#undef RT_LAUNCH
}

static void launch_shade_fast(const DevScene& ds, bool opaque, const DevCamera& cam, const FrameParams& fp, int level,
                              RayQueue cur, RayQueue nxt, const HitRec* hits, Counters* C, float* fb, FixRec* fix, int grid,
                              int num_cu, hipStream_t st) {
  const int n_cells = fp.n_shadow > 1 ? fp.n_shadow * fp.n_shadow : 1;
  const int ch = pick_chunk(n_cells);
  const int n_chunks = (n_cells + ch - 1) / ch;
  switch (ch) {
    case 9: launch_shade_fast_ch<9>(ds, opaque, cam, fp, level, cur, nxt, hits, C, fb, fix, grid, num_cu, st, n_chunks); break;
    case 4: launch_shade_fast_ch<4>(ds, opaque, cam, fp, level, cur, nxt, hits, C, fb, fix, grid, num_cu, st, n_chunks); break;
    default: launch_shade_fast_ch<1>(ds, opaque, cam, fp, level, cur, nxt, hits, C, fb, fix, grid, num_cu, st, n_chunks); break;
  }
}

#endif  // RT_NO_LAUNCHERS

}  // namespace rt
