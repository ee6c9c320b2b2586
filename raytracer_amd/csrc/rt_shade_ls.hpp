// rt_shade_ls.hpp — fp32 shading kernel for small scenes in "light space": the dominant kernel of
// the headline benchmark (pool.txt class: <= MAXS spheres, MAXP planes, MAXB boxes).
//
// Same contract as k_shade_fast (see rt_shade_fast.hpp for why the product over all surface
// crossings equals the reference's shadow walk, ray_tracer.py:247-293, and why the origin-dependent
// terms are formed in fp64 from the exact hit point).  What is particular here:
//
//  * A shadow sample of light L is S = Lp + oi*right + oj*up (ray_tracer.py:224-229), so the
//    UNNORMALISED shadow vector is ts = tl + oi*right + oj*up with tl = Lp - p.  Every dot product
//    the intersection tests need is therefore affine in (oi, oj):
//        oc.ts = A + oi*B + oj*C          (A = oc.tl, B = oc.right, C = oc.up; per hit/light/sphere)
//        N.ts  = D0 + oi*D1 + oj*D2       (per hit/light/plane)
//        ts.ts = T0 + oi*(T1 + oi) + oj*(T2 + oj)     (right, up orthonormal)
//    and the ray p + tau*ts reaches the sample at tau = 1.  A sphere crossing solves
//        tau*|ts|^2 = -(oc.ts) +- sqrt((oc.ts)^2 - |ts|^2*cc),   in range iff 0 < tau*|ts|^2 < |ts|^2.
//    No per-sample normalisation (no rsq, no direction vector), 2 FMAs instead of a 3-term dot per
//    sphere, and only (oi, oj, |ts|^2) per sample: the loop nest is sample-outer / primitive-inner
//    with all primitive terms in registers and no per-sample arrays.
//  * Two samples are processed per iteration in the two halves of 64-bit register pairs
//    (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32).  The per-primitive coefficients are kept as PAIRS OF
//    DIFFERENT scalars — (A, B), (C, cc), (D0, D1), (D2, num), two spheres' root signs — and broadcast
//    into both halves by the instruction's op_sel bits, so a coefficient costs one register, not a
//    splatted pair (lo2 / hi2 below; see ls_light_sum for how the broadcast is kept inside the loop).
//  * The in-range test "0 < u < d2" is ONE unsigned integer compare on the float bit patterns
//    (negative u and NaN from a negative discriminant both have larger bit patterns than any
//    positive finite d2).  The lower bound is 0 instead of the reference's 2e-6 along the ray; the
//    two differ only when the light is behind the hit's own surface, where diffuse is zero.  For an
//    own PLANE the reference lets such rays through when |n.d| >= 0.5 (the own-surface hit at
//    t = 1e-6/|n.d| fails its t > 1e-6 test after the 1e-6 origin shift); that is reproduced exactly
//    by a per-wave slow path (LO) taken only when some lane has that configuration; the same path
//    applies the bound to the hit's own sphere / box when the outcome can depend on it (light inside
//    the own primitive, or transparency anywhere in the scene).
//  * EVERY hit / miss decision is the fp64 mode's decision.  The loop tracks, per (hit, light), how close
//    any of its decisions came to its threshold relative to a bound on the fp32 error of the quantity
//    compared (LsTol): discriminants near zero (a sample grazing a silhouette), roots near 0 (a second
//    surface at the shadow origin), denominators near zero, the 2e-6 bound of the LO path, the 0.9 switch
//    of the light basis; crossings near the sample itself are excluded per (light, primitive) on the host
//    side of the block (s_lstr: the light's square touches the primitive).  A pair with such a decision is
//    shaded with T = 0 and queued (FixRec); k_fix_exact adds its exact fp64 shadow term afterwards.
//    north_star: "fp32 fast mode within <= 1/255 per 8-bit channel" — met on every channel, because no
//    shadow sample can flip (tests/test_gpu_parity.py gates on the maximum, not on a fraction).
//  * (round 3) All-opaque scenes whose candidate spheres have the shadow origin outside them use a loop WITHOUT square
//    roots or reciprocals (ls_light_sum_simple): blocked <=> min3(disc, -oc.ts, oc.ts + |ts|^2) > 0, counted with the
//    clamp bit of a packed multiply; the bit-pattern form above serves scenes with transparency and the LO path.
//  * (round 3) Fused frames (FrameParams::fuse0 == 2, template parameter FALL): the kernel forms the camera ray of its
//    pixel at level 0, traces the children its hits spawn and hands the children's HITS on as 64-byte HitRay records —
//    one launch per level, no k_trace (see the kernel's comments and DESIGN.md section 3).
#pragma once
#include "rt_kernels.hpp"
#include "rt_shade_fast.hpp"

namespace rt {

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat(float a) { v2f r; r.x = a; r.y = a; return r; }
__device__ __forceinline__ v2f lo2(v2f a) { return __builtin_shufflevector(a, a, 0, 0); }   // -> op_sel broadcast
__device__ __forceinline__ v2f hi2(v2f a) { return __builtin_shufflevector(a, a, 1, 1); }
__device__ __forceinline__ v2f mk2(float x, float y) { v2f r; r.x = x; r.y = y; return r; }
__device__ __forceinline__ v2f pk_sqrt(v2f a) { v2f r; r.x = raw_sqrt(a.x); r.y = raw_sqrt(a.y); return r; }
__device__ __forceinline__ v2f pk_rcp(v2f a) { v2f r; r.x = raw_rcp(a.x); r.y = raw_rcp(a.y); return r; }
// 0 <= u < hi for hi positive finite, as ONE unsigned compare of the bit patterns: negative u
// (sign bit set), NaN and +inf all compare above any positive finite hi.
__device__ __forceinline__ bool in_open(float u, float hi) { return __float_as_uint(u) < __float_as_uint(hi); }
__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) { return min(a, min(b, c)); }

// A 16-byte LDS read that is REALLY issued one loop iteration ahead.  Written in C++ ("nb = cells[q + 1]" at the top of
// the iteration, consumed at the top of the next one), hipcc moves the read to where its value is used: the loop then
// starts every iteration with ds_read_b128 + s_waitcnt lgkmcnt(0), a full LDS latency with two waves per SIMD to hide
// it (seen in the ISA of round 2's loops).  lds_issue starts the read, lds_arrive waits for it; between the two the
// compiler treats the destination as already written, so nothing may touch it — the "+v" of lds_arrive orders every use
// behind the wait.  (LDS returns in order and the loop has no other LDS / scalar-memory access, so the compiler's own
// waitcnt bookkeeping stays conservative.)
typedef float ls_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_issue(ls_v4f& dst, const float4* p) {
  const uint32_t a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float4*)p;
  // "+v": destination tied to the variable's current register — the loop-carried variable then lives in ONE register
  // quadruple, and no copy can be placed between the issue and the wait (with "=v" hipcc put the result in a fresh
  // quadruple and copied it to the loop's register right behind the ds_read, before the data had arrived).  Call it
  // after the last use of the previous value.
  asm volatile("ds_read_b128 %0, %1" : "+v"(dst) : "v"(a));
}
__device__ __forceinline__ void lds_arrive(ls_v4f& dst) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dst)); }

constexpr int LS_MAX_CELLS = 1024;   // lights x n^2 entries of the per-block sample table (8 KiB)
constexpr int LS_MAX_LIGHTS = 16;    // lights / materials staged in LDS (uniform global loads inside the
constexpr int LS_MAX_MATS = 32;      // light loop were emitted as vector loads + full-latency waits)
constexpr float LS_U = 6.0e-8f;      // 2^-24, the unit roundoff of fp32
// "not decided in fp32" forced by a special case: any value above every real bound; the audit build tags the cases
// (own-sphere root near 2e-6 / own-plane root / direction parallel to a slab / box edge or sample-end root / own-box root)
#ifdef RT_DEBUG_KERNEL
#define RT_UNC_FORCED(tag) (tag)
#else
#define RT_UNC_FORCED(tag) 3.0e38f
#endif

// Per-hit (cc, sgn, num, box slabs) and per-light (A, B, C, D*) primitive terms live in plain local
// arrays of the kernel (fully unrolled, compile-time indices -> VGPRs); this is a view of references
// so the sample loop can be a function.  (A struct of arrays was partly promoted to LDS by hipcc.)
template <int MAXS, int MAXP, int MAXB, bool OPAQUE>
struct LsPrims {
  // (the coefficient pairs are non-const: ls_light_sum passes them through empty asm statements, values unchanged)
  v2f (&AB)[MAXS];                             // (A, B) = (oc.tl, oc.right)
  v2f (&Cc)[MAXS];                             // (C, cc) = (oc.up, |oc|^2 - r^2)
  v2f (&Sg)[(MAXS + 1) / 2];                   // root signs of spheres (2j, 2j+1): +1 far root (origin inside), -1 near root
  const float (&tauS)[MAXS];
  const float (&kapS)[MAXS];                   // per (hit, light, sphere): error bound of the discriminant per unit of |oc.ts| (LsTol::kap with THIS sphere's |oc|)
  v2f (&D01)[MAXP]; v2f (&D2n)[MAXP];          // (N.tl, N.right), (N.up, offset - p.N)
  const float (&tauP)[MAXP];
  const float (&blo)[MAXB > 0 ? MAXB : 1][3]; const float (&bhi)[MAXB > 0 ? MAXB : 1][3];
  const float (&tauB)[MAXB > 0 ? MAXB : 1]; const bool (&binside)[MAXB > 0 ? MAXB : 1];
};

// Error bounds of one (hit, light), all derived from the magnitudes that enter its fp32 arithmetic.
struct LsTol {
  float kap;      // |disc error| <= kap*|oc.ts| + kap0 around a vanishing discriminant
  float kap0;
  float elo;      // roots u (= distance along the ray times |ts|) up to elo count as "at the shadow origin"
  float eden;     // |N.ts| up to eden counts as "parallel to the plane"
};
struct LsFlags { float unc; uint32_t mlow; float dmin; };   // what the loop tracks (see ls_decided)
__device__ __forceinline__ LsFlags ls_flags_init() { LsFlags f; f.unc = -3.0e38f; f.mlow = 0xFFFFFFFFu; f.dmin = 3.0e38f; return f; }
// true: every decision of the pair kept clear of its threshold by more than the fp32 error bound
__device__ __forceinline__ bool ls_decided(const LsFlags& f, const LsTol& t) {
  return (f.unc < -t.kap0) & (f.mlow > __float_as_uint(t.elo)) & (f.dmin > t.eden);
}

// Transmission sum of one light for one hit: sum over the n x n cells of the product of blocker
// transparencies (opaque scenes: count of unblocked cells).  Two cells per iteration.
//
// Opaque scenes keep the "blocked" state out of the scalar unit: every candidate crossing is
// expressed as u = tau*|ts|^2 and folded into an UNSIGNED MINIMUM of float bit patterns
// (v_min_u32 / v_min3_u32); the cell is blocked iff min_bits < bits(|ts|^2) (see in_open).  A
// compare per test would put ~3 SALU mask operations per VALU compare into the wave's instruction
// stream (measured: 170 SALU vs 134 VALU instructions per pair).
// NS = MAXS: the kernel is instantiated for sphere counts 2, 4, 6; unused slots hold A = B = C = 0,
// cc = 1 (negative discriminant, never hit), so the sphere loop has no guards.
// SDENSE: every one of the NS sphere slots is active (no per-slot test of `act`); NOPLANES: no plane is active (the plane
// block is compiled out).  Both are wave-uniform facts of a (wave, light) pair, known before the loop: tested once there
// instead of once per sample pair on the scalar unit (a compare + branch each, measured 3-5 % of the frame).
template <int NS, int MAXS, int MAXP, int MAXB, bool OPAQUE, bool LO, bool SDENSE = false, bool NOPLANES = false>
__device__ __forceinline__ float ls_light_sum(LsPrims<MAXS, MAXP, MAXB, OPAQUE>& pr, uint32_t act, int nB,
                                              const float4* __restrict__ cells, int n_cells, JitKey key, float cell32,
                                              f3 tl, f3 right, f3 up, float T0, float T1, float T2, float kap,
                                              LsFlags& fl, int own = -1) {
  float total = 0.0f;
  uint32_t n_blocked = 0u;
  const int n_pairs = (n_cells + 1) >> 1;
  // The coefficient pairs pass through an empty asm statement in every iteration (values unchanged): that makes
  // them loop-variant for the optimiser, so the broadcasts lo2() / hi2() stay inside the loop body, where
  // instruction selection folds them into the op_sel bits of the packed instruction.  (Loop-invariant, they are
  // hoisted as splatted register pairs: twice the registers — the kernel then sits at the 256-VGPR limit of
  // 2 waves/SIMD.)  The asm works on the kernel's own arrays, not on copies: a copy would double the registers
  // of the values that stay live for the next light (cc, num, the root signs).
  v2f (&AB)[MAXS] = pr.AB; v2f (&Cc)[MAXS] = pr.Cc; v2f (&Sg)[(MAXS + 1) / 2] = pr.Sg;
  v2f (&D01)[MAXP] = pr.D01; v2f (&D2n)[MAXP] = pr.D2n;
  float unc = fl.unc, dmin = fl.dmin;
  uint32_t mlow = fl.mlow;
  // LO: coefficients of the lane's OWN sphere (selected once; the slow path is rare and must not
  // raise the kernel's register budget, so only the own sphere gets the two-root treatment)
  float oA = 0.0f, oB = 0.0f, oC = 0.0f, occ = 1.0f, otau = 1.0f, okap = kap;
  if (LO) {
#pragma unroll
    for (int k = 0; k < NS; ++k)
      if (own == k) { oA = AB[k].x; oB = AB[k].y; oC = Cc[k].x; occ = Cc[k].y; okap = pr.kapS[k]; if (!OPAQUE) otau = pr.tauS[k]; }
  }
  // cell base offsets: LDS broadcast reads, fetched one iteration ahead (with ~2 waves per SIMD an
  // un-prefetched ds_read + s_waitcnt at the top of every pair stalls the wave for its full latency)
  ls_v4f nb = {0.0f, 0.0f, 0.0f, 0.0f};                                     // (b0.x, b1.x, b0.y, b1.y) of sample pair 0
  lds_issue(nb, cells);
  uint32_t xa = key.a + JIT_PHI_A, xb = key.b + JIT_PHI_B;                  // cell 0 (see jit_ui / jit_uj)
#pragma unroll 1
  for (int q = 0; q < n_pairs; ++q) {
#pragma unroll
    for (int k = 0; k < NS; ++k) { asm("" : "+v"(AB[k])); asm("" : "+v"(Cc[k])); }
#pragma unroll
    for (int k = 0; k < (NS + 1) / 2; ++k) asm("" : "+v"(Sg[k]));
#pragma unroll
    for (int k = 0; k < MAXP; ++k) { asm("" : "+v"(D01[k])); asm("" : "+v"(D2n[k])); }
    const int c0 = 2 * q, c1 = 2 * q + 1;
    const bool valid1 = c1 < n_cells;                                       // wave-uniform
    lds_arrive(nb);
    v2f oi, oj;                                                             // ray_tracer.py:224-225
    oi.x = fmaf((float)xa, cell32, nb.x); oi.y = fmaf((float)(xa + JIT_PHI_A), cell32, nb.y);
    oj.x = fmaf((float)xb, cell32, nb.z); oj.y = fmaf((float)(xb + JIT_PHI_B), cell32, nb.w);
    asm volatile("" : "+v"(oi), "+v"(oj));                                  // (oi, oj are complete before the next read overwrites nb)
    lds_issue(nb, cells + (q + 1 < n_pairs ? q + 1 : 0));                   // the next pair's offsets, in flight during this iteration
    xa += 2u * JIT_PHI_A; xb += 2u * JIT_PHI_B;                             // jitter lattice of the next pair
    const v2f d2 = pk_fma(oi, splat(T1) + oi, pk_fma(oj, splat(T2) + oj, splat(T0)));   // |ts|^2
    uint32_t m0 = 0xFFFFFFFFu, m1 = valid1 ? 0xFFFFFFFFu : 0u;             // min of candidate bit patterns
    uint32_t ml = 0xFFFFFFFFu;                                              // transparent scenes: smallest positive root
    v2f tr = splat(1.0f);
#pragma unroll
    for (int k = 0; k < NS; ++k) {                                          // spheres: sphere.py:49-71
      if (!SDENSE && !(act & (1u << k))) continue;                          // culled for this (wave, light)
      const v2f bt = pk_fma(oj, lo2(Cc[k]), pk_fma(oi, hi2(AB[k]), lo2(AB[k])));                 // oc.ts
      const v2f disc = pk_fma(bt, bt, -(d2 * hi2(Cc[k])));
      const v2f sq = pk_sqrt(disc);                                                              // NaN on a miss
      const v2f sg = (k & 1) ? hi2(Sg[k >> 1]) : lo2(Sg[k >> 1]);
      // a discriminant within its error bound of zero: the sample grazes the silhouette
      unc = fmaxf(fmaxf(unc, fmaf(pr.kapS[k], fabsf(bt.x), -fabsf(disc.x))), fmaf(pr.kapS[k], fabsf(bt.y), -fabsf(disc.y)));   // one v_max3_f32
      if (LO) {                                                             // own sphere: handled after the loop
        if (OPAQUE) {
          v2f u = pk_fma(sg, sq, -bt);
          if (own == k) u = splat(-1.0f);
          m0 = min(m0, __float_as_uint(u.x)); m1 = min(m1, __float_as_uint(u.y));
        } else {
          const v2f u1 = -bt - sq, u2 = sq - bt;
          const float ta = own == k ? 1.0f : pr.tauS[k];
          tr.x *= in_open(u1.x, d2.x) ? ta : 1.0f; tr.y *= in_open(u1.y, d2.y) ? ta : 1.0f;
          tr.x *= in_open(u2.x, d2.x) ? ta : 1.0f; tr.y *= in_open(u2.y, d2.y) ? ta : 1.0f;
          if (own != k) ml = min(ml, min(umin3(__float_as_uint(u1.x), __float_as_uint(u2.x), __float_as_uint(u1.y)), __float_as_uint(u2.y)));
        }
      } else if (OPAQUE) {
        const v2f u = pk_fma(sg, sq, -bt);                                  // far root iff origin inside
        m0 = min(m0, __float_as_uint(u.x)); m1 = min(m1, __float_as_uint(u.y));
      } else {
        const v2f u1 = -bt - sq, u2 = sq - bt;
        const float ta = pr.tauS[k];
        tr.x *= in_open(u1.x, d2.x) ? ta : 1.0f; tr.y *= in_open(u1.y, d2.y) ? ta : 1.0f;
        tr.x *= in_open(u2.x, d2.x) ? ta : 1.0f; tr.y *= in_open(u2.y, d2.y) ? ta : 1.0f;
        ml = min(ml, min(umin3(__float_as_uint(u1.x), __float_as_uint(u2.x), __float_as_uint(u1.y)), __float_as_uint(u2.y)));
      }
    }
    if (LO && OPAQUE) mlow = umin3(mlow, m0, valid1 ? m1 : 0xFFFFFFFFu);   // the other spheres' roots at the shadow origin
    if (LO) {
      // The lane's own sphere with the reference's lower bound t' = u/|ts| > 2e-6 on both roots
      // (u^2 > 4e-12 |ts|^2): a shadow ray that crosses the surface it starts on does so at
      // t' = 1e-6/cos, which the reference does not see when cos >= 0.5 (origin shift 1e-6 + hit threshold
      // 1e-6).  The small root is formed as |ts|^2 cc / (large root) to avoid the cancellation of
      // -bt -+ sqrt(..) at that scale.  occ = 1 (no own sphere): negative discriminant, nothing happens.
      const v2f bt = pk_fma(oj, splat(oC), pk_fma(oi, splat(oB), splat(oA)));
      const v2f disc = pk_fma(bt, bt, -(d2 * splat(occ)));
      const v2f sq = pk_sqrt(disc);
      v2f u2;
      u2.x = bt.x < 0.0f ? sq.x - bt.x : -bt.x - sq.x; u2.y = bt.y < 0.0f ? sq.y - bt.y : -bt.y - sq.y;
      const v2f u1 = d2 * splat(occ) * pk_rcp(u2);                          // u1 * u2 = |ts|^2 cc
      const v2f lo2_ = d2 * splat(T_LO * T_LO);
      const v2f s1 = u1 * u1, s2 = u2 * u2;
      const bool a0 = in_open(u1.x, d2.x) & (s1.x > lo2_.x), a1 = in_open(u1.y, d2.y) & (s1.y > lo2_.y);
      const bool b0 = in_open(u2.x, d2.x) & (s2.x > lo2_.x), b1 = in_open(u2.y, d2.y) & (s2.y > lo2_.y);
      if (own >= 0 && own < NS) {
        // a root within 1e-3 (relative, squared) of the 2e-6 bound is not decided in fp32
        const bool nr = (fabsf(s1.x - lo2_.x) <= 1e-3f * lo2_.x) | (fabsf(s1.y - lo2_.y) <= 1e-3f * lo2_.y) |
                        (fabsf(s2.x - lo2_.x) <= 1e-3f * lo2_.x) | (fabsf(s2.y - lo2_.y) <= 1e-3f * lo2_.y);
        unc = nr ? RT_UNC_FORCED(3.01e38f) : unc;
        unc = fmaxf(unc, fmaxf(fmaf(okap, fabsf(bt.x), -fabsf(disc.x)), fmaf(okap, fabsf(bt.y), -fabsf(disc.y))));
      }
      if (OPAQUE) {
        m0 = (a0 | b0) ? 0u : m0; m1 = (a1 | b1) ? 0u : m1;                 // 0 = "blocked" for the final compare
      } else {
        tr.x *= a0 ? otau : 1.0f; tr.y *= a1 ? otau : 1.0f;
        tr.x *= b0 ? otau : 1.0f; tr.y *= b1 ? otau : 1.0f;
      }
    }
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {                                        // planes: infinite_plane.py:30-42
      if (!NOPLANES && (act & (0x10000u << k))) {
        const v2f den = pk_fma(oj, lo2(D2n[k]), pk_fma(oi, hi2(D01[k]), lo2(D01[k])));   // N.ts
        const v2f tau = hi2(D2n[k]) * pk_rcp(den);                          // den == 0 -> inf/NaN -> miss
        v2f u = tau * d2;                                                   // same scale as the sphere roots
        dmin = fminf(dmin, fminf(fabsf(den.x), fabsf(den.y)));              // ray parallel to the plane within rounding
        if (LO) {   // exact lower bound of the reference: tau*|ts| > 2e-6 (own plane, light behind it)
          const v2f t2 = tau * u;
          const bool nr = (fabsf(t2.x - T_LO * T_LO) <= 1e-3f * T_LO * T_LO) | (fabsf(t2.y - T_LO * T_LO) <= 1e-3f * T_LO * T_LO);
          unc = nr ? RT_UNC_FORCED(3.02e38f) : unc;
          u.x = t2.x > T_LO * T_LO ? u.x : -1.0f; u.y = t2.y > T_LO * T_LO ? u.y : -1.0f;
        }
        if (OPAQUE) { m0 = min(m0, __float_as_uint(u.x)); m1 = min(m1, __float_as_uint(u.y)); }
        else {
          const float ta = pr.tauP[k]; tr.x *= in_open(u.x, d2.x) ? ta : 1.0f; tr.y *= in_open(u.y, d2.y) ? ta : 1.0f;
          if (!LO) ml = umin3(ml, __float_as_uint(u.x), __float_as_uint(u.y));
        }
      }
    }
    if (MAXB > 0) {                                                         // boxes: cube.py:76-107 (slab roots)
      if (nB > 0) {
        const v2f tsx = pk_fma(oj, splat(up.x), pk_fma(oi, splat(right.x), splat(tl.x)));
        const v2f tsy = pk_fma(oj, splat(up.y), pk_fma(oi, splat(right.y), splat(tl.y)));
        const v2f tsz = pk_fma(oj, splat(up.z), pk_fma(oi, splat(right.z), splat(tl.z)));
        const v2f ix = pk_rcp(tsx), iy = pk_rcp(tsy), iz = pk_rcp(tsz);
        const float chord_lim = 6.75e-6f * raw_rsq(T0);
        // a direction component within 1e-3 of parallel to a slab: its 1/x carries the rounding of ts (up to
        // ~1e-3 relative here) -> not decided in fp32
        {
          const float lim = 1e-6f * T0;     // (1e-3 |tl|)^2
          const bool par = (tsx.x * tsx.x < lim) | (tsx.y * tsx.y < lim) | (tsy.x * tsy.x < lim) | (tsy.y * tsy.y < lim) |
                           (tsz.x * tsz.x < lim) | (tsz.y * tsz.y < lim);
          unc = par ? RT_UNC_FORCED(3.03e38f) : unc;
        }
#pragma unroll
        for (int k = 0; k < MAXB; ++k) {
          if (k < nB) {
            const v2f ax = splat(pr.blo[k][0]) * ix, bx = splat(pr.bhi[k][0]) * ix;
            const v2f ay = splat(pr.blo[k][1]) * iy, by = splat(pr.bhi[k][1]) * iy;
            const v2f az = splat(pr.blo[k][2]) * iz, bz = splat(pr.bhi[k][2]) * iz;
            v2f tn, tf;                                                     // slab parameters are tau already
            tn.x = fmaxf(fmaxf(fminf(ax.x, bx.x), fminf(ay.x, by.x)), fminf(az.x, bz.x));
            tn.y = fmaxf(fmaxf(fminf(ax.y, bx.y), fminf(ay.y, by.y)), fminf(az.y, bz.y));
            tf.x = fminf(fminf(fmaxf(ax.x, bx.x), fmaxf(ay.x, by.x)), fmaxf(az.x, bz.x));
            tf.y = fminf(fminf(fmaxf(ax.y, bx.y), fmaxf(ay.y, by.y)), fmaxf(az.y, bz.y));
            {
              // the ray passes an edge of the box within the slabs' rounding (relative 4e-3 with the 1e-3 limit
              // above), or a slab root lies at the sample itself (tau = 1): not decided in fp32
              // (only where it can matter: a degenerate interval [tn, tf] that overlaps the segment's (0, 1), a sample-end
              // root of a box that is not clearly missed — a box whose slab roots all lie beyond the light needs no decision)
              // (+ a chord shorter than ~4e-6 along the ray, whatever its relative size: the reference's walk steps 2e-6 from
              // one crossing to the next and sees such a chord as ONE crossing or none — a ray clipping a box edge right at
              // the hit, scaled scene 1400877.  tau = distance / |ts| and |ts| >= 2/3 |tl| here: 4.5e-6 / |ts| <= chord_lim)
              const float g0 = 4e-3f * (fabsf(tn.x) + fabsf(tf.x)) + chord_lim, g1 = 4e-3f * (fabsf(tn.y) + fabsf(tf.y)) + chord_lim;
              const bool graze0 = (fabsf(tf.x - tn.x) <= g0) & (fminf(tn.x, tf.x) < 1.004f) & (fmaxf(tn.x, tf.x) > -4e-3f);
              const bool graze1 = (fabsf(tf.y - tn.y) <= g1) & (fminf(tn.y, tf.y) < 1.004f) & (fmaxf(tn.y, tf.y) > -4e-3f);
              const bool end0 = (tf.x - tn.x >= -g0) & ((fabsf(tn.x - 1.0f) <= 4e-3f) | (fabsf(tf.x - 1.0f) <= 4e-3f));
              const bool end1 = (tf.y - tn.y >= -g1) & ((fabsf(tn.y - 1.0f) <= 4e-3f) | (fabsf(tf.y - 1.0f) <= 4e-3f));
              const bool edge = graze0 | graze1 | end0 | end1;
              unc = edge ? RT_UNC_FORCED(3.04e38f) : unc;
            }
            if (LO) {                                                      // see the sphere slow path
              const float lo0 = T_LO * T_LO * raw_rcp(d2.x), lo1 = T_LO * T_LO * raw_rcp(d2.y);   // tau^2 > 4e-12 / |ts|^2
              const bool ok0 = tn.x <= tf.x, ok1 = tn.y <= tf.y;
              const bool a0 = ok0 & in_open(tn.x, 1.0f) & (tn.x * tn.x > lo0), a1 = ok1 & in_open(tn.y, 1.0f) & (tn.y * tn.y > lo1);
              const bool b0 = ok0 & in_open(tf.x, 1.0f) & (tf.x * tf.x > lo0), b1 = ok1 & in_open(tf.y, 1.0f) & (tf.y * tf.y > lo1);
              // a root within 1 % of the 2e-6 bound — a POSITIVE root of a box that is hit (a root behind the origin is no crossing)
              const bool nr = (ok0 & (((tn.x > 0.0f) & (fabsf(tn.x * tn.x - lo0) <= 2e-2f * lo0)) | ((tf.x > 0.0f) & (fabsf(tf.x * tf.x - lo0) <= 2e-2f * lo0)))) |
                              (ok1 & (((tn.y > 0.0f) & (fabsf(tn.y * tn.y - lo1) <= 2e-2f * lo1)) | ((tf.y > 0.0f) & (fabsf(tf.y * tf.y - lo1) <= 2e-2f * lo1))));
              unc = nr ? RT_UNC_FORCED(3.05e38f) : unc;
              if (OPAQUE) { m0 = (a0 | b0) ? 0u : m0; m1 = (a1 | b1) ? 0u : m1; }
              else {
                const float ta = pr.tauB[k];
                tr.x *= a0 ? ta : 1.0f; tr.y *= a1 ? ta : 1.0f; tr.x *= b0 ? ta : 1.0f; tr.y *= b1 ? ta : 1.0f;
              }
            } else if (OPAQUE) {
              // entry root, or the exit root when the origin is inside; -1 (never in range) if the slabs miss
              const float t0 = tn.x <= tf.x ? (pr.binside[k] ? tf.x : tn.x) : -1.0f;
              const float t1 = tn.y <= tf.y ? (pr.binside[k] ? tf.y : tn.y) : -1.0f;
              m0 = min(m0, __float_as_uint(t0 * d2.x)); m1 = min(m1, __float_as_uint(t1 * d2.y));
              // The shadow origin ON a face of a box it does not belong to (a floor hit at the foot of a box standing on it:
              // offset 0, -0 or +-1e-16 after the fp64 subtraction): `binside` and the sign of the entry root are then a matter
              // of rounding, while the reference's walk decides by its 1e-6 steps (the origin shifted into the box: the exit
              // face blocks).  An entry root within the bound of zero on EITHER side is therefore not decided in fp32 (the
              // positive side already was: mlow below); -0.0 has the largest bit pattern and used to pass as "behind".
              {
                const float z0 = tn.x <= tf.x ? fabsf(tn.x * d2.x) : 3.0e38f, z1 = tn.y <= tf.y ? fabsf(tn.y * d2.y) : 3.0e38f;
                mlow = umin3(mlow, __float_as_uint(z0), valid1 ? __float_as_uint(z1) : 0xFFFFFFFFu);
              }
            } else {
              const float ta = pr.tauB[k];
              const bool ok0 = tn.x <= tf.x, ok1 = tn.y <= tf.y;
              tr.x *= (ok0 & in_open(tn.x, 1.0f)) ? ta : 1.0f; tr.y *= (ok1 & in_open(tn.y, 1.0f)) ? ta : 1.0f;
              tr.x *= (ok0 & in_open(tf.x, 1.0f)) ? ta : 1.0f; tr.y *= (ok1 & in_open(tf.y, 1.0f)) ? ta : 1.0f;
              ml = min(ml, min(umin3(__float_as_uint(tn.x * d2.x), __float_as_uint(tf.x * d2.x), __float_as_uint(tn.y * d2.y)), __float_as_uint(tf.y * d2.y)));
            }
          }
        }
      }
    }
    // the smallest positive root of the pair (a crossing at the shadow origin is not decided in fp32);
    // LO paths compare their roots with the 2e-6 bound themselves
    if (OPAQUE) { if (!LO) mlow = umin3(mlow, m0, m1 | (valid1 ? 0u : 0xFFFFFFFFu)); }   // (scalar select + v_or + v_min3)
    else mlow = min(mlow, ml);
    // (opaque scenes count BLOCKED samples in an integer: compare + add-with-carry per sample; the second sample of an
    // odd grid's last pair has m1 = 0 and counts as blocked, like the padding it is)
    if (OPAQUE) n_blocked += (uint32_t)(m0 < __float_as_uint(d2.x)) + (uint32_t)(m1 < __float_as_uint(d2.y));
    else total += tr.x + (valid1 ? tr.y : 0.0f);
  }
  lds_arrive(nb);                                                           // (the read issued by the last iteration)
  fl.unc = unc; fl.mlow = mlow; fl.dmin = dmin;
  if (OPAQUE) total = (float)(2 * n_pairs - (int)n_blocked);
  return total;
}

// The same sum for the usual configuration of an all-opaque scene, WITHOUT square roots or reciprocals: every active
// sphere has the shadow origin outside it (or is the hit's own sphere) and every sample of the light outside it.
// Along p + tau*ts the sphere's quadratic is f(tau) = |ts|^2 tau^2 + 2 (oc.ts) tau + cc with f(0) = cc >= 0 and
// f(1) > 0, so the segment crosses the sphere iff both roots lie in (0, 1): a positive discriminant and the vertex
// -(oc.ts)/|ts|^2 inside (0, 1) —
//        blocked  <=>  min( bt^2 - |ts|^2 cc,  -bt,  bt + |ts|^2 ) > 0        (bt = oc.ts; one v_min3_f32 per sample)
// (the hit's own sphere, cc = +0: roots 0 and -2bt/|ts|^2; the far one is in (0, 1) iff bt < 0, given f(1) > 0 — the
// same expression).  A plane: 0 < num/den < 1  <=>  min(num den, den^2 - num den) > 0.  The maximum of those minima
// over the primitives is positive iff the sample is blocked: 11 full-rate instructions per (sample pair, sphere)
// instead of 12 of which two are v_sqrt_f32 (quarter rate), and no rcp for the planes.  The only decision that can
// come within its fp32 error bound is the sign of the discriminant (tracked as before, LsFlags::unc) and a segment
// parallel to a plane (LsFlags::dmin): bt = 0 or bt = -|ts|^2 with a positive discriminant would need cc < 0 or
// f(1) < 0.  The caller checks the preconditions per (wave, light) and uses ls_light_sum otherwise.
template <int NS, int MAXP, bool SDENSE, bool NOPLANES>
__device__ __forceinline__ float ls_light_sum_simple(v2f (&AB)[NS], v2f (&Cc)[NS], const float (&kapS)[NS], v2f (&D01)[MAXP], v2f (&D2n)[MAXP],
                                                     uint32_t act, const float4* __restrict__ cells, int n_cells, JitKey key, float cell32,
                                                     float T0, LsFlags& fl) {
  const int n_pairs = (n_cells + 1) >> 1;
  float unc = fl.unc, dmin = fl.dmin;
  v2f blocked = mk2(0.0f, 0.0f);                                            // blocked samples, counted in floats (see below)
  v2f big; big.x = 8.507059173e37f;                                         // 2^126 (.y is never read: op_sel broadcast)
  ls_v4f nb = {0.0f, 0.0f, 0.0f, 0.0f};
  // loop control kept to a pointer, a counter and a branch: in the one- and two-sphere instantiations, which shade most
  // of a frame's penumbra lanes, 12 scalar instructions per iteration (next-entry select, padding-sample select, counters)
  // were a third of the 28 + 12 instructions a wave issues per sample pair.  The table has an entry behind the last pair
  // (read ahead, never used); the padding sample of an odd grid has NaN offsets in the table: every quantity of that
  // sample is NaN, v_min3 / v_max3 ignore NaN operands (IEEE minNum / maxNum), and clamp(NaN) = 0 — it counts as
  // unblocked without a select, so the sum below starts from n_cells, not from 2 n_pairs
  const float4* cp = cells;
  lds_issue(nb, cp);
  uint32_t xa = key.a + JIT_PHI_A, xb = key.b + JIT_PHI_B;
#pragma unroll 1
  for (int q = n_pairs; q > 0; --q) {
#pragma unroll
    for (int k = 0; k < NS; ++k) { asm("" : "+v"(AB[k])); asm("" : "+v"(Cc[k])); }      // keeps the op_sel broadcasts in the loop (see ls_light_sum)
#pragma unroll
    for (int k = 0; k < MAXP; ++k) { asm("" : "+v"(D01[k])); asm("" : "+v"(D2n[k])); }
    lds_arrive(nb);
    v2f oi, oj;                                                             // ray_tracer.py:224-225
    oi.x = fmaf((float)xa, cell32, nb.x); oi.y = fmaf((float)(xa + JIT_PHI_A), cell32, nb.y);
    oj.x = fmaf((float)xb, cell32, nb.z); oj.y = fmaf((float)(xb + JIT_PHI_B), cell32, nb.w);
    asm volatile("" : "+v"(oi), "+v"(oj));                                  // (complete before the next read overwrites nb)
    ++cp;
    lds_issue(nb, cp);
    xa += 2u * JIT_PHI_A; xb += 2u * JIT_PHI_B;
    // |ts|^2 = |tl|^2 + oi^2 + oj^2: right and up are perpendicular to tl by construction (ray_tracer.py:207-213); the
    // 2 oi (tl.right) + 2 oj (tl.up) of the fp32 basis is below 3e-7 |oi| |tl|, i.e. < 1e-7 relative with the light
    // farther than 3 RL — inside the 54 u the discriminant's bound allows for |ts|^2 cc (LsTol)
    const v2f d2 = pk_fma(oi, oi, pk_fma(oj, oj, splat(T0)));
    v2f W;
    bool first = true;
#pragma unroll
    for (int k = 0; k < NS; ++k) {                                          // spheres: sphere.py:49-71
      if (!SDENSE && !(act & (1u << k))) continue;
      const v2f bt = pk_fma(oj, lo2(Cc[k]), pk_fma(oi, hi2(AB[k]), lo2(AB[k])));         // oc.ts
      const v2f disc = pk_fma(bt, bt, -(d2 * hi2(Cc[k])));
      const v2f s = bt + d2;
      unc = fmaxf(fmaxf(unc, fmaf(kapS[k], fabsf(bt.x), -fabsf(disc.x))), fmaf(kapS[k], fabsf(bt.y), -fabsf(disc.y)));
      const float w0 = fminf(fminf(disc.x, -bt.x), s.x), w1 = fminf(fminf(disc.y, -bt.y), s.y);
      if (SDENSE && first) { W.x = w0; W.y = w1; first = false; }           // (compile time: no initial value to fold in)
      else { if (first) { W = mk2(-1.0f, -1.0f); first = false; } W.x = fmaxf(W.x, w0); W.y = fmaxf(W.y, w1); }
    }
    if (first) W = mk2(-1.0f, -1.0f);
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {                                        // planes: infinite_plane.py:30-42
      if (!NOPLANES && (act & (0x10000u << k))) {
        const v2f den = pk_fma(oj, lo2(D2n[k]), pk_fma(oi, hi2(D01[k]), lo2(D01[k])));   // N.ts
        const v2f x = hi2(D2n[k]) * den;                                    // num den
        const v2f y = pk_fma(den, den, -x);
        dmin = fminf(dmin, fminf(fabsf(den.x), fabsf(den.y)));
        W.x = fmaxf(W.x, fminf(x.x, y.x)); W.y = fmaxf(W.y, fminf(x.y, y.y));
      }
    }
    // blocked <=> W > 0, counted without compares: clamp(W * 2^126) is exactly 1 for every normal W > 0 and exactly 0 for
    // W <= 0 or NaN (one v_pk_mul_f32 with the clamp bit + one v_pk_add_f32 for the two samples); a denormal W > 0 would
    // add a fraction — the caller treats a sum that is not a whole number as undecided
    v2f ind;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0] clamp" : "=v"(ind) : "v"(W), "v"(big));
    blocked += ind;
  }
  lds_arrive(nb);                                                           // (the read issued by the last iteration)
  fl.unc = unc; fl.dmin = dmin;
  return (float)n_cells - (blocked.x + blocked.y);
}

#ifndef LS_WAVES
#define LS_WAVES 2
#endif
template <int MAXS, int MAXP, int MAXB, bool OPAQUE, int FALL = 0>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(LS_WAVES))) void k_shade_ls(DevScene sc, DevCamera cam, FrameParams fp, int level, RayQueue q,
                                                    RayQueue nq, const HitRec* __restrict__ hits, Counters* C,
                                                    float* __restrict__ fb, FixRec* __restrict__ fix) {
  constexpr int MAXN = MAXS + MAXP + MAXB;
  __shared__ Prim s_prims[MAXN];
  // cell base offsets per light and SAMPLE PAIR q (cells 2q, 2q+1): (b0.x, b1.x, b0.y, b1.y), so that one 16-byte LDS
  // read delivers the two packed operands of the pair (the second cell of an odd grid's last pair repeats cell 0)
  __shared__ float4 s_cells[LS_MAX_CELLS / 2 + LS_MAX_LIGHTS];
  __shared__ float s_lightf[LS_MAX_LIGHTS][LIGHT_STRIDE];    // lights as fp32 (the shadow origin's epsilon lives in cc / num)
  __shared__ float4 s_sphf[MAXS];                             // sphere centres + radii as fp32
  __shared__ float4 s_plnf[MAXP > 0 ? MAXP : 1];               // plane normals as fp32 (the light loop reads them per (hit, light))
  // low parts: centre = s_sphf + s_sphl, light position = s_lightf[0..2] + s_lightl to ~1e-15; with the hit point split
  // the same way (pf + pl), oc and tl carry only a few ulp of THEIR size instead of ulps of the coordinates' size
  __shared__ float4 s_sphl[MAXS];
  __shared__ float s_lightl[LS_MAX_LIGHTS][4];
  // per light: spheres that (some sample of) the light may be inside of / outside of — hit independent
  __shared__ uint32_t s_lin[LS_MAX_LIGHTS], s_lout[LS_MAX_LIGHTS];
  // per light: primitives the light's sample square may touch (spheres bit k, planes bit 16+k, boxes bit 24+k): a
  // crossing at the sample end of a shadow segment (tau = 1) is then possible, which fp32 does not decide
  __shared__ uint32_t s_lstr[LS_MAX_LIGHTS];
  __shared__ float s_labs[LS_MAX_LIGHTS];                     // |Lp|_1
  // Whether a root lies before or behind the SAMPLE END of a shadow segment (u < |ts|^2) is an fp32 decision whose error
  // grows with the hit's distance from the sphere (the roots -bt +- sqrt(bt^2 - |ts|^2 cc) cancel); s_lstr only excludes
  // lights within a fixed 1e-4 of a surface.  With w = S - c the margin of that comparison is |ts|^2 g (|w| + r) / |sq + ts.w|
  // >= |ts| g / 2 for a sample at distance g from the surface, its error e_b + 4u |ts|^2 + the square root's, which is
  // <= sqrt(kap |bt| + kap0) once the discriminant is decided; with LsTol's e_b <= 26u Q, kap <= 106u Q (Q = |oc||ts|) that is
  // <= 2.6e-3 Q, and |ts| >= tsm / 2 here: the decision is safe while |oc| < 96 g.  Per light: 80 x the smallest distance of
  // its sample square from a sphere's surface; hits farther than that from the spheres leave the pair to the exact walk
  // unless the loop makes no such decision (ls_light_sum_simple).  (Random scene 200792: a light 0.01 inside a sphere, a
  // reflected ray's hit 434 units away, the far root counted as a second crossing: 2 levels off.)
  __shared__ float s_lend[LS_MAX_LIGHTS];
  // the same per-light values packed for the light loop, which reads a light with THREE 16-byte broadcast reads one light
  // ahead and one more after the sample loop (instead of ~17 four-byte reads in ~6 read + wait groups: a wave's LDS
  // latency is exposed at two waves per SIMD — 37 % of the culled frame's wave cycles were s_waitcnt waits, 250 LDS
  // instructions per 64-hit chunk at ~55 cycles each):
  //   [0] position (high parts), radius   [1] position (low parts), |Lp|_1   [2] bits of s_lin, s_lout, s_lstr, shadow intensity
  //   [3] colour, specular intensity
  __shared__ float4 s_lt[LS_MAX_LIGHTS][4];
  __shared__ double s_matd[LS_MAX_MATS][MAT_STRIDE];
  // exact fp64 geometry of each lane's hit, parked for the spawn step and the re-test queue (hit point, normal, ray
  // direction, weight): cheaper than keeping 24 VGPRs live through the light loop or recomputing the divisions
  __shared__ double s_geo[12][BLOCK];
  __shared__ double s_ct[2][BLOCK];            // fall: t of each lane's transmitted / reflected child that hit something
  __shared__ uint32_t s_ck[2][BLOCK];          //       its record | box face << 8
  const int n_sh = fp.n_shadow;
  const int n_side = n_sh > 1 ? n_sh : 1;
  const int n_cells = n_side * n_side;
  const bool brute = fp.brute != 0u;
  // Level 0 fused with the nearest-hit query (fp.fuse0, LDS-resident scenes; ray_tracer.py:491-519 shades a batch right
  // after its query too): a chunk is 64 pixels — an 8x8 tile —, every lane forms its camera ray, runs the literal fp64
  // nearest() over the staged records and shades its own hit.  No k_trace launch, no 16-byte HitRec per pixel written and
  // read back, no global load at the start of a chunk; pixels that miss get the background here.
  const bool fused = FALL == 2 ? true : FALL == 3 ? false : (level == 0 && (FALL != 0 || fp.fuse0 != 0u));
  // fp.fuse0 == 2: the frame has no k_trace launch at all.  The children of a chunk's hits are traced by the lanes that
  // spawn them (right after the hit's geometry is known, before the light loop), misses add their background term, and
  // the HITS are appended — ray and hit in one 64-byte HitRay — to the next level's queue, which the next launch of this
  // kernel reads instead of HitRec + ray queue.  ray_tracer.py:491-595 as one kernel per level.
  // (FALL is a template parameter so that the instantiations for fused frames carry none of the queue path's arguments
  // and code: ten pointers per ray queue, HitRec loads, slot reservations.  0: frames with k_trace launches; 2: level 0
  // of a fused frame; 3: its deeper levels; 1: any level of a fused frame)
  constexpr bool fall = FALL != 0;
  const HitRay* __restrict__ hin = reinterpret_cast<const HitRay*>(q.ox);
  HitRay* __restrict__ hout = reinterpret_cast<HitRay*>(nq.ox);
  uint32_t* const stripes = fall ? C->stripes3[level % 3] : C->stripes;
  __shared__ uint32_t s_segfirst[NSEG + 1], s_segn[NSEG];      // fall, level >= 1: first chunk id / hits of each queue segment
  const uint32_t seg_cap = fp.queue_cap / NSEG;
  if (!fused && !fall && C->n_hits[level] == 0u) return;        // (uniform over the grid: an empty level)
  if (fall) {
    if (FALL != 2 && level >= 1 && threadIdx.x < 64) {          // lane q: segment q; inclusive scans over the wave (as in k_trace)
      const uint32_t raw = C->segcnt3[level % 3][threadIdx.x * 32];
      const uint32_t nq_ = min(raw, seg_cap);
      uint32_t ch = (nq_ + 63u) / 64u, tot = nq_, worst = raw;
#pragma unroll
      for (int o_ = 1; o_ < 64; o_ <<= 1) {
        const uint32_t a = __shfl_up(ch, o_), b = __shfl_up(tot, o_);
        if ((int)threadIdx.x >= o_) { ch += a; tot += b; }
        worst = max(worst, (uint32_t)__shfl_xor((int)worst, o_));
      }
      s_segfirst[threadIdx.x + 1] = ch; s_segn[threadIdx.x] = nq_;
      if (threadIdx.x == 0) s_segfirst[0] = 0;
      if (blockIdx.x == 0 && threadIdx.x == 63) { C->n_hits[level] = tot; atomicMax(&C->seg_need, worst); }
    }
    if (blockIdx.x == 0) {
      for (int i = threadIdx.x; i < NSEG; i += BLOCK) C->segcnt3[(level + 2) % 3][i * 32] = 0u;
      for (int i = threadIdx.x; i < NSTRIPE; i += BLOCK) C->stripes3[(level + 1) % 3][i * 32] = 0u;
    }
  }
  {
    const int words = sc.n * (int)(sizeof(Prim) / 16);
    const uint4* src = reinterpret_cast<const uint4*>(sc.prims);
    uint4* dst = reinterpret_cast<uint4*>(s_prims);
    for (int i = threadIdx.x; i < words; i += BLOCK) dst[i] = src[i];
    for (int i = threadIdx.x; i < sc.nL * LIGHT_STRIDE; i += BLOCK) (&s_lightf[0][0])[i] = (float)sc.lights[i];
    for (int i = threadIdx.x; i < MAXS; i += BLOCK) {
      s_sphf[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); s_sphl[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (i < sc.nS) {
        const double* a = sc.prims[i].a;
        const float4 h = make_float4((float)a[0], (float)a[1], (float)a[2], (float)a[3]);
        s_sphf[i] = h;
        s_sphl[i] = make_float4((float)(a[0] - (double)h.x), (float)(a[1] - (double)h.y), (float)(a[2] - (double)h.z), 0.0f);
      }
    }
    for (int i = threadIdx.x; i < MAXP; i += BLOCK) {
      s_plnf[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (i < sc.nP) { const double* a = sc.prims[sc.nS + i].a; s_plnf[i] = make_float4((float)a[0], (float)a[1], (float)a[2], 0.0f); }
    }
    for (int i = threadIdx.x; i < sc.nL * 3; i += BLOCK) {
      const int li = i / 3, c = i - li * 3;
      const double v_ = sc.lights[LIGHT_STRIDE * li + c];
      s_lightl[li][c] = (float)(v_ - (double)(float)v_);
    }
    for (int i = threadIdx.x; i < sc.nM * MAT_STRIDE; i += BLOCK) (&s_matd[0][0])[i] = sc.mats[i];
    // per-light table of cell base offsets: -radius/2 + cell*(i or j)   (ray_tracer.py:215, 224-225)
    {
      const int n_pairs_ = (n_cells + 1) >> 1;
      for (int e = threadIdx.x; e < sc.nL * n_pairs_; e += BLOCK) {
        const int li = e / n_pairs_, q_ = e - li * n_pairs_;
        const float radius = (float)sc.lights[LIGHT_STRIDE * li + 8];
        const float cell = radius / (float)n_side, half = -0.5f * radius;
        const bool pad = 2 * q_ + 1 >= n_cells;           // the second sample of an odd grid's last pair: NaN offsets (see ls_light_sum_simple)
        const int c0_ = 2 * q_, c1_ = pad ? 0 : 2 * q_ + 1;
        const int i0 = c0_ / n_side, j0 = c0_ - i0 * n_side, i1 = c1_ / n_side, j1 = c1_ - i1 * n_side;
        const float nan_ = __uint_as_float(0x7FC00000u);
        s_cells[e] = n_sh > 1 ? make_float4(fmaf(cell, (float)i0, half), pad ? nan_ : fmaf(cell, (float)i1, half), fmaf(cell, (float)j0, half), pad ? nan_ : fmaf(cell, (float)j1, half))
                              : make_float4(0.0f, pad ? nan_ : 0.0f, 0.0f, pad ? nan_ : 0.0f);
      }
    }
    for (int li = threadIdx.x; li < sc.nL; li += BLOCK) {
      const double* L = sc.lights + LIGHT_STRIDE * li;
      const float RL = n_sh > 1 ? 0.7072f * (float)L[8] * 1.001f : 0.0f;
      const float labs = (float)(fabs(L[0]) + fabs(L[1]) + fabs(L[2]));
      const double RS = (double)RL + 1e-4 * (1.0 + labs);        // reach of the sample square + slack (see `far` in the light loop)
      uint32_t in_ = 0u, out_ = 0u, str_ = 0u;
      double gapmin = 1e30;
      for (int k = 0; k < sc.nS && k < MAXS; ++k) {
        const double dx = L[0] - sc.prims[k].a[0], dy = L[1] - sc.prims[k].a[1], dz = L[2] - sc.prims[k].a[2];
        const double l2d = (dx * dx + dy * dy) + dz * dz;
        const float l2 = (float)l2d, r = (float)sc.prims[k].a[3];
        const float rin = fmaxf(r - RL, 0.0f);
        if (l2 < (r + RL) * (r + RL) * 1.001f) in_ |= 1u << k;
        if (l2 > rin * rin * 0.999f) out_ |= 1u << k;
        if (fabs(sqrt(l2d) - sc.prims[k].a[3]) <= RS * (1.0 + 1e-5 * sc.prims[k].a[3])) str_ |= 1u << k;
        gapmin = fmin(gapmin, fabs(sqrt(l2d) - sc.prims[k].a[3]));
      }
      for (int k = 0; k < sc.nP && k < MAXP; ++k) {
        const Prim& pl = sc.prims[sc.nS + k];
        if (fabs(((L[0] * pl.a[0] + L[1] * pl.a[1]) + L[2] * pl.a[2]) - pl.a[3]) <= RS * (1.0 + 1e-5 * fabs(pl.a[3]))) str_ |= 0x10000u << k;
      }
      for (int k = 0; k < sc.nB && k < MAXB; ++k) {
        const Prim& b = sc.prims[sc.nS + sc.nP + k];
        bool in_big = true, in_small = true;
        for (int ax = 0; ax < 3; ++ax) {
          const double g = RS * (1.0 + 1e-5 * (fabs(b.a[ax]) + fabs(b.a[3 + ax])));
          in_big &= (L[ax] >= b.a[ax] - g) & (L[ax] <= b.a[3 + ax] + g);
          in_small &= (L[ax] > b.a[ax] + g) & (L[ax] < b.a[3 + ax] - g);
        }
        if (in_big && !in_small) str_ |= 0x1000000u << k;
      }
      s_lin[li] = in_; s_lout[li] = out_; s_lstr[li] = str_; s_labs[li] = labs;
      s_lend[li] = (float)fmin(80.0 * fmax(gapmin - 1.001 * (double)RL, 0.0), 1e30);
      {
        const float hx = (float)L[0], hy = (float)L[1], hz = (float)L[2];
        s_lt[li][0] = make_float4(hx, hy, hz, (float)L[8]);
        s_lt[li][1] = make_float4((float)(L[0] - (double)hx), (float)(L[1] - (double)hy), (float)(L[2] - (double)hz), labs);
        s_lt[li][2] = make_float4(__uint_as_float(in_), __uint_as_float(out_), __uint_as_float(str_), (float)L[7]);
        s_lt[li][3] = make_float4((float)L[3], (float)L[4], (float)L[5], (float)L[6]);
      }
    }
    __syncthreads();
  }
  const Prim* P = s_prims;
  const int nS = sc.nS, nP = sc.nP, nB = sc.nB;
  float umbra_t0_max = 3.0e38f;                              // 1000 * (smallest sphere radius)^2
  for (int k = 0; k < nS; ++k) umbra_t0_max = fminf(umbra_t0_max, 1000.0f * s_sphf[k].w * s_sphf[k].w);
  const bool qin = FALL == 2 ? false : FALL == 3 ? true : (fall && level >= 1);                       // this level's hits come from the HitRay queue
  uint32_t count = fused ? C->n_rays[0] : (qin ? s_segfirst[NSEG] * 64u : C->n_hits[level]);
  if (!qin && count > fp.hit_cap) count = fp.hit_cap;
  uint32_t fused_hits = 0u;                                  // hits of this wave's chunks (fused level 0): one atomic per wave at the end
  uint32_t my_children = 0u;                                 // fall: child rays traced by this wave (wave-uniform)
  const float inv_cells = 1.0f / (float)n_cells;
  const bool spawn_level = level < fp.max_depth;
  const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
  const uint32_t all_prims = ((1u << nS) - 1u) | (((1u << nP) - 1u) << 16);

  // Waves work independently (no block barrier after the staging above): each takes 64-hit chunks — at level 0 an
  // 8x8 pixel tile — until the level's hit list is used up.  Chunk costs differ by several x (penumbra vs fully
  // lit), so chunks are handed out dynamically; a single device counter sustains only ~88 atomics per microsecond
  // (a 4K frame's level 0 has 130k chunks), so the list is cut into NSTRIPE stripes with a cursor each: a wave works
  // through its home stripe, then moves to the stripe with the most chunks left.
  const uint32_t n_chunks = qin ? s_segfirst[NSEG] : (count + 63u) / 64u;
  const uint32_t cps = (n_chunks + NSTRIPE - 1) / NSTRIPE;         // chunks per stripe
  uint32_t stripe = (blockIdx.x * (BLOCK / 64) + wave) % NSTRIPE;
  // RT_DEBUG_STATS bit 1 << 20 (audit build): wave cycles per phase -> Counters::dbg (hit setup, light setup, sample loops, all)
  const bool prof = (RT_DBG(fp) & (1u << 20)) != 0u;
  long long ck_hit = 0, ck_setup = 0, ck_loop = 0, ck_sync = 0;
  const long long ck_begin = prof ? clock64() : 0;
  // A global atomic with return takes ~1 us and a chunk only ~6: the cursor of the NEXT chunk is fetched while the
  // current one is shaded (issued after the chunk's global loads, consumed at its end), and so is the queue slot
  // reservation of the chunk's children.
  auto steal = [&](uint32_t i_) -> uint32_t {      // chunk for cursor value i_ of `stripe`; when that stripe is used up: go to the fullest one
    for (;;) {
      if (i_ < cps && stripe * cps + i_ < n_chunks) return stripe * cps + i_;
      // lane s looks at stripe s (device-scope load: not from this CU's vector cache); the wave moves to the stripe
      // with the most chunks left — one load and one atomic per move, one load to learn that the level is finished
      int left = 0;
      if (lane < (uint32_t)NSTRIPE) {
        const uint32_t cur = __hip_atomic_load(&stripes[lane * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t first = lane * cps;
        const uint32_t n_s = first < n_chunks ? min(cps, n_chunks - first) : 0u;
        left = cur < n_s ? (int)(n_s - cur) : 0;
      }
      int key = (left << 6) | (int)(63u - lane);     // most chunks left, lowest stripe on ties
#pragma unroll
      for (int o_ = 32; o_ > 0; o_ >>= 1) key = max(key, __shfl_xor(key, o_));
      if ((key >> 6) <= 0) return 0xFFFFFFFFu;
      stripe = 63u - (uint32_t)(key & 63);
      uint32_t j_ = 0;
      if (lane == 0) j_ = atomicAdd(&stripes[stripe * 32], 1u);
      i_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)j_);
    }
  };
  uint32_t chunk;
  {
    uint32_t j_ = 0;
    if (lane == 0) j_ = atomicAdd(&stripes[stripe * 32], 1u);
    chunk = steal((uint32_t)__builtin_amdgcn_readfirstlane((int)j_));
  }
  while (chunk != 0xFFFFFFFFu) {
    uint32_t h = chunk * 64u + lane;
    bool active = h < count;
    if (qin) {                                 // chunk -> (segment, offset): the last segment whose first chunk id is <= chunk
      int lo_ = 0, hi_ = NSEG;
      while (hi_ - lo_ > 1) { const int mid = (lo_ + hi_) >> 1; if (s_segfirst[mid] <= chunk) lo_ = mid; else hi_ = mid; }
      const uint32_t off = (chunk - s_segfirst[lo_]) * 64u + lane;
      active = off < s_segn[lo_];
      h = (uint32_t)lo_ * seg_cap + off;       // slot in the queue
    }
    long long ck_t = prof ? clock64() : 0;
    // (the row set's divisors pass through an empty asm statement per chunk: otherwise the reciprocals of the pixel-index
    // divisions — loop invariant, one VGPR each — are kept across the whole chunk loop, where they are spilled to scratch
    // and every chunk waits ~400 cycles for their reload; recomputing them costs ~10 instructions each)
    RowSet rs_l = fp.rs;
    asm volatile("" : "+s"(rs_l.W), "+s"(rs_l.rows), "+s"(rs_l.block_rows));
    // ---------------------------------------------------------------- phase 1: per-hit setup ---
    v2f a_AB[MAXS], a_Cc[MAXS], a_Sg[(MAXS + 1) / 2];
    float a_tauS[MAXS];
    v2f a_D01[MAXP], a_D2n[MAXP];
    float a_tauP[MAXP];
    float a_blo[MAXB > 0 ? MAXB : 1][3], a_bhi[MAXB > 0 ? MAXB : 1][3], a_tauB[MAXB > 0 ? MAXB : 1];
    bool a_binside[MAXB > 0 ? MAXB : 1];
    float a_kapS[MAXS];
    LsPrims<MAXS, MAXP, MAXB, OPAQUE> pr{a_AB, a_Cc, a_Sg, a_tauS, a_kapS, a_D01, a_D2n, a_tauP, a_blo, a_bhi, a_tauB, a_binside};
    d3 p = mk3(0, 0, 0);
    f3 pf = mkf(0, 0, 0), pl = mkf(0, 0, 0);
    f3 nf = mkf(0, 0, 1), v = mkf(0, 0, 1), wgt = mkf(0, 0, 0);
    int mat_idx = 0, own = -1;
    bool own_is_plane = false;
    uint32_t lp = 0, gpix = 0;
    HitRec hr; hr.slot = h; hr.prim = 0; hr.t = 0.0;
    RayIn r; r.o = r.d = r.w = mk3(0, 0, 0); r.lp = 0;
    if (fused) {
      // (the idle lanes of the frame's last chunk trace a copy of its last pixel: nearest_coherent votes per wave)
      r = load_ray(0, active ? h : count - 1u, cam, rs_l, q);
      int k_ = -1;
      hr.t = nearest_coherent(P, nS, nP, nB, r.o, r.d, &k_, fp.plane_thr);
      hr.prim = k_;
      if (active) {
        if (k_ < 0) {                          // ray_tracer.py:500-502 (0 + 1*bg == bg exactly)
          float* dst = fb + 3 * (size_t)r.lp;
          dst[0] = (float)sc.bg[0]; dst[1] = (float)sc.bg[1]; dst[2] = (float)sc.bg[2];
          active = false;
        }
      }
      fused_hits += (uint32_t)__popcll(__ballot(active));
    } else if (active && !qin) {
      hr = hits[h];
      r = load_ray(level, hr.slot, cam, rs_l, q);
    }
    if (fused && !__any(active)) {             // a tile of background pixels: nothing to shade
      uint32_t nc_ = 0;
      if (lane == 0) nc_ = atomicAdd(&stripes[stripe * 32], 1u);
      chunk = steal((uint32_t)__builtin_amdgcn_readfirstlane((int)nc_));
      continue;
    }
    if (active) {
      d3 n, hp;
      if (qin) {
        const HitRay hy = hin[h];
        hp = mk3(hy.hp[0], hy.hp[1], hy.hp[2]);
        r.d = mk3(hy.d[0], hy.d[1], hy.d[2]);
        r.w = mk3((double)hy.w[0], (double)hy.w[1], (double)hy.w[2]);
        r.lp = hy.pk & (HITRAY_MAX_PIXELS - 1u);
        hr.prim = (int)((hy.pk >> HITRAY_PIX_BITS) & (nB > 0 ? 15u : 31u));
        const Prim& op_ = P[hr.prim];
        if (op_.type == 0) {                     // sphere.py:80-90 with the stored hit point
          n = mk3((hp.x - op_.a[0]) / op_.a[3], (hp.y - op_.a[1]) / op_.a[3], (hp.z - op_.a[2]) / op_.a[3]);
          if (dot3(r.d, n) > 0) n = mk3(-n.x, -n.y, -n.z);
        } else if (op_.type == 1) n = plane_normal(op_, r.d);
        else {                                   // the face cube.py:113-134 chose, kept by the lane that traced the ray
          const uint32_t f_ = hy.pk >> 29;
          const double sg_ = (f_ & 4u) ? 1.0 : -1.0;
          n = mk3((f_ & 3u) == 0u ? sg_ : 0.0, (f_ & 3u) == 1u ? sg_ : 0.0, (f_ & 3u) == 2u ? sg_ : 0.0);
        }
        v = normalize_clamped(mkf((float)-r.d.x, (float)-r.d.y, (float)-r.d.z));       // (o - hp) / |o - hp| = -d
      } else {
        const Prim& op_ = P[hr.prim];
        n = prim_normal(op_, r.o, r.d, hr.t);
        hp = mk3(r.o.x + hr.t * r.d.x, r.o.y + hr.t * r.d.y, r.o.z + hr.t * r.d.z);
        v = normalize_clamped(mkf((float)(r.o.x - hp.x), (float)(r.o.y - hp.y), (float)(r.o.z - hp.z)));
      }
      const Prim& op = P[hr.prim];
      p = mk3(hp.x + n.x * EPS, hp.y + n.y * EPS, hp.z + n.z * EPS);
      nf = tof(n);
      mat_idx = op.mat; own = hr.prim; own_is_plane = op.type == 1;
      lp = r.lp; gpix = global_pixel(rs_l, lp);
      wgt = tof(r.w);
      pf = tof(p);
      pl = mkf((float)(p.x - (double)pf.x), (float)(p.y - (double)pf.y), (float)(p.z - (double)pf.z));
      const uint32_t t_ = threadIdx.x;
      s_geo[0][t_] = hp.x; s_geo[1][t_] = hp.y; s_geo[2][t_] = hp.z;
      s_geo[3][t_] = n.x; s_geo[4][t_] = n.y; s_geo[5][t_] = n.z;
      if (spawn_level) {
        s_geo[6][t_] = r.d.x; s_geo[7][t_] = r.d.y; s_geo[8][t_] = r.d.z;
        s_geo[9][t_] = r.w.x; s_geo[10][t_] = r.w.y; s_geo[11][t_] = r.w.z;
      }
    }
    // (all global loads of this chunk are done) fetch the next chunk's cursor and reserve the children's queue slots
    uint32_t next_cursor = 0;
    if (lane == 0) next_cursor = atomicAdd(&stripes[stripe * 32], 1u);
    const double* mat_early = s_matd[mat_idx];
    const bool needT = spawn_level && active && (mat_early[10] > EPS);
    const bool needR = spawn_level && active && (mat_early[6] > EPS || mat_early[7] > EPS || mat_early[8] > EPS);
    const unsigned long long mT = __ballot(needT), mR = __ballot(needR);
    const uint32_t nT = (uint32_t)__popcll(mT), nR = (uint32_t)__popcll(mR);
    const uint32_t seg = spawn_segment(chunk, count);
    uint32_t spawn_base = 0;
    unsigned long long mTh = 0ull, mRh = 0ull;       // fall: children that hit something
    bool hitT = false, hitR = false;
    if (fall) {
      if (nT + nR != 0u) {
        const uint32_t t_ = threadIdx.x;
        // the child rays exactly as the queue path forms them (ray_tracer.py:565-595), traced by the lanes that spawn them:
        // transmitted first, then reflected; a hit is kept (t, record, box face) in LDS until the end of the chunk, when its
        // queue slot is known; a miss adds weight * background (ray_tracer.py:500-502) to the pixel together with the
        // hit's own colour (at level 0 that is a plain store, which must not come after an atomic add of the child's term)
        auto trace_child = [&](bool need, bool refl, bool* hit_) __attribute__((always_inline)) {
          if (!need) return;
          const d3 hp_ = mk3(s_geo[0][t_], s_geo[1][t_], s_geo[2][t_]), n_ = mk3(s_geo[3][t_], s_geo[4][t_], s_geo[5][t_]);
          const d3 d_ = mk3(s_geo[6][t_], s_geo[7][t_], s_geo[8][t_]);
          d3 oc_, dc_;
          if (refl) {
            const double dn = dot3(d_, n_);                                  // reflect_batch, :34-36
            dc_ = mk3(d_.x - (2 * dn) * n_.x, d_.y - (2 * dn) * n_.y, d_.z - (2 * dn) * n_.z);
            oc_ = mk3(hp_.x + n_.x * EPS, hp_.y + n_.y * EPS, hp_.z + n_.z * EPS);
          } else {
            dc_ = d_;
            oc_ = mk3(hp_.x - n_.x * EPS, hp_.y - n_.y * EPS, hp_.z - n_.z * EPS);
          }
          int k_ = -1;
          const double tc_ = nearest_coherent(P, nS, nP, nB, oc_, dc_, &k_, fp.plane_thr);
          if (k_ >= 0) {
            uint32_t face = 0u;
            if (MAXB > 0 && P[k_].type == 2) {
              d3 nb_; box_hit<true>(P[k_], oc_, dc_, &nb_);
              face = nb_.x != 0.0 ? (nb_.x > 0 ? 4u : 0u) : (nb_.y != 0.0 ? (nb_.y > 0 ? 5u : 1u) : (nb_.z > 0 ? 6u : 2u));
            }
            s_ct[refl ? 1 : 0][t_] = tc_;
            s_ck[refl ? 1 : 0][t_] = (uint32_t)k_ | (face << 8);
            *hit_ = true;
          }
        };
        if (nT != 0u) trace_child(needT, false, &hitT);
        if (nR != 0u) trace_child(needR, true, &hitR);
        my_children += nT + nR;
        mTh = __ballot(hitT); mRh = __ballot(hitR);
        const uint32_t nh_ = (uint32_t)__popcll(mTh) + (uint32_t)__popcll(mRh);
        if (lane == 0 && nh_ != 0u) spawn_base = atomicAdd(&C->segcnt3[(level + 1) % 3][seg * 32], nh_);
      }
    } else if (lane == 0 && nT + nR != 0u) spawn_base = atomicAdd(&C->segcnt[(level + 1) & 1][seg * 32], nT + nR);
    bool near_scene = true;    // |oc|^2 within 1000 r^2 of every sphere (see the umbra claim below)
    uint32_t own_in_bit = 0u, own_out_bit = 0u;   // 1 << own if the own sphere is seen from inside / outside
    float oc2max = 0.0f;                          // largest |oc|^2 over the spheres (error bounds)
    uint32_t bad_cc = 0u;                         // spheres the shadow origin is inside of, or whose surface it lies on without being the hit's own (ls_light_sum_simple's precondition)
                                                  // bits 16+: planes through the shadow origin within rounding (see below)
    float a_sgn[MAXS];
    // (LDS reads one slot ahead, here and in the light loop: with two waves per SIMD a read + wait per slot stalls the wave
    // for the LDS latency each time; the record array has MAXS + MAXP + MAXB entries, so slot k + 1 always exists)
    double pc0 = P[0].a[0], pc1 = P[0].a[1], pc2 = P[0].a[2], pc4 = P[0].a[4];
#pragma unroll
    for (int k = 0; k < MAXS; ++k) {
      a_Cc[k] = mk2(0.0f, 1.0f); a_sgn[k] = -1.0f; a_AB[k] = mk2(0.0f, 0.0f);
      const double c0_ = pc0, c1_ = pc1, c2_ = pc2, r2_ = pc4;
      if (k + 1 < MAXS) { pc0 = P[k + 1].a[0]; pc1 = P[k + 1].a[1]; pc2 = P[k + 1].a[2]; pc4 = P[k + 1].a[4]; }
      if (k < nS) {
        const Prim& s_ = P[k];
        const double ox = p.x - c0_, oy = p.y - c1_, oz = p.z - c2_;
        const float cc = (float)(((ox * ox + oy * oy) + oz * oz) - r2_);
        a_Cc[k].y = cc;
        // far root iff the origin is inside — and for the hit's OWN sphere seen from outside: the near root of a ray that
        // re-enters it is the crossing at the shadow origin itself (t' = 1e-6/cos, which the reference may or may not
        // see); the far root decides, as in the reference, unless the light is inside the sphere (LO path)
        a_sgn[k] = (cc < 0.0f || own == k) ? 1.0f : -1.0f;
        near_scene &= fabsf(cc) < 1000.0f * (float)r2_;
        oc2max = fmaxf(oc2max, cc + (float)r2_);
        if (own == k) { own_in_bit = cc < 0.0f ? 1u << k : 0u; own_out_bit = cc < 0.0f ? 0u : 1u << k; }
        // (on the surface: within 5e-5 r + 5e-6 of it — the absolute part covers the reference's 2e-6 bound for small spheres)
        bad_cc |= ((cc < 0.0f) | ((own != k) & (cc < fmaf(1e-4f, (float)r2_, 1e-5f * (float)s_.a[3])))) ? 1u << k : 0u;
        a_tauS[k] = 0.0f; if (!OPAQUE) a_tauS[k] = (float)s_matd[s_.mat][10];
      }
    }
    static_assert(MAXS % 2 == 0, "root signs are stored as pairs");
#pragma unroll
    for (int k = 0; k < MAXS / 2; ++k) a_Sg[k] = mk2(a_sgn[2 * k], a_sgn[2 * k + 1]);
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
      a_D2n[k] = mk2(0.0f, 0.0f); a_D01[k] = mk2(0.0f, 0.0f);
      if (k < nP) {
        const Prim& s_ = P[nS + k];
        a_D2n[k].y = (float)(s_.a[3] - ((p.x * s_.a[0] + p.y * s_.a[1]) + p.z * s_.a[2]));
        a_tauP[k] = 0.0f; if (!OPAQUE) a_tauP[k] = (float)s_matd[s_.mat][10];
        // A FOREIGN plane within 6e-6 of the shadow origin (a floor hit on or next to the line where the floor meets a wall;
        // two nearly parallel planes; the hit's own plane, 1e-6 away, has the slow path LO): its crossing can lie within the
        // reference's 2e-6 bound along the ray (t' = |num| / |N.d| >= |num|) and is then not seen, while
        // ls_light_sum_simple's test 0 < num / den has no lower bound.  Such lanes take ls_light_sum, which leaves roots
        // within ~4e-6 of the origin to fp64.  (Scaled scene 1300544: two planes 2e-6 apart at the hit, 43 levels off.)
        bad_cc |= ((own != nS + k) & (fabsf(a_D2n[k].y) < 6e-6f)) ? 0x10000u << k : 0u;
      }
    }
#pragma unroll
    for (int k = 0; k < MAXB; ++k) {
      if (k < nB) {
        const Prim& s_ = P[nS + nP + k];
        a_blo[k][0] = (float)(s_.a[0] - p.x); a_blo[k][1] = (float)(s_.a[1] - p.y); a_blo[k][2] = (float)(s_.a[2] - p.z);
        a_bhi[k][0] = (float)(s_.a[3] - p.x); a_bhi[k][1] = (float)(s_.a[4] - p.y); a_bhi[k][2] = (float)(s_.a[5] - p.z);
        a_binside[k] = (a_blo[k][0] < 0.0f) & (a_blo[k][1] < 0.0f) & (a_blo[k][2] < 0.0f) &
                        (a_bhi[k][0] > 0.0f) & (a_bhi[k][1] > 0.0f) & (a_bhi[k][2] > 0.0f);
        a_tauB[k] = 0.0f; if (!OPAQUE) a_tauB[k] = (float)s_matd[s_.mat][10];
      }
    }
    const double* mat = s_matd[mat_idx];
    // material terms of this hit in registers (fp32) for the light loop
    const float kd0 = (float)mat[0], kd1 = (float)mat[1], kd2 = (float)mat[2];
    const float ks0 = (float)mat[3], ks1 = (float)mat[4], ks2 = (float)mat[5];
    // (the sequential renderer adds the specular term only for r.v > 0, ray_tracer.py:356-359: with shininess 0 the term
    // is 1 for r.v > 0 and absent otherwise, which is what exp2(1e-30 * log2(r.v)) gives: 1, and 0 for log2(0) = -inf)
    const float shin = ((float)mat[9] == 0.0f && fp.sequential) ? 1e-30f : (float)mat[9];
    const float one_m = 1.0f - (float)mat[10];
    // magnitude for the error bounds (LsTol): largest |oc|
    const float ocm = raw_sqrt(fmaxf(oc2max, 0.0f)) * 1.0001f;

    // ----------------------------------- wave-level pre-cull of (light, sphere) pairs ----------
    // A wave's hits are spatially compact (8x8 pixel tile, or neighbours in the queue).  Bound them by a
    // ball (c, rho); every shadow segment of the wave to light L lies in the hull from that ball to the
    // ball (Lp, RL).  Lane j tests pair j = (light, sphere) against that hull ONCE per wave; for the pairs
    // that fail, the per-lane coefficient / culling arithmetic below (~45 instructions per sphere) is skipped
    // by wave-uniform branches.  Conservative: a pair is only dropped when no lane's fan can touch it.
    unsigned long long pair_can = ~0ull;
    if (nS * sc.nL <= 64 && !brute) {
      float cnt = active ? 1.0f : 0.0f, sx = active ? pf.x : 0.0f, sy = active ? pf.y : 0.0f, sz = active ? pf.z : 0.0f;
#pragma unroll
      for (int o_ = 32; o_ > 0; o_ >>= 1) {
        cnt += __shfl_xor(cnt, o_); sx += __shfl_xor(sx, o_); sy += __shfl_xor(sy, o_); sz += __shfl_xor(sz, o_);
      }
      const float ic = 1.0f / fmaxf(cnt, 1.0f);
      const f3 cw = mkf(sx * ic, sy * ic, sz * ic);
      const f3 dw = mkf(pf.x - cw.x, pf.y - cw.y, pf.z - cw.z);
      float r2 = active ? dotf(dw, dw) : 0.0f;
#pragma unroll
      for (int o_ = 32; o_ > 0; o_ >>= 1) r2 = fmaxf(r2, __shfl_xor(r2, o_));
      const float rho = raw_sqrt(r2) * 1.0001f + 1e-5f * (1.0f + fabsf(cw.x) + fabsf(cw.y) + fabsf(cw.z));
      bool may = false;
      int j = (int)lane;
      asm volatile("" : "+v"(j));           // (likewise: lane j's (light, sphere) and their LDS addresses are formed per chunk, not kept)
      if (j < nS * sc.nL) {
        const int li = j / nS, k = j - li * nS;
        const float* lg = s_lightf[li];
        const float4 sf = s_sphf[k];
        const float RL = (n_sh > 1 ? 0.7072f * lg[8] : 0.0f) * 1.001f;
        const f3 tw = mkf(lg[0] - cw.x, lg[1] - cw.y, lg[2] - cw.z), ow = mkf(cw.x - sf.x, cw.y - sf.y, cw.z - sf.z);
        const float T0w = dotf(tw, tw), Aw = dotf(ow, tw), o2 = dotf(ow, ow);
        const float w0 = sf.w + rho, dl = RL - rho;                          // hull radius w0 + tau*dl around c + tau*tw
        const float qa = T0w - dl * dl, qb = Aw - w0 * dl, qc = o2 - w0 * w0;
        const float tau = fminf(fmaxf(-qb * raw_rcp(qa), 0.0f), 1.0f);
        const float g = fmaf(tau, fmaf(tau, qa, 2.0f * qb), qc);
        may = (qa <= 0.0f) | (g <= 1e-4f * (T0w + o2 + w0 * w0 + 1.0f));
      }
      pair_can = __ballot(may);
    }

    if (prof) { const long long t = clock64(); ck_hit += t - ck_t; ck_t = t; }
    // ------------------------------------------------------------- phase 2: lights & shadows ---
    float dif0 = 0, dif1 = 0, dif2 = 0, spc0 = 0, spc1 = 0, spc2 = 0;
    // (the next light's position is read from LDS while this light's Phong terms are formed, see the end of the loop body)
    float4 nL0 = s_lt[0][0], nL1 = s_lt[0][1], nL2 = s_lt[0][2];
    for (int li = 0; li < sc.nL; ++li) {
      // (compiler barrier: without it the block-invariant LDS reads of the light loop — sphere centres, plane
      // normals — are hoisted out of the chunk loop into ~40 long-lived VGPRs, which costs the second wave per SIMD)
      asm volatile("" ::: "memory");
      const float4 L0 = nL0, L1 = nL1, L2 = nL2;
      const uint32_t l_in = __float_as_uint(L2.x), l_out = __float_as_uint(L2.y), l_str = __float_as_uint(L2.z);
      const float l_shadow = L2.w, l_abs = L1.w;
      const f3 tl = mkf((L0.x - pf.x) + (L1.x - pl.x), (L0.y - pf.y) + (L1.y - pl.y), (L0.z - pf.z) + (L1.z - pl.z));
      const float T0 = dotf(tl, tl);
      const float rl = raw_rsq(T0);
      const f3 ld = scalef(tl, rl);                                         // ray_tracer.py:198-213
      const float radius = L0.w;
      const float nt = dotf(nf, tl);
      const uint32_t wcan = (nS * sc.nL <= 64 && !brute) ? (uint32_t)(pair_can >> (li * nS)) : 0xFFFFFFFFu;   // spheres this wave may touch
      float total = (float)n_cells;
      bool undecided = false;      // this (hit, light) goes to the exact re-test (k_fix_exact) with T = 0 here
      uint32_t dbg_cause = 0u, str_act = 0u;
      bool lane_can = false;     // some primitive may cross some shadow segment of this lane
      bool lane_full = false;    // opaque scenes: one sphere certainly blocks EVERY sample of this lane
      // ---- phase A: conservative per-(wave, light) culling, WITHOUT the light's basis.  A primitive is tested only if
      // it can cross the shadow segment of SOME sample of this light for SOME lane of the wave (results unchanged).
      // Every sample lies within RL of the light centre, so the segment p -> sample stays within RL of the segment
      // p -> Lp.  Only oc.tl, N.tl and per-hit terms enter; where the basis-dependent coefficients B = oc.right,
      // C = oc.up (D1, D2 for planes) used to, their bound |B| + |C| <= sqrt2 |oc - (oc.ld) ld| does.  60 % of the
      // (wave, light) pairs of the pool frame end here (every lane fully lit or in an umbra) and never need the
      // basis, the jitter key, B, C, D1, D2 or the error bounds.
      // Slow path (exact lower bound 2e-6 on every crossing) only if some lane needs it: the light is
      // (partly) behind the hit's own surface AND the outcome can depend on the own-surface crossing the
      // reference does not see — always for an own plane and in scenes with transparency; in all-opaque
      // scenes an own sphere / box blocks through its far side anyway unless the light is on its other side.
      const bool behind = nt < 0.71f * radius;
      // the hit point and (some sample of) the light possibly on different sides of the hit's own sphere /
      // box: then the only crossing between them is the one at the hit point itself
      bool own_light_inside = ((own_out_bit & l_in) | (own_in_bit & l_out)) != 0u;
      uint32_t act = 0;
      const float RL = 0.7072f * radius * 1.001f;
      {
        // Sphere k can cross the segment p -> S of some sample S only if, for some tau in [0,1],
        // |oc + tau*tl| <= r + tau*RL (the segments to all samples fan out inside a cone around
        // p -> Lp):  g(tau) = tau^2 (T0 - RL^2) + 2 tau (A - r RL) + cc <= 0, minimised on [0,1].
        const float qa = T0 - RL * RL, rqa = raw_rcp(qa);
        const float rT0 = rl * rl;                                             // 1 / |tl|^2
        const float hw2 = 0.5005f * radius * 1.4143f;                         // |oi|, |oj| <= radius / 2; |B| + |C| <= sqrt2 sqrt(B^2 + C^2)
        // (the next slot's centre is read from LDS while this one is worked on: a read + wait per slot otherwise)
        float4 sf_n = s_sphf[0], sl_n = s_sphl[0];
#pragma unroll
        for (int k = 0; k < MAXS; ++k) {
          a_AB[k] = mk2(0.0f, 0.0f); a_Cc[k].x = 0.0f;
          const float4 sf = sf_n, sl = sl_n;
          if (k + 1 < MAXS) { sf_n = s_sphf[k + 1]; sl_n = s_sphl[k + 1]; }
          if (k < nS && ((wcan >> k) & 1u)) {
            const f3 oc = mkf((pf.x - sf.x) + (pl.x - sl.x), (pf.y - sf.y) + (pl.y - sl.y), (pf.z - sf.z) + (pl.z - sl.z));
            const float A_ = dotf(oc, tl), cc_ = a_Cc[k].y, r = sf.w;
            a_AB[k].x = A_;
            if (brute) continue;                 // RT_FLAG_BRUTE_FORCE: every primitive is tested for every sample, nothing to decide here
            const float qb = A_ - r * RL;
            const float tau = fminf(fmaxf(-qb * rqa, 0.0f), 1.0f);
            const float g = fmaf(tau, fmaf(tau, qa, 2.0f * qb), cc_);
            // origin outside the sphere and every sample direction leaves it (oc.ts > 0 for all samples):
            // both roots are <= 0 — the lit side of the hit's own sphere, spheres behind the hit.
            // B^2 + C^2 = |oc|^2 - (oc.ld)^2 = cc + r^2 - A^2 / |tl|^2  (+ 0.1 % and a few ulp of |oc|^2 for its rounding)
            const float oc2 = fmaf(r, r, cc_);
            const float perp2 = fmaxf(fmaf(-A_ * rT0, A_, oc2) * 1.001f + 1e-6f * oc2, 0.0f);
            const float bt_lo = fmaf(-hw2, raw_sqrt(perp2), A_);
            const bool away = (cc_ >= 0.0f) & (bt_lo > 1e-5f * (fabsf(A_) + 1e-3f));
            const bool can = ((qa <= 0.0f) | (g <= 1e-5f * (T0 + fabsf(cc_) + r * r))) & !away;
            lane_can |= can;
            if (__any(active & can)) act |= 1u << k;
            if (OPAQUE) {
              // Umbra: at the parameter ts of closest approach of the central segment, the whole fan
              // cross-section (radius ts*RL around p + ts*tl) lies inside the sphere while p is outside
              // it -> every segment p -> S enters the sphere at a parameter in (0, ts), ts < 1.
              const float ts = fminf(fmaxf(-A_ * rT0, 0.0f), 1.0f);
              const float dist2 = fmaf(ts, fmaf(ts, T0, 2.0f * A_), cc_ + r * r);     // |oc + ts*tl|^2
              // fp32 cancellation bound of dist2 (hits far from the scene: |oc|^2 ~ T0 >> r^2 -> no claim)
              const float room = r * 0.999f - ts * RL;
              lane_full |= (cc_ > 0.0f) & (room > 0.0f) & (dist2 < room * room);
            }
          }
        }
        // dist2 above is a difference of terms as large as T0 and |oc|^2: its fp32 cancellation error
        // ~4e-7 (|cc| + r^2 + T0) stays below the 0.2 % margin of `room` only for hits near the scene
        lane_full &= near_scene & (T0 < umbra_t0_max);
        if (OPAQUE && MAXB > 0) {
#pragma unroll
          for (int k = 0; k < MAXB; ++k)      // same for the hit's own box (light vs the box grown / shrunk by RL)
            if (k < nB) {
              const float g_ = a_binside[k] ? -RL : RL;
              const bool in_ = (a_blo[k][0] - g_ < tl.x) & (tl.x < a_bhi[k][0] + g_) & (a_blo[k][1] - g_ < tl.y) &
                               (tl.y < a_bhi[k][1] + g_) & (a_blo[k][2] - g_ < tl.z) & (tl.z < a_bhi[k][2] + g_);
              own_light_inside |= (own == nS + nP + k) & (a_binside[k] ? !in_ : in_);
            }
        }
        lane_full &= !own_light_inside;
        // No umbra claim either when the shadow origin lies on a FOREIGN sphere's surface (bad_cc) while some sample of the
        // light may be inside that sphere: the entry crossing is then closer than the reference's 2e-6 bound and not seen,
        // and the exit lies beyond the sample — the segment is NOT blocked although the whole fan enters the sphere.  (A plane
        // hit next to the circle where the plane leaves a sphere with the light partly inside: scaled scene 900389, one
        // pixel 3 levels off.  The lane then reaches the sample loop, which leaves roots at the origin to fp64.)
        lane_full &= (bad_cc & l_in & 0xFFFFu) == 0u;
        float4 pn4[MAXP > 0 ? MAXP : 1];
#pragma unroll
        for (int k = 0; k < MAXP; ++k) pn4[k] = s_plnf[k];
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
          a_D01[k] = mk2(0.0f, 0.0f); a_D2n[k].x = 0.0f;
          if (k < nP) {
            const float4 nf4 = pn4[k];
            const f3 N = mkf(nf4.x, nf4.y, nf4.z);
            const float D0_ = dotf(N, tl), num_ = a_D2n[k].y;
            a_D01[k].x = D0_;
            if (brute) continue;
            // |N.right| + |N.up| <= sqrt2 |N - (N.ld) ld|, N unit (to 0.1 %)
            const float nperp2 = fmaxf(fmaf(-D0_ * rT0, D0_, 1.002f), 0.0f);
            const float spread = hw2 * raw_sqrt(nperp2) + 1e-6f * fabsf(D0_);
            // tau = num / den in (0, 1) needs num and den of the same sign
            const bool can = num_ < 0.0f ? (D0_ - spread <= 0.0f) : (num_ > 0.0f ? (D0_ + spread >= 0.0f) : true);
            lane_can |= can;
            if (__any(active & can)) act |= 0x10000u << k;
          }
        }
        if (MAXB > 0) lane_can |= nB > 0;
      }
      if (brute) { act = all_prims; lane_can = true; lane_full = false; }
      if (RT_DBG(fp) & 8u) act = 0xFFFFFFFFu;
      if (!brute && !(RT_DBG(fp) & 2u) && !__any(active & lane_can & !lane_full)) {
        // No lane of the wave has a penumbra candidate for this light: every lane is either fully lit
        // (all n^2 terms are exactly 1) or in the umbra of one opaque sphere (all terms exactly 0).
        total = lane_full ? 0.0f : (float)n_cells;
        if ((RT_DBG(fp) & 1u) && lane == 0) atomicAdd(&C->queries, 1ull);
        if (prof) { const long long t = clock64(); ck_setup += t - ck_t; ck_t = t; }
      } else {
      // ---- phase B: the light's basis and what depends on it, for the primitives phase A left
      bool sw = fabsf(ld.x) >= 0.9f;
      if (__any(active & (fabsf(fabsf(ld.x) - 0.9f) < 1e-5f))) {             // within rounding of the switch: decide it in fp64
        const uint32_t t_ = threadIdx.x;
        if (active & (fabsf(fabsf(ld.x) - 0.9f) < 1e-5f))
          sw = basis_switch_exact(sc, mk3(s_geo[0][t_] + s_geo[3][t_] * EPS, s_geo[1][t_] + s_geo[4][t_] * EPS, s_geo[2][t_] + s_geo[5][t_] * EPS), li);
      }
      const f3 ref = sw ? mkf(0, 1, 0) : mkf(1, 0, 0);
      const f3 right = normalize_clamped(crossf(ld, ref));
      const f3 up = normalize_clamped(crossf(right, ld));
      const float T1 = 2.0f * dotf(tl, right), T2 = 2.0f * dotf(tl, up);
      const float cell32 = n_sh > 1 ? (radius / (float)n_side) * JIT_SCALE : 0.0f;
      const JitKey key = jit_light_key(fp.seed, gpix, (uint32_t)level, (uint32_t)li);
      const bool lo_path = (RT_DBG(fp) & 4u) || __any(active & behind & (OPAQUE ? (own_is_plane | own_light_inside) : true));
      // (the slow path reads the coefficients of each lane's OWN sphere, whether or not the wave tests that sphere)
      const uint32_t bc_mask = lo_path ? (wcan | act) : act;
      float4 sfb_n = s_sphf[0], slb_n = s_sphl[0];
#pragma unroll
      for (int k = 0; k < MAXS; ++k) {
        const float4 sf = sfb_n, sl = slb_n;
        if (k + 1 < MAXS) { sfb_n = s_sphf[k + 1]; slb_n = s_sphl[k + 1]; }
        if (k < nS && ((bc_mask >> k) & 1u)) {
          const f3 oc = mkf((pf.x - sf.x) + (pl.x - sl.x), (pf.y - sf.y) + (pl.y - sl.y), (pf.z - sf.z) + (pl.z - sl.z));
          a_AB[k].y = dotf(oc, right); a_Cc[k].x = dotf(oc, up);
        }
      }
#pragma unroll
      for (int k = 0; k < MAXP; ++k) {
        if (k < nP && ((act >> (16 + k)) & 1u)) {
          const float4 nf4 = s_plnf[k];
          const f3 N = mkf(nf4.x, nf4.y, nf4.z);
          a_D01[k].y = dotf(N, right); a_D2n[k].x = dotf(N, up);
        }
      }
      if (prof) { const long long t = clock64(); ck_setup += t - ck_t; ck_t = t; }
      const float4* cells = s_cells + li * ((n_cells + 1) >> 1);
      str_act = act & l_str;
      if (MAXB > 0) str_act |= l_str & 0xFF000000u;
      // scenes with transparency: the reference's walk multiplies at most 20 crossings (ray_tracer.py:257), the sample loop
      // all of them — a (wave, light) whose candidate set could give a segment more than that goes to the exact walk
      // (only the 12- / 16-sphere instantiations can get there; the sequential renderer's walk has no cap)
      const bool over_cap = !OPAQUE && (2 * MAXS + MAXP + 2 * MAXB > WALK_CAP) && !fp.sequential &&
                            2 * __popc(act & 0xFFFFu) + __popc((act >> 16) & 0xFFu) + 2 * nB > WALK_CAP;
      if (str_act != 0u || over_cap) {
        // the light's sample square touches a candidate primitive: crossings at the sample end of a segment
        total = 0.0f; undecided = true;
      } else {
        // fp32 error bounds of this (hit, light): see LsTol.  |ts| <= tsm, |oc| <= ocm.  oc and tl carry <= 2 ulp of
        // their size (two-part subtraction above), the basis vectors <= 4 ulp per component, the sample offsets
        // <= 3 ulp of the light's size; the dots accumulate 3 more.  With |oi|, |oj| <= radius/2 (18 radius = 25 RL):
        //   error of oc.ts          e_b <= u |oc| (9 |ts| + 25 RL)
        //   relative error of |ts|^2 cc    <= 34 u   (light farther than 3 RL: |ts| within a factor 2 of tsm)
        //   => |disc error| <= |oc.ts| (2 e_b + 34 u |oc| |ts|) + e_b^2 when |disc| is small; all with a factor 1.5
        LsTol tol;
        const float tlen = T0 * rl;                                            // |tl|
        const float tsm = tlen + RL;
        const float e_b = 1.5f * LS_U * ocm * (9.0f * tsm + 25.0f * RL);
        tol.kap = 2.0f * e_b + 54.0f * LS_U * ocm * tsm;      // (51 u + 3 u for |ts|^2 without its tl.right / tl.up terms, ls_light_sum_simple)
        {
          // the same bound with each sphere's own |oc| = sqrt(cc + r^2) instead of the largest one (a grazed sphere is
          // usually a near one: 3-5x fewer undecided pairs on the pool scene), never below 1e-3 of it
          const float kap1 = tol.kap * raw_rcp(fmaxf(ocm, 1e-30f));
#pragma unroll
          for (int k = 0; k < MAXS; ++k) {
            a_kapS[k] = tol.kap;
            if (act & (1u << k)) {
              const float r_ = s_sphf[k].w;
              const float ock = raw_sqrt(fmaxf(fmaf(r_, r_, a_Cc[k].y), 0.0f)) * 1.0001f + 1e-3f * ocm;
              a_kapS[k] = fminf(kap1 * ock, tol.kap);
            }
          }
        }
        // (+ (2.2e-6 |ts|)^2: a chord shorter than ~4.4e-6 along the ray — two crossings the reference's walk sees as one —
        // is a discriminant below that, whatever the scene's size; it only exceeds the rounding part for scenes < 1e-3)
        tol.kap0 = fmaf(4.84e-12f * tsm, tsm, 2.0f * e_b * e_b);
        tol.elo = 2.0f * T_LO * tsm + 2.0f * (e_b + tol.kap);
        tol.eden = 12.0f * LS_U * tsm;
        LsFlags fl = ls_flags_init();
        // all-opaque scenes without boxes: the square-root-free loop when every candidate sphere has the wave's shadow origins
        // outside it (or is the hit's own) and the light's samples outside it (ls_light_sum_simple); wave-uniform
        const bool simple = OPAQUE && MAXB == 0 && !lo_path && !(RT_DBG(fp) & 64u) && ((act & 0xFFFFu & l_in) == 0u) &&
                            !__any(active & ((bad_cc & act) != 0u));
        // (sample-end decisions of far hits, see s_lend: only the loops that compare roots with |ts|^2)
        const bool end_bad = simple ? false : (ocm >= s_lend[li]);
        if (lo_path)
          total = ls_light_sum<MAXS, MAXS, MAXP, MAXB, OPAQUE, true>(pr, act, nB, cells, n_cells, key, cell32, tl, right, up,
                                                                     T0, T1, T2, tol.kap, fl, own);
        else if (MAXS > 2 && __popc(act & ((1u << MAXS) - 1u)) <= 2) {
          // At most two spheres left for this (wave, light) — the usual case: their coefficients are moved to dense
          // slots (wave-uniform moves) and the sample loop runs in an instantiation for exactly that many spheres,
          // with or without the plane block: nothing about `act` is tested per sample pair any more.
          v2f gAB[2], gCc[2], gSg[1];
          float gtau[2], gkap[2];
          gAB[0] = gAB[1] = mk2(0.0f, 0.0f); gCc[0] = gCc[1] = mk2(0.0f, 1.0f); gSg[0] = mk2(-1.0f, -1.0f);
          gtau[0] = gtau[1] = 0.0f; gkap[0] = gkap[1] = 0.0f;
          int j = 0;
#pragma unroll
          for (int k = 0; k < MAXS; ++k) {
            if ((act >> k) & 1u) {
              const float sg_ = (k & 1) ? a_Sg[k >> 1].y : a_Sg[k >> 1].x;
              if (j == 0) { gAB[0] = a_AB[k]; gCc[0] = a_Cc[k]; gSg[0].x = sg_; gtau[0] = a_tauS[k]; gkap[0] = a_kapS[k]; }
              else        { gAB[1] = a_AB[k]; gCc[1] = a_Cc[k]; gSg[0].y = sg_; gtau[1] = a_tauS[k]; gkap[1] = a_kapS[k]; }
              ++j;
            }
          }
          const uint32_t act2 = (act & ~((1u << MAXS) - 1u)) | ((1u << j) - 1u);
          const bool nopl = ((act >> 16) & 0xFFu) == 0u;
          if (simple) {
            if (j == 2)
              total = nopl ? ls_light_sum_simple<2, MAXP, true, true>(gAB, gCc, gkap, a_D01, a_D2n, act2, cells, n_cells, key, cell32, T0, fl)
                           : ls_light_sum_simple<2, MAXP, true, false>(gAB, gCc, gkap, a_D01, a_D2n, act2, cells, n_cells, key, cell32, T0, fl);
            else if (j == 1)
              total = nopl ? ls_light_sum_simple<1, MAXP, true, true>(reinterpret_cast<v2f(&)[1]>(gAB), reinterpret_cast<v2f(&)[1]>(gCc), reinterpret_cast<float(&)[1]>(gkap), a_D01, a_D2n, act2, cells, n_cells, key, cell32, T0, fl)
                           : ls_light_sum_simple<1, MAXP, true, false>(reinterpret_cast<v2f(&)[1]>(gAB), reinterpret_cast<v2f(&)[1]>(gCc), reinterpret_cast<float(&)[1]>(gkap), a_D01, a_D2n, act2, cells, n_cells, key, cell32, T0, fl);
            else      // planes only
              total = ls_light_sum_simple<1, MAXP, false, false>(reinterpret_cast<v2f(&)[1]>(gAB), reinterpret_cast<v2f(&)[1]>(gCc), reinterpret_cast<float(&)[1]>(gkap), a_D01, a_D2n, act2, cells, n_cells, key, cell32, T0, fl);
          } else if (j == 2) {
            LsPrims<2, MAXP, MAXB, OPAQUE> pr2{gAB, gCc, gSg, gtau, gkap, a_D01, a_D2n, a_tauP, a_blo, a_bhi, a_tauB, a_binside};
            total = nopl ? ls_light_sum<2, 2, MAXP, MAXB, OPAQUE, false, true, true>(pr2, act2, nB, cells, n_cells, key, cell32, tl, right, up, T0, T1, T2, tol.kap, fl)
                         : ls_light_sum<2, 2, MAXP, MAXB, OPAQUE, false, true, false>(pr2, act2, nB, cells, n_cells, key, cell32, tl, right, up, T0, T1, T2, tol.kap, fl);
          } else {
            LsPrims<1, MAXP, MAXB, OPAQUE> pr1{reinterpret_cast<v2f(&)[1]>(gAB), reinterpret_cast<v2f(&)[1]>(gCc), gSg, reinterpret_cast<float(&)[1]>(gtau),
                                               reinterpret_cast<float(&)[1]>(gkap), a_D01, a_D2n, a_tauP, a_blo, a_bhi, a_tauB, a_binside};
            if (j == 1)
              total = nopl ? ls_light_sum<1, 1, MAXP, MAXB, OPAQUE, false, true, true>(pr1, act2, nB, cells, n_cells, key, cell32, tl, right, up, T0, T1, T2, tol.kap, fl)
                           : ls_light_sum<1, 1, MAXP, MAXB, OPAQUE, false, true, false>(pr1, act2, nB, cells, n_cells, key, cell32, tl, right, up, T0, T1, T2, tol.kap, fl);
            else      // planes / boxes only
              total = ls_light_sum<1, 1, MAXP, MAXB, OPAQUE, false>(pr1, act2, nB, cells, n_cells, key, cell32, tl, right, up, T0, T1, T2, tol.kap, fl);
          }
        } else if (simple) {
          // (every sphere slot active — the brute-force frames, scenes of exactly MAXS spheres in a row —: no per-slot test)
          const bool dense = (act & ((1u << MAXS) - 1u)) == ((1u << MAXS) - 1u), nopl = ((act >> 16) & 0xFFu) == 0u;
          total = dense ? (nopl ? ls_light_sum_simple<MAXS, MAXP, true, true>(a_AB, a_Cc, a_kapS, a_D01, a_D2n, act, cells, n_cells, key, cell32, T0, fl)
                                : ls_light_sum_simple<MAXS, MAXP, true, false>(a_AB, a_Cc, a_kapS, a_D01, a_D2n, act, cells, n_cells, key, cell32, T0, fl))
                        : ls_light_sum_simple<MAXS, MAXP, false, false>(a_AB, a_Cc, a_kapS, a_D01, a_D2n, act, cells, n_cells, key, cell32, T0, fl);
        } else
          total = ls_light_sum<MAXS, MAXS, MAXP, MAXB, OPAQUE, false>(pr, act, nB, cells, n_cells, key, cell32, tl, right, up,
                                                                      T0, T1, T2, tol.kap, fl);
        // also undecided in fp32: the light within 3 sample-square radii of the hit (|ts| is then not comparable
        // to |tl|, which the bounds assume)
        // far hits: the fp32 position of a segment's far end is off by more than the slack s_lstr was built with
        const bool far = 16.0f * LS_U * tsm > 1e-4f * (1.0f + l_abs);
        // (ls_light_sum_simple counts blocked samples in floats: a sum that is not a whole number had a denormal decision value)
        undecided = !ls_decided(fl, tol) | (T0 < 9.0f * RL * RL) | !(T0 > 1e-12f) | far | end_bad | (simple & (total != floorf(total)));
        if (RT_DBG(fp) & (1u << 23))    // cause bits: 1 grazing discriminant, 2 root at the origin, 4 parallel plane, 8 light close, 16 far
          dbg_cause = ((!(fl.unc < -tol.kap0) && fl.unc < 3.0e38f) ? 1u : 0u) | (fl.unc >= 3.005e38f ? (32u << (int)((fl.unc - 3.005e38f) * 1e-36f)) : 0u) | (!(fl.mlow > __float_as_uint(tol.elo)) ? 2u : 0u) | (!(fl.dmin > tol.eden) ? 4u : 0u) |
                      (((T0 < 9.0f * RL * RL) | !(T0 > 1e-12f)) ? 8u : 0u) | (far ? 16u : 0u);
      }
      }   // phase B
      if (prof) { const long long t = clock64(); ck_loop += t - ck_t; ck_t = t; }
      if ((RT_DBG(fp) & 1u) && lane == 0) atomicAdd(&C->queries, 1ull << 32);
      if ((RT_DBG(fp) & 2u) && active) {      // shortcut disabled: audit what it would have claimed
        if (!lane_can && total != (float)n_cells) atomicAdd(&C->dbg[0], 1ull);
        if (lane_full && total != 0.0f) atomicAdd(&C->dbg[1], 1ull);
        if (lane_can && !lane_full) atomicAdd(&C->dbg[2], 1ull);
        atomicAdd(&C->dbg[3], 1ull);
      }
      // the conservative claims are exact whatever the loop computed for the lane: nothing can cross any of its
      // segments (all terms 1) / one opaque sphere swallows its whole fan (all terms 0)
      if (!lane_can) { total = (float)n_cells; undecided = false; }
      if (OPAQUE && lane_full) { total = 0.0f; undecided = false; }
      undecided &= active;
#ifdef RT_DEBUG_KERNEL
      if ((RT_DBG(fp) & 33554432u) && active && !undecided) {
        // audit build, bit 25: every pair this kernel decides itself (sample loop or lit / umbra claim) against the
        // literal fp64 walk of every cell (tests/test_gpu_audit.py)
        const uint32_t t_ = threadIdx.x;
        const d3 pe = mk3(s_geo[0][t_] + s_geo[3][t_] * EPS, s_geo[1][t_] + s_geo[4][t_] * EPS, s_geo[2][t_] + s_geo[5][t_] * EPS);
        double tsum = 0.0; int qd = 0;
        for (int cell_ = 0; cell_ < n_cells; ++cell_) {
          const ExactRay er = exact_shadow_ray(sc, pe, li, fp.n_shadow, fp.seed, gpix, (uint32_t)level, cell_);
          tsum += shadow_walk(sc.prims, sc.nS, sc.nP, sc.nB, sc.mats, er.so, er.dir, er.maxdist, &qd, false, fp.plane_thr);
        }
        atomicAdd(&g_walk_audit[0], 1ull);
        if (fabs(tsum - (double)total) > 1e-3) {
          if (atomicAdd(&g_walk_audit[1], 1ull) < 12ull)
            printf("[walk audit] level %d light %d (k_shade_ls): fp32 sum %.6f exact %.6f  p (%.17g %.17g %.17g) gpix %u lane_can %d lane_full %d\n",
                   level, li, total, tsum, pe.x, pe.y, pe.z, gpix, (int)lane_can, (int)lane_full);
        }
      }
#endif
      if ((RT_DBG(fp) & (1u << 23)) && undecided) {      // audit build: why pairs go to the re-test queue
#ifdef RT_DEBUG_KERNEL
        atomicAdd(&g_fix_cause[0], 1ull);
        if (str_act != 0u) atomicAdd(&g_fix_cause[1], 1ull);
        for (int b_ = 0; b_ < 5; ++b_) if (dbg_cause & (1u << b_)) atomicAdd(&g_fix_cause[2 + b_], 1ull);
        for (int b_ = 0; b_ < 5; ++b_) if (dbg_cause & (32u << b_)) atomicAdd(&g_forced_cause[b_], 1ull);
#endif
      }
      if (undecided) total = 0.0f;
      const float4 L3 = s_lt[li][3];                         // colour, specular intensity: for the Phong terms below
      {
        const int ln = li + 1 < sc.nL ? li + 1 : li;        // the next light, read while this one's Phong terms are formed
        nL0 = s_lt[ln][0]; nL1 = s_lt[ln][1]; nL2 = s_lt[ln][2];
      }
      const float T = total * inv_cells;
      const float I = fmaf(l_shadow, T, 1.0f - l_shadow);                                 // ray_tracer.py:543
      // light direction from the offset point p instead of the un-offset hit point (1e-6 apart): a relative error of
      // 1e-6 / |tl| in the Phong terms — below 1/20 of an 8-bit level for lights farther than 0.01 from the surface.  (Forming
      // it from hp for closer lights behind a wave-uniform test was measured: +2 % on the pool 4K frame for scenes a
      // thousandth of the reference's size; not taken.  DESIGN.md 4.3, "tiny scenes".)
      const float ndl = fmaxf(nt * rl, 0.0f);
      const float s1 = I * ndl;
      dif0 = fmaf(kd0 * L3.x, s1, dif0);
      dif1 = fmaf(kd1 * L3.y, s1, dif1);
      dif2 = fmaf(kd2 * L3.z, s1, dif2);
      const float dn = -nt * rl;                                                          // (-l).n
      const f3 rf = mkf(fmaf(-2.0f * dn, nf.x, -ld.x), fmaf(-2.0f * dn, nf.y, -ld.y), fmaf(-2.0f * dn, nf.z, -ld.z));
      const float rdv = fmaxf(dotf(rf, v), 0.0f);
      // rdv ** shininess as exp2(shininess * log2(rdv)) on the raw hardware transcendentals
      // (v_log_f32 / v_exp_f32, 1 ulp each; log2(0) = -inf -> 0); np.power(0, 0) == 1
      const float pw = (shin == 0.0f) ? 1.0f : __builtin_amdgcn_exp2f(shin * __builtin_amdgcn_logf(rdv));
      const float s2 = I * pw * L3.w;
      spc0 = fmaf(ks0 * L3.x, s2, spc0);
      spc1 = fmaf(ks1 * L3.y, s2, spc1);
      spc2 = fmaf(ks2 * L3.z, s2, spc2);
      if (__any(undecided)) {
        // colour per unit of T of this (hit, light): d(colour)/dI * shadow_intensity, times weight and (1 - transparency)
        const float cm = l_shadow * one_m, sp = pw * L3.w;
        const uint32_t t_ = threadIdx.x;
        const d3 pe = mk3(s_geo[0][t_] + s_geo[3][t_] * EPS, s_geo[1][t_] + s_geo[4][t_] * EPS, s_geo[2][t_] + s_geo[5][t_] * EPS);
        queue_fix(undecided, fix, C, fp, pe, wgt.x * cm * L3.x * fmaf(kd0, ndl, ks0 * sp), wgt.y * cm * L3.y * fmaf(kd1, ndl, ks1 * sp),
                  wgt.z * cm * L3.z * fmaf(kd2, ndl, ks2 * sp), lp, gpix, level, li);
      }
    }

    // ------------------------------------------------- phase 3: accumulate, spawn children -----
    if (active) {
      float b0 = 0.0f, b1 = 0.0f, b2 = 0.0f;
      if (seq_background_term(fp, level, mat)) {                             // ray_tracer.py:370-377
        b0 = (float)(sc.bg[0] * mat[10]); b1 = (float)(sc.bg[1] * mat[10]); b2 = (float)(sc.bg[2] * mat[10]);
      }
      float c0 = wgt.x * ((dif0 + spc0) * one_m + b0), c1 = wgt.y * ((dif1 + spc1) * one_m + b1),
            c2 = wgt.z * ((dif2 + spc2) * one_m + b2);
      if (fall) {                              // children that left the scene: weight * background (ray_tracer.py:500-502)
        const uint32_t t_ = threadIdx.x;
        if (needT && !hitT) {
          c0 += (float)((s_geo[9][t_] * mat[10]) * sc.bg[0]); c1 += (float)((s_geo[10][t_] * mat[10]) * sc.bg[1]); c2 += (float)((s_geo[11][t_] * mat[10]) * sc.bg[2]);
        }
        if (needR && !hitR) {
          c0 += (float)((s_geo[9][t_] * mat[6]) * sc.bg[0]); c1 += (float)((s_geo[10][t_] * mat[7]) * sc.bg[1]); c2 += (float)((s_geo[11][t_] * mat[8]) * sc.bg[2]);
        }
      }
      float* dst = fb + 3 * (size_t)lp;
      if (level == 0) { dst[0] = c0; dst[1] = c1; dst[2] = c2; }
      else { atomicAdd(dst + 0, c0); atomicAdd(dst + 1, c1); atomicAdd(dst + 2, c2); }
    }
    const long long ck_s1 = prof ? clock64() : 0;
    if (fall) {
      if ((mTh | mRh) != 0ull) {
        const uint32_t base_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)spawn_base);
        const uint32_t nTh = (uint32_t)__popcll(mTh);
        const uint32_t oT = base_ + __builtin_amdgcn_mbcnt_hi((uint32_t)(mTh >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mTh, 0u));
        const uint32_t oR = base_ + nTh + __builtin_amdgcn_mbcnt_hi((uint32_t)(mRh >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mRh, 0u));
        const uint32_t t_ = threadIdx.x;
        auto put_child = [&](bool refl, uint32_t o_) __attribute__((always_inline)) {
          if (o_ >= seg_cap) { C->overflow = 1u; return; }
          const d3 hp_ = mk3(s_geo[0][t_], s_geo[1][t_], s_geo[2][t_]), n_ = mk3(s_geo[3][t_], s_geo[4][t_], s_geo[5][t_]);
          const d3 d_ = mk3(s_geo[6][t_], s_geo[7][t_], s_geo[8][t_]);
          d3 oc_, dc_;
          if (refl) {
            const double dn = dot3(d_, n_);
            dc_ = mk3(d_.x - (2 * dn) * n_.x, d_.y - (2 * dn) * n_.y, d_.z - (2 * dn) * n_.z);
            oc_ = mk3(hp_.x + n_.x * EPS, hp_.y + n_.y * EPS, hp_.z + n_.z * EPS);
          } else {
            dc_ = d_;
            oc_ = mk3(hp_.x - n_.x * EPS, hp_.y - n_.y * EPS, hp_.z - n_.z * EPS);
          }
          const double tc_ = s_ct[refl ? 1 : 0][t_];
          const uint32_t kf_ = s_ck[refl ? 1 : 0][t_];
          HitRay y;
          y.hp[0] = oc_.x + tc_ * dc_.x; y.hp[1] = oc_.y + tc_ * dc_.y; y.hp[2] = oc_.z + tc_ * dc_.z;
          y.d[0] = dc_.x; y.d[1] = dc_.y; y.d[2] = dc_.z;
          y.w[0] = (float)(s_geo[9][t_] * (refl ? mat[6] : mat[10])); y.w[1] = (float)(s_geo[10][t_] * (refl ? mat[7] : mat[10]));
          y.w[2] = (float)(s_geo[11][t_] * (refl ? mat[8] : mat[10]));
          y.pk = MAXB > 0 ? (lp | ((kf_ & 15u) << HITRAY_PIX_BITS) | ((kf_ >> 8) << 29)) : (lp | ((kf_ & 31u) << HITRAY_PIX_BITS));
          hout[(size_t)seg * seg_cap + o_] = y;
        };
        if (hitT) put_child(false, oT);
        if (hitR) put_child(true, oR);
      }
    } else
    if (nT + nR != 0u) {                                                   // ray_tracer.py:565-595
      const uint32_t base_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)spawn_base);
      const uint32_t seg_cap = fp.queue_cap / NSEG;
      const uint32_t oT = base_ + __builtin_amdgcn_mbcnt_hi((uint32_t)(mT >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mT, 0u));
      const uint32_t oR = base_ + nT + __builtin_amdgcn_mbcnt_hi((uint32_t)(mR >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mR, 0u));
      uint32_t sT = 0xFFFFFFFFu, sR = 0xFFFFFFFFu;
      if (needT) { if (oT < seg_cap) sT = seg * seg_cap + oT; else C->overflow = 1u; }
      if (needR) { if (oR < seg_cap) sR = seg * seg_cap + oR; else C->overflow = 1u; }
      if (needT || needR) {
        const uint32_t t_ = threadIdx.x;
        const d3 hp = mk3(s_geo[0][t_], s_geo[1][t_], s_geo[2][t_]);
        const d3 n = mk3(s_geo[3][t_], s_geo[4][t_], s_geo[5][t_]);
        RayIn r; r.d = mk3(s_geo[6][t_], s_geo[7][t_], s_geo[8][t_]); r.w = mk3(s_geo[9][t_], s_geo[10][t_], s_geo[11][t_]);
        r.lp = lp;
        if (sT != 0xFFFFFFFFu) {
          nq.ox[sT] = hp.x - n.x * EPS; nq.oy[sT] = hp.y - n.y * EPS; nq.oz[sT] = hp.z - n.z * EPS;
          nq.dx[sT] = r.d.x; nq.dy[sT] = r.d.y; nq.dz[sT] = r.d.z;
          nq.wx[sT] = r.w.x * mat[10]; nq.wy[sT] = r.w.y * mat[10]; nq.wz[sT] = r.w.z * mat[10];
          nq.pix[sT] = r.lp;
        }
        if (sR != 0xFFFFFFFFu) {
          const double dn = dot3(r.d, n);                                // reflect_batch, :34-36
          nq.dx[sR] = r.d.x - (2 * dn) * n.x; nq.dy[sR] = r.d.y - (2 * dn) * n.y; nq.dz[sR] = r.d.z - (2 * dn) * n.z;
          nq.ox[sR] = hp.x + n.x * EPS; nq.oy[sR] = hp.y + n.y * EPS; nq.oz[sR] = hp.z + n.z * EPS;
          nq.wx[sR] = r.w.x * mat[6]; nq.wy[sR] = r.w.y * mat[7]; nq.wz[sR] = r.w.z * mat[8];
          nq.pix[sR] = r.lp;
        }
      }
    }
    chunk = steal((uint32_t)__builtin_amdgcn_readfirstlane((int)next_cursor));
    if (prof) ck_sync += clock64() - ck_s1;
  }
  if (fused && lane == 0 && fused_hits != 0u) atomicAdd(&C->n_hits[0], fused_hits);
  if (fall && lane == 0 && my_children != 0u) atomicAdd(&C->n_rays[level + 1], my_children);
  if (prof && lane == 0) {
    atomicAdd(&C->queries, (unsigned long long)ck_sync);
    atomicAdd(&C->dbg[0], (unsigned long long)ck_hit); atomicAdd(&C->dbg[1], (unsigned long long)ck_setup);
    atomicAdd(&C->dbg[2], (unsigned long long)ck_loop); atomicAdd(&C->dbg[3], (unsigned long long)(clock64() - ck_begin));
  }
}

#ifndef RT_NO_LAUNCHERS   // (single-kernel compile harness: tools/one_kernel.sh)
// Scenes served: up to 6 spheres + 2 planes without boxes, or up to 4 spheres + 2 planes + 4 boxes,
// any shadow grid with lights * n^2 <= LS_MAX_CELLS.  Returns false -> caller uses another kernel.
// Does k_shade_ls serve this scene / frame?  (launch_frame asks before level 0: the kernel then traces the camera rays itself.)
static bool shade_ls_serves(const DevScene& ds, const FrameParams& fp) {
  const int side = fp.n_shadow > 1 ? fp.n_shadow : 1;
  if (ds.nL * side * side > LS_MAX_CELLS || ds.nP > 2 || ds.nL > LS_MAX_LIGHTS || ds.nM > LS_MAX_MATS) return false;
  if (ds.nB == 0) return ds.nS <= 16;
  return ds.nB <= 4 && ds.nS <= 4;
}

static bool launch_shade_ls(const DevScene& ds, bool opaque, const DevCamera& cam, const FrameParams& fp, int level,
                            RayQueue cur, RayQueue nxt, const HitRec* hits, Counters* C, float* fb, FixRec* fix, int grid,
                            int num_cu, hipStream_t st) {
  if (!shade_ls_serves(ds, fp)) return false;
#define RT_LS_K(S, Pn, B, O, F)                                                                                       \
  hipLaunchKernelGGL((k_shade_ls<S, Pn, B, O, F>), dim3(shade_grid<k_shade_ls<S, Pn, B, O, F>>(num_cu, grid)), dim3(BLOCK), 0, st, ds, cam, fp, level, cur, nxt, hits, C, fb, fix)
  // fused frames (fp.fuse0 == 2): one instantiation for level 0 (camera rays, no queue input) and one for the levels that
  // read the HitRay queue — neither carries the other's arguments and code
#define RT_LS(S, Pn, B)                                                                                             \
  do {                                                                                                              \
    if (fp.fuse0 == 2u) {                                                                                           \
      if (opaque) { if (level == 0) RT_LS_K(S, Pn, B, true, 2); else RT_LS_K(S, Pn, B, true, 3); }                  \
      else { if (level == 0) RT_LS_K(S, Pn, B, false, 2); else RT_LS_K(S, Pn, B, false, 3); }                       \
    } else if (opaque) RT_LS_K(S, Pn, B, true, 0);                                                                  \
    else RT_LS_K(S, Pn, B, false, 0);                                                                               \
    return true;                                                                                                    \
  } while (0)
  if (ds.nB == 0) {
    if (ds.nS <= 2) RT_LS(2, 2, 0);
    if (ds.nS <= 4) RT_LS(4, 2, 0);
    if (ds.nS <= 6) RT_LS(6, 2, 0);
    // 7-16 spheres: the per-hit coefficient arrays no longer fit 256 VGPRs (22 / 60 / 108 spilled at 8 / 12 / 16), but
    // the sample loop runs in its dense 1- / 2-sphere instantiations almost always, so the spills sit outside it:
    // pool + 2 spheres at 4K 2.5 ms against 9.7 ms through k_shade_fast
    if (ds.nS <= 8) RT_LS(8, 2, 0);
    if (ds.nS <= 12) RT_LS(12, 2, 0);
    if (ds.nS <= 16) RT_LS(16, 2, 0);
  } else if (ds.nB <= 4) {
    if (ds.nS <= 2) RT_LS(2, 2, 4);
    if (ds.nS <= 4) RT_LS(4, 2, 4);
    // (more spheres with boxes: the box path leaves too many (hit, light) pairs to the exact re-test — slab roots near an
    // edge or a direction nearly parallel to a slab are never decided in fp32 —, measured 16.5 ms against 6.6 ms through
    // k_shade_fast for pool + 2 boxes at 4K)
  }
#undef RT_LS
#undef RT_LS_K
  return false;
}

#endif  // RT_NO_LAUNCHERS

}  // namespace rt
