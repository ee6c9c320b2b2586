#!/usr/bin/env python3
"""CPU diagnosis of one pixel's level-0 shadow samples: exact (oracle) transmission per (light, cell) next to an fp32
emulation of k_shade_fast's per-primitive tests and of its 'undecided' flags.  Finds decisions the flags miss.
    python tools/diag_pixel.py synth:4096:512:3:3 3840x2160 7 1669"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from fp32_vs_fp64 import load
from oracle import oracle
from raytracer_amd.jitter import PixelKeyedJitter
from raytracer_amd.scene import Sphere, InfinitePlane, Cube

spec, size, x, y = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
W, H = (int(v) for v in size.split("x"))
cam, st, mats, surfs, lights = load(spec)
cam.setup(W / H)
n_sh = int(st.root_number_shadow_rays)
# camera ray (camera.py:46-73)
px = (0.5 - (x + 0.5) / W) * cam.screen_width; py = (0.5 - (y + 0.5) / H) * cam.screen_height
P = cam.position + cam.forward * cam.screen_distance + cam.right * px + cam.up * py
d = (P - cam.position); d = d / np.sqrt(d @ d)
o = cam.position.copy()
oracle.build()
t, idx, nrm = oracle.nearest_hit(surfs, o[None], d[None])
t, idx, nrm = t[0], idx[0], nrm[0]
print("hit surface", idx, type(surfs[idx]).__name__, "t", t, "normal", nrm, "material", surfs[idx].material_index)
hp = o + t * d
p = hp + nrm * 1e-6
f32 = np.float32
U8 = f32(12 * 6.0e-8)
jit = PixelKeyedJitter(int(os.environ.get("RT_DIAG_SEED", "1234")))
for li, L in enumerate(lights):
    tl = L.position - p; ldist = np.sqrt(tl @ tl); ld = tl / ldist
    ref = np.array([0, 1.0, 0]) if abs(ld[0]) >= 0.9 else np.array([1.0, 0, 0])
    right = np.cross(ld, ref); right /= max(np.sqrt(right @ right), 1e-6)
    up = np.cross(right, ld); up /= max(np.sqrt(up @ up), 1e-6)
    cellw = L.radius / n_sh
    for cell in range(n_sh * n_sh):
        i, j = divmod(cell, n_sh)
        ui, uj = jit.uniforms(np.array([y * W + x]), 0, li, cell)
        oi = -L.radius / 2 + cellw * (i + ui[0]); oj = -L.radius / 2 + cellw * (j + uj[0])
        S = (L.position + oi * right) + oj * up
        ts = S - p; dist = np.sqrt(ts @ ts); dr = ts / dist
        so = p + dr * 1e-6
        T = oracle.shadow_transmission(mats, surfs, so[None], dr[None], np.array([dist - 1e-6]))[0]
        # fp32 emulation of the sphere / box tests from p
        tr = 1.0; notes = []
        d32 = dr.astype(f32)
        for k, s in enumerate(surfs):
            tau = mats[s.material_index - 1].transparency
            if isinstance(s, Sphere):
                ocd = p - s.position
                cc = f32((ocd @ ocd) - s.radius * s.radius)
                oc = ocd.astype(f32)
                b = f32(oc[2] * d32[2] + f32(oc[1] * d32[1] + f32(oc[0] * d32[0])))
                disc = f32(b * b - cc)
                ocn = np.sqrt(max(cc + f32(s.radius) ** 2, 0)); e_b = U8 * ocn; kap = 4 * e_b; kap0 = 4 * e_b * e_b + 4 * 6e-8 * abs(cc)
                flag = kap * abs(b) + kap0 >= abs(disc)
                if disc >= 0:
                    sq = np.sqrt(disc); t1 = sq - b if b < 0 else -b - sq; t2 = cc / t1
                    in1 = 2e-6 < t1 < dist; in2 = 2e-6 < t2 < dist
                else:
                    in1 = in2 = False
                # exact
                a_ = dr @ dr; oc2 = so - s.position; bb = 2 * (oc2 @ dr); c2 = oc2 @ oc2 - s.radius ** 2; D = bb * bb - 4 * a_ * c2
                if D >= 0:
                    e1 = 1e-6 < (-bb - np.sqrt(D)) / (2 * a_) < dist - 1e-6; e2 = 1e-6 < (-bb + np.sqrt(D)) / (2 * a_) < dist - 1e-6
                else:
                    e1 = e2 = False
                if (in1 + in2) != (e1 + e2) or (flag and (in1 or in2 or e1 or e2 or abs(disc) < 1)):
                    notes.append(f"sph#{k} fp32 in {in1},{in2} exact {e1},{e2} flag {flag} disc {disc:.3e} b {b:.3e} kap|b|+kap0 {kap*abs(b)+kap0:.3e} D/4 {D/4:.3e} tau {tau}")
                if in1: tr *= tau
                if in2: tr *= tau
            elif isinstance(s, Cube):
                lo = (s.min_bound - p).astype(f32); hi = (s.max_bound - p).astype(f32)
                with np.errstate(divide="ignore", invalid="ignore"):
                    jx = f32(1) / d32
                    a3 = lo * jx; b3 = hi * jx
                tn = np.max(np.minimum(a3, b3)); tf = np.min(np.maximum(a3, b3))
                ok = tn <= tf
                in1 = ok and 2e-6 < tn < dist; in2 = ok and 2e-6 < tf < dist
                jm = np.max(np.abs(jx)); rel = U8 * jm + 4 * 6e-8
                flag = abs(tf - tn) <= 2 * rel * (abs(tn) + abs(tf)) or jm > 1e3
                with np.errstate(divide="ignore", invalid="ignore"):
                    A = (s.min_bound - so) / dr; B = (s.max_bound - so) / dr
                tmin = np.max(np.minimum(A, B)); tmax = np.min(np.maximum(A, B))
                v = tmin <= tmax
                e1 = v and 1e-6 <= tmin < dist - 1e-6; e2 = v and 1e-6 <= tmax < dist - 1e-6
                if (in1 + in2) != (e1 + e2) or (flag and abs(tf - tn) < 1):
                    notes.append(f"box#{k} fp32 in {in1},{in2} exact {e1},{e2} flag {flag} tn {tn:.6e} tf {tf:.6e} tmin {tmin:.6e} tmax {tmax:.6e} tau {tau}")
                if in1: tr *= tau
                if in2: tr *= tau
            else:
                N = s.normal; den = dr @ N
                tt = (s.offset - so @ N) / den if abs(den) >= 1e-10 else np.inf
                if 1e-6 < tt < dist - 1e-6: tr *= tau
        mark = "" if abs(tr - T) < 1e-6 else "   <<<<<< MISMATCH"
        if notes or mark:
            print(f"light {li} cell {cell}: exact T {T:.6f} fp32-emulated {tr:.6f}{mark}")
            for nline in notes: print("     ", nline)
