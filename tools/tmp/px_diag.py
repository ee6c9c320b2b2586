import os, sys
ROOT = "/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
src = open(os.path.join(ROOT, "tools", "scaled_survey.py")).read().split("N = int(sys.argv[1])")[0]
ns = {"__file__": os.path.join(ROOT, "tools", "scaled_survey.py")}; exec(compile(src, "h", "exec"), ns)
scaled = ns["scaled"]; random_scene_text = ns["random_scene_text"]
from raytracer_amd import parse_scene_text, separate_objects
from raytracer_amd.scene import Sphere, InfinitePlane, Cube
from raytracer_amd.jitter import PixelKeyedJitter
from oracle import oracle
s, opaque, seed, y, x = float(sys.argv[1]), sys.argv[2] == "1", int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
cam, st, objs = parse_scene_text(scaled(random_scene_text(1000 * int(opaque) + seed, opaque), s))
mats, surfs, lights = separate_objects(objs)
W, H = 56, 40
cam.setup(W / H)
o, d = oracle.camera_rays(cam, W, H)
pix = y * W + x
t, idx, nrm = oracle.nearest_hit(surfs, o[pix][None], d[pix][None]); t, idx, nrm = t[0], idx[0], nrm[0]
hp = o[pix] + t * d[pix]; p = hp + nrm * 1e-6
print("hit surface", idx, type(surfs[idx]).__name__, "t", t, "hp", hp, "n", nrm)
n_sh = int(st.root_number_shadow_rays)
jit = PixelKeyedJitter(seed)
for li, L in enumerate(lights):
    tl = L.position - p; ldist = np.sqrt(tl @ tl); ld = tl / ldist
    ref = np.array([0, 1.0, 0]) if abs(ld[0]) >= 0.9 else np.array([1.0, 0, 0])
    right = np.cross(ld, ref); right /= max(np.sqrt(right @ right), 1e-6)
    up = np.cross(right, ld); up /= max(np.sqrt(up @ up), 1e-6)
    cellw = L.radius / n_sh
    print("light", li, "pos", L.position, "radius", L.radius, "|tl|", ldist, "n.l", nrm @ ld)
    for k, sf in enumerate(surfs):
        if isinstance(sf, Sphere):
            oc = p - sf.position; print("   sphere", k, "r", sf.radius, "|oc|", np.sqrt(oc @ oc), "cc", oc @ oc - sf.radius ** 2, "|L-c|", np.linalg.norm(L.position - sf.position))
    for cell in range(n_sh * n_sh):
        i, j = divmod(cell, n_sh)
        ui, uj = jit.uniforms(np.array([pix]), 0, li, cell)
        oi = -L.radius / 2 + cellw * (i + ui[0]); oj = -L.radius / 2 + cellw * (j + uj[0])
        S = (L.position + oi * right) + oj * up
        ts = S - p; dist = np.sqrt(ts @ ts); dr = ts / dist
        so = p + dr * 1e-6
        T = oracle.shadow_transmission(mats, surfs, so[None], dr[None], np.array([dist - 1e-6]))[0]
        hits = []
        for k, sf in enumerate(surfs):
            tt, _ = oracle.intersect(sf, so[None], dr[None])
            if np.isfinite(tt[0]): hits.append((k, float(tt[0])))
        print("   cell", cell, "T", T, "dist", dist, "first hits (surface, t):", hits)
