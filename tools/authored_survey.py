#!/usr/bin/env python3
"""fp32-vs-oracle sweep over "authored" scenes: everything on a half-unit grid, so surfaces coincide exactly the way
hand-made scenes do — boxes on the floor, stacked, side by side and against walls, spheres resting on boxes / floor /
each other (tangent), spheres cut by planes, lights on grid points (axis-aligned shadow rays), cameras on axes.
tools/authored_survey.py [n_seeds [first_seed]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from raytracer_amd import parse_scene_text, separate_objects
from raytracer_amd.render import render_hip
from raytracer_amd.image import quantize_u8
from oracle import oracle
oracle.set_threads(16)


def authored_scene_text(seed):
    rng = np.random.default_rng(seed)
    transparent = seed % 2 == 1
    h = lambda lo, hi: float(rng.integers(int(lo * 2), int(hi * 2) + 1)) / 2.0      # half-unit grid
    cam_choices = ["cam 0 2 9 0 1 0 0 1 0 1.4 2", "cam 6 3.5 6 0 1 0 0 1 0 1.2 2", "cam 0 8 0.001 0 0 0 0 1 0 1.0 2",
                   "cam 0 1 9 0 1 0 0 1 0 1.4 2", "cam -7 0.5 2 0 1 0 0 1 0 1.4 2"]
    lines = [cam_choices[int(rng.integers(len(cam_choices)))],
             f"set 0.2 0.25 0.3 {int(rng.integers(1, 5))} {int(rng.integers(1, 5))}"]
    taus = [0.0, 0.0, 0.0, 0.0] if not transparent else [float(rng.choice([0.0, 0.3, 0.6, 1.0])) for _ in range(4)]
    for k in range(4):
        c = rng.uniform(0.2, 0.9, 3); refl = float(rng.choice([0.0, 0.3]))
        lines.append(f"mtl {c[0]:.3f} {c[1]:.3f} {c[2]:.3f} 0.5 0.5 0.5 {refl} {refl} {refl} {int(rng.choice([0, 1, 10, 50]))} {taus[k]}")
    lines.append("pln 0 1 0 0 1")
    if rng.random() < 0.6: lines.append("pln 0 0 1 -4 2")
    if rng.random() < 0.3: lines.append("pln 1 0 0 -4 3")
    if rng.random() < 0.2: lines.append("pln 0 -1 0 -6 4")
    tops = {}                                                # column (x, z) -> current height
    if seed % 5 == 4:                                        # large variant: a floor tiled with cubes (shared faces), > 128 surfaces
        n_t = int(rng.integers(12, 20))
        for ix in range(n_t):
            for iz in range(n_t):
                x, z = ix - n_t // 2 + 0.0, iz - n_t // 2 - 2.0
                up_ = 1.0 if rng.random() < 0.08 else 0.0    # some tiles stand one unit proud
                lines.append(f"box {x} {-0.5 + up_} {z} 1.0 {int(rng.integers(1, 5))}")
                tops[(x, z)] = up_
    for _ in range(int(rng.integers(1, 9))):
        x, z = h(-3, 3), h(-3, 3)
        s = float(rng.choice([0.5, 1.0, 1.0, 2.0]))
        y0 = tops.get((x, z), 0.0)
        if rng.random() < 0.15: y0 -= 0.5 * s                # sunk into what is below
        lines.append(f"box {x} {y0 + s / 2} {z} {s} {int(rng.integers(1, 5))}")
        tops[(x, z)] = y0 + s
    for _ in range(int(rng.integers(0, 6))):
        x, z = h(-3, 3), h(-3, 3)
        r = float(rng.choice([0.25, 0.5, 0.5, 1.0]))
        y0 = tops.get((x, z), 0.0)
        mode = rng.random()
        cy = y0 + r if mode < 0.7 else (y0 if mode < 0.85 else y0 + r + h(0, 2))
        lines.append(f"sph {x} {cy} {z} {r} {int(rng.integers(1, 5))}")
        tops[(x, z)] = cy + r
    for _ in range(int(rng.integers(1, 4))):
        lines.append(f"lgt {h(-5, 5)} {h(1, 8)} {h(-3, 6)} {rng.uniform(0.4, 1):.3f} {rng.uniform(0.4, 1):.3f} {rng.uniform(0.4, 1):.3f} 1 {rng.uniform(0.3, 1):.3f} {float(rng.choice([0.0, 0.5, 1.0, 2.0]))}")
    return "\n".join(lines) + "\n"


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    FIRST = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0; worst = 0; retested = 0; pairs = 0
    for seed in range(FIRST, FIRST + N):
        cam, st, objs = parse_scene_text(authored_scene_text(seed))
        mats, surfs, lights = separate_objects(objs)
        W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (80, 60)      # tools/authored_survey.py [n [first_seed [W H]]]
        ref = oracle.render(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed))
        s32 = {}
        img64 = render_hip(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed), verbose=False)
        img = render_hip(cam, st, mats, surfs, lights, W, H, jitter=("pixel", seed), precision="fp32", stats=s32, verbose=False)
        e64 = float((np.abs(img64 - ref) / np.maximum(np.abs(ref), 1e-3)).max())
        d8 = np.abs(quantize_u8(img).astype(int) - quantize_u8(ref).astype(int))
        worst = max(worst, int(d8.max())); retested += s32["retested"]; pairs += s32["hits"] * len(lights)
        if d8.max() > 1 or e64 > 1e-9:
            bad += 1; print("seed", seed, "surfaces", len(surfs), "channels > 1:", int((d8 > 1).sum()), "max", int(d8.max()), "fp64 rel err", e64, flush=True)
    print(f"{N} scenes, {bad} bad, worst fp32 diff {worst}, {retested} of {pairs} pairs re-tested")
