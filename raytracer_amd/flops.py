"""Algorithmic flop model of the hot path — the single source of truth for roofline numbers
(BASELINE.md §3).  Convention: add/sub/mul/div/sqrt/min/max/pow = 1, FMA = 2,
compare/select/abs/neg/integer = 0; brute force, every query tests every surface.
"""

PER_QUERY_BASE = 7      # a = d.d, 4a, 2a hoisted per ray (sphere.py:50,54,67)
PER_SPHERE = 23         # oc 3 + b 6 + c 6 + disc 3 + sqrt 1 + t1 2 + t2 2 (sphere.py:49-68)
PER_PLANE = 12          # two dots 10 + sub + div (infinite_plane.py:30-40)
PER_BOX = 24            # 3 axes x (2 sub + 2 div + min + max + running max + running min) (cube.py:91-107)
PER_RAY = 20            # winner normal / flip
PER_HIT = 41            # hit point, view dir, eps offset, surface colour, accumulate (ray_tracer.py:517-563)
PER_HIT_LIGHT = 112     # 62 Phong (530-557) + 50 basis (198-213)
PER_SHADOW_RAY = 38     # sample generation (224-241)
PER_CHILD = 30          # spawn (565-595)


def flops_per_query(n_sph, n_pln, n_box):
    return PER_QUERY_BASE + PER_SPHERE * n_sph + PER_PLANE * n_pln + PER_BOX * n_box


def algorithmic_flops(*, n_sph, n_pln, n_box, n_lights, n_shadow, rays, hits, children, segments):
    """FLOPs = Q*F_q + 20*R + Hits*(41 + L*112 + L*n^2*38) + 30*children, Q = R + Sh + segments."""
    cells = n_shadow * n_shadow if n_shadow > 1 else 1
    shadow = hits * n_lights * cells
    q = rays + shadow + max(segments, 0)
    return (q * flops_per_query(n_sph, n_pln, n_box) + PER_RAY * rays +
            hits * (PER_HIT + n_lights * PER_HIT_LIGHT + n_lights * cells * PER_SHADOW_RAY) + PER_CHILD * children)


def shade_kernel_flops(*, n_sph, n_pln, n_box, n_lights, n_shadow, hits, segments=0, rays=0, children=0):
    """The share of the model executed by the shade kernel for `hits` hits (its launch's
    algorithmic flops): shadow queries + per-hit shading terms — plus, for the radiance rays the kernel
    traces itself (fused frames: `rays` of them, `children` spawned), their nearest-hit queries, winner
    normals and the spawn arithmetic, i.e. the whole of algorithmic_flops() when every level is fused."""
    cells = n_shadow * n_shadow if n_shadow > 1 else 1
    shadow = hits * n_lights * cells
    return ((shadow + max(segments, 0) + rays) * flops_per_query(n_sph, n_pln, n_box) + PER_RAY * rays + PER_CHILD * children +
            hits * (PER_HIT + n_lights * PER_HIT_LIGHT + n_lights * cells * PER_SHADOW_RAY))


def rays_traced_by_shade_kernel(stats):
    """(rays, children) whose nearest-hit queries ran inside the shade kernel, from an rt_stats dict of a frame rendered
    with RT_FLAG_TIMING: no k_trace launch at all -> every ray; one launch fewer than levels -> the camera rays."""
    levels = [r for r in stats["level_rays"] if r > 0]
    if not levels:
        return 0, 0
    if stats["trace_launches"] == 0:
        return int(stats["rays"]), int(stats["rays"]) - int(stats["level_rays"][0])
    if stats["trace_launches"] < stats["levels"]:
        return int(stats["level_rays"][0]), 0
    return 0, 0
