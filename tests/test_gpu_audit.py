"""Audit of the wave-level shortcuts of k_shade_ls on the headline workload (pool.txt 4K, 5x5, depth 3).

The audit build of the library (make -C raytracer_amd/csrc debug -> librt_hip_debug.so; same
sources with -DRT_DEBUG_KERNEL) can run the sample loop for EVERY (hit, light) pair and compare
its result with what the shortcut would have claimed for the lane ("no primitive can cross this
lane's shadow fan" -> all n^2 terms are 1; "one opaque sphere swallows the whole fan" -> all 0).
Any wrong claim would make the image depend on which hits share a wave."""
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

DEBUG_LIB = os.path.join(ROOT, "raytracer_amd", "librt_hip_debug.so")


def test_shortcut_claims_agree_with_the_sample_loop():
    if not os.path.exists(DEBUG_LIB):
        pytest.fail("librt_hip_debug.so is missing: run __graft_entry__.build() (make -C raytracer_amd/csrc debug)")
    env = dict(os.environ, RT_HIP_LIBRARY=DEBUG_LIB, RT_DEBUG_STATS="2")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "quick_bench.py"), "pool4k", "--frames", "1", "--host-out"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    found = re.findall(r"fully-lit claims wrong (\d+), umbra claims wrong (\d+), penumbra lanes (\d+) of (\d+)", p.stderr)
    assert found, p.stderr[-2000:]            # one line per row band of the host-output frame
    wrong_lit, wrong_umbra, penumbra, total = (sum(int(f[k]) for f in found) for k in range(4))
    print(f"audit: {total} (hit, light) pairs, {penumbra} need the sample loop, wrong claims: {wrong_lit} lit, {wrong_umbra} umbra")
    assert wrong_lit == 0 and wrong_umbra == 0
    assert total == 10979324 * 5 and 0 < penumbra < total


def _audit_counts(stderr):
    found = re.findall(r"culling audit: (\d+) exact \(fp64\) crossings of records by shadow-segment bundles, (\d+) bundles", stderr)
    return sum(int(a) for a, _ in found), sum(int(b) for _, b in found), len(found)


def test_large_scene_culling_never_drops_a_crossed_record():
    """k_shade_fast's large-scene path drops records at four places (wave-level leaf list, wave-level record
    list, per-lane node tests, per-lane fan test).  The audit build checks in fp64, for every (hit, light),
    that every record crossed by one of the lane's shadow segments reached the lane's sample block: on the
    synthetic 4609-primitive scene and on randomised scenes with lights inside the slab, huge light radii,
    cameras among the objects and floor hits thousands of units away (tools/stress_accel.py)."""
    if not os.path.exists(DEBUG_LIB):
        pytest.fail("librt_hip_debug.so is missing: run __graft_entry__.build() (make -C raytracer_amd/csrc debug)")
    env = dict(os.environ, RT_HIP_LIBRARY=DEBUG_LIB, RT_DEBUG_STATS="65536")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "quick_bench.py"), "synth4k", "--scale", "4", "--frames", "1", "--host-out"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    crossings, dropped, frames = _audit_counts(p.stderr)
    print(f"synthetic scene 960x540: {crossings} exact crossings over {frames} frames, {dropped} dropped")
    assert frames >= 1 and crossings > 1000000 and dropped == 0, p.stderr[-2000:]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_accel.py"), "100", "45"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-2000:])
    crossings, dropped, frames = _audit_counts(p.stderr)
    print(f"45 randomised large scenes: {crossings} exact crossings over {frames} frames, {dropped} dropped")
    assert frames >= 45 and crossings > 10000 and dropped == 0, p.stderr[-2000:]


def test_fp32_decided_pairs_agree_with_the_exact_walk():
    """Audit build, RT_DEBUG_STATS bit 25: every (hit, light) pair that k_shade_ls / k_shade_fast decide themselves (not
    queued for the re-test; k_shade_ls: including its fully-lit / umbra claims) is evaluated again with the literal fp64
    walk of every cell;
    a pair whose sums differ by more than 1e-3 is printed.  Randomised small scenes (opaque and transparent, the ones
    with more than 6 spheres / 2 planes / boxes go through k_shade_fast) and large ones: no line may appear.  Sharper
    than the 8-bit image comparison: it sees a wrong decision even where the pixel does not change."""
    if not os.path.exists(DEBUG_LIB):
        pytest.fail("librt_hip_debug.so is missing: run __graft_entry__.build() (make -C raytracer_amd/csrc debug)")
    env = dict(os.environ, RT_HIP_LIBRARY=DEBUG_LIB, RT_DEBUG_STATS=str(1 << 25))
    # the headline kernel on the headline scene (1920x1080; the full 4K frame by hand: 54,853,298 pairs, 0 differ)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "quick_bench.py"), "pool4k", "--scale", "2", "--frames", "1", "--host-out"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    found = re.findall(r"walk audit: (\d+) \(hit, light\) pairs decided in fp32 checked against the exact walk, (\d+) differ", p.stderr)
    assert found, p.stderr[-2000:]
    checked, differ = sum(int(a) for a, _ in found), sum(int(b) for _, b in found)
    print(f"pool.txt 1920x1080: {checked} pairs decided by k_shade_ls checked against the exact walk, {differ} differ")
    assert checked > 13000000 and differ == 0
    # settings_survey seeds 1236 / 1255: samples with two crossings of different primitives < 2e-6 apart (found by this audit
    # in round 2: 3 pairs differed; the merged-crossing test of round 3, rt_api.hip mark_near_pairs, sends them to the exact walk)
    for tool, args in (("random_fp32_survey.py", ["60"]), ("stress_accel.py", ["0", "12"]), ("settings_survey.py", ["1", "1236"]),
                       ("settings_survey.py", ["1", "1255"])):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)] + args, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
        lines = [l for l in (p.stdout + p.stderr).splitlines() if l.startswith("[walk audit] level")]
        print(f"{tool}: {len(lines)} fp32-decided pairs differ from the exact walk")
        assert not lines, lines[:5]
