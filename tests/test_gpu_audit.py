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
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "quick_bench.py"), "pool4k", "--frames", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    m = re.search(r"fully-lit claims wrong (\d+), umbra claims wrong (\d+), penumbra lanes (\d+) of (\d+)", p.stderr)
    assert m, p.stderr[-2000:]
    wrong_lit, wrong_umbra, penumbra, total = map(int, m.groups())
    print(f"audit: {total} (hit, light) pairs, {penumbra} need the sample loop, wrong claims: {wrong_lit} lit, {wrong_umbra} umbra")
    assert wrong_lit == 0 and wrong_umbra == 0
    assert total == 10979324 * 5 and 0 < penumbra < total
