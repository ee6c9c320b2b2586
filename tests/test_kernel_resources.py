"""Build-time guard for the headline kernel: k_shade_ls<6,2,0,true> (pool.txt class, opaque) sits just
below the 256-VGPR limit of 2 waves/SIMD; one more live register halves its occupancy and costs ~60 % of
the frame rate (measured twice while developing).  Compiles that single instantiation (a few seconds,
tools/one_kernel.sh) and checks the compiler's resource report."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="hipcc not available")
@pytest.mark.parametrize("fall,max_spill", [(2, 6), (3, 2)])      # level 0 / deeper levels of a fused frame
def test_headline_kernel_keeps_two_waves_per_simd(fall, max_spill):
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "one_kernel.sh"), f"rt::k_shade_ls<6, 2, 0, true, {fall}>", "rt_shade_ls.hpp"],
                         capture_output=True, text=True, timeout=600).stdout
    occ = re.search(r"Occupancy \[waves/SIMD\]: (\d+)", out)
    spill = re.search(r"VGPRs Spill: (\d+)", out)
    vgprs = re.search(r" VGPRs: (\d+)", out)
    assert occ and vgprs and spill, out
    assert int(occ.group(1)) >= 2, out
    # (round 3, with the level-0 query fused in: ~10 loop-invariant values are parked in scratch before the chunk loop and
    # re-read once per 64-hit chunk, outside the light and sample loops; the deeper levels' instantiation spills nothing)
    assert int(spill.group(1)) <= max_spill, out
    assert int(vgprs.group(1)) <= 256


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="hipcc not available")
@pytest.mark.parametrize("kernel,header,args,min_occ,max_spill", [
    # large-scene shade kernel, 3x3 shadow grid: 2 waves/SIMD (LDS lists + <= 256 VGPRs)
    ("rt::k_shade_fast<9, false, true, false>", "rt_shade_fast.hpp",
     "(rt::DevScene, rt::DevCamera, rt::FrameParams, int, rt::RayQueue, rt::RayQueue, const rt::HitRec*, rt::Counters*, float*, int, rt::FixRec*)", 2, 64),
    # large-scene trace kernel: four workgroups per CU (nodes + hit staging + wave lists <= 40 KiB, <= 128 VGPRs);
    # the hierarchy walk is a latency chain, 3 -> 4 waves/SIMD was worth 14 %
    ("rt::k_trace<float, false>", "rt_kernels.hpp",
     "(rt::DevScene, rt::DevCamera, rt::FrameParams, int, rt::RayQueue, rt::HitRec*, rt::Counters*, float*)", 4, 32),
])
def test_large_scene_kernels_keep_their_occupancy(kernel, header, args, min_occ, max_spill):
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "one_kernel.sh"), kernel, header, args],
                         capture_output=True, text=True, timeout=600).stdout
    occ = re.search(r"Occupancy \[waves/SIMD\]: (\d+)", out)
    spill = re.search(r"VGPRs Spill: (\d+)", out)
    assert occ and spill, out
    assert int(occ.group(1)) >= min_occ, out
    # (round 3: the harness used to cut the report off before this line.  The large-scene shade kernel has spilled ~50
    # VGPRs since round 2 — outside its sample blocks —; the budget keeps that from growing unnoticed)
    assert int(spill.group(1)) <= max_spill, out
