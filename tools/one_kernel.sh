#!/bin/bash
# Compile ONE kernel instantiation (fast iteration on codegen): prints its resource usage and
# leaves the assembly in /tmp/one_kernel.s
#   tools/one_kernel.sh 'rt::k_shade_ls<8, 2, 0, true>' rt_shade_ls.hpp '(rt::DevScene, rt::DevCamera, rt::FrameParams, int, rt::RayQueue, rt::RayQueue, const rt::HitRec*, rt::Counters*, float*)'
cd "$(dirname "$0")/../raytracer_amd/csrc" || exit 1
K=$1; H=$2; ARGS=${3:-'(rt::DevScene, rt::DevCamera, rt::FrameParams, int, rt::RayQueue, rt::RayQueue, const rt::HitRec*, rt::Counters*, float*, rt::FixRec*)'}
printf '#define RT_NO_LAUNCHERS\n#include "%s"\ntemplate __global__ void %s%s;\n' "$H" "$K" "$ARGS" > /tmp/one_kernel.hip
FLAGS="-I. -O3 -std=c++17 -ffp-contract=off -fno-fast-math -munsafe-fp-atomics --offload-arch=gfx950 --cuda-device-only -Wno-bitwise-instead-of-logical"
/opt/rocm/bin/hipcc $FLAGS -Rpass-analysis=kernel-resource-usage -S -o /tmp/one_kernel.s /tmp/one_kernel.hip 2>&1 | sed -E 's/ \[-Rpass-analysis=kernel-resource-usage\]//' | grep -A12 "Function Name: .*k_\(shade\|trace\)" | grep -E "VGPRs:|Spill|Occupancy|LDS Size|ScratchSize|error" 
