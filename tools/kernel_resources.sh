#!/bin/bash
# Compact per-kernel resource table (VGPRs, SGPRs, spills, scratch, occupancy, LDS) from hipcc remarks.
cd "$(dirname "$0")/../raytracer_amd/csrc" || exit 1
make -s resource-usage 2>&1 | sed -E 's/ \[-Rpass-analysis=kernel-resource-usage\]//' | awk '
/Function Name:/ {name=$NF}
/TotalSGPRs:/ {sg=$NF} / VGPRs:/ {vg=$NF} /ScratchSize/ {sc=$NF} /Occupancy/ {oc=$NF} /SGPRs Spill/ {ss=$NF} /VGPRs Spill/ {vs=$NF}
/LDS Size/ {lds=$NF; printf "vgpr %-4s sgpr %-4s sspill %-4s vspill %-3s scratch %-4s occ %-2s lds %-6s %s\n", vg, sg, ss, vs, sc, oc, lds, name}' | \
  sed -E 's/_ZN2rt[0-9]+//; s/EE?vNS_8DevScene.*//; s/ENS_8DevScene.*//' | sort -k16
