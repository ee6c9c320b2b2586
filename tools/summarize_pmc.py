#!/usr/bin/env python3
"""Fold the rocprofv3 --pmc CSVs written by tools/profile_pmc.sh into one JSON summary per kernel.

    python tools/summarize_pmc.py gpurun_out/pmc_pool4k pool4k > profiles/r01_pmc_pool4k.json
Counter values are summed over the dispatches of one frame (the last of the 3 profiled frames'
worth is not separable, so totals are divided by the number of frames).  FETCH_SIZE / WRITE_SIZE are
in KiB per the rocprofv3 derived-counter definition; FETCH_SIZE is doubled (gfx950 counts 128-B
requests as 64 B for wide coalesced reads, MI355X_MICROARCH.md §HBM).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    root, workload = sys.argv[1], sys.argv[2]
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    per = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    newest = {}   # one CSV per pass directory: the most recent (gpurun merges runs into one tree)
    for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
        key = os.path.relpath(path, root).split(os.sep)[0]
        if key not in newest or os.path.getmtime(path) > os.path.getmtime(newest[key]):
            newest[key] = path
    for path in newest.values():
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"].split("(")[0].replace("void ", "")
                per[k][row["Counter_Name"]] += float(row["Counter_Value"])
                calls[k][row["Counter_Name"]] += 1
    out = {"workload": workload, "frames_profiled": frames, "kernels": {}}
    for k, ctr in per.items():
        n_launch = max(calls[k].values()) / frames
        d = {"launches_per_frame": n_launch}
        for name, v in sorted(ctr.items()):
            d[name + "_per_frame"] = v / frames
        if "SQ_INSTS_VALU" in ctr and "SQ_BUSY_CYCLES" in ctr:
            d["valu_insts_per_wave"] = ctr["SQ_INSTS_VALU"] / max(ctr.get("SQ_WAVES", 1), 1)
        if "SQ_ACTIVE_INST_VALU" in ctr and "SQ_WAVE_CYCLES" in ctr and ctr["SQ_WAVE_CYCLES"]:
            d["valu_active_frac_of_wave_cycles"] = ctr["SQ_ACTIVE_INST_VALU"] / ctr["SQ_WAVE_CYCLES"]
        hbm = 0.0
        if "FETCH_SIZE" in ctr:
            d["hbm_read_bytes_per_launch"] = 2.0 * ctr["FETCH_SIZE"] * 1024 / frames / n_launch
            hbm += d["hbm_read_bytes_per_launch"]
        if "WRITE_SIZE" in ctr:
            d["hbm_write_bytes_per_launch"] = ctr["WRITE_SIZE"] * 1024 / frames / n_launch
            hbm += d["hbm_write_bytes_per_launch"]
        if hbm:
            d["hbm_bytes_per_launch"] = hbm
        out["kernels"][k] = d
    shade = [v for k, v in out["kernels"].items() if "k_shade" in k and "hbm_bytes_per_launch" in v]
    if shade:
        out["hbm_bytes_per_shade_launch"] = max(s["hbm_bytes_per_launch"] for s in shade)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
