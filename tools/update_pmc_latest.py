#!/usr/bin/env python3
"""profiles/pmc_latest.json (what bench.py's roofline.executed reads) from a tools/summarize_pmc.py summary:
    python tools/update_pmc_latest.py profiles/r02_pmc_pool4k.json pool4k 2 [packed_static_share]
The SQ_INSTS_VALU_{ADD,MUL,FMA}_F32 counters count a packed instruction (v_pk_fma_f32 ...) once, so the counted flops
are a lower bound; the upper bound doubles them (every counted instruction packed).  packed_static_share = share of
v_pk_* among the fp32 add/mul/fma instructions in the kernel's ISA (tools/one_kernel.sh + grep), recorded beside."""
import json, sys
src, workload, rnd = sys.argv[1], sys.argv[2], int(sys.argv[3])
share = float(sys.argv[4]) if len(sys.argv) > 4 else None
d = json.load(open(src))
shade = [v for k, v in d["kernels"].items() if "k_shade" in k]
fix = [v for k, v in d["kernels"].items() if "k_fix" in k]
def tot(name): return sum(v.get(name + "_per_frame", 0.0) for v in shade + fix)
amf = tot("SQ_INSTS_VALU_ADD_F32") + tot("SQ_INSTS_VALU_MUL_F32") + 2 * tot("SQ_INSTS_VALU_FMA_F32")
lower = 64 * (amf + tot("SQ_INSTS_VALU_TRANS_F32"))
upper = 64 * (2 * amf + tot("SQ_INSTS_VALU_TRANS_F32"))
f64 = 64 * (tot("SQ_INSTS_VALU_ADD_F64") + tot("SQ_INSTS_VALU_MUL_F64") + 2 * tot("SQ_INSTS_VALU_FMA_F64"))
# HBM bytes of the whole frame: every kernel's bytes per launch x its launches per frame (FETCH_SIZE doubled as summarize_pmc.py does)
frame_bytes = sum(v.get("hbm_bytes_per_launch", 0.0) * v.get("launches_per_frame", 0.0) for v in d["kernels"].values())
per_kernel = {k.replace("rt::", "").split("<")[0]: round(v.get("hbm_bytes_per_launch", 0.0) * v.get("launches_per_frame", 0.0))
              for k, v in d["kernels"].items() if v.get("hbm_bytes_per_launch")}
path = "profiles/pmc_latest.json"
try:
    out = json.load(open(path))
except Exception:
    out = {}
out[workload] = {
    "hbm_bytes_per_shade_launch": d.get("hbm_bytes_per_shade_launch"),
    "hbm_bytes_per_frame": frame_bytes,
    "hbm_bytes_per_frame_by_kernel": per_kernel,
    "executed_fp32_flops_per_frame": lower,
    "executed_fp32_flops_per_frame_upper": upper,
    "packed_static_share_of_fp32_add_mul_fma": share,
    "executed_fp64_flops_per_frame": f64,
    "valu_wave_instructions_per_frame": tot("SQ_INSTS_VALU"),
    "salu_wave_instructions_per_frame": tot("SQ_INSTS_SALU"),
    "valu_issue_peak_Tinstr_per_s": 0.58,
    "valu_active_fraction_of_wave_cycles": (tot("SQ_ACTIVE_INST_VALU") / tot("SQ_WAVE_CYCLES")) if tot("SQ_WAVE_CYCLES") else None,
    "waves_per_simd": 2,
    "round": rnd,
    "note": "shade + re-test launches of one frame; SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32 x 64 lanes, FMA = 2; a packed instruction is counted "
            "once (lower bound), upper bound = every add/mul/fma packed; issue peak 0.57-0.59 T wave-instr/s measured by tools/microbench/valu_rates.hip "
            "(profiles/r01_valu_rates_microbench.txt)",
    "source": src,
}
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out[workload], indent=1))
