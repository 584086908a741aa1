"""profiles/<round>/valu_ceiling.json from the microbenchmark's output (tools/ubench_valu_ceiling > valu_ceiling_raw.json) and the three
`rocprofv3 --pmc SQ_INSTS_VALU` passes over its Goldilocks classes (counter_collection.csv each): wave-instructions per unit of the
gl::mul / Poseidon2 lane form / Poseidon2 quad form loops, their best rates, and the summary bench.py's roofline_valu reads.
Usage: python tools/make_valu_ceiling.py r05 <glmul csv> <lane csv> <quad csv>   (tools/run_round_profiles.sh does this)"""
import csv
import json
import os
import sys

ROUND = sys.argv[1]
D = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", ROUND)
raw = json.load(open(os.path.join(D, "valu_ceiling_raw.json")))
# iterations and lane-units per wave-iteration of the three counted classes (tools/ubench_valu_ceiling.hip `modes`)
COUNTED = {"gl::mul x4 chains": (2000, 64 * 64), "p2::permute (lane form)": (200, 64), "p2::Coop4::permute (quad form)": (400, 16)}


def insts_per_unit(path, iters, units):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == "SQ_INSTS_VALU"]
    by = {}
    for r in rows:
        by.setdefault(r["Dispatch_Id"], [0.0, float(r["Grid_Size"])])[0] += float(r["Counter_Value"])
    v, grid = next(iter(by.values()))
    return v / (grid / 64) / iters / units


pmc = {name: insts_per_unit(p, *COUNTED[name]) for name, p in zip(COUNTED, sys.argv[2:5])}
out = {"source": "tools/ubench_valu_ceiling.hip on one MI355X (every CU busy, no memory traffic in the timed loops), 1 / 2 / 4 / 8 waves per SIMD pinned by the "
                 "LDS request; SQ_INSTS_VALU of the last three classes by `rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU -- tools/ubench_valu_ceiling <class>` on the same binary",
       "device_cus": raw["device_cus"], "simds": raw["simds"], "nominal_clock_hz": 2.4e9, "classes": []}
for c in raw["classes"]:
    if c["class"].startswith("v_cndmask_b32_e32"):
        continue  # reads an uninitialised vcc in the microbenchmark: 23 cycles whatever the occupancy, unexplained, not used
    e = {"class": c["class"], "unit": c["unit"], "by_waves_per_simd": c["by_waves_per_simd"]}
    best = max(c["by_waves_per_simd"].values(), key=lambda v: v["units_per_s"])
    e["best_units_per_s"] = best["units_per_s"]
    if c["class"] in pmc:
        e["wave_insts_per_unit"] = pmc[c["class"]]
        e["best_wave_insts_per_s"] = best["units_per_s"] * pmc[c["class"]]
        e["cycles_per_wave_inst_per_simd_at_2.4GHz"] = 2.4e9 / (e["best_wave_insts_per_s"] / raw["simds"])
    else:
        e["best_wave_insts_per_s"] = best["wave_insts_per_s"]
        e["cycles_per_wave_inst_per_simd_at_2.4GHz"] = best["cycles_per_wave_inst_per_simd_at_2.4GHz"]
    out["classes"].append(e)
g = {c["class"]: c for c in out["classes"]}
lane, quad = g["p2::permute (lane form)"], g["p2::Coop4::permute (quad form)"]
out["summary"] = {
    "full_rate_wave_insts_per_s": g["v_add_u32"]["best_wave_insts_per_s"],
    "half_rate_wave_insts_per_s": g["v_mad_u64_u32"]["best_wave_insts_per_s"],
    "full_rate_ops": "v_add_u32 / v_sub_u32 / v_xor_b32 / v_mov_b32 / v_fma_f32 / v_fmac_f32 (VOP2 or VOP3 encoding, a 32-bit literal allowed): ~2.2 cycles per "
                     "wave-instruction per SIMD from 2 waves per SIMD on (one wave alone: 4.6)",
    "half_rate_ops": "v_mad_u64_u32, v_mul_lo/hi_u32, v_mul_u32_u24, v_mad_u32_u24, every shift, v_add3_u32, v_and_or_b32, v_cndmask_b32 on an SGPR mask, every add / "
                     "subtract that writes or reads a carry (VCC or an SGPR pair), v_lshl_add_u64, v_cmp_*_u64, DPP and SDWA moves, v_pk_add_u16, any VALU op with an SGPR "
                     "operand: ~4.1 cycles",
    "goldilocks_mix_wave_insts_per_s": {"p2_lane_form": lane["best_wave_insts_per_s"], "p2_quad_form": quad["best_wave_insts_per_s"],
                                        "gl_mul": g["gl::mul x4 chains"]["best_wave_insts_per_s"]},
    "wave_insts_per_permutation": {"p2_lane_form": lane["wave_insts_per_unit"], "p2_quad_form": quad["wave_insts_per_unit"]},
    "one_wave_per_simd": {"p2_quad_form_perm_per_s": quad["by_waves_per_simd"]["1"]["units_per_s"], "p2_lane_form_perm_per_s": lane["by_waves_per_simd"]["1"]["units_per_s"]},
    "note": "the old roofline_valu peak (256 x 4 x 2.4e9 / 4 = 6.14e11) was the half-rate class's nominal rate; a kernel of full-rate instructions can exceed it. A kernel's "
            "ceiling is the rate of ITS mix: the Goldilocks kernels are ~75 % half-rate instructions (the classes above give the two rates)."}
json.dump(out, open(os.path.join(D, "valu_ceiling.json"), "w"), indent=1)
print(json.dumps(out["summary"]["goldilocks_mix_wave_insts_per_s"]), json.dumps(out["summary"]["wave_insts_per_permutation"]))
