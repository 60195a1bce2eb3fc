#!/usr/bin/env python3
"""VALU issue floor of the conv kernels, priced per instruction class.

Static side (runs anywhere hipcc does): compile csrc/hconv.hip to gfx950 assembly (-S --cuda-device-only), histogram the VALU mnemonics of every kernel and
price each at the issue rate tools/ubench4.hip measured on MI355X (profiles/round2_ubench_instr.txt, 8 workgroups per CU; those figures assume 2.4 GHz, the chip
holds ~2.1 GHz under these kernels, so they are scaled by 2.1 / 2.4): moves / 32-bit adds / logic / v_fma_f32 issue in ~2.4-3.0 cycles per wave64 instruction per
SIMD, multiplies, 64-bit adds, carries, compares, shifts, selects and fp64 in ~4.0-4.3. The kernels are straight-line (fully unrolled butterflies), so the static
class mix is the dynamic one up to the trip counts of the few row-batch loops.
Dynamic side (from a rocprofv3 pass, tools/gpu_r4_valu.sh): SQ_INSTS_VALU per kernel and launch; SQ_ACTIVE_INST_VALU equals it on gfx950 (one quad-cycle per
instruction whatever its class: the counter prices everything at 4 cycles), SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, GRBM_GUI_ACTIVE for the clock.
usage: valu_floor.py static [out.json]            -> per-kernel class histogram and mean cycles per VALU instruction
       valu_floor.py combine <pmc_dir> <convs> [out.json]   -> lane-instructions per conv, counter busy fraction, class-priced issue floor
"""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCALE = 2.1 / 2.4          # ubench cycles assumed 2.4 GHz; the chip runs ~2.1 GHz under load (GRBM_GUI_ACTIVE / duration)
# measured, wgs/cu 8 column of profiles/round2_ubench_instr.txt (cycles per wave64 instruction per SIMD at the assumed clock)
MEASURED = {"v_mov_b32": 2.79, "v_add_u32": 3.41, "v_sub_u32": 3.41, "v_subrev_u32": 3.41, "v_and_b32": 3.32, "v_or_b32": 3.32, "v_xor_b32": 3.33, "v_fma_f32": 2.72,
            "v_add3_u32": 4.88, "v_lshlrev_b32": 4.63, "v_lshrrev_b32": 4.63, "v_alignbit_b32": 4.66, "v_lshl_or_b32": 4.72, "v_and_or_b32": 4.68, "v_bfe_u32": 4.55,
            "v_perm_b32": 4.66, "v_add_co_u32": 4.62, "v_addc_co_u32": 4.87, "v_sub_co_u32": 4.62, "v_subb_co_u32": 4.87, "v_mul_lo_u32": 4.90, "v_mul_hi_u32": 4.54,
            "v_mad_u64_u32": 4.90, "v_min_u32": 4.45, "v_max_u32": 4.45}
FAST_DEFAULT, FULL_DEFAULT = 3.0, 4.7      # unmeasured mnemonics: 32-bit move/logic class vs everything else (64-bit, compares, selects, fp64, DPP, conversions)
FAST_RE = re.compile(r"^v_(mov_b32|not_b32|add_f32|sub_f32|mul_f32|max_f32|min_f32|accvgpr)")


def cycles(mn):
    base = mn.split("_e32")[0].split("_e64")[0].split("_dpp")[0].split("_sdwa")[0]
    if base in MEASURED:
        return MEASURED[base] * SCALE
    return (FAST_DEFAULT if FAST_RE.match(base) else FULL_DEFAULT) * SCALE


def static_histogram():
    asm = "/tmp/hconv_gfx950.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-S", "--cuda-device-only", "-o", asm, os.path.join(ROOT, "optimal_conv_amd", "csrc", "hconv.hip")])
    kernels, cur = {}, None
    for line in open(asm):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1); kernels[cur] = collections.Counter(); continue
        if cur is None:
            continue
        t = line.strip()
        if t.startswith("s_endpgm"):
            cur = None; continue
        m = re.match(r"^(v_\w+)", t)
        if m:
            kernels[cur][m.group(1)] += 1
    out = {}
    for k, h in kernels.items():
        n = sum(h.values())
        if not n:
            continue
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
        cyc = sum(cycles(mn) * c for mn, c in h.items())
        fast = sum(c for mn, c in h.items() if cycles(mn) < 3.5 * SCALE + 0.5)
        out[name] = {"static_valu": n, "mean_cycles_per_valu": cyc / n, "fast_class_share": fast / n, "top": dict(h.most_common(8))}
    return out


def main():
    if sys.argv[1] == "static":
        out = static_histogram()
        for k, v in sorted(out.items()):
            if k.startswith("hc_k_a") or k.startswith("hc_k_b"):
                print(f"{k:28s} {v['static_valu']:6d} VALU, mean {v['mean_cycles_per_valu']:.2f} cycles each, {100 * v['fast_class_share']:.0f} % in the fast class")
        if len(sys.argv) > 2:
            json.dump(out, open(sys.argv[2], "w"), indent=1)
        return
    if sys.argv[1] == "combine":
        import csv
        import glob
        pmc, convs = sys.argv[2], int(sys.argv[3])
        st = static_histogram()
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for f in glob.glob(os.path.join(pmc, "sq1", "**", "*counter_collection.csv"), recursive=True):
            seen = set()
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                if (k, r.get("Dispatch_Id")) not in seen:
                    seen.add((k, r.get("Dispatch_Id"))); n[k] += 1
        dur = collections.defaultdict(float)
        for f in glob.glob(os.path.join(pmc, "stats", "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                dur[r["Kernel_Name"].split("(")[0].replace("void ", "")] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9
        gr = collections.defaultdict(float)
        for f in glob.glob(os.path.join(pmc, "grbm", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    gr[r["Kernel_Name"].split("(")[0].replace("void ", "")] += float(r["Counter_Value"])
        rows, tot_inst, tot_floor_cyc, tot_busy4, tot_simd_cyc, tot_t = {}, 0.0, 0.0, 0.0, 0.0, 0.0
        for k, a in acc.items():
            if not (k.startswith("hc_k_a") or k.startswith("hc_k_b") or k.startswith("hc_k_sb") or k.startswith("hc_k_ctc")):
                continue
            base = k.split("<")[0]
            mean = next((v["mean_cycles_per_valu"] for kk, v in st.items() if kk == k or kk.split("<")[0] == base), FULL_DEFAULT * SCALE)
            inst = a.get("SQ_INSTS_VALU", 0.0)            # wave64 instructions
            t = dur.get(k, 0.0)
            clk = gr[k] / 8.0 / t if t and gr.get(k) else 2.1e9
            simd_cyc = t * clk * 1024
            rows[k] = {"launches": n[k], "wave_valu_instr": inst, "mean_cycles_per_valu_priced": mean, "kernel_seconds": t, "clock_GHz": clk / 1e9,
                       "busy_frac_counter_4cyc": 4 * a.get("SQ_ACTIVE_INST_VALU", 0.0) / simd_cyc if simd_cyc else None,
                       "busy_frac_priced": mean * inst / simd_cyc if simd_cyc else None}
            tot_inst += inst; tot_floor_cyc += mean * inst; tot_busy4 += 4 * a.get("SQ_ACTIVE_INST_VALU", 0.0); tot_simd_cyc += simd_cyc; tot_t += t
        clk = tot_simd_cyc / 1024 / tot_t if tot_t else 2.1e9
        out = {"convs_in_run": convs, "lane_instr_per_conv": 64 * tot_inst / convs, "kernel_ms_per_conv_one_stream": 1e3 * tot_t / convs,
               "busy_frac_counter": tot_busy4 / tot_simd_cyc, "busy_frac_priced": tot_floor_cyc / tot_simd_cyc,
               "issue_floor_ms_counter_4cyc": 1e3 * tot_busy4 / 1024 / clk / convs, "issue_floor_ms": 1e3 * tot_floor_cyc / 1024 / clk / convs, "clock_GHz": clk / 1e9,
               "method": "SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU (rocprofv3 --pmc, one context) per kernel; busy_frac_counter = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x duration x clock) "
                         "(the counter tallies one quad-cycle per VALU instruction on gfx950: = rocprof's VALUBusy); busy_frac_priced / issue_floor_ms price each instruction class at its "
                         "measured issue rate (llvm histogram per kernel x profiles/round2_ubench_instr.txt scaled to the measured clock)", "kernels": rows}
        print(json.dumps({k: v for k, v in out.items() if k != "kernels"}, indent=1))
        if len(sys.argv) > 4:
            json.dump(out, open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
