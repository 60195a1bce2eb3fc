#!/usr/bin/env python3
"""tools/chain_table.py <dir> <images per launch set> <layers in run>: per-kernel table of the convReLU chain from the rocprofv3 passes tools/gpu_r6_chain_counters.sh left under
<dir> (stats, sq, sq2, l2, fetch, write, grbm), both roofs side by side, and <dir>/chain_valu.json: lane-instructions and the VALU issue floor per ciphertext-layer.
Every instruction class these kernels are made of issues at one wave64 instruction per 4 cycles per SIMD (profiles/round2_ubench_instr.txt), and SQ_ACTIVE_INST_VALU counts exactly
that quad-cycle per instruction on gfx950: issue time of a kernel = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x clock)."""
import collections
import csv
import glob
import json
import sys

O, NB, IT = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
layers = NB * IT                                   # ciphertext-layers in the run


def kname(s):
    return s.split("(")[0].replace("void ", "")


def layer_region(rows):
    """the dispatches of the LAYERS: everything from the first convolution on (hc_k_ctc_pairs opens conv_then_pack). What runs before it is set-up - key generation and the
    encoding of the DFT diagonals, 1 279 single-row and 401 batched transforms - and is not a layer's work (rounds 4-5 tables had it in: the launch order is the program's,
    so the cut is the same dispatch in every pass)"""
    ids = [int(r["Dispatch_Id"]) for r in rows if "hc_k_ctc_pairs" in r["Kernel_Name"]]
    first = min(ids) if ids else 0
    return [r for r in rows if int(r["Dispatch_Id"]) >= first]


acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
vg = {}
for p in ("sq", "sq2", "l2", "fetch", "write", "grbm"):
    for f in glob.glob(O + "/" + p + "/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in layer_region(list(csv.DictReader(open(f)))):
            k = kname(r["Kernel_Name"])
            acc[k][p + ":" + r["Counter_Name"]] += float(r["Counter_Value"])
            vg[k] = r.get("VGPR_Count", "?")
            if p == "sq" and (k, r["Dispatch_Id"]) not in seen:
                seen.add((k, r["Dispatch_Id"])); n[k] += 1
dur = collections.defaultdict(float)
calls = collections.Counter()
for f in glob.glob(O + "/stats/**/*kernel_trace.csv", recursive=True):
    for r in layer_region(list(csv.DictReader(open(f)))):
        k = kname(r["Kernel_Name"])
        dur[k] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9
        calls[k] += 1
# the harness' kernels (key generation, encoding, encryption) are not the layer
HARNESS = ("hc_k_swk_", "hc_k_encode", "hc_k_prep_ker", "hc_k_ker_", "hc_k_make_pairs", "hc_k_fill", "hc_k_bl_", "hc_k_pointwise", "hc_k_cols_fwd<", "hc_k_rows_fwd_canon<", "hc_k_cols_inv<", "hc_k_rows_inv<")   # the last four: the single-row L0 transforms of encryption / decryption
rows = []
tot_t = tot_floor = tot_rd = tot_wr = tot_lane = 0.0
for k in sorted(dur, key=lambda x: -dur[x]):
    if not (k.startswith("hc_k_") or k.startswith("__amd_rocclr_copyBuffer")) or k.startswith(HARNESS) or not calls[k]:
        continue
    a, t = acc[k], dur[k]
    clk = a.get("grbm:GRBM_GUI_ACTIVE", 0) / 8.0 / t if a.get("grbm:GRBM_GUI_ACTIVE") else 2.3e9      # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    wc = a.get("sq:SQ_WAVE_CYCLES", 0.0) or 1.0
    wc2 = a.get("sq2:SQ_WAVE_CYCLES", 0.0) or 1.0
    issue = 4 * a.get("sq:SQ_ACTIVE_INST_VALU", 0.0) / (1024 * clk)                               # seconds of VALU issue at 100 % of the pipe
    rd, wr = a.get("fetch:FETCH_SIZE", 0.0) * 1024 * 2.0, a.get("write:WRITE_SIZE", 0.0) * 1024   # KiB; FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md, HBM)
    hit = a.get("l2:TCC_HIT_sum", 0.0); miss = a.get("l2:TCC_MISS_sum", 0.0)
    waves = a.get("sq:SQ_WAVES", 0.0) or 1.0
    rows.append((k, calls[k], 1e6 * t / calls[k], 1e3 * t / layers, vg.get(k, "?"), a.get("sq:SQ_INSTS_VALU", 0.0) / waves, issue / t, 100 * a.get("sq:SQ_WAIT_ANY", 0) / wc,
                 100 * a.get("sq:SQ_WAIT_INST_ANY", 0) / wc, 100 * a.get("sq2:SQ_WAIT_INST_LDS", 0) / wc2,
                 100 * a.get("sq2:SQ_LDS_BANK_CONFLICT", 0) / (a.get("sq2:SQ_LDS_IDX_ACTIVE", 0) or 1.0), hit / ((hit + miss) or 1.0), rd / layers / 1e9, wr / layers / 1e9, (rd + wr) / t / 1e12, clk / 1e9))
    tot_t += t; tot_floor += issue; tot_rd += rd; tot_wr += wr; tot_lane += 64 * a.get("sq:SQ_INSTS_VALU", 0.0)
print("convReLU 5 1 x %d layers at %d images per launch set = %d ciphertext-layers. ms/ctl = kernel time per ciphertext-layer; VALUbusy = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x time x clock);" % (IT, NB, layers))
print("waitAny / waitInst / waitLDS = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_WAIT_INST_LDS over SQ_WAVE_CYCLES; LDScf = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; rd / wr = fabric GB per ciphertext-layer (FETCH_SIZE x 2, WRITE_SIZE)")
print("%-28s %6s %8s %7s %5s %5s %9s %8s %8s %8s %8s %6s %6s %7s %7s %6s %6s" % ("kernel", "calls", "us/call", "ms/ctl", "%", "VGPR", "VALU/wave", "VALUbusy", "waitAny%", "waitIns%", "waitLDS%", "LDScf%", "L2hit", "rdGB", "wrGB", "TB/s", "clkGHz"))
for r in rows:
    print("%-28s %6d %8.1f %7.3f %5.1f %5s %9.0f %8.2f %8.1f %8.1f %8.1f %6.1f %6.2f %7.3f %7.3f %6.2f %6.2f" % (r[0][:28], r[1], r[2], r[3], 100 * r[3] * layers / 1e3 / tot_t, r[4], *r[5:]))
ms = 1e3 * tot_t / layers
print("sum of chain kernel time: %.2f ms per ciphertext-layer; VALU issue floor %.2f ms (valu_frac of the kernel time %.3f); fabric %.2f GB per ciphertext-layer = %.2f TB/s" %
      (ms, 1e3 * tot_floor / layers, tot_floor / tot_t, (tot_rd + tot_wr) / layers / 1e9, (tot_rd + tot_wr) / tot_t / 1e12))
per_kernel = {r[0]: {"read": r[12] * 1e9, "write": r[13] * 1e9, "launches_per_layer_launch_set": r[1] / IT, "ms_per_ct_layer": r[3]} for r in rows}
json.dump({"bytes_per_ciphertext_layer": (tot_rd + tot_wr) / layers, "read_bytes_per_ciphertext_layer": tot_rd / layers, "write_bytes_per_ciphertext_layer": tot_wr / layers,
           "per_kernel_per_ciphertext_layer": per_kernel,
           "measured_with": {"command": "HCONV_IMAGE_BATCH=%d HCONV_SKIP_BL=1 conv convReLU 5 1 %d under rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes)" % (NB, IT),
                             "images_per_launch_set": NB, "method": "FETCH_SIZE x 2 (gfx950) + WRITE_SIZE, KiB units; the LAYERS' dispatches only (from the first hc_k_ctc_pairs on; harness kernels excluded): rounds 4-5 counted the set-up transforms (key generation, DFT diagonals) in"}},
          open(O + "/traffic_convrelu_5_1.json", "w"), indent=1)
json.dump({"issue_floor_ms": 1e3 * tot_floor / layers, "lane_instr_per_ct_layer": tot_lane / layers, "kernel_ms_per_ct_layer": ms, "valu_busy_of_kernel_time": tot_floor / tot_t,
           "fabric_bytes_per_ct_layer": (tot_rd + tot_wr) / layers, "images_per_launch_set": NB, "layers_in_run": IT,
           "method": "tools/gpu_r6_chain_counters.sh: rocprofv3 --kernel-trace --pmc passes over `conv convReLU 5 1 2`; issue floor = sum over the layer's kernels of 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x clock), clock from GRBM_GUI_ACTIVE; harness kernels (key generation, encoding) excluded"},
          open(O + "/chain_valu.json", "w"), indent=1)
