#!/usr/bin/env python3
"""Per-kernel table from the rocprofv3 passes of tools/gpu_r2_pmc.sh. Normalisation (written down because round 1's table was misread):
  * every SQ_* cycle counter here is summed over all waves (SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_*: quad-cycles spent by EACH
    resident wave) or over all SIMD/CU instances (SQ_BUSY_CYCLES); a per-wave percentage such as ACTIVE_INST_VALU / WAVE_CYCLES says
    what fraction of ITS residency a wave spent issuing VALU instructions. The VALU pipe of a SIMD is shared by all waves resident on
    it, so pipe utilisation = (per-wave fraction) x (waves resident per SIMD), estimated as 4 x SQ_INSTS_VALU issue cycles /
    (kernel duration x clock x SIMDs)  [column valu_pipe%: wave64 VALU instructions x 4 cycles each over 1024 SIMDs].
  * FETCH_SIZE is doubled (gfx950: 128-byte requests tallied at 64 bytes, MI355X_MICROARCH.md); WRITE_SIZE as reported; both KiB.
"""
import collections, csv, glob, os, sys
O = sys.argv[1]
def load(name):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(os.path.join(O, name, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            d = (k, r.get("Dispatch_Id"))
            if d not in seen: seen.add(d); cnt[k] += 1
    return acc, cnt
def durations():
    dur = collections.defaultdict(float); calls = collections.Counter()
    for f in glob.glob(os.path.join(O, "stats", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            dur[k] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3; calls[k] += 1
    return dur, calls
fe, _ = load("fetch"); wr, _ = load("write"); tc, _ = load("tcc"); s1, n1 = load("sq1"); s2, _ = load("sq2"); gr, _ = load("grbm")
dur, calls = durations()
ks = [k for k in sorted(dur, key=lambda k: -dur[k]) if k.startswith("hc_k_")][:12]
rows_json = {}
print("%-22s %6s %9s %8s %8s %7s %6s | %9s %7s %7s %7s %8s | %7s %7s" % ("kernel", "calls", "us/call", "rdMiB/c", "wrMiB/c", "TB/s", "L2hit%", "VALU/wave", "actV%", "waitI%", "waitA%", "valu_pipe%", "LDScf%", "clkGHz"))
for k in ks:
    c = max(1, calls[k]); us = dur[k] / c
    rd = fe[k].get("FETCH_SIZE", 0) * 2 / 1024 / max(1, c); w = wr[k].get("WRITE_SIZE", 0) / 1024 / max(1, c)
    tb = (rd + w) * 2 ** 20 / (us * 1e-6) / 1e12 if us else 0
    hit = tc[k].get("TCC_HIT_sum", 0); miss = tc[k].get("TCC_MISS_sum", 0)
    a = s1[k]; wc = a.get("SQ_WAVE_CYCLES", 0) or 1; waves = a.get("SQ_WAVES", 0) or 1
    clk = gr[k].get("GRBM_GUI_ACTIVE", 0) / 8.0 / max(1, c) / (us * 1e-6) / 1e9 if us else 0      # the counter is summed over the 8 XCDs
    pipe = 100.0 * (a.get("SQ_INSTS_VALU", 0) / max(1, c)) * 4 / (us * 1e-6 * (clk or 2.0) * 1e9 * 1024) if us else 0
    b = s2[k]
    print("%-22s %6d %9.1f %8.1f %8.1f %7.2f %6.1f | %9.0f %7.1f %7.1f %7.1f %8.1f | %7.1f %7.2f" % (k[:22], c, us, rd, w, tb, 100 * hit / max(1, hit + miss),
          a.get("SQ_INSTS_VALU", 0) / waves, 100 * a.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * a.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * a.get("SQ_WAIT_ANY", 0) / wc, pipe,
          100 * b.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, b.get("SQ_LDS_IDX_ACTIVE", 0)), clk))
    rows_json[k] = {"calls": c, "us_per_call": us, "read_MiB_per_call": rd, "write_MiB_per_call": w, "fabric_TBps": tb, "l2_hit_pct": 100 * hit / max(1, hit + miss),
                    "valu_instr_per_wave": a.get("SQ_INSTS_VALU", 0) / waves, "wave_active_valu_pct": 100 * a.get("SQ_ACTIVE_INST_VALU", 0) / wc,
                    "wave_wait_inst_pct": 100 * a.get("SQ_WAIT_INST_ANY", 0) / wc, "wave_wait_any_pct": 100 * a.get("SQ_WAIT_ANY", 0) / wc,
                    "valu_pipe_pct": pipe, "lds_conflict_cycles_per_wave": b.get("SQ_LDS_BANK_CONFLICT", 0) / waves, "lds_active_cycles_per_wave": b.get("SQ_LDS_IDX_ACTIVE", 0) / waves, "clock_GHz": clk}
tot_us = sum(dur[k] for k in ks)
print("sum of kernel time: %.1f us; total fabric bytes: %.1f MiB" % (tot_us, sum(fe[k].get("FETCH_SIZE", 0) * 2 / 1024 + wr[k].get("WRITE_SIZE", 0) / 1024 for k in ks)))

import json
convs = int(os.environ.get("PMC_CONVS", "0"))
out = {"method": "rocprofv3 --kernel-trace --pmc, separate passes (FETCH_SIZE x2 gfx950 correction; WRITE_SIZE as reported; SQ counters per wave; "
                 "valu_pipe_pct = wave64 VALU instructions x 4 cycles / (duration x clock x 1024 SIMDs); clock = GRBM_GUI_ACTIVE / 8 XCDs / duration)",
       "command": os.environ.get("PMC_COMMAND", "bench.py --steps 2 --warmup 1 --batch 8 --streams 1"), "kernels": rows_json}
if convs:
    out["convs_in_run"] = convs
    out["fabric_bytes_per_conv"] = sum((v["read_MiB_per_call"] + v["write_MiB_per_call"]) * v["calls"] for k, v in rows_json.items() if k.startswith("hc_k_a") or k.startswith("hc_k_b")) * 2 ** 20 / convs
json.dump(out, open(os.path.join(O, "counters.json"), "w"), indent=1)
