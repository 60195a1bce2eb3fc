#!/usr/bin/env python3
"""Static instruction mix of the gfx950 kernels: compiles csrc/hconv.hip to ISA and counts, per kernel, VALU instructions by
class, then prices them in VALU issue cycles with the per-instruction rates measured on MI355X (profiles/round1_ubench.txt:
v_mad_u64_u32 31.5, v_mul_lo_u32 26.8, v_mul_hi_u32 17.6 T lane-op/s against a plain-VALU peak of 256 CUs x 64 lanes x 2.4 GHz
= 39.3 T/s, i.e. 1.25 / 1.47 / 2.23 issue slots). With the measured per-job time this gives the fraction of the VALU pipe each
kernel keeps busy: the roofline that actually bounds these integer kernels.
Usage: tools/isa_mix.py [per-job-us as name=us ...]   (per-job = kernel duration / jobs, one job = one 2^16-coefficient row)"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK = 256 * 64 * 2.4e9
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", *os.environ.get("ISA_DEFS", "").split(), *([a for a in os.environ.get("ISA_DEFS","").split() if a]), "-I" + ROOT + "/optimal_conv_amd/csrc", "-I" + ROOT + "/include",
                       "-S", "--cuda-device-only", "-o", "/tmp/hconv_isa.s", ROOT + "/optimal_conv_amd/csrc/hconv.hip"])
lines = open("/tmp/hconv_isa.s").read().split("\n")
per_job = dict(a.split("=") for a in sys.argv[1:])
want = ["hc_k_a1ILi1", "hc_k_a2ILi1ELi1", "hc_k_a3ILi1", "hc_k_b1", "hc_k_b2ILi2", "hc_k_b3ILi2", "hc_k_b4ILi1", "hc_k_b5ILi1"]
print(f"{'kernel':14s} {'instr':>6s} {'VALU':>6s} {'mul':>5s} {'mov':>5s} {'nop':>5s} {'vmem':>5s} {'lds':>4s}  issue slots per job -> model time vs measured")
for w in want:
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\d+" + w + r".*:\s", l))
    end = start
    while not lines[end].startswith(".Lfunc_end"):
        end += 1
    cnt = collections.Counter()
    for l in lines[start:end]:
        m = re.match(r"^\s+([a-z_0-9]+)\s", l)
        if m:
            cnt[m.group(1)] += 1
    valu = sum(v for k, v in cnt.items() if k.startswith("v_"))
    mul = sum(v for k, v in cnt.items() if k.startswith(("v_mad_u64", "v_mul_lo", "v_mul_hi")))
    mov = sum(v for k, v in cnt.items() if k.startswith("v_mov"))
    nop = cnt["s_nop"]
    vmem = sum(v for k, v in cnt.items() if k.startswith(("global_", "buffer_")))
    lds = sum(v for k, v in cnt.items() if k.startswith("ds_"))
    name = w.split("ILi")[0].replace("hc_k_", "")
    slots = valu - mul + 1.25 * cnt["v_mad_u64_u32"] + 1.47 * cnt["v_mul_lo_u32"] + 2.23 * cnt["v_mul_hi_u32"]
    lane_slots = slots * 256 * 16         # one job = 16 workgroups x 256 threads (straight-line code, no loops)
    extra = ""
    if name in per_job:
        t_model = lane_slots / PEAK * 1e6
        extra = f"model {t_model:.3f} us/job vs measured {float(per_job[name]):.3f} -> VALU pipe {100 * t_model / float(per_job[name]):4.1f}% busy"
    print(f"{name:14s} {sum(cnt.values()):6d} {valu:6d} {mul:5d} {mov:5d} {nop:5d} {vmem:5d} {lds:4d}  {lane_slots / 1e6:6.2f} M lane-slots/job  {extra}")
