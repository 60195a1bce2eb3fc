#!/usr/bin/env python3
"""Copies what tools/gpu_r6_final.sh left under gpurun_out/<OUT>/ into the tracked profiles/round6_* files (the ones the docs and bench.py cite), stamping the commit the run was made
on (gpurun_out/<OUT>/provenance.txt). usage: python tools/collect_r6_profiles.py r6final"""
import json
import os
import shutil
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out", sys.argv[1])
prof = os.path.join(root, "profiles")
commit = open(os.path.join(out, "provenance.txt")).read().split()[1]


def stamp(src, dst, extra=None):
    d = json.load(open(src))
    d.setdefault("measured_with", {})
    if not isinstance(d["measured_with"], dict):
        d["measured_with"] = {"note": d["measured_with"]}
    d["measured_with"]["commit"] = commit
    d["measured_with"].update(extra or {})
    json.dump(d, open(dst, "w"), indent=1)
    print("wrote", os.path.relpath(dst, root))


def cp(src, dst):
    shutil.copy(src, os.path.join(prof, dst)); print("wrote profiles/" + dst)


pmc = os.path.join(out, "pmc")
conv = json.load(open(os.path.join(pmc, "traffic.json")))
conv["measured_with"] = {"command": open(os.path.join(pmc, "command.txt")).read().strip(), "contexts": 1, "ciphertexts_per_launch_set": 4, "commit": commit, "method": conv.pop("method", "")}
json.dump(conv, open(os.path.join(prof, "round6_traffic_conv_B256.json"), "w"), indent=1)
for src, dst in (("table.txt", "round6_conv33_counters.txt"), ("counters.json", "round6_conv33_counters.json"), ("valu.json", "round6_conv33_valu.json")):
    cp(os.path.join(pmc, src), dst)
cp(os.path.join(pmc, "stats", "run_kernel_stats.csv"), "round6_conv33_kernel_stats_one_stream.csv")
for sub, tag in (("chain", ""), ("chain1", "_n1")):
    ch = os.path.join(out, sub)
    with open(os.path.join(prof, f"round6_chain_counters{tag}.txt"), "w") as f:
        f.write(f"# tools/gpu_r6_chain_counters.sh on one MI355X, commit {commit}: seven separate rocprofv3 passes (--kernel-trace + --pmc only) over `conv convReLU 5 1 2`; the LAYERS' dispatches only\n")
        f.write(open(os.path.join(ch, "chain_counters.txt")).read())
    print(f"wrote profiles/round6_chain_counters{tag}.txt")
    stamp(os.path.join(ch, "traffic_convrelu_5_1.json"), os.path.join(prof, f"round6_traffic_convrelu_5_1{tag}.json"), {"pack32": 2})
    v = json.load(open(os.path.join(ch, "chain_valu.json"))); v["commit"] = commit
    json.dump(v, open(os.path.join(prof, f"round6_chain_valu{tag}.json"), "w"), indent=1); print(f"wrote profiles/round6_chain_valu{tag}.json")
    cp(os.path.join(ch, "stats", "run_kernel_stats.csv"), f"round6_convrelu_kernel_stats{tag}.csv")
cp(os.path.join(out, "resnet_stats", "run_kernel_stats.csv"), "round6_resnet20_kernel_stats.csv")
if os.path.exists(os.path.join(out, "bench_20_5.json")):
    cp(os.path.join(out, "bench_20_5.json"), "round6_bench_conv33_20_5.json")
