#!/usr/bin/env python3
"""Copies what tools/gpu_r5_final.sh left under gpurun_out/<OUT>/ into the tracked profiles/round5_* files (the ones the docs and bench.py cite), stamping the commit the run was
made on (gpurun_out/<OUT>/provenance.txt). usage: python tools/collect_r5_profiles.py r5final3"""
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


pmc = os.path.join(out, "pmc")
stamp(os.path.join(pmc, "traffic_convrelu_5_1_n4.json"), os.path.join(prof, "round5_traffic_convrelu_5_1.json"), {"pack32": 2})
conv = json.load(open(os.path.join(pmc, "traffic.json")))
conv["measured_with"] = {"command": open(os.path.join(pmc, "command.txt")).read().strip(), "contexts": 1, "ciphertexts_per_launch_set": 4, "commit": commit, "method": conv.pop("method", "")}
json.dump(conv, open(os.path.join(prof, "round5_traffic_conv_B256.json"), "w"), indent=1)
for src, dst in (("table.txt", "round5_conv33_counters.txt"), ("counters.json", "round5_conv33_counters.json"), ("valu.json", "round5_conv33_valu.json")):
    shutil.copy(os.path.join(pmc, src), os.path.join(prof, dst)); print("wrote profiles/" + dst)
for src, dst in ((os.path.join(pmc, "stats", "run_kernel_stats.csv"), "round5_conv33_kernel_stats_one_stream.csv"), (os.path.join(pmc, "chain_stats", "run_kernel_stats.csv"), "round5_convrelu_kernel_stats.csv"),
                 (os.path.join(out, "resnet_stats", "run_kernel_stats.csv"), "round5_resnet20_kernel_stats.csv")):
    shutil.copy(src, os.path.join(prof, dst)); print("wrote profiles/" + dst)
log = open(os.path.join(out, "pytest_gpu.log")).read().splitlines()
tail = [l for l in log if "passed" in l or "slowest" in l or l.strip().endswith("s call     " + l.split("call")[-1].strip()) or " call " in l][-14:]
with open(os.path.join(prof, "round5_pytest_gpu_summary.txt"), "w") as f:
    f.write(f"# pytest tests -m gpu -x -q --durations=10 on one MI355X (round 5, commit {commit}; tools/gpu_r5_final.sh)\n")
    f.write("\n".join(tail) + "\n")
print("wrote profiles/round5_pytest_gpu_summary.txt")
if os.path.exists(os.path.join(out, "bench_20_5.json")):
    shutil.copy(os.path.join(out, "bench_20_5.json"), os.path.join(prof, "round5_bench_conv33_20_5.json")); print("wrote profiles/round5_bench_conv33_20_5.json")
