#!/usr/bin/env python3
"""HBM-side traffic per conv from two rocprofv3 passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; kernel-trace only).
FETCH_SIZE is doubled for gfx950 (MI355X_MICROARCH.md, HBM section); both counters are in KiB.
usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <convs in the run> <out.json>"""
import collections
import csv
import json
import sys

fetch_csv, write_csv, convs, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]


def total(path, name, corr):
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if k.startswith("hc_k_a") or k.startswith("hc_k_b") or "pointwise<3>" in k or "pointwise<1>" in k:
                acc[k] += float(r["Counter_Value"]) * 1024 * corr / convs
    return acc


rd, wr = total(fetch_csv, "FETCH_SIZE", 2.0), total(write_csv, "WRITE_SIZE", 1.0)
res = {"bytes_per_conv": sum(rd.values()) + sum(wr.values()), "read_bytes_per_conv": sum(rd.values()), "write_bytes_per_conv": sum(wr.values()),
       "per_kernel": {k: {"read": rd.get(k, 0.0), "write": wr.get(k, 0.0)} for k in sorted(set(rd) | set(wr))},
       "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; FETCH_SIZE x2 (gfx950 correction); KiB units",
       "convs_in_run": convs}
json.dump(res, open(out, "w"), indent=1)
print(f"traffic per conv: {res['bytes_per_conv'] / 1e9:.3f} GB (read {res['read_bytes_per_conv'] / 1e9:.3f}, write {res['write_bytes_per_conv'] / 1e9:.3f})")
