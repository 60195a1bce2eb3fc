#!/usr/bin/env python3
"""BASELINE config 5 (`resnet 3 20 1 n false`, images/hour on N GPUs): images are independent ciphertexts, so N GPUs = N
processes of the `conv resnet` CLI, one per device (HCONV_DEVICE), each classifying its own share of the images with --threads image
threads (own context and non-blocking stream each, cached allocations: 2 is the measured optimum on MI355X); no collective.
Prints one JSON line: images/hour = images / max over ranks of the summed per-image "Total done in" times (context and key
generation, which the reference also keeps outside its per-image timer, are reported separately).
Usage: tools/resnet_throughput.py [--gpus N] [--images M per GPU] [--depth 20] [--ker 3]"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def to_seconds(tok):
    m = re.match(r"([0-9.]+)(µs|ms|s)$", tok)
    return float(m.group(1)) * {"µs": 1e-6, "ms": 1e-3, "s": 1.0}[m.group(2)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--images", type=int, default=24)
    ap.add_argument("--depth", type=int, default=20)
    ap.add_argument("--ker", type=int, default=3)
    ap.add_argument("--procs-per-gpu", type=int, default=1, help="independent CLI processes sharing one device: a layer's launches are mostly one wave of workgroups, so the images of several processes overlap on the CUs")
    ap.add_argument("--threads", type=int, default=2, help="HCONV_IMAGE_THREADS: image threads inside one CLI process (own context and stream each); the images of a GPU are shared among them")
    a = ap.parse_args()
    import golden.gen_resnet_csv as rgen
    cli = os.path.join(ROOT, "optimal_conv_amd", "host", "conv")
    work = tempfile.mkdtemp(prefix="resnet_tp_")
    rgen.write_case(work, a.ker, a.depth, a.images)
    t0 = time.time()
    procs = [subprocess.Popen([cli, "--test-mode", "resnet", str(a.ker), str(a.depth), "1", str(a.images), "false"], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                              env=dict(os.environ, HCONV_DEVICE=str(r // a.procs_per_gpu), HCONV_SEED=str(100 + r), HCONV_IMAGE_THREADS=str(a.threads), HCONV_ASYNC_ALLOC=os.environ.get("HCONV_ASYNC_ALLOC", "1"))) for r in range(a.gpus * a.procs_per_gpu)]
    per_rank, spans = [], []
    for p in procs:
        out = p.communicate()[0]
        if p.returncode:
            sys.exit(f"rank failed ({p.returncode}):\n{out[-2000:]}")
        per_rank.append([to_seconds(t) for t in re.findall(r"^Total done in (\S+) $", out, re.M)])
        m = re.search(r"^All \d+ images done in (\S+)  \(", out, re.M)          # concurrent image threads: the window in which all of this process's images ran
        spans.append(to_seconds(m.group(1)) if m else sum(per_rank[-1]))
    wall = time.time() - t0
    slowest = max(spans)
    print(json.dumps({"metric": "encrypted ResNet inference, images/hour", "value": a.gpus * a.procs_per_gpu * a.images / slowest * 3600.0, "unit": "images/hour", "n_gpus": a.gpus,
                      "config": {"workload": f"resnet {a.ker} {a.depth} 1 {a.images} false", "images_per_gpu": a.images * a.procs_per_gpu, "processes_per_gpu": a.procs_per_gpu, "image_threads_per_process": a.threads, "data": "synthetic weights and images"},
                      "seconds_per_image_latency": [sum(t) / len(t) for t in per_rank], "seconds_for_all_images": spans, "wall_seconds_including_context_and_keys": wall}))


if __name__ == "__main__":
    main()
