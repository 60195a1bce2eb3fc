#!/usr/bin/env python3
"""bench.py — homomorphic convs/sec of the `conv 3 3` hot path (evalConv_BN: conv_then_pack + bias add,
eval.go:250-260) on N MI355X GPUs.

A "step" is one pass of the hot path over one batch of input: one homomorphic convolution on each of the ciphertexts resident
on the GPU (--batch ciphertexts per hc_conv_then_pack_batch call, i.e. per launch set, times --streams contexts), each = one
N=2^16 level-1 ciphertext in, its own B=256 kernel plaintexts, one level-0 ciphertext out (conv.go:522-546 + eval.go:258);
`value` counts convolutions (steps x batch x streams x GPUs / time). Kernel plaintexts are
pre-encoded and excluded exactly as the reference excludes prep_Ker from its "Conv (with BN)" timer (eval.go:244). Inputs are synthetic uniform residues, resident in
HBM before the timed region. Multi-GPU: ciphertexts (images) are independent, so rank r runs its own convolutions
on GPU r with no data-path collective (weak scaling); torch.distributed only provides the barriers and the
max-over-ranks reduction of the elapsed time.

Prints ONE JSON line (rank 0). `roofline.achieved` = algorithmic bytes per conv ((5B-1+2.5*log2B+0.5)
MiB) / average conv duration measured with HIP events on the library's own stream; `cpu_baseline` = the CPU
oracle (a port of the reference's Go/Lattigo path, one thread) timed on this host on one full `conv 3 3`.

The line validates itself (`parity_check`): the oracle runs on the SAME planted inputs as ciphertext 0 of context 0, and its output is compared word for word with what
the last timed step left on the GPU (outputs are cleared before the timed region); the convReLU 5 1 workload's launch set runs once more on the input and keys that were
planted into the reference binary and its SHA-256 digests are compared with the binary's (tests/golden/ref_trace_chain_5_1.json); under N > 1 the sharded convolution
is compared with the oracle too. Any mismatch: the line is still printed, then the process exits with status 1.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

Q0, Q1, P0 = 0x80000000080001, 0x1FFFFFFEA0001, 0x1FFFFFFFFFE00001
N = 65536
BATCHS = [4, 16, 64, 256]          # main.go:578
WIDTHS = [128, 64, 32, 16]         # main.go:579


def algorithmic_mib(B):
    """compulsory HBM traffic of one conv with every phase fused, in MiB."""
    return 5 * B - 1 + 2.5 * np.log2(B) + 0.5


def traffic_from_profiles(B):
    """HBM-side (L2 <-> fabric) bytes per conv from the committed rocprofv3 PMC passes (tools/gpu_r3_pmc.sh -> tools/pmc_traffic.py).
    PMC counters cannot be read from inside this process, so the figure is the one measured with the command recorded beside it
    (`measured_with`: contexts, ciphertexts per launch set, commit) and committed under profiles/. (None, None) when there is none."""
    for name in (f"round6_traffic_conv_B{B}.json", f"round5_traffic_conv_B{B}.json", f"round4_traffic_conv_B{B}.json", f"round3_traffic_conv_B{B}.json", f"traffic_conv_B{B}.json"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
            return d["bytes_per_conv"], {"file": "profiles/" + name, **d.get("measured_with", {"note": d.get("method", "")})}
        except Exception:
            continue
    return None, None


def valu_from_profiles():
    """VALU side of the conv's roofline from the committed counter pass (tools/gpu_r4_pmc.sh -> tools/valu_floor.py): lane-instructions per conv, the counter-based busy
    fraction of the VALU pipe, and the issue floor with every instruction class priced at its measured rate. None when there is no such profile."""
    try:
        name = next(n for n in ("round6_conv33_valu.json", "round5_conv33_valu.json", "round4_conv33_valu.json") if os.path.exists(os.path.join(ROOT, "profiles", n)))
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        return {"lane_instr_per_conv": d["lane_instr_per_conv"], "busy_frac": d["busy_frac_counter"], "busy_frac_class_priced": d["busy_frac_priced"],
                "issue_floor_ms": d["issue_floor_ms"], "issue_floor_ms_at_4_cycles_per_instr": d["issue_floor_ms_counter_4cyc"], "kernel_ms_per_conv_one_stream": d["kernel_ms_per_conv_one_stream"],
                "clock_GHz": d["clock_GHz"], "measured_with": {"file": "profiles/" + name, "method": d["method"]}}
    except Exception:
        return None


def synth_rows(rng, q, shape):
    return (rng.integers(0, 1 << 62, size=shape, dtype=np.uint64) % np.uint64(q)).astype(np.uint64)


def oracle_conv(B, ct_in, pl_ker, keys, bias):
    """One conv_then_pack + bias on the CPU oracle (test infrastructure: the checker and the reported baseline, never the product) on the given planted
    data; returns (output residues (2, N), seconds). `keys` = [(galEl = 2^j + 1, [b_Q0, a_Q0, b_P, a_P])] as handed to hc_evk_load."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle
    O = Oracle()
    evk = np.zeros((16, 4, N), dtype=np.uint64)
    for gal, k4 in keys:
        evk[(gal - 1).bit_length() - 2] = np.stack(k4)          # row j - 1 for galEl = 2^j + 1 (tests/parity_cases.py: load_tree_keys)
    idx = O.idx_plaintexts()
    t0 = time.perf_counter()
    want, _ = O.conv_then_pack(ct_in, 2.0 ** 30, pl_ker, 2.0 ** 30, idx, evk, B, 1, 2.0 ** 30, bias)
    return want, time.perf_counter() - t0


def cpu_baseline(B, dt):
    return {"value": 1.0 / dt, "unit": "conv/s", "cores": 1, "kind": "port",
            "sample": f"1 full conv_then_pack+bias at B={B} (N=2^16) on the C oracle, {dt:.2f} s, host has {os.cpu_count()} cores; the SAME planted inputs as ciphertext 0 of context 0 of the timed region (parity_check compares the two outputs word for word)"}


def first_difference(got, want):
    """'ok' or the first differing word of two residue arrays, as the record the driver keeps"""
    got, want = np.asarray(got).reshape(-1), np.asarray(want).reshape(-1)
    if got.shape != want.shape:
        return f"shape {got.shape} != {want.shape}"
    bad = np.flatnonzero(got != want)
    if bad.size == 0:
        return "ok"
    i = int(bad[0])
    return f"{bad.size} of {got.size} words differ; first at word {i} (poly {i // N}, coefficient {i % N}): gpu {int(got[i])} != oracle {int(want[i])}"


CLI = os.path.join(ROOT, "optimal_conv_amd", "host", "conv")


def _secs(tok):
    import re
    m = re.match(r"([0-9.e+-]+)(µs|ms|s)$", tok)
    return float(m.group(1)) * {"µs": 1e-6, "ms": 1e-3, "s": 1.0}[m.group(2)]


def _write_conv_csv(d, k, i_batch, iters):
    """synthetic test_conv_data/*.csv in the reference's layout (test.go:37-40, 66-68): uniform inputs, kernel / sqrt(k^2 B), BN a in [0.5, 1.5], b in [-0.5, 0.5]"""
    B, W = BATCHS[i_batch], WIDTHS[i_batch]
    raw = W - k // 2
    os.makedirs(d, exist_ok=True)
    for it in range(iters):
        rng = np.random.default_rng(1000 * k + 10 * i_batch + it)
        parts = {"in": rng.uniform(-1, 1, raw * raw * B), "ker": rng.uniform(-1, 1, k * k * B * B) / np.sqrt(k * k * B), "bna": rng.uniform(0.5, 1.5, B), "bnb": rng.uniform(-0.5, 0.5, B)}
        parts["out"] = parts["reluout"] = np.zeros(raw * raw * B)          # the printed precision is not what this run is for (tests/test_gpu_z_cli.py checks it)
        for name, v in parts.items():
            np.savetxt(os.path.join(d, f"test_conv{k}_batch_{B}_{name}_{it}.csv"), v, fmt="%.17g")


def _write_resnet_csv(root, k, depth, n_images):
    """random weights and images of the shapes `resnet k depth 1 n false` reads (test.go:78-80, 128, 171-183, 285, 329); the reference ships none (README.md:23)"""
    blocks = {20: (7, 5, 5), 14: (5, 3, 3), 8: (3, 1, 1)}[depth]
    ch = (16, 32, 64)
    shapes = [(3, ch[0])] + [(ch[0], ch[0])] * (blocks[0] - 1) + [(ch[0], ch[1])] + [(ch[1], ch[1])] * blocks[1] + [(ch[1], ch[2])] + [(ch[2], ch[2])] * blocks[2]
    tag = f"crop_ker{k}_d{depth}_wid1"
    wdir, pdir = os.path.join(root, "Resnet_weights", "weights_" + tag), os.path.join(root, "Resnet_plain_data", tag)
    os.makedirs(wdir, exist_ok=True); os.makedirs(pdir, exist_ok=True)
    rng = np.random.default_rng(20)
    for i, (ib, ob) in enumerate(shapes):
        np.savetxt(os.path.join(wdir, f"w{i}-conv.csv"), rng.uniform(-1, 1, k * k * ib * ob) * 0.6 / np.sqrt(k * k * ib), fmt="%.17g")
        np.savetxt(os.path.join(wdir, f"w{i}-a.csv"), rng.uniform(0.8, 1.2, ob), fmt="%.17g")
        np.savetxt(os.path.join(wdir, f"w{i}-b.csv"), rng.uniform(-0.1, 0.1, ob), fmt="%.17g")
    np.savetxt(os.path.join(wdir, "final-fckernel.csv"), rng.uniform(-1, 1, ch[2] * 10) / 8, fmt="%.17g")
    np.savetxt(os.path.join(wdir, "final-fcbias.csv"), rng.uniform(-0.1, 0.1, 10), fmt="%.17g")
    for it in range(n_images):
        np.savetxt(os.path.join(pdir, f"test_image_{it}.csv"), np.random.default_rng(1000 + it).uniform(-1, 1, 32 * 32 * 3), fmt="%.17g")


def chain_replay_check(device, relu_batch, work):
    """The convReLU 5 1 workload's launch set once more on PLANTED data: `conv --test-mode convReLU 5 1 1` with HCONV_CHAIN_REPLAY=<seed of tests/golden/ref_trace_chain_5_1.json>
    and the timed HCONV_IMAGE_BATCH. Image 0 carries the input and the switching keys `gotrace -chain` planted into the reference binary; the CLI prints the SHA-256 of
    BootstrappConv_CtoS' two results and of the ciphertext the layer hands on, and they must be the binary's (the fixture is data recorded from /root/reference/test_run;
    nothing of the reference is read here). Returns {"ctos0": .., "ctos1": .., "final": .., "match": bool}."""
    import re
    import subprocess
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_trace_chain_5_1.json")))
    ev = ref["events"]
    ctos = next(e for e in ev if e["fn"] == "BootstrappConv_CtoS")["digests"]
    final = [e for e in ev if e["fn"] == "Rescale" and "digests" in e][-1]["digests"][0]
    env = dict(os.environ, HCONV_SKIP_BL="1", HCONV_DEVICE=str(device), HCONV_SEED="31", HCONV_CHAIN_REPLAY=str(ref["seed"]), HCONV_IMAGE_BATCH=str(relu_batch))
    r = subprocess.run([CLI, "--test-mode", "convReLU", "5", "1", "1"], cwd=work, capture_output=True, text=True, timeout=900, env=env)
    if r.returncode != 0:
        return {"match": False, "error": r.stderr[-400:]}
    pat = r"^replay digest(?:\[(\d+)\])? (\S+) level (\d+) scale (\S+) ((?:[0-9a-f]{64} ?)+)$"
    got = {m.group(2): (int(m.group(3)), float(m.group(4)), m.group(5).split()) for m in re.finditer(pat, r.stdout, re.M) if int(m.group(1) or 0) == 0}
    out = {"fixture": "tests/golden/ref_trace_chain_5_1.json (SHA-256 of the reference binary's ciphertexts on the planted input and keys)",
           "command": f"HCONV_IMAGE_BATCH={relu_batch} HCONV_CHAIN_REPLAY={ref['seed']} conv --test-mode convReLU 5 1 1 (image 0 of the launch set)", "match": True}
    for name, want in (("ctos0", ctos[0]), ("ctos1", ctos[1]), ("final", final)):
        g = got.get(name)
        ok = g is not None and (g[0], g[1]) == (want["level"], want["scale"]) and g[2] == want["polys"]
        out[name] = {"level": g[0] if g else None, "sha256": g[2] if g else None, "equals_reference_binary": bool(ok)}
        out["match"] = out["match"] and bool(ok)
    return out


def chain_workloads(device, relu_batch, resnet_batch, resnet_images, relu=True):
    """BASELINE configs 4 and 5 through the PRODUCT path - the C++ host CLI (optimal_conv_amd/host/conv) over libhconv.so - on synthetic CSVs, in the same run as the
    headline: `convReLU 5 1` with HCONV_IMAGE_BATCH ciphertexts per launch set (per ciphertext-layer: convolution + BootstrappConv_CtoS + evalReLU + mask + SlotsToCoeffs)
    and `resnet 3 20 1 n false` (images/hour). Timers are the CLI's own (the reference's: eval.go:463, 479, 565; test.go:361-367). Keys are fresh random ones."""
    import re
    import subprocess
    import tempfile
    if not os.path.exists(CLI):
        import __graft_entry__
        __graft_entry__.build()
    out = {}
    env = dict(os.environ, HCONV_SKIP_BL="1", HCONV_DEVICE=str(device), HCONV_ALG_BYTES="1")
    env.pop("HCONV_SEED", None)
    with tempfile.TemporaryDirectory(prefix="hconv_bench_") as work:
        iters = 3
        _write_conv_csv(os.path.join(work, "test_conv_data"), 5, 1, iters)
        r = None if not relu else subprocess.run([CLI, "convReLU", "5", "1", str(iters)], cwd=work, capture_output=True, text=True, timeout=900, env=dict(env, HCONV_IMAGE_BATCH=str(relu_batch)))
        if r is None:
            pass
        elif r.returncode == 0:
            txt = r.stdout
            conv = [_secs(t) for t in re.findall(r"^Conv \(with BN\) Done in (\S+) ", txt, re.M)]
            ctos = [_secs(t) for t in re.findall(r"^Done in (\S+) $", txt, re.M) if not t.endswith("keys")]
            relu = [_secs(t) for t in re.findall(r"ReLU Done in (\S+) ", txt, re.M)]
            stoc = [_secs(t) for t in re.findall(r"^Boot \(StoC\) Done in (\S+) ", txt, re.M)]
            alg = re.findall(r"^algorithmic traffic of the layer's tail: (\S+) GB per ciphertext \+ (\S+) GB shared", txt, re.M)
            keys = re.search(r"^Generating bootstrapping keys\.\.\.\nDone in (\S+) ", txt, re.M)
            if len(conv) == iters and len(relu) == iters and len(stoc) == iters and len(ctos) >= iters and alg:
                ctos = ctos[-iters:]
                layer = [conv[i] + ctos[i] + relu[i] + stoc[i] for i in range(1, iters)]          # the first iteration allocates the pools: warm-up
                ms = 1e3 * float(np.mean(layer)) / relu_batch
                ab = (float(alg[-1][0]) + float(alg[-1][1]) / relu_batch) * 1e9 + algorithmic_mib(16) * 2 ** 20
                out["convReLU_5_1"] = {"ms_per_ct_layer_throughput": ms, "ms_per_layer_per_ct": ms, "ciphertexts_per_launch_set": relu_batch, "layer_ms": 1e3 * float(np.mean(layer)),
                                       "layer_latency_ms": 1e3 * float(np.mean(layer)),
                                       "figures_note": "ms_per_ct_layer_throughput = layer_latency_ms / ciphertexts_per_launch_set: a THROUGHPUT figure (the images of a batch share every launch); the latency a caller waits for a layer is layer_latency_ms, and ms_per_layer_n1 is one image alone (the reference's own flow); prep_Ker is outside the timers as in the reference (eval.go:244)",
                                       "stages_ms_per_launch_set": {"conv": 1e3 * float(np.mean(conv[1:])), "ctos_sine": 1e3 * float(np.mean(ctos[1:])), "relu": 1e3 * float(np.mean(relu[1:])), "mask_stoc": 1e3 * float(np.mean(stoc[1:]))},
                                       "algorithmic_bytes": ab, "algorithmic_bytes_note": "per ciphertext-layer, counted PER EVALUATOR OPERATION: every operation reads its ciphertext operands and writes its result once (SURVEY 8d's convention), switching keys / diagonals / masks once per launch set (shared by the images), + the convolution's. A hybrid key switch cannot keep its beta x (level+1+alpha) digit rows on chip, so this convention is generous: frac is an UPPER reading of the fraction",
                                       "frac": ab / (ms * 1e-3) / 8e12, "bootstrapping_keys_s": _secs(keys.group(1)) if keys else None,
                                       "command": f"HCONV_IMAGE_BATCH={relu_batch} HCONV_SKIP_BL=1 conv convReLU 5 1 {iters}"}
                tr = traffic_chain_from_profiles()
                if tr:
                    out["convReLU_5_1"].update(tr)
                cv = chain_valu_from_profiles()
                if cv:          # both roofs: issue floor / measured time beside the HBM fraction above
                    out["convReLU_5_1"].update(cv)
                    out["convReLU_5_1"]["valu_frac"] = cv["issue_floor_ms"] / ms
                if relu_batch != 1:          # one image alone: the latency of the reference's own per-image flow (eval.go:446-565's timers)
                    r1 = subprocess.run([CLI, "convReLU", "5", "1", str(iters)], cwd=work, capture_output=True, text=True, timeout=900, env=dict(env, HCONV_IMAGE_BATCH="1"))
                    if r1.returncode == 0:
                        t1 = r1.stdout
                        c1 = [_secs(t) for t in re.findall(r"^Conv \(with BN\) Done in (\S+) ", t1, re.M)]; b1 = [_secs(t) for t in re.findall(r"^Done in (\S+) $", t1, re.M) if not t.endswith("keys")][-iters:]
                        l1 = [_secs(t) for t in re.findall(r"ReLU Done in (\S+) ", t1, re.M)]; s1 = [_secs(t) for t in re.findall(r"^Boot \(StoC\) Done in (\S+) ", t1, re.M)]
                        if len(c1) == iters and len(b1) == iters and len(l1) == iters and len(s1) == iters:
                            out["convReLU_5_1"]["ms_per_layer_n1"] = 1e3 * float(np.mean([c1[i] + b1[i] + l1[i] + s1[i] for i in range(1, iters)]))
            else:
                out["convReLU_5_1"] = {"error": "could not parse the CLI output", "stdout_tail": txt[-400:]}
        else:
            out["convReLU_5_1"] = {"error": r.stderr[-400:]}
        if relu:        # the timed launch set (same batch, same options) on the reference's planted data: digests against the binary's
            try:
                out["convReLU_5_1_replay"] = chain_replay_check(device, relu_batch, work)
            except Exception as e:
                out["convReLU_5_1_replay"] = {"match": False, "error": repr(e)}
        _write_resnet_csv(work, 3, 20, resnet_images)
        r = subprocess.run([CLI, "resnet", "3", "20", "1", str(resnet_images), "false"], cwd=work, capture_output=True, text=True, timeout=1500, env=dict(env, HCONV_IMAGE_BATCH=str(resnet_batch)))
        if r.returncode == 0:
            tot = [(_secs(m.group(1)), int(m.group(2) or 1)) for m in re.finditer(r"^Total done in (\S+) (?:\((\d+) images\))?$", r.stdout, re.M)]
            if len(tot) >= 2:
                steady = tot[1:]          # the first group of images also generates the stride layers' keys on demand and allocates the pools
                sec = sum(t for t, _ in steady); nim = sum(n for _, n in steady)
                out["resnet20"] = {"images_per_hour": 3600.0 * nim / sec, "seconds_per_image": sec / nim, "images_per_launch_set": resnet_batch, "images_timed": nim,
                                   "first_group_seconds": tot[0][0], "command": f"HCONV_IMAGE_BATCH={resnet_batch} conv resnet 3 20 1 {resnet_images} false", "data": "random weights and images of the reference's shapes"}
            else:
                out["resnet20"] = {"error": "fewer than two image groups", "stdout_tail": r.stdout[-400:]}
        else:
            out["resnet20"] = {"error": r.stderr[-400:]}
    return out


def traffic_chain_from_profiles():
    """fabric bytes per convReLU ciphertext-layer from the committed rocprofv3 PMC passes (tools/gpu_relu_traffic.sh), with their provenance"""
    for name in ("round6_traffic_convrelu_5_1.json", "round5_traffic_convrelu_5_1.json", "round4_traffic_convrelu_5_1.json"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
            return {"traffic": d["bytes_per_ciphertext_layer"], "traffic_measured_with": {"file": "profiles/" + name, **d.get("measured_with", {})}}
        except Exception:
            continue
    return None


def chain_valu_from_profiles():
    """the OTHER roof of the convReLU chain: lane-instructions and the VALU issue floor per ciphertext-layer from the committed per-kernel counter pass
    (tools/gpu_r6_chain_counters.sh -> tools/chain_table.py -> profiles/round6_chain_valu.json; the layers' kernels only, set-up excluded)"""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "round6_chain_valu.json")))
        return {"issue_floor_ms": d["issue_floor_ms"], "lane_instr_per_ct_layer": d["lane_instr_per_ct_layer"], "kernel_ms_per_ct_layer_under_rocprof": d["kernel_ms_per_ct_layer"],
                "valu_measured_with": {"file": "profiles/round6_chain_valu.json", "table": "profiles/round6_chain_counters.txt", "images_per_launch_set": d["images_per_launch_set"], "commit": d.get("commit"), "method": d["method"]}}
    except Exception:
        return None


def sharded_conv_timing(rank, world, device, backend):
    """ONE convolution `conv 7 3` (k = 7 only changes prep_Ker: the homomorphic work is that of B = 256) over the N GPUs of the node, both forms of SURVEY 8(e):
    (a) one process per GPU, partial trees gathered over RCCL (optimal_conv_amd/sharded.py) - all ranks; (b) hc_conv_then_pack_sharded driven by rank 0 with one context
    per device and peer copies (xGMI). Median of 5 after one warm-up each; rank 0 returns the figures."""
    import torch
    import torch.distributed as dist
    from optimal_conv_amd import Context
    from optimal_conv_amd.sharded import conv_then_pack_sharded, local_channels
    B = 256
    if B % world or world & (world - 1):
        return {"skipped": f"world size {world} does not divide {B} as a power of two"}
    rng = np.random.default_rng(0x7C0FFEE)           # the same data on every rank
    ct_in = np.stack([np.stack([synth_rows(rng, Q0, N), synth_rows(rng, Q1, N)]) for _ in range(2)])
    pl_ker = np.empty((B, 2, N), dtype=np.uint64)
    pl_ker[:, 0] = synth_rows(rng, Q0, (B, N)); pl_ker[:, 1] = synth_rows(rng, Q1, (B, N))
    keys, step, j = [], B // 2, 16 - ((B // 2).bit_length() - 1)
    while step >= 1:
        keys.append(((1 << j) + 1, [synth_rows(rng, Q0, N), synth_rows(rng, Q0, N), synth_rows(rng, P0, N), synth_rows(rng, P0, N)]))
        step //= 2; j += 1
    bias = synth_rows(rng, Q0, N)

    def make(dev):
        c = Context([Q0, Q1], [P0], device=dev)
        for gal, k4 in keys:
            c.evk_load(gal, k4)
        c.idx_load(None)
        return c
    out = {"workload": "conv 7 3 (one convolution, B = 256, channels i mod N)", "n_gpus": world}
    ctx = make(device)
    kh = ctx.ker_load(pl_ker[local_channels(B, rank, world)])
    cin, bb = ctx.buf(ct_in), ctx.buf(bias)
    dev_s = f"cuda:{device}" if backend == "nccl" else "cpu"
    times = []
    for it in range(6):
        dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        res, _ = conv_then_pack_sharded(ctx, cin, 2.0 ** 30, kh, 2.0 ** 30, B, 2.0 ** 30, bb, device=f"cuda:{device}")
        ctx.sync()
        dist.barrier()
        times.append((time.perf_counter() - t0) * 1e3)
    out["sharded_conv_ms_rccl_gather"] = float(np.median(times[1:]))
    want = None
    if rank == 0:           # the first record from a multi-GPU box is also a correctness record: the last timed sharded convolution against the CPU oracle, all 2N words
        want, _ = oracle_conv(B, ct_in, pl_ker, keys, bias)
        out["parity_sharded_conv_rccl_gather"] = first_difference(res.cpu().numpy().view(np.uint64).reshape(2, N), want)
    ctx.ker_free(kh); ctx.close()
    dist.barrier()
    if rank == 0:
        try:
            ndev = torch.cuda.device_count()
            ctxs = [make(d % ndev) for d in range(world)]          # fewer devices than ranks (the one-GPU dry run): the contexts share devices
            khs = [c.ker_load(pl_ker) for c in ctxs]
            ins = [c.buf(ct_in) for c in ctxs]
            b0, o0 = ctxs[0].buf(bias), ctxs[0].buf(nwords=2 * N)
            ts = []
            for it in range(6):
                for c in ctxs:
                    c.sync()
                t0 = time.perf_counter()
                Context.conv_then_pack_sharded_dev(ctxs, ins, 2.0 ** 30, khs, 2.0 ** 30, B, 2.0 ** 30, b0, o0)
                ctxs[0].sync()
                ts.append((time.perf_counter() - t0) * 1e3)
            out["sharded_conv_ms"] = float(np.median(ts[1:]))
            out["parity_sharded_conv"] = first_difference(o0.download((2, N)), want)
            out["devices_used"] = min(world, ndev)
            out["peer_access"] = "hipMemcpyPeerAsync between the devices' contexts; direct peer access enabled where hipDeviceCanAccessPeer allows (a failure to enable falls back to staged copies, reported on stderr)"
            for c, k in zip(ctxs, khs):
                c.ker_free(k); c.close()
        except Exception as e:
            out["sharded_conv_ms"] = None; out["sharded_conv_error"] = repr(e); out["parity_sharded_conv"] = "not run: " + repr(e)
    dist.barrier()
    _ = dev_s
    return out if rank == 0 else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--i-batch", type=int, default=3, help="reference batch index (main.go:578): 3 => B=256, W=16 = `conv 3 3`")
    ap.add_argument("--ker-wid", type=int, default=3)
    ap.add_argument("--chunk", type=int, default=512, help="jobs (channels / tree nodes, summed over the batch) per kernel launch")
    ap.add_argument("--batch", type=int, default=4, help="ciphertexts per hc_conv_then_pack_batch call (one launch set covers them all); 4 x 4 contexts measured best (8 x 2: -4.5 %, 8 x 3: -1 %)")
    ap.add_argument("--streams", type=int, default=4, help="contexts (HIP streams) per GPU, each with its own batch of resident ciphertexts")
    ap.add_argument("--batch-alt", type=int, default=0, help="experiment: odd-numbered contexts use this batch size instead (desynchronises the streams)")
    ap.add_argument("--opt", action="append", default=[], help="extra context option name=value (hc_set_option), e.g. small_levels=0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-workloads", action="store_true", help="skip the convReLU 5 1 / resnet-20 figures (BASELINE configs 4, 5: the C++ host CLI, ~40 s)")
    ap.add_argument("--relu-batch", type=int, default=4, help="HCONV_IMAGE_BATCH of the convReLU 5 1 workload")
    ap.add_argument("--resnet-batch", type=int, default=8, help="HCONV_IMAGE_BATCH of the resnet-20 workload")
    ap.add_argument("--resnet-images", type=int, default=24)
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    # HC_BENCH_BACKEND=gloo + fewer devices than ranks is the single-GPU dry run of the N>1 path (ranks share GPU 0);
    # the real runs are one rank per GPU over RCCL.
    backend = os.environ.get("HC_BENCH_BACKEND", "nccl" if ndev else "gloo")
    device = local_rank if (ndev == 0 or local_rank < ndev or backend == "nccl") else local_rank % ndev
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(device)
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                device_id=torch.device("cuda", device) if backend == "nccl" else None)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from optimal_conv_amd import Context
    B, W = BATCHS[args.i_batch], WIDTHS[args.i_batch]
    S = max(1, args.streams)
    NB = max(1, min(16, args.batch))
    rng = np.random.default_rng(0xC0FFEE + args.i_batch + 1000 * rank)
    ct_in = np.stack([np.stack([synth_rows(rng, Q0, N), synth_rows(rng, Q1, N)]) for _ in range(2)])
    pl_ker = np.empty((B, 2, N), dtype=np.uint64)
    pl_ker[:, 0] = synth_rows(rng, Q0, (B, N)); pl_ker[:, 1] = synth_rows(rng, Q1, (B, N))
    keys = []
    step = B // 2
    j = 16 - (step.bit_length() - 1)
    while step >= 1:
        keys.append(((1 << j) + 1, [synth_rows(rng, Q0, N), synth_rows(rng, Q0, N), synth_rows(rng, P0, N), synth_rows(rng, P0, N)]))
        step //= 2; j += 1
    bias = synth_rows(rng, Q0, N)
    lanes = []                      # one lane = one context/stream with its own resident ciphertext, keys and kernel plaintexts
    for s_ in range(S):
        ctx = Context([Q0, Q1], [P0], device=device)            # raises if no GPU / no libhconv.so
        ctx.set_option("chunk_nodes", args.chunk)
        for kv in args.opt:
            ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        for gal, k4 in keys:
            ctx.evk_load(gal, k4)
        ctx.idx_load(None)
        # every resident ciphertext has its own input, its own kernel plaintexts (distinct HBM copies, channel-rolled so that no two
        # are equal) and its own output; the BN bias plaintext is the layer's
        L = {"ctx": ctx, "bias": ctx.buf(bias), "ker": [], "in": [], "out": []}
        for z in range(args.batch_alt if (args.batch_alt and s_ % 2) else NB):
            L["ker"].append(ctx.ker_load(np.roll(pl_ker, s_ * NB + z, axis=0)))
            L["in"].append(ctx.buf(np.roll(ct_in, 17 * (s_ * NB + z), axis=2)))          # ciphertext 0 of context 0 = the planted data itself: the oracle's input below
            L["out"].append(ctx.buf(nwords=2 * N))
        lanes.append(L)
    ctx = lanes[0]["ctx"]
    counter = [0]

    def one_step():
        """one pass of the hot path over one batch of input = one convolution on each of the S resident ciphertexts"""
        for L in lanes:
            L["ctx"].conv_then_pack_batch_dev(L["in"], 2.0 ** 30, L["ker"], 2.0 ** 30, B, 1, 2.0 ** 30, [L["bias"]] * len(L["in"]), L["out"])

    def sync_all():
        for L in lanes:
            L["ctx"].sync()

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        sync_all()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    # set-up, not warm-up: every lane's first conv allocates its context's workspaces (hipMalloc); do that outside both the
    # warm-up count and the timed region so that a run with a small --warmup does not time allocations on the cold lanes
    one_step()
    sync_all()
    for _ in range(args.warmup):
        one_step()
    sync_all()
    for L in lanes:         # every output is cleared before the timed region: what is compared with the oracle below was written by the timed steps
        for o in L["out"]:
            o.upload(np.zeros(2 * N, dtype=np.uint64))
    barrier()
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(args.steps):
        one_step()
    sync_all()
    ev_ms = ctx.timer_stop()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{device}" if backend == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    timed_out0 = lanes[0]["out"][0].download((2, N))         # ciphertext 0 of context 0 after the LAST timed step (the region above is closed)
    timed_out_last = lanes[-1]["out"][-1].download((2, N))    # ... and the last ciphertext of the last context (a rolled copy of the same data: must differ from it, and is non-zero)

    # per-kernel HIP-event profile (separate untimed pass: event records between launches perturb the stream)
    counter[0] = 0
    ctx.set_option("profile", 1)
    ctx.profile_reset()
    nprof = max(1, min(3, args.steps))
    L0 = lanes[0]
    for _ in range(nprof):
        ctx.conv_then_pack_batch_dev(L0["in"], 2.0 ** 30, L0["ker"], 2.0 ** 30, B, 1, 2.0 ** 30, [L0["bias"]] * NB, L0["out"])
    prof = ctx.profile()
    ctx.set_option("profile", 0)
    kern = {k: {"ms_per_conv": v[0] / (nprof * NB), "launches_per_batch": v[1] // nprof, "avg_launch_us": 1e3 * v[0] / max(1, v[1])} for k, v in prof.items()}
    dom = max(kern, key=lambda k: kern[k]["ms_per_conv"]) if kern else None

    # ONE convolution alone on an otherwise idle GPU (what the `conv` CLI does): latency, not throughput
    L0["ctx"].sync()
    lone = []
    for _ in range(10):       # HIP events around each lone convolution, the stream drained in between: a latency, nothing pipelines
        ctx.timer_start()
        ctx.conv_then_pack_batch_dev(L0["in"][:1], 2.0 ** 30, L0["ker"][:1], 2.0 ** 30, B, 1, 2.0 ** 30, [L0["bias"]], L0["out"][:1])
        lone.append(ctx.timer_stop())
    single_ms = float(np.median(lone))

    if rank == 0:
        per_step = sum(len(L["in"]) for L in lanes)
        conv_ms_events = ev_ms / (args.steps * per_step)
        alg_bytes = algorithmic_mib(B) * 2 ** 20
        achieved = alg_bytes / (conv_ms_events * 1e-3) / 1e9
        traffic, traffic_how = traffic_from_profiles(B)
        valu = valu_from_profiles() if B == 256 else None
        # `bound` names the roofline `achieved` / `peak` / `frac` are quoted against (HBM: the metric's roofline). The other roof - VALU issue, from measured
        # instruction counts priced per class - is carried beside it as `valu_frac` = issue floor / achieved time per conv. Neither binds alone (issue at ~0.65, fabric at ~0.37
        # of their ceilings): the kernels are latency / overlap bound, `state` says so.
        bound = "hbm"
        valu_frac = (valu["issue_floor_ms"] / conv_ms_events) if valu else None
        mib = 2 ** 20
        loops = {}        # per-loop share of the roofline from the one-stream kernel profile above (SURVEY.md 8d splits the algorithmic bytes the same way)
        if kern:
            la = sum(v["ms_per_conv"] for k, v in kern.items() if k[0] == "a" or k.startswith("ctc"))
            lb = sum(v["ms_per_conv"] for k, v in kern.items() if k[0] == "b")
            for name, ms, alg in (("loop_A", la, (2 * B + 2) * mib), ("loop_B", lb, (3 * (B - 1) + 2.5 * np.log2(B) + 0.5) * mib)):
                if ms > 0:
                    loops[name] = {"ms_per_conv_one_stream": ms, "algorithmic_bytes": alg, "frac": alg / (ms * 1e-3) / 1e9 / 8000.0}
        out = {
            "metric": "homomorphic convs/sec (conv_then_pack + BN bias, k x k, batch B, N=2^16)",
            "value": world * args.steps * per_step / elapsed, "unit": "conv/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"conv {args.ker_wid} {args.i_batch}", "ker_wid": args.ker_wid, "batch": B, "in_wid": W,
                       "logN": 16, "moduli": "ckks.DefaultBootstrapParams[6] Q0,Q1 + P=0x1fffffffffe00001",
                       "convs_per_step_per_gpu": per_step, "ciphertexts_per_launch_set": NB, "contexts_per_gpu": S, "chunk_nodes": args.chunk},
            "roofline": {"bound": bound, "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "valu_frac": valu_frac,
                         "state": "neither roof binds alone: VALU issue floor / achieved = valu_frac, fabric traffic / time well below the ~6.3 TB/s achievable - the kernels are latency / overlap bound (DESIGN.md 5)",
                         "traffic": traffic, "traffic_measured_with": traffic_how,
                         "unit_of_launch": "one conv_then_pack (its share of the batched launch set: all kernels of loop A and of the pack tree)",
                         "algorithmic_bytes_per_conv": alg_bytes, "conv_ms_hip_events": conv_ms_events,
                         "dominant_kernel": dom, "kernels": kern,
                         "kernels_measured_with": {"contexts": 1, "ciphertexts_per_launch_set": NB, "note": "HIP events around every launch of one context, separate untimed pass"},
                         "loops": loops, "single_conv_ms": single_ms, "valu": valu},
        }
        # The record proves its own work: the oracle (2 s, one core) runs on the SAME planted inputs as ciphertext 0 of context 0 and its output is compared word for word with
        # what the timed region left there (conv.go:522-546 + eval.go:258). A mismatch fails the process after the line is printed.
        if not args.no_cpu_baseline:
            want, dt = oracle_conv(B, ct_in, pl_ker, keys, bias)
            verdict = first_difference(timed_out0, want)
            if verdict == "ok" and per_step > 1 and (not timed_out_last.any() or np.array_equal(timed_out_last, timed_out0)):
                verdict = "the last resident ciphertext's output is zero or a copy of ciphertext 0's: the timed steps did not write it"
            out["parity_check"] = {f"conv_{args.ker_wid}_{args.i_batch}": verdict,
                                   "what": f"GPU output of ciphertext 0 / context 0 after the last of the {args.steps} timed steps (outputs cleared before the timed region; B={B}, n={NB}, chunk={args.chunk}, {S} contexts) == the CPU oracle on the same planted inputs, all {2 * N} words"}
            if world == 1:
                out["cpu_baseline"] = cpu_baseline(B, dt)
        else:
            out["parity_check"] = {f"conv_{args.ker_wid}_{args.i_batch}": "skipped (--no-cpu-baseline)"}
    for L in lanes:
        for k in L["ker"]:
            L["ctx"].ker_free(k)
        L["ctx"].close()
    # The headline is measured. What follows are extras (the sharded conv 7 3 under N > 1, the chain workloads through the CLI): they must never cost the line. The N > 1 extras
    # have only ever run as ranks sharing one GPU (no multi-GPU box in this pool), so a watchdog prints the headline alone if they do not come back (a collective that hangs, a
    # peer copy that stalls) and ends the process; every rank carries it so that the launcher is not left waiting for a stuck rank.
    import threading
    emitted = threading.Lock()
    def _give_up():
        if not emitted.acquire(blocking=False):
            return
        if rank == 0:
            out["extras"] = "timed out after %d s: headline only" % extras_timeout
            print(json.dumps(out), flush=True)
        os._exit(0)
    extras_timeout = int(os.environ.get("HC_BENCH_EXTRAS_TIMEOUT", "420" if world > 1 else "900"))
    watchdog = threading.Timer(extras_timeout, _give_up)
    watchdog.daemon = True
    watchdog.start()
    sharded = None
    if world > 1:           # BASELINE config 3 beside the weak-scaling figure: ONE `conv 7 3` (B = 256) split i mod N over the N devices
        try:
            sharded = sharded_conv_timing(rank, world, device, backend)
        except Exception as e:
            sharded = {"error": repr(e)}
    wl = None
    if not args.no_workloads:       # after the headline's contexts are gone: the CLI builds its own (device memory is free again). N > 1: every rank classifies its own images
        try:
            wl = chain_workloads(device, args.relu_batch, args.resnet_batch, args.resnet_images, relu=(rank == 0))
        except Exception as e:      # the headline line must not depend on the secondary workloads
            wl = {"error": repr(e)}
        if world > 1:               # every rank takes part whatever happened to its own run (inf = this rank has no figure)
            spi = wl.get("resnet20", {}).get("seconds_per_image", float("inf")) if isinstance(wl, dict) else float("inf")
            t = torch.tensor([spi], dtype=torch.float64, device=f"cuda:{device}" if backend == "nccl" else "cpu")
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            if rank == 0 and "resnet20" in wl and np.isfinite(float(t.item())):
                wl["resnet20"].update({"images_per_hour": world * 3600.0 / float(t.item()), "n_gpus": world, "note": "images sharded over the ranks (independent ciphertexts, no collective); slowest rank's time per image"})
            elif rank == 0 and "resnet20" in wl:
                wl["resnet20"]["note"] = "another rank produced no figure: this is rank 0's own rate"
    if not emitted.acquire(blocking=False):       # the watchdog is printing the headline: let it end the process
        time.sleep(60)
    watchdog.cancel()
    failed = False
    if rank == 0:
        if wl is not None:
            out["workloads"] = wl
            rp = wl.pop("convReLU_5_1_replay", None) if isinstance(wl, dict) else None
            if rp is not None:
                out["parity_check"]["convReLU_5_1"] = "ok" if rp.get("match") else "MISMATCH"
                out["parity_check"]["convReLU_5_1_digests"] = rp
        if sharded is not None:
            out["sharded_conv"] = sharded
            for k_ in ("parity_sharded_conv", "parity_sharded_conv_rccl_gather"):
                if isinstance(sharded, dict) and k_ in sharded:
                    out["parity_check"][k_[len("parity_"):]] = sharded.pop(k_)
        print(json.dumps(out), flush=True)
        failed = any(isinstance(v, str) and v != "ok" and not v.startswith(("skipped", "not run")) for k_, v in out["parity_check"].items() if k_ != "what")
        if failed:
            print("bench.py: PARITY CHECK FAILED: " + json.dumps({k: v for k, v in out["parity_check"].items() if isinstance(v, str) and k != "what"}), file=sys.stderr, flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()
    if failed:
        sys.exit(1)


if __name__ == "__main__":
    main()
