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
    for name in (f"round3_traffic_conv_B{B}.json", f"traffic_conv_B{B}.json"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
            return d["bytes_per_conv"], {"file": "profiles/" + name, **d.get("measured_with", {"note": d.get("method", "")})}
        except Exception:
            continue
    return None, None


def synth_rows(rng, q, shape):
    return (rng.integers(0, 1 << 62, size=shape, dtype=np.uint64) % np.uint64(q)).astype(np.uint64)


def cpu_baseline(B):
    """Time the CPU oracle (test infrastructure, used here only as the baseline being reported, never as the
    product) on one full conv_then_pack at the bench workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle
    O = Oracle()
    rng = np.random.default_rng(0xC0FFEE)
    ct_in = np.stack([np.stack([synth_rows(rng, Q0, N), synth_rows(rng, Q1, N)]) for _ in range(2)])
    ker = np.empty((B, 2, N), dtype=np.uint64)
    ker[:, 0] = synth_rows(rng, Q0, (B, N)); ker[:, 1] = synth_rows(rng, Q1, (B, N))
    evk = np.zeros((16, 4, N), dtype=np.uint64)
    evk[:, 0] = synth_rows(rng, Q0, (16, N)); evk[:, 1] = synth_rows(rng, Q0, (16, N))
    evk[:, 2] = synth_rows(rng, P0, (16, N)); evk[:, 3] = synth_rows(rng, P0, (16, N))
    idx = O.idx_plaintexts()
    bias = synth_rows(rng, Q0, N)
    t0 = time.perf_counter()
    O.conv_then_pack(ct_in, 2.0 ** 30, ker, 2.0 ** 30, idx, evk, B, 1, 2.0 ** 30, bias)
    dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "conv/s", "cores": 1, "kind": "port",
            "sample": f"1 full conv_then_pack+bias at B={B} (N=2^16) on the C oracle, {dt:.2f} s, host has {os.cpu_count()} cores"}


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
            L["in"].append(ctx.buf(np.roll(ct_in, 17 * (s_ * NB + z) + 1, axis=2)))
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
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": traffic, "traffic_measured_with": traffic_how,
                         "unit_of_launch": "one conv_then_pack (its share of the batched launch set: all kernels of loop A and of the pack tree)",
                         "algorithmic_bytes_per_conv": alg_bytes, "conv_ms_hip_events": conv_ms_events,
                         "dominant_kernel": dom, "kernels": kern,
                         "kernels_measured_with": {"contexts": 1, "ciphertexts_per_launch_set": NB, "note": "HIP events around every launch of one context, separate untimed pass"},
                         "loops": loops, "single_conv_ms": single_ms},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(B)
        print(json.dumps(out), flush=True)
    for L in lanes:
        for k in L["ker"]:
            L["ctx"].ker_free(k)
        L["ctx"].close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
