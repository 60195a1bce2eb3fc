"""The i-mod-G sharded convolution (optimal_conv_amd/sharded.py, BASELINE config `conv 7 3` over 8 GPUs) on a real GPU with the
RCCL backend. The pool's boxes have one GPU, so the world size is 1 here (the exchange degenerates to a gather of one 1 MiB
ciphertext); what this covers on the device is everything else of that path: torch CUDA storage handed to libhconv,
hc_conv_mult_phase + hc_pack_ctxts_strided, the gather on CUDA tensors, bias on the last level. World sizes 2 and 4 run on CPU
(tests/test_sharded_gloo.py)."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r); sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    import parity_cases as pc
    from oracle_lib import Oracle, Q0, Q1, P0, splitmix_rows
    from optimal_conv_amd import Context
    from optimal_conv_amd.sharded import conv_then_pack_sharded, local_channels
    dev = int(os.environ["LOCAL_RANK"]) if os.environ.get("HC_TEST_DEVICE_PER_RANK") else 0      # one GPU per rank on a multi-GPU node
    torch.cuda.set_device(dev)
    backend = os.environ.get("HC_TEST_BACKEND", "nccl")
    dist.init_process_group(backend, rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]), device_id=torch.device("cuda", dev) if backend == "nccl" else None)
    rank, world = dist.get_rank(), dist.get_world_size()
    B, seed = int(os.environ.get("HC_TEST_B", "16")), 0xABCD
    ctx = Context([Q0, Q1], [P0], device=dev)
    ct_in, ker = pc.planted_conv_inputs(seed, B)
    evk_all = pc.load_tree_keys(ctx, seed, B)
    ctx.idx_load(None)
    bias = splitmix_rows(seed + 5, Q0, pc.N)
    kh = ctx.ker_load(ker[local_channels(B, rank, world)])
    res, sc = conv_then_pack_sharded(ctx, ctx.buf(ct_in), 2.0 ** 30, kh, 2.0 ** 30, B, 2.0 ** 30, ctx.buf(bias), device="cuda:%%d" %% dev)
    if rank == 0:
        O = Oracle()
        want, wsc = O.conv_then_pack(ct_in, 2.0 ** 30, ker, 2.0 ** 30, O.idx_plaintexts(), evk_all, B, 1, 2.0 ** 30, bias)
        pc.eq(res.cpu().numpy().view(np.uint64).reshape(2, pc.N), want, "sharded conv_then_pack, world %%d, B %%d" %% (world, B))
        assert sc == wsc
        print("SHARDED_OK")
    dist.barrier()
    dist.destroy_process_group()
""") % (ROOT, HERE)


def _run(tmp_path, world, extra_env):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                          "--master-port", str(29700 + os.getpid() % 200), str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "SHARDED_OK" in out.stdout


def test_sharded_conv_rccl_world1(tmp_path):
    _run(tmp_path, 1, {})


def test_sharded_conv_config3_shape_8_ranks_on_one_gpu(tmp_path):
    """BASELINE config 3's exact shape -- ONE convolution with B = 256 output channels sharded i mod 8 over 8 ranks, one gather of
    8 x 1 MiB, the last 3 tree levels on rank 0 -- executed once on the one GPU this pool has: the 8 ranks share device 0 and talk
    over gloo (the partials are staged through the host; on an 8-GPU node the same code runs over RCCL). Bit-exact vs the oracle."""
    _run(tmp_path, 8, {"HC_TEST_BACKEND": "gloo", "HC_TEST_B": "256"})


# ---- two or more physical GPUs (skipped on the one-GPU boxes of this pool; the driver's multi-GPU node runs them) ----
def _ngpu():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


needs2 = pytest.mark.skipif(_ngpu() < 2, reason="needs two MI355X in the box")


@needs2
def test_sharded_conv_rccl_two_gpus(tmp_path):
    """conv.go:286-297 over two PHYSICAL devices: ranks 0 and 1 on GPUs 0 and 1, the gather of the two partials over RCCL / xGMI"""
    _run(tmp_path, 2, {"HC_TEST_DEVICE_PER_RANK": "1", "HC_TEST_B": "64"})


@needs2
@pytest.mark.parametrize("peer", [1, 0])
def test_conv_sharded_over_two_devices_c_abi(peer):
    """hc_conv_then_pack_sharded with its contexts on DISTINCT devices: cross-device events, hipMemcpyPeerAsync of the partials with
    (peer = 1: hipDeviceEnablePeerAccess must succeed where hipDeviceCanAccessPeer says so) and without direct peer access, == the oracle"""
    import parity_cases as pc
    from oracle_lib import Oracle, P0, Q0, Q1
    from optimal_conv_amd import Context
    G = 2 if _ngpu() < 4 else 4
    devs = iter(range(G))

    def make():
        c = Context([Q0, Q1], [P0], device=next(devs))
        c.set_option("peer_access", peer)
        return c
    pc.case_conv_sharded_abi(make, Oracle(), 64, G)


@needs2
def test_bench_two_gpus_over_rccl(tmp_path):
    """the driver's N = 2 launch line: one rank per GPU over RCCL, ONE JSON line from rank 0 with n_gpus = 2 and about twice the N = 1 rate"""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = ["--steps", "3", "--warmup", "1", "--batch", "2", "--streams", "2", "--no-cpu-baseline"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args, env=env, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(29900 + os.getpid() % 90),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2"] + args, env=env, capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-3000:]
    j1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    j2 = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][-1])
    assert j2["n_gpus"] == 2 and j2["scaling"] == "weak" and j2["unit"] == "conv/s"
    assert 1.5 * j1["value"] < j2["value"] < 2.5 * j1["value"], (j1["value"], j2["value"])


def test_bench_two_gloo_ranks_on_one_gpu_carries_configs_3_and_5(tmp_path):
    """The N > 1 line of bench.py, as far as a one-GPU box can exercise it (no multi-GPU box exists in this pool: a dry run of the plumbing, not a measurement): two gloo
    ranks share GPU 0. The ONE JSON line must carry the weak-scaling headline with n_gpus = 2 AND both multi-GPU configurations of BASELINE.json: config 3 - one `conv 7 3`
    split i mod N over the ranks' contexts, both forms (`sharded_conv`) - and config 5 - every rank classifying its own ResNet-20 images, the whole-job rate from the
    slowest rank (`workloads.resnet20.images_per_hour`, n_gpus = 2) - so that the first SCALE record the driver can take holds them."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", HC_BENCH_BACKEND="gloo")
    args = ["--steps", "2", "--warmup", "1", "--batch", "2", "--streams", "1", "--no-cpu-baseline", "--relu-batch", "2", "--resnet-batch", "2", "--resnet-images", "4"]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(29700 + os.getpid() % 90),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2"] + args, env=env, capture_output=True, text=True, timeout=1200)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-3000:]
    lines = [ln for ln in two.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["unit"] == "conv/s" and j["value"] > 0
    assert "extras" not in j, j.get("extras")                                    # the watchdog did not have to cut the extras off
    sc = j["sharded_conv"]
    assert sc["n_gpus"] == 2 and sc["sharded_conv_ms"] > 0 and sc["sharded_conv_ms_rccl_gather"] > 0, sc
    rn = j["workloads"]["resnet20"]
    assert rn.get("n_gpus") == 2 and rn["images_per_hour"] > 0 and rn["images_per_launch_set"] == 2, rn
    cr = j["workloads"]["convReLU_5_1"]
    assert cr["ms_per_ct_layer_throughput"] > 0 and cr["layer_latency_ms"] >= cr["ms_per_ct_layer_throughput"] and cr.get("ms_per_layer_n1", 0) > 0, cr
    # the first record from a multi-GPU box is also a correctness record (round 6): both sharded forms against the CPU oracle, the chain against the reference binary's digests
    pk = j["parity_check"]
    assert pk["sharded_conv"] == "ok" and pk["sharded_conv_rccl_gather"] == "ok" and pk["convReLU_5_1"] == "ok", pk


def test_bench_line_validates_itself_on_gpu():
    """the one record the driver takes carries its own proof. bench.py (N = 1, the driver's flags but fewer steps) feeds ciphertext 0 of context 0 the planted
    inputs the oracle gets, clears every output before the timed region, and compares all 131 072 words of what the LAST timed step left with the oracle's; the convReLU 5 1
    launch set (4 images) runs once more on the input and keys planted into the reference binary and must print the binary's three SHA-256 digests. rc = 0 only if both hold."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--resnet-images", "16"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    pk = j["parity_check"]
    assert pk["conv_3_3"] == "ok" and pk["convReLU_5_1"] == "ok", pk
    d = pk["convReLU_5_1_digests"]
    assert d["match"] and all(d[k]["equals_reference_binary"] and len(d[k]["sha256"]) == 2 for k in ("ctos0", "ctos1", "final")), d
    assert "B=256, n=4, chunk=512, 4 contexts" in pk["what"]                         # the timed configuration itself
    assert j["cpu_baseline"]["kind"] == "port" and "SAME planted inputs" in j["cpu_baseline"]["sample"]
    cr = j["workloads"]["convReLU_5_1"]
    assert 0 < cr["valu_frac"] < 1 and cr["issue_floor_ms"] > 0 and cr["lane_instr_per_ct_layer"] > 0, {k: cr.get(k) for k in ("valu_frac", "issue_floor_ms")}
