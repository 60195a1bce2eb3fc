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
    torch.cuda.set_device(0)
    backend = os.environ.get("HC_TEST_BACKEND", "nccl")
    dist.init_process_group(backend, rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]), device_id=torch.device("cuda", 0) if backend == "nccl" else None)
    rank, world = dist.get_rank(), dist.get_world_size()
    B, seed = int(os.environ.get("HC_TEST_B", "16")), 0xABCD
    ctx = Context([Q0, Q1], [P0], device=0)
    ct_in, ker = pc.planted_conv_inputs(seed, B)
    evk_all = pc.load_tree_keys(ctx, seed, B)
    ctx.idx_load(None)
    bias = splitmix_rows(seed + 5, Q0, pc.N)
    kh = ctx.ker_load(ker[local_channels(B, rank, world)])
    res, sc = conv_then_pack_sharded(ctx, ctx.buf(ct_in), 2.0 ** 30, kh, 2.0 ** 30, B, 2.0 ** 30, ctx.buf(bias), device="cuda:0")
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
