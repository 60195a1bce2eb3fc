"""N > 1 path on CPU: two processes over gloo run the i-mod-G sharded convolution (optimal_conv_amd/sharded.py)
with the kernel SOURCES executing under the fiber emulator (tests/kernel_emu); rank 0's result must equal the
oracle's single-process conv_then_pack bit for bit. Also covers bench.py's weak-scaling bookkeeping helpers."""
import os
import subprocess
import sys
import textwrap

import pytest

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
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank, world = dist.get_rank(), dist.get_world_size()
    B, seed = int(os.environ["HC_B"]), 0xABCD
    ctx = Context([Q0, Q1], [P0], lib_path=os.environ["HC_EMU_LIB"])
    ct_in, ker = pc.planted_conv_inputs(seed, B)
    evk_all = pc.load_tree_keys(ctx, seed, B)
    ctx.idx_load(None)
    bias = splitmix_rows(seed + 5, Q0, pc.N)
    mine = local_channels(B, rank, world)
    kh = ctx.ker_load(ker[mine])
    res, sc = conv_then_pack_sharded(ctx, ctx.buf(ct_in), 2.0 ** 30, kh, 2.0 ** 30, B, 2.0 ** 30, ctx.buf(bias))
    if rank == 0:
        O = Oracle()
        want, wsc = O.conv_then_pack(ct_in, 2.0 ** 30, ker, 2.0 ** 30, O.idx_plaintexts(), evk_all, B, 1, 2.0 ** 30, bias)
        got = res.numpy().view(np.uint64).reshape(2, pc.N)
        pc.eq(got, want, "sharded conv_then_pack")
        assert sc == wsc
        print("SHARDED_OK")
    dist.barrier()
    dist.destroy_process_group()
""") % (ROOT, HERE)


@pytest.mark.parametrize("world,B", [(2, 4), (2, 8), (4, 8)])
def test_sharded_conv_world(tmp_path, world, B):
    emu_dir = os.path.join(HERE, "kernel_emu")
    emu_lib = os.path.join(emu_dir, "_build", "libhconv_emu.so")
    subprocess.check_call(["make", "-s", "-C", emu_dir, emu_lib])
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, HC_EMU_LIB=emu_lib, HC_B=str(B), MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    port = 29500 + (os.getpid() % 500) + world * 7 + B
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "SHARDED_OK" in out.stdout
