"""One homomorphic convolution split across G GPUs (BASELINE config `conv 7 3` sharded over 8 GPUs).

conv.go:525-531's B products are independent and conv.go:286-297's tree pairs (i, i+step): with output channels
assigned by i mod G every level with step >= G is rank-local; only the last log2 G levels need the G partial
ciphertexts (1 MiB each) in one place. So: loop A + strided local tree on every rank, ONE gather of G x 1 MiB to
rank 0 (RCCL over xGMI on GPUs, gloo in the CPU tests), last log2 G levels + bias there. No other collective.

torch is used only for the process group and as the owner of the buffers that cross ranks; all arithmetic goes
through the C ABI (device pointers are plain pointers, so torch storage can be handed to libhconv directly).
"""
import ctypes as C

import torch
import torch.distributed as dist


def local_channels(B, rank, world):
    """output channels rank `rank` owns: i = rank (mod world)"""
    return list(range(rank, B, world))


def conv_then_pack_sharded(ctx, ct_in_buf, ct_scale, ker_local, ker_scale, B, out_scale, bias_buf=None, device="cpu", group=None):
    """ctx: optimal_conv_amd.Context of this rank; ct_in_buf: replicated level-1 input (DevBuf);
    ker_local: hc_ker handle holding pl_ker[rank + world*m], m = 0..B/world-1; returns (torch.int64 tensor [2*N]
    holding the level-0 result on rank 0, scale) - None on other ranks."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    assert B % world == 0 and world & (world - 1) == 0, "world size must be a power of two dividing B"
    N, nloc = ctx.N, B // world
    log2g = world.bit_length() - 1
    cts = torch.empty(nloc * 2 * N, dtype=torch.int64, device=device)
    p = C.c_void_p(cts.data_ptr())
    # loop A on the local channels. SetScale's target is out_scale/B (conv.go:528): max_ob/norm here is B/world,
    # so the out_scale handed down is out_scale/world - same constant, same single-limb rescale.
    ctx._ck(ctx.L.hc_conv_mult_phase(ctx.h, ct_in_buf.ptr, ct_scale, ker_local, ker_scale, nloc, 1, out_scale / world, p))
    # tree levels with step >= world (local slot m is global channel rank + world*m => stride 2^log2g)
    ctx._ck(ctx.L.hc_pack_ctxts_strided(ctx.h, p, nloc, log2g, None))
    ctx.sync()
    part = cts[: 2 * N]
    # gloo moves host memory only: the single-GPU dry run of the N-rank path (ranks share one device, gloo barriers) stages the
    # 1 MiB partials through the host; over RCCL the CUDA tensors go as they are (xGMI)
    via_host = part.is_cuda and dist.get_backend(group) == "gloo"
    xdev = "cpu" if via_host else device
    if via_host:
        part = part.cpu()
    gathered = [torch.empty(2 * N, dtype=torch.int64, device=xdev) for _ in range(world)] if rank == 0 else None
    dist.gather(part, gathered, dst=0, group=group)          # the only exchange: world x 1 MiB
    if rank != 0:
        return None, None
    allp = torch.cat(gathered).to(device)                       # [world][2][N], slot g = global channel g
    if str(device).startswith("cuda"):
        # torch's gather and cat run on torch's stream, the tree below on libhconv's own (possibly non-blocking) stream:
        # everything torch queued must be complete before the library reads `allp`
        torch.cuda.synchronize()
    ctx._ck(ctx.L.hc_pack_ctxts_strided(ctx.h, C.c_void_p(allp.data_ptr()), world, 0, bias_buf.ptr if bias_buf is not None else None))
    ctx.sync()
    return allp[: 2 * N], out_scale
