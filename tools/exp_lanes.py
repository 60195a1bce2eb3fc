#!/usr/bin/env python3
"""round 3 experiment: ONE `conv 3 3` at a time on G internal lanes (hc_set_option lanes / lane_priority), 10 back to back, ms per convolution.
Run on the GPU box: python tools/exp_lanes.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from optimal_conv_amd import Context
from oracle_lib import Q0, Q1, P0

B, N = 256, 1 << 16
rng = np.random.default_rng(1)
for lanes, prio in ((1, 0), (2, 0), (2, 1), (4, 0), (4, 1), (8, 1)):
    ctx = Context([Q0, Q1], [P0])
    for j in range(1, 17):
        ctx.evk_load((1 << j) + 1, [rng.integers(0, q, N, dtype=np.uint64) for q in (Q0, P0, Q0, P0)])
    ctx.idx_load(None)
    if prio:
        ctx.set_option("lane_priority", 1)
    ctx.set_option("lanes", lanes)
    ker = ctx.ker_load(np.stack([np.stack([rng.integers(0, Q0, N, dtype=np.uint64), rng.integers(0, Q1, N, dtype=np.uint64)]) for _ in range(B)]))
    cin = ctx.buf(np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in (Q0, Q1, Q0, Q1)]))
    bias = ctx.buf(rng.integers(0, Q0, N, dtype=np.uint64))
    out = ctx.buf(nwords=2 * N)
    for _ in range(3):
        ctx.conv_then_pack_dev(cin, 2.0 ** 30, ker, 2.0 ** 30, B, 1, 2.0 ** 30, bias, out)
    ctx.sync()
    ref = out.download()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            ctx.conv_then_pack_dev(cin, 2.0 ** 30, ker, 2.0 ** 30, B, 1, 2.0 ** 30, bias, out)
        ctx.sync()
        best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
    print(f"lanes {lanes} priority {prio}: {best:.3f} ms per convolution  sha-equal-across-configs {hash(ref.tobytes()) & 0xffffffff:08x}", flush=True)
    ctx.ker_free(ker); ctx.close()
