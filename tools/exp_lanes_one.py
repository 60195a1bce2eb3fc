#!/usr/bin/env python3
"""one configuration of tools/exp_lanes.py for a kernel trace: python tools/exp_lanes_one.py LANES PRIORITY"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from optimal_conv_amd import Context
from oracle_lib import Q0, Q1, P0
lanes, prio = int(sys.argv[1]), int(sys.argv[2])
B, N = 256, 1 << 16
rng = np.random.default_rng(1)
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
for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 4):
    ctx.conv_then_pack_dev(cin, 2.0 ** 30, ker, 2.0 ** 30, B, 1, 2.0 ** 30, bias, out)
ctx.sync()
