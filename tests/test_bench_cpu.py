"""bench.py's self-validation helpers on CPU: the oracle leg takes the keys in the form hc_evk_load takes them (row j - 1 of the oracle's key array for galEl = 2^j + 1),
and a differing word is reported, not swallowed."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
import parity_cases as pc  # noqa: E402
from oracle_lib import Oracle  # noqa: E402


def test_oracle_conv_places_the_keys_where_the_oracle_reads_them():
    B, seed = 4, 0xBE7C
    ct_in, ker = pc.planted_conv_inputs(seed, B)
    bias = pc.splitmix_rows(seed + 5, pc.Q0, pc.N)
    keys, evk_all = [], np.zeros((16, 4, pc.N), dtype=np.uint64)
    step, j = B // 2, 16 - ((B // 2).bit_length() - 1)
    while step >= 1:
        k4 = pc.seeded_evk(seed + 7000 + 10 * j)
        keys.append(((1 << j) + 1, list(k4)))
        evk_all[j - 1] = k4
        step //= 2; j += 1
    got, dt = bench.oracle_conv(B, ct_in, ker, keys, bias)
    O = Oracle()
    want, _ = O.conv_then_pack(ct_in, 2.0 ** 30, ker, 2.0 ** 30, O.idx_plaintexts(), evk_all, B, 1, 2.0 ** 30, bias)
    assert np.array_equal(got, want) and dt > 0
    assert bench.first_difference(got, want) == "ok"
    bad = want.copy(); bad[1, 77] ^= np.uint64(1)
    msg = bench.first_difference(bad, want)
    assert msg.startswith("1 of 131072 words differ; first at word 65613 (poly 1, coefficient 77)"), msg
    assert "shape" in bench.first_difference(want[0], want)
