#!/usr/bin/env python3
"""Stage digests of the ORACLE's sparse-slot convReLU tails at full size (N = 2^16), for the geometries `resnet 3 20 1 n false` uses
(test.go:76-370: Conv_sparse at log_sparse 2 / 3 / 4 on 32 / 16 / 8-wide images; StrConv_sparse at log_sparse 1 / 2 on 32 / 16-wide
images) plus the square log_sparse-2 case round 2 tested. The oracle chain takes minutes per case on one core, so it is run once here
(build container) and the GPU tests compare the device chain's SHA-256 with these; `HCONV_TEST_FULL_ORACLE=1` makes them run the oracle
beside the device instead, and tests/test_oracle_ckks.py::test_sparse_tail_digest_fixture (slow) re-derives one entry.

    python tests/golden/gen_sparse_tail_digests.py [case ...]      # e.g. conv_ls3_w16; default: all
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import parity_cases as pc  # noqa: E402

CASES = [("conv", 2, 64), ("conv", 2, 32), ("conv", 3, 16), ("conv", 4, 8), ("conv", 1, 32), ("strconv", 1, 32), ("strconv", 2, 16)]

if __name__ == "__main__":
    want = set(sys.argv[1:])
    path = pc.SPARSE_TAIL_FIXTURE
    doc = json.load(open(path)) if os.path.exists(path) else {"logN": 16, "seed": 3, "made_by": "tests/golden/gen_sparse_tail_digests.py", "cases": {}}
    for kind, ls, w in CASES:
        name = f"{kind}_ls{ls}_w{w}"
        if want and name not in want:
            continue
        t0 = time.time()
        doc["cases"][name] = pc.sparse_tail_oracle_digests(kind, ls, w)
        print(name, f"{time.time() - t0:.0f} s", doc["cases"][name], flush=True)
        json.dump(doc, open(path, "w"), indent=1, sort_keys=True)
