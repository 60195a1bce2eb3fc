#!/usr/bin/env python3
"""Stage digests of the ORACLE's full-slot convReLU tails at full size (N = 2^16) on the planted inputs of tests/parity_cases.py: case_conv_relu_tail (Ours: CtoS + sine,
ReLU, mask, StoC on parameter set [6]) and case_bl_boot_relu (the baseline half: imaginary packing, stock Bootstrapp on set [7], ReLU). The oracle chains take one to two
minutes each on one core; the GPU tests compare the device chain's SHA-256 with these instead of running the oracle beside the device (HCONV_TEST_FULL_ORACLE=1 does that).

    python tests/golden/gen_full_tail_digests.py
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import parity_cases as pc  # noqa: E402

if __name__ == "__main__":
    doc = {"logN": 16, "made_by": "tests/golden/gen_full_tail_digests.py", "cases": {}}
    for name, fn in (("conv_relu_tail", pc.conv_relu_tail_oracle_digests), ("bl_boot_relu", pc.bl_boot_relu_oracle_digests)):
        t0 = time.time()
        doc["cases"][name] = fn()
        print(name, f"{time.time() - t0:.0f} s", doc["cases"][name], flush=True)
        json.dump(doc, open(pc.FULL_TAIL_FIXTURE, "w"), indent=1, sort_keys=True)
