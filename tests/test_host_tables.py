"""Constant tables the C++ host carries must equal the fixtures they were read from."""
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))


def test_sine_coefficient_table_equals_the_reference_trace():
    want = [c[0] for c in json.load(open(os.path.join(HERE, "golden", "ref_trace_cheby_5_1.json")))["events"][0]["pol"]["coeffs"]]
    src = open(os.path.join(HERE, "..", "optimal_conv_amd", "host", "hconv_sine_coeffs.hpp")).read()
    body = src.split("FORK_SINE_COEFFS[63] = {")[1].split("};")[0]
    got = [float.fromhex(t) for t in re.findall(r"-?0x[0-9a-f.]+p[-+]?\d+", body)]
    assert got == want and len(got) == 63
