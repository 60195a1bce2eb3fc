"""Regenerates optimal_conv_amd/host/hconv_sine_coeffs.hpp from tests/golden/ref_trace_cheby_5_1.json (hex float literals, exact)."""
import json, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cs = [c[0] for c in json.load(open(os.path.join(R, "tests", "golden", "ref_trace_cheby_5_1.json")))["events"][0]["pol"]["coeffs"]]
p = os.path.join(R, "optimal_conv_amd", "host", "hconv_sine_coeffs.hpp")
head, tail = open(p).read().split("static const double FORK_SINE_COEFFS[63] = {\n")[0], "};\n}  // namespace hconv\n"
body = "\n".join("    " + ", ".join(float(c).hex() for c in cs[i:i + 4]) + "," for i in range(0, len(cs), 4))
open(p, "w").write(head + "static const double FORK_SINE_COEFFS[63] = {\n" + body + "\n" + tail)
