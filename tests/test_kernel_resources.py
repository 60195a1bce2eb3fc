"""The gfx950 build's register budget, checked without a GPU: hipcc's kernel-resource-usage remarks for the kernels of the hot path. A kernel of the convolution (loops A and B)
or of the batched multi-modulus transforms that starts spilling to scratch loses tens of percent silently (round 5: constant-address-space twiddle loads were harmless in the
transform kernels and cost hc_k_b3 / hc_k_b5m 430-530 bytes of scratch and the convolution 19 % of its rate) - so a spill there fails the CPU suite, not a later bench."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# kernels that must not touch scratch (name prefix after the _Z<len> mangling prefix is stripped). The NS = 8 instantiations of the extension passes (6..8 special primes)
# spilled 208 bytes per lane: they are gone since round 6 and hc_ctx_create refuses np > 5 (test below).
NO_SCRATCH = ["hc_k_a1", "hc_k_a2", "hc_k_a3", "hc_k_b1", "hc_k_b2", "hc_k_b3", "hc_k_b4", "hc_k_b5", "hc_k_sb", "hc_k_rows_fwd_canon_mm", "hc_k_rows_inv_mm",
              "hc_k_cols_inv_canon_mm", "hc_k_cols_fwd_mmILi0E", "hc_k_cols_fwd_mmILi1ELi2E", "hc_k_cols_fwd_mmILi2ELi2E", "hc_k_cols_fwd_mmILi2ELi5E", "hc_k_ks_mac", "hc_k_qp_mul_sum",
              "hc_k_lv_", "hc_k_basis_yv"]
# hc_k_cols_fwd_mm<1, 5> at five wavefronts keeps 12 bytes of scratch in its rarely taken generic-digit path (measured faster than four wavefronts without: profiles/LEDGER.md)
# hc_k_ks_mac_multi<8, 2, false, *>: the rot_fuse = 0 A/B form (not the default) at two images per launch set keeps 68 bytes (eight key pointers and eight accumulator pointers: scalar registers)
SMALL_SCRATCH = {"hc_k_cols_fwd_mmILi1ELi5E": 16, "hc_k_ks_mac_multiILi8ELi2ELb0E": 72}


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_hot_kernels_do_not_spill():
    import __graft_entry__                # the flags the shipped libhconv.so is built with (a hard-coded copy here could drift from the build)
    r = subprocess.run([HIPCC, *__graft_entry__.HIP_FLAGS, "--cuda-device-only", "-S", "-o", os.devnull,
                        os.path.join(ROOT, "optimal_conv_amd", "csrc", "hconv.hip"), "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    names = re.findall(r"Function Name: _Z\d+(\S+)", r.stderr)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
    assert names and len(names) == len(scratch)
    seen = set()
    for name, sc in zip(names, scratch):
        for pre in NO_SCRATCH:
            if name.startswith(pre) and not any(name.startswith(w) for w in SMALL_SCRATCH):
                seen.add(pre)
                assert sc == 0, f"{name}: {sc} bytes of scratch per lane"
        for pre, cap in SMALL_SCRATCH.items():
            if name.startswith(pre):
                seen.add(pre)
                assert sc <= cap, f"{name}: {sc} bytes of scratch per lane (cap {cap})"
    missing = [p for p in list(NO_SCRATCH) + list(SMALL_SCRATCH) if p not in seen]
    assert not missing, f"kernels not found in the build: {missing}"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_no_spilling_instantiation_is_reachable():
    """hc_k_cols_fwd_mm<1, 8> / <2, 8> (208 B of scratch per lane) were reachable through hc_ctx_create(np = 6..8). They are not compiled any more."""
    src = open(os.path.join(ROOT, "optimal_conv_amd", "csrc", "hconv.hip")).read()
    assert "hc_k_cols_fwd_mm<E, 8>" not in src and "np > HC_MAX_NP" in src
