"""CPU-side checks that need no GPU: the C-ABI library exports every symbol include/hconv.h declares, the product
refuses to run without a GPU, and oracle end-to-end semantics (encrypt -> conv -> decrypt ~ float conv)."""
import os
import re

import numpy as np
import pytest

import golden.gen_conv_csv as gen
from oracle_lib import Oracle, P0, Q0, Q1

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    # include/hconv.h = what a Lattigo host binds; include/hconv_test_hooks.h = the one entry point that exists for this repository's replays
    text = open(os.path.join(ROOT, "include", "hconv.h")).read() + open(os.path.join(ROOT, "include", "hconv_test_hooks.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hc_[a-z0-9_]+)\s*\(", text)))


def test_abi_table_matches_header():
    from optimal_conv_amd import SYMBOLS
    assert sorted(SYMBOLS) == header_symbols()


def test_hip_library_exports_every_declared_symbol():
    from optimal_conv_amd import abi
    if not os.path.exists(abi.DEFAULT_LIB):
        import __graft_entry__
        __graft_entry__.build()
    L = abi.load()                       # types every symbol; AttributeError on drift
    assert L.hc_version() >= 2


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from optimal_conv_amd import Context, HconvError
    with pytest.raises(HconvError):
        Context([Q0, Q1], [P0])


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "optimal_conv_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".cc", ".hpp")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle_lib" not in src and "liboracle" not in src and "oracle/" not in src.replace("oracle/pin", ""), f


@pytest.mark.parametrize("k,i_batch,min_bits", [(3, 0, 20), (5, 1, 18), (7, 2, 17)])
def test_oracle_end_to_end_semantics(k, i_batch, min_bits):
    """CLI configs `conv k i 1` on the oracle alone: decrypted result vs the float convolution (test.go:58-71)."""
    O = Oracle()
    B, W, raw, x, ker, bna, bnb = gen.make_case(k, i_batch, 0)
    sk = O.gen_sk(42)
    ct = O.encrypt(sk, O.encode_coeffs(O.prep_input(x.reshape(-1), raw, W), 2.0 ** 30, [0, 1]), 1, 7)
    kc = O.prep_ker_coeffs(ker.reshape(-1), bna, W, k, B, B)
    pl_ker = np.empty((B, 2, O.N), dtype=np.uint64)
    for i in range(B):
        e = O.encode_coeffs(kc[i], 2.0 ** 30, [0, 1])
        pl_ker[i, 0], pl_ker[i, 1] = O.ntt(0, e[0]), O.ntt(1, e[1])
    bias_pt = O.ntt(0, O.encode_coeffs(O.bias_coeffs(bnb, W), 2.0 ** 30, [0])[0])
    evk = np.zeros((16, 4, O.N), dtype=np.uint64)
    step = B // 2
    j = 16 - (step.bit_length() - 1)
    while step >= 1:
        evk[j - 1] = O.gen_galois_key_l0(sk, (1 << j) + 1, 100 + j)
        step //= 2; j += 1
    got, sc = O.conv_then_pack(ct, 2.0 ** 30, pl_ker, 2.0 ** 30, O.idx_plaintexts(), evk, B, 1, 2.0 ** 30, bias_pt)
    out = O.post_process(O.decrypt_decode_l0(sk, got, sc), raw, W)
    err = np.abs(out - gen.plain_conv(x, ker, bna, bnb).reshape(-1))
    prec = -np.log2(np.maximum(err, 2.0 ** -40))
    assert np.median(prec) >= min_bits, f"median precision {np.median(prec):.1f} bits"


def test_resnet_cli_refuses_the_out_of_scope_variants(tmp_path):
    """the wide networks (wide_case 2 / 3) and the CIFAR-100 head are out of scope; the product CLI says so and exits like a Go panic (status 2) before it
    touches a device, instead of carrying code nobody asked for."""
    import subprocess
    cli = os.path.join(ROOT, "optimal_conv_amd", "host", "conv")
    if not os.path.exists(cli):
        import __graft_entry__
        __graft_entry__.build()
    for argv in (["resnet", "3", "20", "2", "1", "false"], ["resnet", "3", "20", "3", "1", "false"], ["resnet", "3", "20", "1", "1", "true"]):
        r = subprocess.run([cli] + argv, cwd=tmp_path, capture_output=True, text=True, timeout=60)
        assert r.returncode == 2 and "out of scope" in r.stderr, (argv, r.returncode, r.stderr[-300:])
    r = subprocess.run([cli, "resnet", "3", "20", "4", "1", "false"], cwd=tmp_path, capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "Wrong wide case!" in r.stderr                       # main.go:628
