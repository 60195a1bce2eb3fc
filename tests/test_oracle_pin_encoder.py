"""The slot encoder pinned against the reference binary: `gotrace -keep-bl -noplant -enc 24` ran /root/reference/test_run
`conv 3 0 1` (its BL baseline half, test_BL.go:16-185) under ptrace and recorded, with nothing planted, SHA-256 digests of
  * the encoder's root table (131 073 complex128 from Go's math.Cos / math.Sin) and
  * for the first 24 ckks.invfft calls the complex128 input vector and the output vector, and
  * the plaintext every ckks.Encode call left (coefficient domain, scaleUpVecExact applied, 2 limbs)
(tests/golden/ref_trace_enc_3_0.json). The data are the run's own: the CSVs tests/golden/gen_conv_csv.py writes, so the oracle
regenerates every input -- the two input images (reshape_input_BL), the BN-bias slots (eval.go:93-99) and the 18 kernel-tap
vectors `postKer` (conv.go:150-164) -- and must reproduce EVERY digest: tests/go_math.py (Go's trig), the special inverse FFT
and the rounding of tests/oracle_bl.py are then the reference's, bit for bit. (CPU only; the GPU encoder is compared with this
oracle in tests/parity_cases.py::case_encode_slots.)"""
import hashlib
import json
import os

import numpy as np

import golden.gen_conv_csv as gen
import oracle_bl as ob

HERE = os.path.dirname(os.path.abspath(__file__))
TRACE = json.load(open(os.path.join(HERE, "golden", "ref_trace_enc_3_0.json")))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def reference_slot_vectors(k=3, i_batch=0):
    """every vector the BL run of `conv k i 1` hands to the encoder, in call order, with its scale (test_BL.go:60-111, eval.go:78-134)"""
    B, W, raw, x, ker, bna, bnb = gen.make_case(k, i_batch, 0)
    pad, in_size = k // 2, W * W
    inp = x.reshape(raw, raw, B)
    pads = [np.zeros((W, W, B // 2)) for _ in range(2)]
    pads[0][:raw, :raw, :] = inp[:, :, : B // 2]; pads[1][:raw, :raw, :] = inp[:, :, B // 2:]
    out = [(np.asarray(ob.reshape_input_BL(p.reshape(-1), W), dtype=np.complex128), 2.0 ** 30) for p in pads]
    kk = ker.reshape(k * k, B, B)
    max_batch = ob.N // (2 * in_size)
    for pos in range(2):
        for inn in range(2):
            ksep = kk[:, inn * (B // 2): (inn + 1) * (B // 2), pos * (B // 2): (pos + 1) * (B // 2)].reshape(-1)
            bn_a = bna[pos * (B // 2): (pos + 1) * (B // 2)]
            bn_b = bnb[pos * (B // 2): (pos + 1) * (B // 2)] if inn == 0 else np.zeros(B // 2)
            max_ker_rs = ob.reshape_ker_BL(ksep, bn_a, k, B // 2, B // 2, max_batch)
            slots = np.zeros(ob.SLOTS, dtype=np.complex128)
            for i, elt in enumerate(bn_b):
                blk = slots[in_size * i: in_size * (i + 1)].reshape(W, W)
                blk[: W - pad, : W - pad] = elt
            out.append((slots, 2.0 ** 60))
            for r in range(max_batch):
                for i in range(k):
                    for j in range(k):
                        out.append((ob.postKer(max_ker_rs, i, j, W, k, r, pad, max_batch), 2.0 ** 30))
    return out


def test_root_table_is_the_reference_binarys():
    tab = next(e for e in TRACE["events"] if e["op"] == "encoder_tables")
    assert tab["M"] == ob.M and tab["roots_len"] == ob.M + 1
    flat = np.empty(2 * (ob.M + 1)); flat[0::2] = ob._ROOT_RE; flat[1::2] = ob._ROOT_IM
    assert sha(flat) == tab["roots"]
    assert [float(v) for v in flat[2:6]] == tab["roots_1_2"]


def test_special_fft_and_encode_match_every_digest_of_the_reference_run():
    inv = [e for e in TRACE["events"] if e["op"] == "invfft"]
    enc = {e["call"]: e for e in TRACE["events"] if e["op"] == "Encode"}
    vecs = reference_slot_vectors()
    assert len(inv) == 24 and len(vecs) >= len(inv)
    bl = ob.BLOracle()
    for e, (v, scale) in zip(inv, vecs):
        assert e["n"] == ob.SLOTS
        assert sha(v) == e["in"], f"slot vector of encoder call {e['call']} (layout functions)"
        assert sha(ob.invfft_special(v)) == e["out"], f"special inverse FFT, call {e['call']}"
        if e["call"] in enc:
            pe = enc[e["call"]]
            assert pe["scale"] == scale
            rows = ob.encode_slots(bl.O, v, pe["pt"]["limbs"] - 1, scale)
            assert sha(rows) == pe["pt"]["sha256"], f"Encode (rounding, both limbs), call {e['call']}"
