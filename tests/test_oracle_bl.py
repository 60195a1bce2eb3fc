"""The BL baseline oracle (tests/oracle_bl.py: the reference's slot-packed convolution, test_BL.go:16-185 with
boot = false, on the pinned residue primitives + a numpy restatement of Lattigo's slot encoder) end to end:
encode -> encrypt -> 4 x evalConv_BN_BL_test -> decrypt -> decode must reproduce the plain convolution to the
precision the reference binary prints for its "Base Line" run at B = 4 (MED 21.4 bits, /tmp run logged in DESIGN.md)."""
import numpy as np

import oracle_bl


def test_encoder_roundtrip():
    rng = np.random.default_rng(3)
    v = rng.normal(size=oracle_bl.SLOTS) + 1j * rng.normal(size=oracle_bl.SLOTS)
    w = oracle_bl.fft_special(oracle_bl.invfft_special(v))
    assert np.max(np.abs(w - v)) < 1e-9


def test_rotation_galois_elements():
    N2 = 2 * oracle_bl.N
    assert oracle_bl.gal_for_rotation(0) == 1 and oracle_bl.gal_for_rotation(1) == 5
    assert oracle_bl.gal_for_rotation(-1) * 5 % N2 == 1
    assert oracle_bl.gal_for_rotation(129) == pow(5, 129, N2)


def test_bl_conv_end_to_end():
    got, want = oracle_bl.testConv_BL_in(3, 0)
    prec = -np.log2(np.abs(got - want) + 2.0 ** -60)
    assert np.median(prec) >= 20.5, np.median(prec)
