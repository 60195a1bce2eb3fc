"""math.Cos / math.Sin as the reference's Go runtime evaluates them (Go 1.16 src/math/sin.go: the pure-Go Cephes port amd64 uses),
restated with Python floats (IEEE double, one rounding per operation, no contraction). TEST INFRASTRUCTURE: the oracle's slot
encoder builds Lattigo's root table with it; pinned by the SHA-256 of the table inside the reference binary
(tests/golden/ref_trace_enc_3_0.json). The product's copy is optimal_conv_amd/csrc/hc_gomath.h."""
_SIN = (1.58962301576546568060e-10, -2.50507477628578072866e-8, 2.75573136213857245213e-6,
        -1.98412698295895385996e-4, 8.33333333332211858878e-3, -1.66666666666666307295e-1)
_COS = (-1.13585365213876817300e-11, 2.08757008419747316778e-9, -2.75573141792967388112e-7,
        2.48015872888517045348e-5, -1.38888888888730564116e-3, 4.16666666666665929218e-2)
PI4A, PI4B, PI4C = 7.85398125648498535156e-1, 3.77489470793079817668e-8, 2.69515142907905952645e-15
FOUR_OVER_PI = 1.2732395447351628


def _reduce(x):
    j = int(x * FOUR_OVER_PI)
    y = float(j)
    if j & 1:
        j += 1
        y += 1.0
    j &= 7
    z = ((x - y * PI4A) - y * PI4B) - y * PI4C
    return j, z


def _psin(z, zz):
    return z + z * zz * ((((((_SIN[0] * zz) + _SIN[1]) * zz + _SIN[2]) * zz + _SIN[3]) * zz + _SIN[4]) * zz + _SIN[5])


def _pcos(zz):
    return 1.0 - 0.5 * zz + zz * zz * ((((((_COS[0] * zz) + _COS[1]) * zz + _COS[2]) * zz + _COS[3]) * zz + _COS[4]) * zz + _COS[5])


def go_cos(x):
    sign = False
    x = abs(x)
    j, z = _reduce(x)
    if j > 3:
        j -= 4
        sign = not sign
    if j > 1:
        sign = not sign
    zz = z * z
    y = _psin(z, zz) if j in (1, 2) else _pcos(zz)
    return -y if sign else y


def go_sin(x):
    sign = False
    if x < 0:
        x, sign = -x, True
    j, z = _reduce(x)
    if j > 3:
        sign = not sign
        j -= 4
    zz = z * z
    y = _pcos(zz) if j in (1, 2) else _psin(z, zz)
    return -y if sign else y
