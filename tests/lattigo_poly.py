"""ckks.(*evaluator).EvaluatePoly of the reference's Lattigo fork (standard basis: computePowerBasis, recurse, splitCoeffs,
evaluatePolyFromPowerBasis; test_run symbols ckks.recurse @52eb00, ckks.evaluatePolyFromPowerBasis @52fd00, ckks.computePowerBasis
@52dbc0), restated over an abstract leveled backend. Control flow, float64 scale arithmetic and integer constants follow the op
log the binary itself produced under `gotrace -poly` (tests/golden/ref_trace_poly_5_1.json): every nested mulRelin / Rescale /
MultByGaussianIntegerAndAdd / Add result of the three sign polynomials of evalReLU (conv.go:460-477) is reproduced bit for bit by
tests/test_oracle_pin_poly.py. TEST INFRASTRUCTURE shared by the oracle (tests/oracle_ckks.py); the product's copy is
optimal_conv_amd/host/hconv_relu.cpp (Boot::evaluate_poly).

Backend protocol (ciphertexts are opaque to this file): level(ct), scale(ct), mul_relin(a, b) [levels aligned to the lower one,
scale = product], rescale(ct, min_scale) [ckks Rescale: drop limbs while scale / q_level >= min_scale / 2], zero(level, scale),
mul_int_add(ct, c, acc) [acc += ct * c for the integer c, acc's scale unchanged], mul_int(ct, c) [ct * c, scale unchanged],
add_rows(a, b, scale) [a + b at the lower level, labelled `scale`], drop(ct, levels), add_const(ct, c), q(level)."""


class Poly:
    def __init__(self, coeffs, max_deg=None, lead=True):
        self.coeffs = [float(c) for c in coeffs]
        self.max_deg = len(self.coeffs) - 1 if max_deg is None else max_deg
        self.lead = lead

    def degree(self):
        return len(self.coeffs) - 1


def split_coeffs(p, split):
    """p = q * X^split + r (ckks.splitCoeffs)"""
    r = Poly(p.coeffs[:split], split - 1 if p.max_deg == p.degree() else p.max_deg - (p.degree() - split + 1), False)
    q = Poly(p.coeffs[split:], p.max_deg, p.lead)
    return q, r


def lattigo_add(be, a, b):
    """evaluator.Add(a, b, a): the operand with the smaller scale is multiplied by uint64(ratio) first (evaluateInPlace)"""
    sa, sb = be.scale(a), be.scale(b)
    if sa > sb:
        k = int(sa / sb)
        if k > 1:          # (k == 1 multiplies by one: the binary's op log shows no MultByConst for it)
            b = be.mul_int(b, k)
        return be.add_rows(a, b, sa)
    if sb > sa:
        k = int(sb / sa)
        if k > 1:          # (k == 1 multiplies by one: the binary's op log shows no MultByConst for it)
            a = be.mul_int(a, k)
        return be.add_rows(a, b, sb)
    return be.add_rows(a, b, sa)


def compute_power_basis(be, C, n, scale):
    if n in C:
        return
    a, b = (n + 1) // 2, n >> 1                       # ceil(n/2), floor(n/2)
    compute_power_basis(be, C, a, scale)
    compute_power_basis(be, C, b, scale)
    C[n] = be.rescale(be.mul_relin(C[a], C[b]), scale)


def evaluate_from_power_basis(be, target_scale, p, C, scale):
    if p.degree() == 0:
        res = be.zero(be.level(C[1]), target_scale)
        if abs(p.coeffs[0]) > 1e-14:
            res = be.add_const(res, p.coeffs[0])
        return res
    lv = be.level(C[p.degree()])
    current_qi = float(be.q(lv))
    res = be.zero(lv, target_scale * current_qi)
    if abs(p.coeffs[0]) > 1e-14:
        res = be.add_const(res, p.coeffs[0])
    for key in range(p.degree(), 0, -1):
        if abs(p.coeffs[key]) > 1e-14:
            const_scale = target_scale * current_qi / be.scale(C[key])
            res = be.mul_int_add(C[key], int(p.coeffs[key] * const_scale), res)      # Go's int64(float64): truncation
    return be.rescale(res, scale)


def recurse(be, target_scale, log_split, log_degree, p, C, scale):
    if p.degree() < (1 << log_split):
        # the leading leaf is split again with a smaller baby set when it is long enough to cost a level more than its neighbours.
        # (Upstream tests maxDeg % 2^(logSplit+1) here; the fork's binary sends [0, c13] of a degree-13 polynomial -- maxDeg 13,
        # 13 % 8 = 5 > 2 -- straight to the leaf, so its test is on the leaf's own degree.)
        if p.lead and log_split > 1 and p.degree() > (1 << (log_split - 1)):
            log_degree = p.degree().bit_length()
            log_split = log_degree >> 1
            return recurse(be, target_scale, log_split, log_degree, p, C, scale)
        return evaluate_from_power_basis(be, target_scale, p, C, scale)
    next_power = 1 << log_split
    while next_power < (p.degree() >> 1) + 1:
        next_power <<= 1
    pq, pr = split_coeffs(p, next_power)
    level = be.level(C[next_power]) - 1
    if p.max_deg >= 1 << (log_degree - 1) and p.lead:
        level += 1
    current_qi = float(be.q(level))
    res = recurse(be, target_scale * current_qi / be.scale(C[next_power]), log_split, log_degree, pq, C, scale)
    tmp = recurse(be, target_scale, log_split, log_degree, pr, C, scale)
    if be.level(res) > be.level(tmp):
        while be.level(res) != be.level(tmp) + 1:
            res = be.drop(res, 1)
    res = be.mul_relin(res, C[next_power])
    if be.level(res) > be.level(tmp):
        res = be.rescale(res, scale)
        res = lattigo_add(be, res, tmp)
    else:
        res = lattigo_add(be, res, tmp)
        res = be.rescale(res, scale)
    return res


def evaluate_poly(be, ct, coeffs, target_scale, scale):
    """EvaluatePoly(ct, NewPoly(coeffs), target_scale) with evaluator.scale = `scale` (params.Scale())"""
    p = Poly(coeffs)
    C = {1: ct}
    log_degree = p.degree().bit_length()
    log_split = log_degree >> 1
    for i in range(2, 1 << log_split):
        compute_power_basis(be, C, i, scale)
    for i in range(log_split, log_degree):
        compute_power_basis(be, C, 1 << i, scale)
    return recurse(be, target_scale, log_split, log_degree, p, C, scale)


# ------------------------------------------------------------------ Chebyshev basis (the sine of evaluateSine)
# ckks.(*evaluator).EvaluateCheby @52d7c0, computePowerBasisCheby @52dee0, splitCoeffsCheby @52e7c0, recurseCheby @52f400; the leaf
# (evaluatePolyFromPowerBasis) is shared with the standard basis. Pinned by tests/golden/ref_trace_cheby_5_1.json (gotrace -cheby 1).
# Additional backend calls: sub_rows(a, b, scale) [a - b at the lower level].
def lattigo_sub(be, a, b):
    """evaluator.Sub(a, b, a) with the same scale matching as Add"""
    sa, sb = be.scale(a), be.scale(b)
    if sa > sb:
        k = int(sa / sb)
        if k > 1:          # (k == 1 multiplies by one: the binary's op log shows no MultByConst for it)
            b = be.mul_int(b, k)
        return be.sub_rows(a, b, sa)
    if sb > sa:
        k = int(sb / sa)
        if k > 1:          # (k == 1 multiplies by one: the binary's op log shows no MultByConst for it)
            a = be.mul_int(a, k)
        return be.sub_rows(a, b, sb)
    return be.sub_rows(a, b, sa)


def compute_power_basis_cheby(be, C, n, scale):
    """C[n] = 2 C[a] C[b] - C[|a-b|], a = ceil(n/2), b = floor(n/2) (C[0] = 1 is not stored)"""
    if n in C:
        return
    a, b = (n + 1) // 2, n >> 1
    c = a - b
    compute_power_basis_cheby(be, C, a, scale)
    compute_power_basis_cheby(be, C, b, scale)
    if c != 0:
        compute_power_basis_cheby(be, C, c, scale)
    t = be.rescale(be.mul_relin(C[a], C[b]), scale)
    t = lattigo_add(be, t, t)
    C[n] = be.add_const(t, -1.0) if c == 0 else lattigo_sub(be, t, C[c])


def split_coeffs_cheby(p, split):
    """p = q * T_split + r in the Chebyshev basis: q_0 = p_split, q_j = 2 p_(split+j), r_(split-j) -= p_(split+j)"""
    r = Poly(p.coeffs[:split], split - 1 if p.max_deg == p.degree() else p.max_deg - (p.degree() - split + 1), False)
    q = Poly([0.0] * (p.degree() - split + 1), p.max_deg, p.lead)
    q.coeffs[0] = p.coeffs[split]
    for i, j in zip(range(split + 1, p.degree() + 1), range(1, p.degree() + 1)):
        q.coeffs[i - split] = 2 * p.coeffs[i]
        r.coeffs[split - j] -= p.coeffs[i]
    return q, r


def recurse_cheby(be, target_scale, log_split, log_degree, p, C, scale):
    if p.degree() < (1 << log_split):
        if p.lead and log_split > 1 and p.max_deg > ((1 << log_degree) - (1 << (log_split - 1))):
            log_degree = p.degree().bit_length()
            log_split = log_degree >> 1
            return recurse_cheby(be, target_scale, log_split, log_degree, p, C, scale)
        return evaluate_from_power_basis(be, target_scale, p, C, scale)
    next_power = 1 << log_split
    while next_power < (p.degree() >> 1) + 1:
        next_power <<= 1
    pq, pr = split_coeffs_cheby(p, next_power)
    level = be.level(C[next_power]) - 1
    if p.max_deg >= 1 << (log_degree - 1) and p.lead:
        level += 1
    current_qi = float(be.q(level))
    res = recurse_cheby(be, target_scale * current_qi / be.scale(C[next_power]), log_split, log_degree, pq, C, scale)
    tmp = recurse_cheby(be, target_scale, log_split, log_degree, pr, C, scale)
    if be.level(res) > be.level(tmp):
        while be.level(res) != be.level(tmp) + 1:
            res = be.drop(res, 1)
    res = be.mul_relin(res, C[next_power])
    if be.level(res) > be.level(tmp):
        res = be.rescale(res, scale)
        res = lattigo_add(be, res, tmp)
    else:
        res = lattigo_add(be, res, tmp)
        res = be.rescale(res, scale)
    return res


def evaluate_cheby(be, ct, coeffs, target_scale, scale, max_deg=None, lead=True):
    """EvaluateCheby(ct, cheby, target_scale): ct already holds T_1 (the change of variable is the caller's, evaluateCheby @508540)"""
    p = Poly(coeffs, max_deg, lead)
    C = {1: ct}
    log_degree = p.degree().bit_length()
    log_split = log_degree >> 1
    for i in range(2, 1 << log_split):
        compute_power_basis_cheby(be, C, i, scale)
    for i in range(log_split, log_degree):
        compute_power_basis_cheby(be, C, 1 << i, scale)
    return recurse_cheby(be, target_scale, log_split, log_degree, p, C, scale)
