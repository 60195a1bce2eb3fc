"""Test infrastructure: ckks.(*evaluator).MultiplyByDiagMatrixBSGS of the reference's Lattigo fork (test_run @52a580, behind LinearTransform
@5264c0), restated over the oracle's primitives. The fork keeps everything in the extended basis QP until the last moment:

  * the baby-step rotations are key-switched WITHOUT the division by P (rotateHoistedNoModDown -> KeyswitchHoistedNoModDown, one digit
    decomposition of c1 for all of them) and P*c0 is added to their first component, so rot_i = (phi_i(P c0 + d0_i), phi_i(d1_i)) mod QP;
  * per giant step j != 0 the products with the plaintext diagonals (encoded mod Q AND mod P) are summed in QP and brought down ONCE
    (ModDownSplitNTTPQ on both components); a rotation-0 diagonal of that giant step multiplies the input itself and is added after the
    division; the second component is key-switched again without the division (SwitchKeysInPlaceNoModDown), the first is permuted straight
    into the result, the key-switch outputs are permuted and accumulated in QP;
  * the giant step 0 adds its products to the same QP accumulators; they are brought down once, added to the result, and the rotation-0
    diagonal of giant step 0 multiplies the input itself.

Go walks the giant steps in map order; every sum is exact modular arithmetic and every ModDown sees a complete sum, so the result does not
depend on it. Pinned by tests/golden/ref_trace_lt_5_1.json (gotrace -lt) in tests/test_oracle_pin_lt.py: every ModDownSplitNTTPQ input and
output, the key-switch input and the returned ciphertext."""
import numpy as np


def bsgs_index(keys, slots, n1):
    index = {}
    for k in keys:
        k &= slots - 1
        index.setdefault(k // n1, []).append(k & (n1 - 1))
    return index


class QP:
    """helpers on rows in the basis Q_0..Q_level, P_0..P_(alpha-1) of an oracle context"""

    def __init__(self, O, level):
        self.O, self.level = O, level
        self.mods = list(range(level + 1)) + [len(O.q) + j for j in range(len(O.p))]
        self.idx = {}

    def perm_index(self, k):
        if k not in self.idx:
            self.idx[k] = self.O.permute_index(pow(5, k % (2 * self.O.N), 2 * self.O.N))
        return self.idx[k]

    def permute(self, k, rows):
        idx = self.perm_index(k)
        return np.stack([self.O.permute(idx, r) for r in rows])

    def mul(self, a, b, nrows=None):
        n = nrows or len(self.mods)
        return np.stack([self.O.mul(self.mods[t], a[t], b[t]).reshape(-1) for t in range(n)])

    def add(self, a, b, nrows=None):
        n = nrows or len(self.mods)
        return np.stack([self.O.add(self.mods[t], a[t], b[t]).reshape(-1) for t in range(n)])


def multiply_by_diag_matrix_bsgs(O, level, ct, diags_qp, n1, slots, key_of_rotation):
    """ct: (2, level+1, N) NTT rows; diags_qp: {rotation N1*j+i: (level+1+alpha, N) NTT rows of the diagonal pre-rotated by -N1*j, mod Q then mod P};
    key_of_rotation(k) -> evk rows [beta][2][level+1+alpha][N]. Returns (2, level+1, N) and a log of the ModDown / key-switch checkpoints."""
    nl, N = level + 1, O.N
    qp = QP(O, level)
    index = bsgs_index(list(diags_qp), slots, n1)
    log = []
    c0, c1 = ct[0], ct[1]
    # P * c0 (MulScalarBigintLvl with P = prod p_j), zero on the P rows
    Pbig = 1
    for p in O.p:
        Pbig *= p
    pc0 = np.zeros((len(qp.mods), N), dtype=np.uint64)
    for l in range(nl):
        pc0[l] = O.mul_scalar(l, c0[l], Pbig % O.q[l]).reshape(-1)
    rot = {}
    for i in sorted({i for js in index.values() for i in js}):
        if i == 0:
            continue
        acc = O.keyswitch_qp(level, c1, key_of_rotation(i))                       # KeyswitchHoistedNoModDown (one decomposition in the fork)
        rot[i] = (qp.permute(i, qp.add(acc[0], pc0)), qp.permute(i, acc[1]))       # phi(P c0 + d0), phi(d1)
    res = np.zeros((2, nl, N), dtype=np.uint64)
    B = [None, None]

    def acc_into(k, x):
        B[k] = x if B[k] is None else qp.add(B[k], x)

    for j in sorted(index):
        if j == 0:
            continue
        A = [None, None]
        for i in index[j]:
            if i == 0:
                continue
            for k in range(2):
                t = qp.mul(diags_qp[n1 * j + i], rot[i][k])
                A[k] = t if A[k] is None else qp.add(A[k], t)
        a = []
        for k in range(2):
            if A[k] is None:
                a.append(np.zeros((nl, N), dtype=np.uint64))
            else:
                out = O.mod_down(level, A[k])
                log.append(("ModDown", A[k][:nl], A[k][nl:], out))
                a.append(out)
        if 0 in index[j]:
            ptq = diags_qp[n1 * j]
            a = [qp.add(a[k], qp.mul(ptq, ct[k], nl), nl) for k in range(2)]
        log.append(("KeySwitchNoModDown", a[1]))
        e = O.keyswitch_qp(level, a[1], key_of_rotation(n1 * j))                  # SwitchKeysInPlaceNoModDown
        res[0] = qp.add(res[0], qp.permute(n1 * j, a[0]), nl)
        for k in range(2):
            acc_into(k, qp.permute(n1 * j, e[k]))
    for i in index.get(0, []):
        if i == 0:
            continue
        for k in range(2):
            acc_into(k, qp.mul(diags_qp[i], rot[i][k]))
    for k in range(2):
        if B[k] is not None:
            out = O.mod_down(level, B[k])
            log.append(("ModDown", B[k][:nl], B[k][nl:], out))
            res[k] = qp.add(res[k], out, nl)
    if 0 in index.get(0, []):
        for k in range(2):
            res[k] = qp.add(res[k], qp.mul(diags_qp[0], ct[k], nl), nl)
    return res, log
