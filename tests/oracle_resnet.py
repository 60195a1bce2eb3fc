"""Oracle side of the `resnet` row (scope 8f-3): the reference's testResNet_crop_sparse (test.go:76-370) and the layer kinds
"Conv_sparse" / "StrConv_sparse" of evalConv_BNRelu_new (eval.go:272-607) on the pinned primitives, plus a plain numpy model
of the same network to compare against. TEST INFRASTRUCTURE.

Reference mapping:
  test.go:76-370      testResNet_crop_sparse            -> ResNetOracle.run
  eval.go:335-392     StrConv_sparse front end (two convolutions at norm/2, X^(norm/4) shift, add, offset monomial) -> layer()
  conv.go:374-414     ext_double_ctxt                   -> ext_double_ctxt
  rot_util.go:557-612 gen_comprs_sparse (log_sparse != 0) -> gen_comprs_sparse
  rot_util.go:179-218 gen_keep_vec_sparse               -> oracle_ckks.gen_keep_vec_sparse
  test.go:285-334     reduce-mean + FC as one convolution -> final_fc
The bootstrapper is oracle_ckks.Bootstrapper(log_sparse): my restatement of the fork-only BootstrappConv_CtoS/_StoC."""
import numpy as np

import oracle_ckks as ck
from oracle_lib import Oracle

Q0, Q1, P0 = ck.Q_SET6[0], ck.Q_SET6[1], ck.P_SET6[0]


def rev_bits(x, nbits):
    r = 0
    for b in range(nbits):
        r |= ((x >> b) & 1) << (nbits - 1 - b)
    return r


def gen_comprs_sparse(vec_size, in_wid, kp_wid, log_sparse, ul=0, pos=0):
    """rot_util.go:557-612, the log_sparse != 0 branch (the only one `resnet 3 20 ...` reaches): masks + rotations of the two
    stages of ext_double_ctxt that keep the stride-2 positions and re-pack them for the next (half-width) block"""
    assert log_sparse != 0 and pos == 0 and in_wid % 2 == 0
    m_idx, r_idx = {}, {}
    batch = 2 * vec_size // (in_wid * in_wid * (1 << log_sparse))
    min_wid = in_wid // 2
    log_in_wid = (in_wid - 1).bit_length()
    rep = 1 << (log_sparse - 1)

    def tile(tmp):
        seg = vec_size // rep
        for k in range(1, rep):
            tmp[k * seg: (k + 1) * seg] = tmp[:seg]
        return tmp
    for j in range(min_wid):
        tmp = np.zeros(vec_size, dtype=np.int64)
        for b in range(batch):
            for i in range(min_wid // 2):
                for k in range(2):
                    if rev_bits(j, log_in_wid - 1) < kp_wid and rev_bits(i, log_in_wid - 2) + k * min_wid // 2 < kp_wid:
                        tmp[k * in_wid * min_wid * batch + in_wid * in_wid * b // 2 + in_wid * j // 2 + i] = 1
        m_idx[j * min_wid // 2] = tile(tmp)
    for b in range(batch):
        tmp = np.zeros(vec_size, dtype=np.int64)
        for j in range(min_wid):
            for i in range(min_wid // 2):
                for k in range(2):
                    tmp[k * in_wid * min_wid * batch + b * in_wid * in_wid // 2 + j * min_wid // 2 + i] = 1
        r_idx[3 * b * min_wid * min_wid // 2] = tile(tmp)
    return m_idx, r_idx


def ext_double_ctxt(C, ct, m_idx, r_idx):
    """conv.go:374-414: sum_rot Rotate(ct * mask, rot) twice (masks at scale sqrt(q_level)), one rescale"""
    L = ct.level
    sq = float(C.Q[L]) ** 0.5
    mid = None
    for rot, m in m_idx.items():
        t = C.rotate(C.mul_plain(ct, C.encode_ntt(m.astype(np.complex128), L, sq), sq), rot)
        mid = t if mid is None else C.add(mid, t)
    res = None
    for rot, m in r_idx.items():
        t = C.rotate(C.mul_plain(mid, C.encode_ntt(m.astype(np.complex128), L, sq), sq), rot)
        res = t if res is None else C.add(res, t)
    return C.rescale(res)


def strconv_relu_tail_sparse(C, btp, ct_conv, pow_, in_wid, kp_next, stages=None):
    """eval.go:437-565 for kind "StrConv_sparse" after the two half convolutions were joined (eval.go:361-387): bootstrapping with
    log_sparse = btp.ls, ReLU, MulByPow2, then ext_double_ctxt (conv.go:374-414) with the gen_comprs_sparse masks (rot_util.go:557-612)
    that keep the stride-2 positions and re-pack them for the next, half-width block, SlotsToCoeffs. Returns level 1, scale 2^30."""
    (boot,) = btp.ctos(ck.Ct(ct_conv.rows, ct_conv.scale * 2.0 ** pow_))
    if stages is not None:
        stages["ctos"] = [boot.copy()]
    r = C.mul_const_int(ck.eval_relu(C, boot, 0.0), 1 << int(pow_))
    if stages is not None:
        stages["relu"] = [r.copy()]
    m_idx, r_idx = gen_comprs_sparse(C.N // 2, in_wid, kp_next, btp.ls)
    ext = ext_double_ctxt(C, r, m_idx, r_idx)
    if stages is not None:
        stages["ext"] = [ext.copy()]
    return btp.stoc(ext, None)


# ---------------------------------------------------------------- plain model
def plain_conv_same(x, ker):
    """x (H, W, Cin), ker (k, k, Cin, Cout): zero-padded 'same' correlation"""
    H, Wd, _ = x.shape
    k = ker.shape[0]
    p = k // 2
    xp = np.zeros((H + 2 * p, Wd + 2 * p, x.shape[2]))
    xp[p:p + H, p:p + Wd] = x
    out = np.zeros((H, Wd, ker.shape[3]))
    for di in range(k):
        for dj in range(k):
            out += xp[di:di + H, dj:dj + Wd] @ ker[di, dj]
    return out


class Net:
    """shapes of testResNet_crop_sparse for a ring of 2^logN (logN = 16 is the reference's network; smaller rings scale the
    widths down and keep batch x norm = N / width^2)"""

    def __init__(self, logN, ker_wid=3, depth=8, fc_out=10, seed=0, wide=1):
        self.logN, self.k, self.fc_out, self.wide = logN, ker_wid, fc_out, wide
        shapes = {16: ((32, 16, 8), (16, 32, 64)), 14: ((16, 8, 4), (16, 32, 64)), 12: ((16, 8, 4), (4, 8, 16))}
        self.in_wids, self.real_batch = [list(t) for t in shapes[logN]]
        if wide in (2, 3):  # testResNet_crop_sparse_wide (test.go:681-691): wide_case 2 = 32/64/128 channels, 3 = 48/96/192; first layer 3 -> 16 -> real_batch[0]
            assert logN == 16
            self.real_batch = [32, 64, 128] if wide == 2 else [48, 96, 192]
        self.raw = [w - ker_wid // 2 for w in self.in_wids]
        self.max_batch = [(1 << logN) // (w * w) for w in self.in_wids]
        self.norm = [m // r for m, r in zip(self.max_batch, self.real_batch)]      # wide 3: 64 // 48 = 1, 256 // 96 = 2, 1024 // 192 = 4 (test.go:688)
        self.blocks = {20: (7, 5, 5), 14: (5, 3, 3), 8: (3, 1, 1)}[depth]
        rng = np.random.default_rng(seed)
        k = ker_wid
        self.layers = []        # (kind, block index of the INPUT, weights (k,k,cin,cout), bn_a, bn_b)
        cin = 3
        for blk in range(3):
            if blk > 0:
                cout = self.real_batch[blk]
                self.layers.append(("StrConv_sparse", blk - 1, self._w(rng, k, cin, cout), rng.uniform(0.8, 1.2, cout), rng.uniform(-0.1, 0.1, cout)))
                cin = cout
            for li in range(self.blocks[blk]):
                cout = 16 if (wide in (2, 3) and blk == 0 and li == 0) else self.real_batch[blk]      # init_batch = 16 (test.go:667)
                self.layers.append(("Conv_sparse", blk, self._w(rng, k, cin, cout), rng.uniform(0.8, 1.2, cout), rng.uniform(-0.1, 0.1, cout)))
                cin = cout
        self.fc_w = rng.uniform(-1, 1, (cin, fc_out)) / np.sqrt(cin)
        self.fc_b = rng.uniform(-0.1, 0.1, fc_out)
        self.image = rng.uniform(-1, 1, (self.raw[0], self.raw[0], 3))

    @staticmethod
    def _w(rng, k, cin, cout):
        return rng.uniform(-1, 1, (k, k, cin, cout)) * (1.5 / np.sqrt(k * k * cin))

    def plain(self, upto=None):
        """float model: conv 'same' on the raw window, BN, ReLU; stride-2 layers keep even positions of the raw window"""
        x = self.image
        acts = []
        for li, (kind, blk, w, a, b) in enumerate(self.layers):
            y = plain_conv_same(x, w) * a + b
            if kind == "StrConv_sparse":
                r = self.raw[blk + 1]
                y = y[0:2 * r:2, 0:2 * r:2]
            x = np.maximum(y, 0)
            acts.append(x)
            if upto is not None and li == upto:
                break
        if upto is not None:
            return acts
        pooled = x.mean(axis=(0, 1))
        return acts, pooled @ self.fc_w + self.fc_b


class ResNetOracle:
    def __init__(self, net, seed=1, Ckks=None, conv_oracle=None):
        self.net = net
        self.C = Ckks if Ckks is not None else ck.Ckks(logN=net.logN, seed=seed, h=192 if net.logN >= 14 else 64)
        self.Oc = conv_oracle if conv_oracle is not None else Oracle(logN=net.logN, q=[Q0, Q1], p=[P0])
        self.N = 1 << net.logN
        self.evk = np.zeros((net.logN, 4, self.N), dtype=np.uint64)
        self.have = set()
        self.idx_pt = None
        self.btp = {}
        self.pow = 6.0 if net.logN >= 16 else 4.0

    def bootstrapper(self, ls):
        if ls not in self.btp:
            self.btp[ls] = ck.Bootstrapper(self.C, log_sparse=ls)
        return self.btp[ls]

    # ---- evalConv_BN (eval.go:224-263) on the conv oracle
    def conv_bn(self, ct, ker_in, bn_a, bn_b, in_wid, ker_wid, real_ib, real_ob, norm, out_scale):
        O, N, logN = self.Oc, self.N, self.net.logN
        max_bat = N // (in_wid * in_wid)
        kc = O.prep_ker_coeffs(np.ascontiguousarray(ker_in).reshape(-1), bn_a, in_wid, ker_wid, real_ib, real_ob, norm)
        pl_ker = np.zeros((max_bat, 2, N), dtype=np.uint64)
        for i in range(0, max_bat, norm):
            e = O.encode_coeffs(kc[i], 2.0 ** 30, [0, 1])
            pl_ker[i, 0], pl_ker[i, 1] = O.ntt(0, e[0]), O.ntt(1, e[1])
        bias_pt = O.ntt(0, O.encode_coeffs(O.bias_coeffs(bn_b, in_wid, norm), out_scale, [0])[0])
        step = max_bat // 2
        j = logN - (step.bit_length() - 1)
        while step >= norm and step >= 1:
            if j not in self.have:
                self.evk[j - 1] = O.gen_galois_key_l0(self.C.sk, (1 << j) + 1, 100 + j)
                self.have.add(j)
            step //= 2
            j += 1
        if self.idx_pt is None:
            self.idx_pt = np.stack([O.ntt(0, np.eye(1, N, 1 << s, dtype=np.uint64)[0]) for s in range(logN)])
        got, sc = O.conv_then_pack(np.ascontiguousarray(ct.rows[:, :2]), ct.scale, pl_ker, 2.0 ** 30, self.idx_pt, self.evk, max_bat, norm, out_scale, bias_pt)
        return ck.Ct(got.reshape(2, 1, N).copy(), sc)

    def mul_monomial_l0(self, ct, idx, sign=1):
        """MulNew(ct, EncodeCoeffs(X^idx at scale 1)) at level 0 (eval.go:361-367, 374-387)"""
        m = np.zeros(self.N, dtype=np.uint64)
        m[idx] = 1 if sign > 0 else Q0 - 1
        pt = self.Oc.ntt(0, m)
        return ck.Ct(np.stack([self.Oc.mul(0, ct.rows[d, 0], pt).reshape(1, -1) for d in range(2)]), ct.scale)

    def layer(self, ct, kind, blk, w, bn_a, bn_b, pow_, stages=None):
        """evalConv_BNRelu_new for kinds Conv_sparse / StrConv_sparse; ct: level >= 1, scale 2^30; returns level 1, scale 2^30"""
        net, C = self.net, self.C
        k, in_wid = net.k, net.in_wids[blk]
        cin, cout = w.shape[2], w.shape[3]
        out_scale = 2.0 ** (round(np.log2(float(Q0))) - (pow_ + 8))
        ker_in = np.ascontiguousarray(w).reshape(-1)          # HWIO flat = ker_in[t*cin*cout + i*cout + o]
        if kind == "Conv_sparse":
            ls = {0: 2, 1: 3, 2: 4}[blk]
            assert (1 << ls) == net.norm[blk], (ls, net.norm)
            ct_conv = self.conv_bn(ct, ker_in, bn_a, bn_b, in_wid, k, cin, cout, net.norm[blk], out_scale)
            if stages is not None:
                stages["conv"] = ct_conv
            return ck.conv_relu_tail_sparse(C, self.bootstrapper(ls), ct_conv, 0.0, pow_, in_wid, net.raw[blk], stages=stages)
        assert kind == "StrConv_sparse"
        ls = {0: 1, 1: 2}[blk]
        norm = net.norm[blk + 1]                               # eval.go is called with the NEXT block's norm (test.go:200)
        kk = w.reshape(k * k, cin, cout)
        halves = []
        for par in range(2):                                   # eval.go:336-359: even / odd output channels, norm/2
            halves.append(self.conv_bn(ct, np.ascontiguousarray(kk[:, :, par::2]).reshape(-1), bn_a[par::2], bn_b[par::2], in_wid, k, cin, cout // 2, norm // 2, out_scale))
        ct_conv = C.add(halves[0], self.mul_monomial_l0(halves[1], norm // 4))        # eval.go:361-369
        max_batch = self.N // (in_wid * in_wid)
        if (in_wid - k // 2) % 2 == 0:                         # eval.go:377-387
            ct_conv = self.mul_monomial_l0(ct_conv, self.N - max_batch * (in_wid + 1), -1)
        if stages is not None:
            stages["conv"] = ct_conv
        return strconv_relu_tail_sparse(C, self.bootstrapper(ls), ct_conv, pow_, in_wid, net.raw[blk + 1], stages=stages)

    # ---- layouts
    def pack_input(self, image):
        """test.go:138-150: sparse pack of the 3-channel image into block 1's layout"""
        net = self.net
        W, mb, nm, raw = net.in_wids[0], net.max_batch[0], net.norm[0], net.raw[0]
        cf = np.zeros(self.N)
        for i in range(raw):
            for j in range(raw):
                for b in range(3):
                    cf[i * W * mb + j * mb + b * nm] = image[i, j, b]
        return cf

    def unpack(self, cf, blk, channels):
        """coefficients -> (raw, raw, channels) activation of block blk (prt_mat_norm's view, main.go:760-)"""
        net = self.net
        W, mb, nm, raw = net.in_wids[blk], net.max_batch[blk], net.norm[blk], net.raw[blk]
        out = np.zeros((raw, raw, channels))
        for i in range(raw):
            for j in range(raw):
                out[i, j] = cf[i * W * mb + j * mb: i * W * mb + j * mb + channels * nm: nm]
        return out

    def final_fc(self, ct):
        """test.go:285-334 (cifar10 branch): reduce-mean over the raw window + FC as ONE convolution whose taps all equal the
        FC weights, BN scale 1/raw^2, bias = FC bias; the class scores sit at the centre pixel"""
        net = self.net
        raw, W = net.raw[2], net.in_wids[2]
        kw = raw if raw % 2 else raw + 1
        cin = net.real_batch[2]
        ker = np.broadcast_to(net.fc_w.reshape(1, cin * net.fc_out), (kw * kw, cin * net.fc_out)).reshape(-1).copy()
        bn_a = np.full(net.fc_out, 1.0 / (raw * raw))
        ct_res = self.conv_bn(ct, ker, bn_a, net.fc_b, W, kw, cin, net.fc_out, net.norm[2], 2.0 ** 30)
        cf = self.Oc.decrypt_decode_l0(self.C.sk, ct_res.rows[:, 0], ct_res.scale)
        c = kw // 2
        mb, nm = net.max_batch[2], net.norm[2]
        return np.array([cf[c * W * mb + c * mb + o * nm] for o in range(net.fc_out)])

    def run(self, verbose=False):
        """the whole encrypted inference; returns (class scores, per-layer max abs error vs the plain model)"""
        net, C = self.net, self.C
        acts, scores = net.plain()
        ct = C.encrypt_coeffs(self.pack_input(net.image), 1, 2.0 ** 30, seed=5)
        errs = []
        for li, (kind, blk, w, a, b) in enumerate(net.layers):
            ct = self.layer(ct, kind, blk, w, a, b, self.pow)
            oblk = blk + 1 if kind == "StrConv_sparse" else blk
            errs.append(float(np.max(np.abs(self.unpack(C.decrypt_coeffs(ct), oblk, w.shape[3]) - acts[li]))))
            if verbose:
                print(li, kind, "max err", errs[-1], flush=True)
        return self.final_fc(ct), scores, errs
