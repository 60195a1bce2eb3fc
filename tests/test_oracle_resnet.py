"""The `resnet` row (scope 8f-3) on the oracle: testResNet_crop_sparse (test.go:76-370) with depth 8 on a 2^12 ring (same
modulus chain, widths 16/8/4, channels 4/8/16, sparse packing norms 4/8/16): three "Conv_sparse" layers, a "StrConv_sparse"
down-sampling layer (two half convolutions, ext_double_ctxt compression with gen_comprs_sparse masks), ... and the final
reduce-mean + FC convolution, against a plain numpy model of the same network."""
import numpy as np

import oracle_resnet as rn


def test_comprs_masks_shape():
    m_idx, r_idx = rn.gen_comprs_sparse(2048, 16, 7, 1)
    assert sorted(m_idx) == [j * 4 for j in range(8)] and all(v.shape == (2048,) for v in m_idx.values())
    assert len(r_idx) == 2 * 2048 // (16 * 16 * 2) and 0 in r_idx
    # second-stage masks partition the compressed positions: disjoint supports
    tot = sum(r_idx.values())
    assert tot.max() == 1


def test_encrypted_resnet_depth8_small_ring():
    net = rn.Net(12, depth=8)
    got, want, errs = rn.ResNetOracle(net).run()
    assert max(errs) < 0.08, errs                    # activations stay within the ReLU approximation error, layer after layer
    assert got.argmax() == want.argmax() and np.max(np.abs(got - want)) < 0.03
