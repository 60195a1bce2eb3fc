#!/usr/bin/env python3
"""Synthetic inputs for `resnet <ker> <depth> <wide_case 1|2> <n> false` in the reference's file layout (test.go:78-80,128,171-183,285,329):
  Resnet_weights/weights_crop_ker{k}_d{depth}_wid1/w{i}-conv.csv (HWIO flat), w{i}-a.csv, w{i}-b.csv, final-fckernel.csv, final-fcbias.csv
  Resnet_plain_data/crop_ker{k}_d{depth}_wid1/test_image_{iter}.csv  (32 x 32 x 3, HWC; only the raw window is read)
plus what the plain float model of the same network outputs: Resnet_plain_data/.../expected_scores_{iter}.csv.
The reference ships neither weights nor images (README.md:23); these are random (seeded) weights of the right shapes."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle_resnet as rn  # noqa: E402


def write_case(root, ker_wid=3, depth=8, n_images=1, seed=0, cf100=False, wide=1, native_image=False):
    """native_image: image 0 is the one rn.Net draws itself (what tests/golden/gen_resnet_digests.py encrypts), not default_rng(1000)'s"""
    net = rn.Net(16, ker_wid=ker_wid, depth=depth, seed=seed, fc_out=100 if cf100 else 10, wide=wide)
    tag = ("cf100_" if cf100 else "") + f"crop_ker{ker_wid}_d{depth}_wid{wide}"
    wdir, pdir = os.path.join(root, "Resnet_weights", "weights_" + tag), os.path.join(root, "Resnet_plain_data", tag)
    os.makedirs(wdir, exist_ok=True)
    os.makedirs(pdir, exist_ok=True)
    for i, (_, _, w, a, b) in enumerate(net.layers):
        np.savetxt(os.path.join(wdir, f"w{i}-conv.csv"), w.reshape(-1), fmt="%.17g")
        np.savetxt(os.path.join(wdir, f"w{i}-a.csv"), a, fmt="%.17g")
        np.savetxt(os.path.join(wdir, f"w{i}-b.csv"), b, fmt="%.17g")
    np.savetxt(os.path.join(wdir, "final-fckernel.csv"), net.fc_w.reshape(-1), fmt="%.17g")
    np.savetxt(os.path.join(wdir, "final-fcbias.csv"), net.fc_b, fmt="%.17g")
    scores = []
    for it in range(n_images):
        if not (native_image and it == 0):
            net.image = np.random.default_rng(1000 + it).uniform(-1, 1, net.image.shape)
        W, raw = net.in_wids[0], net.raw[0]
        full = np.zeros((W, W, 3))
        full[:raw, :raw] = net.image
        np.savetxt(os.path.join(pdir, f"test_image_{it}.csv"), full.reshape(-1), fmt="%.17g")
        acts, sc = net.plain()
        np.savetxt(os.path.join(pdir, f"expected_scores_{it}.csv"), sc, fmt="%.17g")
        scores.append((sc, max(float(np.abs(a).max()) for a in acts)))
    return scores


if __name__ == "__main__":
    root, k, d, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    cf100 = len(sys.argv) > 5 and sys.argv[5] in ("true", "1")
    wide = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    for sc, amax in write_case(root, k, d, n, cf100=cf100, wide=wide):
        print("scores", np.round(sc, 4), "max |activation|", amax)
