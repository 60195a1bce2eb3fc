#!/usr/bin/env python3
"""Synthetic test_conv_data/*.csv generator (the reference ships none: README.md:21-24).

File names and sizes follow the reference's readers (test.go:37-40,66-68; main.go:971-990):
  test_conv{k}_batch_{B}_{in|ker|bna|bnb|out|reluout}_{iter}.csv, whitespace separated floats
  (reluout = max(out, 0), what `convReLU` compares against: test.go:66).
Layouts: input HWC flat  in[(i*raw+j)*B+b]      (main.go:1007-1042 prep_Input)
         kernel HWIO flat ker[o + c*B + t*B*B]  (conv.go:184-202 reshape_ker)
Expected output = zero-padded 'same' correlation on the raw x raw image, times bn_a plus bn_b.
Seeds: numpy default_rng(1000*k + 10*i_batch + iter)  (BASELINE.md section 2).
"""
import argparse
import os
import numpy as np

BATCHS = [4, 16, 64, 256, 1024]   # main.go:578
WIDTHS = [128, 64, 32, 16, 8]     # main.go:579


def make_case(k, i_batch, it):
    B, W = BATCHS[i_batch], WIDTHS[i_batch]
    raw = W - k // 2
    rng = np.random.default_rng(1000 * k + 10 * i_batch + it)
    x = rng.uniform(-1, 1, size=(raw, raw, B))
    ker = rng.uniform(-1, 1, size=(k, k, B, B)) / np.sqrt(k * k * B)
    a = rng.uniform(0.5, 1.5, size=B)
    b = rng.uniform(-0.5, 0.5, size=B)
    return B, W, raw, x, ker, a, b


def plain_conv(x, ker, a, b):
    """'same' zero-padded correlation, HWC x HWIO -> HWO, then *a + b."""
    raw, _, B = x.shape
    k = ker.shape[0]
    p = k // 2
    xp = np.zeros((raw + 2 * p, raw + 2 * p, B))
    xp[p:p + raw, p:p + raw, :] = x
    out = np.zeros((raw, raw, ker.shape[3]))
    for di in range(k):
        for dj in range(k):
            out += xp[di:di + raw, dj:dj + raw, :] @ ker[di, dj]
    return out * a + b


def write_case(outdir, k, i_batch, it):
    B, W, raw, x, ker, a, b = make_case(k, i_batch, it)
    out = plain_conv(x, ker, a, b)
    os.makedirs(outdir, exist_ok=True)
    pre = os.path.join(outdir, f"test_conv{k}_batch_{B}_")
    for name, arr in (("in", x), ("ker", ker), ("bna", a), ("bnb", b), ("out", out), ("reluout", np.maximum(out, 0.0))):
        np.savetxt(f"{pre}{name}_{it}.csv", arr.reshape(-1), fmt="%.17g")
    return B, W, raw


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("outdir")
    ap.add_argument("k", type=int)
    ap.add_argument("i_batch", type=int)
    ap.add_argument("n", type=int)
    args = ap.parse_args()
    os.makedirs(args.outdir, exist_ok=True)
    for it in range(args.n):
        print(write_case(args.outdir, args.k, args.i_batch, it))
