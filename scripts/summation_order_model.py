#!/usr/bin/env python
"""EVIDENCE SCRIPT -- why the device's trunk features carry ~2x the round-off of the CPU executors (profiles/r04_feature_error_*).

The fp32 MFMA accumulates a convolution's K = Cin*KH*KW products as ONE k-ordered fma chain per output element (bit-identical to
the CPU's "native" path where K is short: the stem, K = 147, matches it bit for bit).  MKL's sgemm and oneDNN's blocked
convolutions accumulate K in blocks of a few hundred k (register block), adding each block's sum to the running total.  For a
random-walk partial sum the chain's round-off grows like eps*K/sqrt(2), the blocked form's like eps*K/sqrt(2*n_blocks): this
script measures both on synthetic layer-3-shaped dot products (post-ReLU activations x zero-mean weights, K = 256 ... 2304)
against float64 and prints the ratio -- the factor the stage-wise measurement shows between "device" and "onednn"/"native".

Second part (round 4, after the trunk was chunked and the flip counts did NOT follow): the same comparison for the mutual-NN score
itself, a sum of 1024 non-negative products -- the chain's round-off there (8e-8) is 2.8x the CPU product's and twice what the
trunk-feature error induces in a score, i.e. the score accumulation, not the trunk, decided most of the excess flips.

    python scripts/summation_order_model.py [--out profiles/r04_summation_order_model.json]
"""
import argparse
import json

import numpy as np


def chain(p):                       # sequential float32 accumulation, one rounding per term (fma: the product is exact)
    s = np.zeros(p.shape[0], dtype=np.float32)
    for k in range(p.shape[1]):
        s = (s.astype(np.float64) + p[:, k]).astype(np.float32)
    return s


def blocked(p, kc):
    tot = np.zeros(p.shape[0], dtype=np.float32)
    for k0 in range(0, p.shape[1], kc):
        tot = (tot + chain(p[:, k0:k0 + kc])).astype(np.float32)
    return tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--n", type=int, default=20000)
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    rows = []
    for K in (147, 256, 576, 1024, 1152, 2304):
        x = np.maximum(rng.standard_normal((a.n, K)), 0).astype(np.float32)                  # post-ReLU activations
        w = (rng.standard_normal((a.n, K)) * np.sqrt(2.0 / K)).astype(np.float32)           # kaiming-scaled weights
        p = x.astype(np.float64) * w.astype(np.float64)                                      # exact products (the fma's view)
        ref = p.sum(axis=1)
        e_chain = chain(p).astype(np.float64) - ref
        row = dict(K=K, rms_signal=float(np.sqrt((ref ** 2).mean())), rms_err_chain=float(np.sqrt((e_chain ** 2).mean())))
        for kc in (128, 256, 384):
            e = blocked(p, kc).astype(np.float64) - ref
            row["rms_err_blocked_%d" % kc] = float(np.sqrt((e ** 2).mean()))
            row["chain_over_blocked_%d" % kc] = round(row["rms_err_chain"] / row["rms_err_blocked_%d" % kc], 2)
        rows.append(row)
        print(row)
    # ---- the mutual-NN scores (utils/outil.py:34): sums of C = 1024 NON-NEGATIVE products of L2-normalised post-ReLU features.
    # The partial sum of a chain grows monotonically towards the score, so every later rounding is relative to a value of the
    # order of the score itself; with blocks of 256 each block sum is ~1/4 of it.  Compared: the chain, blocks of 32 ... 512,
    # and the CPU's own float32 product (torch.mm -- what the reference executes); also the score error that a trunk-feature
    # error of the measured size (profiles/r04_feature_error_*: 2.0e-8 ... 4.2e-8 per element) induces, for scale.
    import torch
    torch.manual_seed(0)
    C, nA, nB = 1024, 1500, 600
    A = torch.relu(torch.randn(C, nA)); A = A / A.norm(dim=0, keepdim=True)
    B = torch.relu(A[:, torch.randint(nA, (nB,))] + 0.5 * torch.randn(C, nB)); B = B / B.norm(dim=0, keepdim=True)
    ref = (A.double().t() @ B.double()).numpy()
    mm = (A.t() @ B).numpy().astype(np.float64)
    a64, b64 = A.numpy().astype(np.float64), B.numpy().astype(np.float64)

    def scores(kc):
        tot = np.zeros((nA, nB), np.float32); acc = np.zeros((nA, nB), np.float32)
        for k in range(C):
            acc = (acc.astype(np.float64) + np.outer(a64[k], b64[k])).astype(np.float32)
            if kc and ((k + 1) % kc == 0 or k + 1 == C):
                tot = (tot.astype(np.float64) + acc).astype(np.float32); acc[:] = 0
        return (tot if kc else acc).astype(np.float64)
    rms = lambda x, y: float(np.sqrt(((x - y) ** 2).mean()))
    sc = dict(C=C, nA=nA, nB=nB, mean_score=float(ref.mean()), rms_err_torch_mm=rms(mm, ref))
    for kc in (0, 512, 256, 128, 32):
        x = scores(kc)
        sc["rms_err_%s" % ("chain" if kc == 0 else "blocked_%d" % kc)] = rms(x, ref)
        sc["rms_distance_to_torch_mm_%s" % ("chain" if kc == 0 else "blocked_%d" % kc)] = rms(x, mm)
    for d in (2.0e-8, 2.8e-8, 4.2e-8):        # per-element feature error -> score error (both operands perturbed independently)
        dA = torch.randn(C, nA, dtype=torch.float64) * d; dB = torch.randn(C, nB, dtype=torch.float64) * d
        pert = ((A.double() + dA).t() @ (B.double() + dB)).numpy()
        sc["rms_score_error_from_feature_error_%.1e" % d] = rms(pert, ref)
    print(sc)
    out = dict(scores=sc, note="sequential fp32 accumulation (the MFMA's k-ordered chain) vs K-blocked accumulation (MKL / oneDNN register blocks) "
                    "of the same exact products, error vs float64", samples=a.n, rows=rows)
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
