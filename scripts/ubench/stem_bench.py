#!/usr/bin/env python
"""Micro-benchmark of the two fused stem kernels on the config-3 shapes (64 images per pyramid level), back-to-back launches between
one HIP event pair; checks every library variant (RFX_LIB=...) bit for bit against the un-fused convolution + pooling ops.
    python scripts/ubench/stem_bench.py [--n 64] [--out gpurun_out/r06/stem_bench.json]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ransac-flow_amd"))
import torch  # noqa: E402
from rfx import ops, weights  # noqa: E402
from rfx.ops import ConvPlan, ACT_RELU  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    sd = weights.resnet50_trunk_sd(0, randomize_bn=True)
    plan = ConvPlan(sd["conv1.weight"], {k: sd["bn1." + k] for k in ("weight", "bias", "running_mean", "running_var")}, 2, 3, ACT_RELU, dev)
    rows = []
    for (H, W) in ((960, 1280), (800, 1056), (640, 848), (480, 640), (400, 528), (320, 416), (240, 320)):
        g = torch.Generator(device=dev).manual_seed(H)
        x = torch.randn(a.n, 3, H, W, device=dev, generator=g)
        ref = ops.maxpool2d(plan(x[:2]), 3, 2, 1)
        out = ops.stem_conv7_maxpool(x[:2], plan)
        same = bool(torch.equal(out, ref))
        for _ in range(3):
            ops.stem_conv7_maxpool(x, plan)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            ops.stem_conv7_maxpool(x, plan)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        Hc, Wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        tf = 2.0 * a.n * Hc * Wc * 64 * 147 / ms / 1e9
        row = dict(kernel="stem7", N=a.n, H=H, W=W, ms=round(ms, 3), tflops=round(tf, 1), frac=round(tf / 157.3, 3), bit_identical=same,
                   lib=os.environ.get("RFX_LIB", "librfx.so"))
        rows.append(row)
        print(json.dumps(row), flush=True)
    # the FeatureExtractor stem (conv3x3 3 -> 64 + BN + ReLU + MaxPool(2, 1) + BlurPool/2) on the fine-pass shapes of config 3
    sdf = weights.feature_extractor_sd(1, randomize_bn=True)
    plan3 = ConvPlan(sdf["conv1.weight"], {k: sdf["bn1." + k] for k in ("weight", "bias", "running_mean", "running_var")}, 1, 1, ACT_RELU, dev)
    for (H, W) in ((480, 640), (240, 320)):
        g = torch.Generator(device=dev).manual_seed(H + 1)
        x = torch.randn(a.n, 3, H, W, device=dev, generator=g)
        same = bool(torch.equal(ops.stem_conv_maxblur(x[:2], plan3), ops.maxblurpool2d(plan3(x[:2]), 2)))
        for _ in range(3):
            ops.stem_conv_maxblur(x, plan3)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            ops.stem_conv_maxblur(x, plan3)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        tf = 2.0 * a.n * H * W * 64 * 27 / ms / 1e9
        row = dict(kernel="stem3", N=a.n, H=H, W=W, ms=round(ms, 3), tflops=round(tf, 1), frac=round(tf / 157.3, 3), bit_identical=same,
                   lib=os.environ.get("RFX_LIB", "librfx.so"))
        rows.append(row)
        print(json.dumps(row), flush=True)
    if a.out:
        json.dump(rows, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
