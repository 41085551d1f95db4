#!/usr/bin/env python
"""A/B timing of the stride-2 3x3 layers of a config-3 step: direct stride-2 kernel (rfx_conv3x3_s2_f32) vs the implicit-GEMM
kernel (rfx_conv2d_f32), HIP-event timed, with a bit-identity check.   python scripts/ubench/conv_s2_bench.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ransac-flow_amd"))
import torch  # noqa: E402
from rfx import ops  # noqa: E402

SHAPES = [(64, 64, 240, 320, 128), (64, 128, 120, 160, 256), (64, 128, 240, 320, 128), (64, 256, 120, 160, 256),
          (64, 128, 200, 264, 128), (64, 256, 100, 132, 256), (64, 128, 60, 80, 128), (64, 256, 30, 40, 256)]


def main():
    dev = torch.device("cuda:0")
    rows = []
    for (N, Cin, H, W, Cout) in SHAPES:
        g = torch.Generator().manual_seed(Cin + Cout)
        x = torch.randn(N, Cin, H, W, generator=g).to(dev)
        w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
        bnd = dict(weight=torch.ones(Cout), bias=torch.zeros(Cout), running_mean=torch.zeros(Cout), running_var=torch.ones(Cout))
        plan = ops.ConvPlan(w, bnd, 2, 1, ops.ACT_RELU, dev)
        Ho, Wo = plan.out_hw(H, W)
        out_g = torch.empty((N, Cout, Ho, Wo), device=dev)

        def generic():
            ops._call("rfx_conv2d_f32", dev, ops._p(x), ops._p(plan.wT), ops._p(plan.ktab), ops._p(plan.scale), ops._p(plan.shift),
                      ops._p(None), ops._p(out_g), N, Cin, H, W, Cout, 3, 3, 2, 1, ops.ACT_RELU)
        res = {}
        for name, fn in (("direct_s2", lambda: plan(x)), ("implicit_gemm", generic)):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            res[name] = dict(ms=round(ms, 3), tflops=round(2.0 * N * Ho * Wo * Cout * Cin * 9 / ms / 1e9, 1))
        generic()
        same = bool(torch.equal(plan(x), out_g))
        row = dict(shape=[N, Cin, H, W, Cout], **res, bit_identical=same)
        rows.append(row)
        print(json.dumps(row), flush=True)
    if len(sys.argv) > 1:
        os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
        json.dump(rows, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
