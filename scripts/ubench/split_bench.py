#!/usr/bin/env python
"""rfx_conv1x1_split_f32 (float32 sums from exact bf16 operand pieces, csrc/conv1x1s.hip) against the fp32-MFMA kernel it can replace:
error of both against a float64 convolution on the device, and back-to-back launch times on the trunk's 1x1 shapes.
    python scripts/ubench/split_bench.py [--n 64] [--out gpurun_out/r06/split_bench.json]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ransac-flow_amd"))
import torch  # noqa: E402
from rfx import ops  # noqa: E402
from rfx.ops import ConvPlan, ACT_RELU  # noqa: E402

SHAPES3 = [  # 3x3 / stride 1 / pad 1: (Cin, Cout, H, W, residual)
    (256, 256, 60, 80, False), (512, 256, 60, 80, False), (256, 128, 60, 80, False), (64, 64, 240, 320, True), (128, 128, 120, 160, True),
    (256, 256, 30, 40, True), (256, 256, 50, 66, False), (256, 256, 25, 33, False), (64, 64, 200, 264, False), (49, 512, 60, 80, False),
]
SHAPES3S2 = [  # 3x3 / stride 2 / pad 1
    (128, 128, 240, 320, False), (256, 256, 120, 160, False), (64, 128, 240, 320, False), (128, 256, 120, 160, False), (128, 128, 100, 132, False),
]
SHAPES = [  # 1x1: (Cin, Cout, H, W, residual)
    (256, 1024, 60, 80, True), (1024, 256, 60, 80, False), (512, 128, 120, 160, False), (256, 64, 240, 320, False),
    (64, 256, 240, 320, False), (1024, 256, 50, 66, False), (256, 1024, 50, 66, True), (1024, 256, 25, 33, False), (256, 1024, 25, 33, True),
]


def timed(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--out", default=None)
    ap.add_argument("--k", type=int, default=0, help="3 / 1: only the 3x3 / 1x1 shapes")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    ap_k = [(k_, sh) for k_, sh in [(3, sh) for sh in SHAPES3] + [(32, sh) for sh in SHAPES3S2] + [(1, sh) for sh in SHAPES] if a.k in (0, k_)]
    rows = []
    for ksz, (Cin, Cout, H, W, has_res) in ap_k:
        stride = 2 if ksz == 32 else 1                  # 32: the 3x3 / stride 2 shapes
        ksz = 3 if ksz == 32 else ksz
        Ho, Wo = (H + 2 * (ksz // 2) - ksz) // stride + 1, (W + 2 * (ksz // 2) - ksz) // stride + 1
        g = torch.Generator().manual_seed(Cin + Cout + H)
        w = torch.randn(Cout, Cin, ksz, ksz, generator=g) * (2.0 / (Cout * ksz * ksz)) ** 0.5
        bn = dict(weight=1.0 + 0.2 * (torch.rand(Cout, generator=g) - 0.5), bias=0.1 * torch.randn(Cout, generator=g),
                  running_mean=0.1 * torch.randn(Cout, generator=g), running_var=1.0 + 0.4 * (torch.rand(Cout, generator=g) - 0.5))
        p32 = ConvPlan(w, bn, stride, ksz // 2, ACT_RELU, dev)
        psp = ConvPlan(w, bn, stride, ksz // 2, ACT_RELU, dev, split=True)
        assert psp.wS is not None
        x = torch.relu(torch.randn(a.n, Cin, H, W, generator=g)).to(dev)
        res = torch.randn(a.n, Cout, Ho, Wo, generator=g).to(dev) if has_res else None
        # float64 reference on 2 images
        xs, rs = x[:2], (res[:2] if has_res else None)
        s64 = torch.nn.functional.conv2d(xs.double(), w.double().to(dev), stride=stride, padding=ksz // 2)
        y64 = s64 * p32.scale.double().view(1, -1, 1, 1) + p32.shift.double().view(1, -1, 1, 1)
        if has_res:
            y64 = y64 + rs.double()
        y64 = torch.relu(y64)
        rms = float(y64.pow(2).mean().sqrt())
        e32 = (p32(xs, residual=rs).double() - y64)
        esp = (psp(xs, residual=rs).double() - y64)
        ms32 = timed(lambda: p32(x, residual=res), a.iters)
        mssp = timed(lambda: psp(x, residual=res), a.iters)
        fl = 2.0 * a.n * Ho * Wo * Cin * Cout * ksz * ksz
        row = dict(k=ksz, stride=stride, Cin=Cin, Cout=Cout, H=H, W=W, N=a.n, residual=has_res,
                   fp32_ms=round(ms32, 3), split_ms=round(mssp, 3), speedup=round(ms32 / mssp, 3),
                   fp32_tflops=round(fl / ms32 / 1e9, 1), split_tflops_equiv=round(fl / mssp / 1e9, 1),
                   fp32_rms_err=float(e32.pow(2).mean().sqrt()) / rms, split_rms_err=float(esp.pow(2).mean().sqrt()) / rms,
                   fp32_max_err=float(e32.abs().max()) / rms, split_max_err=float(esp.abs().max()) / rms)
        rows.append(row)
        print(json.dumps(row), flush=True)
    if a.out:
        json.dump(rows, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
