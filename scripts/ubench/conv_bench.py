#!/usr/bin/env python
"""Micro-benchmark of single convolutions of the trunk (rfx_conv2d_f32 / rfx_conv3x3_conv1x1_f32) on the bench's own
shapes, HIP-event timed on the launch stream with rotating inputs.  One line per shape: avg us, algorithmic TFLOP/s,
fraction of the 157.3 TFLOP/s fp32 MFMA peak, and a checksum of the output (variants of the library that must stay
bit-identical are compared through it).

    python scripts/ubench/conv_bench.py [--shapes name ...] [--iters 20] [--out file.json]
    RFX_LIB=ransac-flow_amd/librfx_c3dbg1.so python scripts/ubench/conv_bench.py     # experiments (make c3dbg1)
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ransac-flow_amd"))
import torch  # noqa: E402
from rfx import ops  # noqa: E402

# name: (N, Cin, H, W, Cout, k, stride, Cexp)   Cexp > 0: fused Bottleneck tail (3x3 Cin->Cout, then 1x1 Cout->Cexp + residual)
SHAPES = {
    "fe64_240x320": (128, 64, 240, 320, 64, 3, 1, 0),        # FeatureExtractor BasicBlock convs (model/model.py:32-35)
    "c128_120x160": (128, 128, 120, 160, 128, 3, 1, 0),
    "l3_256_60x80": (128, 256, 60, 80, 256, 3, 1, 0),
    "l3_256_30x40": (128, 256, 30, 40, 256, 3, 1, 0),        # ResNet layer3 conv2 (model/resnet50.py:75)
    "l3_256_25x33": (64, 256, 25, 33, 256, 3, 1, 0),
    "l3_256_34x45": (64, 256, 34, 45, 256, 3, 1, 0),
    "tail64_120x160": (128, 64, 120, 160, 64, 3, 1, 256),    # layer1 conv2 + conv3 fused
    "tail128_60x80": (128, 128, 60, 80, 128, 3, 1, 512),     # layer2 conv2 + conv3 fused
    "tail64_100x132": (64, 64, 100, 132, 64, 3, 1, 256),     # ... at the small pyramid levels
    "tail64_112x148": (64, 64, 112, 148, 64, 3, 1, 256),
    "tail128_50x66": (64, 128, 50, 66, 128, 3, 1, 512),
    "tail128_62x84": (64, 128, 62, 84, 128, 3, 1, 512),
    "l3_256_26x35": (64, 256, 26, 35, 256, 3, 1, 0),
    "l3_256_28x37": (64, 256, 28, 37, 256, 3, 1, 0),
    "l3_256_31x42": (64, 256, 31, 42, 256, 3, 1, 0),
    "pw256_1024_30x40": (128, 256, 30, 40, 1024, 1, 1, 0),   # layer3 conv3 (without its residual)
    "pw256_1024_30x40_res": (128, 256, 30, 40, 1024, 1, 1, 0, True),   # layer3 conv3 + bn3 + residual + relu, as in the trunk
    "pw256_1024_36x48_res": (64, 256, 36, 48, 1024, 1, 1, 0, True),
    "pw256_1024_34x45_res": (64, 256, 34, 45, 1024, 1, 1, 0, True),
    "pw1024_256_30x40": (128, 1024, 30, 40, 256, 1, 1, 0),   # layer3 conv1
    "pw256_1024_34x45": (64, 256, 34, 45, 1024, 1, 1, 0),    # odd plane (scalar pixel path)
    "pw512_128_60x80": (128, 512, 60, 80, 128, 1, 1, 0),
    "s2_64_128_240x320": (128, 64, 240, 320, 128, 3, 2, 0),
    "s2_128_128_120x160": (128, 128, 120, 160, 128, 3, 2, 0),  # layer2.0 conv2 (stride 2): generic implicit-GEMM kernel
    "ds_256_512_s2_120x160": (128, 256, 120, 160, 512, 1, 2, 0),   # layer2.0 downsample (1x1 stride 2)
    "head49_512_60x80": (64, 49, 60, 80, 512, 3, 1, 0),         # NetFlowCoarse conv1 (Cin = 49)
    "stem7_960x1280": (64, 3, 960, 1280, 64, 7, 2, 0),          # ResNet stem fused with its max-pool, config 3's scale-2 level
    "stem7_480x640": (128, 3, 480, 640, 64, 7, 2, 0),
    "tail64_240x320": (64, 64, 240, 320, 64, 3, 1, 256),        # config 3, scale-2 level: the largest fused tail of the step
    "pw256_1024_60x80_res": (64, 256, 60, 80, 1024, 1, 1, 0, True),   # config 3, scale-2 level: layer3 conv3 + residual
    "pw64_256_240x320_res": (64, 64, 240, 320, 256, 1, 1, 0, True),   # layer1 conv3 un-fused (HBM-bound: 14 FLOP/B)
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", nargs="+", default=list(SHAPES))
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--sets", type=int, default=2)
    ap.add_argument("--out", type=str, default=None)
    ap.add_argument("--tag", type=str, default=os.environ.get("RFX_LIB", "librfx.so"))
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rows = []
    warm = False
    for name in a.shapes:
        N, Cin, H, W, Cout, k, stride, Cexp = SHAPES[name][:8]
        with_res = len(SHAPES[name]) > 8 and SHAPES[name][8]
        g = torch.Generator().manual_seed(7)
        w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5
        bn = dict(weight=torch.rand(Cout, generator=g) + 0.5, bias=torch.randn(Cout, generator=g) * 0.1,
                  running_mean=torch.randn(Cout, generator=g) * 0.1, running_var=torch.rand(Cout, generator=g) + 0.5)
        plan = ops.ConvPlan(w, bn, stride=stride, pad=k // 2, act=ops.ACT_RELU, device=dev)
        gd = torch.Generator(device=dev).manual_seed(11)
        xs = [torch.randn(N, Cin, H, W, device=dev, generator=gd) for _ in range(a.sets)]
        if Cexp:
            w3 = torch.randn(Cexp, Cout, 1, 1, generator=g) * (2.0 / Cout) ** 0.5
            bn3 = dict(weight=torch.rand(Cexp, generator=g) + 0.5, bias=torch.randn(Cexp, generator=g) * 0.1,
                       running_mean=torch.randn(Cexp, generator=g) * 0.1, running_var=torch.rand(Cexp, generator=g) + 0.5)
            plan3 = ops.ConvPlan(w3, bn3, act=ops.ACT_RELU, device=dev)
            res = torch.randn(N, Cexp, H, W, device=dev, generator=gd)
            run = lambda x: ops.bottleneck_tail(x, plan, plan3, res)     # noqa: E731
            Ho, Wo = H, W
            flops = 2.0 * N * H * W * (Cout * Cin * 9 + Cexp * Cout)
        elif k == 7 and Cin == 3:
            Ho, Wo = plan.out_hw(H, W)
            run = lambda x: ops.stem_conv7_maxpool(x, plan)             # noqa: E731
            flops = 2.0 * N * Ho * Wo * Cout * Cin * k * k
        else:
            Ho, Wo = plan.out_hw(H, W)
            r1 = torch.randn(N, Cout, Ho, Wo, device=dev, generator=gd) if with_res else None
            run = lambda x: plan(x, residual=r1)                        # noqa: E731
            flops = 2.0 * N * Ho * Wo * Cout * Cin * k * k
        out = run(xs[0])
        torch.cuda.synchronize()
        chk = float(out.double().sum()), float(out.double().abs().max())
        del out
        for _ in range(3 if warm else 25):       # clock ramp on the first shape
            run(xs[0])
        warm = True
        evs = []
        for i in range(a.iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(xs[i % a.sets])
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in evs)
        avg = sum(ts) / len(ts)
        row = dict(tag=a.tag, shape=name, dims=SHAPES[name], avg_us=round(avg, 1), min_us=round(ts[0], 1),
                   tflops=round(flops / avg / 1e6, 1), frac=round(flops / avg / 1e6 / 157.3, 3), checksum=chk)
        rows.append(row)
        print("%-22s %-20s avg %9.1f us  min %9.1f  %6.1f TF  %.3f  sum %.6e" % (os.path.basename(a.tag), name, avg, ts[0],
                                                                              row["tflops"], row["frac"], chk[0]), flush=True)
        del xs
        torch.cuda.empty_cache()
    if a.out:
        with open(a.out, "a") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
