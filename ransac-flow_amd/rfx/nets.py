"""The four networks of the hot path as sequences of librfx kernel launches.

Each class is built from a reference-keyed ``state_dict`` (see rfx/weights.py), packs the conv weights and
folds eval-mode BatchNorm once, and then runs forward-only on device tensors.  Activations stay NCHW fp32
from the first layer to the last: there is no layout conversion between kernels.
"""
import os

import torch

from . import ops
from .ops import ConvPlan, ACT_NONE, ACT_RELU, ACT_SIGMOID


def _bn(sd, p):
    return {k: sd[p + "." + k] for k in ("weight", "bias", "running_mean", "running_var")}


class ResNet50Trunk:
    """ResNet-50 conv1..layer3 (stride 16, 1024 channels): model/resnet50.py:68-104,112-169 as sliced by
    quick_start/coarseAlignFeatMatch.py:34-52.  Bottleneck = 1x1 -> 3x3 (stride here) -> 1x1, BN folded into
    every conv's epilogue, residual add + ReLU fused into the third conv."""

    def __init__(self, sd, device="cuda"):
        self.conv1 = ConvPlan(sd["conv1.weight"], _bn(sd, "bn1"), stride=2, pad=3, act=ACT_RELU, device=device)
        self.blocks = []
        for layer, nblk, stride in (("layer1", 3, 1), ("layer2", 4, 2), ("layer3", 6, 2)):
            for b in range(nblk):
                p = "%s.%d" % (layer, b)
                s = stride if b == 0 else 1
                # split=True (round 6): the long-K 1x1 layers -- conv1 of every block but the first (K = 256 ... 1024) and layer3's
                # conv3 (K = 256, + residual) -- run on rfx_conv1x1_split_f32: float32 sums from exact bf16 operand pieces, closer to
                # the exact sum than the fp32-MFMA kernel and 1.3-1.5x faster (profiles/r06_split_conv_bench.json).  The conv3 of a
                # block whose tail can fuse (layer1 / layer2) stays on the fp32 kernels: fused and two-kernel forms stay bit-identical.
                w1 = sd[p + ".conv1.weight"]
                blk = {
                    "c1": ConvPlan(w1, _bn(sd, p + ".bn1"), 1, 0, ACT_RELU, device, split=w1.shape[1] >= 128),
                    "c2": ConvPlan(sd[p + ".conv2.weight"], _bn(sd, p + ".bn2"), s, 1, ACT_RELU, device, split=s == 2),   # stride 2: rfx_conv3x3_split_s2_f32
                    "c3": ConvPlan(sd[p + ".conv3.weight"], _bn(sd, p + ".bn3"), 1, 0, ACT_RELU, device),
                    "ds": None,
                }
                if not ops.bottleneck_tail_shape(blk["c2"], blk["c3"]):
                    blk["c3"] = ConvPlan(sd[p + ".conv3.weight"], _bn(sd, p + ".bn3"), 1, 0, ACT_RELU, device, split=True)
                    if s == 1:   # layer3's 256 -> 256 3x3 (K = 2304): rfx_conv3x3_split_f32
                        blk["c2"] = ConvPlan(sd[p + ".conv2.weight"], _bn(sd, p + ".bn2"), s, 1, ACT_RELU, device, split=True)
                if (p + ".downsample.0.weight") in sd:
                    wd = sd[p + ".downsample.0.weight"]       # the stride-2 projections (256 -> 512, 512 -> 1024): split kernel, strided pixels
                    blk["ds"] = ConvPlan(wd, _bn(sd, p + ".downsample.1"), s, 0, ACT_NONE, device, split=wd.shape[1] >= 128)
                if ops.bottleneck_tail_shape(blk["c2"], blk["c3"]):
                    if ops.conv_split_enabled() and os.environ.get("RFX_SPLIT_TAILS", "1") != "0":
                        # layer1 / layer2 tails (3x3 64 -> 64 / 128 -> 128, then the 1x1 expansion): two split kernels beat the fused fp32
                        # kernel (the 3x3 runs 1.3-1.45x faster on the bf16 pipe; measured per block in profiles/r06_split_conv_bench.json)
                        blk["c2"] = ConvPlan(sd[p + ".conv2.weight"], _bn(sd, p + ".bn2"), s, 1, ACT_RELU, device, split=True)
                        blk["c3"] = ConvPlan(sd[p + ".conv3.weight"], _bn(sd, p + ".bn3"), 1, 0, ACT_RELU, device,
                                             split=blk["c3"].Cin >= 128)
                    # the tail's 3x3 (K = 576 / 1152) sums in chunks of 4 K steps -- in the fused kernel and, with this argument,
                    # in the stand-alone one (RFX_FUSE_BOTTLENECK=0): the two forms stay bit-identical
                    blk["c2"].k_chunk = 4
                self.blocks.append(blk)

    def __call__(self, x):
        x = ops.stem_conv7_maxpool(x, self.conv1)  # conv1 + BN + ReLU + MaxPool2d(3, 2, 1) in one kernel
        for blk in self.blocks:
            o = blk["c1"](x)
            r = blk["ds"](x) if blk["ds"] is not None else x
            if ops.bottleneck_tail_eligible(blk["c2"], blk["c3"]):
                x = ops.bottleneck_tail(o, blk["c2"], blk["c3"], residual=r)   # conv2 + conv3 in one kernel
            else:
                o = blk["c2"](o)
                x = blk["c3"](o, residual=r)  # relu(bn3(conv3(o)) + r)
        return x


    def forward_group(self, xs, side_streams=True):
        """The same pass over several inputs of DIFFERENT sizes (a single pair's 7 pyramid levels + target), layer by layer
        with one grouped launch per kernel instance and layer (ops.launch_group) instead of one launch per layer and input:
        bit-identical to [self(x) for x in xs]."""
        dev = xs[0].device
        with ops.launch_group(dev, side_streams):
            xs = [ops.stem_conv7_maxpool(x, self.conv1) for x in xs]
        for blk in self.blocks:
            with ops.launch_group(dev, side_streams):          # c1 and the projection shortcut both read x only: one group
                os_ = [blk["c1"](x) for x in xs]
                rs = [blk["ds"](x) for x in xs] if blk["ds"] is not None else xs
            # outputs get NEW names inside a group and are rebound after it: the recorded launches read their inputs when the
            # group ends, so every input tensor must stay referenced until then (launch_group's contract)
            if ops.bottleneck_tail_eligible(blk["c2"], blk["c3"]):
                with ops.launch_group(dev, side_streams):
                    ys = [ops.bottleneck_tail(o, blk["c2"], blk["c3"], residual=r) for o, r in zip(os_, rs)]
            else:
                with ops.launch_group(dev, side_streams):
                    ms = [blk["c2"](o) for o in os_]
                with ops.launch_group(dev, side_streams):
                    ys = [blk["c3"](m, residual=r) for m, r in zip(ms, rs)]
            xs = ys
        return xs


class FeatureExtractorNet:
    """model/model.py:59-125: conv3x3(3->64)+BN+ReLU, MaxPool(2,1)+BlurPool/2, 2xBasicBlock(64),
    2x(128,/2), 2x(256,/2).  Shortcut of the strided blocks = BlurPool/2 -> 1x1 conv -> BN (:92-93)."""

    def __init__(self, sd, device="cuda"):
        self.conv1 = ConvPlan(sd["conv1.weight"], _bn(sd, "bn1"), 1, 1, ACT_RELU, device)
        self.blocks = []
        for layer, stride in (("layer1", 1), ("layer2", 2), ("layer3", 2)):
            for b in range(2):
                p = "%s.%d" % (layer, b)
                s = stride if b == 0 else 1
                blk = {
                    # the stride-1 3x3 convolutions (64 / 128 / 256 channels): rfx_conv3x3_split_f32 (round 6; csrc/conv3x3s.hip)
                    "c1": ConvPlan(sd[p + ".conv1.weight"], _bn(sd, p + ".bn1"), s, 1, ACT_RELU, device, split=True),      # stride 2: ..._split_s2_f32
                    "c2": ConvPlan(sd[p + ".conv2.weight"], _bn(sd, p + ".bn2"), 1, 1, ACT_RELU, device, split=True),
                    "ds": None, "stride": s,
                }
                if (p + ".downsample.1.weight") in sd:
                    blk["ds"] = ConvPlan(sd[p + ".downsample.1.weight"], _bn(sd, p + ".downsample.2"), 1, 0, ACT_NONE,
                                         device)
                self.blocks.append(blk)

    def __call__(self, x):
        x = ops.stem_conv_maxblur(x, self.conv1)  # conv1 + BN + ReLU + MaxPool2d(2, stride 1) + BlurPool/2 in one kernel
        for blk in self.blocks:
            o = blk["c1"](x)
            r = blk["ds"](ops.blurpool2d(x, blk["stride"])) if blk["ds"] is not None else x
            x = blk["c2"](o, residual=r)
        return x


class _HeadTrunk:
    def __init__(self, sd, last_act, device):
        self.c1 = ConvPlan(sd["conv1.weight"], _bn(sd, "bn1"), 1, 1, ACT_RELU, device, split=True)      # 49 -> 512 (a ragged 4th channel block),
        self.c2 = ConvPlan(sd["conv2.weight"], _bn(sd, "bn2"), 1, 1, ACT_RELU, device, split=True)      # 512 -> 256, 256 -> 128:
        self.c3 = ConvPlan(sd["conv3.weight"], _bn(sd, "bn3"), 1, 1, ACT_RELU, device, split=True)      # rfx_conv3x3_split_f32
        self.c4 = ConvPlan(sd["conv4.weight"], None, 1, 1, last_act, device)

    def __call__(self, coef):
        return self.c4(self.c3(self.c2(self.c1(coef))))


class NetFlowCoarseNet:
    """model/model.py:167-249.  up8X=True -> F.upsample_bilinear x8 (align_corners=True, :234)."""

    def __init__(self, sd, kernelSize=7, device="cuda"):
        self.k = kernelSize
        self.trunk = _HeadTrunk(sd, ACT_NONE, device)

    def __call__(self, coef, up8X=True):
        flow = ops.flow_head(self.trunk(coef), self.k)
        if up8X:
            flow = ops.resize_bilinear(flow, (flow.shape[2] * 8, flow.shape[3] * 8), align_corners=True)
        return flow


class NetMatchabilityNet:
    """model/model.py:254-322: sigmoid fused into the last conv's epilogue."""

    def __init__(self, sd, kernelSize=7, device="cuda"):
        self.trunk = _HeadTrunk(sd, ACT_SIGMOID, device)

    def __call__(self, feat, up8X=True):
        m = self.trunk(feat)
        if up8X:
            m = ops.resize_bilinear(m, (m.shape[2] * 8, m.shape[3] * 8), align_corners=True)
        return m
