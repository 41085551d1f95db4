"""Sky segmentation on the device: the forward pass behind ``CoarseAlign.skyFromSeg``
(evaluation/evalHpatch/coarseAlignFeatMatch.py:63-64,152-153) = ``SegNet.getSky`` of segNet/segEval.py:23-43.

Per target image, over the five test scales of segNet/segData.py:57-77 (short side 300 / 375 / 450 / 525 / 600, long side <= 500,
sides rounded UP to multiples of 8, PIL BILINEAR -- on the host, like image decoding: the reference's own pre-processing):
  encoder   segNet/segModel.py:59-127,156-215: conv3x3(3->64, /2) - conv3x3(64->64) - conv3x3(64->128) (BN + ReLU each) - MaxPool(3, 2, 1) -
            layer1 (3 Bottlenecks, 64) - layer2 (4, 128, /2 on the 3x3) - layer3 (6, 256) - layer4 (3, 512), the last two with their
            strides removed and their 3x3 convolutions DILATED (first block of a layer: dilation = padding = d/2, the others d; d = 2 / 4):
            a stride-8 map of 2048 channels;
  decoder   :218-265 PPMDeepsup: four pyramid branches AdaptiveAvgPool(1 / 2 / 3 / 6) - conv1x1(2048->512) - BN - ReLU - bilinear
            up to the conv5 size, concatenated with conv5 (4096 channels) - conv3x3(4096->512) - BN - ReLU - [Dropout2d: identity in
            eval] - conv1x1(512->150) + bias - bilinear up to the ORIGINAL image size - softmax over the classes;
  scores += softmax / 5 (segEval.py:34-35); after the five scales: arg-max over the classes, mask = (pred == segId) or 1 - that.
Every convolution runs on librfx's convolution family (the direct 3x3 kernels, the k-major 1x1 kernel, the stride-2 3x3 kernel, the
implicit GEMM for the 3-channel stem, the stride-2 projection and -- through rfx_conv2d_dilated_f32 -- the dilated 3x3 layers), BN
folded, residual + ReLU fused; pooling / softmax / arg-max are csrc/seg.hip.  No CPU or eager-PyTorch fallback."""
import numpy as np
import torch
import PIL.Image as Image

from . import ops
from .ops import ConvPlan, ACT_NONE, ACT_RELU
from .nets import _bn

IMG_SIZES = (300, 375, 450, 525, 600)     # segNet/segEval.py:19
IMG_MAX_SIZE = 500
PADDING_CONSTANT = 8
MEAN = (0.485, 0.456, 0.406)              # segNet/segData.py:29-31
STD = (0.229, 0.224, 0.225)


def input_sizes(ori_width, ori_height, img_sizes=IMG_SIZES, img_max_size=IMG_MAX_SIZE, pad=PADDING_CONSTANT):
    """segNet/segData.py:63-72: (width, height) of the five inputs."""
    out = []
    for short in img_sizes:
        scale = min(short / float(min(ori_height, ori_width)), img_max_size / float(max(ori_height, ori_width)))
        th, tw = int(ori_height * scale), int(ori_width * scale)
        out.append((((tw - 1) // pad + 1) * pad, ((th - 1) // pad + 1) * pad))
    return out


def img_transform(img):
    """segNet/segData.py:33-38: uint8 HWC -> float32 / 255 -> CHW -> Normalize (host arithmetic, as the reference's)."""
    a = np.float32(np.array(img)) / 255.
    t = torch.from_numpy(a.transpose((2, 0, 1)).copy())
    return (t - torch.tensor(MEAN).view(3, 1, 1)) / torch.tensor(STD).view(3, 1, 1)


class SegEncoder:
    """ResnetDilated(resnet50, dilate_scale=8) (segNet/segModel.py:156-215)."""

    def __init__(self, sd, device="cuda"):
        self.stem = [ConvPlan(sd["conv1.weight"], _bn(sd, "bn1"), 2, 1, ACT_RELU, device),
                     ConvPlan(sd["conv2.weight"], _bn(sd, "bn2"), 1, 1, ACT_RELU, device),
                     ConvPlan(sd["conv3.weight"], _bn(sd, "bn3"), 1, 1, ACT_RELU, device)]
        self.blocks = []
        # (layer, blocks, stride of the first block's 3x3 / projection, dilation of the first block's 3x3, dilation of the others)
        for layer, nblk, stride, d_first, d_rest in (("layer1", 3, 1, 1, 1), ("layer2", 4, 2, 1, 1), ("layer3", 6, 1, 1, 2),
                                                     ("layer4", 3, 1, 2, 4)):
            for b in range(nblk):
                p = "%s.%d" % (layer, b)
                s = stride if b == 0 else 1
                d = d_first if b == 0 else d_rest                    # _nostride_dilate, :184-198
                blk = {"c1": ConvPlan(sd[p + ".conv1.weight"], _bn(sd, p + ".bn1"), 1, 0, ACT_RELU, device),
                       "c2": ConvPlan(sd[p + ".conv2.weight"], _bn(sd, p + ".bn2"), s, d, ACT_RELU, device, dilation=d),
                       "c3": ConvPlan(sd[p + ".conv3.weight"], _bn(sd, p + ".bn3"), 1, 0, ACT_RELU, device), "ds": None}
                if (p + ".downsample.0.weight") in sd:
                    blk["ds"] = ConvPlan(sd[p + ".downsample.0.weight"], _bn(sd, p + ".downsample.1"), s, 0, ACT_NONE, device)
                self.blocks.append(blk)

    def __call__(self, x):
        for c in self.stem:
            x = c(x)
        x = ops.maxpool2d(x, 3, 2, 1)
        for blk in self.blocks:
            o = blk["c2"](blk["c1"](x))
            r = blk["ds"](x) if blk["ds"] is not None else x
            x = blk["c3"](o, residual=r)                                # relu(bn3(conv3(o)) + r)
        return x                                                         # conv5: (N, 2048, H/8, W/8)


class SegDecoder:
    """PPMDeepsup inference path (segNet/segModel.py:250-265): class logits at the conv5 resolution."""

    POOL_SCALES = (1, 2, 3, 6)

    def __init__(self, sd, device="cuda"):
        self.ppm = [ConvPlan(sd["ppm.%d.1.weight" % k], _bn(sd, "ppm.%d.2" % k), 1, 0, ACT_RELU, device) for k in range(4)]
        self.conv_last = ConvPlan(sd["conv_last.0.weight"], _bn(sd, "conv_last.1"), 1, 1, ACT_RELU, device)
        self.classify = ConvPlan(sd["conv_last.4.weight"], None, 1, 0, ACT_NONE, device, bias=sd["conv_last.4.bias"])

    def __call__(self, conv5):
        h, w = conv5.shape[2], conv5.shape[3]
        parts = [conv5]
        for scale, conv in zip(self.POOL_SCALES, self.ppm):
            parts.append(ops.resize_bilinear(conv(ops.adaptive_avgpool2d(conv5, scale)), (h, w), align_corners=False))
        return self.classify(self.conv_last(torch.cat(parts, dim=1)))


class SegNetDevice:
    """``SegNet`` of segNet/segEval.py:6-43 on the MI355X: ``get_sky(image)`` -> float32 (H, W) numpy mask."""

    def __init__(self, encoder_sd, decoder_sd, segId=1, segFg=True, device="cuda", num_class=150):
        self.dev = torch.device(device)
        self.encoder = SegEncoder(encoder_sd, self.dev)
        self.decoder = SegDecoder(decoder_sd, self.dev)
        self.segId, self.segFg, self.num_class = int(segId), bool(segFg), num_class

    def scores(self, img):
        """The averaged class probabilities (1, 150, H, W) of a PIL image (segEval.py:27-35), on the device."""
        img = img.convert("RGB")
        W, H = img.size
        scores = None
        for (tw, th) in input_sizes(W, H):
            x = img_transform(img.resize((tw, th), Image.BILINEAR)).unsqueeze(0).contiguous().to(self.dev)
            logits = self.decoder(self.encoder(x))
            up = ops.resize_bilinear(logits, (H, W), align_corners=False)
            # segEval.py:30 starts from zeros: 0 + p/5 == p/5 exactly, so the first scale stores, the others accumulate
            scores = ops.softmax_accum(up, scores, div=float(len(IMG_SIZES)))
        return scores

    def get_sky_device(self, img, want_pred=False):
        """-> (mask (H, W) float32 device tensor[, pred (H, W) int32]): segEval.py:37-43."""
        out = ops.argmax_mask(self.scores(img), self.segId, complement=self.segFg, want_pred=want_pred)
        return (out[0][0], out[1][0]) if want_pred else out[0]

    def getSky(self, img_or_path):
        img = Image.open(img_or_path) if isinstance(img_or_path, (str, bytes)) or hasattr(img_or_path, "__fspath__") else img_or_path
        return self.get_sky_device(img).cpu().numpy()
